"""Writes rocco_amd/csrc/ziggurat_tables.h: the three 256-entry tables of NumPy's ziggurat sampler for the standard
normal (numpy/random/src/distributions/ziggurat_constants.h: ki_double, wi_double, fi_double; NumPy is BSD-3-Clause).

The tables are NumPy's DATA, needed bit for bit: rocco_amd/csrc/normal.hip reproduces `Generator.standard_normal` on the
device, and a table entry that differs in its last bit gives a different stream.  They are not reproducible from the
published construction (Marsaglia & Tsang 2000 with r = 3.65415288536100879635, 256 layers): the exact construction
evaluated with 80 digits agrees with NumPy's entries to a few units in the last place only, NumPy's were made in double
arithmetic.  So this script reads them out of the static library NumPy ships for extension writers
(numpy/random/lib/libnpyrandom.a, member distributions.c.o, section .rodata), finds the three arrays by comparing with
the exact construction, and checks a pure-Python restatement of the sampler built on them against
`np.random.default_rng(seed).standard_normal` before writing the header.

    python scripts/gen_ziggurat_tables.py            (needs binutils' ar / objcopy and mpmath)
"""
import math
import os
import subprocess
import sys
import tempfile

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_STR = "3.6541528853610087963519472518"
PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645


def exact_tables():
    mp.mp.dps = 80
    x1 = mp.mpf(R_STR)
    fb = lambda x: mp.exp(-x * x / 2)
    area = x1 * fb(x1) + mp.sqrt(mp.pi / 2) * mp.erfc(x1 / mp.sqrt(2))
    two52 = mp.mpf(2) ** 52
    K, W, F = [0] * 256, [0.0] * 256, [0.0] * 256
    K[0] = int(mp.floor(x1 * fb(x1) / area * two52))
    W[0], W[255] = float(area / fb(x1) / two52), float(x1 / two52)
    F[0], F[255] = 1.0, float(fb(x1))
    for i in range(254, 0, -1):
        x = mp.sqrt(-2 * mp.log(area / x1 + fb(x1)))
        K[i + 1], W[i], F[i] = int(mp.floor(x / x1 * two52)), float(x / two52), float(fb(x))
        x1 = x
    return np.array(K, dtype=np.float64), np.array(W), np.array(F)


def numpy_rodata():
    lib = os.path.join(os.path.dirname(np.__file__), "random", "lib", "libnpyrandom.a")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["ar", "x", lib], cwd=tmp, check=True)
        member = [f for f in os.listdir(tmp) if "distributions" in f and "distributions.c" in f.replace("_", ".")]
        member = member or [f for f in os.listdir(tmp) if f.endswith("distributions.c.o")]
        out = os.path.join(tmp, "rodata.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.rodata", os.path.join(tmp, member[0]), out], check=True)
        return open(out, "rb").read()


def find(blob, approx, dtype):
    """The 2048-byte window of `blob` that, read as `dtype`, lies within 1e-9 (relative) of `approx`."""
    for off in range(0, len(blob) - 2047, 8):
        got = np.frombuffer(blob[off:off + 2048], dtype=dtype).astype(np.float64)
        if np.all(np.abs(got - approx) <= 1e-9 * np.maximum(np.abs(approx), 1e-300)):
            return np.frombuffer(blob[off:off + 2048], dtype=dtype).copy()
    raise SystemExit("table not found in NumPy's library")


def restated_normals(seed, count, ki, wi, fi):
    """NumPy's `random_standard_normal` (distributions.c) over PCG64 (pcg64.h: step, then XSL-RR of the new state)."""
    st = np.random.default_rng(seed).bit_generator.state["state"]
    state, inc, mask = st["state"], st["inc"], (1 << 128) - 1
    inv_r, r = 0.27366123732975827203338247596, float(mp.mpf(R_STR))
    out, raws = [], 0

    def nxt():
        nonlocal state, raws
        state = (state * PCG_MULT + inc) & mask
        raws += 1
        x, rot = (state >> 64) ^ (state & ((1 << 64) - 1)), state >> 122
        return ((x >> rot) | (x << ((-rot) & 63))) & ((1 << 64) - 1)

    def dbl():
        return (nxt() >> 11) * (1.0 / 9007199254740992.0)

    while len(out) < count:
        u = nxt()
        idx, u = u & 0xFF, u >> 8
        sign, rabs = u & 1, (u >> 1) & 0x000FFFFFFFFFFFFF
        x = rabs * wi[idx]
        x = -x if sign else x
        if rabs < int(ki[idx]):
            out.append(x)
        elif idx == 0:
            while True:
                xx, yy = -inv_r * math.log1p(-dbl()), -math.log1p(-dbl())
                if yy + yy > xx * xx:
                    out.append(-(r + xx) if (rabs >> 8) & 1 else r + xx)
                    break
        elif (fi[idx - 1] - fi[idx]) * dbl() + fi[idx] < math.exp(-0.5 * x * x):
            out.append(x)
    return np.array(out), raws


def main():
    K, W, F = exact_tables()
    blob = numpy_rodata()
    ki, wi, fi = find(blob, K, np.uint64), find(blob, W, np.float64), find(blob, F, np.float64)
    for seed in (0, 20240, 104729 * 5 + 3):
        mine, raws = restated_normals(seed, 200000, ki, wi, fi)
        if not np.array_equal(mine, np.random.default_rng(seed).standard_normal(200000)):
            raise SystemExit("restated sampler differs from NumPy's")
        a, b = np.random.default_rng(seed), np.random.default_rng(seed)
        a.standard_normal(200000)
        b.bit_generator.advance(raws)
        if not np.array_equal(a.standard_normal(8), b.standard_normal(8)):
            raise SystemExit("raw draws consumed differ from NumPy's")
    path = os.path.join(ROOT, "rocco_amd", "csrc", "ziggurat_tables.h")
    with open(path, "w") as f:
        f.write("// rocco_amd/csrc/ziggurat_tables.h -- written by scripts/gen_ziggurat_tables.py; do not edit.\n"
                "// The tables of NumPy's ziggurat sampler for the standard normal (numpy/random/src/distributions/\n"
                f"// ziggurat_constants.h, BSD-3-Clause), read from NumPy {np.__version__}'s libnpyrandom.a and checked there: a\n"
                "// restatement of Generator.standard_normal on these tables reproduces NumPy's stream bit for bit.\n"
                "#pragma once\n\n#include <cstdint>\n\nnamespace rocco {\n\n")
        f.write("static const uint64_t kZigguratKi[256] = {\n")
        for i in range(0, 256, 4):
            f.write("    " + ", ".join(f"0x{int(v):016X}ull" for v in ki[i:i + 4]) + ",\n")
        f.write("};\n\n// (bit patterns of the doubles)\nstatic const uint64_t kZigguratWiBits[256] = {\n")
        for i in range(0, 256, 4):
            f.write("    " + ", ".join(f"0x{int(v):016X}ull" for v in wi[i:i + 4].view(np.uint64)) + ",\n")
        f.write("};\n\nstatic const uint64_t kZigguratFiBits[256] = {\n")
        for i in range(0, 256, 4):
            f.write("    " + ", ".join(f"0x{int(v):016X}ull" for v in fi[i:i + 4].view(np.uint64)) + ",\n")
        f.write("};\n\n}  // namespace rocco\n")
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
