"""Synthetic signal matrices for benchmarks and parity tests.

Two generators:

* `hash_matrix` / `hash_matrix_device` -- the counter-based generator of
  rocco_amd/csrc/synth.hip restated with NumPy integer arithmetic, so a device-resident matrix of
  any size (the K=100 whole-genome benchmark input is 49 GB and is never materialised on the host)
  can be regenerated slice by slice on the CPU bit-for-bit.
* `survey_matrix` -- the NumPy-RNG recipe of SURVEY.md section 8(d) (gamma background rounded to
  5 decimals + planted peaks with per-sample dropout), used for the CPU-sized golden fixtures.

Also the hg38 chromosome sizes the benchmark configurations are defined on (lengths only: data,
identical to rocco/hg38.sizes:1-24).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Tuple

import numpy as np

HG38_SIZES: Dict[str, int] = {
    "chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555,
    "chr5": 181538259, "chr6": 170805979, "chr7": 159345973, "chr8": 145138636,
    "chr9": 138394717, "chr10": 133797422, "chr11": 135086622, "chr12": 133275309,
    "chr13": 114364328, "chr14": 107043718, "chr15": 101991189, "chr16": 90338345,
    "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
    "chr21": 46709983, "chr22": 50818468, "chrX": 156040895, "chrY": 57227415,
}


def chrom_loci(step: int = 50, chroms=None) -> List[Tuple[str, int]]:
    """[(chrom, n_loci)] with n = ceil(size / step)."""
    names = list(HG38_SIZES) if chroms is None else list(chroms)
    return [(c, -(-HG38_SIZES[c] // int(step))) for c in names]


def chrom_seed(base_seed: int, chrom_index: int) -> int:
    return (int(base_seed) * 1000003 + int(chrom_index)) & 0xFFFFFFFFFFFFFFFF


# --------------------------------------------------------------------------------------------
# counter-based generator (bit-identical to synth.hip)
# --------------------------------------------------------------------------------------------

_U64 = np.uint64


def _mix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + _U64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def hash_matrix(K: int, n: int, seed: int, j0: int = 0, dtype=np.float64) -> np.ndarray:
    """Rows 0..K-1, loci j0..j0+n-1 of the synthetic matrix for `seed` (host, NumPy)."""
    seed_u = _U64(int(seed) & 0xFFFFFFFFFFFFFFFF)
    j = np.arange(j0, j0 + n, dtype=np.uint64)
    out = np.empty((K, n), dtype=np.float64)
    period = j // _U64(1500)
    with np.errstate(over="ignore"):
        hp = _mix64(seed_u ^ (period * _U64(0xA24BAED4963EE407)) ^ _U64(0x5851F42D4C957F2D))
        start = period * _U64(1500) + _U64(300) + (hp % _U64(600))
        width = _U64(4) + ((hp >> _U64(16)) % _U64(37))
    in_peak = (j >= start) & (j < start + width)
    for k in range(K):
        with np.errstate(over="ignore"):
            row_key = _mix64(np.array([seed_u ^ (_U64(k + 1) * _U64(0xD6E8FEB86659FD93))], dtype=np.uint64))[0]
            h = _mix64(row_key + j)
        a = ((h >> _U64(40)).astype(np.uint32) | np.uint32(1)).astype(np.float64)
        mant, e = np.frexp(a)  # a = mant * 2**e, mant in [0.5, 1)
        t = (24 - e).astype(np.float64) + 2.0 * (1.0 - mant)
        bg = t * 0.20794415416798357
        units = np.rint(bg * 100000.0)
        with np.errstate(over="ignore"):
            hs = _mix64(hp ^ (_U64(k + 1) * _U64(0x9FB21C651E98DF25)))
        present = in_peak & ((hs % _U64(10)) < _U64(8))
        u = ((hs >> _U64(20)) & _U64(0xFFFFF)).astype(np.float64) * (1.0 / 1048576.0)
        amp = 2.0 + 8.0 * u
        units = units + np.where(present, np.rint(amp * 100000.0), 0.0)
        out[k] = units / 100000.0
    return out.astype(dtype, copy=False)


def hash_matrix_device(K: int, n: int, seed: int, device=None, dtype=None, out=None):
    """Device-resident synthetic matrix [K, n] via rocco_hip_synth_matrix."""
    import torch

    from . import _native
    from . import dp as _dp

    dtype = dtype or torch.float64
    if out is None:
        out = torch.empty((K, n), dtype=dtype, device=f"cuda:{_dp._device_index(device)}")
    solver = _native.solver_for(out.device.index)
    _native.check(_native.load().rocco_hip_synth_matrix(
        solver.handle, out.data_ptr(), 0 if out.dtype == torch.float64 else 1, K, n, int(out.stride(0)),
        ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _dp._stream_ptr(out)), "rocco_hip_synth_matrix")
    return out


# --------------------------------------------------------------------------------------------
# SURVEY.md section 8(d) recipe (NumPy RNG; CPU-sized problems and golden fixtures)
# --------------------------------------------------------------------------------------------

def survey_matrix(n: int, K: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    m = np.round(rng.gamma(1.0, 0.3, size=(K, n)), 5)
    pos = 0
    while True:
        pos += 1500 + int(rng.integers(-300, 301))
        if pos >= n:
            break
        w = int(rng.integers(4, 41))
        amp = rng.gamma(6.0, 1.0, size=K) * (rng.random(K) < 0.8)
        m[:, pos:pos + w] += amp[:, None]
    return np.round(m, 5)
