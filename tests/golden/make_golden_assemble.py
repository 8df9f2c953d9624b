#!/usr/bin/env python3
"""What the REFERENCE's matrix assembly makes of decoded tracks.

    make -C oracle ref
    python tests/golden/make_golden_assemble.py

Two of the reference's functions are run here, on inputs that are DATA:

* ``generate_chrom_matrix`` (rocco/readtracks.py:521-633) with its two per-file readers replaced IN THE MODULE by
  callables that hand back prepared ``(starts, values)`` lists -- the module guards its optional imports
  (rocco/readtracks.py:17-25), so it imports without pyBigWig / htslib; with ``num_processors=1`` the readers are called
  in process.  Stored per scenario: the file names, every track's lists (or "no data"), the keyword overrides, and what
  the reference returned -- ``(intervals, matrix)``, ``(None, None)`` or the ``ValueError`` text.
* ``get_bigwig_chrom_scores`` (rocco/readtracks.py:94-186) over a stand-in ``pyBigWig`` object whose ``open(...)``
  returns prepared ``chroms()`` / ``intervals(chromosome)``: the five validations, the dense fill of gaps, the constant
  scale, ``np.round(..., digits)``.

Writes tests/golden/assemble_vectors.npz + assemble_vectors.json (data only, no reference source)."""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("REFERENCE", "/root/reference")

pkg = types.ModuleType("rocco")
pkg.__path__ = [os.path.join(REFERENCE, "rocco"), os.path.join(ROOT, "oracle", "_ref")]
sys.modules["rocco"] = pkg
rt = importlib.import_module("rocco.readtracks")
REFERENCE_BIGWIG_READER = rt.get_bigwig_chrom_scores  # (replaced in the module for the generate_chrom_matrix scenarios)

rng = np.random.default_rng(20250)
arrays, meta = {}, {"matrix": [], "bigwig": []}


def values(n, digits=5):
    return np.round(rng.gamma(1.0, 2.0, size=n), digits)


def ragged(n, step, first=1000, keep=0.9):
    a = first + step * np.arange(n, dtype=np.int64)
    return a[rng.random(n) < keep] if keep < 1.0 else a


# ---- generate_chrom_matrix -------------------------------------------------------------------------------------------
def matrix_scenario(name, files, tracks, **kwargs):
    """tracks[k]: (starts, values) or None (the reader found nothing for that file)."""
    table = dict(zip(files, tracks))

    def fake_bam(bam_file, *_a, **_k):
        entry = table[bam_file]
        return (None, None) if entry is None else (np.asarray(entry[0]), np.asarray(entry[1]))

    def fake_bigwig(bigwig_file, *_a, **_k):
        entry = table[bigwig_file]
        return (None, None) if entry is None else (np.asarray(entry[0]), np.asarray(entry[1]))

    rt.get_bam_chrom_reads, rt.get_bigwig_chrom_scores = fake_bam, fake_bigwig
    record = {"name": name, "files": list(files), "kwargs": kwargs, "has": [t is not None for t in tracks]}
    for k, t in enumerate(tracks):
        if t is not None:
            arrays[f"m_{name}_t{k}_starts"] = np.asarray(t[0], dtype=np.int64)
            arrays[f"m_{name}_t{k}_values"] = np.asarray(t[1], dtype=np.float64)
    try:
        intervals, matrix = rt.generate_chrom_matrix("chrT", list(files), "unused.sizes", 50, num_processors=1, **kwargs)
    except ValueError as exc:
        record["error"] = str(exc)
    else:
        if intervals is None:
            record["none"] = True
        else:
            arrays[f"m_{name}_intervals"] = np.asarray(intervals)
            arrays[f"m_{name}_matrix"] = np.asarray(matrix)
            record["intervals_dtype"] = str(np.asarray(intervals).dtype)
            record["matrix_dtype"] = str(np.asarray(matrix).dtype)
    meta["matrix"].append(record)


full = 1000 + 50 * np.arange(400, dtype=np.int64)
matrix_scenario("bam_aligned", ["a.bam", "b.bam", "c.bam"], [(full, values(400)), (full, values(400)), (full, values(400))])
r1, r2, r3 = ragged(300, 50), ragged(310, 50, first=900), ragged(280, 50, first=1250)
matrix_scenario("bam_ragged", ["a.bam", "b.bam", "c.bam"], [(r1, values(r1.size)), (r2, values(r2.size)), (r3, values(r3.size))])
u = rng.permutation(np.concatenate([r1, r1[rng.integers(0, r1.size, size=40)]]))  # unsorted, some loci twice: the last write stays
matrix_scenario("bam_repeated_unsorted", ["a.bam", "b.bam"], [(u, values(u.size)), (r2, values(r2.size))])
matrix_scenario("bam_one_track_empty", ["a.bam", "b.bam", "c.bam"], [(r1, values(r1.size)), None, (r3, values(r3.size))])
matrix_scenario("bam_all_empty", ["a.bam", "b.bam"], [None, None])
matrix_scenario("bam_one_track", ["a.bam"], [(r2, values(r2.size))])  # K = 1: the reshape
matrix_scenario("bam_uneven_steps", ["a.bam", "b.bam"], [(np.array([0, 50, 100, 175, 400]), values(5)), (np.array([50, 175, 900]), values(3))])
matrix_scenario("bam_low_memory", ["a.bam", "b.bam", "c.bam"], [(r1, values(r1.size, 7) * 1.0e3), (r2, values(r2.size, 7)), (r3, values(r3.size, 7) * 1.0e-3)],
                low_memory=True)
matrix_scenario("bam_single_locus", ["a.bam", "b.bam"], [(np.array([700]), np.array([2.5])), (np.array([700]), np.array([0.125]))])
big = [ragged(5000, 10, first=10 * int(rng.integers(0, 40)), keep=0.8) for _ in range(6)]
matrix_scenario("bam_six_ragged_tracks", [f"s{k}.bam" for k in range(6)], [(b, values(b.size)) for b in big])
even, odd = 0 + 50 * np.arange(120, dtype=np.int64), 25 + 50 * np.arange(120, dtype=np.int64)
matrix_scenario("bigwig_two_phases", ["a.bw", "b.bigwig"], [(even, values(120)), (odd, values(120))])  # the union still has ONE step
matrix_scenario("bigwig_aligned_ragged_ends", ["a.bw", "b.bw", "c.bw"], [(full[20:], values(380)), (full[:350], values(350)), (full, values(400))])
matrix_scenario("bigwig_broken_step", ["a.bw", "b.bw"], [(even, values(120)), (np.array([0, 50, 100, 175]), values(4))])
matrix_scenario("bigwig_one_track_with_gap", ["a.bw"], [(np.array([0, 50, 100, 200]), values(4))])  # one track, uneven union
matrix_scenario("bigwig_single_locus", ["a.bw"], [(np.array([350]), np.array([1.5]))])
matrix_scenario("bigwig_low_memory", ["a.bw", "b.bw"], [(full, values(400, 6) * 123.456), (full[5:395], values(390, 6))], low_memory=True)
matrix_scenario("mixed_types", ["a.bam", "b.bw"], [(full, values(400)), (full, values(400))])
matrix_scenario("bigwig_one_track_empty", ["a.bw", "b.bw"], [None, (full[:50], values(50))])


# ---- get_bigwig_chrom_scores -----------------------------------------------------------------------------------------
class FakeBigWig:
    def __init__(self, chroms, intervals):
        self._chroms, self._intervals, self.closed = chroms, intervals, False

    def chroms(self):
        return self._chroms

    def intervals(self, chromosome):
        return self._intervals.get(chromosome)

    def close(self):
        self.closed = True


tmp = tempfile.mkdtemp()
sizes_file = os.path.join(tmp, "t.sizes")
with open(sizes_file, "w", encoding="utf-8") as handle:
    handle.write("chrT\t1000000\nchrU\t5000\n")
bw_file = os.path.join(tmp, "t.bw")
open(bw_file, "w").close()


def bigwig_scenario(name, intervals, chromosome="chrT", chroms=None, **kwargs):
    """intervals: list of (start, end, value) as pyBigWig returns them, or None."""
    chroms = {"chrT": 1000000, "chrU": 5000} if chroms is None else chroms
    handle = FakeBigWig(chroms, {chromosome: intervals} if intervals is not None else {})
    fake = types.ModuleType("pyBigWig")
    fake.open = lambda _path: handle
    rt.pyBigWig = fake
    record = {"name": name, "chromosome": chromosome, "chroms": list(chroms), "kwargs": kwargs, "has_intervals": intervals is not None}
    if intervals is not None:
        arrays[f"b_{name}_starts"] = np.asarray([e[0] for e in intervals], dtype=np.int64)
        arrays[f"b_{name}_ends"] = np.asarray([e[1] for e in intervals], dtype=np.int64)
        arrays[f"b_{name}_values"] = np.asarray([e[2] for e in intervals], dtype=np.float64)
    try:
        out_i, out_v = REFERENCE_BIGWIG_READER(bw_file, chromosome, sizes_file, **kwargs)
    except ValueError as exc:
        record["error"] = str(exc).replace(bw_file, "{file}").replace(sizes_file, "{sizes}")
    else:
        assert handle.closed
        if out_i is None:
            record["none"] = True
        else:
            arrays[f"b_{name}_intervals"] = np.asarray(out_i)
            arrays[f"b_{name}_out"] = np.asarray(out_v)
            record["intervals_dtype"] = str(np.asarray(out_i).dtype)
    meta["bigwig"].append(record)


def track(starts, step, vals):
    return [(int(s), int(s) + step, float(v)) for s, v in zip(starts, vals)]


dense = 2000 + 25 * np.arange(300, dtype=np.int64)
bigwig_scenario("dense", track(dense, 25, values(300, 7)))
holes = dense[rng.random(300) < 0.7]
holes = np.concatenate([dense[:1], holes[(holes > dense[0]) & (holes < dense[-1])], dense[-1:]])
bigwig_scenario("gaps_zero_filled", track(holes, 25, values(holes.size, 7)))
bigwig_scenario("round_two_digits", track(holes, 25, values(holes.size, 7)), round_digits=2)
bigwig_scenario("round_zero_digits", track(holes, 25, values(holes.size, 7) * 10.0), round_digits=0)
bigwig_scenario("scaled_half", track(holes, 25, values(holes.size, 7)), const_scale=0.5)
bigwig_scenario("scaled_by_a_third", track(dense, 25, values(300, 7)), const_scale=1.0 / 3.0, round_digits=6)
bigwig_scenario("scaled_by_zero", track(dense[:40], 25, values(40, 7)), const_scale=0.0)
bigwig_scenario("negative_scale_is_no_scale", track(dense[:40], 25, values(40, 7)), const_scale=-2.0)
bigwig_scenario("half_way_values", track(dense[:8], 25, [0.000005, 0.000015, 0.000025, 2.5, 0.125, 1.0000050000001, 7.0, 0.000035]))
bigwig_scenario("single_interval", track([400], 50, [3.25]))
bigwig_scenario("chromosome_not_in_bigwig", track(dense[:5], 25, values(5)), chromosome="chrU", chroms={"chrT": 1000000})
bigwig_scenario("no_intervals", None)
bigwig_scenario("empty_intervals", [])
bigwig_scenario("chromosome_not_in_sizes", track(dense[:5], 25, values(5)), chromosome="chrZ", chroms={"chrZ": 10})
bigwig_scenario("non_finite_value", track(dense[:6], 25, [1.0, 2.0, float("inf"), 1.0, 0.0, 3.0]))
bigwig_scenario("nan_value", track(dense[:6], 25, [1.0, float("nan"), 2.0, 1.0, 0.0, 3.0]))
bigwig_scenario("non_positive_width", [(0, 25, 1.0), (25, 25, 2.0), (50, 75, 1.0)])
bigwig_scenario("variable_width", [(0, 25, 1.0), (25, 75, 2.0), (75, 100, 1.0)])
bigwig_scenario("misaligned_start", [(0, 25, 1.0), (30, 55, 2.0), (75, 100, 1.0)])
bigwig_scenario("duplicate_bin", [(0, 25, 1.0), (50, 75, 2.0), (50, 75, 3.0)])

np.savez_compressed(os.path.join(HERE, "assemble_vectors.npz"), **arrays)
with open(os.path.join(HERE, "assemble_vectors.json"), "w", encoding="utf-8") as handle:
    json.dump(meta, handle, indent=1, sort_keys=True)
print(f"wrote {len(meta['matrix'])} generate_chrom_matrix scenarios and {len(meta['bigwig'])} get_bigwig_chrom_scores scenarios, "
      f"{len(arrays)} arrays")
for kind in ("matrix", "bigwig"):
    for r in meta[kind]:
        print(f"  {kind:7s} {r['name']:32s} ->", r.get("error") or ("(None, None)" if r.get("none") else "arrays"))
