"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol that
include/rocco_hip.h declares (no compute calls without a GPU), host helpers match NumPy, and the
product refuses to run without its HIP device instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from rocco_amd import _native

    header = open(os.path.join(ROOT, "include", "rocco_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(rocco_hip_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 12
    if not os.path.isfile(_native.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rocco_hip.h but not exported"
    bound = {p[0] for p in _native.PROTOTYPES}
    assert set(declared) == bound, "ctypes prototypes and header disagree"
    assert _native.load().rocco_hip_abi_version() == 1000


def test_struct_layouts_match_header_sizes():
    from rocco_amd import _native

    assert ctypes.sizeof(_native.BudgetTask) == 64
    assert ctypes.sizeof(_native.ProbeStats) == 32
    assert ctypes.sizeof(_native.WindowStats) == 40 + 16 * (8 + 8 + 8 + 8 + 4 + 4)


def test_pairwise_sum_matches_numpy():
    from rocco_amd.dp import sum_constant_like_numpy

    assert np.getbufsize() == 8192, "the emulation assumes NumPy's default ufunc buffer size"
    rng = np.random.default_rng(0)
    for _ in range(300):
        length = int(rng.choice([0, 1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 65537, 934199,
                                 int(rng.integers(1, 3_000_000))]))
        gamma = float(rng.choice([1.0, 0.5, 10.0, 0.73, 1 / 3, rng.uniform(0, 10)]))
        want = float(np.sum(np.full(length, gamma))) if length else 0.0
        assert sum_constant_like_numpy(gamma, length) == want, (length, gamma)


def test_bed_helpers_match_golden(tmp_path):
    from rocco_amd import rocco as rr

    gold = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
    files = []
    for chrom in ("chr2", "chr10", "chr1"):
        path = tmp_path / f"in_{chrom}.bed"
        path.write_text("".join(f"{chrom}\t{a}\t{b}\n" for a, b in ((100, 200), (200, 260), (500, 650), (640, 700))))
        files.append(str(path))
    out = rr.combine_chrom_results(files, str(tmp_path / "combined.bed"))
    assert open(out, "rb").read() == gold["combine_text"].tobytes()
    recs, extra = rr._read_bed_records(out)
    assert not extra and recs[0][0] == "chr1" and recs[2][0] == "chr10"
    assert rr._merge_bed_records([("c", 5, 9), ("c", 0, 5), ("c", 20, 22)], min_length_bp=3) == [("c", 0, 9)]
    with pytest.raises(FileNotFoundError):
        rr.combine_chrom_results([str(tmp_path / "missing.bed")], str(tmp_path / "x.bed"))
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t5\n")
    with pytest.raises(ValueError):
        rr._read_bed_records(str(bad))


def test_synth_generators_are_deterministic():
    from rocco_amd import synth

    a = synth.hash_matrix(4, 3000, seed=7)
    b = synth.hash_matrix(4, 1000, seed=7, j0=1000)
    assert np.array_equal(a[:, 1000:2000], b)
    assert a.min() > 0 and 0.2 < np.median(a) < 0.4 and a.max() > 2.0
    assert np.array_equal(np.round(a, 5), a)
    assert dict(synth.chrom_loci(50))["chr1"] == 4979129 and sum(n for _, n in synth.chrom_loci(50)) == 61765409
    assert sum(n for _, n in synth.chrom_loci(10)) == 308826993


def test_lpt_partition_bounds():
    from rocco_amd import shard, synth

    sizes = [n for _, n in synth.chrom_loci(50)]
    for ranks, bound in ((1, 61765409), (2, 31000000), (8, 8001116)):
        owned = shard.lpt_partition(sizes, ranks)
        assert sorted(i for part in owned for i in part) == list(range(24))
        assert shard.makespan(sizes, owned) <= bound
    assert shard.makespan(sizes, shard.lpt_partition(sizes, 8)) == 8001116


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rocco_amd import score_central_tendency_chrom, solve_chrom_exact

    with pytest.raises(RuntimeError):
        solve_chrom_exact(np.ones(10), budget=0.1)
    with pytest.raises(RuntimeError):
        score_central_tendency_chrom(np.ones((3, 10)))


def test_argument_validation_like_reference():
    from rocco_amd import build_switch_costs, score_central_tendency_chrom

    assert build_switch_costs(np.zeros(1)).shape == (0,)
    assert build_switch_costs(np.zeros(5), gamma=2.0).tolist() == [2.0] * 4
    with pytest.raises(ValueError):
        build_switch_costs(np.zeros((2, 2)))
    with pytest.raises(ValueError):
        score_central_tendency_chrom(np.zeros(5))
