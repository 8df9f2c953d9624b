"""GPU: signal-matrix assembly (generate_chrom_matrix, rocco/readtracks.py:521-633, and get_bigwig_chrom_scores, 94-186;
SURVEY.md section 8 (f), item 2) through the C ABI: against fixtures the REFERENCE's own two functions wrote
(tests/golden/make_golden_assemble.py: its per-file readers replaced in the module, a stand-in pyBigWig) and against the
oracle's NumPy statements on larger random tracks -- arrays bit for bit, errors word for word."""
import os
import types

import numpy as np
import pytest

import assemble_golden as ag

pytestmark = pytest.mark.gpu


def test_generate_chrom_matrix_as_the_reference_ran_it(gpu, monkeypatch):
    """Every scenario of tests/golden/assemble_vectors.json through rocco_amd.readtracks.generate_chrom_matrix, called as the
    reference's function is called and with the per-file readers replaced in the module as the fixture script replaced the
    reference's: ragged, repeated and unsorted starts, a file without data (excluded), none with data ((None, None)),
    bigWig tracks of two phases, a broken step (the reference's ValueError text), --low_memory float32, one track."""
    from rocco_amd import readtracks

    arrays, meta = ag.load()
    for record in meta["matrix"]:
        name = record["name"]
        table = dict(zip(record["files"], ag.matrix_tracks(arrays, record)))
        calls = []

        def reader(track_file, *args, _table=table, _calls=calls):
            _calls.append((track_file, len(args)))
            entry = _table[track_file]
            return (None, None) if entry is None else (entry[0], entry[1])

        monkeypatch.setattr(readtracks, "get_bam_chrom_reads", reader)
        monkeypatch.setattr(readtracks, "get_bigwig_chrom_scores", reader)
        call = lambda: readtracks.generate_chrom_matrix("chrT", list(record["files"]), "unused.sizes", 50, num_processors=1, **record["kwargs"])
        if "error" in record:
            with pytest.raises(ValueError) as info:
                call()
            assert str(info.value) == record["error"], name
            continue
        intervals, matrix = call()
        # (one call per file, in order, with the reference's positional arguments: 16 after the file for BAM, 4 for bigWig)
        assert [c[0] for c in calls] == record["files"] and {c[1] for c in calls} <= {16, 4}, name
        if record.get("none"):
            assert intervals is None and matrix is None, name
            continue
        want_i, want_m = arrays[f"m_{name}_intervals"], arrays[f"m_{name}_matrix"]
        assert intervals.dtype == want_i.dtype and np.array_equal(intervals, want_i), name
        assert matrix.dtype == want_m.dtype and matrix.shape == want_m.shape and matrix.tobytes() == want_m.tobytes(), name


def test_get_bigwig_chrom_scores_as_the_reference_ran_it(gpu, monkeypatch, tmp_path):
    """Every get_bigwig_chrom_scores scenario of the fixture through rocco_amd.readtracks.get_bigwig_chrom_scores over the
    same stand-in pyBigWig object: gaps zero-filled, np.round at 0 / 2 / 5 / 6 digits incl. half-way values, the constant
    scale (1/2, 1/3, 0, negative = none), the five validations and the three early returns, the handle closed every time."""
    from rocco_amd import readtracks

    arrays, meta = ag.load()
    sizes = tmp_path / "t.sizes"
    sizes.write_text("chrT\t1000000\nchrU\t5000\n")
    bw = tmp_path / "t.bw"
    bw.write_bytes(b"")
    for record in meta["bigwig"]:
        name = record["name"]
        intervals = ag.bigwig_intervals(arrays, record)
        chroms = {c: 1 for c in record["chroms"]}
        handle = ag.FakeBigWig(chroms, {record["chromosome"]: intervals} if intervals is not None else {})
        fake = types.ModuleType("pyBigWig")
        fake.open = lambda _path, _handle=handle: _handle
        monkeypatch.setattr(readtracks, "pyBigWig", fake)
        call = lambda: readtracks.get_bigwig_chrom_scores(str(bw), record["chromosome"], str(sizes), **record["kwargs"])
        if "error" in record:
            with pytest.raises(ValueError) as info:
                call()
            assert str(info.value) == record["error"].replace("{file}", str(bw)).replace("{sizes}", str(sizes)), name
            continue
        out_i, out_v = call()
        assert handle.closed, name
        if record.get("none"):
            assert out_i is None and out_v is None, name
            continue
        want_i, want_v = arrays[f"b_{name}_intervals"], arrays[f"b_{name}_out"]
        assert out_i.dtype == want_i.dtype and np.array_equal(out_i, want_i), name
        assert out_v.dtype == want_v.dtype and out_v.tobytes() == want_v.tobytes(), name
    monkeypatch.setattr(readtracks, "pyBigWig", None)
    with pytest.raises(ImportError, match="pyBigWig"):
        readtracks.get_bigwig_chrom_scores(str(bw), "chrT", str(sizes))
    with pytest.raises(FileNotFoundError):
        readtracks.get_bigwig_chrom_scores(str(tmp_path / "missing.bw"), "chrT", str(sizes))
    with pytest.raises(RuntimeError, match="does not decode BAM"):
        readtracks.generate_chrom_matrix("chrT", ["a.bam"], str(sizes), 50)


def make_tracks(rng, K, n, step, ragged, duplicates):
    ints, vals = [], []
    for k in range(K):
        lo = int(rng.integers(0, 5)) if ragged else 0
        hi = n - (int(rng.integers(0, 5)) if ragged else 0)
        a = 1000 + step * np.arange(lo, hi, dtype=np.int64)
        if ragged:
            a = a[rng.random(a.size) > 0.1]
        if duplicates and a.size > 4:
            a = np.concatenate([a, a[rng.integers(0, a.size, size=max(1, a.size // 7))]])  # unsorted, repeated loci
        ints.append(a)
        vals.append(np.round(rng.gamma(1.0, 2.0, size=a.size), 5))
    return ints, vals


@pytest.mark.parametrize("K,n,step", [(1, 1, 50), (2, 5, 50), (3, 1000, 50), (10, 70001, 10), (4, 600000, 50)])
@pytest.mark.parametrize("ragged,duplicates", [(False, False), (True, False), (True, True)])
def test_matrix_equals_numpy_statements(gpu, oracle, K, n, step, ragged, duplicates):
    from rocco_amd.readtracks import assemble_chrom_matrix

    rng = np.random.default_rng(K * 1000 + n + ragged + 2 * duplicates)
    ints, vals = make_tracks(rng, K, n, step, ragged, duplicates)
    for low_memory in (False, True):
        want_i, want_m = oracle.assemble_chrom_matrix(ints, vals, low_memory=low_memory)
        got_i, got_m = assemble_chrom_matrix(ints, vals, low_memory=low_memory)
        assert got_i.dtype == want_i.dtype and np.array_equal(got_i, want_i)
        assert got_m.dtype == want_m.dtype and got_m.shape == want_m.shape and got_m.tobytes() == want_m.tobytes()


def test_bigwig_fixed_step_check(gpu, oracle):
    from rocco_amd.readtracks import assemble_chrom_matrix

    a = np.arange(0, 5000, 50, dtype=np.int64)
    b = np.arange(25, 5000, 50, dtype=np.int64)  # another phase: the union has steps of 25 -> still one step
    c = np.array([0, 50, 100, 175], dtype=np.int64)
    ok_i, ok_m = assemble_chrom_matrix([a, b], [np.ones(a.size), 2 * np.ones(b.size)], track_type="bigwig")
    want_i, want_m = oracle.assemble_chrom_matrix([a, b], [np.ones(a.size), 2 * np.ones(b.size)], track_type="bigwig")
    assert np.array_equal(ok_i, want_i) and ok_m.tobytes() == want_m.tobytes()
    with pytest.raises(ValueError):
        oracle.assemble_chrom_matrix([a, c], [np.ones(a.size), np.ones(c.size)], track_type="bigwig", chromosome="chrT")
    with pytest.raises(ValueError, match="chrT do not share one fixed binning scheme"):
        assemble_chrom_matrix([a, c], [np.ones(a.size), np.ones(c.size)], track_type="bigwig", chromosome="chrT")
    # the same tracks pass as BAM counts (no check, readtracks.py:615)
    i1, m1 = assemble_chrom_matrix([a, c], [np.ones(a.size), np.ones(c.size)])
    i2, m2 = oracle.assemble_chrom_matrix([a, c], [np.ones(a.size), np.ones(c.size)])
    assert np.array_equal(i1, i2) and m1.tobytes() == m2.tobytes()


def test_assembled_matrix_feeds_the_scoring_kernel(gpu):
    import torch

    from rocco_amd.readtracks import assemble_chrom_matrix_device
    from rocco_amd.rocco import score_central_tendency_chrom_device

    rng = np.random.default_rng(4)
    ints, vals = make_tracks(rng, 5, 20000, 50, True, False)
    common_t, matrix_t = assemble_chrom_matrix_device(ints, vals)
    scores = score_central_tendency_chrom_device(matrix_t).cpu().numpy()
    assert np.array_equal(scores, np.median(matrix_t.cpu().numpy(), axis=0))
    assert common_t.dtype == torch.int64 and common_t.shape[0] == matrix_t.shape[1]


@pytest.mark.parametrize("n,step,gaps", [(1, 50, False), (7, 50, True), (1000, 10, True), (200000, 50, True), (50000, 25, False)])
def test_bigwig_dense_fill_equals_numpy_statements(gpu, oracle, n, step, gaps):
    from rocco_amd.readtracks import bigwig_dense_fill

    rng = np.random.default_rng(n + step)
    grid = 3000 + step * np.arange(n, dtype=np.int64)
    keep = np.ones(n, dtype=bool)
    if gaps and n > 3:
        keep[1:-1] = rng.random(n - 2) > 0.3  # missing bins are filled with zeros
    starts = grid[keep]
    vals = rng.gamma(1.0, 3.0, size=starts.size) * 10.0 ** rng.integers(-6, 3, size=starts.size)
    vals[rng.random(vals.size) < 0.05] = 0.125  # exact ties for round-half-to-even
    for const_scale, digits in ((1.0, 5), (0.37, 2), (-1.0, 0), (2.5, -1), (0.0, 3)):
        want_i, want_v = oracle.bigwig_dense_fill(starts, starts + step, vals, const_scale, digits)
        got_i, got_v = bigwig_dense_fill(starts, starts + step, vals, const_scale, digits)
        assert got_i.dtype == want_i.dtype and np.array_equal(got_i, want_i)
        assert got_v.tobytes() == want_v.tobytes(), (n, const_scale, digits)


def test_bigwig_dense_fill_errors_like_the_reference(gpu, oracle):
    from rocco_amd.readtracks import bigwig_dense_fill

    s = np.array([0, 50, 100, 150], dtype=np.int64)
    v = np.ones(4)
    cases = {
        "non-finite": (s, s + 50, np.array([1.0, np.nan, 1.0, 1.0])),
        "non-positive widths": (s, np.array([50, 50, 150, 200]), v),
        "variable-width": (s, np.array([50, 100, 160, 200]), v),
        "not aligned": (np.array([0, 50, 110, 160]), np.array([50, 100, 160, 210]), v),
        "overlapping or duplicate": (np.array([0, 50, 50, 100]), np.array([50, 100, 100, 150]), v),
    }
    for needle, (a, b, c) in cases.items():
        with pytest.raises(ValueError, match=needle):
            oracle.bigwig_dense_fill(a, b, c, bigwig_file="t.bw", chromosome="chrZ")
        with pytest.raises(ValueError, match=needle):
            bigwig_dense_fill(a, b, c, bigwig_file="t.bw", chromosome="chrZ")
