"""GPU: signal-matrix assembly (the tail of generate_chrom_matrix, rocco/readtracks.py:614-633; SURVEY.md section 8
(f), item 2) through the C ABI against the same NumPy statements: exact.  (The reference's tests reach this code
only through file readers that are not installed here, so there is no reference fixture for it.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
