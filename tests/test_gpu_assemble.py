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
