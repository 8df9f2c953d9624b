"""GPU: centred-WLS locus scores through the C ABI (rocco_hip_score_centered_wls_f64) against the golden
vectors of the reference's backend and against the CPU oracle: every track bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wls_vectors.npz")
ORDER = ("mean", "raw", "prior", "mod", "se", "scores")


def stack(res):
    scores, mean, raw, prior, mod, se, df, window = res
    return np.stack([mean, raw, prior, mod, se, scores]), np.array([df, window])


def test_golden_wls_tracks_bit_for_bit(gpu):
    from rocco_amd.inference import score_centered_wls

    gold = np.load(GOLD)
    for name in gold["names"]:
        lbz, pdf, me, use_me, win, pfr = gold[f"{name}_params"]
        tracks, dfw = stack(score_centered_wls(gold[f"{name}_matrix"], lower_bound_z=lbz, prior_df=pdf,
                                               min_effect=(me if use_me else None), spatial_window=int(win),
                                               precision_floor_ratio=pfr))
        want = gold[f"{name}_tracks"]
        for t, label in enumerate(ORDER):
            assert tracks[t].tobytes() == want[t].tobytes(), (name, label)
        assert np.array_equal(dfw, gold[f"{name}_df_window"]), name


@pytest.mark.parametrize("K,n", [(1, 5), (2, 31), (3, 32), (17, 257), (16, 287), (33, 4099), (5, 70001), (3, 1200003)])
def test_random_matrices_match_oracle(gpu, oracle, K, n):
    from rocco_amd.inference import score_centered_wls

    rng = np.random.default_rng(K * 7919 + n)
    m = rng.normal(0.0, 1.0, size=(K, n)) * np.exp(rng.normal(0.0, 0.5, size=(1, n)))
    cases = [(m, {}), (np.round(m, 1), {"min_effect": 0.25}),  # rounded: ties in |value| and in the variances
             (m, {"spatial_window": 7, "prior_df": 0.0, "precision_floor_ratio": 0.5}),
             (m, {"spatial_window": 63, "lower_bound_z": 0.0})]
    if n > 100000:
        cases = cases[:2]
    for mat, kw in cases:
        got, dfw = stack(score_centered_wls(mat, **kw))
        want, dfw_o = stack(oracle.score_centered_wls(mat, **kw))
        for t, label in enumerate(ORDER):
            assert got[t].tobytes() == want[t].tobytes(), (K, n, kw, label)
        assert np.array_equal(dfw, dfw_o)


def test_constant_and_zero_rows(gpu, oracle):
    from rocco_amd.inference import score_centered_wls

    m = np.zeros((3, 500))
    m[1] = 2.5
    m[2, 100:200] = np.linspace(-1, 1, 100)
    got, _ = stack(score_centered_wls(m))
    want, _ = stack(oracle.score_centered_wls(m))
    assert got.tobytes() == want.tobytes()


def test_argument_errors(gpu):
    from rocco_amd.inference import _score_centered_wls_matrix, score_centered_wls

    with pytest.raises(ValueError):
        score_centered_wls(np.zeros((0, 5)))
    with pytest.raises(ValueError):
        _score_centered_wls_matrix(np.zeros(5))
    bad = np.ones((2, 100))
    bad[1, 7] = np.nan
    with pytest.raises(ValueError):
        score_centered_wls(bad)


@pytest.mark.parametrize("K,n,window", [(2, 1000, 65), (3, 1000, 101), (1, 64, 64), (4, 5000, 1001), (2, 300, 10**6), (3, 70001, 255)])
def test_windows_above_the_tiled_kernel_limit(gpu, oracle, K, n, window):
    """The reference takes any spatial window (wls_backend.c:610-742, resolved at 232-260); above 63 loci the
    rolling sums run straight from memory (slower) with the same results."""
    from rocco_amd.inference import score_centered_wls

    rng = np.random.default_rng(n + window)
    m = np.round(rng.normal(0.0, 1.0, size=(K, n)) * np.exp(rng.normal(0.0, 0.5, size=(1, n))), 3)
    got, dfw = stack(score_centered_wls(m, spatial_window=window))
    want, dfw_o = stack(oracle.score_centered_wls(m, spatial_window=window))
    for t, label in enumerate(ORDER):
        assert got[t].tobytes() == want[t].tobytes(), (K, n, window, label)
    assert np.array_equal(dfw, dfw_o) and dfw[1] >= 63


def test_wrapper_details(gpu, oracle):
    from rocco_amd.inference import _score_centered_wls_matrix

    rng = np.random.default_rng(5)
    m = rng.normal(size=(6, 3000))
    scores, details = _score_centered_wls_matrix(m, min_effect=0.1)
    o = oracle.score_centered_wls(m, min_effect=0.1)
    assert scores.tobytes() == o[0].tobytes()
    assert details["mean"].tobytes() == o[1].tobytes() and details["standard_error"].tobytes() == o[5].tobytes()
    assert details["prior_spatial_window"] == 31.0 and details["min_effect"] == 0.1
    assert np.all(details["degrees_of_freedom"] == o[6])


@pytest.mark.parametrize("kind", ["quantised", "few_levels", "zeros_and_noise", "boundary_tie_only"])
def test_trend_fit_with_runs_of_equal_values(gpu, oracle, kind):
    """Rows of 4096 loci or more fit their variance trend without sorting the pairs: a few order statistics of |value| give the
    bins' boundaries, and a pair knows its bin from its |value| alone -- unless a boundary falls inside a run of equal
    |value|, where the reference's order inside the run (by variance) decides: such rows take the sorted path.  Rows
    full of ties, rows with a single tie exactly at a boundary, and clean rows side by side in one matrix."""
    from rocco_amd.inference import score_centered_wls, wls_sorted_rows

    rng = np.random.default_rng(sum(map(ord, kind)))
    K, n = 5, 30011
    m = rng.normal(0.0, 0.8, size=(K, n))
    if kind == "quantised":
        m = np.round(m, 1)  # ~60 distinct |values|: every boundary inside a run
    elif kind == "few_levels":
        m = rng.choice(np.array([-2.0, -0.5, 0.0, 0.5, 1.5]), size=(K, n))
    elif kind == "zeros_and_noise":
        m[:, rng.random(n) < 0.6] = 0.0
        m[3] = rng.normal(size=n)  # one clean row among them
    else:
        # rows 0 and 2 clean; row 1: exactly two equal |values| placed at the ranks around one bin boundary
        bins = int(np.floor(1.0 + np.log2(n + 1.0)))
        order = np.argsort(np.abs(m[1]))
        left = (7 * n) // bins
        m[1, order[left]] = -m[1, order[left - 1]]
    got = score_centered_wls(m)
    assert wls_sorted_rows() == {"quantised": 5, "few_levels": 5, "zeros_and_noise": 4, "boundary_tie_only": 1}[kind]
    want = oracle.score_centered_wls(m)
    for g, w in zip(got[:6], want[:6]):
        assert np.asarray(g).tobytes() == np.asarray(w).tobytes(), kind
    assert got[6:] == want[6:]


@pytest.mark.parametrize("kind", ["wide_range", "dense_cell", "tiny_values", "two_binades"])
def test_trend_fit_rank_finder_shapes(gpu, oracle, kind):
    """The order statistics of |value| the bins need come from two histogram passes (14 key bits, then 6..11 more inside
    the buckets that hold a wanted rank) and a gather of the cells that hold one (csrc/wls.hip, "the x side"): rows whose
    ranks spread over more than a hundred buckets (narrow second digit), rows with more than 8192 distinct values in one
    cell (sorted path instead), rows of subnormal and zero values, rows inside two binades (wide second digit)."""
    from rocco_amd.inference import score_centered_wls, wls_sorted_rows

    rng = np.random.default_rng(11)
    K, n = 3, 70001
    if kind == "wide_range":
        m = np.exp(rng.normal(0.0, 12.0, size=(K, n))) * rng.choice([-1.0, 1.0], size=(K, n))
    elif kind == "dense_cell":
        n = 200003
        m = rng.normal(0.0, 0.8, size=(K, n))
        # 9000 distinct values that share their top 40 key bits, inside one bin of row 1 (its median among them)
        bins = int(np.floor(1.0 + np.log2(n + 1.0)))
        width = n // bins
        order = np.argsort(np.abs(m[1]))
        lo = 6 * n // bins + (width - 9000) // 2
        base = abs(m[1, order[lo]])
        m[1, order[lo:lo + 9000]] = base * (1.0 + np.arange(9000) * 2.0 ** -50)
        assert np.unique(np.abs(m[1])).size == n
    elif kind == "tiny_values":
        m = rng.normal(0.0, 1.0, size=(K, n)) * 1.0e-310
        m[0, rng.random(n) < 0.001] = 0.0
    else:
        m = rng.uniform(1.0, 4.0, size=(K, n)) * rng.choice([-1.0, 1.0], size=(K, n))
    got = score_centered_wls(m)
    assert wls_sorted_rows() == (1 if kind == "dense_cell" else 0)  # the sort-free path is the one that ran
    want = oracle.score_centered_wls(m)
    for g, w in zip(got[:6], want[:6]):
        assert np.asarray(g).tobytes() == np.asarray(w).tobytes(), kind
    assert got[6:] == want[6:]


@pytest.mark.parametrize("group", [1, 2, 4, 8])
@pytest.mark.parametrize("K,n,window", [(1, 64, 31), (3, 700, 5), (9, 5000, 31), (17, 1300, 63), (8, 129, 63), (5, 40000, 7)])
def test_rolling_variances_every_group_shape(gpu, oracle, monkeypatch, group, K, n, window):
    """The rolling kernel runs 1, 2, 4 or 8 rows per workgroup (tiles of 512 / 256 / 128 / 64 start positions) depending on how
    many rows a call has; here every shape on the same matrices: row counts that do not fill the last group, rows shorter
    than a tile, windows at both ends of the tiled range, enough tiles for the ring of LDS lines to wrap several times."""
    from rocco_amd.inference import score_centered_wls

    monkeypatch.setenv("ROCCO_HIP_ROLLING_GROUP", str(group))
    rng = np.random.default_rng(K * 1000 + n + window)
    m = np.round(rng.normal(0.0, 1.0, size=(K, n)) * np.exp(rng.normal(0.0, 0.5, size=(1, n))), 3)
    got, dfw = stack(score_centered_wls(m, spatial_window=window))
    want, dfw_o = stack(oracle.score_centered_wls(m, spatial_window=window))
    for t, label in enumerate(ORDER):
        assert got[t].tobytes() == want[t].tobytes(), (group, K, n, window, label)
    assert np.array_equal(dfw, dfw_o)
