"""GPU: the score-track budget estimate and the automatic switch cost (rocco_amd/budget.py) against outputs of the
reference's own functions on the same tracks (tests/golden/make_golden_budget.py).  Given the same host multipliers
every statistic is the reference's bit for bit -- the order statistics come from a device sort, the means are summed
in NumPy's order -- except the autocorrelation time and what is derived from it: the reference takes the
autocovariances from an FFT, the device sums lagged products (tolerance 1e-9; the truncation lag must be equal)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "budget_vectors.npz")
FROM_AUTOCORRELATION = {"autocorrelation_time", "effective_total_count", "effective_count"}


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_score_track_estimate_matches_the_reference(gpu, gold):
    from rocco_amd.budget import _resolve_chrom_gamma, estimate_budget_nonnull_fraction_from_score_track

    for name in gold["names"]:
        scores = gold[f"{name}_scores"]
        draws, hint = (int(v) for v in gold[f"{name}_params"])
        fraction, details = estimate_budget_nonnull_fraction_from_score_track(
            scores, dependence_lag_hint=None if hint < 0 else hint, num_null_draws=draws, return_details=True)
        want = json.loads(str(gold[f"{name}_details"][0]))
        assert set(details) == set(want), name
        for key, value in want.items():
            if isinstance(value, (str, bool)):
                assert details[key] == value, (name, key)
            elif key in FROM_AUTOCORRELATION:
                assert np.isclose(details[key], value, rtol=1e-9, atol=1e-12), (name, key, details[key], value)
            else:
                assert details[key] == value, (name, key, details[key], value)
        assert fraction == float(gold[f"{name}_fraction"][0]), name
        gamma, meta = _resolve_chrom_gamma("chrT", {"gamma": None}, scores, details)
        g_want = gold[f"{name}_gamma"]
        assert meta["characteristic_run_length"] == int(g_want[2]) and meta["positive_score_count"] == int(g_want[4]), name
        assert meta["positive_score_median"] == g_want[3] and gamma == g_want[0] and meta["gamma_raw"] == g_want[5], name


def test_effective_sample_size_matches_the_reference(gpu, gold):
    from rocco_amd.budget import _estimate_effective_sample_size

    for name in gold["names"]:
        scores = gold[f"{name}_scores"]
        want_det = json.loads(str(gold[f"{name}_details"][0]))
        soft = np.clip(scores - want_det["null_center"], 0.0, None) / max(want_det["null_scale"], 1.0e-6)
        ess, tau, lags = _estimate_effective_sample_size(soft, int(want_det["ess_max_lag"]))
        w_ess, w_tau, w_lags = gold[f"{name}_ess"]
        assert lags == int(w_lags), name
        assert np.isclose(tau, w_tau, rtol=1e-9) and np.isclose(ess, w_ess, rtol=1e-9), (name, tau, w_tau)


def test_fixed_gamma_and_device_tensor_input(gpu, gold):
    import torch
    from rocco_amd.budget import _resolve_chrom_gamma, build_chrom_cache_from_tracks, _resolve_budgets

    assert _resolve_chrom_gamma("c", {"gamma": 2.5}, np.zeros(3), {}) == (2.5, None)
    with pytest.raises(ValueError):
        _resolve_chrom_gamma("c", {"gamma": -1.0}, np.zeros(3), {})
    # the bigWig branch of the cache builder on matrices held in HBM: medians -> estimate -> switch cost -> pooled budgets
    rng = np.random.default_rng(2)
    tracks = {}
    for k, n in enumerate((6000, 9000, 4000, 7000)):
        m = np.round(rng.gamma(1.0, 0.3, size=(3, n)), 5)
        for p in rng.integers(0, n, size=n // 300):
            m[:, p:p + 20] += rng.gamma(5.0, 1.0)
        tracks[f"chr{k + 1}"] = (np.arange(n) * 50, torch.from_numpy(m).to(gpu))
    cache = build_chrom_cache_from_tracks(tracks, {"budget_null_draws": 10, "gamma": None})
    for chrom, (_iv, m_t) in tracks.items():
        entry = cache[chrom]
        assert np.array_equal(entry["scores"].cpu().numpy(), np.median(m_t.cpu().numpy(), axis=0))
        assert 0.5 <= entry["gamma"] <= 10.0 and entry["gamma_meta"]["method"] == "auto_score_autocorr"
        assert 1.0 <= entry["total_count"] <= entry["num_loci"] and 0.0 <= entry["budget_count_hat"] <= entry["total_count"]
    budgets, meta = _resolve_budgets(cache, {"budget_posterior_quantile": 0.01, "budget": None, "scale_chrom_budgets": 1.0})
    assert meta["prior_fit_method"] == "beta_binomial_mle" and all(0.005 <= b <= 0.1 for b in budgets.values())


def test_host_threads_do_not_change_the_estimate(gpu, gold, monkeypatch):
    """`num_processes` > 1 runs the FFT convolutions of the draws in host threads, several draws ahead of the one being
    consumed (the draws' normals still come from the one generator in order): every statistic must be what one worker gives
    -- also when a draw is degenerate, where the generator is put back and the rest runs in sequence."""
    from rocco_amd import budget

    for name in list(gold["names"])[:6]:
        scores = gold[f"{name}_scores"]
        draws, hint = (int(v) for v in gold[f"{name}_params"])
        kw = dict(dependence_lag_hint=None if hint < 0 else hint, num_null_draws=draws, return_details=True)
        one = budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_processes=1, **kw)
        for workers in (2, 5):
            many = budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_processes=workers, **kw)
            assert many == one, (name, workers)
    # a degenerate draw in the middle: the smoothing of the third draw "fails", signs are drawn instead
    scores = gold[f"{gold['names'][0]}_scores"]
    real, calls = budget._smooth_and_standardise, {"n": 0}

    def sometimes_degenerate(normals, taps):
        calls["n"] += 1
        return None if normals[0] > 1.2 else real(normals, taps)  # (a property of the draw itself: the same draws in any order)

    monkeypatch.setattr(budget, "_smooth_and_standardise", sometimes_degenerate)
    one = budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_null_draws=40, min_null_draws=40, num_processes=1, return_details=True)
    many = budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_null_draws=40, min_null_draws=40, num_processes=4, return_details=True)
    assert many == one and calls["n"] >= 80
