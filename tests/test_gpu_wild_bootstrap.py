"""GPU: the count-matrix budget estimator (rocco_amd.budget.estimate_budget_nonnull_fraction_from_wild_bootstrap_null,
the device form of rocco/inference.py:988-1148 over 719-985) against outputs of the reference's own function on the
same centred matrices (tests/golden/make_golden_composed.py, part 1).  The multipliers come from NumPy's generator on
the host in the reference's order, the rescoring and every n-long statistic run on the device: every entry of the
details equals the reference's bit for bit except the autocorrelation time and what is derived from it (the reference
takes the autocovariances from an FFT, the device sums lagged products: 1e-9, truncation lag equal)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wild_bootstrap_vectors.npz")
FROM_AUTOCORRELATION = {"autocorrelation_time", "effective_total_count", "effective_count"}


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _compare(name, fraction, details, gold):
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


def test_every_golden_case_matches_the_reference(gpu, gold):
    from rocco_amd.budget import estimate_budget_nonnull_fraction_from_wild_bootstrap_null as estimate

    assert len(gold["names"]) >= 11
    for name in gold["names"]:
        kwargs = json.loads(str(gold[f"{name}_kwargs"][0]))
        observed = gold[f"{name}_observed"] if f"{name}_observed" in gold.files else None
        fraction, details = estimate(gold[f"{name}_centered"], observed_scores=observed, return_details=True, **kwargs)
        _compare(name, fraction, details, gold)
        assert estimate(gold[f"{name}_centered"], observed_scores=observed, **kwargs) == fraction  # the bare return value


def test_the_references_own_expectations(gpu, gold):
    """tests/test_rocco.py:462-502, assertion by assertion, on the centred matrix the reference made from its test input."""
    from rocco_amd import budget

    name = "reference_test"
    fraction, meta = budget.estimate_budget_nonnull_fraction_from_empirical_null(
        gold[f"{name}_centered"], observed_scores=gold[f"{name}_observed"], dependence_lag_hint=16, num_null_draws=6,
        return_details=True)
    assert 0.0 < fraction <= 1.0
    assert np.isclose(fraction, meta["nonnull_fraction"])
    assert 0.0 <= meta["observed_positive_fraction"] <= 1.0
    assert 0.0 <= meta["null_positive_fraction"] <= 1.0
    assert meta["observed_excess_mass"] > meta["null_excess_mass"] > 0.0
    assert meta["observed_excess_units"] > meta["null_excess_units"] > 0.0
    assert meta["effective_count"] > 0.0
    assert 1.0 <= meta["effective_total_count"] <= meta["num_loci"]
    assert meta["autocorrelation_time"] >= 1.0
    assert meta["ess_max_lag"] == 64.0
    assert meta["null_method"] == "dependent_wild_residual_bootstrap"
    assert meta["num_null_draws"] == 6.0
    assert meta["max_null_draws"] == 6.0
    assert not meta["adaptive_stop"]
    assert meta["wild_bandwidth"] >= 8.0
    assert meta["null_excess_units_sd"] > 0.0
    assert meta["null_reference_mean_positive_consensus"] >= 0.0
    assert meta["negative_support_size"] > 0.0
    assert 0.0 < meta["negative_fraction"] <= 1.0
    # the third name of the same estimator (rocco/inference.py:1455-1485): the hint travels as `mean_block_length`
    again = budget.estimate_budget_nonnull_fraction_from_resampled_null(
        gold[f"{name}_centered"], observed_scores=gold[f"{name}_observed"], mean_block_length=16, num_null_draws=6)
    assert again == fraction


def test_device_tensors_in_and_errors(gpu, gold):
    import torch
    from rocco_amd.budget import estimate_budget_nonnull_fraction_from_wild_bootstrap_null as estimate

    name = "k3_n5000_pool4"
    kwargs = json.loads(str(gold[f"{name}_kwargs"][0]))
    centered_t = torch.from_numpy(gold[f"{name}_centered"]).to(gpu)
    observed_t = torch.from_numpy(gold[f"{name}_observed"]).to(gpu)
    fraction, details = estimate(centered_t, observed_scores=observed_t, return_details=True, **kwargs)
    _compare(name, fraction, details, gold)
    with pytest.raises(ValueError):  # rocco/inference.py:764-767
        estimate(centered_t, observed_scores=observed_t[:-1], **kwargs)
    with pytest.raises(ValueError):  # rocco/inference.py:1044-1045
        estimate(np.zeros((2, 3, 4)))
    with pytest.raises(ValueError):  # rocco/inference.py:1048-1049 / 242-243
        estimate(np.zeros((2, 0)))


def test_random_matrices_against_the_oracle(gpu, oracle):
    """Sizes and settings the fixtures do not hold, against the oracle's restatement of the estimator (itself checked
    against the same fixtures in tests/test_oracle_golden.py)."""
    from rocco_amd.budget import estimate_budget_nonnull_fraction_from_wild_bootstrap_null as estimate

    rng = np.random.default_rng(5)
    for K, n, kwargs in ((2, 8191, dict(num_null_draws=9, num_processes=2)),
                         (7, 33000, dict(num_null_draws=10, dependence_lag_hint=101, random_seed=3)),
                         (1, 100, dict(num_null_draws=4, min_effect=0.1, prior_df=6.0))):
        e = rng.normal(0.0, 0.5, size=(K, n + 4))
        centered = np.stack([np.convolve(row, np.ones(5) / 2.0, mode="valid") for row in e])
        for p in rng.integers(0, max(1, n - 30), size=max(1, n // 400)):
            centered[:, p:p + 20] += 2.0
        fraction, details = estimate(centered, return_details=True, **kwargs)
        o_fraction, o_details = oracle.estimate_budget_nonnull_fraction_from_wild_bootstrap_null(
            centered, return_details=True, **kwargs)
        assert set(details) == set(o_details)
        for key, value in o_details.items():
            if key in FROM_AUTOCORRELATION:
                assert np.isclose(details[key], value, rtol=1e-9, atol=1e-12), (K, n, key)
            else:
                assert details[key] == value, (K, n, key, details[key], value)
        assert fraction == o_fraction
