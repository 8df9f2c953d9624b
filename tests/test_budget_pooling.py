"""Host side of the budget estimation (rocco_amd/budget.py): empirical-Bayes pooling, the final clipping and the scalar
rules, against outputs of the reference's own functions (tests/golden/make_golden_budget.py).  The pooling is scalar
code over at most 24 pairs calling the same SciPy routines as the reference, so agreement is to the last place."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "budget_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_pooled_budgets_and_metadata(gold):
    from rocco_amd.budget import _resolve_budgets, estimate_empirical_bayes_budgets

    for key in gold["eb_names"]:
        q = float(key.rsplit("_q", 1)[1])
        counts = {f"c{i}": float(v) for i, v in enumerate(gold[f"{key}_counts"])}
        totals = {f"c{i}": float(v) for i, v in enumerate(gold[f"{key}_totals"])}
        budgets, meta = estimate_empirical_bayes_budgets(counts, totals, posterior_quantile=q)
        want_meta = json.loads(str(gold[f"{key}_meta"][0]))
        assert set(meta) == set(want_meta), key
        for name, value in want_meta.items():
            if isinstance(value, (str, bool)):
                assert meta[name] == value, (key, name)
            else:
                assert np.isclose(meta[name], value, rtol=1e-12, atol=0.0), (key, name, meta[name], value)
        assert np.allclose([budgets[c] for c in counts], gold[f"{key}_budgets"], rtol=1e-12, atol=0.0), key
        for budget_arg, scale in ((None, 1.0), (0.03, 1.5)):
            cache = {c: {"budget_count_hat": counts[c], "total_count": totals[c]} for c in counts}
            final, _ = _resolve_budgets(cache, {"budget_posterior_quantile": q, "budget": budget_arg, "scale_chrom_budgets": scale})
            got = np.array([final[c] for c in counts])
            assert np.allclose(got, gold[f"{key}_final_{budget_arg}_{scale}"], rtol=1e-12, atol=0.0), (key, budget_arg)
            assert np.all((got >= 0.005) & (got <= 0.1))  # rocco/rocco.py:1132-1141


def test_pooling_shrinks_towards_the_genome_wide_rate():
    from rocco_amd.budget import estimate_empirical_bayes_budgets

    counts = {"a": 6.0, "b": 48.0, "c": 14.0}
    totals = {"a": 1000.0, "b": 1000.0, "c": 1000.0}
    budgets, meta = estimate_empirical_bayes_budgets(counts, totals)
    assert meta["posterior_summary"] == "beta_quantile" and np.isclose(meta["posterior_quantile"], 0.01)
    assert meta["prior_strength"] > 0 and meta["prior_dispersion"] >= meta["min_prior_dispersion"]
    assert budgets["a"] < budgets["c"] < budgets["b"] and budgets["b"] < counts["b"] / totals["b"]
    low, _ = estimate_empirical_bayes_budgets(counts, totals, posterior_quantile=0.2)
    high, _ = estimate_empirical_bayes_budgets(counts, totals, posterior_quantile=0.4)
    assert all(low[c] <= high[c] for c in counts)  # a lower posterior quantile is the more conservative budget
    lone, meta1 = estimate_empirical_bayes_budgets({"x": 0}, {"x": 0})
    assert np.isclose(meta1["genome_wide_budget"], 0.05) and 0.0 < lone["x"] < 0.05
    with pytest.raises(ValueError):
        estimate_empirical_bayes_budgets({"a": 1.0, "b": 2.0}, {"b": 10.0, "a": 10.0})
    with pytest.raises(ValueError):
        estimate_empirical_bayes_budgets(counts, totals, posterior_quantile=1.0)


def test_scalar_rules():
    from rocco_amd import budget as b

    assert b._resolve_budget_ess_max_lag(512, 16) == 64 and b._resolve_budget_ess_max_lag(10**6) == 404
    assert b._resolve_budget_ess_max_lag(10, None) == 9 and b._resolve_budget_ess_max_lag(1) == 0
    assert b._resolve_budget_bootstrap_bandwidth(1) == 1 and b._resolve_budget_bootstrap_bandwidth(1000) == 10
    assert b._resolve_budget_bootstrap_bandwidth(100, 3) == 8 and b._resolve_budget_bootstrap_bandwidth(5, 50) == 4
    taps = b._build_budget_bootstrap_kernel(8)
    assert taps.shape == (17,) and np.isclose(np.sum(taps * taps), 1.0) and np.allclose(taps, taps[::-1]) and taps.argmax() == 8
    w = b._generate_dependent_wild_weights(5000, taps, np.random.default_rng(3))
    assert w.shape == (5000,) and abs(w.mean()) < 1e-12 and np.isclose(w.std(), 1.0)
    assert b._generate_dependent_wild_weights(1, taps, np.random.default_rng(3)).tolist() == [1.0]
    run = b._Running()
    for v in (1.0, 2.0, 4.0):
        run.add(v)
    assert np.isclose(run.mean, 7.0 / 3.0) and np.isclose(run.sd(), np.std([1.0, 2.0, 4.0], ddof=1))
