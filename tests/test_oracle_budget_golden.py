"""CPU: the oracle's restatement of the budget / switch-cost estimation and of the composed driver
(oracle/budget_oracle.py) against outputs of the reference's own functions (tests/golden/make_golden_budget.py,
make_golden_composed.py).  Same NumPy / SciPy calls in the same order, so everything is bit for bit here -- including
the FFT-derived autocorrelation time, which the device path only reproduces to 1e-9."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(got: dict, want: dict, where):
    assert set(got) == set(want), where
    for key, value in want.items():
        assert got[key] == value, (where, key, got[key], value)


def test_score_track_estimate_and_switch_cost(oracle):
    gold = np.load(os.path.join(GOLDEN, "budget_vectors.npz"))
    for name in gold["names"]:
        scores = gold[f"{name}_scores"]
        draws, hint = (int(v) for v in gold[f"{name}_params"])
        fraction, details = oracle.estimate_budget_nonnull_fraction_from_score_track(
            scores, dependence_lag_hint=None if hint < 0 else hint, num_null_draws=draws, return_details=True)
        _same(details, json.loads(str(gold[f"{name}_details"][0])), name)
        assert fraction == float(gold[f"{name}_fraction"][0])
        gamma, meta = oracle.resolve_chrom_gamma({"gamma": None}, scores, details)
        want = gold[f"{name}_gamma"]
        assert [gamma, meta["autocorrelation_time"], meta["characteristic_run_length"], meta["positive_score_median"],
                meta["positive_score_count"], meta["gamma_raw"]] == want.tolist(), name


def test_pooled_budgets(oracle):
    gold = np.load(os.path.join(GOLDEN, "budget_vectors.npz"))
    for key in gold["eb_names"]:
        q = float(key.rsplit("_q", 1)[1])
        counts = {f"c{i}": float(v) for i, v in enumerate(gold[f"{key}_counts"])}
        totals = {f"c{i}": float(v) for i, v in enumerate(gold[f"{key}_totals"])}
        budgets, meta = oracle.estimate_empirical_bayes_budgets(counts, totals, posterior_quantile=q)
        assert [budgets[c] for c in counts] == gold[f"{key}_budgets"].tolist(), key
        _same(meta, json.loads(str(gold[f"{key}_meta"][0])), key)
        for budget_arg, scale in ((None, 1.0), (0.03, 1.5)):
            cache = {c: {"budget_count_hat": counts[c], "total_count": totals[c]} for c in counts}
            final, _ = oracle.resolve_budgets(cache, {"budget_posterior_quantile": q, "budget": budget_arg,
                                                      "scale_chrom_budgets": scale})
            assert [final[c] for c in counts] == gold[f"{key}_final_{budget_arg}_{scale}"].tolist(), key


def test_wild_bootstrap_estimate(oracle):
    gold = np.load(os.path.join(GOLDEN, "wild_bootstrap_vectors.npz"))
    for name in gold["names"]:
        kwargs = json.loads(str(gold[f"{name}_kwargs"][0]))
        observed = gold[f"{name}_observed"] if f"{name}_observed" in gold.files else None
        fraction, details = oracle.estimate_budget_nonnull_fraction_from_wild_bootstrap_null(
            gold[f"{name}_centered"], observed_scores=observed, return_details=True, **kwargs)
        _same(details, json.loads(str(gold[f"{name}_details"][0])), name)
        assert fraction == float(gold[f"{name}_fraction"][0])


@pytest.mark.parametrize("fixture", ["bigwig", "counts_exact_log", "counts_general", "counts_low_memory"])
def test_composed_driver(oracle, fixture):
    """Cache -> pooled budgets -> solve -> combined BED: every number of the reference's cache, the budgets, every
    chromosome's penalty / count / BED text and the combined BED text."""
    gold = np.load(os.path.join(GOLDEN, "composed_vectors.npz"))
    chroms = [str(c) for c in gold[f"{fixture}_chroms"]]
    args = json.loads(str(gold[f"{fixture}_args"][0]))
    inputs = {c: (gold[f"{fixture}_{c}_intervals"], gold[f"{fixture}_{c}_matrix"]) for c in chroms}
    cache, budgets, solved, combined = oracle.run_chromosomes(chroms, inputs, args)
    assert list(cache) == chroms
    for c in chroms:
        entry = cache[c]
        assert np.array_equal(entry["scores"], gold[f"{fixture}_{c}_scores"]), c
        assert [entry["gamma"], entry["budget_count_hat"], entry["budget_fraction_hat"], entry["total_count"],
                entry["num_loci"], budgets[c]] == gold[f"{fixture}_{c}_numbers"].tolist(), c
        _same(entry["budget_rate_meta"], json.loads(str(gold[f"{fixture}_{c}_rate_meta"][0])), c)
        want_gamma_meta = json.loads(str(gold[f"{fixture}_{c}_gamma_meta"][0]))
        assert (entry["gamma_meta"] is None) == (want_gamma_meta is None)
        details, records = solved[c]
        penalty, count, _objective, penalized = gold[f"{fixture}_{c}_solve"]
        assert details["selection_penalty"] == penalty and details["selected_count"] == int(count), c
        assert details["penalized_objective"] == penalized, c
        assert oracle.bed_text(records) == str(gold[f"{fixture}_{c}_bed"][0]), c
    assert oracle.bed_text(combined) == str(gold[f"{fixture}_combined_bed"][0])
