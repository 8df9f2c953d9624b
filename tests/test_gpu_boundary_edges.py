"""GPU: corners of the boundary that round 2 left as limits -- locus starts that do not increase (the reference walks
them locus by locus, rocco/rocco.py:180-190) and autocorrelation lag caps above 1023 (rocco/inference.py:446-517 has no
cap) -- against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["decreasing", "shuffled", "constant"])
def test_records_of_locus_starts_that_do_not_increase(gpu, oracle, kind):
    from rocco_amd import rocco as impl

    rng = np.random.default_rng(4)
    n = 400
    solution = (rng.random(n) < 0.3).astype(np.uint8)
    solution[100:140] = 1
    if kind == "decreasing":
        intervals = (n - np.arange(n)) * 50  # constant step -50: passes the reference's gap check
        check = True
    elif kind == "shuffled":
        intervals = rng.permutation(n) * 50
        check = False
    else:
        intervals = np.full(n, 700)
        check = True
    got = impl.chrom_solution_records("chrZ", intervals, solution, check_gaps_intervals=check, min_length_bp=None)
    want = oracle.chrom_solution_records("chrZ", intervals, solution, check_gaps_intervals=check)
    assert got == want
    got = impl.chrom_solution_records("chrZ", intervals, solution, check_gaps_intervals=check, min_length_bp=60)
    assert got == oracle.chrom_solution_records("chrZ", intervals, solution, check_gaps_intervals=check, min_length_bp=60)


@pytest.mark.parametrize("n,hint", [(9000, 300), (40000, 1000), (3000, 5000)])
def test_effective_sample_size_with_many_lags(gpu, oracle, n, hint):
    from rocco_amd import budget

    rng = np.random.default_rng(n)
    e = rng.normal(size=n + 199)
    x = np.clip(np.convolve(e, np.ones(200) / 14.0, mode="valid"), 0.0, None)  # long memory: the Geyer sum runs far
    lag = budget._resolve_budget_ess_max_lag(n, hint)
    assert lag > 1023
    ess, tau, used = budget._estimate_effective_sample_size(x, lag)
    o_ess, o_tau, o_used = oracle.effective_sample_size(x, lag)
    assert used == o_used and np.isclose(tau, o_tau, rtol=1e-9) and np.isclose(ess, o_ess, rtol=1e-9)
    fraction, details = budget.estimate_budget_nonnull_fraction_from_score_track(x - 0.5, dependence_lag_hint=hint,
                                                                                num_null_draws=5, return_details=True)
    o_fraction, o_details = oracle.estimate_budget_nonnull_fraction_from_score_track(x - 0.5, dependence_lag_hint=hint,
                                                                                     num_null_draws=5, return_details=True)
    assert fraction == o_fraction and details["ess_lags_used"] == o_details["ess_lags_used"]
    assert np.isclose(details["autocorrelation_time"], o_details["autocorrelation_time"], rtol=1e-9)
