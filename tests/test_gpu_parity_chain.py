"""GPU parity of the chain solve through the C ABI (librocco_hip.so) against the CPU oracle.

Bit-exact on solution / count / selection penalty (integer and IEEE-ordered work);
objective_value within 1e-9 relative (BLAS-ordered dot in the reference, rocco/dp.py:33-34).
"""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bruteforce(scores, costs, lam):
    """tests/test_rocco.py:50-70 of the reference, restated."""
    n = len(scores)
    best = None
    for bits in itertools.product([0, 1], repeat=n):
        sol = np.asarray(bits, dtype=np.uint8)
        val = scores @ sol - np.sum(costs * np.abs(np.diff(sol))) - lam * np.sum(sol)
        cnt = int(sol.sum())
        if best is None or val > best[1] or (np.isclose(val, best[1]) and cnt < best[2]):
            best = (sol, float(val), cnt)
    return best


def test_exact_dp_matches_bruteforce(gpu):
    """Reference test_exact_dp_matches_bruteforce (tests/test_rocco.py:398-415): seed 7, n=9,
    non-constant switch costs, four penalties."""
    from rocco_amd import solve_penalized_chain

    rng = np.random.default_rng(7)
    scores = rng.normal(size=9)
    costs = rng.uniform(0.2, 1.3, size=8)
    for lam in (-0.5, 0.0, 0.6, 1.4):
        sol, val, cnt = solve_penalized_chain(scores, costs, lam)
        b_sol, b_val, b_cnt = _bruteforce(scores, costs, lam)
        assert sol.dtype == np.uint8
        assert np.array_equal(sol, b_sol)
        assert np.isclose(val, b_val)
        assert cnt == b_cnt


def test_solve_chrom_exact_respects_budget(gpu, oracle):
    """Reference test_solve_chrom_exact_respects_budget (tests/test_rocco.py:419-437) + its known
    answer: [0,0,0,0,1,1,0,0], objective -3.8, penalty 1.05."""
    from rocco_amd import build_switch_costs, objective_value, solve_chrom_exact

    scores = np.array([0.5, 1.5, 1.4, -0.2, 3.0, 2.8, -0.1, 0.1])
    solution, objective, details = solve_chrom_exact(scores, budget=0.375, gamma=1.0, return_details=True)
    assert solution.dtype == np.uint8
    assert np.sum(solution) <= 3
    assert np.isclose(objective, objective_value(solution, scores, build_switch_costs(scores, gamma=1.0)))
    assert details["selected_fraction"] <= 0.375
    assert solution.tolist() == [0, 0, 0, 0, 1, 1, 0, 0]
    assert np.isclose(objective, -3.8)
    assert details["selection_penalty"] == 1.05
    assert set(details) == {"penalized_objective", "selected_count", "selected_fraction", "selection_penalty"}


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 64, 65, 255, 256, 257, 1000, 4099])
def test_fixed_penalty_random_vs_oracle(gpu, oracle, n):
    from rocco_amd import solve_penalized_chain

    rng = np.random.default_rng(100 + n)
    scores = np.round(rng.gamma(1.0, 0.3, size=n), 5)
    costs = rng.uniform(0.0, 2.0, size=max(n - 1, 0))
    for lam in (-1.0, 0.0, 0.31, 0.9, 3.0):
        sol, val, cnt = solve_penalized_chain(scores, costs, lam)
        o_sol, o_val, o_cnt = oracle.solve_penalized_chain(scores, costs, lam)
        assert np.array_equal(sol, o_sol)
        assert cnt == o_cnt
        assert np.isclose(val, o_val, rtol=1e-12, atol=1e-12)


def test_integer_scores_exact_ties(gpu, oracle):
    """Integer-valued scores and costs make exact value ties common; the count tie-break
    (rocco/_chain_dp.c:133-134, 147-148, 167-168) must be reproduced."""
    from rocco_amd import solve_penalized_chain

    rng = np.random.default_rng(5)
    for _ in range(20):
        n = int(rng.integers(2, 400))
        scores = rng.integers(-3, 4, size=n).astype(np.float64)
        costs = rng.integers(0, 3, size=n - 1).astype(np.float64)
        lam = float(rng.integers(-1, 2))
        sol, val, cnt = solve_penalized_chain(scores, costs, lam)
        o_sol, o_val, o_cnt = oracle.solve_penalized_chain(scores, costs, lam)
        assert np.array_equal(sol, o_sol)
        assert (val, cnt) == (o_val, o_cnt)


@pytest.mark.parametrize("n,budget,gamma", [
    (50, 0.1, 1.0), (777, 0.02, 0.5), (5000, 0.05, 2.0), (20000, 0.02, 1.0), (20000, 0.005, 10.0),
    (1, 0.5, 1.0), (2, 0.5, 1.0), (10, 1.0, 1.0), (10, 0.0, 1.0),
])
def test_budget_solve_vs_oracle(gpu, oracle, n, budget, gamma):
    from rocco_amd import solve_chrom_exact

    rng = np.random.default_rng(n * 7 + int(gamma * 10))
    scores = np.round(rng.gamma(1.0, 0.3, size=n), 5)
    scores[rng.integers(0, n, size=max(1, n // 50))] += rng.gamma(6.0, 1.0, size=max(1, n // 50))
    sol, obj, det = solve_chrom_exact(scores, budget=budget, gamma=gamma, return_details=True)
    o_sol, o_obj, o_det = oracle.solve_chrom_exact(scores, budget=budget, gamma=gamma, return_details=True)
    assert np.array_equal(sol, o_sol)
    assert det["selected_count"] == o_det["selected_count"]
    assert abs(det["selection_penalty"] - o_det["selection_penalty"]) <= 1e-9
    assert np.isclose(obj, o_obj, rtol=1e-9, atol=1e-9)
    assert np.isclose(det["penalized_objective"], o_det["penalized_objective"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("offset,n,gamma,budget", [
    (1.0e3, 40000, 1.0, 0.05), (3.0e4, 300000, 0.5, 0.1), (-1.0e6, 3000, 3.0, 0.05), (1.0e9, 100, 1.0, 0.3),
    (1.0e3, 1200000, 1.0, 0.02),
])
def test_scores_far_from_zero(gpu, oracle, offset, n, gamma, budget):
    """Scores whose magnitude dwarfs their spread (the reference adds the score before it subtracts the penalty,
    rocco/_chain_dp.c:120,125,127-128, so it rounds at the magnitude of the score): bit-exact all the same."""
    from rocco_amd import dp

    rng = np.random.default_rng([n, int(abs(offset)) % 1000])
    s = offset + rng.gamma(1.0, 1.0, n)
    target = int(np.floor(n * budget))
    ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), target)
    got = dp.calibrate_selection_penalty(s, gamma, target)
    assert got[0] == ref[0] and got[3] == ref[3] and np.array_equal(got[1], ref[1])


def test_errors_like_reference(gpu):
    from rocco_amd import solve_chrom_exact, solve_penalized_chain

    with pytest.raises(ValueError):
        solve_penalized_chain(np.zeros((2, 2)), np.zeros(1), 0.0)
    with pytest.raises(ValueError):
        solve_penalized_chain(np.zeros(0), np.zeros(0), 0.0)
    with pytest.raises(ValueError):
        solve_penalized_chain(np.zeros(5), np.zeros(3), 0.0)
    with pytest.raises(ValueError):
        solve_chrom_exact(np.zeros(0), budget=0.1)
