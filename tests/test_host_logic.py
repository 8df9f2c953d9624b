"""The product's host-side search / certification logic (rocco_amd/csrc/search.cpp, the very file
compiled into librocco_hip.so) driven on the CPU by an evaluator backed by the oracle
(tests/host_logic/harness.cpp).  Whatever path it takes, its answer must be the reference's:
solution and count bit-exact, penalty within 1e-9 (bit-exact when no decision was left open)."""
import os

import numpy as np
import pytest

import hostlogic as hl

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_golden_budget_cases(gold):
    for name in gold["bud_names"]:
        scores = gold[f"bud_{name}_scores"]
        budget, gamma = gold[f"bud_{name}_params"]
        target = int(np.floor(len(scores) * float(budget)))
        pen, sol, val, cnt, info = hl.calibrate(scores, float(gamma), target)
        pobj, count, frac, penalty = gold[f"bud_{name}_details"]
        assert np.array_equal(sol, gold[f"bud_{name}_solution"]), (name, info)
        assert cnt == int(count)
        assert abs(pen - penalty) <= 1e-9
        assert abs(val - pobj) <= 1e-9 * max(1.0, abs(pobj))
        assert info["evaluations"] == 62


def test_budget8_known_answer(gold):
    pen, sol, val, cnt, info = hl.calibrate(gold["budget8_scores"], 1.0, 3)
    assert sol.tolist() == [0, 0, 0, 0, 1, 1, 0, 0]
    assert pen == 1.05 and cnt == 2


@pytest.mark.parametrize("seed", range(8))
def test_random_problems_match_oracle(oracle, seed):
    from rocco_amd.synth import hash_matrix, survey_matrix

    rng = np.random.default_rng(seed)
    n = int(rng.choice([500, 5000, 40000]))
    K = int(rng.choice([1, 3, 10]))
    m = survey_matrix(n, K, 500 + seed) if seed % 2 == 0 else hash_matrix(K, n, seed=seed)
    s = np.median(m, axis=0) if K > 1 else m[0]
    budget = float(rng.choice([0.005, 0.02, 0.1]))
    gamma = float(rng.choice([0.5, 1.0, 10.0]))
    target = int(np.floor(n * budget))
    ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), target)
    for depth in (1, 3):
        pen, sol, val, cnt, info = hl.calibrate(s, gamma, target, spec_depth=depth)
        assert np.array_equal(sol, ref[1]), info
        assert cnt == ref[3]
        assert abs(pen - ref[0]) <= 1e-9
    # forcing the exact evaluator gives the reference bit for bit
    pen, sol, val, cnt, info = hl.calibrate(s, gamma, target, force_exact=True)
    assert (pen, val, cnt) == (ref[0], ref[2], ref[3]) and np.array_equal(sol, ref[1])
    assert info["path"] == 2


_DEVICE_SIDE_BEHAVIOURS = [
    {"NOW": "1"},                                                      # a compacted copy already exists: adopted at once
    {"NOW": "1", "SLACK": "1e-5", "TILE": "64"},                       # ... built at a lower penalty, tile borders kept
    {"POINTS": "64", "PILOT": "0.05", "TILE": "8192"},                 # 64 penalties a round, a good pilot
    {"POINTS": "7", "PILOT": "3.0", "PILOT_ROUNDS": "1", "PILOT_POINTS": "64"},  # a pilot that is wrong by 300 %
    {"COMPACT": "0"},                                                  # no compaction at all
]


@pytest.mark.parametrize("behaviour", range(len(_DEVICE_SIDE_BEHAVIOURS)))
def test_search_does_not_depend_on_what_the_device_side_does(oracle, monkeypatch, behaviour):
    """The evaluator of the harness imitates what the HIP evaluator may do behind the search's back -- adopt an
    existing compacted level (at once, built lower than asked, with tile borders kept), ask for few or many
    penalties a round, feed it pilot estimates of any quality -- and the answer must stay the reference's."""
    for key, value in _DEVICE_SIDE_BEHAVIOURS[behaviour].items():
        monkeypatch.setenv("ROCCO_HOSTLOGIC_" + key, value)
    kinds = ("round5", "int", "normal", "offset")
    for it in range(24):
        rng = np.random.default_rng([behaviour, it])
        n = int(rng.choice([33, 1000, 8193, 30000]))
        kind = kinds[it % len(kinds)]
        if kind == "round5":
            s = np.round(rng.gamma(1.0, 0.3, n), 5)
            s[rng.integers(0, n, max(1, n // 50))] += rng.gamma(6.0, 1.0, max(1, n // 50))
        elif kind == "int":
            s = rng.integers(-3, 6, n).astype(float)
        elif kind == "normal":
            s = rng.normal(0.2, 1.0, n)
        else:
            s = 1.0e3 + rng.gamma(1.0, 1.0, n)
        gamma = float(rng.choice([0.5, 1.0, 3.0]))
        target = int(np.floor(n * float(rng.choice([0.01, 0.05, 0.2]))))
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), target)
        pen, sol, val, cnt, info = hl.calibrate(s, gamma, target, spec_depth=1 + it % 3)
        assert pen == ref[0] and cnt == ref[3] and np.array_equal(sol, ref[1]), (kind, n, gamma, target, info)


def test_edge_cases_match_oracle(oracle):
    rng = np.random.default_rng(0)
    s = np.round(rng.gamma(1.0, 0.3, size=200), 5)
    for target in (0, 1, 199, 200, 1000):
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 1.0), target)
        pen, sol, val, cnt, info = hl.calibrate(s, 1.0, target)
        assert np.array_equal(sol, ref[1]) and cnt == ref[3], (target, info)
        assert abs(pen - ref[0]) <= 1e-9
    # degenerate switch cost: the fast path must refuse and the exact evaluator answer
    ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 0.0), 10)
    pen, sol, val, cnt, info = hl.calibrate(s, 0.0, 10)
    assert info["path"] == 2 and (pen, cnt) == (ref[0], ref[3]) and np.array_equal(sol, ref[1])
    # one locus
    ref = oracle.calibrate_selection_penalty(np.array([0.7]), np.zeros(0), 0)
    pen, sol, val, cnt, info = hl.calibrate(np.array([0.7]), 1.0, 0)
    assert np.array_equal(sol, ref[1]) and cnt == ref[3]


@pytest.mark.parametrize("offset,n,gamma,budget", [
    (1.0e3, 40000, 1.0, 0.05), (1.0e3, 3000, 1.0, 0.1), (3.0e4, 40000, 0.5, 0.1), (-1.0e6, 3000, 3.0, 0.05),
    (1.0e9, 100, 1.0, 0.3),
])
def test_scores_far_from_zero(oracle, offset, n, gamma, budget):
    """Scores whose magnitude dwarfs their spread: the reference adds the score BEFORE it subtracts the penalty
    (rocco/_chain_dp.c:120,125,127-128), so its roundings happen at the magnitude of the score, which the rounding
    model has to price whatever the running value is."""
    for seed in range(4):
        rng = np.random.default_rng([seed, n])
        s = offset + rng.gamma(1.0, 1.0, n)
        target = int(np.floor(n * budget))
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), target)
        pen, sol, val, cnt, info = hl.calibrate(s, gamma, target)
        assert pen == ref[0] and cnt == ref[3] and np.array_equal(sol, ref[1]), (seed, info)


def test_single_large_peak_at_the_end(oracle):
    """Target 0 on scores of magnitude 1e6: the penalty that stops selecting the last locus is decided by one
    rounding at that magnitude (a case the compacted copy of the chromosome once got wrong in the harness)."""
    bad = 0
    for it in range(300):
        rng = np.random.default_rng(it)
        n = int(rng.choice([5, 7, 12]))
        s = rng.normal(0, 1e6, n)
        gamma = float(rng.choice([0.5, 1.0, 3.0, 10.0, float(abs(rng.normal()) * 2)]))
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, gamma), 0)
        pen, sol, val, cnt, info = hl.calibrate(s, gamma, 0, spec_depth=1 + it % 3)
        bad += not (pen == ref[0] and cnt == ref[3] and np.array_equal(sol, ref[1]))
    assert bad == 0


def test_integer_scores_many_ties(oracle):
    """Integer data makes exact value ties the rule; whichever path is taken the answer is the
    reference's (count tie-break of rocco/_chain_dp.c:133-179)."""
    rng = np.random.default_rng(4)
    for _ in range(6):
        n = int(rng.integers(30, 3000))
        s = rng.integers(0, 6, size=n).astype(np.float64)
        target = int(n * 0.1)
        ref = oracle.calibrate_selection_penalty(s, oracle.build_switch_costs(s, 1.0), target)
        pen, sol, val, cnt, info = hl.calibrate(s, 1.0, target)
        assert np.array_equal(sol, ref[1]) and cnt == ref[3], info
        assert abs(pen - ref[0]) <= 1e-9


def test_fixed_penalty_matches_oracle(oracle):
    rng = np.random.default_rng(9)
    for _ in range(10):
        n = int(rng.integers(1, 5000))
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        gamma = float(rng.choice([0.5, 1.0, 3.0]))
        lam = float(rng.choice([-1.0, 0.0, 0.31, 2.0, 50.0]))
        sol, val, cnt, info = hl.solve_fixed(s, gamma, lam)
        o_sol, o_val, o_cnt = oracle.solve_penalized_chain(s, oracle.build_switch_costs(s, gamma), lam)
        assert np.array_equal(sol, o_sol) and cnt == o_cnt, (n, gamma, lam, info)
        assert abs(val - o_val) <= 1e-9 * max(1.0, abs(o_val))
    # vector costs
    n = 700
    s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
    c = rng.uniform(0.2, 1.3, size=n - 1)
    sol, val, cnt, info = hl.solve_fixed(s, c, 0.4)
    o_sol, o_val, o_cnt = oracle.solve_penalized_chain(s, c, 0.4)
    assert np.array_equal(sol, o_sol) and cnt == o_cnt


def test_delta_definition_self_consistency(oracle):
    """The sequential delta-form definition agrees with the exact DP wherever it claims certainty:
    with no uncertain locus its fill equals the reference solution."""
    rng = np.random.default_rng(21)
    agree = 0
    for _ in range(30):
        n = int(rng.integers(10, 4000))
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        lam = float(rng.uniform(0.0, 1.0))
        sol, st = oracle.delta_chain(s, 1.0, lam)
        o_sol, _, o_cnt = oracle.solve_penalized_chain(s, oracle.build_switch_costs(s, 1.0), lam)
        if st["uncertain"] == 0 and not st["overflow"]:
            assert np.array_equal(sol, o_sol) and st["count"] == o_cnt
            agree += 1
        else:
            assert abs(st["count"] - o_cnt) <= st["effect"]
    assert agree >= 20


def test_host_logic_under_address_and_undefined_behaviour_sanitizers():
    """The product's search.cpp (1 400 lines of index-heavy host logic with prefetches ahead of the problem being planned) in
    ONE executable with the harness, the oracle's C sources and a driver, all compiled with -fsanitize=address,undefined
    (tests/host_logic/san_driver.cpp): 160 random calibrations + 320 fixed-penalty solves over six kinds of score arrays
    and eight imitated device-side behaviours, each compared with the oracle's sequential calibration.  Exit code 0 and an
    empty sanitizer log (CPU only: the pool has no GPU sanitizers)."""
    import os
    import subprocess

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_logic")
    subprocess.run(["make", "-C", here, "hostlogic_san"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([os.path.join(here, "hostlogic_san"), "160"], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "160 cases, 0 mismatches" in run.stdout
    assert "Sanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr
