"""GPU path through the C ABI against the committed golden vectors (produced by the reference) and
against the CPU oracle: scoring, solve, decode, BED text; plus size-independent properties at
larger sizes (certified path == forced exact path)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_golden_budget_cases_end_to_end(gpu, gold, tmp_path, monkeypatch):
    from rocco_amd import chrom_solution_to_bed, score_central_tendency_chrom, solve_chrom_exact

    monkeypatch.chdir(tmp_path)
    for name in gold["bud_names"]:
        m = gold[f"bud_{name}_matrix"]
        budget, gamma = gold[f"bud_{name}_params"]
        scores = score_central_tendency_chrom(m, method="quantile", quantile=0.5)
        assert np.array_equal(scores, gold[f"bud_{name}_scores"])
        sol, obj, det = solve_chrom_exact(scores, budget=float(budget), gamma=float(gamma), return_details=True)
        pobj, count, frac, penalty = gold[f"bud_{name}_details"]
        assert sol.dtype == np.uint8 and np.array_equal(sol, gold[f"bud_{name}_solution"]), name
        assert det["selected_count"] == int(count) and det["selected_fraction"] == frac
        assert abs(det["selection_penalty"] - penalty) <= 1e-9
        assert abs(obj - float(gold[f"bud_{name}_objective"])) <= 1e-9 * max(1.0, abs(obj))  # north_star: 1e-4
        assert abs(det["penalized_objective"] - pobj) <= 1e-9 * max(1.0, abs(pobj))
        intervals = np.arange(m.shape[1], dtype=np.int64) * 50
        bed = chrom_solution_to_bed("chrT", intervals, sol, ID=name)
        assert bed == f"rocco_{name}_chrT.bed"
        assert open(bed, "rb").read() == gold[f"bud_{name}_bed"].tobytes()
        bed150 = chrom_solution_to_bed("chrT", intervals, sol, ID=name + "m", min_length_bp=150)
        assert open(bed150, "rb").read() == gold[f"bud_{name}_bed_min150"].tobytes()


def test_golden_fixed_penalty_cases(gpu, gold):
    from rocco_amd import solve_penalized_chain

    for i in range(int(gold["fixed_n"])):
        s, c, lam = gold[f"fixed_{i}_scores"], gold[f"fixed_{i}_costs"], float(gold[f"fixed_{i}_lambda"])
        sol, val, cnt = solve_penalized_chain(s, c, lam)
        assert np.array_equal(sol, gold[f"fixed_{i}_solution"]), i
        assert cnt == int(gold[f"fixed_{i}_count"])
        assert np.isclose(val, float(gold[f"fixed_{i}_value"]), rtol=1e-12, atol=1e-12)


def test_median_expectation_and_validation(gpu, gold):
    from rocco_amd import score_central_tendency_chrom

    assert score_central_tendency_chrom(gold["median2_matrix"]).tolist() == [0.0, 2.5, 1.5, 0.0]
    with pytest.raises(ValueError):
        score_central_tendency_chrom(np.zeros(4))
    with pytest.raises(ValueError):
        score_central_tendency_chrom(np.zeros((3, 4)), method="no-such-method")
    assert score_central_tendency_chrom(np.zeros((3, 4)), method="tmean").tolist() == [0.0] * 4
    assert score_central_tendency_chrom(np.full((3, 4), 3.0), power=2.0).tolist() == [9.0] * 4
    one = np.arange(5.0)[None, :]
    assert np.array_equal(score_central_tendency_chrom(one), one[0])


@pytest.mark.parametrize("K", [2, 3, 5, 10, 33, 100, 101])
def test_quantile_and_mean_branches_equal_numpy(gpu, K):
    """rocco/rocco.py:267-272 (np.quantile(..., method="nearest")) and 298-299 (np.mean(axis=0)): bit for bit."""
    from rocco_amd import score_central_tendency_chrom

    rng = np.random.default_rng(K)
    n = 20011
    m = np.round(rng.gamma(1.0, 0.3, size=(K, n)), 2)  # plenty of ties
    m[rng.integers(0, K), rng.integers(0, n, size=20)] = np.nan
    for q in (0.0, 0.1, 0.25, 0.37, 0.5 + 1e-9, 0.75, 0.9, 0.995, 1.0):
        want = np.quantile(m, q, axis=0, method="nearest")
        got = score_central_tendency_chrom(m, method="quantile", quantile=q)
        assert np.array_equal(got, want, equal_nan=True), (K, q)
    for mat in (m, m.astype(np.float32)):
        want = np.mean(np.asarray(mat, dtype=float), axis=0)
        got = score_central_tendency_chrom(mat, method="mean")
        assert got.tobytes() == want.tobytes(), K


@pytest.mark.parametrize("K", [2, 3, 7, 8, 9, 33, 100, 129, 137, 300])
def test_trimmed_mean_and_power_branches(gpu, K):
    """rocco/rocco.py:273-297 (per column: stats.tmean between the nearest-rank quantiles at tprop and 1 - tprop) bit
    for bit against SciPy itself; `power` (255, 304): 2 is a square as in NumPy, other exponents to the last places."""
    from scipy import stats

    from rocco_amd import score_central_tendency_chrom

    rng = np.random.default_rng(K)
    n = 3001
    m = np.round(rng.gamma(1.0, 0.3, size=(K, n)), 2)  # plenty of ties at the limits
    m[:, 4] = 0.5
    for tprop in (0.05, 0.2, 0.0, 0.5):
        lo = np.quantile(m, tprop, axis=0, method="nearest")
        hi = np.quantile(m, 1.0 - tprop, axis=0, method="nearest")
        want = np.array([stats.tmean(m[:, i], limits=(lo[i], hi[i]), inclusive=(True, True)) for i in range(n)])
        got = score_central_tendency_chrom(m, method="tmean", tprop=tprop)
        assert got.tobytes() == want.tobytes(), (K, tprop)
    med = np.median(m, axis=0)
    assert score_central_tendency_chrom(m, power=2.0).tobytes() == np.power(med, 2.0).tobytes()
    assert np.allclose(score_central_tendency_chrom(m, power=0.5), np.power(med, 0.5), rtol=1e-14, atol=0.0)
    one = np.round(rng.gamma(1.0, 0.3, size=(1, 50)), 3)
    assert score_central_tendency_chrom(one, power=2.0).tobytes() == np.power(one[0], 2.0).tobytes()


@pytest.mark.parametrize("K", [2, 3, 4, 5, 7, 8, 9, 10, 11, 16, 17, 25, 33, 50, 51, 64, 77, 100, 101, 128, 130, 151, 160, 199, 200, 201,
                               202, 229, 255, 256, 257, 258, 299, 300, 301, 400, 555, 600, 601, 1000, 1200, 1201])
def test_median_kernel_all_sizes(gpu, K):
    import torch
    from rocco_amd.rocco import score_central_tendency_chrom_device

    rng = np.random.default_rng(K)
    n = 1537
    m = np.round(rng.gamma(1.0, 0.3, size=(K, n)), 5)
    m[:, 5] = 1.25  # all equal
    m[0, 7] = np.inf
    if K > 2:
        m[1, 9] = np.nan
    if K > 110:  # the second half of the two-half kernel (rows 100 ...): NaN, +-inf, ties
        m[105, 11] = np.nan
        m[103, 13] = -np.inf
        m[100:, 15] = m[0, 15]
    if K > 200:  # the parts-in-LDS kernel (4 x 64 / 3 x 100): ties across parts, a column of few distinct values
        m[:, 17] = np.round(m[:, 17], 1)
        m[:, 19] = np.where(np.arange(K) % 2 == 0, 0.25, 0.75)
        m[64:128, 21] = -np.inf
        m[200, 23] = np.nan
    got = score_central_tendency_chrom_device(torch.from_numpy(m).to(gpu)).cpu().numpy()
    want = np.median(m, axis=0)
    assert np.array_equal(got, want, equal_nan=True)
    m32 = m.astype(np.float32)
    got32 = score_central_tendency_chrom_device(torch.from_numpy(m32).to(gpu)).cpu().numpy()
    assert np.array_equal(got32, np.median(np.asarray(m32, dtype=float), axis=0), equal_nan=True)


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 4095, 4096, 4097, 100003])
def test_decode_runs_vs_python_loop(gpu, oracle, n):
    import torch
    from rocco_amd.rocco import chrom_solution_records

    rng = np.random.default_rng(n)
    intervals = np.arange(n, dtype=np.int64) * 50 + 1000
    for p in (0.0, 0.02, 0.5, 1.0):
        sol = (rng.random(n) < p).astype(np.uint8)
        for min_len in (None, 100):
            got = chrom_solution_records("chrQ", intervals, torch.from_numpy(sol).to(gpu), min_length_bp=min_len)
            want = oracle.chrom_solution_records("chrQ", intervals, sol, min_length_bp=min_len) if n > 1 else []
            assert got == want, (n, p, min_len)
    with pytest.raises(ValueError):
        chrom_solution_records("chrQ", intervals[:-1] if n > 1 else np.arange(2), np.zeros(n, dtype=np.uint8))


def test_objective_value_matches_reference_expression(gpu, oracle):
    from rocco_amd import objective_value

    rng = np.random.default_rng(3)
    for n in (1, 2, 1000, 50001):
        s = rng.normal(size=n)
        z = (rng.random(n) < 0.3).astype(np.uint8)
        assert np.isclose(objective_value(z, s, 0.7), oracle.objective_value(z, s, 0.7), rtol=1e-12, atol=1e-12)
        if n > 1:
            c = rng.uniform(0, 2, size=n - 1)
            assert np.isclose(objective_value(z, s, c), oracle.objective_value(z, s, c), rtol=1e-12, atol=1e-12)


def test_synth_device_matches_host(gpu):
    import torch
    from rocco_amd import synth

    for K, n, seed, dt in ((5, 10001, 3, torch.float64), (3, 4097, 99, torch.float32)):
        d = synth.hash_matrix_device(K, n, seed, dtype=dt).cpu().numpy()
        h = synth.hash_matrix(K, n, seed, dtype=np.float64 if dt == torch.float64 else np.float32)
        assert np.array_equal(d, h)


def test_pipeline_three_chromosomes_vs_oracle(gpu, oracle):
    from rocco_amd import pipeline, synth

    works, hosts = [], []
    for idx, (name, n, K, budget, gamma) in enumerate((("chrA", 60000, 10, 0.02, 1.0), ("chrB", 33333, 3, 0.05, 0.5),
                                                        ("chrC", 8000, 4, 0.01, 2.0))):
        seed = synth.chrom_seed(77, idx)
        works.append(pipeline.ChromWork(name, synth.hash_matrix_device(K, n, seed), budget, gamma, step=50))
        hosts.append(synth.hash_matrix(K, n, seed))
    results = pipeline.solve_rank(works)
    # grouped solves (scoring of a group overlapped with the solve of the one before; score-first ordering) and the
    # ungrouped default return the same things
    for groups, score_first in ((2, 0), (3, 0), (3, 1)):
        old = pipeline.SCORE_FIRST
        pipeline.SCORE_FIRST = score_first
        try:
            again = pipeline.solve_rank(works, groups=groups)
        finally:
            pipeline.SCORE_FIRST = old
        for a, b in zip(results, again):
            assert a["selection_penalty"] == b["selection_penalty"] and a["selected_count"] == b["selected_count"]
            assert np.array_equal(a["solution"].cpu().numpy(), b["solution"].cpu().numpy())
            assert np.array_equal(a["begin"].cpu().numpy(), b["begin"].cpu().numpy())
    for w, m, r in zip(works, hosts, results):
        s = np.median(m, axis=0)
        o_sol, o_obj, o_det = oracle.solve_chrom_exact(s, budget=w.budget, gamma=w.gamma, return_details=True)
        assert np.array_equal(r["solution"].cpu().numpy(), o_sol), w.name
        assert r["selected_count"] == o_det["selected_count"]
        assert abs(r["selection_penalty"] - o_det["selection_penalty"]) <= 1e-9
        want = oracle.chrom_solution_records(w.name, np.arange(w.n, dtype=np.int64) * 50, o_sol)
        assert pipeline.runs_to_records(r) == want


def test_certified_path_equals_forced_exact_path(gpu):
    """Size-independent property at a size the oracle would need seconds for: the certified
    parallel path and the sequential exact kernel must return the same solution and count."""
    import torch
    from rocco_amd import _native, dp, synth
    from rocco_amd.rocco import score_central_tendency_chrom_device

    m_t = synth.hash_matrix_device(10, 934200, seed=2121)
    s_t = score_central_tendency_chrom_device(m_t)
    sol_a, obj_a, det_a = dp.solve_chrom_exact_device(s_t, budget=0.02, gamma=1.0)
    solver = _native.solver_for(s_t.device.index)
    solver.set("force_exact", 1)
    try:
        sol_b, obj_b, det_b = dp.solve_chrom_exact_device(s_t, budget=0.02, gamma=1.0)
    finally:
        solver.set("force_exact", 0)
    assert det_b["_path"] == 2
    assert torch.equal(sol_a, sol_b)
    assert det_a["selected_count"] == det_b["selected_count"]
    assert abs(det_a["selection_penalty"] - det_b["selection_penalty"]) <= 1e-9
    assert abs(obj_a - obj_b) <= 1e-9 * max(1.0, abs(obj_b))
