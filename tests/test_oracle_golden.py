"""The CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py, run in the build container with /root/reference mounted), and --
when the reference is mounted -- directly against the reference.  Bit-exact on solutions, counts,
values and penalties (same IEEE operations in the same order); BED text byte-identical."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_bruteforce_vectors(oracle, gold):
    """Reference tests/test_rocco.py:398-415 inputs; expected = what the reference returned."""
    for k in range(4):
        sol, val, cnt = oracle.solve_penalized_chain(gold[f"brute_{k}_scores"], gold[f"brute_{k}_costs"],
                                                     float(gold[f"brute_{k}_lambda"]))
        assert np.array_equal(sol, gold[f"brute_{k}_solution"])
        assert val == float(gold[f"brute_{k}_value"])
        assert cnt == int(gold[f"brute_{k}_count"])


def test_budget_vector(oracle, gold):
    """Reference tests/test_rocco.py:419-437: [0,0,0,0,1,1,0,0], objective -3.8, penalty 1.05."""
    sol, obj, det = oracle.solve_chrom_exact(gold["budget8_scores"], budget=0.375, gamma=1.0, return_details=True)
    assert sol.tolist() == [0, 0, 0, 0, 1, 1, 0, 0] == gold["budget8_solution"].tolist()
    assert obj == float(gold["budget8_objective"])
    want = gold["budget8_details"]
    assert [det["penalized_objective"], det["selected_count"], det["selected_fraction"],
            det["selection_penalty"]] == want.tolist()
    assert det["selection_penalty"] == 1.05


def test_fixed_penalty_vectors(oracle, gold):
    for i in range(int(gold["fixed_n"])):
        sol, val, cnt = oracle.solve_penalized_chain(gold[f"fixed_{i}_scores"], gold[f"fixed_{i}_costs"],
                                                     float(gold[f"fixed_{i}_lambda"]))
        assert np.array_equal(sol, gold[f"fixed_{i}_solution"]), i
        assert val == float(gold[f"fixed_{i}_value"]) and cnt == int(gold[f"fixed_{i}_count"])


def test_budgeted_matrices_and_bed_text(oracle, gold):
    for name in gold["bud_names"]:
        m = gold[f"bud_{name}_matrix"]
        budget, gamma = gold[f"bud_{name}_params"]
        scores = oracle.score_central_tendency_chrom(m)
        assert np.array_equal(scores, gold[f"bud_{name}_scores"])
        assert np.array_equal(scores, np.median(m, axis=0)) or m.shape[0] == 1
        sol, obj, det = oracle.solve_chrom_exact(scores, budget=float(budget), gamma=float(gamma), return_details=True)
        assert np.array_equal(sol, gold[f"bud_{name}_solution"]), name
        assert obj == float(gold[f"bud_{name}_objective"])
        assert [det["penalized_objective"], det["selected_count"], det["selected_fraction"],
                det["selection_penalty"]] == gold[f"bud_{name}_details"].tolist()
        intervals = np.arange(m.shape[1], dtype=np.int64) * 50
        text = oracle.bed_text(oracle.chrom_solution_records("chrT", intervals, sol))
        assert text.encode() == gold[f"bud_{name}_bed"].tobytes()
        text150 = oracle.bed_text(oracle.chrom_solution_records("chrT", intervals, sol, min_length_bp=150))
        assert text150.encode() == gold[f"bud_{name}_bed_min150"].tobytes()


def test_median_expectation(oracle, gold):
    """Reference tests/test_rocco.py:838-897 expects [0, 2.5, 1.5, 0] for the two-row matrix."""
    got = oracle.score_central_tendency_chrom(gold["median2_matrix"])
    assert got.tolist() == [0.0, 2.5, 1.5, 0.0] == gold["median2_scores"].tolist()


def test_combine_text(oracle, gold):
    per_chrom = [[(c, 100, 200), (c, 200, 260), (c, 500, 650), (c, 640, 700)] for c in ("chr2", "chr10", "chr1")]
    text = oracle.bed_text(oracle.combine_records(per_chrom))
    assert text.encode() == gold["combine_text"].tobytes()
    assert text.splitlines()[0].startswith("chr1\t") and text.splitlines()[2].startswith("chr10\t")


def test_median_nan_and_f32(oracle):
    m = np.array([[1.0, np.nan, 3.0], [2.0, 5.0, 1.0], [9.0, 4.0, 2.0]])
    got = oracle.score_central_tendency_chrom(m)
    want = np.median(m, axis=0)
    assert np.array_equal(got, want, equal_nan=True)
    rng = np.random.default_rng(0)
    m32 = rng.random((7, 50)).astype(np.float32)
    assert np.array_equal(oracle.score_central_tendency_chrom(m32), np.median(np.asarray(m32, dtype=float), axis=0))


def test_oracle_against_mounted_reference(oracle):
    """Direct cross-check when /root/reference is mounted (build container only)."""
    import ref_loader

    if not ref_loader.reference_available():
        pytest.skip("reference not mounted")
    dp = ref_loader.load_reference_dp()
    rng = np.random.default_rng(11)
    for _ in range(12):
        n = int(rng.integers(1, 4000))
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        c = rng.uniform(0, 2, size=max(n - 1, 0))
        lam = float(rng.uniform(-1, 2))
        a = dp.solve_penalized_chain(s, c, lam)
        b = oracle.solve_penalized_chain(s, c, lam)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]
    for _ in range(6):
        n = int(rng.integers(50, 20000))
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        budget = float(rng.uniform(0.005, 0.1))
        gamma = float(rng.choice([0.5, 1.0, 2.0, 0.73]))
        a = dp.solve_chrom_exact(s, budget=budget, gamma=gamma, return_details=True)
        b = oracle.solve_chrom_exact(s, budget=budget, gamma=gamma, return_details=True)
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2]
