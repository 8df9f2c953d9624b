"""GPU: the hot path at BASELINE.json's full sizes (hg38 chr1; 50 bp bins with K = 100 and 10 bp bins with K = 50).

At 50 bp the oracle still finishes in seconds (62 chain evaluations of 5 M loci), so the whole budgeted solve is
compared bit for bit.  At 10 bp (24.9 M loci) one oracle evaluation at the calibrated penalty is compared and the
rest is checked through size-independent properties: sorted-column check of the medians on slices, the budget
bracket (count at the returned penalty <= target), penalised value == -objective - penalty * count,
solution -> runs -> solution round trip, idempotence of the fixed-penalty solve, run-to-run determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CHR1_BP = 248956422  # rocco/hg38.sizes:1


def column_medians_by_sort(matrix_t, lo, hi):
    """Independent check of the scoring kernel on a slice: sort each column, take the middle (mean of two)."""
    import torch

    block = torch.sort(matrix_t[:, lo:hi], dim=0).values
    K = block.shape[0]
    if K % 2:
        return block[K // 2]
    return (block[K // 2 - 1] + block[K // 2]) / 2.0  # np.median: mean of the two middle values


def runs_to_solution(begin, end, n):
    sol = np.zeros(n, dtype=np.uint8)
    delta = np.zeros(n + 1, dtype=np.int64)
    np.add.at(delta, begin, 1)
    np.add.at(delta, end, -1)
    sol[:] = np.cumsum(delta[:-1]) > 0
    return sol


def solve(matrix_t, name, budget, gamma, step):
    from rocco_amd import pipeline

    scores = []
    res = pipeline.solve_rank([pipeline.ChromWork(name, matrix_t, budget, gamma, step=step)], scores_out=scores)[0]
    return res, scores[0]


def check_properties(res, scores_t, matrix_t, n, budget, gamma):
    import torch

    from rocco_amd import dp

    # medians: exact order statistics
    for lo in (0, n // 2 - 50000, n - 100000):
        hi = lo + 100000
        assert torch.equal(scores_t[lo:hi], column_medians_by_sort(matrix_t, lo, hi))
    target = int(np.floor(n * budget))
    sol = res["solution"].cpu().numpy()
    count = int(sol.sum())
    assert count == res["selected_count"] <= target
    # penalised value == -objective - penalty * count (rocco/dp.py:216-227), objective recomputed on the device
    objective = dp.objective_value(res["solution"], scores_t, gamma)
    assert abs(res["penalized_objective"] - (-objective - res["selection_penalty"] * count)) <= 1e-9 * max(1.0, abs(objective))
    # runs -> solution round trip (the last locus is never emitted, rocco/rocco.py:180)
    begin, end = res["begin"].cpu().numpy(), res["end"].cpu().numpy()
    assert np.all(begin[1:] > end[:-1]) and np.all(end > begin)
    assert np.array_equal(runs_to_solution(begin, end, n)[: n - 1], sol[: n - 1])
    # idempotence: the fixed-penalty solve at the calibrated penalty returns the calibrated solution
    again, _value, cnt, _path = dp.solve_penalized_chain_device(scores_t, gamma, res["selection_penalty"])
    assert cnt == count and torch.equal(again, res["solution"])
    return sol, count


def test_chr1_50bp_k100_against_the_oracle(gpu, oracle):
    from rocco_amd import synth

    n = -(-CHR1_BP // 50)
    assert n == 4979129
    K, budget, gamma = 100, 0.02, 1.0
    matrix_t = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, 0))
    res, scores_t = solve(matrix_t, "chr1", budget, gamma, 50)
    sol, count = check_properties(res, scores_t, matrix_t, n, budget, gamma)
    # the whole budgeted solve on the CPU oracle (2 + 60 chain evaluations), same scores
    o_sol, o_obj, o_det = oracle.solve_chrom_exact(scores_t.cpu().numpy(), budget=budget, gamma=gamma, return_details=True)
    assert o_det["selection_penalty"] == res["selection_penalty"]
    assert o_det["selected_count"] == count
    assert np.array_equal(o_sol, sol)
    assert abs(o_det["penalized_objective"] - res["penalized_objective"]) <= 1e-9 * abs(o_det["penalized_objective"])
    # determinism
    res2, _ = solve(matrix_t, "chr1", budget, gamma, 50)
    assert res2["selection_penalty"] == res["selection_penalty"] and np.array_equal(res2["solution"].cpu().numpy(), sol)


def test_chr1_10bp_k50_properties_and_one_oracle_evaluation(gpu, oracle):
    from rocco_amd import synth

    n = -(-CHR1_BP // 10)
    assert n == 24895643
    K, budget, gamma = 50, 0.02, 1.0
    matrix_t = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, 0))
    res, scores_t = solve(matrix_t, "chr1", budget, gamma, 10)
    sol, count = check_properties(res, scores_t, matrix_t, n, budget, gamma)
    del matrix_t
    scores = scores_t.cpu().numpy()
    costs = oracle.build_switch_costs(scores, gamma)
    o_sol, o_value, o_count = oracle.solve_penalized_chain(scores, costs, res["selection_penalty"])
    assert o_count == count and np.array_equal(o_sol, sol)
    assert abs(o_value - res["penalized_objective"]) <= 1e-9 * max(1.0, abs(o_value))


def test_whole_genome_50bp_k100_every_chromosome_against_the_oracle(gpu, oracle):
    """BASELINE.json's headline configuration (hg38, chr1-22 + X + Y, 50 bp bins, K = 100: 61.8 M loci, 49 GB of
    signal in HBM), all chromosomes in the same device passes as in bench.py; every chromosome's calibrated
    penalty, count, solution and BED3 records against the CPU oracle run on the same scores."""
    from rocco_amd import pipeline, synth

    K, budget, gamma, step = 100, 0.02, 1.0, 50
    genome = synth.chrom_loci(step)
    assert len(genome) == 24 and sum(n for _, n in genome) == 61765409
    works = [pipeline.ChromWork(name, synth.hash_matrix_device(K, n, synth.chrom_seed(20240, idx)), budget, gamma, step=step)
             for idx, (name, n) in enumerate(genome)]
    scores = []
    results = pipeline.solve_rank(works, scores_out=scores)
    del works
    for (name, n), res, s_t in zip(genome, results, scores):
        s_h = s_t.cpu().numpy()
        o_sol, _o_obj, o_det = oracle.solve_chrom_exact(s_h, budget=budget, gamma=gamma, return_details=True)
        assert res["selection_penalty"] == o_det["selection_penalty"], name
        assert res["selected_count"] == o_det["selected_count"] <= int(np.floor(n * budget)), name
        assert np.array_equal(res["solution"].cpu().numpy(), o_sol), name
        assert abs(res["penalized_objective"] - o_det["penalized_objective"]) <= 1e-9 * abs(o_det["penalized_objective"]), name
        want = oracle.chrom_solution_records(name, np.arange(n, dtype=np.int64) * step, o_sol)
        assert pipeline.runs_to_records(res) == want, name
