"""GPU: BASELINE.json's configurations at their stated sizes, and the drop-in entry points that had no GPU test.

config 1  chr21, K = 3, 50 bp (n = 934 200)             whole budgeted solve vs the oracle
config 2  chr1, K = 10, 50 bp (n = 4 979 129)            scores vs np.median, whole budgeted solve vs the oracle
config 3  all autosomes, K = 10, 50 bp (57.5 M loci)     every chromosome's budgeted solve vs the oracle
config 5  whole genome, K = 50, 10 bp (308.8 M loci, 123.5 GB of signal in HBM): medians on sampled columns, every
          chromosome's budgeted solve vs the oracle's whole calibration (penalty bit for bit), runs <-> solution
(config 4, the headline, is tests/test_gpu_full_size.py.)  The oracle runs in spawned worker processes."""
import numpy as np
import pytest

import oracle_pool

pytestmark = pytest.mark.gpu


def records_numpy(name, n, step, solution):
    """The merged BED3 records of rocco/rocco.py:139-191 for contiguous loci, vectorised (the oracle's restatement is
    the reference's per-locus Python loop: seconds per 5 M loci): selected loci among 0..n-2, touching ones merged."""
    sel = (np.asarray(solution[: n - 1]) > 0).astype(np.int8)
    d = np.diff(np.concatenate([[0], sel, [0]]))
    return [(name, int(a) * step, int(b) * step) for a, b in zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1))]


def _solve_one(name, idx, K, step, budget=0.02, gamma=1.0):
    from rocco_amd import pipeline, synth

    n = dict(synth.chrom_loci(step))[name]
    m_t = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, idx))
    scores = []
    res = pipeline.solve_rank([pipeline.ChromWork(name, m_t, budget, gamma, step=step)], scores_out=scores)[0]
    return m_t, res, scores[0], n


def _check_against_budgeted_oracle(oracle, res, s_h, n, budget, gamma, name, step):
    from rocco_amd import pipeline

    o_sol, _o_obj, o_det = oracle.solve_chrom_exact(s_h, budget=budget, gamma=gamma, return_details=True)
    assert res["selection_penalty"] == o_det["selection_penalty"], name
    assert res["selected_count"] == o_det["selected_count"] <= int(np.floor(n * budget)), name
    assert np.array_equal(res["solution"].cpu().numpy(), o_sol), name
    assert abs(res["penalized_objective"] - o_det["penalized_objective"]) <= 1e-9 * abs(o_det["penalized_objective"]), name
    assert pipeline.runs_to_records(res) == oracle.chrom_solution_records(name, np.arange(n, dtype=np.int64) * step, o_sol), name


def test_config1_chr21_k3(gpu, oracle):
    m_t, res, s_t, n = _solve_one("chr21", 20, 3, 50)
    assert n == 934200
    s_h = s_t.cpu().numpy()
    assert np.array_equal(s_h, np.median(m_t.cpu().numpy(), axis=0))
    _check_against_budgeted_oracle(oracle, res, s_h, n, 0.02, 1.0, "chr21", 50)


def test_config2_chr1_k10(gpu, oracle):
    m_t, res, s_t, n = _solve_one("chr1", 0, 10, 50)
    assert n == 4979129
    s_h = s_t.cpu().numpy()
    assert np.array_equal(s_h, np.median(m_t.cpu().numpy(), axis=0))
    _check_against_budgeted_oracle(oracle, res, s_h, n, 0.02, 1.0, "chr1", 50)


def test_config3_autosomes_k10_every_chromosome(gpu, oracle):
    from rocco_amd import pipeline, synth

    K, budget, gamma, step = 10, 0.02, 1.0, 50
    genome = [(i, name, n) for i, (name, n) in enumerate(synth.chrom_loci(step)) if name not in ("chrX", "chrY")]
    assert len(genome) == 22 and sum(n for _, _, n in genome) == 57500042
    works = [pipeline.ChromWork(name, synth.hash_matrix_device(K, n, synth.chrom_seed(20240, i)), budget, gamma, step=step)
             for i, name, n in genome]
    scores = []
    results = pipeline.solve_rank(works, scores_out=scores)
    for w, s_t in list(zip(works, scores))[-2:]:  # np.median on the two shortest (the scoring kernel is the same for all)
        assert np.array_equal(s_t.cpu().numpy(), np.median(w.matrix_t.cpu().numpy(), axis=0)), w.name
    del works
    host_scores = [s.cpu().numpy() for s in scores]
    expected = oracle_pool.run([(s, budget, gamma) for s in host_scores], "budgeted")
    for (i, name, n), res, ((pen, cnt, pobj), o_sol) in zip(genome, results, expected):
        assert res["selection_penalty"] == pen and res["selected_count"] == cnt <= int(np.floor(n * budget)), name
        assert np.array_equal(res["solution"].cpu().numpy(), o_sol), name
        assert abs(res["penalized_objective"] - pobj) <= 1e-9 * abs(pobj), name
        assert pipeline.runs_to_records(res) == records_numpy(name, n, step, o_sol), name
    # the vectorised record builder against the oracle's loop on the shortest chromosome
    (i, name, n), (_r, o_sol) = genome[-1], expected[-1]
    assert records_numpy(name, n, step, o_sol) == oracle.chrom_solution_records(name, np.arange(n, dtype=np.int64) * step, o_sol)


def test_config5_whole_genome_k50_10bp(gpu, oracle):
    import torch

    from rocco_amd import pipeline, synth

    K, budget, gamma, step = 50, 0.02, 1.0, 10
    genome = synth.chrom_loci(step)
    assert len(genome) == 24 and sum(n for _, n in genome) == 308826993
    free, _total = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs 150 GB of free HBM")
    works = [pipeline.ChromWork(name, synth.hash_matrix_device(K, n, synth.chrom_seed(20240, i)), budget, gamma, step=step)
             for i, (name, n) in enumerate(genome)]
    scores = []
    results = pipeline.solve_rank(works, scores_out=scores)
    rng = np.random.default_rng(5)
    for w, s_t in zip(works, scores):  # medians: 20 000 sampled columns of every chromosome against np.median
        cols = torch.from_numpy(np.sort(rng.choice(w.n, size=20000, replace=False))).to(gpu)
        assert np.array_equal(s_t[cols].cpu().numpy(), np.median(w.matrix_t[:, cols].cpu().numpy(), axis=0)), w.name
    del works
    torch.cuda.empty_cache()
    host_scores = [s.cpu().numpy() for s in scores]
    # the reference's whole calibration (2 bracket + 60 bisection evaluations, rocco/dp.py:89-164) per chromosome on the
    # host cores (~70 core-seconds in all): the penalty must be the reference's bit for bit, as in configs 1-4
    expected = oracle_pool.run([(s, budget, gamma) for s in host_scores], "budgeted")
    for (name, n), res, s_h, ((o_penalty, o_count, o_value), o_sol) in zip(genome, results, host_scores, expected):
        sol = res["solution"].cpu().numpy()
        assert res["selection_penalty"] == o_penalty, (name, res["selection_penalty"], o_penalty)
        assert res["selected_count"] == o_count == int(sol.sum()) <= int(np.floor(n * budget)), name
        assert np.array_equal(sol, o_sol), name
        assert abs(o_value - res["penalized_objective"]) <= 1e-9 * max(1.0, abs(o_value)), name
        begin, end = res["begin"].cpu().numpy(), res["end"].cpu().numpy()
        marks = np.zeros(n + 1, dtype=np.int32)
        np.add.at(marks, begin, 1)
        np.add.at(marks, end, -1)
        assert np.array_equal((np.cumsum(marks[:-1]) > 0)[: n - 1], sol[: n - 1] > 0), name
        assert res["path"] in (1, 4), name  # certified or exact spine: never the sequential last resort


def test_config5_count_matrices_scored_on_one_gpu(gpu):
    """BASELINE config 5 as count matrices: K = 50, 10 bp bins, 308.8 M loci = 123.5 GB of counts -- the count-path scoring
    (rocco/inference.py:302-379 for every chromosome) in ONE call on one GPU: the matrices are centred in place and every
    pipeline walks its chromosomes in chunks that the free memory allows (rocco_amd.inference.score_loci_wls_batch_device).
    Every track finite; the chunked call's tracks for three chromosomes bit for bit the single-matrix call's."""
    import torch

    from rocco_amd import inference, synth

    K, step = 50, 10
    genome = synth.chrom_loci(step)
    free, _total = torch.cuda.mem_get_info()
    if free < 250e9:
        pytest.skip("needs 250 GB of free HBM")
    mats = []
    for i, (_name, n) in enumerate(genome):
        m = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, i))
        m.mul_(20.0).round_()
        mats.append(m)
    check = [len(genome) - 1, len(genome) - 3, 12]  # chr21, chrY and one of middle length: kept to compare
    kept = {i: mats[i].clone() for i in check}
    out = inference.score_loci_wls_batch_device(mats, overwrite_input=True)
    assert inference.last_batch_growths_in_flight == 0
    for (scores, details), (_name, n) in zip(out, genome):
        assert tuple(scores.shape) == (n,) and bool(torch.isfinite(scores).all())
        assert details["centered_matrix"].shape == (K, n)
    tracks = {i: (out[i][0].clone(), out[i][1]["standard_error"].clone(), out[i][1]["centered_matrix"][:, :4096].clone()) for i in check}
    del out, mats
    inference.release_batch_workers()
    torch.cuda.empty_cache()
    for i in check:
        scores, details = inference.score_loci_wls_device(kept[i], overwrite_input=True)
        assert torch.equal(scores, tracks[i][0]) and torch.equal(details["standard_error"], tracks[i][1])
        assert torch.equal(details["centered_matrix"][:, :4096], tracks[i][2])


# ---- entry points ------------------------------------------------------------------------------------------

def _cache(rng, sizes, gamma=1.0):
    cache = {}
    for k, n in enumerate(sizes):
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        for p in rng.integers(0, n, size=max(1, n // 300)):
            s[p:p + int(rng.integers(3, 25))] += rng.gamma(5.0, 1.0)
        cache[f"chr{k + 1}"] = {"scores": s, "intervals": np.arange(n, dtype=np.int64) * 50 + 1000, "gamma": gamma}
    return cache


def test_solve_cached_chromosomes_matches_the_oracle(gpu, oracle, tmp_path, monkeypatch):
    from rocco_amd.rocco import solve_cached_chromosomes

    rng = np.random.default_rng(11)
    cache = _cache(rng, [5000, 40000, 9000])
    budgets = {"chr1": 0.03, "chr2": 0.01, "chr3": 0.08}
    got = solve_cached_chromosomes(cache, budgets, write_files=False)
    assert [g[0] for g in got] == list(cache)
    for chrom, objective, details, records in got:
        s = cache[chrom]["scores"]
        o_sol, o_obj, o_det = oracle.solve_chrom_exact(s, budget=budgets[chrom], gamma=1.0, return_details=True)
        assert details["selection_penalty"] == o_det["selection_penalty"] and details["selected_count"] == o_det["selected_count"]
        assert set(details) == {"penalized_objective", "selected_count", "selected_fraction", "selection_penalty"}
        assert abs(objective - o_obj) <= 1e-9 * max(1.0, abs(o_obj))
        assert records == oracle.chrom_solution_records(chrom, cache[chrom]["intervals"], o_sol)
    # files, with a minimum length; and a fixed selection penalty for every chromosome (rocco/rocco.py:915-921)
    monkeypatch.chdir(tmp_path)
    files = solve_cached_chromosomes(cache, budgets, selection_penalty=0.9, min_length_bp=150, run_id="t", write_files=True)
    for chrom, _objective, details, path in files:
        s = cache[chrom]["scores"]
        o_sol, _v, o_cnt = oracle.solve_penalized_chain(s, oracle.build_switch_costs(s, 1.0), 0.9)
        assert details["selection_penalty"] == 0.9 and details["selected_count"] == o_cnt
        want = oracle.chrom_solution_records(chrom, cache[chrom]["intervals"], o_sol, min_length_bp=150)
        with open(path) as fh:
            assert fh.read() == "".join(f"{c}\t{a}\t{b}\n" for c, a, b in want)


@pytest.mark.parametrize("field,value,text", [
    ("scores", np.array([0.1, np.nan, 0.3]), "scores contain non-finite values"),
    ("budget", "a lot", "budget could not be read as a finite number"),
    ("gamma", None, "gamma could not be read as a finite number"),
    ("budget", -0.1, "budget must be finite and non-negative"),
    ("gamma", float("inf"), "gamma must be finite and non-negative"),
])
def test_solve_cached_chromosomes_rejects_what_the_reference_rejects(gpu, field, value, text):
    """rocco/rocco.py:897-914, in the reference's order (scores, then budget, then gamma)."""
    from rocco_amd.rocco import solve_cached_chromosomes

    cache = {"chrT": {"scores": np.array([0.1, 0.2, 0.3]), "intervals": np.array([0, 50, 100]), "gamma": 1.0}}
    budgets = {"chrT": 0.5}
    if field == "budget":
        budgets["chrT"] = value
    else:
        cache["chrT"][field] = value
    with pytest.raises(ValueError, match=text):
        solve_cached_chromosomes(cache, budgets, write_files=False)


@pytest.mark.parametrize("n", [2, 33, 1000, 8193, 70000])
def test_calibrate_selection_penalty_with_a_cost_vector(gpu, oracle, n):
    """The public signature takes any switch-cost vector (rocco/dp.py:89-94; the reference's own brute-force test
    uses non-constant costs, tests/test_rocco.py:398-415)."""
    from rocco_amd import calibrate_selection_penalty

    rng = np.random.default_rng(n)
    s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
    s[rng.integers(0, n, size=max(1, n // 200))] += 4.0
    costs = rng.uniform(0.2, 1.7, size=n - 1)
    for target in sorted({0, 1, max(1, n // 50), max(1, n // 5)}):
        g = calibrate_selection_penalty(s, costs, target)
        o = oracle.calibrate_selection_penalty(s, costs, target)
        assert g[0] == o[0] and g[3] == o[3] and np.array_equal(g[1], o[1]), (n, target)
        assert abs(g[2] - o[2]) <= 1e-9 * max(1.0, abs(o[2]))


def test_random_solves_fixed_seeds(gpu, oracle):
    """A bounded slice of tests/tools/fuzz_parity.py: 150 random budgeted / fixed-penalty solves (eight score
    distributions, scalar and vector costs, sizes around the chunk and tile sizes) against the oracle, bit for bit."""
    from rocco_amd import dp

    bad = []
    for it in range(150):
        rng = np.random.default_rng(777000 + it)
        n = int(rng.choice([1, 2, 3, 5, 31, 32, 33, 100, 1000, 8191, 8192, 8193, 20000, 70000, 300000]))
        kind = rng.choice(["normal", "int", "round5", "heavy", "const", "sparse", "tiny", "huge"])
        s = {"normal": lambda: rng.normal(0.2, 1.0, n), "int": lambda: rng.integers(-3, 6, n).astype(float),
             "round5": lambda: np.round(rng.gamma(1.0, 0.3, n), 5), "heavy": lambda: rng.standard_cauchy(n),
             "const": lambda: np.full(n, float(rng.normal())),
             "sparse": lambda: np.where(rng.random(n) < 0.02, rng.gamma(6.0, 1.0, n), 0.0),
             "tiny": lambda: rng.normal(0, 1e-9, n), "huge": lambda: rng.normal(0, 1e6, n)}[kind]()
        gamma = float(rng.choice([0.0, 0.5, 1.0, 3.0, 10.0]))
        use_vec = n > 1 and rng.random() < 0.25
        costs = rng.gamma(1.0, gamma + 0.1, n - 1) if use_vec else gamma
        o_costs = costs if use_vec else oracle.build_switch_costs(s, gamma)
        if rng.random() < 0.6:
            target = int(np.floor(n * float(rng.choice([0.005, 0.02, 0.1, 0.3]))))
            g = dp.calibrate_selection_penalty(s, costs, target)
            o = oracle.calibrate_selection_penalty(s, o_costs, target)
            ok = g[0] == o[0] and np.array_equal(g[1], o[1]) and g[3] == o[3] and abs(g[2] - o[2]) <= 1e-9 * max(1.0, abs(o[2]))
        else:
            lam = float(rng.choice([0.0, float(np.median(s)), float(rng.normal()), float(np.max(s)) + 1.0]))
            g = dp.solve_penalized_chain(s, costs, lam)
            o = oracle.solve_penalized_chain(s, o_costs, lam)
            ok = np.array_equal(g[0], o[0]) and g[2] == o[2] and abs(g[1] - o[1]) <= 1e-9 * max(1.0, abs(o[1]))
        if not ok:
            bad.append((it, n, kind, gamma, use_vec))
    assert not bad, bad
