"""GPU: the count-path scoring glue `score_loci_wls` (rocco/inference.py:302-379).

Everything downstream of the logarithm equals the reference bit for bit, and the logarithm itself is correctly
rounded on the device (tests/test_gpu_log2.py).  So: counts of the form 2^k - 1 (log2(count + 1) exact on any platform)
reproduce the oracle's composition exactly; general counts reproduce it exactly when the oracle is handed the correctly
rounded log matrix; and against NumPy's own log2 -- what the reference calls, not correctly rounded and not the same
on every host -- the log scale is within one ulp, equal in all but a fraction of a per cent of the entries."""
import numpy as np
import pytest

from log2_truth import log2_correctly_rounded

pytestmark = pytest.mark.gpu
TRACKS = ("mean", "raw_variance", "prior_variance", "moderated_variance", "standard_error", "z_scores",
          "degrees_of_freedom", "centered_matrix")


def test_reference_known_answers(gpu):
    """tests/test_rocco.py:235-260 of the reference."""
    from rocco_amd.inference import score_loci_wls

    scores, details = score_loci_wls(np.array([[1.0, 15.0]]), lower_bound_z=0.0, return_details=True)
    assert details["input_scale"] == "log2p1"
    assert "sample_intercepts" not in details and "sample_baselines" not in details
    assert np.allclose(details["mean"], np.array([-1.5, 1.5]))
    assert np.allclose(details["z_scores"], np.array([-0.67449076, 0.67449076]))
    assert np.allclose(scores, np.array([-0.67449076, 0.67449076]))
    scores, details = score_loci_wls(np.array([[1.0, 15.0]]), min_effect=0.5, return_details=True)
    assert np.isclose(details["min_effect"], 0.5)
    assert scores[1] < details["z_scores"][1] and scores[0] < details["z_scores"][0]
    scores, details = score_loci_wls(np.array([[1.0, 3.0, 7.0], [1.2, 2.8, 6.5]]), low_memory=True, return_details=True)
    assert scores.dtype == np.float64 and details["centered_matrix"].dtype == np.float32
    assert np.all(np.isfinite(details["centered_matrix"]))
    assert score_loci_wls(np.array([[1.0, 3.0, 7.0]])).shape == (3,)


@pytest.mark.parametrize("K,n", [(1, 2), (2, 24), (3, 25), (4, 26), (5, 1000), (7, 40001), (3, 600000)])
def test_power_of_two_counts_bit_for_bit(gpu, oracle, K, n):
    from rocco_amd.inference import score_loci_wls

    rng = np.random.default_rng(K * 31 + n)
    peaks = rng.random(n) < 0.05
    k = rng.integers(0, 6, size=(K, n)) + peaks[None, :] * rng.integers(2, 9, size=(K, n))
    counts = np.ldexp(1.0, k) - 1.0
    counts[rng.random(counts.shape) < 0.01] *= -1.0  # negative counts are clipped to zero
    for kw in ({}, {"min_effect": 0.3, "prior_df": 2.0, "precision_floor_ratio": 0.2}):
        got, gd = score_loci_wls(counts, return_details=True, **kw)
        want, wd = oracle.score_loci_wls(counts, **kw)
        assert got.tobytes() == want.tobytes(), (K, n, kw)
        for key in TRACKS:
            assert np.asarray(gd[key]).tobytes() == np.asarray(wd[key]).tobytes(), (K, n, kw, key)
        for key in ("input_scale", "local_baseline_window", "local_baseline_lambda", "min_effect",
                    "precision_floor_ratio", "prior_spatial_window"):
            assert gd[key] == wd[key], key


def test_log_scale_and_pilot_offset(gpu):
    import torch

    from rocco_amd.inference import log_scale_center_rows_device

    rng = np.random.default_rng(2)
    for n in (1, 2, 7, 1000, 100001):
        counts = rng.gamma(0.7, 8.0, size=(4, n))
        counts[0, :: 3] = 0.0
        counts[1] = -counts[1]
        c_t, off_t = log_scale_center_rows_device(torch.from_numpy(counts).cuda())
        log_ref = log2_correctly_rounded(np.clip(counts, 0.0, None) + 1.0)
        med = np.median(log_ref, axis=1)
        assert np.array_equal(off_t.cpu().numpy(), med)
        assert np.array_equal(c_t.cpu().numpy(), log_ref - med[:, None])
        host = np.log2(np.clip(counts, 0.0, None) + 1.0)  # this host's NumPy: one ulp at most, rarely
        assert np.all(np.abs(host - log_ref) <= np.spacing(np.abs(log_ref))) and (host != log_ref).mean() < 2.0e-3
    with pytest.raises(ValueError):
        log_scale_center_rows_device(torch.tensor([[1.0, float("inf")]], dtype=torch.float64).cuda())


@pytest.mark.parametrize("kind", ["fractional", "integer"])
def test_general_counts_bit_for_bit_given_the_correctly_rounded_log(gpu, oracle, kind):
    from rocco_amd.inference import score_loci_wls

    rng = np.random.default_rng(9)
    K, n = 6, 50000
    counts = rng.gamma(0.8, 6.0, size=(K, n)) * (1.0 + 4.0 * (rng.random(n) < 0.03))[None, :]
    if kind == "integer":
        counts = np.floor(counts * 3.0)
    got, gd = score_loci_wls(counts, return_details=True)
    want, wd = oracle.score_loci_wls(counts, log_matrix=log2_correctly_rounded(np.clip(counts, 0.0, None) + 1.0))
    assert got.tobytes() == want.tobytes()
    for key in TRACKS:
        assert np.asarray(gd[key]).tobytes() == np.asarray(wd[key]).tobytes(), key
    # with this host's NumPy log2 in the oracle instead: the same to the last places
    host, hd = oracle.score_loci_wls(counts)
    assert np.abs(got - host).max() <= 1e-9 * np.abs(host).max()


def test_golden_vectors_of_the_reference_function(gpu):
    """tests/golden/score_loci_wls_vectors.npz (the reference's own `score_loci_wls`): bit for bit from the
    log-scaled matrix on, and from the counts wherever log2(count + 1) is exact ("pow2" cases)."""
    import os

    from rocco_amd.inference import score_loci_wls

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "score_loci_wls_vectors.npz"))
    from_counts = 0
    for name in gold["names"]:
        name = str(name)
        lbz, pdf, me, pfr = gold[f"{name}_params"]
        kw = dict(lower_bound_z=lbz, prior_df=pdf, min_effect=None if np.isnan(me) else me, precision_floor_ratio=pfr)
        runs = [(gold[f"{name}_log"], "log2p1")]
        # the log scale NumPy produced on the generating host against the correctly rounded one: one ulp at most --
        # and where they are equal (every one of the 48 cases: this NumPy's log2 first leaves the correctly rounded value
        # at log2(7957)) the reference's tracks must come out of the raw counts bit for bit as well
        cr = log2_correctly_rounded(np.clip(gold[f"{name}_counts"], 0.0, None) + 1.0)
        assert np.all(np.abs(cr - gold[f"{name}_log"]) <= np.spacing(np.abs(cr))), name
        if "_pow2_" in name or np.array_equal(cr, gold[f"{name}_log"]):
            runs.append((gold[f"{name}_counts"], "counts"))
            from_counts += 1
        for matrix, scale in runs:
            scores, details = score_loci_wls(matrix, return_details=True, input_scale=scale, **kw)
            assert scores.tobytes() == gold[f"{name}_scores"].tobytes(), (name, scale)
            for key in TRACKS:
                assert np.asarray(details[key], dtype=np.float64).tobytes() == gold[f"{name}_{key}"].tobytes(), (name, scale, key)
            scalars = np.array([details["local_baseline_window"], details["local_baseline_lambda"], details["min_effect"],
                                details["precision_floor_ratio"], details["prior_spatial_window"]], dtype=np.float64)
            assert np.array_equal(scalars, gold[f"{name}_scalars"]), name
    assert from_counts == 48  # all of them, not only the 16 exact-log cases


def test_count_path_pipeline_end_to_end(gpu, oracle):
    """Counts -> score_loci_wls -> budgeted solve -> BED3 records -> summit offsets, every step in HBM
    (rocco/rocco.py:1009-1018, 890-930, 139-191, 809-872), against the oracle's composition.  The counts are
    2^k - 1 so that log2(count + 1) is exact and everything downstream must agree bit for bit."""
    import torch

    from rocco_amd import pipeline

    rng = np.random.default_rng(21)
    K, n, step, budget, gamma = 6, 120000, 50, 0.03, 1.0
    peaks = np.zeros(n, dtype=bool)
    for c in rng.integers(0, n - 60, size=n // 1500):
        peaks[c:c + int(rng.integers(4, 40))] = True
    k = rng.integers(0, 4, size=(K, n)) + peaks[None, :] * rng.integers(2, 7, size=(K, n))
    counts = np.ldexp(1.0, k) - 1.0
    params = {"lower_bound_z": 1.0, "prior_df": 5.0, "min_effect": None, "precision_floor_ratio": 0.01}
    work = pipeline.ChromWork("chrC", torch.from_numpy(counts).cuda(), budget, gamma, step=step, start=1000,
                              scoring="wls", wls_params=params)
    scores = []
    res = pipeline.solve_rank([work], scores_out=scores)[0]
    o_scores, o_details = oracle.score_loci_wls(counts, **params)
    assert scores[0].cpu().numpy().tobytes() == o_scores.tobytes()
    o_sol, _o_obj, o_det = oracle.solve_chrom_exact(o_scores, budget=budget, gamma=gamma, return_details=True)
    assert res["selection_penalty"] == o_det["selection_penalty"] and res["selected_count"] == o_det["selected_count"]
    assert np.array_equal(res["solution"].cpu().numpy(), o_sol)
    intervals = 1000 + step * np.arange(n, dtype=np.int64)
    want_records = oracle.chrom_solution_records("chrC", intervals, o_sol)
    assert pipeline.runs_to_records(res) == want_records and len(want_records) > 10
    track = oracle.narrowpeak_summit_track(intervals, o_details["mean"])
    want_offsets = oracle.narrowpeak_summit_offsets(want_records, {"chrC": track})
    assert pipeline.summit_offsets(res) == want_offsets
    assert any(off > 0 for _, off in want_offsets)


def _ulps(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float64).view(np.int64)
    a = np.where(a < 0, np.int64(-2 ** 63) - a, a)
    b = np.where(b < 0, np.int64(-2 ** 63) - b, b)
    return np.abs(a - b)


def test_raw_counts_downstream_of_the_one_ulp_log_freedom(gpu, oracle):
    """What the device's correctly rounded log2 changes against the reference's np.log2, measured end to end: raw counts
    -> score_loci_wls -> solve_chrom_exact -> BED3 records, (i) for the 32 reference-written golden cases whose
    log2(count + 1) is not exact, against the reference's own scores, and (ii) for a chromosome-sized Poisson count
    matrix against the oracle's composition with this host's np.log2.  Asserted: the BED records are identical
    everywhere; the scores stay within a few hundred ulps (a last-place change of a log propagates through the baseline
    solve).  The measured numbers go to gpurun_out/a2_divergence.json and from there to INTEGRATION.md section 5."""
    import json
    import os

    from rocco_amd.dp import solve_chrom_exact
    from rocco_amd.inference import score_loci_wls
    from rocco_amd.rocco import chrom_solution_records

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = np.load(os.path.join(root, "tests", "golden", "score_loci_wls_vectors.npz"))
    report = {"golden_cases": [], "numpy": np.__version__}
    for name in (str(v) for v in gold["names"]):
        if "_pow2_" in name:
            continue
        lbz, pdf, me, pfr = gold[f"{name}_params"]
        kw = dict(lower_bound_z=lbz, prior_df=pdf, min_effect=None if np.isnan(me) else me, precision_floor_ratio=pfr)
        counts = gold[f"{name}_counts"]
        want = gold[f"{name}_scores"]
        got = score_loci_wls(counts, **kw)
        n = int(want.shape[0])
        cr = log2_correctly_rounded(np.clip(counts, 0.0, None) + 1.0)
        entry = {"case": name, "entries": int(counts.size), "log_entries_one_ulp_off": int(np.sum(cr != gold[f"{name}_log"])),
                 "scores_differing": int(np.sum(got != want)), "max_score_ulps": int(_ulps(got, want).max())}
        if n >= 8:
            budget, gamma = 0.1, 1.0
            g_sol, _g_obj, g_det = solve_chrom_exact(got, budget=budget, gamma=gamma, return_details=True)
            o_sol, _o_obj, o_det = oracle.solve_chrom_exact(want, budget=budget, gamma=gamma, return_details=True)
            intervals = np.arange(n, dtype=np.int64) * 50
            same_bed = chrom_solution_records("chrG", intervals, g_sol) == oracle.chrom_solution_records("chrG", intervals, o_sol)
            entry.update(penalty_equal=bool(g_det["selection_penalty"] == o_det["selection_penalty"]),
                         penalty_rel_diff=float(abs(g_det["selection_penalty"] - o_det["selection_penalty"])
                                                / max(abs(o_det["selection_penalty"]), 1e-300)),
                         solution_equal=bool(np.array_equal(g_sol, o_sol)), bed_equal=bool(same_bed))
            assert same_bed, name
        assert entry["max_score_ulps"] <= 4096, entry
        report["golden_cases"].append(entry)
    assert len(report["golden_cases"]) == 32

    rng = np.random.default_rng(1234)
    K, n, budget, gamma = 10, 500000, 0.02, 1.0
    counts = rng.poisson(3.0, size=(K, n)).astype(np.float64)
    for p in range(400, n - 100, 1500):
        counts[:, p:p + int(rng.integers(6, 40))] += (rng.random((K, 1)) < 0.8) * rng.poisson(rng.gamma(6.0, 6.0), size=(K, 1))
    got = score_loci_wls(counts)
    host, _hd = oracle.score_loci_wls(counts)  # this host's np.log2, as the reference computes it here
    cr = log2_correctly_rounded(counts + 1.0)
    g_sol, _g_obj, g_det = solve_chrom_exact(got, budget=budget, gamma=gamma, return_details=True)
    o_sol, _o_obj, o_det = oracle.solve_chrom_exact(host, budget=budget, gamma=gamma, return_details=True)
    rel = np.abs(got - host) / np.abs(host).max()
    big = {"shape": [K, n], "log_entries_one_ulp_off": int(np.sum(cr != np.log2(counts + 1.0))),
           "log_entries": int(counts.size), "scores_differing": int(np.sum(got != host)),
           "max_score_ulps": int(_ulps(got, host).max()), "max_score_rel_diff": float(rel.max()),
           "penalty_equal": bool(g_det["selection_penalty"] == o_det["selection_penalty"]),
           "penalty_rel_diff": float(abs(g_det["selection_penalty"] - o_det["selection_penalty"]) / abs(o_det["selection_penalty"])),
           "selected_count_equal": bool(g_det["selected_count"] == o_det["selected_count"]),
           "solution_loci_differing": int(np.sum(g_sol != o_sol))}
    intervals = np.arange(n, dtype=np.int64) * 50
    g_records = chrom_solution_records("chrP", intervals, g_sol)
    big["bed_records"] = len(g_records)
    big["bed_equal"] = bool(g_records == records_of(o_sol, "chrP", 50))
    report["poisson_matrix"] = big
    # (iii) a matrix built to contain the freedom: on this host np.log2 and the correctly rounded log2 first part at
    # log2(7957) (29 integers below 2^22), so the enriched stretches take their counts from exactly those integers - 1
    grid = np.arange(1, 1 << 18, dtype=np.float64)
    off = grid[np.log2(grid) != log2_correctly_rounded(grid)]
    report["host_log2"] = {"integers_checked": int(grid.size), "not_correctly_rounded": int(off.size),
                           "smallest": None if off.size == 0 else float(off[0])}
    if off.size:
        K, n = 6, 60000
        counts = rng.poisson(3.0, size=(K, n)).astype(np.float64)
        for p in range(300, n - 100, 900):
            counts[:, p:p + int(rng.integers(6, 40))] = rng.choice(off - 1.0, size=(K, 1))
        got = score_loci_wls(counts)
        host, _hd = oracle.score_loci_wls(counts)
        g_sol, _g_obj, g_det = solve_chrom_exact(got, budget=budget, gamma=gamma, return_details=True)
        o_sol, _o_obj, o_det = oracle.solve_chrom_exact(host, budget=budget, gamma=gamma, return_details=True)
        iv = np.arange(n, dtype=np.int64) * 50
        stress = {"shape": [K, n], "log_entries_one_ulp_off": int(np.sum(log2_correctly_rounded(counts + 1.0) != np.log2(counts + 1.0))),
                  "scores_differing": int(np.sum(got != host)), "max_score_ulps": int(_ulps(got, host).max()),
                  # (against the scores' own scale: a score that crosses zero has no meaningful relative error of its own)
                  "max_score_diff_over_score_scale": float(np.abs(got - host).max() / np.abs(host).max()),
                  "penalty_equal": bool(g_det["selection_penalty"] == o_det["selection_penalty"]),
                  "penalty_rel_diff": float(abs(g_det["selection_penalty"] - o_det["selection_penalty"]) / abs(o_det["selection_penalty"])),
                  "solution_loci_differing": int(np.sum(g_sol != o_sol)),
                  "bed_equal": bool(chrom_solution_records("chrS", iv, g_sol) == records_of(o_sol, "chrS", 50))}
        report["stress_matrix"] = stress
        assert stress["log_entries_one_ulp_off"] > 0 and stress["max_score_diff_over_score_scale"] <= 1e-9, stress
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "a2_divergence.json"), "w") as handle:
        json.dump(report, handle, indent=1)
    assert big["max_score_rel_diff"] <= 1e-9 and big["bed_equal"], big


def records_of(solution, name, step):
    sel = (np.asarray(solution[:-1]) > 0).astype(np.int8)
    d = np.diff(np.concatenate([[0], sel, [0]]))
    return [(name, int(a) * step, int(b) * step) for a, b in zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1))]


@pytest.mark.parametrize("kind", ["continuous", "integer_ties", "mixed", "one_binade", "constant_row", "huge_cell"])
@pytest.mark.parametrize("n", [1, 2, 7, 64, 4097, 70000, 70001])
def test_row_medians_by_select_and_gathered_cell(gpu, kind, n):
    """Every row's median (rocco/inference.py:330-331) is a radix select: two counting passes, then -- when the median's
    22-bit cell holds at most 4096 values -- the cell is gathered and sorted, else four more passes and one for the upper
    middle value.  Signed continuous rows (settled from the cell), rows of few distinct values (never settled), both in
    one matrix, rows inside one binade, constant rows and rows whose cell is just too large: np.median, bit for bit."""
    import torch
    from rocco_amd.inference import log_scale_center_rows_device

    rng = np.random.default_rng(n + len(kind))
    K = 6
    m = rng.normal(0.0, 3.0, size=(K, n))
    if kind == "integer_ties":
        m = np.round(m)
    elif kind == "mixed":
        m[::2] = np.round(m[::2])
    elif kind == "one_binade":
        m = rng.uniform(1.0, 2.0, size=(K, n)) * rng.choice([-1.0, 1.0], size=(K, 1))
    elif kind == "constant_row":
        m[1] = 2.5
        m[4] = -0.0
    elif kind == "huge_cell":
        m = 1.0 + rng.integers(0, 6000, size=(K, n)) * 2.0 ** -40  # ~6000 distinct values inside one 22-bit cell
    centred, offsets = log_scale_center_rows_device(torch.from_numpy(m).cuda(), apply_log2=False)
    want = np.median(m, axis=1)
    assert np.array_equal(offsets.cpu().numpy(), want), (kind, n)
    assert np.array_equal(centred.cpu().numpy(), m - want[:, None])
