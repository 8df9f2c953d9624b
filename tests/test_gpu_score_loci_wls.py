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
    for name in gold["names"]:
        name = str(name)
        lbz, pdf, me, pfr = gold[f"{name}_params"]
        kw = dict(lower_bound_z=lbz, prior_df=pdf, min_effect=None if np.isnan(me) else me, precision_floor_ratio=pfr)
        runs = [(gold[f"{name}_log"], "log2p1")]
        if "_pow2_" in name:
            runs.append((gold[f"{name}_counts"], "counts"))
        # the log scale NumPy produced on the generating host against the correctly rounded one: one ulp at most
        cr = log2_correctly_rounded(np.clip(gold[f"{name}_counts"], 0.0, None) + 1.0)
        assert np.all(np.abs(cr - gold[f"{name}_log"]) <= np.spacing(np.abs(cr))), name
        for matrix, scale in runs:
            scores, details = score_loci_wls(matrix, return_details=True, input_scale=scale, **kw)
            assert scores.tobytes() == gold[f"{name}_scores"].tobytes(), (name, scale)
            for key in TRACKS:
                assert np.asarray(details[key], dtype=np.float64).tobytes() == gold[f"{name}_{key}"].tobytes(), (name, scale, key)
            scalars = np.array([details["local_baseline_window"], details["local_baseline_lambda"], details["min_effect"],
                                details["precision_floor_ratio"], details["prior_spatial_window"]], dtype=np.float64)
            assert np.array_equal(scalars, gold[f"{name}_scalars"]), name


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
