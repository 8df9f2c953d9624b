"""GPU: both budget estimators with the bootstrap multipliers made on the device (`multipliers="device"`,
rocco_amd/csrc/normal.hip) against the reference-written fixtures.  The innovations are NumPy's own stream (bit for bit
but for the last place of tail values), the smoothing is a direct sum where the reference calls SciPy's FFT: the
multipliers agree to ~1e-15 of their scale, so the estimates cannot be the reference's bits.  What must hold: every
discrete outcome -- draws used, adaptive stop, truncation lag, support sizes, strings -- is the reference's, and every
statistic is within 1e-9 relative; how many statistics keep their bits is printed."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DISCRETE = {"num_null_draws", "max_null_draws", "adaptive_stop", "ess_max_lag", "ess_lags_used", "num_loci",
            "negative_support_size", "wild_bandwidth", "wild_process", "null_method"}


def _against(want, details, name):
    assert set(details) == set(want), name
    same, worst = 0, 0.0
    for key, value in want.items():
        if isinstance(value, (str, bool)) or key in DISCRETE:
            assert details[key] == value, (name, key, details[key], value)
            same += 1
            continue
        assert np.isclose(details[key], value, rtol=1e-9, atol=1e-12), (name, key, details[key], value)
        same += int(details[key] == value)
        if value != 0.0:
            worst = max(worst, abs(details[key] - value) / abs(value))
    return same, len(want), worst


def test_count_matrix_estimator_on_the_reference_written_cases(gpu):
    from rocco_amd.budget import estimate_budget_nonnull_fraction_from_wild_bootstrap_null as estimate

    gold = np.load(os.path.join(HERE, "golden", "wild_bootstrap_vectors.npz"))
    report = []
    for name in gold["names"]:
        kwargs = json.loads(str(gold[f"{name}_kwargs"][0]))
        observed = gold[f"{name}_observed"] if f"{name}_observed" in gold.files else None
        fraction, details = estimate(gold[f"{name}_centered"], observed_scores=observed, return_details=True,
                                     multipliers="device", **kwargs)
        want = json.loads(str(gold[f"{name}_details"][0]))
        same, total, worst = _against(want, details, str(name))
        assert np.isclose(fraction, float(gold[f"{name}_fraction"][0]), rtol=1e-9, atol=1e-12), name
        report.append((str(name), same, total, worst, fraction == float(gold[f"{name}_fraction"][0])))
    for row in report:
        print("wild bootstrap, device multipliers: %-28s %2d / %2d entries keep the reference's bits, worst relative "
              "difference %.2e, enriched fraction %s" % (row[0], row[1], row[2], row[3], "equal" if row[4] else "differs"))


def test_score_track_estimator_on_the_reference_written_tracks(gpu):
    from rocco_amd.budget import estimate_budget_nonnull_fraction_from_score_track as estimate

    gold = np.load(os.path.join(HERE, "golden", "budget_vectors.npz"))
    for name in gold["names"]:
        scores = gold[f"{name}_scores"]
        draws, hint = (int(v) for v in gold[f"{name}_params"])
        fraction, details = estimate(scores, dependence_lag_hint=None if hint < 0 else hint, num_null_draws=draws,
                                     return_details=True, multipliers="device")
        want = json.loads(str(gold[f"{name}_details"][0]))
        same, total, worst = _against(want, details, str(name))
        assert np.isclose(fraction, float(gold[f"{name}_fraction"][0]), rtol=1e-9, atol=1e-12), name
        print("score track, device multipliers: %-24s %2d / %2d entries keep the reference's bits, worst relative "
              "difference %.2e" % (str(name), same, total, worst))


def test_environment_switch_and_bad_value(gpu, monkeypatch):
    from rocco_amd import budget

    rng = np.random.default_rng(3)
    scores = rng.normal(0.0, 1.0, 20000)
    host = budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_null_draws=6, return_details=True)
    monkeypatch.setenv("ROCCO_BUDGET_MULTIPLIERS", "device")
    dev = budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_null_draws=6, return_details=True)
    assert np.isclose(dev[0], host[0], rtol=1e-9, atol=1e-12) and dev[1]["num_null_draws"] == host[1]["num_null_draws"]
    # an explicit argument wins over the environment
    again = budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_null_draws=6, return_details=True,
                                                                     multipliers="host")
    assert again[0] == host[0] and again[1] == host[1]
    with pytest.raises(ValueError):
        budget.estimate_budget_nonnull_fraction_from_score_track(scores, num_null_draws=6, multipliers="gpu")


def test_count_branch_budget_estimates_side_by_side_give_the_same_bed(gpu, tmp_path, monkeypatch):
    """The composed driver's count branch with device multipliers runs the chromosomes' budget estimates on worker streams
    side by side (rocco_amd.rocco._estimates_side_by_side); ROCCO_BUDGET_NULL_STREAMS=1 runs them one after another
    as the reference's loop does (rocco/rocco.py:1027-1048): same cache entries, same combined BED."""
    import torch

    from rocco_amd import rocco as impl

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(2718)
    inputs = {}
    for chrom, n in (("chr5", 30011), ("chr11", 8200), ("chr2", 45001), ("chr20", 12000)):
        m = rng.poisson(3.0, size=(6, n)).astype(np.float64)
        for p in range(200, n - 100, 900):
            m[:, p:p + int(rng.integers(5, 40))] += rng.poisson(25.0, size=(6, 1))
        inputs[chrom] = (np.arange(n, dtype=np.int64) * 50, torch.from_numpy(m).cuda())
    args = {"input_track_type": "bam", "budget_null_draws": 12, "threads": -1, "gamma": None, "budget": None,
            "scale_chrom_budgets": 1.0, "budget_posterior_quantile": 0.01, "selection_penalty": None, "min_length_bp": None,
            "score_lower_bound_z": 1.0, "score_prior_df": 5.0, "score_min_effect": None, "score_precision_floor_ratio": 0.01,
            "low_memory": False, "narrowPeak": False, "budget_null_multipliers": "device"}
    beds, caches = {}, {}
    for streams in ("1", "3"):
        monkeypatch.setenv("ROCCO_BUDGET_NULL_STREAMS", streams)
        a = dict(args)
        a["output"] = str(tmp_path / f"out{streams}.bed")
        beds[streams] = open(impl.run_chromosomes(list(inputs), inputs, a, run_id=streams), "rb").read()
        caches[streams] = impl._build_chrom_cache(list(inputs), inputs, dict(args))
    assert beds["1"] == beds["3"] and len(beds["1"]) > 0
    assert list(caches["1"]) == list(caches["3"]) == list(inputs)
    for c in inputs:
        for key in ("budget_count_hat", "budget_fraction_hat", "gamma", "total_count"):
            assert caches["1"][c][key] == caches["3"][c][key], (c, key)
        assert caches["1"][c]["budget_rate_meta"] == caches["3"][c]["budget_rate_meta"], c


def test_inputs_handed_over_give_the_same_bed_and_are_centred_in_place(gpu, tmp_path, monkeypatch):
    """`args["consume_inputs"] = True` (round 5; not in the reference): CUDA count matrices the caller no longer needs are
    centred in place instead of in copies.  Default: the caller's tensors are never written to.  Same combined BED."""
    import torch

    from rocco_amd import rocco as impl

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(99)
    hosts = {}
    for chrom, n in (("chr3", 21011), ("chr8", 9300), ("chr14", 15000)):
        m = rng.poisson(2.0, size=(5, n)).astype(np.float64)
        for p in range(300, n - 100, 1100):
            m[:, p:p + int(rng.integers(5, 40))] += rng.poisson(20.0, size=(5, 1))
        hosts[chrom] = m
    args = {"input_track_type": "bam", "budget_null_draws": 8, "threads": -1, "gamma": None, "budget": None,
            "scale_chrom_budgets": 1.0, "budget_posterior_quantile": 0.01, "selection_penalty": None, "min_length_bp": None,
            "score_lower_bound_z": 1.0, "score_prior_df": 5.0, "score_min_effect": None, "score_precision_floor_ratio": 0.01,
            "low_memory": False, "narrowPeak": False, "budget_null_multipliers": "device"}
    beds = {}
    for consume in (False, True):
        inputs = {c: (np.arange(m.shape[1], dtype=np.int64) * 50, torch.from_numpy(m).cuda()) for c, m in hosts.items()}
        a = dict(args, consume_inputs=consume, output=str(tmp_path / f"out{int(consume)}.bed"))
        beds[consume] = open(impl.run_chromosomes(list(inputs), inputs, a, run_id=str(int(consume))), "rb").read()
        untouched = all(bool(torch.equal(inputs[c][1].cpu(), torch.from_numpy(hosts[c]))) for c in hosts)
        assert untouched == (not consume)
    assert beds[False] == beds[True] and len(beds[True]) > 0


def test_device_multipliers_refuse_a_source_that_has_drawn_ahead(gpu):
    """A `TrackWeightsAhead` made with ahead > 1 has already taken normals from its generator: continuing that generator on
    the device would put every draw somewhere else in the stream than the reference's -- refused, not done silently."""
    import torch

    from rocco_amd import budget

    rng = np.random.default_rng(5)
    track = torch.from_numpy(rng.normal(0.2, 1.0, size=6000)).cuda()
    ahead = budget.TrackWeightsAhead(6000, None, 6, random_seed=0, ahead=3)
    try:
        with pytest.raises(ValueError, match="drawn ahead"):
            budget.estimate_budget_nonnull_fraction_from_score_track(track, num_null_draws=6, multipliers="device", weights_source=ahead)
    finally:
        ahead.close()
    fresh = budget.TrackWeightsAhead(6000, None, 6, random_seed=0, ahead=1)
    got = budget.estimate_budget_nonnull_fraction_from_score_track(track, num_null_draws=6, multipliers="device", weights_source=fresh)
    want = budget.estimate_budget_nonnull_fraction_from_score_track(track, num_null_draws=6, multipliers="device")
    assert got == want


def test_track_branch_budget_estimates_side_by_side_give_the_same_bed(gpu, tmp_path, monkeypatch):
    """The same for score tracks (rocco/rocco.py:994-1008)."""
    from rocco_amd import rocco as impl

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(31415)
    inputs = {}
    for chrom, n in (("chr4", 40003), ("chr9", 9000), ("chr1", 65000), ("chr21", 12001), ("chrX", 20000)):
        m = np.round(rng.gamma(1.0, 0.3, size=(4, n)), 5)
        for p in range(300, n - 100, 800):
            m[:, p:p + int(rng.integers(5, 40))] += rng.gamma(6.0, 0.8, size=(4, 1))
        inputs[chrom] = (np.arange(n, dtype=np.int64) * 50, m)
    args = {"input_track_type": "bigwig", "budget_null_draws": 12, "threads": -1, "gamma": None, "budget": None,
            "scale_chrom_budgets": 1.0, "budget_posterior_quantile": 0.01, "selection_penalty": None, "min_length_bp": None,
            "score_lower_bound_z": 1.0, "score_prior_df": 5.0, "score_precision_floor_ratio": 0.01, "low_memory": False,
            "narrowPeak": False, "budget_null_multipliers": "device"}
    beds, caches = {}, {}
    for streams in ("1", "3"):
        monkeypatch.setenv("ROCCO_BUDGET_NULL_STREAMS", streams)
        a = dict(args)
        a["output"] = str(tmp_path / f"out{streams}.bed")
        beds[streams] = open(impl.run_chromosomes(list(inputs), inputs, a, run_id=streams), "rb").read()
        caches[streams] = impl._build_chrom_cache(list(inputs), inputs, dict(args))
    assert beds["1"] == beds["3"] and len(beds["1"]) > 0
    for c in inputs:
        for key in ("budget_count_hat", "budget_fraction_hat", "gamma", "total_count"):
            assert caches["1"][c][key] == caches["3"][c][key], (c, key)
        assert caches["1"][c]["budget_rate_meta"] == caches["3"][c]["budget_rate_meta"], c
