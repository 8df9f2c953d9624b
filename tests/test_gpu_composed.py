"""GPU: the composed driver -- rocco_amd.rocco._build_chrom_cache -> _resolve_budgets -> _solve_cached_chromosomes ->
combine_chrom_results (the device form of rocco/rocco.py:933-1196, 194-240) -- against what the REFERENCE's own
composition wrote for the same in-memory matrices (tests/golden/make_golden_composed.py, part 2: its
`generate_chrom_matrix` replaced by the matrices, exactly as its own tests replace it), and against the caches the
reference's builder made of stand-in callables (tests/golden/make_golden_seam.py: the seam its own tests use,
tests/test_rocco.py:566-689, 838-897).

What must be equal: the scores, the switch costs, every statistic of the budget estimate that does not come from the
autocorrelation time, every chromosome's BED bytes and the combined BED bytes.  The autocorrelation time comes from an
FFT in the reference and from lagged products here (1e-9), and the effective totals, pooled budgets and therefore the
budgets inherit that last-digit freedom (1e-9).  The solve sees a budget only through floor(n * budget): that integer
must be the reference's -- how far n * budget lies from the next integer is asserted against the 1e-9 freedom -- and on
the fixtures whose scores and switch costs are the reference's bits the calibrated penalty must then be the reference's
double.  `counts_general` starts from general integer counts; they stay below 7957, where the device's correctly rounded
log2 and this image's np.log2 first differ, so its scores are the reference's bits too;
its budgets come from `args["budget"]` scaled by the pooled estimate, so its penalties are held to 1e-9."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "composed_vectors.npz")
FROM_AUTOCORRELATION = {"autocorrelation_time", "effective_total_count", "effective_count"}
EXACT_FIXTURES = ("bigwig", "counts_exact_log", "counts_low_memory")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _inputs(gold, fixture):
    chroms = [str(c) for c in gold[f"{fixture}_chroms"]]
    args = json.loads(str(gold[f"{fixture}_args"][0]))
    return chroms, args, {c: (gold[f"{fixture}_{c}_intervals"], gold[f"{fixture}_{c}_matrix"]) for c in chroms}


@pytest.mark.parametrize("fixture", ["bigwig", "counts_exact_log", "counts_general", "counts_low_memory"])
def test_composed_driver_reproduces_the_references_bed(gpu, gold, fixture, tmp_path, monkeypatch):
    from rocco_amd import rocco as impl

    chroms, args, inputs = _inputs(gold, fixture)
    monkeypatch.chdir(tmp_path)  # the per-chromosome files go to the working directory, as in the reference
    exact = fixture in EXACT_FIXTURES
    cache = impl._build_chrom_cache(chroms, inputs, args)
    assert list(cache) == chroms
    budgets, budget_meta = impl._resolve_budgets(cache, args)
    want_meta = json.loads(str(gold[f"{fixture}_budget_meta"][0]))
    assert budget_meta["prior_fit_method"] == want_meta["prior_fit_method"]
    assert budget_meta["prior_dispersion_at_floor"] == want_meta["prior_dispersion_at_floor"]
    for c in chroms:
        entry = cache[c]
        want_scores = gold[f"{fixture}_{c}_scores"]
        assert isinstance(entry["scores"], np.ndarray) and entry["scores"].dtype == np.float64
        assert np.array_equal(entry["scores"], want_scores), c  # (counts_general too: every count is below 7957)
        gamma, count_hat, fraction_hat, total, n_loci, budget = gold[f"{fixture}_{c}_numbers"]
        want_rate = json.loads(str(gold[f"{fixture}_{c}_rate_meta"][0]))
        assert set(entry["budget_rate_meta"]) == set(want_rate), c
        for key, value in want_rate.items():
            got = entry["budget_rate_meta"][key]
            if isinstance(value, (str, bool)):
                assert got == value, (c, key)
            elif key in FROM_AUTOCORRELATION or not exact:
                assert np.isclose(got, value, rtol=1e-9, atol=1e-12), (c, key, got, value)
            else:
                assert got == value, (c, key, got, value)
        assert entry["num_loci"] == int(n_loci)
        if exact:
            assert entry["gamma"] == gamma and entry["budget_fraction_hat"] == fraction_hat, c
        assert np.isclose(entry["gamma"], gamma, rtol=1e-9) and np.isclose(entry["budget_fraction_hat"], fraction_hat, rtol=1e-9, atol=1e-15)
        assert np.isclose(entry["total_count"], total, rtol=1e-9) and np.isclose(entry["budget_count_hat"], count_hat, rtol=1e-9)
        assert np.isclose(budgets[c], budget, rtol=1e-9), (c, budgets[c], budget)
        want_gamma_meta = json.loads(str(gold[f"{fixture}_{c}_gamma_meta"][0]))
        assert entry["gamma_meta"]["characteristic_run_length"] == want_gamma_meta["characteristic_run_length"]
        assert entry["gamma_meta"]["positive_score_count"] == want_gamma_meta["positive_score_count"]
    files = impl._solve_cached_chromosomes(cache, budgets, args, "77")
    assert [os.path.basename(f) for f in files] == [f"rocco_77_{c}.bed" for c in chroms]
    for c, f in zip(chroms, files):
        assert open(f).read() == str(gold[f"{fixture}_{c}_bed"][0]), c
    final = impl.combine_chrom_results(files, str(tmp_path / "combined.bed"), name_features=False)
    assert open(final).read() == str(gold[f"{fixture}_combined_bed"][0])
    # What the solve sees of a budget is floor(n * budget) (rocco/dp.py:197).  The pooled budgets carry the 1e-9 freedom of
    # the autocorrelation time; the integer must not: n * budget has to lie further from an integer than that freedom.
    solved = impl.solve_cached_chromosomes(cache, budgets, selection_penalty=None, min_length_bp=args["min_length_bp"],
                                           write_files=False)
    for (c, _objective, details, _records) in solved:
        penalty, count, _obj, _pen = gold[f"{fixture}_{c}_solve"]
        n_loci, want_budget = int(gold[f"{fixture}_{c}_numbers"][4]), float(gold[f"{fixture}_{c}_numbers"][5])
        product = n_loci * want_budget
        assert int(np.floor(n_loci * budgets[c])) == int(np.floor(product)), c
        clearance = min(product - np.floor(product), np.ceil(product) - product) if product != np.floor(product) else 1.0
        assert clearance > 1.0e3 * (1.0e-9 * product), (c, product, clearance)  # three orders above the tolerance's reach
        assert details["selected_count"] == int(count), c
        if exact:
            # same scores, same switch cost, same target count: the calibration is the reference's, step for step
            assert details["selection_penalty"] == penalty, (c, details["selection_penalty"], penalty)
        else:
            assert np.isclose(details["selection_penalty"], penalty, rtol=1e-9, atol=1e-12), c


def test_run_chromosomes_end_to_end_and_cleans_up(gpu, gold, tmp_path, monkeypatch):
    from rocco_amd import rocco as impl

    chroms, args, inputs = _inputs(gold, "counts_exact_log")
    monkeypatch.chdir(tmp_path)
    args["output"] = str(tmp_path / "peaks.bed")
    args["narrowPeak"] = True  # summit tracks are written for the count branch and removed at the end
    final = impl.run_chromosomes(chroms + ["chrMissing"], inputs, args, run_id="5")
    assert open(final).read() == str(gold["counts_exact_log_combined_bed"][0])
    assert sorted(os.listdir(tmp_path)) == ["peaks.bed"]


def test_composed_driver_against_the_oracle_on_fresh_matrices(gpu, oracle, tmp_path, monkeypatch):
    """Inputs no fixture holds (other sizes, a fixed penalty, min_length_bp) against the oracle's composition, which
    tests/test_oracle_budget_golden.py pins to the reference's."""
    from rocco_amd import rocco as impl

    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(99)
    inputs = {}
    for chrom, n in (("chr7", 20000), ("chr12", 8191), ("chr3", 16385)):
        m = np.round(rng.gamma(1.0, 0.3, size=(5, n)) - 0.3 + rng.normal(0.0, 0.15, size=(5, n)), 5)
        for p in range(300, n - 100, 700):
            m[:, p:p + int(rng.integers(5, 40))] += rng.gamma(6.0, 0.8, size=(5, 1)) * (rng.random((5, 1)) < 0.8)
        inputs[chrom] = (1000 + np.arange(n, dtype=np.int64) * 25, np.round(m, 5))
    args = {"input_track_type": "bigwig", "budget_null_draws": 9, "threads": 1, "gamma": None, "budget": None,
            "scale_chrom_budgets": 0.8, "budget_posterior_quantile": 0.05, "selection_penalty": None, "min_length_bp": 75,
            "score_lower_bound_z": 1.0, "score_prior_df": 6.0, "score_precision_floor_ratio": 0.01}
    o_cache, o_budgets, o_solved, o_combined = oracle.run_chromosomes(list(inputs), inputs, args)
    args["output"] = str(tmp_path / "out.bed")
    final = impl.run_chromosomes(list(inputs), inputs, args, run_id="9")
    assert open(final).read() == oracle.bed_text(o_combined)
    cache = impl._build_chrom_cache(list(inputs), inputs, args)
    for c in inputs:
        assert np.array_equal(cache[c]["scores"], o_cache[c]["scores"])
        assert cache[c]["gamma"] == o_cache[c]["gamma"]
        assert cache[c]["budget_fraction_hat"] == o_cache[c]["budget_fraction_hat"]


# ------------------------------------------------------------------------------------------------------------------
# the seam: stand-in callables in the module namespace (tests/golden/make_golden_seam.py ran the reference on the same)
# ------------------------------------------------------------------------------------------------------------------
SEAM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seam_vectors.npz")


def _stand_ins(seam, name, chroms, seen):
    """The scenario's three callables, made from its data; every call notes the keyword names it was given."""
    data = {}
    for c in chroms:
        if f"{name}_{c}_starts" in seam.files:
            fake = json.loads(str(seam[f"{name}_{c}_fake"][0]))
            data[c] = dict(starts=seam[f"{name}_{c}_starts"], matrix=seam[f"{name}_{c}_matrix"], fraction=fake["fraction"],
                           meta=fake["meta"], window=fake["window"],
                           scores=seam[f"{name}_{c}_fake_scores"] if f"{name}_{c}_fake_scores" in seam.files else None,
                           mean=seam[f"{name}_{c}_fake_mean"] if f"{name}_{c}_fake_mean" in seam.files else None)

    def generate(chrom, *a, **k):
        seen["generate"].append(sorted(k))
        entry = data.get(chrom)
        return (None, None) if entry is None else (entry["starts"].copy(), entry["matrix"].copy())

    def wls(matrix, **k):
        seen["wls"].append(sorted(k))
        entry = next(e for e in data.values() if tuple(e["matrix"].shape) == tuple(matrix.shape))
        return entry["scores"].copy(), {"centered_matrix": np.zeros_like(entry["matrix"]), "local_baseline_window": entry["window"],
                                        "mean": entry["mean"].copy()}

    def estimate(first, *a, **k):
        seen["estimate"].append(sorted(k))
        n = np.asarray(k.get("observed_scores", first)).shape[-1]
        entry = next(e for e in data.values() if len(e["starts"]) == n)
        return entry["fraction"], dict(entry["meta"])

    return generate, wls, estimate


def test_cache_built_from_stand_ins_is_the_references(gpu, monkeypatch):
    from rocco_amd import rocco as impl

    seam = np.load(SEAM)
    assert len(seam["names"]) >= 5
    for name in (str(v) for v in seam["names"]):
        args = json.loads(str(seam[f"{name}_args"][0]))
        chroms = [str(c) for c in seam[f"{name}_chroms"]]
        seen = {"generate": [], "wls": [], "estimate": []}
        generate, wls, estimate = _stand_ins(seam, name, chroms, seen)
        monkeypatch.setattr(impl, "generate_chrom_matrix", generate)
        monkeypatch.setattr(impl, "score_loci_wls", wls)
        monkeypatch.setattr(impl, "estimate_budget_nonnull_fraction_from_wild_bootstrap_null", estimate)
        monkeypatch.setattr(impl, "estimate_budget_nonnull_fraction_from_score_track", estimate)
        cache = impl._build_chrom_cache(chroms, [], args)
        assert list(cache) == [str(c) for c in seam[f"{name}_cached"]], name
        # the stand-ins were called as the reference calls them: as often, with the same keywords
        assert seen == json.loads(str(seam[f"{name}_seen"][0])), name
        for c, entry in cache.items():
            assert sorted(entry) == [str(k) for k in seam[f"{name}_{c}_cache_keys"]], (name, c)
            assert np.array_equal(np.asarray(entry["scores"]), seam[f"{name}_{c}_cache_scores"]), (name, c)
            gamma, count_hat, fraction_hat, total, n_loci = seam[f"{name}_{c}_cache_numbers"]
            assert (entry["gamma"], entry["budget_count_hat"], entry["budget_fraction_hat"], entry["total_count"],
                    float(entry["num_loci"])) == (gamma, count_hat, fraction_hat, total, n_loci), (name, c)
            want_meta = json.loads(str(seam[f"{name}_{c}_cache_gamma_meta"][0]))
            assert (entry["gamma_meta"] is None) == (want_meta is None), (name, c)
            for key, value in (want_meta or {}).items():
                assert entry["gamma_meta"][key] == value, (name, c, key)


def test_cache_builder_errors(gpu):
    from rocco_amd import rocco as impl

    args = {"input_track_type": "bigwig", "gamma": 1.0, "threads": 1, "budget_null_draws": 4, "score_lower_bound_z": 1.0,
            "score_prior_df": 5.0, "score_precision_floor_ratio": 0.01}
    bad = np.ones((2, 50))
    bad[1, 7] = np.nan
    with pytest.raises(ValueError, match="chrQ matrix contains non-finite values"):
        impl._build_chrom_cache(["chrQ"], {"chrQ": (np.arange(50) * 50, bad)}, args)
    with pytest.raises(RuntimeError):  # a list of file names: decoding is the reference's readers' job
        impl._build_chrom_cache(["chrQ"], ["a.bw"], args)
    assert impl._build_chrom_cache(["chrNone"], {}, args) == {}
