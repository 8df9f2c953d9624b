#!/usr/bin/env python3
"""Expected outputs of the REFERENCE for (1) the count-matrix budget estimator and (2) the composed driver:

    make -C oracle ref
    python tests/golden/make_golden_composed.py

(1) rocco.inference.estimate_budget_nonnull_fraction_from_wild_bootstrap_null (rocco/inference.py:988-1148 over
    719-985) on centred matrices: the reference's own test input (tests/test_rocco.py:462-502) and random ones --
    one and several rows, few and many loci, draws that stop early, the worker-pool batching of the stopping rule
    (`num_processes`), a minimum effect, float32 (`--low_memory`) matrices, scores given or fitted.
(2) rocco.rocco._build_chrom_cache -> _resolve_budgets -> _solve_cached_chromosomes -> combine_chrom_results
    (rocco/rocco.py:933-1196, 194-240) with `generate_chrom_matrix` replaced by in-memory matrices exactly as the
    reference's own tests replace it (tests/test_rocco.py:566-689, 838-897): three fixtures -- bigWig tracks, count
    matrices whose log2(x + 1) is exact, general count matrices -- of three or four chromosomes each, automatic switch
    cost, data-driven budgets.  Stored: the matrices, every cache entry's numbers, the pooled budgets, each
    chromosome's BED text and the combined BED text.

`rocco.inference` / `rocco.rocco` are imported under an empty package object with a dummy `pysam` (the package import
fails on the absent pysam; no pysam code is on any path used here).  Writes tests/golden/wild_bootstrap_vectors.npz
and tests/golden/composed_vectors.npz -- data only, no reference source."""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("REFERENCE", "/root/reference")

pkg = types.ModuleType("rocco")
pkg.__path__ = [os.path.join(REFERENCE, "rocco"), os.path.join(ROOT, "oracle", "_ref")]
sys.modules["rocco"] = pkg
dummy = types.ModuleType("pysam")
dummy.AlignedSegment = type("AlignedSegment", (), {})
sys.modules.setdefault("pysam", dummy)
inference = importlib.import_module("rocco.inference")
impl = importlib.import_module("rocco.rocco")
assert inference._wls_native is not None and inference._baseline_native is not None


def plain(meta):
    return {k: (v if isinstance(v, (str, bool)) or v is None else float(v)) for k, v in meta.items()}


# ------------------------------------------------------------------------------------------------------------------
# (1) the estimator
# ------------------------------------------------------------------------------------------------------------------
out, names = {}, []


def centred(K, n, seed):
    """A centred K x n matrix the way the count path leaves one: dependent noise around zero plus shared stretches."""
    rng = np.random.default_rng(seed)
    e = rng.normal(0.0, 0.6, size=(K, n + 8))
    m = np.stack([np.convolve(row, np.ones(9) / 3.0, mode="valid") for row in e])
    for p in rng.integers(0, max(1, n - 30), size=max(1, n // 500)):
        m[:, p:p + int(rng.integers(5, 30))] += rng.gamma(4.0, 0.5) * (rng.random((K, 1)) < 0.8)
    return m


def record(name, centered, observed, kwargs):
    fraction, details = inference.estimate_budget_nonnull_fraction_from_wild_bootstrap_null(
        centered, observed_scores=observed, return_details=True, **kwargs)
    names.append(name)
    out[f"{name}_centered"] = centered
    if observed is not None:
        out[f"{name}_observed"] = observed
    out[f"{name}_kwargs"] = np.array([json.dumps(kwargs)])
    out[f"{name}_fraction"] = np.array([fraction])
    out[f"{name}_details"] = np.array([json.dumps(plain(details))])
    print(f"  {name}: fraction {fraction:.6g}, draws {details['num_null_draws']:.0f}/{details['max_null_draws']:.0f}")


# the reference's own test (tests/test_rocco.py:462-502): three tracks with two shared bumps
x = np.arange(512, dtype=np.float64)
peak1 = 6.0 * np.exp(-0.5 * ((x - 120.0) / 15.0) ** 2)
peak2 = 5.5 * np.exp(-0.5 * ((x - 320.0) / 15.0) ** 2)
chrom_matrix = np.vstack([0.25 + peak1 + peak2 + 0.05 * np.sin(x / 13.0),
                          0.20 + 0.95 * peak1 + 1.05 * peak2 + 0.04 * np.cos(x / 15.0),
                          0.22 + 1.1 * peak1 + 0.9 * peak2 + 0.05 * np.sin(x / 17.0)])
scores, det = inference.score_loci_wls(chrom_matrix, return_details=True)
record("reference_test", det["centered_matrix"], scores, dict(dependence_lag_hint=16, num_null_draws=6))

cases = [
    ("k1_n30", 1, 30, dict(num_null_draws=5)),
    ("k2_n700", 2, 700, dict(num_null_draws=25, dependence_lag_hint=25)),
    ("k3_n5000_pool4", 3, 5000, dict(num_null_draws=25, dependence_lag_hint=101, num_processes=4)),
    ("k3_n5000_pool1", 3, 5000, dict(num_null_draws=25, dependence_lag_hint=101, num_processes=1)),
    ("k5_n20000", 5, 20000, dict(num_null_draws=16, dependence_lag_hint=101, num_processes=3, random_seed=11)),
    ("k4_n9000_tuned", 4, 9000, dict(num_null_draws=9, lower_bound_z=0.5, prior_df=6.0, min_effect=0.15,
                                     precision_floor_ratio=0.05, dependence_lag_hint=40)),
    ("k2_n3", 2, 3, dict(num_null_draws=4)),
    ("k3_n1", 3, 1, dict(num_null_draws=4)),
]
for idx, (name, K, n, kw) in enumerate(cases):
    c = centred(K, n, 100 + idx)
    obs = None if idx % 2 else inference._score_centered_wls_matrix(c, lower_bound_z=kw.get("lower_bound_z", 1.0),
                                                                    prior_df=kw.get("prior_df", 5.0),
                                                                    min_effect=kw.get("min_effect"),
                                                                    precision_floor_ratio=kw.get("precision_floor_ratio", 0.01))[0]
    record(name, c, obs, kw)
record("k3_n6000_f32", centred(3, 6000, 77).astype(np.float32), None, dict(num_null_draws=12, dependence_lag_hint=101))
record("one_dimensional", centred(1, 800, 78)[0], None, dict(num_null_draws=8))
out["names"] = np.array(names)
path = os.path.join(HERE, "wild_bootstrap_vectors.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(names)} cases, {os.path.getsize(path) / 1e6:.2f} MB")


# ------------------------------------------------------------------------------------------------------------------
# (2) the composed driver
# ------------------------------------------------------------------------------------------------------------------
def signal_tracks(K, n, rng):
    """bigWig-like tracks (log-ratio style: background around zero), 5 decimals, enriched stretches in most samples."""
    m = np.round(rng.gamma(1.0, 0.3, size=(K, n)) - 0.3 + rng.normal(0.0, 0.2, size=(K, n)), 5)
    pos = 200
    while pos < n - 60:
        width = int(rng.integers(4, 40))
        m[:, pos:pos + width] += rng.gamma(6.0, 1.0, size=(K, 1)) * (rng.random((K, 1)) < 0.8)
        pos += int(1500 + rng.integers(-300, 300)) // 3
    return np.round(m, 5)


def count_matrix(K, n, rng, exact_log):
    """Read-count-like matrices.  `exact_log`: every entry is 2^k - 1, so log2(x + 1) is exact whatever log2 is used."""
    if exact_log:
        m = (2.0 ** rng.integers(0, 4, size=(K, n))) - 1.0
    else:
        m = rng.poisson(3.0, size=(K, n)).astype(np.float64)
    pos = 150
    while pos < n - 60:
        width = int(rng.integers(6, 40))
        on = rng.random((K, 1)) < 0.85
        if exact_log:
            m[:, pos:pos + width] = np.where(on, (2.0 ** rng.integers(4, 8, size=(K, width))) - 1.0, m[:, pos:pos + width])
        else:
            m[:, pos:pos + width] += on * rng.poisson(rng.gamma(6.0, 6.0), size=(K, width))
        pos += int(500 + rng.integers(-100, 100))
    return m


BASE_ARGS = {
    "chrom_sizes_file": None, "step": 50, "round_digits": 5, "effective_genome_size": None, "norm_method": "RPGC",
    "min_mapping_score": 0, "flag_include": None, "flag_exclude": None, "extend_reads": 0, "center_reads": False,
    "ignore_for_norm": [], "scale_factor": 1.0, "score_lower_bound_z": 1.0, "score_prior_df": 6.0,
    "score_min_effect": None, "score_precision_floor_ratio": 0.01, "gamma": None, "budget": None,
    "scale_chrom_budgets": 1.0, "budget_posterior_quantile": 0.01, "selection_penalty": None, "min_length_bp": None,
    "narrowPeak": False, "low_memory": False,
}

fixtures = {
    # name: (track type, [(chrom, n, K)], overrides)
    "bigwig": ("bigwig", [("chr1", 12000, 4), ("chr10", 7000, 4), ("chr2", 9000, 4), ("chrX", 5000, 4)],
               dict(budget_null_draws=25, threads=1)),
    "counts_exact_log": ("bam", [("chr1", 9000, 4), ("chr10", 5000, 4), ("chr2", 7000, 4)],
                         dict(budget_null_draws=12, threads=4, min_length_bp=100)),
    "counts_general": ("bam", [("chr2", 8000, 5), ("chr11", 6000, 5), ("chr1", 10000, 5)],
                       dict(budget_null_draws=10, threads=1, budget=0.03, scale_chrom_budgets=1.2)),
    "counts_low_memory": ("bam", [("chr3", 6000, 3), ("chr4", 5000, 3), ("chr5", 4000, 3), ("chr6", 4500, 3)],
                          dict(budget_null_draws=16, threads=4, low_memory=True)),
}
comp = {}
for fname, (track_type, chroms, overrides) in fixtures.items():
    rng = np.random.default_rng(sum(map(ord, fname)))
    data = {}
    for chrom, n, K in chroms:
        start = int(rng.integers(0, 2000)) * 50
        intervals = start + np.arange(n, dtype=np.int64) * 50
        if track_type == "bigwig":
            matrix = signal_tracks(K, n, rng)
        else:
            matrix = count_matrix(K, n, rng, exact_log=(fname in ("counts_exact_log", "counts_low_memory")))
            if overrides.get("low_memory"):
                matrix = matrix.astype(np.float32)  # readtracks.py:621: --low_memory matrices are float32
        data[chrom] = (intervals, matrix)
    args = dict(BASE_ARGS, input_track_type=track_type, **overrides)
    impl.generate_chrom_matrix = lambda chrom, *a, _d=data, **k: (_d[chrom][0].copy(), _d[chrom][1].copy())
    order = [c for c, _, _ in chroms]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            args["output"] = os.path.join(tmp, "combined.bed")
            cache = impl._build_chrom_cache(order, [], args)
            budgets, budget_meta = impl._resolve_budgets(cache, args)
            files = impl._solve_cached_chromosomes(cache, budgets, args, "77")
            final = impl.combine_chrom_results(files, args["output"], name_features=False)
            comp[f"{fname}_combined_bed"] = np.array([open(final).read()])
            for chrom, f in zip(order, files):
                comp[f"{fname}_{chrom}_bed"] = np.array([open(f).read()])
        finally:
            os.chdir(cwd)
    comp[f"{fname}_chroms"] = np.array(order)
    comp[f"{fname}_args"] = np.array([json.dumps({k: v for k, v in args.items() if k != "output"})])
    comp[f"{fname}_budget_meta"] = np.array([json.dumps(plain(budget_meta))])
    for chrom in order:
        entry = cache[chrom]
        comp[f"{fname}_{chrom}_intervals"] = data[chrom][0]
        comp[f"{fname}_{chrom}_matrix"] = data[chrom][1]
        comp[f"{fname}_{chrom}_scores"] = np.asarray(entry["scores"], dtype=np.float64)
        comp[f"{fname}_{chrom}_numbers"] = np.array([entry["gamma"], entry["budget_count_hat"], entry["budget_fraction_hat"],
                                                     entry["total_count"], entry["num_loci"], budgets[chrom]])
        comp[f"{fname}_{chrom}_rate_meta"] = np.array([json.dumps(plain(entry["budget_rate_meta"]))])
        comp[f"{fname}_{chrom}_gamma_meta"] = np.array([json.dumps(None if entry["gamma_meta"] is None else plain(entry["gamma_meta"]))])
        sol, obj, det = impl.solve_chrom_exact(entry["scores"], budget=budgets[chrom], gamma=entry["gamma"], return_details=True)
        comp[f"{fname}_{chrom}_solve"] = np.array([det["selection_penalty"], det["selected_count"], obj, det["penalized_objective"]])
        print(f"  {fname} {chrom}: n {entry['num_loci']}, gamma {entry['gamma']:.4g}, budget {budgets[chrom]:.5f}, "
              f"draws {entry['budget_rate_meta']['num_null_draws']:.0f}, selected {det['selected_count']}")
    print(f"  {fname}: {comp[f'{fname}_combined_bed'][0].count(chr(10))} combined intervals")
comp["fixtures"] = np.array(list(fixtures))
path = os.path.join(HERE, "composed_vectors.npz")
np.savez_compressed(path, **comp)
print(f"wrote {path}: {len(fixtures)} fixtures, {os.path.getsize(path) / 1e6:.2f} MB")
