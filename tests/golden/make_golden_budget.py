#!/usr/bin/env python3
"""Expected outputs of the REFERENCE's budget / switch-cost estimation around the solve:

    make -C oracle ref
    python tests/golden/make_golden_budget.py

rocco.inference.estimate_budget_nonnull_fraction_from_score_track (+ details), _estimate_effective_sample_size,
estimate_empirical_bayes_budgets, and rocco.rocco._resolve_chrom_gamma / _resolve_budgets on this repository's own
inputs.  `rocco.inference` / `rocco.rocco` are imported under an empty package object (the package import itself
fails on the absent pysam; a dummy module stands in for it: no pysam code is on any path used here).  Writes
tests/golden/budget_vectors.npz -- data only, no reference source."""
import importlib
import json
import os
import sys
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

out, names = {}, []


def track(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "signal":  # non-negative medians of signal tracks (the bigWig case): background + enriched stretches
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        for p in rng.integers(0, n, size=max(1, n // 400)):
            s[p:p + int(rng.integers(4, 40))] += rng.gamma(6.0, 1.0)
        return np.round(s, 5)
    if kind == "centred":  # scores on both sides of zero, short-range dependent
        e = rng.normal(size=n + 20)
        s = np.convolve(e, np.ones(21) / np.sqrt(21.0), mode="valid")
        s[rng.integers(0, n, size=max(1, n // 300))] += 6.0
        return s
    if kind == "all_negative":
        return -np.round(rng.gamma(2.0, 0.5, size=n), 4) - 0.1
    return np.where(rng.random(n) < 0.5, 0.0, rng.normal(size=n))  # "zeros": many exact zeros


cases = [("signal", 30, 1), ("signal", 4000, 2), ("signal", 120000, 3), ("centred", 500, 4), ("centred", 60000, 5),
         ("all_negative", 3000, 6), ("zeros", 9000, 7), ("centred", 3, 8), ("signal", 1, 9)]
for kind, n, seed in cases:
    for draws, hint in ((25, None), (6, 16)):
        s = track(kind, n, seed)
        frac, det = inference.estimate_budget_nonnull_fraction_from_score_track(
            s, dependence_lag_hint=hint, num_null_draws=draws, return_details=True)
        name = f"{kind}_n{n}_d{draws}_h{hint}"
        names.append(name)
        out[f"{name}_scores"] = s
        out[f"{name}_fraction"] = np.array([frac])
        out[f"{name}_details"] = np.array([json.dumps({k: (v if isinstance(v, (str, bool)) else float(v)) for k, v in det.items()})])
        out[f"{name}_params"] = np.array([draws, -1 if hint is None else hint])
        soft = np.clip(s - det["null_center"], 0.0, None) / max(det["null_scale"], 1.0e-6)
        out[f"{name}_ess"] = np.array(inference._estimate_effective_sample_size(soft, int(det["ess_max_lag"])), dtype=np.float64)
        gamma, meta = impl._resolve_chrom_gamma("chrT", {"gamma": None}, s, det)
        out[f"{name}_gamma"] = np.array([gamma, meta["autocorrelation_time"], meta["characteristic_run_length"],
                                         meta["positive_score_median"], meta["positive_score_count"], meta["gamma_raw"]])
out["names"] = np.array(names)

# empirical-Bayes pooling and the final clipping: one, three and several chromosomes; dispersion at and above the floor
eb_cases = {
    "single": ({"chr1": 12.5}, {"chr1": 900.0}),
    "empty_single": ({"chr1": 0.0}, {"chr1": 0.0}),
    "three": ({"a": 4.0, "b": 61.0, "c": 17.5}, {"a": 1200.0, "b": 950.0, "c": 1010.0}),
    "eight": ({f"c{i}": v for i, v in enumerate([3.0, 40.0, 11.0, 26.5, 0.0, 88.0, 15.0, 7.25])},
              {f"c{i}": v for i, v in enumerate([1000.0, 1500.0, 800.0, 1200.0, 600.0, 2000.0, 900.0, 700.0])}),
    "at_floor": ({f"c{i}": 20.0 for i in range(6)}, {f"c{i}": 1000.0 for i in range(6)}),
}
eb_names = []
for name, (counts, totals) in eb_cases.items():
    for q in (0.01, 0.3):
        budgets, meta = inference.estimate_empirical_bayes_budgets(counts, totals, posterior_quantile=q)
        key = f"eb_{name}_q{q}"
        eb_names.append(key)
        out[f"{key}_counts"] = np.array(list(counts.values()))
        out[f"{key}_totals"] = np.array(list(totals.values()))
        out[f"{key}_budgets"] = np.array([budgets[c] for c in counts])
        out[f"{key}_meta"] = np.array([json.dumps({k: (v if isinstance(v, (str, bool)) else float(v)) for k, v in meta.items()})])
        for budget_arg, scale in ((None, 1.0), (0.03, 1.5)):
            cache = {c: {"budget_count_hat": counts[c], "total_count": totals[c]} for c in counts}
            final, _ = impl._resolve_budgets(cache, {"budget_posterior_quantile": q, "budget": budget_arg, "scale_chrom_budgets": scale})
            out[f"{key}_final_{budget_arg}_{scale}"] = np.array([final[c] for c in counts])
out["eb_names"] = np.array(eb_names)
path = os.path.join(HERE, "budget_vectors.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(names)} track cases, {len(eb_names)} pooling cases, {os.path.getsize(path) / 1e6:.2f} MB")
