#!/usr/bin/env python3
"""Golden vectors of the per-draw part of the wild-bootstrap budget null, produced
by the REFERENCE's own functions `_fit_budget_null_residual_template` and `_compute_budget_null_draw`
(rocco/inference.py:688-722, 628-685) over its own compiled extension modules (see make_golden_score_loci_wls.py
for how rocco.inference is imported without the package's pysam dependency):

    make -C oracle ref
    python tests/golden/make_golden_budget_null.py

The multipliers of every draw are regenerated with the reference's `_generate_dependent_wild_weights` from the same
NumPy generator stream `_compute_budget_null_draw` uses and stored next to its four statistics, so a checker can
replay the draw without NumPy's generator.  Only data is written -- no reference source.
"""
import importlib
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
inf = importlib.import_module("rocco.inference")
assert inf._wls_native is not None

rng0 = np.random.default_rng(31337)
out, names = {}, []
for K, n in ((1, 40), (3, 500), (1, 8191), (1, 8200), (1, 16400)):
    for label, kw in (("default", dict(lower_bound_z=1.0, prior_df=5.0, min_effect=None, precision_floor_ratio=0.01)),
                      ("tuned", dict(lower_bound_z=0.5, prior_df=2.0, min_effect=0.2, precision_floor_ratio=0.1))):
        centered = rng0.normal(0.0, 0.7, size=(K, n)) + 1.5 * (rng0.random((1, n)) < 0.04)
        template, fitted_scores, positive = inf._fit_budget_null_residual_template(centered, **kw)
        null_center = float(np.median(fitted_scores))
        null_soft_scale = float(max(inf._robust_scale(fitted_scores), 1.0e-6))
        null_threshold = float(null_center + 1.5 * null_soft_scale)
        kernel = inf._build_budget_bootstrap_kernel(inf._resolve_budget_bootstrap_bandwidth(n, 25))
        base_seed = 7 + K + n
        inf._init_budget_null_process(template, kw["lower_bound_z"], kw["prior_df"], kw["min_effect"],
                                      kw["precision_floor_ratio"], null_center, null_soft_scale, null_threshold, kernel,
                                      base_seed)
        name = f"k{K}_n{n}_{label}"
        names.append(name)
        out[f"{name}_centered"] = centered
        out[f"{name}_template"] = template
        out[f"{name}_fitted_scores"] = fitted_scores
        out[f"{name}_positive"] = positive
        out[f"{name}_params"] = np.array([kw["lower_bound_z"], kw["prior_df"],
                                          np.nan if kw["min_effect"] is None else kw["min_effect"],
                                          kw["precision_floor_ratio"], null_center, null_soft_scale, null_threshold])
        for draw in range(2):
            stats = inf._compute_budget_null_draw(draw)
            rng = np.random.default_rng(base_seed + (104729 * (draw + 1)))  # inference.py:655
            weights = np.stack([inf._generate_dependent_wild_weights(n, kernel=kernel, rng=rng) for _ in range(K)])
            out[f"{name}_draw{draw}_weights"] = weights
            out[f"{name}_draw{draw}_stats"] = np.array(stats, dtype=np.float64)
out["names"] = np.array(names)
path = os.path.join(HERE, "budget_null_vectors.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(names)} cases, {os.path.getsize(path) / 1e6:.2f} MB")
