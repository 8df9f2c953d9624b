#!/usr/bin/env python3
"""Golden vectors of `score_loci_wls`, produced by the REFERENCE's own Python
function over its own compiled extension modules:

    make -C oracle ref          # builds oracle/_ref/_wls*.so and _baseline*.so from rocco/_wls.c, rocco/_baseline.c
    python tests/golden/make_golden_score_loci_wls.py

`import rocco` fails in this container (rocco/scores.py imports pysam, which is not installed), so
`rocco.inference` is imported alone under an empty package object whose search path is the reference's
package directory followed by oracle/_ref (where the two extension modules were built).  Writes
tests/golden/score_loci_wls_vectors.npz: count matrices, the log-scaled matrix NumPy produced on the generating
host (so a checker can separate the last-place freedom of log2 from everything downstream), scores and detail
tracks.  Only data is written -- no reference source.
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
inference = importlib.import_module("rocco.inference")
assert inference._wls_native is not None and inference._baseline_native is not None

TRACKS = ("mean", "raw_variance", "prior_variance", "moderated_variance", "standard_error", "z_scores",
          "degrees_of_freedom", "centered_matrix")
rng = np.random.default_rng(777)
out, names = {}, []
for K, n in ((1, 2), (2, 3), (1, 24), (3, 25), (2, 26), (4, 300), (2, 1201), (2, 3000)):
    for variant in ("counts", "pow2", "fractional"):
        lam = rng.gamma(2.0, 2.0, size=(K, 1)) * (1.0 + 6.0 * (rng.random((1, n)) < 0.05))
        counts = rng.poisson(lam).astype(np.float64)
        if variant == "pow2":
            counts = np.ldexp(1.0, rng.integers(0, 7, size=(K, n))) - 1.0  # log2(count + 1) exact everywhere
        elif variant == "fractional":
            counts = counts * rng.random((K, n)) - 0.05  # a few negative values: clipped to zero
        for label, kw in (("default", {}), ("tuned", {"lower_bound_z": 0.5, "prior_df": 2.0, "min_effect": 0.2,
                                                        "precision_floor_ratio": 0.1})):
            scores, details = inference.score_loci_wls(counts, return_details=True, **kw)
            name = f"k{K}_n{n}_{variant}_{label}"
            names.append(name)
            out[f"{name}_counts"] = counts
            out[f"{name}_log"] = inference._log_scale_wls_matrix(counts)
            out[f"{name}_params"] = np.array([kw.get("lower_bound_z", 1.0), kw.get("prior_df", 5.0),
                                              kw.get("min_effect", np.nan), kw.get("precision_floor_ratio", 0.01)])
            out[f"{name}_scores"] = scores
            for key in TRACKS:
                out[f"{name}_{key}"] = np.asarray(details[key], dtype=np.float64)
            out[f"{name}_scalars"] = np.array([details["local_baseline_window"], details["local_baseline_lambda"],
                                               details["min_effect"], details["precision_floor_ratio"],
                                               details["prior_spatial_window"]], dtype=np.float64)
out["names"] = np.array(names)
path = os.path.join(HERE, "score_loci_wls_vectors.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(names)} cases, {os.path.getsize(path) / 1e6:.2f} MB")
