#!/usr/bin/env python3
"""Expected outputs of the REFERENCE's `_score_centered_wls_matrix` and `_consenrich_crossfit_whittaker_baseline`
(rocco/inference.py:231-299, 145-182) on a handful of hand-built situations, for tests/test_gpu_count_path_cases.py:

    make -C oracle ref
    python tests/golden/make_golden_count_path_cases.py

The situations (inputs are this repository's own): an all-zero matrix (every value tied), a quiet and a locally
noisy copy of one signal, three near-identical tracks scored with and without a precision floor, a narrow spike on
a broad hump for the baseline.  Writes tests/golden/count_path_cases.npz -- data only, no reference source.
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

TRACKS = ("mean", "raw_variance", "prior_variance", "moderated_variance", "standard_error", "z_scores")
out = {}


def wls_case(name, centered, **kw):
    scores, details = inference._score_centered_wls_matrix(centered, **kw)
    out[f"{name}_centered"] = np.asarray(centered, dtype=np.float64)
    out[f"{name}_kwargs"] = np.array([kw.get("lower_bound_z", 1.0), kw.get("prior_df", 5.0), kw.get("spatial_window", 31),
                                      kw.get("precision_floor_ratio", 0.01)])
    out[f"{name}_scores"] = scores
    for key in TRACKS:
        out[f"{name}_{key}"] = np.asarray(details[key], dtype=np.float64)


# every value tied
wls_case("zeros", np.zeros((4, 40000)), lower_bound_z=1.0, prior_df=5.0)
# one quiet and one locally noisy copy of a smooth signal
t = np.linspace(0.0, 6.0, 701)
smooth = 0.7 * np.cos(1.3 * t) + 0.2 * np.sin(3.1 * t)
noisy = smooth.copy()
noisy[250:420] += 0.6 * (1.0 - 2.0 * (np.arange(170) % 2))
wls_case("noisy_pair", np.vstack([smooth, noisy]), lower_bound_z=0.0, prior_df=6.0, spatial_window=31)
out["noisy_pair_region"] = np.array([250, 420])
# near-identical tracks with and without a floor on the precisions
trio = np.array([[0.03, 0.9, 1.1, 0.9, 0.02, 0.01], [0.05, 1.0, 1.0, 1.0, 0.04, 0.02], [0.04, 1.1, 0.9, 1.1, 0.05, 0.03]])
wls_case("trio_no_floor", trio, prior_df=6.0, precision_floor_ratio=0.0)
wls_case("trio_floor", trio, prior_df=6.0, precision_floor_ratio=0.3)
# narrow spike on a broad hump: the baseline follows the hump, not the spike
x = np.arange(151, dtype=np.float64)
hump = 2.0 * np.exp(-0.5 * ((x - 75.0) / 22.0) ** 2)
signal = hump + 4.0 * np.exp(-0.5 * ((x - 75.0) / 2.0) ** 2)
out["baseline_signal"] = signal
out["baseline_hump"] = hump
out["baseline_block"] = np.array([45])
out["baseline_expected"] = inference._consenrich_crossfit_whittaker_baseline(signal, block_size=45)
out["baseline_short_expected"] = inference._consenrich_crossfit_whittaker_baseline(np.full(20, 3.0))
path = os.path.join(HERE, "count_path_cases.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB")
