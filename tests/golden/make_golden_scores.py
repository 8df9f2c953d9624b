#!/usr/bin/env python3
"""Expected outputs of the REFERENCE's post-hoc peak-scoring helpers (rocco/scores.py) on this repository's own inputs:

    python tests/golden/make_golden_scores.py

`_peak_signal_stat` per peak, `EmpiricalNull.survival` per length bin, `_assign_length_bins`, and
scipy.stats.false_discovery_control (what score_peaks calls at rocco/scores.py:583) -- the parts of `score_peaks` that do
not touch a BAM file (the function itself cannot run here: it counts reads with pysam).  `rocco.scores` is imported
with a dummy `pysam` module in place.  Writes tests/golden/scores_vectors.npz -- data only."""
import importlib
import os
import sys
import types

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("REFERENCE", "/root/reference")
pkg = types.ModuleType("rocco")
pkg.__path__ = [os.path.join(REFERENCE, "rocco"), os.path.join(ROOT, "oracle", "_ref")]
sys.modules["rocco"] = pkg
dummy = types.ModuleType("pysam")
dummy.AlignedSegment = type("AlignedSegment", (), {})
sys.modules.setdefault("pysam", dummy)
scores = importlib.import_module("rocco.scores")

rng = np.random.default_rng(31)
out = {}
for name, P, K in (("small", 40, 3), ("wide", 2500, 12), ("many", 20000, 5), ("one_sample", 300, 1)):
    lengths = rng.integers(50, 4000, size=P).astype(np.float64)
    counts = rng.gamma(2.0, 30.0, size=(P, K)) * (lengths[:, None] / 500.0)
    counts[rng.random((P, K)) < 0.05] = 0.0
    binned, reps = scores._assign_length_bins(lengths, max_bins=24)
    nulls = {int(r): rng.gamma(2.0, 1.2, size=int(rng.integers(20, 500))) for r in reps}
    sig = np.array([scores._peak_signal_stat(counts[i], lengths[i], row_scale=1000, pc=1) for i in range(P)])
    pvals = np.array([scores.EmpiricalNull(nulls[int(binned[i])]).survival(sig[i]) for i in range(P)])
    qvals = stats.false_discovery_control(pvals, method="bh")
    out[f"{name}_counts"], out[f"{name}_lengths"], out[f"{name}_binned"] = counts, lengths, binned
    out[f"{name}_null_keys"] = np.array(sorted(nulls))
    for k, v in nulls.items():
        out[f"{name}_null_{k}"] = v
    out[f"{name}_sig"], out[f"{name}_pvals"], out[f"{name}_qvals"] = sig, pvals, qvals
out["names"] = np.array(["small", "wide", "many", "one_sample"])
path = os.path.join(HERE, "scores_vectors.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")
