#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE in the build container.

Run here (where /root/reference is mounted) after `make -C oracle ref`:
    python tests/golden/make_golden.py
Writes tests/golden/*.npz / *.json: inputs and the reference's own outputs for the hot path
(rocco/dp.py, rocco/rocco.py scoring + BED helpers).  Only data is written -- no reference source.
The reference's own test vectors (tests/test_rocco.py:398-437, 838-897) are included verbatim as
inputs together with what the reference returns for them.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

import ref_loader  # noqa: E402
from rocco_amd.synth import hash_matrix, survey_matrix  # noqa: E402

dp = ref_loader.load_reference_dp()
rr = ref_loader.load_reference_rocco()

out = {}

# ---- 1. reference test_exact_dp_matches_bruteforce inputs (tests/test_rocco.py:398-415) ----
rng = np.random.default_rng(7)
scores = rng.normal(size=9)
costs = rng.uniform(0.2, 1.3, size=8)
for k, lam in enumerate((-0.5, 0.0, 0.6, 1.4)):
    sol, val, cnt = dp.solve_penalized_chain(scores, costs, lam)
    out[f"brute_{k}_scores"], out[f"brute_{k}_costs"] = scores, costs
    out[f"brute_{k}_lambda"] = np.float64(lam)
    out[f"brute_{k}_solution"], out[f"brute_{k}_value"], out[f"brute_{k}_count"] = sol, np.float64(val), np.int64(cnt)

# ---- 2. reference test_solve_chrom_exact_respects_budget vector (tests/test_rocco.py:419-437) ----
s8 = np.array([0.5, 1.5, 1.4, -0.2, 3.0, 2.8, -0.1, 0.1])
sol, obj, det = dp.solve_chrom_exact(s8, budget=0.375, gamma=1.0, return_details=True)
out["budget8_scores"], out["budget8_solution"], out["budget8_objective"] = s8, sol, np.float64(obj)
out["budget8_details"] = np.array([det["penalized_objective"], det["selected_count"], det["selected_fraction"],
                                   det["selection_penalty"]])

# ---- 3. fixed-penalty solves: random scores, vector costs, integer ties ----
cases = []
rng = np.random.default_rng(2024)
for i in range(12):
    n = int(rng.choice([1, 2, 7, 33, 257, 1500, 4099]))
    if i % 3 == 2:
        s = rng.integers(-3, 4, size=n).astype(np.float64)
        c = rng.integers(0, 3, size=max(n - 1, 0)).astype(np.float64)
        lam = float(rng.integers(-1, 2))
    else:
        s = np.round(rng.gamma(1.0, 0.3, size=n), 5)
        c = rng.uniform(0.0, 2.0, size=max(n - 1, 0))
        lam = float(rng.uniform(-0.5, 1.5))
    sol, val, cnt = dp.solve_penalized_chain(s, c, lam)
    out[f"fixed_{i}_scores"], out[f"fixed_{i}_costs"], out[f"fixed_{i}_lambda"] = s, c, np.float64(lam)
    out[f"fixed_{i}_solution"], out[f"fixed_{i}_value"], out[f"fixed_{i}_count"] = sol, np.float64(val), np.int64(cnt)
out["fixed_n"] = np.int64(12)

# ---- 4. budgeted solves on synthetic K x n matrices (median-scored), incl. BED text ----
budget_cases = [
    ("survey_k3", survey_matrix(30000, 3, 20240 + 21), 0.02, 1.0),
    ("survey_k10", survey_matrix(20000, 10, 20240 + 1), 0.05, 0.5),
    ("hash_k6", hash_matrix(6, 25000, seed=99), 0.03, 2.0),
    ("hash_k1", hash_matrix(1, 9000, seed=5), 0.01, 10.0),
    ("tiny", np.round(np.random.default_rng(3).gamma(1.0, 0.3, size=(2, 40)), 5), 0.1, 1.0),
]
names = []
cwd = os.getcwd()
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    for name, m, budget, gamma in budget_cases:
        scores = rr.score_central_tendency_chrom(m, method="quantile", quantile=0.5)
        sol, obj, det = dp.solve_chrom_exact(scores, budget=budget, gamma=gamma, return_details=True)
        intervals = np.arange(m.shape[1], dtype=np.int64) * 50
        bed = rr.chrom_solution_to_bed("chrT", intervals, sol, ID=name, min_length_bp=None)
        bed_min = rr.chrom_solution_to_bed("chrT", intervals, sol, ID=name + "_min", min_length_bp=150)
        out[f"bud_{name}_matrix"] = m
        out[f"bud_{name}_scores"] = scores
        out[f"bud_{name}_params"] = np.array([budget, gamma])
        out[f"bud_{name}_solution"] = sol
        out[f"bud_{name}_objective"] = np.float64(obj)
        out[f"bud_{name}_details"] = np.array([det["penalized_objective"], det["selected_count"],
                                               det["selected_fraction"], det["selection_penalty"]])
        out[f"bud_{name}_bed"] = np.frombuffer(open(bed, "rb").read(), dtype=np.uint8)
        out[f"bud_{name}_bed_min150"] = np.frombuffer(open(bed_min, "rb").read(), dtype=np.uint8)
        names.append(name)
    # ---- 5. combine_chrom_results on per-chromosome BED files (lexicographic chromosome order) ----
    files = []
    for chrom in ("chr2", "chr10", "chr1"):
        recs = [(chrom, 100, 200), (chrom, 200, 260), (chrom, 500, 650), (chrom, 640, 700)]
        path = f"in_{chrom}.bed"
        with open(path, "w") as fh:
            for c, a, b in recs:
                fh.write(f"{c}\t{a}\t{b}\n")
        files.append(path)
    combined = rr.combine_chrom_results(files, "combined.bed")
    out["combine_text"] = np.frombuffer(open(combined, "rb").read(), dtype=np.uint8)
    os.chdir(cwd)
out["bud_names"] = np.array(names)

# ---- 6. the reference's bigWig-median expectation (tests/test_rocco.py:838-897) ----
m2 = np.array([[0.0, 2.0, 1.0, 0.0], [0.0, 3.0, 2.0, 0.0]])
out["median2_matrix"] = m2
out["median2_scores"] = rr.score_central_tendency_chrom(m2, method="quantile", quantile=0.5)

np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
meta = {"generator": "tests/golden/make_golden.py", "reference_version": "1.11.0",
        "numpy": np.__version__, "n_arrays": len(out)}
with open(os.path.join(HERE, "reference_vectors.json"), "w") as fh:
    json.dump(meta, fh, indent=1)
print("wrote", len(out), "arrays", os.path.getsize(os.path.join(HERE, "reference_vectors.npz")), "bytes")
