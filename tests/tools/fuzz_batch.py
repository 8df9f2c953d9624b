"""Ad-hoc fuzzing: batched / grouped calibration (pipeline.solve_rank) against the CPU oracle, bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import pyoracle as po
from rocco_amd import pipeline
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + seconds
t_said = time.time()
it = bad = chroms = 0
while time.time() < t_end:
    rng = np.random.default_rng(990000 + it)
    C = int(rng.integers(1, 9))
    works, hosts = [], []
    for c in range(C):
        n = int(rng.choice([2, 40, 1000, 8192, 8193, 30000, 150000, 600000])) + int(rng.integers(0, 50))
        K = int(rng.choice([1, 2, 3, 4, 7, 10]))
        kind = rng.choice(["gamma", "int", "normal"])
        if kind == "gamma":
            m = np.round(rng.gamma(1.0, 0.3, (K, n)), 5) + (rng.random((1, n)) < 0.02) * rng.gamma(6.0, 1.0, (K, n))
        elif kind == "int":
            m = rng.integers(0, 4, (K, n)).astype(float)
        else:
            m = rng.normal(0, 1, (K, n))
        budget = float(rng.choice([0.005, 0.02, 0.05, 0.1])); gamma = float(rng.choice([0.0, 0.5, 1.0, 3.0]))
        works.append(pipeline.ChromWork(f"c{c}", torch.from_numpy(m).cuda(), budget, gamma, step=50))
        hosts.append((m, budget, gamma, n))
    res = pipeline.solve_rank(works, groups=int(rng.integers(1, 5)))
    for r, (m, budget, gamma, n) in zip(res, hosts):
        s = np.median(m, axis=0)
        o_sol, _obj, o_det = po.solve_chrom_exact(s, budget=budget, gamma=gamma, return_details=True)
        ok = (r["selection_penalty"] == o_det["selection_penalty"] and r["selected_count"] == o_det["selected_count"]
              and np.array_equal(r["solution"].cpu().numpy(), o_sol)
              and pipeline.runs_to_records(r) == po.chrom_solution_records(r["name"], np.arange(n, dtype=np.int64) * 50, o_sol))
        chroms += 1
        if not ok:
            bad += 1; print(f"MISMATCH it={it} chrom={r['name']} n={n} budget={budget} gamma={gamma}", flush=True)
    it += 1
    if time.time() - t_said > 60.0:  # (a GPU box takes a command that says nothing for minutes to be hung)
        t_said = time.time(); print(f"... {it} iterations so far", flush=True)
print(f"{it} batches, {chroms} chromosomes, {bad} mismatches")
