"""Ad-hoc: the count-matrix budget estimator (estimate_budget_nonnull_fraction_from_wild_bootstrap_null) on one matrix with 1 and
several host workers for the multipliers: wall time and the estimate (must be the same).
   python scripts/wild_bootstrap_probe.py [K] [n] [draws] [workers,workers,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import inference, budget

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
draws = int(sys.argv[3]) if len(sys.argv) > 3 else 8
workers = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,4,8").split(",")]
rng = np.random.default_rng(4)
counts = np.round(rng.gamma(1.0, 3.0, size=(K, n)))
for p in range(500, n - 200, 5000):
    counts[:, p:p + 60] += rng.poisson(20.0, size=(K, 1))
scores, details = inference.score_loci_wls_device(torch.from_numpy(counts).to("cuda:0"))
centred = details["centered_matrix"]
for w in workers:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frac, meta = budget.estimate_budget_nonnull_fraction_from_wild_bootstrap_null(
        centred, observed_scores=scores, num_null_draws=draws, num_processes=w, return_details=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"K={K} n={n} draws={draws} workers={w}: {dt:.2f} s; fraction {frac!r}; draws used {meta.get('num_null_draws')}", flush=True)
