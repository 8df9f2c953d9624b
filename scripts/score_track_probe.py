"""Ad-hoc: the score-track budget estimator on one long track with 1 and several host workers.
   python scripts/score_track_probe.py [n] [draws] [workers,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import budget
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 16
workers = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,4,16").split(",")]
rng = np.random.default_rng(3)
s = rng.gamma(1.0, 0.3, size=n) - 0.3
for p in range(1000, n - 200, 7000):
    s[p:p + 60] += rng.gamma(6.0, 0.8)
s_t = torch.from_numpy(s).to("cuda:0")
for w in workers:
    t0 = time.perf_counter()
    frac, meta = budget.estimate_budget_nonnull_fraction_from_score_track(s_t, num_null_draws=draws, min_null_draws=draws, num_processes=w, return_details=True)
    print(f"n={n} draws={draws} workers={w}: {time.perf_counter() - t0:.2f} s; fraction {frac!r}; draws used {meta['num_null_draws']}", flush=True)
