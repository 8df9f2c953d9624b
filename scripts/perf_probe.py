"""Ad-hoc timing probe (not part of the bench contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, dp, rocco as rr

def timeit(fn, reps=3):
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        t=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t)
    return min(ts)

for K in (10, 100):
    n = 4979129
    m = synth.hash_matrix_device(K, n, seed=1)
    out = torch.empty(n, dtype=torch.float64, device='cuda')
    t = timeit(lambda: rr.score_central_tendency_chrom_device(m, out))
    print(f"median K={K} n={n}: {t*1e3:.2f} ms  {K*n*8/t/1e9:.1f} GB/s")
    if K == 10:
        s = out.clone()
    del m
n = s.shape[0]
t = timeit(lambda: dp.solve_penalized_chain_device(s, 1.0, 0.3, want_solution=False), reps=2)
print(f"exact single-lambda count-only n={n}: {t*1e3:.1f} ms")
t = timeit(lambda: dp.solve_penalized_chain_device(s, 1.0, 0.3, want_solution=True), reps=2)
print(f"exact single-lambda with solution n={n}: {t*1e3:.1f} ms")
t0=time.perf_counter(); r = dp.solve_chrom_exact_device(s, budget=0.02, gamma=1.0); torch.cuda.synchronize(); t=time.perf_counter()-t0
print(f"exact budget solve n={n}: {t*1e3:.1f} ms  details={r[2]}")
sol = r[0]
t = timeit(lambda: rr.decode_runs_device(sol))
print(f"decode n={n}: {t*1e3:.3f} ms runs={rr.decode_runs_device(sol)[0].shape[0]}")
