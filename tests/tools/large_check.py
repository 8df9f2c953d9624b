"""One-off: parity at BASELINE config-5 scale (hg38 chr1 at 10 bp = 24.9 M loci) against the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
from rocco_amd import synth, dp, rocco as rr
import pyoracle
n, K = 24_895_643, 10
dev = torch.device("cuda:0")
m = synth.hash_matrix_device(K, n, 77, device=dev)
s = rr.score_central_tendency_chrom_device(m)
del m
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sol, obj, det = dp.solve_chrom_exact_device(s, budget=0.02, gamma=1.0)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"GPU solve n={n}: {t*1e3:.1f} ms  {det}")
t0 = time.perf_counter()
ref_sol, ref_obj, ref_det = pyoracle.solve_chrom_exact(s.cpu().numpy(), budget=0.02, gamma=1.0, return_details=True)
print(f"oracle: {time.perf_counter()-t0:.1f} s {ref_det}")
print("solution identical:", np.array_equal(sol.cpu().numpy(), ref_sol), "penalty diff", det["selection_penalty"] - ref_det["selection_penalty"],
      "count", det["selected_count"], ref_det["selected_count"])
b, e = rr.decode_runs_device(sol)
print("intervals", int(b.numel()))
