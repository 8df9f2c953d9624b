"""Ad-hoc: scoring throughput above K = 100 (two-half kernel up to 200, rank counting beyond): GB/s of algorithmic bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, rocco as rr
dev = torch.device("cuda:0")
n = 4979129
n_large = 1000000  # (K > 300: a 1 M-locus matrix, 8 GB at K = 1000)
for K in (33, 64, 90, 100, 128, 160, 200, 256, 300, 400, 600, 1000):
    n = n if K <= 300 else n_large
    m = synth.hash_matrix_device(K, n, 7, device=dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(2):
        rr.score_central_tendency_chrom_device(m, out)
    torch.cuda.synchronize()
    reps = 5 if K <= 200 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        rr.score_central_tendency_chrom_device(m, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"K={K}: {1e3 * dt:.3f} ms, {(8 * K + 8) * n / dt / 1e9:.0f} GB/s ({(8 * K + 8) * n / dt / 8e12:.3f} of 8 TB/s)")
    del m, out
