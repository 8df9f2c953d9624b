"""Ad-hoc: is the K=100 median kernel bound by HBM or by its min/max network?  Same kernel on a matrix that
fits the 256 MiB Infinity Cache (no HBM traffic after the first pass) and on the benchmark's chr1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, rocco as rr
dev = torch.device("cuda:0")
for n in (131072, 262144, 4979129):
    m = synth.hash_matrix_device(100, n, 5, device=dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(5):
        rr.score_central_tendency_chrom_device(m, out)
    torch.cuda.synchronize()
    reps = 50 if n < 1e6 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        rr.score_central_tendency_chrom_device(m, out)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    print(f"n={n}: {t*1e6:.1f} us/launch, {n/t/1e9:.2f} G loci/s, {808*n/t/1e12:.2f} TB/s equivalent")
