"""Ad-hoc: the batched rolling-variance launch alone (rocco_hip_wls_rolling_variances_batch_f64) over a genome's worth of
rows at K = 100: wall time per call and ns per locus of the longest row."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, inference

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
names = None if len(sys.argv) <= 2 or sys.argv[2] == "all" else sys.argv[2].split(",")
genome = synth.chrom_loci(50, names)
dev = torch.device("cuda:0")
mats = [synth.hash_matrix_device(K, n, synth.chrom_seed(7, idx), device=dev) - 0.3 for idx, (_, n) in enumerate(genome)]
longest = max(n for _, n in genome)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = inference.wls_rolling_variances_batch_device(mats, spatial_window=31)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{len(mats)} matrices, K={K}: {dt * 1e3:.1f} ms = {dt / longest * 1e9:.1f} ns per locus of the longest row; checksum {float(out[0][0, :1000].sum()):.12g}", flush=True)
    del out
