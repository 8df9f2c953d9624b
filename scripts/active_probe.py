"""Ad-hoc: budget solve of one benchmark chromosome with ROCCO_HIP_DEBUG survey output."""
import os, sys, time
os.environ["ROCCO_HIP_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, dp, rocco as rr

chrom = sys.argv[1] if len(sys.argv) > 1 else "chr1"
K = 100
genome = synth.chrom_loci(50, None)
idx = [i for i, (nm, _) in enumerate(genome) if nm == chrom][0]
n = genome[idx][1]
m = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, idx), device=torch.device("cuda"))
s = torch.empty(n, dtype=torch.float64, device="cuda")
rr.score_central_tendency_chrom_device(m, s)
del m
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = dp.solve_chrom_exact_device(s, budget=0.02, gamma=1.0)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{chrom} n={n}: {t*1e3:.1f} ms details={r[2]}", flush=True)
