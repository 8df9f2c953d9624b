"""Ad-hoc: time the cross-fit Whittaker baseline on a benchmark-sized matrix and a CPU sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
from rocco_amd import synth, inference
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4979129
dev = torch.device("cuda:0")
m = synth.hash_matrix_device(K, n, 11, device=dev)
m = torch.log2(m + 1.0)
m = (m - m.mean(dim=0, keepdim=True)).contiguous()
lam = inference._consenrich_whittaker_lambda(101)
out = torch.empty_like(m)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inference.crossfit_whittaker_baseline_device(m, lam, out)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"GPU baseline K={K} n={n}: {t*1e3:.1f} ms  ({K*n/t/1e9:.3f} G values/s, {5*8*K*n/t/1e9:.1f} GB/s of row traffic)")
import pyoracle
ns = min(n, 200000)
sample = m[: min(K, 4), :ns].contiguous().cpu().numpy()
t0 = time.perf_counter(); ref = pyoracle.crossfit_whittaker_baseline(sample, lam); t = time.perf_counter() - t0
print(f"CPU oracle (1 core) {sample.shape}: {t*1e3:.1f} ms ({sample.size/t/1e6:.2f} M values/s)")
got = inference.crossfit_whittaker_baseline_device(m[: min(K, 4), :ns].contiguous(), lam).cpu().numpy()
print("bit-exact on the sample:", got.tobytes() == ref.tobytes())
