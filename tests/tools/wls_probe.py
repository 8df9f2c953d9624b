"""Ad-hoc: time the centred-WLS scoring on a benchmark-sized matrix and a CPU sample."""
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
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = inference.score_centered_wls_device(m)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"GPU wls K={K} n={n}: {t*1e3:.1f} ms  ({K*n/t/1e9:.3f} G values/s)", flush=True)
import pyoracle
ns = min(n, 1000000)
Ks = min(K, 4)
sample_t = m[:Ks, :ns].contiguous()
sample = sample_t.cpu().numpy()
t0 = time.perf_counter(); ref = pyoracle.score_centered_wls(sample); t = time.perf_counter() - t0
print(f"CPU oracle (1 core) {sample.shape}: {t*1e3:.1f} ms ({sample.size/t/1e6:.2f} M values/s)")
got = inference.score_centered_wls_device(sample_t)
print("bit-exact on the sample:", all(g.cpu().numpy().tobytes() == r.tobytes() for g, r in zip(got[:6], ref[:6])))
