"""Ad-hoc: time the count-path scoring (rows a2-a4: score_loci_wls) on a benchmark-sized matrix, with a CPU sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from rocco_amd import synth, inference
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4979129
dev = torch.device("cuda:0")
counts = synth.hash_matrix_device(K, n, 11, device=dev)
counts = (counts * 20.0).contiguous()  # count-like magnitudes
for rep in range(3):
    work = counts.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    scores, details = inference.score_loci_wls_device(work, overwrite_input=True)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"GPU score_loci_wls K={K} n={n}: {t*1e3:.1f} ms ({K*n/t/1e9:.3f} G values/s, {n/t/1e6:.2f} M loci/s)", flush=True)
    del work, details
import pyoracle
ns, Ks = min(n, 400000), min(K, 4)
sample_t = counts[:Ks, :ns].contiguous()
sample = sample_t.cpu().numpy()
t0 = time.perf_counter(); ref_scores, ref = pyoracle.score_loci_wls(sample); t = time.perf_counter() - t0
print(f"CPU oracle (1 core) {sample.shape}: {t*1e3:.1f} ms ({sample.size/t/1e6:.2f} M values/s)")
got, gd = inference.score_loci_wls_device(sample_t)
d = np.abs(got.cpu().numpy() - ref_scores).max() / np.abs(ref_scores).max()
print(f"max |score difference| / max |score| on the sample: {d:.3e} (log2 differs in the last place)")
