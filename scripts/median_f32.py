"""Ad-hoc: median kernel on chr1, K = 100: float64 vs float32 (--low_memory) input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, rocco as rr
n, K = 4979129, 100
m64 = synth.hash_matrix_device(K, n, 7)
m32 = m64.to(torch.float32)
out = torch.empty(n, dtype=torch.float64, device="cuda")
for name, m, bpl in (("float64", m64, 8 * K + 8), ("float32", m32, 4 * K + 8)):
    for _ in range(5): rr.score_central_tendency_chrom_device(m, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record(); rr.score_central_tendency_chrom_device(m, out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); t = ts[len(ts) // 2]
    print(f"{name}: {t*1e3:.1f} us per launch, {bpl*n/t/1e9:.2f} TB/s of algorithmic traffic, {n/t/1e6:.2f} G loci/s")
ref = torch.median(m32[:, :100000].to(torch.float64), dim=0)
