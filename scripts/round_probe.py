"""Ad-hoc: time single rounds (map / probe M=1 / probe M=3 / window) on one large mapped problem."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, delta, rocco as rr

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 32_000_000
dev = torch.device("cuda:0")
m = synth.hash_matrix_device(8, n, 77, device=dev)
s = torch.empty(n, dtype=torch.float64, device=dev)
rr.score_central_tendency_chrom_device(m, s)
del m
lam = 0.45
def T(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
emap = delta.delta_build_map_device(s, 1.0, lam, 12.0)
r = delta.delta_probe_device(s, 1.0, [lam], emap)
print("count", r[0], "blocks", (n + 8191) // 8192)
print("map   M=1 %.3f ms" % T(lambda: delta.delta_build_map_device(s, 1.0, lam, 12.0)))
print("probe M=1 %.3f ms" % T(lambda: delta.delta_probe_device(s, 1.0, [lam], emap)))
print("probe M=3 %.3f ms" % T(lambda: delta.delta_probe_device(s, 1.0, [lam, lam * 1.001, lam * 0.999], emap)))
print("nomap M=1 %.3f ms" % T(lambda: delta.delta_probe_device(s, 1.0, [lam], None)))
print("nomap M=3 %.3f ms" % T(lambda: delta.delta_probe_device(s, 1.0, [lam, lam * 1.001, lam * 0.999], None)))
print("window    %.3f ms" % T(lambda: delta.delta_window_device(s, 1.0, lam, lam * 1.00001, emap)))
