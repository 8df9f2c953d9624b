"""Ad-hoc (run under rocprofv3 --kernel-trace --stats): lean_model_kernel on one array of T tiles with P penalties per call,
for several T and P -- how its duration depends on the workgroups in flight and on what each carries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import delta

rng = np.random.default_rng(7)
for T in (8, 18, 64, 128, 256, 420):
    n = 8192 * T
    s = np.round(rng.gamma(1.0, 0.3, n), 5)
    at = rng.integers(0, n, max(1, n // 40))
    s[at] += np.round(rng.gamma(6.0, 1.0, at.size), 5)
    s_t = torch.from_numpy(s).cuda()
    lam_ref = float(np.quantile(s, 0.9))
    margin = 1.0 + float(np.ptp(s)) + float(np.max(np.abs(s))) + 4.0
    emap = delta.delta_build_map_device(s_t, 1.0, lam_ref, margin)
    for P in (1, 2, 4, 7, 8):
        lams = list(lam_ref + 1e-3 * rng.uniform(-1, 1, P))
        for _ in range(5):
            delta.delta_model_lean_device(s_t, 1.0, lams, emap)
        torch.cuda.synchronize()
        print(f"T={T} P={P} done", flush=True)
