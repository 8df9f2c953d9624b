"""Ad-hoc: throughput of the bootstrap multipliers made on the device against NumPy / SciPy on the host cores.
One draw of a K = 100 chromosome-1-sized matrix is 100 rows x (4 979 129 + 2 x 101) normals."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import budget

n, K, hint = 4979129, 100, 101
taps = budget._build_budget_bootstrap_kernel(budget._resolve_budget_bootstrap_bandwidth(n, hint))
count = K * (n + taps.size - 1)
rng = np.random.default_rng(1)
budget.device_standard_normal(rng, 1 << 20)
torch.cuda.synchronize()
for what in ("normals", "multipliers"):
    ts = []
    for rep in range(4):
        rng = np.random.default_rng(100 + rep)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = budget.device_standard_normal(rng, count) if what == "normals" else budget.device_multipliers(rng, K, n, taps)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        del out
    print(f"device {what}: {count} values in {min(ts) * 1e3:.1f} ms = {count / min(ts) / 1e9:.2f} G values/s (best of {len(ts)})", flush=True)
rows = 4
t0 = time.perf_counter()
rng = np.random.default_rng(100)
ref = rng.standard_normal(rows * (n + taps.size - 1))
t_norm = time.perf_counter() - t0
rng = np.random.default_rng(100)
t0 = time.perf_counter()
for _ in range(rows):
    budget._generate_dependent_wild_weights(n, taps, rng)
t_mult = time.perf_counter() - t0
print(f"host (one core), {rows} of the {K} rows: normals {t_norm:.2f} s = {rows * n / t_norm / 1e6:.1f} M values/s; "
      f"multipliers (normals + SciPy FFT smoothing + scaling) {t_mult:.2f} s = {rows * n / t_mult / 1e6:.1f} M values/s", flush=True)
