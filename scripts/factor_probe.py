"""Ad-hoc: time of the first cross-fit Whittaker call of a process for a long row (the factor of that length is built
inside it) and of the second (factor cached), and the check against the oracle on a shorter one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd.inference import crossfit_whittaker_baseline_batch_device, _consenrich_whittaker_lambda
lam = _consenrich_whittaker_lambda(101)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_979_000
m = torch.randn((2, n), dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
for label in ("first call (factor built)", "second call"):
    t0 = time.perf_counter()
    out = crossfit_whittaker_baseline_batch_device([m], lam)
    torch.cuda.synchronize()
    print(f"{label}: {1e3 * (time.perf_counter() - t0):.1f} ms for 2 x {n}, lambda {lam:g}")
