"""Ad-hoc: log scale + row medians + centring of one 100 x 5M matrix (rocco_hip_log_scale_center_rows_f64), rows of small integer
counts (the median is a run of equal values: seven passes) against rows of scaled counts (settled from the gathered cell after
two passes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, inference
dev = torch.device("cuda:0")
u = synth.hash_matrix_device(100, 4_979_129, 5, device=dev)
for name, x in {"integer counts 0..20": torch.round(u * 20.0), "scaled counts": u * 37.123}.items():
    inference.log_scale_center_rows_device(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        c, off = inference.log_scale_center_rows_device(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:22s}: {dt * 1e3:6.2f} ms per call; first medians {off[:3].tolist()}", flush=True)
