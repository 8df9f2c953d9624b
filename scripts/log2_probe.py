"""Ad-hoc: the log-scale kernel alone (rocco_hip_log_scale_f64) on 2.5e8 values of three kinds: small integer counts,
scaled counts (fractions), wide-range positive values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, inference
dev = torch.device("cuda:0")
u = synth.hash_matrix_device(50, 5_000_000, 99, device=dev)
kinds = {"integer counts 0..20": torch.round(u * 20.0), "scaled counts": u * 37.123, "wide range": torch.exp((u - 0.3) * 40.0)}
for name, x in kinds.items():
    x = x.reshape(-1).contiguous()
    inference.log_scale_device(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = inference.log_scale_device(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:22s}: {dt * 1e3:7.2f} ms for {x.numel()} values = {x.numel() / dt / 1e9:6.1f} G values/s; checksum {float(y[:100000].sum()):.10g}", flush=True)
