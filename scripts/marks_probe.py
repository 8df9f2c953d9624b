"""Ad-hoc: host-side event log (ROCCO_HIP_TIMING=2) of one genome calibration after warm-up.
    python scripts/marks_probe.py [ENV=VALUE ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
import numpy as np, torch
from rocco_amd import synth, dp
from rocco_amd import rocco as rr

device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
scores = []
only = os.environ.get("AB_CHROMS")
for idx, (name, n) in enumerate(genome):
    if only and name not in only.split(","):
        continue
    m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
    scores.append(rr.score_central_tendency_chrom_batch_device([m])[0])
    del m
targets = [int(np.floor(s.shape[0] * 0.02)) for s in scores]
for _ in range(6):
    dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
torch.cuda.synchronize()
for rep in range(3):
    os.environ["ROCCO_HIP_TIMING"] = "2"
    sys.stderr.write(f"---- rep {rep}\n")
    dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
    torch.cuda.synchronize()
    os.environ.pop("ROCCO_HIP_TIMING")
    dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
