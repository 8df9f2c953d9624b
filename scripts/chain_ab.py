"""Ad-hoc: several settings of the chained search interleaved in one process (same box, same clocks): medians of N reps.
    python scripts/chain_ab.py "A=;B=ROCCO_HIP_CHAIN_PILOT_WGS=1024;C=ROCCO_HIP_CHAIN_SOFT=0.9,ROCCO_HIP_CHAIN_PILOT_WGS=1024" [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, dp
from rocco_amd import rocco as rr

device = torch.device("cuda:0")
configs = {}
for item in sys.argv[1].split(";"):
    name, _, envs = item.partition("=")
    configs[name] = dict(kv.split("=", 1) for kv in envs.split(",") if "=" in kv)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
genome = synth.chrom_loci(50, None)
only = os.environ.get("AB_CHROMS")  # e.g. "chr1,chr9,chr17" (a rank's shard); default: the whole genome
scores = []
for idx, (name, n) in enumerate(genome):
    if only and name not in only.split(","):
        continue
    m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
    scores.append(rr.score_central_tendency_chrom_batch_device([m])[0])
    del m
targets = [int(np.floor(s.shape[0] * 0.02)) for s in scores]
keys = sorted({k for c in configs.values() for k in c})


def run(env):
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


times = {k: [] for k in configs}
for rep in range(reps + 3):
    for k, env in configs.items():
        t = run(env)
        if rep >= 3:
            times[k].append(t)
for k, v in times.items():
    print(f"{k:10s} median {np.median(v):.3f} ms  min {min(v):.3f}  {configs[k]}", flush=True)
