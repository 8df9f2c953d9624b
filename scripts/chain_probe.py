"""Ad-hoc: the calibration of a set of chromosomes (scores precomputed) with the threshold search sequenced by the host
(ROCCO_HIP_CHAIN=0) and by the device (chain.hip), same process, alternating; results must be identical.
    python scripts/chain_probe.py [chr1,chr15,chr21 | all] [reps]
Environment of the chained runs can be varied with CHAIN_ENV="ROCCO_HIP_CHAIN_LEVELS=2.2;ROCCO_HIP_CHAIN_PILOT_ROUNDS=3"."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, dp
from rocco_amd import rocco as rr

device = torch.device("cuda:0")
arg = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
genome = synth.chrom_loci(50, None)
names = [name for name, _n in genome] if arg == "all" else arg.split(",")
index = {name: i for i, (name, _n) in enumerate(genome)}
K = int(os.environ.get("PROBE_K", "100"))
scores = []
for n in names:
    m = synth.hash_matrix_device(K, genome[index[n]][1], synth.chrom_seed(20240, index[n]), device=device)
    scores.append(rr.score_central_tendency_chrom_batch_device([m])[0])
    del m
torch.cuda.synchronize()
budget = float(os.environ.get("PROBE_BUDGET", "0.02"))
targets = [int(np.floor(s.shape[0] * budget)) for s in scores]
gammas = [1.0] * len(scores)


def run(env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = dp.calibrate_batch_device(scores, gammas, targets)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0), out
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


chain_env = {"ROCCO_HIP_CHAIN": "1"}
for kv in os.environ.get("CHAIN_ENV", "").split(";"):
    if "=" in kv:
        k, v = kv.split("=", 1)
        chain_env[k] = v
host_env = {"ROCCO_HIP_CHAIN": "0"}
for _ in range(3):
    run(host_env)
    run(chain_env)
th, tc = [], []
ref = None
for r in range(reps):
    t, oh = run(host_env)
    th.append(t)
    t, oc = run(chain_env)
    tc.append(t)
    for a, b_ in zip(oh, oc):
        assert a[0] == b_[0], (a[0], b_[0])
        assert a[3] == b_[3], (a[3], b_[3])
        assert torch.equal(a[1], b_[1])
        assert b_[4]["path"] == a[4]["path"], (a[4]["path"], b_[4]["path"])
print(f"{len(names)} chromosomes, {sum(int(s.shape[0]) for s in scores)} loci, K={K}")
print(f"host-sequenced search : min {min(th):.3f} median {np.median(th):.3f} ms   passes {[o[4]['passes'] for o in oh]}")
print(f"device-chained search : min {min(tc):.3f} median {np.median(tc):.3f} ms   passes {[o[4]['passes'] for o in oc]}")
print("paths", sorted(set(o[4]["path"] for o in oc)), "results identical")
if os.environ.get("PROBE_DEBUG"):
    os.environ["ROCCO_HIP_DEBUG"] = "1"
    os.environ["ROCCO_HIP_TIMING"] = "1"
    run(chain_env)
