"""Ad-hoc (round 5): one whole-genome calibration batch with the chained rounds as three launches / one launch each
(ROCCO_HIP_CHAIN_FUSED), results compared.   python scripts/fused_probe.py [chroms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, dp, rocco as rr
names = None if len(sys.argv) < 2 or sys.argv[1] == "all" else sys.argv[1].split(",")
genome = synth.chrom_loci(50, names)
dev = torch.device("cuda:0")
scores = [rr.score_central_tendency_chrom_device(synth.hash_matrix_device(10, n, synth.chrom_seed(20240, i), device=dev)) for i, (_, n) in enumerate(genome)]
targets = [int(np.floor(n * 0.02)) for _, n in genome]
ref = None
for mode in os.environ.get("MODES", "0,3,2,1").split(","):
    os.environ["ROCCO_HIP_CHAIN_FUSED"] = mode
    print(f"mode {mode} ...", flush=True)
    times = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    key = [(o[0], o[3], o[4]["path"], o[4]["passes"]) for o in out]
    sols = [o[1].clone() for o in out]
    if ref is None:
        ref = (key, sols)
    same = key == ref[0] and all(torch.equal(a, b) for a, b in zip(sols, ref[1]))
    print(f"mode {mode}: {1e3 * np.median(times[1:]):.3f} ms (calls {[round(1e3 * t, 3) for t in times]}), same results as mode {os.environ.get('MODES', '0')[0]}: {same}, paths {sorted(set(k[2] for k in key))}", flush=True)
