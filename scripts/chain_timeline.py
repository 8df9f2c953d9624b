"""Ad-hoc: one chained calibration for a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 scripts/chain_timeline.py chr1,chr15,chr21
then  python scripts/kernel_timeline.py OUT/.../*_kernel_trace.csv stats_partial 200  on the LAST calibration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, dp
from rocco_amd import rocco as rr
device = torch.device("cuda:0")
arg = sys.argv[1] if len(sys.argv) > 1 else "all"
genome = synth.chrom_loci(50, None)
names = [name for name, _n in genome] if arg == "all" else arg.split(",")
index = {name: i for i, (name, _n) in enumerate(genome)}
scores = []
for n in names:
    m = synth.hash_matrix_device(100, genome[index[n]][1], synth.chrom_seed(20240, index[n]), device=device)
    scores.append(rr.score_central_tendency_chrom_batch_device([m])[0])
    del m
targets = [int(np.floor(s.shape[0] * 0.02)) for s in scores]
for _ in range(4):
    dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
torch.cuda.synchronize()
time.sleep(0.05)
t0 = time.perf_counter()
out = dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
torch.cuda.synchronize()
print(f"calibrate {len(names)} chromosomes: {1e3 * (time.perf_counter() - t0):.3f} ms, passes {[o[4]['passes'] for o in out]}")
