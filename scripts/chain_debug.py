"""Ad-hoc: one chained calibration with the director's records printed (ROCCO_HIP_CHAIN_DEBUG)."""
import os, sys
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
print("targets", targets, flush=True)
os.environ["ROCCO_HIP_CHAIN_DEBUG"] = "1"
os.environ["ROCCO_HIP_DEBUG"] = "1"
os.environ["ROCCO_SEARCH_DEBUG"] = "1"
out = dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
print([(o[0], o[3], o[4]["passes"]) for o in out])
