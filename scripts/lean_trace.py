"""Ad-hoc: round-by-round trace (ROCCO_HIP_DEBUG / ROCCO_SEARCH_DEBUG) and wall time of the whole-genome
solve alone (scores precomputed), one group and four groups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline, dp
from rocco_amd import rocco as rr
device = torch.device("cuda:0")
K = int(os.environ.get("TRACE_K", "100"))
genome = synth.chrom_loci(50, None)
scores, names = [], []
for idx, (name, n) in enumerate(genome):
    m = synth.hash_matrix_device(K, n, synth.chrom_seed(20240, idx), device=device)
    scores.append(rr.score_central_tendency_chrom_device(m))
    names.append(name)
    del m
torch.cuda.synchronize()
targets = [int(np.floor(s.shape[0] * 0.02)) for s in scores]
gammas = [1.0] * len(scores)
which = [int(x) for x in os.environ.get("TRACE_CHROMS", "0").split(",")]
# trace of the selected chromosomes alone
os.environ["ROCCO_HIP_DEBUG"] = "1"
os.environ["ROCCO_SEARCH_DEBUG"] = "1"
dp.calibrate_batch_device([scores[i] for i in which], [gammas[i] for i in which], [targets[i] for i in which])
torch.cuda.synchronize()
del os.environ["ROCCO_HIP_DEBUG"]
del os.environ["ROCCO_SEARCH_DEBUG"]
for label, sel in (("chr1 alone", [0]), ("whole genome, one batch", list(range(len(scores))))):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = dp.calibrate_batch_device([scores[i] for i in sel], [gammas[i] for i in sel], [targets[i] for i in sel])
        torch.cuda.synchronize()
        print(f"{label}: {1e3 * (time.perf_counter() - t0):.3f} ms  passes {[o[4]['passes'] for o in out][:6]}", flush=True)
