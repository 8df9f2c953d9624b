"""Ad-hoc: wall time of the phases of a one-group step with a synchronisation after each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline, dp
from rocco_amd import rocco as rr
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]
mats = [w.matrix_t for w in works]
targets = [int(np.floor(w.n * w.budget)) for w in works]
gammas = [1.0] * len(works)
def sync(): torch.cuda.synchronize()
for rep in range(6):
    sync(); t0 = time.perf_counter()
    scores = rr.score_central_tendency_chrom_batch_device(mats); sync(); t1 = time.perf_counter()
    solved = dp.calibrate_batch_device(scores, gammas, targets); sync(); t2 = time.perf_counter()
    runs = rr.decode_runs_batch_device([s[1] for s in solved], capacities=[max(1024, w.n // 64) for w in works]); sync(); t3 = time.perf_counter()
    print(f"median {1e3*(t1-t0):.3f}  solve {1e3*(t2-t1):.3f}  decode {1e3*(t3-t2):.3f}  total {1e3*(t3-t0):.3f} ms")
for rep in range(4):
    sync(); t0 = time.perf_counter()
    pipeline.solve_rank(works, groups=1); sync(); t1 = time.perf_counter()
    print(f"solve_rank(groups=1) {1e3*(t1-t0):.3f} ms")
