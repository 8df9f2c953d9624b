"""Ad-hoc: whole-genome step time against the number of chromosome groups calibrating side by side."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import numpy as np, torch
from rocco_amd import synth, pipeline
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]
ref = None
settings = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,6,8").split(",")]
for g in settings:  # warm-up: solvers, streams, scratch buffers
    pipeline.solve_rank(works, groups=g)
torch.cuda.synchronize(); gc.collect(); gc.freeze()
times = {g: [] for g in settings}
for rep in range(8):
    for g in settings:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = pipeline.solve_rank(works, groups=g)
        torch.cuda.synchronize(); times[g].append((time.perf_counter() - t0) * 1e3)
        sig = [(r["selection_penalty"], r["selected_count"], int(r["begin"].numel())) for r in res]
        if ref is None: ref = sig
        assert sig == ref, f"groups={g}: results differ"
for g in settings:
    print(f"groups={g}: median {statistics.median(times[g]):.2f} ms  min {min(times[g]):.2f} ms")
