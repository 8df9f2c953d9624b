"""Ad-hoc: per-step wall time of the whole-genome step, to spot outliers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = []
for idx, (name, n) in enumerate(genome):
    m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
    works.append(pipeline.ChromWork(name, m, 0.02, 1.0, step=50))
torch.cuda.synchronize()
ts = []
import gc
if os.environ.get("NOGC"): gc.disable()
for rep in range(40):
    t0 = time.perf_counter()
    res = pipeline.solve_rank(works)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts))
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_reserved() / 2**30)
