"""Ad-hoc A/B timing of whole-genome steps in ONE process: configs are env-var settings read by the
library at each solve; steps alternate between configs so clock drift hits all equally."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline

configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in sys.argv[1:]] or [{}]
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = []
for idx, (name, n) in enumerate(genome):
    m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
    works.append(pipeline.ChromWork(name, m, 0.02, 1.0, step=50))
torch.cuda.synchronize()
times = [[] for _ in configs]
keys = sorted({k for c in configs for k in c})
for rep in range(7):
    for ci, c in enumerate(configs):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipeline.solve_rank(works)
        torch.cuda.synchronize(); times[ci].append((time.perf_counter() - t0) * 1e3)
for c, t in zip(configs, times):
    t = t[1:]
    print(c, "min %.2f med %.2f max %.2f ms" % (min(t), statistics.median(t), max(t)))
