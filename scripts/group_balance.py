"""Ad-hoc: how long each chromosome group of a whole-genome solve_rank takes (critical path = the slowest group)."""
import gc, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx)), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]
orig = pipeline._solve_group
log = []
def timed(chroms, scores):
    t0 = time.perf_counter()
    out = orig(chroms, scores)
    torch.cuda.current_stream().synchronize()
    log.append((time.perf_counter() - t0, [c.name for c in chroms], [r["path"] for r in out], sum(c.n for c in chroms)))
    return out
pipeline._solve_group = timed
for rep in range(3):
    pipeline.solve_rank(works, groups=4)
gc.collect(); gc.freeze()
totals, per_group = [], {}
for rep in range(10):
    log.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipeline.solve_rank(works, groups=4)
    torch.cuda.synchronize(); totals.append(time.perf_counter() - t0)
    for t, names, paths, loci in log:
        per_group.setdefault(tuple(names), ([], paths, loci))[0].append(t)
print(f"step median {statistics.median(totals)*1e3:.2f} ms")
for names, (ts, paths, loci) in sorted(per_group.items(), key=lambda kv: -statistics.median(kv[1][0])):
    print(f"  group median {statistics.median(ts)*1e3:6.2f} ms  loci {loci:9d}  spine chromosomes {[n for n, p in zip(names, paths) if p == 4]}  {list(names)}")
