"""Ad-hoc: step time with every chromosome scored in one launch before the groups solve (ROCCO_SCORE_FIRST=1) against
group-by-group scoring, for a few group counts, alternating."""
import os, sys, time, statistics, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]
settings = [(sf, g) for sf in (0, 1) for g in (2, 3, 4, 6)]
for sf, g in settings:
    pipeline.SCORE_FIRST = sf
    pipeline.solve_rank(works, groups=g)
torch.cuda.synchronize(); gc.collect(); gc.freeze()
times = {s: [] for s in settings}
for rep in range(8):
    for sf, g in settings:
        pipeline.SCORE_FIRST = sf
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipeline.solve_rank(works, groups=g)
        torch.cuda.synchronize(); times[(sf, g)].append((time.perf_counter() - t0) * 1e3)
for sf, g in settings:
    print(f"score_first={sf} groups={g}: median {statistics.median(times[(sf, g)]):.2f} ms  min {min(times[(sf, g)]):.2f} ms")
