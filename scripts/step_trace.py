"""Ad-hoc: one whole-genome step under rocprofv3 (run as: rocprofv3 --kernel-trace ... -- python3 scripts/step_trace.py G)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline
device = torch.device("cuda:0")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 4
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]
for _ in range(3):
    pipeline.solve_rank(works, groups=g)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipeline.solve_rank(works, groups=g)
torch.cuda.synchronize()
print(f"groups={g}: {1e3 * (time.perf_counter() - t0):.2f} ms")
