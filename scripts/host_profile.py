"""Ad-hoc: where the host time of a whole-genome step goes (cProfile over 20 steps of the bench's step function)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]


def one_step():
    res = pipeline.solve_rank(works, groups=int(os.environ.get("PROFILE_GROUPS", "1")))
    flat = torch.cat([torch.stack([r["begin"], r["end"]], dim=1) for r in res if r["begin"].numel()]).cpu().numpy()
    return flat


for _ in range(5):
    one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    one_step()
torch.cuda.synchronize()
print(f"step {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    one_step()
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
print(out.getvalue())
