"""Ad-hoc: where the host's time goes in a step of the headline (pipeline.solve_rank), cProfile over N steps."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import pipeline, synth

device = torch.device("cuda:0")
names = os.environ.get("AB_CHROMS")
genome = synth.chrom_loci(50, names.split(",") if names else None)
works = []
for idx, (name, n) in enumerate(genome):
    m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
    works.append(pipeline.ChromWork(name, m, 0.02, 1.0, step=50))
units = list(range(len(works)))
for _ in range(5):
    pipeline.solve_rank(works, units=units)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    pipeline.solve_rank(works, units=units)
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / N:.3f} ms per step")
prof = cProfile.Profile()
prof.enable()
for _ in range(N):
    pipeline.solve_rank(works, units=units)
torch.cuda.synchronize()
prof.disable()
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(22)
