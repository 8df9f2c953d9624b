"""Ad-hoc: wall time of the three phases of a whole-genome step (scoring, batched solve, decode)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline, dp as _dp, rocco as _rocco
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = []
for idx, (name, n) in enumerate(genome):
    m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
    works.append(pipeline.ChromWork(name, m, 0.02, 1.0, step=50))
torch.cuda.synchronize()
T = {"median": [], "solve": [], "decode": [], "to_host": []}
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    scores = [_rocco.score_central_tendency_chrom_device(c.matrix_t) for c in works]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    targets = [int(np.floor(c.n * c.budget)) for c in works]
    solved = _dp.calibrate_batch_device(scores, [c.gamma for c in works], targets)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    runs = [_rocco.decode_runs_device(s[1], capacity=max(1024, c.n // 64)) for c, s in zip(works, solved)]
    torch.cuda.synchronize(); t3 = time.perf_counter()
    host = [torch.stack([b, e], dim=1).cpu().numpy() for b, e in runs]
    t4 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        T[k].append(v * 1e3)
for k, v in T.items():
    print(k, "med %.2f ms" % statistics.median(v[1:]))
