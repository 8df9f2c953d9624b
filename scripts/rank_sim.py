"""Ad-hoc: step time of the chromosomes one rank of an N-GPU run owns (LPT partition), on one GPU."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline, shard
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
owned = shard.lpt_partition([n for _, n in genome], N)
for rank in range(min(N, 3)):
    works = []
    for idx in owned[rank]:
        name, n = genome[idx]
        m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
        works.append(pipeline.ChromWork(name, m, 0.02, 1.0, step=50))
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipeline.solve_rank(works); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"N={N} rank {rank}: {[genome[i][0] for i in owned[rank]]} loci {sum(genome[i][1] for i in owned[rank])}: med {statistics.median(ts[1:]):.2f} ms")
    del works
