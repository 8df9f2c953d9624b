"""Ad-hoc: one rank's workload (N-GPU LPT shard `rank`) solved a few times, for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline, shard
N, rank, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
owned = shard.lpt_partition([n for _, n in genome], N)
works = []
for idx in owned[rank]:
    name, n = genome[idx]
    works.append(pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device), 0.02, 1.0, step=50))
for rep in range(reps):
    pipeline.solve_rank(works); torch.cuda.synchronize()
