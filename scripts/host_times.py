"""Ad-hoc: host-side time split of the batched whole-genome solve (ROCCO_HIP_DEBUG output)."""
import os, sys
os.environ["ROCCO_HIP_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = []
for idx, (name, n) in enumerate(genome):
    m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device)
    works.append(pipeline.ChromWork(name, m, 0.02, 1.0, step=50))
for rep in range(3):
    pipeline.solve_rank(works)
    torch.cuda.synchronize()
