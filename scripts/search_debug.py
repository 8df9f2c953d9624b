"""Ad-hoc: print the threshold-search trace (ROCCO_SEARCH_DEBUG=1) of one chromosome."""
import os, sys
os.environ["ROCCO_SEARCH_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline
name = sys.argv[1] if len(sys.argv) > 1 else "chr3"
genome = dict(synth.chrom_loci(50, None))
idx = [n for n, _ in synth.chrom_loci(50, None)].index(name)
w = pipeline.ChromWork(name, synth.hash_matrix_device(100, genome[name], synth.chrom_seed(20240, idx)), 0.02, 1.0, step=50)
res = pipeline.solve_rank([w], groups=1)
print(res[0]["selection_penalty"], res[0]["info"])
