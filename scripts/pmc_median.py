"""Workload for the PMC passes of the roofline `traffic` figure (profiles/README.md): one calibration
launch with a known byte count in the same access pattern (8 B per lane, row-coalesced: the K = 1 copy
kernel over 100 M loci = 800 MB read, 800 MB written, past the 256 MiB Infinity Cache), then the median
launches of one benchmark step at N = 1 (the whole genome, K = 100, one launch per group of chromosomes, the
groups of rocco_amd/pipeline.py), twice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, pipeline, rocco as rr

dev = torch.device("cuda:0")
n_cal = 100_000_000
cal = torch.rand(1, n_cal, dtype=torch.float64, device=dev)
out = torch.empty(n_cal, dtype=torch.float64, device=dev)
rr.score_central_tendency_chrom_device(cal, out)
torch.cuda.synchronize()
del cal, out
genome = synth.chrom_loci(50, None)
K = 100
mats = [synth.hash_matrix_device(K, n, synth.chrom_seed(20240, idx), device=dev) for idx, (name, n) in enumerate(genome)]
order = sorted(range(len(mats)), key=lambda i: -int(mats[i].shape[1]))
groups = pipeline._group_chunks(order, pipeline.SOLVE_GROUPS)
for _ in range(2):
    for g in groups:
        rr.score_central_tendency_chrom_batch_device([mats[i] for i in g])
torch.cuda.synchronize()
print("calibration bytes read", n_cal * 8, "written", n_cal * 8)
for g in groups:
    loci = sum(int(mats[i].shape[1]) for i in g)
    print("median batch K=%d loci=%d algorithmic bytes read %d written %d" % (K, loci, K * loci * 8, loci * 8))
