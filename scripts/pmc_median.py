"""Workload for the PMC passes of the roofline `traffic` figure (profiles/README.md): one calibration
launch with a known byte count in the same access pattern (8 B per lane, row-coalesced: the K = 1 copy
kernel over 100 M loci = 800 MB read, 800 MB written, past the 256 MiB Infinity Cache), then the K = 100
median kernel on the benchmark's largest chromosome (chr1), three launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, rocco as rr

dev = torch.device("cuda:0")
n_cal = 100_000_000
cal = torch.rand(1, n_cal, dtype=torch.float64, device=dev)
out = torch.empty(n_cal, dtype=torch.float64, device=dev)
rr.score_central_tendency_chrom_device(cal, out)
torch.cuda.synchronize()
del cal, out
genome = synth.chrom_loci(50, None)
n = genome[0][1]
m = synth.hash_matrix_device(100, n, synth.chrom_seed(20240, 0), device=dev)
out = torch.empty(n, dtype=torch.float64, device=dev)
for _ in range(3):
    rr.score_central_tendency_chrom_device(m, out)
torch.cuda.synchronize()
print("calibration bytes read", n_cal * 8, "written", n_cal * 8)
print("median K=100 n=%d algorithmic bytes read %d written %d" % (n, 100 * n * 8, n * 8))
