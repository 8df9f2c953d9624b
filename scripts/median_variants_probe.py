"""Where the production median launch loses against scripts/ubench/median_lab's copy of the same kernel: 24 allocations
vs one, the data, the launch path.  Times with events on the launching stream, 8 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, rocco as rr

dev = torch.device("cuda:0")
K = 100
genome = synth.chrom_loci(50, None)
total = sum(n for _, n in genome)


def timed(label, fn, reps=8):
    fn(); fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    print(f"{label:58s} best {ms[0]:.3f}  median {ms[len(ms)//2]:.3f}  worst {ms[-1]:.3f} ms   {(8*K+8)*total/ms[0]/1e6:.0f} GB/s", flush=True)


mats = [synth.hash_matrix_device(K, n, synth.chrom_seed(20240, i), device=dev) for i, (_, n) in enumerate(genome)]
timed("24 chromosomes, 24 allocations, benchmark data", lambda: rr.score_central_tendency_chrom_batch_device(mats))
order = sorted(range(len(mats)), key=lambda i: -mats[i].shape[1])
timed("the same, longest first", lambda: rr.score_central_tendency_chrom_batch_device([mats[i] for i in order]))
for m in mats:
    m.uniform_(0.0, 1.0)
timed("24 chromosomes, 24 allocations, uniform data", lambda: rr.score_central_tendency_chrom_batch_device(mats))
del mats
torch.cuda.empty_cache()
big = torch.rand((K, total), dtype=torch.float64, device=dev)
out = torch.empty(total, dtype=torch.float64, device=dev)
timed("one 100 x 61.77M matrix (one task), uniform data", lambda: rr.score_central_tendency_chrom_batch_device([big]))
timed("the same through the single-matrix entry", lambda: rr.score_central_tendency_chrom_device(big, out))
views, at = [], 0
for _, n in genome:
    views.append(big[:, at:at + n])
    at += n
timed("24 tasks that are column ranges of that one matrix", lambda: rr.score_central_tendency_chrom_batch_device(views))
del big, views
torch.cuda.empty_cache()
# 24 allocations again, every row padded to a multiple of 16 columns (128 bytes): row starts on cache-line boundaries
padded = []
for i, (_, n) in enumerate(genome):
    stride = (n + 15) // 16 * 16
    buf = torch.rand((K, stride), dtype=torch.float64, device=dev)
    padded.append(buf[:, :n])
timed("24 allocations, rows padded to 128-byte multiples", lambda: rr.score_central_tendency_chrom_batch_device(padded))
