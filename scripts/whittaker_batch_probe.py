"""Ad-hoc: the cross-fit Whittaker baseline of a genome's worth of K-row matrices in ONE pair of launches
(rocco_hip_crossfit_whittaker_baseline_batch_f64) -- time, ns per locus of the longest row, a row checked against the
CPU oracle.   python scripts/whittaker_batch_probe.py [K] [chrom,chrom,... | all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
from rocco_amd import synth, inference

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
names = None if len(sys.argv) <= 2 or sys.argv[2] == "all" else sys.argv[2].split(",")
genome = synth.chrom_loci(50, names)
dev = torch.device("cuda:0")
lam = inference._consenrich_whittaker_lambda(101)
mats = []
for idx, (name, n) in enumerate(genome):
    m = synth.hash_matrix_device(K, n, synth.chrom_seed(7, idx), device=dev)
    mats.append(torch.log2(torch.round(m * 20.0) + 1.0))
outs = [torch.empty_like(m) for m in mats]
offsets = [m.median(dim=1).values.contiguous() for m in mats]
total = sum(n for _, n in genome)
longest = max(n for _, n in genome)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get("PROBE_RESIDUAL"):  # (the residual form: offsets subtracted on the way in, the baselines on the way out)
        inference.crossfit_whittaker_residual_batch_device(mats, offsets, lam, outs=outs)
    else:
        inference.crossfit_whittaker_baseline_batch_device(mats, lam, outs=outs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {len(mats)} matrices, K={K}, {total} loci: {dt * 1e3:.1f} ms  = {dt / (2 * longest) * 1e9:.1f} ns per locus and sweep of "
          f"the longest row; {3 * 8 * K * total * 2 / dt / 1e9:.0f} GB/s of 2 sweeps x 24 B per value", flush=True)
import pyoracle as po
i = int(np.argmin([n for _, n in genome]))
rows = [0, K // 2, K - 1]
if os.environ.get("PROBE_RESIDUAL"):
    g = mats[i][rows].cpu().numpy() - offsets[i][rows].cpu().numpy()[:, None]
    want = g - po.crossfit_whittaker_baseline(g, lam)
else:
    want = po.crossfit_whittaker_baseline(mats[i][rows].cpu().numpy(), lam)
print("rows of the shortest chromosome equal the oracle's:", bool(np.array_equal(outs[i][rows].cpu().numpy(), want)))
