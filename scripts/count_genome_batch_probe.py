"""Ad-hoc: whole-genome count-path scoring (score_loci_wls over every chromosome) -- one chromosome after the other
against rocco_amd.inference.score_loci_wls_batch_device.   python scripts/count_genome_batch_probe.py [K] [chroms|all] [workers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, inference

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
names = None if len(sys.argv) <= 2 or sys.argv[2] == "all" else sys.argv[2].split(",")
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
genome = synth.chrom_loci(50, names)
dev = torch.device("cuda:0")
total = sum(n for _, n in genome)


def make():
    return [torch.round(synth.hash_matrix_device(K, n, synth.chrom_seed(7, idx), device=dev) * 20.0) for idx, (_, n) in enumerate(genome)]


import threading
low = [1e30]
def watch():
    while True:
        free, total = torch.cuda.mem_get_info()
        low[0] = min(low[0], free)
        time.sleep(0.01)
threading.Thread(target=watch, daemon=True).start()
mats = make()
inference.score_loci_wls_device(mats[0].clone())  # the Whittaker factor of the longest row, once per process
torch.cuda.synchronize()
for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
    mats = make()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = inference.score_loci_wls_batch_device(mats, overwrite_input=True, workers=workers)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch ({workers} workers): {len(mats)} matrices K={K}, {total} loci: {dt:.3f} s = {total / dt / 1e6:.1f} M loci/s, "
          f"{K * total / dt / 1e9:.2f} G values/s; least free device memory so far {low[0] / 1e9:.1f} GB", flush=True)
    ref = [o[0] for o in out]
    del out
if os.environ.get("PROBE_BATCH_ONLY"):
    sys.exit(0)
mats = make()
torch.cuda.synchronize(); t0 = time.perf_counter()
seq = [inference.score_loci_wls_device(m, overwrite_input=True)[0] for m in mats]
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"one after the other: {dt:.3f} s = {total / dt / 1e6:.1f} M loci/s; same scores: {all(torch.equal(a, b) for a, b in zip(ref, seq))}")
