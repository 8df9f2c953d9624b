"""Ad-hoc: does the whole-genome median launch run slower after a few ms of light work / idling (as inside a step)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rocco_amd import synth, rocco as rr
dev = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
mats = [synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=dev) for idx, (name, n) in enumerate(genome)]
small = torch.zeros(1 << 20, device=dev)


def run(label, between):
    ts = []
    for rep in range(12):
        between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rr.score_central_tendency_chrom_batch_device(mats)
        e1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            ts.append(e0.elapsed_time(e1))
    print(f"{label}: {sum(ts) / len(ts):.3f} ms (min {min(ts):.3f}, max {max(ts):.3f})", flush=True)


run("back to back", lambda: None)
run("4 ms idle before", lambda: time.sleep(0.004))
run("20 ms idle before", lambda: time.sleep(0.020))


def light():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.004:
        small.add_(1.0)
        torch.cuda.synchronize()


run("4 ms of tiny kernels with synchronisations before", light)
run("back to back again", lambda: None)
