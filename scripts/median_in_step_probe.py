"""Ad-hoc: what the whole-genome median launch costs INSIDE a step against alone.
Variants of what precedes the launch on the stream, event-bracketed, same process, interleaved:
  alone            -- the launch after a synchronisation, nothing before it
  after_calib      -- a full calibration + decode of the previous step before it (what a step does), no synchronisation between
  after_calib_sync -- the same, host synchronises before launching (the launch starts on an idle device)
  after_dummy      -- a 200 us streaming kernel over 1 GB right in front (memory system busy, pages / clocks warm)
  after_calib_dummy-- calibration + decode, then the 200 us kernel, then the launch"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, dp, pipeline
from rocco_amd import rocco as rr

dev = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
mats = [synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=dev) for idx, (name, n) in enumerate(genome)]
scores = rr.score_central_tendency_chrom_batch_device(mats)
targets = [int(np.floor(s.shape[0] * 0.02)) for s in scores]
big = torch.empty(1 << 27, dtype=torch.float64, device=dev)  # 1 GB


def calib():
    solved = dp.calibrate_batch_device(scores, [1.0] * len(scores), targets)
    rr.decode_runs_table_device([sol for (_p, sol, _v, _c, _i) in solved])


def median_ms(before):
    before()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rr.score_central_tendency_chrom_batch_device(mats)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


variants = {
    "alone": lambda: torch.cuda.synchronize(),
    "after_calib": calib,
    "after_calib_sync": lambda: (calib(), torch.cuda.synchronize()),
    "after_dummy": lambda: (torch.cuda.synchronize(), big.mul_(1.0)),
    "after_calib_dummy": lambda: (calib(), big.mul_(1.0)),
}
for _ in range(3):
    for f in variants.values():
        median_ms(f)
times = {k: [] for k in variants}
for rep in range(12):
    for k, f in variants.items():
        times[k].append(median_ms(f))
for k, v in times.items():
    print(f"{k:18s} mean {np.mean(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}", flush=True)
