"""Ad-hoc: one whole-genome step in sequence (one group) with device events between the phases and host clocks
around them: medians | calibration | decode | interval transfer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline, dp
from rocco_amd import rocco as rr
device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, n, synth.chrom_seed(20240, idx), device=device), 0.02, 1.0, step=50)
         for idx, (name, n) in enumerate(genome)]
targets = [int(np.floor(w.n * w.budget)) for w in works]
gammas = [w.gamma for w in works]
caps = [max(1024, w.n // 64) for w in works]
acc = np.zeros(6)
reps = 20
for rep in range(reps + 5):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    h0 = time.perf_counter()
    ev[0].record()
    scores = rr.score_central_tendency_chrom_batch_device([w.matrix_t for w in works])
    ev[1].record()
    h1 = time.perf_counter()
    solved = dp.calibrate_batch_device(scores, gammas, targets)
    ev[2].record()
    h2 = time.perf_counter()
    if os.environ.get("STEP_PHASES_OLD_DECODE"):
        runs = rr.decode_runs_batch_device([s[1] for s in solved], capacities=caps)
        ev[3].record()
        flat = torch.cat([torch.stack([b, e], dim=1) for b, e in runs if b.numel()]).cpu().numpy()
    else:  # round 3: one table of (unit, begin, end) rows, in pinned host memory when the call returns
        selected = sum(int(s[3]) for s in solved)
        table_t, offsets, flat = rr.decode_runs_table_device([s[1] for s in solved], capacity_rows=max(1024, selected // 2 + 64),
                                                             eager_rows=max(4096, selected // 5))
        ev[3].record()
    ev[4].record()
    torch.cuda.synchronize()
    h3 = time.perf_counter()
    if rep >= 5:
        acc += np.array([ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), ev[3].elapsed_time(ev[4]),
                         1e3 * (h3 - h0), 1e3 * (h2 - h1)])
acc /= reps
print(f"device: medians {acc[0]:.3f} ms, calibration {acc[1]:.3f} ms, decode {acc[2]:.3f} ms, interval transfer {acc[3]:.3f} ms; "
      f"host: whole step {acc[4]:.3f} ms, inside calibrate_batch_device {acc[5]:.3f} ms")
