"""Ad-hoc: where one rank's step goes on a small shard (the chromosomes rank 0 owns at N = 8 by default): each
phase alone between device synchronisations, then the pipelined step.
    python scripts/shard_phases.py chr1,chr15,chr21"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline, dp
from rocco_amd import rocco as rr
device = torch.device("cuda:0")
names = (sys.argv[1] if len(sys.argv) > 1 else "chr1,chr15,chr21").split(",")
genome = synth.chrom_loci(50, None)
index = {name: i for i, (name, _n) in enumerate(genome)}
works = [pipeline.ChromWork(name, synth.hash_matrix_device(100, genome[index[name]][1], synth.chrom_seed(20240, index[name]), device=device),
                            0.02, 1.0, step=50) for name in names]
torch.cuda.synchronize()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out


t_med, scores = timed(lambda: rr.score_central_tendency_chrom_batch_device([w.matrix_t for w in works]))
targets = [int(np.floor(w.n * w.budget)) for w in works]
gammas = [w.gamma for w in works]
t_cal, solved = timed(lambda: dp.calibrate_batch_device(scores, gammas, targets))
t_cal1, _ = timed(lambda: dp.calibrate_batch_device(scores[:1], gammas[:1], targets[:1]))
t_dec, runs = timed(lambda: rr.decode_runs_batch_device([s[1] for s in solved], capacities=[max(1024, w.n // 64) for w in works]))
print(f"{names}: medians {t_med:.3f} ms, calibrate (one batch) {t_cal:.3f} ms [first chromosome alone {t_cal1:.3f}], "
      f"decode {t_dec:.3f} ms, passes {[s[4]['passes'] for s in solved]}")
for g in (1, 3):
    t, res = timed(lambda: pipeline.solve_rank(works, groups=g))
    t_host, _ = timed(lambda: torch.cat([torch.stack([r["begin"], r["end"]], dim=1) for r in res if r["begin"].numel()]).cpu().numpy())
    print(f"solve_rank groups={g}: {t:.3f} ms   (host transfer of the intervals afterwards: {t_host:.3f} ms)")
