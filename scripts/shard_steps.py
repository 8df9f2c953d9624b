"""Ad-hoc (round 5): every rank's owned set at N = 2 / 4 / 8 (LPT over the genome's chromosomes,
rocco_amd.shard.lpt_partition) run ALONE on this one GPU -- medians, calibration, decode, the pipelined step -- and the host
side of the interval gather on tables of that size.  PER-SHARD TIMINGS ON ONE GPU, NOT A SCALING CURVE: no two ranks run at
once, no collective crosses a link.   python scripts/shard_steps.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rocco_amd import synth, pipeline, dp, shard
from rocco_amd import rocco as rr

device = torch.device("cuda:0")
genome = synth.chrom_loci(50, None)
sizes = [n for _, n in genome]
K = 100


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out


print("per-shard timings on ONE GPU (each rank's chromosomes alone), not a scaling curve; ms", flush=True)
whole = None
for N in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    owned = shard.lpt_partition(sizes, N)
    worst = 0.0
    for rank, part in enumerate(owned):
        works = [pipeline.ChromWork(genome[i][0], synth.hash_matrix_device(K, genome[i][1], synth.chrom_seed(20240, i), device=device),
                                    0.02, 1.0, step=50) for i in part]
        t_med, scores = timed(lambda: rr.score_central_tendency_chrom_batch_device([w.matrix_t for w in works]))
        targets = [int(np.floor(w.n * w.budget)) for w in works]
        t_cal, solved = timed(lambda: dp.calibrate_batch_device(scores, [1.0] * len(works), targets))
        t_step, res = timed(lambda: pipeline.solve_rank(works))
        loci = sum(w.n for w in works)
        worst = max(worst, t_step)
        print(f"N={N} rank {rank}: {len(part):2d} chromosomes {loci:9d} loci ({', '.join(genome[i][0] for i in part)}): medians {t_med:6.3f}  "
              f"calibration {t_cal:6.3f} (passes {max(s[4]['passes'] for s in solved)})  step {t_step:6.3f}", flush=True)
        del works, scores, solved, res
        torch.cuda.empty_cache()
    if N == 1:
        whole = worst
    print(f"N={N}: slowest rank's step {worst:.3f} ms" + (f" = {whole / worst:.2f} x the whole genome's step on one GPU ({whole:.3f}); LPT bound {sum(sizes) / shard.makespan(sizes, owned):.2f}" if whole else ""), flush=True)
