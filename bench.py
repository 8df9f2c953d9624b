#!/usr/bin/env python3
"""bench.py -- loci/sec of the per-chromosome solve path on MI355X.

Metric (BASELINE.json): loci/sec to converged solve, hg38, 50 bp bins, K = 100 samples; BED3
intervals bit-exact.  One "step" = one pass of the hot path over the whole genome held by the job:
K x n median scoring of every chromosome, the budgeted chain solve (the reference's 2 + 60 chain
evaluations per chromosome, rocco/dp.py:89-164) and the run-length decode to merged intervals.
Inputs (synthetic K x n matrices, rocco_amd/synth.py) are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 starts the N ranks itself (python -m torch.distributed.run, one process per GPU,
before anything in this process touches the GPU) and relays rank 0's JSON line; under a launcher
(WORLD_SIZE set) it runs as one rank.  N > 1: the 24 chromosomes of ONE genome are assigned to ranks by
LPT on their locus counts (strong scaling, no data-path collective); the interval lists are gathered to
every rank with two small all_gathers over RCCL inside the timed region.  Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=100, help="K")
    ap.add_argument("--step-bp", type=int, default=50)
    ap.add_argument("--budget", type=float, default=0.02)
    ap.add_argument("--gamma", type=float, default=1.0)
    ap.add_argument("--chroms", type=str, default="", help="comma list (default: whole genome)")
    ap.add_argument("--seed", type=int, default=20240)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-composed", action="store_true", help="skip the composed-driver leg (next_rows.composed_driver)")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed steps and their roofline block only (for profiler runs: the kernel statistics then "
                         "hold the steps' launches and nothing else)")
    ap.add_argument("--cpu-sample", type=str, default="chr1")
    ap.add_argument("--rehearse", action="store_true",
                    help="no device work: every rank fabricates its interval lists and runs the launch, partition, "
                         "gather, timing and reporting code only (CPU test of the N > 1 path, backend gloo)")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """Start `args.gpus` ranks as fresh processes and relay rank 0's line.  Nothing here touches the GPU."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for text in proc.stdout.splitlines():
        if text.startswith("{") and '"metric"' in text:
            line = text
        else:
            print(text, file=sys.stderr)
    if line is not None:
        print(line)
    return proc.returncode if line is not None or proc.returncode != 0 else 1


def _cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for text in fh:
                if text.lower().startswith("model name"):
                    model = text.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, usable, os.cpu_count() or usable


def _cpu_solve_worker(job):
    """One chromosome's budgeted solve on the CPU restatement of the reference (oracle/): a process-pool worker of
    the cpu_baseline leg (no torch, no GPU in these processes)."""
    path, budget, gamma = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po

    s = np.load(path)
    t0 = time.perf_counter()
    sol, obj, det = po.solve_chrom_exact(s, budget=budget, gamma=gamma, return_details=True)
    return time.perf_counter() - t0, int(det["selected_count"])


def _file_sha256(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def rehearse(args, rank, world):
    """The N > 1 host path without a device: partition, fabricated intervals, gather, timing, JSON line."""
    import torch
    import torch.distributed as dist

    from rocco_amd import shard, synth

    if world > 1:
        dist.init_process_group(backend=os.environ.get("ROCCO_BENCH_BACKEND", "gloo"))
    names = [c for c in args.chroms.split(",") if c] or None
    genome = synth.chrom_loci(args.step_bp, names)
    sizes = [n for _, n in genome]
    owned = shard.lpt_partition(sizes, world)
    mine = owned[rank]

    def one_step():
        local = {}
        for idx in mine:
            m = 3 + idx % 5  # deterministic stand-in for a chromosome's merged runs
            local[idx] = np.stack([np.arange(m) * 1000 + idx, np.arange(m) * 1000 + idx + 50], axis=1).astype(np.int64)
        return shard.gather_intervals(local) if world > 1 else local

    for _ in range(args.warmup):
        merged = one_step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        merged = one_step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # after the timed region every rank checks its shortest chromosome (here: that the gathered table holds exactly the
    # rows it fabricated for it; the real run checks the solve against the oracle as well) and the verdicts are summed
    ok = 1.0
    if mine:
        idx = min(mine, key=lambda i: (sizes[i], i))
        m = 3 + idx % 5
        want = np.stack([np.arange(m) * 1000 + idx, np.arange(m) * 1000 + idx + 50], axis=1).astype(np.int64)
        ok = float(idx in merged and np.array_equal(np.asarray(merged[idx]), want))
    ranks_seen, parity_ranks_ok = 1, int(ok)
    if world > 1:
        t = torch.tensor([elapsed, 1.0, ok], dtype=torch.float64)
        dist.all_reduce(t[0:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:3], op=dist.ReduceOp.SUM)
        elapsed, ranks_seen, parity_ranks_ok = float(t[0]), int(round(float(t[1]))), int(round(float(t[2])))
    if rank == 0:
        total = int(sum(sizes))
        print(json.dumps({
            "metric": "REHEARSAL (no device work): loci/sec to converged solve", "value": total / (elapsed / max(1, args.steps)),
            "unit": "loci/s", "n_gpus": world, "ranks_seen": ranks_seen, "parity_ranks_ok": parity_ranks_ok,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "fabricated", "rehearsal": True,
            "config": {"workload": f"{len(genome)} chromosomes, {total} loci", "parallelism": f"chromosome-sharded x{world} (LPT)"},
            "chromosomes_gathered": len(merged), "intervals_gathered": int(sum(v.shape[0] for v in merged.values())),
            "shard_loci": [int(sum(sizes[i] for i in part)) for part in owned]}))
    if world > 1:
        dist.destroy_process_group()


def _solve_by_track_kind_leg(args, genome, device, po):
    """What the budgeted solve (rocco/dp.py:89-164 through rocco_hip_solve_budget_batch_f64) costs OFF the benchmark's one
    input distribution: chr1 alone and the whole genome in one batch, for (i) score tracks made by SURVEY.md section 8(d)'s
    recipe (K = 10: gamma background rounded to 5 decimals, a peak every 1500 +- 300 loci of width 4..40 and amplitude
    gamma(6, 1) in 80 % of the samples; column medians), (ii) zero-inflated tracks (63 % exact zeros, the same peaks on top),
    (iii) integer-valued tracks, (iv) the benchmark's hash tracks at the corners of the driver's ranges, gamma in {0.5, 10} x
    budget in {0.005, 0.1}.  Per case: milliseconds (median of three calls), the search's passes, which path every chromosome
    took, binade maps and zone iterations; the shortest chromosome's penalty / count / solution against the oracle."""
    import torch

    from rocco_amd import dp as _dp
    from rocco_amd import rocco as rr
    from rocco_amd import synth

    def peaks_on(base, gen, integer=False):
        """base [K, n]: + a peak every 1500 +- 300 loci, width 4 .. 40, amplitude gamma(6, 1) in 80 % of the rows."""
        K, n = base.shape
        count = max(1, n // 1500)
        starts = (torch.arange(count, device=device) * 1500 + torch.randint(0, 601, (count,), device=device, generator=gen)).clamp_(max=n - 1)
        widths = torch.randint(4, 41, (count,), device=device, generator=gen)
        amp = torch._standard_gamma(torch.full((K, count), 6.0, device=device, dtype=torch.float64), generator=gen)
        amp = amp * (torch.rand((K, count), device=device, generator=gen) < 0.8)
        if integer:
            amp = torch.floor(amp)
        locus = torch.arange(n, device=device)
        which = (torch.searchsorted(starts, locus, right=True) - 1).clamp_(min=0)
        inside = (locus >= starts[which]) & (locus < starts[which] + widths[which])
        return base + amp[:, which] * inside

    def track(kind, n, seed):
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed))
        if kind == "hash":
            return rr.score_central_tendency_chrom_device(synth.hash_matrix_device(10, n, seed, device=device))
        K = 10
        gamma1 = torch._standard_gamma(torch.ones((K, n), device=device, dtype=torch.float64), generator=gen)
        if kind == "survey":
            m = peaks_on(torch.round(gamma1 * 0.3 * 1e5) / 1e5, gen)
        elif kind == "zero_inflated":
            zero = torch.rand((K, n), device=device, generator=gen) < 0.6  # (the column median of ten is then zero at ~63 % of the loci)
            m = peaks_on(torch.where(zero, torch.zeros_like(gamma1), torch.round(gamma1 * 0.3 * 1e5) / 1e5), gen)
        else:  # integers
            m = peaks_on(torch.floor(gamma1 * 1.5), gen, integer=True)
        return rr.score_central_tendency_chrom_device(m.contiguous())

    def solve(scores, gamma, budget):
        targets = [int(np.floor(int(s.shape[0]) * budget)) for s in scores]
        times, out = [], None
        for _rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = _dp.calibrate_batch_device(scores, [gamma] * len(scores), targets)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(times[1:]))
        paths = {}
        for r in out:
            paths[str(r[4]["path"])] = paths.get(str(r[4]["path"]), 0) + 1
        loci = sum(int(s.shape[0]) for s in scores)
        return {"ms": round(ms, 3), "ns_per_locus": round(1e6 * ms / loci, 4), "passes_max": max(r[4]["passes"] for r in out),
                "paths": paths, "maps": sum(r[4]["maps"] for r in out), "zone_iters": sum(max(0, r[4]["zone_iters"]) for r in out)}, out

    small = int(np.argmin([n for _, n in genome]))
    leg, cases = {}, [("hash", 1.0, 0.02), ("survey", 1.0, 0.02), ("zero_inflated", 1.0, 0.02), ("integers", 1.0, 0.02),
                      ("hash", 0.5, 0.005), ("hash", 0.5, 0.1), ("hash", 10.0, 0.005), ("hash", 10.0, 0.1)]
    made = {}
    for kind, gamma, budget in cases:
        if kind not in made:
            made[kind] = [track(kind, n, synth.chrom_seed(args.seed + 77, idx)) for idx, (_name, n) in enumerate(genome)]
        scores = made[kind]
        entry = {"gamma": gamma, "budget": budget, "exact_zero_fraction": round(float((scores[0] == 0).double().mean()), 3)}
        entry["chr1"], _ = solve(scores[:1], gamma, budget)
        entry["genome"], out = solve(scores, gamma, budget)
        # the shortest chromosome against the oracle's sequential calibration
        s_h = scores[small].cpu().numpy()
        o_sol, _o_obj, o_det = po.solve_chrom_exact(s_h, budget=budget, gamma=gamma, return_details=True)
        pen, sol_t, _val, cnt, _info = out[small]
        entry["oracle_check"] = {"chromosome": genome[small][0], "penalty_equal": pen == float(o_det["selection_penalty"]),
                                 "count_equal": cnt == int(o_det["selected_count"]),
                                 "solution_equal": bool(np.array_equal(sol_t.cpu().numpy(), o_sol))}
        leg[f"{kind}, gamma {gamma:g}, budget {budget:g}"] = entry
    ref = leg["hash, gamma 1, budget 0.02"]["genome"]["ns_per_locus"]
    for entry in leg.values():
        entry["genome"]["vs_hash_tracks"] = round(entry["genome"]["ns_per_locus"] / ref, 2)
    return leg


def _composed_driver_leg(args, genome, device, works, mine, po):
    """`rocco_amd.rocco.run_chromosomes` -- cache (scores, budget estimates, switch costs), pooled budgets, solve, BED text --
    on the whole genome: K = 10 bigWig-style tracks with the bootstrap multipliers made on the host (the reference's own
    NumPy / SciPy calls: every statistic bit for bit) and on the device (normal.hip), and K = 100 count matrices with
    device multipliers; host multipliers of the count branch on one short chromosome only (they take hours on the genome);
    the oracle's composition (CPU, one chromosome) beside them."""
    import torch

    from rocco_amd import budget, synth
    from rocco_amd import inference
    from rocco_amd import rocco as rr

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import budget_oracle as bo

    base = {"budget_null_draws": 25, "threads": -1, "gamma": None, "budget": None, "scale_chrom_budgets": 1.0,
            "budget_posterior_quantile": 0.01, "selection_penalty": None, "min_length_bp": None, "score_lower_bound_z": 1.0,
            "score_prior_df": 5.0, "score_min_effect": None, "score_precision_floor_ratio": 0.01, "low_memory": False,
            "narrowPeak": False}
    names = [n for n, _ in genome]
    total = sum(n for _, n in genome)

    def run(inputs, chroms, track_type, multipliers, tmp, **more):
        phases, mult = {}, {}
        budget.collect_timings(mult)
        run_args = dict(base, input_track_type=track_type, budget_null_multipliers=multipliers, _phase_seconds=phases,
                        output=os.path.join(tmp, f"{track_type}_{multipliers}{'_consumed' if more else ''}.bed"), **more)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = rr.run_chromosomes(chroms, inputs, run_args, run_id="b")
        seconds = time.perf_counter() - t0
        budget.collect_timings(None)
        loci = sum(int(inputs[c][1].shape[1]) for c in chroms)
        cache_s = phases.get("cache_s", 0.0)
        estimates = phases.get("budget_estimates_s", 0.0)
        made = mult.get("multipliers_host_s", 0.0) + mult.get("multipliers_device_s", 0.0)
        return {"seconds": round(seconds, 3), "loci_per_s": round(loci / seconds, 1), "chromosomes": len(chroms), "loci": loci,
                "multipliers": multipliers, "intervals": sum(1 for _ in open(out)),
                "split_s": {"matrices_to_device": round(phases.get("gather_s", 0.0), 3), "scoring": round(phases.get("scoring_s", 0.0), 3),
                            "budget_null_multipliers_" + multipliers: round(made, 3),
                            "budget_null_device_rest": round(max(0.0, estimates - made), 3),
                            "switch_costs_and_cache_rest": round(max(0.0, cache_s - phases.get("gather_s", 0.0) - phases.get("scoring_s", 0.0) - estimates), 3),
                            "pooled_budgets": round(phases.get("pooled_budgets_s", 0.0), 3),
                            "solve_and_chromosome_bed_text": round(phases.get("solve_and_chromosome_files_s", 0.0), 3),
                            "combined_bed_text": round(phases.get("combine_s", 0.0), 3)}}, out

    leg = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory(prefix="rocco_composed_") as tmp:
        os.chdir(tmp)
        try:
            # (a) bigWig-style tracks, K = 10, whole genome
            tracks = {}
            for idx, (name, n) in enumerate(genome):
                tracks[name] = (np.arange(n, dtype=np.int64) * args.step_bp,
                                synth.hash_matrix_device(10, n, synth.chrom_seed(args.seed, idx), device=device))
            run({names[-1]: tracks[names[-1]]}, [names[-1]], "bigwig", "device", tmp)  # (first-call allocations)
            on_device, bed_dev = run(tracks, names, "bigwig", "device", tmp)
            on_host, bed_host = run(tracks, names, "bigwig", "host", tmp)
            on_device["bed_equal_to_host_multipliers_run"] = open(bed_dev).read() == open(bed_host).read()
            # the oracle's composition (CPU restatement of the reference's driver) on one chromosome
            small = min(genome, key=lambda g: g[1])[0]
            inputs_h = {small: (tracks[small][0], tracks[small][1].cpu().numpy())}
            t0 = time.perf_counter()
            _c, _b, _s, o_combined = bo.run_chromosomes([small], inputs_h, dict(base, input_track_type="bigwig"))
            t_cpu = time.perf_counter() - t0
            one, bed_one = run({small: tracks[small]}, [small], "bigwig", "host", tmp)
            leg["bigwig_K10_whole_genome"] = {
                "device_multipliers": on_device, "host_multipliers": on_host,
                "cpu_oracle_one_chromosome": {"chromosome": small, "seconds": round(t_cpu, 3),
                                              "loci_per_s": round(inputs_h[small][1].shape[1] / t_cpu, 1),
                                              "gpu_seconds_same_chromosome_host_multipliers": one["seconds"],
                                              "bed_identical": open(bed_one).read() == po.bed_text(o_combined)}}
            del tracks
            # (b) count matrices, K = 100, whole genome: the benchmark's matrices turned into counts
            counts = {}
            for idx, w in enumerate(works):
                synth.hash_matrix_device(args.samples, w.n, synth.chrom_seed(args.seed, mine[idx]), out=w.matrix_t)
                w.matrix_t.mul_(20.0).round_()
                counts[w.name] = (np.arange(w.n, dtype=np.int64) * args.step_bp, w.matrix_t)
            order = [w.name for w in works]
            small = min(works, key=lambda w: w.n).name
            torch.cuda.reset_peak_memory_stats()
            first_run, _bed = run(counts, order, "bam", "device", tmp)
            bed_first = open(_bed).read()
            # (the first run of a process allocates the scoring's blocks -- ~200 GB, seconds of the runtime's time, box to box;
            # as in the count-path leg the run after it is the one reported, both are listed)
            genome_counts, _bed = run(counts, order, "bam", "device", tmp)
            genome_counts["first_run_seconds"] = first_run["seconds"]
            genome_counts["first_run_split_s"] = first_run["split_s"]
            genome_counts["peak_torch_memory_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
            bed_kept_inputs = open(_bed).read()  # (the one-chromosome runs below write to the same name)
            genome_counts["bed_equal_to_the_first_run"] = bed_kept_inputs == bed_first
            one_dev, bed_dev = run({small: counts[small]}, [small], "bam", "device", tmp)
            one_host, bed_host = run({small: counts[small]}, [small], "bam", "host", tmp)
            one_dev["bed_equal_to_host_multipliers_run"] = open(bed_dev).read() == open(bed_host).read()
            # the same with the matrices handed over (`consume_inputs`: centred in place, no copies; they are the caller's no more)
            torch.cuda.reset_peak_memory_stats()
            consumed, bed_consumed = run(counts, order, "bam", "device", tmp, consume_inputs=True)
            consumed["peak_torch_memory_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
            consumed["bed_equal_to_the_run_that_kept_its_inputs"] = open(bed_consumed).read() == bed_kept_inputs
            leg[f"counts_K{args.samples}_whole_genome"] = {
                "device_multipliers": genome_counts, "device_multipliers_inputs_consumed": consumed,
                "one_chromosome": {"chromosome": small, "device_multipliers": one_dev, "host_multipliers": one_host,
                                   "note": "host multipliers (NumPy normals + SciPy FFT for K rows per draw) on the whole genome take "
                                           "hours; their cost is shown on the shortest chromosome"}}
            inference.release_batch_workers()
        finally:
            os.chdir(cwd)
    return leg


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # not under a launcher: start the ranks (fresh processes) before this one has touched the GPU
        raise SystemExit(self_launch(args))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse:
        return rehearse(args, rank, world)

    import torch
    import torch.distributed as dist

    from rocco_amd import pipeline, shard, synth
    from rocco_amd import rocco as rr

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (rocco_amd has no CPU path)")
    # one process per GPU over RCCL; ROCCO_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with
    # fewer GPUs than ranks (ranks then share devices; timings of such a run mean nothing)
    backend = os.environ.get("ROCCO_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    names = [c for c in args.chroms.split(",") if c] or None
    genome = synth.chrom_loci(args.step_bp, names)  # [(name, n)]
    sizes = [n for _, n in genome]
    owned = shard.lpt_partition(sizes, world)
    mine = owned[rank]
    total_loci = int(sum(sizes))
    K = args.samples

    # ---- inputs resident in HBM (not timed) ----
    works = []
    for idx in mine:
        name, n = genome[idx]
        m_t = synth.hash_matrix_device(K, n, synth.chrom_seed(args.seed, idx), device=device)
        works.append(pipeline.ChromWork(name, m_t, args.budget, args.gamma, step=args.step_bp))
    torch.cuda.synchronize()

    median_events = []

    def one_step(timing=None):
        # the decode leaves every owned chromosome's runs in ONE table of rows (unit, begin, end) -- unit = the
        # chromosome's genome-wide index -- on the device and, after the decode's one synchronisation, in pinned host memory
        res = pipeline.solve_rank(works, median_timing=timing, units=mine)
        if world > 1:
            # the rows go to the gather as they are (RCCL: device tensors, ONE transfer to the host after the exchange;
            # Gloo, the CPU rehearsal of this path: the host copy)
            if backend == "nccl":
                merged = shard.gather_interval_rows(pipeline.interval_rows(res, host=False))
            else:
                merged = shard.gather_interval_rows(torch.from_numpy(np.ascontiguousarray(pipeline.interval_rows(res, host=True))))
            return res, merged
        # per chromosome: its rows of the pinned table (a view: valid until the next step's decode)
        local = {idx: r["rows_host"][r["row_range"][0]:r["row_range"][1], 1:] for idx, r in zip(mine, res)}
        return res, local

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res, merged = one_step()
    # a full (generation-2) pass of Python's cyclic GC costs ~30 ms here and would otherwise fire once,
    # a few steps into the run: take it now and keep the survivors out of later passes
    gc.collect()
    gc.freeze()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, merged = one_step(median_events if rank == 0 else None)
    sync_all()
    elapsed = time.perf_counter() - t0
    # ---- every rank checks its SHORTEST chromosome against the oracle (after the timed region; N > 1 only -- at N = 1
    # the `parity` block below does the same on the longest): scores against np.median, penalty / count / solution
    # against the reference's calibration restated on the CPU, and the rows that came back through the gather against
    # the records of that solution.  The verdicts are summed into `parity_ranks_ok`: a SCALE record then proves that
    # every rank produced, and the exchange delivered, the reference's answer.
    rank_ok, rank_check = 1.0, None
    if world > 1 and works and not args.headline_only:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po

        pos = min(range(len(works)), key=lambda k: (works[k].n, k))
        w, r, unit = works[pos], res[pos], mine[pos]
        s_h = r["scores"].cpu().numpy()
        scores_ok = bool(np.array_equal(s_h, np.median(w.matrix_t.cpu().numpy(), axis=0)))
        o_sol, _o_obj, o_det = po.solve_chrom_exact(s_h, budget=w.budget, gamma=w.gamma, return_details=True)
        sel = (o_sol[: w.n - 1] > 0).astype(np.int8)
        edges = np.diff(np.concatenate([[0], sel, [0]]))
        o_rows = np.stack([np.flatnonzero(edges == 1), np.flatnonzero(edges == -1)], axis=1).astype(np.int64)
        got_rows = np.asarray(merged.get(unit, np.zeros((0, 2), dtype=np.int64)))
        rank_check = {"chromosome": w.name, "scores_bit_exact": scores_ok,
                      "penalty_equal": bool(r["selection_penalty"] == o_det["selection_penalty"]),
                      "count_equal": bool(r["selected_count"] == o_det["selected_count"]),
                      "solution_bit_exact": bool(np.array_equal(r["solution"].cpu().numpy(), o_sol)),
                      "gathered_rows_identical": bool(got_rows.shape == o_rows.shape and np.array_equal(got_rows, o_rows))}
        rank_ok = float(all(v for k, v in rank_check.items() if k != "chromosome"))
    ranks_seen, parity_ranks_ok = 1, int(rank_ok)
    if world > 1:
        t = torch.tensor([elapsed, 1.0, rank_ok], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t[0:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:3], op=dist.ReduceOp.SUM)
        elapsed, ranks_seen, parity_ranks_ok = float(t[0].item()), int(round(float(t[1].item()))), int(round(float(t[2].item())))
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    value = total_loci / (elapsed / max(1, args.steps))

    # ---- roofline ----
    # dominant kernel: the K x n median scoring, one launch per group of chromosomes (rocco_amd/pipeline.py); its
    # launches INSIDE the timed steps were bracketed with HIP events on the stream they went to
    roofline = None
    paths = {}
    if rank == 0 and works and median_events:
        esize = works[0].matrix_t.element_size()
        durs = [a.elapsed_time(b) for a, b, _ in median_events]  # ms
        launches_per_step = len(median_events) // max(1, args.steps)
        dur_ms = float(np.mean(durs))
        alg_bytes = float(np.mean([nb for _, _, nb in median_events]))  # (8K + 8) B per locus x loci per launch
        achieved = alg_bytes / (dur_ms * 1e-3) / 1e9
        my_loci = int(sum(w.n for w in works))
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE,
        # separate runs; profiles/README.md) -- only while they describe THIS build of the kernel and these launches
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r03_pmc_median.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            same_kernel = pmc.get("median_hip_sha256") == _file_sha256(os.path.join(ROOT, "rocco_amd", "csrc", "median.hip"))
            same_work = (int(pmc.get("K", -1)) == K and int(pmc.get("loci_per_step", -1)) == my_loci and esize == 8
                         and int(pmc.get("launches_per_step", -1)) == launches_per_step)
            if same_kernel and same_work:
                traffic = int(round(pmc["traffic_bytes_per_launch"]))
                traffic_src = "profiles/r03_pmc_median.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; same median.hip, same launches)"
        step_bytes = (esize * K + 8 + 8 + 1) * total_loci  # scoring + one read of the scores + the solution
        roofline = {"bound": "hbm", "kernel": f"median_batch_kernel<K={K}> ({launches_per_step} launches per step over {my_loci} loci)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_kernel_ms": round(dur_ms, 4),
                    "algorithmic_bytes_per_launch": int(alg_bytes), "traffic": traffic, "traffic_source": traffic_src,
                    "launches_timed": len(median_events),
                    # the whole step against the same roof: 817 B per locus at K = 100
                    "step": {"algorithmic_bytes": int(step_bytes), "ms": round(ms_per_step, 3),
                             "achieved": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                             "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        for r in res:
            paths[r["name"]] = {"path": r["path"], "passes": r["info"]["passes"], "maps": r["info"].get("maps", 0),
                                "zone_iters": r["info"].get("zone_iters", -1)}

    # ---- the two halves of a step on their own (rank 0; after the timed region) ----
    if rank == 0 and world == 1 and roofline is not None and not args.headline_only:
        from rocco_amd import dp as _dp

        mats = [w.matrix_t for w in works]
        for _ in range(2):
            scores = rr.score_central_tendency_chrom_batch_device(mats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            scores = rr.score_central_tendency_chrom_batch_device(mats)
        torch.cuda.synchronize()
        t_med = (time.perf_counter() - t0) / reps
        targets = [int(np.floor(w.n * w.budget)) for w in works]
        gammas = [w.gamma for w in works]
        _dp.calibrate_batch_device(scores, gammas, targets)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            solved = _dp.calibrate_batch_device(scores, gammas, targets)
        torch.cuda.synchronize()
        t_solve = (time.perf_counter() - t0) / reps
        solve_bytes = 9 * total_loci  # compulsory: the scores once, the solution once
        roofline["scoring_alone"] = {"ms": round(1e3 * t_med, 3), "achieved": round((esize * K + 8) * total_loci / t_med / 1e9, 1),
                                     "frac": round((esize * K + 8) * total_loci / t_med / 1e9 / HBM_PEAK_GBS, 4),
                                     "note": "every chromosome in one launch"}
        roofline["solve"] = {"ms_alone_one_batch": round(1e3 * t_solve, 3), "compulsory_bytes": int(solve_bytes),
                             "achieved": round(solve_bytes / t_solve / 1e9, 1),
                             "frac": round(solve_bytes / t_solve / 1e9 / HBM_PEAK_GBS, 5),
                             "passes": int(max(s[4]["passes"] for s in solved)),
                             "note": "the calibration is a chain of launch-latency-bound rounds (DESIGN.md section 8); "
                                     "one of them reads every score (8 B per locus), the others a few per cent of them"}
        del scores, solved

    # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 semantics) ----
    cpu_baseline = None
    parity = None
    m_t = None
    if args.headline_only:
        args.no_cpu_baseline = True
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po

        sample = args.cpu_sample if any(n == args.cpu_sample for n, _ in genome) else genome[-1][0]
        sidx = [n for n, _ in genome].index(sample)
        n_s = genome[sidx][1]
        for w in works:
            if w.name == sample:
                m_t = w.matrix_t
        if m_t is None:
            m_t = synth.hash_matrix_device(K, n_s, synth.chrom_seed(args.seed, sidx), device=device)
        m_h = m_t.cpu().numpy()
        t0 = time.perf_counter()
        s_h = np.median(m_h, axis=0)  # the reference's own scoring call (rocco/rocco.py:265)
        t_score = time.perf_counter() - t0
        t0 = time.perf_counter()
        o_sol, o_obj, o_det = po.solve_chrom_exact(s_h, budget=args.budget, gamma=args.gamma, return_details=True)
        t_solve = time.perf_counter() - t0
        intervals = np.arange(n_s, dtype=np.int64) * args.step_bp
        t0 = time.perf_counter()
        o_recs = po.chrom_solution_records(sample, intervals, o_sol)
        t_decode = time.perf_counter() - t0
        cpu_total = t_score + t_solve + t_decode
        model, usable, logical = _cpu_info()
        # the solve stage of the whole genome under the reference's process policies (rocco/rocco.py:792-806: at most
        # 4 forked workers; here spawned, fork after HIP initialisation being unsafe): 1 process, 4, every usable core.
        # The scores are the GPU's medians (bit-identical to np.median: the parity block below checks the sample's).
        policies = {}
        with tempfile.TemporaryDirectory(prefix="rocco_cpu_") as tmp:
            jobs = []
            for w in sorted(works, key=lambda w: -w.n):
                path = os.path.join(tmp, w.name + ".npy")
                np.save(path, rr.score_central_tendency_chrom_device(w.matrix_t).cpu().numpy())
                jobs.append((path, args.budget, args.gamma))
            import multiprocessing as mp

            ctx = mp.get_context("spawn")
            # every usable core, one chromosome per core -- there are only as many jobs as chromosomes
            share = max(1, min(usable, len(jobs)))
            for label, procs in (("one_process", 1), ("four_processes_reference_policy", min(4, usable)), ("all_cores_one_chromosome_per_core", share)):
                if procs <= 4 and label.startswith("all_cores"):
                    continue
                t0 = time.perf_counter()
                if procs == 1:
                    out = [_cpu_solve_worker(j) for j in jobs]
                else:
                    with ctx.Pool(procs) as pool:
                        out = pool.map(_cpu_solve_worker, jobs, chunksize=1)
                wall = time.perf_counter() - t0
                policies[label] = {"processes": procs, "wall_s": round(wall, 3), "loci_per_s": round(sum(w.n for w in works) / wall, 1),
                                   "sum_of_chromosome_solves_s": round(sum(o[0] for o in out), 3)}
        cpu_baseline = {"value": round(n_s / cpu_total, 1), "unit": "loci/s", "cores": 1, "kind": "port",
                        "sample": f"{sample} (n={n_s}), K={K}, whole path on one core: np.median {t_score:.2f}s + oracle chain solve "
                                  f"(62 evaluations) {t_solve:.2f}s + python decode {t_decode:.2f}s",
                        "cpu_model": model, "cores_usable": usable, "cores_logical": logical,
                        "solve_stage_whole_genome": policies}
        # parity of the GPU path on the same sample against the oracle
        w = pipeline.ChromWork(sample, m_t, args.budget, args.gamma, step=args.step_bp)
        scores = []
        g = pipeline.solve_rank([w], scores_out=scores)[0]
        g_recs = pipeline.runs_to_records(g)
        parity = {
            "sample": sample,
            "scores_bit_exact": bool(np.array_equal(scores[0].cpu().numpy(), s_h)),
            "solution_bit_exact": bool(np.array_equal(g["solution"].cpu().numpy(), o_sol)),
            "bed3_identical": bool(g_recs == o_recs),
            "penalty_abs_diff": abs(g["selection_penalty"] - o_det["selection_penalty"]),
            "path": g["path"],
        }

    # ---- the count-path scoring rows on the same sample: reported beside the headline, not part of it ----
    next_rows = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from rocco_amd import inference

        counts_t = torch.round(m_t * 20.0).contiguous()  # integer counts of count-like magnitude from the sample's tracks
        inference.score_loci_wls_device(counts_t)  # first call: Whittaker factor for this penalty, scratch buffers
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wls_scores, _ = inference.score_loci_wls_device(counts_t)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        ks, ns = min(K, 4), min(n_s, 200000)
        sub = counts_t[:ks, :ns].contiguous()
        sub_h = sub.cpu().numpy()
        t0 = time.perf_counter()
        o_scores, _ = po.score_loci_wls(sub_h)
        t_cpu = time.perf_counter() - t0
        g_scores = inference.score_loci_wls_device(sub)[0].cpu().numpy()
        # the oracle above takes this host's np.log2, as the reference does; the device's log2 is correctly rounded: on
        # integer counts below 7956 the two agree (tests/test_gpu_score_loci_wls.py)
        next_rows = {"score_loci_wls": {
            "value": round(n_s / t_gpu, 1), "unit": "loci/s", "workload": f"{sample} (n={n_s}), K={K} count matrix",
            "cpu_oracle_values_per_s": round(sub_h.size / t_cpu, 1), "cpu_sample": f"{ks} x {ns}",
            "gpu_values_per_s": round(K * n_s / t_gpu, 1),
            "scores_bit_exact_vs_oracle_numpy_log2": bool(np.array_equal(g_scores, o_scores)),
            "max_rel_score_diff_vs_oracle": float(np.abs(g_scores - o_scores).max() / np.abs(o_scores).max())}}
        del counts_t, wls_scores
        # the same scoring over EVERY chromosome of the workload at once (rocco_amd.inference.score_loci_wls_batch_device:
        # per pipeline the baselines of its chromosomes in one pair of launches, their rolling variances in one launch, the
        # rest in launches over whole matrices): the benchmark's matrices are turned into counts in place, so this leg comes
        # last; the first call also allocates its scratch (tens of GB) and the second still warms the allocator's pools: the
        # last of five is the one reported, all are listed (round 5: with the blocks of the pipelines in PyTorch's pool the
        # allocator takes a call or two more to settle)
        if len(works) > 1:
            t_calls = []
            for _call in range(5):
                mats = []
                for idx, w in enumerate(works):
                    synth.hash_matrix_device(K, w.n, synth.chrom_seed(args.seed, mine[idx]), out=w.matrix_t)
                    w.matrix_t.mul_(20.0).round_()
                    mats.append(w.matrix_t)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                batch = inference.score_loci_wls_batch_device(mats, overwrite_input=True)
                torch.cuda.synchronize()
                t_calls.append(time.perf_counter() - t0)
                finite = bool(all(bool(torch.isfinite(sc).all()) for sc, _d in batch))
                del batch, mats
            t_batch = t_calls[-1]
            # bytes a value moves over the passes the path makes (round 5): log scale 16 (it counts the row medians' first pass on
            # its way), row medians (2 more passes) 16, forward
            # sweep 24 (reads the log matrix minus the row medians, writes both parities), backward sweep 32 (reads both
            # parities and the log matrix, writes the centred matrix: both subtractions ride on the sweeps), rolling
            # variances 16, the x ranks 24, dealing 24, segment medians (2 counting passes + the gathered cells) 24,
            # accumulation 16 = 192 B per value (round 4: 248 -- the offsets' and the baselines' subtractions were passes of
            # their own, the segment medians seven passes, the log scale was read back for the first count)
            passes_bytes = 192 * K * total_loci
            next_rows["score_loci_wls_whole_workload"] = {
                "value": round(total_loci / t_batch, 1), "unit": "loci/s", "seconds": round(t_batch, 3),
                "first_call_seconds": round(t_calls[0], 3), "seconds_by_call": [round(t, 3) for t in t_calls],
                "workload": f"{len(works)} chromosomes, {total_loci} loci, K={K} count matrices, one call",
                "gpu_values_per_s": round(K * total_loci / t_batch, 1),
                "hbm_floor": {"bytes": int(passes_bytes), "achieved_GBps": round(passes_bytes / t_batch / 1e9, 1),
                              "frac_of_peak": round(passes_bytes / t_batch / 1e9 / HBM_PEAK_GBS, 4),
                              "traffic_bytes_per_value": 223.8,
                              "traffic_source": "profiles/r05_pmc_count_path.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over "
                                                "the same call with one pipeline: 1.38 TB per call; above the passes' count by the sweeps' "
                                                "warm-up re-reads, +12 B, and the row medians' later passes, +13 B)",
                              "note": "192 B per value over the passes the path makes today (round 4: 248); ~16 B per value is "
                                      "compulsory (matrix in, centred matrix out). The baseline sweeps run in verified segments "
                                      "since round 5 and are bandwidth-bound; the rolling sums remain one sequential chain per row "
                                      "(bit-exactness pins their order) and take the latency of the longest row, ~25 % of the call"},
                "scores_finite": finite}

    # ---- the composed driver (rocco/rocco.py:933-1196, 1269-1300: matrices -> scores -> data-driven budgets and switch costs ->
    # solve -> combined BED) on the BASELINE configurations, reported beside the headline, not part of it ----
    if next_rows is not None and names is None and not args.no_composed:
        next_rows["solve_by_track_kind"] = _solve_by_track_kind_leg(args, genome, device, po)
        next_rows["composed_driver"] = _composed_driver_leg(args, genome, device, works, mine, po)

    if rank == 0:
        line = {
            "metric": "loci/sec to converged solve, hg38 50bp K=%d; BED3 intervals bit-exact" % K,
            "value": round(value, 1),
            "unit": "loci/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "parity_ranks_ok": parity_ranks_ok if world > 1 else (None if parity is None else int(
                parity["scores_bit_exact"] and parity["solution_bit_exact"] and parity["bed3_identical"]
                and parity["penalty_abs_diff"] == 0.0)),
            "rank0_parity_check": rank_check,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"hg38 {'whole genome' if names is None else ','.join(names)}, K={K}, "
                                   f"{args.step_bp} bp bins, budget={args.budget}, gamma={args.gamma}, "
                                   f"{len(genome)} chromosomes, {total_loci} loci",
                       "parallelism": f"chromosome-sharded x{world} (LPT)"},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "parity": parity,
            "next_rows": next_rows,
            "solve_paths": paths,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
