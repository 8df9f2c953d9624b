#!/usr/bin/env python3
"""bench.py -- loci/sec of the per-chromosome solve path on MI355X.

Metric (BASELINE.json): loci/sec to converged solve, hg38, 50 bp bins, K = 100 samples; BED3
intervals bit-exact.  One "step" = one pass of the hot path over the whole genome held by the job:
K x n median scoring of every chromosome, the budgeted chain solve (the reference's 2 + 60 chain
evaluations per chromosome, rocco/dp.py:89-164) and the run-length decode to merged intervals.
Inputs (synthetic K x n matrices, rocco_amd/synth.py) are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: the 24 chromosomes of ONE genome are assigned to ranks by LPT on their locus counts (strong
scaling, no data-path collective); the interval lists are gathered to every rank with two small
all_gathers over RCCL inside the timed region.
Prints one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse_args():
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
    ap.add_argument("--cpu-sample", type=str, default="chr1")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from rocco_amd import pipeline, shard, synth
    from rocco_amd import rocco as rr

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    def one_step():
        res = pipeline.solve_rank(works)
        # intervals of every owned chromosome to the host in ONE transfer
        counts = [int(r["begin"].numel()) for r in res]
        if sum(counts):
            flat = torch.cat([torch.stack([r["begin"], r["end"]], dim=1) for r in res if r["begin"].numel()]).cpu().numpy()
        else:
            flat = np.zeros((0, 2), dtype=np.int64)
        local, at = {}, 0
        for idx, c in zip(mine, counts):
            local[idx] = flat[at:at + c]
            at += c
        merged = shard.gather_intervals(local, device=device if backend == "nccl" else None) if world > 1 else local
        return res, merged

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
        res, merged = one_step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    value = total_loci / (elapsed / max(1, args.steps))

    # ---- dominant kernel: K x n median scoring; HIP events on the stream it is launched on ----
    roofline = None
    paths = {}
    if rank == 0 and works:
        big = max(works, key=lambda w: w.n)
        out = torch.empty(big.n, dtype=torch.float64, device=device)
        for _ in range(2):
            rr.score_central_tendency_chrom_device(big.matrix_t, out)
        reps = 10
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for i in range(reps):
            ev0[i].record()
            rr.score_central_tendency_chrom_device(big.matrix_t, out)
            ev1[i].record()
        torch.cuda.synchronize()
        dur_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
        esize = big.matrix_t.element_size()
        alg_bytes = (esize * K + 8) * big.n  # 8K read + 8 written per locus
        achieved = alg_bytes / (dur_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x 2 on gfx950 +
        # WRITE_SIZE, separate runs; profiles/README.md) -- only when they were taken on this very launch
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_median.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            if int(pmc.get("K", -1)) == K and int(pmc.get("n", -1)) == big.n and esize == 8:
                traffic = int(round(pmc["traffic_bytes_per_launch"]))
                traffic_src = "profiles/r01_pmc_median.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        roofline = {"bound": "hbm", "kernel": f"median_kernel<K={K}> on {big.name} (n={big.n})",
                    "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "avg_kernel_ms": round(dur_ms, 4),
                    "algorithmic_bytes_per_launch": int(alg_bytes), "traffic": traffic,
                    "traffic_source": traffic_src}
        for r in res:
            paths[r["name"]] = {"path": r["path"], "passes": r["info"]["passes"], "maps": r["info"].get("maps", 0),
                                "zone_iters": r["info"].get("zone_iters", -1)}

    # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 semantics) ----
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po

        sample = args.cpu_sample if any(n == args.cpu_sample for n, _ in genome) else genome[-1][0]
        sidx = [n for n, _ in genome].index(sample)
        n_s = genome[sidx][1]
        m_t = None
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
        cpu_baseline = {"value": round(n_s / cpu_total, 1), "unit": "loci/s", "cores": 1, "kind": "port",
                        "sample": f"{sample} (n={n_s}), K={K}: np.median {t_score:.2f}s + oracle chain solve "
                                  f"(62 evaluations) {t_solve:.2f}s + python decode {t_decode:.2f}s"}
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

        counts_t = (m_t * 20.0).contiguous()  # count-like magnitudes from the sample's tracks
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
        next_rows = {"score_loci_wls": {
            "value": round(n_s / t_gpu, 1), "unit": "loci/s", "workload": f"{sample} (n={n_s}), K={K} count matrix",
            "cpu_oracle_values_per_s": round(sub_h.size / t_cpu, 1), "cpu_sample": f"{ks} x {ns}",
            "gpu_values_per_s": round(K * n_s / t_gpu, 1),
            "max_rel_score_diff_vs_oracle": float(np.abs(g_scores - o_scores).max() / np.abs(o_scores).max()),
            "note": "log2 differs from NumPy's in the last place; everything downstream is bit-exact (tests)"}}
        del counts_t, wls_scores

    if rank == 0:
        line = {
            "metric": "loci/sec to converged solve, hg38 50bp K=%d; BED3 intervals bit-exact" % K,
            "value": round(value, 1),
            "unit": "loci/s",
            "n_gpus": world,
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
