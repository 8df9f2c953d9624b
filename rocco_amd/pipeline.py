"""Device-resident per-rank pipeline: K x n matrices -> scores -> budgeted solve -> merged runs.

This is the GPU form of the reference's `_build_chrom_cache` scoring step for bigWig inputs
(rocco/rocco.py:983-991), `_solve_cached_chromosomes` (rocco/rocco.py:1146-1196) and the decode in
`chrom_solution_to_bed` (rocco/rocco.py:139-191), for the chromosomes owned by one rank: every
array stays in HBM, every chromosome of the rank shares each device pass of the solve, and only the
interval lists (a few thousand index pairs per chromosome) come back to the host.
"""
from __future__ import annotations

import concurrent.futures
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from . import dp as _dp
from . import rocco as _rocco
from . import shard as _shard

# The calibration is a sequence of short device passes separated by host decisions: latency-bound once the
# active set has shrunk.  Chromosomes are independent, so a rank's chromosomes CAN be split into groups that
# calibrate side by side -- one host thread, HIP stream and solver handle (scratch buffers) per group; the
# native call releases the GIL -- and fill each other's gaps, each group solving while the next is scored.  Results do
# not depend on the grouping.  Default 1 (no groups: every median in one launch at the kernel's full bandwidth, then one
# batch): with the calibration of round 2 (about 12 rounds instead of 25, most of them on 2 % of the loci) the two
# phases overlapped add up to the same time on the whole genome (12.39 against 12.44 ms, five alternating pairs) and
# to MORE on the shards a rank owns at N = 2 / 4 / 8 (7.06 / 3.92 / 2.51 ms against 7.25 / 4.35 / 2.65 with three
# groups; scripts/groups_ab.sh, scripts/groups_ab_shard.sh): the kernels compete for the same wavefront slots.
SOLVE_GROUPS = int(os.environ.get("ROCCO_SOLVE_GROUPS", "1"))
# 1: score every chromosome of the rank in ONE launch before any group starts solving (the median kernel then runs at
# its full bandwidth); 0: score group after group so that the first groups solve while the later ones are scored
SCORE_FIRST = int(os.environ.get("ROCCO_SCORE_FIRST", "0"))
# 1: the median launch also reduces min / max / sum |.| of its scores, which the calibration starts from, instead of
# the calibration's own pass over the scores.  Measured on the whole genome: the reduction at
# the end of every wavefront of the bandwidth-bound kernel costs what the separate pass costs -- off by default.
MEDIAN_STATS = bool(int(os.environ.get("ROCCO_MEDIAN_STATS", "0")))
_pool: Optional[concurrent.futures.ThreadPoolExecutor] = None
_group_state: Dict[Tuple[int, int], tuple] = {}  # (device, group) -> (Solver, torch.cuda.Stream)
_group_lock = threading.Lock()  # the groups' solver handles are per process: one grouped solve at a time


def _group_resources(device_index: int, group: int):
    import torch

    key = (int(device_index), int(group))
    if key not in _group_state:
        _group_state[key] = (_native.Solver(int(device_index)), torch.cuda.Stream(device=int(device_index)))
    return _group_state[key]


class ChromWork:
    """One chromosome owned by this rank."""

    def __init__(self, name: str, matrix_t, budget: float, gamma: float, step: int = 50, start: int = 0,
                 scoring: str = "median", wls_params: Optional[dict] = None):
        """`scoring`: "median" -- column medians of signal tracks, the bigWig branch of `_build_chrom_cache`
        (rocco/rocco.py:977-991); "wls" -- `score_loci_wls` on a count matrix, its BAM branch (1009-1018), with
        `wls_params` = lower_bound_z / prior_df / min_effect / precision_floor_ratio."""
        if scoring not in ("median", "wls"):
            raise ValueError("scoring must be 'median' or 'wls'")
        self.name = name
        self.matrix_t = matrix_t  # [K, n] float64 / float32 CUDA tensor
        self.budget = float(budget)
        self.gamma = float(gamma)
        self.step = int(step)
        self.start = int(start)
        self.n = int(matrix_t.shape[1])
        self.scoring = scoring
        self.wls_params = dict(wls_params or {})


def _score_wls(c: ChromWork):
    import torch

    from . import inference as _inference

    s_t, details = _inference.score_loci_wls_device(c.matrix_t.to(torch.float64), **c.wls_params)
    c._effect_mean = details["mean"]  # rocco/rocco.py:1090 `effect_mean`, the summit track's source
    return s_t


def _solve_group(chroms: Sequence[ChromWork], scores: list, score_stats=None, units: Optional[Sequence[int]] = None) -> list:
    """Calibrate and decode the given chromosomes on the calling thread's current stream / solver; count-path
    chromosomes (scores[i] is None) are scored here first, so that the groups' chain kernels -- one latency-bound
    wavefront per row and parity -- run side by side as well."""
    if any(s is None for s in scores):
        score_stats = None
    scores = [s if s is not None else _score_wls(c) for c, s in zip(chroms, scores)]
    targets = [int(np.floor(c.n * c.budget)) for c in chroms]  # rocco/dp.py:197
    solved = _dp.calibrate_batch_device(scores, [c.gamma for c in chroms], targets, score_stats=score_stats)
    out = []
    # every chromosome's runs as ONE table of (unit, begin, end) rows: three launches, one synchronisation, and the
    # table reaches pinned host memory in the same breath (a run has at least one selected locus and on real tracks
    # about twenty: a fifth of the selected loci, never less than 4096 rows, travels before the synchronisation)
    selected = sum(int(count) for (_p, _s, _v, count, _i) in solved)
    if len(chroms) <= 48:
        table_t, offsets, host_rows = _rocco.decode_runs_table_device(
            [sol_t for (_p, sol_t, _v, _c, _i) in solved], units=units, capacity_rows=max(1024, selected // 2 + 64),
            eager_rows=max(4096, selected // 5))
        runs = [(table_t[offsets[i]:offsets[i + 1], 1], table_t[offsets[i]:offsets[i + 1], 2]) for i in range(len(chroms))]
    else:
        table_t, offsets, host_rows = None, None, None
        runs = _rocco.decode_runs_batch_device([sol_t for (_p, sol_t, _v, _c, _i) in solved],
                                               capacities=[max(1024, c.n // 64) for c in chroms])
    for i, (c, s_t, (penalty, sol_t, value, count, info), (begin_t, end_t)) in enumerate(zip(chroms, scores, solved, runs)):
        out.append({
            "name": c.name, "n": c.n, "selected_count": count, "selection_penalty": penalty,
            "penalized_objective": value, "path": info["path"], "info": info,
            "begin": begin_t, "end": end_t, "solution": sol_t, "step": c.step, "start": c.start,
            "effect_mean": getattr(c, "_effect_mean", None), "scores": s_t,
            # the group's whole table (device; host view valid until this device's next decode) and this chromosome's rows
            "rows": table_t, "rows_host": host_rows, "row_range": None if offsets is None else (offsets[i], offsets[i + 1]),
        })
    return out


def _group_chunks(order: List[int], n_groups: int) -> List[List[int]]:
    """Consecutive chunks of `order` (chromosomes by descending length), about equally many chromosomes each:
    the first groups hold the long chromosomes, the last one the shortest -- what remains to be solved once
    the last median has been taken is as little as possible."""
    k = len(order)
    base, extra = divmod(k, n_groups)
    out, at = [], 0
    for g in range(n_groups):
        size = base + (1 if g < extra else 0)
        out.append(order[at:at + size])
        at += size
    return [c for c in out if c]


def solve_rank(chroms: Sequence[ChromWork], scores_out: Optional[list] = None, groups: Optional[int] = None,
               median_timing: Optional[list] = None, units: Optional[Sequence[int]] = None):
    """Score, solve and decode every chromosome of this rank.

    Returns a list of dicts (in the order of `chroms`): name, n, selected_count, selection_penalty,
    penalized_objective, path, begin / end (int64 CUDA tensors: half-open locus index pairs of the merged runs),
    the solution and the score tensors.  `groups` (default ROCCO_SOLVE_GROUPS): how many groups of chromosomes
    calibrate side by side (see SOLVE_GROUPS above).

    With one group (the default) every median of the rank is one launch and the calibration one batch behind it.  With
    more, scoring and solving overlap: the medians (bandwidth-bound) are issued group after group on the caller's
    stream, longest chromosomes first, and every group starts calibrating (launch-latency-bound rounds on its own
    stream and host thread) as soon as ITS chromosomes are scored, while the later groups' medians are still
    running.  Count-path scoring (latency-bound chain kernels) runs inside the groups.

    `median_timing`: a list that receives one (start event, end event, algorithmic bytes) per median launch,
    recorded on the stream the launch goes to (for bench.py's roofline block).

    `units`: the number every chromosome's interval rows carry in their first column (default: its position in
    `chroms`) -- a rank of a sharded run passes the chromosomes' genome-wide indices, and the rows go to the gather as
    they are (`interval_rows`).
    """
    import torch

    global _pool
    if len(chroms) == 0:
        return []
    device = chroms[0].matrix_t.device
    # (never more group streams than hardware queues besides the caller's: _native.max_side_streams)
    n_groups = max(1, min(int(groups if groups is not None else SOLVE_GROUPS), len(chroms), _native.max_side_streams()))
    out: List[Optional[dict]] = [None] * len(chroms)
    units = list(range(len(chroms))) if units is None else [int(u) for u in units]

    def score_all(members: Sequence[ChromWork]):
        # the medians of a group in one launch (count-path chromosomes are scored inside their group), together
        # with the statistics the calibration starts from (min, max, sum |.| of every score array: reduced in the
        # same launch and copied to pinned host memory behind it -- valid once the caller has waited for the stream)
        med = [c for c in members if c.scoring != "wls"]
        if median_timing is not None and med:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(device))
        outs, stats_d = [], None
        if med and MEDIAN_STATS:
            outs, stats_d = _rocco.score_central_tendency_chrom_batch_device([c.matrix_t for c in med], with_stats=True)
        elif med:
            outs = _rocco.score_central_tendency_chrom_batch_device([c.matrix_t for c in med])
        got = dict(zip((id(c) for c in med), outs))
        if median_timing is not None and med:
            ev1.record(torch.cuda.current_stream(device))
            # K elements read + 8 bytes written per locus
            nbytes = sum((c.matrix_t.element_size() * int(c.matrix_t.shape[0]) + 8) * c.n for c in med)
            median_timing.append((ev0, ev1, nbytes))
        stats_h = None
        if stats_d is not None and len(med) == len(members):
            stats_h = torch.empty((len(med), 3), dtype=torch.float64, pin_memory=True)
            stats_h.copy_(stats_d, non_blocking=True)
        for c in med:
            c._effect_mean = got[id(c)]  # rocco/rocco.py:995-997: the bigWig branch uses the scores themselves
        return [got.get(id(c)) for c in members], stats_h

    if n_groups == 1:
        scores, stats_h = score_all(chroms)
        if stats_h is not None:
            torch.cuda.current_stream(device).synchronize()
        out = _solve_group(chroms, scores, stats_h, units)
    else:
        order = sorted(range(len(chroms)), key=lambda i: -chroms[i].n)
        members = _group_chunks(order, n_groups)
        with torch.cuda.device(device):
            caller_stream = torch.cuda.current_stream(device)

            def work(group: int, idx: List[int], scores: list, scored, stats_h=None, stats_rows=None):
                solver, stream = _group_resources(device.index, group)
                with torch.cuda.device(device), torch.cuda.stream(stream), _native.use_solver(solver):
                    if stats_h is not None:
                        scored.synchronize()  # the statistics are read on the host ...
                        if stats_rows is not None:  # ... so a group's rows are taken only once the copy has landed
                            stats_h = stats_h[stats_rows]
                    else:
                        stream.wait_event(scored)
                    res = _solve_group([chroms[i] for i in idx], scores, stats_h, [units[i] for i in idx])
                    stream.synchronize()
                    # the pinned table is the group solver's: the next decode there overwrites it
                    kept = None if not res or res[0]["rows_host"] is None else res[0]["rows_host"].copy()
                    for r in res:
                        r["rows_host"] = kept
                # the results were allocated on the group's stream and are handed to the caller's
                for r in res:
                    for key in ("begin", "end", "solution", "scores", "effect_mean", "rows"):
                        t = r.get(key)
                        if t is not None and hasattr(t, "record_stream"):
                            t.record_stream(caller_stream)
                return idx, res

            with _group_lock:
                if _pool is None:
                    # (one thread per group stream, never more than the side streams allowed)
                    _pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, _native.max_side_streams()),
                                                                  thread_name_prefix="rocco-solve")
                futures = []
                if SCORE_FIRST:
                    every, stats_all = score_all([chroms[i] for i in order])
                    by_index = dict(zip(order, every))
                    row = {i: k for k, i in enumerate(order)}
                    scored = torch.cuda.Event()
                    scored.record(caller_stream)
                    for g, idx in enumerate(members):
                        rows = [row[i] for i in idx] if stats_all is not None else None
                        futures.append(_pool.submit(work, g, idx, [by_index[i] for i in idx], scored, stats_all, rows))
                else:
                    for g, idx in enumerate(members):
                        scores, stats_h = score_all([chroms[i] for i in idx])
                        scored = torch.cuda.Event()
                        scored.record(caller_stream)
                        futures.append(_pool.submit(work, g, idx, scores, scored, stats_h))
                # every group is waited for (the solver handles are shared) before any failure is reported
                first_error = None
                for f in futures:
                    try:
                        idx, res = f.result()
                        for i, r in zip(idx, res):
                            out[i] = r
                    except BaseException as exc:  # noqa: BLE001
                        if first_error is None:
                            first_error = exc
                if first_error is not None:
                    raise first_error
    if scores_out is not None:
        scores_out.extend(r["scores"] for r in out)
    return out


def interval_rows(results: Sequence[dict], host: bool = True):
    """All interval rows (unit, begin, end) of a `solve_rank` result as one int64 array [m, 3]: the NumPy view in pinned
    host memory (`host=True`; valid until the device's next decode) or the CUDA tensor the gather takes.  With one
    group (the default) this is the table the decode wrote, untouched; with several groups their tables are joined."""
    import torch

    tables, seen = [], set()
    for r in results:
        t = r["rows_host"] if host else r["rows"]
        if t is None:  # more chromosomes than one table holds: rows from the per-chromosome tensors
            unit = torch.full_like(r["begin"], -1)
            rows = torch.stack([unit, r["begin"], r["end"]], dim=1)
            tables.append(rows.cpu().numpy() if host else rows)
        elif id(t) not in seen:
            seen.add(id(t))
            tables.append(t)
    if len(tables) == 1:
        return tables[0]
    if host:
        return np.concatenate(tables, axis=0) if tables else np.zeros((0, 3), dtype=np.int64)
    return torch.cat(tables, dim=0)


def runs_to_records(result: dict, min_length_bp: Optional[int] = None) -> List[Tuple[str, int, int]]:
    """BED3 records of one chromosome result (fixed-step loci: start + index * step)."""
    b = result["begin"].cpu().numpy()
    e = result["end"].cpu().numpy()
    step, start = result["step"], result["start"]
    recs = []
    for bi, ei in zip(b.tolist(), e.tolist()):
        s_bp, e_bp = start + bi * step, start + ei * step
        if min_length_bp is None or (e_bp - s_bp) >= int(min_length_bp):
            recs.append((result["name"], int(s_bp), int(e_bp)))
    return recs


def summit_offsets(result: dict, min_length_bp: Optional[int] = None) -> List[Tuple[str, int]]:
    """(peak name, summit offset) of every record of `runs_to_records(result)` -- what
    `_write_narrowpeak_summit_offsets` (rocco/rocco.py:838-872) writes for this chromosome -- from the result's
    `effect_mean` track, all on the device."""
    import torch

    records = runs_to_records(result, min_length_bp=min_length_bp)
    if not records:
        return []
    dev = result["begin"].device
    n, step, start = result["n"], result["step"], result["start"]
    intervals_t = start + step * torch.arange(n, dtype=torch.int64, device=dev)
    starts_t = torch.tensor([r[1] for r in records], dtype=torch.int64, device=dev)
    ends_t = torch.tensor([r[2] for r in records], dtype=torch.int64, device=dev)
    mean_t = result["effect_mean"].to(torch.float64).contiguous()
    off = _rocco.narrowpeak_summit_offsets_device(intervals_t, mean_t, starts_t, ends_t).cpu().numpy().tolist()
    return [(f"{c}_{s}_{e}", int(o)) for (c, s, e), o in zip(records, off)]
