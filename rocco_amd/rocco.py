"""Scoring, solve driver and BED decode on MI355X -- drop-in for the hot-path functions of the
reference's `rocco/rocco.py`:

    score_central_tendency_chrom   rocco/rocco.py:243-304  (median branch 264-265; call-site 983-991)
    chrom_solution_to_bed          rocco/rocco.py:139-191
    _merge_bed_records             rocco/rocco.py:74-95
    _write_bed_records             rocco/rocco.py:98-110
    _read_bed_records              rocco/rocco.py:53-71
    combine_chrom_results          rocco/rocco.py:194-240
    solve_cached_chromosomes       rocco/rocco.py:890-930, 1146-1196 (one process per GPU, every
                                   chromosome of the rank batched into the same device passes,
                                   instead of a fork pool of <= 4 workers)
    _cpy_narrowpeak_summit_track   rocco/rocco.py:809-835
    _write_narrowpeak_summit_offsets  rocco/rocco.py:838-872 (per-peak arg-max on the device)

The K x n -> n scoring, the chain solve and the run-length decode run in librocco_hip.so; text
formatting / file writing of the (small) interval lists stays on the host as in the reference.
"""
from __future__ import annotations

import ctypes
import logging
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from . import dp as _dp

from .budget import (estimate_budget_nonnull_fraction_from_score_track,  # noqa: E402,F401  (looked up by name in
                     estimate_budget_nonnull_fraction_from_wild_bootstrap_null)  # _build_chrom_cache, as rocco/rocco.py:23-31)
from .inference import score_loci_wls  # noqa: E402,F401

logger = logging.getLogger(__name__)

Record = Tuple[str, int, int]


# --------------------------------------------------------------------------------------------
# BED record helpers (host; interval lists are ~1e5 records per genome)
# --------------------------------------------------------------------------------------------

def _read_bed_records(bed_file: str) -> Tuple[List[Record], bool]:
    """rocco/rocco.py:53-71."""
    records: List[Record] = []
    extra = False
    with open(bed_file, "r", encoding="utf-8") as handle:
        for line_num, line in enumerate(handle, start=1):
            text = line.strip()
            if text == "":
                continue
            fields = text.split("\t")
            if len(fields) < 3:
                raise ValueError(f"BED row {line_num} in {bed_file} has fewer than 3 columns.")
            if len(fields) > 3:
                extra = True
            records.append((str(fields[0]), int(fields[1]), int(fields[2])))
    return records, extra


def _merge_bed_records(records: Sequence[Record], min_length_bp: Optional[int] = None) -> List[Record]:
    """rocco/rocco.py:74-95: order by (chromosome string, start, end) -- so chr10 sorts before
    chr2 -- merge records whose start is <= the previous end, then apply the length filter."""
    merged: List[List] = []
    for chrom, start, end in sorted(records, key=lambda rec: (rec[0], rec[1], rec[2])):
        if merged and merged[-1][0] == chrom and int(start) <= int(merged[-1][2]):
            if int(end) > int(merged[-1][2]):
                merged[-1][2] = int(end)
            continue
        merged.append([chrom, int(start), int(end)])
    return [(str(c), int(s), int(e)) for c, s, e in merged
            if min_length_bp is None or (int(e) - int(s)) >= int(min_length_bp)]


def _write_bed_records(records: Sequence[Record], output_file: str, name_features: bool = False) -> str:
    """rocco/rocco.py:98-110."""
    with open(output_file, "w", encoding="utf-8") as handle:
        for chrom, start, end in records:
            if name_features:
                handle.write(f"{chrom}\t{start}\t{end}\t{chrom}_{start}_{end}\n")
            else:
                handle.write(f"{chrom}\t{start}\t{end}\n")
    return output_file


def combine_chrom_results(chrom_bed_files: list, output_file: str, name_features: bool = False) -> str:
    """Combine per-chromosome BED files into one sorted, merged BED file (rocco/rocco.py:194-240)."""
    if os.path.exists(output_file):
        logger.info("Removing existing output file: %s", output_file)
        try:
            os.remove(output_file)
        except OSError:
            logger.info("Could not remove existing output file: %s.", output_file)
    combined: List[Record] = []
    warned = False
    for bed in chrom_bed_files:
        if not os.path.exists(bed):
            raise FileNotFoundError(f"File does not exist: {bed}")
        records, extra = _read_bed_records(bed)
        if extra and not warned:
            logger.info("More than 3 columns detected in the input BED files. Extra columns will be ignored.")
            warned = True
        combined.extend(records)
    return _write_bed_records(_merge_bed_records(combined), output_file, name_features=name_features)


# --------------------------------------------------------------------------------------------
# scoring
# --------------------------------------------------------------------------------------------

def score_central_tendency_chrom_device(matrix_t, out_t=None, method: str = "median", rank: int = 0, rank_hi: int = 0):
    """Column-wise median of a [K, n] float64/float32 CUDA tensor -> float64 CUDA tensor [n]
    (``method="rank"``: the order statistic `rank`; ``method="mean"``: the column mean; ``method="tmean"``: the mean
    of the values between the order statistics `rank` and `rank_hi`)."""
    import torch

    if matrix_t.ndim != 2:
        raise ValueError("`chrom_matrix` must be a 2D array.")
    if matrix_t.dtype not in (torch.float64, torch.float32):
        matrix_t = matrix_t.to(torch.float64)
    if matrix_t.stride(1) != 1:
        matrix_t = matrix_t.contiguous()
    K, n = int(matrix_t.shape[0]), int(matrix_t.shape[1])
    if out_t is None:
        out_t = torch.empty(n, dtype=torch.float64, device=matrix_t.device)
    if n == 0:
        return out_t
    solver = _native.solver_for(matrix_t.device.index)
    row_stride = int(matrix_t.stride(0)) if K > 1 else n
    dtype = 0 if matrix_t.dtype == torch.float64 else 1
    lib, stream = _native.load(), _dp._stream_ptr(matrix_t)
    if method == "rank":
        _native.check(lib.rocco_hip_score_order_statistic(solver.handle, matrix_t.data_ptr(), dtype, K, n,
                                                          max(row_stride, n), int(rank), out_t.data_ptr(), stream),
                      "rocco_hip_score_order_statistic")
    elif method == "mean":
        _native.check(lib.rocco_hip_score_mean(solver.handle, matrix_t.data_ptr(), dtype, K, n, max(row_stride, n),
                                               out_t.data_ptr(), stream), "rocco_hip_score_mean")
    elif method == "tmean":
        _native.check(lib.rocco_hip_score_trimmed_mean(solver.handle, matrix_t.data_ptr(), dtype, K, n, max(row_stride, n),
                                                       int(rank), int(rank_hi), out_t.data_ptr(), stream),
                      "rocco_hip_score_trimmed_mean")
    else:
        _native.check(lib.rocco_hip_score_median(solver.handle, matrix_t.data_ptr(), dtype, K, n, max(row_stride, n),
                                                 out_t.data_ptr(), stream), "rocco_hip_score_median")
    return out_t


def score_central_tendency_chrom_batch_device(matrices, with_stats: bool = False):
    """Column medians of several [K, n_i] CUDA tensors of one K and dtype in ONE launch
    (rocco_hip_score_median_batch); returns the float64 score tensors.  Falls back to one launch each when the
    matrices differ in K or dtype.

    `with_stats`: returns ``(scores, stats)`` where `stats` is a [count, 3] float64 CUDA tensor of np.min, np.max and
    sum |.| of every score array, reduced inside the same launch (None when the launch could not be batched):
    what dp.calibrate_batch_device otherwise reads every score once more for."""
    import ctypes

    import torch

    matrices = list(matrices)
    if not matrices:
        return []
    for m in matrices:
        if m.ndim != 2:
            raise ValueError("`chrom_matrix` must be a 2D array.")
    same = all(m.dtype == matrices[0].dtype and m.shape[0] == matrices[0].shape[0] and m.stride(1) == 1
               and m.device == matrices[0].device for m in matrices)
    if (not same or matrices[0].dtype not in (torch.float64, torch.float32) or int(matrices[0].shape[0]) < 2
            or (with_stats and any(int(m.shape[1]) == 0 for m in matrices))):
        outs = [score_central_tendency_chrom_device(m) for m in matrices]
        return (outs, None) if with_stats else outs
    K = int(matrices[0].shape[0])
    count = len(matrices)
    outs = [torch.empty(int(m.shape[1]), dtype=torch.float64, device=m.device) for m in matrices]
    ptrs = (ctypes.c_void_p * count)(*[m.data_ptr() for m in matrices])
    optrs = (ctypes.c_void_p * count)(*[o.data_ptr() for o in outs])
    ns = (ctypes.c_size_t * count)(*[int(m.shape[1]) for m in matrices])
    strides = (ctypes.c_size_t * count)(*[max(int(m.stride(0)), int(m.shape[1])) for m in matrices])
    solver = _native.solver_for(matrices[0].device.index)
    dtype = 0 if matrices[0].dtype == torch.float64 else 1
    if with_stats:
        stats = torch.empty((count, 3), dtype=torch.float64, device=matrices[0].device)
        _native.check(_native.load().rocco_hip_score_median_batch_stats(solver.handle, ptrs, dtype, K, ns, strides, optrs,
                                                                        count, stats.data_ptr(), _dp._stream_ptr(matrices[0])),
                      "rocco_hip_score_median_batch_stats")
        return outs, stats
    _native.check(_native.load().rocco_hip_score_median_batch(solver.handle, ptrs, dtype, K, ns, strides, optrs, count,
                                                              _dp._stream_ptr(matrices[0])),
                  "rocco_hip_score_median_batch")
    return outs


def score_central_tendency_chrom(chrom_matrix, method="quantile", quantile=0.50, tprop=0.05, power=1.0):
    r"""Return a column-wise location summary across samples (rocco/rocco.py:243-304).

    The median branch -- the only one the reference's driver reaches (rocco/rocco.py:983-991) -- runs the
    selection-network kernel; the nearest-rank quantile, the mean and the trimmed mean (summed the way SciPy 1.15's
    stats.tmean sums) run simple kernels, all bit-exact against NumPy / SciPy.  `power`: 1 (the driver's value) is the
    identity and 2 a square, as in NumPy; any other exponent goes through the device's pow (NumPy's own pow is not the
    same function on every host, see include/rocco_hip.h).
    """
    import torch

    _native.load()
    if _dp._is_tensor(chrom_matrix):
        matrix_t = chrom_matrix
        if matrix_t.ndim != 2:
            raise ValueError("`chrom_matrix` must be a 2D array.")
    else:
        arr = np.asarray(chrom_matrix)
        if arr.dtype != np.float32:
            arr = np.asarray(arr, dtype=float)
        if arr.ndim != 2:
            raise ValueError("`chrom_matrix` must be a 2D array.")
        matrix_t = torch.from_numpy(np.ascontiguousarray(arr)).to(f"cuda:{_dp._device_index()}")
    method_ = str(method).strip().lower().replace("-", "").replace("_", "")
    kernel, rank, rank_hi = "median", 0, 0
    if matrix_t.shape[0] > 1:
        if method_ == "quantile":
            if not 0.0 <= quantile <= 1.0:
                logger.warning("`quantile` must be in [0, 1]. Using the median instead.")
                quantile = 0.50
            if quantile != 0.50:
                # the sorted position np.quantile(..., method="nearest") picks, by NumPy's own rounding rule
                K = int(matrix_t.shape[0])
                kernel, rank = "rank", int(np.quantile(np.arange(K, dtype=float), quantile, method="nearest"))
        elif method_ == "mean":
            kernel = "mean"
        elif method_ == "tmean":
            # the sorted positions np.quantile(..., method="nearest") picks at tprop and 1 - tprop (rocco/rocco.py:275-286)
            K = int(matrix_t.shape[0])
            ramp = np.arange(K, dtype=float)
            kernel = "tmean"
            rank = int(np.quantile(ramp, tprop, method="nearest"))
            rank_hi = int(np.quantile(ramp, 1.0 - tprop, method="nearest"))
            if rank > rank_hi:
                raise ValueError("No array values within given limits")  # what SciPy's tmean raises for an empty range
        else:
            raise ValueError(f"Central tendency method not recognized: {method}")
    if not matrix_t.is_cuda:
        matrix_t = matrix_t.to(f"cuda:{_dp._device_index()}")
    if kernel != "median" and matrix_t.dtype != torch.float64:
        matrix_t = matrix_t.to(torch.float64)  # np.asarray(chrom_matrix, dtype=float), rocco/rocco.py:251
    out = score_central_tendency_chrom_device(matrix_t, method=kernel, rank=rank, rank_hi=rank_hi)
    if power != 1.0:  # np.power(central_tendency, power), rocco/rocco.py:255, 304
        solver = _native.solver_for(out.device.index)
        _native.check(_native.load().rocco_hip_power_f64(solver.handle, out.data_ptr(), float(power), out.data_ptr(),
                                                         int(out.shape[0]), _dp._stream_ptr(out)), "rocco_hip_power_f64")
    return out.cpu().numpy()


# --------------------------------------------------------------------------------------------
# decode
# --------------------------------------------------------------------------------------------

def decode_runs_device(solution_t, capacity: Optional[int] = None):
    """Maximal runs of selected loci among loci 0..n-2 as two int64 CUDA tensors (begin, end)."""
    import torch

    n = int(solution_t.shape[0])
    if solution_t.dtype != torch.uint8:
        solution_t = (solution_t > 0.5).to(torch.uint8)
    solution_t = solution_t.contiguous()
    solver = _native.solver_for(solution_t.device.index)
    lib = _native.load()
    cap = int(capacity) if capacity is not None else max(1024, n // 64)
    while True:
        begin_t = torch.empty(cap, dtype=torch.int64, device=solution_t.device)
        end_t = torch.empty(cap, dtype=torch.int64, device=solution_t.device)
        n_runs = ctypes.c_size_t(0)
        _native.check(lib.rocco_hip_decode_runs(solver.handle, solution_t.data_ptr(), n,
                                                begin_t.data_ptr(), end_t.data_ptr(), cap,
                                                ctypes.byref(n_runs), _dp._stream_ptr(solution_t)),
                      "rocco_hip_decode_runs")
        if n_runs.value <= cap:
            return begin_t[: n_runs.value], end_t[: n_runs.value]
        cap = int(n_runs.value)


def decode_runs_batch_device(solutions, capacities=None):
    """`decode_runs_device` for several uint8 CUDA solution tensors in three launches and one synchronisation
    (rocco_hip_decode_runs_batch); returns a list of (begin, end) tensor pairs."""
    import torch

    solutions = [s.contiguous() for s in solutions]
    count = len(solutions)
    if count == 0:
        return []
    if any(s.dtype != torch.uint8 for s in solutions):
        return [decode_runs_device(s) for s in solutions]
    dev = solutions[0].device
    ns = [int(s.shape[0]) for s in solutions]
    caps = [int(c) for c in capacities] if capacities is not None else [max(1024, n // 64) for n in ns]
    solver = _native.solver_for(dev.index)
    lib = _native.load()
    out = [None] * count
    todo = list(range(count))
    while todo:
        begins = [torch.empty(caps[i], dtype=torch.int64, device=dev) for i in todo]
        ends = [torch.empty(caps[i], dtype=torch.int64, device=dev) for i in todo]
        k = len(todo)
        n_runs = (ctypes.c_size_t * k)()
        _native.check(lib.rocco_hip_decode_runs_batch(
            solver.handle, k, (ctypes.c_void_p * k)(*[solutions[i].data_ptr() for i in todo]),
            (ctypes.c_size_t * k)(*[ns[i] for i in todo]), (ctypes.c_void_p * k)(*[b.data_ptr() for b in begins]),
            (ctypes.c_void_p * k)(*[e.data_ptr() for e in ends]), (ctypes.c_size_t * k)(*[caps[i] for i in todo]),
            n_runs, _dp._stream_ptr(solutions[0])), "rocco_hip_decode_runs_batch")
        again = []
        for slot, i in enumerate(todo):
            if n_runs[slot] <= caps[i]:
                out[i] = (begins[slot][: n_runs[slot]], ends[slot][: n_runs[slot]])
            else:
                caps[i] = int(n_runs[slot])
                again.append(i)
        todo = again
    return out


def decode_runs_table_device(solutions, units=None, capacity_rows: Optional[int] = None, eager_rows: Optional[int] = None,
                             to_host: bool = True):
    """The maximal runs of several uint8 CUDA solution tensors (at most 48) as ONE int64 table of rows
    (unit, begin, end), solution after solution (rocco_hip_decode_runs_table): three launches and one synchronisation.

    Returns ``(table_t, offsets, host_rows)``: the rows of solution i are ``table_t[offsets[i]:offsets[i + 1]]``;
    ``host_rows`` is the same table as a NumPy array in pinned memory owned by the solver -- a VIEW that the next
    decode on this device overwrites (copy it to keep it) -- or None with ``to_host=False``.  The rows that exist travel to
    the host in front of the synchronisation, copied by a kernel that reads their number on the device (`eager_rows` is
    accepted and ignored)."""
    import torch

    solutions = [s.contiguous() for s in solutions]
    count = len(solutions)
    if count == 0 or count > 48 or any(s.dtype != torch.uint8 for s in solutions):
        raise ValueError("decode_runs_table_device takes 1 to 48 uint8 solution tensors")
    dev = solutions[0].device
    ns = [int(s.shape[0]) for s in solutions]
    units = list(range(count)) if units is None else [int(u) for u in units]
    cap = int(capacity_rows) if capacity_rows is not None else max(1024, sum(ns) // 128)
    solver = _native.solver_for(dev.index)
    lib = _native.load()
    while True:
        table_t = torch.empty((cap, 3), dtype=torch.int64, device=dev)
        offsets = (ctypes.c_size_t * (count + 1))()
        host_ptr = ctypes.c_void_p()
        eager = cap if eager_rows is None else min(int(eager_rows), cap)
        _native.check(lib.rocco_hip_decode_runs_table(
            solver.handle, count, (ctypes.c_void_p * count)(*[s.data_ptr() for s in solutions]),
            (ctypes.c_size_t * count)(*ns), (ctypes.c_longlong * count)(*units), table_t.data_ptr(), cap, eager if to_host else 0,
            offsets, ctypes.byref(host_ptr) if to_host else None, _dp._stream_ptr(solutions[0])), "rocco_hip_decode_runs_table")
        total = int(offsets[count])
        if total <= cap:
            break
        cap = total
    host_rows = None
    if to_host:
        if total == 0:
            host_rows = np.zeros((0, 3), dtype=np.int64)
        else:
            raw = (ctypes.c_int64 * (3 * total)).from_address(host_ptr.value)
            host_rows = np.frombuffer(raw, dtype=np.int64).reshape(total, 3)
    return table_t[:total], [int(v) for v in offsets], host_rows


def chrom_solution_records(chromosome, intervals, solution, check_gaps_intervals=True,
                           min_length_bp=None) -> List[Record]:
    """The merged records chrom_solution_to_bed writes (rocco/rocco.py:165-190), as a list."""
    import torch

    n = len(intervals)
    n_sol = int(solution.shape[0]) if hasattr(solution, "shape") else len(solution)
    if n != n_sol:
        raise ValueError(
            f"Intervals and solution must have the same length at the pre-merge stage: {n} != {n_sol}")
    intervals_ = np.asarray(intervals)
    if check_gaps_intervals and n > 1:
        diffs = np.diff(intervals_)
        if np.any(diffs != diffs[0]):
            raise ValueError(f"Intervals must be contiguous: {set(diffs.tolist())}")
    monotone = not (n > 2 and np.any(np.diff(intervals_) <= 0))
    if _dp._is_tensor(solution):
        sol_t = solution
        if not sol_t.is_cuda:
            sol_t = sol_t.to(f"cuda:{_dp._device_index()}")
    else:
        sol_np = np.ascontiguousarray(np.asarray(solution) > 0.50, dtype=np.uint8)
        sol_t = torch.from_numpy(sol_np).to(f"cuda:{_dp._device_index()}")
    begin_t, end_t = decode_runs_device(sol_t)
    begins = begin_t.cpu().numpy()
    ends = end_t.cpu().numpy()
    chrom = str(chromosome)
    if not monotone:
        # locus starts that do not increase (only reachable with check_gaps_intervals=False or a non-positive step): a
        # run of selected loci is then not one interval.  The reference emits (intervals[i], intervals[i + 1]) locus by
        # locus and lets its merge sort them (rocco/rocco.py:180-190); the same here over the selected loci only.
        loci = np.concatenate([np.arange(b, e) for b, e in zip(begins.tolist(), ends.tolist())]) if len(begins) \
            else np.zeros(0, dtype=np.int64)
        return _merge_bed_records([(chrom, int(intervals_[i]), int(intervals_[i + 1])) for i in loci.tolist()],
                                  min_length_bp=min_length_bp)
    starts_bp = intervals_[begins] if len(begins) else np.zeros(0, dtype=np.int64)
    ends_bp = intervals_[ends] if len(ends) else np.zeros(0, dtype=np.int64)
    return [(chrom, int(s), int(e)) for s, e in zip(starts_bp, ends_bp)
            if min_length_bp is None or (int(e) - int(s)) >= int(min_length_bp)]


def chrom_solution_to_bed(chromosome, intervals, solution, ID=None, check_gaps_intervals=True,
                          min_length_bp=None) -> str:
    r"""Convert the vector of decision variables of one chromosome to a BED file
    (rocco/rocco.py:139-191): loci 0..n-2 with solution > 0.5 become (intervals[i], intervals[i+1]),
    touching records are merged, records shorter than `min_length_bp` are dropped, and the file
    `rocco_{ID}_{chromosome}.bed` (or `rocco_{chromosome}.bed`) is written in the working directory.
    """
    records = chrom_solution_records(chromosome, intervals, solution,
                                     check_gaps_intervals=check_gaps_intervals,
                                     min_length_bp=min_length_bp)
    output_file = f"rocco_{chromosome}.bed" if ID is None else f"rocco_{ID}_{chromosome}.bed"
    return _write_bed_records(records, output_file)


# --------------------------------------------------------------------------------------------
# narrowPeak summit offsets
# --------------------------------------------------------------------------------------------

def narrowpeak_summit_offsets_device(intervals_t, effect_mean_t, peak_start_t, peak_end_t, centers_t=None):
    """Summit offsets of the peaks of one chromosome (rocco/rocco.py:852-870) as an int64 CUDA tensor:
    ``intervals_t`` int64 locus starts, ``effect_mean_t`` float64 WLS mean per locus, peaks in base pairs.
    With ``centers_t`` the pair (intervals_t, centers_t) is a stored summit track's (starts, centers)."""
    import torch

    checks = [(intervals_t, torch.int64), (effect_mean_t, torch.float64), (peak_start_t, torch.int64),
              (peak_end_t, torch.int64)]
    if centers_t is not None:
        checks.append((centers_t, torch.int64))
        if centers_t.shape != intervals_t.shape:
            raise ValueError("starts and centers of a summit track must have the same length")
    for t, dt in checks:
        if t.dtype != dt or not t.is_cuda or not t.is_contiguous() or t.dim() != 1:
            raise ValueError("summit inputs must be contiguous one-dimensional CUDA tensors (int64 / float64)")
    if peak_start_t.shape != peak_end_t.shape:
        raise ValueError("peak starts and ends must have the same length")
    out_t = torch.empty_like(peak_start_t)
    solver = _native.solver_for(intervals_t.device.index)
    _native.check(_native.load().rocco_hip_narrowpeak_summit_offsets(
        solver.handle, intervals_t.data_ptr(), int(intervals_t.shape[0]),
        None if centers_t is None else centers_t.data_ptr(), effect_mean_t.data_ptr(),
        int(effect_mean_t.shape[0]), peak_start_t.data_ptr(), peak_end_t.data_ptr(), int(peak_start_t.shape[0]),
        out_t.data_ptr(), _dp._stream_ptr(intervals_t)), "rocco_hip_narrowpeak_summit_offsets")
    return out_t


def _cpy_narrowpeak_summit_track(chrom: str, intervals, effect_mean) -> Optional[str]:
    """The per-chromosome summit track `_write_narrowpeak_summit_offsets` reads back (rocco/rocco.py:809-835): an
    .npz with the left edge and the centre of every locus that has a right neighbour, and the float32 effect mean
    there.  Returns the path of a temporary file the caller owns, or None when no locus qualifies."""
    import tempfile

    edges = np.asarray(intervals).astype(np.int64, copy=False)
    track = np.asarray(effect_mean).astype(np.float32, copy=False)
    n_bins = min(edges.shape[0] - 1, track.shape[0])  # a locus needs its right edge to have a centre
    if n_bins < 1:
        return None
    left, right = edges[:n_bins], edges[1:n_bins + 1]
    with tempfile.NamedTemporaryFile(prefix=f"rocco_summit_track_{chrom}_", suffix=".npz", delete=False) as handle:
        np.savez(handle, starts=left, centers=(left + right) // 2, mean=track[:n_bins])
    return handle.name


def _write_narrowpeak_summit_offsets(peak_file: str, chrom_cache: dict, output_file: str) -> str:
    """rocco/rocco.py:838-872: one line ``{chrom}_{start}_{end}\t{offset}`` per peak of ``peak_file``.  The peaks
    of a chromosome go to the device together (binary searches + first arg-max per peak in one launch)."""
    import torch

    _native.load()
    records, _ = _read_bed_records(peak_file)
    offsets = [-1] * len(records)
    by_chrom: Dict[str, List[int]] = {}
    for idx, (chrom, _start, _end) in enumerate(records):
        by_chrom.setdefault(chrom, []).append(idx)
    device = f"cuda:{_dp._device_index()}"
    for chrom, indices in by_chrom.items():
        summit_track_file = chrom_cache.get(chrom, {}).get("summit_track_file")
        if summit_track_file is None:
            continue
        with np.load(summit_track_file) as summit_track:
            starts = np.asarray(summit_track["starts"], dtype=np.int64)
            centers = np.asarray(summit_track["centers"], dtype=np.int64)
            mean_track = np.asarray(summit_track["mean"], dtype=np.float64)
        if starts.shape[0] == 0:
            continue
        peak_start = np.array([records[i][1] for i in indices], dtype=np.int64)
        peak_end = np.array([records[i][2] for i in indices], dtype=np.int64)
        out_t = narrowpeak_summit_offsets_device(
            torch.from_numpy(np.ascontiguousarray(starts)).to(device),
            torch.from_numpy(np.ascontiguousarray(mean_track)).to(device), torch.from_numpy(peak_start).to(device),
            torch.from_numpy(peak_end).to(device), centers_t=torch.from_numpy(np.ascontiguousarray(centers)).to(device))
        for i, value in zip(indices, out_t.cpu().numpy().tolist()):
            offsets[i] = int(value)
    with open(output_file, "w", encoding="utf-8") as handle:
        for (chrom, start, end), summit_offset in zip(records, offsets):
            handle.write(f"{chrom}_{start}_{end}\t{summit_offset}\n")
    return output_file


# --------------------------------------------------------------------------------------------
# chromosome cache: matrix -> scores, budget estimate, switch cost (rocco/rocco.py:933-1110)
# --------------------------------------------------------------------------------------------

def generate_chrom_matrix(chromosome, signal_inputs, *_args, **_kwargs):
    """Where the reference decodes BAM / bigWig files into (locus starts, K x n matrix) per chromosome
    (rocco/readtracks.py:521-633).  File decoding stays with the reference's readers (out of
    scope); here `signal_inputs` is a mapping ``{chromosome: (intervals, matrix)}`` of matrices already in memory
    (NumPy arrays or CUDA tensors) and a chromosome without an entry gives ``(None, None)`` as the reference's reader
    does for one without data.  `_build_chrom_cache` looks this name up in the module at call time, so an integrator
    (or a test, as the reference's own tests do) can put the real reader in its place."""
    if not hasattr(signal_inputs, "get"):
        raise RuntimeError("rocco_amd does not decode BAM / bigWig files: pass {chromosome: (intervals, matrix)} or "
                           "replace rocco_amd.rocco.generate_chrom_matrix with the reference's reader")
    entry = signal_inputs.get(chromosome)
    return (None, None) if entry is None else (entry[0], entry[1])


def _resolve_parallel_process_count(item_count: int, thread_limit: int) -> int:
    """rocco/rocco.py:792-806: how many workers the reference would fork -- at most 4, the items, the cores.  Nothing
    is forked here; the number only sets how many bootstrap draws pass between two looks at the stopping rule
    (rocco/inference.py:804-805, 889-937), which decides how many draws a chromosome's budget estimate uses."""
    import multiprocessing as mp

    cores = max(1, os.cpu_count() or 1) if int(thread_limit) <= 0 else max(1, int(thread_limit))
    if int(item_count) <= 1 or cores <= 1 or "fork" not in mp.get_all_start_methods():
        return 1
    return int(min(int(item_count), cores, 4))


def _all_finite(values) -> bool:
    import torch

    if _dp._is_tensor(values):
        return bool(torch.isfinite(values).all())
    return bool(np.all(np.isfinite(values)))


def _host_scores(values):
    """The cache's score array: NumPy as in the reference, with its HBM twin attached when there is one."""
    if _dp._is_tensor(values):
        return _dp.host_with_device(values) if values.is_cuda else values.numpy()
    return np.asarray(values, dtype=np.float64)


def _matrix_to_device(chrom_matrix):
    import torch

    if _dp._is_tensor(chrom_matrix):
        return chrom_matrix if chrom_matrix.is_cuda else chrom_matrix.to(f"cuda:{_dp._device_index()}")
    arr = np.asarray(chrom_matrix)
    if arr.dtype != np.float32:
        arr = np.asarray(arr, dtype=np.float64)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(f"cuda:{_dp._device_index()}")


class _CachePlan:
    """What one cache build reads from `args`, read once (rocco/rocco.py:939-947, 1010-1048)."""

    def __init__(self, args: dict):
        self.bigwig = args["input_track_type"] == "bigwig"
        self.low_memory = bool(args.get("low_memory", False))
        # (not in the reference: a caller that hands over CUDA count matrices it no longer needs lets the scoring centre them
        # in place instead of in copies -- `args["consume_inputs"] = True`; default off: a caller's tensor is never written to)
        self.consume_inputs = bool(args.get("consume_inputs", False))
        self.warned_many_tracks = False
        self.draws = args["budget_null_draws"]
        workers = 1 if self.low_memory else _resolve_parallel_process_count(int(self.draws), int(args["threads"]))
        self.null_processes = min(int(self.draws), int(workers))
        self.wls = dict(lower_bound_z=args["score_lower_bound_z"], prior_df=args["score_prior_df"],
                        min_effect=args.get("score_min_effect"), precision_floor_ratio=args["score_precision_floor_ratio"])
        self.multipliers = args.get("budget_null_multipliers")  # None: the estimators' default (host)
        self.narrow_peak = bool(args.get("narrowPeak", False)) and not self.bigwig


def _gather_chromosomes(chroms_to_process: list, signal_inputs, args: dict, plan: _CachePlan, generate):
    """Every chromosome's (name, locus starts, matrix in HBM), in the caller's order: the reference's call of
    `generate_chrom_matrix` with the reference's keywords, its skip of chromosomes without data and its finiteness check
    (rocco/rocco.py:948-974).  A generator: under `--low_memory` the caller holds one matrix at a time."""
    for name in chroms_to_process:
        logger.info("Generating chromosome matrix: %s", name)
        starts, matrix = generate(
            name, signal_inputs, args.get("chrom_sizes_file"), args.get("step"),
            round_digits=args.get("round_digits"), effective_genome_size=args.get("effective_genome_size"),
            norm_method=args.get("norm_method"), min_mapping_score=args.get("min_mapping_score"),
            flag_include=args.get("flag_include"), flag_exclude=args.get("flag_exclude"),
            extend_reads=args.get("extend_reads"), center_reads=args.get("center_reads"),
            ignore_for_norm=args.get("ignore_for_norm"), scale_factor=args.get("scale_factor"),
            num_processors=args.get("threads"), low_memory=plan.low_memory)
        if starts is None or matrix is None:
            logger.warning("Skipping chromosome %s... no data found.", name)
            continue
        logger.info("Chromosome %s matrix: %s", name, tuple(matrix.shape))
        if not _all_finite(matrix):
            raise ValueError(f"{name} matrix contains non-finite values")
        if plan.bigwig and matrix.shape[0] > 1 and not plan.warned_many_tracks:
            plan.warned_many_tracks = True  # (the reference logs this per chromosome, rocco/rocco.py:977-981: once per run is enough)
            logger.warning("Multiple bigwig tracks detected (first: %s): aggregated by the column-wise median, not WLS.", name)
        matrix_t = _matrix_to_device(matrix)
        yield name, starts, matrix_t, matrix_t is not matrix  # (a matrix that came in as a CUDA tensor is the caller's: never written to)


def _score_gathered(batch: list, plan: _CachePlan, own_wls: bool, wls):
    """Scores (and, for count matrices, the details with the centred matrix) of the gathered chromosomes, all of them on
    the device at once: bigWig tracks -- every column median of the batch in ONE launch (rocco_hip_score_median_batch; a
    single track is its own score, rocco/rocco.py:254-255); count matrices -- ONE `score_loci_wls_batch_device` call,
    which shares the baseline and rolling launches between the chromosomes.  `wls` replaced by
    the caller (the reference's tests do): one call per matrix through it, with the reference's keywords."""
    import torch

    from . import inference as _inf

    if plan.bigwig:
        scores = [None] * len(batch)
        many = [i for i, (_n, _s, m, _o) in enumerate(batch) if m.ndim == 2 and m.shape[0] > 1]
        for i, (_n, _s, m, _o) in enumerate(batch):
            if m.ndim != 2:
                raise ValueError("`chrom_matrix` must be a 2D array.")
            if m.shape[0] == 1:
                scores[i] = m[0].to(torch.float64).contiguous()
        if many:
            for i, out in zip(many, score_central_tendency_chrom_batch_device([batch[i][2] for i in many])):
                scores[i] = out
        return [(s_t, {"mean": s_t}) for s_t in scores]
    if own_wls and len(batch) > 1 and not plan.low_memory:
        mats, ours = [], []
        for _n, _s, m, owned in batch:
            if m.ndim != 2:
                raise ValueError("`chrom_matrix` must be two-dimensional")
            m64 = m.to(torch.float64).contiguous()
            mats.append(m64)
            ours.append(plan.consume_inputs or owned or m64 is not m)  # centred in place where the matrix is a copy this call made (or handed over)
        # (keep_blocks: the pipelines' blocks stay allocated for the budget estimates that follow, which compute their draws in
        # them; reserve_bytes: what those estimates hold beside the blocks -- per stream a residual template and the scoring's
        # own scratch, about four times their chromosome's matrix.  Without it a K = 100 genome whose inputs are kept filled the
        # device to the last GB, and a kernel whose private segment the runtime could not place aborted the process)
        largest = max(8 * int(m.numel()) for m in mats)
        streams = min(len(mats), _native_side_streams(), int(os.environ.get("ROCCO_BUDGET_NULL_STREAMS", "3")))
        return _inf.score_loci_wls_batch_device(mats, overwrite_input=ours, keep_blocks=True, reserve_bytes=4 * streams * largest, **plan.wls)  # noqa: E501
    return [wls(m, low_memory=plan.low_memory, return_details=True, resident=True, **plan.wls) if own_wls else
            wls(m, low_memory=plan.low_memory, return_details=True, **plan.wls) for _n, _s, m, _o in batch]


def _batches_within_memory(items, plan: _CachePlan):
    """The gathered chromosomes in batches the device can score at once: a count matrix needs about four times its size
    beside itself while it is scored (log scale, baselines, rank finder, dealing), and its centred matrix stays until its
    budget estimate is done; a bigWig batch only its scores.  `--low_memory`: one chromosome at a time."""
    import torch

    batch, held = [], 0
    for item in items:
        size = int(item[2].numel()) * 8
        if batch:
            from . import inference as _inf

            free, _total = torch.cuda.mem_get_info(item[2].device)
            device = item[2].device
            # what the batch holds so far is counted as held, the rest must fit beside it; blocks PyTorch has cached and the
            # scratch the scoring pipelines kept from an earlier call are there to be used again (a second run of a process
            # saw neither as room and scored one chromosome at a time)
            cached = max(0, int(torch.cuda.memory_reserved(device)) - int(torch.cuda.memory_allocated(device)))
            room = free + cached + _inf.batch_scratch_bytes() + held
            if plan.low_memory or (not plan.bigwig and 5 * (held + size) > room) or (plan.bigwig and held + size > room):
                yield batch
                batch, held = [], 0
        batch.append(item)
        held += size
    if batch:
        yield batch


def _native_side_streams() -> int:
    from . import _native

    return int(_native.max_side_streams())


def _estimates_side_by_side(ready: list, plan, estimate, streams: int) -> list:
    """The budget estimates of a batch (rocco/rocco.py:994-1008 for tracks, 1027-1048 for count matrices: one call per
    chromosome) with the chromosomes' draws running side by side.  One estimate is a sequence of 8-25 draws -- for a
    count matrix each a K x n product and a WLS rescoring whose rolling and trend-fit launches occupy a fraction of the
    device (one workgroup per row), for a score track a handful of short reductions -- so `streams` host threads, each
    with a stream and a solver handle of its own (the count-path batch's), work through the chromosomes longest first.
    Every estimate is what the loop computes: its generators are seeded per draw, nothing is shared.  Returns
    [(fraction, meta)] in the order of `ready`; the centred matrices are released as they are done."""
    import concurrent.futures
    import threading

    import torch

    from . import _native
    from . import inference as _inf

    device = torch.device(f"cuda:{_dp._device_index()}")
    for entry in ready:
        twin = entry[4] if entry[4] is not None else _dp._resident_tensor(entry[2])
        if twin is not None:
            device = twin.device
            break
    caller = torch.cuda.current_stream(device)
    start = torch.cuda.Event()
    start.record(caller)
    order = iter(sorted(range(len(ready)), key=lambda i: -int(ready[i][2].shape[0])))
    lock = threading.Lock()
    out = [None] * len(ready)

    def work(slot):
        solver, stream = _inf._batch_worker(device.index, slot)
        with torch.cuda.device(device), torch.cuda.stream(stream), _native.use_solver(solver):
            stream.wait_event(start)
            _budget.set_null_workspace(carvers[slot])
            # (the draws' rolling launches run beside the other streams' bandwidth-bound kernels: four rows per workgroup at
            # least -- a quarter of the workgroups, each of which holds a CU's registers while its rows' chains run: estimates
            # of a K = 100 genome 5.0 -> 4.5 s)
            solver.set("rolling_group_min", 4)
            try:
                while True:
                    with lock:
                        i = next(order, None)
                    if i is None:
                        break
                    name, starts, scores, details, centred = ready[i]
                    for t in (centred, _dp._resident_tensor(scores)):
                        if t is not None:
                            t.record_stream(stream)
                    try:
                        out[i] = estimate(scores, details, centred)
                    except (torch.OutOfMemoryError, MemoryError):
                        # (a fragmented cache: whole free segments go back to the runtime, the estimate -- a function of its
                        # arguments, its generators seeded per draw -- runs once more)
                        stream.synchronize()
                        torch.cuda.empty_cache()
                        out[i] = estimate(scores, details, centred)
                    del centred
                    ready[i] = (name, starts, scores, details, None)
                    logger.info("Budget null %s: %s draws", name, out[i][1].get("num_null_draws"))
                stream.synchronize()
            finally:
                _budget.set_null_workspace(None)
                solver.set("rolling_group_min", 1)

    # Count matrices: how many draws an estimate computes together is a question of memory (a draw holds three K x n
    # blocks), asked ONCE here for all the estimates that are about to run side by side: estimates that each ask what is
    # free when they start, while the others allocate, see numbers that mean little.  (Handing the batch scoring's scratch
    # back first -- twice the matrices' bytes, idle during the estimates -- was measured: the next scoring call then spends
    # 1.7-6 s allocating it again, more than the larger groups of draws gain.)
    from . import budget as _budget

    hinted = False
    carvers = [None] * streams
    blocks = [int(e[4].shape[0]) * int(e[4].shape[1]) * 8 for e in ready if e[4] is not None and _dp._is_tensor(e[4]) and e[4].is_cuda]
    if blocks and _budget._resolve_multipliers(plan.multipliers) == "device":
        torch.cuda.synchronize(device)
        want = 4 * 3 * sum(sorted(blocks, reverse=True)[:streams])

        def usable():
            free_now, _total = torch.cuda.mem_get_info(device)
            return int(free_now) + max(0, int(torch.cuda.memory_reserved(device)) - int(torch.cuda.memory_allocated(device)))

        _budget.set_null_memory_hint((7 * usable()) // (10 * max(1, streams)))
        hinted = True
        # the blocks the batch scoring keeps for its next call (its pipelines' arenas and sweep scratch: three times the
        # matrices' bytes, idle now) are lent to the estimates, dealt largest first to the thread that holds least
        lent = _inf.borrow_batch_blocks(device.index)
        if lent:
            shares, totals = [[] for _ in range(streams)], [0] * streams
            for block in sorted(lent, key=lambda t: -int(t.numel())):
                j = totals.index(min(totals))
                shares[j].append(block)
                totals[j] += int(block.numel())
            carvers = [(_inf.BlockCarver(share) if share else None) for share in shares]
        del lent
        if os.environ.get("ROCCO_BATCH_TRACE"):
            free_now, total_now = torch.cuda.mem_get_info(device)
            print(f"[budget null] {len(blocks)} count matrices, four draws of the {streams} largest want {want / 1e9:.1f} GB; free {free_now / 1e9:.1f} of "
                  f"{total_now / 1e9:.1f} GB, torch holds {torch.cuda.memory_allocated(device) / 1e9:.1f} GB in tensors and {torch.cuda.memory_reserved(device) / 1e9:.1f} GB "
                  f"in all, the pipelines' solvers {_inf.batch_scratch_bytes() / 1e9:.1f} GB; per estimate {_budget._null_memory_hint / 1e9:.1f} GB", flush=True)
    try:
        with concurrent.futures.ThreadPoolExecutor(max_workers=streams, thread_name_prefix="rocco-budget") as pool:
            for future in [pool.submit(work, slot) for slot in range(streams)]:
                future.result()
    finally:
        if hinted:
            _budget.set_null_memory_hint(None)
            _inf.return_batch_blocks(device.index)
        carvers = None
    return out


def _build_chrom_cache(chroms_to_process: list, signal_inputs, args: dict) -> dict:
    """The chromosome cache of rocco/rocco.py:933-1110 for matrices in memory, built the way a GPU wants it built:
    the matrices are gathered first; ALL chromosomes of a batch are scored by one device call (one median launch, or one
    batched WLS scoring); then the budget estimates run chromosome after chromosome with the NEXT chromosomes' bootstrap
    multipliers already being made on host threads (`budget.TrackWeightsAhead`), or with the multipliers made on the
    device (`args["budget_null_multipliers"] = "device"`, normal.hip); then switch costs and summit tracks.  Same keys
    and values as the reference's cache ("scores" is a NumPy array that keeps its copy in HBM, `dp.ResidentArray`); the
    reference's checks, with its messages, in its order within each phase.  `generate_chrom_matrix`, `score_loci_wls`
    and the two budget estimators are taken from this module's namespace when called, as the reference's tests replace
    them there; a replaced one is called exactly as the reference calls it."""
    from . import budget as _budget
    from . import inference as _inf

    import time as _time

    this = globals()
    plan = _CachePlan(args)
    phases = args.get("_phase_seconds")  # (bench.py: a dict that receives where the time goes; the device is waited for
                                         # at phase ends only when it is given)

    def lap(key, since):
        if phases is not None:
            import torch

            torch.cuda.synchronize()
            phases[key] = phases.get(key, 0.0) + (_time.perf_counter() - since)
        return _time.perf_counter()

    wls = this["score_loci_wls"]
    track_estimate = this["estimate_budget_nonnull_fraction_from_score_track"]
    count_estimate = this["estimate_budget_nonnull_fraction_from_wild_bootstrap_null"]
    own_track = track_estimate is _budget.estimate_budget_nonnull_fraction_from_score_track
    own_count = count_estimate is _budget.estimate_budget_nonnull_fraction_from_wild_bootstrap_null
    host_multipliers = _budget._resolve_multipliers(plan.multipliers) == "host"
    cache = {}
    gathered = _gather_chromosomes(chroms_to_process, signal_inputs, args, plan, this["generate_chrom_matrix"])
    clock = _time.perf_counter()
    for batch in _batches_within_memory(gathered, plan):
        clock = lap("gather_s", clock)
        scored = _score_gathered(batch, plan, wls is _inf.score_loci_wls, wls)
        clock = lap("scoring_s", clock)
        # the reference's checks on what the scoring returned, chromosome by chromosome (rocco/rocco.py:992, 1019, 1024)
        ready = []
        for (name, starts, _matrix, _owned), (scores, details) in zip(batch, scored):
            if not _all_finite(scores):
                raise ValueError(f"{name} direct scores contain non-finite values" if plan.bigwig
                                 else f"{name} scores contain non-finite values")
            centred = None
            if not plan.bigwig:
                centred = details.pop("centered_matrix")
                # the reference casts what the scorer returned (rocco/rocco.py:1020-1023): float32 under --low_memory, float64
                # otherwise -- a replaced scorer may hand back either
                if _dp._is_tensor(centred):
                    import torch as _torch

                    want = _torch.float32 if plan.low_memory else _torch.float64
                    centred = centred if centred.dtype == want else centred.to(want)
                else:
                    centred = np.asarray(centred, dtype=np.float32 if plan.low_memory else np.float64)
                if not _all_finite(centred):
                    raise ValueError(f"{name} centered matrix contains non-finite values")
            ready.append((name, starts, _host_scores(scores), details, centred))
        del batch, scored
        # multipliers of the coming score tracks, made ahead on host threads while earlier estimates run
        ahead = {}
        depth = max(1, min(8, (os.cpu_count() or 1) // max(1, plan.null_processes)))

        def start_ahead(upto):
            for j in range(min(upto, len(ready))):
                if j not in ahead and plan.bigwig and own_track and host_multipliers and plan.null_processes > 1:
                    ahead[j] = _budget.TrackWeightsAhead(int(ready[j][2].shape[0]), None, int(max(1, plan.draws)),
                                                         random_seed=0, ahead=plan.null_processes)

        # the multipliers made on the device: nothing of an estimate waits for the host, the chromosomes' estimates run side by side
        side_streams = min(len(ready), _native_side_streams(), int(os.environ.get("ROCCO_BUDGET_NULL_STREAMS", "3")))
        together = None
        if not host_multipliers and side_streams > 1 and (own_track if plan.bigwig else own_count):
            shared = {} if plan.multipliers is None else {"multipliers": plan.multipliers}
            if plan.bigwig:
                def one(scores, details, centred):
                    return track_estimate(scores, num_null_draws=plan.draws, progress_label=None, num_processes=plan.null_processes,
                                          return_details=True, **shared)
            else:
                def one(scores, details, centred):
                    return count_estimate(centred, observed_scores=scores,
                                          dependence_lag_hint=max(25, int(details.get("local_baseline_window", 101))),
                                          num_null_draws=plan.draws, progress_label=None, num_processes=plan.null_processes,
                                          return_details=True, **plan.wls, **shared)
            together = _estimates_side_by_side(ready, plan, one, side_streams)
        try:
            for i, (name, starts, scores, details, centred) in enumerate(ready):
                start_ahead(i + 1 + depth)
                extra = {} if plan.multipliers is None else {"multipliers": plan.multipliers}
                if together is not None:
                    fraction, meta = together[i]
                elif plan.bigwig:
                    if own_track and i in ahead:
                        extra["weights_source"] = ahead.pop(i)
                    fraction, meta = track_estimate(scores, num_null_draws=plan.draws, progress_label=f"Budget null {name}",
                                                    num_processes=plan.null_processes, return_details=True,
                                                    **(extra if own_track else {}))
                else:
                    fraction, meta = count_estimate(
                        centred, observed_scores=scores, dependence_lag_hint=max(25, int(details.get("local_baseline_window", 101))),
                        num_null_draws=plan.draws, progress_label=f"Budget null {name}", num_processes=plan.null_processes,
                        return_details=True, **plan.wls, **(extra if own_count else {}))
                    centred = None
                    ready[i] = (name, starts, scores, details, None)  # the centred matrix goes as soon as its estimate is done
                if not np.isfinite(fraction):
                    raise ValueError(f"{name} budget estimate is not finite")
                n_loci = int(scores.shape[0])
                total = float(np.clip(meta.get("effective_total_count", n_loci), 1.0, n_loci))
                logger.info("%s raw budget estimate: %s", name, meta)
                gamma, gamma_meta = _budget._resolve_chrom_gamma(name, args, scores, meta)
                cache[name] = {
                    "intervals": starts,
                    "scores": scores,
                    "effect_mean": details.get("mean", scores),
                    "gamma": gamma,
                    "gamma_meta": gamma_meta,
                    "budget_count_hat": float(np.clip(fraction * total, 0.0, total)),
                    "budget_fraction_hat": float(fraction),
                    "budget_rate_meta": meta,
                    "total_count": total,
                    "num_loci": n_loci,
                }
        finally:
            for source in ahead.values():
                source.close()
            _inf.drop_batch_blocks()  # (what the batch scoring kept for these estimates goes back to the allocator)
        clock = lap("budget_estimates_s", clock)
    for name, entry in cache.items():
        effect = entry.pop("effect_mean", None)
        if plan.narrow_peak:
            effect = effect.cpu().numpy() if _dp._is_tensor(effect) else np.asarray(effect, dtype=np.float64)
            entry["summit_track_file"] = _cpy_narrowpeak_summit_track(name, entry["intervals"], effect)
    return cache


def _resolve_budgets(chrom_cache: dict, args: dict):
    """rocco/rocco.py:1113-1143 (the pooling itself: rocco_amd/budget.py)."""
    from . import budget as _budget

    return _budget._resolve_budgets(chrom_cache, args)


def _solve_cached_chromosomes(chrom_cache: dict, chrom_budgets: dict, args: dict, run_id: str) -> list:
    """rocco/rocco.py:1146-1196 with the reference's arguments and return value (the per-chromosome BED files, in cache
    order, written to the working directory): the chromosomes share the device passes of one calibration instead of a
    pool of <= 4 forked workers."""
    for chrom_, chrom_data in chrom_cache.items():
        logger.info("%s: budget=%s gamma=%s", chrom_, round(float(chrom_budgets[chrom_]), 6),
                    round(float(chrom_data["gamma"]), 6))
    solved = solve_cached_chromosomes(chrom_cache, chrom_budgets, selection_penalty=args["selection_penalty"],
                                      min_length_bp=args["min_length_bp"], run_id=run_id, write_files=True)
    return [out for _chrom, _objective, _details, out in solved]


def _world_size(group=None) -> int:
    import torch.distributed as dist

    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _run_chromosomes_sharded(chroms_to_process: list, signal_inputs, args: dict, run_id: str, group=None) -> str:
    """`run_chromosomes` over the ranks of a torch.distributed job, one process per GPU: the
    chromosomes are dealt longest first to the least loaded rank (`shard.lpt_partition`); every rank builds the cache
    entries of ITS chromosomes; exchange 1 -- every chromosome's (budget_count_hat, total_count), the only two numbers
    the pooling reads (rocco/rocco.py:1117-1127) -- then the SAME empirical-Bayes fit on every rank, in the order of
    `chroms_to_process`; every rank solves its chromosomes; exchange 2 -- the merged intervals as (unit, start, end)
    rows; rank 0 writes the per-chromosome files, combines them and removes them.  Every rank returns the output path."""
    import torch
    import torch.distributed as dist

    from . import shard as _shard

    this = globals()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    exchange_device = f"cuda:{_dp._device_index()}" if on_gpu else "cpu"
    # loci per chromosome where the caller's inputs say so (matrices in memory); else every chromosome weighs the same
    sizes = []
    for chrom_ in chroms_to_process:
        entry = signal_inputs.get(chrom_) if isinstance(signal_inputs, dict) else None
        sizes.append(int(len(entry[0])) if entry is not None and entry[0] is not None else 1)
    mine = sorted(_shard.lpt_partition(sizes, world)[rank])
    my_chroms = [chroms_to_process[u] for u in mine]
    chrom_cache = this["_build_chrom_cache"](my_chroms, signal_inputs, args)
    unit_of = {chrom_: u for u, chrom_ in enumerate(chroms_to_process)}
    pairs = _shard.gather_budget_counts({unit_of[c]: (d["budget_count_hat"], d["total_count"]) for c, d in chrom_cache.items()},
                                        len(chroms_to_process), device=exchange_device, group=group)
    pooled_view = {chroms_to_process[u]: {"budget_count_hat": pairs[u][0], "total_count": pairs[u][1]} for u in sorted(pairs)}
    chrom_budgets, _ = this["_resolve_budgets"](pooled_view, args)
    solved = solve_cached_chromosomes(chrom_cache, {c: chrom_budgets[c] for c in chrom_cache},
                                      selection_penalty=args["selection_penalty"], min_length_bp=args["min_length_bp"],
                                      run_id=run_id, write_files=False, rows_for_units=unit_of)
    rows_t = solved.rows.to(exchange_device)  # (unit, start, end) in base pairs: the decode's own table, never a Python list
    everyone = _shard.gather_interval_rows(rows_t, group=group)
    if rank == 0:
        files = []
        for u in sorted(pairs):  # every chromosome that had data, in the caller's order (also those without intervals)
            chrom_ = chroms_to_process[u]
            records = [(chrom_, int(a), int(b)) for a, b in everyone.get(u, np.zeros((0, 2), dtype=np.int64))]
            files.append(_write_bed_records(records, f"rocco_{run_id}_{chrom_}.bed"))
        combine_chrom_results(files, args["output"], name_features=False)
        for tmp_file in files:
            try:
                os.remove(tmp_file)
            except OSError as exc:
                logger.info("Could not remove chromosome-specific temp. file %s\n%s", tmp_file, exc)
    for chrom_data in chrom_cache.values():
        summit_track_file = chrom_data.pop("summit_track_file", None)
        if summit_track_file is not None:
            try:
                os.remove(summit_track_file)
            except OSError as exc:
                logger.info("Could not remove narrowPeak summit temp. file %s\n%s", summit_track_file, exc)
    dist.barrier(group=group)
    return args["output"]


def run_chromosomes(chroms_to_process: list, signal_inputs, args: dict, run_id: Optional[str] = None, group=None) -> str:
    """What the reference's `main` does between its argument parsing and its narrowPeak step (rocco/rocco.py:1269-1300):
    cache -> pooled budgets -> solve -> combined BED at `args["output"]`; the per-chromosome files are removed.  Returns
    the path of the combined BED.  Inside an initialised torch.distributed job of more than one rank the chromosomes are
    sharded over the ranks (`_run_chromosomes_sharded`: same file, written by rank 0); `run_id` must then be given, the
    same on every rank.  (`rocco_amd.pipeline.solve_rank` is the multi-GPU form with user-given budgets.)"""
    import uuid

    run_id = str(int(uuid.uuid4().hex[:5], base=16)) if run_id is None else str(run_id)
    this = globals()
    if _world_size(group) > 1:
        return _run_chromosomes_sharded(chroms_to_process, signal_inputs, args, run_id, group)
    import time as _time

    phases = args.get("_phase_seconds")
    t0 = _time.perf_counter()
    chrom_cache = this["_build_chrom_cache"](chroms_to_process, signal_inputs, args)
    t1 = _time.perf_counter()
    chrom_budgets, _ = this["_resolve_budgets"](chrom_cache, args)
    t2 = _time.perf_counter()
    tmp_chrom_bed_files = this["_solve_cached_chromosomes"](chrom_cache, chrom_budgets, args, run_id)
    t3 = _time.perf_counter()
    final_output = combine_chrom_results(tmp_chrom_bed_files, args["output"], name_features=False)
    if phases is not None:
        phases.update(cache_s=t1 - t0, pooled_budgets_s=t2 - t1, solve_and_chromosome_files_s=t3 - t2,
                      combine_s=_time.perf_counter() - t3)
    for tmp_file in tmp_chrom_bed_files:
        try:
            os.remove(tmp_file)
        except OSError as exc:
            logger.info("Could not remove chromosome-specific temp. file %s\n%s", tmp_file, exc)
    for chrom_data in chrom_cache.values():  # rocco/rocco.py:875-887
        summit_track_file = chrom_data.pop("summit_track_file", None)
        if summit_track_file is not None:
            try:
                os.remove(summit_track_file)
            except OSError as exc:
                logger.info("Could not remove narrowPeak summit temp. file %s\n%s", summit_track_file, exc)
    return final_output


# --------------------------------------------------------------------------------------------
# per-rank solve driver
# --------------------------------------------------------------------------------------------

class _Solved(list):
    """The list `solve_cached_chromosomes` returns; `rows` (with `rows_for_units`): every chromosome's merged intervals as
    one int64 CUDA tensor of (unit, start, end) rows in base pairs."""
    rows = None


def _interval_rows_device(chroms, chrom_cache, solutions, unit_of, min_length_bp):
    """(unit, start bp, end bp) rows of the solved chromosomes, made on the device from the decode's table of (unit, first
    locus, locus behind the last) rows: a chromosome's locus starts are equally spaced (checked, as
    chrom_solution_to_bed checks them), so a locus index becomes base pairs by one multiply-add per row.  Chromosomes
    whose starts are not equally spaced and increasing take the per-chromosome route and are appended."""
    import torch

    dev = solutions[0].device
    regular, odd_rows = [], []
    first_bp, step_bp = {}, {}
    for chrom, sol_t in zip(chroms, solutions):
        starts = np.asarray(chrom_cache[chrom]["intervals"])
        if int(starts.shape[0]) != int(sol_t.shape[0]):
            raise ValueError("Intervals and solution must have the same length at the pre-merge stage: "
                             f"{int(starts.shape[0])} != {int(sol_t.shape[0])}")
        diffs = np.diff(starts) if starts.shape[0] > 1 else np.zeros(0, dtype=np.int64)
        if diffs.size and np.any(diffs != diffs[0]):
            raise ValueError(f"Intervals must be contiguous: {set(diffs.tolist())}")
        if diffs.size and diffs[0] <= 0:
            records = chrom_solution_records(chrom, starts, sol_t, check_gaps_intervals=True, min_length_bp=min_length_bp)
            odd_rows.extend((unit_of[chrom], int(a), int(b)) for _c, a, b in records)
            continue
        regular.append((chrom, sol_t))
        first_bp[unit_of[chrom]] = int(starts[0]) if starts.shape[0] else 0
        step_bp[unit_of[chrom]] = int(diffs[0]) if diffs.size else 0
    tables = []
    for at in range(0, len(regular), 48):
        part = regular[at:at + 48]
        table_t, _offsets, _host = decode_runs_table_device([sol_t for _c, sol_t in part], units=[unit_of[c] for c, _s in part],
                                                            to_host=False)
        tables.append(table_t)
    if tables:
        table_t = torch.cat(tables, dim=0) if len(tables) > 1 else tables[0]
        size = max(unit_of.values()) + 1
        first_t = torch.zeros(size, dtype=torch.int64, device=dev)
        step_t = torch.zeros(size, dtype=torch.int64, device=dev)
        for unit, value in first_bp.items():
            first_t[unit] = value
            step_t[unit] = step_bp[unit]
        unit_col = table_t[:, 0]
        # the locus behind a run's last one exists (the chain's last locus is never reported selected, rocco/rocco.py:180)
        rows = torch.stack([unit_col, first_t[unit_col] + table_t[:, 1] * step_t[unit_col],
                            first_t[unit_col] + table_t[:, 2] * step_t[unit_col]], dim=1)
        if min_length_bp is not None:
            rows = rows[(rows[:, 2] - rows[:, 1]) >= int(min_length_bp)]
    else:
        rows = torch.zeros((0, 3), dtype=torch.int64, device=dev)
    if odd_rows:
        rows = torch.cat([rows, torch.tensor(odd_rows, dtype=torch.int64, device=dev).reshape(-1, 3)], dim=0)
    return rows


def solve_cached_chromosomes(chrom_cache: Dict[str, dict], chrom_budgets: Dict[str, float],
                             selection_penalty: Optional[float] = None,
                             min_length_bp: Optional[int] = None,
                             run_id: Optional[str] = None, write_files: bool = True, rows_for_units: Optional[dict] = None):
    """Solve every chromosome in `chrom_cache` on the current GPU (rocco/rocco.py:890-930 and
    1146-1196).  `chrom_cache[chrom]` holds "scores" (NumPy or CUDA tensor), "intervals" and
    "gamma" as in the reference's cache.  Returns a list of
    (chrom, objective, details, bed_path_or_records) in cache order.  With `rows_for_units` (chromosome -> unit number;
    what the sharded driver passes) no per-chromosome records are made: the list's `rows` attribute holds every
    interval as a (unit, start, end) row of ONE device table, the form the gather takes.
    """
    import torch

    chroms = list(chrom_cache)
    scores_list = []
    gammas = []
    targets = []
    for chrom in chroms:
        data = chrom_cache[chrom]
        # the reference's checks in the reference's order (rocco/rocco.py:897-914): scores, budget, gamma
        s_t = _dp._to_device_f64(data["scores"])
        if not bool(torch.isfinite(s_t).all()):
            raise ValueError(f"{chrom} scores contain non-finite values")
        try:
            budget = float(chrom_budgets[chrom])
        except (TypeError, ValueError) as exc:
            raise ValueError(f"{chrom} budget could not be read as a finite number") from exc
        try:
            gamma = float(data["gamma"])
        except (TypeError, ValueError) as exc:
            raise ValueError(f"{chrom} gamma could not be read as a finite number") from exc
        if not np.isfinite(budget) or budget < 0.0:
            raise ValueError(f"{chrom} budget must be finite and non-negative")
        if not np.isfinite(gamma) or gamma < 0.0:
            raise ValueError(f"{chrom} gamma must be finite and non-negative")
        scores_list.append(s_t)
        gammas.append(gamma)
        targets.append(int(np.floor(int(s_t.shape[0]) * budget)))  # rocco/dp.py:197

    results = _Solved()
    if selection_penalty is None:
        solved = _dp.calibrate_batch_device(scores_list, gammas, targets)
    else:
        solved = []
        for s_t, gamma in zip(scores_list, gammas):
            sol_t, value, count, path = _dp.solve_penalized_chain_device(s_t, gamma, float(selection_penalty))
            solved.append((float(selection_penalty), sol_t, value, count, {"path": path}))
    for chrom, s_t, gamma, (penalty, sol_t, value, count, info) in zip(chroms, scores_list, gammas, solved):
        objective = _dp.objective_value(sol_t, s_t, gamma)
        details = {
            "penalized_objective": float(value),
            "selected_count": int(count),
            "selected_fraction": float(count / int(s_t.shape[0])),
            "selection_penalty": float(penalty),
        }
        data = chrom_cache[chrom]
        if rows_for_units is not None:
            out = None
        elif write_files:
            out = chrom_solution_to_bed(chrom, data["intervals"], sol_t, run_id,
                                        check_gaps_intervals=True, min_length_bp=min_length_bp)
        else:
            out = chrom_solution_records(chrom, data["intervals"], sol_t, check_gaps_intervals=True,
                                         min_length_bp=min_length_bp)
        logger.info("%s solve: selected=%s (%.6f), selection_penalty=%.6f, objective=%.4f", chrom,
                    details["selected_count"], details["selected_fraction"],
                    details["selection_penalty"], objective)
        results.append((chrom, float(objective), details, out))
    if rows_for_units is not None:
        results.rows = _interval_rows_device(chroms, chrom_cache, [sol_t for (_p, sol_t, _v, _c, _i) in solved], rows_for_units,
                                             min_length_bp) if chroms else torch.zeros((0, 3), dtype=torch.int64)
    return results
