"""Count-path scoring pieces of rocco/inference.py on the MI355X.

Built so far: the cross-fit Whittaker baseline (row a3) -- ``crossfit_whittaker_baseline`` is the
device replacement of the reference's ``rocco._baseline.crossfit_whittaker_baseline``
(rocco/_baseline.c:16-104 over rocco/native/baseline_backend.c) and
``_estimate_local_background_matrix`` mirrors rocco/inference.py:185-229 around it -- and the centred-WLS
scores (row a4): ``score_centered_wls`` replaces ``rocco._wls.score_centered_wls`` (rocco/_wls.c over
rocco/native/wls_backend.c:744-947), ``_score_centered_wls_matrix`` mirrors rocco/inference.py:231-299.
Every track equals the reference's bit for bit from the log-scaled matrix on.  The log scale itself (log2 of
count + 1) follows NumPy's to the last place or the one next to it: NumPy's own log2 is not correctly rounded and
differs between its SVML and libm builds, so no fixed algorithm reproduces it on
every host.  There is no CPU fallback: without the library or a GPU these raise.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Tuple

import numpy as np

from . import _native
from . import dp as _dp


def _resolve_local_baseline_window(n_loci: int, target_window: int = 101) -> int:
    """Window of the local baseline (rocco/inference.py:50-62): the requested size, at least 3, made odd and
    fitted into the track; no baseline (0) for tracks shorter than 25 loci."""
    n = int(n_loci)
    if n < 25:
        return 0
    w = min(max(3, int(target_window)), n)
    if w % 2:
        return w
    return w + 1 if w < n else w - 1  # even: the next odd size, unless the track ends there


_BLOCK_TO_BANDWIDTH = 0.15915494  # the reference's constant, digit for digit (rocco/inference.py:75)


def _consenrich_whittaker_lambda(block_size: int) -> float:
    """Whittaker penalty of a smoothing block (rocco/inference.py:65-76): 7 (odd block size x 0.15915494)^4."""
    odd_block = max(3, int(block_size)) | 1
    return 7.0 * (float(odd_block) * _BLOCK_TO_BANDWIDTH) ** 4


def crossfit_whittaker_baseline_device(values_t, penalty_lambda: float, out_t=None):
    """Device-resident form: ``values_t`` is a contiguous float64 CUDA tensor [rows, cols] (or [cols]);
    returns a new tensor of the same shape (or fills ``out_t``, which must not alias the input)."""
    import torch

    if values_t.dim() not in (1, 2):
        raise ValueError("`values` must be one-dimensional or two-dimensional")
    if values_t.dtype != torch.float64 or not values_t.is_cuda or not values_t.is_contiguous():
        raise ValueError("values_t must be a contiguous float64 CUDA tensor")
    rows = 1 if values_t.dim() == 1 else int(values_t.shape[0])
    cols = int(values_t.shape[-1])
    if out_t is None:
        out_t = torch.empty_like(values_t)
    elif out_t.shape != values_t.shape or out_t.dtype != torch.float64 or not out_t.is_contiguous() \
            or out_t.data_ptr() == values_t.data_ptr():
        raise ValueError("out_t must be a distinct contiguous float64 tensor of the same shape")
    if rows * cols == 0:
        return out_t
    solver = _native.solver_for(values_t.device.index)
    _native.check(_native.load().rocco_hip_crossfit_whittaker_baseline_matrix_f64(
        solver.handle, values_t.data_ptr(), rows, cols, float(penalty_lambda), out_t.data_ptr(),
        _dp._stream_ptr(values_t)), "rocco_hip_crossfit_whittaker_baseline_matrix_f64")
    return out_t


def crossfit_whittaker_baseline_batch_device(values_list, penalty_lambda: float, outs=None):
    """The cross-fit baselines of several [rows_i, cols_i] float64 CUDA matrices of ONE penalty -- the chromosomes of a
    genome -- in one pair of launches (rocco_hip_crossfit_whittaker_baseline_batch_f64): every group of 8 rows of every
    matrix is a workgroup whose chain wavefront carries the 16 chains (8 rows x 2 parities) in lockstep, and the pair of
    launches lasts as long as the longest row (27 ns per locus and sweep).  Returns the list of baseline tensors (`outs` when given: distinct tensors of the
    same shapes)."""
    import ctypes

    import torch

    values_list = list(values_list)
    if not values_list:
        return []
    for v in values_list:
        if v.dim() != 2 or v.dtype != torch.float64 or not v.is_cuda or not v.is_contiguous():
            raise ValueError("every matrix must be a contiguous two-dimensional float64 CUDA tensor")
    if outs is None:
        outs = [torch.empty_like(v) for v in values_list]
    for v, o in zip(values_list, outs):
        if o.shape != v.shape or o.dtype != torch.float64 or not o.is_contiguous() or o.data_ptr() == v.data_ptr():
            raise ValueError("every output must be a distinct contiguous float64 tensor of its matrix's shape")
    count = len(values_list)
    solver = _native.solver_for(values_list[0].device.index)
    _native.check(_native.load().rocco_hip_crossfit_whittaker_baseline_batch_f64(
        solver.handle, count, (ctypes.c_void_p * count)(*[v.data_ptr() for v in values_list]),
        (ctypes.c_size_t * count)(*[int(v.shape[0]) for v in values_list]),
        (ctypes.c_size_t * count)(*[int(v.shape[1]) for v in values_list]), float(penalty_lambda),
        (ctypes.c_void_p * count)(*[o.data_ptr() for o in outs]), _dp._stream_ptr(values_list[0])),
        "rocco_hip_crossfit_whittaker_baseline_batch_f64")
    return outs


def crossfit_whittaker_baseline(values, penalty_lambda: float) -> np.ndarray:
    """Same call as the reference's extension (rocco/_baseline.c:16-104): ``values`` 1-D or 2-D,
    returns a float64 array of the same shape with the cross-fit baseline of every row."""
    _native.load()
    arr = np.ascontiguousarray(values, dtype=np.float64)
    if arr.ndim not in (1, 2):
        raise ValueError("`values` must be one-dimensional or two-dimensional")
    if arr.size == 0:
        return np.zeros(arr.shape, dtype=np.float64)
    values_t = _dp._to_device_f64(arr.reshape(-1)).reshape(arr.shape)
    return crossfit_whittaker_baseline_device(values_t, float(penalty_lambda)).cpu().numpy()


def _consenrich_crossfit_whittaker_baseline(y_vals, block_size: int = 101) -> np.ndarray:
    """rocco/inference.py:135-165: broad local baseline of one track (zeros for fewer than 25 loci)."""
    y_arr = np.asarray(y_vals, dtype=np.float64)
    if y_arr.ndim != 1:
        raise ValueError("`y_vals` must be one-dimensional")
    window = _resolve_local_baseline_window(int(y_arr.size), target_window=block_size)
    if window == 0:
        return np.zeros_like(y_arr, dtype=np.float64)
    return crossfit_whittaker_baseline(y_arr, penalty_lambda=_consenrich_whittaker_lambda(window))


def _estimate_local_background_matrix(centered_matrix, target_window: int = 101) -> Tuple[np.ndarray, int, float]:
    """rocco/inference.py:185-229: (local baselines [K, n], window, penalty)."""
    matrix = np.asarray(centered_matrix, dtype=np.float64)
    if matrix.ndim != 2:
        raise ValueError("`centered_matrix` must be two-dimensional")
    _, n_loci = matrix.shape
    window = _resolve_local_baseline_window(n_loci, target_window=target_window)
    if window == 0:
        return np.zeros_like(matrix, dtype=np.float64), 0, 0.0
    penalty_lambda = _consenrich_whittaker_lambda(window)
    local_baselines = crossfit_whittaker_baseline(matrix, penalty_lambda=penalty_lambda)
    if not np.all(np.isfinite(local_baselines)):
        raise ValueError("Local baseline fit produced non-finite values")
    return local_baselines, window, penalty_lambda


def wls_spatial_window(n: int, requested: int = 31) -> int:
    """The spatial window the scoring resolves for a row of n loci (wls_backend.c:229-259): odd, at least 5, at most n;
    0 for fewer than 5 loci."""
    n = int(n)
    if n < 5:
        return 0
    w = int(requested) if int(requested) > 0 else 31
    w = min(max(w, 5), n)
    if w % 2 == 0:
        w = w - 1 if w == n else w + 1
    return w if w >= 5 else 0


def wls_rolling_variances_batch_device(centered_list, spatial_window: int = 31, arena=None):
    """Every row's rolling AR(1) innovation variances (wls_backend.c:610-742) for the rows of SEVERAL centred matrices in
    one launch (rocco_hip_wls_rolling_variances_batch_f64).  Returns one tensor [K_i, n_i - window_i + 1] per matrix
    (None where the scoring does not use them: fewer than 5 loci).  ``arena``: a flat float64 CUDA tensor of at least
    the matrices' values, which the results are then views of (the count-path batch hands each pipeline ONE block for
    its baselines and, after them, its variances: no allocation while the pipelines run)."""
    import ctypes

    import torch

    centered_list = list(centered_list)
    count = len(centered_list)
    outs, used = [], 0
    for c in centered_list:
        K, n = int(c.shape[0]), int(c.shape[1])
        w = wls_spatial_window(n, spatial_window)
        if not (w > 0 and n >= 4):
            outs.append(None)
        elif arena is None:
            outs.append(torch.empty((K, n - w + 1), dtype=torch.float64, device=c.device))
        else:
            outs.append(arena[used:used + K * (n - w + 1)].view(K, n - w + 1))
            used += K * (n - w + 1)
    if count == 0:
        return outs
    solver = _native.solver_for(centered_list[0].device.index)
    _native.check(_native.load().rocco_hip_wls_rolling_variances_batch_f64(
        solver.handle, count, (ctypes.c_void_p * count)(*[c.data_ptr() for c in centered_list]),
        (ctypes.c_size_t * count)(*[int(c.shape[0]) for c in centered_list]),
        (ctypes.c_size_t * count)(*[int(c.shape[1]) for c in centered_list]), int(spatial_window),
        (ctypes.c_void_p * count)(*[(o.data_ptr() if o is not None else None) for o in outs]),
        _dp._stream_ptr(centered_list[0])), "rocco_hip_wls_rolling_variances_batch_f64")
    return outs


def wls_sorted_rows(device_index: int = 0) -> int:
    """Rows of the calling thread's last centred-WLS call on this device whose trend fit took the sorted path
    (rocco_hip_wls_sorted_rows): 0 for matrices of long rows without runs of equal |value| at the bin boundaries."""
    return int(_native.load().rocco_hip_wls_sorted_rows(_native.solver_for(int(device_index)).handle))


def score_centered_wls_device(centered_t, lower_bound_z: float = 1.0, prior_df: float = 5.0, min_effect=None,
                              spatial_window: int = 31, precision_floor_ratio: float = 0.01, variances_t=None):
    """Device-resident form of the reference extension's ``score_centered_wls`` (rocco/_wls.c over
    rocco/native/wls_backend.c:744-947): ``centered_t`` is a contiguous float64 CUDA tensor [K, n]; returns
    (scores, mean, raw_variance, prior_variance, moderated_variance, standard_error, total_df, window)
    with the six tracks as CUDA tensors of n doubles.  ``variances_t``: the rows' rolling variances when a batched launch
    has made them already (`wls_rolling_variances_batch_device`)."""
    import ctypes

    import torch

    if centered_t.dim() != 2:
        raise ValueError("`centered_matrix` must be two-dimensional")
    if centered_t.dtype != torch.float64 or not centered_t.is_cuda or not centered_t.is_contiguous():
        raise ValueError("centered_t must be a contiguous float64 CUDA tensor")
    K, n = int(centered_t.shape[0]), int(centered_t.shape[1])
    if K == 0 or n == 0:
        raise ValueError("`centered_matrix` must be non-empty")
    tracks = torch.empty((6, n), dtype=torch.float64, device=centered_t.device)
    df, win = ctypes.c_double(), ctypes.c_int()
    solver = _native.solver_for(centered_t.device.index)
    _native.check(_native.load().rocco_hip_score_centered_wls_given_variances_f64(
        solver.handle, centered_t.data_ptr(), K, n, float(lower_bound_z), float(prior_df),
        float(0.0 if min_effect is None else min_effect), 0 if min_effect is None else 1, int(spatial_window),
        float(precision_floor_ratio), None if variances_t is None else variances_t.data_ptr(),
        *[tracks[i].data_ptr() for i in range(6)], ctypes.byref(df), ctypes.byref(win),
        _dp._stream_ptr(centered_t)), "rocco_hip_score_centered_wls_f64")
    mean, raw, prior, mod, se, scores = (tracks[i] for i in range(6))
    return scores, mean, raw, prior, mod, se, float(df.value), int(win.value)


def score_centered_wls(centered_matrix, lower_bound_z: float = 1.0, prior_df: float = 5.0, min_effect=None,
                       spatial_window: int = 31, precision_floor_ratio: float = 0.01):
    """Same call as the reference's extension ``rocco._wls.score_centered_wls`` on a host array."""
    _native.load()
    arr = np.ascontiguousarray(centered_matrix, dtype=np.float64)
    if arr.ndim != 2 or arr.shape[0] == 0 or arr.shape[1] == 0:
        raise ValueError("`centered_matrix` must be a non-empty two-dimensional array")
    centered_t = _dp._to_device_f64(arr.reshape(-1)).reshape(arr.shape)
    out = score_centered_wls_device(centered_t, lower_bound_z, prior_df, min_effect, spatial_window,
                                    precision_floor_ratio)
    return tuple(t.cpu().numpy() for t in out[:6]) + out[6:]


def _score_centered_wls_matrix(centered_matrix, lower_bound_z: float = 1.0, prior_df: float = 5.0, min_effect=None,
                               spatial_window=None, precision_floor_ratio: float = 0.01):
    """rocco/inference.py:231-299: (scores, details) of the EB-moderated centred WLS."""
    centered = np.asarray(centered_matrix, dtype=np.float64)
    if centered.ndim != 2:
        raise ValueError("`centered_matrix` must be two-dimensional")
    if centered.shape[0] == 0 or centered.shape[1] == 0:
        raise ValueError("`centered_matrix` must be non-empty")
    precision_floor_ratio_ = float(max(precision_floor_ratio, 0.0))
    scores, mean, raw, prior, mod, se, total_df, window = score_centered_wls(
        centered, lower_bound_z=float(lower_bound_z), prior_df=float(prior_df), min_effect=min_effect,
        spatial_window=31 if spatial_window is None else int(spatial_window),
        precision_floor_ratio=precision_floor_ratio_)
    details = {
        "mean": mean,
        "raw_variance": raw,
        "prior_variance": prior,
        "moderated_variance": mod,
        "standard_error": se,
        "z_scores": mean / np.maximum(se, 1.0e-8),
        "min_effect": float(0.0 if min_effect is None else max(min_effect, 0.0)),
        "precision_floor_ratio": float(precision_floor_ratio_),
        "degrees_of_freedom": np.full(centered.shape[1], float(total_df), dtype=np.float64),
        "prior_spatial_window": float(window),
    }
    if not all(np.all(np.isfinite(a)) for a in (scores, mean, raw, prior, mod, se, details["z_scores"])):
        raise ValueError("EB scoring produced non-finite values")
    return scores, details


def log_scale_device(values_t, pseudocount: float = 1.0):
    """log2(max(values, 0) + pseudocount) of a float64 CUDA tensor, correctly rounded (rocco_hip_log_scale_f64): the
    device form of `_log_scale_wls_matrix` (rocco/inference.py:40-47)."""
    import torch

    if values_t.dtype != torch.float64 or not values_t.is_cuda or not values_t.is_contiguous():
        raise ValueError("values_t must be a contiguous float64 CUDA tensor")
    out_t = torch.empty_like(values_t)
    solver = _native.solver_for(values_t.device.index)
    _native.check(_native.load().rocco_hip_log_scale_f64(solver.handle, values_t.data_ptr(), int(values_t.numel()),
                                                         float(pseudocount), out_t.data_ptr(), _dp._stream_ptr(values_t)),
                  "rocco_hip_log_scale_f64")
    return out_t


def log_scale_center_rows_device(counts_t, pseudocount: float = 1.0, out_t=None, apply_log2: bool = True):
    """rocco/inference.py:40-47 + 330-331 on the device: log2(max(counts, 0) + pseudocount) with every row's
    median subtracted (``apply_log2=False``: the matrix is already log-scaled).  Returns (centred tensor
    [K, n], row medians [K])."""
    import torch

    if counts_t.dim() != 2:
        raise ValueError("`chrom_matrix` must be two-dimensional")
    if counts_t.dtype != torch.float64 or not counts_t.is_cuda or not counts_t.is_contiguous():
        raise ValueError("counts_t must be a contiguous float64 CUDA tensor")
    K, n = int(counts_t.shape[0]), int(counts_t.shape[1])
    if K == 0 or n == 0:
        raise ValueError("`chrom_matrix` must be non-empty")
    if out_t is None:
        out_t = torch.empty_like(counts_t)
    offsets = torch.empty(K, dtype=torch.float64, device=counts_t.device)
    solver = _native.solver_for(counts_t.device.index)
    _native.check(_native.load().rocco_hip_log_scale_center_rows_f64(
        solver.handle, counts_t.data_ptr(), K, n, float(pseudocount), 1 if apply_log2 else 0, out_t.data_ptr(),
        offsets.data_ptr(),
        _dp._stream_ptr(counts_t)), "rocco_hip_log_scale_center_rows_f64")
    return out_t, offsets


def log_scale_row_offsets_device(counts_t, pseudocount: float = 1.0, out_t=None, apply_log2: bool = True):
    """`log_scale_center_rows_device` WITHOUT the subtraction: (log matrix [K, n], row medians [K]) -- for the batched baseline
    sweeps that subtract the medians on their way (`crossfit_whittaker_residual_batch_device`): one pass over the matrix less
    (rocco_hip_log_scale_row_offsets_f64)."""
    import torch

    if counts_t.dim() != 2:
        raise ValueError("`chrom_matrix` must be two-dimensional")
    if counts_t.dtype != torch.float64 or not counts_t.is_cuda or not counts_t.is_contiguous():
        raise ValueError("counts_t must be a contiguous float64 CUDA tensor")
    K, n = int(counts_t.shape[0]), int(counts_t.shape[1])
    if K == 0 or n == 0:
        raise ValueError("`chrom_matrix` must be non-empty")
    if out_t is None:
        out_t = torch.empty_like(counts_t)
    offsets = torch.empty(K, dtype=torch.float64, device=counts_t.device)
    solver = _native.solver_for(counts_t.device.index)
    _native.check(_native.load().rocco_hip_log_scale_row_offsets_f64(
        solver.handle, counts_t.data_ptr(), K, n, float(pseudocount), 1 if apply_log2 else 0, out_t.data_ptr(),
        offsets.data_ptr(), _dp._stream_ptr(counts_t)), "rocco_hip_log_scale_row_offsets_f64")
    return out_t, offsets


def whittaker_batch_scratch_bytes(shapes) -> int:
    """Bytes of scratch the batched sweeps need for matrices of these (rows, cols) shapes (twice the matrices + records)."""
    import ctypes

    shapes = list(shapes)
    count = len(shapes)
    return int(_native.load().rocco_hip_whittaker_batch_scratch_bytes(
        count, (ctypes.c_size_t * count)(*[int(r) for r, _c in shapes]), (ctypes.c_size_t * count)(*[int(c) for _r, c in shapes])))


def crossfit_whittaker_residual_batch_device(values_list, offsets_list, penalty_lambda: float, outs=None, scratch=None):
    """rocco/inference.py:330-338 for several matrices of ONE penalty in the baseline sweeps' own launches
    (rocco_hip_crossfit_whittaker_residual_batch_f64): out_i = (values_i - offsets_i[row]) - baseline(values_i - offsets_i[row]),
    rounded as the reference's separate statements round.  `offsets_list`: one [K_i] tensor or None per matrix (or None for
    all).  Raises ValueError("Local baseline fit produced non-finite values") as the reference does (207-208)."""
    import ctypes

    import torch

    values_list = list(values_list)
    if not values_list:
        return []
    offsets_list = [None] * len(values_list) if offsets_list is None else list(offsets_list)
    for v, o in zip(values_list, offsets_list):
        if v.dim() != 2 or v.dtype != torch.float64 or not v.is_cuda or not v.is_contiguous():
            raise ValueError("every matrix must be a contiguous two-dimensional float64 CUDA tensor")
        if o is not None and (o.dtype != torch.float64 or not o.is_cuda or not o.is_contiguous() or int(o.numel()) != int(v.shape[0])):
            raise ValueError("offsets must be one contiguous float64 CUDA value per row")
    if outs is None:
        outs = [torch.empty_like(v) for v in values_list]
    for v, o in zip(values_list, outs):
        if o.shape != v.shape or o.dtype != torch.float64 or not o.is_contiguous() or o.data_ptr() == v.data_ptr():
            raise ValueError("every output must be a distinct contiguous float64 tensor of its matrix's shape")
    count = len(values_list)
    solver = _native.solver_for(values_list[0].device.index)
    common = (solver.handle, count, (ctypes.c_void_p * count)(*[v.data_ptr() for v in values_list]),
              (ctypes.c_void_p * count)(*[(None if o is None else o.data_ptr()) for o in offsets_list]),
              (ctypes.c_size_t * count)(*[int(v.shape[0]) for v in values_list]),
              (ctypes.c_size_t * count)(*[int(v.shape[1]) for v in values_list]), float(penalty_lambda),
              (ctypes.c_void_p * count)(*[o.data_ptr() for o in outs]))
    if scratch is not None:
        # (`scratch`: a CUDA tensor of at least whittaker_batch_scratch_bytes(...) bytes -- the sweeps' scratch in the caller's pool)
        rc = _native.load().rocco_hip_crossfit_whittaker_residual_batch_scratch_f64(
            *common, scratch.data_ptr(), int(scratch.numel()) * int(scratch.element_size()), _dp._stream_ptr(values_list[0]))
    else:
        rc = _native.load().rocco_hip_crossfit_whittaker_residual_batch_f64(*common, _dp._stream_ptr(values_list[0]))
    if rc == _native.EINVAL and "non-finite" in _native.last_error():
        raise ValueError("Local baseline fit produced non-finite values")
    _native.check(rc, "rocco_hip_crossfit_whittaker_residual_batch_f64")
    return outs


def score_loci_wls_device(counts_t, lower_bound_z: float = 1.0, prior_df: float = 5.0, min_effect=None,
                          precision_floor_ratio: float = 0.01, overwrite_input: bool = False,
                          input_scale: str = "counts"):
    """Device-resident rocco/inference.py:302-379: ``counts_t`` is a contiguous float64 CUDA tensor [K, n].
    Returns (score tensor [n], details) where the details hold CUDA tensors (``centered_matrix`` included).
    ``input_scale="log2p1"`` (not in the reference) takes a matrix that is already log2(count + 1)."""
    import torch

    if input_scale not in ("counts", "log2p1"):
        raise ValueError("input_scale must be 'counts' or 'log2p1'")
    global_centered, _ = log_scale_center_rows_device(counts_t, 1.0, counts_t if overwrite_input else None,
                                                      apply_log2=(input_scale == "counts"))
    K, n = int(global_centered.shape[0]), int(global_centered.shape[1])
    window = _resolve_local_baseline_window(n, target_window=101)
    penalty_lambda = 0.0
    solver = _native.solver_for(counts_t.device.index)
    if window == 0:
        centered = global_centered  # zero baselines (rocco/inference.py:195-196)
    else:
        penalty_lambda = _consenrich_whittaker_lambda(window)
        baselines = crossfit_whittaker_baseline_device(global_centered, penalty_lambda)
        if not bool(torch.isfinite(baselines).all()):
            raise ValueError("Local baseline fit produced non-finite values")
        _native.check(_native.load().rocco_hip_subtract_f64(
            solver.handle, global_centered.data_ptr(), baselines.data_ptr(), global_centered.data_ptr(), K * n,
            _dp._stream_ptr(global_centered)), "rocco_hip_subtract_f64")
        centered = global_centered
        del baselines
    precision_floor_ratio_ = float(max(precision_floor_ratio, 0.0))
    scores, mean, raw, prior, mod, se, total_df, resolved_window = score_centered_wls_device(
        centered, lower_bound_z=float(lower_bound_z), prior_df=float(prior_df), min_effect=min_effect,
        spatial_window=31, precision_floor_ratio=precision_floor_ratio_)
    z_scores = mean / torch.clamp_min(se, 1.0e-8)
    tracks = torch.stack([scores, mean, raw, prior, mod, se, z_scores])
    if not bool(torch.isfinite(tracks).all()):
        raise ValueError("EB scoring produced non-finite values")
    details = {
        "input_scale": "log2p1",
        "local_baseline_window": int(window),
        "local_baseline_lambda": float(penalty_lambda),
        "mean": mean,
        "raw_variance": raw,
        "prior_variance": prior,
        "moderated_variance": mod,
        "standard_error": se,
        "z_scores": z_scores,
        "min_effect": float(0.0 if min_effect is None else max(min_effect, 0.0)),
        "precision_floor_ratio": float(precision_floor_ratio_),
        "prior_spatial_window": int(resolved_window),
        "degrees_of_freedom": torch.full((n,), float(total_df), dtype=torch.float64, device=counts_t.device),
        "centered_matrix": centered,
    }
    return scores, details


# ---- the count-path scoring of several chromosomes at once ---------------------------------------------------------
# Pipelines of the batch: one (solver handle, HIP stream, host thread) each, created on first use and kept -- the native
# calls synchronise their stream and release the GIL, so the pipelines overlap on the device.
_batch_lock = threading.Lock()  # one batch at a time per process: the pipelines and their scratch are shared
_batch_pool = None
_batch_workers = {}
# The pipelines' big blocks -- per pipeline one for baselines / variances, one for the sweeps' scratch -- are KEPT between
# calls (round 5): allocating a K = 100 genome's 150 GB takes the runtime 4-6 s, ten times what the scoring takes, and a
# caching allocator that has meanwhile cut them up for other tensors allocates them afresh.  Between scoring calls a caller
# may borrow them (`borrow_batch_blocks`: the composed driver's budget null computes its draws in them).
_batch_blocks = {}        # device index -> [tensor, ...] (arenas first, then the sweeps' scratch blocks)
_batch_blocks_lent = set()
last_batch_growths_in_flight = 0  # solver buffers that grew while the last batch's pipelines were running (must be 0)


def _batch_worker(device_index: int, slot: int):
    import torch

    key = (int(device_index), int(slot))
    if key not in _batch_workers:
        _batch_workers[key] = (_native.Solver(int(device_index)), torch.cuda.Stream(device=int(device_index)))
    return _batch_workers[key]


def release_batch_workers() -> None:
    """Destroy the pipelines' solver handles and with them their scratch buffers (tens of GB after a genome-sized batch: the
    grouped baseline and rolling launches keep their scratch for the next call).  The next batch creates them again."""
    global _batch_workers
    with _batch_lock:
        workers, _batch_workers = _batch_workers, {}
    for solver, stream in workers.values():
        stream.synchronize()
        solver.close()
    with _batch_lock:
        _batch_blocks.clear()


def batch_scratch_bytes() -> int:
    """Device memory the pipelines' solvers hold between calls (`release_batch_workers` hands it back)."""
    with _batch_lock:
        return sum(int(_native.load().rocco_hip_solver_device_bytes(solver.handle)) for solver, _stream in _batch_workers.values())


class BlockCarver:
    """First-fit carving of flat float64 tensors out of a few big ones (no allocation, no freeing: `reset` starts over)."""

    def __init__(self, pieces):
        self.pieces = [p.reshape(-1) for p in pieces]
        self.used = [0] * len(self.pieces)
        self.base = [0] * len(self.pieces)

    def reset(self):
        """Everything taken since the last `mark` (or `clear`) is given up."""
        self.used = list(self.base)

    def mark(self):
        """What has been taken so far stays taken across `reset`."""
        self.base = list(self.used)

    def clear(self):
        self.base = [0] * len(self.pieces)
        self.used = [0] * len(self.pieces)

    def capacity(self) -> int:
        return sum(int(p.numel()) for p in self.pieces)

    def fits(self, requests) -> bool:
        """Whether these element counts, taken in this order, would all be served."""
        used = list(self.used)
        for want in requests:
            for j, p in enumerate(self.pieces):
                if int(p.numel()) - used[j] >= int(want):
                    used[j] += (int(want) + 63) // 64 * 64  # (as `take` steps)
                    break
            else:
                return False
        return True

    def take(self, numel: int):
        numel = int(numel)
        for j, p in enumerate(self.pieces):
            if int(p.numel()) - self.used[j] >= numel:
                out = p[self.used[j]:self.used[j] + numel]
                self.used[j] += (numel + 63) // 64 * 64  # (512-byte steps)
                return out
        raise MemoryError("the borrowed blocks do not hold this request")


def borrow_batch_blocks(device_index: int):
    """The blocks the last batch scoring on this device kept (flat float64 CUDA tensors; [] when there are none or they are
    lent already).  Until `return_batch_blocks` a scoring call allocates blocks of its own."""
    with _batch_lock:
        if int(device_index) in _batch_blocks_lent:
            return []
        blocks = list(_batch_blocks.get(int(device_index), []))
        if blocks:
            _batch_blocks_lent.add(int(device_index))
        return blocks


def return_batch_blocks(device_index: int) -> None:
    with _batch_lock:
        _batch_blocks_lent.discard(int(device_index))


def drop_batch_blocks(device_index: Optional[int] = None) -> None:
    """The blocks a `keep_blocks=True` scoring call kept go back to the allocator (its cache: a next call of the same
    shapes gets them again without asking the runtime)."""
    with _batch_lock:
        for index in ([int(device_index)] if device_index is not None else list(_batch_blocks)):
            if index not in _batch_blocks_lent:
                _batch_blocks.pop(index, None)


def score_loci_wls_batch_device(counts_list, lower_bound_z: float = 1.0, prior_df: float = 5.0, min_effect=None,
                                precision_floor_ratio: float = 0.01, overwrite_input: bool = False,
                                input_scale: str = "counts", workers: int = 2, memory_budget_bytes: Optional[int] = None,
                                keep_blocks: bool = False, reserve_bytes: int = 0):
    """`score_loci_wls_device` for several [K_i, n_i] float64 CUDA count matrices -- the chromosomes a rank owns
    (the loop of rocco/rocco.py:948-1018 around rocco/inference.py:302-379).  Returns one (scores, details) pair per
    matrix, bit for bit what the single-matrix call returns.  What is shared: the matrices are dealt to `workers`
    pipelines (host thread + stream + solver handle each, at most `_native.max_side_streams()`: one hardware queue per
    stream, see there); inside a pipeline every matrix whose local-baseline window gives the same Whittaker penalty (all
    of 101 loci or more) has its baselines fitted in ONE pair of launches (`crossfit_whittaker_baseline_batch_device`:
    the pair lasts as long as the longest row, not as long as all rows one after the other) and the rolling variances
    of every row come from ONE launch (`wls_rolling_variances_batch_device`); the per-matrix steps (log scale + row
    medians; rank finding, dealing, selects and accumulation of the trend fit) are launches over whole matrices.
    Measured (MI355X, 24 chromosomes of 50 bp loci, K = 100, 6.2e9 values), round 5: 0.42-0.43 s with one pipeline (log scale
    and row medians 0.06, baselines with both subtractions folded into their sweeps 0.09 -- rows cut into verified segments,
    whittaker.hip --, rolling sums 0.11, trend fits 0.16), 0.40-0.42 s with two of equal parts, the second starting when the
    first has its baselines behind it (the default: its bandwidth-bound phases then run under the first one's rolling launch,
    the one phase left that lasts as long as its longest row) (round 4: 0.59, of which the baselines' longest rows were 0.29).

    Memory: beside its matrices a pipeline holds, for the chunk of C bytes it works on, ONE block of C bytes (the log matrix,
    then in its place the rolling variances) and 2 C for the forward sweep's two parities -- both blocks of PyTorch's allocator
    since round 5 (`keep_blocks=True` keeps them for `borrow_batch_blocks` and the next call; `reserve_bytes`: memory the caller
    needs afterwards, left out of the plan) -- and the solver's own scratch for one matrix's trend fit.  A pipeline therefore walks its matrices in CHUNKS whose size the device's free memory allows
    (`memory_budget_bytes`: what the whole call may hold beside the inputs and the tracks it returns; default 88 % of what is
    free or cached when it starts): everything at once when that fits (the K = 100 genome: 49 GB of matrices, ~200 GB in
    all), several chunks one after the other when it does not (K = 50 at 10 bp, 123 GB of matrices, centred in place)."""
    import concurrent.futures

    import torch

    global _batch_pool
    counts_list = list(counts_list)
    if not counts_list:
        return []
    if input_scale not in ("counts", "log2p1"):
        raise ValueError("input_scale must be 'counts' or 'log2p1'")
    for c in counts_list:
        if c.dim() != 2 or c.dtype != torch.float64 or not c.is_cuda or not c.is_contiguous():
            raise ValueError("every matrix must be a contiguous two-dimensional float64 CUDA tensor")
        if int(c.shape[0]) == 0 or int(c.shape[1]) == 0:
            raise ValueError("`chrom_matrix` must be non-empty")
    device = counts_list[0].device
    # (`overwrite_input`: one flag, or one per matrix -- a caller that owns only some of them)
    overwrite = [bool(v) for v in overwrite_input] if isinstance(overwrite_input, (list, tuple)) else [bool(overwrite_input)] * len(counts_list)
    if len(overwrite) != len(counts_list):
        raise ValueError("`overwrite_input` must be one flag or one per matrix")
    workers = max(1, min(int(workers), len(counts_list), _native.max_side_streams()))  # never more streams than hardware queues
    caller_stream = torch.cuda.current_stream(device)
    sizes = [int(c.shape[0]) * int(c.shape[1]) for c in counts_list]
    order = sorted(range(len(counts_list)), key=lambda i: -sizes[i])
    # Pipelines: the matrices are dealt, longest first, to `workers` groups holding 1, 2, ... parts in 1 + 2 + ... of the
    # values.  Each group is one host thread, stream and solver walking ITS matrices through every phase; the chains of a
    # group's longest row set how long its baseline and rolling launches last while using a fraction of the device, and
    # the bandwidth-bound phases of the other groups (log scale and row medians before, rank finding, dealing, selects
    # and accumulation after) run under them.  The group with the longest rows therefore holds the fewest values.
    # (round 5: equal parts -- the pipelines are staggered on their baselines, see run_group_body; until the baselines were cut
    # into segments the parts were 1 : 2 : 3 with the longest rows in the smallest)
    shares = [1 for _g in range(workers)]
    if os.environ.get("ROCCO_BATCH_SHARES"):  # (experiments: other splits, e.g. "2,3,4")
        given = [float(x) for x in os.environ["ROCCO_BATCH_SHARES"].split(",")]
        if len(given) == workers and all(x > 0 for x in given):
            shares = given
    bounds = [sum(shares[:g + 1]) / float(sum(shares)) for g in range(workers)]
    groups, g, running, total = [[] for _ in range(workers)], 0, 0, float(sum(sizes))
    for i in order:
        groups[g].append(i)
        running += sizes[i]
        if g < workers - 1 and running >= bounds[g] * total:
            g += 1
    groups = [idx for idx in groups if idx]
    result = [None] * len(counts_list)
    trace = os.environ.get("ROCCO_BATCH_TRACE")
    import time as _time

    def run_group(slot, idx, start, t_start):
        solver, stream = _batch_worker(device.index, slot)

        def stamp(what):
            if trace:
                stream.synchronize()
                print(f"[batch] group {slot} ({len(idx)} matrices): {what} at {1e3 * (_time.perf_counter() - t_start):.1f} ms", flush=True)

        try:
            run_group_body(slot, idx, start, solver, stream, stamp)
        finally:
            baselines_done[slot].set()  # (whatever happened: the next pipeline must not wait for ever)

    def run_group_body(slot, idx, start, solver, stream, stamp):
        with torch.cuda.device(device), torch.cuda.stream(stream), _native.use_solver(solver):
            stream.wait_event(start)
            if stagger and slot > 0:
                # Round 5: since the baselines are cut into segments their launches fill the device, and the one phase that
                # does not is the rolling sums (the latency of the longest row): a pipeline starts when the one before it
                # has its baselines behind it, so that ITS bandwidth-bound phases run under the other's rolling launch
                baselines_done[slot - 1].wait()
                stream.wait_event(baselines_event[slot - 1])
            for k, part in enumerate(chunks_of[slot]):
                run_chunk(solver, stream, part, stamp, slot, first=(k == 0))
            if not baselines_done[slot].is_set():  # (a pipeline without a baseline phase still lets the next one go)
                baselines_event[slot].record(stream)
                baselines_done[slot].set()
            stream.synchronize()
            stamp("done")

    def run_chunk(solver, stream, idx, stamp, slot, first=True):
        windows = {i: _resolve_local_baseline_window(int(counts_list[i].shape[1]), target_window=101) for i in idx}
        penalties = {i: (0.0 if windows[i] == 0 else _consenrich_whittaker_lambda(windows[i])) for i in idx}
        arena = arenas[slot]
        if fused:
            # Round 5: the two elementwise statements around the baseline fit (330-331: minus the row medians; 338: minus the
            # baselines) ride on the sweeps.  phase 1: log scale and row medians (325, 330), the LOG matrix into the pipeline's
            # block; phase 2: the sweeps read it minus the medians and write the centred matrix -- into the caller's tensor when it
            # may be overwritten, a fresh one otherwise -- 40 bytes per value and 24 waits for the stream less per genome
            logs, offsets, centred, used = {}, {}, {}, 0
            for i in idx:
                k_i, n_i = int(counts_list[i].shape[0]), int(counts_list[i].shape[1])
                if windows[i] == 0:  # (too short for a baseline, 198-199: the centred matrix is the log matrix minus its medians)
                    centred[i] = log_scale_center_rows_device(counts_list[i], 1.0, counts_list[i] if overwrite[i] else None,
                                                              apply_log2=(input_scale == "counts"))[0]
                    continue
                view = arena[used:used + k_i * n_i].view(k_i, n_i)
                used += k_i * n_i
                logs[i], offsets[i] = log_scale_row_offsets_device(counts_list[i], 1.0, view, apply_log2=(input_scale == "counts"))
            stamp("baselines start")
            for lam in sorted({penalties[i] for i in idx if windows[i] != 0}):
                same = [i for i in idx if windows[i] != 0 and penalties[i] == lam]
                outs = [counts_list[i] if overwrite[i] else torch.empty_like(counts_list[i]) for i in same]
                crossfit_whittaker_residual_batch_device([logs[i] for i in same], [offsets[i] for i in same], lam, outs=outs,
                                                         scratch=scratches[slot])
                centred.update(zip(same, outs))
            del logs, offsets
        else:
            # phase 1: log scale, pilot offset (rocco/inference.py:325, 333-334)
            centred = {i: log_scale_center_rows_device(counts_list[i], 1.0, counts_list[i] if overwrite[i] else None,
                                                       apply_log2=(input_scale == "counts"))[0] for i in idx}
            stamp("baselines start")
            # phase 2: local baselines (335), the group's matrices of one penalty together, and their subtraction (338)
            for lam in sorted({penalties[i] for i in idx if windows[i] != 0}):
                same = [i for i in idx if windows[i] != 0 and penalties[i] == lam]
                views, used = [], 0
                for i in same:  # (the pipeline's one block: the baselines now, the rolling variances after them)
                    k_i, n_i = int(centred[i].shape[0]), int(centred[i].shape[1])
                    views.append(arena[used:used + k_i * n_i].view(k_i, n_i))
                    used += k_i * n_i
                baselines = crossfit_whittaker_baseline_batch_device([centred[i] for i in same], lam, outs=views)
                for i, b in zip(same, baselines):
                    c = centred[i]
                    rc = _native.load().rocco_hip_subtract_finite_f64(solver.handle, c.data_ptr(), b.data_ptr(), c.data_ptr(),
                                                                      int(c.shape[0]) * int(c.shape[1]), stream.cuda_stream)
                    if rc == _native.EINVAL:
                        raise ValueError("Local baseline fit produced non-finite values")
                    _native.check(rc, "rocco_hip_subtract_finite_f64")
                del baselines
        if first and not baselines_done[slot].is_set():
            baselines_event[slot].record(stream)
            baselines_done[slot].set()
        stamp("rolling variances start")
        # phase 3: the centred WLS (342-348): the rolling variances of every row of the group in one launch (one
        # workgroup per row), then rank finding, trend fits and accumulation matrix by matrix
        variances = dict(zip(idx, wls_rolling_variances_batch_device([centred[i] for i in idx], spatial_window=31, arena=arena)))
        stamp("trend fits and accumulation start")
        for i in idx:
            c = centred[i]
            n = int(c.shape[1])
            floor_ratio = float(max(precision_floor_ratio, 0.0))
            scores, mean, raw, prior, mod, se, total_df, resolved_window = score_centered_wls_device(
                c, lower_bound_z=float(lower_bound_z), prior_df=float(prior_df), min_effect=min_effect, spatial_window=31,
                precision_floor_ratio=floor_ratio, variances_t=variances.pop(i))
            z_scores = mean / torch.clamp_min(se, 1.0e-8)
            if not bool(torch.isfinite(torch.stack([scores, mean, raw, prior, mod, se, z_scores])).all()):
                raise ValueError("EB scoring produced non-finite values")
            details = {
                "input_scale": "log2p1", "local_baseline_window": int(windows[i]), "local_baseline_lambda": float(penalties[i]),
                "mean": mean, "raw_variance": raw, "prior_variance": prior, "moderated_variance": mod, "standard_error": se,
                "z_scores": z_scores, "min_effect": float(0.0 if min_effect is None else max(min_effect, 0.0)),
                "precision_floor_ratio": floor_ratio, "prior_spatial_window": int(resolved_window),
                "degrees_of_freedom": torch.full((n,), float(total_df), dtype=torch.float64, device=c.device),
                "centered_matrix": c,
            }
            for t in (scores, mean, raw, prior, mod, se, z_scores, details["degrees_of_freedom"], c):
                t.record_stream(caller_stream)
            result[i] = (scores, details)
        del centred, variances

    with _batch_lock:
        if _batch_pool is None:
            _batch_pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(4, _native.max_side_streams()),
                                                                thread_name_prefix="rocco-count")
        # Every pipeline's solver is sized HERE, on the calling thread, for the matrices it is about to see (scratch of the
        # baseline sweeps, the rolling task table, the trend fit's scratch, the Whittaker factor of its longest row): no
        # worker thread allocates or frees device memory -- both synchronise the whole device -- while the others are in flight.
        # chunks: beside the inputs a pipeline holds, for the chunk of C bytes it works on, C of tensors (the baselines, then
        # the rolling variances in their place) and 2 C of solver scratch (the forward sweep's two parities: the backward
        # sweep's segments read beyond their own ends, so it does not run in place); the solver keeps its scratch
        # between calls, so only what it has to GROW counts against what is free
        fused = os.environ.get("ROCCO_BATCH_FUSED_RESIDUAL", "1") != "0"  # (0: medians and baselines subtracted by passes of their own)
        torch.cuda.synchronize(device)
        free_now, _total = torch.cuda.mem_get_info(device)
        cached = max(0, int(torch.cuda.memory_reserved(device)) - int(torch.cuda.memory_allocated(device)))
        holds = [int(_native.load().rocco_hip_solver_device_bytes(_batch_worker(device.index, slot)[0].handle))
                 for slot in range(len(groups))]
        lent = device.index in _batch_blocks_lent
        kept = [] if lent else _batch_blocks.get(device.index, [])
        kept_bytes = sum(8 * int(t.numel()) for t in kept)  # (ours to use again)
        # (`reserve_bytes`: device memory the caller needs for what it does next with the results -- left out of this call's plan)
        budget = (int(0.88 * (free_now + cached + kept_bytes)) - int(max(0, reserve_bytes))) if memory_budget_bytes is None else int(memory_budget_bytes)
        copies = sum(8 * sizes[i] for i in range(len(counts_list)) if not overwrite[i])
        # (what the call returns stays allocated: eight tracks of n doubles per matrix -- 20 GB for 309 M loci -- + slack)
        results = sum(10 * 8 * int(c.shape[1]) for c in counts_list) + (2 << 30)
        everything = max(1, sum(8 * sizes[i] for idx in groups for i in idx))
        chunks_of = []
        for slot, idx in enumerate(groups):
            # (a pipeline's share of the budget is its share of the values)
            per_group = max(0, budget - copies - (results if memory_budget_bytes is None else 0)) * sum(8 * sizes[i] for i in idx) // everything

            def need(chunk_bytes, largest_bytes, slot=slot):
                if fused:
                    # the chunk's block + the sweeps' scratch (2 C + records), both the allocator's; the solver's own scratch is
                    # what ONE matrix's trend fit takes (its dealt values, tables: ~1.3 x the chunk's largest matrix)
                    grow = int(1.3 * largest_bytes) + (64 << 20) - (holds[slot] if memory_budget_bytes is None else 0)
                    return chunk_bytes + int(2.1 * chunk_bytes) + max(0, grow)
                grow = int(2.2 * chunk_bytes) + (64 << 20) - (holds[slot] if memory_budget_bytes is None else 0)  # (2 C + 1/16 + records)
                return chunk_bytes + max(0, grow)

            parts, part, held = [], [], 0
            for i in idx:  # (longest first: a chunk's first matrix holds its longest rows)
                if part and need(held + 8 * sizes[i], 8 * sizes[part[0]]) > per_group:
                    parts.append(part)
                    part, held = [], 0
                part.append(i)
                held += 8 * sizes[i]
            parts.append(part)
            chunks_of.append(parts)
        if trace:
            print(f"[batch] {len(groups)} pipelines, chunks per pipeline {[len(p) for p in chunks_of]}, budget {budget / 1e9:.1f} GB", flush=True)
        # each pipeline's block for its baselines and (after them) its rolling variances: its largest chunk, allocated here
        # ... and, round 5, the baseline sweeps' scratch (the forward sweep's two parities: twice the chunk) as a block of the
        # framework's allocator too, not of the solver's: what this call leaves idle when it returns is then there for the
        # caller's next step (the composed driver's budget null computes several draws at once in it) instead of sitting in
        # a pool the allocator cannot see
        wanted = [max(sum(sizes[i] for i in part) for part in parts) for parts in chunks_of]
        if fused:
            wanted += [(max(whittaker_batch_scratch_bytes([tuple(counts_list[i].shape) for i in part]) for part in parts) + 7) // 8
                       for parts in chunks_of]

        def make_arenas():
            # (the blocks kept from the last call serve again where they are large enough -- largest wish, largest block)
            pool = sorted(kept, key=lambda t: -int(t.numel()))
            made = [None] * len(wanted)
            for j in sorted(range(len(wanted)), key=lambda j: -wanted[j]):
                if pool and int(pool[0].numel()) >= wanted[j]:
                    made[j] = pool.pop(0)
            del pool
            for j in range(len(wanted)):
                if made[j] is None:
                    made[j] = torch.empty(wanted[j], dtype=torch.float64, device=device)
            return made

        try:
            blocks = make_arenas()
        except torch.OutOfMemoryError:
            # (the budget counted what the allocator caches as usable: cached blocks of other sizes are not -- hand them back,
            # the kept blocks that did not serve included)
            kept = []
            if not lent:
                _batch_blocks.pop(device.index, None)
            torch.cuda.empty_cache()
            blocks = make_arenas()
        if not lent:
            _batch_blocks.pop(device.index, None)  # (this call owns them now; they are registered again when it has come through)
        arenas, scratches = blocks[:len(chunks_of)], (blocks[len(chunks_of):] if fused else [None] * len(chunks_of))
        del blocks, kept
        for slot, parts in enumerate(chunks_of):
            solver, _stream = _batch_worker(device.index, slot)
            for idx in parts:
                rows_a = (ctypes.c_size_t * len(idx))(*[int(counts_list[i].shape[0]) for i in idx])
                cols_a = (ctypes.c_size_t * len(idx))(*[int(counts_list[i].shape[1]) for i in idx])
                _native.check(_native.load().rocco_hip_count_path_reserve_ex(solver.handle, len(idx), rows_a, cols_a, 0.0, 1 if fused else 0,
                                                                             caller_stream.cuda_stream), "rocco_hip_count_path_reserve")
                # ... and the Whittaker factor of every penalty the chunk holds (contigs of 25 .. 100 loci have windows -- and
                # penalties -- of their own): the device keeps one factor per penalty, all of them built here
                by_lam = {}
                for i in idx:
                    w = _resolve_local_baseline_window(int(counts_list[i].shape[1]), target_window=101)
                    if w != 0:
                        by_lam.setdefault(_consenrich_whittaker_lambda(w), []).append(i)
                for lam, same in by_lam.items():
                    r_s = (ctypes.c_size_t * len(same))(*[int(counts_list[i].shape[0]) for i in same])
                    c_s = (ctypes.c_size_t * len(same))(*[int(counts_list[i].shape[1]) for i in same])
                    _native.check(_native.load().rocco_hip_count_path_reserve_ex(solver.handle, len(same), r_s, c_s, float(lam), 1 if fused else 0,
                                                                                 caller_stream.cuda_stream), "rocco_hip_count_path_reserve")
        import threading

        stagger = os.environ.get("ROCCO_BATCH_STAGGER", "1") != "0" and len(groups) > 1

        baselines_event = [torch.cuda.Event() for _ in groups]
        baselines_done = [threading.Event() for _ in groups]
        start = torch.cuda.Event()
        start.record(caller_stream)
        t_start = _time.perf_counter()
        global last_batch_growths_in_flight
        grown_before = int(_native.load().rocco_hip_buffer_growths())
        futures = [_batch_pool.submit(run_group, slot, idx, start, t_start) for slot, idx in enumerate(groups)]
        first_error = None
        for f in futures:
            try:
                f.result()
            except BaseException as exc:  # noqa: BLE001
                first_error = first_error or exc
        last_batch_growths_in_flight = int(_native.load().rocco_hip_buffer_growths()) - grown_before
        if keep_blocks and not lent and first_error is None:
            # (`keep_blocks`: the pipelines' big blocks stay for `borrow_batch_blocks` and the next call; otherwise -- and after
            # any error -- they go back to the allocator's cache here, whole, where the next call finds them again)
            _batch_blocks[device.index] = list(arenas) + [t for t in scratches if t is not None]
        del arenas, scratches
        if first_error is not None:
            raise first_error
        return result


def score_loci_wls(chrom_matrix, lower_bound_z: float = 1.0, prior_df: float = 5.0, min_effect=None,
                   precision_floor_ratio: float = 0.01, low_memory: bool = False, return_details: bool = False,
                   input_scale: str = "counts", resident: bool = False):
    """rocco/inference.py:302-379 with the same signature, NumPy in and out (``input_scale``: see
    ``score_loci_wls_device``).  ``chrom_matrix`` may also be a CUDA tensor.  ``resident=True`` (not in the
    reference; what the chromosome cache asks for) leaves every track and the centred matrix in HBM and returns CUDA
    tensors -- the centred matrix as float32 under ``low_memory``, as the reference stores it."""
    import torch

    _native.load()
    if _dp._is_tensor(chrom_matrix):
        if chrom_matrix.ndim != 2:
            raise ValueError("`chrom_matrix` must be two-dimensional")
        if chrom_matrix.shape[0] == 0 or chrom_matrix.shape[1] == 0:
            raise ValueError("`chrom_matrix` must be non-empty")
        counts_t = chrom_matrix if chrom_matrix.is_cuda else chrom_matrix.to(f"cuda:{_dp._device_index()}")
        owned = counts_t.dtype != torch.float64 or not counts_t.is_contiguous() or counts_t is not chrom_matrix
        counts_t = counts_t.to(torch.float64).contiguous()
        if not bool(torch.isfinite(counts_t).all()):
            raise ValueError("`chrom_matrix` contains non-finite values")
    else:
        matrix = np.ascontiguousarray(chrom_matrix, dtype=np.float64)
        if np.any(~np.isfinite(matrix)):
            raise ValueError("`chrom_matrix` contains non-finite values")  # rocco/inference.py:45-46
        if matrix.ndim != 2:
            raise ValueError("`chrom_matrix` must be two-dimensional")
        if matrix.shape[0] == 0 or matrix.shape[1] == 0:
            raise ValueError("`chrom_matrix` must be non-empty")
        counts_t = _dp._to_device_f64(matrix.reshape(-1)).reshape(matrix.shape)
        owned = True
    scores_t, details_t = score_loci_wls_device(counts_t, lower_bound_z=lower_bound_z, prior_df=prior_df,
                                                min_effect=min_effect, precision_floor_ratio=precision_floor_ratio,
                                                overwrite_input=owned, input_scale=input_scale)
    if resident:
        if low_memory:
            details_t["centered_matrix"] = details_t["centered_matrix"].to(torch.float32)
        return (scores_t, details_t) if return_details else scores_t
    scores = scores_t.cpu().numpy()
    if not return_details:
        return scores
    details = {}
    for key, value in details_t.items():
        if isinstance(value, torch.Tensor):
            if key == "centered_matrix" and low_memory:
                value = value.to(torch.float32)
            details[key] = value.cpu().numpy()
        else:
            details[key] = value
    return scores, details


# --------------------------------------------------------------------------------------------
# the K x n part of the wild-bootstrap budget null (rocco/inference.py:628-722)
# --------------------------------------------------------------------------------------------

def numpy_sum_device(x_t) -> float:
    """np.sum of a contiguous one-dimensional float64 CUDA tensor in NumPy's own summation order (bit for bit)."""
    import ctypes

    import torch

    if x_t.dim() != 1 or x_t.dtype != torch.float64 or not x_t.is_cuda or not x_t.is_contiguous():
        raise ValueError("x_t must be a contiguous one-dimensional float64 CUDA tensor")
    out = ctypes.c_double(0.0)
    solver = _native.solver_for(x_t.device.index)
    _native.check(_native.load().rocco_hip_numpy_sum_f64(solver.handle, x_t.data_ptr(), int(x_t.shape[0]),
                                                         ctypes.byref(out), _dp._stream_ptr(x_t)),
                  "rocco_hip_numpy_sum_f64")
    return float(out.value)


def fit_budget_null_residual_template_device(centered_t, lower_bound_z: float = 1.0, prior_df: float = 5.0,
                                             min_effect=None, precision_floor_ratio: float = 0.01, residual_out=None):
    """rocco/inference.py:688-722 on the device: (residual_template [K, n], observed_scores [n],
    positive_consensus [n]) with residual = centered - max(mu_hat, 0)."""
    import torch

    scores, mean, _raw, _prior, _mod, _se, _df, _win = score_centered_wls_device(
        centered_t, lower_bound_z=float(lower_bound_z), prior_df=float(prior_df), min_effect=min_effect,
        spatial_window=31, precision_floor_ratio=float(max(precision_floor_ratio, 0.0)))
    K, n = int(centered_t.shape[0]), int(centered_t.shape[1])
    residual = residual_out if residual_out is not None else torch.empty_like(centered_t)  # (`residual_out`: the caller's [K, n] block)
    if residual.shape != centered_t.shape or residual.dtype != torch.float64 or not residual.is_contiguous():
        raise ValueError("`residual_out` must be a contiguous float64 tensor shaped as the centred matrix")
    solver = _native.solver_for(centered_t.device.index)
    _native.check(_native.load().rocco_hip_subtract_positive_row_f64(
        solver.handle, centered_t.data_ptr(), mean.data_ptr(), K, n, residual.data_ptr(),
        _dp._stream_ptr(centered_t)), "rocco_hip_subtract_positive_row_f64")
    return residual, scores, torch.clamp_min(mean, 0.0)


def compute_budget_null_draws_device(residual_template_t, weights_list, lower_bound_z: float, prior_df: float, min_effect,
                                     precision_floor_ratio: float, null_center: float, null_soft_scale: float,
                                     null_threshold: float, variances_arena=None):
    """Several draws of the budget null at once (round 5): what `compute_budget_null_draw_device` computes for each of
    `weights_list` (one [K, n] float64 CUDA tensor of multipliers per draw; they are OVERWRITTEN by the draws' products),
    with the rolling variances of every draw's K rows in ONE launch -- that launch lasts as long as one row whatever the
    number of rows (csrc/wls.hip), and it is most of a draw: four draws of a chromosome cost little more than one.  The
    reference's worker pool computes `num_processes` draws side by side before it looks at its stopping rule
    (rocco/inference.py:871-937), so batching that many changes nothing it decides.  Returns one 4-tuple per draw."""
    import ctypes

    import torch

    K, n = int(residual_template_t.shape[0]), int(residual_template_t.shape[1])
    lib, solver, stream = _native.load(), _native.solver_for(residual_template_t.device.index), _dp._stream_ptr(residual_template_t)
    for w in weights_list:
        if w.shape != residual_template_t.shape or w.dtype != torch.float64 or not w.is_cuda or not w.is_contiguous():
            raise ValueError("every draw's multipliers must be a contiguous float64 CUDA tensor shaped as the residual template")
        _native.check(lib.rocco_hip_multiply_f64(solver.handle, residual_template_t.data_ptr(), w.data_ptr(), w.data_ptr(), K * n, stream),
                      "rocco_hip_multiply_f64")
    variances = wls_rolling_variances_batch_device(weights_list, spatial_window=31, arena=variances_arena)  # (`variances_arena`: the caller's block)
    out = []
    for boot, var in zip(weights_list, variances):
        scores = score_centered_wls_device(boot, lower_bound_z=float(lower_bound_z), prior_df=float(prior_df),
                                           min_effect=(None if min_effect is None else float(max(min_effect, 0.0))), spatial_window=31,
                                           precision_floor_ratio=float(max(precision_floor_ratio, 0.0)), variances_t=var)[0]
        if not bool(torch.isfinite(scores).all()):
            raise ValueError("EB scoring produced non-finite values")
        stats = (ctypes.c_double * 4)()
        _native.check(lib.rocco_hip_budget_null_draw_stats_f64(solver.handle, scores.data_ptr(), n, float(null_center), float(null_soft_scale),
                                                               float(null_threshold), stats, stream), "rocco_hip_budget_null_draw_stats_f64")
        out.append((float(stats[0]), float(stats[1]), float(stats[2]), float(stats[3])))
    return out


def compute_budget_null_draw_device(residual_template_t, wild_weights_t, lower_bound_z: float, prior_df: float,
                                    min_effect, precision_floor_ratio: float, null_center: float,
                                    null_soft_scale: float, null_threshold: float, work_t=None):
    """One draw of the budget null given its multipliers (rocco/inference.py:656-685; the multipliers of 646-664
    come from NumPy's generator on the host): bootstrap = residual_template * wild_weights, WLS rescoring, and the
    four means -- mean positive excess, the same in null-scale units, fraction positive, fraction above the null
    threshold -- summed in NumPy's order."""
    import ctypes

    import torch

    if residual_template_t.shape != wild_weights_t.shape or residual_template_t.dim() != 2:
        raise ValueError("residual template and multipliers must be [K, n] tensors of the same shape")
    for t in (residual_template_t, wild_weights_t):
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("residual template and multipliers must be contiguous float64 CUDA tensors")
    K, n = int(residual_template_t.shape[0]), int(residual_template_t.shape[1])
    boot = work_t if work_t is not None else torch.empty_like(residual_template_t)
    lib, solver, stream = _native.load(), _native.solver_for(residual_template_t.device.index), \
        _dp._stream_ptr(residual_template_t)
    _native.check(lib.rocco_hip_multiply_f64(solver.handle, residual_template_t.data_ptr(), wild_weights_t.data_ptr(),
                                             boot.data_ptr(), K * n, stream), "rocco_hip_multiply_f64")
    scores = score_centered_wls_device(boot, lower_bound_z=float(lower_bound_z), prior_df=float(prior_df),
                                       min_effect=(None if min_effect is None else float(max(min_effect, 0.0))),
                                       spatial_window=31,
                                       precision_floor_ratio=float(max(precision_floor_ratio, 0.0)))[0]
    if not bool(torch.isfinite(scores).all()):
        raise ValueError("EB scoring produced non-finite values")
    stats = (ctypes.c_double * 4)()
    _native.check(lib.rocco_hip_budget_null_draw_stats_f64(solver.handle, scores.data_ptr(), n, float(null_center),
                                                           float(null_soft_scale), float(null_threshold), stats,
                                                           stream), "rocco_hip_budget_null_draw_stats_f64")
    return float(stats[0]), float(stats[1]), float(stats[2]), float(stats[3])
