"""Signal-matrix assembly of rocco/readtracks.py on the MI355X.

File decoding (BAM / bigWig) stays with the reference's readers; what this module replaces is the tail of
``generate_chrom_matrix`` (rocco/readtracks.py:603-633): the union of the tracks' locus starts, the fixed-step
check for bigWig inputs and the scatter of every track's values into the dense K x m matrix -- done in HBM, so
only the per-track (start, value) lists cross PCIe and the matrix is born where the scoring kernels read it.
There is no CPU fallback: without the library or a GPU these raise.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from . import dp as _dp


def assemble_chrom_matrix_device(interval_matrix: Sequence, vals_matrix: Sequence, track_type: str = "bam",
                                 low_memory: bool = False, chromosome: str = "", device=None):
    """``interval_matrix[k]`` / ``vals_matrix[k]``: locus starts (integers) and values of track k (NumPy arrays or
    CUDA tensors).  Returns (common_intervals int64 CUDA tensor [m], matrix CUDA tensor [K, m], float64 or float32
    with ``low_memory``), as rocco/readtracks.py:614-633 does on the host."""
    import torch

    _native.load()
    if len(interval_matrix) != len(vals_matrix):
        raise ValueError("one value list per interval list is required")
    K = len(interval_matrix)
    if K == 0:
        raise ValueError("no tracks")
    dev = torch.device(device) if device is not None else torch.device(f"cuda:{_dp._device_index()}")

    def to_dev(a, dtype):
        if _dp._is_tensor(a):
            return a.to(device=dev, dtype=dtype).contiguous().reshape(-1)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=dtype_np[dtype])).to(dev)

    dtype_np = {torch.int64: np.int64, torch.float64: np.float64}
    ints = [to_dev(a, torch.int64) for a in interval_matrix]
    vals = [to_dev(v, torch.float64) for v in vals_matrix]
    for a, v in zip(ints, vals):
        if a.shape[0] != v.shape[0]:
            raise ValueError("shape mismatch: value array cannot be broadcast to indexing result")  # NumPy's error
    offsets = np.zeros(K + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([int(a.shape[0]) for a in ints])
    total = int(offsets[-1])
    ints_cat = torch.cat(ints) if total else torch.empty(0, dtype=torch.int64, device=dev)
    vals_cat = torch.cat(vals) if total else torch.empty(0, dtype=torch.float64, device=dev)
    lib, solver, stream = _native.load(), _native.solver_for(dev.index), _dp._stream_ptr(ints_cat)
    common_full = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    m, fixed = ctypes.c_size_t(0), ctypes.c_int(1)
    _native.check(lib.rocco_hip_union_intervals(solver.handle, ints_cat.data_ptr(), total, common_full.data_ptr(),
                                                ctypes.byref(m), ctypes.byref(fixed), stream),
                  "rocco_hip_union_intervals")
    common = common_full[: m.value]
    if track_type == "bigwig" and m.value > 1 and not fixed.value:
        raise ValueError(f"bigWig inputs for {chromosome} do not share one fixed binning scheme")
    matrix = torch.empty((K, m.value), dtype=torch.float32 if low_memory else torch.float64, device=dev)
    off_c = (ctypes.c_size_t * (K + 1))(*[int(x) for x in offsets])
    _native.check(lib.rocco_hip_scatter_tracks(solver.handle, common.data_ptr(), m.value, ints_cat.data_ptr(),
                                               vals_cat.data_ptr(), off_c, K, 1 if low_memory else 0,
                                               matrix.data_ptr(), stream), "rocco_hip_scatter_tracks")
    return common, matrix


def assemble_chrom_matrix(interval_matrix: Sequence, vals_matrix: Sequence, track_type: str = "bam",
                          low_memory: bool = False, chromosome: str = "") -> Tuple[np.ndarray, np.ndarray]:
    """NumPy in and out: (common_intervals as ``int`` array, count_matrix), the return value of
    ``generate_chrom_matrix`` (rocco/readtracks.py:633) for tracks already decoded."""
    common_t, matrix_t = assemble_chrom_matrix_device(interval_matrix, vals_matrix, track_type=track_type,
                                                      low_memory=low_memory, chromosome=chromosome)
    return common_t.cpu().numpy().astype(int), matrix_t.cpu().numpy()


_BW_ERRORS = (
    (1, "bigWig values for {f} {c} contain non-finite entries"),
    (2, "bigWig intervals for {f} {c} contain non-positive widths"),
    (4, "bigWig file {f} uses variable-width bins on {c}; ROCCO expects a fixed-width binning scheme"),
    (8, "bigWig starts for {f} {c} are not aligned to a single fixed binning scheme"),
    (16, "bigWig file {f} has overlapping or duplicate bins on {c}"),
)


def bigwig_dense_fill_device(starts, ends, vals, const_scale: float = 1.0, round_digits: int = 5,
                             bigwig_file: str = "", chromosome: str = "", device=None):
    """What ``get_bigwig_chrom_scores`` does with a track's intervals once pyBigWig has returned them
    (rocco/readtracks.py:141-186), on the device: validation (same ValueErrors, same order), dense fill of the
    fixed-step grid between the first and the last start, constant scaling, ``np.round(..., round_digits)``.
    ``starts`` must ascend (as pyBigWig returns them).  Returns (first_start, step, values CUDA tensor)."""
    import torch

    _native.load()
    dev = torch.device(device) if device is not None else torch.device(f"cuda:{_dp._device_index()}")

    def to_dev(a, dtype, np_dtype):
        if _dp._is_tensor(a):
            return a.to(device=dev, dtype=dtype).contiguous().reshape(-1)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np_dtype)).to(dev)

    starts_t, ends_t = to_dev(starts, torch.int64, np.int64), to_dev(ends, torch.int64, np.int64)
    vals_t = to_dev(vals, torch.float64, np.float64)
    count = int(starts_t.shape[0])
    if count == 0 or int(ends_t.shape[0]) != count or int(vals_t.shape[0]) != count:
        raise ValueError("starts, ends and values must be non-empty and of one length")
    lib, solver, stream = _native.load(), _native.solver_for(dev.index), _dp._stream_ptr(starts_t)
    first, step = ctypes.c_longlong(0), ctypes.c_longlong(0)
    n_full, flags = ctypes.c_size_t(0), ctypes.c_int(0)

    def call(out_t, capacity):
        _native.check(lib.rocco_hip_bigwig_dense_fill_f64(
            solver.handle, starts_t.data_ptr(), ends_t.data_ptr(), vals_t.data_ptr(), count, float(const_scale),
            int(round_digits), None if out_t is None else out_t.data_ptr(), capacity, ctypes.byref(first),
            ctypes.byref(step), ctypes.byref(n_full), ctypes.byref(flags), stream), "rocco_hip_bigwig_dense_fill_f64")
        for bit, message in _BW_ERRORS:
            if flags.value & bit:
                raise ValueError(message.format(f=bigwig_file, c=chromosome))

    call(None, 0)
    full_t = torch.empty(int(n_full.value), dtype=torch.float64, device=dev)
    call(full_t, int(n_full.value))
    return int(first.value), int(step.value), full_t


def bigwig_dense_fill(starts, ends, vals, const_scale: float = 1.0, round_digits: int = 5, bigwig_file: str = "",
                      chromosome: str = "") -> Tuple[np.ndarray, np.ndarray]:
    """NumPy in and out: the (full_intervals, rounded full_vals) pair ``get_bigwig_chrom_scores`` returns
    (rocco/readtracks.py:175-186)."""
    first, step, full_t = bigwig_dense_fill_device(starts, ends, vals, const_scale, round_digits, bigwig_file, chromosome)
    full_intervals = np.arange(first, first + step * int(full_t.shape[0]), step, dtype=np.int64)
    return full_intervals.astype(int), full_t.cpu().numpy()
