"""Post-hoc peak scoring: the arithmetic after the counting (rocco/scores.py:120-149,
180-194, 381-639).  The reference counts reads over the peaks and over random background regions with pysam (BAM
work, not reproduced here); given those counts, what remains is per-peak arithmetic -- the signal statistic, its
survival under the empirical null of the peak's length bin, Benjamini-Hochberg q-values, the narrowPeak columns --
and that runs on the device (peakscore.hip).  The last three formatting statements (-log10, rounding to four
decimals, the UCSC score) are NumPy calls on the short per-peak vectors, as in the reference.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence

import numpy as np

from . import _native
from . import dp as _dp


class EmpiricalNull:
    """Finite-sample empirical null of one length bin (rocco/scores.py:120-149): right-tail survival with a plus-one
    correction, and the plain empirical CDF."""

    def __init__(self, values):
        ordered = np.sort(np.asarray(values, dtype=np.float64))
        if ordered.ndim != 1 or ordered.size == 0:
            raise ValueError("`values` must be a non-empty one-dimensional array.")
        self.values, self.size = ordered, int(ordered.size)

    def survival(self, x):
        x_ = np.asarray(x, dtype=np.float64)
        below = np.searchsorted(self.values, x_, side="left")
        out = (self.size - below + 1.0) / (self.size + 1.0)
        return float(out) if x_.ndim == 0 else out

    def evaluate(self, x):
        x_ = np.asarray(x, dtype=np.float64)
        out = np.searchsorted(self.values, x_, side="right") / float(self.size)
        return float(out) if x_.ndim == 0 else out


def _device(arr, dtype):
    import torch

    if _dp._is_tensor(arr):
        t = arr if arr.is_cuda else arr.to(f"cuda:{_dp._device_index()}")
        return t.to(dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(arr)).to(f"cuda:{_dp._device_index()}").to(dtype).contiguous()


def peak_signal_stat_device(counts_t, lengths_t, row_scale: float = 1000.0, pc: float = 1.0, percentile: float = 75.0):
    """`_peak_signal_stat` (rocco/scores.py:180-194) of every row of a [peaks, samples] float64 CUDA tensor."""
    import torch

    P, K = int(counts_t.shape[0]), int(counts_t.shape[1])
    out = torch.empty(P, dtype=torch.float64, device=counts_t.device)
    solver = _native.solver_for(counts_t.device.index)
    _native.check(_native.load().rocco_hip_peak_signal_stat_f64(
        solver.handle, counts_t.data_ptr(), lengths_t.data_ptr(), P, K, float(row_scale), float(pc), float(percentile),
        out.data_ptr(), _dp._stream_ptr(counts_t)), "rocco_hip_peak_signal_stat_f64")
    return out


def _peak_signal_stat(vals, length, row_scale: float = 1000.0, pc: float = 1.0, percentile: float = 75.0) -> float:
    """Same call as the reference's helper, for one peak."""
    import torch

    counts = _device(np.asarray(vals, dtype=np.float64).reshape(1, -1), torch.float64)
    lengths = _device(np.array([float(length)]), torch.float64)
    return float(peak_signal_stat_device(counts, lengths, row_scale, pc, percentile)[0])


def benjamini_hochberg_device(pvals_t):
    """scipy.stats.false_discovery_control(ps, method="bh") (rocco/scores.py:583) on a float64 CUDA tensor."""
    import torch

    m = int(pvals_t.shape[0])
    if m <= 1:
        return pvals_t.clone()
    out = torch.empty_like(pvals_t)
    solver = _native.solver_for(pvals_t.device.index)
    _native.check(_native.load().rocco_hip_bh_adjust_f64(solver.handle, pvals_t.data_ptr(), m, out.data_ptr(), _dp._stream_ptr(pvals_t)),
                  "rocco_hip_bh_adjust_f64")
    return out


def score_peak_counts(count_matrix, lengths, binned_lengths, ecdf_dict: Dict[int, EmpiricalNull], row_scale: float = 1000.0,
                      pc: float = 1.0, ucsc_base: int = 250) -> dict:
    """Everything `score_peaks` computes after its counting and scaling (rocco/scores.py:560-625).

    `count_matrix`: [peaks, samples] scaled counts; `lengths`: peak lengths in bp; `binned_lengths[p]`: the key of the
    peak's length bin in `ecdf_dict` (`_assign_length_bins`); `ecdf_dict[key]`: the bin's EmpiricalNull (or its values).
    Returns the arrays the reference writes: signal values, p-values, q-values, the UCSC score column and the rounded
    -log10 columns."""
    import torch

    counts_t = _device(count_matrix, torch.float64)
    if counts_t.dim() != 2:
        raise ValueError("`count_matrix` must be two-dimensional (peaks x samples)")
    P = int(counts_t.shape[0])
    lengths_h = np.asarray(lengths, dtype=np.float64)
    if lengths_h.shape != (P,):
        raise ValueError("`lengths` must hold one entry per peak")
    keys = sorted(ecdf_dict)
    position = {int(k): i for i, k in enumerate(keys)}
    nulls = [np.sort(np.asarray(getattr(ecdf_dict[k], "values", ecdf_dict[k]), dtype=np.float64)) for k in keys]
    if any(v.size == 0 for v in nulls):
        raise ValueError("`values` must be a non-empty one-dimensional array.")
    offsets = np.concatenate([[0], np.cumsum([v.size for v in nulls])]).astype(np.int64)
    bins_h = np.array([position[int(b)] for b in np.asarray(binned_lengths)], dtype=np.int32)
    sig_t = peak_signal_stat_device(counts_t, _device(lengths_h, torch.float64), row_scale, pc)
    pvals_t = torch.empty_like(sig_t)
    solver = _native.solver_for(counts_t.device.index)
    null_t, off_t, bin_t = _device(np.concatenate(nulls), torch.float64), _device(offsets, torch.int64), _device(bins_h, torch.int32)
    _native.check(_native.load().rocco_hip_ecdf_survival_f64(solver.handle, sig_t.data_ptr(), bin_t.data_ptr(), null_t.data_ptr(),
                                                             off_t.data_ptr(), P, pvals_t.data_ptr(), _dp._stream_ptr(sig_t)),
                  "rocco_hip_ecdf_survival_f64")
    qvals_t = benjamini_hochberg_device(pvals_t)
    sig, pvals, qvals = sig_t.cpu().numpy(), pvals_t.cpu().numpy(), qvals_t.cpu().numpy()
    # narrowPeak columns (rocco/scores.py:604-616): NumPy on the per-peak vectors, as the reference
    bed6 = np.minimum(np.array(ucsc_base + sig / np.quantile(sig, q=0.99) * (1000 - ucsc_base), dtype=int), 1000)
    return {"signal": sig, "pvals": pvals, "qvals": qvals, "bed6_scores": bed6,
            "signal_out": np.round(sig, 4), "pvals_out": np.round(-np.log10(pvals + 1e-10), 4),
            "qvals_out": np.round(-np.log10(qvals + 1e-10), 4)}


def write_scored_peaks(bed_strings: Sequence[str], names: Sequence[str], lengths, scored: dict, output_file: str,
                       summit_offsets: Optional[Dict[str, int]] = None) -> str:
    """The narrowPeak-like rows `score_peaks` writes (rocco/scores.py:618-637)."""
    offsets = summit_offsets or {}
    with open(output_file, "w") as fh:
        for i, peak in enumerate(bed_strings):
            summit = int(offsets.get(names[i], -1))
            if summit >= 0:
                summit = int(np.clip(summit, 0, max(int(lengths[i]) - 1, 0)))
            fh.write(f"{peak}\t{names[i]}\t{scored['bed6_scores'][i]}\t.\t{scored['signal_out'][i]}\t"
                     f"{scored['pvals_out'][i]}\t{scored['qvals_out'][i]}\t{summit}\n")
    return output_file
