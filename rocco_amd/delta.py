"""Python face of the delta-form evaluation entry points (rocco_hip_delta_probe_f64 /
rocco_hip_delta_window_f64): count-only evaluation of the chain solve at several penalties in one
pass, and the joint window evaluation used to certify the final solution.
These replace the reference's one-call-per-penalty use of rocco/_chain_dp.c (rocco/dp.py:113-162)."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence

import numpy as np

from . import _native
from . import dp as _dp


def delta_build_map_device(scores_t, switch_costs, lambda_ref: float, margin: float):
    """Per-chunk binade codes (uint8 CUDA tensor of ceil(n / 32) entries) at penalty lambda_ref."""
    import torch

    n = int(scores_t.shape[0])
    costs_t, gamma = _dp._costs_arg(switch_costs, n, scores_t.device)
    emap_t = torch.empty((n + 31) // 32, dtype=torch.uint8, device=scores_t.device)
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_delta_build_map_f64(
        solver.handle, scores_t.data_ptr(), costs_t.data_ptr() if costs_t is not None else None, gamma, n,
        float(lambda_ref), float(margin), emap_t.data_ptr(), _dp._stream_ptr(scores_t)),
        "rocco_hip_delta_build_map_f64")
    return emap_t


def delta_build_map_lean_device(scores_t, gamma: float, lambda_ref: float, margin: float):
    """The same codes from the lean kernels (rocco_hip_delta_build_map_lean_f64): scalar switch cost, n >= 2."""
    import torch

    n = int(scores_t.shape[0])
    emap_t = torch.empty((n + 31) // 32, dtype=torch.uint8, device=scores_t.device)
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_delta_build_map_lean_f64(
        solver.handle, scores_t.data_ptr(), float(gamma), n, float(lambda_ref), float(margin), emap_t.data_ptr(),
        _dp._stream_ptr(scores_t)), "rocco_hip_delta_build_map_lean_f64")
    return emap_t


def delta_probe_device(scores_t, switch_costs, lambdas: Sequence[float], emap_t=None) -> List[Dict[str, int]]:
    n = int(scores_t.shape[0])
    costs_t, gamma = _dp._costs_arg(switch_costs, n, scores_t.device)
    lam = (ctypes.c_double * len(lambdas))(*[float(x) for x in lambdas])
    stats = (_native.ProbeStats * max(1, len(lambdas)))()
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_delta_probe_f64(
        solver.handle, scores_t.data_ptr(), costs_t.data_ptr() if costs_t is not None else None, gamma, n,
        emap_t.data_ptr() if emap_t is not None else None, lam, len(lambdas), stats,
        _dp._stream_ptr(scores_t)), "rocco_hip_delta_probe_f64")
    return [{"count": int(s.count), "uncertain": int(s.uncertain), "effect": int(s.effect),
             "max_run": int(s.max_run)} for s in stats[:len(lambdas)]]


def delta_model_lean_device(scores_t, gamma: float, lambdas: Sequence[float], emap_t):
    """Rounding-model counts from the lean kernel (rocco_hip_delta_model_lean_f64): per penalty (count, open) where
    open != 0 means the count is not certified equal to the reference's."""
    n = int(scores_t.shape[0])
    lam = (ctypes.c_double * max(1, len(lambdas)))(*[float(x) for x in lambdas])
    counts = (ctypes.c_longlong * max(1, len(lambdas)))()
    opens = (ctypes.c_longlong * max(1, len(lambdas)))()
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_delta_model_lean_f64(
        solver.handle, scores_t.data_ptr(), float(gamma), n, emap_t.data_ptr(), lam, len(lambdas), counts, opens,
        _dp._stream_ptr(scores_t)), "rocco_hip_delta_model_lean_f64")
    return [(int(counts[i]), int(opens[i])) for i in range(len(lambdas))]


def delta_bound_rounds_device(scores_t, gamma: float, rounds: Sequence[Sequence[float]]):
    """Exact-arithmetic counts (rocco_hip_delta_bound_rounds_f64) for rounds of penalties; returns per round
    (penalties as evaluated, counts, loci of the array the round ran on)."""
    n = int(scores_t.shape[0])
    flat = [float(x) for r in rounds for x in r]
    sizes = (ctypes.c_int * len(rounds))(*[len(r) for r in rounds])
    lam = (ctypes.c_double * max(1, len(flat)))(*flat)
    used = (ctypes.c_double * max(1, len(flat)))()
    counts = (ctypes.c_longlong * max(1, len(flat)))()
    lens = (ctypes.c_longlong * max(1, len(rounds)))()
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_delta_bound_rounds_f64(
        solver.handle, scores_t.data_ptr(), float(gamma), n, lam, sizes, len(rounds), used, counts, lens,
        _dp._stream_ptr(scores_t)), "rocco_hip_delta_bound_rounds_f64")
    out, at = [], 0
    for k, r in enumerate(rounds):
        out.append((list(used[at:at + len(r)]), [int(c) for c in counts[at:at + len(r)]], int(lens[k])))
        at += len(r)
    return out


def delta_window_device(scores_t, switch_costs, lambda_lo: float, lambda_hi: float, emap_t=None):
    import torch

    n = int(scores_t.shape[0])
    costs_t, gamma = _dp._costs_arg(switch_costs, n, scores_t.device)
    sol_t = torch.empty(n, dtype=torch.uint8, device=scores_t.device)
    st = _native.WindowStats()
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_delta_window_f64(
        solver.handle, scores_t.data_ptr(), costs_t.data_ptr() if costs_t is not None else None, gamma, n,
        emap_t.data_ptr() if emap_t is not None else None, float(lambda_lo), float(lambda_hi),
        sol_t.data_ptr(), ctypes.byref(st), _dp._stream_ptr(scores_t)),
        "rocco_hip_delta_window_f64")
    listed = min(int(st.n_diff), 16)
    return sol_t, {
        "count_lo": int(st.count_lo), "count_hi": int(st.count_hi), "n_diff": int(st.n_diff),
        "diff_adjacent": bool(st.diff_adjacent), "overflow": bool(st.overflow), "max_run": int(st.max_run),
        "diffs": [{"locus": int(st.diff_locus[k]), "margin_lo": float(st.diff_margin_lo[k]),
                   "margin_hi": float(st.diff_margin_hi[k]), "run": int(st.diff_run[k]),
                   "cls_lo": int(st.diff_cls_lo[k]), "cls_hi": int(st.diff_cls_hi[k])} for k in range(listed)],
    }


def delta_spine_device(scores_t, switch_costs, lambdas: Sequence[float], emap_t, solution_index: int = -1):
    """Exact counts (and optionally one exact solution) through the spine."""
    import torch

    n = int(scores_t.shape[0])
    costs_t, gamma = _dp._costs_arg(switch_costs, n, scores_t.device)
    lam = (ctypes.c_double * len(lambdas))(*[float(x) for x in lambdas])
    counts = (ctypes.c_longlong * len(lambdas))()
    sol_t = torch.empty(n, dtype=torch.uint8, device=scores_t.device) if solution_index >= 0 else None
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_delta_spine_f64(
        solver.handle, scores_t.data_ptr(), costs_t.data_ptr() if costs_t is not None else None, gamma, n,
        emap_t.data_ptr(), lam, len(lambdas), int(solution_index),
        sol_t.data_ptr() if sol_t is not None else None, counts, _dp._stream_ptr(scores_t)),
        "rocco_hip_delta_spine_f64")
    return [int(c) for c in counts], sol_t
