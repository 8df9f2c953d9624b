"""ctypes binding of librocco_hip.so (C ABI declared in include/rocco_hip.h).

This is the reference-side stub a ROCCO maintainer would add in place of `from . import _chain_dp`
(rocco/dp.py:10-13): load the shared library, declare the prototypes, and raise the same
RuntimeError the reference raises when its native extension is missing (rocco/dp.py:73-74).
There is no CPU fallback: if the library or a GPU is missing, every call fails loudly.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROCCO_HIP_LIBRARY") or os.path.join(_HERE, "librocco_hip.so")  # override: A/B builds

OK, ENOMEM, EINVAL, EHIP = 0, -1, -2, -3
PATH_CERTIFIED, PATH_EXACT, PATH_TRIVIAL, PATH_SPINE = 1, 2, 3, 4

c_double_p = ctypes.POINTER(ctypes.c_double)
c_ll_p = ctypes.POINTER(ctypes.c_longlong)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_size_p = ctypes.POINTER(ctypes.c_size_t)


class BudgetTask(ctypes.Structure):
    _fields_ = [
        ("scores_dev", ctypes.c_void_p),
        ("switch_costs_dev", ctypes.c_void_p),
        ("gamma", ctypes.c_double),
        ("n", ctypes.c_size_t),
        ("target_count", ctypes.c_longlong),
        ("sum_costs", ctypes.c_double),
        ("max_iter", ctypes.c_int),
        ("solution_dev", ctypes.c_void_p),
    ]


class BudgetResult(ctypes.Structure):
    _fields_ = [
        ("selection_penalty", ctypes.c_double),
        ("penalized_value", ctypes.c_double),
        ("selected_count", ctypes.c_longlong),
        ("evaluations", ctypes.c_int),
        ("path", ctypes.c_int),
        ("passes", ctypes.c_int),
        ("zone_iters", ctypes.c_int),
        ("n_diff", ctypes.c_longlong),
        ("maps", ctypes.c_int),
    ]


class ProbeStats(ctypes.Structure):
    _fields_ = [
        ("count", ctypes.c_longlong),
        ("uncertain", ctypes.c_longlong),
        ("effect", ctypes.c_longlong),
        ("max_run", ctypes.c_longlong),
    ]


class WindowStats(ctypes.Structure):
    _fields_ = [
        ("count_lo", ctypes.c_longlong),
        ("count_hi", ctypes.c_longlong),
        ("n_diff", ctypes.c_longlong),
        ("diff_adjacent", ctypes.c_int),
        ("overflow", ctypes.c_int),
        ("max_run", ctypes.c_longlong),
        ("diff_locus", ctypes.c_longlong * 16),
        ("diff_margin_lo", ctypes.c_double * 16),
        ("diff_margin_hi", ctypes.c_double * 16),
        ("diff_run", ctypes.c_longlong * 16),
        ("diff_cls_lo", ctypes.c_int * 16),
        ("diff_cls_hi", ctypes.c_int * 16),
    ]


# every symbol include/rocco_hip.h declares: (name, restype, argtypes)
PROTOTYPES = [
    ("rocco_hip_abi_version", ctypes.c_int, []),
    ("rocco_hip_last_error", ctypes.c_char_p, []),
    ("rocco_hip_solver_create", ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    ("rocco_hip_solver_destroy", None, [ctypes.c_void_p]),
    ("rocco_hip_solver_set", ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_longlong]),
    ("rocco_hip_score_trimmed_mean", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
      ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_power_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_score_median_batch", ctypes.c_int,
     [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_size_t, c_size_p, c_size_p,
      ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_score_median_batch_stats", ctypes.c_int,
     [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_size_t, c_size_p, c_size_p,
      ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_score_median", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
      ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_score_order_statistic", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
      ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_score_mean", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
      ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_solve_penalized_chain_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t,
      ctypes.c_double, ctypes.c_void_p, c_double_p, c_ll_p, c_int_p, ctypes.c_void_p]),
    ("rocco_hip_solve_budget_batch_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(BudgetTask), ctypes.POINTER(BudgetResult),
      ctypes.c_void_p]),
    ("rocco_hip_solve_budget_batch_stats_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(BudgetTask), ctypes.c_void_p, ctypes.POINTER(BudgetResult),
      ctypes.c_void_p]),
    ("rocco_hip_delta_model_lean_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p, c_double_p, ctypes.c_size_t,
      c_ll_p, c_ll_p, ctypes.c_void_p]),
    ("rocco_hip_delta_probe_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p,
      c_double_p, ctypes.c_size_t, ctypes.POINTER(ProbeStats), ctypes.c_void_p]),
    ("rocco_hip_delta_build_map_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t,
      ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_delta_build_map_lean_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
      ctypes.c_void_p]),
    ("rocco_hip_delta_bound_rounds_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t, c_double_p, c_int_p, ctypes.c_int,
      c_double_p, c_ll_p, c_ll_p, ctypes.c_void_p]),
    ("rocco_hip_delta_spine_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p,
      c_double_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, c_ll_p, ctypes.c_void_p]),
    ("rocco_hip_delta_window_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p,
      ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.POINTER(WindowStats), ctypes.c_void_p]),
    ("rocco_hip_objective_value_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
      ctypes.c_size_t, c_double_p, ctypes.c_void_p]),
    ("rocco_hip_sort_f64", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_sorted_probe_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_ll_p, ctypes.c_size_t, c_double_p, ctypes.c_double, c_double_p,
      ctypes.c_size_t, c_ll_p, c_ll_p, ctypes.c_void_p]),
    ("rocco_hip_autocovariance_sums_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int, c_double_p, ctypes.c_void_p]),
    ("rocco_hip_negative_part_f64", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_soft_counts_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_peak_signal_stat_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_double,
      ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_ecdf_survival_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
      ctypes.c_void_p]),
    ("rocco_hip_bh_adjust_f64", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_log_scale_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_decode_runs_table", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), c_size_p, c_ll_p, ctypes.c_void_p, ctypes.c_size_t,
      ctypes.c_size_t, c_size_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    ("rocco_hip_decode_runs_batch", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), c_size_p, ctypes.POINTER(ctypes.c_void_p),
      ctypes.POINTER(ctypes.c_void_p), c_size_p, c_size_p, ctypes.c_void_p]),
    ("rocco_hip_decode_runs", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
      ctypes.c_size_t, c_size_p, ctypes.c_void_p]),
    ("rocco_hip_crossfit_whittaker_baseline_batch_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), c_size_p, c_size_p, ctypes.c_double,
      ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    ("rocco_hip_crossfit_whittaker_residual_batch_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_size_p, c_size_p,
      ctypes.c_double, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    ("rocco_hip_crossfit_whittaker_baseline_matrix_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_void_p,
      ctypes.c_void_p]),
    ("rocco_hip_score_centered_wls_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_double,
      ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_double_p, c_int_p, ctypes.c_void_p]),
    ("rocco_hip_wls_rolling_variances_batch_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), c_size_p, c_size_p, ctypes.c_int,
      ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    ("rocco_hip_score_centered_wls_given_variances_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_double,
      ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_double_p, c_int_p, ctypes.c_void_p]),
    ("rocco_hip_log_scale_center_rows_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_int,
      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_log_scale_row_offsets_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_int,
      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_wls_sorted_rows", ctypes.c_int, [ctypes.c_void_p]),
    ("rocco_hip_log2_selfcheck", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p]),
    ("rocco_hip_subtract_finite_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_subtract_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_narrowpeak_summit_offsets", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_union_intervals", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, c_size_p, c_int_p, ctypes.c_void_p]),
    ("rocco_hip_scatter_tracks", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, c_size_p, ctypes.c_size_t,
      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    ("rocco_hip_numpy_sum_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_double_p, ctypes.c_void_p]),
    ("rocco_hip_budget_null_draw_stats_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_double,
      c_double_p, ctypes.c_void_p]),
    ("rocco_hip_count_path_reserve", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.c_double,
      ctypes.c_void_p]),
    ("rocco_hip_count_path_reserve_ex", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.c_double,
      ctypes.c_int, ctypes.c_void_p]),
    ("rocco_hip_whittaker_batch_scratch_bytes", ctypes.c_size_t, [ctypes.c_size_t, c_size_p, c_size_p]),
    ("rocco_hip_crossfit_whittaker_residual_batch_scratch_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_size_p, c_size_p,
      ctypes.c_double, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_buffer_growths", ctypes.c_longlong, []),
    ("rocco_hip_whittaker_seam_repairs", ctypes.c_longlong, []),
    ("rocco_hip_model_chain_counters", None, [ctypes.POINTER(ctypes.c_longlong)]),
    ("rocco_hip_model_chain_written_counters", None, [ctypes.POINTER(ctypes.c_longlong)]),
    ("rocco_hip_solver_device_bytes", ctypes.c_longlong, [ctypes.c_void_p]),
    ("rocco_hip_pcg64_standard_normal_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_size_t,
      ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_void_p]),
    ("rocco_hip_bartlett_multipliers_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, c_double_p, ctypes.c_size_t, ctypes.c_void_p,
      ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    ("rocco_hip_multiply_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    ("rocco_hip_subtract_positive_row_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
      ctypes.c_void_p]),
    ("rocco_hip_bigwig_dense_fill_f64", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int,
      ctypes.c_void_p, ctypes.c_size_t, c_ll_p, c_ll_p, c_size_p, c_int_p, ctypes.c_void_p]),
    ("rocco_hip_synth_matrix", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
      ctypes.c_size_t, ctypes.c_uint64, ctypes.c_void_p]),
]

class _Library:
    """The loaded library with one policy in front of every status-returning entry point: a device allocation of the
    library's own (hipMalloc for a solver's scratch) can fail while PyTorch's caching allocator sits on blocks it does not
    use -- the two do not see each other's reserves.  On ROCCO_HIP_ENOMEM the cached blocks are handed back to the runtime
    and the call is made once more (every entry point is a function of its arguments: a failed reserve has launched
    nothing).  Round 5: the composed driver at K = 100 on the whole genome ran out of memory in its second run of a
    process this way."""

    def __init__(self, cdll: ctypes.CDLL):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "_wrapped", {})

    def __getattr__(self, name):
        wrapped = self._wrapped.get(name)
        if wrapped is not None:
            return wrapped
        fn = getattr(self._cdll, name)
        if getattr(fn, "restype", None) is not ctypes.c_int:
            self._wrapped[name] = fn
            return fn

        def call(*args, _fn=fn):
            rc = _fn(*args)
            if rc == ENOMEM:
                import torch

                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    rc = _fn(*args)
            return rc

        call.__name__ = name
        self._wrapped[name] = call
        return call


_lib: Optional[_Library] = None


def load() -> _Library:
    """Load librocco_hip.so and bind every prototype.  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own HIP runtime: load it first, so that this library binds to the same one
    # (two HIP runtimes in one process do not see each other's devices, streams or allocations)
    import torch  # noqa: F401

    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "Make sure native HIP extension is built and available "
            f"({LIB_PATH} missing; run `python -c 'import __graft_entry__ as g; g.build()'`)")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # missing ROCm runtime etc.
        raise RuntimeError(f"Make sure native HIP extension is built and available: {exc}") from exc
    for name, restype, argtypes in PROTOTYPES:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = _Library(lib)
    return _lib


def last_error() -> str:
    msg = load().rocco_hip_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc == OK:
        return
    if rc == ENOMEM:
        raise MemoryError(f"{what}: device/host allocation failed ({last_error()})")
    if rc == EINVAL:
        raise ValueError(f"{what}: invalid argument ({last_error()})")
    raise RuntimeError(f"{what}: HIP runtime error ({last_error()})")


class Solver:
    """Owns one `rocco_hip_solver*` bound to a HIP device."""

    def __init__(self, device: int = 0):
        self._lib = load()
        handle = ctypes.c_void_p()
        check(self._lib.rocco_hip_solver_create(ctypes.byref(handle), int(device)),
              "rocco_hip_solver_create")
        self.handle = handle
        self.device = int(device)

    def set(self, key: str, value: int) -> None:
        check(self._lib.rocco_hip_solver_set(self.handle, key.encode(), int(value)),
              f"rocco_hip_solver_set({key})")

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle:
            self._lib.rocco_hip_solver_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_solvers: dict = {}
_tls = threading.local()


@contextlib.contextmanager
def use_solver(solver: Solver):
    """Calls made by this thread inside the block use `solver` (its scratch buffers) instead of the process-wide
    handle of its device: concurrent host threads need one solver each (rocco_amd.pipeline)."""
    previous = getattr(_tls, "solver", None)
    _tls.solver = solver
    try:
        yield solver
    finally:
        _tls.solver = previous


def max_side_streams() -> int:
    """How many HIP streams besides the caller's may carry work at the same time.

    The HIP runtime multiplexes a process's streams onto a fixed number of hardware queues (4 unless GPU_MAX_HW_QUEUES
    says otherwise).  Measured on MI355X / ROCm 7.2 (round 3, the count path of a genome at K = 100): with the caller's
    stream plus THREE worker streams everything completes; with a fourth worker -- five streams on four queues -- two of
    the workers' streams stop for good (their host threads never return from hipStreamSynchronize; the process has to
    be killed), and the same run with GPU_MAX_HW_QUEUES=8 completes.  So the side streams are capped at the number of
    hardware queues minus the caller's."""
    if os.environ.get("ROCCO_MAX_SIDE_STREAMS"):  # (experiments: lift or lower the cap)
        try:
            return max(1, int(os.environ["ROCCO_MAX_SIDE_STREAMS"]))
        except ValueError:
            pass
    try:
        queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        queues = 4
    return max(1, queues - 1)


def solver_for(device: int) -> Solver:
    """Process-wide solver handle per device (scratch buffers are reused across calls), or the calling
    thread's own (`use_solver`)."""
    own = getattr(_tls, "solver", None)
    if own is not None and own.device == int(device):
        return own
    s = _solvers.get(int(device))
    if s is None:
        s = Solver(int(device))
        _solvers[int(device)] = s
    return s
