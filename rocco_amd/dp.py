"""Per-chromosome chain solve on MI355X -- drop-in for the reference's `rocco/dp.py`.

Same public names, argument meaning, return types and error behaviour as the reference module:

    objective_value              rocco/dp.py:16-34
    build_switch_costs           rocco/dp.py:37-46
    solve_penalized_chain        rocco/dp.py:49-86   (-> rocco/_chain_dp.c:9-213)
    calibrate_selection_penalty  rocco/dp.py:89-164
    solve_chrom_exact            rocco/dp.py:167-228

NumPy inputs are moved to the GPU, solved by librocco_hip.so (hand-written HIP, see
rocco_amd/csrc/) and returned as NumPy arrays exactly like the reference; torch CUDA tensors are
used in place (the `*_device` functions keep results on the device for the pipeline in
rocco_amd/rocco.py).  PyTorch is only the allocator / stream provider here.
"""
from __future__ import annotations

import ctypes
import functools
import logging
from typing import Dict, Optional, Tuple, Union

import numpy as np

from . import _native

logger = logging.getLogger(__name__)

ArrayLike = Union[np.ndarray, "torch.Tensor", list, tuple]  # noqa: F821


def _torch():
    import torch

    return torch


def _device_index(device=None) -> int:
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError(
            "Make sure native HIP extension is built and available (no HIP device visible; "
            "rocco_amd has no CPU path)")
    if device is None:
        return torch.cuda.current_device()
    if isinstance(device, int):
        return device
    dev = torch.device(device)
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class ResidentArray(np.ndarray):
    """A read-only host array that remembers the CUDA tensor it was copied from.

    The reference's chromosome cache holds NumPy arrays (rocco/rocco.py:1087-1098) and everything downstream indexes,
    prints and compares them as such; the solve wants the same numbers in HBM.  An entry made by `host_with_device`
    serves both: NumPy sees an ordinary float64 array, `_to_device_f64` / the budget estimators take the tensor and
    skip the upload.  Any array derived from it (a slice, a copy, arithmetic) is a plain ndarray again."""

    device_tensor = None

    def __array_finalize__(self, obj):
        self.device_tensor = None  # only the array made by host_with_device itself keeps its twin


def host_with_device(t) -> "ResidentArray":
    """Host copy of a CUDA tensor as a `ResidentArray` (read-only: the tensor is the other half of the pair)."""
    host = t.detach().cpu().numpy().view(ResidentArray)
    host.device_tensor = t
    host.flags.writeable = False
    return host


def _resident_tensor(x):
    """The CUDA twin of a `ResidentArray`, else None."""
    return getattr(x, "device_tensor", None) if isinstance(x, ResidentArray) else None


def _to_device_f64(x, device=None, name="scores"):
    """1-D float64 contiguous CUDA tensor from NumPy / list / tensor input."""
    torch = _torch()
    if _resident_tensor(x) is not None:
        x = _resident_tensor(x)
    if _is_tensor(x):
        t = x
        if t.ndim != 1:
            raise ValueError(f"`{name}` must be one-dimensional")
        if not t.is_cuda:
            t = t.to(f"cuda:{_device_index(device)}")
        return t.to(torch.float64).contiguous()
    arr = np.ascontiguousarray(x, dtype=np.float64)
    if arr.ndim != 1:
        raise ValueError(f"`{name}` must be one-dimensional")
    return torch.from_numpy(arr).to(f"cuda:{_device_index(device)}")


def _stream_ptr(t) -> int:
    torch = _torch()
    return int(torch.cuda.current_stream(t.device).cuda_stream)


# --------------------------------------------------------------------------------------------
# NumPy's pairwise summation of a constant vector, without materialising the vector
# --------------------------------------------------------------------------------------------

_NUMPY_BUFSIZE = 8192  # NumPy's default ufunc buffer size, in elements


def _seq_sum(value: float, k: int) -> float:
    acc = value
    for _ in range(k - 1):
        acc = acc + value
    return acc


def _pairwise_constant(value: float, m: int) -> float:
    """What NumPy's pairwise_sum (8 accumulators, blocks of 128, halving split rounded down to a
    multiple of 8) returns for m copies of `value`."""

    @functools.lru_cache(maxsize=None)
    def rec(k: int) -> float:
        if k < 8:
            acc = 0.0
            for _ in range(k):
                acc = acc + value
            return acc
        if k <= 128:
            r = _seq_sum(value, k // 8)
            res = ((r + r) + (r + r)) + ((r + r) + (r + r))
            for _ in range(k % 8):
                res = res + value
            return res
        k2 = k // 2
        k2 -= k2 % 8
        return rec(k2) + rec(k - k2)

    return rec(int(m))


@functools.lru_cache(maxsize=4096)
def sum_constant_like_numpy(value: float, length: int) -> float:
    """np.sum(np.full(length, value)) bit-for-bit (rocco/dp.py:110-111 uses np.sum(switch_costs)
    on the constant vector of rocco/dp.py:37-46).  np.add.reduce walks the array in chunks of the
    ufunc buffer size (np.getbufsize() == 8192 elements by default), adding the pairwise sum of
    each chunk to a running total."""
    value = float(value)
    if length <= 0:
        return 0.0
    full, rest = divmod(int(length), _NUMPY_BUFSIZE)
    acc = 0.0
    if full:
        chunk = _pairwise_constant(value, _NUMPY_BUFSIZE)
        for _ in range(full):
            acc = acc + chunk
    if rest:
        acc = acc + _pairwise_constant(value, rest)
    return acc


# --------------------------------------------------------------------------------------------
# reference surface
# --------------------------------------------------------------------------------------------

def build_switch_costs(scores, gamma: float = 1.0) -> np.ndarray:
    """rocco/dp.py:37-46 (host array, provided for API compatibility; the solver itself takes the
    scalar gamma and never materialises this vector)."""
    if _is_tensor(scores):
        shape = tuple(scores.shape)
        ndim = scores.ndim
    else:
        scores_ = np.asarray(scores, dtype=np.float64)
        shape, ndim = scores_.shape, scores_.ndim
    if ndim != 1:
        raise ValueError("`scores` must be a one-dimensional array")
    if shape[0] <= 1:
        return np.zeros(0, dtype=np.float64)
    return np.full(shape[0] - 1, float(gamma), dtype=np.float64)


def _costs_arg(switch_costs, n: int, device):
    """(costs_tensor_or_None, gamma) from a scalar, NumPy vector or tensor."""
    if switch_costs is None:
        raise ValueError("`switch_costs` is required")
    if np.isscalar(switch_costs):
        return None, float(switch_costs)
    t = _to_device_f64(switch_costs, device, name="switch_costs")
    if n > 1 and t.shape[0] != n - 1:
        raise ValueError("`switch_costs` must have length len(scores) - 1")
    if n <= 1:
        return None, 0.0
    return t, 0.0


def objective_value(solution, scores, switch_costs) -> float:
    """rocco/dp.py:16-34, evaluated on the device."""
    torch = _torch()
    scores_t = _to_device_f64(scores)
    n = int(scores_t.shape[0])
    if _is_tensor(solution):
        sol_t = solution.to(scores_t.device)
    else:
        sol_t = torch.from_numpy(np.ascontiguousarray(np.asarray(solution) > 0.5, dtype=np.uint8)).to(scores_t.device)
    if sol_t.dtype != torch.uint8:
        sol_t = (sol_t > 0.5).to(torch.uint8)
    sol_t = sol_t.contiguous()
    if sol_t.shape[0] != n:
        raise ValueError("`solution` and `scores` must have the same length")
    costs_t, gamma = _costs_arg(switch_costs, n, scores_t.device)
    solver = _native.solver_for(scores_t.device.index)
    out = ctypes.c_double(0.0)
    _native.check(_native.load().rocco_hip_objective_value_f64(
        solver.handle, sol_t.data_ptr(), scores_t.data_ptr(),
        costs_t.data_ptr() if costs_t is not None else None, gamma, n, ctypes.byref(out),
        _stream_ptr(scores_t)), "rocco_hip_objective_value_f64")
    return float(out.value)


def solve_penalized_chain_device(scores_t, switch_costs, selection_penalty: float, want_solution=True):
    """Device-resident form of solve_penalized_chain: returns (uint8 CUDA tensor, value, count, path)."""
    torch = _torch()
    n = int(scores_t.shape[0])
    if n <= 0:
        raise ValueError("`scores` cannot be empty")
    costs_t, gamma = _costs_arg(switch_costs, n, scores_t.device)
    sol_t = torch.empty(n, dtype=torch.uint8, device=scores_t.device) if want_solution else None
    value = ctypes.c_double(0.0)
    count = ctypes.c_longlong(0)
    path = ctypes.c_int(0)
    solver = _native.solver_for(scores_t.device.index)
    _native.check(_native.load().rocco_hip_solve_penalized_chain_f64(
        solver.handle, scores_t.data_ptr(), costs_t.data_ptr() if costs_t is not None else None,
        gamma, n, float(selection_penalty), sol_t.data_ptr() if sol_t is not None else None,
        ctypes.byref(value), ctypes.byref(count), ctypes.byref(path), _stream_ptr(scores_t)),
        "rocco_hip_solve_penalized_chain_f64")
    return sol_t, float(value.value), int(count.value), int(path.value)


def solve_penalized_chain(scores, switch_costs, selection_penalty: float) -> Tuple[np.ndarray, float, int]:
    r"""Solve the penalized binary chain problem for one chromosome (rocco/dp.py:49-86).

    .. math::
       \max_{z \in \{0,1\}^n} \sum_j (s_j - \lambda) z_j - \sum_j c_j |z_{j+1} - z_j|

    Ties go to the labeling with fewer selected loci, exactly as rocco/_chain_dp.c:133-179.
    Returns ``(uint8[n], penalized_objective, selected_count)``.
    """
    _native.load()  # RuntimeError if the extension is missing (rocco/dp.py:73-74)
    if not _is_tensor(scores):
        probe = np.asarray(scores)
        if probe.ndim != 1:
            raise ValueError("`scores` must be one-dimensional")
        if probe.shape[0] <= 0:
            raise ValueError("`scores` cannot be empty")
    if not np.isscalar(switch_costs) and not _is_tensor(switch_costs):
        if np.asarray(switch_costs).ndim != 1:
            raise ValueError("`switch_costs` must be one-dimensional")
    scores_t = _to_device_f64(scores)
    sol_t, value, count, _ = solve_penalized_chain_device(scores_t, switch_costs, selection_penalty)
    return sol_t.cpu().numpy().astype(np.uint8, copy=False), float(value), int(count)


def _sum_costs(n: int, costs_t, gamma: float) -> float:
    """np.sum(switch_costs) of rocco/dp.py:110-111, bit-for-bit."""
    if costs_t is None:
        return sum_constant_like_numpy(gamma, n - 1)
    # general vector: NumPy's own pairwise summation on the host copy (API-completeness path)
    return float(np.sum(costs_t.cpu().numpy()))


def calibrate_selection_penalty_device(scores_t, switch_costs, target_count: int, max_iter: int = 60):
    """Device-resident calibrate: returns (penalty, uint8 CUDA tensor, value, count, info dict)."""
    return calibrate_batch_device([scores_t], [switch_costs], [target_count], max_iter=max_iter)[0]


def calibrate_batch_device(scores_list, switch_costs_list, target_counts, max_iter: int = 60, score_stats=None):
    """One device launch sequence for several chromosomes (rocco_hip_solve_budget_batch_f64).

    `score_stats`: optional [k, 3] float64 HOST array (NumPy, or a CPU tensor whose copy from the device has
    completed) of np.min, np.max and sum |.| of every score array as score_central_tendency_chrom_batch_device
    (with_stats=True) produced them: the solve then skips its own pass over the scores."""
    torch = _torch()
    lib = _native.load()
    k = len(scores_list)
    if k == 0:
        return []
    device = scores_list[0].device
    tasks = (_native.BudgetTask * k)()
    results = (_native.BudgetResult * k)()
    keep = []
    sols = []
    for i, (s_t, costs, target) in enumerate(zip(scores_list, switch_costs_list, target_counts)):
        n = int(s_t.shape[0])
        if n == 0:
            raise ValueError("`scores` cannot be empty")
        costs_t, gamma = _costs_arg(costs, n, device)
        sol_t = torch.empty(n, dtype=torch.uint8, device=device)
        keep.append((s_t, costs_t))
        sols.append(sol_t)
        tasks[i].scores_dev = s_t.data_ptr()
        tasks[i].switch_costs_dev = costs_t.data_ptr() if costs_t is not None else None
        tasks[i].gamma = gamma
        tasks[i].n = n
        tasks[i].target_count = int(target)
        tasks[i].sum_costs = _sum_costs(n, costs_t, gamma)
        tasks[i].max_iter = int(max_iter)
        tasks[i].solution_dev = sol_t.data_ptr()
    solver = _native.solver_for(device.index)
    if score_stats is not None:
        stats_h = np.ascontiguousarray(score_stats.numpy() if _is_tensor(score_stats) else score_stats, dtype=np.float64)
        if stats_h.shape != (k, 3):
            raise ValueError("`score_stats` must have shape [len(scores_list), 3]")
        _native.check(lib.rocco_hip_solve_budget_batch_stats_f64(solver.handle, k, tasks, stats_h.ctypes.data, results,
                                                                 _stream_ptr(scores_list[0])),
                      "rocco_hip_solve_budget_batch_stats_f64")
    else:
        _native.check(lib.rocco_hip_solve_budget_batch_f64(solver.handle, k, tasks, results,
                                                           _stream_ptr(scores_list[0])),
                      "rocco_hip_solve_budget_batch_f64")
    out = []
    for i in range(k):
        r = results[i]
        out.append((float(r.selection_penalty), sols[i], float(r.penalized_value), int(r.selected_count),
                    {"evaluations": int(r.evaluations), "path": int(r.path), "passes": int(r.passes),
                     "zone_iters": int(r.zone_iters), "n_diff": int(r.n_diff), "maps": int(r.maps)}))
    return out


def calibrate_selection_penalty(scores, switch_costs, target_count: int,
                                max_iter: int = 60) -> Tuple[float, np.ndarray, float, int]:
    r"""Find a selection penalty whose solution selects at most `target_count` loci, by the
    reference's bracket + 60-step bisection (rocco/dp.py:89-164).  Returns
    ``(upper, best_solution, best_value, best_count)``."""
    _native.load()
    if not _is_tensor(scores) and np.asarray(scores).shape[0] == 0:
        raise ValueError("`scores` cannot be empty")
    scores_t = _to_device_f64(scores)
    penalty, sol_t, value, count, _ = calibrate_selection_penalty_device(
        scores_t, switch_costs, target_count, max_iter=max_iter)
    return float(penalty), sol_t.cpu().numpy().astype(np.uint8, copy=False), float(value), int(count)


def solve_chrom_exact_device(scores_t, budget: Optional[float] = None, gamma: float = 1.0,
                             selection_penalty: Optional[float] = None):
    """Device-resident solve_chrom_exact: returns (uint8 CUDA tensor, objective, details)."""
    n = int(scores_t.shape[0])
    info: Dict[str, float] = {}
    if selection_penalty is None:
        if budget is None:
            penalty_ = 0.0
            sol_t, penalized, count, path = solve_penalized_chain_device(scores_t, float(gamma), penalty_)
            info = {"path": path}
        else:
            target_count = int(np.floor(n * float(budget)))  # rocco/dp.py:197
            penalty_, sol_t, penalized, count, info = calibrate_selection_penalty_device(
                scores_t, float(gamma), target_count)
    else:
        penalty_ = float(selection_penalty)
        sol_t, penalized, count, path = solve_penalized_chain_device(scores_t, float(gamma), penalty_)
        info = {"path": path}
    objective = objective_value(sol_t, scores_t, float(gamma))
    details = {
        "penalized_objective": float(penalized),
        "selected_count": int(count),
        "selected_fraction": float(count / n),
        "selection_penalty": float(penalty_),
    }
    details.update({f"_{k}": v for k, v in info.items()})
    return sol_t, float(objective), details


def solve_chrom_exact(scores, budget: Optional[float] = None, gamma: float = 1.0,
                      selection_penalty: Optional[float] = None, return_details: bool = False):
    r"""Solve one chromosome with the exact penalized-chain solver (rocco/dp.py:167-228).

    If ``selection_penalty`` is not supplied and ``budget`` is, a penalty is calibrated so that at
    most ``floor(n * budget)`` loci are selected; if ``selection_penalty`` is supplied the chain is
    solved directly with it; with neither, the penalty is 0.
    Returns ``(uint8[n], objective)`` or ``(uint8[n], objective, details)``.
    """
    _native.load()
    scores_t = _to_device_f64(scores)
    if scores_t.shape[0] == 0:
        raise ValueError("`scores` cannot be empty")
    sol_t, objective, details = solve_chrom_exact_device(scores_t, budget=budget, gamma=gamma,
                                                         selection_penalty=selection_penalty)
    solution = sol_t.cpu().numpy().astype(np.uint8, copy=False)
    if not return_details:
        return solution, objective
    public = {k: v for k, v in details.items() if not k.startswith("_")}
    return solution, objective, public
