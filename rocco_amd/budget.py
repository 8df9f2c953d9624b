"""Budget and switch-cost estimation around the solve, for score tracks held in HBM.

What stands where in the reference:

    estimate_budget_nonnull_fraction_from_score_track   rocco/inference.py:1312-1421 (+ the direct-score null, 1151-1309)
    _estimate_effective_sample_size                     rocco/inference.py:446-501
    estimate_empirical_bayes_budgets                    rocco/inference.py:1593-1737 (+ 1488-1590)
    _resolve_chrom_gamma                                rocco/rocco.py:751-789
    _resolve_budgets                                    rocco/rocco.py:1113-1143
    the bigWig branch of _build_chrom_cache             rocco/rocco.py:977-1008, 1049-1097

Split: every n-long step runs on the device -- the order statistics (np.median of the residual template, the MAD of
its mirrored non-positive part, the median of the positive scores) are read off ONE radix sort of the vector, the
means are summed in NumPy's own order (npsum.hip: equal to np.mean bit for bit), the per-draw product and its four
statistics are the kernels of the count-path null, the autocovariances are summed in a fixed order
(budget_stats.hip).  The dependent multipliers of the bootstrap draws come from NumPy's generator and SciPy's
fftconvolve on the host, draw by draw as the reference makes them (their streams cannot be reproduced elsewhere;
SURVEY.md section 8 (f)), and the scalar logic (Welford updates, the stopping rule, Geyer's truncation, the beta-binomial
fit through SciPy) is host code.  Given the same multipliers every statistic equals the reference's bit for bit except
the autocorrelation time, which the reference takes from an FFT (agreement ~1e-13; tests/golden/make_golden_budget.py).
"""
from __future__ import annotations

import ctypes
import logging
import threading
from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import _native
from . import dp as _dp

logger = logging.getLogger(__name__)

_MAD_TO_SIGMA = 1.4826  # rocco/inference.py:37


# --------------------------------------------------------------------------------------------------------------
# device helpers
# --------------------------------------------------------------------------------------------------------------

def _as_score_tensor(score_track):
    import torch

    if _dp._resident_tensor(score_track) is not None:
        score_track = _dp._resident_tensor(score_track)
    if _dp._is_tensor(score_track):
        t = score_track
        if not t.is_cuda:
            t = t.to(f"cuda:{_dp._device_index()}")
        return t.to(torch.float64).contiguous()
    arr = np.asarray(score_track, dtype=np.float64)
    if arr.ndim != 1:
        raise ValueError("`score_track` must be one-dimensional")
    return torch.from_numpy(np.ascontiguousarray(arr)).to(f"cuda:{_dp._device_index()}")


def _lib_solver_stream(t):
    return _native.load(), _native.solver_for(t.device.index), _dp._stream_ptr(t)


def sort_device(x_t):
    """Ascending sorted copy of a one-dimensional float64 CUDA tensor (rocco_hip_sort_f64)."""
    import torch

    out = torch.empty_like(x_t)
    lib, solver, stream = _lib_solver_stream(x_t)
    _native.check(lib.rocco_hip_sort_f64(solver.handle, x_t.data_ptr(), int(x_t.shape[0]), out.data_ptr(), stream),
                  "rocco_hip_sort_f64")
    return out


def sorted_probe(sorted_t, ranks=(), thresholds=(), shift: float = 0.0):
    """Values at `ranks` of a sorted tensor; for each threshold t the counts of (x - shift) <= t and < t."""
    lib, solver, stream = _lib_solver_stream(sorted_t)
    nr, nt = len(ranks), len(thresholds)
    r = (ctypes.c_longlong * max(1, nr))(*[int(v) for v in ranks])
    vals = (ctypes.c_double * max(1, nr))()
    th = (ctypes.c_double * max(1, nt))(*[float(v) for v in thresholds])
    le = (ctypes.c_longlong * max(1, nt))()
    lt = (ctypes.c_longlong * max(1, nt))()
    _native.check(lib.rocco_hip_sorted_probe_f64(solver.handle, sorted_t.data_ptr(), int(sorted_t.shape[0]), r, nr, vals,
                                                 float(shift), th, nt, le, lt, stream), "rocco_hip_sorted_probe_f64")
    return [float(v) for v in vals[:nr]], [int(v) for v in le[:nt]], [int(v) for v in lt[:nt]]


def _median_of_sorted_range(sorted_t, first: int, count: int) -> float:
    """np.median of sorted_t[first : first + count] (count >= 1): the middle value or the mean of the two middle ones."""
    lo, hi = first + (count - 1) // 2, first + count // 2
    (a, b), _, _ = sorted_probe(sorted_t, ranks=(lo, hi))
    return a if lo == hi else (a + b) / 2.0


def _draw_stats(scores_t, center: float, soft_scale: float, threshold: float) -> Tuple[float, float, float, float]:
    """np.mean(pos), np.mean(pos / soft_scale), np.mean(pos > 0), np.mean(scores > threshold), pos = clip(scores - center, 0)."""
    lib, solver, stream = _lib_solver_stream(scores_t)
    out = (ctypes.c_double * 4)()
    _native.check(lib.rocco_hip_budget_null_draw_stats_f64(solver.handle, scores_t.data_ptr(), int(scores_t.shape[0]),
                                                           float(center), float(soft_scale), float(threshold), out, stream),
                  "rocco_hip_budget_null_draw_stats_f64")
    return float(out[0]), float(out[1]), float(out[2]), float(out[3])


def _numpy_mean(x_t) -> float:
    from .inference import numpy_sum_device

    return numpy_sum_device(x_t) / float(x_t.shape[0])


# --------------------------------------------------------------------------------------------------------------
# scalar rules of the estimate (host)
# --------------------------------------------------------------------------------------------------------------

def _resolve_budget_ess_max_lag(n_loci: int, dependence_lag_hint: Optional[int] = None) -> int:
    """Lag cap of the autocorrelation sum (rocco/inference.py:504-517): four times the dependence scale (101 loci
    unless hinted), at least 16, below the track length."""
    n = max(1, int(n_loci))
    scale = min(n, 101) if dependence_lag_hint is None else max(1, min(n, int(dependence_lag_hint)))
    return int(min(n - 1, max(16, 4 * scale)))


def _resolve_budget_bootstrap_bandwidth(n_loci: int, dependence_lag_hint: Optional[int] = None) -> int:
    """Bandwidth of the dependent multipliers (rocco/inference.py:520-531): the hint, else n^(1/3), at least 8."""
    n = max(1, int(n_loci))
    if n <= 1:
        return 1
    wanted = round(n ** (1.0 / 3.0)) if dependence_lag_hint is None else int(dependence_lag_hint)
    return int(min(n - 1, max(8, wanted)))


def _build_budget_bootstrap_kernel(bandwidth: int) -> np.ndarray:
    """Bartlett weights on -b..b, scaled to unit sum of squares (rocco/inference.py:534-543)."""
    b = max(1, int(bandwidth))
    taps = np.maximum(1.0 - np.abs(np.arange(-b, b + 1, dtype=np.float64)) / float(b + 1), 0.0)
    taps /= np.sqrt(np.sum(taps * taps))
    return taps


def _generate_dependent_wild_weights(n_loci: int, kernel: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """One draw of the multiplier process (rocco/inference.py:546-575): smoothed standard normals, centred and scaled
    to unit variance -- on the host, with the reference's own generator calls in the reference's own order (what makes a
    draw reproducible is the generator's stream, which no other implementation shares)."""
    from scipy import signal

    n = max(1, int(n_loci))
    if n == 1:
        return np.ones(1, dtype=np.float64)
    taps = np.asarray(kernel, dtype=np.float64)
    w = _smooth_and_standardise(rng.standard_normal(n + taps.size - 1), taps)
    if w is not None:
        return w
    signs = rng.choice(np.array([-1.0, 1.0]), size=n)  # degenerate smoothing: Rademacher signs instead
    signs -= float(np.mean(signs))
    return signs / max(float(np.std(signs)), 1.0e-6)


def _smooth_and_standardise(normals: np.ndarray, taps: np.ndarray):
    """The part of a draw that does not touch the generator (rocco/inference.py:561-569): FFT convolution with the taps,
    centring, scaling to unit variance; None when the smoothed series is degenerate (the caller then draws signs)."""
    from scipy import signal

    w = np.asarray(signal.fftconvolve(normals, taps, mode="valid"), dtype=np.float64)
    w -= float(np.mean(w))
    spread = float(np.std(w))
    if np.isfinite(spread) and spread > 1.0e-8:
        return w / spread
    return None


# Where a composed run's time goes (bench.py's composed-driver leg): a dict handed to `collect_timings` receives the
# seconds the estimators spend obtaining multipliers (host: waiting for the threads that make them; device: making them).
_TIMINGS = None


def collect_timings(store: Optional[dict]) -> None:
    global _TIMINGS
    _TIMINGS = store


def _note(key: str, seconds: float) -> None:
    if _TIMINGS is not None:
        _TIMINGS[key] = _TIMINGS.get(key, 0.0) + float(seconds)


def device_standard_normal(rng: np.random.Generator, count: int, out_t=None, device_index: Optional[int] = None):
    """``rng.standard_normal(count)`` made on the device (rocco_hip_pcg64_standard_normal_f64): a float64 CUDA tensor of
    NumPy's own values -- its ziggurat over PCG64, continued from `rng`'s state -- and `rng` advanced by the raw draws
    they consumed, so later host draws continue the same stream.  Bit for bit NumPy's except tail values (|x| > 3.654),
    which may differ in the last place (see include/rocco_hip.h)."""
    import ctypes

    import torch

    state = rng.bit_generator.state
    if state.get("bit_generator") != "PCG64":
        raise ValueError("device multipliers need a PCG64 generator (np.random.default_rng)")
    count = int(count)
    index = _dp._device_index() if device_index is None else int(device_index)
    if out_t is None:
        out_t = torch.empty(count, dtype=torch.float64, device=f"cuda:{index}")
    if out_t.dtype != torch.float64 or not out_t.is_contiguous() or out_t.numel() < count:
        raise ValueError("`out_t` must be a contiguous float64 CUDA tensor of at least `count` elements")
    if count == 0:
        return out_t[:0]
    s128, i128 = int(state["state"]["state"]), int(state["state"]["inc"])
    mask = (1 << 64) - 1
    solver = _native.solver_for(out_t.device.index)
    raws = ctypes.c_ulonglong(0)
    _native.check(_native.load().rocco_hip_pcg64_standard_normal_f64(
        solver.handle, s128 >> 64, s128 & mask, i128 >> 64, i128 & mask, count, out_t.data_ptr(), ctypes.byref(raws),
        _lib_solver_stream(out_t)[2]), "rocco_hip_pcg64_standard_normal_f64")
    rng.bit_generator.advance(int(raws.value))
    return out_t.reshape(-1)[:count]


# bytes one estimate of the count branch may hold for its draws (set by the composed driver around its estimates; None: every
# estimate asks the device what is free when it starts)
_null_memory_hint = None


def set_null_memory_hint(bytes_per_estimate) -> None:
    global _null_memory_hint
    _null_memory_hint = None if bytes_per_estimate is None else int(bytes_per_estimate)


# A thread's workspace for the count branch's draws: an `inference.BlockCarver` over blocks borrowed from the batch scoring
# (set by the composed driver for each of its estimate threads).  With one, the draws' three big tensors -- innovations,
# multipliers / products, rolling variances -- are carved from it: no allocation, and as many draws at once as it holds.
_null_workspace = threading.local()


def set_null_workspace(carver) -> None:
    _null_workspace.carver = carver


def device_multipliers(rng: np.random.Generator, rows: int, n_loci: int, kernel: np.ndarray, out=None, innovations_out=None):
    """What `rows` successive calls of ``_generate_dependent_wild_weights(n_loci, kernel, rng)`` return, as a [rows,
    n_loci] float64 CUDA tensor made on the device: the innovations are NumPy's own stream (`device_standard_normal`),
    the smoothing is a direct sum instead of SciPy's FFT (same values to ~1e-15 of their scale, not bit for bit), centring
    and scaling per row.  `rng` ends where the host calls would leave it.  None: a row came out degenerate (the reference
    then draws signs, rocco/inference.py:565-569) -- `rng` is put back and the caller takes the host path for this draw."""
    import ctypes

    import torch

    rows, n = int(rows), max(1, int(n_loci))
    taps = np.ascontiguousarray(kernel, dtype=np.float64)
    if n == 1:
        return torch.ones((rows, 1), dtype=torch.float64, device=f"cuda:{_dp._device_index()}")
    before = rng.bit_generator.state
    width = n + taps.size - 1
    # (`out` [rows, n] / `innovations_out` (rows x (n + taps - 1) elements): the caller's blocks instead of fresh tensors)
    innovations = device_standard_normal(rng, rows * width, out_t=innovations_out)
    weights = out if out is not None else torch.empty((rows, n), dtype=torch.float64, device=innovations.device)
    if tuple(weights.shape) != (rows, n) or weights.dtype != torch.float64 or not weights.is_contiguous():
        raise ValueError("`out` must be a contiguous float64 [rows, n_loci] CUDA tensor")
    solver = _native.solver_for(weights.device.index)
    degenerate = ctypes.c_int(0)
    _native.check(_native.load().rocco_hip_bartlett_multipliers_f64(
        solver.handle, innovations.data_ptr(), rows, n, taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), taps.size,
        weights.data_ptr(), ctypes.byref(degenerate), _lib_solver_stream(weights)[2]), "rocco_hip_bartlett_multipliers_f64")
    if degenerate.value != 0:
        rng.bit_generator.state = before
        return None
    return weights


def _resolve_multipliers(multipliers: Optional[str]) -> str:
    """"host" (NumPy + SciPy, the reference's own calls: every statistic bit for bit) or "device" (normal.hip: the same
    generator stream, direct smoothing; estimates equal to ~1e-12).  None: the environment's ROCCO_BUDGET_MULTIPLIERS, else
    "host"."""
    import os

    choice = multipliers if multipliers is not None else os.environ.get("ROCCO_BUDGET_MULTIPLIERS", "host")
    if choice not in ("host", "device"):
        raise ValueError("`multipliers` must be 'host' or 'device'")
    return choice


class TrackWeightsAhead:
    """The multipliers of ONE score track's bootstrap draws, made on host threads ahead of their use.

    The draws of a track share one generator (rocco/inference.py:1195-1213), so their normals are drawn one after the
    other; what costs -- SciPy's FFT convolution, 2.5 s per draw of a 5 M-locus track against 0.07 s for its normals --
    does not touch the generator and runs in `ahead` threads, up to that many draws before the one being consumed.  A
    draw whose smoothed series is degenerate takes signs from the generator BEFORE the next draw's normals: the
    generator is put back to that draw's start and the rest runs in sequence.  Draws past the stopping rule's verdict
    are wasted host work, nothing else.  The object starts filling when it is made: the composed driver makes the
    next chromosomes' objects while the current chromosome's estimate runs (rocco_amd/rocco.py)."""

    def __init__(self, n_loci: int, dependence_lag_hint: Optional[int], max_draws: int, random_seed: int = 0, ahead: int = 1):
        self.n = int(n_loci)
        self.taps = _build_budget_bootstrap_kernel(_resolve_budget_bootstrap_bandwidth(self.n, dependence_lag_hint))
        self.max_draws = int(max(1, max_draws))
        self.rng = np.random.default_rng(int(random_seed))
        self.ahead = int(max(1, ahead)) if self.n > 1 else 1
        self.issued = 0
        self.queue = []
        self.pool = None
        if self.ahead > 1:
            import concurrent.futures
            import os

            self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=min(self.ahead, os.cpu_count() or 1),
                                                              thread_name_prefix="rocco-null")
            self._fill()

    def _fill(self) -> None:
        width = self.n + self.taps.size - 1
        while self.issued < self.max_draws and len(self.queue) < self.ahead:
            state = self.rng.bit_generator.state
            self.queue.append((state, self.pool.submit(_smooth_and_standardise, self.rng.standard_normal(width), self.taps)))
            self.issued += 1

    def next(self) -> np.ndarray:
        if self.pool is None:
            return _generate_dependent_wild_weights(self.n, self.taps, self.rng)
        self._fill()
        state, future = self.queue.pop(0)
        weights = future.result()
        if weights is None:  # degenerate: back to this draw's start, in sequence from here on
            self.close()
            self.rng.bit_generator.state = state
            return _generate_dependent_wild_weights(self.n, self.taps, self.rng)
        return weights

    def close(self) -> None:
        if self.pool is not None:
            for _state, later in self.queue:
                later.cancel()
            self.queue.clear()
            self.pool.shutdown(wait=True)
            self.pool = None


class _Running:
    """Welford mean / sum of squared deviations (rocco/inference.py:578-590)."""

    def __init__(self):
        self.count, self.mean, self.m2 = 0, 0.0, 0.0

    def add(self, value: float) -> None:
        self.count += 1
        delta = float(value) - self.mean
        self.mean += delta / float(self.count)
        self.m2 += delta * (float(value) - self.mean)

    def sd(self) -> float:
        return float(np.sqrt(max(self.m2 / float(max(self.count - 1, 1)), 0.0)))

    def stderr(self) -> float:
        return float(np.sqrt(max(self.m2 / float(max(self.count - 1, 1)), 0.0) / float(max(self.count, 1))))


def _stable_enough(run: _Running, min_draws: int, abs_tol: float, rel_tol: float) -> bool:
    """Stopping rule of the draws (rocco/inference.py:593-606)."""
    if run.count < max(2, int(min_draws)):
        return False
    return run.stderr() <= max(abs_tol, rel_tol * max(abs(run.mean), 1.0e-6))


def _estimate_effective_sample_size_device(values_t, max_lag: int) -> Tuple[float, float, int]:
    """n / tau_int with Geyer's positive-pair truncation (rocco/inference.py:446-501); autocovariances on the device."""
    n = int(values_t.shape[0])
    if n < 4:
        return float(max(1, n)), 1.0, 0
    mean = _numpy_mean(values_t)
    lag_cap = int(min(max(2, int(max_lag)), n - 1))
    lib, solver, stream = _lib_solver_stream(values_t)
    sums = (ctypes.c_double * (lag_cap + 1))()
    _native.check(lib.rocco_hip_autocovariance_sums_f64(solver.handle, values_t.data_ptr(), n, float(mean), lag_cap, sums, stream),
                  "rocco_hip_autocovariance_sums_f64")
    acov = np.array(sums[:lag_cap + 1], dtype=np.float64)
    var0 = acov[0] / float(n)  # np.mean(centered * centered)
    if not np.isfinite(var0) or var0 <= 1.0e-12:
        return float(n), 1.0, 0
    acov /= np.arange(n, n - lag_cap - 1, -1, dtype=np.float64)
    if not np.isfinite(acov[0]) or acov[0] <= 1.0e-12:
        return float(n), 1.0, 0
    acf = np.clip(acov[1:] / acov[0], -1.0, 1.0)
    tau, used = 1.0, 0
    for k in range(0, acf.size, 2):
        pair = float(acf[k]) + (float(acf[k + 1]) if k + 1 < acf.size else 0.0)
        if not np.isfinite(pair) or pair <= 0.0:
            break
        tau += 2.0 * pair
        used = int(min(lag_cap, k + 2))
    return float(np.clip(n / max(tau, 1.0), 1.0, n)), float(tau), int(used)


def _estimate_effective_sample_size(values, max_lag: int) -> Tuple[float, float, int]:
    """Same call as the reference's (host array in, three numbers out)."""
    arr = np.asarray(values, dtype=np.float64)
    if arr.ndim != 1:
        raise ValueError("`values` must be one-dimensional")
    if arr.size < 4:
        return float(max(1, arr.size)), 1.0, 0
    return _estimate_effective_sample_size_device(_as_score_tensor(arr), max_lag)


# --------------------------------------------------------------------------------------------------------------
# the estimate
# --------------------------------------------------------------------------------------------------------------

def estimate_budget_nonnull_fraction_from_score_track(score_track, dependence_lag_hint: Optional[int] = None,
                                                      num_null_draws: int = 25, random_seed: int = 0,
                                                      progress_label: Optional[str] = None, num_processes: int = 1,
                                                      return_details: bool = False, min_null_draws: Optional[int] = None,
                                                      stability_abs_tol: float = 5.0e-3, stability_rel_tol: float = 5.0e-2,
                                                      multipliers: Optional[str] = None, weights_source=None):
    """Conservative enriched fraction of a score track (rocco/inference.py:1312-1421): tail occupancy of the observed
    scores above the null's threshold minus that of dependent-wild-bootstrap draws of the one-sided residual track.
    `score_track`: NumPy array or float64 CUDA tensor.  Same return value and details keys as the reference.
    `multipliers` (not in the reference): see `_resolve_multipliers`.  `weights_source` (not in the reference): a
    `TrackWeightsAhead` made for this track earlier, whose host threads have been filling it since."""
    import torch

    s_t = _as_score_tensor(score_track)
    if s_t.dim() != 1:
        raise ValueError("`score_track` must be one-dimensional")
    n = int(s_t.shape[0])
    if n == 0:
        raise ValueError("`score_track` must contain at least one locus")
    lib, solver, stream = _lib_solver_stream(s_t)

    # ---- the null's reference: residual = score - max(score, 0); centre = its median; scale from the mirrored
    # non-positive side (rocco/inference.py:1170-1188).  One sort serves every order statistic.
    template_t = torch.empty_like(s_t)
    _native.check(lib.rocco_hip_negative_part_f64(solver.handle, s_t.data_ptr(), template_t.data_ptr(), n, stream),
                  "rocco_hip_negative_part_f64")
    sorted_template = sort_device(template_t)
    null_center = _median_of_sorted_range(sorted_template, 0, n)
    # residuals (template - centre) <= 0 are the first `m` of the sorted template (x - c is monotone in x)
    _, (m,), (n_below,) = sorted_probe(sorted_template, thresholds=(0.0,), shift=null_center)
    if m == 0:  # cannot happen (the median has at least one element at or below it); kept for the reference's branch
        raise ValueError("Direct-score budget null fit produced non-finite values")
    # magnitudes = -(template - centre) of those, ascending = the sorted residuals read backwards; the mirrored sample
    # (-mag, +mag) has median 0 and absolute deviations (mag, mag): its median is the mean of two magnitudes
    k_lo, k_hi = (m - 1) // 2, m // 2
    (r_lo, r_hi), _, _ = sorted_probe(sorted_template, ranks=(m - 1 - k_lo, m - 1 - k_hi))
    mad = ((-(r_lo - null_center)) + (-(r_hi - null_center))) / 2.0
    null_scale = float(max(mad * _MAD_TO_SIGMA, 1.0e-6))
    if not np.isfinite(null_center) or not np.isfinite(null_scale):
        raise ValueError("Direct-score budget null fit produced non-finite values")
    soft_scale = float(max(null_scale, 1.0e-6))
    null_threshold = float(null_center + 2.0 * null_scale)

    # ---- the draws (rocco/inference.py:1190-1262) ----
    bandwidth = _resolve_budget_bootstrap_bandwidth(n, dependence_lag_hint)
    taps = _build_budget_bootstrap_kernel(bandwidth)
    max_draws = int(max(1, num_null_draws))
    min_draws = int(min(max_draws, max(4, 8 if min_null_draws is None else min_null_draws)))
    mass, units, fraction, tail = _Running(), _Running(), _Running(), _Running()
    product_t = torch.empty_like(s_t)
    weights_dev = torch.empty_like(s_t)
    # Host multipliers come from a `TrackWeightsAhead` (made here, or handed in by a caller that started it earlier);
    # device multipliers (normal.hip) continue the same generator on the device, draw after draw.
    on_device = _resolve_multipliers(multipliers) == "device" and n > 1
    source = weights_source
    if source is None:
        source = TrackWeightsAhead(n, dependence_lag_hint, max_draws, random_seed=random_seed,
                                   ahead=1 if on_device else int(max(1, num_processes)))
    elif source.n != n or source.max_draws != max_draws or not np.array_equal(source.taps, taps):
        raise ValueError("`weights_source` was made for another track")
    elif on_device and (source.pool is not None or source.issued > 0):
        # (a source that makes draws ahead has already taken normals from the generator the device would continue: every
        # draw would come from another place of the stream than the reference's, silently)
        raise ValueError("`weights_source` has drawn ahead on the host: device multipliers need one made with ahead=1 and unused")
    rng = source.rng
    next_weights = source.next
    import time as _time

    for _draw in range(max_draws):
        t_mult = _time.perf_counter()
        made = device_multipliers(rng, 1, n, taps) if on_device else None  # (None: degenerate, generator put back)
        if made is not None:
            weights_dev = made[0]
        else:
            weights_dev.copy_(torch.from_numpy(next_weights()))
        _note("multipliers_device_s" if made is not None else "multipliers_host_s", _time.perf_counter() - t_mult)
        _native.check(lib.rocco_hip_multiply_f64(solver.handle, template_t.data_ptr(), weights_dev.data_ptr(),
                                                 product_t.data_ptr(), n, stream), "rocco_hip_multiply_f64")
        d_mass, d_units, d_fraction, d_tail = _draw_stats(product_t, null_center, soft_scale, null_threshold)
        mass.add(d_mass)
        units.add(d_units)
        fraction.add(d_fraction)
        tail.add(d_tail)
        if _stable_enough(units, min_draws, stability_abs_tol, stability_rel_tol):
            break
    source.close()
    draws_used = units.count

    # ---- observed side and the effective sample size (rocco/inference.py:1340-1366) ----
    obs_mass, obs_units, obs_pos_fraction, obs_tail = _draw_stats(s_t, null_center, soft_scale, null_threshold)
    sorted_scores = sort_device(s_t)
    _, _, (n_neg,) = sorted_probe(sorted_scores, thresholds=(0.0,), shift=null_center)  # residual < 0
    obs_neg_fraction = float(n_neg) / float(n)
    soft_t = torch.empty_like(s_t)
    _native.check(lib.rocco_hip_soft_counts_f64(solver.handle, s_t.data_ptr(), float(null_center), soft_scale, soft_t.data_ptr(),
                                                n, stream), "rocco_hip_soft_counts_f64")
    ess_max_lag = _resolve_budget_ess_max_lag(n, dependence_lag_hint)
    effective_total, tau_int, ess_lags_used = _estimate_effective_sample_size_device(soft_t, ess_max_lag)
    nonnull_fraction = float(np.clip(obs_tail - tail.mean, 0.0, 1.0))
    if not (np.isfinite(nonnull_fraction) and np.isfinite(effective_total) and np.isfinite(tau_int)):
        raise ValueError("Direct-score budget initialization produced non-finite values")
    # positive consensus = clip(scores, 0): its mean / max (rocco/inference.py:1305-1306)
    pos_mean, _u, _f, _t = _draw_stats(s_t, 0.0, 1.0, 0.0)
    (top,), _, _ = sorted_probe(sorted_scores, ranks=(n - 1,))
    details: Dict[str, Any] = {
        "observed_positive_fraction": float(obs_pos_fraction),
        "observed_negative_fraction": float(obs_neg_fraction),
        "null_positive_fraction": float(fraction.mean),
        "observed_excess_mass": float(obs_mass),
        "null_excess_mass": float(mass.mean),
        "observed_excess_units": float(obs_units),
        "null_excess_units": float(units.mean),
        "null_excess_units_sd": float(units.sd()),
        "null_excess_units_stderr": float(units.stderr()),
        "null_threshold": float(null_threshold),
        "observed_tail_occupancy": float(obs_tail),
        "null_tail_occupancy": float(tail.mean),
        "null_tail_occupancy_sd": float(tail.sd()),
        "null_tail_occupancy_stderr": float(tail.stderr()),
        "null_center": float(null_center),
        "null_scale": float(null_scale),
        "nonnull_fraction": float(nonnull_fraction),
        "effective_count": float(nonnull_fraction * effective_total),
        "effective_total_count": float(effective_total),
        "autocorrelation_time": float(tau_int),
        "ess_max_lag": float(ess_max_lag),
        "ess_lags_used": float(ess_lags_used),
        "num_loci": float(n),
        "negative_support_size": float(m),
        "negative_fraction": float(m / max(n, 1)),
        "num_null_draws": float(draws_used),
        "max_null_draws": float(max_draws),
        "adaptive_stop": bool(draws_used < max_draws),
        "wild_bandwidth": float(bandwidth),
        "wild_process": "bartlett_multiplier",
        "null_method": "dependent_wild_score_bootstrap",
        "null_reference_mean_positive_consensus": float(pos_mean),
        "null_reference_max_positive_consensus": float(max(top, 0.0)),
    }
    if return_details:
        return nonnull_fraction, details
    return nonnull_fraction


# --------------------------------------------------------------------------------------------------------------
# the count-matrix (BAM) branch: dependent wild residual bootstrap of the centred K x n matrix
# (rocco/inference.py:719-985 and 988-1148; call-site rocco/rocco.py:1027-1048)
# --------------------------------------------------------------------------------------------------------------

def _as_centered_tensor(centered_matrix):
    """[K, n] float64 CUDA tensor of a centred matrix (a float32 one -- `--low_memory` -- is widened as
    np.asarray(..., dtype=np.float64) widens it, rocco/inference.py:750, 1041)."""
    import torch

    if _dp._resident_tensor(centered_matrix) is not None:
        centered_matrix = _dp._resident_tensor(centered_matrix)
    if _dp._is_tensor(centered_matrix):
        t = centered_matrix
        if not t.is_cuda:
            t = t.to(f"cuda:{_dp._device_index()}")
    else:
        arr = np.asarray(centered_matrix)
        if arr.dtype not in (np.float64, np.float32):
            arr = np.asarray(arr, dtype=np.float64)
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(f"cuda:{_dp._device_index()}")
    if t.dim() == 1:
        t = t[None, :]
    if t.dim() != 2:
        raise ValueError("`centered_matrix` must be one- or two-dimensional")
    return t.to(torch.float64).contiguous()


def _null_reference_of(sorted_t) -> Tuple[float, float, int]:
    """Centre, scale and support of a null's reference scores from their sorted copy (rocco/inference.py:776-787,
    1176-1188): the median; 1.4826 x the median magnitude of the residuals at or below it (the mirrored sample
    (-m, +m) has median 0 and absolute deviations (m, m)); the number of those residuals."""
    n = int(sorted_t.shape[0])
    center = _median_of_sorted_range(sorted_t, 0, n)
    _, (m,), _ = sorted_probe(sorted_t, thresholds=(0.0,), shift=center)  # residuals (x - centre) <= 0 come first
    if m == 0:  # the median always has an element at or below it; the reference's branch for it takes |residuals|
        raise ValueError("Budget null fit produced non-finite values")
    k_lo, k_hi = (m - 1) // 2, m // 2
    (r_lo, r_hi), _, _ = sorted_probe(sorted_t, ranks=(m - 1 - k_lo, m - 1 - k_hi))
    mad = ((-(r_lo - center)) + (-(r_hi - center))) / 2.0
    return float(center), float(max(mad * _MAD_TO_SIGMA, 1.0e-6)), int(m)


def _estimate_wild_bootstrap_score_null(centered_matrix, lower_bound_z: float = 1.0, prior_df: float = 5.0, min_effect=None,
                                        precision_floor_ratio: float = 0.01, observed_scores=None,
                                        dependence_lag_hint: Optional[int] = None, num_null_draws: int = 25,
                                        random_seed: int = 0, progress_label: Optional[str] = None, num_processes: int = 1,
                                        min_null_draws: Optional[int] = None, stability_abs_tol: float = 5.0e-3,
                                        stability_rel_tol: float = 5.0e-2, multipliers: Optional[str] = None) -> Dict[str, Any]:
    """Score null of one chromosome by the dependent wild residual bootstrap (rocco/inference.py:719-985).

    Same arguments and the same dictionary as the reference's (`observed_scores` comes back as a float64 CUDA tensor).
    On the device: the two WLS scorings of the fit (centred matrix, residual template), ONE sort of the null's reference
    scores for its centre and scale, and per draw the K x n product with the multipliers, the WLS rescoring and the four
    means in NumPy's order.  On the host: the multipliers -- draw d takes `default_rng(seed + 104729 (d + 1))` and makes
    row after row with it, as rocco/inference.py:654-664 does -- and the running moments / stopping rule.  The
    reference's worker pool (`num_processes`) only changes how many draws it finishes between two looks at the
    stopping rule; that batching is kept, the draws themselves run one after another on the GPU."""
    import os
    import sys
    import time as _time

    import torch

    from . import inference as _inf

    centered_t = _as_centered_tensor(centered_matrix)
    K, n = int(centered_t.shape[0]), int(centered_t.shape[1])
    if K == 0 or n == 0:
        raise ValueError("`centered_matrix` must be non-empty")
    floor_ratio = float(max(precision_floor_ratio, 0.0))
    # (a workspace of borrowed blocks: the residual template is its first tenant and stays for the whole estimate, when it
    # and at least one draw's three blocks fit)
    carver, template_out = getattr(_null_workspace, "carver", None), None
    if carver is not None:
        carver.clear()
        if centered_t.dtype == torch.float64 and carver.fits([K * n, K * (n + 1024), K * n, K * n]):
            template_out = carver.take(K * n).view(K, n)
            carver.mark()
    template_t, fitted_scores_t, positive_t = _inf.fit_budget_null_residual_template_device(
        centered_t, lower_bound_z=lower_bound_z, prior_df=prior_df, min_effect=min_effect,
        precision_floor_ratio=floor_ratio, residual_out=template_out)
    fit_tracks = torch.stack([fitted_scores_t, positive_t])
    if not bool(torch.isfinite(fit_tracks).all()):
        raise ValueError("EB scoring produced non-finite values")
    if observed_scores is None:
        observed_t = fitted_scores_t
    else:
        observed_t = _as_score_tensor(observed_scores)
        if observed_t.dim() != 1 or int(observed_t.shape[0]) != n:
            raise ValueError("`observed_scores` must have the same number of loci as `centered_matrix`")

    reference_t = _inf.score_centered_wls_device(template_t, lower_bound_z=float(lower_bound_z), prior_df=float(prior_df),
                                                 min_effect=min_effect, spatial_window=31,
                                                 precision_floor_ratio=floor_ratio)[0]
    if not bool(torch.isfinite(reference_t).all()):
        raise ValueError("EB scoring produced non-finite values")
    null_center, null_scale, support = _null_reference_of(sort_device(reference_t))
    if not np.isfinite(null_center) or not np.isfinite(null_scale):
        raise ValueError("Budget null fit produced non-finite values")
    soft_scale = float(max(null_scale, 1.0e-6))
    null_threshold = float(null_center + (2.0 * null_scale))

    bandwidth = _resolve_budget_bootstrap_bandwidth(n, dependence_lag_hint)
    taps = _build_budget_bootstrap_kernel(bandwidth)
    max_draws = int(max(1, num_null_draws))
    min_draws = int(min(max_draws, max(4, 8 if min_null_draws is None else min_null_draws)))
    look_every = int(max(1, min(max(1, int(num_processes)), max_draws)))  # the reference's pool batch
    draw_min_effect = None if min_effect is None else float(max(min_effect, 0.0))
    mass, units, fraction, tail = _Running(), _Running(), _Running(), _Running()
    on_device = _resolve_multipliers(multipliers) == "device" and n > 1
    weights_t = None if on_device else torch.empty_like(template_t)  # (device multipliers arrive in their own tensor)
    product_t = None  # (made when the first draw computed by itself needs it: draws computed together overwrite their multipliers)
    staging = torch.empty((2, n), dtype=torch.float64).pin_memory()
    copies = [torch.cuda.Event(), torch.cuda.Event()]
    # The multipliers of a draw are host work (NumPy's generator and SciPy's FFT convolution) and cost
    # ~100x the draw's device work on a chromosome-sized matrix.  The reference hands the draws of a batch to a pool of
    # `num_processes` workers; here the draws of a batch are generated side by side in threads (both libraries release the
    # GIL in their loops), each into a K x n host array, and consumed IN DRAW ORDER, so the running moments see the same
    # sequence.  One worker (`--low_memory`): the rows stream through two pinned rows and no K x n host array exists.
    pool = None
    if look_every > 1 and not on_device:
        import concurrent.futures

        pool = concurrent.futures.ThreadPoolExecutor(max_workers=min(look_every, os.cpu_count() or 1), thread_name_prefix="rocco-null")

    def host_weights(draw):
        rng = np.random.default_rng(int(random_seed) + (104729 * (draw + 1)))
        block = np.empty((K, n), dtype=np.float64)
        for row in range(K):  # the generator's stream runs through the rows in order
            block[row] = _generate_dependent_wild_weights(n, taps, rng)
        return block

    # at most `inflight` K x n host arrays at a time: half of the host memory that is free now, and never more than the pool
    inflight = look_every
    if pool is not None:
        try:
            import psutil

            inflight = int(max(1, min(look_every, (psutil.virtual_memory().available // 2) // max(1, K * n * 8))))
        except Exception:  # noqa: BLE001 (no psutil: the pool's size is the bound)
            pass
    # device multipliers: the draws between two looks at the stopping rule are computed together, as many at a time as the
    # free memory holds (each keeps its K x n product and its rolling variances): their rolling launch is ONE
    at_once = 1
    workspace = None
    if on_device and look_every > 1 and os.environ.get("ROCCO_BUDGET_NULL_DRAWS_AT_ONCE", "") != "1":
        per_draw = max(1, 3 * K * n * 8)  # (innovations while the multipliers are made, then product and rolling variances)
        carver = getattr(_null_workspace, "carver", None)
        if carver is not None:
            # (blocks borrowed from the batch scoring: as many draws at once as they hold -- innovations, then one block of
            # multipliers per draw, then the variances of all of them)
            for a in range(int(look_every), 0, -1):
                carver.reset()
                if carver.fits([K * (n + len(taps) - 1)] + [K * n] * a + [a * K * n]):
                    workspace, at_once = carver, a
                    break
        if workspace is not None:
            pass
        elif _null_memory_hint is not None:
            # (the composed driver looked at the device once, before its estimates started side by side: estimates that each
            # ask what is free NOW, while the others allocate, see numbers that mean little)
            at_once = int(max(1, min(look_every, int(_null_memory_hint) // per_draw)))
        else:
            free_now, _total = torch.cuda.mem_get_info(template_t.device)
            cached = max(0, int(torch.cuda.memory_reserved(template_t.device)) - int(torch.cuda.memory_allocated(template_t.device)))
            share = max(1, int(os.environ.get("ROCCO_BUDGET_NULL_STREAMS", "3")))  # (estimates may run side by side on that many streams)
            at_once = int(max(1, min(look_every, (6 * (free_now + cached) // (10 * share)) // per_draw)))
    for first in range(0, max_draws, look_every):
        batch = list(range(first, min(max_draws, first + look_every)))
        pending, to_submit = {}, list(batch)
        if at_once > 1 or workspace is not None:
            at = 0
            while at < len(batch):
                group, weights = batch[at:at + at_once], []
                t_mult = _time.perf_counter()
                try:
                    variances_block = None
                    if workspace is not None:
                        workspace.reset()
                        innovations_block = workspace.take(K * (n + len(taps) - 1))
                        blocks = [workspace.take(K * n).view(K, n) for _draw in group]
                        variances_block = workspace.take(len(group) * K * n)
                    for j, draw in enumerate(group):
                        rng_draw = np.random.default_rng(int(random_seed) + (104729 * (draw + 1)))
                        if workspace is not None:
                            made = device_multipliers(rng_draw, K, n, taps, out=blocks[j], innovations_out=innovations_block)
                        else:
                            made = device_multipliers(rng_draw, K, n, taps)
                        if made is None:  # (a degenerate row: the reference's own calls)
                            made = torch.from_numpy(host_weights(draw)).to(template_t.device)
                            if workspace is not None:
                                blocks[j].copy_(made)
                                made = blocks[j]
                        weights.append(made)
                    _note("multipliers_device_s", _time.perf_counter() - t_mult)
                    results = _inf.compute_budget_null_draws_device(
                        template_t, weights, lower_bound_z, prior_df, draw_min_effect, floor_ratio, null_center, soft_scale, null_threshold,
                        variances_arena=variances_block)
                except (torch.OutOfMemoryError, MemoryError):
                    # (the draws are seeded one by one: the same group again, fewer at a time, gives the same numbers)
                    if at_once == 1:
                        raise
                    del weights
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    at_once = max(1, at_once // 2)
                    continue
                for d_mass, d_units, d_fraction, d_tail in results:
                    mass.add(d_mass)
                    units.add(d_units)
                    fraction.add(d_fraction)
                    tail.add(d_tail)
                del weights, results
                torch.cuda.current_stream().synchronize()
                if progress_label:
                    sys.stderr.write(f"\r{progress_label}: {units.count}/{max_draws}")
                    sys.stderr.flush()
                at += at_once
            if _stable_enough(units, min_draws, stability_abs_tol, stability_rel_tol):
                break
            continue
        for draw in batch:
            made = None
            t_mult = _time.perf_counter()
            if on_device:
                # the draw's generator is seeded on the host, its stream runs through the rows on the device (normal.hip)
                made = device_multipliers(np.random.default_rng(int(random_seed) + (104729 * (draw + 1))), K, n, taps)
            if made is not None:
                draw_weights = made
            elif on_device:
                draw_weights = torch.from_numpy(host_weights(draw)).to(template_t.device)  # (a degenerate row: the reference's own calls)
            elif pool is not None:
                while to_submit and len(pending) < inflight:
                    ahead = to_submit.pop(0)
                    pending[ahead] = pool.submit(host_weights, ahead)
                weights_t.copy_(torch.from_numpy(pending.pop(draw).result()))
                draw_weights = weights_t
            else:
                draw_weights = weights_t
                rng = np.random.default_rng(int(random_seed) + (104729 * (draw + 1)))
                for row in range(K):  # uploads overlap the next row's generation
                    slot = row & 1
                    if row >= 2:
                        copies[slot].synchronize()
                    staging[slot].numpy()[:] = _generate_dependent_wild_weights(n, taps, rng)
                    weights_t[row].copy_(staging[slot], non_blocking=True)
                    copies[slot].record()
            _note("multipliers_device_s" if made is not None else "multipliers_host_s", _time.perf_counter() - t_mult)
            if product_t is None:
                product_t = torch.empty_like(template_t)
            d_mass, d_units, d_fraction, d_tail = _inf.compute_budget_null_draw_device(
                template_t, draw_weights, lower_bound_z, prior_df, draw_min_effect, floor_ratio, null_center, soft_scale,
                null_threshold, work_t=product_t)
            del made, draw_weights
            torch.cuda.current_stream().synchronize()
            mass.add(d_mass)
            units.add(d_units)
            fraction.add(d_fraction)
            tail.add(d_tail)
            if progress_label:
                sys.stderr.write(f"\r{progress_label}: {units.count}/{max_draws}")
                sys.stderr.flush()
        if _stable_enough(units, min_draws, stability_abs_tol, stability_rel_tol):
            break
    if pool is not None:
        pool.shutdown(wait=True)
    if progress_label:
        sys.stderr.write("\n")
        sys.stderr.flush()
    draws_used = units.count
    return {
        "observed_scores": observed_t,
        "null_center": float(null_center),
        "null_scale": float(null_scale),
        "null_positive_mass": float(mass.mean),
        "null_positive_units": float(units.mean),
        "null_positive_fraction": float(fraction.mean),
        "null_positive_units_sd": float(units.sd()),
        "null_positive_units_stderr": float(units.stderr()),
        "null_threshold": float(null_threshold),
        "null_tail_occupancy": float(tail.mean),
        "null_tail_occupancy_sd": float(tail.sd()),
        "null_tail_occupancy_stderr": float(tail.stderr()),
        "negative_support_size": int(support),
        "negative_fraction": float(support / max(n, 1)),
        "num_null_draws": int(draws_used),
        "max_null_draws": int(max_draws),
        "adaptive_stop": bool(draws_used < max_draws),
        "wild_bandwidth": int(bandwidth),
        "wild_process": "bartlett_multiplier",
        "null_method": "dependent_wild_residual_bootstrap",
        "null_reference_mean_positive_consensus": float(_numpy_mean(positive_t)),
        "null_reference_max_positive_consensus": float(positive_t.max().item()),
    }


def estimate_budget_nonnull_fraction_from_wild_bootstrap_null(centered_matrix, observed_scores=None, lower_bound_z: float = 1.0,
                                                              prior_df: float = 5.0, min_effect=None,
                                                              precision_floor_ratio: float = 0.01,
                                                              dependence_lag_hint: Optional[int] = None,
                                                              num_null_draws: int = 25, random_seed: int = 0,
                                                              progress_label: Optional[str] = None, num_processes: int = 1,
                                                              return_details: bool = False, multipliers: Optional[str] = None):
    """Conservative enriched fraction of a chromosome from its centred K x n matrix (rocco/inference.py:988-1148): the
    tail occupancy of the observed scores above the fitted null's threshold (centre + 2 scale) minus the average of the
    same statistic over dependent-wild-bootstrap draws of the residual template, clipped to [0, 1].
    `centered_matrix`: NumPy array or CUDA tensor (float64 or float32), [K, n] or [n]; `observed_scores`: NumPy array or
    float64 CUDA tensor.  Same return value and details keys as the reference."""
    import torch

    centered_t = _as_centered_tensor(centered_matrix)
    n = int(centered_t.shape[1])
    if n <= 0:
        raise ValueError("`centered_matrix` must contain at least one locus")
    null = _estimate_wild_bootstrap_score_null(
        centered_t, lower_bound_z=lower_bound_z, prior_df=prior_df, min_effect=min_effect,
        precision_floor_ratio=precision_floor_ratio, observed_scores=observed_scores,
        dependence_lag_hint=dependence_lag_hint, num_null_draws=num_null_draws, random_seed=random_seed,
        progress_label=progress_label, num_processes=num_processes, multipliers=multipliers)
    s_t = null["observed_scores"]
    null_center, null_scale = float(null["null_center"]), float(null["null_scale"])
    soft_scale = float(max(null_scale, 1.0e-6))
    null_threshold = float(null["null_threshold"])
    lib, solver, stream = _lib_solver_stream(s_t)
    obs_mass, obs_units, obs_pos_fraction, obs_tail = _draw_stats(s_t, null_center, soft_scale, null_threshold)
    _, _, (n_neg,) = sorted_probe(sort_device(s_t), thresholds=(0.0,), shift=null_center)  # residual < 0
    obs_neg_fraction = float(n_neg) / float(n)
    soft_t = torch.empty_like(s_t)
    _native.check(lib.rocco_hip_soft_counts_f64(solver.handle, s_t.data_ptr(), float(null_center), soft_scale, soft_t.data_ptr(),
                                                n, stream), "rocco_hip_soft_counts_f64")
    ess_max_lag = _resolve_budget_ess_max_lag(n, dependence_lag_hint)
    effective_total, tau_int, ess_lags_used = _estimate_effective_sample_size_device(soft_t, ess_max_lag)
    nonnull_fraction = float(np.clip(obs_tail - float(null["null_tail_occupancy"]), 0.0, 1.0))
    if not (np.isfinite(nonnull_fraction) and np.isfinite(effective_total) and np.isfinite(tau_int)):
        raise ValueError("Budget initialization produced non-finite values")
    details: Dict[str, Any] = {
        "observed_positive_fraction": float(obs_pos_fraction),
        "observed_negative_fraction": float(obs_neg_fraction),
        "null_positive_fraction": float(null["null_positive_fraction"]),
        "observed_excess_mass": float(obs_mass),
        "null_excess_mass": float(null["null_positive_mass"]),
        "observed_excess_units": float(obs_units),
        "null_excess_units": float(null["null_positive_units"]),
        "null_excess_units_sd": float(null["null_positive_units_sd"]),
        "null_excess_units_stderr": float(null["null_positive_units_stderr"]),
        "null_threshold": float(null_threshold),
        "observed_tail_occupancy": float(obs_tail),
        "null_tail_occupancy": float(null["null_tail_occupancy"]),
        "null_tail_occupancy_sd": float(null["null_tail_occupancy_sd"]),
        "null_tail_occupancy_stderr": float(null["null_tail_occupancy_stderr"]),
        "null_center": float(null_center),
        "null_scale": float(null_scale),
        "nonnull_fraction": float(nonnull_fraction),
        "effective_count": float(nonnull_fraction * effective_total),
        "effective_total_count": float(effective_total),
        "autocorrelation_time": float(tau_int),
        "ess_max_lag": float(ess_max_lag),
        "ess_lags_used": float(ess_lags_used),
        "num_loci": float(n),
        "negative_support_size": float(null["negative_support_size"]),
        "negative_fraction": float(null["negative_fraction"]),
        "num_null_draws": float(null["num_null_draws"]),
        "max_null_draws": float(null["max_null_draws"]),
        "adaptive_stop": bool(null["adaptive_stop"]),
        "wild_bandwidth": float(null["wild_bandwidth"]),
        "wild_process": str(null["wild_process"]),
        "null_method": str(null["null_method"]),
        "null_reference_mean_positive_consensus": float(null["null_reference_mean_positive_consensus"]),
        "null_reference_max_positive_consensus": float(null["null_reference_max_positive_consensus"]),
    }
    if return_details:
        return nonnull_fraction, details
    return nonnull_fraction


def estimate_budget_nonnull_fraction_from_empirical_null(centered_matrix, observed_scores=None, lower_bound_z: float = 1.0,
                                                         prior_df: float = 5.0, min_effect=None,
                                                         precision_floor_ratio: float = 0.01,
                                                         dependence_lag_hint: Optional[int] = None, num_null_draws: int = 25,
                                                         random_seed: int = 0, progress_label: Optional[str] = None,
                                                         num_processes: int = 1, return_details: bool = False):
    """The reference's older name of the estimator above (rocco/inference.py:1424-1452)."""
    return estimate_budget_nonnull_fraction_from_wild_bootstrap_null(
        centered_matrix, observed_scores=observed_scores, lower_bound_z=lower_bound_z, prior_df=prior_df,
        min_effect=min_effect, precision_floor_ratio=precision_floor_ratio, dependence_lag_hint=dependence_lag_hint,
        num_null_draws=num_null_draws, random_seed=random_seed, progress_label=progress_label,
        num_processes=num_processes, return_details=return_details)


def estimate_budget_nonnull_fraction_from_resampled_null(centered_matrix, observed_scores=None, lower_bound_z: float = 1.0,
                                                         prior_df: float = 5.0, min_effect=None,
                                                         precision_floor_ratio: float = 0.01, num_null_draws: int = 25,
                                                         mean_block_length: Optional[int] = None,
                                                         null_threshold_scale: float = 1.0, random_seed: int = 0,
                                                         progress_label: Optional[str] = None, num_processes: int = 1,
                                                         return_details: bool = False):
    """Another older name (rocco/inference.py:1455-1485): `mean_block_length` is the dependence hint, the threshold
    scale is ignored there too."""
    return estimate_budget_nonnull_fraction_from_wild_bootstrap_null(
        centered_matrix, observed_scores=observed_scores, lower_bound_z=lower_bound_z, prior_df=prior_df,
        min_effect=min_effect, precision_floor_ratio=precision_floor_ratio, dependence_lag_hint=mean_block_length,
        num_null_draws=num_null_draws, random_seed=random_seed, progress_label=progress_label,
        num_processes=num_processes, return_details=return_details)


# --------------------------------------------------------------------------------------------------------------
# switch cost (rocco/rocco.py:751-789)
# --------------------------------------------------------------------------------------------------------------

def _resolve_chrom_gamma(chrom: str, args: dict, chrom_scores, budget_rate_meta: dict):
    """A fixed `--gamma`, or 0.5 x ceil(autocorrelation time) x median of the positive scores clipped to [0.5, 10];
    same return value (gamma, metadata or None) as the reference."""
    if args["gamma"] is not None:
        fixed = float(args["gamma"])
        if not np.isfinite(fixed) or fixed < 0.0:
            raise ValueError("`--gamma` must be finite and non-negative")
        return fixed, None
    s_t = _as_score_tensor(chrom_scores)
    n = int(s_t.shape[0])
    positive_count, positive_median = 0, 1.0
    if n > 0:
        ordered = sort_device(s_t)
        _, (not_positive,), _ = sorted_probe(ordered, thresholds=(0.0,))  # scores <= 0 come first
        positive_count = n - not_positive
        if positive_count > 0:
            positive_median = _median_of_sorted_range(ordered, not_positive, positive_count)
    tau = max(1.0, float(budget_rate_meta.get("autocorrelation_time", 1.0)))
    run_length = int(np.ceil(tau))
    raw = 0.5 * float(run_length) * float(positive_median)
    gamma = float(np.clip(raw, 0.5, 10.0))
    meta = {"method": "auto_score_autocorr", "autocorrelation_time": float(tau), "characteristic_run_length": int(run_length),
            "positive_score_median": float(positive_median), "positive_score_count": int(positive_count),
            "gamma_raw": float(raw), "gamma_clipped": float(gamma), "gamma_clip_min": 0.5, "gamma_clip_max": 10.0}
    logger.info("%s auto gamma estimate: %s", chrom, meta)
    return gamma, meta


# --------------------------------------------------------------------------------------------------------------
# empirical-Bayes pooling across chromosomes (rocco/inference.py:1488-1737) -- scalar host code over <= 24 pairs
# --------------------------------------------------------------------------------------------------------------

def fit_beta_prior_mle(successes, totals, init_center: float = 0.05, init_strength: float = 10.0) -> Tuple[float, float]:
    """Beta prior of the per-chromosome rates by beta-binomial maximum likelihood (rocco/inference.py:1488-1562)."""
    from scipy import optimize, special

    x = np.asarray(successes, dtype=np.float64)
    t = np.asarray(totals, dtype=np.float64)
    if x.shape != t.shape:
        raise ValueError("`successes` and `totals` must have the same shape")
    if x.size == 0:
        return 1.0, 1.0
    center = min(max(float(init_center), 1.0e-6), 1.0 - 1.0e-6)
    pooled, seen_var, floor_var = _rate_dispersion(x, t)
    if seen_var <= floor_var + 1.0e-12:  # no dispersion beyond the binomial: a (practically) degenerate prior at the pooled rate
        strength = float(max(1.0e12, 100.0 * np.max(t)))
        return pooled * strength, (1.0 - pooled) * strength

    def negative_log_likelihood(theta):
        a, b = float(np.exp(theta[0])), float(np.exp(theta[1]))
        return float(-np.sum(special.betaln(x + a, t - x + b) - special.betaln(a, b)))

    start = np.log(np.array([center * float(init_strength), (1.0 - center) * float(init_strength)], dtype=np.float64))
    fit = optimize.minimize(negative_log_likelihood, start, method="L-BFGS-B")
    if not fit.success:
        logger.warning("Falling back to a weak beta prior while fitting EB budgets: %s", fit.message)
        return center * float(init_strength), (1.0 - center) * float(init_strength)
    return float(np.exp(fit.x[0])), float(np.exp(fit.x[1]))


def _rate_dispersion(x: np.ndarray, t: np.ndarray) -> Tuple[float, float, float]:
    """Pooled rate, the variance of the raw rates across chromosomes, and its binomial floor."""
    rates = x / np.maximum(t, 1.0)
    pooled = float(np.clip(np.sum(x) / max(np.sum(t), 1.0), 1.0e-6, 1.0 - 1.0e-6))
    seen = float(np.var(rates, ddof=1)) if rates.size > 1 else 0.0
    floor = float(pooled * (1.0 - pooled) * np.mean(1.0 / np.maximum(t, 1.0)))
    return pooled, seen, floor


def _posterior_budget(x: float, t: float, a: float, b: float, quantile: float, lo: float, hi: float) -> float:
    """Quantile of the Beta(x + a, t - x + b) posterior, clipped to the budget range (rocco/inference.py:1565-1590)."""
    from scipy import stats

    q = float(np.clip(quantile, 1.0e-6, 1.0 - 1.0e-6))
    value = float(stats.beta.ppf(q, float(max(1.0e-12, x + a)), float(max(1.0e-12, (t - x) + b))))
    return float(np.clip(value, lo, hi))


def estimate_empirical_bayes_budgets(chrom_candidate_counts: Dict[str, float], chrom_total_counts: Dict[str, float],
                                     min_budget: float = 1.0e-4, max_budget: float = 0.5, init_center: float = 0.05,
                                     init_strength: float = 10.0, posterior_quantile: float = 0.01):
    """Per-chromosome budgets shrunk towards a genome-wide beta prior (rocco/inference.py:1593-1737); same return value
    (budgets, metadata) and metadata keys as the reference."""
    chroms = list(chrom_candidate_counts)
    if chroms != list(chrom_total_counts):
        raise ValueError("`chrom_candidate_counts` and `chrom_total_counts` must share keys in the same order")
    x = np.array([chrom_candidate_counts[c] for c in chroms], dtype=np.float64)
    t = np.array([chrom_total_counts[c] for c in chroms], dtype=np.float64)
    pooled, seen_var, floor_var = _rate_dispersion(x, t)
    q = float(posterior_quantile)
    if not (0.0 < q < 1.0):
        raise ValueError("`posterior_quantile` must lie strictly between 0 and 1")
    at_floor = bool(seen_var <= floor_var + 1.0e-12)
    if len(chroms) <= 1:  # nothing to pool: the default prior
        a, b = float(init_center) * float(init_strength), (1.0 - float(init_center)) * float(init_strength)
        method, genome_wide, strength = "single_chrom_default", float(init_center), float(init_strength)
        dispersion, flag = float(1.0 / (1.0 + a + b)), False
    elif len(chroms) <= 3:  # too few chromosomes to fit a dispersion: a weak prior at the pooled rate
        a, b = float(pooled) * float(init_strength), (1.0 - float(pooled)) * float(init_strength)
        method, genome_wide, strength = "weak_pooled_prior", float(pooled), float(a + b)
        dispersion, flag = float(max(0.0, 1.0 / (1.0 + strength))), at_floor
    else:
        a, b = fit_beta_prior_mle(x, t, init_center=init_center, init_strength=init_strength)
        method, genome_wide, strength = "beta_binomial_mle", float(a / (a + b)), float(a + b)
        dispersion, flag = float(max(0.0, 1.0 / (1.0 + strength))), at_floor
    budgets = {c: _posterior_budget(float(x[i]), float(t[i]), a, b, q, min_budget, max_budget) for i, c in enumerate(chroms)}
    meta = {"alpha": float(a), "beta": float(b), "genome_wide_budget": float(genome_wide), "prior_strength": float(strength),
            "prior_dispersion": float(dispersion), "min_prior_dispersion": 0.0, "observed_raw_budget_var": float(seen_var),
            "theoretical_min_raw_budget_var": float(floor_var), "prior_dispersion_at_floor": bool(flag),
            "posterior_summary": "beta_quantile", "posterior_quantile": float(q), "prior_fit_method": method}
    return budgets, meta


def _resolve_budgets(chrom_cache: dict, args: dict):
    """Budgets of every chromosome of the cache (rocco/rocco.py:1113-1143): the empirical-Bayes estimates, rescaled to
    a requested genome-wide `--budget`, times `--scale_chrom_budgets`, clipped to [0.005, 0.1]."""
    counts = {c: chrom_cache[c]["budget_count_hat"] for c in chrom_cache}
    totals = {c: chrom_cache[c]["total_count"] for c in chrom_cache}
    budgets, meta = estimate_empirical_bayes_budgets(counts, totals, posterior_quantile=args["budget_posterior_quantile"])
    rescale = 1.0
    if args["budget"] is not None and meta["genome_wide_budget"] > 0:
        rescale = float(args["budget"]) / meta["genome_wide_budget"]
    scale = float(args["scale_chrom_budgets"])
    budgets = {c: min(max(budgets[c] * rescale * scale, 0.005), 0.1) for c in budgets}
    logger.info("Empirical-Bayes budget prior: %s", meta)
    return budgets, meta


# --------------------------------------------------------------------------------------------------------------
# the bigWig branch of _build_chrom_cache (rocco/rocco.py:977-1008, 1049-1097) for matrices held in HBM
# --------------------------------------------------------------------------------------------------------------

def build_chrom_cache_from_tracks(chrom_tracks: Dict[str, Tuple[Any, Any]], args: dict) -> dict:
    """`chrom_tracks[chrom] = (intervals, matrix)` with `matrix` a [K, n] NumPy array or CUDA tensor of signal tracks.
    Returns the reference's cache entries (scores as float64 CUDA tensors): scores, gamma, gamma_meta,
    budget_count_hat, budget_fraction_hat, budget_rate_meta, total_count, num_loci, intervals."""
    import torch

    from . import rocco as _rocco

    cache = {}
    for chrom, (intervals, matrix) in chrom_tracks.items():
        m_t = matrix if _dp._is_tensor(matrix) else torch.from_numpy(np.ascontiguousarray(matrix)).to(f"cuda:{_dp._device_index()}")
        if not bool(torch.isfinite(m_t).all()):
            raise ValueError(f"{chrom} matrix contains non-finite values")
        scores_t = _rocco.score_central_tendency_chrom_device(m_t)
        if not bool(torch.isfinite(scores_t).all()):
            raise ValueError(f"{chrom} direct scores contain non-finite values")
        fraction, rate_meta = estimate_budget_nonnull_fraction_from_score_track(
            scores_t, num_null_draws=args["budget_null_draws"], return_details=True)
        if not np.isfinite(fraction):
            raise ValueError(f"{chrom} budget estimate is not finite")
        n = int(scores_t.shape[0])
        total = float(np.clip(rate_meta.get("effective_total_count", n), 1.0, n))
        count = float(np.clip(fraction * total, 0.0, total))
        gamma, gamma_meta = _resolve_chrom_gamma(chrom, args, scores_t, rate_meta)
        cache[chrom] = {"intervals": intervals, "scores": scores_t, "gamma": gamma, "gamma_meta": gamma_meta,
                        "budget_count_hat": count, "budget_fraction_hat": float(fraction), "budget_rate_meta": rate_meta,
                        "total_count": total, "num_loci": n}
    return cache
