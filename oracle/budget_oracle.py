"""oracle/budget_oracle.py -- TEST INFRASTRUCTURE ONLY: CPU restatement (NumPy / SciPy) of what stands between the
scores and the solve in the reference, and of the driver that strings the whole path together.

    null fit + dependent wild bootstrap of a score track       rocco/inference.py:1151-1309, 1312-1421
    the same over a centred K x n matrix (count path)          rocco/inference.py:719-985, 988-1148
    multipliers, running moments, stopping rule                rocco/inference.py:520-601
    effective sample size (FFT autocovariance, Geyer pairs)    rocco/inference.py:446-517
    automatic switch cost                                      rocco/rocco.py:751-789
    beta-binomial pooling of the chromosome budgets            rocco/inference.py:1488-1737, rocco/rocco.py:1113-1143
    cache -> budgets -> solve -> BED records                   rocco/rocco.py:933-1110, 1146-1196, 194-240

Own code in own form; the arithmetic follows the reference statement by statement wherever a rounding could differ
(NumPy's reductions, the FFT, the generator's call order).  Pinned against outputs of the reference itself:
tests/golden/budget_vectors.npz, wild_bootstrap_vectors.npz, composed_vectors.npz (tests/test_oracle_golden.py).
Imported by tests/, bench.py's checker legs and __graft_entry__.smoke() only -- never by rocco_amd/."""
from __future__ import annotations

import multiprocessing
import os
from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import numpy as np

_MAD_SIGMA = 1.4826


# ---------------------------------------------------------------------------------------------------------------------
# pieces shared by both estimators
# ---------------------------------------------------------------------------------------------------------------------

def robust_scale(values, floor: float = 1.0e-6) -> float:
    """rocco/inference.py:32-37."""
    v = np.asarray(values, dtype=np.float64)
    if v.size == 0:
        return float(floor)
    return float(max(np.median(np.abs(v - np.median(v))) * _MAD_SIGMA, floor))


def ess_max_lag(n_loci: int, hint: Optional[int] = None) -> int:
    """rocco/inference.py:504-517."""
    n = int(max(1, n_loci))
    scale = min(n, 101) if hint is None else max(1, min(n, int(hint)))
    return int(min(n - 1, max(16, 4 * scale)))


def bootstrap_bandwidth(n_loci: int, hint: Optional[int] = None) -> int:
    """rocco/inference.py:520-530."""
    n = int(max(1, n_loci))
    if n <= 1:
        return 1
    if hint is None:
        return int(min(n - 1, max(8, round(n ** (1.0 / 3.0)))))
    return int(min(n - 1, max(8, int(hint))))


def bartlett_kernel(bandwidth: int) -> np.ndarray:
    """rocco/inference.py:533-541."""
    b = int(max(1, bandwidth))
    support = np.arange(-b, b + 1, dtype=np.float64)
    k = np.maximum(1.0 - (np.abs(support) / float(b + 1)), 0.0)
    k /= np.sqrt(np.sum(k * k))
    return k


def dependent_wild_weights(n_loci: int, kernel: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """rocco/inference.py:544-570: the generator calls and the reductions in the reference's order."""
    from scipy import signal

    n = int(max(1, n_loci))
    if n == 1:
        return np.ones(1, dtype=np.float64)
    k = np.asarray(kernel, dtype=np.float64)
    w = np.asarray(signal.fftconvolve(rng.standard_normal(n + k.size - 1), k, mode="valid"), dtype=np.float64)
    w -= float(np.mean(w))
    sd = float(np.std(w))
    if not np.isfinite(sd) or sd <= 1.0e-8:
        flip = rng.choice(np.array([-1.0, 1.0], dtype=np.float64), size=n)
        flip -= float(np.mean(flip))
        return flip / max(float(np.std(flip)), 1.0e-6)
    return w / sd


class Moments:
    """Welford updates of rocco/inference.py:573-585 and the summaries taken from them (943-958)."""

    def __init__(self):
        self.count, self.mean, self.m2 = 0, 0.0, 0.0

    def push(self, value: float) -> None:
        self.count += 1
        d = float(value) - self.mean
        self.mean = self.mean + (d / float(self.count))
        self.m2 = self.m2 + (d * (float(value) - self.mean))

    def sd(self) -> float:
        return float(np.sqrt(max(self.m2 / float(max(self.count - 1, 1)), 0.0)))

    def stderr(self) -> float:
        return float(np.sqrt(max(self.m2 / float(max(self.count - 1, 1)), 0.0) / float(max(self.count, 1))))

    def settled(self, min_draws: int, abs_tol: float, rel_tol: float) -> bool:
        """rocco/inference.py:588-601."""
        if self.count < int(max(2, min_draws)):
            return False
        return bool(self.stderr() <= float(max(abs_tol, rel_tol * max(abs(self.mean), 1.0e-6))))


def effective_sample_size(values, max_lag: int) -> Tuple[float, float, int]:
    """rocco/inference.py:446-501."""
    v = np.asarray(values, dtype=np.float64)
    if v.ndim != 1:
        raise ValueError("`values` must be one-dimensional")
    n = int(v.size)
    if n < 4:
        return float(max(1, n)), 1.0, 0
    c = v - float(np.mean(v))
    var0 = float(np.mean(c * c))
    if not np.isfinite(var0) or var0 <= 1.0e-12:
        return float(n), 1.0, 0
    lag = int(min(max(2, max_lag), n - 1))
    n_fft = 1 << int(np.ceil(np.log2((2 * n) - 1)))
    spec = np.fft.rfft(c, n=n_fft)
    acov = np.fft.irfft(spec * np.conjugate(spec), n=n_fft)[: lag + 1]
    acov /= np.arange(n, n - lag - 1, -1, dtype=np.float64)
    if not np.isfinite(acov[0]) or acov[0] <= 1.0e-12:
        return float(n), 1.0, 0
    acf = np.clip(acov[1:] / acov[0], -1.0, 1.0)
    tau, used = 1.0, 0
    for i in range(0, int(acf.size), 2):
        pair = float(acf[i]) + (float(acf[i + 1]) if (i + 1) < acf.size else 0.0)
        if not np.isfinite(pair) or pair <= 0.0:
            break
        tau += 2.0 * pair
        used = int(min(lag, i + 2))
    return float(np.clip(n / max(tau, 1.0), 1.0, n)), float(tau), int(used)


def _null_fit(reference_scores: np.ndarray, what: str) -> Tuple[float, float, int]:
    """Centre, scale and support of the fitted null (rocco/inference.py:776-789, 1172-1184)."""
    center = float(np.median(reference_scores))
    resid = reference_scores - center
    below = resid[resid <= 0.0]
    mags = np.abs(resid) if below.size == 0 else -below
    if mags.size == 0:
        mags = np.array([0.0], dtype=np.float64)
    scale = float(robust_scale(np.concatenate((-mags, mags))))
    if not np.isfinite(center) or not np.isfinite(scale):
        raise ValueError(f"{what} null fit produced non-finite values")
    return center, scale, int(mags.size)


def _draw_statistics(scores: np.ndarray, center: float, soft_scale: float, threshold: float):
    pos = np.clip(scores - center, 0.0, None)
    return (float(np.mean(pos)), float(np.mean(pos / soft_scale)), float(np.mean(pos > 0.0)),
            float(np.mean(scores > threshold)))


def _finish(observed: np.ndarray, n_loci: int, hint, null: Dict[str, Any], what: str):
    """The observed side, the effective sample size and the details dictionary (rocco/inference.py:1064-1148, 1336-1421)."""
    center, scale = float(null["null_center"]), float(null["null_scale"])
    soft = float(max(scale, 1.0e-6))
    resid = observed - center
    excess = np.clip(resid, 0.0, None)
    shortfall = np.clip(-resid, 0.0, None)
    soft_counts = excess / soft
    lag_cap = ess_max_lag(n_loci, hint)
    eff_total, tau, lags_used = effective_sample_size(soft_counts, lag_cap)
    obs_tail = float(np.mean(observed > float(null["null_threshold"])))
    fraction = float(np.clip(obs_tail - float(null["null_tail_occupancy"]), 0.0, 1.0))
    if not np.isfinite(fraction) or not np.isfinite(eff_total) or not np.isfinite(tau):
        raise ValueError(f"{what} initialization produced non-finite values")
    details = {
        "observed_positive_fraction": float(np.mean(excess > 0.0)),
        "observed_negative_fraction": float(np.mean(shortfall > 0.0)),
        "null_positive_fraction": float(null["null_positive_fraction"]),
        "observed_excess_mass": float(np.mean(excess)),
        "null_excess_mass": float(null["null_positive_mass"]),
        "observed_excess_units": float(np.mean(soft_counts)),
        "null_excess_units": float(null["null_positive_units"]),
        "null_excess_units_sd": float(null["null_positive_units_sd"]),
        "null_excess_units_stderr": float(null["null_positive_units_stderr"]),
        "null_threshold": float(null["null_threshold"]),
        "observed_tail_occupancy": obs_tail,
        "null_tail_occupancy": float(null["null_tail_occupancy"]),
        "null_tail_occupancy_sd": float(null["null_tail_occupancy_sd"]),
        "null_tail_occupancy_stderr": float(null["null_tail_occupancy_stderr"]),
        "null_center": center,
        "null_scale": scale,
        "nonnull_fraction": fraction,
        "effective_count": float(fraction * eff_total),
        "effective_total_count": float(eff_total),
        "autocorrelation_time": float(tau),
        "ess_max_lag": float(lag_cap),
        "ess_lags_used": float(lags_used),
        "num_loci": float(n_loci),
        "negative_support_size": float(null["negative_support_size"]),
        "negative_fraction": float(null["negative_fraction"]),
        "num_null_draws": float(null["num_null_draws"]),
        "max_null_draws": float(null["max_null_draws"]),
        "adaptive_stop": bool(null["adaptive_stop"]),
        "wild_bandwidth": float(null["wild_bandwidth"]),
        "wild_process": "bartlett_multiplier",
        "null_method": str(null["null_method"]),
        "null_reference_mean_positive_consensus": float(null["null_reference_mean_positive_consensus"]),
        "null_reference_max_positive_consensus": float(null["null_reference_max_positive_consensus"]),
    }
    return fraction, details


def _run_draws(draw: Callable[[int], Tuple[float, float, float, float]], max_draws: int, min_draws: int, look_every: int,
               abs_tol: float, rel_tol: float):
    """Draw until the mean of the second statistic has settled, looking every `look_every` draws."""
    mass, units, fraction, tail = Moments(), Moments(), Moments(), Moments()
    for first in range(0, max_draws, look_every):
        for d in range(first, min(max_draws, first + look_every)):
            a, b, c, e = draw(d)
            mass.push(a)
            units.push(b)
            fraction.push(c)
            tail.push(e)
        if units.settled(min_draws, abs_tol, rel_tol):
            break
    return mass, units, fraction, tail


def _null_summary(center, scale, support, n_ref, moments, max_draws, bandwidth, method, positive_consensus):
    mass, units, fraction, tail = moments
    return {
        "null_center": center, "null_scale": scale, "null_threshold": float(center + (2.0 * scale)),
        "null_positive_mass": mass.mean, "null_positive_units": units.mean, "null_positive_fraction": fraction.mean,
        "null_positive_units_sd": units.sd(), "null_positive_units_stderr": units.stderr(),
        "null_tail_occupancy": tail.mean, "null_tail_occupancy_sd": tail.sd(), "null_tail_occupancy_stderr": tail.stderr(),
        "negative_support_size": support, "negative_fraction": float(support / max(int(n_ref), 1)),
        "num_null_draws": units.count, "max_null_draws": max_draws, "adaptive_stop": bool(units.count < max_draws),
        "wild_bandwidth": bandwidth, "null_method": method,
        "null_reference_mean_positive_consensus": float(np.mean(positive_consensus)),
        "null_reference_max_positive_consensus": float(np.max(positive_consensus)),
    }


# ---------------------------------------------------------------------------------------------------------------------
# the two estimators
# ---------------------------------------------------------------------------------------------------------------------

def estimate_budget_nonnull_fraction_from_score_track(score_track, dependence_lag_hint=None, num_null_draws: int = 25,
                                                      random_seed: int = 0, progress_label=None, num_processes: int = 1,
                                                      return_details: bool = False):
    """rocco/inference.py:1312-1421 over 1151-1309."""
    s = np.asarray(score_track, dtype=np.float64)
    if s.ndim != 1:
        raise ValueError("`score_track` must be one-dimensional")
    if s.size == 0:
        raise ValueError("`score_track` must contain at least one locus")
    positive = np.clip(s, 0.0, None)
    template = s - positive
    center, scale, support = _null_fit(template, "Direct-score budget")
    soft, threshold = float(max(scale, 1.0e-6)), float(center + (2.0 * scale))
    bandwidth = bootstrap_bandwidth(s.size, dependence_lag_hint)
    kernel = bartlett_kernel(bandwidth)
    max_draws = int(max(1, num_null_draws))
    min_draws = int(min(max_draws, max(4, 8)))
    rng = np.random.default_rng(int(random_seed))  # ONE stream for all draws here (inference.py:1206)

    def draw(_d):
        return _draw_statistics(template * dependent_wild_weights(s.size, kernel, rng), center, soft, threshold)

    moments = _run_draws(draw, max_draws, min_draws, 1, 5.0e-3, 5.0e-2)
    null = _null_summary(center, scale, support, s.size, moments, max_draws, bandwidth, "dependent_wild_score_bootstrap",
                         positive)
    fraction, details = _finish(s, int(s.size), dependence_lag_hint, null, "Direct-score budget")
    return (fraction, details) if return_details else fraction


def estimate_budget_nonnull_fraction_from_wild_bootstrap_null(centered_matrix, observed_scores=None, lower_bound_z: float = 1.0,
                                                              prior_df: float = 5.0, min_effect=None,
                                                              precision_floor_ratio: float = 0.01, dependence_lag_hint=None,
                                                              num_null_draws: int = 25, random_seed: int = 0,
                                                              progress_label=None, num_processes: int = 1,
                                                              return_details: bool = False):
    """rocco/inference.py:988-1148 over 719-985; the WLS scorings through the oracle's own backend restatement."""
    import pyoracle as po

    c = np.asarray(centered_matrix, dtype=np.float64)
    if c.ndim == 1:
        c = c[np.newaxis, :]
    if c.ndim != 2:
        raise ValueError("`centered_matrix` must be one- or two-dimensional")
    K, n = c.shape
    if n <= 0:
        raise ValueError("`centered_matrix` must contain at least one locus")
    floor_ratio = float(max(precision_floor_ratio, 0.0))
    template, fitted, positive = po.fit_budget_null_residual_template(c, lower_bound_z, prior_df, min_effect, floor_ratio)
    if observed_scores is None:
        observed = fitted
    else:
        observed = np.asarray(observed_scores, dtype=np.float64)
        if observed.shape[0] != n:
            raise ValueError("`observed_scores` must have the same number of loci as `centered_matrix`")
    reference = po.score_centered_wls(template, lower_bound_z=lower_bound_z, prior_df=prior_df, min_effect=min_effect,
                                      spatial_window=31, precision_floor_ratio=floor_ratio)[0]
    center, scale, support = _null_fit(np.asarray(reference, dtype=np.float64), "Budget")
    soft, threshold = float(max(scale, 1.0e-6)), float(center + (2.0 * scale))
    bandwidth = bootstrap_bandwidth(n, dependence_lag_hint)
    kernel = bartlett_kernel(bandwidth)
    max_draws = int(max(1, num_null_draws))
    min_draws = int(min(max_draws, max(4, 8)))
    look_every = int(max(1, min(max(1, num_processes), max_draws)))

    def draw(d):
        rng = np.random.default_rng(int(random_seed) + (104729 * (int(d) + 1)))  # a stream per draw (inference.py:654)
        weights = np.stack([dependent_wild_weights(n, kernel, rng) for _ in range(K)])
        return po.compute_budget_null_draw(template, weights, lower_bound_z, prior_df, min_effect, floor_ratio, center, soft,
                                           threshold)

    moments = _run_draws(draw, max_draws, min_draws, look_every, 5.0e-3, 5.0e-2)
    null = _null_summary(center, scale, support, reference.size, moments, max_draws, bandwidth,
                         "dependent_wild_residual_bootstrap", positive)
    fraction, details = _finish(observed, int(n), dependence_lag_hint, null, "Budget")
    return (fraction, details) if return_details else fraction


# ---------------------------------------------------------------------------------------------------------------------
# switch cost, pooled budgets
# ---------------------------------------------------------------------------------------------------------------------

def resolve_chrom_gamma(args: dict, scores, rate_meta: dict):
    """rocco/rocco.py:751-789 -> (gamma, metadata or None)."""
    if args["gamma"] is not None:
        g = float(args["gamma"])
        if not np.isfinite(g) or g < 0.0:
            raise ValueError("`--gamma` must be finite and non-negative")
        return g, None
    s = np.asarray(scores, dtype=np.float64)
    pos = s[s > 0.0]
    scale, count = (1.0, 0) if pos.size == 0 else (float(np.median(pos)), int(pos.size))
    tau = max(1.0, float(rate_meta.get("autocorrelation_time", 1.0)))
    run = int(np.ceil(tau))
    raw = 0.5 * float(run) * float(scale)
    g = float(np.clip(raw, 0.5, 10.0))
    return g, {"method": "auto_score_autocorr", "autocorrelation_time": float(tau), "characteristic_run_length": run,
               "positive_score_median": float(scale), "positive_score_count": count, "gamma_raw": float(raw),
               "gamma_clipped": g, "gamma_clip_min": 0.5, "gamma_clip_max": 10.0}


def _beta_prior(x: np.ndarray, t: np.ndarray, init_center: float, init_strength: float) -> Tuple[float, float]:
    """rocco/inference.py:1488-1562."""
    from scipy import optimize, special

    center0 = min(max(float(init_center), 1.0e-6), 1.0 - 1.0e-6)
    rates = x / np.maximum(t, 1.0)
    pooled = float(np.clip(np.sum(x) / max(np.sum(t), 1.0), 1.0e-6, 1.0 - 1.0e-6))
    seen = float(np.var(rates, ddof=1)) if rates.size > 1 else 0.0
    floor = float(pooled * (1.0 - pooled) * np.mean(1.0 / np.maximum(t, 1.0)))
    if seen <= floor + 1.0e-12:
        strength = float(max(1.0e12, 100.0 * np.max(t)))
        return pooled * strength, (1.0 - pooled) * strength

    def nll(theta):
        a, b = float(np.exp(theta[0])), float(np.exp(theta[1]))
        return float(-np.sum(special.betaln(x + a, t - x + b) - special.betaln(a, b)))

    start = np.log(np.array([center0 * float(init_strength), (1.0 - center0) * float(init_strength)], dtype=np.float64))
    fit = optimize.minimize(nll, start, method="L-BFGS-B")
    if not fit.success:
        return center0 * float(init_strength), (1.0 - center0) * float(init_strength)
    return float(np.exp(fit.x[0])), float(np.exp(fit.x[1]))


def estimate_empirical_bayes_budgets(counts: Dict[str, float], totals: Dict[str, float], min_budget: float = 1.0e-4,
                                     max_budget: float = 0.5, init_center: float = 0.05, init_strength: float = 10.0,
                                     posterior_quantile: float = 0.01):
    """rocco/inference.py:1593-1737 -> (budgets, metadata)."""
    from scipy import stats

    chroms = list(counts.keys())
    if chroms != list(totals.keys()):
        raise ValueError("`chrom_candidate_counts` and `chrom_total_counts` must share keys in the same order")
    x = np.array([counts[c] for c in chroms], dtype=np.float64)
    t = np.array([totals[c] for c in chroms], dtype=np.float64)
    rates = x / np.maximum(t, 1.0)
    pooled = float(np.clip(np.sum(x) / max(np.sum(t), 1.0), 1.0e-6, 1.0 - 1.0e-6))
    seen = float(np.var(rates, ddof=1)) if rates.size > 1 else 0.0
    floor = float(pooled * (1.0 - pooled) * np.mean(1.0 / np.maximum(t, 1.0)))
    q = float(posterior_quantile)
    if not (0.0 < q < 1.0):
        raise ValueError("`posterior_quantile` must lie strictly between 0 and 1")
    if len(chroms) <= 1:
        a, b = float(init_center) * float(init_strength), (1.0 - float(init_center)) * float(init_strength)
        method, wide, strength, disp, at_floor = ("single_chrom_default", float(init_center), float(init_strength),
                                                  float(1.0 / (1.0 + a + b)), False)
    elif len(chroms) <= 3:
        a, b = float(pooled) * float(init_strength), (1.0 - float(pooled)) * float(init_strength)
        strength = float(a + b)
        method, wide, disp, at_floor = "weak_pooled_prior", float(pooled), float(max(0.0, 1.0 / (1.0 + strength))), \
            bool(seen <= floor + 1.0e-12)
    else:
        a, b = _beta_prior(x, t, init_center, init_strength)
        strength = float(a + b)
        method, wide, disp, at_floor = "beta_binomial_mle", float(a / (a + b)), float(max(0.0, 1.0 / (1.0 + strength))), \
            bool(seen <= floor + 1.0e-12)
    qq = float(np.clip(q, 1.0e-6, 1.0 - 1.0e-6))
    budgets = {}
    for i, c in enumerate(chroms):
        pa, pb = float(max(1.0e-12, x[i] + a)), float(max(1.0e-12, (t[i] - x[i]) + b))
        budgets[c] = float(np.clip(float(stats.beta.ppf(qq, pa, pb)), min_budget, max_budget))
    meta = {"alpha": float(a), "beta": float(b), "genome_wide_budget": float(wide), "prior_strength": float(strength),
            "prior_dispersion": float(disp), "min_prior_dispersion": 0.0, "observed_raw_budget_var": float(seen),
            "theoretical_min_raw_budget_var": float(floor), "prior_dispersion_at_floor": bool(at_floor),
            "posterior_summary": "beta_quantile", "posterior_quantile": float(q), "prior_fit_method": method}
    return budgets, meta


def resolve_budgets(chrom_cache: dict, args: dict):
    """rocco/rocco.py:1113-1143."""
    budgets, meta = estimate_empirical_bayes_budgets({c: chrom_cache[c]["budget_count_hat"] for c in chrom_cache},
                                                     {c: chrom_cache[c]["total_count"] for c in chrom_cache},
                                                     posterior_quantile=args["budget_posterior_quantile"])
    rescale = 1.0
    if args["budget"] is not None and meta["genome_wide_budget"] > 0:
        rescale = float(args["budget"]) / meta["genome_wide_budget"]
    return {c: min(max(budgets[c] * rescale * float(args["scale_chrom_budgets"]), 0.005), 0.1) for c in budgets}, meta


# ---------------------------------------------------------------------------------------------------------------------
# the driver
# ---------------------------------------------------------------------------------------------------------------------

def parallel_process_count(item_count: int, thread_limit: int) -> int:
    """rocco/rocco.py:792-806."""
    cores = max(1, os.cpu_count() or 1) if int(thread_limit) <= 0 else max(1, int(thread_limit))
    if int(item_count) <= 1 or int(cores) <= 1 or "fork" not in multiprocessing.get_all_start_methods():
        return 1
    return int(min(int(item_count), int(cores), 4))


def build_chrom_cache(chroms: Sequence[str], inputs: Dict[str, tuple], args: dict, log_matrix_of=None) -> dict:
    """rocco/rocco.py:933-1110 for in-memory `inputs[chrom] = (intervals, matrix)`.  `log_matrix_of(counts)` replaces
    NumPy's log2(x + 1) of the count branch (tests hand in the correctly rounded one to separate that freedom)."""
    import pyoracle as po

    cache = {}
    low_memory = bool(args.get("low_memory", False))
    pool = 1 if low_memory else parallel_process_count(int(args["budget_null_draws"]), int(args["threads"]))
    pool = min(int(args["budget_null_draws"]), int(pool))
    for chrom in chroms:
        if chrom not in inputs:
            continue
        intervals, matrix = inputs[chrom]
        if not np.all(np.isfinite(matrix)):
            raise ValueError(f"{chrom} matrix contains non-finite values")
        if args["input_track_type"] == "bigwig":
            scores = np.asarray(po.score_central_tendency_chrom(matrix, method="quantile", quantile=0.50, power=1.0),
                                dtype=np.float64)
            if not np.all(np.isfinite(scores)):
                raise ValueError(f"{chrom} direct scores contain non-finite values")
            effect = scores
            fraction, rate_meta = estimate_budget_nonnull_fraction_from_score_track(
                scores, num_null_draws=args["budget_null_draws"], num_processes=pool, return_details=True)
        else:
            scores, det = po.score_loci_wls(matrix, lower_bound_z=args["score_lower_bound_z"], prior_df=args["score_prior_df"],
                                            min_effect=args.get("score_min_effect"),
                                            precision_floor_ratio=args["score_precision_floor_ratio"],
                                            log_matrix=None if log_matrix_of is None else log_matrix_of(matrix))
            if not np.all(np.isfinite(scores)):
                raise ValueError(f"{chrom} scores contain non-finite values")
            centered = np.asarray(det["centered_matrix"], dtype=np.float32 if low_memory else np.float64)
            effect = det["mean"]
            fraction, rate_meta = estimate_budget_nonnull_fraction_from_wild_bootstrap_null(
                centered, observed_scores=scores, lower_bound_z=args["score_lower_bound_z"], prior_df=args["score_prior_df"],
                min_effect=args.get("score_min_effect"), precision_floor_ratio=args["score_precision_floor_ratio"],
                dependence_lag_hint=max(25, int(det.get("local_baseline_window", 101))),
                num_null_draws=args["budget_null_draws"], num_processes=pool, return_details=True)
        if not np.isfinite(fraction):
            raise ValueError(f"{chrom} budget estimate is not finite")
        n = int(scores.shape[0])
        total = float(np.clip(rate_meta.get("effective_total_count", n), 1.0, n))
        gamma, gamma_meta = resolve_chrom_gamma(args, scores, rate_meta)
        cache[chrom] = {"intervals": intervals, "scores": scores, "effect_mean": np.asarray(effect, dtype=np.float64),
                        "gamma": gamma, "gamma_meta": gamma_meta,
                        "budget_count_hat": float(np.clip(fraction * total, 0.0, total)), "budget_fraction_hat": float(fraction),
                        "budget_rate_meta": rate_meta, "total_count": total, "num_loci": n}
    return cache


def run_chromosomes(chroms: Sequence[str], inputs: Dict[str, tuple], args: dict, log_matrix_of=None):
    """Cache -> budgets -> per-chromosome solve and records -> combined records (rocco/rocco.py:1269-1287).  Returns
    (cache, budgets, {chrom: (details, records)}, combined records)."""
    import pyoracle as po

    cache = build_chrom_cache(chroms, inputs, args, log_matrix_of=log_matrix_of)
    budgets, _ = resolve_budgets(cache, args)
    solved = {}
    for chrom, entry in cache.items():
        solution, _objective, details = po.solve_chrom_exact(entry["scores"], budget=budgets[chrom], gamma=entry["gamma"],
                                                             selection_penalty=args["selection_penalty"], return_details=True)
        solved[chrom] = (details, po.chrom_solution_records(chrom, entry["intervals"], solution, min_length_bp=args["min_length_bp"]))
    return cache, budgets, solved, po.combine_records([records for _d, records in solved.values()])
