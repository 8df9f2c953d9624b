/*
 * oracle/delta_oracle.c -- sequential definition of the delta-form ("fast path") evaluation.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference's forward pass (rocco/_chain_dp.c:115-165) keeps two path values prev0/prev1.
 * In exact arithmetic only their difference matters:
 *      delta_0 = a_0,   delta_i = clamp(delta_{i-1}, -c_{i-1}, +c_{i-1}) + a_i,   a_i = s_i - lambda
 * with  bt0[i] = [delta_{i-1} > c], bt1[i] = [delta_{i-1} >= -c]  (rocco/_chain_dp.c:133-159) and
 * terminal state [delta_{n-1} > 0] (rocco/_chain_dp.c:167-179), so the backtrack
 * (rocco/_chain_dp.c:181-186) is a backward fill over per-locus classes ONE / ZERO / COPY.
 *
 * SPEC (shared with rocco_amd/csrc/chain_fast.hip):
 *  - grid: a_i and c_i are rounded to multiples of q = 2^qexp by rn(x) = (x + M) - M,
 *    M = 1.5 * 2^(52+qexp); every later add/min/max is then exact in double precision, the
 *    recursion is exactly associative, and any parallel decomposition gives identical bits.
 *  - noise model: P16 = sum floor(16 * max(a_i, 0)) and npos = #{a_i > 0} (exact integers);
 *    Pb = 2 * ((P16 + npos) / 16 + cmax + sabs + |lambda| + 1) bounds every intermediate of the
 *    reference's pass; h = 2^(ilogb(Pb) - 53); tau_step = 4h + q; tau0 = 9h + 2q.
 *    guard = ORACLE_GUARD = 2^-16 (constant): a clamp is "clear" when |delta| - c > guard.
 *  - chunks of ORACLE_CHUNK = 32 loci: inside a chunk two extreme chains start from the clamp
 *    bounds (+c, -c); a locus is "known" once they agree; a known locus with |delta| - c > guard is a
 *    provable clear clamp; m_j = j - 1 - (last provable clear clamp before j).
 *  - tau_j = tau0 + tau_step * m_j; locus j is certain iff tau_j <= guard and ||delta_j| - c_j| > tau_j
 *    (tau_j > guard = "overflow": the tolerance model no longer covers that locus).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

enum { CLS_ZERO = 0, CLS_COPY = 1, CLS_ONE = 2 };

static inline double grid_round(double x, double magic)
{
    volatile double t = x + magic; /* volatile: keep the two roundings exactly as written */
    return t - magic;
}

static inline double clampd(double x, double c) { return fmin(fmax(x, -c), c); }

static void fill_backward(const uint8_t *cls, size_t n, uint8_t *solution, long long *count)
{
    uint8_t state = 0; /* cls[n-1] is never COPY (terminal rule) */
    long long total = 0;
    for (size_t j = n; j-- > 0;) {
        if (cls[j] != CLS_COPY) {
            state = (uint8_t)(cls[j] == CLS_ONE);
        }
        if (solution != NULL) {
            solution[j] = state;
        }
        total += state;
    }
    *count = total;
}

void oracle_noise_model(const double *scores, size_t n, double lambda, int qexp, double cmax,
                        double sabs, oracle_noise *out)
{
    const double magic = ldexp(1.5, 52 + qexp);
    long long p16 = 0, npos = 0;
    for (size_t j = 0; j < n; ++j) {
        const double a = grid_round(scores[j] - lambda, magic);
        if (a > 0.0) {
            p16 += (long long)(16.0 * a);
            ++npos;
        }
    }
    const double pb = 2.0 * ((double)(p16 + npos) * 0.0625 + cmax + sabs + fabs(lambda) + 1.0);
    const double h = ldexp(1.0, ilogb(pb) - 53);
    const double q = ldexp(1.0, qexp);
    out->p16 = p16;
    out->npos = npos;
    out->tau_step = 4.0 * h + q;
    out->tau0 = 9.0 * h + 2.0 * q;
    out->guard = ORACLE_GUARD;
}

int oracle_delta_chain_f64(const double *scores, const double *switch_costs, double gamma, size_t n,
                           double selection_penalty, int qexp, double cmax, double sabs,
                           uint8_t *solution, oracle_delta_stats *stats)
{
    if (scores == NULL || n == 0 || stats == NULL) {
        return -2;
    }
    uint8_t *cls = (uint8_t *)malloc(n);
    if (cls == NULL) {
        return -1;
    }
    oracle_noise nz;
    oracle_noise_model(scores, n, selection_penalty, qexp, cmax, sabs, &nz);
    const double magic = ldexp(1.5, 52 + qexp);
    const double lam = selection_penalty;
    const double gq = grid_round(gamma, magic);
    double delta = 0.0, up = 0.0, dn = 0.0;
    long long last_clear = -1;
    long long uncertain = 0, effect = 0, max_run = 0;
    int overflow = 0;

    for (size_t j = 0; j < n; ++j) {
        const double a = grid_round(scores[j] - lam, magic);
        if (j == 0) {
            delta = up = dn = a;
        } else {
            const double c_prev = (switch_costs != NULL) ? grid_round(switch_costs[j - 1], magic) : gq;
            delta = clampd(delta, c_prev) + a;
            if (j % ORACLE_CHUNK == 0) { /* extremes restart at every chunk boundary */
                up = c_prev + a;
                dn = -c_prev + a;
            } else {
                up = clampd(up, c_prev) + a;
                dn = clampd(dn, c_prev) + a;
            }
        }
        const long long m = (long long)j - 1 - last_clear;
        if (m > max_run) {
            max_run = m;
        }
        const double tau = nz.tau0 + nz.tau_step * (double)m;
        int certain;
        uint8_t k;
        if (j + 1 < n) {
            const double cj = (switch_costs != NULL) ? grid_round(switch_costs[j], magic) : gq;
            const double e = fabs(delta) - cj;
            certain = (tau <= nz.guard) && (e > tau || e < -tau);
            k = (delta > cj) ? CLS_ONE : ((delta < -cj) ? CLS_ZERO : CLS_COPY);
            if (up == dn && e > nz.guard) {
                last_clear = (long long)j;
            }
        } else {
            const double e = fabs(delta);
            certain = (tau <= nz.guard) && (e > tau);
            k = (delta > 0.0) ? CLS_ONE : CLS_ZERO;
        }
        if (tau > nz.guard) {
            overflow = 1;
        }
        if (!certain) {
            ++uncertain;
            effect += m + 1;
        }
        cls[j] = k;
    }
    fill_backward(cls, n, solution, &stats->count);
    stats->uncertain = uncertain;
    stats->effect = effect;
    stats->max_run = max_run;
    stats->overflow = overflow;
    free(cls);
    return 0;
}

int oracle_delta_window_f64(const double *scores, const double *switch_costs, double gamma, size_t n,
                            double lambda_lo, double lambda_hi, int qexp, double cmax, double sabs,
                            uint8_t *solution, oracle_window_stats *stats,
                            oracle_window_diff *diffs, int diff_capacity)
{
    if (scores == NULL || n == 0 || stats == NULL || !(lambda_lo <= lambda_hi)) {
        return -2;
    }
    uint8_t *lo_cls = (uint8_t *)malloc(n);
    uint8_t *hi_cls = (uint8_t *)malloc(n);
    if (lo_cls == NULL || hi_cls == NULL) {
        free(lo_cls);
        free(hi_cls);
        return -1;
    }
    oracle_noise nz;
    oracle_noise_model(scores, n, lambda_lo, qexp, cmax, sabs, &nz);
    const double magic = ldexp(1.5, 52 + qexp);
    const double gq = grid_round(gamma, magic);
    /* "lo" chain uses lambda_lo (larger delta), "hi" chain uses lambda_hi (smaller delta) */
    double d_lo = 0.0, d_hi = 0.0, up_lo = 0.0, dn_lo = 0.0, up_hi = 0.0, dn_hi = 0.0;
    long long last_clear = -1, max_run = 0, n_diff = 0;
    int adjacent = 1, overflow = 0;

    for (size_t j = 0; j < n; ++j) {
        const double a_lo = grid_round(scores[j] - lambda_lo, magic);
        const double a_hi = grid_round(scores[j] - lambda_hi, magic);
        if (j == 0) {
            d_lo = up_lo = dn_lo = a_lo;
            d_hi = up_hi = dn_hi = a_hi;
        } else {
            const double c_prev = (switch_costs != NULL) ? grid_round(switch_costs[j - 1], magic) : gq;
            d_lo = clampd(d_lo, c_prev) + a_lo;
            d_hi = clampd(d_hi, c_prev) + a_hi;
            if (j % ORACLE_CHUNK == 0) {
                up_lo = c_prev + a_lo;
                dn_lo = -c_prev + a_lo;
                up_hi = c_prev + a_hi;
                dn_hi = -c_prev + a_hi;
            } else {
                up_lo = clampd(up_lo, c_prev) + a_lo;
                dn_lo = clampd(dn_lo, c_prev) + a_lo;
                up_hi = clampd(up_hi, c_prev) + a_hi;
                dn_hi = clampd(dn_hi, c_prev) + a_hi;
            }
        }
        const long long m = (long long)j - 1 - last_clear;
        if (m > max_run) {
            max_run = m;
        }
        const double tau = nz.tau0 + nz.tau_step * (double)m;
        uint8_t lo, hi;
        double cj = 0.0;
        if (j + 1 < n) {
            cj = (switch_costs != NULL) ? grid_round(switch_costs[j], magic) : gq;
            /* lowest class the reference could take anywhere in the zone */
            if (d_hi + cj < tau) {
                lo = CLS_ZERO;
            } else if (d_hi - cj > tau) {
                lo = CLS_ONE;
            } else {
                lo = CLS_COPY;
            }
            /* highest class */
            if (d_lo - cj > -tau) {
                hi = CLS_ONE;
            } else if (d_lo + cj < -tau) {
                hi = CLS_ZERO;
            } else {
                hi = CLS_COPY;
            }
            /* clear for every lambda in the zone: even the smallest delta is far above +c, or even
             * the largest delta is far below -c; both chains must be exactly known here */
            if (up_lo == dn_lo && up_hi == dn_hi && (d_hi - cj > nz.guard || -d_lo - cj > nz.guard)) {
                last_clear = (long long)j;
            }
        } else {
            lo = (d_hi > tau) ? CLS_ONE : CLS_ZERO;
            hi = (d_lo > -tau) ? CLS_ONE : CLS_ZERO;
        }
        if (tau > nz.guard) { /* tolerance model no longer valid */
            overflow = 1;
        }
        if (lo != hi) {
            /* decision boundary between the two classes: -c (ZERO|COPY), +c (COPY|ONE), 0 (terminal) */
            const double bound = (j + 1 < n) ? ((hi == CLS_ONE) ? cj : -cj) : 0.0;
            if (diffs != NULL && n_diff < diff_capacity) {
                diffs[n_diff].locus = (long long)j;
                diffs[n_diff].margin_lo = d_lo - bound;
                diffs[n_diff].margin_hi = d_hi - bound;
                diffs[n_diff].run = m;
                diffs[n_diff].cls_lo = lo;
                diffs[n_diff].cls_hi = hi;
            }
            ++n_diff;
            if (j + 1 < n && (int)hi - (int)lo != 1) {
                adjacent = 0;
            }
        }
        lo_cls[j] = lo;
        hi_cls[j] = hi;
    }
    fill_backward(hi_cls, n, NULL, &stats->count_hi);
    fill_backward(lo_cls, n, solution, &stats->count_lo);
    stats->n_diff = n_diff;
    stats->diff_adjacent = adjacent;
    stats->overflow = overflow;
    stats->max_run = max_run;
    free(lo_cls);
    free(hi_cls);
    return 0;
}
