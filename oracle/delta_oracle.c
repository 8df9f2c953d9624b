/*
 * oracle/delta_oracle.c -- sequential definition of the delta-form recursion.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference's forward pass (rocco/_chain_dp.c:115-165) keeps two path values prev0/prev1.
 * In exact arithmetic only their difference matters:
 *      delta_0 = s_0 - lambda,   delta_i = clamp(delta_{i-1}, -c_{i-1}, +c_{i-1}) + (s_i - lambda)
 * with  bt0[i] = [delta_{i-1} > c], bt1[i] = [delta_{i-1} >= -c]  (rocco/_chain_dp.c:133-159) and
 * terminal state [delta_{n-1} > 0] (rocco/_chain_dp.c:167-179), so the backtrack
 * (rocco/_chain_dp.c:181-186) is a backward fill over per-locus classes ONE / ZERO / COPY.
 *
 * This file is the *sequential* statement of that recursion, including the certification
 * bookkeeping (run length m since the last clear clamp, tolerance tau = tau0 + tau_step * m).
 * The HIP kernels evaluate the same recursion in parallel and must agree with this file bit for
 * bit (counts, classes, flags); the relation to the reference itself is by certification
 *  and is checked against chain_oracle.c / oracle/_ref in tests/.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

enum { CLS_ZERO = 0, CLS_COPY = 1, CLS_ONE = 2 };

static void fill_backward(const uint8_t *cls, size_t n, uint8_t *solution, long long *count)
{
    /* cls[n-1] is never COPY (terminal rule). */
    uint8_t state = 0;
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

int oracle_delta_chain_f64(const double *scores, const double *switch_costs, double gamma, size_t n,
                           double selection_penalty, double tau0, double tau_step, double guard,
                           int m_cap, uint8_t *solution, oracle_delta_stats *stats)
{
    if (scores == NULL || n == 0 || stats == NULL) {
        return -2;
    }
    uint8_t *cls = (uint8_t *)malloc(n);
    if (cls == NULL) {
        return -1;
    }
    const double lam = selection_penalty;
    double delta = 0.0;
    double e_prev = 0.0;
    long long m = 0;
    long long uncertain = 0, effect = 0, max_run = 0;

    for (size_t j = 0; j < n; ++j) {
        const double a = scores[j] - lam;
        if (j == 0) {
            delta = a;
            m = 0;
        } else {
            const double c_prev = (switch_costs != NULL) ? switch_costs[j - 1] : gamma;
            m = (e_prev > guard) ? 0 : m + 1;
            delta = fmin(fmax(delta, -c_prev), c_prev) + a;
        }
        if (m > max_run) {
            max_run = m;
        }
        const double tau = tau0 + tau_step * (double)m;
        int certain;
        uint8_t k;
        if (j + 1 < n) {
            const double cj = (switch_costs != NULL) ? switch_costs[j] : gamma;
            const double e = fabs(delta) - cj;
            certain = (m <= m_cap) && (e > tau || e < -tau);
            k = (delta > cj) ? CLS_ONE : ((delta < -cj) ? CLS_ZERO : CLS_COPY);
            e_prev = e;
        } else {
            const double e = fabs(delta);
            certain = (m <= m_cap) && (e > tau);
            k = (delta > 0.0) ? CLS_ONE : CLS_ZERO;
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
    free(cls);
    return 0;
}

int oracle_delta_window_f64(const double *scores, const double *switch_costs, double gamma, size_t n,
                            double lambda_lo, double lambda_hi, double tau0, double tau_step,
                            double guard, int m_cap, uint8_t *solution, oracle_window_stats *stats)
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
    double d_lo = 0.0, d_hi = 0.0; /* delta at lambda_lo (larger) and at lambda_hi (smaller) */
    double clear_one_prev = 0.0, clear_zero_prev = 0.0;
    long long m = 0, max_run = 0, n_diff = 0, first_diff = -1;
    int adjacent = 1;

    for (size_t j = 0; j < n; ++j) {
        const double a_lo = scores[j] - lambda_lo;
        const double a_hi = scores[j] - lambda_hi;
        if (j == 0) {
            d_lo = a_lo;
            d_hi = a_hi;
            m = 0;
        } else {
            const double c_prev = (switch_costs != NULL) ? switch_costs[j - 1] : gamma;
            /* clear for every lambda in the zone: even the smallest delta is far above +c, or even
             * the largest delta is far below -c */
            m = (clear_one_prev > guard || clear_zero_prev > guard) ? 0 : m + 1;
            d_lo = fmin(fmax(d_lo, -c_prev), c_prev) + a_lo;
            d_hi = fmin(fmax(d_hi, -c_prev), c_prev) + a_hi;
        }
        if (m > max_run) {
            max_run = m;
        }
        const double tau = tau0 + tau_step * (double)m;
        uint8_t lo, hi;
        if (j + 1 < n) {
            const double cj = (switch_costs != NULL) ? switch_costs[j] : gamma;
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
            clear_one_prev = d_hi - cj;
            clear_zero_prev = -d_lo - cj;
        } else {
            lo = (d_hi > tau) ? CLS_ONE : CLS_ZERO;
            hi = (d_lo > -tau) ? CLS_ONE : CLS_ZERO;
        }
        if (m > m_cap) { /* tolerance model no longer valid: force a (non-adjacent) difference */
            lo = CLS_ZERO;
            hi = CLS_ONE;
        }
        if (lo != hi) {
            if (n_diff == 0) {
                first_diff = (long long)j;
            }
            ++n_diff;
            if ((int)hi - (int)lo != 1) {
                adjacent = 0;
            }
        }
        lo_cls[j] = lo;
        hi_cls[j] = hi;
    }
    /* terminal entries are never COPY by construction */
    fill_backward(hi_cls, n, NULL, &stats->count_hi);
    fill_backward(lo_cls, n, solution, &stats->count_lo);
    stats->n_diff = n_diff;
    stats->first_diff = first_diff;
    stats->diff_adjacent = adjacent;
    stats->max_run = max_run;
    free(lo_cls);
    free(hi_cls);
    return 0;
}
