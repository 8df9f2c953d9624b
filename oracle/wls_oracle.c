/*
 * oracle/wls_oracle.c -- CPU restatement of the centred-WLS scoring backend.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Follows rocco/native/wls_backend.c:
 *   232-260  odd spatial window (default 31, at least 5, at most n)
 *   610-742  rolling AR(1) innovation variance: three running sums updated by subtract-then-add
 *   394-608  monotone variance-vs-|value| trend: pairs sorted by (x, y), ~log2(n) equal-count bins,
 *            bin medians, pool-adjacent-violators (262-339), knots, linear interpolation (341-392)
 *   744-947  per row: variance track, trend, EB shrink, precision sums; then per locus mean / SE / score
 * Same IEEE operations in the same order wherever order matters (running sums, pooled means, row
 * accumulation); order statistics (medians) are taken from a full sort instead of the reference's
 * quick-select -- the values are the same.  Pinned against the reference's own file compiled in place
 * (oracle/_ref/libwls_ref.so) and golden vectors (tests/test_wls_oracle.py).
 * Inputs must be finite (the reference's Python caller guarantees it, rocco/inference.py:40-46).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double x, y;
} xy_pair;

static int cmp_xy(const void *a, const void *b)
{
    const xy_pair *l = (const xy_pair *)a, *r = (const xy_pair *)b;
    if (l->x < r->x) return -1;
    if (l->x > r->x) return 1;
    if (l->y < r->y) return -1;
    if (l->y > r->y) return 1;
    return 0;
}

static int cmp_d(const void *a, const void *b)
{
    const double l = *(const double *)a, r = *(const double *)b;
    return (l < r) ? -1 : ((l > r) ? 1 : 0);
}

/* wls_backend.c:121-145 on an ascending array */
static double median_sorted(const double *v, size_t n)
{
    if (n == 0) return 0.0;
    if (n == 1) return v[0];
    if (n & 1U) return v[n / 2];
    return 0.5 * (v[n / 2 - 1] + v[n / 2]);
}

size_t oracle_wls_spatial_window(size_t n, int requested)
{
    size_t w;
    if (n < 5) return 0;
    w = requested > 0 ? (size_t)requested : 31U;
    if (w < 5) w = 5;
    if (w > n) w = n;
    if ((w & 1U) == 0) w = (w == n) ? (w - 1) : (w + 1);
    return (w < 5) ? 0 : w;
}

/* wls_backend.c:610-742 */
int oracle_rolling_ar1_innovation_variance_f64(const double *v, size_t n, size_t window, double *out)
{
    if (v == NULL || out == NULL || n == 0) return -2;
    window = oracle_wls_spatial_window(n, (int)window);
    if (window == 0 || n < 4) {
        memset(out, 0, n * sizeof(double));
        return 0;
    }
    const size_t half = window / 2, max_start = n - window;
    double *at_start = (double *)malloc((max_start + 1) * sizeof(double));
    if (at_start == NULL) return -1;
    double sum_y = 0.0, sum_sq = 0.0, sum_lag = 0.0;
    for (size_t i = 0; i < window; ++i) {
        sum_y += v[i];
        sum_sq += v[i] * v[i];
        if (i < window - 1) sum_lag += v[i] * v[i + 1];
    }
    const double wd = (double)window, pairs = (double)(window - 1);
    for (size_t s = 0; s <= max_start; ++s) {
        const double leaving = v[s], entering = v[s + window - 1];
        const double sum_x_seq = sum_y - entering, sum_y_seq = sum_y - leaving;
        const double mean_all = sum_y / wd;
        double g0n = sum_sq - (wd * mean_all * mean_all);
        if (g0n < 0.0) g0n = 0.0;
        const double g1n = sum_lag - (mean_all * sum_x_seq) - (mean_all * sum_y_seq) + (pairs * mean_all * mean_all);
        const double lambda_eff = 1.0 / (wd + 1.0);
        const double scale_floor = 1.0e-4 * (g0n + 1.0);
        const double denom = (g0n * (1.0 + lambda_eff)) + scale_floor;
        const double eps = 1.0e-12 * (g0n + 1.0);
        double beta1 = 0.0;
        if (denom > eps) beta1 = g1n / denom;
        if (beta1 > 0.99) beta1 = 0.99;
        else if (beta1 < 0.0) beta1 = 0.0;
        const double gamma0 = g0n / wd;
        double omb = 1.0 - (beta1 * beta1);
        if (omb < 0.0) omb = 0.0;
        at_start[s] = fmax(gamma0 * omb, 0.0);
        if (s < max_start) {
            const double next = v[s + window], lag_left = v[s + window - 1], lag_right = v[s + 1];
            sum_y = (sum_y - leaving) + next;
            sum_sq = sum_sq - (leaving * leaving) + (next * next);
            sum_lag = sum_lag - (leaving * lag_right) + (lag_left * next);
        }
    }
    for (size_t i = 0; i < n; ++i) {
        size_t c = (i < half) ? 0 : (i - half);
        if (c > max_start) c = max_start;
        out[i] = at_start[c];
    }
    free(at_start);
    return 0;
}

/* wls_backend.c:262-339 */
static void pava(const double *values, const double *weights, size_t n, double *out)
{
    double *bv = (double *)malloc(n * sizeof(double)), *bw = (double *)malloc(n * sizeof(double));
    size_t *bl = (size_t *)malloc(n * sizeof(size_t));
    size_t nb = 0;
    for (size_t i = 0; i < n; ++i) {
        bv[nb] = values[i];
        bw[nb] = fmax(weights[i], 1.0e-8);
        bl[nb] = 1;
        ++nb;
        while (nb >= 2 && bv[nb - 2] > bv[nb - 1]) {
            const double tw = bw[nb - 2] + bw[nb - 1];
            const double mv = ((bv[nb - 2] * bw[nb - 2]) + (bv[nb - 1] * bw[nb - 1])) / tw;
            bv[nb - 2] = mv;
            bw[nb - 2] = tw;
            bl[nb - 2] += bl[nb - 1];
            --nb;
        }
    }
    size_t cur = 0;
    for (size_t b = 0; b < nb; ++b)
        for (size_t r = 0; r < bl[b]; ++r) out[cur++] = bv[b];
    free(bv);
    free(bw);
    free(bl);
}

/* wls_backend.c:341-392 */
static double interp(const double *xs, const double *ys, size_t n, double t)
{
    if (n == 0) return 1.0e-8;
    if (n == 1 || t <= xs[0]) return ys[0];
    if (t >= xs[n - 1]) return ys[n - 1];
    size_t left = 0, right = n - 1;
    while (right - left > 1) {
        const size_t mid = left + (right - left) / 2;
        if (xs[mid] <= t) left = mid;
        else right = mid;
    }
    if (xs[right] <= xs[left]) return fmax(ys[right], ys[left]);
    const double w = (t - xs[left]) / (xs[right] - xs[left]);
    return ys[left] + (w * (ys[right] - ys[left]));
}

/* wls_backend.c:394-608 (finite inputs) */
int oracle_monotone_variance_trend_f64(const double *cov, const double *raw_var, size_t n, double *trend)
{
    if (cov == NULL || raw_var == NULL || trend == NULL) return -2;
    xy_pair *pairs = (xy_pair *)malloc((n ? n : 1) * sizeof(xy_pair));
    double *work = (double *)malloc((n ? n : 1) * sizeof(double));
    if (pairs == NULL || work == NULL) {
        free(pairs);
        free(work);
        return -1;
    }
    for (size_t i = 0; i < n; ++i) {
        pairs[i].x = fabs(cov[i]);
        pairs[i].y = fmax(raw_var[i], 1.0e-8);
        work[i] = pairs[i].y;
    }
    double fallback = 1.0e-6;
    if (n > 0) {
        qsort(work, n, sizeof(double), cmp_d);
        fallback = fmax(median_sorted(work, n), 1.0e-8);
    }
    if (n < 4) {
        for (size_t i = 0; i < n; ++i) trend[i] = fallback;
        free(pairs);
        free(work);
        return 0;
    }
    qsort(pairs, n, sizeof(xy_pair), cmp_xy);
    const size_t bins = (size_t)fmax(4.0, floor(1.0 + (log((double)n + 1.0) / log(2.0))));
    double *bc = (double *)malloc(6 * bins * sizeof(double));
    double *bvar = bc + bins, *bw = bc + 2 * bins, *fit = bc + 3 * bins, *kc = bc + 4 * bins, *kv = bc + 5 * bins;
    size_t used = 0;
    for (size_t b = 0; b < bins; ++b) {
        const size_t left = (b * n) / bins, right = ((b + 1) * n) / bins;
        if (right <= left) continue;
        const size_t width = right - left;
        if (width & 1U) bc[used] = pairs[left + width / 2].x;
        else bc[used] = 0.5 * (pairs[left + width / 2 - 1].x + pairs[left + width / 2].x);
        for (size_t k = 0; k < width; ++k) work[k] = pairs[left + k].y;
        qsort(work, width, sizeof(double), cmp_d);
        bvar[used] = median_sorted(work, width);
        bw[used] = (double)width;
        ++used;
    }
    if (used == 0) {
        for (size_t i = 0; i < n; ++i) trend[i] = fallback;
    } else if (used == 1) {
        const double c = fmax(bvar[0], 1.0e-8);
        for (size_t i = 0; i < n; ++i) trend[i] = c;
    } else {
        pava(bvar, bw, used, fit);
        size_t knots = 0;
        for (size_t b = 0; b < used; ++b) {
            const double cv = bc[b], vv = fmax(fit[b], 1.0e-8);
            if (knots > 0 && cv <= kc[knots - 1]) {
                kv[knots - 1] = fmax(kv[knots - 1], vv);
                continue;
            }
            kc[knots] = cv;
            kv[knots] = vv;
            ++knots;
        }
        if (knots == 0) {
            for (size_t i = 0; i < n; ++i) trend[i] = fallback;
        } else if (knots == 1) {
            const double c = fmax(kv[0], 1.0e-8);
            for (size_t i = 0; i < n; ++i) trend[i] = c;
        } else {
            for (size_t i = 0; i < n; ++i) trend[i] = fmax(interp(kc, kv, knots, fabs(cov[i])), 1.0e-8);
        }
    }
    free(bc);
    free(pairs);
    free(work);
    return 0;
}

/* wls_backend.c:177-205 */
static double robust_scale(double *work, size_t n)
{
    if (n == 0) return 1.0e-6;
    qsort(work, n, sizeof(double), cmp_d);
    const double med = median_sorted(work, n);
    for (size_t i = 0; i < n; ++i) work[i] = fabs(work[i] - med);
    qsort(work, n, sizeof(double), cmp_d);
    double mad = median_sorted(work, n);
    mad *= 1.4826;
    if (!(mad > 1.0e-6)) return 1.0e-6;
    return mad;
}

/* wls_backend.c:744-947 */
int oracle_score_centered_wls_f64(const double *centered, size_t K, size_t n, double lower_bound_z, double prior_df,
                                  double min_effect, int use_min_effect, int spatial_window,
                                  double precision_floor_ratio, double *mean, double *raw_var, double *prior_var,
                                  double *mod_var, double *se, double *scores, double *df_out, int *window_out)
{
    if (centered == NULL || mean == NULL || raw_var == NULL || prior_var == NULL || mod_var == NULL || se == NULL ||
        scores == NULL || K == 0 || n == 0)
        return -2;
    const double pdf = fmax(prior_df, 0.0), floor_ratio = fmax(precision_floor_ratio, 0.0);
    const size_t window = oracle_wls_spatial_window(n, spatial_window);
    const double local_df = window > 0 ? fmax(4.0, (double)window - 3.0) : 1.0;
    const double total_df = local_df + pdf;
    if (df_out) *df_out = total_df;
    if (window_out) *window_out = (int)window;
    double *buf = (double *)calloc(7 * n, sizeof(double));
    if (buf == NULL) return -1;
    double *obs = buf, *prior = buf + n, *wsum = buf + 2 * n, *psum = buf + 3 * n, *rsum = buf + 4 * n,
           *qsum = buf + 5 * n, *work = buf + 6 * n;
    for (size_t k = 0; k < K; ++k) {
        const double *row = centered + k * n;
        if (window == 0 || n < 4) {
            memcpy(work, row, n * sizeof(double));
            double sf = robust_scale(work, n);
            sf = fmax(sf * sf, 1.0e-8);
            for (size_t i = 0; i < n; ++i) obs[i] = prior[i] = sf;
        } else {
            int rc = oracle_rolling_ar1_innovation_variance_f64(row, n, window, obs);
            if (rc != 0) {
                free(buf);
                return rc == -1 ? -1 : -2;
            }
            for (size_t i = 0; i < n; ++i) obs[i] = fmax(obs[i], 1.0e-8);
            rc = oracle_monotone_variance_trend_f64(row, obs, n, prior);
            if (rc != 0) {
                free(buf);
                return rc == -1 ? -1 : -2;
            }
        }
        for (size_t i = 0; i < n; ++i) {
            const double ov = fmax(obs[i], 1.0e-8), pv = fmax(prior[i], 1.0e-8);
            double post = ((local_df * ov) + (pdf * pv)) / fmax(total_df, 1.0);
            const double vfloor = floor_ratio * pv;
            if (post < vfloor) post = vfloor;
            post = fmax(post, 1.0e-8);
            const double prec = 1.0 / post;
            rsum[i] += 1.0 / ov;
            qsum[i] += 1.0 / pv;
            psum[i] += prec;
            wsum[i] += prec * row[i];
        }
    }
    for (size_t i = 0; i < n; ++i) {
        const double lp = fmax(psum[i], 1.0e-8);
        mean[i] = wsum[i] / lp;
        raw_var[i] = (double)K / fmax(rsum[i], 1.0e-8);
        prior_var[i] = (double)K / fmax(qsum[i], 1.0e-8);
        mod_var[i] = (double)K / lp;
        se[i] = sqrt(1.0 / lp);
        const double z = mean[i] / fmax(se[i], 1.0e-8);
        if (use_min_effect != 0) scores[i] = (mean[i] - fmax(min_effect, 0.0)) / fmax(se[i], 1.0e-8);
        else scores[i] = z - lower_bound_z;
    }
    free(buf);
    return 0;
}
