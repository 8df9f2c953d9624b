/*
 * oracle/baseline_oracle.c -- CPU restatement of the cross-fit Whittaker baseline.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Follows rocco/native/baseline_backend.c:
 *   79-173   LDL^T factorisation of the symmetric pentadiagonal system + forward / diagonal / backward solves
 *   175-250  bands of  W + lambda D^T D  (second differences) with a parity mask in W, right-hand side W y
 *   252-303  even fit, odd fit, average;  fewer than 25 values -> zeros
 *   305-334  row loop
 * Same IEEE operations in the same order; organised differently: the factor (d, l1, l2) does not depend
 * on the data, so it is computed once per (length, penalty, parity) and every row is solved against it
 * -- which is also how the device path works (rocco_amd/csrc/whittaker.hip).
 * Pinned against the reference's own file compiled in place (oracle/_ref/libbaseline_ref.so) and the
 * golden vectors (tests/test_oracle_golden.py).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* diagonal / first / second off-diagonal of W + lambda D^T D at row i (baseline_backend.c:198-232) */
static double band_a0(size_t i, size_t n, int parity, double lambda)
{
    const double w = ((i & 1U) == (size_t)parity) ? 1.0 : 0.0;
    if (i == 0 || i == n - 1) {
        return w + lambda;
    }
    if (i == 1 || i == n - 2) {
        return w + (5.0 * lambda);
    }
    return w + (6.0 * lambda);
}

static double band_a1(size_t i, size_t n, double lambda)
{
    return (i == 0 || i == n - 2) ? (-2.0 * lambda) : (-4.0 * lambda);
}

/* baseline_backend.c:105-140: d[n], l1[n-1], l2[n-2] */
void oracle_whittaker_factor_f64(size_t n, int parity, double lambda, double *d, double *l1, double *l2)
{
    if (n < 3) {
        return;
    }
    d[0] = band_a0(0, n, parity, lambda);
    l1[0] = band_a1(0, n, lambda) / d[0];
    l2[0] = lambda / d[0];
    d[1] = band_a0(1, n, parity, lambda) - ((l1[0] * l1[0]) * d[0]);
    {
        const double t1 = ((l2[0] * d[0]) * l1[0]);
        l1[1] = (band_a1(1, n, lambda) - t1) / d[1];
    }
    if (n > 3) {
        l2[1] = lambda / d[1];
    }
    for (size_t i = 2; i < n; ++i) {
        double t1 = ((l1[i - 1] * l1[i - 1]) * d[i - 1]);
        const double t2 = ((l2[i - 2] * l2[i - 2]) * d[i - 2]);
        d[i] = band_a0(i, n, parity, lambda) - t1 - t2;
        if (i + 2 <= n) {
            t1 = ((l2[i - 1] * d[i - 1]) * l1[i - 1]);
            l1[i] = (band_a1(i, n, lambda) - t1) / d[i];
        }
        if (i + 3 <= n) {
            l2[i] = lambda / d[i];
        }
    }
}

/* one row against a factor: rhs = W y (baseline_backend.c:200-216), then 142-172 */
static void solve_row(const double *y, size_t n, int parity, const double *d, const double *l1, const double *l2,
                      double *work, double *x)
{
    /* right-hand side: the end entries are selected, the interior ones multiplied by the 0/1 weight */
    for (size_t i = 0; i < n; ++i) {
        const int mine = ((i & 1U) == (size_t)parity);
        if (i < 2 || i + 2 >= n) {
            work[i] = mine ? y[i] : 0.0;
        } else {
            work[i] = (mine ? 1.0 : 0.0) * y[i];
        }
    }
    /* forward: L f = rhs (in place) */
    work[1] = work[1] - (l1[0] * work[0]);
    for (size_t i = 2; i < n; ++i) {
        const double t1 = l1[i - 1] * work[i - 1];
        const double t2 = l2[i - 2] * work[i - 2];
        work[i] = work[i] - t1 - t2;
    }
    /* diagonal */
    for (size_t i = 0; i < n; ++i) {
        work[i] = work[i] / d[i];
    }
    /* backward: L^T x = z */
    x[n - 1] = work[n - 1];
    x[n - 2] = work[n - 2] - (l1[n - 2] * x[n - 1]);
    for (size_t i = n - 2; i-- > 0;) {
        const double t1 = l1[i] * x[i + 1];
        const double t2 = l2[i] * x[i + 2];
        x[i] = work[i] - t1 - t2;
    }
}

int oracle_crossfit_whittaker_baseline_matrix_f64(const double *matrix, size_t rows, size_t cols, double lambda,
                                                  double *out)
{
    if (matrix == NULL || out == NULL) {
        return -1;
    }
    if (cols < 25) { /* baseline_backend.c:265-272 */
        for (size_t i = 0; i < rows * cols; ++i) {
            out[i] = 0.0;
        }
        return 0;
    }
    const size_t n = cols;
    double *buf = (double *)malloc((size_t)(8 * n) * sizeof(double));
    if (buf == NULL) {
        return -1;
    }
    double *d[2] = {buf, buf + n}, *l1[2] = {buf + 2 * n, buf + 3 * n}, *l2[2] = {buf + 4 * n, buf + 5 * n};
    double *work = buf + 6 * n, *odd = buf + 7 * n;
    for (int parity = 0; parity < 2; ++parity) {
        oracle_whittaker_factor_f64(n, parity, lambda, d[parity], l1[parity], l2[parity]);
    }
    for (size_t r = 0; r < rows; ++r) {
        const double *y = matrix + r * n;
        double *b = out + r * n;
        solve_row(y, n, 0, d[0], l1[0], l2[0], work, b);
        solve_row(y, n, 1, d[1], l1[1], l2[1], work, odd);
        for (size_t i = 0; i < n; ++i) { /* baseline_backend.c:296-299 */
            b[i] = 0.5 * (b[i] + odd[i]);
        }
    }
    free(buf);
    return 0;
}
