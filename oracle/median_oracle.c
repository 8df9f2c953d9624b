/*
 * oracle/median_oracle.c -- column-wise median over K samples.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restates what rocco/rocco.py:264-265 obtains from np.median(chrom_matrix, axis=0) on the
 * bigWig path (call-site rocco/rocco.py:983-991): the middle order statistic for odd K, the mean
 * of the two middle order statistics ((a + b) / 2) for even K, NaN if the column holds a NaN.
 * NumPy 2.2 is the arithmetic being restated (third-party, not vendored in the reference;
 * version unpinned there, setup.py:387-392); tests pin this file against np.median directly.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

static int cmp_double(const void *pa, const void *pb)
{
    const double a = *(const double *)pa;
    const double b = *(const double *)pb;
    return (a > b) - (a < b);
}

int oracle_median_columns(const void *matrix, int is_f32, size_t K, size_t n, double *scores_out)
{
    if (matrix == NULL || scores_out == NULL || K == 0) {
        return -2;
    }
    double *col = (double *)malloc(K * sizeof(double));
    if (col == NULL) {
        return -1;
    }
    const float *mf = (const float *)matrix;
    const double *md = (const double *)matrix;
    for (size_t j = 0; j < n; ++j) {
        int has_nan = 0;
        for (size_t k = 0; k < K; ++k) {
            const double v = is_f32 ? (double)mf[k * n + j] : md[k * n + j];
            if (v != v) {
                has_nan = 1;
            }
            col[k] = v;
        }
        if (has_nan) {
            scores_out[j] = NAN;
            continue;
        }
        qsort(col, K, sizeof(double), cmp_double);
        if (K & 1U) {
            scores_out[j] = col[K / 2];
        } else {
            scores_out[j] = (col[K / 2 - 1] + col[K / 2]) / 2.0;
        }
    }
    free(col);
    return 0;
}
