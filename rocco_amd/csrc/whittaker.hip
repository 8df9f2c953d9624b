// rocco_amd/csrc/whittaker.hip -- cross-fit Whittaker baseline of every row of a K x n matrix, gfx950.
//
// Replaces rocco/native/baseline_backend.c:305-334 (rocco_crossfit_whittaker_baseline_matrix_f64; row
// kernel 252-303, band setup 175-250, LDL^T solve 79-173), called from rocco/inference.py:185-209.
//
// Per row the reference solves  (W_p + lambda D^T D) b = W_p y  for the two parity masks p and averages
// the two fits.  Results must equal the reference bit for bit, so every recurrence runs in the
// reference's own order (linear recurrences with rounding are not associative).  What can be shared and
// what can run side by side:
//   * the LDL^T factor (d, l1, l2) does not depend on the data: it is computed ONCE per (n, lambda,
//     parity) -- the reference recomputes it for every row -- by one lane per parity (factor kernel);
//   * the rows are independent: forward and backward substitution run with one lane per row, both
//     parities in the same lane (two independent dependent-chains interleave), 64 rows per wavefront;
//     the row-major matrix is moved through LDS in 64 x 64 tiles so that global accesses are coalesced
//     (a lane reading its own row directly would touch 64 cache lines per instruction).
// The chains are latency-bound by construction (one multiply-subtract-subtract per locus and parity);
// the work per locus is 5 loads/stores of 8 bytes per row.
#include "kernels.h"

namespace rocco {

namespace {

constexpr int kTile = 64;          // loci per tile and rows per wavefront
constexpr int kStride = kTile + 1;  // LDS row stride (odd: lanes reading a column hit distinct banks)

__device__ __forceinline__ double band_a0(long long i, long long n, int parity, double lambda)
{
    // baseline_backend.c:198-216
    const double w = ((i & 1LL) == (long long)parity) ? 1.0 : 0.0;
    if (i == 0 || i == n - 1) {
        return w + lambda;
    }
    if (i == 1 || i == n - 2) {
        return w + (5.0 * lambda);
    }
    return w + (6.0 * lambda);
}

__device__ __forceinline__ double band_a1(long long i, long long n, double lambda)
{
    // baseline_backend.c:219-224
    return (i == 0 || i == n - 2) ? (-2.0 * lambda) : (-4.0 * lambda);
}

// factor[p] = d | l1 | l2, each n doubles (l1[n-1], l2[n-2], l2[n-1] unused)
__global__ __launch_bounds__(64) void whittaker_factor_kernel(long long n, double lambda, double *factor)
{
    const int parity = threadIdx.x;
    if (parity >= 2) {
        return;
    }
    double *__restrict__ d = factor + (long long)parity * 3 * n;
    double *__restrict__ l1 = d + n;
    double *__restrict__ l2 = l1 + n;
    // baseline_backend.c:105-121
    double d_m2 = band_a0(0, n, parity, lambda);
    double l1_m2 = band_a1(0, n, lambda) / d_m2;
    double l2_m2 = lambda / d_m2;
    d[0] = d_m2;
    l1[0] = l1_m2;
    l2[0] = l2_m2;
    double d_m1 = band_a0(1, n, parity, lambda) - ((l1_m2 * l1_m2) * d_m2);
    double l1_m1 = (band_a1(1, n, lambda) - ((l2_m2 * d_m2) * l1_m2)) / d_m1;
    double l2_m1 = (n > 3) ? (lambda / d_m1) : 0.0;
    d[1] = d_m1;
    l1[1] = l1_m1;
    l2[1] = l2_m1;
    // baseline_backend.c:123-140
    for (long long i = 2; i < n; ++i) {
        double t1 = ((l1_m1 * l1_m1) * d_m1);
        const double t2 = ((l2_m2 * l2_m2) * d_m2);
        const double di = band_a0(i, n, parity, lambda) - t1 - t2;
        double l1i = 0.0, l2i = 0.0;
        if (i <= n - 2) {
            t1 = ((l2_m1 * d_m1) * l1_m1);
            l1i = (band_a1(i, n, lambda) - t1) / di;
        }
        if (i <= n - 3) {
            l2i = lambda / di;
        }
        d[i] = di;
        l1[i] = l1i;
        l2[i] = l2i;
        d_m2 = d_m1;
        l2_m2 = l2_m1;
        d_m1 = di;
        l1_m1 = l1i;
        l2_m1 = l2i;
    }
}

struct Factor {
    const double *d, *l1, *l2;
};

__device__ __forceinline__ Factor factor_of(const double *factor, long long n, int parity)
{
    Factor f;
    f.d = factor + (long long)parity * 3 * n;
    f.l1 = f.d + n;
    f.l2 = f.l1 + n;
    return f;
}

// rhs = W_p y (baseline_backend.c:200-216): the two entries at either end are selected, the others are
// multiplied by the 0/1 weight (kept as a product: it decides the sign of a zero)
__device__ __forceinline__ double rhs_value(double y, long long i, long long n, int parity)
{
    const bool mine = ((i & 1LL) == (long long)parity);
    if (i < 2 || i + 2 >= n) {
        return mine ? y : 0.0;
    }
    return (mine ? 1.0 : 0.0) * y;
}

// forward substitution L f = rhs and the diagonal solve z = f / d, both parities
// (baseline_backend.c:142-156): z of parity 0 -> z0 (the output buffer), parity 1 -> z1 (scratch)
__global__ __launch_bounds__(kTile) void whittaker_forward_kernel(const double *__restrict__ matrix, long long rows,
                                                                 long long n, const double *__restrict__ factor,
                                                                 double *__restrict__ z0, double *__restrict__ z1)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *tile_y = smem;
    double *tile_a = tile_y + kTile * kStride;
    double *tile_b = tile_a + kTile * kStride;
    double(*coef)[kTile + 2] = reinterpret_cast<double(*)[kTile + 2]>(tile_b + kTile * kStride);
    // coef: d, l1(i-1), l2(i-2) of both parities for the tile's loci
    const int lane = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kTile;
    const int nrows = (int)((rows - row0 < kTile) ? (rows - row0) : kTile);
    const Factor f0 = factor_of(factor, n, 0), f1 = factor_of(factor, n, 1);
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;  // f[i-1], f[i-2] of parity 0 / 1
    for (long long base = 0; base < n; base += kTile) {
        const int T = (int)((n - base < kTile) ? (n - base) : kTile);
        __syncthreads();
        for (int r = 0; r < nrows; ++r) {
            if (lane < T) {
                tile_y[r * kStride + lane] = matrix[(row0 + r) * n + base + lane];
            }
        }
        if (lane < T) {
            const long long i = base + lane;
            coef[0][lane] = f0.d[i];
            coef[1][lane] = (i >= 1) ? f0.l1[i - 1] : 0.0;
            coef[2][lane] = (i >= 2) ? f0.l2[i - 2] : 0.0;
            coef[3][lane] = f1.d[i];
            coef[4][lane] = (i >= 1) ? f1.l1[i - 1] : 0.0;
            coef[5][lane] = (i >= 2) ? f1.l2[i - 2] : 0.0;
        }
        __syncthreads();
        if (lane < nrows) {
            for (int t = 0; t < T; ++t) {
                const long long i = base + t;
                const double y = tile_y[lane * kStride + t];
                const double ra = rhs_value(y, i, n, 0), rb = rhs_value(y, i, n, 1);
                double fa, fb;
                if (i == 0) {
                    fa = ra;
                    fb = rb;
                } else if (i == 1) {
                    fa = ra - (coef[1][t] * a1);
                    fb = rb - (coef[4][t] * b1);
                } else {
                    const double ta1 = coef[1][t] * a1, ta2 = coef[2][t] * a2;
                    const double tb1 = coef[4][t] * b1, tb2 = coef[5][t] * b2;
                    fa = ra - ta1 - ta2;
                    fb = rb - tb1 - tb2;
                }
                tile_a[lane * kStride + t] = fa / coef[0][t];
                tile_b[lane * kStride + t] = fb / coef[3][t];
                a2 = a1;
                a1 = fa;
                b2 = b1;
                b1 = fb;
            }
        }
        __syncthreads();
        for (int r = 0; r < nrows; ++r) {
            if (lane < T) {
                z0[(row0 + r) * n + base + lane] = tile_a[r * kStride + lane];
                z1[(row0 + r) * n + base + lane] = tile_b[r * kStride + lane];
            }
        }
    }
}

// backward substitution L^T x = z for both parities and the cross-fit average
// (baseline_backend.c:158-172, 296-299); out holds z of parity 0 on entry, the baseline on exit
__global__ __launch_bounds__(kTile) void whittaker_backward_kernel(long long rows, long long n,
                                                                  const double *__restrict__ factor,
                                                                  double *__restrict__ out, const double *__restrict__ z1)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *tile_a = smem;
    double *tile_b = tile_a + kTile * kStride;
    double(*coef)[kTile + 2] = reinterpret_cast<double(*)[kTile + 2]>(tile_b + kTile * kStride);
    // coef: l1(i), l2(i) of both parities
    const int lane = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kTile;
    const int nrows = (int)((rows - row0 < kTile) ? (rows - row0) : kTile);
    const Factor f0 = factor_of(factor, n, 0), f1 = factor_of(factor, n, 1);
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;  // x[i+1], x[i+2] of parity 0 / 1
    const long long last_base = ((n - 1) / kTile) * kTile;
    for (long long base = last_base; base >= 0; base -= kTile) {
        const int T = (int)((n - base < kTile) ? (n - base) : kTile);
        __syncthreads();
        for (int r = 0; r < nrows; ++r) {
            if (lane < T) {
                tile_a[r * kStride + lane] = out[(row0 + r) * n + base + lane];
                tile_b[r * kStride + lane] = z1[(row0 + r) * n + base + lane];
            }
        }
        if (lane < T) {
            const long long i = base + lane;
            coef[0][lane] = f0.l1[i];
            coef[1][lane] = f0.l2[i];
            coef[2][lane] = f1.l1[i];
            coef[3][lane] = f1.l2[i];
        }
        __syncthreads();
        if (lane < nrows) {
            for (int t = T - 1; t >= 0; --t) {
                const long long i = base + t;
                const double za = tile_a[lane * kStride + t], zb = tile_b[lane * kStride + t];
                double xa, xb;
                if (i == n - 1) {
                    xa = za;
                    xb = zb;
                } else if (i == n - 2) {
                    xa = za - (coef[0][t] * a1);
                    xb = zb - (coef[2][t] * b1);
                } else {
                    const double ta1 = coef[0][t] * a1, ta2 = coef[1][t] * a2;
                    const double tb1 = coef[2][t] * b1, tb2 = coef[3][t] * b2;
                    xa = za - ta1 - ta2;
                    xb = zb - tb1 - tb2;
                }
                tile_a[lane * kStride + t] = 0.5 * (xa + xb);
                a2 = a1;
                a1 = xa;
                b2 = b1;
                b1 = xb;
            }
        }
        __syncthreads();
        for (int r = 0; r < nrows; ++r) {
            if (lane < T) {
                out[(row0 + r) * n + base + lane] = tile_a[r * kStride + lane];
            }
        }
    }
}

__global__ void zero_kernel(double *out, long long count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        out[i] = 0.0;
    }
}

}  // namespace

size_t whittaker_scratch_bytes(size_t rows, size_t cols)
{
    return (6 * cols + rows * cols) * sizeof(double) + 256;
}

int launch_crossfit_whittaker(const double *matrix_dev, size_t rows, size_t cols, double penalty_lambda,
                              double *baseline_out_dev, void *scratch_dev, hipStream_t stream)
{
    if (rows == 0 || cols == 0) {
        return ROCCO_HIP_OK;
    }
    if (cols < 25) {  // baseline_backend.c:265-272
        const long long count = (long long)(rows * cols);
        hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                           baseline_out_dev, count);
        ROCCO_HIP_TRY(hipGetLastError());
        return ROCCO_HIP_OK;
    }
    double *factor = (double *)scratch_dev;
    double *z1 = factor + 6 * cols;
    const unsigned row_groups = (unsigned)((rows + kTile - 1) / kTile);
    const size_t lds_fwd = (size_t)(3 * kTile * kStride + 6 * (kTile + 2)) * sizeof(double);
    const size_t lds_bwd = (size_t)(2 * kTile * kStride + 4 * (kTile + 2)) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_forward_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fwd));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_backward_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
        attr_set = true;
    }
    hipLaunchKernelGGL(whittaker_factor_kernel, dim3(1), dim3(64), 0, stream, (long long)cols, penalty_lambda, factor);
    hipLaunchKernelGGL(whittaker_forward_kernel, dim3(row_groups), dim3(kTile), lds_fwd, stream, matrix_dev,
                       (long long)rows, (long long)cols, factor, baseline_out_dev, z1);
    hipLaunchKernelGGL(whittaker_backward_kernel, dim3(row_groups), dim3(kTile), lds_bwd, stream, (long long)rows,
                       (long long)cols, factor, baseline_out_dev, z1);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
