// rocco_amd/csrc/whittaker.hip -- cross-fit Whittaker baseline of every row of a K x n matrix, gfx950.
//
// Replaces rocco/native/baseline_backend.c:305-334 (rocco_crossfit_whittaker_baseline_matrix_f64; row
// kernel 252-303, band setup 175-250, LDL^T solve 79-173), called from rocco/inference.py:185-209.
//
// Per row the reference solves  (W_p + lambda D^T D) b = W_p y  for the two parity masks p and averages
// the two fits.  Results must equal the reference bit for bit, so every recurrence runs in the
// reference's own order (linear recurrences with rounding are not associative).  What can be shared and
// what can run side by side:
//   * the LDL^T factor (d, l1, l2) does not depend on the data, and its entry i does not depend on the
//     length n except for the last two loci (the bands differ only there): it is computed ONCE per
//     penalty for the longest length seen (one lane per parity; the reference recomputes it for every
//     row), kept in the solver, and a tiny kernel recomputes the two end entries for each length;
//   * the diagonal solve z = f / d is elementwise: a separate, fully parallel kernel (a division inside
//     a dependent chain costs more than the chain);
//   * the rows are independent: forward and backward substitution run with one lane per row, both
//     parities in the same lane (two dependent chains interleave).  A workgroup takes 16 rows and moves
//     them through LDS in tiles of 256 loci with coalesced 512-byte accesses (a lane reading its own row
//     directly would touch one cache line per lane per instruction), all 64 lanes moving data, 16 of
//     them running chains: the chains are latency-bound by construction (multiply, subtract, subtract
//     per locus and parity), so narrow workgroups on many CUs beat wide ones.
#include "kernels.h"

namespace rocco {

namespace {

constexpr int kRows = 16;             // rows (chains) per workgroup
constexpr int kTile = 256;            // loci per tile
constexpr int kStride = kTile + 1;    // LDS row stride (odd multiple of 8 bytes: no bank conflicts)
constexpr int kLanes = 64;
constexpr int kMoves = kRows * kTile / kLanes;  // 8-byte elements each lane moves per tile

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

// factor[p] = d | l1 | l2, each `cap` doubles, computed for length `cap` (baseline_backend.c:105-140)
__global__ __launch_bounds__(64) void whittaker_factor_kernel(long long cap, double lambda, double *factor)
{
    const int parity = threadIdx.x;
    if (parity >= 2) {
        return;
    }
    const long long n = cap;
    double *__restrict__ d = factor + (long long)parity * 3 * cap;
    double *__restrict__ l1 = d + cap;
    double *__restrict__ l2 = l1 + cap;
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

// The factor of length n from the one computed for cap >= n: entries 0 .. n-3 coincide (same bands, same
// recurrence); tail[p] = { d[n-2], d[n-1], l1[n-2] } are recomputed with the end bands of length n.
__global__ __launch_bounds__(64) void whittaker_tail_kernel(long long n, long long cap, double lambda,
                                                           const double *__restrict__ factor, double *__restrict__ tail)
{
    const int parity = threadIdx.x;
    if (parity >= 2) {
        return;
    }
    const double *d = factor + (long long)parity * 3 * cap;
    const double *l1 = d + cap;
    const double *l2 = l1 + cap;
    double *out = tail + 3 * parity;
    if (n == cap) {
        out[0] = d[n - 2];
        out[1] = d[n - 1];
        out[2] = l1[n - 2];
        return;
    }
    // i = n - 2 (baseline_backend.c:125-134 with the bands of 204-206, 224)
    double t1 = ((l1[n - 3] * l1[n - 3]) * d[n - 3]);
    double t2 = ((l2[n - 4] * l2[n - 4]) * d[n - 4]);
    const double d_n2 = band_a0(n - 2, n, parity, lambda) - t1 - t2;
    t1 = ((l2[n - 3] * d[n - 3]) * l1[n - 3]);
    const double l1_n2 = (band_a1(n - 2, n, lambda) - t1) / d_n2;
    // i = n - 1
    t1 = ((l1_n2 * l1_n2) * d_n2);
    t2 = ((l2[n - 3] * l2[n - 3]) * d[n - 3]);
    out[0] = d_n2;
    out[1] = band_a0(n - 1, n, parity, lambda) - t1 - t2;
    out[2] = l1_n2;
}

struct Factor {
    const double *d, *l1, *l2;
    double d_n2, d_n1, l1_n2;  // the entries that depend on the length
    long long n;
    __device__ __forceinline__ double dd(long long i) const
    {
        return (i == n - 2) ? d_n2 : ((i == n - 1) ? d_n1 : d[i]);
    }
    __device__ __forceinline__ double ll1(long long i) const
    {
        return (i == n - 2) ? l1_n2 : ((i > n - 2) ? 0.0 : l1[i]);
    }
    __device__ __forceinline__ double ll2(long long i) const { return (i > n - 3) ? 0.0 : l2[i]; }
};

__device__ __forceinline__ Factor factor_of(const double *factor, const double *tail, long long n, long long cap,
                                            int parity)
{
    Factor f;
    f.d = factor + (long long)parity * 3 * cap;
    f.l1 = f.d + cap;
    f.l2 = f.l1 + cap;
    f.d_n2 = tail[3 * parity + 0];
    f.d_n1 = tail[3 * parity + 1];
    f.l1_n2 = tail[3 * parity + 2];
    f.n = n;
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

// v - c1 * p1 - c2 * p2 in the reference's order (baseline_backend.c:146-151 and 167-172)
__device__ __forceinline__ double chain_step(double v, double c1, double c2, double p1, double p2)
{
    const double t1 = c1 * p1;
    const double t2 = c2 * p2;
    return v - t1 - t2;
}

// move a tile between global memory (row-major, row length n) and LDS ([row][kStride]); every instruction
// of the wavefront touches 64 consecutive loci of one row
template <bool TO_LDS>
__device__ __forceinline__ void move_tile(double *__restrict__ lds, double *__restrict__ global, long long row0,
                                          int nrows, long long n, long long base, int T)
{
    const int lane = threadIdx.x;
    if (nrows == kRows && T == kTile) {
        // full tile (workgroup-uniform): unconditional accesses, all in flight together -- loads inside
        // per-element conditionals are waited for one by one
        if (TO_LDS) {
            double v[kMoves];
#pragma unroll
            for (int k = 0; k < kMoves; ++k) {
                const int e = k * kLanes + lane;
                v[k] = global[(row0 + e / kTile) * n + base + e % kTile];
            }
#pragma unroll
            for (int k = 0; k < kMoves; ++k) {
                const int e = k * kLanes + lane;
                lds[(e / kTile) * kStride + e % kTile] = v[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kMoves; ++k) {
                const int e = k * kLanes + lane;
                global[(row0 + e / kTile) * n + base + e % kTile] = lds[(e / kTile) * kStride + e % kTile];
            }
        }
        return;
    }
#pragma unroll 8
    for (int k = 0; k < kMoves; ++k) {
        const int e = k * kLanes + lane;
        const int r = e / kTile, c = e % kTile;
        if (r < nrows && c < T) {
            if (TO_LDS) {
                lds[r * kStride + c] = global[(row0 + r) * n + base + c];
            } else {
                global[(row0 + r) * n + base + c] = lds[r * kStride + c];
            }
        }
    }
}

// forward substitution L f = rhs for both parities (baseline_backend.c:142-151): f of parity 0 -> fa_out
// (the output buffer), parity 1 -> fb_out (scratch)
__global__ __launch_bounds__(kLanes) void whittaker_forward_kernel(const double *__restrict__ matrix, long long rows,
                                                                  long long n, long long cap,
                                                                  const double *__restrict__ factor,
                                                                  const double *__restrict__ tail,
                                                                  double *__restrict__ fa_out, double *__restrict__ fb_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *tile_y = smem;
    double *tile_a = tile_y + kRows * kStride;
    double *tile_b = tile_a + kRows * kStride;
    double *coef = tile_b + kRows * kStride;  // [4][kTile]: l1(i-1), l2(i-2) of parity 0, then of parity 1
    const int lane = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kRows;
    const int nrows = (int)((rows - row0 < kRows) ? (rows - row0) : kRows);
    const Factor f0 = factor_of(factor, tail, n, cap, 0), f1 = factor_of(factor, tail, n, cap, 1);
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;  // f[i-1], f[i-2] of parity 0 / 1
    for (long long base = 0; base < n; base += kTile) {
        const int T = (int)((n - base < kTile) ? (n - base) : kTile);
        __syncthreads();
        move_tile<true>(tile_y, const_cast<double *>(matrix), row0, nrows, n, base, T);
        for (int c = lane; c < T; c += kLanes) {
            const long long i = base + c;
            coef[0 * kTile + c] = (i >= 1) ? f0.ll1(i - 1) : 0.0;
            coef[1 * kTile + c] = (i >= 2) ? f0.ll2(i - 2) : 0.0;
            coef[2 * kTile + c] = (i >= 1) ? f1.ll1(i - 1) : 0.0;
            coef[3 * kTile + c] = (i >= 2) ? f1.ll2(i - 2) : 0.0;
        }
        __syncthreads();
        if (lane < nrows) {
            const double *__restrict__ yrow = tile_y + lane * kStride;
            double *__restrict__ arow = tile_a + lane * kStride;
            double *__restrict__ brow = tile_b + lane * kStride;
            if (T == kTile && base >= 2 && base + kTile + 2 <= n) {
                // interior tile: no end cases; unrolled so that the LDS reads run ahead of the chains.
                // rhs = w * y, w = 1 on the locus' own parity and 0 on the other (baseline_backend.c:208-209)
                // Inputs of eight loci are read into registers first: LDS reads issued one by one between
                // the (possibly aliasing) LDS writes would each cost a full LDS round trip inside the chain.
#pragma unroll 1
                for (int t0 = 0; t0 < kTile; t0 += 8) {
                    double yv[8], c0[8], c1[8], c2[8], c3[8], fa[8], fb[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // the factor is the same for every lane: wave-uniform (scalar) loads, no LDS traffic
                        const long long i = base + t0 + k;
                        yv[k] = yrow[t0 + k];
                        c0[k] = f0.l1[i - 1];
                        c1[k] = f0.l2[i - 2];
                        c2[k] = f1.l1[i - 1];
                        c3[k] = f1.l2[i - 2];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // even loci (t0 is even) carry weight 1 for parity 0 and 0 for parity 1; odd the reverse
                        const double wa = (k & 1) ? 0.0 : 1.0, wb = (k & 1) ? 1.0 : 0.0;
                        fa[k] = chain_step(wa * yv[k], c0[k], c1[k], a1, a2);
                        fb[k] = chain_step(wb * yv[k], c2[k], c3[k], b1, b2);
                        a2 = a1;
                        a1 = fa[k];
                        b2 = b1;
                        b1 = fb[k];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        arow[t0 + k] = fa[k];
                        brow[t0 + k] = fb[k];
                    }
                }
            } else {
                for (int t = 0; t < T; ++t) {
                    const long long i = base + t;
                    const double y = yrow[t];
                    const double ra = rhs_value(y, i, n, 0), rb = rhs_value(y, i, n, 1);
                    double fa, fb;
                    if (i == 0) {
                        fa = ra;
                        fb = rb;
                    } else if (i == 1) {
                        fa = ra - (coef[0 * kTile + t] * a1);
                        fb = rb - (coef[2 * kTile + t] * b1);
                    } else {
                        fa = chain_step(ra, coef[0 * kTile + t], coef[1 * kTile + t], a1, a2);
                        fb = chain_step(rb, coef[2 * kTile + t], coef[3 * kTile + t], b1, b2);
                    }
                    arow[t] = fa;
                    brow[t] = fb;
                    a2 = a1;
                    a1 = fa;
                    b2 = b1;
                    b1 = fb;
                }
            }
        }
        __syncthreads();
        move_tile<false>(tile_a, fa_out, row0, nrows, n, base, T);
        move_tile<false>(tile_b, fb_out, row0, nrows, n, base, T);
    }
}

// diagonal solve z = f / d (baseline_backend.c:153-156), every element independently
__global__ __launch_bounds__(256) void whittaker_diagonal_kernel(long long rows, long long n, long long cap,
                                                                 const double *__restrict__ factor,
                                                                 const double *__restrict__ tail,
                                                                 double *__restrict__ z0, double *__restrict__ z1)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const double d0 = factor_of(factor, tail, n, cap, 0).dd(i), d1 = factor_of(factor, tail, n, cap, 1).dd(i);
    for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
        z0[r * n + i] = z0[r * n + i] / d0;
        z1[r * n + i] = z1[r * n + i] / d1;
    }
}

// backward substitution L^T x = z for both parities and the cross-fit average
// (baseline_backend.c:158-172, 296-299); out holds z of parity 0 on entry, the baseline on exit
__global__ __launch_bounds__(kLanes) void whittaker_backward_kernel(long long rows, long long n, long long cap,
                                                                   const double *__restrict__ factor,
                                                                   const double *__restrict__ tail,
                                                                   double *__restrict__ out, const double *__restrict__ z1)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *tile_a = smem;
    double *tile_b = tile_a + kRows * kStride;
    double *coef = tile_b + kRows * kStride;  // [4][kTile]: l1(i), l2(i) of parity 0, then of parity 1
    const int lane = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kRows;
    const int nrows = (int)((rows - row0 < kRows) ? (rows - row0) : kRows);
    const Factor f0 = factor_of(factor, tail, n, cap, 0), f1 = factor_of(factor, tail, n, cap, 1);
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;  // x[i+1], x[i+2] of parity 0 / 1
    const long long last_base = ((n - 1) / kTile) * kTile;
    for (long long base = last_base; base >= 0; base -= kTile) {
        const int T = (int)((n - base < kTile) ? (n - base) : kTile);
        __syncthreads();
        move_tile<true>(tile_a, out, row0, nrows, n, base, T);
        move_tile<true>(tile_b, const_cast<double *>(z1), row0, nrows, n, base, T);
        for (int c = lane; c < T; c += kLanes) {
            const long long i = base + c;
            coef[0 * kTile + c] = f0.ll1(i);
            coef[1 * kTile + c] = f0.ll2(i);
            coef[2 * kTile + c] = f1.ll1(i);
            coef[3 * kTile + c] = f1.ll2(i);
        }
        __syncthreads();
        if (lane < nrows) {
            double *__restrict__ arow = tile_a + lane * kStride;
            const double *__restrict__ brow = tile_b + lane * kStride;
            if (T == kTile && base + kTile + 2 <= n) {
                // interior tile: x[i] = z[i] - l1[i] x[i+1] - l2[i] x[i+2]
#pragma unroll 1
                for (int t0 = kTile - 8; t0 >= 0; t0 -= 8) {
                    double za[8], zb[8], c0[8], c1[8], c2[8], c3[8], xm[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const long long i = base + t0 + k;
                        za[k] = arow[t0 + k];
                        zb[k] = brow[t0 + k];
                        c0[k] = f0.l1[i];
                        c1[k] = f0.l2[i];
                        c2[k] = f1.l1[i];
                        c3[k] = f1.l2[i];
                    }
#pragma unroll
                    for (int k = 7; k >= 0; --k) {
                        const double xa = chain_step(za[k], c0[k], c1[k], a1, a2);
                        const double xb = chain_step(zb[k], c2[k], c3[k], b1, b2);
                        xm[k] = 0.5 * (xa + xb);
                        a2 = a1;
                        a1 = xa;
                        b2 = b1;
                        b1 = xb;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        arow[t0 + k] = xm[k];
                    }
                }
            } else {
                for (int t = T - 1; t >= 0; --t) {
                    const long long i = base + t;
                    const double za = arow[t], zb = brow[t];
                    double xa, xb;
                    if (i == n - 1) {
                        xa = za;
                        xb = zb;
                    } else if (i == n - 2) {
                        xa = za - (coef[0 * kTile + t] * a1);
                        xb = zb - (coef[2 * kTile + t] * b1);
                    } else {
                        xa = chain_step(za, coef[0 * kTile + t], coef[1 * kTile + t], a1, a2);
                        xb = chain_step(zb, coef[2 * kTile + t], coef[3 * kTile + t], b1, b2);
                    }
                    arow[t] = 0.5 * (xa + xb);
                    a2 = a1;
                    a1 = xa;
                    b2 = b1;
                    b1 = xb;
                }
            }
        }
        __syncthreads();
        move_tile<false>(tile_a, out, row0, nrows, n, base, T);
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
    return (rows * cols + 8) * sizeof(double) + 256;
}

int launch_whittaker_factor(size_t cap, double penalty_lambda, double *factor_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(whittaker_factor_kernel, dim3(1), dim3(64), 0, stream, (long long)cap, penalty_lambda,
                       factor_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_crossfit_whittaker(const double *matrix_dev, size_t rows, size_t cols, double penalty_lambda,
                              const double *factor_dev, size_t factor_cap, double *baseline_out_dev,
                              void *scratch_dev, hipStream_t stream)
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
    if (factor_dev == nullptr || factor_cap < cols) {
        return ROCCO_HIP_EINVAL;
    }
    double *tail = (double *)scratch_dev;  // 6 doubles
    double *z1 = tail + 8;
    const unsigned row_groups = (unsigned)((rows + kRows - 1) / kRows);
    const size_t lds_fwd = (size_t)(3 * kRows * kStride + 4 * kTile) * sizeof(double);
    const size_t lds_bwd = (size_t)(2 * kRows * kStride + 4 * kTile) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_forward_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fwd));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_backward_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bwd));
        attr_set = true;
    }
    const long long n = (long long)cols, cap = (long long)factor_cap;
    hipLaunchKernelGGL(whittaker_tail_kernel, dim3(1), dim3(64), 0, stream, n, cap, penalty_lambda, factor_dev, tail);
    hipLaunchKernelGGL(whittaker_forward_kernel, dim3(row_groups), dim3(kLanes), lds_fwd, stream, matrix_dev,
                       (long long)rows, n, cap, factor_dev, tail, baseline_out_dev, z1);
    {
        const unsigned gx = (unsigned)((cols + 255) / 256);
        const unsigned gy = (unsigned)((rows < 64) ? rows : 64);
        hipLaunchKernelGGL(whittaker_diagonal_kernel, dim3(gx, gy), dim3(256), 0, stream, (long long)rows, n, cap,
                           factor_dev, tail, baseline_out_dev, z1);
    }
    hipLaunchKernelGGL(whittaker_backward_kernel, dim3(row_groups), dim3(kLanes), lds_bwd, stream, (long long)rows, n,
                       cap, factor_dev, tail, baseline_out_dev, z1);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
