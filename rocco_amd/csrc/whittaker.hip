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
//   * the diagonal solve z = f / d and the cross-fit average are elementwise: they ride on the write-back
//     of the sweeps (a division inside a dependent chain costs more than the chain);
//   * the rows are independent and so are the two parities: one workgroup per row, one wavefront per
//     parity running the recurrence in a single lane, two more wavefronts feeding them through LDS (see
//     "forward / backward substitution" below).  The chains are issue-bound by construction (two multiplies
//     and two subtractions per locus, a lone wavefront issues one FP64 instruction per ~2.6 ns), so the
//     design keeps everything else off the chain wavefront and lets K rows x 2 parities run side by side.
#include "kernels.h"

#include <algorithm>

namespace rocco {

namespace {

constexpr int kLanes = 64;

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

// factor[p] = d | l1 | l2, each `cap` doubles, computed for length `cap` (baseline_backend.c:105-140).
// start > 0 resumes a factor whose entries 0 .. start-1 are already in place and do not depend on the length
// (start <= the previous length - 2: the last two entries of a factor carry its end bands).
__global__ __launch_bounds__(64) void whittaker_factor_kernel(long long cap, double lambda, double *factor, long long start)
{
    const int parity = threadIdx.x;
    if (parity >= 2) {
        return;
    }
    const long long n = cap;
    double *__restrict__ d = factor + (long long)parity * 3 * cap;
    double *__restrict__ l1 = d + cap;
    double *__restrict__ l2 = l1 + cap;
    double d_m2, l1_m2, l2_m2, d_m1, l1_m1, l2_m1;
    long long first = 2;
    if (start >= 4) {
        first = start;
        d_m2 = d[start - 2];
        l2_m2 = l2[start - 2];
        d_m1 = d[start - 1];
        l1_m1 = l1[start - 1];
        l2_m1 = l2[start - 1];
    } else {
    d_m2 = band_a0(0, n, parity, lambda);
    l1_m2 = band_a1(0, n, lambda) / d_m2;
    l2_m2 = lambda / d_m2;
    d[0] = d_m2;
    l1[0] = l1_m2;
    l2[0] = l2_m2;
    d_m1 = band_a0(1, n, parity, lambda) - ((l1_m2 * l1_m2) * d_m2);
    l1_m1 = (band_a1(1, n, lambda) - ((l2_m2 * d_m2) * l1_m2)) / d_m1;
    l2_m1 = (n > 3) ? (lambda / d_m1) : 0.0;
    d[1] = d_m1;
    l1[1] = l1_m1;
    l2[1] = l2_m1;
    }
    for (long long i = first; i < n; ++i) {
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

// ---- forward / backward substitution ------------------------------------------------------------------
// One workgroup per row: wavefront p (p = 0, 1) runs the recurrence of parity p in lane 0 -- a lone wavefront
// issues about one FP64 instruction per 2.6 ns whether 1 or 64 lanes are active, so the sweep time is
// (instructions per locus of the chain wavefront) x n and everything that is not the chain is moved off it:
// wavefronts 2 and 3 stage the next tile's inputs AND factor entries in LDS (coalesced loads issued before,
// stored after their other work; a load issued by the chain wavefront itself would put a memory latency into
// every batch) and carry the previous tile's results to global memory, applying the elementwise steps on the
// way (z = f / d after the forward sweep, baseline_backend.c:153-156; 0.5 (x0 + x1) after the backward sweep,
// 296-299).  The end cases of the recurrences are staged as zero coefficients (x - 0 * 0 - 0 * 0 == x exactly,
// signed zeros included), and so are the loci past the row's end, so the chain wavefront runs the same
// branch-free loop on every tile.  Three LDS tiles rotate (chain / write-back / staging), one barrier per tile.
constexpr int kSweepTile = 512;
constexpr int kSweepHelpers = 2 * kLanes;
constexpr int kSweepMoves = kSweepTile / kSweepHelpers;

struct SweepTile {
    double in[2][kSweepTile];
    double coef[2][kSweepTile][2];  // the multipliers of the previous and the one-before-previous value
    double out[2][kSweepTile];
};

template <bool BACKWARD>
__device__ __forceinline__ void sweep_chain(SweepTile &tile, int parity, double &p1, double &p2)
{
    const double *__restrict__ in = tile.in[parity];
    const double(*__restrict__ coef)[2] = tile.coef[parity];
    double *__restrict__ out = tile.out[parity];
    // batches of 8 loci; batch j covers [t(j), t(j) + 8), in sweep order
    auto batch_start = [](int j) { return BACKWARD ? (kSweepTile - 8 - 8 * j) : (8 * j); };
    constexpr int kBatches = kSweepTile / 8;
    double v[8], a[8], b[8], v2[8], a2[8], b2[8], r[8];
    auto fetch = [&](double(&vv)[8], double(&aa)[8], double(&bb)[8], int j) {
        const int t0 = batch_start((j < kBatches) ? j : (kBatches - 1));  // (past the tile's end: re-read the last batch)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            vv[k] = in[t0 + k];
            aa[k] = coef[t0 + k][0];
            bb[k] = coef[t0 + k][1];
        }
    };
    auto run = [&](const double(&vv)[8], const double(&aa)[8], const double(&bb)[8], int j) {
        const int t0 = batch_start(j);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int kk = BACKWARD ? (7 - k) : k;
            r[kk] = chain_step(vv[kk], aa[kk], bb[kk], p1, p2);
            p2 = p1;
            p1 = r[kk];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            out[t0 + k] = r[k];
        }
    };
    fetch(v, a, b, 0);
#pragma unroll 1
    for (int j = 0; j < kBatches; j += 2) {
        // two batches per trip, the operand registers ping-pong: the next batch's operands are fetched while
        // this batch's chain runs
        fetch(v2, a2, b2, j + 1);
        run(v, a, b, j);
        fetch(v, a, b, j + 2);
        run(v2, a2, b2, j + 1);
    }
}

// forward (BACKWARD = false): src0 = the matrix (rhs = W_p y), dst0 / dst1 = z of parity 0 / 1 (f / d);
// backward: src0 / src1 = z of parity 0 / 1, dst0 = the baseline (dst0 may be src0: every tile is read
// before it is written)
template <bool BACKWARD>
__global__ __launch_bounds__(2 * kLanes + kSweepHelpers) void whittaker_sweep_kernel(
    const double *src0, const double *src1, long long n, long long cap, const double *__restrict__ factor,
    const double *__restrict__ tail, double *dst0, double *dst1)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    SweepTile *tiles = reinterpret_cast<SweepTile *>(smem);  // [3]
    const int wave = threadIdx.x / kLanes, lane = threadIdx.x % kLanes;
    const bool helper = wave >= 2;
    const int hl = (int)threadIdx.x - 2 * kLanes;
    const long long row_off = (long long)blockIdx.x * n;
    const Factor f0 = factor_of(factor, tail, n, cap, 0), f1 = factor_of(factor, tail, n, cap, 1);
    const long long n_tiles = (n + kSweepTile - 1) / kSweepTile;
    // tile k of the sweep covers loci [base(k), base(k) + T(k))
    auto tile_base = [&](long long k) { return (BACKWARD ? (n_tiles - 1 - k) : k) * kSweepTile; };
    double s0[kSweepMoves], s1[kSweepMoves], ca[2][kSweepMoves], cb[2][kSweepMoves];
    auto stage_load = [&](long long k) {
        const long long b = tile_base(k);
#pragma unroll
        for (int j = 0; j < kSweepMoves; ++j) {
            const long long i = b + hl + j * kSweepHelpers;
            const long long ii = (i < n) ? i : (n - 1);  // loads without branches
            s0[j] = src0[row_off + ii];
            s1[j] = BACKWARD ? src1[row_off + ii] : 0.0;
            // the multipliers of locus i, end cases as zeros: forward l1[i-1], l2[i-2] (baseline_backend.c:142-151),
            // backward l1[i], l2[i] (158-172); Factor::ll1 / ll2 hold the entries that depend on the length
            const long long i1 = BACKWARD ? ii : ((ii >= 1) ? (ii - 1) : 0), i2 = BACKWARD ? ii : ((ii >= 2) ? (ii - 2) : 0);
            const bool has1 = BACKWARD ? (i < n - 1) : (i >= 1 && i < n), has2 = BACKWARD ? (i < n - 2) : (i >= 2 && i < n);
            const double a0 = f0.ll1(i1), a1 = f1.ll1(i1), b0 = f0.ll2(i2), b1 = f1.ll2(i2);
            ca[0][j] = has1 ? a0 : 0.0;
            ca[1][j] = has1 ? a1 : 0.0;
            cb[0][j] = has2 ? b0 : 0.0;
            cb[1][j] = has2 ? b1 : 0.0;
        }
    };
    auto stage_store = [&](long long k) {
        SweepTile &t = tiles[k % 3];
        const long long b = tile_base(k);
#pragma unroll
        for (int j = 0; j < kSweepMoves; ++j) {
            const int c = hl + j * kSweepHelpers;
            const long long i = b + c;
            const bool inside = i < n;
            const long long ic = inside ? i : (n - 1);
            if (BACKWARD) {
                t.in[0][c] = inside ? s0[j] : 0.0;
                t.in[1][c] = inside ? s1[j] : 0.0;
            } else {
                t.in[0][c] = inside ? rhs_value(s0[j], ic, n, 0) : 0.0;
                t.in[1][c] = inside ? rhs_value(s0[j], ic, n, 1) : 0.0;
            }
            t.coef[0][c][0] = ca[0][j];
            t.coef[0][c][1] = cb[0][j];
            t.coef[1][c][0] = ca[1][j];
            t.coef[1][c][1] = cb[1][j];
        }
    };
    auto write_back = [&](long long k) {
        const SweepTile &t = tiles[k % 3];
        const long long b = tile_base(k);
#pragma unroll
        for (int j = 0; j < kSweepMoves; ++j) {
            const int c = hl + j * kSweepHelpers;
            const long long i = b + c;
            if (i < n) {
                if (BACKWARD) {
                    dst0[row_off + i] = 0.5 * (t.out[0][c] + t.out[1][c]);
                } else {
                    dst0[row_off + i] = t.out[0][c] / f0.dd(i);
                    dst1[row_off + i] = t.out[1][c] / f1.dd(i);
                }
            }
        }
    };
    if (helper) {
        stage_load(0);
        stage_store(0);
    }
    __syncthreads();
    double p1 = 0.0, p2 = 0.0;
    for (long long k = 0; k < n_tiles; ++k) {
        if (!helper) {
#ifndef SWEEP_NO_CHAIN
            if (lane == 0) {
                sweep_chain<BACKWARD>(tiles[k % 3], wave, p1, p2);
            }
#endif
        } else {
#ifndef SWEEP_NO_HELPER
            if (k + 1 < n_tiles) {
                stage_load(k + 1);
            }
            if (k > 0) {
                write_back(k - 1);
            }
            if (k + 1 < n_tiles) {
                stage_store(k + 1);
            }
#endif
        }
        __syncthreads();
    }
    if (helper) {
        write_back(n_tiles - 1);
    }
}

// ---- the same sweeps with several chains per wavefront (round 3) ----------------------------------------------------
// A chain advances one locus per ~25 ns whatever its wavefront's other 63 lanes do (three dependent FP64 operations of
// ~19 cycles each), so a wavefront that carries ONE chain wastes them: above, K rows keep K workgroups busy for n x 25 ns,
// and the 2 400 rows of a genome's count matrices take sum(n) x 25 ns / (workgroups in flight).  Here lane L of the chain
// wavefront is chain (parity L / G, row L % G) of a group of G rows: both parities of a row read the same input, and all
// 2 G chains step through the loci in lockstep with the same instructions as one did before.  The rows of EVERY matrix of a
// batch (the chromosomes of a genome) are groups of one launch; the launch lasts as long as its longest row.  G = 8: with
// 32 rows (all 64 lanes busy) the four helper wavefronts, each streaming 8 far-apart rows in and out, need 39 ns per locus
// against the chain's 28 and the launch runs at 36; with 8 rows they stay under the chain and the launch runs at 27 ns per
// locus (a genome's 2 400 rows are 312 workgroups, all resident); 16 rows 29 ns, 4 rows 45 ns (too few lanes per LDS access).
//   * the helpers load a tile of 64 loci of each row (512 contiguous bytes per wavefront instruction) and store it
//     TRANSPOSED in LDS -- element (locus t, chain L) at t * (2 G + 1) + L: the chain wavefront's lanes read consecutive
//     doubles, a helper's 64 lanes (one row, 64 loci) write at an odd stride: no conflicts either way;
//   * the multipliers of a locus are the same for every row: one (a, b) pair per parity and locus, read as a broadcast;
//   * two input and two output tiles alternate (chain on k, staging of k + 1, write-back of k - 1), one barrier per tile.
constexpr int kRowTile = 64;    // loci per tile
#ifndef ROCCO_GROUP_ROWS
#define ROCCO_GROUP_ROWS 8
#endif
constexpr int kGroupRows = ROCCO_GROUP_ROWS;  // rows per workgroup (x 2 parities <= 64 lanes of the chain wavefront)
constexpr int kPitch = 2 * kGroupRows + 1;
static_assert(2 * kGroupRows <= kLanes, "one lane per chain");
#ifndef ROCCO_ROW_HELPERS
#define ROCCO_ROW_HELPERS 4
#endif
constexpr int kRowHelpers = ROCCO_ROW_HELPERS;  // helper wavefronts

struct RowTiles {
    double in[2][kRowTile * kPitch];
    double out[2][kRowTile * kPitch];
    double coef[2][2][kRowTile][2];  // [buffer][parity][locus] = multipliers of the previous / one-before-previous value
};

struct RowRegs {  // what a helper lane holds of one tile between its loads and its LDS stores
    double x0[kGroupRows / kRowHelpers], x1[kGroupRows / kRowHelpers];
    double d0, d1;        // backward: the diagonal entries of the lane's locus (z = f / d on the way into LDS)
    double ca[2], cb[2];  // wavefront 0 of the helpers: the multipliers of the lane's locus, per parity
};

// forward (BACKWARD = false): src0 = the matrix (rhs = W_p y); dst0 / dst1 = f of parity 0 / 1, the forward substitution
// BEFORE its division by the diagonal (baseline_backend.c:142-156) -- the division rides on the backward sweep's staging,
// where the diagonal entry arrives with the tile's other loads instead of queueing for three tiles in registers.
// backward: src0 / src1 = f of parity 0 / 1, dst0 = the baseline 0.5 (x0 + x1) (158-172, 296-299); dst0 may be src0.
template <bool BACKWARD>
__global__ __launch_bounds__(kLanes *(1 + kRowHelpers)) void whittaker_rows_kernel(const WhittakerRowTask *__restrict__ tasks,
                                                                                  long long cap, const double *__restrict__ factor)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    RowTiles &T = *reinterpret_cast<RowTiles *>(smem);
    const WhittakerRowTask task = tasks[blockIdx.x];
    const int wave = threadIdx.x / kLanes, lane = threadIdx.x % kLanes;
    const int h = wave - 1;  // helper number
    const long long n = task.n;
    const Factor f0 = factor_of(factor, task.tail, n, cap, 0), f1 = factor_of(factor, task.tail, n, cap, 1);
    const long long n_tiles = (n + kRowTile - 1) / kRowTile;
    auto tile_base = [&](long long k) { return (BACKWARD ? (n_tiles - 1 - k) : k) * kRowTile; };
    constexpr int kMine = kGroupRows / kRowHelpers;  // rows a helper wavefront moves per tile

    // helpers: lane = locus of the tile.  A tile is loaded into registers TWO trips before it is stored to LDS (two
    // register sets take turns), every load of a trip issued before anything waits for one: a load has two chains of a
    // tile -- several memory latencies -- to arrive.
    auto stage_load = [&](long long k, RowRegs &R) {
        const long long i = tile_base(k) + lane;
        const long long ii = (i < n) ? i : (n - 1);  // loads without branches
#pragma unroll
        for (int q = 0; q < kMine; ++q) {
            const int r = h + q * kRowHelpers;
            const long long at = (long long)(task.row0 + ((r < task.rows) ? r : 0)) * n + ii;
            R.x0[q] = task.src0[at];
            R.x1[q] = BACKWARD ? task.src1[at] : 0.0;
        }
        if (BACKWARD) {
            R.d0 = f0.dd(ii);
            R.d1 = f1.dd(ii);
        }
        if (h == 0) {
            // the multipliers of locus i, end cases as zeros: forward l1[i-1], l2[i-2] (baseline_backend.c:142-151),
            // backward l1[i], l2[i] (158-172); Factor::ll1 / ll2 hold the entries that depend on the length
            const long long i1 = BACKWARD ? ii : ((ii >= 1) ? (ii - 1) : 0), i2 = BACKWARD ? ii : ((ii >= 2) ? (ii - 2) : 0);
            const bool has1 = BACKWARD ? (i < n - 1) : (i >= 1 && i < n), has2 = BACKWARD ? (i < n - 2) : (i >= 2 && i < n);
            const double a0 = f0.ll1(i1), a1 = f1.ll1(i1), b0 = f0.ll2(i2), b1 = f1.ll2(i2);
            R.ca[0] = has1 ? a0 : 0.0;
            R.ca[1] = has1 ? a1 : 0.0;
            R.cb[0] = has2 ? b0 : 0.0;
            R.cb[1] = has2 ? b1 : 0.0;
        }
    };
    auto stage_store = [&](long long k, const RowRegs &R) {
        const int buf = (int)(k & 1);
        const long long i = tile_base(k) + lane;
        const bool inside = i < n;
        const long long ii = inside ? i : (n - 1);
        double *__restrict__ in = T.in[buf];
#pragma unroll
        for (int q = 0; q < kMine; ++q) {
            const int r = h + q * kRowHelpers;
            const bool live = inside && r < task.rows;
            double v0, v1;
            if (BACKWARD) {
                v0 = R.x0[q] / R.d0;  // z = f / d (baseline_backend.c:153-156)
                v1 = R.x1[q] / R.d1;
            } else {
                v0 = rhs_value(R.x0[q], ii, n, 0);
                v1 = rhs_value(R.x0[q], ii, n, 1);
            }
            in[lane * kPitch + r] = live ? v0 : 0.0;
            in[lane * kPitch + kGroupRows + r] = live ? v1 : 0.0;
        }
        if (h == 0) {
            T.coef[buf][0][lane][0] = R.ca[0];
            T.coef[buf][0][lane][1] = R.cb[0];
            T.coef[buf][1][lane][0] = R.ca[1];
            T.coef[buf][1][lane][1] = R.cb[1];
        }
    };
    auto write_back = [&](long long k) {
        const double *__restrict__ out = T.out[k & 1];
        const long long i = tile_base(k) + lane;
        if (i >= n) {
            return;
        }
        for (int r = h; r < task.rows; r += kRowHelpers) {
            const long long at = (long long)(task.row0 + r) * n + i;
            if (BACKWARD) {
                task.dst0[at] = 0.5 * (out[lane * kPitch + r] + out[lane * kPitch + kGroupRows + r]);
            } else {
                task.dst0[at] = out[lane * kPitch + r];
                task.dst1[at] = out[lane * kPitch + kGroupRows + r];
            }
        }
    };

    double p1 = 0.0, p2 = 0.0;
    const bool chain_lane = lane < 2 * kGroupRows;  // (2 x kGroupRows chains; the other lanes of the chain wavefront idle)
    const int col = chain_lane ? lane : 0;
    const int parity = col / kGroupRows;
    auto chain = [&](long long k) {
        const double *__restrict__ in = T.in[k & 1];
        double *__restrict__ out = T.out[k & 1];
        const double(*__restrict__ coef)[2] = T.coef[k & 1][parity];
        // batches of 8 loci in sweep order; the operands of the next batch are fetched while this one's chain runs
        double v[8], a[8], b[8], v2[8], a2[8], b2[8], r[8];
        auto start = [](int j) { return BACKWARD ? (kRowTile - 8 - 8 * j) : (8 * j); };
        auto fetch = [&](double(&vv)[8], double(&aa)[8], double(&bb)[8], int j) {
            const int t0 = start((j < kRowTile / 8) ? j : (kRowTile / 8 - 1));
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                vv[q] = in[(t0 + q) * kPitch + col];
                aa[q] = coef[t0 + q][0];
                bb[q] = coef[t0 + q][1];
            }
        };
        auto run = [&](const double(&vv)[8], const double(&aa)[8], const double(&bb)[8], int j) {
            const int t0 = start(j);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int qq = BACKWARD ? (7 - q) : q;
                r[qq] = chain_step(vv[qq], aa[qq], bb[qq], p1, p2);
                p2 = p1;
                p1 = r[qq];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (chain_lane) {
                    out[(t0 + q) * kPitch + col] = r[q];
                }
            }
        };
        fetch(v, a, b, 0);
#pragma unroll 1
        for (int j = 0; j < kRowTile / 8; j += 2) {
            fetch(v2, a2, b2, j + 1);
            run(v, a, b, j);
            fetch(v, a, b, j + 2);
            run(v2, a2, b2, j + 1);
        }
    };
    // one trip: the chain wavefront runs tile k; the helpers store tile k + 1 (in `set` since two trips), refill `set`
    // with tile k + 3 and carry tile k - 1 to memory
    auto trip = [&](long long k, RowRegs &set) {
#ifdef ROCCO_ROWS_NOCHAIN  // (timing experiments only)
        if (false) {
#else
        if (wave == 0) {
#endif
            chain(k);
#ifdef ROCCO_ROWS_NOHELP
        } else if (false) {
#else
        } else if (wave > 0) {
#endif
            if (k + 1 < n_tiles) {
                stage_store(k + 1, set);
            }
            if (k + 3 < n_tiles) {
                stage_load(k + 3, set);
            }
            if (k > 0) {
                write_back(k - 1);
            }
        }
        __syncthreads();
    };

    RowRegs A, B;
    if (wave > 0) {
        stage_load(0, A);
        stage_store(0, A);
        if (n_tiles > 1) {
            stage_load(1, B);
        }
        if (n_tiles > 2) {
            stage_load(2, A);
        }
    } else {
        __builtin_amdgcn_s_setprio(3);  // the chain wavefront shares its SIMD with a helper: its instructions go first
    }
    __syncthreads();
    for (long long k = 0; k < n_tiles; k += 2) {
        trip(k, B);  // even trips store odd tiles (set B), odd trips even tiles (set A)
        if (k + 1 < n_tiles) {
            trip(k + 1, A);
        }
    }
    if (wave > 0) {
        write_back(n_tiles - 1);
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
    return whittaker_batch_scratch_bytes(&rows, &cols, 1);
}

int launch_whittaker_factor(size_t cap, double penalty_lambda, double *factor_dev, hipStream_t stream,
                            const double *old_factor_dev, size_t old_cap)
{
    // a longer factor of the same penalty continues the old one: its entries 0 .. old_cap-3 carry over
    long long start = 0;
    if (old_factor_dev != nullptr && old_cap >= 8 && old_cap < cap) {
        start = (long long)old_cap - 2;
        for (int a = 0; a < 6; ++a) {
            ROCCO_HIP_TRY(hipMemcpyAsync(factor_dev + (size_t)a * cap, old_factor_dev + (size_t)a * old_cap,
                                         (size_t)start * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
    }
    hipLaunchKernelGGL(whittaker_factor_kernel, dim3(1), dim3(64), 0, stream, (long long)cap, penalty_lambda,
                       factor_dev, start);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

namespace {

int configure_rows_kernels()
{
    static bool attr_set = false;
    if (!attr_set) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_rows_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RowTiles)));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_rows_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RowTiles)));
        attr_set = true;
    }
    return ROCCO_HIP_OK;
}

}  // namespace

int whittaker_group_rows() { return kGroupRows; }

size_t whittaker_batch_scratch_bytes(const size_t *rows, const size_t *cols, size_t count)
{
    // per matrix: the z of parity 1 (rows x cols doubles) and 8 doubles of end entries; per group of rows two task records
    size_t bytes = 256;
    for (size_t i = 0; i < count; ++i) {
        bytes += (rows[i] * cols[i] + 8) * sizeof(double) + 2 * ((rows[i] + kGroupRows - 1) / kGroupRows) * sizeof(WhittakerRowTask) + 256;
    }
    return bytes;
}

int launch_crossfit_whittaker_batch(const double *const *matrices_dev, const size_t *rows, const size_t *cols, size_t count,
                                    double penalty_lambda, const double *factor_dev, size_t factor_cap, double *const *baselines_dev,
                                    void *scratch_dev, WhittakerRowTask *tasks_host_pinned, hipStream_t stream)
{
    int rc;
    if (count == 1 && rows[0] > 0 && cols[0] >= 25) {
        // ONE matrix: a workgroup per row with one chain per wavefront steps ~25 ns per locus, the grouped wavefronts below
        // ~27 ns -- both last as long as one row, so the lone matrix takes the faster step; two matrices or more take the
        // launch that lasts as long as its longest
        if (factor_dev == nullptr || factor_cap < cols[0]) {
            return ROCCO_HIP_EINVAL;
        }
        double *tail = (double *)scratch_dev;  // 6 doubles
        double *z1 = tail + 8;
        const long long n = (long long)cols[0], cap = (long long)factor_cap;
        hipLaunchKernelGGL(whittaker_tail_kernel, dim3(1), dim3(64), 0, stream, n, cap, penalty_lambda, factor_dev, tail);
        const dim3 block(2 * kLanes + kSweepHelpers);
        const size_t lds = 3 * sizeof(SweepTile);
        static bool attr_set = false;
        if (!attr_set) {
            ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_sweep_kernel<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_sweep_kernel<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(whittaker_sweep_kernel<false>, dim3((unsigned)rows[0]), block, lds, stream, matrices_dev[0],
                           (const double *)nullptr, n, cap, factor_dev, tail, baselines_dev[0], z1);
        hipLaunchKernelGGL(whittaker_sweep_kernel<true>, dim3((unsigned)rows[0]), block, lds, stream,
                           (const double *)baselines_dev[0], (const double *)z1, n, cap, factor_dev, tail, baselines_dev[0],
                           (double *)nullptr);
        ROCCO_HIP_TRY(hipGetLastError());
        return ROCCO_HIP_OK;
    }
    if ((rc = configure_rows_kernels()) != ROCCO_HIP_OK) return rc;
    // scratch: [tasks forward | tasks backward] then per matrix [tail (8 doubles) | z1]
    size_t n_tasks = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] > 0 && cols[i] >= 25) {
            n_tasks += (rows[i] + kGroupRows - 1) / kGroupRows;
        }
    }
    char *at = (char *)scratch_dev;
    WhittakerRowTask *tasks_dev = (WhittakerRowTask *)at;
    at += ((2 * n_tasks * sizeof(WhittakerRowTask) + 255) / 256) * 256;
    size_t t = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] == 0 || cols[i] == 0) {
            continue;
        }
        if (cols[i] < 25) {  // baseline_backend.c:265-272
            const long long total = (long long)(rows[i] * cols[i]);
            hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, baselines_dev[i], total);
            continue;
        }
        if (factor_dev == nullptr || factor_cap < cols[i]) {
            return ROCCO_HIP_EINVAL;
        }
        double *tail = (double *)at;
        double *z1 = tail + 8;
        at += ((rows[i] * cols[i] + 8) * sizeof(double) + 255) / 256 * 256;
        hipLaunchKernelGGL(whittaker_tail_kernel, dim3(1), dim3(64), 0, stream, (long long)cols[i], (long long)factor_cap, penalty_lambda,
                           factor_dev, tail);
        for (size_t r0 = 0; r0 < rows[i]; r0 += kGroupRows, ++t) {
            WhittakerRowTask &fw = tasks_host_pinned[t], &bw = tasks_host_pinned[n_tasks + t];
            fw.src0 = matrices_dev[i];
            fw.src1 = nullptr;
            fw.dst0 = baselines_dev[i];  // z of parity 0 lives in the output until the backward sweep replaces it
            fw.dst1 = z1;
            fw.n = (long long)cols[i];
            fw.row0 = (int)r0;
            fw.rows = (int)std::min<size_t>(kGroupRows, rows[i] - r0);
            fw.tail = tail;
            bw = fw;
            bw.src0 = baselines_dev[i];
            bw.src1 = z1;
            bw.dst0 = baselines_dev[i];
            bw.dst1 = nullptr;
        }
    }
    if (n_tasks == 0) {
        ROCCO_HIP_TRY(hipGetLastError());
        return ROCCO_HIP_OK;
    }
    ROCCO_HIP_TRY(hipMemcpyAsync(tasks_dev, tasks_host_pinned, 2 * n_tasks * sizeof(WhittakerRowTask), hipMemcpyHostToDevice, stream));
    const dim3 block(kLanes * (1 + kRowHelpers));
    hipLaunchKernelGGL(whittaker_rows_kernel<false>, dim3((unsigned)n_tasks), block, sizeof(RowTiles), stream,
                       (const WhittakerRowTask *)tasks_dev, (long long)factor_cap, factor_dev);
    hipLaunchKernelGGL(whittaker_rows_kernel<true>, dim3((unsigned)n_tasks), block, sizeof(RowTiles), stream,
                       (const WhittakerRowTask *)(tasks_dev + n_tasks), (long long)factor_cap, factor_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
