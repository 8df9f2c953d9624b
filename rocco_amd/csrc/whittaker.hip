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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    // the same from a value loaded earlier from d[i] / l1[i] / l2[i] (i < n): a select on a value just loaded waits for the
    // load, so the staging wavefronts load the table entries untouched and apply the end cases when they store the tile
    __device__ __forceinline__ double dd_of(double raw, long long i) const { return (i == n - 2) ? d_n2 : ((i == n - 1) ? d_n1 : raw); }
    __device__ __forceinline__ double ll1_of(double raw, long long i) const { return (i == n - 2) ? l1_n2 : ((i > n - 2) ? 0.0 : raw); }
    __device__ __forceinline__ double ll2_of(double raw, long long i) const { return (i > n - 3) ? 0.0 : raw; }
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
    lds_barrier();
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
        lds_barrier();
    }
    if (helper) {
        write_back(n_tiles - 1);
    }
}

// ---- the same sweeps with several chains per wavefront (round 3), over SEGMENTS of the rows (round 5) -----------------------
// A chain advances one locus per ~25 ns whatever its wavefront's other 63 lanes do (three dependent FP64 operations), so a
// wavefront that carries ONE chain wastes them: above, K rows keep K workgroups busy for n x 25 ns, and the 2 400 rows of a
// genome's count matrices take sum(n) x 25 ns / (workgroups in flight).  Here lane L of the chain wavefront is chain
// (parity L / G, row L % G) of a group of G rows: both parities of a row read the same input, and all 2 G chains step through
// the loci in lockstep with the same instructions as one did before.  The rows of EVERY matrix of a batch (the chromosomes of
// a genome) are groups of one launch.  G = 8: with 32 rows (all 64 lanes busy) the four helper wavefronts, each streaming 8
// far-apart rows in and out, need 39 ns per locus against the chain's 28 and the launch runs at 36; with 8 rows they stay
// under the chain and the launch runs at 27 ns per locus; 16 rows 29 ns, 4 rows 45 ns (too few lanes per LDS access).
//   * the helpers load a tile of 64 loci of each row (512 contiguous bytes per wavefront instruction) and store it
//     TRANSPOSED in LDS -- element (locus t, chain L) at t * (2 G + 1) + L: the chain wavefront's lanes read consecutive
//     doubles, a helper's 64 lanes (one row, 64 loci) write at an odd stride: no conflicts either way;
//   * the multipliers of a locus are the same for every row: one (a, b) pair per parity and locus, read as a broadcast;
//   * two input and two output tiles alternate (chain on k, staging of k + 1, write-back of k - 1), one barrier per tile.
// Round 5, once rows are cut into segments the launch fills the device and lasts what its slowest ROLE takes per tile, and
// that was the helpers (89 ms per K = 100 genome with the chain compiled out, 93 with it): the wavefronts have loops of their own
// by role -- chain | loaders of the odd tiles | loaders of the even tiles | storers (see kRowHelpers) -- and G = 16 with two
// loaders per group and two storers (7 wavefronts, 71 KB of LDS: two workgroups per CU) runs both sweeps in 80 ms, 88 with
// the residual folded in (G = 32 with 4 + 4 + 4: 89 / 97).  Still the helpers' pace: 3.4-3.7 TB/s over ~25 thousand
// row streams advancing 512 bytes per tile.
//
// Segments (round 5).  Until round 5 a launch lasted as long as its longest row: 4.98 M loci x 27 ns per sweep for a genome,
// with most of the device idle.  The substitution is a CONTRACTING recurrence (its homogeneous part decays by ~0.96 per
// locus at the count path's penalty), so a chain started from a zero state some way ahead of a locus forgets its start:
// its values approach the row's own, and -- the values being doubles -- after a while they ARE the row's own, bit for
// bit, from some locus on and for good (two consecutive values equal => everything behind them equal: the recurrence has
// no other memory).  Measured on six kinds of rows, both sweeps (scripts/ubench/coalesce.c): the two sequences meet after
// 8-20 thousand loci on average, 99 % within 55 thousand, none later than 93 thousand.  So a row is cut into segments; a
// workgroup walks `warm_tiles` tiles ahead of its segment from a zero state without writing anything, notes the state it
// has reached at the segment's start (`spec`), goes on through its segment writing results, and notes the state at the
// segment's end (`edge`).  whittaker_seam_kernel then compares, seam by seam, the state a segment started from with the
// state its predecessor ended in: EQUAL BITS PROVE the segment's values are the sequential sweep's (induction from the
// row's first segment, which starts from the reference's own start).  A seam that differs (not seen at the default
// warm-up, forced in tests/test_gpu_baseline.py) is recomputed from the true state until the recomputed values meet the
// stored ones.  Results are the sequential sweep's bits in every case; only the time depends on the data.
constexpr int kRowTile = 64;    // loci per tile
#ifndef ROCCO_GROUP_ROWS
#define ROCCO_GROUP_ROWS 16
#endif
constexpr int kGroupRows = ROCCO_GROUP_ROWS;  // rows per workgroup (x 2 parities <= 64 lanes of the chain wavefront)
constexpr int kPitch = 2 * kGroupRows + 1;
static_assert(2 * kGroupRows <= kLanes, "one lane per chain");
// Helper wavefronts come in two kinds since round 5: LOADERS (memory -> registers -> LDS) and STORERS (LDS -> memory).  A
// wavefront that has loads AND stores in flight cannot wait for "the loads of two trips ago": on this target both count on
// vmcnt and may complete out of order with respect to each other, so the compiler's wait for a loaded register is
// vmcnt(0) -- every trip then waited for the stores it had just issued, the kernels ran at the helpers' pace (2.1 us per
// tile against the chain's 1.5) and at 3.3 TB/s.  A loader has only loads in flight (precise in-order counts: the loads
// of two trips ago, the newer ones still flying), a storer only stores (it never waits for them).
#ifndef ROCCO_ROW_HELPERS
#define ROCCO_ROW_HELPERS 2
#endif
#ifndef ROCCO_ROW_STORERS
#define ROCCO_ROW_STORERS 2
#endif
constexpr int kRowHelpers = ROCCO_ROW_HELPERS;  // loader wavefronts per tile parity (two groups: odd tiles, even tiles)
constexpr int kRowStorers = ROCCO_ROW_STORERS;  // storer wavefronts
constexpr int kRowThreads = kLanes * (1 + 2 * kRowHelpers + kRowStorers);
static_assert(kGroupRows % kRowHelpers == 0 && kGroupRows % kRowStorers == 0, "rows dealt evenly to the helper wavefronts");

struct RowTiles {
    double in[2][kRowTile * kPitch];
    double out[2][kRowTile * kPitch];
    double coef[2][2][kRowTile][2];  // [buffer][parity][locus] = multipliers of the previous / one-before-previous value
    double off[kGroupRows];          // the offsets of the group's rows (0.0 where the task has none; registers are short here)
};

struct RowRegs {  // what a helper lane holds of one tile between its loads and its LDS stores
    double x0[kGroupRows / kRowHelpers], x1[kGroupRows / kRowHelpers];
    double d0, d1;        // backward: the diagonal entries of the lane's locus (z = f / d on the way into LDS)
    double ca[2], cb[2];  // wavefront 0 of the helpers: the multipliers of the lane's locus, per parity
};

// the multipliers of locus i, end cases as zeros: forward l1[i-1], l2[i-2] (baseline_backend.c:142-151), backward l1[i],
// l2[i] (158-172); Factor::ll1 / ll2 hold the entries that depend on the length
template <bool BACKWARD>
__device__ __forceinline__ void multiplier_loci(long long i, long long n, long long &i1, long long &i2)
{
    const long long ii = (i < n) ? i : (n - 1);
    i1 = BACKWARD ? ii : ((ii >= 1) ? (ii - 1) : 0);
    i2 = BACKWARD ? ii : ((ii >= 2) ? (ii - 2) : 0);
}

template <bool BACKWARD>
__device__ __forceinline__ void multipliers_from(const Factor &f, long long i, long long n, double raw1, double raw2, double &a, double &b)
{
    long long i1, i2;
    multiplier_loci<BACKWARD>(i, n, i1, i2);
    const bool has1 = BACKWARD ? (i < n - 1) : (i >= 1 && i < n), has2 = BACKWARD ? (i < n - 2) : (i >= 2 && i < n);
    const double a0 = f.ll1_of(raw1, i1), b0 = f.ll2_of(raw2, i2);
    a = has1 ? a0 : 0.0;
    b = has2 ? b0 : 0.0;
}

template <bool BACKWARD>
__device__ __forceinline__ void multipliers_at(const Factor &f, long long i, long long n, double &a, double &b)
{
    long long i1, i2;
    multiplier_loci<BACKWARD>(i, n, i1, i2);
    multipliers_from<BACKWARD>(f, i, n, f.l1[(i1 < n - 1) ? i1 : 0], f.l2[(i2 < n - 2) ? i2 : 0], a, b);
}

// forward (BACKWARD = false): src0 = the matrix (rhs = W_p y); dst0 / dst1 = f of parity 0 / 1, the forward substitution
// BEFORE its division by the diagonal (baseline_backend.c:142-156) -- the division rides on the backward sweep's staging,
// where the diagonal entry arrives with the tile's other loads instead of queueing for three tiles in registers.
// backward: src0 / src1 = f of parity 0 / 1, dst0 = the baseline 0.5 (x0 + x1) (158-172, 296-299); dst0 must be neither
// src0 nor src1 once rows are cut into segments (a neighbour's warm-up reads what this segment would overwrite).
// RESIDUAL (backward only, round 5): dst0 receives (minus - offset of the row) - baseline, the centred matrix of
// rocco/inference.py:335, instead of the baseline; a non-finite baseline raises task.bad (inference.py:207-208).  The
// forward sweep subtracts the row's offset (the pilot offset of inference.py:330-331; 0.0 when the task has none: x - 0.0 is x)
// on its way into LDS.  Both used to be passes of their own over the matrix (24 + 16 bytes per value).
template <bool BACKWARD, bool RESIDUAL = false>
__global__ __launch_bounds__(kRowThreads) void whittaker_rows_kernel(const WhittakerRowTask *__restrict__ tasks,
                                                                                  long long cap, const double *__restrict__ factor)
{
    static_assert(BACKWARD || !RESIDUAL, "the residual is formed where the baseline is");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    RowTiles &T = *reinterpret_cast<RowTiles *>(smem);
    const WhittakerRowTask task = tasks[blockIdx.x];
    // (pointers read from a record are generic to the compiler: flat loads and stores, which count on the LDS counter too
    // and force full waits; these are global)
    typedef const __attribute__((address_space(1))) double *gconst_t;
    typedef __attribute__((address_space(1))) double *gmut_t;
    const gconst_t src0 = (gconst_t)task.src0, src1 = (gconst_t)task.src1;
    const gmut_t dst0 = (gmut_t)task.dst0, dst1 = (gmut_t)task.dst1;
    const int wave = threadIdx.x / kLanes, lane = threadIdx.x % kLanes;
    const int h = (wave - 1) % kRowHelpers;  // loader number within its group
    const long long n = task.n;
    const Factor f0 = factor_of(factor, task.tail, n, cap, 0), f1 = factor_of(factor, task.tail, n, cap, 1);
    const long long all_tiles = (n + kRowTile - 1) / kRowTile;
    // the tiles this workgroup walks, in sweep order: `warm` tiles ahead of the segment, then the segment's own
    const long long first = BACKWARD ? (((task.tile_end + task.warm_tiles < all_tiles) ? (task.tile_end + task.warm_tiles) : all_tiles) - 1)
                                     : ((task.tile_begin > task.warm_tiles) ? (task.tile_begin - task.warm_tiles) : 0);
    const long long warm = BACKWARD ? (first - (task.tile_end - 1)) : (task.tile_begin - first);
    const long long n_tiles = warm + (task.tile_end - task.tile_begin);
    auto tile_base = [&](long long k) { return (BACKWARD ? (first - k) : (first + k)) * kRowTile; };
    constexpr int kMine = kGroupRows / kRowHelpers;   // rows a loader moves per tile
    constexpr int kStore = kGroupRows / kRowStorers;  // rows a storer moves per tile
    const int hs = wave - 1 - 2 * kRowHelpers;        // storer number
    const gconst_t minus = (gconst_t)task.minus;
    if (threadIdx.x < kGroupRows) {
        const int r = (int)threadIdx.x;
        T.off[r] = (task.offsets != nullptr && r < task.rows) ? task.offsets[task.row0 + r] : 0.0;
    }
    lds_barrier();

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
            R.x0[q] = src0[at];
            R.x1[q] = BACKWARD ? src1[at] : 0.0;
        }
        // (table entries as they are: the end cases are applied in stage_store, two trips later)
        if (BACKWARD) {
            R.d0 = f0.d[ii];
            R.d1 = f1.d[ii];
        }
        if (__builtin_amdgcn_readfirstlane(h) == 0) {
            long long i1, i2;
            multiplier_loci<BACKWARD>(i, n, i1, i2);
            const long long j1 = (i1 < n - 1) ? i1 : 0, j2 = (i2 < n - 2) ? i2 : 0;  // (l1 has n - 1 entries, l2 n - 2)
            R.ca[0] = f0.l1[j1];
            R.ca[1] = f1.l1[j1];
            R.cb[0] = f0.l2[j2];
            R.cb[1] = f1.l2[j2];
        }
    };
    auto stage_store = [&](long long k, const RowRegs &R, int buf) {  // (buf: k & 1 -- but for a tile staged a second time past the end)
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
                v0 = R.x0[q] / f0.dd_of(R.d0, ii);  // z = f / d (baseline_backend.c:153-156)
                v1 = R.x1[q] / f1.dd_of(R.d1, ii);
            } else {
                const double y = R.x0[q] - T.off[r];
                v0 = rhs_value(y, ii, n, 0);
                v1 = rhs_value(y, ii, n, 1);
            }
            in[lane * kPitch + r] = live ? v0 : 0.0;
            in[lane * kPitch + kGroupRows + r] = live ? v1 : 0.0;
        }
        if (__builtin_amdgcn_readfirstlane(h) == 0) {
            double a0, b0, a1, b1;
            multipliers_from<BACKWARD>(f0, i, n, R.ca[0], R.cb[0], a0, b0);
            multipliers_from<BACKWARD>(f1, i, n, R.ca[1], R.cb[1], a1, b1);
            T.coef[buf][0][lane][0] = a0;
            T.coef[buf][0][lane][1] = b0;
            T.coef[buf][1][lane][0] = a1;
            T.coef[buf][1][lane][1] = b1;
        }
    };
    // RESIDUAL: the values the baselines of tile k are subtracted from -- loaded at the start of the trip that writes the tile
    // back, used at its end (a load used at once would cost the helper a memory latency per tile)
    auto load_minus = [&](long long k, double(&m)[kStore]) {
        if (!RESIDUAL || k < warm) {
            return;
        }
        const long long i = tile_base(k) + lane;
        const long long ii = (i < n) ? i : (n - 1);
#pragma unroll
        for (int q = 0; q < kStore; ++q) {
            const int r = hs + q * kRowStorers;
            m[q] = minus[(long long)(task.row0 + ((r < task.rows) ? r : 0)) * n + ii];
        }
    };
    auto write_back = [&](long long k, const double(&m)[kStore]) {
        if (k < warm) {
            return;  // (ahead of the segment: another workgroup's loci)
        }
        const double *__restrict__ out = T.out[k & 1];
        const long long i = tile_base(k) + lane;
        if (i >= n) {
            return;
        }
        if (RESIDUAL) {
            // every result first, then every store: with stores and the loads of `m` in flight together a wait for a loaded
            // value is a wait for everything (they complete out of order with respect to each other) -- one such wait per
            // tile, for operations a whole tile old, instead of one in front of every store
            double res[kStore];
            bool bad = false;
#pragma unroll
            for (int q = 0; q < kStore; ++q) {
                const int r = hs + q * kRowStorers, rr = (r < task.rows) ? r : 0;
                const double base = 0.5 * (out[lane * kPitch + rr] + out[lane * kPitch + kGroupRows + rr]);
                bad = bad || (r < task.rows && !isfinite(base));
                res[q] = (m[q] - T.off[rr]) - base;
            }
#pragma unroll
            for (int q = 0; q < kStore; ++q) {
                asm volatile("" : "+v"(res[q]));  // (computed HERE: the compiler otherwise sinks each result into its store's branch, wait and all)
            }
#pragma unroll
            for (int q = 0; q < kStore; ++q) {
                const int r = hs + q * kRowStorers;
                if (r < task.rows) {
                    dst0[(long long)(task.row0 + r) * n + i] = res[q];
                }
            }
            if (bad) {
                atomicOr(task.bad, 1);
            }
            return;
        }
        for (int r = hs; r < task.rows; r += kRowStorers) {
            const long long at = (long long)(task.row0 + r) * n + i;
            if (BACKWARD) {
                dst0[at] = 0.5 * (out[lane * kPitch + r] + out[lane * kPitch + kGroupRows + r]);
            } else {
                dst0[at] = out[lane * kPitch + r];
                dst1[at] = out[lane * kPitch + kGroupRows + r];
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
    // The chain wavefront and the helpers walk the tiles in loops of their own (round 5: one loop with a branch per trip kept
    // the helpers' two register sets alive across the chain's code -- 254 registers, none left for the residual form); they
    // meet at one barrier per tile.  The branch is on a scalar (readfirstlane): every wavefront runs one of the loops only,
    // and both pass 1 + n_tiles barriers.
    // One trip: the chain wavefront runs tile k; the helpers carry tile k - 1 to memory, store tile k + 1 (in `set` since two
    // trips) and refill `set` with tile k + 3
#ifdef ROCCO_ROWS_NOCHAIN  // (timing experiments only)
    constexpr bool kRunChain = false;
#else
    constexpr bool kRunChain = true;
#endif
    constexpr bool kRunHelpers = true;  // (never compiled out: without their values every seam differs and is recomputed in full -- minutes)
    if (__builtin_amdgcn_readfirstlane(wave) == 0) {
        __builtin_amdgcn_s_setprio(3);  // the chain wavefront shares its SIMD with a helper: its instructions go first
        lds_barrier();
        for (long long k = 0; k < n_tiles; ++k) {
            if (kRunChain) {
                if (k == warm && warm > 0 && chain_lane) {
                    // the state the warm-up has reached where the segment starts
                    task.spec[2 * col] = p1;
                    task.spec[2 * col + 1] = p2;
                }
                chain(k);
            }
            lds_barrier();
        }
        if (chain_lane && task.edge != nullptr) {
            task.edge[2 * col] = p1;  // the state at the segment's far end: what the next segment of the sweep must start from
            task.edge[2 * col + 1] = p2;
        }
    } else if (__builtin_amdgcn_readfirstlane(wave) <= 2 * kRowHelpers) {
        // Loaders, two groups: one serves the odd tiles, one the even ones.  A loader works every other trip -- tile k + 1 (in
        // its registers since two trips) into LDS, the registers refilled with tile k + 3 -- and has ONE set of loads in
        // flight, two trips to land.  (One wavefront alternating two sets had the compiler drain every load in flight once
        // per trip: where paths join it counts the loads in flight conservatively, and it reuses a set's dead registers --
        // a write-after-write wait on everything newer.  Here a full drain is exact: nothing newer is in flight.)
        // Past the end the last tile is staged again, into the buffer nobody reads.
        RowRegs R;
        const long long last = n_tiles - 1;
        const bool even_tiles = __builtin_amdgcn_readfirstlane(wave) > kRowHelpers;
        auto work = [&](long long k) {
            stage_store((k + 1 < last) ? (k + 1) : last, R, (int)((k + 1) & 1));
            stage_load((k + 3 < last) ? (k + 3) : last, R);
        };
        if (even_tiles) {
            stage_load(0, R);
            stage_store(0, R, 0);
            stage_load((2 < last) ? 2 : last, R);
            lds_barrier();
            for (long long k = 0; k < n_tiles; k += 2) {
                lds_barrier();  // (trip k: the other group's)
                if (k + 1 < n_tiles) {
                    work(k + 1);
                    lds_barrier();
                }
            }
        } else {
            stage_load((1 < last) ? 1 : last, R);
            lds_barrier();
            for (long long k = 0; k < n_tiles; k += 2) {
                work(k);
                lds_barrier();
                if (k + 1 < n_tiles) {
                    lds_barrier();  // (trip k + 1: the other group's)
                }
            }
        }
    } else {
        // storers: tile k - 1 to memory.  RESIDUAL: the values its baselines are subtracted from were loaded a trip earlier
        // (the one kind of load a storer has; its wait for them is a wait for the stores of the trip before too -- both a
        // whole tile old by then)
        double minus_set[kStore] = {};
        lds_barrier();
        load_minus(0, minus_set);
        lds_barrier();
        for (long long k = 1; k < n_tiles; ++k) {
            if (kRunHelpers) {
                write_back(k - 1, minus_set);
                load_minus(k, minus_set);
            }
            lds_barrier();
        }
        write_back(n_tiles - 1, minus_set);
    }
}

// ---- the seams between segments ---------------------------------------------------------------------------------------------
// One wavefront per 64 rows of a matrix walks the rows' seams in sweep order, lane = row.  At each seam and for each
// parity: the state the following segment started from (`spec`, reached by its warm-up) against the state the segment
// before it ended in (`edge`).  Equal bits: the following segment's stored values are the sequential sweep's (see above).
// Otherwise that segment is recomputed from the true state, side by side with the sequence the workgroup had computed
// (restarted from `spec`), storing the true values until both sequences give the same two values in a row -- from there
// on the stored values are right; a segment recomputed to its end hands its true end state to the next seam's comparison.
// A recomputation is one lane's sequential work; the wavefront's other lanes fetch its operands -- 64 loci at a time,
// through LDS -- and carry its results to memory.
__device__ __forceinline__ bool same_bits(double x, double y) { return __double_as_longlong(x) == __double_as_longlong(y); }

template <bool BACKWARD>
__global__ __launch_bounds__(kLanes) void whittaker_seam_kernel(const WhittakerSeamMatrix *__restrict__ matrices, double *__restrict__ states,
                                                                long long tasks_per_sweep, long long cap, const double *__restrict__ factor,
                                                                unsigned long long *__restrict__ repairs)
{
    __shared__ double s_v[2][kRowTile], s_a[2][kRowTile], s_b[2][kRowTile], s_out[2][kRowTile];
    const WhittakerSeamMatrix m = matrices[blockIdx.y];
    if (m.n_seg < 2) {
        return;
    }
    const int lane = threadIdx.x;
    const int row = (int)(blockIdx.x * kLanes + threadIdx.x);
    const bool valid = row < m.rows;
    const long long n = m.n, all_tiles = (n + kRowTile - 1) / kRowTile;
    const Factor f[2] = {factor_of(factor, m.tail, n, cap, 0), factor_of(factor, m.tail, n, cap, 1)};
    const int my = valid ? row : 0, group = my / m.group_rows, r = my % m.group_rows;
    // states: [sweep][task][spec | edge][chain][2]; chain = parity * G + row of the group
    double *const sweep_states = states + (BACKWARD ? tasks_per_sweep : 0) * (long long)(8 * kGroupRows);
    auto spec_of = [&](int seg, int parity) { return sweep_states + ((m.task_base + (long long)group * m.n_seg + seg) * 2 + 0) * (4 * kGroupRows) + 2 * (parity * kGroupRows + r); };
    auto edge_of = [&](int seg, int parity) { return sweep_states + ((m.task_base + (long long)group * m.n_seg + seg) * 2 + 1) * (4 * kGroupRows) + 2 * (parity * kGroupRows + r); };
    for (int step = 1; step < m.n_seg; ++step) {
        // forward: the seam below segment `step`; backward: the seam above segment n_seg - 1 - step
        const int seg = BACKWARD ? (m.n_seg - 1 - step) : step, before = BACKWARD ? (seg + 1) : (seg - 1);
        double tp1[2], tp2[2], sp1[2], sp2[2];
        int open[2];
        for (int p = 0; p < 2; ++p) {
            tp1[p] = edge_of(before, p)[0];
            tp2[p] = edge_of(before, p)[1];
            sp1[p] = spec_of(seg, p)[0];
            sp2[p] = spec_of(seg, p)[1];
            open[p] = (valid && !(same_bits(tp1[p], sp1[p]) && same_bits(tp2[p], sp2[p]))) ? 1 : 0;
        }
        unsigned long long todo = __ballot((open[0] | open[1]) != 0);
        if (todo == 0ULL) {
            continue;
        }
        const long long t_lo = (long long)seg * m.seg_tiles, t_hi = (seg + 1 == m.n_seg) ? all_tiles : (t_lo + m.seg_tiles);
        const long long lo = t_lo * kRowTile, hi = (t_hi * kRowTile < n) ? (t_hi * kRowTile) : n;  // the segment's loci [lo, hi)
        while (todo != 0ULL) {
            const int owner = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1ULL;
            const int orow = __shfl(row, owner);
            const bool oopen[2] = {__shfl(open[0], owner) != 0, __shfl(open[1], owner) != 0};
            // backward writes 0.5 (x0 + x1): an open parity needs the other parity's values too, so both are walked there
            const bool walk[2] = {oopen[0] || (BACKWARD && oopen[1]), oopen[1] || (BACKWARD && oopen[0])};
            const long long at_row = (long long)orow * n;
            const double off_o = (m.offsets != nullptr) ? m.offsets[orow] : 0.0;  // (the sweeps' input is src - offset of the row)
            int met[2] = {oopen[0] ? 0 : 2, oopen[1] ? 0 : 2};  // consecutive loci at which the two sequences agreed (the owner's count)
            long long walked = 0;
            bool done = false;
            for (long long base = 0; base < hi - lo && !done; base += kRowTile) {
                // the tile's loci in sweep order: lane j holds the j-th
                const long long i = BACKWARD ? (hi - 1 - base - lane) : (lo + base + lane);
                const bool inside = BACKWARD ? (i >= lo) : (i < hi);
                const long long ii = inside ? i : (BACKWARD ? lo : (hi - 1));
                for (int p = 0; p < 2; ++p) {
                    if (!walk[p]) {
                        continue;
                    }
                    double a, b, v;
                    multipliers_at<BACKWARD>(f[p], ii, n, a, b);
                    if (BACKWARD) {
                        v = (p == 0 ? m.src0 : m.src1)[at_row + ii] / f[p].dd(ii);
                    } else {
                        v = rhs_value(m.src0[at_row + ii] - off_o, ii, n, p);
                    }
                    s_v[p][lane] = v;
                    s_a[p][lane] = a;
                    s_b[p][lane] = b;
                }
                __syncthreads();
                const long long left = hi - lo - base;
                const int count = (left < kRowTile) ? (int)left : kRowTile;
                int upto = count;
                if (lane == owner) {
                    for (int j = 0; j < count; ++j) {
                        for (int p = 0; p < 2; ++p) {
                            if (!walk[p]) {
                                continue;
                            }
                            const double v = s_v[p][j], a = s_a[p][j], b = s_b[p][j];
                            const double tr = chain_step(v, a, b, tp1[p], tp2[p]);
                            tp2[p] = tp1[p];
                            tp1[p] = tr;
                            s_out[p][j] = tr;
                            if (oopen[p]) {
                                const double sr = chain_step(v, a, b, sp1[p], sp2[p]);
                                sp2[p] = sp1[p];
                                sp1[p] = sr;
                                met[p] = (met[p] >= 2) ? 2 : (same_bits(sr, tr) ? (met[p] + 1) : 0);
                            }
                        }
                        if (met[0] >= 2 && met[1] >= 2) {
                            upto = j + 1;
                            break;
                        }
                    }
                }
                __syncthreads();
                // (the same in every lane, and said so: a loop exit the compiler takes for divergent must not hold barriers)
                upto = __builtin_amdgcn_readfirstlane(__shfl(upto, owner));
                done = __builtin_amdgcn_readfirstlane(__shfl((met[0] >= 2 && met[1] >= 2) ? 1 : 0, owner)) != 0;
                if (lane < upto) {
                    if (BACKWARD) {
                        const double base = 0.5 * (s_out[0][lane] + s_out[1][lane]);
                        if (m.minus != nullptr) {  // (the residual form: what whittaker_rows_kernel<true, true> writes)
                            if (!isfinite(base)) {
                                atomicOr(m.bad, 1);
                            }
                            m.dst0[at_row + i] = (m.minus[at_row + i] - off_o) - base;
                        } else {
                            m.dst0[at_row + i] = base;
                        }
                    } else {
                        if (oopen[0]) {
                            m.dst0[at_row + i] = s_out[0][lane];
                        }
                        if (oopen[1]) {
                            m.dst1[at_row + i] = s_out[1][lane];
                        }
                    }
                }
                walked += upto;
                __syncthreads();
            }
            if (lane == owner) {
                atomicAdd(repairs, (unsigned long long)((oopen[0] ? 1 : 0) + (oopen[1] ? 1 : 0)));
                atomicAdd(repairs + 1, (unsigned long long)walked);  // loci recomputed
                for (int p = 0; p < 2; ++p) {
                    if (oopen[p] && met[p] < 2) {
                        // recomputed to its end without meeting: its true end state for the next seam
                        edge_of(seg, p)[0] = tp1[p];
                        edge_of(seg, p)[1] = tp2[p];
                    }
                }
            }
        }
    }
}

// out = matrix - offset of the row (rows too short for a baseline: the baseline is zero, baseline_backend.c:265-272)
__global__ void offset_rows_kernel(const double *matrix, const double *offsets, long long n, long long count, double *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        out[i] = (matrix[i] - ((offsets != nullptr) ? offsets[i / n] : 0.0)) - 0.0;
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

// workgroups of the rows kernels the device holds at once (registers and LDS decide how many per CU): the segment plan
// cuts a batch into about that many
std::atomic<long long> g_rows_resident{0};

int configure_rows_kernels()
{
    static bool attr_set = false;
    if (!attr_set) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_rows_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RowTiles)));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_rows_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RowTiles)));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(whittaker_rows_kernel<true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RowTiles)));
        int per_cu[2] = {1, 1}, device = 0, cus = 256;
        ROCCO_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[0], reinterpret_cast<const void *>(whittaker_rows_kernel<false>),
                                                                   kRowThreads, sizeof(RowTiles)));
        ROCCO_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu[1], reinterpret_cast<const void *>(whittaker_rows_kernel<true>),
                                                                   kRowThreads, sizeof(RowTiles)));
        ROCCO_HIP_TRY(hipGetDevice(&device));
        ROCCO_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        g_rows_resident.store((long long)std::max(1, std::min(per_cu[0], per_cu[1])) * (long long)std::max(1, cus));
        attr_set = true;
    }
    return ROCCO_HIP_OK;
}

}  // namespace

int whittaker_group_rows() { return kGroupRows; }

namespace {

std::atomic<long long> g_seam_repairs{0};

long long env_loci(const char *name, long long fallback)
{
    const char *v = std::getenv(name);
    return (v != nullptr && *v != '\0') ? std::atoll(v) : fallback;
}

// How a batch's rows are cut.  A workgroup's time is (warm-up + segment) x the chain's step, so a row is cut into as many
// segments as keep every workgroup of the batch resident at once (resident_workgroups()) -- but none shorter than the warm-up, which
// would then be most of the work.  ROCCO_HIP_WHITTAKER_SEGMENT_LOCI: a fixed segment length (0: rows are never cut);
// ROCCO_HIP_WHITTAKER_WARM_LOCI: the warm-up.  Measured on a genome's 66 thousand seams (K = 100, both sweeps): 2.5 % had
// not met after 65 536 loci, 0.045 % after 131 072 (an e-fold per ~16 thousand) -- the default of 196 608 leaves about
// one seam in a hundred thousand to be recomputed (whittaker_seam_kernel).
constexpr long long kWarmLoci = 196608;

long long resident_workgroups()
{
    const long long forced = env_loci("ROCCO_HIP_WHITTAKER_RESIDENT", 0);
    const long long known = g_rows_resident.load();
    return (forced > 0) ? forced : ((known > 0) ? known : 256);
}

struct SegmentPlan {
    long long seg_tiles = 0;  // tiles per segment
    int n_seg = 1;
    long long warm_tiles = 0;
};

// rows of a matrix are dealt to ceil(rows / G) workgroups evenly (100 rows, G = 32: four groups of 25, not 32 + 32 + 32 + 4:
// a workgroup's helpers move its rows, so even groups finish together)
size_t groups_of(size_t rows) { return (rows + kGroupRows - 1) / kGroupRows; }
size_t group_rows_of(size_t rows) { return (rows + groups_of(rows) - 1) / std::max<size_t>(1, groups_of(rows)); }

long long batch_group_loci(const size_t *rows, const size_t *cols, size_t count)
{
    long long total = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] > 0 && cols[i] >= 25) {
            total += (long long)groups_of(rows[i]) * (long long)cols[i];
        }
    }
    return total;
}

long long warm_loci() { return std::max<long long>(kRowTile, env_loci("ROCCO_HIP_WHITTAKER_WARM_LOCI", kWarmLoci)); }

// a row of `cols` loci cut into `pieces` segments of whole tiles
SegmentPlan cut(size_t cols, long long pieces)
{
    SegmentPlan plan;
    const long long all_tiles = ((long long)cols + kRowTile - 1) / kRowTile;
    plan.warm_tiles = (warm_loci() + kRowTile - 1) / kRowTile;
    plan.seg_tiles = all_tiles;
    if (pieces > 1) {
        plan.seg_tiles = (all_tiles + pieces - 1) / pieces;
        plan.n_seg = (int)((all_tiles + plan.seg_tiles - 1) / plan.seg_tiles);
    }
    if (plan.n_seg < 2) {
        plan.n_seg = 1;
        plan.seg_tiles = all_tiles;
    }
    return plan;
}

// The plan of a batch is ONE number, the loci a workgroup may walk (`budget`): a row no longer than that stays whole, a
// longer one is cut into the fewest pieces whose length + warm-up fit (a piece never shorter than the warm-up).  A launch
// lasts (workgroups / resident, rounded up) x budget chain steps: among one, two and three rounds of resident workgroups
// the smallest such product wins.  ROCCO_HIP_WHITTAKER_SEGMENT_LOCI fixes the piece length instead (0: rows are never cut).
long long pieces_of(size_t cols, long long budget)
{
    const long long fixed = env_loci("ROCCO_HIP_WHITTAKER_SEGMENT_LOCI", -1), warm = warm_loci(), n = (long long)cols;
    if (fixed >= 0) {
        return (fixed > 0 && n >= 2 * fixed) ? ((n + fixed / 2) / fixed) : 1;
    }
    if (budget <= 0 || n <= budget) {
        return 1;
    }
    const long long room = std::max(warm, budget - warm);
    return std::max<long long>(1, std::min((n + room - 1) / room, n / warm));
}

long long batch_budget_loci(const size_t *rows, const size_t *cols, size_t count)
{
    if (env_loci("ROCCO_HIP_WHITTAKER_SEGMENT_LOCI", -1) >= 0) {
        return 0;
    }
    const long long resident = resident_workgroups(), warm = warm_loci();
    long long longest = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] > 0 && cols[i] >= 25) {
            longest = std::max(longest, (long long)cols[i]);
        }
    }
    auto tasks_at = [&](long long budget) {
        long long tasks = 0;
        for (size_t i = 0; i < count; ++i) {
            if (rows[i] > 0 && cols[i] >= 25) {
                tasks += (long long)groups_of(rows[i]) * pieces_of(cols[i], budget);
            }
        }
        return tasks;
    };
    long long best_budget = longest, best_cost = ((tasks_at(longest) + resident - 1) / resident) * longest;
    for (long long rounds = 1; rounds <= 3; ++rounds) {
        long long lo = 2 * warm, hi = longest;  // the smallest budget whose workgroups fit `rounds` rounds
        if (lo >= hi || tasks_at(hi) > rounds * resident) {
            continue;
        }
        while (hi - lo > kRowTile) {
            const long long mid = lo + (hi - lo) / 2;
            if (tasks_at(mid) <= rounds * resident) {
                hi = mid;
            } else {
                lo = mid;
            }
        }
        const long long cost = ((tasks_at(hi) + resident - 1) / resident) * hi;
        if (cost < best_cost) {
            best_cost = cost;
            best_budget = hi;
        }
    }
    return best_budget;
}

SegmentPlan plan_segments(size_t cols, long long budget) { return cut(cols, pieces_of(cols, budget)); }

size_t batch_tasks(const size_t *rows, const size_t *cols, size_t count, long long budget)
{
    size_t n_tasks = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] > 0 && cols[i] >= 25) {
            n_tasks += groups_of(rows[i]) * (size_t)plan_segments(cols[i], budget).n_seg;
        }
    }
    return n_tasks;
}

// an upper bound of batch_tasks() that does not depend on what else is in the batch (a solver's scratch is sized before the
// batches are formed): no plan cuts a row into more than cols / (shortest segment) + 1 pieces
size_t batch_tasks_bound(const size_t *rows, const size_t *cols, size_t count)
{
    const long long warm = warm_loci();
    const long long fixed = env_loci("ROCCO_HIP_WHITTAKER_SEGMENT_LOCI", -1);
    const long long shortest = (fixed < 0) ? warm : std::max<long long>(fixed / 2, 1);
    size_t n_tasks = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] > 0 && cols[i] >= 25) {
            const size_t pieces = (fixed == 0) ? 1 : (size_t)((long long)cols[i] / shortest + 1);
            n_tasks += groups_of(rows[i]) * pieces;
        }
    }
    return n_tasks;
}

constexpr size_t kStateDoubles = 8 * kGroupRows;  // per task and sweep: spec | edge, 2 G chains x 2 values each

size_t round256(size_t bytes) { return (bytes + 255) / 256 * 256; }

}  // namespace

long long whittaker_seam_repairs() { return g_seam_repairs.load(std::memory_order_relaxed); }

// what launch_crossfit_whittaker_batch copies up from pinned memory: both sweeps' task records, the seam records, and
// (coming back) the count of recomputed seams in the first 64 bytes
size_t whittaker_batch_stage_bytes(const size_t *rows, const size_t *cols, size_t count)
{
    return 64 + round256(2 * batch_tasks_bound(rows, cols, count) * sizeof(WhittakerRowTask)) + round256(2 * count * sizeof(WhittakerSeamMatrix)) + 256;
}

size_t whittaker_batch_scratch_bytes(const size_t *rows, const size_t *cols, size_t count)
{
    // the staged records, the seam states of both sweeps, a counter; per matrix: the forward sweep's f of both parities
    // (2 x rows x cols doubles) and 8 doubles of end entries
    const size_t n_tasks = batch_tasks_bound(rows, cols, count);
    size_t bytes = whittaker_batch_stage_bytes(rows, cols, count) + round256(2 * n_tasks * kStateDoubles * sizeof(double)) + 512;
    for (size_t i = 0; i < count; ++i) {
        bytes += round256((2 * rows[i] * cols[i] + 8) * sizeof(double));
    }
    return bytes;
}

int launch_crossfit_whittaker_batch(const double *const *matrices_dev, const size_t *rows, const size_t *cols, size_t count,
                                    double penalty_lambda, const double *factor_dev, size_t factor_cap, double *const *baselines_dev,
                                    void *scratch_dev, void *tasks_host_pinned, hipStream_t stream,
                                    const double *const *offsets_dev, int residual)
{
    int rc;
    if ((rc = configure_rows_kernels()) != ROCCO_HIP_OK) return rc;
    std::memset(tasks_host_pinned, 0, 64);  // (the counters whittaker_collect_repairs reads, whichever way this call goes)
    const long long seg_loci = batch_budget_loci(rows, cols, count);
    auto offsets_of = [&](size_t i) { return (offsets_dev != nullptr) ? offsets_dev[i] : (const double *)nullptr; };
    if (residual == 0 && offsets_dev == nullptr && count == 1 && rows[0] > 0 && cols[0] >= 25 && plan_segments(cols[0], seg_loci).n_seg < 2) {
        // ONE matrix of rows too short to cut: a workgroup per row with one chain per wavefront steps ~25 ns per locus, the
        // grouped wavefronts below ~27 ns -- both last as long as one row, so the lone matrix takes the faster step
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
    const size_t n_tasks = batch_tasks(rows, cols, count, seg_loci);
    // scratch: [counter | tasks forward | tasks backward | seam records forward | backward] [states] then per matrix
    // [tail (8 doubles) | f of parity 0 | f of parity 1]; the pinned staging area mirrors the first part
    const size_t stage_bytes = 64 + round256(2 * n_tasks * sizeof(WhittakerRowTask)) + round256(2 * count * sizeof(WhittakerSeamMatrix));
    char *at = (char *)scratch_dev;
    unsigned long long *repairs_dev = (unsigned long long *)at;
    int *bad_dev = (int *)(repairs_dev + 2);  // (the counter block's third word)
    WhittakerRowTask *tasks_dev = (WhittakerRowTask *)(at + 64);
    WhittakerSeamMatrix *seams_dev = (WhittakerSeamMatrix *)(at + 64 + round256(2 * n_tasks * sizeof(WhittakerRowTask)));
    at += round256(stage_bytes);
    double *states_dev = (double *)at;
    at += round256(2 * n_tasks * kStateDoubles * sizeof(double));
    char *stage = (char *)tasks_host_pinned;
    std::memset(stage, 0, 64);
    WhittakerRowTask *tasks_host = (WhittakerRowTask *)(stage + 64);
    WhittakerSeamMatrix *seams_host = (WhittakerSeamMatrix *)(stage + 64 + round256(2 * n_tasks * sizeof(WhittakerRowTask)));
    size_t t = 0, n_seams = 0, most_rows = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] == 0 || cols[i] == 0) {
            continue;
        }
        if (cols[i] < 25) {  // baseline_backend.c:265-272
            const long long total = (long long)(rows[i] * cols[i]);
            if (residual != 0) {
                hipLaunchKernelGGL(offset_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, matrices_dev[i], offsets_of(i),
                                   (long long)cols[i], total, baselines_dev[i]);
            } else {
                hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, baselines_dev[i], total);
            }
            continue;
        }
        if (factor_dev == nullptr || factor_cap < cols[i]) {
            return ROCCO_HIP_EINVAL;
        }
        double *tail = (double *)at;
        double *z0 = tail + 8;
        double *z1 = z0 + rows[i] * cols[i];
        at += round256((2 * rows[i] * cols[i] + 8) * sizeof(double));
        hipLaunchKernelGGL(whittaker_tail_kernel, dim3(1), dim3(64), 0, stream, (long long)cols[i], (long long)factor_cap, penalty_lambda,
                           factor_dev, tail);
        const SegmentPlan plan = plan_segments(cols[i], seg_loci);
        const size_t group_rows = group_rows_of(rows[i]);
        const long long all_tiles = ((long long)cols[i] + kRowTile - 1) / kRowTile;
        if (plan.n_seg > 1) {
            WhittakerSeamMatrix &fw = seams_host[n_seams], &bw = seams_host[count + n_seams];
            fw.src0 = matrices_dev[i];
            fw.src1 = nullptr;
            fw.dst0 = z0;
            fw.dst1 = z1;
            fw.n = (long long)cols[i];
            fw.rows = (int)rows[i];
            fw.n_seg = plan.n_seg;
            fw.group_rows = (int)group_rows;
            fw.seg_tiles = plan.seg_tiles;
            fw.task_base = (long long)t;
            fw.tail = tail;
            fw.offsets = offsets_of(i);
            fw.minus = nullptr;
            fw.bad = bad_dev;
            bw = fw;
            bw.minus = (residual != 0) ? matrices_dev[i] : nullptr;
            bw.src0 = z0;
            bw.src1 = z1;
            bw.dst0 = baselines_dev[i];
            bw.dst1 = nullptr;
            ++n_seams;
            most_rows = std::max(most_rows, rows[i]);
        }
        for (size_t r0 = 0; r0 < rows[i]; r0 += group_rows) {
            for (int seg = 0; seg < plan.n_seg; ++seg, ++t) {
                WhittakerRowTask &fw = tasks_host[t], &bw = tasks_host[n_tasks + t];
                fw.src0 = matrices_dev[i];
                fw.src1 = nullptr;
                fw.dst0 = z0;
                fw.dst1 = z1;
                fw.n = (long long)cols[i];
                fw.row0 = (int)r0;
                fw.rows = (int)std::min<size_t>(group_rows, rows[i] - r0);
                fw.tail = tail;
                fw.tile_begin = (long long)seg * plan.seg_tiles;
                fw.tile_end = (seg + 1 == plan.n_seg) ? all_tiles : (fw.tile_begin + plan.seg_tiles);
                fw.warm_tiles = (seg > 0) ? plan.warm_tiles : 0;
                fw.spec = states_dev + (2 * t + 0) * (kStateDoubles / 2);
                fw.edge = states_dev + (2 * t + 1) * (kStateDoubles / 2);
                fw.offsets = offsets_of(i);
                fw.minus = nullptr;
                fw.bad = bad_dev;
                bw = fw;
                bw.minus = (residual != 0) ? matrices_dev[i] : nullptr;
                bw.src0 = z0;
                bw.src1 = z1;
                bw.dst0 = baselines_dev[i];
                bw.dst1 = nullptr;
                bw.warm_tiles = (seg + 1 < plan.n_seg) ? plan.warm_tiles : 0;
                bw.spec = states_dev + (2 * (n_tasks + t) + 0) * (kStateDoubles / 2);
                bw.edge = states_dev + (2 * (n_tasks + t) + 1) * (kStateDoubles / 2);
            }
        }
    }
    if (n_tasks == 0 || t == 0) {
        ROCCO_HIP_TRY(hipGetLastError());
        return ROCCO_HIP_OK;
    }
    if (env_loci("ROCCO_HIP_WHITTAKER_TRACE", 0) != 0) {
        std::fprintf(stderr, "[whittaker] %zu matrices: a workgroup walks <= %lld loci, warm-up %lld, %zu workgroups of <= %d rows (%lld resident), %zu matrices cut\n",
                     count, seg_loci, env_loci("ROCCO_HIP_WHITTAKER_WARM_LOCI", kWarmLoci), t, kGroupRows, resident_workgroups(), n_seams);
    }
    ROCCO_HIP_TRY(hipMemcpyAsync(scratch_dev, stage, stage_bytes, hipMemcpyHostToDevice, stream));
    const dim3 block(kRowThreads);
    const dim3 seam_grid((unsigned)((most_rows + kLanes - 1) / kLanes), (unsigned)n_seams);
    hipLaunchKernelGGL(whittaker_rows_kernel<false>, dim3((unsigned)t), block, sizeof(RowTiles), stream,  // (t <= n_tasks records are filled)
                       (const WhittakerRowTask *)tasks_dev, (long long)factor_cap, factor_dev);
    if (n_seams > 0) {
        hipLaunchKernelGGL(whittaker_seam_kernel<false>, seam_grid, dim3(kLanes), 0, stream, (const WhittakerSeamMatrix *)seams_dev,
                           states_dev, (long long)n_tasks, (long long)factor_cap, factor_dev, repairs_dev);
    }
    if (residual != 0) {
        hipLaunchKernelGGL((whittaker_rows_kernel<true, true>), dim3((unsigned)t), block, sizeof(RowTiles), stream,
                           (const WhittakerRowTask *)(tasks_dev + n_tasks), (long long)factor_cap, factor_dev);
    } else {
        hipLaunchKernelGGL((whittaker_rows_kernel<true, false>), dim3((unsigned)t), block, sizeof(RowTiles), stream,
                           (const WhittakerRowTask *)(tasks_dev + n_tasks), (long long)factor_cap, factor_dev);
    }
    if (n_seams > 0) {
        hipLaunchKernelGGL(whittaker_seam_kernel<true>, seam_grid, dim3(kLanes), 0, stream, (const WhittakerSeamMatrix *)(seams_dev + count),
                           states_dev, (long long)n_tasks, (long long)factor_cap, factor_dev, repairs_dev);
    }
    if (n_seams > 0 || residual != 0) {
        // the count of recomputed seams (and the residual form's flag) come back into the staging area's first words (read
        // by whittaker_collect_repairs once the caller has waited for the stream)
        ROCCO_HIP_TRY(hipMemcpyAsync(stage, repairs_dev, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int whittaker_collect_repairs(const void *tasks_host_pinned)
{
    const unsigned long long *back = (const unsigned long long *)tasks_host_pinned;
    g_seam_repairs.fetch_add((long long)back[0], std::memory_order_relaxed);
    if (back[0] != 0 && env_loci("ROCCO_HIP_WHITTAKER_TRACE", 0) != 0) {
        std::fprintf(stderr, "[whittaker] %llu seams recomputed over %llu loci\n", back[0], back[1]);
    }
    return (back[2] != 0ULL) ? 1 : 0;
}

}  // namespace rocco
