// rocco_amd/csrc/wls.hip -- centred-WLS scoring of a K x n matrix, gfx950.
//
// Replaces rocco_score_centered_wls_f64 (rocco/native/wls_backend.c:744-947; wrapper rocco/_wls.c; caller
// rocco/inference.py:231-299).  Results equal the reference's bit for bit on finite inputs, so every
// order-dependent sum keeps the reference's order:
//   * rolling AR(1) innovation variance (610-742): the three running sums are updated by subtract-then-add
//     along the row -- one lane per running sum, eight rows per workgroup, tiles of 64 start positions through
//     LDS -- and helper wavefronts turn the sums of a tile into variances (the divisions are not part of the
//     dependent chain);
//   * monotone variance trend (394-608): the reference sorts the (|value|, variance) pairs of a row with
//     qsort under a total order (x, then y), so the sorted sequence is unique: two stable device radix
//     sorts (by y, then by x) give the same sequence; bin medians are order statistics, taken from a
//     segmented sort of each bin's variances; pooling (262-339) and knots run in one thread per row;
//   * rows are accumulated into the per-locus precision sums in row order (858-910), one launch per row.
#include "kernels.h"

#include <vector>
#include "log2_cr.h"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdlib>

namespace rocco {

namespace {

constexpr int kLanes = 64;
constexpr int kMaxWindow = 63;  // the rolling kernel's tiles hold windows up to here
constexpr int kMaxBins = 64;
constexpr long long kTrendSelectMin = 4096;  // rows at least this long fit their trend without sorting the pairs

// ---- rolling AR(1) innovation variance (wls_backend.c:610-742) --------------------------------------
// vas[row][s] for s = 0 .. max_start (the caller indexes it with clamp(i - half, 0, max_start), 727-738).
// The three running sums of the reference are independent chains  s <- (s - P[t]) + P[t + off]  over the streams
// P0 = v, P1 = v*v, P2 = v[i]*v[i+1]  (off = window, window, window - 1): two dependent additions per start position, in
// the order of the reference (711-722); the variances (667-709; their divisions are off the chains) follow from the sums.
// G rows per workgroup (round 3; G = 8, 4, 2 or 1): lane L < 3 G of the chain wavefront runs chain L % 3 of row L / 3 -- the
// chains of a wavefront cost what one costs (two dependent FP64 additions per start position whatever the lane count), so
// one row's time per locus serves G rows -- over a ring of 4 TILE lines in LDS, line t mod 4 TILE holding (v, v^2, v v_next)
// of locus t for the G rows at 3 r + c.  Eight helper wavefronts (512 lanes = G rows x TILE loci of a chunk) store the chunk
// two tiles ahead, load the one after it, and turn the previous tile's sums into variances (their long dependent FP64
// sequences -- three divisions per variance -- overlap across the wavefronts); one barrier per tile of TILE = 512 / G
// start positions.  The launch lasts what its longest row lasts as long as every workgroup is resident (two per CU by LDS
// for G >= 4, one for G <= 2), so G is the smallest that makes the rows of a call fit: a genome's 2 400 rows take G = 8 and
// 23 ns per locus of the longest row, one matrix of 100 rows G = 1 and a tile of 512 (fewer barriers per locus).
constexpr int kRowsHelpers = 8 * kLanes;

template <int G>
struct RollingShape {
    static constexpr int kTile = kRowsHelpers / G;        // start positions per tile = loci per staged chunk
    static constexpr int kRing = 4 * kTile;               // lines: the chunks of the tiles k - 1 .. k + 2
    static constexpr int kPitch = (3 * G) % 2 ? 3 * G + 2 : 3 * G + 1;  // doubles per line, odd: lane = locus accesses spread over the banks
    static_assert(kTile >= kMaxWindow + 1, "a tile's chains read at most one chunk ahead");
    static_assert(kPitch % 2 == 1, "odd pitch");
};

template <int G>
struct RollingRows {
    double S[2][RollingShape<G>::kTile][RollingShape<G>::kPitch];  // (first: within reach of immediate offsets in every shape)
    // + a copy of the ring's first 8 lines behind its last one: a batch of 8 consecutive lines that starts inside the ring
    // never needs its line indices masked one by one (round 5; until then every access of every fourth tile was masked:
    // three integer instructions per operand on the chain wavefront)
    double P[RollingShape<G>::kRing + 8][RollingShape<G>::kPitch];
};

// the chains of one tile: batches of 8 start positions; past the row's last start the updates read staged zeros, results
// unused.  Round 5 (scripts/ubench/chain_ladder.hip, chain_mix.hip): a lone wavefront's LDS instructions do not overlap its
// dependent FP64 chain -- a ds_read_b64 costs the chain ~7 cycles, a ds_write_b64 ~17 when the next instruction overwrites
// its source -- and a wait placed by the compiler between a batch's reads and its chain costs an LDS round trip per batch.
// So: ONE explicit wait per batch, behind it the next batch's reads and the last batch's sums (from registers nothing
// overwrites for a whole batch), then the batch's 16 additions on registers alone.
#define ROCCO_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)  // s_waitcnt lgkmcnt(0) (gfx9 encoding; vmcnt / expcnt left open)
template <int G, bool WRAP>
__device__ __forceinline__ void rolling_rows_tile(RollingRows<G> &T, long long tile, int a0, int b0, int col, double &sum)
{
    constexpr int kTile = RollingShape<G>::kTile, kPitch = RollingShape<G>::kPitch;
    double(*__restrict__ S)[kPitch] = T.S[tile & 1];
    const double(*__restrict__ A)[kPitch] = T.P + a0;
    auto leaving = [&](int t) -> double { return A[t][col]; };
    constexpr int kRing = RollingShape<G>::kRing;
    double a[2][8], b[2][8], s[2][8];
    constexpr int kBatches = kTile / 8;
    static_assert(kBatches % 2 == 0, "two batches per trip");
    auto fetch = [&](int set, int batch) {
        const int t = 8 * ((batch < kBatches) ? batch : (kBatches - 1));  // (past the tile's end: the last batch again)
        // the batch's first entering line (inside the ring; its seven successors may run into the copy behind the ring's end)
        const double(*__restrict__ B)[kPitch] = T.P + (WRAP ? ((b0 + t) & (kRing - 1)) : (b0 + t));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[set][u] = leaving(t + u);
            b[set][u] = B[u][col];
        }
    };
    // (Round 5, measured and dropped: the chain wavefront storing only every eighth sum -- ROCCO_ROLL_FLUSH=8, 48.8 -> 33 cycles per
    // position -- with the helper wavefronts re-running the batches from those checkpoints: the re-runs' LDS traffic, 24 of 64
    // lanes per instruction, cost the helpers 1000 cycles per tile and the chain its gain: 113 -> 133 ms.  DESIGN.md 13.3)
    auto flush = [&](int set, int batch) {
#if defined(ROCCO_ROLL_FLUSH) && ROCCO_ROLL_FLUSH == 8  // (timing experiments only: what the chain wavefront's LDS writes cost)
        S[8 * batch][col] = s[set][0];
#else
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            S[8 * batch + u][col] = s[set][u];
        }
#endif
    };
    auto run = [&](int set) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s[set][u] = sum;
            sum = (sum - a[set][u]) + b[set][u];  // wls_backend.c:711-722
        }
    };
    fetch(0, 0);
#pragma unroll 1
    for (int j = 0; j < kBatches; j += 2) {
        ROCCO_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        fetch(1, j + 1);
        if (j > 0) {
            flush(1, j - 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        run(0);
        __builtin_amdgcn_sched_barrier(0);
        ROCCO_WAIT_LGKM0();
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, j + 2);
        flush(0, j);
        __builtin_amdgcn_sched_barrier(0);
        run(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    flush(1, kBatches - 1);
}

template <int G>
__global__ __launch_bounds__(kLanes + kRowsHelpers) void wls_rolling_rows_kernel(const WlsRollingTask *__restrict__ tasks)
{
    constexpr int kTile = RollingShape<G>::kTile, kRing = RollingShape<G>::kRing;
    extern __shared__ __align__(16) unsigned char rolling_rows_lds[];
    RollingRows<G> &T = *reinterpret_cast<RollingRows<G> *>(rolling_rows_lds);
    const WlsRollingTask task = tasks[blockIdx.x];
    const long long n = task.n;
    const int window = task.window, rows = task.rows;
    const long long max_start = n - window, stride_out = max_start + 1;
    const long long n_tiles = (max_start + kTile) / kTile;  // ceil((max_start + 1) / kTile)
    const int lane = threadIdx.x, hl = (int)threadIdx.x - kLanes;
    const bool helper = hl >= 0;
    const int ht = hl & (kTile - 1), hr = hl / kTile;  // helper: locus within the chunk, row of the group
    // a chunk is loaded kAhead trips before it is stored to LDS (round 5: with one trip between load and store every tile
    // waited for a memory latency -- the launch took 22 ns per locus with the chain AND the variances compiled out)
    constexpr int kAhead = 4;
    double rv[kAhead], rnx[kAhead];
#pragma unroll
    for (int q = 0; q < kAhead; ++q) {
        rv[q] = 0.0;
        rnx[q] = 0.0;
    }
    typedef const __attribute__((address_space(1))) double *global_row_t;  // (global_load, not flat_load: its own counter)
    const global_row_t my_row = (global_row_t)(task.row + (long long)((helper && hr < rows) ? hr : 0) * n);

    // (the loaded values stay untouched in their registers until store_chunk: a select on a value just loaded is a wait
    // for the load -- a memory latency per trip, which is what the launch took until round 5)
    auto load_chunk = [&](long long chunk, double &v_out, double &nx_out) {
        const long long i = chunk * kTile + ht;
        const long long i0 = (i < n) ? i : (n - 1), i1 = (i + 1 < n) ? (i + 1) : (n - 1);
        v_out = my_row[i0];
        nx_out = my_row[i1];
    };
    auto store_chunk = [&](long long chunk, double v_raw, double nx_raw) {  // (zeros past the row's end and for rows the task does not have)
        const long long i = chunk * kTile + ht;
        const double v = (hr < rows && i < n) ? v_raw : 0.0, nx = (hr < rows && i + 1 < n) ? nx_raw : 0.0;
        const int at = (int)(i & (kRing - 1));
        double *__restrict__ line = T.P[at];
        line[3 * hr] = v;
        line[3 * hr + 1] = v * v;
        line[3 * hr + 2] = v * nx;
        if (at < 8) {  // (the copy behind the ring's end)
            double *__restrict__ copy = T.P[kRing + at];
            copy[3 * hr] = v;
            copy[3 * hr + 1] = v * v;
            copy[3 * hr + 2] = v * nx;
        }
    };
    // the variances of tile `tile` from its sums (wls_backend.c:667-709; the divisions are off the chains)
    auto variances = [&](long long tile) {
        const long long t = tile * kTile + ht;
        if (t > max_start || hr >= rows) {
            return;
        }
        const double wd = (double)window, pair_count = (double)(window - 1);
        const double *__restrict__ sums = T.S[tile & 1][ht] + 3 * hr;
        const double sy = sums[0], ssq = sums[1], slag = sums[2];
        const double leaving = T.P[(int)(t & (kRing - 1))][3 * hr], entering = T.P[(int)((t + window - 1) & (kRing - 1))][3 * hr];
        const double sum_x_seq = sy - entering, sum_y_seq = sy - leaving;
        const double mean_all = sy / wd;
        double g0n = ssq - (wd * mean_all * mean_all);
        if (g0n < 0.0) {
            g0n = 0.0;
        }
        const double g1n = slag - (mean_all * sum_x_seq) - (mean_all * sum_y_seq) + (pair_count * mean_all * mean_all);
        const double lambda_eff = 1.0 / (wd + 1.0);
        const double scale_floor = 1.0e-4 * (g0n + 1.0);
        const double denom = (g0n * (1.0 + lambda_eff)) + scale_floor;
        const double eps = 1.0e-12 * (g0n + 1.0);
        double beta1 = 0.0;
        if (denom > eps) {
            beta1 = g1n / denom;
        }
        if (beta1 > 0.99) {
            beta1 = 0.99;
        } else if (beta1 < 0.0) {
            beta1 = 0.0;
        }
        const double gamma0 = g0n / wd;
        double omb = 1.0 - (beta1 * beta1);
        if (omb < 0.0) {
            omb = 0.0;
        }
        ((__attribute__((address_space(1))) double *)task.out)[(long long)hr * stride_out + t] = fmax(gamma0 * omb, 0.0);  // (global_store: flat_store would tie the load counter to the LDS counter)
    };

    if (helper) {
        double v0, nx0;
        load_chunk(0, v0, nx0);
        store_chunk(0, v0, nx0);
        load_chunk(1, v0, nx0);
        store_chunk(1, v0, nx0);
#pragma unroll
        for (int q = 0; q < kAhead; ++q) {
            load_chunk(2 + q, rv[q], rnx[q]);  // (chunk c waits in set (c - 2) % kAhead)
        }
    }
    lds_barrier();
    const int chain = lane % 3;
    const int off = (chain == 2) ? (window - 1) : window;
    const bool runs = !helper && lane < 3 * G;
    const int col = runs ? lane : 0;
    double sum = 0.0;
    if (!helper) {
        __builtin_amdgcn_s_setprio(3);  // the chain wavefront shares its SIMD with helpers: its instructions go first
    }
    if (runs) {
        for (int i = 0; i < off; ++i) {  // wls_backend.c:652-661 (window terms, window - 1 for the lagged products)
            sum += T.P[i][col];
        }
    }
#ifdef ROCCO_ROLL_STAMPS  // (timing experiments only: where the wavefronts of workgroup 0 spend their cycles)
    long long st_work = 0, st_wait = 0, st_a = 0, st_b = 0, st_c = 0;
#define ROCCO_STAMP(x) const long long x = __builtin_readcyclecounter()
#else
#define ROCCO_STAMP(x)
#endif
    auto trip = [&](long long tile, double &set_v, double &set_nx) {
        ROCCO_STAMP(t0);
#ifdef ROCCO_ROLL_NOCHAIN  // (timing experiments only)
        if (false) {
#else
        if (runs) {
#endif
            // a tile's own lines are consecutive ring lines; the lines `off` further on are consecutive too unless the
            // ring's end falls among them (every fourth tile): only then is the line index masked per access
            // a tile's own lines are consecutive ring lines; the lines `off` further on are consecutive too unless the
            // ring's end falls among them (every fourth tile): only then is the line index masked, once per batch
            const int a0 = (int)((tile * kTile) & (kRing - 1)), b0 = (a0 + off) & (kRing - 1);
            if (b0 + kTile <= kRing) {
                rolling_rows_tile<G, false>(T, tile, a0, b0, col, sum);
            } else {
                rolling_rows_tile<G, true>(T, tile, a0, b0, col, sum);
            }
        } else if (helper) {
            store_chunk(tile + 2, set_v, set_nx);  // (in the registers since kAhead trips)
            ROCCO_STAMP(ta);
            load_chunk(tile + 2 + kAhead, set_v, set_nx);
            ROCCO_STAMP(tb);
#ifndef ROCCO_ROLL_NOVAR
            if (tile > 0) {
                variances(tile - 1);
            }
#endif
#ifdef ROCCO_ROLL_STAMPS
            ROCCO_STAMP(tc);
            st_a += ta - t0;
            st_b += tb - ta;
            st_c += tc - tb;
#endif
        }
        ROCCO_STAMP(t1);
        lds_barrier();
#ifdef ROCCO_ROLL_STAMPS
        ROCCO_STAMP(t2);
        st_work += t1 - t0;
        st_wait += t2 - t1;
#endif
    };
    for (long long tile = 0; tile < n_tiles; tile += kAhead) {
#pragma unroll
        for (int q = 0; q < kAhead; ++q) {
            if (tile + q < n_tiles) {
                trip(tile + q, rv[q], rnx[q]);
            }
        }
    }
    if (helper) {
        variances(n_tiles - 1);
    }
#ifdef ROCCO_ROLL_STAMPS
    if (blockIdx.x == 0 && (threadIdx.x % kLanes) == 0) {
        const int w = threadIdx.x / kLanes;
        if (w < 3) {
            printf("[roll stamps] wave %d: tiles %lld, cycles per tile: work %lld (store %lld, load %lld, variances %lld), barrier wait %lld\n", w, n_tiles,
                   st_work / n_tiles, st_a / n_tiles, st_b / n_tiles, st_c / n_tiles, st_wait / n_tiles);
        }
    }
#endif
}

// ---- any window (above the LDS-tiled kernel's 63 loci): the same three chains straight from global memory, one
// wavefront per row, then the variances elementwise.  Slower (every step waits for its own loads) but the
// reference takes any window (wls_backend.c:610-742), so this must too.
__global__ __launch_bounds__(kLanes) void wls_rolling_general_sums_kernel(const double *__restrict__ matrix, long long n, int window,
                                                                         double *__restrict__ sums)
{
    const int lane = threadIdx.x;
    if (lane >= 3) {
        return;
    }
    const double *__restrict__ row = matrix + (long long)blockIdx.x * n;
    const long long max_start = n - window;
    double *__restrict__ S = sums + ((long long)blockIdx.x * 3 + lane) * (max_start + 1);
    const int off = (lane == 2) ? (window - 1) : window;
    auto term = [&](long long i) -> double {
        // past the row's end: zeros (those sums are never used)
        const double v = (i < n) ? row[i] : 0.0;
        if (lane == 0) {
            return v;
        }
        if (lane == 1) {
            return v * v;
        }
        const double nx = (i + 1 < n) ? row[i + 1] : 0.0;
        return v * nx;
    };
    double sum = 0.0;
    for (int i = 0; i < off; ++i) {  // wls_backend.c:652-661
        sum += term(i);
    }
    for (long long t = 0; t <= max_start; ++t) {
        S[t] = sum;
        sum = (sum - term(t)) + term(t + off);  // wls_backend.c:711-722
    }
}

__global__ __launch_bounds__(256) void wls_rolling_general_variance_kernel(const double *__restrict__ matrix, long long n, int window,
                                                                          const double *__restrict__ sums, double *__restrict__ vas)
{
    const long long max_start = n - window;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > max_start) {
        return;
    }
    const long long r = blockIdx.y;
    const double *__restrict__ row = matrix + r * n;
    const double *__restrict__ S = sums + r * 3 * (max_start + 1);
    const double wd = (double)window, pair_count = (double)(window - 1);
    const double sy = S[t], ssq = S[(max_start + 1) + t], slag = S[2 * (max_start + 1) + t];
    const double leaving = row[t], entering = row[t + window - 1];
    const double sum_x_seq = sy - entering, sum_y_seq = sy - leaving;
    const double mean_all = sy / wd;
    double g0n = ssq - (wd * mean_all * mean_all);
    if (g0n < 0.0) {
        g0n = 0.0;
    }
    const double g1n = slag - (mean_all * sum_x_seq) - (mean_all * sum_y_seq) + (pair_count * mean_all * mean_all);
    const double lambda_eff = 1.0 / (wd + 1.0);
    const double scale_floor = 1.0e-4 * (g0n + 1.0);
    const double denom = (g0n * (1.0 + lambda_eff)) + scale_floor;
    const double eps = 1.0e-12 * (g0n + 1.0);
    double beta1 = 0.0;
    if (denom > eps) {
        beta1 = g1n / denom;
    }
    if (beta1 > 0.99) {
        beta1 = 0.99;
    } else if (beta1 < 0.0) {
        beta1 = 0.0;
    }
    const double gamma0 = g0n / wd;
    double omb = 1.0 - (beta1 * beta1);
    if (omb < 0.0) {
        omb = 0.0;
    }
    vas[r * (max_start + 1) + t] = fmax(gamma0 * omb, 0.0);
}

__device__ __forceinline__ double obs_variance_at(const double *__restrict__ vas_row, long long i, long long half,
                                                  long long max_start)
{
    // wls_backend.c:727-738 then 865-868
    long long c = (i < half) ? 0 : (i - half);
    if (c > max_start) {
        c = max_start;
    }
    return fmax(vas_row[c], 1.0e-8);
}

// pairs of one row: keys (bit patterns of non-negative doubles order like the numbers)
__global__ __launch_bounds__(256) void wls_pairs_kernel(const double *__restrict__ row, const double *__restrict__ vas_row,
                                                       long long n, long long half, long long max_start,
                                                       unsigned long long *__restrict__ key_y,
                                                       unsigned long long *__restrict__ val_x, int *__restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const double x = fabs(row[i]);
    const double y = fmax(obs_variance_at(vas_row, i, half, max_start), 1.0e-8);  // wls_backend.c:429-430
    if (!isfinite(x) || !isfinite(y)) {
        atomicOr(bad, 1);
    }
    key_y[i] = (unsigned long long)__double_as_longlong(y);
    val_x[i] = (unsigned long long)__double_as_longlong(x);
}

__global__ __launch_bounds__(256) void wls_iota_kernel(unsigned *__restrict__ idx, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        idx[i] = (unsigned)i;
    }
}

// perm[r] = position in the y-sorted sequence of the pair of rank r in (x, y) order; the pair's trend bin
// is the b with (b n) / bins <= r < ((b + 1) n) / bins (wls_backend.c:474-476)
__global__ __launch_bounds__(256) void wls_bin_scatter_kernel(const unsigned *__restrict__ perm, long long n, int bins,
                                                             unsigned char *__restrict__ bin_of_yrank)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) {
        return;
    }
    long long b = (r * bins) / n;
    while (b + 1 < bins && ((b + 1) * n) / bins <= r) {
        ++b;
    }
    while (b > 0 && (b * n) / bins > r) {
        --b;
    }
    bin_of_yrank[perm[r]] = (unsigned char)b;
}

struct TrendFit {
    int mode;         // 0: constant `value`, 2: interpolate the knots
    int knots;
    double value;
    double kc[kMaxBins], kv[kMaxBins];
};

__device__ __forceinline__ double bits_to_double(unsigned long long b) { return __longlong_as_double((long long)b); }

// one thread: bins -> pooled fit -> knots (wls_backend.c:436-439, 472-565)
__global__ void wls_knots_kernel(const unsigned long long *__restrict__ ys_sorted_all,  // all y ascending
                                 const unsigned long long *__restrict__ xs,             // x of the (x, y)-sorted pairs
                                 const unsigned long long *__restrict__ ys_bins,        // y sorted inside each bin
                                 long long n, int bins, TrendFit *fit)
{
    __shared__ double s_bc[kMaxBins], s_bv[kMaxBins], s_bw[kMaxBins];
    __shared__ double s_fallback;
    {
        // one lane per bin fetches the bin's medians (wls_backend.c:472-520)
        const int b = threadIdx.x;
        if (b < bins) {
            const long long left = ((long long)b * n) / bins, right = ((long long)(b + 1) * n) / bins;
            const long long width = right - left;
            double c = 0.0, v = 0.0;
            if (width > 0) {
                if (width & 1LL) {
                    c = bits_to_double(xs[left + width / 2]);
                    v = bits_to_double(ys_bins[left + width / 2]);
                } else {
                    c = 0.5 * (bits_to_double(xs[left + width / 2 - 1]) + bits_to_double(xs[left + width / 2]));
                    v = 0.5 * (bits_to_double(ys_bins[left + width / 2 - 1]) + bits_to_double(ys_bins[left + width / 2]));
                }
            }
            s_bc[b] = c;
            s_bv[b] = v;
            s_bw[b] = (double)width;
        }
        if (b == kMaxBins - 1) {
            // fallback: median of every y (wls_backend.c:436-439)
            double f;
            if (n & 1LL) {
                f = bits_to_double(ys_sorted_all[n / 2]);
            } else {
                f = 0.5 * (bits_to_double(ys_sorted_all[n / 2 - 1]) + bits_to_double(ys_sorted_all[n / 2]));
            }
            s_fallback = fmax(f, 1.0e-8);
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) {
        return;
    }
    const double fallback = s_fallback;
    // the pooling's working arrays live in LDS, not in the thread's private memory: indexed by run-time values they
    // would be 2.6 KB of scratch per lane -- a kernel whose scratch demand (per lane x every wavefront slot of the device)
    // exceeds the runtime's per-queue limit takes the "large scratch" path of the HSA runtime on every dispatch, and with
    // four or more streams dispatching it side by side (the count path of a genome, one stream per chromosome) streams
    // stopped for good (round 3: two of four workers never returned; the process had to be killed)
    __shared__ double bc[kMaxBins], bv[kMaxBins], bw[kMaxBins], fitv[kMaxBins];
    __shared__ long long bl[kMaxBins];
    int used = 0;
    for (int b = 0; b < bins; ++b) {
        if (s_bw[b] > 0.0) {
            bc[used] = s_bc[b];
            bv[used] = s_bv[b];
            bw[used] = s_bw[b];
            ++used;
        }
    }
    fit->mode = 0;
    fit->knots = 0;
    if (used == 0) {
        fit->value = fallback;
        return;
    }
    if (used == 1) {
        fit->value = fmax(bv[0], 1.0e-8);
        return;
    }
    // pool adjacent violators (wls_backend.c:262-339)
    int nb = 0;
    for (int i = 0; i < used; ++i) {
        fitv[nb] = bv[i];
        bw[nb] = fmax(bw[i], 1.0e-8);
        bl[nb] = 1;
        ++nb;
        while (nb >= 2 && fitv[nb - 2] > fitv[nb - 1]) {
            const double tw = bw[nb - 2] + bw[nb - 1];
            const double mv = ((fitv[nb - 2] * bw[nb - 2]) + (fitv[nb - 1] * bw[nb - 1])) / tw;
            fitv[nb - 2] = mv;
            bw[nb - 2] = tw;
            bl[nb - 2] += bl[nb - 1];
            --nb;
        }
    }
    // expand blocks and build the knots (wls_backend.c:541-552)
    int knots = 0;
    int idx = 0;
    for (int b = 0; b < nb; ++b) {
        for (long long r = 0; r < bl[b]; ++r, ++idx) {
            const double cv = bc[idx], vv = fmax(fitv[b], 1.0e-8);
            if (knots > 0 && cv <= fit->kc[knots - 1]) {
                fit->kv[knots - 1] = fmax(fit->kv[knots - 1], vv);
                continue;
            }
            fit->kc[knots] = cv;
            fit->kv[knots] = vv;
            ++knots;
        }
    }
    fit->knots = knots;
    if (knots == 0) {
        fit->value = fallback;
    } else if (knots == 1) {
        fit->value = fmax(fit->kv[0], 1.0e-8);
    } else {
        fit->mode = 2;
    }
}

__device__ __forceinline__ double key_to_double(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    return __longlong_as_double((long long)b);
}

// ---- np.median of every row without sorting it (round 3) ----------------------------------------------------------------
// The median is one order statistic (two for an even length): a radix SELECT over the order-preserving keys finds the key
// of rank (n - 1) / 2 of EVERY row in six passes over the matrix -- digits of 11, 11, 11, 11, 11 and 9 bits from the top,
// each pass counting, per row, the digits of the keys that still match the row's prefix -- and one more pass finds the
// next key above it.  56 bytes read per value and 14 launches per MATRIX, against a full radix sort (8 passes, 16 bytes
// moved per value and pass) and ~12 launches per ROW: 2 400 sorts and 30 000 launches less per genome.
constexpr int kSelectBuckets = 2048;
constexpr int kSelectChunk = 8192;  // values per workgroup of a pass

struct RowSelect {  // per row, in device memory
    unsigned long long prefix;   // the bits of the wanted key found so far (right-aligned)
    long long rank;              // rank of the wanted key among the keys that match the prefix
    unsigned long long above;    // smallest key above the wanted one (~0: none)
    long long count_le;          // keys at or below the wanted one
};

__device__ __forceinline__ unsigned long long order_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}

// One count per active lane into an LDS histogram.  The keys of a row of counts, or the variances of one bin, crowd into a few
// buckets: 64 lanes adding to ONE counter are serialised lane by lane.  So the two most common buckets of the wavefront are
// counted by one lane each (a ballot of the lanes that share the leader's bucket), whoever is left adds for itself.
__device__ __forceinline__ void lds_count(unsigned *__restrict__ local, unsigned bucket, bool active)
{
    unsigned long long todo = __ballot(active);
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (todo == 0ULL) {
            break;
        }
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned b0 = (unsigned)__shfl((int)bucket, leader);
        const unsigned long long same = __ballot(active && bucket == b0) & todo;
        if (lane == leader) {
            atomicAdd(&local[b0], (unsigned)__popcll(same));
        }
        todo &= ~same;
    }
    if ((todo >> lane) & 1ULL) {
        atomicAdd(&local[bucket], 1u);
    }
}

// pass `p` looks at the `width` bits above bit `low`; hist[row][digit] += keys of the row matching its prefix.
// SPAN (round 5, the passes behind the gathered-cell shortcut): also the smallest and the largest key that matches the
// prefix -- a row of counts has its median inside a RUN of equal values, which no cell of kCellMax values holds; when the
// two agree every key of the cell is the wanted one and the row is complete without the passes that remain
// (row_select_span_kernel): three passes instead of six over such rows.
template <bool SPAN>
__global__ __launch_bounds__(256) void row_select_count_kernel(const double *__restrict__ matrix, long long n, int low, int width,
                                                              const RowSelect *__restrict__ state, unsigned *__restrict__ hist,
                                                              unsigned long long *__restrict__ span, const unsigned *__restrict__ complete)
{
    __shared__ unsigned local[kSelectBuckets];
    __shared__ unsigned long long wave_lo[4], wave_hi[4];
    const long long row = blockIdx.y;
    if (state[row].rank < 0 || (complete != nullptr && complete[row] != 0u)) {  // settled from its gathered cell (row_gather_kernel), or complete
        return;
    }
    unsigned long long lo_key = ~0ULL, hi_key = 0ULL;
    for (int b = threadIdx.x; b < kSelectBuckets; b += 256) {
        local[b] = 0u;
    }
    __syncthreads();
    const unsigned long long prefix = state[row].prefix;
    const int above_bits = low + width;  // the prefix holds bits above_bits .. 63
    const unsigned long long mask = (1ULL << width) - 1ULL;
    const double *__restrict__ x = matrix + row * n;
    const long long base = (long long)blockIdx.x * kSelectChunk;
    // (eight loads in flight per thread, at clamped positions: a load under `if (i < n)` is waited for before the next is issued)
    for (int j0 = 0; j0 < kSelectChunk / 256; j0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = base + threadIdx.x + 256LL * (j0 + u);
            v[u] = x[(i < n) ? i : (n - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = base + threadIdx.x + 256LL * (j0 + u);
            const unsigned long long k = order_key(v[u]);
            const bool counts = i < n && (above_bits >= 64 || (k >> above_bits) == prefix);
            lds_count(local, (unsigned)((k >> low) & mask), counts);
            if (SPAN && counts) {
                lo_key = (k < lo_key) ? k : lo_key;
                hi_key = (k > hi_key) ? k : hi_key;
            }
        }
    }
    if (SPAN) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long l2 = __shfl_xor(lo_key, off), h2 = __shfl_xor(hi_key, off);
            lo_key = (l2 < lo_key) ? l2 : lo_key;
            hi_key = (h2 > hi_key) ? h2 : hi_key;
        }
        if ((threadIdx.x & 63) == 0) {
            wave_lo[threadIdx.x >> 6] = lo_key;
            wave_hi[threadIdx.x >> 6] = hi_key;
        }
    }
    __syncthreads();
    unsigned *__restrict__ mine = hist + row * kSelectBuckets;
    for (int b = threadIdx.x; b < kSelectBuckets; b += 256) {
        if (local[b] != 0u) {
            atomicAdd(&mine[b], local[b]);
        }
    }
    if (SPAN && threadIdx.x == 0) {
        unsigned long long l = wave_lo[0], h = wave_hi[0];
        for (int w = 1; w < 4; ++w) {
            l = (wave_lo[w] < l) ? wave_lo[w] : l;
            h = (wave_hi[w] > h) ? wave_hi[w] : h;
        }
        if (l <= h) {
            atomicMin(&span[2 * row], l);
            atomicMax(&span[2 * row + 1], h);
        }
    }
}

// behind a SPAN pass and its pick: a row whose matching keys were all ONE key has that key for its median's lower middle
// element -- its prefix is complete, the counting passes that remain skip it (the pass for the upper middle element does not)
__global__ __launch_bounds__(256) void row_select_span_kernel(RowSelect *__restrict__ state, unsigned long long *__restrict__ span,
                                                             unsigned *__restrict__ complete, long long rows, int init)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) {
        return;
    }
    const unsigned long long l = span[2 * row], h = span[2 * row + 1];
    span[2 * row] = ~0ULL;  // (empty again for the next pass)
    span[2 * row + 1] = 0ULL;
    if (init == 0 && state[row].rank >= 0 && complete[row] == 0u && l == h) {
        state[row].prefix = l;
        complete[row] = 1u;
    }
}

// one workgroup per row: the digit whose bucket holds the wanted rank joins the prefix; the histogram is cleared for the next pass
// (`bucket`, may be null: how many keys the chosen digit's bucket holds; rows whose rank is negative are settled already)
__global__ __launch_bounds__(256) void row_select_pick_kernel(RowSelect *__restrict__ state, unsigned *__restrict__ hist, int width,
                                                             unsigned *__restrict__ bucket, const unsigned *__restrict__ complete = nullptr)
{
    __shared__ unsigned part[256];
    __shared__ unsigned long long chosen[2];
    const long long row = blockIdx.x;
    if (state[row].rank < 0 || (complete != nullptr && complete[row] != 0u)) {
        return;
    }
    unsigned *__restrict__ mine = hist + row * kSelectBuckets;
    const int per = kSelectBuckets / 256;  // 8 consecutive buckets per thread
    unsigned c[8], sum = 0u;
#pragma unroll
    for (int q = 0; q < per; ++q) {
        c[q] = mine[threadIdx.x * per + q];
        sum += c[q];
        mine[threadIdx.x * per + q] = 0u;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long rank = state[row].rank;
        int t = 0;
        while (t < 255 && (long long)part[t] <= rank) {
            rank -= (long long)part[t];
            ++t;
        }
        chosen[0] = (unsigned long long)t;
        chosen[1] = (unsigned long long)rank;
    }
    __syncthreads();
    if (threadIdx.x == (unsigned)chosen[0]) {
        long long rank = (long long)chosen[1];
        int q = 0;
        while (q < per - 1 && (long long)c[q] <= rank) {
            rank -= (long long)c[q];
            ++q;
        }
        const unsigned long long digit = (unsigned long long)(threadIdx.x * per + q);
        state[row].prefix = (width >= 64) ? digit : ((state[row].prefix << width) | digit);
        state[row].rank = rank;
        if (bucket != nullptr) {
            bucket[row] = c[q];
        }
    }
}

// with the wanted key known: how many keys lie at or below it, and the smallest key above it
__global__ __launch_bounds__(256) void row_select_above_kernel(const double *__restrict__ matrix, long long n,
                                                              RowSelect *__restrict__ state)
{
    const long long row = blockIdx.y;
    if (state[row].rank < 0) {
        return;
    }
    const unsigned long long want = state[row].prefix;
    const double *__restrict__ x = matrix + row * n;
    const long long base = (long long)blockIdx.x * kSelectChunk;
    long long le = 0;
    unsigned long long above = ~0ULL;
#pragma unroll 4
    for (int j = 0; j < kSelectChunk / 256; ++j) {
        const long long i = base + threadIdx.x + 256LL * j;
        if (i < n) {
            const unsigned long long k = order_key(x[i]);
            le += (k <= want) ? 1 : 0;
            above = (k > want && k < above) ? k : above;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        le += __shfl_xor(le, off);
        const unsigned long long o = __shfl_xor(above, off);
        above = (o < above) ? o : above;
    }
    if ((threadIdx.x & 63) == 0) {
        if (le != 0) {
            atomicAdd((unsigned long long *)&state[row].count_le, (unsigned long long)le);
        }
        if (above != ~0ULL) {
            atomicMin(&state[row].above, above);
        }
    }
}

__global__ __launch_bounds__(256) void row_select_init_kernel(RowSelect *__restrict__ state, unsigned *__restrict__ hist, long long rows,
                                                             long long n, unsigned *__restrict__ filled)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) {
        filled[i] = 0u;
        state[i].prefix = 0ULL;
        state[i].rank = (n - 1) / 2;  // the lower middle element
        state[i].above = ~0ULL;
        state[i].count_le = 0;
    }
    if (i < rows * kSelectBuckets) {
        hist[i] = 0u;
    }
}

// np.median: the middle value, or the mean of the two middle values (the upper one is the wanted key again when it occurs
// more than once beyond its rank, else the next key above it)
__global__ __launch_bounds__(256) void row_select_median_kernel(const RowSelect *__restrict__ state, long long rows, long long n,
                                                               double *__restrict__ med)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) {
        return;
    }
    const double lo = key_to_double(state[i].prefix);
    if (n & 1LL) {
        med[i] = lo;
    } else {
        const double hi = (state[i].count_le >= n / 2 + 1) ? lo : key_to_double(state[i].above);
        med[i] = (lo + hi) / 2.0;
    }
}

// ---- the trend fit without sorting the pairs (round 3) ------------------------------------------------------------------
// What the fit needs of the (x, y)-sorted sequence (wls_backend.c:454-520) is, per bin, the median x and the median y.  The
// bins are contiguous rank ranges of the x order, so a handful of order statistics of x per bin (found for every row of the
// matrix at once, see "the x side" below) give every bin's first x (its boundary) and median x; with the boundaries every
// pair knows its bin from its x alone -- unless a boundary falls inside a run of equal x, where the reference's order
// inside the run (by y) decides: such a row (`tie`) takes the sorted path below, as do short rows; the y values of ALL
// rows of the matrix are then dealt into per-bin segments (order inside a segment does not matter) and every segment's
// median is one radix select (the row-median kernels of the count-path glue, per segment).  Per value: 24 B read for the
// x ranks, 24 B to deal, 56 B to select, in ~30 launches per MATRIX; the sorted path moves ~600 B per value in 40 launches
// per ROW.
struct TrendRow {
    double bound[kMaxBins];  // bound[b] (b >= 1): the x of the first pair of bin b
    double cov[kMaxBins];    // median x of the bin
    double var[kMaxBins];    // median y of the bin
    int tie;                 // 1: this row takes the sorted path
    int pad;
};

// ---- the x side: the wanted order statistics of every row without sorting it ------------------------------------------------
// Per bin the fit reads FOUR ranks of the row's |x| order: the last of the bin before and the first of the bin (the
// boundary and its tie check) and the one or two middle ones (the bin's median x) -- at most 128 ranks for the 32 bins a
// row of fewer than 2^32 loci can have.  Three passes over the MATRIX find them all: (0) a histogram of the top 14 key
// bits of every row (bits 62..49: exponent + 3 mantissa bits; the sign bit of |x| is 0), from which every rank knows its
// bucket; (1) inside the buckets that hold a rank ("slots", at most 128, in practice ~30), a histogram of the next
// 6..11 bits, all slots at once in 8192 LDS counters; (2) the values of the ~22-bit cells that hold a rank -- about
// n / 2^11 each, some 2 % of the row in all -- are gathered, and one workgroup per cell sorts its few hundred values in
// LDS and reads the ranks off.  24 bytes read per value and 8 launches per matrix, against a keys-only radix sort per row
// (8 passes, 16 bytes moved per value and pass, ~25 launches and memsets per row).  A row with a cell of more than 8192
// values (heavy runs of equal x) takes the sorted path like a row with a tie at a boundary.
constexpr int kRankTargets = 128;     // 4 per bin
constexpr int kRankBuckets0 = 16384;  // first digit: key bits 62..49
constexpr int kRankShift0 = 49;
constexpr int kRankCells1 = 8192;     // counters of the second pass: slots << width1
constexpr int kRankChunk = 32768;     // values per workgroup of the histogram passes
constexpr int kGatherChunk = 16384;   // values per workgroup of the gathering pass
constexpr int kRankSortMax = 8192;    // values of one cell

struct RankRow {
    int slots0, width1, slots1, pad;
    unsigned short bucket0[kRankTargets];  // the first digit of slot s
    unsigned short cell1[kRankTargets];    // (slot << width1 | second digit) of cell c
    unsigned count1[kRankTargets], offset1[kRankTargets], cursor1[kRankTargets];
    long long rank[kRankTargets];          // -1: unused; the rank inside the slot / cell as the passes go
    unsigned char slot[kRankTargets];
    unsigned long long key[kRankTargets];  // the answer
};

__device__ __forceinline__ long long rank_of_target(int t, long long n, int bins)
{
    const int b = t >> 2, which = t & 3;
    if (b >= bins) {
        return -1;
    }
    const long long left = ((long long)b * n) / bins, right = ((long long)(b + 1) * n) / bins, w = right - left;
    if (w <= 0) {
        return -1;
    }
    switch (which) {
    case 0: return (b >= 1 && left >= 1) ? left - 1 : -1;
    case 1: return (b >= 1) ? left : -1;
    case 2: return ((w & 1LL) == 0) ? left + w / 2 - 1 : -1;
    default: return left + w / 2;
    }
}

__global__ __launch_bounds__(256) void rank_init_kernel(RankRow *__restrict__ rr, unsigned *__restrict__ hist0, TrendRow *__restrict__ trows,
                                                       long long rows, long long n, int bins)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < rows * kRankBuckets0) {
        hist0[i] = 0u;
    }
    if (i < rows * kRankTargets) {
        rr[i / kRankTargets].rank[i % kRankTargets] = rank_of_target((int)(i % kRankTargets), n, bins);
    }
    if (i < rows) {
        trows[i].tie = 0;
    }
}

constexpr int kRankThreads = 1024;  // the histogram passes: 16 wavefronts per workgroup (their 48-64 KB of LDS allow two or three per CU)

__global__ __launch_bounds__(kRankThreads) void rank_hist0_kernel(const double *__restrict__ matrix, long long n, unsigned *__restrict__ hist0,
                                                                 int *__restrict__ bad)
{
    __shared__ unsigned local[kRankBuckets0];
    const long long row = blockIdx.y;
    for (int b = threadIdx.x; b < kRankBuckets0; b += kRankThreads) {
        local[b] = 0u;
    }
    __syncthreads();
    const double *__restrict__ x = matrix + row * n;
    const long long base = (long long)blockIdx.x * kRankChunk;
    bool finite = true;
    constexpr int kBatch = 16;  // loads of a batch are issued before its LDS atomics
#pragma unroll 1
    for (int j0 = 0; j0 < kRankChunk / kRankThreads; j0 += kBatch) {
        double v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const long long i = base + threadIdx.x + (long long)kRankThreads * (j0 + j);
            v[j] = (i < n) ? fabs(x[i]) : -1.0;
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            if (v[j] >= 0.0 || v[j] != v[j]) {  // (a NaN is counted too: the call fails with the flag below)
                finite = finite && isfinite(v[j]);
                atomicAdd(&local[(unsigned)((unsigned long long)__double_as_longlong(v[j]) >> kRankShift0) & (kRankBuckets0 - 1)], 1u);
            }
        }
    }
    if (!finite) {
        atomicOr(bad, 1);
    }
    __syncthreads();
    unsigned *__restrict__ mine = hist0 + row * kRankBuckets0;
    for (int b = threadIdx.x; b < kRankBuckets0; b += kRankThreads) {
        if (local[b] != 0u) {
            atomicAdd(&mine[b], local[b]);
        }
    }
}

// one workgroup per row: which counter of `cells` (count of them a multiple of 256 * per) holds each rank.  On return
// found[t] = the counter, and rank[t] the rank inside it.  `origin[t]`: the counter the rank is counted from.
template <int PER>
__device__ __forceinline__ void rank_locate(const unsigned *__restrict__ cells, const int *origin, long long *rank, int *found,
                                            unsigned long long *part_base)
{
    unsigned long long sum = 0;
    for (int q = 0; q < PER; ++q) {
        sum += cells[threadIdx.x * PER + q];
    }
    part_base[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int c = 0; c < 256; ++c) {
            const unsigned long long v = part_base[c];
            part_base[c] = run;
            run += v;
        }
        part_base[256] = run;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < kRankTargets) {
        found[t] = -1;
        if (rank[t] >= 0) {
            const unsigned long long want = part_base[origin[t] / PER] + (unsigned long long)rank[t];  // (origins sit on part boundaries)
            int lo = 0, hi = 256;  // the last part whose base is <= want
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (part_base[mid] <= want) {
                    lo = mid;
                } else {
                    hi = mid;
                }
            }
            unsigned long long run = part_base[lo];
            int c = lo * PER;
            for (int q = 0; q < PER; ++q, ++c) {
                const unsigned long long here = cells[c];
                if (want < run + here) {
                    break;
                }
                run += here;
            }
            found[t] = (c < 256 * PER) ? c : -1;  // (-1: the rank lies beyond the counted values -- cannot happen)
            rank[t] = (long long)(want - run);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void rank_plan0_kernel(RankRow *__restrict__ rr, const unsigned *__restrict__ hist0,
                                                        unsigned *__restrict__ hist1, TrendRow *__restrict__ trows)
{
    __shared__ unsigned long long part_base[257];
    __shared__ long long rank[kRankTargets];
    __shared__ int origin[kRankTargets], found[kRankTargets];
    RankRow &r = rr[blockIdx.x];
    const int t = threadIdx.x;
    if (t < kRankTargets) {
        rank[t] = r.rank[t];
        origin[t] = 0;
    }
    for (int c = t; c < kRankCells1; c += 256) {
        hist1[(long long)blockIdx.x * kRankCells1 + c] = 0u;
    }
    __syncthreads();
    rank_locate<kRankBuckets0 / 256>(hist0 + (long long)blockIdx.x * kRankBuckets0, origin, rank, found, part_base);
    if (t == 0) {
        int slots = 0, prev = -1, lost = 0;
        for (int q = 0; q < kRankTargets; ++q) {  // the ranks ascend with q, so do their buckets
            if (rank[q] < 0) {
                continue;
            }
            if (found[q] < 0) {
                lost = 1;
                continue;
            }
            if (found[q] != prev) {
                prev = found[q];
                r.bucket0[slots++] = (unsigned short)prev;
            }
            r.slot[q] = (unsigned char)(slots - 1);
        }
        int width = 11;
        while (width > 0 && (slots << width) > kRankCells1) {
            --width;
        }
        r.slots0 = lost ? 0 : slots;
        r.width1 = width;
        r.slots1 = 0;
        if (lost) {
            trows[blockIdx.x].tie = 1;
        }
    }
    if (t < kRankTargets) {
        r.rank[t] = rank[t];
    }
}

__device__ __forceinline__ void rank_slot_map(const RankRow &r, unsigned char *map0)
{
    for (int b = threadIdx.x; b < kRankBuckets0 / 4; b += blockDim.x) {
        ((unsigned *)map0)[b] = 0xFFFFFFFFu;
    }
    __syncthreads();
    if ((int)threadIdx.x < r.slots0) {
        map0[r.bucket0[threadIdx.x]] = (unsigned char)threadIdx.x;
    }
}

__global__ __launch_bounds__(kRankThreads) void rank_hist1_kernel(const double *__restrict__ matrix, long long n,
                                                                 const RankRow *__restrict__ rr, unsigned *__restrict__ hist1)
{
    __shared__ unsigned local[kRankCells1];
    __shared__ unsigned char map0[kRankBuckets0];
    const long long row = blockIdx.y;
    const RankRow &r = rr[row];
    if (r.slots0 == 0) {
        return;
    }
    const int width = r.width1, cells = r.slots0 << width;
    for (int b = threadIdx.x; b < cells; b += kRankThreads) {
        local[b] = 0u;
    }
    rank_slot_map(r, map0);
    __syncthreads();
    const double *__restrict__ x = matrix + row * n;
    const long long base = (long long)blockIdx.x * kRankChunk;
    const int low = kRankShift0 - width;
    const unsigned mask = (1u << width) - 1u;
    constexpr int kBatch = 16;
#pragma unroll 1
    for (int j0 = 0; j0 < kRankChunk / kRankThreads; j0 += kBatch) {
        unsigned long long k[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const long long i = base + threadIdx.x + (long long)kRankThreads * (j0 + j);
            k[j] = (i < n) ? (unsigned long long)__double_as_longlong(fabs(x[i])) : ~0ULL;  // (no |x| has its sign bit set)
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            if (k[j] != ~0ULL) {
                const unsigned s = map0[(unsigned)(k[j] >> kRankShift0) & (kRankBuckets0 - 1)];
                if (s != 255u) {
                    atomicAdd(&local[(s << width) | ((unsigned)(k[j] >> low) & mask)], 1u);
                }
            }
        }
    }
    __syncthreads();
    unsigned *__restrict__ mine = hist1 + row * kRankCells1;
    for (int b = threadIdx.x; b < cells; b += kRankThreads) {
        if (local[b] != 0u) {
            atomicAdd(&mine[b], local[b]);
        }
    }
}

__global__ __launch_bounds__(256) void rank_plan1_kernel(RankRow *__restrict__ rr, const unsigned *__restrict__ hist1, long long capacity,
                                                        TrendRow *__restrict__ trows)
{
    __shared__ unsigned long long part_base[257];
    __shared__ long long rank[kRankTargets];
    __shared__ int origin[kRankTargets], found[kRankTargets];
    RankRow &r = rr[blockIdx.x];
    if (r.slots0 == 0) {
        return;
    }
    const int t = threadIdx.x;
    if (t < kRankTargets) {
        rank[t] = r.rank[t];
        origin[t] = (rank[t] >= 0) ? ((int)r.slot[t] << r.width1) : 0;  // (width1 >= 6: a multiple of the 32 counters of a part)
    }
    __syncthreads();
    const unsigned *__restrict__ cells = hist1 + (long long)blockIdx.x * kRankCells1;  // (zero beyond slots0 << width1)
    rank_locate<kRankCells1 / 256>(cells, origin, rank, found, part_base);
    if (t == 0) {
        int count = 0, prev = -1, lost = 0;
        unsigned long long total = 0;
        for (int q = 0; q < kRankTargets; ++q) {
            if (rank[q] < 0) {
                continue;
            }
            if (found[q] < 0 || (found[q] >> r.width1) != (int)r.slot[q]) {
                lost = 1;  // (cannot happen: the rank lies inside its slot)
                continue;
            }
            if (found[q] != prev) {
                prev = found[q];
                const unsigned here = cells[prev];
                r.cell1[count] = (unsigned short)prev;
                r.count1[count] = here;
                r.offset1[count] = (unsigned)total;
                r.cursor1[count] = 0u;
                total += here;
                if (here > (unsigned)kRankSortMax) {
                    lost = 1;  // a cell too full to sort in LDS: runs of equal or nearly equal x
                }
                ++count;
            }
            r.slot[q] = (unsigned char)(count - 1);
        }
        if (lost || total > (unsigned long long)capacity) {
            trows[blockIdx.x].tie = 1;
            count = 0;
        }
        r.slots1 = count;
    }
    if (t < kRankTargets) {
        r.rank[t] = rank[t];
    }
}

// the values of the cells that hold a rank, cell by cell, into the row's room of `cand`
__global__ __launch_bounds__(256) void rank_gather_kernel(const double *__restrict__ matrix, long long n, RankRow *__restrict__ rr,
                                                         unsigned long long *__restrict__ cand, long long capacity)
{
    __shared__ unsigned char map0[kRankBuckets0], map1[kRankCells1];
    __shared__ unsigned cnt[kRankTargets], base_of[kRankTargets];
    const long long row = blockIdx.y;
    RankRow &r = rr[row];
    if (r.slots1 == 0) {
        return;
    }
    const int width = r.width1, t = threadIdx.x;
    for (int b = t; b < kRankCells1 / 4; b += 256) {
        ((unsigned *)map1)[b] = 0xFFFFFFFFu;
    }
    if (t < kRankTargets) {
        cnt[t] = 0u;
    }
    rank_slot_map(r, map0);
    __syncthreads();
    if (t < r.slots1) {
        map1[r.cell1[t]] = (unsigned char)t;
    }
    __syncthreads();
    const double *__restrict__ x = matrix + row * n;
    const long long first = (long long)blockIdx.x * kGatherChunk;
    const int low = kRankShift0 - width;
    const unsigned mask = (1u << width) - 1u;
    unsigned char mine[kGatherChunk / 256];
    // (eight loads in flight per thread, at clamped positions: a load under `if (i < n)` is waited for by itself)
    double held[8];
#pragma unroll
    for (int j = 0; j < kGatherChunk / 256; ++j) {
        if ((j & 7) == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long iu = first + t + 256LL * (j + u);
                held[u] = x[(iu < n) ? iu : (n - 1)];
            }
        }
        const long long i = first + t + 256LL * j;
        unsigned c = 255u;
        if (i < n) {
            const unsigned long long k = (unsigned long long)__double_as_longlong(fabs(held[j & 7]));
            const unsigned s = map0[(unsigned)(k >> kRankShift0) & (kRankBuckets0 - 1)];
            if (s != 255u) {
                c = map1[(s << width) | ((unsigned)(k >> low) & mask)];
                if (c != 255u) {
                    atomicAdd(&cnt[c], 1u);
                }
            }
        }
        mine[j] = (unsigned char)c;
    }
    __syncthreads();
    if (t < r.slots1) {
        const unsigned c = cnt[t];
        base_of[t] = (c != 0u) ? atomicAdd(&r.cursor1[t], c) : 0u;
        cnt[t] = 0u;
    }
    __syncthreads();
    unsigned long long *__restrict__ out = cand + row * capacity;
#pragma unroll
    for (int j = 0; j < kGatherChunk / 256; ++j) {
        const unsigned c = mine[j];
        if (c != 255u) {
            const unsigned at = base_of[c] + atomicAdd(&cnt[c], 1u);
            if (at < r.count1[c]) {  // (never false: the count is the histogram's)
                out[(long long)r.offset1[c] + at] = (unsigned long long)__double_as_longlong(fabs(x[first + t + 256LL * j]));
            }
        }
    }
}

// one workgroup per (cell, row): bitonic sort of the cell's values in LDS, then the ranks that live in it
__global__ __launch_bounds__(256) void rank_sort_kernel(RankRow *__restrict__ rr, const unsigned long long *__restrict__ cand,
                                                       long long capacity, TrendRow *__restrict__ trows)
{
    __shared__ unsigned long long v[kRankSortMax];
    const long long row = blockIdx.y;
    const int c = blockIdx.x, t = threadIdx.x;
    RankRow &r = rr[row];
    if (c >= r.slots1) {
        return;
    }
    const int count = (int)r.count1[c];
    if (r.cursor1[c] != r.count1[c]) {  // (guard: every counted value must have arrived)
        if (t == 0) {
            atomicOr(&trows[row].tie, 1);
        }
        return;
    }
    int padded = 64;
    while (padded < count) {
        padded <<= 1;
    }
    const unsigned long long *__restrict__ in = cand + row * capacity + r.offset1[c];
    for (int i = t; i < padded; i += 256) {
        v[i] = (i < count) ? in[i] : ~0ULL;
    }
    __syncthreads();
    for (int k = 2; k <= padded; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < padded; i += 256) {
                const int partner = i ^ j;
                if (partner > i) {
                    const unsigned long long a = v[i], b = v[partner];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        v[i] = b;
                        v[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (t < kRankTargets && r.rank[t] >= 0 && (int)r.slot[t] == c) {
        r.key[t] = (r.rank[t] < count) ? v[r.rank[t]] : ~0ULL;
    }
}

// one wavefront per row: lane b = bin b; the boundaries, the tie check and the median x from the ranks' keys
__global__ __launch_bounds__(64) void rank_finish_kernel(const RankRow *__restrict__ rr, long long n, int bins, TrendRow *__restrict__ trows)
{
    const long long row = blockIdx.x;
    const int b = threadIdx.x;
    const RankRow &r = rr[row];
    TrendRow &tr = trows[row];
    if (b >= bins || tr.tie != 0) {
        return;
    }
    const long long left = ((long long)b * n) / bins, right = ((long long)(b + 1) * n) / bins, w = right - left;
    if (w <= 0) {
        atomicOr(&tr.tie, 1);  // an empty bin
        return;
    }
    const unsigned long long *key = r.key + 4 * b;
    tr.cov[b] = (w & 1LL) ? bits_to_double(key[3]) : 0.5 * (bits_to_double(key[2]) + bits_to_double(key[3]));
    tr.bound[b] = (b >= 1) ? bits_to_double(key[1]) : -1.0;
    if (b >= 1 && left >= 1 && key[0] == key[1]) {
        atomicOr(&tr.tie, 1);  // a boundary inside a run of equal x
    }
}

constexpr int kDealChunk = 4096;  // pairs per workgroup of the dealing kernel (16 per thread)

// every pair's y to its bin's segment of ypart (row-major, bin b of a row at [left_b, right_b)); cursor[row][b] counts
__global__ __launch_bounds__(256) void wls_deal_kernel(const double *__restrict__ matrix, const double *__restrict__ vas, long long n,
                                                      long long half, long long max_start, int bins, const TrendRow *__restrict__ rows,
                                                      unsigned *__restrict__ cursor, double *__restrict__ ypart, int *__restrict__ bad)
{
    __shared__ double sb[kMaxBins];
    __shared__ unsigned cnt[kMaxBins], base[kMaxBins], before[kMaxBins], total;
    __shared__ long long left_of[kMaxBins], right_of[kMaxBins];
    __shared__ double stage[kDealChunk];
    __shared__ unsigned char stage_bin[kDealChunk];
    const long long r = blockIdx.y;
    const TrendRow &tr = rows[r];
    if (tr.tie != 0) {
        return;
    }
    const int t = threadIdx.x;
    if (t < kMaxBins) {
        sb[t] = (t >= 1 && t < bins) ? tr.bound[t] : ((t == 0) ? -1.0 : INFINITY);
        cnt[t] = 0u;
    }
    __syncthreads();
    const double *__restrict__ row = matrix + r * n;
    const double *__restrict__ vas_row = vas + r * (max_start + 1);
    const long long first = (long long)blockIdx.x * kDealChunk;
    unsigned char mine[kDealChunk / 256];
    double held[8];  // (eight loads in flight per thread, at clamped positions: a load under `if (i < n)` is waited for by itself)
#pragma unroll
    for (int j = 0; j < kDealChunk / 256; ++j) {
        if ((j & 7) == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long iu = first + t + 256LL * (j + u);
                held[u] = row[(iu < n) ? iu : (n - 1)];
            }
        }
        const long long i = first + t + 256LL * j;
        int b = 0;
        if (i < n) {
            const double x = fabs(held[j & 7]);
            // the bin: how many boundaries lie at or below x (sb ascending, +inf beyond the last bin)
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                b += (b + step < kMaxBins && sb[b + step] <= x) ? step : 0;
            }
            atomicAdd(&cnt[b], 1u);
        }
        mine[j] = (unsigned char)b;
    }
    __syncthreads();
    // Round 5: the block's values go to memory SORTED BY BIN (through LDS), a bin's values side by side -- a wavefront's store is then a
    // few runs of consecutive addresses instead of 64 lone 8-byte writes (each lane's value used to go wherever its bin's cursor stood)
    if (t < kMaxBins) {  // (one wavefront: the bins' counts -> their places in the block's sorted order, and in the row's segments)
        const unsigned c = (t < bins) ? cnt[t] : 0u;
        base[t] = (c != 0u) ? atomicAdd(&cursor[r * kMaxBins + t], c) : 0u;
        unsigned run = c;
#pragma unroll
        for (int off = 1; off < kMaxBins; off <<= 1) {
            const unsigned up = __shfl_up(run, off);
            run += (t >= off) ? up : 0u;
        }
        before[t] = run - c;
        left_of[t] = ((long long)t * n) / bins;
        right_of[t] = ((long long)(t + 1) * n) / bins;
        cnt[t] = 0u;
        if (t == kMaxBins - 1) {
            total = run;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kDealChunk / 256; ++j) {
        if ((j & 7) == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long iu = first + t + 256LL * (j + u);
                held[u] = obs_variance_at(vas_row, (iu < n) ? iu : (n - 1), half, max_start);
            }
        }
        const long long i = first + t + 256LL * j;
        if (i < n) {
            const int b = mine[j];
            const double y = fmax(held[j & 7], 1.0e-8);  // wls_backend.c:429-430
            if (!isfinite(y)) {
                atomicOr(bad, 1);
            }
            const unsigned slot = before[b] + atomicAdd(&cnt[b], 1u);
            stage[slot] = y;
            stage_bin[slot] = (unsigned char)b;
        }
    }
    __syncthreads();
    double *__restrict__ out = ypart + r * n;
    for (unsigned e = (unsigned)t; e < total; e += 256u) {
        const int b = stage_bin[e];
        const long long pos = left_of[b] + (long long)base[b] + (long long)(e - before[b]);
        if (pos < right_of[b]) {  // (never false when the boundaries are what they should be: checked by the cursor afterwards)
            out[pos] = stage[e];
        }
    }
}

// median of every (row, bin) segment of ypart: the radix select of the row medians, one "row" per segment
typedef RowSelect SegSelect;

__device__ __forceinline__ void segment_of(long long seg, long long n, int bins, long long &offset, long long &width)
{
    const long long r = seg / bins, b = seg % bins;
    const long long left = (b * n) / bins, right = ((b + 1) * n) / bins;
    offset = r * n + left;
    width = right - left;
}

__global__ __launch_bounds__(256) void seg_select_init_kernel(SegSelect *__restrict__ state, unsigned *__restrict__ hist, long long segs,
                                                             long long n, int bins, unsigned *__restrict__ cursor, long long rows,
                                                             unsigned *__restrict__ filled)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < segs) {
        long long off, w;
        segment_of(i, n, bins, off, w);
        filled[i] = 0u;
        state[i].prefix = 0ULL;
        state[i].rank = (w - 1) / 2;
        state[i].above = ~0ULL;
        state[i].count_le = 0;
    }
    if (i < segs * kSelectBuckets) {
        hist[i] = 0u;
    }
    if (i < rows * kMaxBins) {
        cursor[i] = 0u;
    }
}

__global__ __launch_bounds__(256) void seg_select_count_kernel(const double *__restrict__ ypart, long long n, int bins, int low, int width,
                                                              const SegSelect *__restrict__ state, const TrendRow *__restrict__ rows,
                                                              unsigned *__restrict__ hist)
{
    __shared__ unsigned local[kSelectBuckets];
    const long long seg = blockIdx.y;
    if (rows[seg / bins].tie != 0 || state[seg].rank < 0) {  // (a negative rank: settled from its gathered cell already)
        return;
    }
    long long off, w;
    segment_of(seg, n, bins, off, w);
    const long long base = (long long)blockIdx.x * kSelectChunk;
    if (base >= w) {
        return;
    }
    for (int b = threadIdx.x; b < kSelectBuckets; b += 256) {
        local[b] = 0u;
    }
    __syncthreads();
    const unsigned long long prefix = state[seg].prefix;
    const int above_bits = low + width;
    const unsigned long long mask = (1ULL << width) - 1ULL;
    const double *__restrict__ y = ypart + off;
    // (eight loads in flight per thread, at clamped positions: a load under `if (i < w)` is waited for before the next is issued)
    for (int j0 = 0; j0 < kSelectChunk / 256; j0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = base + threadIdx.x + 256LL * (j0 + u);
            v[u] = y[(i < w) ? i : (w - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = base + threadIdx.x + 256LL * (j0 + u);
            const unsigned long long k = (unsigned long long)__double_as_longlong(v[u]);  // y > 0: bit order = numeric order
            lds_count(local, (unsigned)((k >> low) & mask), i < w && (above_bits >= 64 || (k >> above_bits) == prefix));
        }
    }
    __syncthreads();
    unsigned *__restrict__ mine = hist + seg * kSelectBuckets;
    for (int b = threadIdx.x; b < kSelectBuckets; b += 256) {
        if (local[b] != 0u) {
            atomicAdd(&mine[b], local[b]);
        }
    }
}

__global__ __launch_bounds__(256) void seg_select_above_kernel(const double *__restrict__ ypart, long long n, int bins,
                                                              SegSelect *__restrict__ state, const TrendRow *__restrict__ rows)
{
    const long long seg = blockIdx.y;
    if (rows[seg / bins].tie != 0 || state[seg].rank < 0) {
        return;
    }
    long long off, w;
    segment_of(seg, n, bins, off, w);
    const long long base = (long long)blockIdx.x * kSelectChunk;
    if (base >= w) {
        return;
    }
    const unsigned long long want = state[seg].prefix;
    const double *__restrict__ y = ypart + off;
    long long le = 0;
    unsigned long long above = ~0ULL;
#pragma unroll 4
    for (int j = 0; j < kSelectChunk / 256; ++j) {
        const long long i = base + threadIdx.x + 256LL * j;
        if (i < w) {
            const unsigned long long k = (unsigned long long)__double_as_longlong(y[i]);
            le += (k <= want) ? 1 : 0;
            above = (k > want && k < above) ? k : above;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        le += __shfl_xor(le, o);
        const unsigned long long other = __shfl_xor(above, o);
        above = (other < above) ? other : above;
    }
    if ((threadIdx.x & 63) == 0) {
        if (le != 0) {
            atomicAdd((unsigned long long *)&state[seg].count_le, (unsigned long long)le);
        }
        if (above != ~0ULL) {
            atomicMin(&state[seg].above, above);
        }
    }
}

// After two passes (22 key bits) the cell that holds a segment's median has a few hundred values when the variances are
// spread as variances are: instead of four more counting passes and one for the upper middle value, ONE pass gathers the cell
// (and finds the smallest key of the cells above it), and a workgroup per segment sorts the cell in LDS and reads the two
// middle values off.  A cell with more than kCellMax values (runs of equal variances) leaves its segment to the remaining
// passes as before; settled segments carry a negative rank and those passes skip them.
constexpr int kCellMax = 4096;
constexpr int kCellBits = 22;  // key bits known after two passes

__global__ __launch_bounds__(256) void seg_gather_kernel(const double *__restrict__ ypart, long long n, int bins, SegSelect *__restrict__ state,
                                                        const TrendRow *__restrict__ rows, const unsigned *__restrict__ bucket,
                                                        unsigned long long *__restrict__ cand, unsigned *__restrict__ filled)
{
    const long long seg = blockIdx.y;
    if (rows[seg / bins].tie != 0 || bucket[seg] > (unsigned)kCellMax) {
        return;
    }
    long long off, w;
    segment_of(seg, n, bins, off, w);
    const long long base = (long long)blockIdx.x * kSelectChunk;
    if (base >= w) {
        return;
    }
    const unsigned long long prefix = state[seg].prefix;
    const double *__restrict__ y = ypart + off;
    unsigned long long above = ~0ULL;
    for (int j0 = 0; j0 < kSelectChunk / 256; j0 += 8) {
      double v[8];  // (eight loads in flight per thread, at clamped positions)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long iu = base + threadIdx.x + 256LL * (j0 + u);
        v[u] = y[(iu < w) ? iu : (w - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long i = base + threadIdx.x + 256LL * (j0 + u);
        if (i < w) {
            const unsigned long long k = (unsigned long long)__double_as_longlong(v[u]);
            const unsigned long long cell = k >> (64 - kCellBits);
            if (cell == prefix) {
                const unsigned at = atomicAdd(&filled[seg], 1u);
                if (at < (unsigned)kCellMax) {
                    cand[seg * kCellMax + at] = k;
                }
            } else if (cell > prefix && k < above) {
                above = k;
            }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(above, o);
        above = (other < above) ? other : above;
    }
    if ((threadIdx.x & 63) == 0 && above != ~0ULL) {
        atomicMin(&state[seg].above, above);
    }
}

__global__ __launch_bounds__(256) void seg_settle_kernel(SegSelect *__restrict__ state, const TrendRow *__restrict__ rows, int bins,
                                                        const unsigned *__restrict__ bucket, const unsigned long long *__restrict__ cand,
                                                        const unsigned *__restrict__ filled)
{
    __shared__ unsigned long long v[kCellMax];
    const long long seg = blockIdx.x;
    const int count = (int)bucket[seg];
    if ((rows != nullptr && rows[seg / bins].tie != 0) || count > kCellMax || count == 0 || (int)filled[seg] != count) {
        return;  // (left to the remaining passes)
    }
    int padded = 64;
    while (padded < count) {
        padded <<= 1;
    }
    for (int i = threadIdx.x; i < padded; i += 256) {
        v[i] = (i < count) ? cand[seg * kCellMax + i] : ~0ULL;
    }
    __syncthreads();
    for (int k = 2; k <= padded; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < padded; i += 256) {
                const int partner = i ^ j;
                if (partner > i) {
                    const unsigned long long a = v[i], b = v[partner];
                    if ((a > b) == ((i & k) == 0)) {
                        v[i] = b;
                        v[partner] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        SegSelect &st = state[seg];
        const long long rank = st.rank;  // inside the cell
        if (rank >= 0 && rank < count) {
            st.prefix = v[rank];
            if (rank + 1 < count) {
                st.above = v[rank + 1];  // the upper middle value (equal to the lower one inside a run)
            }
            st.count_le = 0;  // (the finish kernel then takes `above` for the upper middle value)
            st.rank = -1;     // settled
        }
    }
}

// the same for the row medians of the count-path glue (keys: order_key, any sign); `seg_settle_kernel` with rows == nullptr
// settles them
__global__ __launch_bounds__(256) void row_gather_kernel(const double *__restrict__ matrix, long long n, RowSelect *__restrict__ state,
                                                        const unsigned *__restrict__ bucket, unsigned long long *__restrict__ cand,
                                                        unsigned *__restrict__ filled)
{
    const long long row = blockIdx.y;
    if (bucket[row] > (unsigned)kCellMax) {
        return;
    }
    const unsigned long long prefix = state[row].prefix;
    const double *__restrict__ x = matrix + row * n;
    const long long base = (long long)blockIdx.x * kSelectChunk;
    unsigned long long above = ~0ULL;
#pragma unroll 4
    for (int j = 0; j < kSelectChunk / 256; ++j) {
        const long long i = base + threadIdx.x + 256LL * j;
        if (i < n) {
            const unsigned long long k = order_key(x[i]);
            const unsigned long long cell = k >> (64 - kCellBits);
            if (cell == prefix) {
                const unsigned at = atomicAdd(&filled[row], 1u);
                if (at < (unsigned)kCellMax) {
                    cand[row * kCellMax + at] = k;
                }
            } else if (cell > prefix && k < above) {
                above = k;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(above, o);
        above = (other < above) ? other : above;
    }
    if ((threadIdx.x & 63) == 0 && above != ~0ULL) {
        atomicMin(&state[row].above, above);
    }
}

// the segment medians into the rows' tables; a row whose pairs did not fill every bin exactly (cannot happen with clean
// boundaries; kept as a guard) is sent to the sorted path
__global__ __launch_bounds__(64) void seg_select_finish_kernel(const SegSelect *__restrict__ state, long long n, int bins,
                                                              const unsigned *__restrict__ cursor, TrendRow *__restrict__ rows)
{
    const long long r = blockIdx.x;
    const int b = threadIdx.x;
    if (b >= bins || rows[r].tie != 0) {
        return;
    }
    long long off, w;
    segment_of(r * bins + b, n, bins, off, w);
    if ((long long)cursor[r * kMaxBins + b] != w) {
        atomicOr(&rows[r].tie, 1);
        return;
    }
    const SegSelect &st = state[r * bins + b];
    const double lo = bits_to_double(st.prefix);
    double med = lo;
    if ((w & 1LL) == 0) {
        const double hi = (st.count_le >= w / 2 + 1) ? lo : bits_to_double(st.above);
        med = 0.5 * (lo + hi);
    }
    rows[r].var[b] = med;
}

__global__ __launch_bounds__(256) void wls_tie_flags_kernel(const TrendRow *__restrict__ rows, long long count, int *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < count) {
        out[i] = rows[i].tie;
    }
}

// one workgroup per row: bins -> pooled fit -> knots, from the row's table (the same steps as wls_knots_kernel below)
__global__ __launch_bounds__(64) void wls_knots_rows_kernel(const TrendRow *__restrict__ rows, long long n, int bins, TrendFit *fits)
{
    __shared__ double bc[kMaxBins], bv[kMaxBins], bw[kMaxBins], fitv[kMaxBins];
    __shared__ long long bl[kMaxBins];
    const long long r = blockIdx.x;
    const TrendRow &tr = rows[r];
    if (tr.tie != 0 || threadIdx.x != 0) {
        return;
    }
    TrendFit *fit = fits + r;
    int used = 0;
    for (int b = 0; b < bins; ++b) {
        const long long width = (((long long)(b + 1) * n) / bins) - (((long long)b * n) / bins);
        if (width > 0) {
            bc[used] = tr.cov[b];
            bv[used] = tr.var[b];
            bw[used] = (double)width;
            ++used;
        }
    }
    fit->mode = 0;
    fit->knots = 0;
    if (used == 1) {
        fit->value = fmax(bv[0], 1.0e-8);
        return;
    }
    int nb = 0;
    for (int i = 0; i < used; ++i) {  // pool adjacent violators (wls_backend.c:262-339)
        fitv[nb] = bv[i];
        bw[nb] = fmax(bw[i], 1.0e-8);
        bl[nb] = 1;
        ++nb;
        while (nb >= 2 && fitv[nb - 2] > fitv[nb - 1]) {
            const double tw = bw[nb - 2] + bw[nb - 1];
            const double mv = ((fitv[nb - 2] * bw[nb - 2]) + (fitv[nb - 1] * bw[nb - 1])) / tw;
            fitv[nb - 2] = mv;
            bw[nb - 2] = tw;
            bl[nb - 2] += bl[nb - 1];
            --nb;
        }
    }
    int knots = 0, idx = 0;
    for (int b = 0; b < nb; ++b) {  // expand blocks and build the knots (wls_backend.c:541-552)
        for (long long q = 0; q < bl[b]; ++q, ++idx) {
            const double cv = bc[idx], vv = fmax(fitv[b], 1.0e-8);
            if (knots > 0 && cv <= fit->kc[knots - 1]) {
                fit->kv[knots - 1] = fmax(fit->kv[knots - 1], vv);
                continue;
            }
            fit->kc[knots] = cv;
            fit->kv[knots] = vv;
            ++knots;
        }
    }
    fit->knots = knots;
    if (knots == 1) {
        fit->value = fmax(fit->kv[0], 1.0e-8);
    } else {
        fit->mode = 2;
    }
}

// wls_backend.c:341-392
__device__ __forceinline__ double linear_interp(const double *xs, const double *ys, int count, double t)
{
    if (count == 1 || t <= xs[0]) {
        return ys[0];
    }
    if (t >= xs[count - 1]) {
        return ys[count - 1];
    }
    int left = 0, right = count - 1;
    while (right - left > 1) {
        const int mid = left + (right - left) / 2;
        if (xs[mid] <= t) {
            left = mid;
        } else {
            right = mid;
        }
    }
    if (xs[right] <= xs[left]) {
        return fmax(ys[right], ys[left]);
    }
    const double w = (t - xs[left]) / (xs[right] - xs[left]);
    return ys[left] + (w * (ys[right] - ys[left]));
}

// sums = weighted | precision | raw | prior, n each
// every row into the per-locus sums, one thread per locus walking the rows in order (the order the reference adds them in,
// wls_backend.c:885-910) with the four sums in registers: 16 B read per value instead of 80 (the sums no longer go
// through memory once per row), one launch per matrix instead of one per row
__global__ __launch_bounds__(256) void wls_accumulate_rows_kernel(const double *__restrict__ matrix, const double *__restrict__ vas,
                                                                 long long rows, long long n, long long half, long long max_start,
                                                                 const TrendFit *__restrict__ fits, double local_df, double prior_df,
                                                                 double total_df, double floor_ratio, double *__restrict__ sums)
{
    __shared__ double kc[2][kMaxBins], kv[2][kMaxBins];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long vas_stride = max_start + 1;
    if (threadIdx.x < kMaxBins) {
        kc[0][threadIdx.x] = fits[0].kc[threadIdx.x];
        kv[0][threadIdx.x] = fits[0].kv[threadIdx.x];
    }
    double weighted = 0.0, precision = 0.0, raw = 0.0, prior = 0.0;
    for (long long k = 0; k < rows; ++k) {
        __syncthreads();  // row k's knots are in place; everyone is done with row k - 1's
        const int cur = (int)(k & 1LL);
        if (k + 1 < rows && threadIdx.x < kMaxBins) {
            kc[cur ^ 1][threadIdx.x] = fits[k + 1].kc[threadIdx.x];
            kv[cur ^ 1][threadIdx.x] = fits[k + 1].kv[threadIdx.x];
        }
        if (i < n) {
            const TrendFit &fit = fits[k];
            const double value = matrix[k * n + i];
            const double obs_value = fmax(obs_variance_at(vas + k * vas_stride, i, half, max_start), 1.0e-8);
            double prior_track = fit.value;
            if (fit.mode == 2) {
                prior_track = fmax(linear_interp(kc[cur], kv[cur], fit.knots, fabs(value)), 1.0e-8);  // wls_backend.c:590-592
            }
            const double prior_value = fmax(prior_track, 1.0e-8);
            double posterior_variance = ((local_df * obs_value) + (prior_df * prior_value)) / fmax(total_df, 1.0);
            const double variance_floor = floor_ratio * prior_value;
            if (posterior_variance < variance_floor) {
                posterior_variance = variance_floor;
            }
            posterior_variance = fmax(posterior_variance, 1.0e-8);
            const double posterior_precision = 1.0 / posterior_variance;
            raw += 1.0 / obs_value;
            prior += 1.0 / prior_value;
            precision += posterior_precision;
            weighted += posterior_precision * value;
        }
    }
    if (i < n) {
        sums[2 * n + i] = raw;
        sums[3 * n + i] = prior;
        sums[1 * n + i] = precision;
        sums[0 * n + i] = weighted;
    }
}

// rows of fewer than 5 loci (no window): robust scale of the row for every locus (wls_backend.c:834-851)
__global__ void wls_tiny_kernel(const double *__restrict__ matrix, long long rows, long long n, double local_df,
                                double prior_df, double total_df, double floor_ratio, double *__restrict__ sums)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) {
        return;
    }
    for (long long k = 0; k < rows; ++k) {
        double w[4];
        for (long long i = 0; i < n; ++i) {
            w[i] = matrix[k * n + i];
        }
        auto median_small = [&](double *v) {  // order statistics of at most 4 values
            for (int a = 1; a < (int)n; ++a) {
                const double key = v[a];
                int b = a;
                while (b > 0 && v[b - 1] > key) {
                    v[b] = v[b - 1];
                    --b;
                }
                v[b] = key;
            }
            if (n == 1) {
                return v[0];
            }
            return (n & 1LL) ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
        };
        const double med = median_small(w);
        for (long long i = 0; i < n; ++i) {
            w[i] = fabs(matrix[k * n + i] - med);
        }
        // (the reference takes |work - median| of the partially ordered work buffer: same multiset)
        double mad = median_small(w);
        mad *= 1.4826;
        double sf = (mad > 1.0e-6) ? mad : 1.0e-6;
        sf = fmax(sf * sf, 1.0e-8);
        for (long long i = 0; i < n; ++i) {
            const double obs_value = fmax(sf, 1.0e-8), prior_value = fmax(sf, 1.0e-8);
            double pv = ((local_df * obs_value) + (prior_df * prior_value)) / fmax(total_df, 1.0);
            const double vf = floor_ratio * prior_value;
            if (pv < vf) {
                pv = vf;
            }
            pv = fmax(pv, 1.0e-8);
            const double prec = 1.0 / pv;
            sums[2 * n + i] += 1.0 / obs_value;
            sums[3 * n + i] += 1.0 / prior_value;
            sums[1 * n + i] += prec;
            sums[0 * n + i] += prec * matrix[k * n + i];
        }
    }
}

// wls_backend.c:913-937
__global__ __launch_bounds__(256) void wls_final_kernel(const double *__restrict__ sums, long long n, double sample_count,
                                                       double lower_bound_z, double min_effect, int use_min_effect,
                                                       double *__restrict__ mean, double *__restrict__ raw_var,
                                                       double *__restrict__ prior_var, double *__restrict__ mod_var,
                                                       double *__restrict__ se, double *__restrict__ scores)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const double locus_precision = fmax(sums[1 * n + i], 1.0e-8);
    const double m = sums[0 * n + i] / locus_precision;
    mean[i] = m;
    raw_var[i] = sample_count / fmax(sums[2 * n + i], 1.0e-8);
    prior_var[i] = sample_count / fmax(sums[3 * n + i], 1.0e-8);
    mod_var[i] = sample_count / locus_precision;
    const double s = sqrt(1.0 / locus_precision);
    se[i] = s;
    const double z = m / fmax(s, 1.0e-8);
    if (use_min_effect != 0) {
        scores[i] = (m - fmax(min_effect, 0.0)) / fmax(s, 1.0e-8);
    } else {
        scores[i] = z - lower_bound_z;
    }
}

// Self-check of the two-stage log2 (log2_cr.h): `count` inputs of a family -- 0: any positive finite bit pattern; 1: 1 +- tiny
// (the one region where the result is small); 2: the integers first .. first + count - 1 (counts + pseudocount); 3: mantissas
// next to the table's cell boundaries and centres in random binades; 4: uniform in [0.5, 4) -- through both stages and
// through the full evaluation alone: out[0] += inputs whose results differ.
__global__ __launch_bounds__(256) void log2_selfcheck_kernel(int family, unsigned long long seed, unsigned long long first, long long count,
                                                            unsigned long long *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) {
        return;
    }
    unsigned long long x = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i + 1);  // splitmix64
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= x >> 31;
    double t;
    if (family == 0) {
        unsigned long long b = x & 0x7FFFFFFFFFFFFFFFULL;
        if ((b >> 52) == 0x7FF) {
            b &= 0x7FEFFFFFFFFFFFFFULL;
        }
        t = __longlong_as_double((long long)(b == 0 ? 1ULL : b));
    } else if (family == 1) {
        const int shift = (int)(x >> 58) % 40;                       // distance from 1: 2^-13 .. 2^-52
        const long long steps = (long long)((x >> 12) & ((1ULL << (52 - 13)) - 1)) >> shift;
        t = ((x >> 11) & 1ULL) ? 1.0 + (double)steps * 0x1p-52 : 1.0 - (double)steps * 0x1p-53;
    } else if (family == 2) {
        t = (double)(first + (unsigned long long)i);
    } else if (family == 3) {
        const int cell = (int)(x % 194ULL);                          // boundaries (odd) and centres (even) of the 97 cells
        const double edge = 0.75 + (double)cell * 0.00390625 - 0.00390625;
        const long long ulps = (long long)((x >> 8) & 0xFFFF) - 0x8000;
        const double m = edge + (double)ulps * 0x1p-53;
        const int e = (int)((x >> 24) % 2046ULL) - 1022;
        t = ldexp(m < 0.7421875 ? 0.7421875 : m, e);
        if (!(t > 0.0) || !(t < INFINITY)) {
            t = m;
        }
    } else {
        t = 0.5 + (double)(x >> 11) * (3.5 * 0x1p-53);
    }
    if (log2_correctly_rounded(t) != log2_cr_full(t)) {
        atomicAdd(out, 1ULL);
    }
}

// ---- row a2 glue: log scale, pilot offset, centring (rocco/inference.py:40-47, 330-336) ----------------
__global__ __launch_bounds__(256) void log_scale_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                       long long count, double pseudocount, int apply_log,
                                                       int *__restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) {
        return;
    }
    const double v = in[i];
    if (!isfinite(v)) {
        atomicOr(bad, 1);
    }
    // np.log2(np.clip(matrix, 0.0, None) + pseudocount): correctly rounded (log2_cr.h; NumPy's own log2 is one ulp off
    // that in ~0.03 % of counts, differently on its SVML and libm builds)
    if (apply_log) {
        const double t = fmax(v, 0.0) + pseudocount;
        out[i] = (t > 0.0 && t < INFINITY) ? log2_correctly_rounded(t) : log2(t);
    } else {
        out[i] = v;
    }
}

// The log scale and the row medians' first counting pass in one (round 5): what log_scale_kernel computes and stores is
// counted on the way -- top 11 bits of its order key, row_select_count_kernel's pass 0 -- instead of being read back for
// it: one pass over the matrix less.  grid: (chunks of kSelectChunk loci, rows), as the counting passes'.
__global__ __launch_bounds__(256) void log_scale_count_kernel(const double *__restrict__ in, double *__restrict__ out, long long n,
                                                             double pseudocount, int apply_log, int *__restrict__ bad,
                                                             unsigned *__restrict__ hist)
{
    __shared__ unsigned local[kSelectBuckets];
    for (int b = threadIdx.x; b < kSelectBuckets; b += 256) {
        local[b] = 0u;
    }
    __syncthreads();
    const long long row = blockIdx.y, base = (long long)blockIdx.x * kSelectChunk;
    const double *__restrict__ x = in + row * n;
    double *__restrict__ y = out + row * n;
    constexpr int kLow = 53;  // (the first pass: bits 53 .. 63 of the key)
#pragma unroll 4
    for (int j = 0; j < kSelectChunk / 256; ++j) {
        const long long i = base + threadIdx.x + 256LL * j;
        double r = 0.0;
        if (i < n) {
            const double v = x[i];
            if (!isfinite(v)) {
                atomicOr(bad, 1);
            }
            r = v;
            if (apply_log) {
                const double t = fmax(v, 0.0) + pseudocount;
                r = (t > 0.0 && t < INFINITY) ? log2_correctly_rounded(t) : log2(t);
            }
            y[i] = r;
        }
        if (i < n) {  // (a plain add: this kernel is bound by its logarithm, and lds_count's ballots cost it 6 ms per genome)
            atomicAdd(&local[(unsigned)(order_key(r) >> kLow)], 1u);
        }
    }
    __syncthreads();
    unsigned *__restrict__ mine = hist + row * kSelectBuckets;
    for (int b = threadIdx.x; b < kSelectBuckets; b += 256) {
        if (local[b] != 0u) {
            atomicAdd(&mine[b], local[b]);
        }
    }
}

// grid: (chunks of 1024 loci, rows); four loci per thread (no division to find the row)
__global__ __launch_bounds__(256) void subtract_row_offset_kernel(double *__restrict__ matrix, const double *__restrict__ med,
                                                                 long long n)
{
    const long long row = blockIdx.y;
    const double offset = med[row];
    double *__restrict__ x = matrix + row * n;
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long i = base + 256LL * j;
        if (i < n) {
            x[i] = x[i] - offset;
        }
    }
}

// out = a - b; `bad` (may be null) is raised when b holds a non-finite value (the check the reference makes on its local
// baselines, rocco/inference.py:207-208, without a pass of its own)
__global__ __launch_bounds__(256) void subtract_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                      double *__restrict__ out, long long count, int *__restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const double bv = b[i];
        if (bad != nullptr && !isfinite(bv)) {
            atomicOr(bad, 1);
        }
        out[i] = a[i] - bv;
    }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

int wls_spatial_window(size_t n, int requested)
{
    // wls_backend.c:232-260
    if (n < 5) {
        return 0;
    }
    size_t w = requested > 0 ? (size_t)requested : 31U;
    if (w < 5) {
        w = 5;
    }
    if (w > n) {
        w = n;
    }
    if ((w & 1U) == 0) {
        w = (w == n) ? (w - 1) : (w + 1);
    }
    return (w < 5) ? 0 : (int)w;
}

int wls_max_window() { return kMaxWindow; }

namespace {

__global__ void wls_rolling_tasks_kernel(WlsRollingTask *tasks, const double *matrix, long long K, long long n, int window, double *vas,
                                         int group)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g * group < K) {
        const long long first = g * group;
        tasks[g].row = matrix + first * n;
        tasks[g].n = n;
        tasks[g].window = window;
        tasks[g].rows = (int)((K - first < group) ? (K - first) : group);
        tasks[g].out = vas + first * (n - window + 1);
    }
}

// the rolling variances of one matrix: its groups' task records are written by a kernel (no host staging, no synchronisation)
int launch_wls_rolling_matrix(const double *matrix, size_t K, size_t n, int window, double *vas, WlsRollingTask *tasks_dev,
                              hipStream_t stream)
{
    const int group = wls_rolling_group_rows(K);
    const size_t groups = (K + group - 1) / group;
    hipLaunchKernelGGL(wls_rolling_tasks_kernel, dim3((unsigned)((groups + 63) / 64)), dim3(64), 0, stream, tasks_dev, matrix, (long long)K,
                       (long long)n, window, vas, group);
    return launch_wls_rolling_batch(tasks_dev, groups, group, stream);
}

}  // namespace

int trend_bins(size_t n)
{
    // wls_backend.c:461 -- the same expression, evaluated on the host like the reference's
    return (int)std::fmax(4.0, std::floor(1.0 + (std::log((double)n + 1.0) / std::log(2.0))));
}

using u64 = unsigned long long;

size_t sort_temp_bytes(size_t n)
{
    size_t a = 0, b = 0, c = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const u64 *)nullptr, (u64 *)nullptr, (const u64 *)nullptr,
                                             (u64 *)nullptr, (int)n);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const u64 *)nullptr, (u64 *)nullptr, (const unsigned *)nullptr,
                                             (unsigned *)nullptr, (int)n);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, c, (const unsigned char *)nullptr, (unsigned char *)nullptr,
                                             (const u64 *)nullptr, (u64 *)nullptr, (int)n, 0, 6);
    return align_up(std::max(a, std::max(b, c)), 256);
}

// room per row for the values of the cells that hold a rank (about 2 % of the row; a row that needs more sorts instead)
static size_t rank_capacity(size_t n) { return std::min(n, n / 4 + 65536); }

size_t wls_scratch_bytes(size_t K, size_t n, int spatial_window, bool own_variances)
{
    // windows above the tiled kernel's limit keep the three running sums of every row in memory
    const size_t general = (wls_spatial_window(n, spatial_window) > kMaxWindow) ? align_up(3 * K * n * 8, 256) : 0;
    // the sort-free trend fit (rows of kTrendSelectMin loci or more): the dealt y of every row, the rows' tables, the
    // segment selects' state and histograms, the cursors, one fit per row
    const size_t segs = K * (size_t)kMaxBins;
    const size_t dealt = (n >= (size_t)kTrendSelectMin)
                             ? align_up(K * n * 8, 256) + align_up(K * sizeof(TrendRow), 256) + align_up(segs * sizeof(RowSelect), 256) +
                                   align_up(segs * kSelectBuckets * sizeof(unsigned), 256) + 3 * align_up(segs * sizeof(unsigned), 256) +
                                   align_up(segs * (size_t)kCellMax * 8, 256) +
                                   align_up(K * sizeof(RankRow), 256) + align_up(K * (size_t)kRankBuckets0 * sizeof(unsigned), 256) +
                                   align_up(K * (size_t)kRankCells1 * sizeof(unsigned), 256) + align_up(K * rank_capacity(n) * 8, 256)
                             : 0;
    return general + dealt + align_up(K * sizeof(TrendFit), 256) + (own_variances ? align_up(K * n * 8, 256) : 0) + 6 * align_up(n * 8, 256) +
           2 * align_up(n * 4, 256) + 2 * align_up(n, 256) + align_up(4 * n * 8, 256) + sort_temp_bytes(n) + 4096 +
           align_up(K * sizeof(WlsRollingTask), 256);
}

int launch_score_centered_wls(const double *centered_dev, size_t K, size_t n, double lower_bound_z, double prior_df,
                              double min_effect, int use_min_effect, int spatial_window,
                              double precision_floor_ratio, double *mean_dev, double *raw_var_dev,
                              double *prior_var_dev, double *mod_var_dev, double *se_dev, double *scores_dev,
                              void *scratch_dev, double *df_out, int *window_out, hipStream_t stream, int *flag_host_pinned,
                              const double *vas_given, int *sorted_rows_out)
{
    if (sorted_rows_out != nullptr) {
        *sorted_rows_out = 0;
    }
    const double pdf = std::fmax(prior_df, 0.0), floor_ratio = std::fmax(precision_floor_ratio, 0.0);
    const int window = wls_spatial_window(n, spatial_window);
    const double local_df = window > 0 ? std::fmax(4.0, (double)window - 3.0) : 1.0;
    const double total_df = local_df + pdf;
    if (df_out != nullptr) {
        *df_out = total_df;
    }
    if (window_out != nullptr) {
        *window_out = window;
    }
    const long long nn = (long long)n;
    char *sc = (char *)scratch_dev;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        char *p = sc + off;
        off += align_up(bytes, 256);
        return p;
    };
    // given: the local variances of a batched rolling launch (launch_wls_rolling_batch), no room of their own in the scratch
    double *vas = (vas_given != nullptr) ? const_cast<double *>(vas_given) : (double *)carve(K * n * 8);
    u64 *key_a = (u64 *)carve(n * 8), *key_b = (u64 *)carve(n * 8), *val_a = (u64 *)carve(n * 8);
    u64 *val_b = (u64 *)carve(n * 8), *key_c = (u64 *)carve(n * 8), *key_d = (u64 *)carve(n * 8);
    unsigned *iota = (unsigned *)carve(n * 4), *perm = (unsigned *)carve(n * 4);
    unsigned char *bin_a = (unsigned char *)carve(n), *bin_b = (unsigned char *)carve(n);
    double *sums = (double *)carve(4 * n * 8);
    TrendFit *fits = (TrendFit *)carve(K * sizeof(TrendFit));  // one per row
    int *bad = (int *)carve(256);
    void *rolling_tasks = carve(K * sizeof(WlsRollingTask));  // (one record per group of 1 .. 8 rows)
    double *general_sums = (window > kMaxWindow) ? (double *)carve(3 * K * n * 8) : nullptr;
    const bool select_path = nn >= kTrendSelectMin && window > 0;
    const size_t segs = K * (size_t)kMaxBins;
    double *ypart = select_path ? (double *)carve(K * n * 8) : nullptr;
    TrendRow *trows = select_path ? (TrendRow *)carve(K * sizeof(TrendRow)) : nullptr;
    RowSelect *seg_state = select_path ? (RowSelect *)carve(segs * sizeof(RowSelect)) : nullptr;
    unsigned *seg_hist = select_path ? (unsigned *)carve(segs * kSelectBuckets * sizeof(unsigned)) : nullptr;
    unsigned *cursor = select_path ? (unsigned *)carve(segs * sizeof(unsigned)) : nullptr;
    unsigned *seg_bucket = select_path ? (unsigned *)carve(segs * sizeof(unsigned)) : nullptr;
    unsigned *seg_filled = select_path ? (unsigned *)carve(segs * sizeof(unsigned)) : nullptr;
    unsigned long long *seg_cand = select_path ? (unsigned long long *)carve(segs * (size_t)kCellMax * 8) : nullptr;
    RankRow *rank_rows = select_path ? (RankRow *)carve(K * sizeof(RankRow)) : nullptr;
    unsigned *rank_hist0 = select_path ? (unsigned *)carve(K * (size_t)kRankBuckets0 * sizeof(unsigned)) : nullptr;
    unsigned *rank_hist1 = select_path ? (unsigned *)carve(K * (size_t)kRankCells1 * sizeof(unsigned)) : nullptr;
    unsigned long long *rank_cand = select_path ? (unsigned long long *)carve(K * rank_capacity(n) * 8) : nullptr;
    void *tmp = sc + off;
    const size_t tmp_bytes = sort_temp_bytes(n);
    ROCCO_HIP_TRY(hipMemsetAsync(sums, 0, 4 * n * 8, stream));
    ROCCO_HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), stream));
    const unsigned blocks256 = (unsigned)((n + 255) / 256);
    if (window == 0 || n < 4) {
        hipLaunchKernelGGL(wls_tiny_kernel, dim3(1), dim3(64), 0, stream, centered_dev, (long long)K, nn, local_df, pdf,
                           total_df, floor_ratio, sums);
    } else {
        const long long half = window / 2, max_start = nn - window;
        const size_t vas_stride = (size_t)(max_start + 1);
        if (window <= kMaxWindow) {
            if (vas_given == nullptr) {
                int rc = launch_wls_rolling_matrix(centered_dev, K, n, window, vas, (WlsRollingTask *)rolling_tasks, stream);
                if (rc != ROCCO_HIP_OK) return rc;
            }
        } else {
            hipLaunchKernelGGL(wls_rolling_general_sums_kernel, dim3((unsigned)K), dim3(kLanes), 0, stream, centered_dev, nn, window,
                               general_sums);
            hipLaunchKernelGGL(wls_rolling_general_variance_kernel, dim3((unsigned)((max_start + 256) / 256), (unsigned)K), dim3(256), 0,
                               stream, centered_dev, nn, window, general_sums, vas);
        }
        const int bins = trend_bins(n);
        if (bins > kMaxBins) {  // (floor(1 + log2(n + 1)) <= 64 for every n < 2^63: unreachable, kept as a guard)
            set_last_error("rocco_hip_score_centered_wls_f64: more than 64 trend bins");
            return ROCCO_HIP_EINVAL;
        }
        // the sorted path of one row (short rows; rows whose bin boundaries fall into runs of equal |value|): the reference
        // sorts the pairs under the total order (x, then y), so the sorted sequence is unique
        auto sorted_row = [&](size_t k) -> int {
            const double *row = centered_dev + k * n;
            const double *vas_row = vas + k * vas_stride;
            hipLaunchKernelGGL(wls_pairs_kernel, dim3(blocks256), dim3(256), 0, stream, row, vas_row, nn, half, max_start,
                               key_a, val_a, bad);
            // by y carrying x (key_b = every y ascending, also the fallback median) ...
            size_t t = tmp_bytes;
            ROCCO_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, t, key_a, key_b, val_a, val_b, (int)n, 0, 64, stream));
            // ... then -- stable -- by x carrying the y-rank: key_c = x in (x, y) order, perm = y-rank by (x, y)-rank
            t = tmp_bytes;
            ROCCO_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, t, val_b, key_c, iota, perm, (int)n, 0, 64, stream));
            // bin medians of y: a stable sort of the y-sorted sequence by bin leaves every bin's y ascending
            hipLaunchKernelGGL(wls_bin_scatter_kernel, dim3(blocks256), dim3(256), 0, stream, perm, nn, bins, bin_a);
            t = tmp_bytes;
            ROCCO_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, t, bin_a, bin_b, key_b, key_d, (int)n, 0, 6, stream));
            hipLaunchKernelGGL(wls_knots_kernel, dim3(1), dim3(kMaxBins), 0, stream, key_b, key_c, key_d, nn, bins, fits + k);
            return ROCCO_HIP_OK;
        };
        hipLaunchKernelGGL(wls_iota_kernel, dim3(blocks256), dim3(256), 0, stream, iota, nn);
        std::vector<int> tie(K, 1);
        if (select_path) {
            // x side, every row of the matrix at once: the ranks the bins need (their first and median x), three passes
            hipLaunchKernelGGL(rank_init_kernel, dim3((unsigned)((K * (size_t)kRankBuckets0 + 255) / 256)), dim3(256), 0, stream, rank_rows,
                               rank_hist0, trows, (long long)K, nn, bins);
            const dim3 rank_grid((unsigned)((nn + kRankChunk - 1) / kRankChunk), (unsigned)K);
            hipLaunchKernelGGL(rank_hist0_kernel, rank_grid, dim3(kRankThreads), 0, stream, centered_dev, nn, rank_hist0, bad);
            hipLaunchKernelGGL(rank_plan0_kernel, dim3((unsigned)K), dim3(256), 0, stream, rank_rows, (const unsigned *)rank_hist0, rank_hist1,
                               trows);
            hipLaunchKernelGGL(rank_hist1_kernel, rank_grid, dim3(kRankThreads), 0, stream, centered_dev, nn, (const RankRow *)rank_rows, rank_hist1);
            hipLaunchKernelGGL(rank_plan1_kernel, dim3((unsigned)K), dim3(256), 0, stream, rank_rows, (const unsigned *)rank_hist1,
                               (long long)rank_capacity(n), trows);
            hipLaunchKernelGGL(rank_gather_kernel, dim3((unsigned)((nn + kGatherChunk - 1) / kGatherChunk), (unsigned)K), dim3(256), 0, stream,
                               centered_dev, nn, rank_rows, rank_cand, (long long)rank_capacity(n));
            hipLaunchKernelGGL(rank_sort_kernel, dim3(kRankTargets, (unsigned)K), dim3(256), 0, stream, rank_rows,
                               (const unsigned long long *)rank_cand, (long long)rank_capacity(n), trows);
            hipLaunchKernelGGL(rank_finish_kernel, dim3((unsigned)K), dim3(64), 0, stream, (const RankRow *)rank_rows, nn, bins, trows);
            // y side, every row of the matrix at once: deal the y to the bins' segments, select every segment's median
            const long long n_segs = (long long)K * bins;
            hipLaunchKernelGGL(seg_select_init_kernel, dim3((unsigned)((n_segs * kSelectBuckets + 255) / 256)), dim3(256), 0, stream,
                               seg_state, seg_hist, n_segs, nn, bins, cursor, (long long)K, seg_filled);
            hipLaunchKernelGGL(wls_deal_kernel, dim3((unsigned)((nn + kDealChunk - 1) / kDealChunk), (unsigned)K), dim3(256), 0, stream,
                               centered_dev, (const double *)vas, nn, half, max_start, bins, (const TrendRow *)trows, cursor, ypart, bad);
            const long long widest = (nn + bins - 1) / bins + 1;
            const dim3 seg_grid((unsigned)((widest + kSelectChunk - 1) / kSelectChunk), (unsigned)n_segs);
            const int lows[6] = {53, 42, 31, 20, 9, 0}, widths[6] = {11, 11, 11, 11, 11, 9};
            for (int p = 0; p < 6; ++p) {
                hipLaunchKernelGGL(seg_select_count_kernel, seg_grid, dim3(256), 0, stream, (const double *)ypart, nn, bins, lows[p],
                                   widths[p], (const RowSelect *)seg_state, (const TrendRow *)trows, seg_hist);
                hipLaunchKernelGGL(row_select_pick_kernel, dim3((unsigned)n_segs), dim3(256), 0, stream, seg_state, seg_hist,
                                   (p == 0) ? 64 : widths[p], (p == 1) ? seg_bucket : (unsigned *)nullptr);
                if (p == 1) {  // 22 bits known: gather the cells, settle the segments whose cell is small (nearly all)
                    hipLaunchKernelGGL(seg_gather_kernel, seg_grid, dim3(256), 0, stream, (const double *)ypart, nn, bins, seg_state,
                                       (const TrendRow *)trows, (const unsigned *)seg_bucket, seg_cand, seg_filled);
                    hipLaunchKernelGGL(seg_settle_kernel, dim3((unsigned)n_segs), dim3(256), 0, stream, seg_state, (const TrendRow *)trows, bins,
                                       (const unsigned *)seg_bucket, (const unsigned long long *)seg_cand, (const unsigned *)seg_filled);
                }
            }
            hipLaunchKernelGGL(seg_select_above_kernel, seg_grid, dim3(256), 0, stream, (const double *)ypart, nn, bins, seg_state,
                               (const TrendRow *)trows);
            hipLaunchKernelGGL(seg_select_finish_kernel, dim3((unsigned)K), dim3(64), 0, stream, (const RowSelect *)seg_state, nn, bins,
                               (const unsigned *)cursor, trows);
            hipLaunchKernelGGL(wls_knots_rows_kernel, dim3((unsigned)K), dim3(64), 0, stream, (const TrendRow *)trows, nn, bins, fits);
            // which rows must take the sorted path after all: K flags through the solver's pinned memory (behind the
            // first 64 ints of it, where the non-finite flag lands at the end)
            int *tie_dev = (int *)cursor;  // (the cursors have been checked: their room serves the flags)
            hipLaunchKernelGGL(wls_tie_flags_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, stream, (const TrendRow *)trows,
                               (long long)K, tie_dev);
            ROCCO_HIP_TRY(hipMemcpyAsync(flag_host_pinned + 64, tie_dev, K * sizeof(int), hipMemcpyDeviceToHost, stream));
            ROCCO_HIP_TRY(hipStreamSynchronize(stream));
            for (size_t k = 0; k < K; ++k) {
                tie[k] = flag_host_pinned[64 + k];
            }
        }
        for (size_t k = 0; k < K; ++k) {
            if (tie[k] != 0) {
                int rc = sorted_row(k);
                if (rc != ROCCO_HIP_OK) return rc;
                if (sorted_rows_out != nullptr) {
                    ++*sorted_rows_out;
                }
            }
        }
        hipLaunchKernelGGL(wls_accumulate_rows_kernel, dim3(blocks256), dim3(256), 0, stream, centered_dev, (const double *)vas, (long long)K,
                           nn, half, max_start, (const TrendFit *)fits, local_df, pdf, total_df, floor_ratio, sums);
    }
    hipLaunchKernelGGL(wls_final_kernel, dim3(blocks256), dim3(256), 0, stream, sums, nn, (double)K, lower_bound_z,
                       min_effect, use_min_effect, mean_dev, raw_var_dev, prior_var_dev, mod_var_dev, se_dev, scores_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    // (the flag lands in the solver's pinned memory: several host threads run this at once, each with a solver of its
    // own, and a device-to-host copy into a thread's stack page has the runtime pin that page on the fly)
    *flag_host_pinned = 0;
    ROCCO_HIP_TRY(hipMemcpyAsync(flag_host_pinned, bad, sizeof(int), hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));  // also: the scratch buffer is the solver's
    if (*flag_host_pinned != 0) {
        set_last_error("rocco_hip_score_centered_wls_f64: non-finite values in the centred matrix");
        return ROCCO_HIP_EINVAL;
    }
    return ROCCO_HIP_OK;
}

namespace {

template <int G>
int launch_rolling_rows(const WlsRollingTask *tasks_dev, size_t n_tasks, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wls_rolling_rows_kernel<G>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RollingRows<G>)));
        attr_set = true;
    }
    hipLaunchKernelGGL(wls_rolling_rows_kernel<G>, dim3((unsigned)n_tasks), dim3(kLanes + kRowsHelpers), sizeof(RollingRows<G>), stream,
                       tasks_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace

// rows per workgroup for a launch over `total_rows` rows: the smallest group whose workgroups are all resident at once
// (one workgroup per CU for groups of 1 or 2 rows, two for 4 or 8)
namespace {
thread_local int tl_rolling_group_min = 1;  // (the calling solver's "rolling_group_min", set for the duration of its entry points)
}

void wls_set_rolling_group_min(int rows_per_workgroup) { tl_rolling_group_min = rows_per_workgroup; }

int wls_rolling_group_rows(size_t total_rows)
{
    if (const char *e = std::getenv("ROCCO_HIP_ROLLING_GROUP")) {  // (tests: every shape of the kernel on small inputs)
        const int g = std::atoi(e);
        if (g == 1 || g == 2 || g == 4 || g == 8) {
            return g;
        }
    }
    // (half the device: a second pipeline of the count-path batch may be running its rolling launch at the same time)
    const int by_rows = (total_rows <= 128) ? 1 : (total_rows <= 256) ? 2 : (total_rows <= 1024) ? 4 : kWlsRollingGroup;
    return (by_rows > tl_rolling_group_min) ? by_rows : tl_rolling_group_min;
}

int launch_wls_rolling_batch(const WlsRollingTask *tasks_dev, size_t n_tasks, int group_rows, hipStream_t stream)
{
    if (n_tasks == 0) {
        return ROCCO_HIP_OK;
    }
    switch (group_rows) {
    case 1: return launch_rolling_rows<1>(tasks_dev, n_tasks, stream);
    case 2: return launch_rolling_rows<2>(tasks_dev, n_tasks, stream);
    case 4: return launch_rolling_rows<4>(tasks_dev, n_tasks, stream);
    case 8: return launch_rolling_rows<8>(tasks_dev, n_tasks, stream);
    default: set_last_error("launch_wls_rolling_batch: groups of 1, 2, 4 or 8 rows"); return ROCCO_HIP_EINVAL;
    }
}

size_t log_scale_scratch_bytes(size_t K, size_t n)
{
    (void)n;
    return align_up(K * 8, 256) + align_up(K * sizeof(RowSelect), 256) + align_up(K * kSelectBuckets * sizeof(unsigned), 256) + 512 +
           2 * align_up(K * sizeof(unsigned), 256) + align_up(K * (size_t)kCellMax * 8, 256) +  // gathered cells of the medians
           align_up(K * 16, 256) + align_up(K * sizeof(unsigned), 256);                        // the cells' spans, the complete rows
}

int launch_log_scale_center_rows(const double *counts_dev, size_t K, size_t n, double pseudocount, int apply_log,
                                 double *centered_out_dev, double *row_offsets_out_dev, void *scratch_dev,
                                 hipStream_t stream, int *flag_host_pinned, int center)
{
    const long long count = (long long)(K * n), nn = (long long)n, rows = (long long)K;
    char *sc = (char *)scratch_dev;
    double *med = (double *)sc;
    RowSelect *state = (RowSelect *)(sc + align_up(K * 8, 256));
    unsigned *hist = (unsigned *)((char *)state + align_up(K * sizeof(RowSelect), 256));
    int *bad = (int *)((char *)hist + align_up(K * kSelectBuckets * sizeof(unsigned), 256));
    unsigned *bucket = (unsigned *)((char *)bad + 512);
    unsigned *filled = (unsigned *)((char *)bucket + align_up(K * sizeof(unsigned), 256));
    unsigned long long *cand = (unsigned long long *)((char *)filled + align_up(K * sizeof(unsigned), 256));
    unsigned long long *span = (unsigned long long *)((char *)cand + align_up(K * (size_t)kCellMax * 8, 256));
    unsigned *complete = (unsigned *)((char *)span + align_up(K * 16, 256));
    ROCCO_HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), stream));
    ROCCO_HIP_TRY(hipMemsetAsync(complete, 0, K * sizeof(unsigned), stream));
    // the median of every row: radix select over the whole matrix, six passes + one (see row_select_count_kernel) -- the
    // first of them rides on the log scale (log_scale_count_kernel)
    hipLaunchKernelGGL(row_select_init_kernel, dim3((unsigned)((rows * kSelectBuckets + 255) / 256)), dim3(256), 0, stream, state, hist,
                       rows, nn, filled);
    const dim3 grid((unsigned)((nn + kSelectChunk - 1) / kSelectChunk), (unsigned)K);
    const int lows[6] = {53, 42, 31, 20, 9, 0}, widths[6] = {11, 11, 11, 11, 11, 9};
    static_assert(kSelectBuckets >= (1 << 11), "the first pass counts 11 bits");
    hipLaunchKernelGGL(log_scale_count_kernel, grid, dim3(256), 0, stream, counts_dev, centered_out_dev, nn, pseudocount, apply_log, bad, hist);
    (void)count;
    hipLaunchKernelGGL(row_select_span_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, state, span, complete, rows, 1);  // (span := empty)
    for (int p = 0; p < 6; ++p) {
        if (p == 0) {
            // (counted by log_scale_count_kernel)
        } else if (p >= 2) {
            hipLaunchKernelGGL(row_select_count_kernel<true>, grid, dim3(256), 0, stream, (const double *)centered_out_dev, nn, lows[p], widths[p],
                               (const RowSelect *)state, hist, span, (const unsigned *)complete);
        } else {
            hipLaunchKernelGGL(row_select_count_kernel<false>, grid, dim3(256), 0, stream, (const double *)centered_out_dev, nn, lows[p], widths[p],
                               (const RowSelect *)state, hist, (unsigned long long *)nullptr, (const unsigned *)nullptr);
        }
        hipLaunchKernelGGL(row_select_pick_kernel, dim3((unsigned)K), dim3(256), 0, stream, state, hist, (p == 0) ? 64 : widths[p],
                           (p == 1) ? bucket : (unsigned *)nullptr, (const unsigned *)complete);
        if (p >= 2 && p < 5) {
            hipLaunchKernelGGL(row_select_span_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, state, span, complete, rows, 0);
        }
        if (p == 1) {
            // 22 key bits known: a row whose median's cell holds at most kCellMax values (continuous signal; not a row of small
            // integer counts, whose median is a run of equal values) is settled from the gathered cell and skips the rest
            hipLaunchKernelGGL(row_gather_kernel, grid, dim3(256), 0, stream, (const double *)centered_out_dev, nn, state,
                               (const unsigned *)bucket, cand, filled);
            hipLaunchKernelGGL(seg_settle_kernel, dim3((unsigned)K), dim3(256), 0, stream, state, (const TrendRow *)nullptr, 1,
                               (const unsigned *)bucket, (const unsigned long long *)cand, (const unsigned *)filled);
        }
    }
    if ((nn & 1LL) == 0) {
        hipLaunchKernelGGL(row_select_above_kernel, grid, dim3(256), 0, stream, (const double *)centered_out_dev, nn, state);
    }
    hipLaunchKernelGGL(row_select_median_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, (const RowSelect *)state, rows,
                       nn, med);
    if (center != 0) {  // (0: the caller's next pass subtracts the offsets on its way -- the batched baseline sweeps do)
        hipLaunchKernelGGL(subtract_row_offset_kernel, dim3((unsigned)((nn + 1023) / 1024), (unsigned)K), dim3(256), 0, stream, centered_out_dev, med,
                           nn);
    }
    if (row_offsets_out_dev != nullptr) {
        ROCCO_HIP_TRY(hipMemcpyAsync(row_offsets_out_dev, med, K * 8, hipMemcpyDeviceToDevice, stream));
    }
    ROCCO_HIP_TRY(hipGetLastError());
    *flag_host_pinned = 0;
    ROCCO_HIP_TRY(hipMemcpyAsync(flag_host_pinned, bad, sizeof(int), hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));  // also: the scratch buffer is the solver's
    if (*flag_host_pinned != 0) {
        set_last_error("`chrom_matrix` contains non-finite values");
        return ROCCO_HIP_EINVAL;
    }
    return ROCCO_HIP_OK;
}

int launch_log2_selfcheck(int family, unsigned long long seed, unsigned long long first, size_t count, unsigned long long *out_dev,
                          hipStream_t stream)
{
    if (count > 0) {
        hipLaunchKernelGGL(log2_selfcheck_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, family, seed, first,
                           (long long)count, out_dev);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

int launch_log_scale(const double *in_dev, double *out_dev, size_t count, double pseudocount, int *bad_dev, hipStream_t stream)
{
    if (count == 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(log_scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, in_dev, out_dev,
                       (long long)count, pseudocount, 1, bad_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_subtract(const double *a_dev, const double *b_dev, double *out_dev, size_t count, hipStream_t stream, int *bad_dev)
{
    if (count == 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(subtract_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, a_dev, b_dev, out_dev,
                       (long long)count, bad_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
