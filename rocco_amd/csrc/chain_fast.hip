// rocco_amd/csrc/chain_fast.hip -- parallel delta-form evaluation of the chain solve on gfx950.
//
// What it computes (sequential definition: oracle/delta_oracle.c; derivation: DESIGN.md section 4):
// the reference's two-state Viterbi pass (rocco/_chain_dp.c:115-186) depends on its two running
// values only through their difference, which obeys
//     delta_j = clamp(delta_{j-1}, -c, +c) + a_j ,   classes ONE / ZERO / COPY,  backward fill.
// x -> clamp(x + a, lo, hi) is closed under composition, and with every input on a fixed-point
// grid each add/min/max is exact, so the composition is exactly associative: the scan below returns
// the same bits as the sequential recursion.  In "clean" chunks the inputs are rounded the way the
// reference's own arithmetic rounds them (grid u of the running value's binade, two-stage:
// rn_u(s) + rn_u(-lambda)), which reproduces the reference's decisions exactly; elsewhere a
// tolerance weight per step bounds the difference.
//
// One round = a handful of launches, all streaming / scan work (no MFMA; BLAS-1 class, HBM-bound):
//   K1 aggregate : each lane owns a 32-locus chunk staged through LDS (coalesced 16-B loads),
//                  builds its chunk function for every penalty of the task, workgroup-reduces it
//   K2 blockscan : per chain, the (short) sequence of workgroup functions -> incoming delta per
//                  workgroup; per slot, incoming clear-clamp index / tolerance weight / gain
//   K3 apply     : re-stage the tile, in-workgroup scans -> exact incoming state per lane, run the
//                  recursion, classify, certify, per-chunk fill summaries, window: write fill(LO)
//   K4 fillscan  : per slot, backward over workgroups: fill value entering from the right, counts
//   K5 patch     : window slots only: trailing undetermined loci of a workgroup take that value
//   K6 mapcodes  : map slots only: binade code per chunk from the running stay-off value
// The scores are read twice per round (K1, K3): 16 B / locus for all penalties of the round.
#include "chain_fast.h"

#include <cmath>

namespace rocco {

namespace {

constexpr int kLdsStride = kChunk + 2;  // doubles per chunk row in LDS (16-B aligned, conflict-free)
constexpr int kTileDoubles = kFastThreads * kLdsStride;
enum { kClsZero = 0, kClsCopy = 1, kClsOne = 2 };
constexpr int kFvNone = 2;

struct Fn {
    double a, lo, hi;  // x -> min(max(x + a, lo), hi)
};

__device__ __forceinline__ double grid_round(double x, double magic) { return (x + magic) - magic; }

__device__ __forceinline__ double clampc(double x, double c) { return fmin(fmax(x, -c), c); }

// apply f, then g
__device__ __forceinline__ Fn compose(const Fn &f, const Fn &g, double big)
{
    Fn r;
    r.a = fmin(fmax(f.a + g.a, -big), big);
    r.lo = fmin(fmax(f.lo + g.a, g.lo), g.hi);
    r.hi = fmin(fmax(f.hi + g.a, g.lo), g.hi);
    return r;
}

__device__ __forceinline__ double apply_fn(const Fn &f, double x) { return fmin(fmax(x + f.a, f.lo), f.hi); }

__device__ __forceinline__ Fn shfl_down_fn(const Fn &f, int off)
{
    Fn r;
    r.a = __shfl_down(f.a, off);
    r.lo = __shfl_down(f.lo, off);
    r.hi = __shfl_down(f.hi, off);
    return r;
}

__device__ __forceinline__ Fn shfl_up_fn(const Fn &f, int off)
{
    Fn r;
    r.a = __shfl_up(f.a, off);
    r.lo = __shfl_up(f.lo, off);
    r.hi = __shfl_up(f.hi, off);
    return r;
}

// ---- per-chunk arithmetic mode (mirrors make_mode / step_inputs of oracle/delta_oracle.c) --------
struct Mode {
    bool clean;
    bool mapped;  // false: no map code -> weights are counted in steps and scaled once e_global is known
    double magic_u, half_u, w_tie, w_step, base, nlam;
};

__device__ __forceinline__ Mode make_mode(int code, bool force_hazard, int e_global, const FastTask &task,
                                          double lambda)
{
    Mode m;
    const bool none = (code == kMapNone);
    int e = none ? e_global : ((code & 0x7F) - kMapBias);
    bool hazard = force_hazard || none || ((code & 0x80) != 0);
    if (!none && (code & 0x80) != 0) {
        // a hazard chunk's intermediates (value + s, before lambda comes off) leave the stay-off value by up to
        // 2 cmax + 2 sabs + |lambda| whatever that value is: oracle_hazard_floor
        e = max(e, ilogb(2.0 * task.cmax + 2.0 * task.sabs + fabs(lambda) + 2.0));
    }
    if (!hazard && e - 52 < task.qexp) {
        hazard = true;
    }
    m.magic_u = ldexp(1.5, e);
    m.half_u = ldexp(1.0, e - 53);
    m.nlam = grid_round(-lambda, m.magic_u);
    if (!hazard) {
        if (fabs(-lambda - m.nlam) == m.half_u) {
            hazard = true;
        }
        if (task.switch_costs == nullptr && fabs(task.gamma - grid_round(task.gamma, m.magic_u)) == m.half_u) {
            hazard = true;
        }
    }
    const double hb = ldexp(1.0, e + 2 - 53);
    m.clean = !hazard;
    m.mapped = !none;
    m.w_step = 4.0 * hb + task.qstep;
    m.w_tie = 2.0 * m.half_u;
    m.base = hazard ? (9.0 * hb + 2.0 * task.qstep) : 0.0;
    return m;
}

__device__ __forceinline__ double cost_on_grid(const Mode &m, double c_raw, double magic_q)
{
    return m.clean ? grid_round(c_raw, m.magic_u) : grid_round(c_raw, magic_q);
}

// a_j of one chain on the chunk's grid
__device__ __forceinline__ double step_a(const Mode &m, double s, double lambda, double magic_q)
{
    if (m.clean) {
        return grid_round(s, m.magic_u) + m.nlam;
    }
    return grid_round(s - lambda, magic_q);
}

// tolerance weight of step j (unit 1.0 for unmapped chunks; scaled later)
template <bool HAS_COSTS>
__device__ __forceinline__ double step_w(const Mode &m, double s, double c_raw_prev, bool first_locus)
{
    if (m.clean) {
        double w = (fabs(s - grid_round(s, m.magic_u)) == m.half_u) ? m.w_tie : 0.0;
        if (HAS_COSTS && !first_locus && fabs(c_raw_prev - grid_round(c_raw_prev, m.magic_u)) == m.half_u) {
            w += m.w_tie;
        }
        return w;
    }
    return m.mapped ? m.w_step : 1.0;
}

// ---- tile staging -----------------------------------------------------------------------------
// The workgroup's 8192 loci are loaded with coalesced 16-B accesses and laid out one 32-locus
// chunk per LDS row (row stride 34 doubles), so that every lane then reads its own row with
// 16-B LDS reads without bank conflicts.
// Tile staging in two phases, so that other loads can be put in flight between them: `issue` sends the sixteen
// 16-byte loads of a whole in-range tile (the common case) and `commit` writes them to LDS; every other tile is staged
// by `commit` alone.
struct TileLoads {
    double2 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15;
    bool fast;
};

template <bool HAS_COSTS>
__device__ __forceinline__ void stage_tile_issue(const FastTask &task, int local_block, TileLoads &t)
{
    const long long base = (long long)local_block * kFastBlockLoci;
    const double *__restrict__ s = task.scores;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(s) & 15U) == 0);
    t.fast = !HAS_COSTS && aligned16 && base + kFastBlockLoci <= task.n;
    if (t.fast) {
        const double2 *__restrict__ src = reinterpret_cast<const double2 *>(s + base) + threadIdx.x;
        t.v0 = src[0 * kFastThreads]; t.v1 = src[1 * kFastThreads]; t.v2 = src[2 * kFastThreads]; t.v3 = src[3 * kFastThreads];
        t.v4 = src[4 * kFastThreads]; t.v5 = src[5 * kFastThreads]; t.v6 = src[6 * kFastThreads]; t.v7 = src[7 * kFastThreads];
        t.v8 = src[8 * kFastThreads]; t.v9 = src[9 * kFastThreads]; t.v10 = src[10 * kFastThreads]; t.v11 = src[11 * kFastThreads];
        t.v12 = src[12 * kFastThreads]; t.v13 = src[13 * kFastThreads]; t.v14 = src[14 * kFastThreads]; t.v15 = src[15 * kFastThreads];
    }
}

template <bool HAS_COSTS>
__device__ __forceinline__ void stage_tile(const FastTask &task, int local_block, double *lds_s, double *lds_c);

template <bool HAS_COSTS>
__device__ __forceinline__ void stage_tile_commit(const FastTask &task, int local_block, double *lds_s, double *lds_c,
                                                  const TileLoads &t)
{
    if (!t.fast) {
        stage_tile<HAS_COSTS>(task, local_block, lds_s, lds_c);
        return;
    }
    static_assert(kChunk / 2 == 16, "tile staging is written out for 16 loads per lane");
    // bound rounds (exact arithmetic on the grid q at penalties that are multiples of q): round the
    // scores once here instead of once per step and penalty -- rn_q(s - x) == rn_q(s) - x
    const bool pre = task.pre_round != 0;
    const double mq = task.magic;
    auto put = [&](int r, const double2 &v) {
        const int e = 2 * (r * kFastThreads + (int)threadIdx.x);
        double2 w = v;
        if (pre) {
            w.x = (v.x + mq) - mq;
            w.y = (v.y + mq) - mq;
        }
        *reinterpret_cast<double2 *>(lds_s + (e >> 5) * kLdsStride + (e & 31)) = w;
    };
    put(0, t.v0); put(1, t.v1); put(2, t.v2); put(3, t.v3); put(4, t.v4); put(5, t.v5); put(6, t.v6); put(7, t.v7);
    put(8, t.v8); put(9, t.v9); put(10, t.v10); put(11, t.v11); put(12, t.v12); put(13, t.v13); put(14, t.v14);
    put(15, t.v15);
    __syncthreads();
}

template <bool HAS_COSTS>
__device__ __forceinline__ void stage_tile(const FastTask &task, int local_block, double *lds_s,
                                           double *lds_c)
{
    const long long base = (long long)local_block * kFastBlockLoci;
    const double *__restrict__ s = task.scores;
    const double *__restrict__ cs = task.switch_costs;
    const long long n = task.n;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(s) & 15U) == 0);
    if (!HAS_COSTS && aligned16 && base + kFastBlockLoci <= n) {
        // whole tile in range (workgroup-uniform): unconditional 16-byte loads, all in flight together.
        // (Loads inside per-element conditionals are waited for one by one.)
        const double2 *__restrict__ src = reinterpret_cast<const double2 *>(s + base) + threadIdx.x;
        const double2 v0 = src[0 * kFastThreads], v1 = src[1 * kFastThreads], v2 = src[2 * kFastThreads],
                      v3 = src[3 * kFastThreads], v4 = src[4 * kFastThreads], v5 = src[5 * kFastThreads],
                      v6 = src[6 * kFastThreads], v7 = src[7 * kFastThreads], v8 = src[8 * kFastThreads],
                      v9 = src[9 * kFastThreads], v10 = src[10 * kFastThreads], v11 = src[11 * kFastThreads],
                      v12 = src[12 * kFastThreads], v13 = src[13 * kFastThreads], v14 = src[14 * kFastThreads],
                      v15 = src[15 * kFastThreads];
        static_assert(kChunk / 2 == 16, "tile staging is written out for 16 loads per lane");
        // bound rounds (exact arithmetic on the grid q at penalties that are multiples of q): round the
        // scores once here instead of once per step and penalty -- rn_q(s - x) == rn_q(s) - x
        const bool pre = task.pre_round != 0;
        const double mq = task.magic;
        auto put = [&](int r, const double2 &v) {
            const int e = 2 * (r * kFastThreads + (int)threadIdx.x);
            double2 w = v;
            if (pre) {
                w.x = (v.x + mq) - mq;
                w.y = (v.y + mq) - mq;
            }
            *reinterpret_cast<double2 *>(lds_s + (e >> 5) * kLdsStride + (e & 31)) = w;
        };
        put(0, v0); put(1, v1); put(2, v2); put(3, v3); put(4, v4); put(5, v5); put(6, v6); put(7, v7);
        put(8, v8); put(9, v9); put(10, v10); put(11, v11); put(12, v12); put(13, v13); put(14, v14); put(15, v15);
        __syncthreads();
        return;
    }
#pragma unroll 4
    for (int r = 0; r < kChunk / 2; ++r) {
        const int e = 2 * (r * kFastThreads + (int)threadIdx.x);  // even element index in the tile
        const long long j = base + e;
        double2 v = make_double2(0.0, 0.0);
        if (j + 1 < n && aligned16) {
            v = *reinterpret_cast<const double2 *>(s + j);
        } else {
            if (j < n) v.x = s[j];
            if (j + 1 < n) v.y = s[j + 1];
        }
        if (task.pre_round != 0) {
            v.x = (v.x + task.magic) - task.magic;
            v.y = (v.y + task.magic) - task.magic;
        }
        *reinterpret_cast<double2 *>(lds_s + (e >> 5) * kLdsStride + (e & 31)) = v;
        if (HAS_COSTS) {
            double2 c = make_double2(0.0, 0.0);
            if (j < n - 1) c.x = cs[j];
            if (j + 1 < n - 1) c.y = cs[j + 1];
            *reinterpret_cast<double2 *>(lds_c + (e >> 5) * kLdsStride + (e & 31)) = c;
        }
    }
    __syncthreads();
}

// Per-lane view of its chunk (raw values, read from the workgroup's LDS tile): sv[i] scores,
// cost(i) = cost between loci j0+i and j0+i+1, c_prev0 = cost between j0-1 and j0.
template <bool HAS_COSTS>
struct ChunkData {
    const double *sv;
    const double *cv;
    double c_prev0;
};

template <bool HAS_COSTS>
__device__ __forceinline__ void load_chunk(const FastTask &task, long long j0, const double *lds_s,
                                           const double *lds_c, ChunkData<HAS_COSTS> &d)
{
    const int t = threadIdx.x;
    d.sv = lds_s + t * kLdsStride;
    d.cv = lds_c + t * kLdsStride;
    if (HAS_COSTS) {
        double cp = 0.0;
        if (j0 > 0 && j0 < task.n) {
            cp = (t > 0) ? lds_c[(t - 1) * kLdsStride + (kChunk - 1)] : task.switch_costs[j0 - 1];
        }
        d.c_prev0 = cp;
    } else {
        d.c_prev0 = task.gamma;
    }
}

template <bool HAS_COSTS>
__device__ __forceinline__ double raw_cost_at(const FastTask &task, const ChunkData<HAS_COSTS> &d, int i)
{
    if (HAS_COSTS) {
        return d.cv[i];
    }
    return task.gamma;
}

__device__ __forceinline__ int chunk_code(const FastTask &task, const FastSlot &slot, long long chunk, bool valid)
{
    // the load is unconditional (clamped index; the null test is uniform) so that it can be in flight together
    // with the caller's other loads
    const int raw = (task.emap != nullptr) ? (int)task.emap[valid ? chunk : 0] : kMapNone;
    if (!valid || slot.mode == kModeMap || slot.mode == kModeBound || task.emap == nullptr) {
        return kMapNone;
    }
    return raw;
}

// ---- lean paths ---------------------------------------------------------------------------------
// Nearly every wavefront of a mapped round holds only interior chunks (not the first, not the one
// with the last locus) that are clean and free of tolerance weight, and map rounds certify nothing at
// all.  For those the per-step work collapses to the recursion itself; the general loops below
// (same results, every corner case) run for the remaining wavefronts.

// K1 of a bound slot (exact arithmetic, nothing to certify): the tile already holds rn_q(score) and the
// penalty is a multiple of q, so a step input is one subtraction.
template <bool HAS_COSTS>
__device__ __forceinline__ void bound_aggregate_steps(const double *__restrict__ sv, const double *__restrict__ cv,
                                                      double c_prev0_raw, double gamma, double mq, double lambda,
                                                      double big, Fn &f, int &pstar)
{
    const double c_const = (gamma + mq) - mq;
    double c_prev = HAS_COSTS ? ((c_prev0_raw + mq) - mq) : c_const;
    bool known = false;
    int ps = kChunk;
    double fa = 0.0, lo = -big, hi = big;
#pragma unroll 1
    for (int i0 = 0; i0 < kChunk; i0 += 8) {
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            const double a = sv[i] - lambda;
            fa = fmin(fmax(fa + a, -big), big);
            lo = fmin(fmax(lo, -c_prev), c_prev) + a;
            hi = fmin(fmax(hi, -c_prev), c_prev) + a;
            if (!known && lo == hi) {
                known = true;
                ps = i;
            }
            if (HAS_COSTS) {
                c_prev = (cv[i] + mq) - mq;
            }
        }
    }
    f.a = fa;
    f.lo = lo;
    f.hi = hi;
    pstar = ps;
}

// K1, one chain, any interior chunk.  Per-lane parameters select the arithmetic: a = rn_mg(s - sub) + add
// is rn_u(s) + rn_u(-lambda) for a clean chunk (sub = 0) and rn_q(s - lambda) for a hazard chunk
// (add = 0); half_u < 0 switches the rounding-tie test off (hazard grids have none); w_const is the
// tolerance weight of every step (0 for clean chunks).  `tie` reports a rounding tie, in which case the
// caller repeats the wavefront on the general path.
template <bool HAS_COSTS, bool NOISE>
__device__ __forceinline__ void mid_aggregate_steps(const double *__restrict__ sv, const double *__restrict__ cv,
                                                    double c_prev0_raw, double gamma, double mg, double sub,
                                                    double add, double half_u, double w_const, double big, Fn &f,
                                                    int &pstar, int &lc, double &wsum_out, bool &tie,
                                                    long long &p16, long long &npos)
{
    const double c_const = (gamma + mg) - mg;
    double c_prev = HAS_COSTS ? ((c_prev0_raw + mg) - mg) : c_const;
    bool tied = HAS_COSTS && (fabs(c_prev0_raw - c_prev) == half_u);
    bool known = false;
    int ps = kChunk, last_clear = -1;
    double wsum = 0.0;
    // x -> clamp(x, -big, big) is the identity on every reachable value: the first step needs no case
    double fa = 0.0, lo = -big, hi = big;
#pragma unroll 1
    for (int i0 = 0; i0 < kChunk; i0 += 8) {
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            const double x = sv[i] - sub;
            const double rs = (x + mg) - mg;
            tied = tied || (fabs(x - rs) == half_u);
            const double a = rs + add;
            double cj = c_const;
            if (HAS_COSTS) {
                const double craw = cv[i];
                cj = (craw + mg) - mg;
                tied = tied || (fabs(craw - cj) == half_u);
            }
            fa = fmin(fmax(fa + a, -big), big);
            lo = fmin(fmax(lo, -c_prev), c_prev) + a;
            hi = fmin(fmax(hi, -c_prev), c_prev) + a;
            if (!known && lo == hi) {
                known = true;
                ps = i;
            }
            if (NOISE && a > 0.0) {
                p16 += (long long)(16.0 * a);
                ++npos;
            }
            wsum += w_const;
            if (known && (fabs(hi) - cj > kGuard)) {
                last_clear = i;
                wsum = 0.0;
            }
            c_prev = cj;
        }
    }
    f.a = fa;
    f.lo = lo;
    f.hi = hi;
    pstar = ps;
    lc = last_clear;
    wsum_out = wsum;
    tie = tied;
}

// K3, one chain, any interior chunk whose weights are the same at every step (hazard chunks, and
// clean chunks without rounding ties): the general loop without its case distinctions.
template <bool HAS_COSTS, bool GAIN>
__device__ __forceinline__ void mid_apply_steps(const double *__restrict__ sv, const double *__restrict__ cv,
                                                double c_prev0_raw, double gamma, double mg, double sub, double add,
                                                double w_const, double base, int pstar, int mrun, double wacc,
                                                double &delta, double &gain, unsigned &D, unsigned &V,
                                                long long &uncertain_out, long long &effect_out,
                                                long long &max_run_out, int &overflow_out)
{
    const double c_const = (gamma + mg) - mg;
    double c_prev = HAS_COSTS ? ((c_prev0_raw + mg) - mg) : c_const;
    double dl = delta, g = 0.0;
    unsigned dm = 0, vm = 0;
    int uncertain = 0, max_run = 0;
    long long effect = 0;
    bool overflow = false;
#pragma unroll 1
    for (int i0 = 0; i0 < kChunk; i0 += 8) {
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            const double x = sv[i] - sub;
            const double a = ((x + mg) - mg) + add;
            const double cj = HAS_COSTS ? ((cv[i] + mg) - mg) : c_const;
            if (GAIN) {
                g += fmax(0.0, dl - c_prev);
            }
            dl = fmin(fmax(dl, -c_prev), c_prev) + a;
            wacc += w_const;
            const double tau = wacc + base;
            const bool toler = tau > 0.0;
            max_run = (toler && mrun > max_run) ? mrun : max_run;
            const bool over = tau > kGuard;
            overflow = overflow || over;
            const double e = fabs(dl) - cj;
            const bool unc = over || (toler && !(fabs(e) > tau));
            uncertain += unc ? 1 : 0;
            effect += unc ? (long long)(mrun + 1) : 0LL;
            const bool one = dl > cj;
            const bool zero = dl <= -cj;
            dm |= ((one || zero) ? 1U : 0U) << i;
            vm |= (one ? 1U : 0U) << i;
            const bool clear = (i >= pstar) && (e > kGuard);
            mrun = clear ? 0 : (mrun + 1);
            wacc = clear ? 0.0 : wacc;
            c_prev = cj;
        }
    }
    delta = dl;
    gain = g;
    D = dm;
    V = vm;
    uncertain_out = uncertain;
    effect_out = effect;
    max_run_out = max_run;
    overflow_out = overflow ? 1 : 0;
}

// K3, one chain: the recursion from the true incoming delta, classes, optional gain.
template <bool CLEAN, bool HAS_COSTS, bool GAIN, bool CLASSES, bool PRE = false>
__device__ __forceinline__ void lean_apply_steps(const double *__restrict__ sv, const double *__restrict__ cv,
                                                 double c_prev0_raw, double gamma, double mg, double nl, double &delta,
                                                 double &gain, unsigned &D, unsigned &V)
{
    const double c_const = (gamma + mg) - mg;
    double c_prev = HAS_COSTS ? ((c_prev0_raw + mg) - mg) : c_const;
    double dl = delta, g = 0.0;
    unsigned dm = 0, vm = 0;
#pragma unroll 1
    for (int i0 = 0; i0 < kChunk; i0 += 8) {
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
        const int i = i0 + ii;
        const double sj = sv[i];
        // PRE: the tile holds rn_q(score) and nl (the penalty) is a multiple of q
        const double a = PRE ? (sj - nl) : (CLEAN ? (((sj + mg) - mg) + nl) : (((sj - nl) + mg) - mg));
        const double cj = HAS_COSTS ? ((cv[i] + mg) - mg) : c_const;
        if (GAIN) {
            g += fmax(0.0, dl - c_prev);
        }
        dl = fmin(fmax(dl, -c_prev), c_prev) + a;
        if (CLASSES) {
            const bool one = dl > cj;
            const bool zero = dl <= -cj;
            dm |= ((one || zero) ? 1U : 0U) << i;
            vm |= (one ? 1U : 0U) << i;
        }
        c_prev = cj;
    }
    }
    delta = dl;
    gain = g;
    D = dm;
    V = vm;
}

// ---- K1: chunk functions, provable clear clamps, tolerance weights, noise sums --------------------
template <int NCH, bool HAS_COSTS>
__device__ __forceinline__ void aggregate_slot(const FastTask &task, const FastSlot &slot, int slot_index,
                                               const FastChain *chains, const FastBuffers &buf,
                                               const ChunkData<HAS_COSTS> &d, long long chunk, long long j0,
                                               int local_block, double *lds_red)
{
    const long long n = task.n;
    const double magic = task.magic;
    const double big = task.big;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool valid = j0 < n;

    double lam[NCH];
    lam[0] = chains[slot.chain_a].lambda;
    if (NCH == 2) {
        lam[NCH - 1] = chains[slot.chain_b].lambda;
    }
    const int code = chunk_code(task, slot, chunk, valid);
    // map and bound rounds certify nothing
    const bool need_noise = (code == kMapNone) && slot.mode != kModeMap && slot.mode != kModeBound;
    Mode mode[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        mode[k] = make_mode(code, false, 0, task, lam[k]);
    }
    if (NCH == 2 && mode[0].clean != mode[NCH - 1].clean) {
        mode[0] = make_mode(code, true, 0, task, lam[0]);
        mode[NCH - 1] = make_mode(code, true, 0, task, lam[NCH - 1]);
    }
    Fn f[NCH];
    int pstar[NCH];
    bool known[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        f[k].a = 0.0;
        f[k].lo = -big;
        f[k].hi = big;
        pstar[k] = kChunk;
        known[k] = false;
    }
    int lc = -1;
    double wsum = 0.0;
    long long p16 = 0;
    long long npos = 0;

    // streamlined path: the whole wavefront holds interior chunks (not the first of the chromosome,
    // not the one with its last locus); a rounding tie sends the wavefront to the general loop
    bool done = false;
    bool anyw = false;
    if (NCH == 1) {
        const bool interior = valid && j0 > 0 && (j0 + kChunk < n);
        const bool noise_all = __all(need_noise);
        if (slot.mode == kModeBound && task.pre_round != 0 && __all(interior)) {
            bound_aggregate_steps<HAS_COSTS>(d.sv, d.cv, d.c_prev0, task.gamma, magic, lam[0], big, f[0], pstar[0]);
            known[0] = pstar[0] < kChunk;
            done = true;
            anyw = true;  // (bound slots carry no tolerance model at all; K3 never consults this)
        } else if (__all(interior) && (noise_all || !__any(need_noise))) {
            const bool clean = mode[0].clean;
            const double mg = clean ? mode[0].magic_u : magic;
            const double sub = clean ? 0.0 : lam[0];
            const double add = clean ? mode[0].nlam : 0.0;
            const double half_u = clean ? mode[0].half_u : -1.0;
            const double w_const = clean ? 0.0 : (mode[0].mapped ? mode[0].w_step : 1.0);
            bool tie = false;
            long long p16_l = 0, npos_l = 0;
            double wsum_l = 0.0;
            if (noise_all) {
                mid_aggregate_steps<HAS_COSTS, true>(d.sv, d.cv, d.c_prev0, task.gamma, mg, sub, add, half_u, w_const,
                                                     big, f[0], pstar[0], lc, wsum_l, tie, p16_l, npos_l);
            } else {
                mid_aggregate_steps<HAS_COSTS, false>(d.sv, d.cv, d.c_prev0, task.gamma, mg, sub, add, half_u,
                                                      w_const, big, f[0], pstar[0], lc, wsum_l, tie, p16_l, npos_l);
            }
            if (!__any(tie)) {
                done = true;
                known[0] = pstar[0] < kChunk;
                wsum = wsum_l;
                p16 = p16_l;
                npos = npos_l;
                anyw = (w_const != 0.0);
            } else {
                f[0].a = 0.0;
                f[0].lo = -big;
                f[0].hi = big;
                pstar[0] = kChunk;
                lc = -1;
            }
        }
    }

    if (valid && !done) {
#pragma unroll 1
        for (int i = 0; i < kChunk; ++i) {
            const long long j = j0 + i;
            if (j < n) {
                const double c_raw_prev = (i == 0) ? d.c_prev0 : raw_cost_at(task, d, i - 1);
                const double c_prev = cost_on_grid(mode[0], c_raw_prev, magic);
                const double cj = cost_on_grid(mode[0], raw_cost_at(task, d, i), magic);
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const double a = step_a(mode[k], d.sv[i], lam[k], magic);
                    if (j == 0) {
                        f[k].a = 0.0;
                        f[k].lo = a;
                        f[k].hi = a;
                    } else if (i == 0) {
                        f[k].a = a;
                        f[k].lo = -c_prev + a;
                        f[k].hi = c_prev + a;
                    } else {
                        f[k].a = fmin(fmax(f[k].a + a, -big), big);
                        f[k].lo = clampc(f[k].lo, c_prev) + a;
                        f[k].hi = clampc(f[k].hi, c_prev) + a;
                    }
                    if (!known[k] && f[k].lo == f[k].hi) {
                        known[k] = true;
                        pstar[k] = i;
                    }
                    if (k == 0 && need_noise && a > 0.0) {
                        p16 += (long long)(16.0 * a);
                        ++npos;
                    }
                }
                const double w_step_here = step_w<HAS_COSTS>(mode[0], d.sv[i], c_raw_prev, j == 0);
                anyw = anyw || (w_step_here != 0.0);
                wsum += w_step_here;
                if (j + 1 < n) {
                    bool clear;
                    if (NCH == 1) {
                        clear = known[0] && (fabs(f[0].hi) - cj > kGuard);
                    } else {
                        clear = known[0] && known[NCH - 1] &&
                                ((f[NCH - 1].hi - cj > kGuard) || (-f[0].hi - cj > kGuard));
                    }
                    if (clear) {
                        lc = i;
                        wsum = 0.0;
                    }
                }
            }
        }
    }

    if (valid) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const FastChain &ch = chains[k == 0 ? slot.chain_a : slot.chain_b];
            buf.agg_a[ch.chunk_off + chunk] = f[k].a;
            buf.agg_lo[ch.chunk_off + chunk] = f[k].lo;
            buf.agg_hi[ch.chunk_off + chunk] = f[k].hi;
            buf.pstar[ch.chunk_off + chunk] = (uint8_t)(pstar[k] | ((k == 0 && anyw) ? 0x80 : 0));
        }
        buf.lc_chunk[slot.chunk_off + chunk] = (int8_t)lc;
        buf.w_chunk[slot.chunk_off + chunk] = wsum;
    }

    // workgroup-ordered composition of the chunk functions (per chain)
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        Fn g = f[k];
        const double last_lo = __shfl(f[k].lo, 63), last_hi = __shfl(f[k].hi, 63);
        if (last_lo == last_hi) {
            // the wavefront's last chunk function is constant: so is the composition, with that value
            g.a = 0.0;
            g.lo = last_lo;
            g.hi = last_hi;
        } else {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const Fn p = shfl_down_fn(g, off);
                if (lane + off < 64) {
                    g = compose(g, p, big);
                }
            }
        }
        if (lane == 0) {
            lds_red[(k * 4 + wave) * 3 + 0] = g.a;
            lds_red[(k * 4 + wave) * 3 + 1] = g.lo;
            lds_red[(k * 4 + wave) * 3 + 2] = g.hi;
        }
    }
    // ordered reduction of (has_clear, weight), max clear index, noise sums
    long long lcg = (lc >= 0) ? (j0 + lc) : -1;
    int wflag = (lc >= 0) ? 1 : 0;
    double wval = wsum;
    if (!__any(wsum != 0.0 || p16 != 0 || npos != 0)) {
        // no weights and no noise sums in this wavefront: only the clear-clamp index is reduced
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long pl = __shfl_down(lcg, off);
            const int pf = __shfl_down(wflag, off);
            if (lane + off < 64) {
                lcg = (pl > lcg) ? pl : lcg;
                wflag = wflag | pf;
            }
        }
    } else {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int pf = __shfl_down(wflag, off);
            const double pv = __shfl_down(wval, off);
            const long long pl = __shfl_down(lcg, off);
            const long long pp = __shfl_down(p16, off);
            const long long pn = __shfl_down(npos, off);
            if (lane + off < 64) {
                // own is the left operand, partner the right one
                wval = pf ? pv : (wval + pv);
                wflag = wflag | pf;
                lcg = (pl > lcg) ? pl : lcg;
                p16 += pp;
                npos += pn;
            }
        }
    }
    long long *lds_ll = reinterpret_cast<long long *>(lds_red + 24);
    if (lane == 0) {
        lds_ll[wave * 4 + 0] = lcg;
        lds_ll[wave * 4 + 1] = p16;
        lds_ll[wave * 4 + 2] = npos;
        lds_ll[wave * 4 + 3] = (long long)wflag;
        lds_red[40 + wave] = wval;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            Fn g;
            g.a = lds_red[(k * 4) * 3 + 0];
            g.lo = lds_red[(k * 4) * 3 + 1];
            g.hi = lds_red[(k * 4) * 3 + 2];
            for (int w = 1; w < 4; ++w) {
                Fn p;
                p.a = lds_red[(k * 4 + w) * 3 + 0];
                p.lo = lds_red[(k * 4 + w) * 3 + 1];
                p.hi = lds_red[(k * 4 + w) * 3 + 2];
                g = compose(g, p, big);
            }
            const FastChain &ch = chains[k == 0 ? slot.chain_a : slot.chain_b];
            buf.blk_a[ch.block_off + local_block] = g.a;
            buf.blk_lo[ch.block_off + local_block] = g.lo;
            buf.blk_hi[ch.block_off + local_block] = g.hi;
        }
        long long m = -1, sp = 0, sn = 0;
        int bf = 0;
        double bw = 0.0;
        for (int w = 0; w < 4; ++w) {
            m = (lds_ll[w * 4] > m) ? lds_ll[w * 4] : m;
            sp += lds_ll[w * 4 + 1];
            sn += lds_ll[w * 4 + 2];
            const int f2 = (int)lds_ll[w * 4 + 3];
            bw = f2 ? lds_red[40 + w] : (bw + lds_red[40 + w]);
            bf |= f2;
        }
        buf.lc_block[slot.block_off + local_block] = (int)m;
        buf.w_block[slot.block_off + local_block] = bw;
        if (sp != 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&buf.results[slot_index].p16),
                      (unsigned long long)sp);
        }
        if (sn != 0) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&buf.results[slot_index].npos),
                      (unsigned long long)sn);
        }
    }
    __syncthreads();
}

// The slot and chain descriptors of a task, fetched by the first lanes while the tile is being staged (two
// dependent global loads per slot, which would otherwise sit at the head of every pass of the slot loop).
struct SlotDesc {
    FastSlot slot;
    FastChain a, b;
};
constexpr int kMaxStagedSlots = 64;

__device__ __forceinline__ void stage_slot_descs(const FastLaunch &L, const FastTask &task, SlotDesc *desc)
{
    const int t = threadIdx.x;
    if (t < task.slot_count && t < kMaxStagedSlots) {
        const FastSlot s = L.slots[task.slot_begin + t];
        desc[t].slot = s;
        desc[t].a = L.chains[s.chain_a];
        desc[t].b = L.chains[s.chain_b];
    }
}

// slot `si` of the task with its two chains as a local two-element table (slot.chain_a / chain_b index it);
// from_memory: bypass the LDS copies (before the barrier that publishes them)
__device__ __forceinline__ void fetch_slot(const FastLaunch &L, const FastTask &task, const SlotDesc *desc, int si,
                                           FastSlot &slot, FastChain (&two)[2], bool from_memory = false)
{
    if (!from_memory && si < kMaxStagedSlots) {
        slot = desc[si].slot;
        two[0] = desc[si].a;
        two[1] = desc[si].b;
    } else {
        slot = L.slots[task.slot_begin + si];
        two[0] = L.chains[slot.chain_a];
        two[1] = L.chains[slot.chain_b];
    }
    slot.chain_a = 0;
    slot.chain_b = 1;
}

template <bool HAS_COSTS>
__global__ __launch_bounds__(kFastThreads, 2) void fast_aggregate_kernel(FastLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds_s = smem;
    double *lds_c = HAS_COSTS ? (smem + kTileDoubles) : smem;
    double *lds_red = smem + (HAS_COSTS ? 2 : 1) * kTileDoubles;

    const int2 bm = L.blockmap[blockIdx.x / L.slot_groups];
    const FastTask task = L.tasks[bm.x];
    if (HAS_COSTS != (task.switch_costs != nullptr)) {
        return;
    }
    const int local_block = bm.y;
    SlotDesc *desc = reinterpret_cast<SlotDesc *>(lds_red + 64);
    TileLoads tile;
    stage_tile_issue<HAS_COSTS>(task, local_block, tile);  // the tile's loads first: they only need the task
    stage_slot_descs(L, task, desc);
    stage_tile_commit<HAS_COSTS>(task, local_block, lds_s, lds_c, tile);  // (ends with the barrier that publishes desc too)
    const long long chunk = (long long)local_block * kFastThreads + threadIdx.x;
    const long long j0 = chunk * kChunk;
    ChunkData<HAS_COSTS> d;
    load_chunk<HAS_COSTS>(task, j0, lds_s, lds_c, d);
    for (int si = (int)(blockIdx.x % L.slot_groups); si < task.slot_count; si += L.slot_groups) {
        const int slot_index = task.slot_begin + si;
        FastSlot slot;
        FastChain two[2];
        fetch_slot(L, task, desc, si, slot, two);
        if (slot.mode == kModeWindow) {
            aggregate_slot<2, HAS_COSTS>(task, slot, slot_index, two, L.buf, d, chunk, j0, local_block, lds_red);
        } else {
            aggregate_slot<1, HAS_COSTS>(task, slot, slot_index, two, L.buf, d, chunk, j0, local_block, lds_red);
        }
    }
}

// ---- K2: per chain incoming delta per workgroup; per slot incoming clear index / weight / gain ------
__global__ __launch_bounds__(64) void fast_blockscan_kernel(FastLaunch L)
{
    // one wavefront per chain / slot; the (short) per-workgroup sequences are scanned 64 at a time
    const int idx = blockIdx.x;
    const int lane = threadIdx.x;
    if (idx < L.n_chains) {
        const FastChain ch = L.chains[idx];
        const FastTask &task = L.tasks[ch.task];
        const int nb = task.n_blocks;
        const double big = task.big;
        double v = 0.0;  // delta at the last locus before the tile (unused for block 0)
        for (int base = 0; base < nb; base += 64) {
            const int b = base + lane;
            Fn f;
            f.a = 0.0;
            f.lo = -big;
            f.hi = big;
            if (b < nb) {
                if (task.frz.flag != nullptr && task.frz.flag[b]) {
                    // frozen block: constant function with the closed-form value at this penalty
                    const double nlam = grid_round(-ch.lambda, ldexp(1.5, (int)task.frz.e[b]));
                    const double v_const = task.frz.B[b] + (double)task.frz.m[b] * nlam;
                    f.a = 0.0;
                    f.lo = v_const;
                    f.hi = v_const;
                } else {
                    f.a = L.buf.blk_a[ch.block_off + b];
                    f.lo = L.buf.blk_lo[ch.block_off + b];
                    f.hi = L.buf.blk_hi[ch.block_off + b];
                }
            }
            Fn inc = f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const Fn p = shfl_up_fn(inc, off);
                if (lane >= off) {
                    inc = compose(p, inc, big);
                }
            }
            Fn ex = shfl_up_fn(inc, 1);
            if (lane == 0) {
                ex.a = 0.0;
                ex.lo = -big;
                ex.hi = big;
            }
            if (b < nb) {
                L.buf.din[ch.block_off + b] = apply_fn(ex, v);
            }
            Fn tot;
            tot.a = __shfl(inc.a, 63);
            tot.lo = __shfl(inc.lo, 63);
            tot.hi = __shfl(inc.hi, 63);
            v = apply_fn(tot, v);
        }
    } else if (idx < L.n_chains + L.n_slots) {
        const int si = idx - L.n_chains;
        const FastSlot slot = L.slots[si];
        const FastTask &task = L.tasks[slot.task];
        const int nb = task.n_blocks;
        int run = -1;
        double w = 0.0;
        for (int base = 0; base < nb; base += 64) {
            const int b = base + lane;
            const long long at = slot.block_off + b;
            int x = -1;
            double wv = 0.0;
            if (b < nb) {
                if (task.frz.flag != nullptr && task.frz.flag[b]) {
                    x = task.frz.lc[b];
                    wv = 0.0;
                } else {
                    x = L.buf.lc_block[at];
                    wv = L.buf.w_block[at];
                }
            }
            int inc = x;
            int flag = (x >= 0) ? 1 : 0;
            double val = wv;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int p = __shfl_up(inc, off);
                const int pf = __shfl_up(flag, off);
                const double pv = __shfl_up(val, off);
                if (lane >= off) {
                    inc = (p > inc) ? p : inc;
                    if (!flag) {
                        val = pv + val;
                    }
                    flag |= pf;
                }
            }
            int ex = __shfl_up(inc, 1);
            int exf = __shfl_up(flag, 1);
            double exv = __shfl_up(val, 1);
            if (lane == 0) {
                ex = -1;
                exf = 0;
                exv = 0.0;
            }
            if (b < nb) {
                L.buf.lcin_block[at] = (ex > run) ? ex : run;
                L.buf.win_block[at] = exf ? exv : (w + exv);
            }
            const int tinc = __shfl(inc, 63);
            const int tflag = __shfl(flag, 63);
            const double tval = __shfl(val, 63);
            run = (tinc > run) ? tinc : run;
            w = tflag ? tval : (w + tval);
        }
        if (lane == 0) {
            // global exponent (oracle_global_exponent): Pb bounds every running value of the reference
            FastSlotResult &res = L.buf.results[si];
            const double lam = L.chains[slot.chain_a].lambda;
            const double pb =
                2.0 * ((double)(res.p16 + res.npos) * 0.0625 + task.cmax + task.sabs + fabs(lam) + 1.0);
            res.e_global = ilogb(pb);
        }
    }
}

// ---- K3 helpers ---------------------------------------------------------------------------------
// Within-chunk backward fill on bit masks: det = determined loci, val = class ONE among them.
// Every locus below a determined one takes the nearest determined value above it; loci above the
// highest determined one stay undetermined.
__device__ __forceinline__ void smear_fill(unsigned &det, unsigned &val)
{
    val &= det;
#pragma unroll
    for (int k = 1; k < 32; k <<= 1) {
        const unsigned take = ~det & (det >> k);
        val |= take & (val >> k);
        det |= take;
    }
}

__device__ __forceinline__ unsigned long long spread_bits_to_bytes(unsigned x8)
{
    unsigned long long t = ((unsigned long long)(x8 & 0xFFU) * 0x0101010101010101ULL) & 0x8040201008040201ULL;
    t = ((t + 0x7F7F7F7F7F7F7F7FULL) >> 7) & 0x0101010101010101ULL;
    return t;
}

struct FillOut {
    unsigned zbits;  // fill values of the chunk (pending loci as 0)
};

// Workgroup-level backward fill for one class variant.  D/V: determined / ONE masks of the lane's
// chunk (already restricted to valid loci).  Writes the workgroup summary (fv, pend, base).
__device__ __forceinline__ FillOut block_fill(unsigned D, unsigned V, unsigned validmask, uint8_t *bfv,
                                              unsigned *bpend, unsigned *bbase, long long block_index,
                                              unsigned *lds_u)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned det = D, val = V;
    smear_fill(det, val);
    const unsigned tailmask = ~det & validmask;
    const int fv = (D != 0U) ? (int)((V >> (__ffs(D) - 1)) & 1U) : kFvNone;
    const unsigned long long has = __ballot(fv != kFvNone);
    const unsigned long long one = __ballot(fv == 1);
    int r = kFvNone;  // value of the nearest determined class to the right of this chunk
    const unsigned long long right = (lane == 63) ? 0ULL : (has & (~0ULL << (lane + 1)));
    if (right != 0ULL) {
        const int u = __ffsll((long long)right) - 1;
        r = (int)((one >> u) & 1ULL);
    }
    if (lane == 0) {
        lds_u[wave * 2 + 0] = (has != 0ULL) ? 1U : 0U;
        lds_u[wave * 2 + 1] = (has != 0ULL) ? (unsigned)((one >> (__ffsll((long long)has) - 1)) & 1ULL) : 0U;
    }
    __syncthreads();
    if (r == kFvNone) {
        for (int w = wave + 1; w < 4; ++w) {
            if (lds_u[w * 2]) {
                r = (int)lds_u[w * 2 + 1];
                break;
            }
        }
    }
    int block_fv = kFvNone;
    for (int w = 0; w < 4; ++w) {
        if (lds_u[w * 2]) {
            block_fv = (int)lds_u[w * 2 + 1];
            break;
        }
    }
    __syncthreads();
    FillOut out;
    const bool pending = (r == kFvNone) && (tailmask != 0U);
    out.zbits = (val & det & validmask) | ((r == 1) ? tailmask : 0U);
    unsigned base = (unsigned)__popc(out.zbits);
    unsigned pend = pending ? (unsigned)__popc(tailmask) : 0U;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        base += __shfl_down(base, off);
        pend += __shfl_down(pend, off);
    }
    if (lane == 0) {
        lds_u[8 + wave * 2 + 0] = base;
        lds_u[8 + wave * 2 + 1] = pend;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned b = 0, p = 0;
        for (int w = 0; w < 4; ++w) {
            b += lds_u[8 + w * 2];
            p += lds_u[8 + w * 2 + 1];
        }
        bfv[block_index] = (uint8_t)block_fv;
        bpend[block_index] = p;
        bbase[block_index] = b;
    }
    __syncthreads();
    return out;
}

// exclusive in-workgroup scan of the chunk functions of one chain, evaluated at the workgroup's
// incoming delta: returns the delta entering this lane's chunk
// (f: this lane's chunk function, the identity for lanes past the end; din_block: the delta entering the workgroup --
// both fetched by the caller together with everything else the slot pass needs from memory)
__device__ __forceinline__ double incoming_delta(const Fn &f, double din_block, double big, double *lds_red)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Fn inc = f;
    // anything composed with a constant function is that constant: when every chunk function of the
    // wavefront has coalesced (the usual case) the inclusive scan is the functions themselves
    if (!__all(f.lo == f.hi)) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const Fn p = shfl_up_fn(inc, off);
            if (lane >= off) {
                inc = compose(p, inc, big);
            }
        }
    }
    if (lane == 63) {
        lds_red[wave * 3 + 0] = inc.a;
        lds_red[wave * 3 + 1] = inc.lo;
        lds_red[wave * 3 + 2] = inc.hi;
    }
    __syncthreads();
    Fn ex = shfl_up_fn(inc, 1);
    if (lane == 0) {
        ex.a = 0.0;
        ex.lo = -big;
        ex.hi = big;
    }
    Fn pre;
    pre.a = 0.0;
    pre.lo = -big;
    pre.hi = big;
    for (int w = 0; w < wave; ++w) {
        Fn p;
        p.a = lds_red[w * 3 + 0];
        p.lo = lds_red[w * 3 + 1];
        p.hi = lds_red[w * 3 + 2];
        pre = compose(pre, p, big);
    }
    __syncthreads();
    const Fn total = compose(pre, ex, big);
    return apply_fn(total, din_block);
}

// exclusive in-workgroup scans of the clear-clamp index (max) and of the tolerance weight
// (segmented sum: a chunk with a clear clamp restarts the sum)
__device__ __forceinline__ void incoming_clear(long long own_lc, double own_w, bool own_any_weight,
                                               long long block_lc, double block_w, long long *lds_ll,
                                               double *lds_w, long long &lc_in, double &w_in)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (__syncthreads_and((!own_any_weight && block_w == 0.0) ? 1 : 0)) {
        // no tolerance weight in or before this workgroup's chunks: tau is zero everywhere in it, and
        // the clear-clamp index (only used where tau > 0) does not matter
        lc_in = block_lc;
        w_in = 0.0;
        return;
    }
    long long inc = own_lc;
    int flag = (own_lc >= 0) ? 1 : 0;
    double val = own_w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long p = __shfl_up(inc, off);
        const int pf = __shfl_up(flag, off);
        const double pv = __shfl_up(val, off);
        if (lane >= off) {
            inc = (p > inc) ? p : inc;
            if (!flag) {
                val = pv + val;
            }
            flag |= pf;
        }
    }
    if (lane == 63) {
        lds_ll[wave * 2] = inc;
        lds_ll[wave * 2 + 1] = (long long)flag;
        lds_w[wave] = val;
    }
    __syncthreads();
    long long ex = __shfl_up(inc, 1);
    int exf = __shfl_up(flag, 1);
    double exv = __shfl_up(val, 1);
    if (lane == 0) {
        ex = -1;
        exf = 0;
        exv = 0.0;
    }
    long long pre = block_lc;
    double prew = block_w;
    for (int w = 0; w < wave; ++w) {
        pre = (lds_ll[w * 2] > pre) ? lds_ll[w * 2] : pre;
        prew = lds_ll[w * 2 + 1] ? lds_w[w] : (prew + lds_w[w]);
    }
    __syncthreads();
    lc_in = (ex > pre) ? ex : pre;
    w_in = exf ? exv : (prew + exv);
}

// ---- K3: apply ------------------------------------------------------------------------------------
// What one K3 slot pass reads from memory (written by K1 / K2 of the same round): fetched one pass ahead, so that the
// round trip -- 3-4 us under load -- hides behind the previous pass (or behind the tile staging for the first one).
struct SlotInputs {
    Fn f[2];          // this lane's chunk function per chain
    double din[2];    // delta entering the workgroup per chain
    int praw[2];      // pstar byte per chain
    int lc_raw;
    double w_raw;
    long long lcin_b;
    double win_b;
    int e_global;
    int raw_code;     // binade code of the chunk (kMapNone without a map)
};

__device__ __forceinline__ void load_slot_inputs(const FastTask &task, const FastSlot &slot, int slot_index,
                                                 const FastChain *chains, const FastBuffers &buf, long long chunk,
                                                 bool valid, int local_block, SlotInputs &in)
{
    const long long cq = valid ? chunk : 0;  // clamped indices instead of branches: the loads go out together
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const FastChain &ch = chains[k == 0 ? slot.chain_a : slot.chain_b];
        in.f[k].a = buf.agg_a[ch.chunk_off + cq];
        in.f[k].lo = buf.agg_lo[ch.chunk_off + cq];
        in.f[k].hi = buf.agg_hi[ch.chunk_off + cq];
        in.praw[k] = (int)buf.pstar[ch.chunk_off + cq];
        in.din[k] = buf.din[ch.block_off + local_block];
    }
    in.lc_raw = (int)buf.lc_chunk[slot.chunk_off + cq];
    in.w_raw = buf.w_chunk[slot.chunk_off + cq];
    in.lcin_b = (long long)buf.lcin_block[slot.block_off + local_block];
    in.win_b = buf.win_block[slot.block_off + local_block];
    in.e_global = buf.results[slot_index].e_global;
    in.raw_code = (task.emap != nullptr) ? (int)task.emap[cq] : kMapNone;
}

template <int NCH, bool HAS_COSTS>
__device__ __forceinline__ void apply_slot(const FastTask &task, const FastSlot &slot, int slot_index,
                                           const FastChain *chains, const FastBuffers &buf,
                                           const ChunkData<HAS_COSTS> &d, long long chunk, long long j0,
                                           int local_block, double *lds_red, const SlotInputs &in)
{
    const long long n = task.n;
    const double magic = task.magic;
    const double big = task.big;
    const bool valid = j0 < n;
    const int lane = threadIdx.x & 63;
    FastSlotResult &res = buf.results[slot_index];
    long long *lds_ll = reinterpret_cast<long long *>(lds_red + 24);
    double *lds_w = lds_red + 36;
    unsigned *lds_u = reinterpret_cast<unsigned *>(lds_red + 44);

    // (everything this pass needs from memory was fetched a pass ahead: `in`)
    Fn f_in[NCH];
    double din_block[NCH], lam[NCH];
    int praw[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const FastChain &ch = chains[k == 0 ? slot.chain_a : slot.chain_b];
        lam[k] = ch.lambda;
        f_in[k] = in.f[k];
        praw[k] = in.praw[k];
        din_block[k] = in.din[k];
    }
    const int lc_raw = in.lc_raw;
    const double w_raw = in.w_raw;
    const long long lcin_b = in.lcin_b;
    const double win_b = in.win_b;
    const int e_global = in.e_global;
    const int code = (!valid || slot.mode == kModeMap || slot.mode == kModeBound || task.emap == nullptr) ? kMapNone
                                                                                                        : in.raw_code;

    double delta[NCH], delta_in[NCH];
    double delta_in0 = 0.0;
    int pstar = 0;
    bool anyw = false;  // K1: some step of this chunk carries tolerance weight
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        Fn f = f_in[k];
        if (!valid) {
            f.a = 0.0;
            f.lo = -big;
            f.hi = big;
        }
        delta[k] = incoming_delta(f, din_block[k], big, lds_red);
        delta_in[k] = delta[k];
        if (k == 0) {
            delta_in0 = delta[0];
        }
        if (valid) {
            const int p = praw[k] & 0x7F;
            pstar = (p > pstar) ? p : pstar;
            anyw = anyw || (k == 0 && (praw[k] & 0x80) != 0);
        }
    }
    Mode mode[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        mode[k] = make_mode(code, false, e_global, task, lam[k]);
    }
    if (NCH == 2 && mode[0].clean != mode[NCH - 1].clean) {
        mode[0] = make_mode(code, true, e_global, task, lam[0]);
        mode[NCH - 1] = make_mode(code, true, e_global, task, lam[NCH - 1]);
    }
    // unmapped chunks counted their weights in steps
    const double wscale = mode[0].mapped ? 1.0 : mode[0].w_step;
    const long long own_lc = (valid && lc_raw >= 0) ? (j0 + lc_raw) : -1;
    const double own_w = valid ? w_raw : 0.0;
    long long lc;
    double wacc;
    incoming_clear(own_lc, own_w, valid && anyw && slot.mode != kModeBound, lcin_b, win_b, lds_ll, lds_w, lc, wacc);
    // a task is either mapped everywhere or nowhere, so one scale applies to the incoming weight too
    wacc *= wscale;

    unsigned D_lo = 0, V_lo = 0, D_hi = 0, V_hi = 0;
    long long uncertain = 0, effect = 0, max_run = 0;
    int overflow = 0, nonadjacent = 0;
    unsigned validmask = 0;
    double gain = 0.0, gain_b = 0.0;
    double wsum_chunk = 0.0;
    const bool survey = (NCH == 2) && (task.frz_out.flag != nullptr);

    // lean path: every chunk of the wavefront is interior and either belongs to a map round (nothing
    // is certified) or is clean with no tolerance anywhere (tau == 0: every class is certain)
    bool done = false;
    if (NCH == 1) {
        const bool interior = valid && j0 > 0 && (j0 + kChunk < n);
        const bool map_round = (slot.mode == kModeMap);
        const bool bound_round = (slot.mode == kModeBound);
        const bool lane_lean =
            interior && (map_round || bound_round || (mode[0].clean && mode[0].mapped && !anyw && wacc == 0.0));
        if (__all(lane_lean)) {
            done = true;
            validmask = 0xFFFFFFFFU;
            if (bound_round && task.pre_round != 0) {
                lean_apply_steps<false, HAS_COSTS, false, true, true>(d.sv, d.cv, d.c_prev0, task.gamma, magic,
                                                                      lam[0], delta[0], gain, D_lo, V_lo);
            } else if (bound_round) {
                lean_apply_steps<false, HAS_COSTS, false, true>(d.sv, d.cv, d.c_prev0, task.gamma, magic, lam[0],
                                                                delta[0], gain, D_lo, V_lo);
            } else if (map_round) {
                lean_apply_steps<false, HAS_COSTS, true, false>(d.sv, d.cv, d.c_prev0, task.gamma, magic, lam[0],
                                                                delta[0], gain, D_lo, V_lo);
            } else if (slot.mode == kModeRecord) {
                lean_apply_steps<true, HAS_COSTS, true, true>(d.sv, d.cv, d.c_prev0, task.gamma, mode[0].magic_u,
                                                              mode[0].nlam, delta[0], gain, D_lo, V_lo);
            } else {
                lean_apply_steps<true, HAS_COSTS, false, true>(d.sv, d.cv, d.c_prev0, task.gamma, mode[0].magic_u,
                                                               mode[0].nlam, delta[0], gain, D_lo, V_lo);
            }
        }
    }

    bool stats_pending = !done;
    if (NCH == 1 && !done && n < (1LL << 31)) {
        const bool interior = valid && j0 > 0 && (j0 + kChunk < n);
        if (__all(interior && (!mode[0].clean || !anyw))) {
            done = true;
            validmask = 0xFFFFFFFFU;
            const bool clean = mode[0].clean;
            const double mg = clean ? mode[0].magic_u : magic;
            const double sub = clean ? 0.0 : lam[0];
            const double add = clean ? mode[0].nlam : 0.0;
            const double w_const = clean ? 0.0 : ((mode[0].mapped ? mode[0].w_step : 1.0) * wscale);
            const int mrun0 = (int)(j0 - 1 - lc);
            if (slot.mode == kModeMap || slot.mode == kModeRecord) {
                mid_apply_steps<HAS_COSTS, true>(d.sv, d.cv, d.c_prev0, task.gamma, mg, sub, add, w_const,
                                                 mode[0].base, pstar, mrun0, wacc, delta[0], gain, D_lo, V_lo,
                                                 uncertain, effect, max_run, overflow);
            } else {
                mid_apply_steps<HAS_COSTS, false>(d.sv, d.cv, d.c_prev0, task.gamma, mg, sub, add, w_const,
                                                  mode[0].base, pstar, mrun0, wacc, delta[0], gain, D_lo, V_lo,
                                                  uncertain, effect, max_run, overflow);
            }
            wsum_chunk = w_const;
        }
    }

    if (valid && !done) {
#pragma unroll 1
        for (int i = 0; i < kChunk; ++i) {
            const long long j = j0 + i;
            if (j < n) {
                validmask |= 1U << i;
                const double c_raw_prev = (i == 0) ? d.c_prev0 : raw_cost_at(task, d, i - 1);
                const double c_prev = cost_on_grid(mode[0], c_raw_prev, magic);
                const double cj = cost_on_grid(mode[0], raw_cost_at(task, d, i), magic);
                if ((slot.mode == kModeMap || slot.mode == kModeRecord || survey) && j > 0) {
                    gain += fmax(0.0, delta[0] - c_prev);
                    if (survey) {
                        gain_b += fmax(0.0, delta[NCH - 1] - c_prev);
                    }
                }
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const double a = step_a(mode[k], d.sv[i], lam[k], magic);
                    delta[k] = (j == 0) ? a : (clampc(delta[k], c_prev) + a);
                }
                const double w_here = step_w<HAS_COSTS>(mode[0], d.sv[i], c_raw_prev, j == 0) * wscale;
                wacc += w_here;
                wsum_chunk += w_here;
                const long long m = j - 1 - lc;
                const double tau = wacc + mode[0].base;
                if (tau > 0.0) {  // diagnostic: longest run behind a locus that carries tolerance
                    max_run = (m > max_run) ? m : max_run;
                }
                const bool over = tau > kGuard;
                overflow |= over ? 1 : 0;
                const bool last = (j + 1 >= n);
                if (NCH == 1) {
                    const double dl = delta[0];
                    int cls;
                    bool certain;
                    if (!last) {
                        const double e = fabs(dl) - cj;
                        certain = !over && (tau == 0.0 || e > tau || e < -tau);
                        cls = (dl > cj) ? kClsOne : ((dl <= -cj) ? kClsZero : kClsCopy);
                        if (i >= pstar && e > kGuard) {
                            lc = j;
                            wacc = 0.0;
                        }
                    } else {
                        certain = !over && (tau == 0.0 || fabs(dl) > tau);
                        cls = (dl > 0.0) ? kClsOne : kClsZero;
                    }
                    if (!certain) {
                        ++uncertain;
                        effect += m + 1;
                    }
                    D_lo |= (cls != kClsCopy ? 1U : 0U) << i;
                    V_lo |= (cls == kClsOne ? 1U : 0U) << i;
                } else {
                    const double dlo = delta[0];        // at lambda_lo (larger)
                    const double dhi = delta[NCH - 1];  // at lambda_hi (smaller)
                    int lo, hi;
                    if (!last) {
                        lo = (dhi + cj <= tau) ? kClsZero : ((dhi - cj > tau) ? kClsOne : kClsCopy);
                        const bool hi_one = (tau > 0.0) ? (dlo - cj >= -tau) : (dlo - cj > 0.0);
                        const bool hi_zero = (tau > 0.0) ? (dlo + cj < -tau) : (dlo + cj <= 0.0);
                        hi = hi_one ? kClsOne : (hi_zero ? kClsZero : kClsCopy);
                        if (i >= pstar && ((dhi - cj > kGuard) || (-dlo - cj > kGuard))) {
                            lc = j;
                            wacc = 0.0;
                        }
                    } else {
                        lo = (dhi > tau) ? kClsOne : kClsZero;
                        hi = ((tau > 0.0) ? (dlo >= -tau) : (dlo > 0.0)) ? kClsOne : kClsZero;
                    }
                    if (lo != hi) {
                        const double bound = !last ? ((hi == kClsOne) ? cj : -cj) : 0.0;
                        if (!last && (hi - lo != 1)) {
                            nonadjacent = 1;
                        }
                        const unsigned long long at =
                            atomicAdd(reinterpret_cast<unsigned long long *>(&res.n_diff), 1ULL);
                        if (at < (unsigned long long)kMaxDiffs) {
                            FastDiff &fd = res.diffs[at];
                            fd.locus = j;
                            fd.margin_lo = dlo - bound;
                            fd.margin_hi = dhi - bound;
                            fd.run = m;
                            fd.cls_lo = lo;
                            fd.cls_hi = hi;
                        }
                    }
                    D_lo |= (lo != kClsCopy ? 1U : 0U) << i;
                    V_lo |= (lo == kClsOne ? 1U : 0U) << i;
                    D_hi |= (hi != kClsCopy ? 1U : 0U) << i;
                    V_hi |= (hi == kClsOne ? 1U : 0U) << i;
                }
            }
        }
    }

    // certification statistics (a lean wavefront has none)
    if (stats_pending) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uncertain += __shfl_down(uncertain, off);
        effect += __shfl_down(effect, off);
        const long long o = __shfl_down(max_run, off);
        max_run = (o > max_run) ? o : max_run;
        overflow |= __shfl_down(overflow, off);
        nonadjacent |= __shfl_down(nonadjacent, off);
    }
    if (lane == 0) {
        if (uncertain) atomicAdd(reinterpret_cast<unsigned long long *>(&res.uncertain), (unsigned long long)uncertain);
        if (effect) atomicAdd(reinterpret_cast<unsigned long long *>(&res.effect), (unsigned long long)effect);
        if (max_run > 0) atomicMax(reinterpret_cast<long long *>(&res.max_run), max_run);
        if (overflow) atomicOr(&res.overflow, 1);
        if (nonadjacent) atomicOr(&res.nonadjacent, 1);
    }
    }

    // record slots: what the exact spine needs per chunk, laid out [chunk][slot of the task]
    if (slot.mode == kModeRecord) {
        const bool exact_chunk = !valid || (mode[0].clean && wsum_chunk == 0.0);
        if (valid) {
            const long long at = task.rec_off + chunk * task.slot_count + (slot_index - task.slot_begin);
            buf.rec_din[at] = delta_in0;
            buf.rec_gain[at] = gain;
            buf.rec_d[at] = D_lo;
            buf.rec_v[at] = V_lo;
            buf.rec_flags[at] = (uint8_t)(exact_chunk ? 1 : 0);
        }
        // summary of each group of 32 chunks (half a wavefront): the spine adds whole groups at once
        double gs = valid ? gain : 0.0;
        int ok = exact_chunk ? 1 : 0;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            gs += __shfl_down(gs, off, 32);
            ok &= __shfl_down(ok, off, 32);
        }
        if ((lane & 31) == 0 && valid) {
            const long long gat = task.rec_goff + (chunk / 32) * task.slot_count + (slot_index - task.slot_begin);
            buf.rec_gsum[gat] = gs;
            buf.rec_gok[gat] = (uint8_t)ok;
        }
    }

    // map slots: gain of this chunk and of the workgroup (fixed reduction order)
    if (slot.mode == kModeMap) {
        if (valid) {
            buf.gain_chunk[slot.chunk_off + chunk] = gain;
        }
        double g = gain;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            g += __shfl_down(g, off);
        }
        if (lane == 0) {
            lds_w[4 + (threadIdx.x >> 6)] = g;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            buf.gain_block[slot.block_off + local_block] = ((lds_w[4] + lds_w[5]) + lds_w[6]) + lds_w[7];
        }
        __syncthreads();
    }

    // backward fill
    const long long bidx = slot.block_off + local_block;
    const FillOut f_lo = block_fill(D_lo, V_lo, validmask, buf.bfv_lo, buf.bpend_lo, buf.bbase_lo, bidx, lds_u);
    if (NCH == 2) {
        (void)block_fill(D_hi, V_hi, validmask, buf.bfv_hi, buf.bpend_hi, buf.bbase_hi, bidx, lds_u);
        // materialise fill(LO); pending tails are written as 0 and patched by K5
        if (valid) {
            uint8_t *z = task.solution + j0;
            const unsigned bits = f_lo.zbits;
            if (j0 + kChunk <= n && ((reinterpret_cast<uintptr_t>(z) & 15U) == 0)) {
                uint4 w0, w1;
                const unsigned long long q0 = spread_bits_to_bytes(bits);
                const unsigned long long q1 = spread_bits_to_bytes(bits >> 8);
                const unsigned long long q2 = spread_bits_to_bytes(bits >> 16);
                const unsigned long long q3 = spread_bits_to_bytes(bits >> 24);
                w0.x = (unsigned)q0;
                w0.y = (unsigned)(q0 >> 32);
                w0.z = (unsigned)q1;
                w0.w = (unsigned)(q1 >> 32);
                w1.x = (unsigned)q2;
                w1.y = (unsigned)(q2 >> 32);
                w1.z = (unsigned)q3;
                w1.w = (unsigned)(q3 >> 32);
                reinterpret_cast<uint4 *>(z)[0] = w0;
                reinterpret_cast<uint4 *>(z)[1] = w1;
            } else {
                for (int i = 0; i < kChunk && j0 + i < n; ++i) {
                    z[i] = (uint8_t)((bits >> i) & 1U);
                }
            }
        }
        // survey: can this workgroup's block be frozen for every penalty of [lambda_lo, lambda_hi]?
        if (survey) {
            const int wave = threadIdx.x >> 6;
            const long long bstart = (long long)local_block * kFastBlockLoci;
            long long bend = bstart + kFastBlockLoci;
            bend = ((bend < n) ? bend : n) - 1;  // the block's last locus
            if (threadIdx.x == 0) {
                lds_u[0] = (unsigned)code;  // lane 0's chunk is always valid
            }
            __syncthreads();
            const unsigned code0 = lds_u[0];
            int ok = 1;
            long long lastpos = -1;
            if (valid) {
                ok = (mode[0].clean && mode[NCH - 1].clean && mode[0].mapped && wsum_chunk == 0.0 && D_lo == D_hi &&
                      V_lo == V_hi && (unsigned)code == code0)
                         ? 1
                         : 0;
                unsigned dm = D_lo;
                if (bend - j0 < kChunk) {  // the chunk holding the block's last locus
                    dm &= ~(1U << (int)(bend - j0));
                    lds_red[52] = delta[0];
                    lds_red[53] = delta[NCH - 1];
                }
                if (dm != 0U) {
                    lastpos = j0 + 31 - __clz(dm);
                }
            }
            double g0 = gain, g1 = gain_b;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                g0 += __shfl_down(g0, off);
                g1 += __shfl_down(g1, off);
                const long long o = __shfl_down(lastpos, off);
                lastpos = (o > lastpos) ? o : lastpos;
            }
            if (lane == 0) {
                lds_red[54 + wave] = g0;
                lds_red[58 + wave] = g1;
                lds_ll[wave] = lastpos;
            }
            const int all_ok = __syncthreads_and(ok);
            if (threadIdx.x == 0) {
                long long lp = lds_ll[0];
                for (int w = 1; w < 4; ++w) {
                    lp = (lds_ll[w] > lp) ? lds_ll[w] : lp;
                }
                const FastChain &ca = chains[slot.chain_a];
                const FastChain &cb = chains[slot.chain_b];
                const double dlo = lds_red[52], dhi = lds_red[53];
                const double fa_lo = buf.blk_lo[ca.block_off + local_block];
                const double fa_hi = buf.blk_hi[ca.block_off + local_block];
                const double fb_lo = buf.blk_lo[cb.block_off + local_block];
                const double fb_hi = buf.blk_hi[cb.block_off + local_block];
                const int m = (int)(bend - lp);
                bool frozen = all_ok && lp >= bstart && fa_lo == fa_hi && fb_lo == fb_hi && fa_lo == dlo &&
                              fb_lo == dhi;
                // the closed form must reproduce both ends exactly
                const double B = dlo - (double)m * mode[0].nlam;
                frozen = frozen && (B + (double)m * mode[NCH - 1].nlam == dhi);
                // gain without the first step (whose incoming delta belongs to the previous block): it
                // is linear in rn_u(-lambda) with an integer slope while the classes stay put
                const double g_lo = ((lds_red[54] + lds_red[55]) + lds_red[56]) + lds_red[57];
                const double g_hi = ((lds_red[58] + lds_red[59]) + lds_red[60]) + lds_red[61];
                const double cg = cost_on_grid(mode[0], d.c_prev0, magic);
                const double gx_lo = g_lo - ((j0 > 0) ? fmax(0.0, delta_in[0] - cg) : 0.0);
                const double gx_hi = g_hi - ((j0 > 0) ? fmax(0.0, delta_in[NCH - 1] - cg) : 0.0);
                const double dn = mode[0].nlam - mode[NCH - 1].nlam;
                double mg = 0.0;
                if (dn != 0.0) {
                    mg = (gx_lo - gx_hi) / dn;
                    frozen = frozen && (mg == floor(mg)) && (mg * dn == gx_lo - gx_hi) && mg >= 0.0 && mg < 0x1p40;
                } else {
                    frozen = frozen && (gx_lo == gx_hi);
                }
                const FrozenArrays &fo = task.frz_out;
                fo.gx_lo[local_block] = gx_lo;
                fo.mg[local_block] = mg;
                fo.cprev[local_block] = cg;
                fo.flag[local_block] = frozen ? 1 : 0;
                fo.B[local_block] = B;
                fo.m[local_block] = m;
                fo.e[local_block] = (int8_t)((int)(code0 & 0x7FU) - kMapBias);
                fo.gain_lo[local_block] = g_lo;
                fo.gain_hi[local_block] = g_hi;
                fo.lam_lo[local_block] = lam[0];
                fo.lam_hi[local_block] = lam[NCH - 1];
                fo.lc[local_block] = buf.lc_block[bidx];
                fo.fv[local_block] = buf.bfv_lo[bidx];
                fo.pend[local_block] = buf.bpend_lo[bidx];
                fo.base[local_block] = buf.bbase_lo[bidx];
            }
            __syncthreads();
        }
    }
}

template <bool HAS_COSTS>
__global__ __launch_bounds__(kFastThreads, 2) void fast_apply_kernel(FastLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds_s = smem;
    double *lds_c = HAS_COSTS ? (smem + kTileDoubles) : smem;
    double *lds_red = smem + (HAS_COSTS ? 2 : 1) * kTileDoubles;

    const int2 bm = L.blockmap[blockIdx.x / L.slot_groups];
    const FastTask task = L.tasks[bm.x];
    if (HAS_COSTS != (task.switch_costs != nullptr)) {
        return;
    }
    const int local_block = bm.y;
    const long long chunk = (long long)local_block * kFastThreads + threadIdx.x;
    const long long j0 = chunk * kChunk;
    const bool valid = j0 < task.n;
    SlotDesc *desc = reinterpret_cast<SlotDesc *>(lds_red + 64);
    TileLoads tile;
    stage_tile_issue<HAS_COSTS>(task, local_block, tile);  // the tile's loads first: they only need the task
    stage_slot_descs(L, task, desc);
    // the first pass's inputs go out while the tile is in flight (its descriptors straight from memory: the LDS
    // copies are not published yet), every later pass's during the pass before it
    int si = (int)(blockIdx.x % L.slot_groups);
    FastSlot slot;
    FastChain two[2];
    SlotInputs in;
    if (si < task.slot_count) {
        fetch_slot(L, task, desc, si, slot, two, true);
        load_slot_inputs(task, slot, task.slot_begin + si, two, L.buf, chunk, valid, local_block, in);
    }
    stage_tile_commit<HAS_COSTS>(task, local_block, lds_s, lds_c, tile);  // (ends with the barrier that publishes desc too)
    ChunkData<HAS_COSTS> d;
    load_chunk<HAS_COSTS>(task, j0, lds_s, lds_c, d);
    for (; si < task.slot_count; si += L.slot_groups) {
        const int slot_index = task.slot_begin + si;
        const int si_next = si + L.slot_groups;
        FastSlot slot_next = slot;
        FastChain two_next[2] = {two[0], two[1]};
        SlotInputs in_next = in;
        if (si_next < task.slot_count) {
            fetch_slot(L, task, desc, si_next, slot_next, two_next);
            load_slot_inputs(task, slot_next, task.slot_begin + si_next, two_next, L.buf, chunk, valid, local_block, in_next);
        }
        if (slot.mode == kModeWindow) {
            apply_slot<2, HAS_COSTS>(task, slot, slot_index, two, L.buf, d, chunk, j0, local_block, lds_red, in);
        } else {
            apply_slot<1, HAS_COSTS>(task, slot, slot_index, two, L.buf, d, chunk, j0, local_block, lds_red, in);
        }
        slot = slot_next;
        two[0] = two_next[0];
        two[1] = two_next[1];
        in = in_next;
    }
}

// ---- K4: backward over workgroups (fill) and forward (map gains) ---------------------------------
__global__ __launch_bounds__(64) void fast_fillscan_kernel(FastLaunch L)
{
    // one wavefront per (slot, variant)
    const int idx = blockIdx.x;
    const int lane = threadIdx.x;
    if (idx >= 2 * L.n_slots) {
        return;
    }
    const int si = idx >> 1;
    const int variant = idx & 1;  // 0 = LO (probe: the exact-rule classes), 1 = HI (window) / gains (map)
    const FastSlot slot = L.slots[si];
    const FastTask &task = L.tasks[slot.task];
    const int nb = task.n_blocks;
    if (variant == 1 && slot.mode == kModeMap) {
        double carry = 0.0;
        for (int base = 0; base < nb; base += 64) {
            const int b = base + lane;
            double g = 0.0;
            if (b < nb) {
                if (task.frz.flag != nullptr && task.frz.flag[b]) {
                    // maps only need the stay-off value to within their margin: interpolate
                    const double lam = L.chains[slot.chain_a].lambda;
                    const double span = task.frz.lam_hi[b] - task.frz.lam_lo[b];
                    const double t = (span > 0.0) ? (lam - task.frz.lam_lo[b]) / span : 0.0;
                    g = task.frz.gain_lo[b] + t * (task.frz.gain_hi[b] - task.frz.gain_lo[b]);
                } else {
                    g = L.buf.gain_block[slot.block_off + b];
                }
            }
            double inc = g;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double p = __shfl_up(inc, off);
                if (lane >= off) {
                    inc += p;
                }
            }
            if (b < nb) {
                L.buf.gainin_block[slot.block_off + b] = carry + (inc - g);
            }
            carry += __shfl(inc, 63);
        }
        return;
    }
    if (variant == 1 && slot.mode != kModeWindow) {
        return;
    }
    const uint8_t *bfv = variant ? L.buf.bfv_hi : L.buf.bfv_lo;
    const unsigned *bpend = variant ? L.buf.bpend_hi : L.buf.bpend_lo;
    const unsigned *bbase = variant ? L.buf.bbase_hi : L.buf.bbase_lo;
    long long count = 0;
    int carry_r = 0;  // the last workgroup never has pending loci (the terminal locus is determined)
    for (int base = ((nb - 1) / 64) * 64; base >= 0; base -= 64) {
        const int b = base + lane;
        const long long at = slot.block_off + b;
        int fv = kFvNone;
        long long pend = 0, bs = 0;
        if (b < nb) {
            if (task.frz.flag != nullptr && task.frz.flag[b]) {
                fv = (int)task.frz.fv[b];
                pend = (long long)task.frz.pend[b];
                bs = (long long)task.frz.base[b];
            } else {
                fv = (int)bfv[at];
                pend = (long long)bpend[at];
                bs = (long long)bbase[at];
            }
        }
        const unsigned long long has = __ballot(fv != kFvNone);
        const unsigned long long one = __ballot(fv == 1);
        int r = carry_r;
        const unsigned long long right = (lane == 63) ? 0ULL : (has & (~0ULL << (lane + 1)));
        if (right != 0ULL) {
            r = (int)((one >> (__ffsll((long long)right) - 1)) & 1ULL);
        }
        if (b < nb) {
            if (variant == 0) {
                L.buf.rin_lo[at] = (uint8_t)r;
            }
            count += bs + (r ? pend : 0LL);
        }
        if (has != 0ULL) {
            carry_r = (int)((one >> (__ffsll((long long)has) - 1)) & 1ULL);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        count += __shfl_down(count, off);
    }
    if (lane == 0) {
        if (variant == 0) {
            L.buf.results[si].count_lo = count;
        } else {
            L.buf.results[si].count_hi = count;
        }
    }
}

// ---- K5: patch pending tails of fill(LO) ------------------------------------------------------------
__global__ __launch_bounds__(kFastThreads) void fast_patch_kernel(FastLaunch L)
{
    const int2 bm = L.blockmap_all[blockIdx.x];
    const FastTask task = L.tasks[bm.x];
    const int local_block = bm.y;
    const bool skipped = task.frz.flag != nullptr && task.frz.flag[local_block];
    for (int si = 0; si < task.slot_count; ++si) {
        const FastSlot slot = L.slots[task.slot_begin + si];
        if (slot.mode != kModeWindow) {
            continue;
        }
        const long long at = slot.block_off + local_block;
        const unsigned pend = skipped ? task.frz.pend[local_block] : L.buf.bpend_lo[at];
        if (pend == 0U) {
            continue;
        }
        const uint8_t value = L.buf.rin_lo[at];
        long long end = (long long)(local_block + 1) * kFastBlockLoci;
        end = (end < task.n) ? end : task.n;
        for (long long j = end - pend + threadIdx.x; j < end; j += kFastThreads) {
            task.solution[j] = value;
        }
    }
}

// ---- K6: binade codes from the running stay-off value (oracle_binade_code) ---------------------------
__device__ __forceinline__ uint8_t binade_code(double p0_lo, double p0_hi, double margin)
{
    const double top = fmax(p0_hi, 1.0);
    int e = ilogb(top);
    if (e > 60) {
        e = 60;
    }
    bool clean = false;
    if (p0_lo > 0.0) {
        const double lo_edge = ldexp(1.0, e), hi_edge = ldexp(1.0, e + 1);
        clean = (p0_lo - lo_edge > margin) && (hi_edge - p0_hi > margin);
    }
    return (uint8_t)((clean ? 0 : 0x80) | (e + kMapBias));
}

__global__ __launch_bounds__(kFastThreads) void fast_mapcode_kernel(FastLaunch L)
{
    __shared__ double wsum[4];
    const int2 bm = L.blockmap[blockIdx.x];
    const FastTask task = L.tasks[bm.x];
    const int local_block = bm.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int si = 0; si < task.slot_count; ++si) {
        const FastSlot slot = L.slots[task.slot_begin + si];
        if (slot.mode != kModeMap || task.emap_out == nullptr) {
            continue;
        }
        const long long chunk = (long long)local_block * kFastThreads + threadIdx.x;
        const bool valid = chunk * kChunk < task.n;
        const double g = valid ? L.buf.gain_chunk[slot.chunk_off + chunk] : 0.0;
        double inc = g;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double p = __shfl_up(inc, off);
            if (lane >= off) {
                inc += p;
            }
        }
        if (lane == 63) {
            wsum[wave] = inc;
        }
        __syncthreads();
        double pre = L.buf.gainin_block[slot.block_off + local_block];
        for (int w = 0; w < wave; ++w) {
            pre += wsum[w];
        }
        __syncthreads();
        const double p0_end = pre + inc;
        const double p0_start = p0_end - g;
        if (valid) {
            task.emap_out[chunk] = binade_code(p0_start, p0_end, task.map_margin);
        }
    }
}

// ---- stats: min / max of scores and costs, sum of |score| ------------------------------------------
// out[5 * t + {0,1,2,3,4}] = smin, smax, cmin, cmax, sum |s| (the last one only bounds running values:
// any summation order will do)
constexpr int kStat = 5;

__device__ __forceinline__ void stats_reduce(double (&v)[kStat], double (*red)[kStat])
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v[0] = fmin(v[0], __shfl_down(v[0], off));
        v[1] = fmax(v[1], __shfl_down(v[1], off));
        v[2] = fmin(v[2], __shfl_down(v[2], off));
        v[3] = fmax(v[3], __shfl_down(v[3], off));
        v[4] += __shfl_down(v[4], off);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        for (int k = 0; k < kStat; ++k) {
            red[w][k] = v[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            red[0][0] = fmin(red[0][0], red[w][0]);
            red[0][1] = fmax(red[0][1], red[w][1]);
            red[0][2] = fmin(red[0][2], red[w][2]);
            red[0][3] = fmax(red[0][3], red[w][3]);
            red[0][4] += red[w][4];
        }
    }
}

__global__ __launch_bounds__(kFastThreads) void stats_partial_kernel(const StatsTask *tasks, const int2 *blockmap,
                                                                   double *partials)
{
    __shared__ double red[4][kStat];
    const int2 bm = blockmap[blockIdx.x];
    const StatsTask t = tasks[bm.x];
    const long long base = (long long)bm.y * kFastBlockLoci;
    double v[kStat] = {INFINITY, -INFINITY, INFINITY, -INFINITY, 0.0};
    if (t.switch_costs == nullptr && base + kFastBlockLoci <= t.n && ((reinterpret_cast<uintptr_t>(t.scores + base) & 15U) == 0)) {
        // a whole tile of scores only: sixteen unconditional 16-byte loads per lane, all in flight together (a load under
        // a per-element condition is waited for before the next one is issued), four independent accumulations
        const double2 *__restrict__ src = reinterpret_cast<const double2 *>(t.scores + base) + threadIdx.x;
        double2 x[kChunk / 2];
#pragma unroll
        for (int r = 0; r < kChunk / 2; ++r) {
            x[r] = src[r * kFastThreads];
        }
        double lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        double sum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < kChunk / 2; ++r) {
            const int k = r & 3;
            lo[k] = fmin(lo[k], fmin(x[r].x, x[r].y));
            hi[k] = fmax(hi[k], fmax(x[r].x, x[r].y));
            sum[k] += fabs(x[r].x) + fabs(x[r].y);
        }
        v[0] = fmin(fmin(lo[0], lo[1]), fmin(lo[2], lo[3]));
        v[1] = fmax(fmax(hi[0], hi[1]), fmax(hi[2], hi[3]));
        v[4] = (sum[0] + sum[1]) + (sum[2] + sum[3]);
        stats_reduce(v, red);
        if (threadIdx.x == 0) {
            for (int k = 0; k < kStat; ++k) {
                partials[(long long)kStat * blockIdx.x + k] = red[0][k];
            }
        }
        return;
    }
    for (int r = 0; r < kChunk; ++r) {
        const long long j = base + r * kFastThreads + threadIdx.x;
        if (j < t.n) {
            const double x = t.scores[j];
            v[0] = fmin(v[0], x);
            v[1] = fmax(v[1], x);
            v[4] += fabs(x);
            if (t.switch_costs != nullptr && j < t.n - 1) {
                const double c = t.switch_costs[j];
                v[2] = fmin(v[2], c);
                v[3] = fmax(v[3], c);
            }
        }
    }
    stats_reduce(v, red);
    if (threadIdx.x == 0) {
        for (int k = 0; k < kStat; ++k) {
            partials[(long long)kStat * blockIdx.x + k] = red[0][k];
        }
    }
}

__global__ __launch_bounds__(kFastThreads) void stats_final_kernel(const int2 *blockmap, int n_blocks_total,
                                                                 const double *partials, double *out)
{
    // one workgroup per task: reduce the partials of that task's workgroups
    __shared__ double red[4][kStat];
    const int task = blockIdx.x;
    double v[kStat] = {INFINITY, -INFINITY, INFINITY, -INFINITY, 0.0};
    for (int b = threadIdx.x; b < n_blocks_total; b += kFastThreads) {
        if (blockmap[b].x == task) {
            v[0] = fmin(v[0], partials[(long long)kStat * b + 0]);
            v[1] = fmax(v[1], partials[(long long)kStat * b + 1]);
            v[2] = fmin(v[2], partials[(long long)kStat * b + 2]);
            v[3] = fmax(v[3], partials[(long long)kStat * b + 3]);
            v[4] += partials[(long long)kStat * b + 4];
        }
    }
    stats_reduce(v, red);
    if (threadIdx.x == 0) {
        for (int k = 0; k < kStat; ++k) {
            out[(long long)kStat * task + k] = red[0][k];
        }
    }
}

// broadcast lane `src` (a compile-time constant once the caller's loop is unrolled) of a double
__device__ __forceinline__ double read_lane_f64(double x, int src)
{
    const long long bits = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xFFFFFFFFLL), src);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- exact spine ----------------------------------------------------------------------------------
// One wavefront per chromosome, lane = record slot (penalty).  The reference's running values
// (prev0 = P0, prev1 = P1, rocco/_chain_dp.c:117-165) are carried EXACTLY: over chunks where the
// parallel recursion is exact (clean, no rounding tie) and the lane is synchronised with it, P0 just
// receives the chunk's exact gain; every other chunk is stepped with the reference's own
// operations on absolute values, and stepping continues until P1 - P0 equals the parallel
// recursion's incoming delta of an exact chunk again.  Stepped chunks get their class words
// rewritten with the reference's actual decisions (ties: the state-1 path always holds more
// selected loci, so "leave" needs a strict win and "enter" wins ties).
__global__ __launch_bounds__(64) void spine_kernel(FastLaunch L)
{
    const FastTask task = L.tasks[blockIdx.x];
    if (task.slot_count == 0 || L.slots[task.slot_begin].mode != kModeRecord) {
        return;
    }
    const int lane = threadIdx.x;
    const int S = task.slot_count;
    const bool active = lane < S;
    const double lam = active ? L.chains[L.slots[task.slot_begin + lane].chain_a].lambda : 0.0;
    const long long n = task.n;
    const long long nchunks = (n + kChunk - 1) / kChunk;
    const int nb = task.n_blocks;
    const double *__restrict__ sc = task.scores;
    const double *__restrict__ cs = task.switch_costs;
    const bool has_costs = (cs != nullptr);
    const FastBuffers &buf = L.buf;
    const long long lane_block_off =
        active ? L.chains[L.slots[task.slot_begin + lane].chain_a].block_off : 0;

    // One memory round trip per group of 32 chunks: the lanes fetch the group's record entries
    // (their own column) and, together, its scores / costs into LDS; everything after that is served
    // from LDS.  (The record arrays were written by other compute units: every global load here is a
    // trip to memory, so dependent loads are what this kernel must avoid.)
    constexpr int kGroup = 32;
    constexpr int kGroupLoci = kGroup * kChunk;
    __shared__ double sh_s[kGroupLoci + 1];       // scores of the group and the first one after it
    __shared__ double sh_c[kGroupLoci + 1];       // sh_c[x]: cost between loci g0 + x - 1 and g0 + x
    __shared__ double sh_din[kGroup + 1][64];
    __shared__ double sh_gain[kGroup][64];
    __shared__ uint8_t sh_fl[kGroup + 1][64];

    double P0 = 0.0, P1 = 0.0;
    bool stepping = true;  // chunk 0 is always stepped (the map never marks it clean)
    unsigned D = 0, V = 0;  // class word under construction for the chunk being stepped
#ifdef SPINE_PROF
    long long t_fetch = 0, t_loop = 0, t_step = 0, n_slow = 0;
#define SPINE_T0() const long long t0_ = wall_clock64()
#define SPINE_T1(acc) acc += wall_clock64() - t0_
#else
#define SPINE_T0()
#define SPINE_T1(acc)
#endif
    long long pend_at = -1;  // class word of the previous chunk, waiting for its bit 31
    unsigned pend_d = 0, pend_v = 0;
    long long stepped = 0;

    __shared__ double sh_gs[8][64];
    __shared__ uint8_t sh_ok[8][64];
    const long long ngroups = (nchunks + kGroup - 1) / kGroup;
    constexpr int kGroupsPerBlock = kFastThreads / kGroup;  // 8

    for (long long b = 0; b < nb; ++b) {
        if (task.frz.flag != nullptr) {
            // Blocks that were not evaluated this round: their classes are settled and their gain has a
            // closed form.  (The host evaluates the block after every active block, so a stepping
            // lane has resynchronised before it gets here; otherwise the round is repeated in full.)
            // The lanes look at the next 64 block flags together: one memory latency per run.
            const bool mine = (b + lane < nb) && (task.frz.flag[b + lane] != 0);
            const unsigned long long run_mask = __ballot(mine);
            const int run = (~run_mask == 0ULL) ? 64 : (__ffsll((long long)~run_mask) - 1);
            if (run > 0) {
                if (active) {
                    if (stepping) {
                        atomicOr(&buf.results[task.slot_begin + lane].overflow, 1);
                        stepping = false;
                    }
                    // all loads of the run are independent; the sums are exact in any order
                    for (int r0 = 0; r0 < run; r0 += 4) {
                        double add[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const long long bb = b + ((r0 + q < run) ? (r0 + q) : (run - 1));
                            const double magic_u = ldexp(1.5, (int)task.frz.e[bb]);
                            const double nl = grid_round(-lam, magic_u);
                            const double nl_lo = grid_round(-task.frz.lam_lo[bb], magic_u);
                            const double first = fmax(0.0, buf.din[lane_block_off + bb] - task.frz.cprev[bb]);
                            add[q] = first + (task.frz.gx_lo[bb] + task.frz.mg[bb] * (nl - nl_lo));
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            P0 += (r0 + q < run) ? add[q] : 0.0;
                        }
                    }
                }
                b += run - 1;
                continue;
            }
        }
        // the block's eight group summaries: one memory latency
        {
            double gs[kGroupsPerBlock];
            unsigned ok[kGroupsPerBlock];
#pragma unroll
            for (int q = 0; q < kGroupsPerBlock; ++q) {
                const long long g = b * kGroupsPerBlock + q;
                const bool in = active && (g < ngroups);
                const long long gat = task.rec_goff + (in ? g : 0) * S + (active ? lane : 0);
                gs[q] = in ? buf.rec_gsum[gat] : 0.0;
                ok[q] = in ? buf.rec_gok[gat] : 1U;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kGroupsPerBlock; ++q) {
                sh_gs[q][lane] = gs[q];
                sh_ok[q][lane] = (uint8_t)ok[q];
            }
            __syncthreads();
        }
#pragma unroll 1
        for (int q = 0; q < kGroupsPerBlock; ++q) {
        const long long k = (b * kGroupsPerBlock + q) * kGroup;  // first chunk of the group
        if (k >= nchunks) {
            break;
        }
        if (!__any(active && stepping) && __all(sh_ok[q][lane] != 0)) {
            P0 += sh_gs[q][lane];  // every chunk exact for every lane: the group's gain in one add
            continue;
        }
        // ---- fetch the group's details: one memory latency ----
        const long long g0 = k * kChunk;  // first locus of the group
#ifdef SPINE_PROF
        ++n_slow;
        const long long tf0 = wall_clock64();
#endif
        {
            unsigned fl[kGroup + 1];
            double dv[kGroup + 1];
            double gv[kGroup];
#pragma unroll
            // unconditional loads from clamped (always valid) addresses, masked afterwards: loads inside
            // per-element conditionals would be waited for one by one
            for (int c = 0; c <= kGroup; ++c) {
                const bool in = active && (k + c < nchunks);
                const long long kc = (k + c < nchunks) ? (k + c) : (nchunks - 1);
                const long long a2 = task.rec_off + kc * S + (active ? lane : 0);
                const unsigned flv = buf.rec_flags[a2];
                const double dvv = buf.rec_din[a2];
                fl[c] = in ? flv : 1U;
                dv[c] = in ? dvv : 0.0;
                if (c < kGroup) {
                    const double gvv = buf.rec_gain[a2];
                    gv[c] = in ? gvv : 0.0;
                }
            }
            double sv[kGroupLoci / 64], cv[kGroupLoci / 64];
#pragma unroll
            for (int qq = 0; qq < kGroupLoci / 64; ++qq) {
                const long long j = g0 + qq * 64 + lane;
                const long long jc = (j < n) ? j : (n - 1);
                const double svv = sc[jc];
                sv[qq] = (j < n) ? svv : 0.0;
                if (has_costs) {
                    const long long jj = (jc > 0) ? (jc - 1) : 0;
                    const double cvv = cs[(jj < n - 1) ? jj : ((n > 1) ? (n - 2) : 0)];
                    cv[qq] = (j > 0 && j < n) ? cvv : 0.0;
                } else {
                    cv[qq] = task.gamma;
                }
            }
            const long long jx = g0 + kGroupLoci;
            const double s_x = (jx < n) ? sc[jx] : 0.0;
            const double c_x = has_costs ? ((jx < n) ? cs[jx - 1] : 0.0) : task.gamma;
            __syncthreads();  // the previous group's readers are done
#pragma unroll
            for (int c = 0; c <= kGroup; ++c) {
                sh_fl[c][lane] = (uint8_t)(fl[c] & 1U);
                sh_din[c][lane] = dv[c];
                if (c < kGroup) {
                    sh_gain[c][lane] = gv[c];
                }
            }
#pragma unroll
            for (int qq = 0; qq < kGroupLoci / 64; ++qq) {
                sh_s[qq * 64 + lane] = sv[qq];
                sh_c[qq * 64 + lane] = cv[qq];
            }
            if (lane == 0) {
                sh_s[kGroupLoci] = s_x;
                sh_c[kGroupLoci] = c_x;
            }
            __syncthreads();
        }
#ifdef SPINE_PROF
        t_fetch += wall_clock64() - tf0;
        const long long tl0 = wall_clock64();
#endif
        // ---- chunk by chunk, from LDS ----
        const int c_end = (int)((nchunks - k < kGroup) ? (nchunks - k) : kGroup);
#pragma unroll 1
        for (int c = 0; c < c_end; ++c) {
            const long long kk = k + c;
            const long long at = task.rec_off + kk * S + lane;
            const bool exact_here = active ? (sh_fl[c][lane] != 0) : true;
            const bool need = active && (stepping || !exact_here);
            if (active && !need) {
                P0 += sh_gain[c][lane];
            }
            if (!__any(need)) {
                continue;
            }
            const long long j0 = kk * kChunk;
            const long long j_end = j0 + kChunk;  // first locus of the next chunk
            const bool fresh = need && !stepping;  // was synchronised: enter with the parallel delta
            const bool has_next = j_end < n;
            const bool next_exact = need && has_next && (sh_fl[c + 1][lane] != 0);
            const double din_next = sh_din[c + 1][lane];
            if (fresh) {
                P1 = P0 + sh_din[c][lane];
                stepping = true;
                D = 0;
                V = 0;
            }
            if (need) {
                ++stepped;
            }
            const int steps = (int)((n - j0 < kChunk) ? (n - j0) : kChunk);
#ifdef SPINE_PROF
            const long long ts0 = wall_clock64();
#endif
            double srow[kChunk], crow[kChunk];  // the chunk's rows, read ahead of the dependent chain
#pragma unroll
            for (int i = 0; i < kChunk; ++i) {
                srow[i] = sh_s[c * kChunk + i];
                crow[i] = sh_c[c * kChunk + i];
            }
            // step 0 classifies the last locus of the previous chunk (or starts the chromosome)
            if (need) {
                const double s_j = srow[0];
                const double c_prev = crow[0];
                if (j0 == 0) {
                    P0 = 0.0;
                    P1 = s_j - lam;  // rocco/_chain_dp.c:109-112
                } else {
                    const double leave = P1 - c_prev;
                    const double keep = P1 + s_j - lam;
                    const double enter = P0 - c_prev + s_j - lam;
                    const bool tl = leave > P0;
                    const bool te = enter >= keep;
                    const unsigned dbit = (tl || te) ? 1U : 0U;  // class of locus j-1: ONE / ZERO / COPY
                    const unsigned vbit = tl ? 1U : 0U;
                    const long long prev = at - S;
                    if (pend_at == prev) {
                        buf.rec_d[prev] = pend_d | (dbit << 31);
                        buf.rec_v[prev] = pend_v | (vbit << 31);
                        pend_at = -1;
                    } else {
                        // the previous chunk was not stepped by this lane: patch its word in place
                        if (dbit) {
                            atomicOr(&buf.rec_d[prev], 0x80000000U);
                        } else {
                            atomicAnd(&buf.rec_d[prev], 0x7FFFFFFFU);
                        }
                        if (vbit) {
                            atomicOr(&buf.rec_v[prev], 0x80000000U);
                        } else {
                            atomicAnd(&buf.rec_v[prev], 0x7FFFFFFFU);
                        }
                    }
                    P0 = tl ? leave : P0;
                    P1 = te ? enter : keep;
                }
            }
            // steps 1..: every lane runs them on copies (no per-step masking); only the lanes that
            // needed the chunk keep the result
            double p0 = P0, p1 = P1;
            unsigned dd = D, vv = V;
            if (steps == kChunk) {
#pragma unroll
                for (int i = 1; i < kChunk; ++i) {  // fully unrolled: the rows stay in registers
                    const double s_j = srow[i];
                    const double c_prev = crow[i];
                    const double leave = p1 - c_prev;
                    const double keep = p1 + s_j - lam;
                    const double enter = p0 - c_prev + s_j - lam;
                    const bool tl = leave > p0;
                    const bool te = enter >= keep;
                    dd |= ((tl || te) ? 1U : 0U) << (i - 1);
                    vv |= (tl ? 1U : 0U) << (i - 1);
                    p0 = tl ? leave : p0;
                    p1 = te ? enter : keep;
                }
            } else {
                // the chromosome's last, partial chunk: straight from LDS
#pragma unroll 1
                for (int i = 1; i < steps; ++i) {
                    const double s_j = sh_s[c * kChunk + i];
                    const double c_prev = sh_c[c * kChunk + i];
                    const double leave = p1 - c_prev;
                    const double keep = p1 + s_j - lam;
                    const double enter = p0 - c_prev + s_j - lam;
                    const bool tl = leave > p0;
                    const bool te = enter >= keep;
                    dd |= ((tl || te) ? 1U : 0U) << (i - 1);
                    vv |= (tl ? 1U : 0U) << (i - 1);
                    p0 = tl ? leave : p0;
                    p1 = te ? enter : keep;
                }
            }
            if (need) {
                P0 = p0;
                P1 = p1;
                D = dd;
                V = vv;
            }
#ifdef SPINE_PROF
            t_step += wall_clock64() - ts0;
#endif
            // end of chunk kk for the stepping lanes
            if (!has_next) {
                if (need) {
                    const int il = (int)(n - 1 - j0);
                    const unsigned one = (P1 > P0) ? 1U : 0U;  // rocco/_chain_dp.c:167-179, tie -> state 0
                    D |= 1U << il;
                    V |= one << il;
                    buf.rec_d[at] = D;
                    buf.rec_v[at] = V;
                }
            } else if (need) {
                const bool resync = next_exact && ((P1 - P0) == din_next);
                if (resync) {
                    // decision of the next step classifies this chunk's last locus; state untouched
                    const double s_next = sh_s[(c + 1) * kChunk];
                    const double c_last = sh_c[(c + 1) * kChunk];
                    const double leave = P1 - c_last;
                    const double keep = P1 + s_next - lam;
                    const double enter = P0 - c_last + s_next - lam;
                    const bool tl = leave > P0;
                    const bool te = enter >= keep;
                    D |= ((tl || te) ? 1U : 0U) << 31;
                    V |= (tl ? 1U : 0U) << 31;
                    stepping = false;
                    buf.rec_d[at] = D;
                    buf.rec_v[at] = V;
                } else {
                    // bit 31 is filled in by the next chunk's first step
                    pend_at = at;
                    pend_d = D;
                    pend_v = V;
                }
                D = 0;
                V = 0;
            }
        }
#ifdef SPINE_PROF
        t_loop += wall_clock64() - tl0;
#endif
        }  // groups of the block
    }
#ifdef SPINE_PROF
    if (lane == 0) {
        FastSlotResult &r0 = buf.results[task.slot_begin];
        r0.p16 = t_fetch;
        r0.npos = t_loop;
        r0.max_run = t_step;
        r0.n_diff = n_slow;
    }
#endif
    if (active) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&buf.results[task.slot_begin + lane].uncertain),
                  (unsigned long long)stepped);  // diagnostic: chunks stepped exactly
    }
}

// Rebuild fill summaries (and one solution per task) from the record class words.
__global__ __launch_bounds__(kFastThreads) void fill_from_classes_kernel(FastLaunch L, const int *solution_slot)
{
    __shared__ unsigned lds_u[16];
    const int2 bm = L.blockmap[blockIdx.x / L.slot_groups];
    const FastTask task = L.tasks[bm.x];
    const int local_block = bm.y;
    const long long chunk = (long long)local_block * kFastThreads + threadIdx.x;
    const long long j0 = chunk * kChunk;
    const bool valid = j0 < task.n;
    const int S = task.slot_count;
    const int sol_slot = solution_slot[bm.x];
    unsigned validmask = 0;
    if (valid) {
        const long long left = task.n - j0;
        validmask = (left >= kChunk) ? 0xFFFFFFFFU : ((1U << left) - 1U);
    }
    for (int si = (int)(blockIdx.x % L.slot_groups); si < S; si += L.slot_groups) {
        const FastSlot slot = L.slots[task.slot_begin + si];
        if (slot.mode != kModeRecord) {
            continue;
        }
        unsigned D = 0, V = 0;
        if (valid) {
            const long long at = task.rec_off + chunk * S + si;
            D = L.buf.rec_d[at] & validmask;
            V = L.buf.rec_v[at] & validmask;
        }
        const long long bidx = slot.block_off + local_block;
        const FillOut f = block_fill(D, V, validmask, L.buf.bfv_lo, L.buf.bpend_lo, L.buf.bbase_lo, bidx, lds_u);
        if (task.slot_begin + si == sol_slot && valid) {
            uint8_t *z = task.solution + j0;
            const unsigned bits = f.zbits;
            if (j0 + kChunk <= task.n && ((reinterpret_cast<uintptr_t>(z) & 15U) == 0)) {
                uint4 w0, w1;
                const unsigned long long q0 = spread_bits_to_bytes(bits);
                const unsigned long long q1 = spread_bits_to_bytes(bits >> 8);
                const unsigned long long q2 = spread_bits_to_bytes(bits >> 16);
                const unsigned long long q3 = spread_bits_to_bytes(bits >> 24);
                w0.x = (unsigned)q0;
                w0.y = (unsigned)(q0 >> 32);
                w0.z = (unsigned)q1;
                w0.w = (unsigned)(q1 >> 32);
                w1.x = (unsigned)q2;
                w1.y = (unsigned)(q2 >> 32);
                w1.z = (unsigned)q3;
                w1.w = (unsigned)(q3 >> 32);
                reinterpret_cast<uint4 *>(z)[0] = w0;
                reinterpret_cast<uint4 *>(z)[1] = w1;
            } else {
                for (int i = 0; i < kChunk && j0 + i < task.n; ++i) {
                    z[i] = (uint8_t)((bits >> i) & 1U);
                }
            }
        }
    }
}

__global__ __launch_bounds__(kFastThreads) void spine_patch_kernel(FastLaunch L, const int *solution_slot)
{
    const int2 bm = L.blockmap_all[blockIdx.x];
    const FastTask task = L.tasks[bm.x];
    const int local_block = bm.y;
    const int sol_slot = solution_slot[bm.x];
    if (sol_slot < 0) {
        return;
    }
    const FastSlot slot = L.slots[sol_slot];
    const long long at = slot.block_off + local_block;
    const bool skipped = task.frz.flag != nullptr && task.frz.flag[local_block];
    const unsigned pend = skipped ? task.frz.pend[local_block] : L.buf.bpend_lo[at];
    if (pend == 0U) {
        return;
    }
    const uint8_t value = L.buf.rin_lo[at];
    long long end = (long long)(local_block + 1) * kFastBlockLoci;
    end = (end < task.n) ? end : task.n;
    for (long long j = end - pend + threadIdx.x; j < end; j += kFastThreads) {
        task.solution[j] = value;
    }
}

}  // namespace

int launch_fast_round(const FastLaunch &L, hipStream_t stream)
{
    if (L.n_slots == 0 || L.n_blocks_total == 0) {
        return ROCCO_HIP_OK;
    }
    const size_t lds_desc = (size_t)kMaxStagedSlots * sizeof(SlotDesc);
    const size_t lds_plain = (size_t)(kTileDoubles + 64) * sizeof(double) + lds_desc;
    const size_t lds_costs = (size_t)(2 * kTileDoubles + 64) * sizeof(double) + lds_desc;
    static bool attr_set = false;
    if (!attr_set) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fast_aggregate_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_costs));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fast_apply_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_costs));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fast_aggregate_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_plain));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fast_apply_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_plain));
        attr_set = true;
    }
    const dim3 grid((unsigned)L.n_blocks_total), block(kFastThreads);
    const dim3 grid_split((unsigned)(L.n_blocks_total * L.slot_groups));  // K1 / K3: slots spread over workgroups
    const unsigned scan_blocks = (unsigned)(L.n_chains + L.n_slots);
    const unsigned fill_blocks = (unsigned)(2 * L.n_slots);

    if (L.any_plain) {
        hipLaunchKernelGGL((fast_aggregate_kernel<false>), grid_split, block, lds_plain, stream, L);
    }
    if (L.any_costs) {
        hipLaunchKernelGGL((fast_aggregate_kernel<true>), grid_split, block, lds_costs, stream, L);
    }
    hipLaunchKernelGGL(fast_blockscan_kernel, dim3(scan_blocks), dim3(64), 0, stream, L);
    if (L.any_plain) {
        hipLaunchKernelGGL((fast_apply_kernel<false>), grid_split, block, lds_plain, stream, L);
    }
    if (L.any_costs) {
        hipLaunchKernelGGL((fast_apply_kernel<true>), grid_split, block, lds_costs, stream, L);
    }
    hipLaunchKernelGGL(fast_fillscan_kernel, dim3(fill_blocks), dim3(64), 0, stream, L);
    if (L.any_window) {
        hipLaunchKernelGGL(fast_patch_kernel, dim3((unsigned)L.n_blocks_all), block, 0, stream, L);
    }
    if (L.any_map) {
        hipLaunchKernelGGL(fast_mapcode_kernel, grid, block, 0, stream, L);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

// Last tree of a bisection: walk it with the exact counts (rocco/dp.py:139-151: count > target moves the
// lower end, otherwise the upper end) and name the slot whose penalty is the final upper end.
__global__ void spine_select_kernel(FastLaunch L, int *solution_slot)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.n_tasks) {
        return;
    }
    const FastTask &task = L.tasks[t];
    if (task.sel_depth <= 0 || task.slot_count == 0 || L.slots[task.slot_begin].mode != kModeRecord) {
        return;
    }
    int i = 0;
    int answer = task.sel_has_upper ? (task.slot_count - 1) : -1;  // the current upper end, if it was evaluated
    for (int level = 0; level < task.sel_depth; ++level) {
        if (L.buf.results[task.slot_begin + i].count_lo > task.sel_target) {
            i = 2 * i + 2;
        } else {
            answer = i;
            i = 2 * i + 1;
        }
    }
    solution_slot[t] = (answer >= 0) ? (task.slot_begin + answer) : -1;
    L.buf.results[task.slot_begin].e_global = answer;  // reported to the host
}

int launch_spine(const FastLaunch &L, int *solution_slot_dev, bool any_select, hipStream_t stream)
{
    if (L.n_tasks == 0 || L.n_blocks_total == 0) {
        return ROCCO_HIP_OK;
    }
    const dim3 grid((unsigned)L.n_blocks_total), block(kFastThreads);
    const unsigned fill_blocks = (unsigned)(2 * L.n_slots);
    hipLaunchKernelGGL(spine_kernel, dim3((unsigned)L.n_tasks), dim3(64), 0, stream, L);
    hipLaunchKernelGGL(fill_from_classes_kernel, dim3((unsigned)(L.n_blocks_total * L.slot_groups)), block, 0, stream,
                       L, solution_slot_dev);
    hipLaunchKernelGGL(fast_fillscan_kernel, dim3(fill_blocks), dim3(64), 0, stream, L);
    if (any_select) {
        // the counts are known now: pick the answer on the device and materialise it
        hipLaunchKernelGGL(spine_select_kernel, dim3((unsigned)((L.n_tasks + 63) / 64)), dim3(64), 0, stream, L,
                           solution_slot_dev);
        hipLaunchKernelGGL(fill_from_classes_kernel, dim3((unsigned)(L.n_blocks_total * L.slot_groups)), block, 0,
                           stream, L, solution_slot_dev);
    }
    hipLaunchKernelGGL(spine_patch_kernel, dim3((unsigned)L.n_blocks_all), block, 0, stream, L, solution_slot_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_stats(const StatsTask *tasks_dev, int n_tasks, const int2 *blockmap_dev, int n_blocks_total,
                 double *partials_dev, double *out_dev, hipStream_t stream)
{
    if (n_tasks == 0 || n_blocks_total == 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(stats_partial_kernel, dim3((unsigned)n_blocks_total), dim3(kFastThreads), 0, stream,
                       tasks_dev, blockmap_dev, partials_dev);
    hipLaunchKernelGGL(stats_final_kernel, dim3((unsigned)n_tasks), dim3(kFastThreads), 0, stream, blockmap_dev,
                       n_blocks_total, partials_dev, out_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
