// rocco_amd/csrc/chain.hip -- see chain.h.  gfx950 only.
//
// chain_director_kernel: ONE workgroup of sixteen wavefronts, a wavefront per problem of the batch (lane = penalty of
// the round; the scalar decisions in lane 0).  Between two rounds of the chain it
//   1. reads what the finish kernel left for its problem (counts and compacted lengths per penalty), records every
//      certified (penalty, count) pair, moves the thresholds G / L (or the pilot's estimated interval) exactly as
//      search.cpp does when the host sequences the rounds, and applies the same stop rule;
//   2. decides the compaction the next round starts with (the penalty behind G, if it at least halves the level) and
//      carves the child level out of the problem's share of the pool;
//   3. picks the next penalties -- pilot: equally spaced in the estimated interval; first certified round: where the
//      pilot's estimate crosses a few multiples of the target; then between the thresholds, clustered geometrically
//      around the log-linear estimate of the crossing (the counts behind both thresholds are known), as many per
//      problem as keep the round within one wave of workgroups;
//   4. writes the round's task descriptors, the compaction descriptors and the launch sizes (LeanRoundCtl).
// The per-problem state (ChainHot) is moved into LDS as a whole when the kernel starts and back when it ends, and the
// loads of the round's results do not depend on it: a launch costs about one memory round trip, not one per decision.
// Steps 2-4 need sums over the problems (tickets, record offsets): thread 0 does them in LDS between two barriers.
#include "chain.h"

#include <cmath>

namespace rocco {

namespace {

constexpr int kDirectorWaves = 16;
constexpr int kDirectorThreads = 64 * kDirectorWaves;

__device__ __forceinline__ double snap_to_grid(double x, int qexp) { return ldexp(rint(ldexp(x, -qexp)), qexp); }

// ceil(log2(x)) of a finite x > 0, exactly
__device__ __forceinline__ int ceil_log2_exact(double x)
{
    const int e = ilogb(x);
    return (x == ldexp(1.0, e)) ? e : e + 1;
}

__device__ __forceinline__ unsigned long long align_up_dev(unsigned long long x, unsigned long long a) { return (x + a - 1) / a * a; }

// the problem's share of the level pool is a stack (lean_alloc in budget.hip)
__device__ __forceinline__ char *pool_alloc(const ChainArgs &A, ChainHot &H, unsigned long long bytes)
{
    const unsigned long long at = align_up_dev(H.pool_at, 256);
    if (at + bytes > H.pool_end) {
        return nullptr;
    }
    H.pool_at = at + bytes;
    return A.pool + at;
}

// penalty at which the pilot's estimated count crosses `want` (log-linear between its samples, which are sorted by
// penalty); false: no sample pair brackets it.  Called by a whole wavefront: lane i looks at the pair (i, i + 1).
__device__ bool pilot_crossing(const ChainPilot &W, int n_pilot, double want, int lane, double *x_out)
{
    for (int base = 0; base + 1 < n_pilot; base += 64) {
        const int i = base + lane;
        const bool valid = i + 1 < n_pilot;
        const double x0 = valid ? W.x[i] : 0.0, x1 = valid ? W.x[i + 1] : 0.0;
        const double c0 = valid ? W.c[i] : 0.0, c1 = valid ? W.c[i + 1] : 0.0;
        const unsigned long long hit = __ballot(valid && c0 >= want && c1 < want);
        if (hit != 0ull) {
            const double l0 = log(fmax(c0, 0.5)), l1 = log(fmax(c1, 0.5));
            const double f = (l0 > l1) ? (l0 - log(want)) / (l0 - l1) : 0.5;
            *x_out = __shfl(x0 + f * (x1 - x0), __builtin_ctzll(hit));
            return true;
        }
    }
    return false;
}

// statistics -> grid, epsilon, whether the threshold search applies (search.cpp: fast_path_applicable, bound_epsilon,
// the initial state of calibrate_batch).  One lane.
__device__ void init_problem(const ChainArgs &A, int b, ChainHot &H)
{
    const ChainInput in = A.inputs[b];
    const double smin = A.stats[5 * b + 0], smax = A.stats[5 * b + 1], sabs_sum = A.stats[5 * b + 4];
    H.scores = in.scores;
    H.n = in.n;
    H.gamma = in.gamma;
    H.target = in.target;
    H.pool_end = in.pool_end;
    H.pool_at = in.pool_begin;
    H.smin = smin;
    H.smax = smax;
    H.sabs_sum = sabs_sum;
    H.eps = 0.0;
    H.pg = H.pl = 0.0;
    H.pilot_scale = 1.0;
    H.est_pg = (double)in.n;
    H.est_pl = 0.0;
    H.G = H.L = 0.0;
    H.cG = H.cL = 0;
    H.open_before = 0x7FFFFFFFFFFFFFFFLL;
    H.soft_hi = H.soft_span = H.pilot_res = H.soft_count = 0.0;
    H.lv_s = in.scores;
    H.lv_orig = nullptr;
    H.lv_m = in.n;
    H.lv_bits = nullptr;
    H.lv_tile_off = nullptr;
    H.lv_cap = 0;
    H.n_levels = 0;
    H.qexp = 0;
    H.searching = 0;
    H.done = 0;
    H.rounds = 0;
    H.pilots = 0;
    H.n_evals = 0;
    H.phase = 0;
    H.pilot_left = 0;
    H.pilot_hint = 0;
    H.n_pilot = 0;
    H.G_real = H.L_real = 0;
    H.kind = 0;
    H.np = 0;
    H.have_soft = 0;
    H.can_pilot = in.can_pilot;
    H.pilot_ready = 0;
    H.pad = 0;
    if (in.allowed == 0 || in.n < 2 || !(in.target < in.n) || in.target < 0) {
        return;
    }
    const double c = in.gamma;
    const double sabs = fmax(fabs(smin), fabs(smax));
    if (!(c >= 1.0e-3) || !(c <= 1.0e6) || !isfinite(smin) || !isfinite(smax) || !(smax - smin <= 1.0e9) || !(sabs <= 1.0e12)) {
        return;
    }
    const double lambda = smin - 1.0;
    const double n = (double)in.n;
    const double sum_abs = sabs_sum * (1.0 + 1e-9) + 1.0;
    const double pos = sum_abs + n * fmax(0.0, -lambda);
    const double pb = 2.0 * (pos + n * 0.0625 + fmax(c, 0.0) + sabs + fabs(lambda) + 1.0);
    if (!(pb > 0.0) || !isfinite(pb)) {
        return;
    }
    const int e = ilogb(pb);
    const double r = fmax(c, 0.0) + (smax - smin) + 2.0;
    const int qexp = ceil_log2_exact(8.0 * r) - 52;
    const double hb = ldexp(1.0, e + 2 - 53);
    const double q = ldexp(1.0, qexp);
    if (!(hb >= q)) {
        return;
    }
    H.qexp = qexp;
    H.eps = 16.0 * hb + 4.0 * q;
    H.searching = 1;
    H.G = smin - 1.0;
    H.L = smax + 1.0;
    H.cG = in.n;
    H.cL = 0;
    H.pg = H.G;
    H.pl = H.L;
    // level 0: the caller's array
    H.n_levels = 1;
    ChainLevelReport l0;
    l0.s = in.scores;
    l0.orig = nullptr;
    l0.m = in.n;
    l0.base = -INFINITY;
    l0.sep = 0.0;
    l0.pool_mark = in.pool_begin;
    l0.bits = nullptr;
    l0.tile_off = nullptr;
    l0.cap_points = 0;
    l0.pad = 0;
    A.probs[b].levels[0] = l0;
    A.probs[b].n_above = 0;
    A.probs[b].n_below = 0;
    H.phase = 2;
    if (A.tune.pilot_rounds > 0 && in.target > 0 && in.can_pilot != 0 && (H.pl - H.pg > 64.0 * H.eps)) {
        H.phase = 1;
        H.pilot_left = A.tune.pilot_rounds;
    }
}

__device__ __forceinline__ void give_up(ChainHot &H)
{
    H.done = 2;
    H.phase = 3;
}

// Results of the round that just ended (one wavefront per problem, lane = penalty of the round), and -- certified
// rounds -- the compaction the next round starts with: at the penalty behind G, if it was evaluated in this round and
// its selected loci (with separators) are at most half of the level.  Returns the workgroup slots of that compaction.
__device__ int consume(const ChainArgs &A, int b, ChainHot &H, double x, long long count, long long child, int lane,
                       LeanCompactTask *ct, bool plan_next)
{
    ChainProb &P = A.probs[b];
    const int np = H.np, kind = H.kind;
    const long long target = H.target;
    const double eps = H.eps;
    const bool on = lane < np;
    if (kind == 1) {
        // pilot: estimates only steer (search.cpp, "pilot_round").  The new samples lie inside one gap of the list, which
        // is kept sorted by penalty: make room there.
        ChainPilot &W = A.pilot[b];
        const int n_old = H.n_pilot;
        const double x_first = __shfl(x, 0);
        const double est = (double)llrint((double)count * H.pilot_scale);
        int at = 0;
        double keep_x[kChainMaxPilot / 64], keep_c[kChainMaxPilot / 64];
#pragma unroll
        for (int k = 0; k < kChainMaxPilot / 64; ++k) {
            const int i = k * 64 + lane;
            keep_x[k] = (i < n_old) ? W.x[i] : 0.0;
            keep_c[k] = (i < n_old) ? W.c[i] : 0.0;
            at += __builtin_popcountll(__ballot(i < n_old && keep_x[k] < x_first));
        }
        const int room = min(np, kChainMaxPilot - n_old);
#pragma unroll
        for (int k = 0; k < kChainMaxPilot / 64; ++k) {
            const int i = k * 64 + lane;
            if (i < n_old && i >= at) {
                W.x[i + room] = keep_x[k];
                W.c[i + room] = keep_c[k];
            }
        }
        if (lane < room) {
            W.x[at + lane] = x;
            W.c[at + lane] = est;
        }
        const bool more = on && est > (double)target;
        // the interval's ends: the highest penalty whose estimate exceeds the target, the lowest whose does not, and the
        // estimates there (penalties ascend with the lanes, estimates do not rise)
        const unsigned long long more_b = __ballot(more), less_b = __ballot(on && !more);
        const int i_up = more_b ? 63 - __builtin_clzll(more_b) : 0, i_down = less_b ? __builtin_ctzll(less_b) : 0;
        const double up = more_b ? __shfl(x, i_up) : -INFINITY, down = less_b ? __shfl(x, i_down) : INFINITY;
        const double est_up = __shfl(est, i_up), est_down = __shfl(est, i_down);
        if (lane == 0) {
            ++H.pilots;
            H.n_pilot = n_old + room;
            if (up > H.pg) {
                H.pg = up;
                H.est_pg = est_up;
            }
            if (down < H.pl) {
                H.pl = down;
                H.est_pl = est_down;
            }
            const double pg = H.pg, pl = H.pl;
            const int left = H.pilot_left - 1;
            H.pilot_left = left;
            // Enough when the estimates at the two ends are within a factor of 2.5 of each other: the first certified
            // round takes its penalties from a log-linear interpolation between neighbouring samples, which is only as good
            // as the estimate is smooth between them (on tracks with a noise floor the count falls by orders of magnitude
            // within a hundredth of the range).  Or when the interval is as narrow as epsilon allows, or the rounds are
            // used up.
            const bool smooth = H.est_pg <= 2.5 * fmax(H.est_pl, 1.0);
            // (the pilots of a batch end together -- see the director: the pass over every locus that follows is one launch
            // for all of them, not one per group of stragglers)
            H.pilot_ready = (left <= 0 || !(pl - pg > 64.0 * eps) || smooth) ? 1 : 0;
        }
        return 0;
    }
    // certified counts on the deepest level: records, thresholds (search.cpp, "bound_round"), stop rule
    const int n_evals = H.n_evals;
    if (on && n_evals + lane < kChainMaxEvals) {
        A.evals[b].x[n_evals + lane] = x;
        A.evals[b].c[n_evals + lane] = count;
    }
    // penalties ascend and counts do not rise: the thresholds move to the last "more than the target" and the first
    // "at most the target" of the round
    const unsigned long long more = __ballot(on && count > target), less = __ballot(on && count <= target);
    const int i_more = more ? 63 - __builtin_clzll(more) : 0, i_less = less ? __builtin_ctzll(less) : 0;
    const double x_more = __shfl(x, i_more), x_less = __shfl(x, i_less);
    const long long c_more = __shfl(count, i_more), c_less = __shfl(count, i_less);
    const long long child_more = __shfl(child, i_more);
    // the report keeps the last kChainKeep counts on each side (see ChainProb): this round's highest penalties above
    // the target behind what is there, its lowest ones at or below it likewise
    {
        const int n_more = __builtin_popcountll(more), n_less = __builtin_popcountll(less);
        const int take_more = min(n_more, kChainKeep), take_less = min(n_less, kChainKeep);
        const int old_above = P.n_above, old_below = P.n_below;
        const int keep_above = min(old_above, kChainKeep - take_more), keep_below = min(old_below, kChainKeep - take_less);
        // survivors of the old entries move to the front (the newest are at the end)
        double ax = 0.0, bx = 0.0;
        long long ac = 0, bc = 0;
        if (lane < keep_above) {
            ax = P.above_x[old_above - keep_above + lane];
            ac = P.above_c[old_above - keep_above + lane];
        }
        if (lane < keep_below) {
            bx = P.below_x[old_below - keep_below + lane];
            bc = P.below_c[old_below - keep_below + lane];
        }
        if (lane < keep_above) {
            P.above_x[lane] = ax;
            P.above_c[lane] = ac;
        }
        if (lane < keep_below) {
            P.below_x[lane] = bx;
            P.below_c[lane] = bc;
        }
        // above the target: the round's last take_more such lanes, ascending; at or below: its first take_less lanes,
        // stored so that the tightest (lowest penalty) comes last
        const int rank_more = __builtin_popcountll(more & ((1ull << lane) - 1ull));  // my index among the lanes above
        if (((more >> lane) & 1ull) && rank_more >= n_more - take_more) {
            const int slot = keep_above + (rank_more - (n_more - take_more));
            P.above_x[slot] = x;
            P.above_c[slot] = count;
        }
        const int rank_less = __builtin_popcountll(less & ((1ull << lane) - 1ull));
        if (((less >> lane) & 1ull) && rank_less < take_less) {
            const int slot = keep_below + (take_less - 1 - rank_less);
            P.below_x[slot] = x;
            P.below_c[slot] = count;
        }
        if (lane == 0) {
            P.n_above = keep_above + take_more;
            P.n_below = keep_below + take_less;
        }
    }
    int pre_tiles = 0;
    if (lane == 0) {
        ++H.rounds;
        H.n_evals = min(kChainMaxEvals, n_evals + np);
        bool moved = false;
        if (more != 0ull && (!H.G_real || x_more - eps > H.G)) {
            H.G = x_more - eps;
            H.cG = c_more;
            H.G_real = 1;
            moved = true;
        }
        if (less != 0ull && (!H.L_real || x_less + eps < H.L)) {
            H.L = x_less + eps;
            H.cL = c_less;
            H.L_real = 1;
        }
        if (!H.G_real && H.n_pilot > 0 && H.n_levels == 1 && H.rounds <= 2 && H.can_pilot != 0 && x_less > H.pg + 64.0 * eps) {
            // the pilot's hints all fell at or below the target (its estimate was too coarse where the count falls
            // steeply): two more pilot rounds below the lowest certified penalty, then new hints
            H.pl = fmin(H.pl, x_less);
            H.est_pl = fmin(H.est_pl, (double)c_less);
            H.phase = 1;
            H.pilot_left = 2;
            H.pilot_ready = 0;
        }
        if (n_evals + np + kLeanMaxPoints > kChainMaxEvals) {
            give_up(H);  // (no room for another round's records: the host goes on)
        } else {
            const long long blocks = H.n / 8192 + 1;
            const long long open_now = (H.G_real && H.L_real) ? (H.cG - H.cL) : H.n;
            const bool stalled = open_now >= H.open_before && (double)open_now <= A.tune.survey_gate * (double)blocks;
            H.open_before = open_now;
            const bool few_left = stalled || (double)open_now <= A.tune.search_gate * (double)blocks;
            if (few_left || H.L - H.G <= 8.0 * eps || H.rounds >= 48) {
                H.done = 1;
                H.phase = 3;
            }
        }
        const long long cl = child_more;
        // (only while another round follows: a level that is registered must also be built)
        if (plan_next && H.phase == 2 && moved && H.n_levels < kChainMaxLevels && cl >= 1 && 2 * cl <= H.lv_m && H.lv_bits != nullptr) {
            const unsigned long long mark = H.pool_at;
            double *cs = (double *)pool_alloc(A, H, (unsigned long long)cl * sizeof(double));
            int *co = (int *)pool_alloc(A, H, (unsigned long long)cl * sizeof(int));
            if (cs == nullptr || co == nullptr) {
                H.pool_at = mark;
            } else {
                const int nt = (int)((H.lv_m + kLeanTile - 1) / kLeanTile);
                ct->s = H.lv_s;
                ct->orig = H.lv_orig;
                ct->m = H.lv_m;
                ct->n_tiles = nt;
                ct->block_begin = 0;
                ct->bits = H.lv_bits + (long long)i_more * nt * kLeanThreads;
                ct->tile_off = H.lv_tile_off + (long long)i_more * nt;
                ct->sep = floor(x_more - 2.0 * H.gamma - 2.0);
                ct->out_s = cs;
                ct->out_orig = co;
                ct->capacity = cl;
                ChainLevelReport child_lv;
                child_lv.s = cs;
                child_lv.orig = co;
                child_lv.m = cl;
                child_lv.base = x_more;
                child_lv.sep = ct->sep;
                child_lv.pool_mark = mark;
                child_lv.bits = nullptr;
                child_lv.tile_off = nullptr;
                child_lv.cap_points = 0;
                child_lv.pad = 0;
                P.levels[H.n_levels] = child_lv;
                ++H.n_levels;
                H.lv_s = cs;
                H.lv_orig = co;
                H.lv_m = cl;
                H.lv_bits = nullptr;
                H.lv_tile_off = nullptr;
                H.lv_cap = 0;
                pre_tiles = nt;
            }
        }
    }
    return pre_tiles;
}

// penalties of the next certified round into out[] (ascending, on the grid, strictly between the thresholds); a whole
// wavefront, lane = candidate
__device__ int plan_points(const ChainArgs &A, int b, ChainHot &H, int want, double *out, int lane)
{
    const double G = H.G, L = H.L, eps = H.eps;
    const int qexp = H.qexp;
    const double target = (double)H.target;
    if (H.pilot_hint != 0) {
        const ChainPilot &W = A.pilot[b];
        const int n_pilot = H.n_pilot;
        // where the pilot's estimate crosses a few multiples of the target (lane k: multiple k)
        double mine = INFINITY;
        for (int k = 0; k < A.tune.n_mults; ++k) {
            double x = 0.0;
            if (pilot_crossing(W, n_pilot, A.tune.mults[k] * target, lane, &x)) {
                x = snap_to_grid(x, qexp);
                if (lane == k && x - eps > G && x + eps < L) {
                    mine = x;
                }
            }
        }
        // ascending, duplicates dropped: a candidate's slot is the number of smaller ones that are kept
        bool dup = false;
        for (int k = 0; k < kChainMaxMults; ++k) {
            const double other = __shfl(mine, k);
            dup = dup || (k < lane && other == mine);
        }
        const double kept = (lane < kChainMaxMults && !dup) ? mine : INFINITY;
        int rank = 0;
        for (int k = 0; k < kChainMaxMults; ++k) {
            rank += (__shfl(kept, k) < kept) ? 1 : 0;
        }
        const bool keep = kept != INFINITY;
        const int np = __builtin_popcountll(__ballot(keep));
        if (keep) {
            out[rank] = kept;
        }
        // how far above G the search reaches while no evaluation has certified an upper threshold: to where the
        // pilot saw somewhat less than the target
        double xh = 0.0;
        const bool have = pilot_crossing(W, n_pilot, A.tune.soft_mult * target, lane, &xh);
        if (lane == 0) {
            H.pilot_hint = 0;
            H.pilot_res = H.pl - H.pg;
            H.have_soft = have ? 1 : 0;
            H.soft_hi = xh;
            H.soft_count = A.tune.soft_mult * target;
            H.soft_span = 0.0;
        }
        if (np > 0) {
            return np;
        }
    }
    // (H is in LDS: what lane 0 wrote above is what every lane reads here -- same wavefront, program order)
    double lo = G, hi = L;
    bool soft = false;
    if (H.L_real == 0 && H.G_real != 0 && H.have_soft != 0) {
        const double span = H.soft_span;
        double reach = (span > 0.0) ? 4.0 * span : (H.soft_hi - G);
        reach = fmax(reach, fmax(64.0 * eps, 2.0 * H.pilot_res));
        if (G + reach + 2.0 * eps < L) {
            hi = G + reach;
            soft = true;
            if (lane == 0) {
                H.soft_span = reach;
            }
        }
    }
    const double width = hi - lo;
    // Equally spaced penalties narrow the interval by their number + 1 whatever the counts do.  When the counts behind
    // both ends are known, every other penalty goes near the linear estimate of the crossing instead, at distances in
    // geometric progression: where the count is locally smooth the interval shrinks by far more, where it is not
    // (plateaus, cliffs) the equally spaced half still does its part.
    const bool cluster = A.tune.interpolate != 0 && !soft && H.G_real != 0 && H.L_real != 0 && want >= 8 &&
                         width * A.tune.spread > 16.0 * eps;
    const int n_even = cluster ? want - want / 2 : want;
    double u;
    if (lane < n_even) {
        u = soft ? (double)(lane + 1) / (double)n_even : (double)(lane + 1) / (double)(n_even + 1);
    } else {
        const double cg = (double)H.cG, cl = (double)H.cL;
        double uc = (cg > cl) ? (cg - target) / (cg - cl) : 0.5;
        uc = fmin(fmax(uc, 0.02), 0.98);
        const int m = want - n_even;        // clustered penalties: half below the estimate, half above
        const int j = lane - n_even;
        const int half = m / 2;
        const bool below = j < half;
        const int k = below ? (half - 1 - j) : (j - half);  // 0: innermost
        const int top = below ? (half - 1) : (m - half - 1);
        const double frac = (top > 0) ? (double)k / (double)top : 0.0;
        const double sigma = exp(log(A.tune.spread) * (1.0 - frac)) * 0.5;  // spread / 2 ... 1 / 2 of the way to the end
        u = below ? uc - uc * sigma : uc + (1.0 - uc) * sigma;
    }
    double x = (lane < want) ? snap_to_grid(lo + u * width, qexp) : INFINITY;
    if (cluster) {
        // ascending order over the wavefront: a lane's slot is the number of smaller penalties (ties: lower lane first)
        int rank = 0;
        for (int k = 0; k < want; ++k) {
            const double other = __shfl(x, k);
            rank += (other < x || (other == x && k < lane)) ? 1 : 0;
        }
        // lane r takes the penalty ranked r
        double sorted = INFINITY;
        for (int k = 0; k < want; ++k) {
            const double other = __shfl(x, k);
            const int other_rank = __shfl(rank, k);
            sorted = (other_rank == lane) ? other : sorted;
        }
        x = sorted;
    }
    const double before = __shfl_up(x, 1);
    const bool ok = lane < want && x - eps > G && x + eps < L && (lane == 0 || x > before);
    const unsigned long long mask = __ballot(ok);
    if (ok) {
        out[__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = x;
    }
    return __builtin_popcountll(mask);
}

__global__ __launch_bounds__(kDirectorThreads) void chain_director_kernel(ChainArgs A, int round, int last)
{
    __shared__ ChainHot sh[kChainMaxProblems];
    __shared__ LeanCompactTask s_ct[kChainMaxProblems];
    __shared__ int s_nt[kChainMaxProblems], s_np[kChainMaxProblems], s_flex[kChainMaxProblems], s_pre[kChainMaxProblems];
    __shared__ int s_task[kChainMaxProblems], s_unit[kChainMaxProblems], s_rec[kChainMaxProblems], s_stride[kChainMaxProblems];
    __shared__ int s_pre_idx[kChainMaxProblems], s_pre_block[kChainMaxProblems];
    __shared__ int s_want, s_pilot_want, s_batch;
    lean_round_reset(A.reset, A.ctl);  // (what the round before left: tickets, error word, progress counters)
    if (round > 0 && A.ctl->all_done != 0) {
        return;  // every search has ended: the remaining launches of the chain find all sizes zero
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int B = A.n_problems;
    long long *trace = (A.trace != nullptr && threadIdx.x == 0) ? A.trace + 8 * round : nullptr;
    if (trace) trace[0] = (long long)wall_clock64();
    constexpr int kHotWords = (int)(sizeof(ChainHot) / 8);
    if (round > 0) {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(A.hot);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(sh);
        for (int i = threadIdx.x; i < B * kHotWords; i += kDirectorThreads) {
            dst[i] = src[i];
        }
    }
    for (int i = threadIdx.x; i < kChainMaxProblems; i += kDirectorThreads) {
        s_nt[i] = 0;
        s_np[i] = 0;
        s_flex[i] = 0;
        s_pre[i] = 0;
        s_stride[i] = 1;
    }
    __syncthreads();
    if (trace) trace[1] = (long long)wall_clock64();
    // ---- results of the last round; the next compaction; tiles of the level the next round evaluates ----
    for (int b = wave; b < B; b += kDirectorWaves) {
        ChainHot &H = sh[b];
        int pre_tiles = 0;
        if (round == 0) {
            if (lane == 0) {
                init_problem(A, b, H);
            }
        } else if (H.kind != 0) {
            // (the addresses of the round's results depend on nothing the state holds)
            const double rx = A.points[(long long)b * kLeanMaxPoints + lane];
            const LeanResult rr = A.results[(long long)b * kLeanMaxPoints + lane];
            pre_tiles = consume(A, b, H, rx, rr.count, rr.child_len, lane, &s_ct[b], last == 0);
        }
        if (lane == 0) {
            H.kind = 0;
            H.np = 0;
            const int phase = H.phase;
            if (!last && H.searching != 0 && (phase == 1 || phase == 2)) {
                if (phase == 1) {
                    // every stride-th tile of the caller's array, each a chain of its own (lean_enqueue: pilot)
                    const long long n = H.n;
                    const long long all_tiles = (n + kLeanTile - 1) / kLeanTile;
                    const int stride = (int)max(4LL, all_tiles / max(2, A.tune.pilot_tiles));
                    const int nt = (int)((all_tiles + stride - 1) / stride);
                    const long long last_at = (long long)(nt - 1) * stride * kLeanTile;
                    const long long sampled = (long long)(nt - 1) * kLeanTile + min((long long)kLeanTile, n - last_at);
                    H.pilot_scale = (double)n / (double)max(1LL, sampled);
                    s_nt[b] = nt;
                    s_stride[b] = stride;
                    s_flex[b] = 2;
                } else {
                    s_pre[b] = pre_tiles;
                    const int nt = (int)((H.lv_m + kLeanTile - 1) / kLeanTile);
                    s_nt[b] = nt;
                    s_flex[b] = (H.n_levels == 1 && nt >= 128) ? 0 : 1;
                }
            }
        }
    }
    __syncthreads();
    {
        // The pilots of a batch end together: while one of them still needs a round the others refine theirs (a round
        // costs the same with them in it), and the pass over every locus that follows is ONE launch.
        __shared__ int s_waiting;
        if (threadIdx.x == 0) {
            s_waiting = 0;
        }
        __syncthreads();
        for (int b = threadIdx.x; b < B; b += kDirectorThreads) {
            if (s_flex[b] == 2 && sh[b].pilot_ready == 0) {
                atomicOr(&s_waiting, 1);
            }
        }
        __syncthreads();
        if (s_waiting == 0) {
            for (int b = threadIdx.x; b < B; b += kDirectorThreads) {
                if (s_flex[b] == 2) {
                    ChainHot &H = sh[b];
                    H.pilot_left = 0;
                    H.pilot_hint = 1;
                    H.pilot_ready = 0;
                    H.phase = 2;
                    s_nt[b] = (int)((H.lv_m + kLeanTile - 1) / kLeanTile);
                    s_stride[b] = 1;
                    s_flex[b] = (H.n_levels == 1 && s_nt[b] >= 128) ? 0 : 1;
                }
            }
        }
        __syncthreads();
    }
    if (trace) trace[2] = (long long)wall_clock64();
    if (wave == 0) {
        // penalties per problem: as many groups of eight as keep the round within the workgroups it should fill
        long long tiles = 0;
        for (int k = lane; k < B; k += 64) {
            tiles += (s_flex[k] != 0) ? s_nt[k] : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tiles += __shfl_xor(tiles, off);
        }
        if (lane == 0) {
            const long long groups = max(1LL, min(8LL, (long long)A.tune.wgs / max(1LL, tiles)));
            const long long pilot_groups = max(1LL, min(8LL, (long long)A.tune.pilot_wgs / max(1LL, tiles)));
            s_want = (int)(8 * groups);
            s_pilot_want = (int)max(2LL, min((long long)kLeanMaxPoints, (long long)A.tune.pilot_points * pilot_groups));
        }
    }
    __syncthreads();
    // ---- the next round's penalties ----
    for (int b = wave; b < B; b += kDirectorWaves) {
        if (s_nt[b] <= 0) {
            continue;
        }
        ChainHot &H = sh[b];
        double *pts = A.points + (long long)b * kLeanMaxPoints;
        int np = 0;
        if (s_flex[b] == 2) {
            const int want = s_pilot_want;
            const double pg = H.pg, pl = H.pl;
            const double x = snap_to_grid(pg + (pl - pg) * (double)(lane + 1) / (double)(want + 1), H.qexp);
            const double before = __shfl_up(x, 1);
            const bool ok = lane < want && x > pg && x < pl && (lane == 0 || x > before);
            const unsigned long long mask = __ballot(ok);
            if (ok) {
                pts[__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = x;
            }
            np = __builtin_popcountll(mask);
            if (np == 0 && lane == 0 && H.pilot_ready == 0) {
                give_up(H);  // (a pilot that is ready and has no room left for samples sits the round out)
            }
        } else {
            const int want = (s_flex[b] == 1) ? s_want : max(1, min(8, A.tune.big_points));
            np = plan_points(A, b, H, want, pts, lane);
            if (lane == 0) {
                if (np == 0) {
                    // the thresholds are as close as epsilon allows
                    H.done = 1;
                    H.phase = 3;
                } else if (H.lv_cap < np) {
                    // storage of this level's evaluations
                    const int cap = max(np, (H.n_levels > 1) ? kLeanMaxPoints : 8);
                    const unsigned long long nt = (unsigned long long)s_nt[b];
                    unsigned *bits = (unsigned *)pool_alloc(A, H, (unsigned long long)cap * nt * kLeanThreads * sizeof(unsigned));
                    unsigned *tile_off = (unsigned *)pool_alloc(A, H, (unsigned long long)cap * nt * sizeof(unsigned));
                    if (bits == nullptr || tile_off == nullptr) {
                        np = 0;
                        give_up(H);
                    } else {
                        H.lv_bits = bits;
                        H.lv_tile_off = tile_off;
                        H.lv_cap = cap;
                        ChainLevelReport &lv = A.probs[b].levels[H.n_levels - 1];
                        lv.bits = bits;
                        lv.tile_off = tile_off;
                        lv.cap_points = cap;
                    }
                }
            }
            np = __shfl(np, 0);
        }
        if (lane == 0) {
            s_np[b] = np;
        }
    }
    __syncthreads();
    if (trace) trace[3] = (long long)wall_clock64();
    if (wave == 0) {
        // penalties per workgroup: fewer while the whole round still fits one wave (lean_enqueue: rebatch); then every
        // task's first ticket and first record, every compaction's first workgroup slot: scans over the problems, lane
        // k taking problems k and k + 64
        static_assert(kChainMaxProblems <= 128, "two problems per lane");
        const int k0 = lane, k1 = lane + 64;
        const int np0 = (k0 < B) ? s_np[k0] : 0, np1 = (k1 < B) ? s_np[k1] : 0;
        const int nt0 = (k0 < B) ? s_nt[k0] : 0, nt1 = (k1 < B) ? s_nt[k1] : 0;
        const int pre0 = (k0 < B) ? s_pre[k0] : 0, pre1 = (k1 < B) ? s_pre[k1] : 0;
        long long u2 = (long long)nt0 * ((max(np0, 0) + 1) / 2) + (long long)nt1 * ((max(np1, 0) + 1) / 2);
        long long u4 = (long long)nt0 * ((max(np0, 0) + 3) / 4) + (long long)nt1 * ((max(np1, 0) + 3) / 4);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            u2 += __shfl_xor(u2, off);
            u4 += __shfl_xor(u4, off);
        }
        const int batch = (u2 <= 512) ? 2 : ((u4 <= 512) ? 4 : kLeanBatch);
        // records first: a task whose records would not fit is dropped (its problem is left to the host), which must
        // not move the others' offsets -- so the scan runs over what fits, decided in problem order
        long long rec0 = (np0 > 0) ? (long long)nt0 * np0 : 0, rec1 = (np1 > 0) ? (long long)nt1 * np1 : 0;
        auto scan = [&](long long a, long long b_, long long &ex0, long long &ex1, long long &total) {
            // exclusive prefix over the order (lane 0's first, ..., lane 63's first, lane 0's second, ...)
            long long inc = a;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const long long v = __shfl_up(inc, off);
                inc += (lane >= off) ? v : 0;
            }
            const long long first_total = __shfl(inc, 63);
            ex0 = inc - a;
            long long inc2 = b_;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const long long v = __shfl_up(inc2, off);
                inc2 += (lane >= off) ? v : 0;
            }
            ex1 = first_total + inc2 - b_;
            total = first_total + __shfl(inc2, 63);
        };
        long long r_ex0, r_ex1, r_total;
        scan(rec0, rec1, r_ex0, r_ex1, r_total);
        const bool fit0 = np0 > 0 && r_ex0 + rec0 <= (long long)A.rec_capacity;
        const bool fit1 = np1 > 0 && r_ex1 + rec1 <= (long long)A.rec_capacity;
        // (a dropped task keeps its place in the record scan: the space stays unused, nothing overlaps)
        long long t_ex0, t_ex1, n_tasks, un_ex0, un_ex1, units, pr_ex0, pr_ex1, pairs, pi_ex0, pi_ex1, n_pre, pb_ex0, pb_ex1, pre_blocks;
        scan(fit0 ? 1 : 0, fit1 ? 1 : 0, t_ex0, t_ex1, n_tasks);
        scan(fit0 ? (long long)nt0 * ((np0 + batch - 1) / batch) : 0, fit1 ? (long long)nt1 * ((np1 + batch - 1) / batch) : 0, un_ex0, un_ex1, units);
        scan(fit0 ? np0 : 0, fit1 ? np1 : 0, pr_ex0, pr_ex1, pairs);
        scan(pre0 > 0 ? 1 : 0, pre1 > 0 ? 1 : 0, pi_ex0, pi_ex1, n_pre);
        scan(pre0, pre1, pb_ex0, pb_ex1, pre_blocks);
        if (k0 < B) {
            s_task[k0] = fit0 ? (int)t_ex0 : -1;
            s_unit[k0] = (int)un_ex0;
            s_rec[k0] = (int)r_ex0;
            s_pre_idx[k0] = (pre0 > 0) ? (int)pi_ex0 : -1;
            s_pre_block[k0] = (int)pb_ex0;
            if (np0 > 0 && !fit0) {
                s_np[k0] = -1;
            }
        }
        if (k1 < B) {
            s_task[k1] = fit1 ? (int)t_ex1 : -1;
            s_unit[k1] = (int)un_ex1;
            s_rec[k1] = (int)r_ex1;
            s_pre_idx[k1] = (pre1 > 0) ? (int)pi_ex1 : -1;
            s_pre_block[k1] = (int)pb_ex1;
            if (np1 > 0 && !fit1) {
                s_np[k1] = -1;
            }
        }
        if (lane == 0) {
            s_batch = batch;
            A.ctl->n_tasks = (int)n_tasks;
            A.ctl->n_units = (int)units;
            A.ctl->n_pairs = (int)pairs;
            A.ctl->n_pre_tasks = (int)n_pre;
            A.ctl->n_pre_blocks = (int)pre_blocks;
            A.ctl->round = round + 1;
            A.ctl->all_done = (n_tasks == 0 && n_pre == 0) ? 1 : 0;
        }
    }
    __syncthreads();
    if (trace) trace[4] = (long long)wall_clock64();
    // ---- descriptors, report ----
    for (int b = wave; b < B; b += kDirectorWaves) {
        if (lane != 0) {
            continue;
        }
        ChainHot &H = sh[b];
        if (s_np[b] < 0) {
            give_up(H);
        }
        if (s_pre_idx[b] >= 0) {
            LeanCompactTask ct = s_ct[b];
            ct.block_begin = s_pre_block[b];
            A.pre[s_pre_idx[b]] = ct;
        }
        if (s_task[b] >= 0) {
            const int np = s_np[b];
            const bool pilot = (s_flex[b] == 2);
            const int qexp = H.qexp;
            LeanTask t;
            t.s = pilot ? H.scores : H.lv_s;
            t.m = pilot ? H.n : H.lv_m;
            t.c_raw = H.gamma;
            t.magic = ldexp(1.5, 52 + qexp);
            t.big = ldexp(1.0, 50 + qexp);
            t.n_tiles = s_nt[b];
            t.n_points = np;
            t.n_groups = (np + s_batch - 1) / s_batch;
            t.unit_begin = s_unit[b];
            t.point_begin = b * kLeanMaxPoints;
            t.rec_begin = s_rec[b];
            t.bits_begin = pilot ? 0 : (long long)(H.lv_bits - (const unsigned *)A.pool);
            t.off_begin = pilot ? 0 : (long long)(H.lv_tile_off - (const unsigned *)A.pool);
            t.result_begin = b * kLeanMaxPoints;
            t.tile_stride = pilot ? s_stride[b] : 1;
            t.independent = pilot ? 1 : 0;
            t.store = pilot ? 0 : 1;
            t.emap = nullptr;
            t.wcap = nullptr;
            t.clean_chunks = nullptr;
            t.cmax = 0.0;
            t.sabs = 0.0;
            t.qexp = qexp;
            t.batch = s_batch;
            A.tasks[s_task[b]] = t;
            H.kind = pilot ? 1 : 2;
            H.np = np;
        }
        ChainProb &P = A.probs[b];
        P.smin = H.smin;
        P.smax = H.smax;
        P.sabs_sum = H.sabs_sum;
        P.eps = H.eps;
        P.qexp = H.qexp;
        P.searching = H.searching;
        P.done = H.done;
        P.rounds = H.rounds;
        P.pilots = H.pilots;
        P.n_evals = H.n_evals;
        P.n_levels = H.n_levels;
        P.pad = 0;
        P.pool_at = H.pool_at;
    }
    __syncthreads();
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(A.hot);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(sh);
        for (int i = threadIdx.x; i < B * kHotWords; i += kDirectorThreads) {
            dst[i] = src[i];
        }
    }
    if (trace) trace[5] = (long long)wall_clock64();
    // ---- the host follows the chain: hand it the report as soon as nothing is left to run ----
    if (A.follow != nullptr && (last != 0 || A.ctl->all_done != 0)) {
        __syncthreads();  // the report is complete in device memory (this workgroup wrote the last of it)
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(A.probs);
        unsigned long long *dst = A.follow + 32;
        for (int i = threadIdx.x; i < A.follow_words; i += kDirectorThreads) {
            dst[i] = src[i];
        }
        if (threadIdx.x < (int)(sizeof(LeanRoundCtl) / 8)) {
            dst[A.follow_ctl_word + threadIdx.x] = reinterpret_cast<const unsigned long long *>(A.ctl)[threadIdx.x];
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(A.follow, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace

int launch_chain_director(const ChainArgs &A, int round, int last, hipStream_t stream)
{
    hipLaunchKernelGGL(chain_director_kernel, dim3(1), dim3(kDirectorThreads), 0, stream, A, round, last);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
