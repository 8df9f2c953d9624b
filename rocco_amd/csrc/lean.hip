// rocco_amd/csrc/lean.hip -- see lean.h.  gfx950 only.
//
// lean_eval_kernel: one workgroup per (tile of 8192 loci, batch of <= 8 penalties), ONE launch per search
// round for every chromosome of the batch, the scores read once from HBM:
//   1. the tile is staged through LDS (coalesced 16-B loads, rounded to the grid q on the way; one 32-locus
//      chunk per LDS row, row stride 34 doubles -> conflict-free 16-B LDS reads);
//   2. every lane runs the two extreme trajectories of its chunk (from -c and from +c) for the penalties of
//      the batch interleaved in registers (independent FP64 chains hide each other's latency), which gives
//      the chunk's function x -> clamp(x + a, lo, hi); the functions are composed across the workgroup
//      (wavefront shuffles, then the four wavefront aggregates through LDS);
//   3. the tile's function is handed to the next tile through 8-byte self-flagging granules (a function
//      with lo == hi -- nearly every tile -- fixes the outgoing delta without waiting for anything; tiles
//      take tickets in launch order, so a tile only ever waits for one that is already running);
//   4. every lane reruns its chunk from its true incoming delta, collects the class bits with two
//      instructions per class and locus (sign bit of c - delta / of -delta - c shifted into a mask), and
//      closes the backward fill of its 32 loci with one 64-bit addition (the carry chain of NZ + ONE is the
//      fill recurrence z_j = ONE_j | (COPY_j & z_{j+1}));
//   5. per tile and penalty: the count for an incoming fill value of 0, the length of the run that copies
//      it, and the kept-locus bits (incoming fill value taken as 1: a superset).
// lean_finish_kernel closes the fill across tiles (one wavefront per chromosome and penalty) and lays
// out the compaction each penalty would give; lean_compact_kernel writes the chosen one.
// All of it is integer / exact FP64 min-max-add work bound by HBM reads of 8 B per locus and round (level
// 0) or by launch latency (deeper levels); no MFMA -- the path is BLAS-1.
#include "lean.h"
#include "model_chain.h"

#include <cmath>

namespace rocco {

namespace {

constexpr int kStride = kLeanChunk + 2;  // doubles per chunk row in LDS
constexpr int kTileLds = kLeanThreads * kStride;
constexpr unsigned long long kSentinel = ~0ull;  // granule not written yet (a NaN: never a value)
constexpr unsigned kSpinLimit = 1u << 22;

struct Fn {
    double a, lo, hi;  // x -> min(max(x + a, lo), hi)
};

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// f first, then g
__device__ __forceinline__ Fn compose(const Fn &f, const Fn &g)
{
    Fn r;
    r.a = f.a + g.a;
    r.lo = clampd(f.lo + g.a, g.lo, g.hi);
    r.hi = clampd(f.hi + g.a, g.lo, g.hi);
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

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void granule_store(unsigned long long *p, double v)
{
    __hip_atomic_store(p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bounded wait for a granule; gives up (and reports it) rather than hang the device
__device__ __forceinline__ double granule_wait(const unsigned long long *p, unsigned *error)
{
    unsigned long long v = granule_load(p);
    unsigned spins = 0;
    while (v == kSentinel) {
        __builtin_amdgcn_s_sleep(2);
        v = granule_load(p);
        if (++spins > kSpinLimit) {
            atomicOr(error, 1u);
            return 0.0;
        }
    }
    return __longlong_as_double((long long)v);
}

// ---- which task a ticket (workgroup slot, pair) belongs to ---------------------------------------------------
// The tasks of a launch are a short table in memory (one per chromosome and level); a workgroup finds its own from
// the first ticket of every task.  Walking the table costs one DEPENDENT memory round trip per task -- up to two
// dozen before a workgroup's first useful load, and in a round of a single wave of workgroups every workgroup is the
// first on its CU, with cold caches: that walk was most of the ~25 us such a round took beyond its work.  Here the
// lanes of a wavefront fetch the tasks' first tickets side by side (one round trip per 64 tasks) and a ballot picks
// the last one that does not exceed the key.  Every lane of the wavefront must call it with the same key.
template <typename First>
__device__ __forceinline__ int find_task(int n_tasks, int key, First first_of)
{
    const int lane = threadIdx.x & 63;
    int ti = 0;
    for (int base = 0; base < n_tasks; base += 64) {
        const int i = base + lane;
        const int v = (i < n_tasks) ? first_of(i) : 0x7FFFFFFF;
        const unsigned long long le = __ballot(v <= key);
        if (le == 0ull) {
            break;
        }
        ti = base + 63 - __builtin_clzll(le);
        if (le != ~0ull) {
            break;
        }
    }
    return ti;
}

// ---- tile staging: rn_q(score) into LDS --------------------------------------------------------
template <bool RAW>
__device__ __forceinline__ void stage_tile(const double *__restrict__ s, long long m, long long base, double magic,
                                           double *lds)
{
    const int t = threadIdx.x;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(s) & 15U) == 0);
    if (aligned16 && base + kLeanTile <= m) {
        // whole tile in range: sixteen unconditional 16-byte loads per lane, all in flight together
        const double2 *__restrict__ src = reinterpret_cast<const double2 *>(s + base) + t;
        double2 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = src[r * kLeanThreads];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = 2 * (r * kLeanThreads + t);
            double2 w;
            w.x = RAW ? v[r].x : (v[r].x + magic) - magic;
            w.y = RAW ? v[r].y : (v[r].y + magic) - magic;
            *reinterpret_cast<double2 *>(lds + (e >> 5) * kStride + (e & 31)) = w;
        }
    } else {
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int e = 2 * (r * kLeanThreads + t);
            const long long j = base + e;
            double2 v = make_double2(0.0, 0.0);
            if (j < m) v.x = s[j];
            if (j + 1 < m) v.y = s[j + 1];
            if (!RAW) {
                v.x = (v.x + magic) - magic;
                v.y = (v.y + magic) - magic;
            }
            *reinterpret_cast<double2 *>(lds + (e >> 5) * kStride + (e & 31)) = v;
        }
    }
    __syncthreads();
}

struct Scratch {
    double wave_fn[4][kLeanBatch][3];  // wavefront aggregates
    double tile_in[kLeanBatch];        // incoming delta of the tile
    unsigned red[kLeanBatch][4];       // base, tail, cells, (unused)
    unsigned char wave_pass[4][kLeanBatch], wave_v[4][kLeanBatch];
    unsigned first_word[4][kLeanBatch];  // kept-locus word of every wavefront's first lane
    unsigned last_word[kLeanBatch];      // ... and of the tile's last lane
    unsigned uncertain[kLeanBatch];      // model tasks: a class of the tile is not certified
    int ticket;
    int pad[3];
};

// One chunk, PB penalties: extreme trajectories + sum (the chunk's function).
template <int PB>
__device__ __forceinline__ void chunk_function(const double *__restrict__ row, const double (&x)[PB], double c,
                                               double c_first, double big, Fn (&f)[PB])
{
    double lo[PB], hi[PB], fa[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        fa[p] = 0.0;
        lo[p] = -big;
        hi[p] = big;
    }
#pragma unroll 1
    for (int i0 = 0; i0 < kLeanChunk; i0 += 8) {
        double rs[8];
#pragma unroll
        for (int ii = 0; ii < 8; ii += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(row + i0 + ii);
            rs[ii] = v.x;
            rs[ii + 1] = v.y;
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const double cc = (i0 + ii == 0) ? c_first : c;  // (clamp(+-big, -c, c) = +-c: the extremes start at the bounds)
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const double a = rs[ii] - x[p];
                fa[p] += a;
                lo[p] = fmin(fmax(lo[p], -cc), cc) + a;
                hi[p] = fmin(fmax(hi[p], -cc), cc) + a;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        f[p].a = fa[p];
        f[p].lo = lo[p];
        f[p].hi = hi[p];
    }
}

// One chunk, PB penalties, from the true incoming delta: class masks (locus i of the chunk at bit 31 - i).
// `valid` = loci of the chunk that exist (32 except in the last tile); `last` = index of the chain's last locus
// in this chunk (-1: not here).
template <int PB, bool EDGE>
__device__ __forceinline__ void chunk_classes(const double *__restrict__ row, const double (&x)[PB], double c,
                                              double c_first, const double (&din)[PB], int valid, int last,
                                              unsigned (&one)[PB], unsigned (&nz)[PB])
{
    double d[PB];
    const double nc = -c;
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        one[p] = 0u;
        nz[p] = 0u;
        d[p] = din[p];
    }
#pragma unroll 1
    for (int i0 = 0; i0 < kLeanChunk; i0 += 8) {
        double rs[8];
#pragma unroll
        for (int ii = 0; ii < 8; ii += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(row + i0 + ii);
            rs[ii] = v.x;
            rs[ii + 1] = v.y;
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            const double cc = (i == 0) ? c_first : c;
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const double a = rs[ii] - x[p];
                d[p] = fmin(fmax(d[p], -cc), cc) + a;
                // sign(c - d) = 1  <=>  d > c (ONE);  sign(-c - d) = 1  <=>  d > -c (not ZERO); x - x = +0 at equality
                double t1 = c - d[p];
                double t2 = nc - d[p];
                if (EDGE) {
                    if (i == last) {  // terminal rule: ONE iff delta > 0, else ZERO
                        t1 = 0.0 - d[p];
                        t2 = t1;
                    }
                    if (i >= valid) {  // padding: ZERO
                        t1 = 0.0;
                        t2 = 0.0;
                    }
                }
                one[p] = __builtin_amdgcn_alignbit(one[p], (unsigned)__double2hiint(t1), 31);
                nz[p] = __builtin_amdgcn_alignbit(nz[p], (unsigned)__double2hiint(t2), 31);
            }
        }
    }
}


// ---- rounding-model tasks: the reference's own arithmetic, chunk by chunk (oracle/delta_oracle.c) --------------
// A lane owns one chunk of 32 loci and therefore one arithmetic mode per penalty:
//   clean  (the chunk's running values stay inside one binade, grid u = 2^(e-52)):  a = rn_u(s) + rn_u(-lambda),
//          c = rn_u(gamma), no tolerance unless rn_u(s) is an exact half-way tie;
//   hazard (binade edge nearby, or a tie in rn_u(-lambda) / rn_u(gamma), or u < q):  a = rn_q(s - lambda), c = rn_q(gamma),
//          every step adds weight 4 hb + q to the tolerance, 9 hb + 2 q on top, hb = 2^(e+2-53).
// A class is certified when its distance from the decision boundary exceeds the tolerance accumulated since the last
// clear clamp (|delta| - c > 2^-16: reference and model clamp to the same bound there, whatever came before).  This
// kernel certifies conservatively -- a wavefront's first lane inherits `wcap`, the sum of ALL hazard weights of the
// array, and a tie flags the whole tile -- and reports per penalty whether every class was certified; a penalty
// that is not goes through the full kernels of chain_fast.hip.
constexpr double kModelGuard = 0x1p-16;  // == ORACLE_GUARD
constexpr int kModelMapBias = 64;        // == ORACLE_MAP_BIAS

// the exponent e whose grid u = 2^(e-52) has x exactly half-way between two of its points: the lowest set bit of x is
// 2^(e-53).  (x == 0: none, returns a value outside every map code.)
__device__ __forceinline__ int tie_exponent(double x)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x) & 0x7FFFFFFFFFFFFFFFull;
    if (bits == 0ull) {
        return 1 << 20;
    }
    const int ex = (int)(bits >> 52);
    const unsigned long long frac = bits & 0xFFFFFFFFFFFFFull;
    if (ex == 0) {  // subnormal: value = frac * 2^-1074
        return -1074 + __builtin_ctzll(frac) + 53;
    }
    const unsigned long long sig = frac | (1ull << 52);  // value = sig * 2^(ex - 1075)
    return (ex - 1075) + __builtin_ctzll(sig) + 53;
}

template <int PB>
struct LaneModel {
    double magic_u, half_u, magic_q;
    double nlam[PB];  // rn_u(-lambda)
    double c[PB];     // cost on the lane's grid
    double w[PB];     // weight per step (0: clean)
    double base[PB];  // hazard: 9 hb + 2 q
    bool hz[PB];
    bool tie[PB];     // hazard only because -lambda or gamma rounds as a tie on u: not covered by `wcap`, never certified
};

template <int PB>
__device__ __forceinline__ void lane_model(const LeanTask &task, int code, const double (&x)[PB], LaneModel<PB> &lm)
{
    int e = (code & 0x7F) - kModelMapBias;
    const bool code_hz = (code & 0x80) != 0;
    const double q = ldexp(1.0, task.qexp);
    lm.magic_q = task.magic;
    // (the floor of a hazard chunk's exponent depends on |lambda|: take the largest of this workgroup's penalties)
    double lam_abs = 0.0;
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        lam_abs = fmax(lam_abs, fabs(x[p]));
    }
    if (code_hz) {
        e = max(e, ilogb(2.0 * task.cmax + 2.0 * task.sabs + lam_abs + 2.0));
    }
    lm.magic_u = ldexp(1.5, e);
    lm.half_u = ldexp(1.0, e - 53);
    const double hb = ldexp(1.0, e + 2 - 53);
    const double cu = (task.c_raw + lm.magic_u) - lm.magic_u;
    const double cq = (task.c_raw + lm.magic_q) - lm.magic_q;
    const bool gamma_tie = fabs(task.c_raw - cu) == lm.half_u;
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        lm.nlam[p] = (-x[p] + lm.magic_u) - lm.magic_u;
        const bool coded = code_hz || (e - 52 < task.qexp);
        const bool hz = coded || (fabs(-x[p] - lm.nlam[p]) == lm.half_u) || gamma_tie;
        lm.hz[p] = hz;
        lm.tie[p] = hz && !coded;
        lm.c[p] = hz ? cq : cu;
        lm.w[p] = hz ? (4.0 * hb + q) : 0.0;
        lm.base[p] = hz ? (9.0 * hb + 2.0 * q) : 0.0;
    }
}

template <int PB>
__device__ __forceinline__ void chunk_function_model(const double *__restrict__ row, const double (&x)[PB],
                                                     const LaneModel<PB> &lm, bool first_unclamped, double big, Fn (&f)[PB])
{
    double lo[PB], hi[PB], fa[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        fa[p] = 0.0;
        lo[p] = -big;
        hi[p] = big;
    }
#pragma unroll 1
    for (int i0 = 0; i0 < kLeanChunk; i0 += 8) {
        double rs[8];
#pragma unroll
        for (int ii = 0; ii < 8; ii += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(row + i0 + ii);
            rs[ii] = v.x;
            rs[ii + 1] = v.y;
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const double su = (rs[ii] + lm.magic_u) - lm.magic_u;
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const double ah = ((rs[ii] - x[p]) + lm.magic_q) - lm.magic_q;
                const double a = lm.hz[p] ? ah : (su + lm.nlam[p]);
                const double cc = (first_unclamped && i0 + ii == 0) ? big : lm.c[p];
                fa[p] += a;
                lo[p] = fmin(fmax(lo[p], -cc), cc) + a;
                hi[p] = fmin(fmax(hi[p], -cc), cc) + a;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        f[p].a = fa[p];
        f[p].lo = lo[p];
        f[p].hi = hi[p];
    }
}

// what a lane reports about its tolerances: see eval_body
template <int PB>
struct LaneTolerance {
    bool clear[PB];          // the lane holds a clear clamp
    double tail[PB];         // weight after its last clear clamp (its whole weight if it has none)
    double slack_all[PB];    // min over the loci up to the first clear clamp of (margin - local tolerance)
    double slack_pos[PB];    // ... over those of them whose local tolerance is positive
    double tau_pre[PB];      // largest local tolerance among them
    unsigned bad[PB];        // 1: a class after the first clear clamp is not certified;
                             // 32: the lane is hazard because of a tie of -lambda / gamma; 64: tolerance beyond the guard
};

template <int PB, bool EDGE>
__device__ __forceinline__ void chunk_classes_model(const double *__restrict__ row, const double (&x)[PB],
                                                    const LaneModel<PB> &lm, bool first_unclamped, double big,
                                                    const double (&din)[PB], int valid, int last, unsigned (&one)[PB],
                                                    unsigned (&nz)[PB], LaneTolerance<PB> &tol)
{
    double d[PB], wacc[PB];
    bool pre[PB];
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        one[p] = 0u;
        nz[p] = 0u;
        d[p] = din[p];
        wacc[p] = 0.0;
        pre[p] = true;
        tol.slack_all[p] = inf;
        tol.slack_pos[p] = inf;
        tol.tau_pre[p] = 0.0;
        tol.bad[p] = 0u;
    }
#pragma unroll 1
    for (int i0 = 0; i0 < kLeanChunk; i0 += 8) {
        double rs[8];
#pragma unroll
        for (int ii = 0; ii < 8; ii += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(row + i0 + ii);
            rs[ii] = v.x;
            rs[ii + 1] = v.y;
        }
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            const double su = (rs[ii] + lm.magic_u) - lm.magic_u;
            const bool exists = !EDGE || i < valid;
            const bool su_tie = exists && (fabs(rs[ii] - su) == lm.half_u);
            const bool terminal = EDGE && (i == last);
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const double ah = ((rs[ii] - x[p]) + lm.magic_q) - lm.magic_q;
                const double a = lm.hz[p] ? ah : (su + lm.nlam[p]);
                const double c = lm.c[p];
                const double cc = (first_unclamped && i == 0) ? big : c;
                d[p] = fmin(fmax(d[p], -cc), cc) + a;
                double t1 = c - d[p];
                double t2 = -c - d[p];
                if (EDGE) {
                    if (terminal) {
                        t1 = 0.0 - d[p];
                        t2 = t1;
                    }
                    if (i >= valid) {
                        t1 = 0.0;
                        t2 = 0.0;
                    }
                }
                one[p] = __builtin_amdgcn_alignbit(one[p], (unsigned)__double2hiint(t1), 31);
                nz[p] = __builtin_amdgcn_alignbit(nz[p], (unsigned)__double2hiint(t2), 31);
                // tolerance of this class (a clean step whose rn_u(s) is an exact half-way tie weighs u)
                const double over = fabs(d[p]) - c;                          // > 0: beyond a clamp bound
                const double margin = terminal ? fabs(d[p]) : fabs(over);   // distance from the decision boundary
                if (exists) {
                    wacc[p] += lm.hz[p] ? lm.w[p] : (su_tie ? 2.0 * lm.half_u : 0.0);
                    const double tau = wacc[p] + lm.base[p];
                    if (pre[p]) {
                        const double slack = margin - tau;
                        tol.slack_all[p] = fmin(tol.slack_all[p], slack);
                        tol.slack_pos[p] = (tau > 0.0) ? fmin(tol.slack_pos[p], slack) : tol.slack_pos[p];
                        tol.tau_pre[p] = fmax(tol.tau_pre[p], tau);
                    } else {
                        tol.bad[p] |= (tau > kModelGuard) ? 64u : 0u;
                        tol.bad[p] |= (tau > 0.0 && !(margin > tau)) ? 1u : 0u;
                    }
                    if (!terminal && over > kModelGuard) {
                        wacc[p] = 0.0;
                        pre[p] = false;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        tol.clear[p] = !pre[p];
        tol.tail[p] = wacc[p];
        (void)valid;
    }
}

constexpr int kZeroBytes = 16384;  // per workgroup of the zero launch

__global__ __launch_bounds__(256) void lean_zero_batch_kernel(const LeanScatterTask *__restrict__ tasks, int n_tasks)
{
    const int ti = find_task(n_tasks, (int)blockIdx.x, [&](int i) { return tasks[i].zero_begin; });
    const LeanScatterTask task = tasks[ti];
    const long long at = (long long)((int)blockIdx.x - task.zero_begin) * kZeroBytes;
    uint8_t *p = task.full + at;
    const long long len = min((long long)kZeroBytes, task.n - at);
    if (len == kZeroBytes && ((reinterpret_cast<uintptr_t>(p) & 15U) == 0)) {
        uint4 *q = reinterpret_cast<uint4 *>(p);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int r = 0; r < kZeroBytes / 16 / 256; ++r) {
            q[r * 256 + threadIdx.x] = z;
        }
    } else {
        for (long long i = threadIdx.x; i < len; i += 256) {
            p[i] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void lean_scatter_batch_kernel(const LeanScatterTask *__restrict__ tasks, int n_tasks)
{
    const int ti = find_task(n_tasks, (int)blockIdx.x, [&](int i) { return tasks[i].scatter_begin; });
    const LeanScatterTask task = tasks[ti];
    const long long i = (long long)((int)blockIdx.x - task.scatter_begin) * 256 + threadIdx.x;
    if (i < task.m) {
        const int o = task.orig[i];
        if (o >= 0) {
            task.full[o] = task.level_solution[i];
        }
    }
}

template <int PB, bool MODEL>
__device__ __forceinline__ void eval_body(const LeanLaunch &L, const LeanTask &task, int tile, int p0, int np, double *lds,
                                          Scratch *sc)
{
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const long long base = (long long)tile * task.tile_stride * kLeanTile;
    const bool indep = task.independent != 0;

    const double c = (task.c_raw + task.magic) - task.magic;
    const double big = task.big;
    double x[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        x[p] = L.points[task.point_begin + p0 + min(p, np - 1)];
    }
    const long long j0 = base + (long long)t * kLeanChunk;
    // the chain's first locus takes its input unclamped (pilot tasks: every tile is a chain of its own)
    const bool first_unclamped = (j0 == 0 || (indep && t == 0));
    const double c_first = first_unclamped ? big : c;
    const double *row = lds + t * kStride;
    LaneModel<PB> lm;
    if (MODEL) {
        const long long chunk = j0 / kLeanChunk;
        const long long last_chunk = (task.m - 1) / kLeanChunk;
        lane_model<PB>(task, (int)task.emap[min(chunk, last_chunk)], x, lm);
    }

    // ---- 2. chunk functions, composed across the workgroup ----
    Fn f[PB];
    if (MODEL) {
        chunk_function_model<PB>(row, x, lm, first_unclamped, big, f);
    } else {
        chunk_function<PB>(row, x, c, c_first, big, f);
    }
    Fn inc[PB];  // inclusive within the wavefront
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        inc[p] = f[p];
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const Fn prev = shfl_up_fn(inc[p], off);
            if (lane >= off) {
                inc[p] = compose(prev, inc[p]);
            }
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            sc->wave_fn[wave][p][0] = inc[p].a;
            sc->wave_fn[wave][p][1] = inc[p].lo;
            sc->wave_fn[wave][p][2] = inc[p].hi;
        }
    }
    __syncthreads();

    // ---- 3. hand-off between tiles: one lane per penalty ----
    if (t < np) {
        const int p = t;
        Fn agg;
        agg.a = sc->wave_fn[0][p][0];
        agg.lo = sc->wave_fn[0][p][1];
        agg.hi = sc->wave_fn[0][p][2];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            Fn g;
            g.a = sc->wave_fn[w][p][0];
            g.lo = sc->wave_fn[w][p][1];
            g.hi = sc->wave_fn[w][p][2];
            agg = compose(agg, g);
        }
        const long long rec = (long long)task.rec_begin + (long long)(p0 + p) * task.n_tiles + tile;
        unsigned long long *mine = L.look + rec * 4;
        const bool more = (tile + 1 < task.n_tiles) && !indep;
        if (more) {
            granule_store(mine + 0, agg.lo);
            granule_store(mine + 1, agg.hi);
            if (agg.lo == agg.hi) {
                granule_store(mine + 3, agg.lo);
            }
        }
        double din = 0.0;  // chain start: delta_0 = a_0 (the first clamp is the identity)
        if (tile > 0 && !indep) {
            const unsigned long long *prev = L.look + (rec - 1) * 4;
            const double plo = granule_wait(prev + 0, L.error);
            const double phi = granule_wait(prev + 1, L.error);
            din = (plo == phi) ? plo : granule_wait(prev + 3, L.error);
        }
        sc->tile_in[p] = din;
        if (more && agg.lo != agg.hi) {
            granule_store(mine + 3, clampd(din + agg.a, agg.lo, agg.hi));
        }
    }
    __syncthreads();

    // ---- 4. true incoming delta of every lane, classes, fill ----
    double din[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        // exclusive prefix: wavefronts before mine, then the lanes before mine
        Fn e = shfl_up_fn(inc[p], 1);
        if (lane == 0) {
            e.a = 0.0;
            e.lo = -big;
            e.hi = big;
        }
        double v = sc->tile_in[min(p, np - 1)];
        for (int w = 0; w < wave; ++w) {
            v = clampd(v + sc->wave_fn[w][p][0], sc->wave_fn[w][p][1], sc->wave_fn[w][p][2]);
        }
        din[p] = clampd(v + e.a, e.lo, e.hi);
    }
    const bool edge_tile = (base + kLeanTile >= task.m);
    const int valid = (int)max(0LL, min((long long)kLeanChunk, task.m - j0));
    const int last = (task.m - 1 >= j0 && task.m - 1 < j0 + kLeanChunk) ? (int)(task.m - 1 - j0) : -1;
    unsigned one[PB], nz[PB];
    if (MODEL) {
        LaneTolerance<PB> tol;
        if (edge_tile) {
            chunk_classes_model<PB, true>(row, x, lm, first_unclamped, big, din, valid, last, one, nz, tol);
        } else {
            chunk_classes_model<PB, false>(row, x, lm, first_unclamped, big, din, valid, last, one, nz, tol);
        }
        // tolerance a lane inherits: scan of (clear, tail) over the wavefront; its first lane takes the cap (nothing
        // at the very start of the chain)
        const double wcap = *task.wcap;
        // -gamma or -lambda may round as an exact tie on ONE grid u = 2^(e-52) (the one whose half step is the lowest
        // set bit of the number): the clean chunks of that exponent then run in hazard mode (lane_model) and their steps
        // weigh like any hazard step -- added to what a wavefront's first lane must assume
        const double q_step = ldexp(1.0, task.qexp);
        const int e_gamma = tie_exponent(task.c_raw);
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            double extra = 0.0;
            const int e_lam = tie_exponent(x[p]);
            if (e_lam + kModelMapBias >= 0 && e_lam + kModelMapBias < 128) {
                const unsigned chunks = task.clean_chunks[e_lam + kModelMapBias];
                const double hb = ldexp(1.0, e_lam + 2 - 53);
                extra += (chunks != 0u) ? (double)chunks * (double)kLeanChunk * (4.0 * hb + q_step) + 9.0 * hb + 2.0 * q_step : 0.0;
            }
            if (e_gamma != e_lam && e_gamma + kModelMapBias >= 0 && e_gamma + kModelMapBias < 128) {
                const unsigned chunks = task.clean_chunks[e_gamma + kModelMapBias];
                const double hb = ldexp(1.0, e_gamma + 2 - 53);
                extra += (chunks != 0u) ? (double)chunks * (double)kLeanChunk * (4.0 * hb + q_step) + 9.0 * hb + 2.0 * q_step : 0.0;
            }
            const double wave_in = (base == 0 && wave == 0) ? 0.0 : wcap + extra;
            bool cl = tol.clear[p];
            double tl = tol.tail[p];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int pc = __shfl_up((int)cl, off);
                const double pt = __shfl_up(tl, off);
                if (lane >= off) {
                    tl = cl ? tl : pt + tl;
                    cl = cl || (pc != 0);
                }
            }
            int ec = __shfl_up((int)cl, 1);
            double et = __shfl_up(tl, 1);
            if (lane == 0) {
                ec = 0;
                et = 0.0;
            }
            const double w_in = ec ? et : wave_in + et;
            // why a class is not certified (reported for diagnosis): 1 a class behind the lane's first clear clamp or
            // a tie, 2 tolerance beyond the guard, 4 / 8 a class before the first clear clamp within the inherited
            // tolerance (without / with local weights).  Lanes past the array's end hold no locus: their minima are
            // +inf and they never report.
            unsigned why = tol.bad[p];
            why |= (w_in + tol.tau_pre[p] > kModelGuard) ? 2u : 0u;
            why |= (w_in > 0.0 && !(tol.slack_all[p] > w_in)) ? 4u : 0u;
            why |= !(tol.slack_pos[p] > w_in) ? 8u : 0u;
            if (p >= np) {
                why = 0u;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                why |= __shfl_xor(why, off);
            }
            if (why != 0u && lane == 0) {
                atomicOr(&sc->uncertain[p], why);
            }
        }
    } else if (edge_tile) {
        chunk_classes<PB, true>(row, x, c, c_first, din, valid, last, one, nz);
    } else {
        chunk_classes<PB, false>(row, x, c, c_first, din, valid, last, one, nz);
    }

    unsigned zw[PB];    // kept-locus word (fill value entering the tile taken as 1)
    bool dep[PB];       // my incoming fill value is the tile's
    unsigned cnt0[PB], tail[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const unsigned det = one[p] | ~nz[p];
        const bool pass = (det == 0u);
        tail[p] = pass ? 32u : (unsigned)__builtin_ctz(det);
        const unsigned long long sum0 = (unsigned long long)nz[p] + (unsigned long long)one[p];
        const unsigned z0 = (unsigned)((sum0 ^ nz[p] ^ one[p]) >> 1);
        cnt0[p] = (unsigned)__builtin_popcount(z0);
        const bool v = (z0 >> 31) != 0u;
        const unsigned long long pass_b = __ballot(pass), v_b = __ballot(v);
        // first lane to my right that does not copy
        const unsigned long long right = (lane == 63) ? 0ull : (~pass_b & (~0ull << (lane + 1)));
        const bool in_wave = (right != 0ull);
        const unsigned zin_wave = in_wave ? (unsigned)((v_b >> __builtin_ctzll(right)) & 1ull) : 0u;
        if (lane == 0) {
            sc->wave_pass[wave][p] = (unsigned char)(pass_b == ~0ull);
            sc->wave_v[wave][p] = (pass_b == ~0ull) ? 0 : (unsigned char)((v_b >> __builtin_ctzll(~pass_b)) & 1ull);
        }
        zw[p] = zin_wave;       // (temporarily: the fill value found inside the wavefront)
        dep[p] = !in_wave;      // (temporarily: not found inside the wavefront)
    }
    __syncthreads();
    unsigned cells_lane[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        unsigned zin = zw[p];
        bool d = dep[p];
        if (d) {
            for (int w = wave + 1; w < 4; ++w) {
                if (!sc->wave_pass[w][p]) {
                    zin = sc->wave_v[w][p];
                    d = false;
                    break;
                }
            }
        }
        dep[p] = d;
        const unsigned zeff = d ? 1u : zin;
        const unsigned long long sum = (unsigned long long)nz[p] + (unsigned long long)one[p] + zeff;
        zw[p] = (unsigned)((sum ^ nz[p] ^ one[p]) >> 1);
        if (MODEL && task.store == 2 && p < np) {
            // solution words: the lane's selected loci for either value entering the tile (lanes whose fill does not
            // come from the tile's right edge: the same word twice)
            const unsigned long long sum0 = (unsigned long long)nz[p] + (unsigned long long)one[p] + (d ? 0u : zin);
            const unsigned z_if0 = (unsigned)((sum0 ^ nz[p] ^ one[p]) >> 1);
            const long long plane = (long long)task.n_points * task.n_tiles * kLeanThreads;
            const long long at = task.bits_begin + ((long long)(p0 + p) * task.n_tiles + tile) * kLeanThreads + t;
            L.bits[at] = zw[p];
            L.bits[at + plane] = z_if0;
        }
        if (!d) {
            cnt0[p] += tail[p] * zin;
            tail[p] = 0u;
        }
        if (lane == 0) {
            sc->first_word[wave][p] = zw[p];
        }
        if (t == kLeanThreads - 1) {
            sc->last_word[p] = zw[p];
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        // run ends inside the tile: kept locus whose right neighbour is not kept (the neighbour of the tile's
        // last locus is taken as kept; the finish kernel settles tile borders); no separator follows the
        // chain's last locus
        unsigned nxt = __shfl_down(zw[p], 1);
        if (lane == 63) {
            nxt = (wave < 3) ? sc->first_word[wave + 1][p] : 0x80000000u;
        }
        unsigned ends = zw[p] & ~((zw[p] << 1) | (nxt >> 31));
        if (last >= 0) {
            ends &= ~(0x80000000u >> last);
        }
        cells_lane[p] = (unsigned)__builtin_popcount(zw[p]) + (unsigned)__builtin_popcount(ends);
    }

    // ---- 5. per tile and penalty ----
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        unsigned b = cnt0[p], tl = tail[p], ce = cells_lane[p];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            b += __shfl_down(b, off);
            tl += __shfl_down(tl, off);
            ce += __shfl_down(ce, off);
        }
        if (lane == 0 && p < np) {
            atomicAdd(&sc->red[p][0], b);
            atomicAdd(&sc->red[p][1], tl);
            atomicAdd(&sc->red[p][2], ce);
        }
        if (p < np && task.store == 1) {
            L.bits[task.bits_begin + ((long long)(p0 + p) * task.n_tiles + tile) * kLeanThreads + t] = zw[p];
        }
    }
    __syncthreads();
    if (t < np) {
        const int p = t;
        bool all_pass = true;
        unsigned v = 0u;
        for (int w = 0; w < 4; ++w) {
            if (!sc->wave_pass[w][p]) {
                all_pass = false;
                v = sc->wave_v[w][p];
                break;
            }
        }
        LeanTileRec r;
        r.base = sc->red[p][0];
        r.tail = sc->red[p][1];
        r.cells = sc->red[p][2];
        const unsigned firstw = sc->first_word[0][p];
        r.flags = (all_pass ? 1u : 0u) | (v << 1) | ((firstw >> 31) << 2) | ((sc->last_word[p] & 1u) << 3) |
                  (MODEL ? (sc->uncertain[p] << 4) : 0u);
        L.recs[(long long)task.rec_begin + (long long)(p0 + p) * task.n_tiles + tile] = r;
    }
}

__device__ __forceinline__ void finish_task_pair(const LeanLaunch &L, const LeanTask &task, int p, int lane, bool round_too);

// One wavefront per (task, penalty): close the fill across tiles, lay out the compaction.
__device__ __forceinline__ void finish_pair(const LeanLaunch &L, int n_tasks, int pair)
{
    // the task whose penalties hold this pair: the tasks' penalty counts summed side by side (see find_task)
    const int lane = threadIdx.x;
    int ti = 0, acc = 0;
    for (int base = 0; base < n_tasks; base += 64) {
        const int i = base + lane;
        const int np_i = (i < n_tasks) ? L.tasks[i].n_points : 0;
        int incl = np_i;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            incl += (lane >= off) ? v : 0;
        }
        const unsigned long long beyond = __ballot(i < n_tasks && acc + incl > pair);  // first task whose end passes the pair
        if (beyond != 0ull) {
            const int at = __builtin_ctzll(beyond);
            ti = base + at;
            acc += __shfl(incl - np_i, at);
            break;
        }
        acc += __shfl(incl, 63);
        ti = min(n_tasks - 1, base + 63);
    }
    const LeanTask task = L.tasks[ti];
    finish_task_pair(L, task, pair - acc, lane, pair == 0);
}

// penalty `p` of `task`, by one wavefront (`lane` = its lane); `round_too`: also what is restored once per round
__device__ __forceinline__ void finish_task_pair(const LeanLaunch &L, const LeanTask &task, int p, int lane, bool round_too)
{
    const LeanTileRec *recs = L.recs + (long long)task.rec_begin + (long long)p * task.n_tiles;
    const int nt = task.n_tiles;

    // backward: fill value entering every tile from the right
    unsigned carry = 0u;  // beyond the chain's end (the last locus never copies)
    unsigned long long total = 0ull;
    unsigned uncertain = 0u;
    for (int hi = nt; hi > 0; hi -= 64) {
        const int k = hi - 64 + lane;  // lanes ascend with the tiles
        LeanTileRec r = {0u, 0u, 0u, 1u};
        if (k >= 0) {
            r = recs[k];
        }
        {
            unsigned why = (k >= 0) ? (r.flags >> 4) : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                why |= __shfl_xor(why, off);
            }
            uncertain |= why;
        }
        const bool pass = (r.flags & 1u) != 0u;
        const bool v = (r.flags & 2u) != 0u;
        const unsigned long long pass_b = __ballot(pass), v_b = __ballot(v);
        const unsigned long long right = (lane == 63) ? 0ull : (~pass_b & (~0ull << (lane + 1)));
        const unsigned zin = right ? (unsigned)((v_b >> __builtin_ctzll(right)) & 1ull) : carry;
        if (task.store == 2 && k >= 0) {
            L.tile_off[task.off_begin + (long long)p * nt + k] = zin;  // (what lean_write_solutions_kernel needs of this pair)
        }
        unsigned long long part = (k >= 0) ? ((unsigned long long)r.base + (unsigned long long)r.tail * zin) : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            part += __shfl_down(part, off);
        }
        total += __shfl(part, 0);
        if (~pass_b != 0ull) {
            carry = (unsigned)((v_b >> __builtin_ctzll(~pass_b)) & 1ull);
        }
    }

    // forward: offsets of the tiles in the compacted array
    unsigned *off_out = L.tile_off + task.off_begin + (long long)p * nt;
    const unsigned lead = ((recs[0].flags >> 2) & 1u) ? 0u : 1u;  // a separator leads unless locus 0 is kept
    unsigned long long running = lead;
    for (int lo = 0; lo < nt; lo += 64) {
        const int k = lo + lane;
        unsigned cells = 0u;
        if (k < nt) {
            const LeanTileRec r = recs[k];
            cells = r.cells;
            if (k + 1 < nt) {
                const unsigned next_first = (recs[k + 1].flags >> 2) & 1u;
                cells += ((r.flags >> 3) & 1u) & (next_first ^ 1u);  // run end at the tile's last locus
            }
        }
        unsigned incl = cells;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned u = __shfl_up(incl, off);
            if (lane >= off) {
                incl += u;
            }
        }
        if (k < nt && task.store == 1) {
            off_out[k] = (unsigned)(running + incl - cells);
        }
        running += __shfl(incl, 63);
    }
    if (lane == 0) {
        LeanResult res;
        res.count = (long long)total;
        res.child_len = (long long)running;
        res.flags = (long long)uncertain;
        L.results[task.result_begin + p] = res;
    }
    if (L.self_reset != 0) {
        // leave the round's scratch as the next round expects it: granules and tickets all-ones, error word zero
        unsigned long long *mine = L.look + ((long long)task.rec_begin + (long long)p * nt) * 4;
        for (int k = lane; k < 4 * nt; k += 64) {
            mine[k] = kSentinel;
        }
        if (round_too && lane == 0) {
            const unsigned e = __hip_atomic_load(L.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (L.error_out != nullptr) {
                *L.error_out = e;
            }
            if (L.ctl != nullptr && e != 0u) {
                atomicOr(&L.ctl->error, e);
            }
            *L.error = 0u;
            L.ticket[0] = 0xFFFFFFFFu;
            L.ticket[2] = 0xFFFFFFFFu;
        }
    }
}

__global__ __launch_bounds__(64) void lean_finish_kernel(LeanLaunch L, int n_pairs)
{
    const int pair = blockIdx.x;
    if (pair >= n_pairs) {
        return;
    }
    finish_pair(L, L.n_tasks, pair);
}

// chained rounds: the number of pairs is on the device, the grid is fixed
__global__ __launch_bounds__(64) void lean_finish_chain_kernel(LeanLaunch L)
{
    const int n_pairs = L.ctl->n_pairs, n_tasks = L.ctl->n_tasks;
    for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        finish_pair(L, n_tasks, pair);
    }
}

struct CompactShared {
    unsigned wave_sum[4];
    unsigned total_s;
    int src[kLeanTile + kLeanTile / 2 + 8];  // at most one separator per two kept loci
};

// one workgroup per tile of the parent level.  Every lane first lists the output cells of its 32-locus
// word in LDS (source locus, or -1 for the separator behind a run end) -- no memory access in that loop --,
// then the cells are dealt to the threads one by one: independent loads, coalesced stores.
__device__ __forceinline__ void compact_block(const LeanCompactTask *tasks, int n_tasks, unsigned *error, int block,
                                              CompactShared &sh)
{
    unsigned(&wave_sum)[4] = sh.wave_sum;
    unsigned &total_s = sh.total_s;
    int *src = sh.src;
    const int ti = find_task(n_tasks, block, [&](int i) { return tasks[i].block_begin; });
    const LeanCompactTask task = tasks[ti];
    const int tile = block - task.block_begin;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned z = task.bits[(long long)tile * kLeanThreads + t];
    unsigned nxt;
    if (t + 1 < kLeanThreads) {
        nxt = task.bits[(long long)tile * kLeanThreads + t + 1];
    } else if (tile + 1 < task.n_tiles) {
        nxt = task.bits[(long long)(tile + 1) * kLeanThreads];
    } else {
        nxt = 0x80000000u;  // (beyond the chain's end: never a run end, see below)
    }
    const long long tile0 = (long long)tile * kLeanTile;
    const long long j0 = tile0 + (long long)t * kLeanChunk;
    unsigned ends = z & ~((z << 1) | (nxt >> 31));
    if (task.m - 1 >= j0 && task.m - 1 < j0 + kLeanChunk) {
        ends &= ~(0x80000000u >> (int)(task.m - 1 - j0));  // no separator after the chain's last locus
    }
    const unsigned cells = (unsigned)__builtin_popcount(z) + (unsigned)__builtin_popcount(ends);
    unsigned incl = cells;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o);
        if (lane >= o) {
            incl += u;
        }
    }
    if (lane == 63) {
        wave_sum[wave] = incl;
    }
    __syncthreads();
    unsigned before = 0u;
    for (int w = 0; w < wave; ++w) {
        before += wave_sum[w];
    }
    unsigned at = before + incl - cells;
    if (t == kLeanThreads - 1) {
        total_s = before + incl;
    }
    unsigned rest = z;
    while (rest != 0u) {
        const int b = 31 - __builtin_clz(rest);
        rest &= ~(1u << b);
        src[at++] = t * kLeanChunk + (31 - b);
        if ((ends >> b) & 1u) {
            src[at++] = -1;
        }
    }
    __syncthreads();
    const unsigned total = total_s;
    const long long base = (long long)task.tile_off[tile];
    if (tile == 0 && t == 0 && base == 1) {
        task.out_s[0] = task.sep;  // leading separator
        task.out_orig[0] = -1;
    }
    if (base + total > task.capacity) {
        if (t == 0 && total != 0u) {
            atomicOr(error, 2u);
        }
        return;
    }
    for (unsigned k = (unsigned)t; k < total; k += kLeanThreads) {
        const int sidx = src[k];
        const long long pos = base + k;
        if (sidx < 0) {
            task.out_s[pos] = task.sep;
            task.out_orig[pos] = -1;
        } else {
            const long long j = tile0 + sidx;
            task.out_s[pos] = task.s[j];
            task.out_orig[pos] = (task.orig != nullptr) ? task.orig[j] : (int)j;
        }
    }
}

__global__ __launch_bounds__(kLeanThreads) void lean_compact_kernel(const LeanCompactTask *tasks, int n_tasks,
                                                                    unsigned *error)
{
    __shared__ CompactShared sh;
    compact_block(tasks, n_tasks, error, (int)blockIdx.x, sh);
}

// chained rounds: tasks and workgroup slots are counted on the device, the grid is fixed
__global__ __launch_bounds__(kLeanThreads) void lean_compact_chain_kernel(const LeanCompactTask *tasks, LeanRoundCtl *ctl)
{
    __shared__ CompactShared sh;
    const int n_blocks = ctl->n_pre_blocks, n_tasks = ctl->n_pre_tasks;
    for (int block = blockIdx.x; block < n_blocks; block += gridDim.x) {
        compact_block(tasks, n_tasks, &ctl->error, block, sh);
        __syncthreads();
    }
}

// ---- binade map by lean kernels (lean.h) ---------------------------------------------------------------------------
// One tile, one penalty: a = rn_q(s - lambda) on raw scores (chain_fast.hip: lean_apply_steps<false, ...>, step_a of an
// unmapped chunk), the chunk functions composed and handed from tile to tile exactly as eval_body does, then the
// recursion from the true incoming delta with the gain of chain_fast.hip's K3: sum over the chunk's loci j (0 < j < m) of
// max(0, delta_{j-1} - c).  Loci past the array's end take no step (K3's edge loop skips them).
__device__ __forceinline__ void map_body(const LeanLaunch &L, const LeanTask &task, int tile, double *lds, Scratch *sc,
                                         const LeanMapOut &out)
{
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const long long base = (long long)tile * kLeanTile;
    const double magic = task.magic;
    const double c = (task.c_raw + magic) - magic;
    const double big = task.big;
    const double lam = L.points[task.point_begin];
    const long long j0 = base + (long long)t * kLeanChunk;
    const int valid = (int)max(0LL, min((long long)kLeanChunk, task.m - j0));
    const double *row = lds + t * kStride;

    // chunk function (identity for a lane past the array)
    Fn f;
    f.a = 0.0;
    f.lo = -big;
    f.hi = big;
    for (int i = 0; i < valid; ++i) {
        const double cc = (j0 + i == 0) ? big : c;
        const double a = ((row[i] - lam) + magic) - magic;
        f.a += a;
        f.lo = fmin(fmax(f.lo, -cc), cc) + a;
        f.hi = fmin(fmax(f.hi, -cc), cc) + a;
    }
    Fn inc = f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Fn prev = shfl_up_fn(inc, off);
        if (lane >= off) {
            inc = compose(prev, inc);
        }
    }
    if (lane == 63) {
        sc->wave_fn[wave][0][0] = inc.a;
        sc->wave_fn[wave][0][1] = inc.lo;
        sc->wave_fn[wave][0][2] = inc.hi;
    }
    __syncthreads();
    if (t == 0) {
        Fn agg;
        agg.a = sc->wave_fn[0][0][0];
        agg.lo = sc->wave_fn[0][0][1];
        agg.hi = sc->wave_fn[0][0][2];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            Fn g;
            g.a = sc->wave_fn[w][0][0];
            g.lo = sc->wave_fn[w][0][1];
            g.hi = sc->wave_fn[w][0][2];
            agg = compose(agg, g);
        }
        const long long rec = (long long)task.rec_begin + tile;
        unsigned long long *mine = L.look + rec * 4;
        const bool more = (tile + 1 < task.n_tiles);
        if (more) {
            granule_store(mine + 0, agg.lo);
            granule_store(mine + 1, agg.hi);
            if (agg.lo == agg.hi) {
                granule_store(mine + 3, agg.lo);
            }
        }
        double din = 0.0;
        if (tile > 0) {
            const unsigned long long *prev = L.look + (rec - 1) * 4;
            const double plo = granule_wait(prev + 0, L.error);
            const double phi = granule_wait(prev + 1, L.error);
            din = (plo == phi) ? plo : granule_wait(prev + 3, L.error);
        }
        sc->tile_in[0] = din;
        if (more && agg.lo != agg.hi) {
            granule_store(mine + 3, clampd(din + agg.a, agg.lo, agg.hi));
        }
        // what the finish kernel reads of this tile (it only restores the round scratch here: nothing is counted)
        LeanTileRec r;
        r.base = 0u;
        r.tail = 0u;
        r.cells = 0u;
        r.flags = 1u;
        L.recs[rec] = r;
    }
    __syncthreads();
    // true incoming delta of this lane: wavefronts before mine, then the lanes before mine
    Fn e = shfl_up_fn(inc, 1);
    if (lane == 0) {
        e.a = 0.0;
        e.lo = -big;
        e.hi = big;
    }
    double v = sc->tile_in[0];
    for (int w = 0; w < wave; ++w) {
        v = clampd(v + sc->wave_fn[w][0][0], sc->wave_fn[w][0][1], sc->wave_fn[w][0][2]);
    }
    double d = clampd(v + e.a, e.lo, e.hi);
    double g = 0.0;
    for (int i = 0; i < valid; ++i) {
        const double a = ((row[i] - lam) + magic) - magic;
        if (j0 + i > 0) {
            g += fmax(0.0, d - c);
            d = fmin(fmax(d, -c), c) + a;
        } else {
            d = a;
        }
    }
    const long long rec = (long long)task.rec_begin + tile;
    out.gain_chunk[rec * kLeanThreads + t] = g;
    // the tile's gain: chain_fast.hip's reduction order (K3: shfl_down tree per wavefront, then ((w0 + w1) + w2) + w3)
    double gs = g;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gs += __shfl_down(gs, off);
    }
    __syncthreads();  // (wave_fn is read above by every lane)
    if (lane == 0) {
        sc->wave_fn[wave][1][0] = gs;
    }
    __syncthreads();
    if (t == 0) {
        out.gain_block[rec] = ((sc->wave_fn[0][1][0] + sc->wave_fn[1][1][0]) + sc->wave_fn[2][1][0]) + sc->wave_fn[3][1][0];
    }
}

__global__ __launch_bounds__(kLeanThreads, 2) void lean_map_kernel(LeanLaunch L, LeanMapOut out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds = smem;
    Scratch *sc = reinterpret_cast<Scratch *>(smem + kTileLds);
    const int t = threadIdx.x;
    if (t == 0) {
        sc->ticket = (int)(atomicAdd(L.ticket, 1u) + 1u);
    }
    __syncthreads();
    const int ticket = sc->ticket;
    if (ticket >= L.n_units) {
        return;
    }
    const int ti = find_task(L.n_tasks, ticket, [&](int i) { return L.tasks[i].unit_begin; });
    const LeanTask task = L.tasks[ti];
    const int tile = ticket - task.unit_begin;
    stage_tile<true>(task.s, task.m, (long long)tile * kLeanTile, task.magic, lds);
    map_body(L, task, tile, lds, sc, out);
}

// chain_fast.hip: binade_code (K6) -- the code of a chunk from the stay-off value at its two ends
__device__ __forceinline__ uint8_t lean_binade_code(double p0_lo, double p0_hi, double margin)
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
    return (uint8_t)((clean ? 0 : 0x80) | (e + kModelMapBias));
}

// One workgroup per tile: the gain before the tile in chain_fast.hip's K4 order (tiles in groups of 64: inclusive scan by
// shuffles, running carry), then K6: the chunks' inclusive scan inside the tile, the codes.
__global__ __launch_bounds__(kLeanThreads) void lean_mapcode_kernel(const LeanMapCodeTask *__restrict__ tasks, int n_tasks)
{
    __shared__ double wsum[4];
    __shared__ double pre_s;
    const int ti = find_task(n_tasks, (int)blockIdx.x, [&](int i) { return tasks[i].block_begin; });
    const LeanMapCodeTask task = tasks[ti];
    const int tile = (int)blockIdx.x - task.block_begin;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        double carry = 0.0;
        for (int base = 0; base <= tile; base += 64) {
            const int b = base + lane;
            const double g = (b < task.n_tiles) ? task.gain_block[b] : 0.0;
            double inc = g;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double p = __shfl_up(inc, off);
                if (lane >= off) {
                    inc += p;
                }
            }
            if (b == tile) {
                pre_s = carry + (inc - g);
            }
            carry += __shfl(inc, 63);
        }
    }
    const long long chunk = (long long)tile * kLeanThreads + threadIdx.x;
    const bool valid = chunk * kLeanChunk < task.m;
    const double g = valid ? task.gain_chunk[chunk] : 0.0;
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
    double pre = pre_s;
    for (int w = 0; w < wave; ++w) {
        pre += wsum[w];
    }
    const double p0_end = pre + inc;
    const double p0_start = p0_end - g;
    if (valid) {
        task.emap[chunk] = lean_binade_code(p0_start, p0_end, task.margin);
    }
}

// The 0/1 solution of a level from the words a store == 2 evaluation left: lane word -> 32 bytes (bit 31 = the lane's
// first locus).  One workgroup per tile.
__global__ __launch_bounds__(kLeanThreads) void lean_write_solutions_kernel(const LeanWriteTask *__restrict__ tasks,
                                                                            const int *__restrict__ n_tasks_dev)
{
    const int n_tasks = *n_tasks_dev;
    if (n_tasks <= 0) {
        return;
    }
    const int total = tasks[n_tasks - 1].block_begin + tasks[n_tasks - 1].n_tiles;
    for (int block = blockIdx.x; block < total; block += gridDim.x) {
        const int ti = find_task(n_tasks, block, [&](int i) { return tasks[i].block_begin; });
        const LeanWriteTask task = tasks[ti];
        const int tile = block - task.block_begin;
        const int t = threadIdx.x;
        const unsigned entering = task.entering[tile];
        const unsigned z = (entering != 0u ? task.word1 : task.word0)[(long long)tile * kLeanThreads + t];
        const long long j0 = (long long)tile * kLeanTile + (long long)t * kLeanChunk;
        if (j0 + kLeanChunk <= task.m && ((reinterpret_cast<uintptr_t>(task.solution + j0) & 15u) == 0u)) {
            uint4 lo, hi;
            unsigned *w = reinterpret_cast<unsigned *>(&lo);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned v = 0u;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    v |= ((z >> (31 - (4 * q + b))) & 1u) << (8 * b);
                }
                w[q] = v;
            }
            w = reinterpret_cast<unsigned *>(&hi);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned v = 0u;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    v |= ((z >> (31 - (16 + 4 * q + b))) & 1u) << (8 * b);
                }
                w[q] = v;
            }
            *reinterpret_cast<uint4 *>(task.solution + j0) = lo;
            *reinterpret_cast<uint4 *>(task.solution + j0 + 16) = hi;
        } else {
            for (int b = 0; b < kLeanChunk && j0 + b < task.m; ++b) {
                task.solution[j0 + b] = (uint8_t)((z >> (31 - b)) & 1u);
            }
        }
    }
}

__global__ __launch_bounds__(256) void lean_scatter_kernel(const uint8_t *__restrict__ level, const int *__restrict__ orig,
                                                           long long m, uint8_t *__restrict__ full)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        const int o = orig[i];
        if (o >= 0) {
            full[o] = level[i];
        }
    }
}

__global__ __launch_bounds__(kLeanThreads, 2) void lean_eval_kernel(LeanLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds = smem;
    Scratch *sc = reinterpret_cast<Scratch *>(smem + kTileLds);
    const int t = threadIdx.x;
    if (t == 0) {
        sc->ticket = (int)(atomicAdd(L.ticket, 1u) + 1u);
    }
    if (t < kLeanBatch * 4) {
        (&sc->red[0][0])[t] = 0u;
    }
    __syncthreads();
    const int ticket = sc->ticket;
    if (ticket >= L.n_units) {
        return;
    }
    const int ti = find_task(L.n_tasks, ticket, [&](int i) { return L.tasks[i].unit_begin; });
    const LeanTask task = L.tasks[ti];
    const int unit = ticket - task.unit_begin;
    const int tile = unit / task.n_groups, group = unit % task.n_groups;
    const int p0 = group * task.batch;
    const int np = min(task.batch, task.n_points - p0);
    stage_tile<false>(task.s, task.m, (long long)tile * task.tile_stride * kLeanTile, task.magic, lds);
    // penalties of this workgroup, in registers: 1, 2, 4 or 8 interleaved chains per lane
    if (np > 4) {
        eval_body<8, false>(L, task, tile, p0, np, lds, sc);
    } else if (np > 2) {
        eval_body<4, false>(L, task, tile, p0, np, lds, sc);
    } else if (np > 1) {
        eval_body<2, false>(L, task, tile, p0, np, lds, sc);
    } else {
        eval_body<1, false>(L, task, tile, p0, np, lds, sc);
    }
}

// chained rounds (chain.hip): the launch is queued before the round's tasks exist.  A fixed grid of resident workgroups
// reads the sizes the director left on the device and keeps taking tickets until they run out; tickets are still taken
// in tile order, so a tile only ever waits for one that a running workgroup holds.
__global__ __launch_bounds__(kLeanThreads, 2) void lean_eval_chain_kernel(LeanLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds = smem;
    Scratch *sc = reinterpret_cast<Scratch *>(smem + kTileLds);
    const int t = threadIdx.x;
    const int n_units = L.ctl->n_units, n_tasks = L.ctl->n_tasks;
    if (n_units <= 0) {
        return;  // (before the ticket is touched: no finish kernel would restore it)
    }
    for (;;) {
        if (t == 0) {
            sc->ticket = (int)(atomicAdd(L.ticket, 1u) + 1u);
        }
        if (t < kLeanBatch * 4) {
            (&sc->red[0][0])[t] = 0u;
        }
        __syncthreads();
        const int ticket = sc->ticket;
        if (ticket >= n_units) {
            return;
        }
        const int ti = find_task(n_tasks, ticket, [&](int i) { return L.tasks[i].unit_begin; });
        const LeanTask task = L.tasks[ti];
        const int unit = ticket - task.unit_begin;
        const int tile = unit / task.n_groups, group = unit % task.n_groups;
        const int p0 = group * task.batch;
        const int np = min(task.batch, task.n_points - p0);
        stage_tile<false>(task.s, task.m, (long long)tile * task.tile_stride * kLeanTile, task.magic, lds);
        if (np > 4) {
            eval_body<8, false>(L, task, tile, p0, np, lds, sc);
        } else if (np > 2) {
            eval_body<4, false>(L, task, tile, p0, np, lds, sc);
        } else if (np > 1) {
            eval_body<2, false>(L, task, tile, p0, np, lds, sc);
        } else {
            eval_body<1, false>(L, task, tile, p0, np, lds, sc);
        }
        __syncthreads();  // the tile and the scratch are reused
    }
}

// rounding-model tasks: raw scores in LDS, up to kLeanModelBatch penalties per workgroup
__global__ __launch_bounds__(kLeanThreads, 2) void lean_model_kernel(LeanLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds = smem;
    Scratch *sc = reinterpret_cast<Scratch *>(smem + kTileLds);
    const int t = threadIdx.x;
    if (t == 0) {
        sc->ticket = (int)(atomicAdd(L.ticket, 1u) + 1u);
    }
    if (t < kLeanBatch * 4) {
        (&sc->red[0][0])[t] = 0u;
    }
    if (t < kLeanBatch) {
        sc->uncertain[t] = 0u;
    }
    __syncthreads();
    const int ticket = sc->ticket;
    if (ticket >= L.n_units) {
        return;
    }
    const int ti = find_task(L.n_tasks, ticket, [&](int i) { return L.tasks[i].unit_begin; });
    const LeanTask task = L.tasks[ti];
    const int unit = ticket - task.unit_begin;
    const int tile = unit / task.n_groups, group = unit % task.n_groups;
    const int p0 = group * task.batch;
    const int np = min(task.batch, task.n_points - p0);
    stage_tile<true>(task.s, task.m, (long long)tile * kLeanTile, task.magic, lds);
    if (np > 2) {
        eval_body<4, true>(L, task, tile, p0, np, lds, sc);
    } else if (np > 1) {
        eval_body<2, true>(L, task, tile, p0, np, lds, sc);
    } else {
        eval_body<1, true>(L, task, tile, p0, np, lds, sc);
    }
}

// the same for a chained round (model_chain.hip): sizes on the device, a fixed grid of workgroups that take tickets
// until they run out
__global__ __launch_bounds__(kLeanThreads, 2) void lean_model_chain_kernel(LeanLaunch L)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds = smem;
    Scratch *sc = reinterpret_cast<Scratch *>(smem + kTileLds);
    const int t = threadIdx.x;
    const int n_units = L.ctl->n_units, n_tasks = L.ctl->n_tasks;
    if (n_units <= 0) {
        return;  // (before the ticket is touched: no finish kernel would restore it)
    }
    for (;;) {
        if (t == 0) {
            sc->ticket = (int)(atomicAdd(L.ticket, 1u) + 1u);
        }
        if (t < kLeanBatch * 4) {
            (&sc->red[0][0])[t] = 0u;
        }
        if (t < kLeanBatch) {
            sc->uncertain[t] = 0u;
        }
        __syncthreads();
        const int ticket = sc->ticket;
        if (ticket >= n_units) {
            return;
        }
        const int ti = find_task(n_tasks, ticket, [&](int i) { return L.tasks[i].unit_begin; });
        const LeanTask task = L.tasks[ti];
        const int unit = ticket - task.unit_begin;
        const int tile = unit / task.n_groups, group = unit % task.n_groups;
        const int p0 = group * task.batch;
        const int np = min(task.batch, task.n_points - p0);
        stage_tile<true>(task.s, task.m, (long long)tile * kLeanTile, task.magic, lds);
        if (np > 2) {
            eval_body<4, true>(L, task, tile, p0, np, lds, sc);
        } else if (np > 1) {
            eval_body<2, true>(L, task, tile, p0, np, lds, sc);
        } else {
            eval_body<1, true>(L, task, tile, p0, np, lds, sc);
        }
        __syncthreads();  // the tile and the scratch are reused
    }
}

// A chained round as ONE launch (lean.h: LeanRoundReset): the compactions' blocks by tickets, then -- once every block is
// written -- the evaluation's tickets; the workgroup that completes the last tile of a (task, penalty) pair finishes the
// pair with one of its wavefronts.  Every wait is for work a RUNNING workgroup holds (a block or a tile is taken only by a
// workgroup that is running), so the launch cannot wait for a workgroup that has not started.
template <bool MODEL, bool FINISH>
__global__ __launch_bounds__(kLeanThreads, 2) void lean_round_chain_kernel(LeanLaunch L, const LeanCompactTask *pre, unsigned *progress)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *lds = smem;
    Scratch *sc = reinterpret_cast<Scratch *>(smem + kTileLds);
    __shared__ int s_block, s_nfin, s_fin[kLeanBatch];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n_units = L.ctl->n_units, n_tasks = L.ctl->n_tasks;
    const int n_pre_blocks = (MODEL || pre == nullptr) ? 0 : L.ctl->n_pre_blocks, n_pre_tasks = L.ctl->n_pre_tasks;
    if (n_units <= 0 && n_pre_blocks <= 0) {
        return;  // (before a ticket is touched)
    }
    if (n_pre_blocks > 0) {
        static_assert(sizeof(CompactShared) <= (size_t)kTileLds * sizeof(double), "the compaction's lists fit the tile");
        CompactShared &sh = *reinterpret_cast<CompactShared *>(smem);
        // (the loop's condition is a scalar and thread 0's extra work sits INSIDE the body: with the ticket taken at the top of
        // a for (;;) that is left by `break`, the compiler split the loop by lane -- thread 0 never took a second ticket and
        // the other lanes ran the same block for ever)
        if (t == 0) {
            s_block = (int)atomicAdd(&progress[0], 1u);
        }
        __syncthreads();
        int block = __builtin_amdgcn_readfirstlane(s_block);
        while (block < n_pre_blocks) {
            compact_block(pre, n_pre_tasks, &L.ctl->error, block, sh);
            __syncthreads();  // (every thread's stores are issued; the lists are free again)
            if (t == 0) {
                __threadfence();
                atomicAdd(&progress[1], 1u);
                s_block = (int)atomicAdd(&progress[0], 1u);
            }
            __syncthreads();
            block = __builtin_amdgcn_readfirstlane(s_block);
        }
        // the levels the evaluation reads are complete when every block is: all of them are held by running workgroups
        if (t == 0) {
            unsigned spins = 0;  // (bounded, as every wait of these kernels: gives up and reports rather than hang the device)
            while (__hip_atomic_load(&progress[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)n_pre_blocks) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kSpinLimit) {
                    atomicOr(L.error, 1u);
                    break;
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (n_units <= 0) {
        return;
    }
    unsigned *ticket_word = L.ticket;
    for (;;) {
        if (t == 0) {
            sc->ticket = (int)(atomicAdd(ticket_word, 1u) + 1u);
        }
        if (t < kLeanBatch * 4) {
            (&sc->red[0][0])[t] = 0u;
        }
        if (MODEL && t < kLeanBatch) {
            sc->uncertain[t] = 0u;
        }
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(sc->ticket);
        if (ticket >= n_units) {
            return;
        }
        const int ti = find_task(n_tasks, ticket, [&](int i) { return L.tasks[i].unit_begin; });
        const LeanTask task = L.tasks[ti];
        const int unit = ticket - task.unit_begin;
        const int tile = unit / task.n_groups, group = unit % task.n_groups;
        const int p0 = group * task.batch;
        const int np = min(task.batch, task.n_points - p0);
        if (MODEL) {
            stage_tile<true>(task.s, task.m, (long long)tile * kLeanTile, task.magic, lds);
            if (np > 2) {
                eval_body<4, true>(L, task, tile, p0, np, lds, sc);
            } else if (np > 1) {
                eval_body<2, true>(L, task, tile, p0, np, lds, sc);
            } else {
                eval_body<1, true>(L, task, tile, p0, np, lds, sc);
            }
        } else {
            stage_tile<false>(task.s, task.m, (long long)tile * task.tile_stride * kLeanTile, task.magic, lds);
            if (np > 4) {
                eval_body<8, false>(L, task, tile, p0, np, lds, sc);
            } else if (np > 2) {
                eval_body<4, false>(L, task, tile, p0, np, lds, sc);
            } else if (np > 1) {
                eval_body<2, false>(L, task, tile, p0, np, lds, sc);
            } else {
                eval_body<1, false>(L, task, tile, p0, np, lds, sc);
            }
        }
        __syncthreads();  // the tile's records (and words) are written
        if (!FINISH) {
            continue;  // (the pairs are finished by a launch of their own behind this one)
        }
        // which of this unit's pairs are complete with it
        if (t < 64) {
            // (lane p counts for penalty p: the counters' round trips side by side, not one after the other)
            __threadfence();
            bool last = false;
            if (t < np) {
                last = atomicAdd(&progress[kLeanProgressPairs + task.result_begin + p0 + t], 1u) + 1u == (unsigned)task.n_tiles;
            }
            const unsigned long long lasts = __ballot(last);
            if (last) {
                s_fin[__builtin_popcountll(lasts & ((1ull << t) - 1ull))] = p0 + t;
            }
            if (t == 0) {
                s_nfin = __builtin_popcountll(lasts);
            }
        }
        __syncthreads();
        const int nf = __builtin_amdgcn_readfirstlane(s_nfin);
        if (nf > 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other tiles' records, written by other workgroups
            for (int k = wave; k < nf; k += kLeanThreads / 64) {
                finish_task_pair(L, task, s_fin[k], lane, false);
            }
        }
        __syncthreads();  // the tile and the scratch are reused
    }
}

// What a locus can inherit at most (see lean_model_kernel): the weights of every hazard chunk's steps and of every
// exact half-way tie rn_u(s) in a clean chunk, plus the largest hazard base.  Counted per exponent (integers: no
// order dependence) by one workgroup per tile, summed by one thread per task, which also clears the counters.
__global__ __launch_bounds__(kLeanThreads) void lean_wcap_count_kernel(const LeanWcapTask *__restrict__ tasks, int n_tasks)
{
    __shared__ unsigned steps[128], ties[128], clean[128];
    __shared__ double magic_u[kLeanThreads], half_u[kLeanThreads];  // per chunk of the tile (0: hazard, no tie test)
    const int ti = find_task(n_tasks, (int)blockIdx.x, [&](int i) { return tasks[i].block_begin; });
    const LeanWcapTask task = tasks[ti];
    const int t = threadIdx.x;
    if (t < 128) {
        steps[t] = 0u;
        ties[t] = 0u;
        clean[t] = 0u;
    }
    __syncthreads();
    const long long base = (long long)((int)blockIdx.x - task.block_begin) * kLeanTile;
    const long long n_chunks = (task.m + kLeanChunk - 1) / kLeanChunk;
    const long long chunk = base / kLeanChunk + t;
    int my_bin = 0;
    magic_u[t] = 0.0;
    half_u[t] = 0.0;
    if (chunk < n_chunks) {
        const int code = task.emap[chunk];
        int e = (code & 0x7F) - kModelMapBias;
        const bool code_hz = (code & 0x80) != 0;
        if (code_hz) {
            e = max(e, task.e_floor);
        }
        my_bin = min(max(e + kModelMapBias, 0), 127);
        if (code_hz || e - 52 < task.qexp) {
            atomicAdd(&steps[my_bin], (unsigned)kLeanChunk);
        } else {
            magic_u[t] = ldexp(1.5, e);
            half_u[t] = ldexp(1.0, e - 53);
            atomicAdd(&clean[my_bin], 1u);
        }
    }
    __syncthreads();
    // coalesced pass over the tile's scores: element r * 256 + t lies in chunk (r * 256 + t) / 32 of the tile
#pragma unroll 4
    for (int r = 0; r < kLeanChunk; ++r) {
        const int el = r * kLeanThreads + t;
        const long long j = base + el;
        const int c = el / kLeanChunk;
        if (j < task.m && half_u[c] != 0.0) {
            const double x = task.s[j];
            const double su = (x + magic_u[c]) - magic_u[c];
            if (fabs(x - su) == half_u[c]) {
                // (the chunk's bin: recomputed from its grid)
                atomicAdd(&ties[min(max(ilogb(half_u[c]) + 53 + kModelMapBias, 0), 127)], 1u);
            }
        }
    }
    __syncthreads();
    if (t < 128) {
        if (steps[t] != 0u) {
            atomicAdd(&task.counters[t], steps[t]);
        }
        if (ties[t] != 0u) {
            atomicAdd(&task.counters[128 + t], ties[t]);
        }
        if (clean[t] != 0u) {
            atomicAdd(&task.counters[256 + t], clean[t]);
        }
    }
}

// one workgroup of 128 threads per task: every thread fetches (and clears) the counters of its exponent at once, then
// ONE thread adds the terms in ascending exponent order from LDS -- the same additions in the same order as a lone
// thread walking the counters in memory, without its 128 dependent round trips (39 us per launch -> 3)
__global__ __launch_bounds__(128) void lean_wcap_sum_kernel(const LeanWcapTask *__restrict__ tasks, int n_tasks)
{
    __shared__ unsigned st_s[128], tn_s[128];
    const int ti = blockIdx.x;
    if (ti >= n_tasks) {
        return;
    }
    const LeanWcapTask task = tasks[ti];
    const int b = threadIdx.x;
    st_s[b] = task.counters[b];
    tn_s[b] = task.counters[128 + b];
    task.clean_chunks[b] = task.counters[256 + b];  // kept: a penalty that ties on this grid turns them hazard
    task.counters[b] = 0u;
    task.counters[128 + b] = 0u;
    task.counters[256 + b] = 0u;
    __syncthreads();
    if (b == 0) {
        const double q = ldexp(1.0, task.qexp);
        double sum = 0.0, base = 0.0;
        for (int k = 0; k < 128; ++k) {
            const int e = k - kModelMapBias;
            const unsigned st = st_s[k], tn = tn_s[k];
            if (st != 0u) {
                const double hb = ldexp(1.0, e + 2 - 53);
                sum += (double)st * (4.0 * hb + q);
                base = 9.0 * hb + 2.0 * q;  // exponents ascend: the largest base stays
            }
            if (tn != 0u) {
                sum += (double)tn * ldexp(1.0, e - 52);
            }
        }
        task.wcap[0] = sum + base;
    }
}

int launch_eval_all(const LeanLaunch &L, hipStream_t stream)
{
    static bool configured = false;
    const size_t lds = (size_t)kTileLds * sizeof(double) + sizeof(Scratch);
    if (!configured) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_eval_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(lean_eval_kernel, dim3((unsigned)L.n_units), dim3(kLeanThreads), lds, stream, L);
    return ROCCO_HIP_OK;
}

}  // namespace

int launch_lean_eval(const LeanLaunch &L, hipStream_t stream)
{
    if (L.n_units <= 0) {
        return ROCCO_HIP_OK;
    }
    int rc = launch_eval_all(L, stream);
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_eval_chain(const LeanLaunch &L, int grid, hipStream_t stream)
{
    static bool configured = false;
    const size_t lds = (size_t)kTileLds * sizeof(double) + sizeof(Scratch);
    if (!configured) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_eval_chain_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(lean_eval_chain_kernel, dim3((unsigned)grid), dim3(kLeanThreads), lds, stream, L);
    return ROCCO_HIP_OK;
}

int launch_lean_finish_chain(const LeanLaunch &L, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(lean_finish_chain_kernel, dim3((unsigned)grid), dim3(64), 0, stream, L);
    return ROCCO_HIP_OK;
}

int launch_lean_compact_chain(const LeanCompactTask *tasks_dev, LeanRoundCtl *ctl, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(lean_compact_chain_kernel, dim3((unsigned)grid), dim3(kLeanThreads), 0, stream, tasks_dev, ctl);
    return ROCCO_HIP_OK;
}

int launch_lean_round_chain(const LeanLaunch &L, const LeanCompactTask *pre, unsigned *progress, int grid, int model, hipStream_t stream, int finish)
{
    static bool configured = false;
    const size_t lds = (size_t)kTileLds * sizeof(double) + sizeof(Scratch);
    if (!configured) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_round_chain_kernel<false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_round_chain_kernel<false, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_round_chain_kernel<true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    if (model) {
        hipLaunchKernelGGL((lean_round_chain_kernel<true, true>), dim3((unsigned)grid), dim3(kLeanThreads), lds, stream, L, pre, progress);
    } else if (finish) {
        hipLaunchKernelGGL((lean_round_chain_kernel<false, true>), dim3((unsigned)grid), dim3(kLeanThreads), lds, stream, L, pre, progress);
    } else {
        hipLaunchKernelGGL((lean_round_chain_kernel<false, false>), dim3((unsigned)grid), dim3(kLeanThreads), lds, stream, L, pre, progress);
    }
    return ROCCO_HIP_OK;
}

int launch_lean_model(const LeanLaunch &L, hipStream_t stream)
{
    if (L.n_units <= 0) {
        return ROCCO_HIP_OK;
    }
    static bool configured = false;
    const size_t lds = (size_t)kTileLds * sizeof(double) + sizeof(Scratch);
    if (!configured) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_model_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(lean_model_kernel, dim3((unsigned)L.n_units), dim3(kLeanThreads), lds, stream, L);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_model_chain(const LeanLaunch &L, int grid, hipStream_t stream)
{
    static bool configured = false;
    const size_t lds = (size_t)kTileLds * sizeof(double) + sizeof(Scratch);
    if (!configured) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_model_chain_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(lean_model_chain_kernel, dim3((unsigned)grid), dim3(kLeanThreads), lds, stream, L);
    return ROCCO_HIP_OK;
}

int launch_lean_map(const LeanLaunch &L, const LeanMapOut &out, hipStream_t stream)
{
    if (L.n_units <= 0) {
        return ROCCO_HIP_OK;
    }
    static bool configured = false;
    const size_t lds = (size_t)kTileLds * sizeof(double) + sizeof(Scratch);
    if (!configured) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lean_map_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(lean_map_kernel, dim3((unsigned)L.n_units), dim3(kLeanThreads), lds, stream, L, out);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_mapcode(const LeanMapCodeTask *tasks_dev, int n_tasks, int n_blocks, hipStream_t stream)
{
    if (n_tasks <= 0 || n_blocks <= 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(lean_mapcode_kernel, dim3((unsigned)n_blocks), dim3(kLeanThreads), 0, stream, tasks_dev, n_tasks);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_write_solutions(const LeanWriteTask *tasks_dev, const int *n_tasks_dev, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(lean_write_solutions_kernel, dim3((unsigned)grid), dim3(kLeanThreads), 0, stream, tasks_dev, n_tasks_dev);
    return ROCCO_HIP_OK;
}

int launch_lean_wcap(const LeanWcapTask *tasks_dev, int n_tasks, int n_blocks, hipStream_t stream)
{
    if (n_tasks <= 0 || n_blocks <= 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(lean_wcap_count_kernel, dim3((unsigned)n_blocks), dim3(kLeanThreads), 0, stream, tasks_dev, n_tasks);
    hipLaunchKernelGGL(lean_wcap_sum_kernel, dim3((unsigned)n_tasks), dim3(128), 0, stream, tasks_dev, n_tasks);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_finish(const LeanLaunch &L, int n_pairs, hipStream_t stream)
{
    if (n_pairs <= 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(lean_finish_kernel, dim3((unsigned)n_pairs), dim3(64), 0, stream, L, n_pairs);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_compact(const LeanCompactTask *tasks_dev, int n_tasks, int n_blocks, unsigned *error_dev, hipStream_t stream)
{
    if (n_blocks <= 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(lean_compact_kernel, dim3((unsigned)n_blocks), dim3(kLeanThreads), 0, stream, tasks_dev, n_tasks,
                       error_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_scatter_batch(const LeanScatterTask *tasks_dev, int n_tasks, int zero_blocks, int scatter_blocks,
                              hipStream_t stream)
{
    if (n_tasks <= 0) {
        return ROCCO_HIP_OK;
    }
    if (zero_blocks > 0) {
        hipLaunchKernelGGL(lean_zero_batch_kernel, dim3((unsigned)zero_blocks), dim3(256), 0, stream, tasks_dev, n_tasks);
    }
    if (scatter_blocks > 0) {
        hipLaunchKernelGGL(lean_scatter_batch_kernel, dim3((unsigned)scatter_blocks), dim3(256), 0, stream, tasks_dev, n_tasks);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_lean_scatter(const uint8_t *solution_level, const int *orig, long long m, uint8_t *solution_full, hipStream_t stream)
{
    if (m <= 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(lean_scatter_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, solution_level, orig, m,
                       solution_full);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
