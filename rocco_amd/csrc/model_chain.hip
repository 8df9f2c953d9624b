// rocco_amd/csrc/model_chain.hip -- see model_chain.h: the director of the chained rounding-model rounds.
//
// One workgroup of 1024 lanes, 8 to 64 of them per problem (a power of two above the round's penalties), every problem
// of the batch side by side.  Between two rounds it
//   1. writes the evaluated penalties of the round that just ended into pinned host memory (the host follows the chain
//      while it runs) and walks the reference's bisection through them (search.cpp: the probe walk of calibrate_batch
//      followed by advance_analytic),
//   2. picks the depth of the next round by the rule the host would apply (search.cpp: spec_depth of compacted batches;
//      budget.hip: probe_depth),
//   3. lists the next round's penalties -- the open nodes of the bisection tree below the new bracket (search.cpp:
//      build_open_tree), one lane per node, each lane replaying its own path from the root --,
//   4. sets the per-round fields of the tasks and writes the round's sizes for the launches behind it.
// Nothing here decides anything: a penalty the host would not have asked about costs time, not correctness.
#include "model_chain.h"

namespace rocco {

namespace {

constexpr int kDirectorThreads = 1024;

// search.cpp: known_count (analytic_count, then the certified thresholds); the terms in the host's order
__device__ __forceinline__ bool known_outcome(const ModelChainWalk &in, double lambda, bool *greater)
{
    const double mag = (double)in.n * (fabs(lambda) + in.sabs + in.cost_max + 1.0);
    if (mag < 1.0e15 && in.cost_ok != 0) {
        if (lambda >= in.none_from) {
            *greater = false;
            return true;
        }
        if (lambda <= in.all_upto) {
            *greater = in.n > in.target;
            return true;
        }
    }
    if (in.G_real != 0 && lambda <= in.G) {
        *greater = in.cG > in.target;
        return true;
    }
    if (in.L_real != 0 && lambda >= in.L) {
        *greater = in.cL > in.target;
        return true;
    }
    return false;
}

// budget.hip: probe_depth (the evaluator's wish from the tiles of the round before), search.cpp: max with the floor
__device__ __forceinline__ int next_depth(const ModelChainArgs &A, long long tiles_before)
{
    int wish = A.depth_fixed;
    if (wish <= 0) {
        wish = 3;
        while (wish < 6 && tiles_before * ((((1LL << (wish + 1)) - 1) + kLeanModelBatch - 1) / kLeanModelBatch) <= 512) {
            ++wish;
        }
    }
    return min(kModelChainMaxDepth, max(A.depth_floor, wish));
}

struct DirectorShared {
    ModelChainState st[kModelChainMaxProblems];  // every problem's state, global -> LDS -> global once per call
    int np[kModelChainMaxProblems];
    long long count[kDirectorThreads];  // the round's results of the problems of one pass, lane by lane
    int open[kDirectorThreads];
    unsigned long long tiles_before;
    int asked, stopped, depth;
};

// G lanes per problem (a power of two above the round's open nodes), 1024 / G problems per pass.
// The round that ended: its evaluated penalties go to the host, the bisection is walked through them.
template <int G>
__device__ __forceinline__ void consume_round(const ModelChainArgs &A, int round, DirectorShared &sh)
{
    constexpr int kGroups = kDirectorThreads / G;
    const int g = threadIdx.x / G, j = threadIdx.x % G;
    const int B = A.n_problems;
    for (int i0 = 0; i0 < B; i0 += kGroups) {
        const int i = i0 + g;
        const bool have = i < B;
        const bool asked = have && sh.st[i].active != 0 && sh.st[i].n_points > 0;
        LeanResult res = {0, 0, 1};
        if (asked && j < sh.st[i].n_points) {
            res = A.results[i * kLeanMaxPoints + j];
            ModelChainFact f;
            f.penalty = A.points[i * kLeanMaxPoints + j];
            f.count = res.count;
            f.flags = res.flags;
            A.facts[((size_t)(round - 1) * B + i) * kLeanMaxPoints + j] = f;
        }
        sh.count[threadIdx.x] = res.count;
        sh.open[threadIdx.x] = (res.flags != 0) ? 1 : 0;
        __syncthreads();
        if (have && j == 0) {
            ModelChainState st = sh.st[i];
            A.n_points_out[(size_t)(round - 1) * B + i] = asked ? st.n_points : 0;
            if (asked) {
                const ModelChainWalk in = A.walk[i];
                atomicAdd(&sh.asked, 1);
                atomicAdd(&sh.tiles_before, (unsigned long long)in.n_tiles);
                // open steps read their count, known steps cost nothing (search.cpp: the probe walk, then advance_analytic)
                const int depth = A.globals[0];
                const long long nt = in.n_tiles, np = st.n_points;
                const long long pairs_at = (long long)(round - 1) * A.cap_pairs + st.rec_begin;
                double lo = st.lower, hi = st.upper;
                int left = st.iters_left, open = 0, h = 0;
                while (left > 0) {
                    const double mid = (lo + hi) / 2.0;  // rocco/dp.py:143
                    bool greater = false;
                    int slot = -1;
                    if (!known_outcome(in, mid, &greater)) {
                        if (open >= depth || ((st.mask >> h) & 1ull) == 0ull) {
                            break;
                        }
                        slot = __popcll(st.mask & ((1ull << h) - 1ull));
                        if (sh.open[g * G + slot] != 0) {
                            st.active = 0;  // not certified: the host's own machinery takes this problem from here
                            st.stopped = 1;
                            atomicAdd(&sh.stopped, 1);
                            break;
                        }
                        greater = sh.count[g * G + slot] > in.target;
                        ++open;
                        h = 2 * h + 1 + (greater ? 1 : 0);
                    }
                    if (greater) {
                        lo = mid;
                    } else {
                        hi = mid;
                        // the evaluation behind the new upper end, if it kept what writes a solution
                        st.upper_known = (slot >= 0 && st.stored != 0) ? 1 : 0;
                        if (st.upper_known != 0) {
                            st.upper_w1 = pairs_at * (2 * 256) + ((long long)slot * nt) * 256;
                            st.upper_w0 = st.upper_w1 + np * nt * 256;
                            st.upper_zin = pairs_at + (long long)slot * nt;
                            st.upper_count = sh.count[g * G + slot];
                        }
                    }
                    --left;
                }
                st.lower = lo;
                st.upper = hi;
                st.iters_left = left;
            }
            st.n_points = 0;
            st.mask = 0ull;
            sh.st[i] = st;
        }
        __syncthreads();
    }
}

// The next round's penalties: the open nodes of the bisection tree below the bracket (search.cpp: build_open_tree), one
// lane per node of the heap, each lane replaying its own path from the root.
template <int G>
__device__ __forceinline__ void plan_round(const ModelChainArgs &A, int depth, DirectorShared &sh)
{
    constexpr int kGroups = kDirectorThreads / G;
    const int g = threadIdx.x / G, j = threadIdx.x % G;
    const int lane = threadIdx.x & 63;
    const int B = A.n_problems;
    for (int i0 = 0; i0 < B; i0 += kGroups) {
        const int i = i0 + g;
        const bool have = i < B && sh.st[i].active != 0;
        bool exists = false;
        double mid_out = 0.0;
        const int idx = j + 1;          // heap index, one-based
        const int k = 31 - __clz(idx);  // open steps above this node
        if (have && k < depth) {
            const ModelChainWalk in = A.walk[i];
            double lo = sh.st[i].lower, hi = sh.st[i].upper;
            int left = sh.st[i].iters_left;
            for (int level = 0; level <= k; ++level) {
                bool open_here = false;
                double mid = 0.0;
                while (left > 0) {
                    mid = (lo + hi) / 2.0;
                    bool greater = false;
                    if (!known_outcome(in, mid, &greater)) {
                        open_here = true;
                        break;
                    }
                    if (greater) {
                        lo = mid;
                    } else {
                        hi = mid;
                    }
                    --left;
                }
                if (!open_here) {
                    break;  // the reference's steps end above this node
                }
                if (level == k) {
                    exists = true;
                    mid_out = mid;
                    break;
                }
                if ((idx >> (k - 1 - level)) & 1) {
                    lo = mid;  // "more than the target": the node's right child
                } else {
                    hi = mid;
                }
                --left;
            }
        }
        const unsigned long long wave_bits = __ballot(exists);
        const unsigned long long mask = (G == 64) ? wave_bits : ((wave_bits >> ((lane / G) * G)) & ((1ull << (G & 63)) - 1ull));
        const int np = __popcll(mask);
        if (exists) {
            A.points[i * kLeanMaxPoints + __popcll(mask & ((1ull << j) - 1ull))] = mid_out;
        }
        __syncthreads();  // (the state above was read by every lane of the group)
        if (i < B && j == 0) {
            sh.st[i].mask = mask;
            sh.st[i].active = (have && np > 0) ? 1 : 0;  // (np == 0: every step left is known, nothing more to ask)
            sh.st[i].n_points = np;
            sh.np[i] = np;
        }
    }
}

__device__ __forceinline__ int lanes_for(int depth) { return depth <= 3 ? 8 : (depth == 4 ? 16 : (depth == 5 ? 32 : 64)); }

__device__ __forceinline__ int wave_exclusive(int v, int lane, int *total)
{
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off);
        incl += (lane >= off) ? u : 0;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

__global__ __launch_bounds__(kDirectorThreads) void model_chain_director_kernel(ModelChainArgs A, int round, int last)
{
    __shared__ DirectorShared sh;
    const int B = A.n_problems;
    const int t = threadIdx.x;
    lean_round_reset(A.reset, A.ctl);  // (what the round before left: tickets, error word, progress counters)
    if (t == 0) {
        sh.tiles_before = 0ull;
        sh.asked = 0;
        sh.stopped = 0;
    }
    // ---- the problems' state ----
    constexpr int kStateWords = (int)(sizeof(ModelChainState) / 8);
    if (round == 0) {
        for (int i = t; i < B; i += kDirectorThreads) {
            const ModelChainWalk in = A.walk[i];
            ModelChainState st;
            st.lower = in.lower;
            st.upper = in.upper;
            st.mask = 0ull;
            st.iters_left = in.iters_left;
            st.active = 1;
            st.n_points = 0;
            st.stopped = 0;
            st.upper_w1 = st.upper_w0 = st.upper_zin = st.upper_count = 0;
            st.upper_known = 0;  // (the host's upper end: no evaluation of this chain behind it)
            st.rec_begin = 0;
            st.stored = 0;
            st.pad = 0;
            sh.st[i] = st;
        }
    } else {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(A.state);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(sh.st);
        for (int i = t; i < B * kStateWords; i += kDirectorThreads) {
            dst[i] = src[i];
        }
    }
    __syncthreads();

    // ---- 1. the round that ended ----
    if (round > 0) {
        switch (lanes_for(A.globals[0])) {
        case 8: consume_round<8>(A, round, sh); break;
        case 16: consume_round<16>(A, round, sh); break;
        case 32: consume_round<32>(A, round, sh); break;
        default: consume_round<64>(A, round, sh); break;
        }
        // the host follows the chain: this round's facts are in place
        __threadfence_system();
        __syncthreads();
        if (t == 0) {
            __hip_atomic_store(&A.report->published, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (t == 0) {
        int stopped_all = A.globals[2], asked_rounds = A.globals[1];
        if (round == 0) {
            stopped_all = 0;
            asked_rounds = 0;
        }
        stopped_all += sh.stopped;
        asked_rounds += (sh.asked > 0) ? 1 : 0;
        A.globals[1] = asked_rounds;
        A.globals[2] = stopped_all;
        sh.depth = (round == 0) ? min(kModelChainMaxDepth, A.depth0) : next_depth(A, (long long)sh.tiles_before);
        if (last == 0) {
            A.globals[0] = sh.depth;
        }
    }
    __syncthreads();
    if (last != 0) {
        // ---- the chain's end: what every problem's bisection came to, and the solutions that can be written ----
        __shared__ int cand_s[kModelChainMaxProblems], tiles_s[kModelChainMaxProblems], rank_s[kModelChainMaxProblems],
            block_s[kModelChainMaxProblems];
        if (t < kModelChainMaxProblems) {
            int cand = 0, tiles = 0;
            if (t < B) {
                const ModelChainState &st = sh.st[t];
                cand = (st.iters_left == 0 && st.stopped == 0 && st.upper_known != 0 && A.walk[t].solution != nullptr) ? 1 : 0;
                tiles = cand ? A.walk[t].n_tiles : 0;
            }
            cand_s[t] = cand;
            tiles_s[t] = tiles;
        }
        __syncthreads();
        if (t < 64) {
            int ca, cb, ta, tb;
            const int xa = wave_exclusive(cand_s[t], t, &ca), xb = wave_exclusive(cand_s[t + 64], t, &cb);
            const int ya = wave_exclusive(tiles_s[t], t, &ta), yb = wave_exclusive(tiles_s[t + 64], t, &tb);
            rank_s[t] = xa;
            rank_s[t + 64] = ca + xb;
            block_s[t] = ya;
            block_s[t + 64] = ta + yb;
            if (t == 0) {
                *A.n_writes = ca + cb;
            }
        }
        __syncthreads();
        if (t < B) {
            const ModelChainState &st = sh.st[t];
            ModelChainFinal f;
            f.lower = st.lower;
            f.upper = st.upper;
            f.count = st.upper_count;
            f.iters_left = st.iters_left;
            f.written = cand_s[t];
            if (cand_s[t] != 0) {
                const ModelChainWalk in = A.walk[t];
                LeanWriteTask w;
                w.word1 = A.bits + st.upper_w1;
                w.word0 = A.bits + st.upper_w0;
                w.entering = A.entering + st.upper_zin;
                w.solution = in.solution;
                w.m = in.m;
                w.n_tiles = in.n_tiles;
                w.block_begin = block_s[t];
                A.writes[rank_s[t]] = w;
            }
            A.finals[t] = f;
        }
        __threadfence_system();
        __syncthreads();
        if (t == 0) {
            A.report->error = atomicOr(&A.ctl->error, 0u);
            A.report->stopped = A.globals[2];
            A.report->rounds_run = A.globals[1];
            __threadfence_system();
            __hip_atomic_store(&A.report->finished, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }

    // ---- 2. the next round's penalties ----
    const int depth = sh.depth;
    switch (lanes_for(depth)) {
    case 8: plan_round<8>(A, depth, sh); break;
    case 16: plan_round<16>(A, depth, sh); break;
    case 32: plan_round<32>(A, depth, sh); break;
    default: plan_round<64>(A, depth, sh); break;
    }
    __syncthreads();

    // ---- 3. sizes: the tasks' tickets and records side by side (wavefront 0, two problems per lane) ----
    if (t < 64) {
        const int a = t, b = t + 64;
        const int np_a = (a < B) ? sh.np[a] : 0, np_b = (b < B) ? sh.np[b] : 0;
        const int nt_a = (a < B) ? A.walk[a].n_tiles : 0, nt_b = (b < B) ? A.walk[b].n_tiles : 0;
        int two_a, two_b, four_a, four_b, rec_a, rec_b, pair_a, pair_b;
        const int x2a = wave_exclusive(nt_a * ((np_a + 1) / 2), t, &two_a), x2b = wave_exclusive(nt_b * ((np_b + 1) / 2), t, &two_b);
        const int x4a = wave_exclusive(nt_a * ((np_a + kLeanModelBatch - 1) / kLeanModelBatch), t, &four_a);
        const int x4b = wave_exclusive(nt_b * ((np_b + kLeanModelBatch - 1) / kLeanModelBatch), t, &four_b);
        const int xra = wave_exclusive(nt_a * np_a, t, &rec_a), xrb = wave_exclusive(nt_b * np_b, t, &rec_b);
        (void)wave_exclusive(np_a, t, &pair_a);
        (void)wave_exclusive(np_b, t, &pair_b);
        const bool by_two = A.adapt_batch != 0 && two_a + two_b <= 512;
        const int batch = by_two ? 2 : kLeanModelBatch;
        // solution words are kept while the round's (tile, penalty) pairs fit their scratch
        const bool keep = A.cap_pairs > 0 && rec_a + rec_b <= A.cap_pairs;
        const long long round_pairs = (long long)round * A.cap_pairs;
        if (a < B) {
            LeanTask &task = A.tasks[a];
            task.n_points = np_a;
            task.n_groups = (np_a + batch - 1) / batch;
            task.unit_begin = by_two ? x2a : x4a;
            task.rec_begin = xra;
            task.batch = batch;
            task.store = keep ? 2 : 0;
            task.bits_begin = (round_pairs + xra) * (2 * 256);
            task.off_begin = round_pairs + xra;
            sh.st[a].rec_begin = xra;
            sh.st[a].stored = keep ? 1 : 0;
        }
        if (b < B) {
            LeanTask &task = A.tasks[b];
            task.n_points = np_b;
            task.n_groups = (np_b + batch - 1) / batch;
            task.unit_begin = (by_two ? two_a + x2b : four_a + x4b);
            task.rec_begin = rec_a + xrb;
            task.batch = batch;
            task.store = keep ? 2 : 0;
            task.bits_begin = (round_pairs + rec_a + xrb) * (2 * 256);
            task.off_begin = round_pairs + rec_a + xrb;
            sh.st[b].rec_begin = rec_a + xrb;
            sh.st[b].stored = keep ? 1 : 0;
        }
        if (t == 0) {
            LeanRoundCtl c;
            c.n_tasks = B;
            c.n_units = by_two ? two_a + two_b : four_a + four_b;
            c.n_pairs = pair_a + pair_b;
            c.n_pre_tasks = 0;
            c.n_pre_blocks = 0;
            c.error = (round == 0) ? 0u : atomicOr(&A.ctl->error, 0u);
            c.round = round + 1;
            c.all_done = (pair_a + pair_b == 0) ? 1 : 0;
            *A.ctl = c;
        }
    }
    __syncthreads();
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(A.state);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(sh.st);
        for (int i = t; i < B * kStateWords; i += kDirectorThreads) {
            dst[i] = src[i];
        }
    }
}

}  // namespace

int launch_model_chain_director(const ModelChainArgs &A, int round, int last, hipStream_t stream)
{
    if (A.n_problems < 1 || A.n_problems > kModelChainMaxProblems) {
        set_last_error("model chain: problem count out of range");
        return ROCCO_HIP_EINVAL;
    }
    hipLaunchKernelGGL(model_chain_director_kernel, dim3(1), dim3(kDirectorThreads), 0, stream, A, round, last);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
