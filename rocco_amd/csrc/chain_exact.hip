// rocco_amd/csrc/chain_exact.hip -- exact emulation of the reference chain solver on gfx950.
//
// One wavefront per chromosome; lane l evaluates selection penalty lambda[l], so up to 64 penalties
// (e.g. six levels of the reference's bisection tree, evaluated speculatively) share one pass over
// the scores.  Every lane executes the reference's floating-point operations in the reference's
// own order -- IEEE double add / subtract / compare only, compiled with -ffp-contract=off -- so
// values, counts and decisions are bit-identical to rocco/_chain_dp.c:109-186 for any input:
//     leave   = on - c                      (_chain_dp.c:120)
//     keep_on = (on + s) - lambda           (_chain_dp.c:125)
//     enter   = ((off - c) + s) - lambda    (_chain_dp.c:127-128)
//     pick larger value, then fewer selected loci, then "stay"   (_chain_dp.c:133-159)
// This kernel is latency-bound by construction (a dependent chain per lane); it is the
// always-correct path behind the certified delta-form kernels in chain_fast.hip.
#include "kernels.h"

namespace rocco {

namespace {

constexpr int kTile = 256;  // loci staged per LDS tile (4 per lane)

struct Path {
    double val;
    int cnt;
};

__device__ __forceinline__ bool beats(double av, int ac, double bv, int bc)
{
    return (av > bv) || (av == bv && ac < bc);
}

}  // namespace

__global__ __launch_bounds__(64) void chain_exact_kernel(const ExactTask *__restrict__ tasks)
{
    const ExactTask task = tasks[blockIdx.x];
    const int lane = threadIdx.x;
    const long long n = task.n;
    const double *__restrict__ s = task.scores;
    const double *__restrict__ cs = task.switch_costs;
    const bool has_costs = (cs != nullptr);
    const double lam = task.lambdas[lane < task.n_lambda ? lane : 0];
    const bool record = (task.decision_words != nullptr) && (lane == task.record_lane);

    __shared__ double lds_s[2][kTile];
    __shared__ double lds_c[2][kTile];

    double off_v = 0.0;
    int off_c = 0;
    double on_v = s[0] - lam;  // _chain_dp.c:111
    int on_c = 1;

    unsigned long long word = 0ULL;
    long long words_written = 0;

    // prefetch the first tile (loci 1 .. kTile) into registers
    double rs[kTile / 64];
    double rc[kTile / 64];
#pragma unroll
    for (int k = 0; k < kTile / 64; ++k) {
        const long long i = 1 + k * 64 + lane;
        rs[k] = (i < n) ? s[i] : 0.0;
        rc[k] = (has_costs && i < n) ? cs[i - 1] : task.gamma;
    }

    int buf = 0;
    for (long long base = 1; base < n; base += kTile) {
#pragma unroll
        for (int k = 0; k < kTile / 64; ++k) {
            lds_s[buf][k * 64 + lane] = rs[k];
            lds_c[buf][k * 64 + lane] = rc[k];
        }
        __syncthreads();
        // issue the next tile's global loads; they complete while this tile is processed
        const long long next = base + kTile;
#pragma unroll
        for (int k = 0; k < kTile / 64; ++k) {
            const long long i = next + k * 64 + lane;
            rs[k] = (i < n) ? s[i] : 0.0;
            rc[k] = (has_costs && i < n) ? cs[i - 1] : task.gamma;
        }
        const int steps = (int)((n - base < kTile) ? (n - base) : kTile);
#pragma unroll 4
        for (int t = 0; t < steps; ++t) {
            const double sv = lds_s[buf][t];
            const double c = lds_c[buf][t];
            const double leave = on_v - c;
            const double keep = on_v + sv - lam;
            const double enter = off_v - c + sv - lam;
            const bool take_leave = beats(leave, on_c, off_v, off_c);
            const bool take_enter = beats(enter, off_c, keep, on_c);  // counts both +1
            const double n_off_v = take_leave ? leave : off_v;
            const int n_off_c = take_leave ? on_c : off_c;
            const double n_on_v = take_enter ? enter : keep;
            const int n_on_c = (take_enter ? off_c : on_c) + 1;
            if (record) {
                const unsigned bits = (unsigned)take_leave | ((unsigned)(!take_enter) << 1);
                const int slot = (int)((base + t - 1) & 31);
                word |= (unsigned long long)bits << (slot * 2);
                if (slot == 31) {
                    task.decision_words[words_written++] = word;
                    word = 0ULL;
                }
            }
            off_v = n_off_v;
            off_c = n_off_c;
            on_v = n_on_v;
            on_c = n_on_c;
        }
        buf ^= 1;
    }

    const bool end_on = beats(on_v, on_c, off_v, off_c);  // _chain_dp.c:167-179
    if (lane < task.n_lambda) {
        task.values_out[lane] = end_on ? on_v : off_v;
        task.counts_out[lane] = (long long)(end_on ? on_c : off_c);
    }

    if (record) {
        if (((n - 1) & 31) != 0) {
            task.decision_words[words_written++] = word;
        }
        // backtrack (_chain_dp.c:181-186): same lane reads back the words it wrote
        uint8_t *__restrict__ z = task.solution;
        int state = end_on ? 1 : 0;
        z[n - 1] = (uint8_t)state;
        for (long long i = n - 1; i > 0; --i) {
            const unsigned long long w = task.decision_words[(i - 1) >> 5];
            const unsigned bits = (unsigned)(w >> (((i - 1) & 31) * 2)) & 3U;
            state = (state == 0) ? (int)(bits & 1U) : (int)((bits >> 1) & 1U);
            z[i - 1] = (uint8_t)state;
        }
    }
}

int launch_chain_exact(const ExactTask *tasks_dev, int n_tasks, hipStream_t stream)
{
    if (n_tasks <= 0) {
        return ROCCO_HIP_OK;
    }
    hipLaunchKernelGGL(chain_exact_kernel, dim3(n_tasks), dim3(64), 0, stream, tasks_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
