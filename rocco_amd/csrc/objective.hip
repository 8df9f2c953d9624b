// rocco_amd/csrc/objective.hip -- objective_value(solution, scores, switch_costs)  (rocco/dp.py:16-34)
//
//   -(scores . solution) + switch_costs . |diff(solution)|
//
// Two launches with a fixed summation tree (per-lane serial, wave shuffle, per-tile partial, one
// final workgroup), so the result is deterministic run to run.  The reference uses BLAS dot whose
// order is implementation-defined, hence parity is to a relative tolerance, never bitwise.
#include "kernels.h"

namespace rocco {

namespace {

constexpr int kThreads = 256;
constexpr int kPerLane = 8;
constexpr int kTile = kThreads * kPerLane;

__device__ __forceinline__ double block_sum(double v, double *smem)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off);
    }
    if ((threadIdx.x & 63) == 0) {
        smem[threadIdx.x >> 6] = v;
    }
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0) {
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            total += smem[w];
        }
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(kThreads) void objective_partial_kernel(
    const uint8_t *__restrict__ z, const double *__restrict__ s, const double *__restrict__ cs,
    double gamma, long long n, double *__restrict__ partial)
{
    __shared__ double smem[kThreads / 64];
    const long long i0 = (long long)blockIdx.x * kTile + (long long)threadIdx.x * kPerLane;
    double gain = 0.0, pen = 0.0;
#pragma unroll
    for (int t = 0; t < kPerLane; ++t) {
        const long long i = i0 + t;
        if (i < n) {
            const double zi = (z[i] != 0) ? 1.0 : 0.0;
            gain += s[i] * zi;
            if (i + 1 < n) {
                const double zn = (z[i + 1] != 0) ? 1.0 : 0.0;
                const double c = (cs != nullptr) ? cs[i] : gamma;
                pen += c * fabs(zn - zi);
            }
        }
    }
    const double g = block_sum(gain, smem);
    const double p = block_sum(pen, smem);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = g;
        partial[2 * blockIdx.x + 1] = p;
    }
}

__global__ __launch_bounds__(kThreads) void objective_final_kernel(const double *__restrict__ partial,
                                                                  long long n_tiles,
                                                                  double *__restrict__ out)
{
    __shared__ double smem[kThreads / 64];
    double gain = 0.0, pen = 0.0;
    for (long long t = threadIdx.x; t < n_tiles; t += kThreads) {
        gain += partial[2 * t];
        pen += partial[2 * t + 1];
    }
    const double g = block_sum(gain, smem);
    const double p = block_sum(pen, smem);
    if (threadIdx.x == 0) {
        out[0] = -g + p;
    }
}

// Several problems in two launches: the tiles of every task side by side, then one workgroup per task.  Each
// task's sum follows the same fixed tree as the single-problem launch (bit-identical results).
__global__ __launch_bounds__(kThreads) void objective_partial_batch_kernel(const ObjectiveTask *__restrict__ tasks, int n_tasks,
                                                                          double *__restrict__ partial)
{
    __shared__ double smem[kThreads / 64];
    int ti = 0;
    while (ti + 1 < n_tasks && tasks[ti + 1].tile_begin <= (long long)blockIdx.x) {
        ++ti;
    }
    const ObjectiveTask task = tasks[ti];
    const long long tile = (long long)blockIdx.x - task.tile_begin;
    const uint8_t *__restrict__ z = task.solution;
    const double *__restrict__ s = task.scores;
    const double *__restrict__ cs = task.switch_costs;
    const long long n = task.n;
    const long long i0 = tile * kTile + (long long)threadIdx.x * kPerLane;
    double gain = 0.0, pen = 0.0;
#pragma unroll
    for (int t = 0; t < kPerLane; ++t) {
        const long long i = i0 + t;
        if (i < n) {
            const double zi = (z[i] != 0) ? 1.0 : 0.0;
            gain += s[i] * zi;
            if (i + 1 < n) {
                const double zn = (z[i + 1] != 0) ? 1.0 : 0.0;
                const double c = (cs != nullptr) ? cs[i] : task.gamma;
                pen += c * fabs(zn - zi);
            }
        }
    }
    const double g = block_sum(gain, smem);
    const double p = block_sum(pen, smem);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = g;
        partial[2 * blockIdx.x + 1] = p;
    }
}

__global__ __launch_bounds__(kThreads) void objective_final_batch_kernel(const ObjectiveTask *__restrict__ tasks,
                                                                        const double *__restrict__ partial,
                                                                        double *__restrict__ out)
{
    __shared__ double smem[kThreads / 64];
    const ObjectiveTask task = tasks[blockIdx.x];
    const long long n_tiles = (task.n + kTile - 1) / kTile;
    const double *__restrict__ mine = partial + 2 * task.tile_begin;
    double gain = 0.0, pen = 0.0;
    for (long long t = threadIdx.x; t < n_tiles; t += kThreads) {
        gain += mine[2 * t];
        pen += mine[2 * t + 1];
    }
    const double g = block_sum(gain, smem);
    const double p = block_sum(pen, smem);
    if (threadIdx.x == 0) {
        out[blockIdx.x] = -g + p;
    }
}

}  // namespace

long long objective_tiles(size_t n) { return (long long)((n + kTile - 1) / kTile); }

int launch_objective_batch(const ObjectiveTask *tasks_dev, int n_tasks, long long total_tiles, double *partial_dev,
                           double *out_dev, hipStream_t stream)
{
    if (n_tasks <= 0) {
        return ROCCO_HIP_OK;
    }
    if (total_tiles > 0) {
        hipLaunchKernelGGL(objective_partial_batch_kernel, dim3((unsigned)total_tiles), dim3(kThreads), 0, stream, tasks_dev,
                           n_tasks, partial_dev);
    }
    hipLaunchKernelGGL(objective_final_batch_kernel, dim3((unsigned)n_tasks), dim3(kThreads), 0, stream, tasks_dev,
                       partial_dev, out_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

size_t objective_scratch_bytes(size_t n)
{
    const size_t tiles = (n + kTile - 1) / kTile;
    return (2 * tiles + 2) * sizeof(double);
}

int launch_objective(const uint8_t *solution_dev, const double *scores_dev,
                     const double *switch_costs_dev, double gamma, size_t n, void *scratch_dev,
                     double *objective_host_pinned, hipStream_t stream, bool synchronize)
{
    if (n == 0) {
        *objective_host_pinned = 0.0;
        return ROCCO_HIP_OK;
    }
    const long long tiles = (long long)((n + kTile - 1) / kTile);
    double *out = (double *)scratch_dev;
    double *partial = out + 2;
    hipLaunchKernelGGL(objective_partial_kernel, dim3((unsigned)tiles), dim3(kThreads), 0, stream,
                       solution_dev, scores_dev, switch_costs_dev, gamma, (long long)n, partial);
    hipLaunchKernelGGL(objective_final_kernel, dim3(1), dim3(kThreads), 0, stream, partial, tiles, out);
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipMemcpyAsync(objective_host_pinned, out, sizeof(double), hipMemcpyDeviceToHost,
                                 stream));
    if (synchronize) {
        ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    }
    return ROCCO_HIP_OK;
}

}  // namespace rocco
