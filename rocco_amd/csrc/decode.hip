// rocco_amd/csrc/decode.hip -- 0/1 solution vector -> ordered maximal runs (merged BED3 intervals).
//
// Replaces the per-locus Python loop of rocco/rocco.py:180-186 plus the adjacency merge of
// rocco/rocco.py:74-95: loci 0..n-2 with solution > 0.5 (non-zero bytes here) are selected, the
// last locus is never emitted (rocco.py:180), touching records merge.  For contiguous loci that
// is exactly "maximal runs of selected loci among 0..n-2".
//
// Three small launches, all integer/byte work bound by HBM reads of n bytes:
//   count  : per 4096-locus tile, number of run starts and run ends (16 loci per lane as one 16-B load)
//   scan   : exclusive scan of the per-tile counts (one workgroup)
//   write  : ordered compaction of run begin / end indices
#include "kernels.h"

namespace rocco {

namespace {

constexpr int kThreads = 256;
constexpr int kPerLane = 16;
constexpr int kTileLoci = kThreads * kPerLane;

struct LaneBits {
    unsigned starts;
    unsigned ends;
};

// Selected-mask of the lane's 16 loci and the run start / end bits among them.
__device__ __forceinline__ LaneBits lane_bits(const uint8_t *__restrict__ z, long long n, long long i0)
{
    const long long limit = n - 1;  // loci >= n-1 are never emitted
    unsigned mask = 0;
    if (i0 + kPerLane <= n && ((reinterpret_cast<uintptr_t>(z + i0) & 15U) == 0)) {
        const uint4 w = *reinterpret_cast<const uint4 *>(z + i0);
        const unsigned words[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                mask |= (((words[q] >> (8 * b)) & 0xFFU) != 0U ? 1U : 0U) << (q * 4 + b);
            }
        }
    } else {
        for (int t = 0; t < kPerLane; ++t) {
            if (i0 + t < n && z[i0 + t] != 0) {
                mask |= 1U << t;
            }
        }
    }
    // drop loci >= n-1
    if (i0 + kPerLane > limit) {
        const long long keep = limit - i0;  // may be <= 0
        mask = (keep <= 0) ? 0U : (mask & ((1U << keep) - 1U));
    }
    const unsigned prev = (i0 > 0 && i0 - 1 < limit && z[i0 - 1] != 0) ? 1U : 0U;
    const unsigned next = (i0 + kPerLane < limit && z[i0 + kPerLane] != 0) ? 1U : 0U;
    LaneBits r;
    r.starts = mask & ~((mask << 1) | prev) & 0xFFFFU;
    r.ends = mask & ~((mask >> 1) | (next << (kPerLane - 1))) & 0xFFFFU;
    return r;
}

__global__ __launch_bounds__(kThreads) void decode_count_kernel(const uint8_t *__restrict__ z,
                                                               long long n,
                                                               unsigned *__restrict__ tile_counts)
{
    const long long i0 = ((long long)blockIdx.x * kThreads + threadIdx.x) * kPerLane;
    unsigned packed = 0;  // starts in low 16 bits, ends in high 16 bits (<= 16 each per lane)
    if (i0 < n) {
        const LaneBits b = lane_bits(z, n, i0);
        packed = (unsigned)__popc(b.starts) | ((unsigned)__popc(b.ends) << 16);
    }
    // block reduction (counts <= 4096 fit in 16 bits... use two 32-bit sums to be safe)
    __shared__ unsigned s_starts[kThreads / 64];
    __shared__ unsigned s_ends[kThreads / 64];
    unsigned st = packed & 0xFFFFU;
    unsigned en = packed >> 16;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        st += __shfl_down(st, off);
        en += __shfl_down(en, off);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_starts[wave] = st;
        s_ends[wave] = en;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned a = 0, b = 0;
        for (int w = 0; w < kThreads / 64; ++w) {
            a += s_starts[w];
            b += s_ends[w];
        }
        tile_counts[2 * blockIdx.x] = a;
        tile_counts[2 * blockIdx.x + 1] = b;
    }
}

// One workgroup: exclusive scan of (starts, ends) per tile -> tile_offsets; totals -> totals[0..1].
__global__ __launch_bounds__(1024) void decode_scan_kernel(const unsigned *__restrict__ tile_counts,
                                                           long long n_tiles,
                                                           unsigned long long *__restrict__ tile_offsets,
                                                           unsigned long long *__restrict__ totals)
{
    __shared__ unsigned long long carry[2];
    __shared__ unsigned long long wave_sums[2][16];
    if (threadIdx.x == 0) {
        carry[0] = 0;
        carry[1] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (long long base = 0; base < n_tiles; base += blockDim.x) {
        const long long t = base + threadIdx.x;
        unsigned long long v0 = (t < n_tiles) ? tile_counts[2 * t] : 0ULL;
        unsigned long long v1 = (t < n_tiles) ? tile_counts[2 * t + 1] : 0ULL;
        unsigned long long i0 = v0, i1 = v1;  // inclusive scans within the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long a = __shfl_up(i0, off);
            const unsigned long long b = __shfl_up(i1, off);
            if (lane >= off) {
                i0 += a;
                i1 += b;
            }
        }
        if (lane == 63) {
            wave_sums[0][wave] = i0;
            wave_sums[1][wave] = i1;
        }
        __syncthreads();
        unsigned long long p0 = carry[0], p1 = carry[1];
        for (int w = 0; w < wave; ++w) {
            p0 += wave_sums[0][w];
            p1 += wave_sums[1][w];
        }
        if (t < n_tiles) {
            tile_offsets[2 * t] = p0 + i0 - v0;
            tile_offsets[2 * t + 1] = p1 + i1 - v1;
        }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) {
            carry[0] = p0 + i0;
            carry[1] = p1 + i1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = carry[0];
        totals[1] = carry[1];
    }
}

__global__ __launch_bounds__(kThreads) void decode_write_kernel(
    const uint8_t *__restrict__ z, long long n, const unsigned long long *__restrict__ tile_offsets,
    int64_t *__restrict__ run_begin, int64_t *__restrict__ run_end, unsigned long long capacity)
{
    const long long i0 = ((long long)blockIdx.x * kThreads + threadIdx.x) * kPerLane;
    LaneBits b = {0U, 0U};
    if (i0 < n) {
        b = lane_bits(z, n, i0);
    }
    const unsigned ns = (unsigned)__popc(b.starts);
    const unsigned ne = (unsigned)__popc(b.ends);
    // exclusive scan across the workgroup
    unsigned is = ns, ie = ne;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned a = __shfl_up(is, off);
        const unsigned c = __shfl_up(ie, off);
        if (lane >= off) {
            is += a;
            ie += c;
        }
    }
    __shared__ unsigned ws[kThreads / 64], we[kThreads / 64];
    if (lane == 63) {
        ws[wave] = is;
        we[wave] = ie;
    }
    __syncthreads();
    unsigned long long ps = tile_offsets[2 * blockIdx.x];
    unsigned long long pe = tile_offsets[2 * blockIdx.x + 1];
    for (int w = 0; w < wave; ++w) {
        ps += ws[w];
        pe += we[w];
    }
    ps += is - ns;
    pe += ie - ne;
    unsigned sb = b.starts;
    while (sb) {
        const int t = __ffs(sb) - 1;
        sb &= sb - 1;
        if (ps < capacity) {
            run_begin[ps] = (int64_t)(i0 + t);
        }
        ++ps;
    }
    unsigned eb = b.ends;
    while (eb) {
        const int t = __ffs(eb) - 1;
        eb &= eb - 1;
        if (pe < capacity) {
            run_end[pe] = (int64_t)(i0 + t + 1);
        }
        ++pe;
    }
}

// ---- several solutions in three launches (the chromosomes of a group): same per-tile work, a task table in
// the kernel arguments, one scan workgroup per task ----
__global__ __launch_bounds__(kThreads) void decode_count_batch_kernel(DecodeBatch batch, unsigned *__restrict__ tile_counts)
{
    int ti = 0;
    while (ti + 1 < batch.n_tasks && batch.tasks[ti + 1].tile_begin <= (long long)blockIdx.x) {
        ++ti;
    }
    const DecodeTask &task = batch.tasks[ti];
    const long long tile = (long long)blockIdx.x - task.tile_begin;
    const long long i0 = (tile * kThreads + threadIdx.x) * kPerLane;
    unsigned st = 0, en = 0;
    if (i0 < task.n) {
        const LaneBits b = lane_bits(task.solution, task.n, i0);
        st = (unsigned)__popc(b.starts);
        en = (unsigned)__popc(b.ends);
    }
    __shared__ unsigned s_starts[kThreads / 64];
    __shared__ unsigned s_ends[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        st += __shfl_down(st, off);
        en += __shfl_down(en, off);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_starts[wave] = st;
        s_ends[wave] = en;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned a = 0, b = 0;
        for (int w = 0; w < kThreads / 64; ++w) {
            a += s_starts[w];
            b += s_ends[w];
        }
        tile_counts[2 * blockIdx.x] = a;
        tile_counts[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(1024) void decode_scan_batch_kernel(DecodeBatch batch, const unsigned *__restrict__ tile_counts,
                                                                 unsigned long long *__restrict__ tile_offsets,
                                                                 unsigned long long *__restrict__ totals)
{
    const DecodeTask &task = batch.tasks[blockIdx.x];
    const long long n_tiles = (task.n + kTileLoci - 1) / kTileLoci;
    const unsigned *counts = tile_counts + 2 * task.tile_begin;
    unsigned long long *offsets = tile_offsets + 2 * task.tile_begin;
    __shared__ unsigned long long carry[2];
    __shared__ unsigned long long wave_sums[2][16];
    if (threadIdx.x == 0) {
        carry[0] = 0;
        carry[1] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (long long base = 0; base < n_tiles; base += blockDim.x) {
        const long long t = base + threadIdx.x;
        unsigned long long v0 = (t < n_tiles) ? counts[2 * t] : 0ULL;
        unsigned long long v1 = (t < n_tiles) ? counts[2 * t + 1] : 0ULL;
        unsigned long long i0 = v0, i1 = v1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long a = __shfl_up(i0, off);
            const unsigned long long b = __shfl_up(i1, off);
            if (lane >= off) {
                i0 += a;
                i1 += b;
            }
        }
        if (lane == 63) {
            wave_sums[0][wave] = i0;
            wave_sums[1][wave] = i1;
        }
        __syncthreads();
        unsigned long long p0 = carry[0], p1 = carry[1];
        for (int w = 0; w < wave; ++w) {
            p0 += wave_sums[0][w];
            p1 += wave_sums[1][w];
        }
        if (t < n_tiles) {
            offsets[2 * t] = p0 + i0 - v0;
            offsets[2 * t + 1] = p1 + i1 - v1;
        }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) {
            carry[0] = p0 + i0;
            carry[1] = p1 + i1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[2 * blockIdx.x] = carry[0];
        totals[2 * blockIdx.x + 1] = carry[1];
    }
}

__global__ __launch_bounds__(kThreads) void decode_write_batch_kernel(DecodeBatch batch,
                                                                     const unsigned long long *__restrict__ tile_offsets)
{
    int ti = 0;
    while (ti + 1 < batch.n_tasks && batch.tasks[ti + 1].tile_begin <= (long long)blockIdx.x) {
        ++ti;
    }
    const DecodeTask &task = batch.tasks[ti];
    const long long tile = (long long)blockIdx.x - task.tile_begin;
    const long long i0 = (tile * kThreads + threadIdx.x) * kPerLane;
    LaneBits b = {0U, 0U};
    if (i0 < task.n) {
        b = lane_bits(task.solution, task.n, i0);
    }
    const unsigned ns = (unsigned)__popc(b.starts);
    const unsigned ne = (unsigned)__popc(b.ends);
    unsigned is = ns, ie = ne;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned a = __shfl_up(is, off);
        const unsigned c = __shfl_up(ie, off);
        if (lane >= off) {
            is += a;
            ie += c;
        }
    }
    __shared__ unsigned ws[kThreads / 64], we[kThreads / 64];
    if (lane == 63) {
        ws[wave] = is;
        we[wave] = ie;
    }
    __syncthreads();
    unsigned long long ps = tile_offsets[2 * blockIdx.x];
    unsigned long long pe = tile_offsets[2 * blockIdx.x + 1];
    for (int w = 0; w < wave; ++w) {
        ps += ws[w];
        pe += we[w];
    }
    ps += is - ns;
    pe += ie - ne;
    unsigned sb = b.starts;
    while (sb) {
        const int t = __ffs(sb) - 1;
        sb &= sb - 1;
        if (ps < task.capacity) {
            task.run_begin[ps] = (int64_t)(i0 + t);
        }
        ++ps;
    }
    unsigned eb = b.ends;
    while (eb) {
        const int t = __ffs(eb) - 1;
        eb &= eb - 1;
        if (pe < task.capacity) {
            task.run_end[pe] = (int64_t)(i0 + t + 1);
        }
        ++pe;
    }
}

// Table form of the write: every task's runs as rows (unit, begin, end) of ONE table, task after task.  A task's first
// row is the number of runs of the tasks before it (the scan kernel's per-task totals: at most kDecodeBatchMax
// independent loads); inside the task the tile offsets are the scan's.
__global__ __launch_bounds__(kThreads) void decode_write_table_kernel(DecodeBatch batch,
                                                                     const unsigned long long *__restrict__ tile_offsets,
                                                                     const unsigned long long *__restrict__ totals)
{
    int ti = 0;
    while (ti + 1 < batch.n_tasks && batch.tasks[ti + 1].tile_begin <= (long long)blockIdx.x) {
        ++ti;
    }
    const DecodeTask &task = batch.tasks[ti];
    unsigned long long row0 = 0;
    for (int q = 0; q < ti; ++q) {
        row0 += totals[2 * q];
    }
    const long long tile = (long long)blockIdx.x - task.tile_begin;
    const long long i0 = (tile * kThreads + threadIdx.x) * kPerLane;
    LaneBits b = {0U, 0U};
    if (i0 < task.n) {
        b = lane_bits(task.solution, task.n, i0);
    }
    const unsigned ns = (unsigned)__popc(b.starts);
    const unsigned ne = (unsigned)__popc(b.ends);
    unsigned is = ns, ie = ne;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned a = __shfl_up(is, off);
        const unsigned c = __shfl_up(ie, off);
        if (lane >= off) {
            is += a;
            ie += c;
        }
    }
    __shared__ unsigned ws[kThreads / 64], we[kThreads / 64];
    if (lane == 63) {
        ws[wave] = is;
        we[wave] = ie;
    }
    __syncthreads();
    unsigned long long ps = row0 + tile_offsets[2 * blockIdx.x];
    unsigned long long pe = row0 + tile_offsets[2 * blockIdx.x + 1];
    for (int w = 0; w < wave; ++w) {
        ps += ws[w];
        pe += we[w];
    }
    ps += is - ns;
    pe += ie - ne;
    int64_t *table = task.run_begin;
    unsigned sb = b.starts;
    while (sb) {
        const int t = __ffs(sb) - 1;
        sb &= sb - 1;
        if (ps < task.capacity) {
            table[3 * ps] = (int64_t)task.unit;
            table[3 * ps + 1] = (int64_t)(i0 + t);
        }
        ++ps;
    }
    unsigned eb = b.ends;
    while (eb) {
        const int t = __ffs(eb) - 1;
        eb &= eb - 1;
        if (pe < task.capacity) {
            table[3 * pe + 2] = (int64_t)(i0 + t + 1);
        }
        ++pe;
    }
}

// the rows that exist -- their number is on the device -- from the table to pinned host memory, 16 bytes per lane and contiguous
// (streaming writes over the host link; a copy sized on the host would have to guess the number or wait for it)
__global__ __launch_bounds__(256) void decode_table_to_host_kernel(const int64_t *__restrict__ table_dev, int64_t *__restrict__ table_host,
                                                                  const unsigned long long *__restrict__ totals, int n_tasks,
                                                                  unsigned long long capacity)
{
    unsigned long long rows = 0;
    for (int q = 0; q < n_tasks; ++q) {
        rows += totals[2 * q];
    }
    if (rows > capacity) {
        rows = capacity;
    }
    const unsigned long long pairs = (rows * 3ULL + 1ULL) / 2ULL;  // 16-byte pieces (an odd tail moves one word of padding too)
    const unsigned long long words = rows * 3ULL;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (unsigned long long)gridDim.x * blockDim.x) {
        const int64_t a = table_dev[2 * i];
        table_host[2 * i] = a;
        if (2 * i + 1 < words) {
            table_host[2 * i + 1] = table_dev[2 * i + 1];
        }
    }
}

}  // namespace

long long decode_tiles(size_t n) { return (long long)((n + kTileLoci - 1) / kTileLoci); }

size_t decode_batch_scratch_bytes(long long total_tiles, int n_tasks)
{
    return (size_t)total_tiles * (2 * sizeof(unsigned) + 2 * sizeof(unsigned long long)) + (size_t)n_tasks * 16 + 256;
}

int launch_decode_runs_batch(const DecodeBatch &batch, long long total_tiles, void *scratch_dev,
                             unsigned long long *totals_host_pinned, hipStream_t stream)
{
    if (batch.n_tasks <= 0) {
        return ROCCO_HIP_OK;
    }
    char *base = (char *)scratch_dev;
    unsigned long long *totals = (unsigned long long *)base;
    unsigned long long *offsets = (unsigned long long *)(base + (((size_t)batch.n_tasks * 16 + 255) / 256) * 256);
    unsigned *counts = (unsigned *)((char *)offsets + (size_t)total_tiles * 2 * sizeof(unsigned long long));
    if (total_tiles > 0) {
        hipLaunchKernelGGL(decode_count_batch_kernel, dim3((unsigned)total_tiles), dim3(kThreads), 0, stream, batch, counts);
    }
    hipLaunchKernelGGL(decode_scan_batch_kernel, dim3((unsigned)batch.n_tasks), dim3(1024), 0, stream, batch, counts, offsets,
                       totals);
    if (total_tiles > 0) {
        hipLaunchKernelGGL(decode_write_batch_kernel, dim3((unsigned)total_tiles), dim3(kThreads), 0, stream, batch, offsets);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipMemcpyAsync(totals_host_pinned, totals, (size_t)batch.n_tasks * 2 * sizeof(unsigned long long),
                                 hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    return ROCCO_HIP_OK;
}

int launch_decode_runs_table(const DecodeBatch &batch, long long total_tiles, void *scratch_dev, int64_t *table_dev,
                             unsigned long long *totals_host_pinned, int64_t *table_host_pinned, size_t eager_rows,
                             hipStream_t stream)
{
    if (batch.n_tasks <= 0) {
        return ROCCO_HIP_OK;
    }
    char *base = (char *)scratch_dev;
    unsigned long long *totals = (unsigned long long *)base;
    unsigned long long *offsets = (unsigned long long *)(base + (((size_t)batch.n_tasks * 16 + 255) / 256) * 256);
    unsigned *counts = (unsigned *)((char *)offsets + (size_t)total_tiles * 2 * sizeof(unsigned long long));
    if (total_tiles > 0) {
        hipLaunchKernelGGL(decode_count_batch_kernel, dim3((unsigned)total_tiles), dim3(kThreads), 0, stream, batch, counts);
    }
    hipLaunchKernelGGL(decode_scan_batch_kernel, dim3((unsigned)batch.n_tasks), dim3(1024), 0, stream, batch, counts, offsets,
                       totals);
    if (total_tiles > 0) {
        hipLaunchKernelGGL(decode_write_table_kernel, dim3((unsigned)total_tiles), dim3(kThreads), 0, stream, batch, offsets,
                           (const unsigned long long *)totals);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipMemcpyAsync(totals_host_pinned, totals, (size_t)batch.n_tasks * 2 * sizeof(unsigned long long),
                                 hipMemcpyDeviceToHost, stream));
    if (table_host_pinned != nullptr && eager_rows > 0) {  // (eager_rows: the rows the host buffer has room for)
        hipLaunchKernelGGL(decode_table_to_host_kernel, dim3(256), dim3(256), 0, stream, (const int64_t *)table_dev, table_host_pinned,
                           (const unsigned long long *)totals, batch.n_tasks, (unsigned long long)eager_rows);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    return ROCCO_HIP_OK;
}

size_t decode_scratch_bytes(size_t n)
{
    const size_t tiles = (n + kTileLoci - 1) / kTileLoci;
    // counts (2 x u32) + offsets (2 x u64) per tile + totals (2 x u64)
    return tiles * (2 * sizeof(unsigned) + 2 * sizeof(unsigned long long)) + 64;
}

int launch_decode_runs(const uint8_t *solution_dev, size_t n, int64_t *run_begin_dev,
                       int64_t *run_end_dev, size_t capacity, void *scratch_dev,
                       unsigned long long *n_runs_host_pinned, hipStream_t stream)
{
    if (n <= 1) {
        *n_runs_host_pinned = 0;
        return ROCCO_HIP_OK;
    }
    const long long tiles = (long long)((n + kTileLoci - 1) / kTileLoci);
    char *base = (char *)scratch_dev;
    unsigned long long *totals = (unsigned long long *)base;
    unsigned long long *offsets = (unsigned long long *)(base + 64);
    unsigned *counts = (unsigned *)(base + 64 + (size_t)tiles * 2 * sizeof(unsigned long long));
    hipLaunchKernelGGL(decode_count_kernel, dim3((unsigned)tiles), dim3(kThreads), 0, stream,
                       solution_dev, (long long)n, counts);
    hipLaunchKernelGGL(decode_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, tiles, offsets,
                       totals);
    hipLaunchKernelGGL(decode_write_kernel, dim3((unsigned)tiles), dim3(kThreads), 0, stream,
                       solution_dev, (long long)n, offsets, run_begin_dev, run_end_dev,
                       (unsigned long long)capacity);
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipMemcpyAsync(n_runs_host_pinned, totals, sizeof(unsigned long long),
                                 hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    return ROCCO_HIP_OK;
}

}  // namespace rocco
