// rocco_amd/csrc/peakscore.hip -- the per-peak arithmetic of the post-hoc peak scoring (SURVEY.md section 8 (f) item 4;
// rocco/scores.py:180-194, 128-141, 560-583), gfx950.  Counting reads over peaks and over the random background regions
// of the empirical nulls is BAM work and stays with the reference's readers; what follows it is here:
//   * signal value of a peak: the 75th percentile (np.percentile's linear interpolation) over the samples of
//     log2(max(count * row_scale / length + pc, pc)) -- one lane per peak, the two order statistics by rank counting
//     (K samples: tens), the logarithm correctly rounded (log2_cr.h);
//   * p-value: finite-sample right-tail survival against the sorted null of the peak's length bin,
//     (size - lower_bound + 1) / (size + 1);
//   * q-values: Benjamini-Hochberg as scipy.stats.false_discovery_control applies it -- sort, p * (m / rank),
//     running minimum from the largest rank down, back to the input order, clip to [0, 1].
#include "kernels.h"
#include "log2_cr.h"

#include <hipcub/hipcub.hpp>

namespace rocco {

namespace {

// transformed[p][k] = log2(max(counts[p][k] * row_scale / max(int(length[p]), 1) + pc, pc))
__global__ __launch_bounds__(256) void peak_transform_kernel(const double *__restrict__ counts, const double *__restrict__ lengths,
                                                            long long P, int K, double row_scale, double pc,
                                                            double *__restrict__ transformed)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * K) {
        return;
    }
    const long long p = i / K;
    const long long len_i = (long long)lengths[p];
    const double factor = row_scale / (double)((len_i > 1) ? len_i : 1);  // float(row_scale) / float(max(int(length), 1))
    const double t = fmax(counts[i] * factor + pc, pc);
    transformed[i] = (t > 0.0 && t < INFINITY) ? log2_correctly_rounded(t) : log2(t);
}

// np.percentile(row, q), method "linear", of every row of a [P][K] matrix: one lane per row, the two order statistics by
// rank counting
__global__ __launch_bounds__(256) void row_percentile_kernel(const double *__restrict__ x, long long P, int K, double percentile,
                                                            double *__restrict__ out)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) {
        return;
    }
    const double *__restrict__ row = x + p * K;
    // virtual index n q + (1 - q) - 1 in NumPy's own expression (alpha = beta = 1)
    const double q = percentile / 100.0;
    const double vi = (double)K * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
    int prev = (int)floor(vi);
    const double gamma = vi - (double)prev;
    int next = prev + 1;
    prev = min(max(prev, 0), K - 1);
    next = min(max(next, 0), K - 1);
    double a = 0.0, b = 0.0;
    bool has_nan = false;
    for (int i = 0; i < K; ++i) {
        const double v = row[i];
        if (v != v) {
            has_nan = true;
            continue;
        }
        int less = 0, equal = 0;
        for (int j = 0; j < K; ++j) {
            const double y = row[j];
            less += (y < v);
            equal += (y == v);
        }
        if (prev >= less && prev < less + equal) {
            a = v;
        }
        if (next >= less && next < less + equal) {
            b = v;
        }
    }
    const double diff = b - a;  // numpy's _lerp
    double r = a + diff * gamma;
    if (gamma >= 0.5) {
        r = b - diff * (1.0 - gamma);
    }
    out[p] = has_nan ? __longlong_as_double(0x7FF8000000000000LL) : r;
}

__global__ __launch_bounds__(256) void survival_kernel(const double *__restrict__ stat, const int *__restrict__ bin,
                                                      const double *__restrict__ null_values, const long long *__restrict__ null_offsets,
                                                      long long P, double *__restrict__ out)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) {
        return;
    }
    const long long lo0 = null_offsets[bin[p]], hi0 = null_offsets[bin[p] + 1];
    const double x = stat[p];
    long long lo = lo0, hi = hi0;  // np.searchsorted(values, x, side="left")
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        if (null_values[mid] < x) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    const double size = (double)(hi0 - lo0);
    out[p] = (size - (double)(lo - lo0) + 1.0) / (size + 1.0);
}

using u64 = unsigned long long;

__global__ __launch_bounds__(256) void bh_keys_kernel(const double *__restrict__ p, u64 *__restrict__ key, unsigned *__restrict__ idx,
                                                     long long m)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) {
        key[i] = (u64)__double_as_longlong(p[i]);  // p >= 0: the bit patterns order like the numbers
        idx[i] = (unsigned)i;
    }
}

// one workgroup: adjusted[k] = p_sorted[k] * (m / (k + 1)), running minimum from the end, scatter to the input order, clip
__global__ __launch_bounds__(1024) void bh_finish_kernel(const u64 *__restrict__ sorted_key, const unsigned *__restrict__ sorted_idx,
                                                        long long m, double *__restrict__ scratch, double *__restrict__ out)
{
    __shared__ double wave_min[16];
    __shared__ double carry;
    if (threadIdx.x == 0) {
        carry = INFINITY;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long top = m; top > 0; top -= blockDim.x) {
        // thread t takes position top - 1 - t: positions descend with the thread index, so an inclusive minimum scan over
        // the threads is the running minimum from the end
        const long long k = top - 1 - (long long)threadIdx.x;
        double v = INFINITY;
        if (k >= 0) {
            const double pk = __longlong_as_double((long long)sorted_key[k]);
            v = pk * ((double)m / (double)(k + 1));
        }
        double incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double u = __shfl_up(incl, off);
            if (lane >= off) {
                incl = fmin(incl, u);
            }
        }
        if (lane == 63) {
            wave_min[wave] = incl;
        }
        __syncthreads();
        double before = carry;
        for (int w = 0; w < wave; ++w) {
            before = fmin(before, wave_min[w]);
        }
        const double res = fmin(incl, before);
        if (k >= 0) {
            scratch[k] = res;
        }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) {
            carry = res;
        }
        __syncthreads();
    }
    for (long long k = threadIdx.x; k < m; k += blockDim.x) {
        out[sorted_idx[k]] = fmin(fmax(scratch[k], 0.0), 1.0);
    }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

int launch_peak_signal(const double *counts_dev, const double *lengths_dev, size_t P, size_t K, double row_scale, double pc,
                       double percentile, double *out_dev, void *scratch_dev, hipStream_t stream)
{
    if (P > 0) {
        double *transformed = (double *)scratch_dev;  // P * K doubles
        hipLaunchKernelGGL(peak_transform_kernel, dim3((unsigned)((P * K + 255) / 256)), dim3(256), 0, stream, counts_dev, lengths_dev,
                           (long long)P, (int)K, row_scale, pc, transformed);
        hipLaunchKernelGGL(row_percentile_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, transformed, (long long)P,
                           (int)K, percentile, out_dev);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

int launch_ecdf_survival(const double *stat_dev, const int *bin_dev, const double *null_values_dev, const long long *null_offsets_dev,
                         size_t P, double *out_dev, hipStream_t stream)
{
    if (P > 0) {
        hipLaunchKernelGGL(survival_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, stat_dev, bin_dev, null_values_dev,
                           null_offsets_dev, (long long)P, out_dev);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

size_t bh_scratch_bytes(size_t m)
{
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const u64 *)nullptr, (u64 *)nullptr, (const unsigned *)nullptr,
                                             (unsigned *)nullptr, (int)m);
    return 3 * align_up(m * 8, 256) + 2 * align_up(m * 4, 256) + align_up(temp, 256) + 256;
}

int launch_bh_adjust(const double *pvals_dev, size_t m, double *qvals_out_dev, void *scratch_dev, hipStream_t stream)
{
    if (m == 0) {
        return ROCCO_HIP_OK;
    }
    char *sc = (char *)scratch_dev;
    u64 *key_a = (u64 *)sc, *key_b = (u64 *)(sc + align_up(m * 8, 256));
    double *work = (double *)(sc + 2 * align_up(m * 8, 256));
    unsigned *idx_a = (unsigned *)(sc + 3 * align_up(m * 8, 256)), *idx_b = (unsigned *)((char *)idx_a + align_up(m * 4, 256));
    void *temp = (char *)idx_b + align_up(m * 4, 256);
    size_t temp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const u64 *)nullptr, (u64 *)nullptr, (const unsigned *)nullptr,
                                             (unsigned *)nullptr, (int)m);
    hipLaunchKernelGGL(bh_keys_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, pvals_dev, key_a, idx_a, (long long)m);
    ROCCO_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const u64 *)key_a, key_b, (const unsigned *)idx_a, idx_b, (int)m,
                                                     0, 64, stream));
    hipLaunchKernelGGL(bh_finish_kernel, dim3(1), dim3(1024), 0, stream, key_b, idx_b, (long long)m, work, qvals_out_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
