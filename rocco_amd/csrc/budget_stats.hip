// rocco_amd/csrc/budget_stats.hip -- n-long pieces of the score-track budget estimate (SURVEY.md section 8 (f) item 1;
// rocco/inference.py:1151-1421, 446-501; rocco/rocco.py:751-789), gfx950.
//
// The reference's estimate is a handful of order statistics (np.median of the residual template, the MAD of its
// mirrored non-positive part, the median of the positive scores for the switch cost), means of elementwise functions
// (npsum.hip: summed in NumPy's order) and the autocovariances of one n-long series at up to a few hundred lags.
// Here: a radix sort of the vector (hipcub) from which any order statistic / count below a threshold is read
// off, and the lagged products summed in a fixed order (segments of 4096 loci staged through LDS with their halo, one
// lane per lag; then the segments in order).  The reference gets the autocovariances through an FFT (np.fft.rfft /
// irfft), whose rounding no other summation order reproduces: they agree to ~1e-13 relative, the truncation lag of
// the integrated autocorrelation time (Geyer's positive pairs) exactly unless a pair sum is that close to zero.
// The dependent multipliers of the bootstrap draws stay with NumPy's generator and SciPy's fftconvolve on the host
// (their streams are not reproducible elsewhere).
#include "kernels.h"

#include <hipcub/hipcub.hpp>

namespace rocco {

namespace {

using u64 = unsigned long long;

__global__ __launch_bounds__(256) void to_key_kernel(const double *__restrict__ in, u64 *__restrict__ key, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u64 b = (u64)__double_as_longlong(in[i]);
        key[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ULL);  // order-preserving (negative: all bits flipped)
    }
}

__global__ __launch_bounds__(256) void from_key_kernel(const u64 *__restrict__ key, double *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const u64 k = key[i];
        const u64 b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
        out[i] = __longlong_as_double((long long)b);
    }
}

// values at ranks; per threshold t: how many elements satisfy (x - shift) <= t and (x - shift) < t (x - shift is
// monotone in x, so a binary search on the sorted x finds both)
__global__ __launch_bounds__(64) void sorted_probe_kernel(const double *__restrict__ sorted, long long n, SortedProbe p,
                                                         double *__restrict__ values_out, long long *__restrict__ counts_out)
{
    const int t = threadIdx.x;
    if (t < p.n_ranks) {
        const long long r = p.ranks[t];
        values_out[t] = (r >= 0 && r < n) ? sorted[r] : 0.0;
    }
    if (t < p.n_thresholds) {
        const double thr = p.thresholds[t];
        for (int strict = 0; strict < 2; ++strict) {
            long long lo = 0, hi = n;  // first index whose shifted value is NOT (<= thr / < thr)
            while (lo < hi) {
                const long long mid = lo + (hi - lo) / 2;
                const double v = sorted[mid] - p.shift;
                const bool inside = strict ? (v < thr) : (v <= thr);
                if (inside) {
                    lo = mid + 1;
                } else {
                    hi = mid;
                }
            }
            counts_out[2 * t + strict] = lo;
        }
    }
}

constexpr int kSeg = 4096;       // loci per workgroup
constexpr int kMaxLag = 1023;    // lags 0..kMaxLag

// partial[(block) * (L + 1) + k] = sum over the block's loci i (in order) of c_i * c_{i+k0+k}, c = x - mean, i + k0 + k < n:
// the lags k0 .. k0 + L of one launch (k0 = 0 for the first 1024 lags: both factors then come from one staged segment)
__global__ __launch_bounds__(256) void autocov_partial_kernel(const double *__restrict__ x, long long n, double mean, int k0, int L,
                                                             double *__restrict__ partial)
{
    extern __shared__ double c[];  // k0 == 0: kSeg + L values; else kSeg values, then kSeg + L values from k0 on
    const long long base = (long long)blockIdx.x * kSeg;
    const int first = (k0 == 0) ? (kSeg + L) : kSeg;
    for (int i = threadIdx.x; i < first; i += blockDim.x) {
        const long long j = base + i;
        c[i] = (j < n) ? (x[j] - mean) : 0.0;
    }
    const double *ahead = c;
    if (k0 != 0) {
        for (int i = threadIdx.x; i < kSeg + L; i += blockDim.x) {
            const long long j = base + k0 + i;
            c[kSeg + i] = (j < n) ? (x[j] - mean) : 0.0;
        }
        ahead = c + kSeg;
    }
    __syncthreads();
    const long long here = (n - base < kSeg) ? (n - base) : kSeg;
    for (int k = threadIdx.x; k <= L; k += blockDim.x) {
        double acc = 0.0;
        for (int i = 0; i < here; ++i) {
            if (base + i + k0 + k < n) {
                acc += c[i] * ahead[i + k];
            }
        }
        partial[(long long)blockIdx.x * (L + 1) + k] = acc;
    }
}

__global__ __launch_bounds__(256) void autocov_final_kernel(const double *__restrict__ partial, int n_blocks, int L,
                                                           double *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= L) {
        double acc = 0.0;
        for (int b = 0; b < n_blocks; ++b) {
            acc += partial[(long long)b * (L + 1) + k];
        }
        out[k] = acc;
    }
}

// elementwise helpers of the estimate
__global__ __launch_bounds__(256) void negative_part_kernel(const double *__restrict__ s, double *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double v = s[i];
        out[i] = v - ((v > 0.0) ? v : 0.0);  // scores - np.clip(scores, 0, None)   (inference.py:1171-1172)
    }
}

__global__ __launch_bounds__(256) void soft_count_kernel(const double *__restrict__ s, double center, double scale,
                                                        double *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double r = s[i] - center;
        out[i] = ((r > 0.0) ? r : 0.0) / scale;  // np.clip(scores - null_center, 0, None) / null_soft_scale (1344-1347)
    }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

size_t sort_f64_scratch_bytes(size_t n)
{
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, temp, (const u64 *)nullptr, (u64 *)nullptr, (int)n);
    return 2 * align_up(n * sizeof(u64), 256) + align_up(temp, 256) + 256;
}

int launch_sort_f64(const double *x_dev, size_t n, double *sorted_out_dev, void *scratch_dev, hipStream_t stream)
{
    if (n == 0) {
        return ROCCO_HIP_OK;
    }
    u64 *keys = (u64 *)scratch_dev;
    u64 *keys_out = (u64 *)((char *)scratch_dev + align_up(n * sizeof(u64), 256));
    void *temp = (char *)scratch_dev + 2 * align_up(n * sizeof(u64), 256);
    size_t temp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, (const u64 *)nullptr, (u64 *)nullptr, (int)n);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(to_key_kernel, dim3(blocks), dim3(256), 0, stream, x_dev, keys, (long long)n);
    ROCCO_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, (const u64 *)keys, keys_out, (int)n, 0, 64, stream));
    hipLaunchKernelGGL(from_key_kernel, dim3(blocks), dim3(256), 0, stream, keys_out, sorted_out_dev, (long long)n);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_sorted_probe(const double *sorted_dev, size_t n, const SortedProbe &probe, double *values_out_dev,
                        long long *counts_out_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(sorted_probe_kernel, dim3(1), dim3(64), 0, stream, sorted_dev, (long long)n, probe, values_out_dev,
                       counts_out_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

size_t autocov_scratch_bytes(size_t n, int max_lag)
{
    const size_t blocks = (n + kSeg - 1) / kSeg;
    const size_t lags = (size_t)((max_lag < kMaxLag) ? max_lag : kMaxLag) + 1;  // one chunk of lags at a time
    return blocks * lags * sizeof(double) + 256;
}

int launch_autocov(const double *x_dev, size_t n, double mean, int max_lag, double *sums_out_dev, void *scratch_dev,
                   hipStream_t stream)
{
    if (n == 0 || max_lag < 0) {
        return ROCCO_HIP_EINVAL;
    }
    static bool lds_raised = false;  // lag chunks beyond the first stage two segments: 72 KB of LDS
    if (!lds_raised) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(autocov_partial_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * kSeg + kMaxLag) * sizeof(double))));
        lds_raised = true;
    }
    const int blocks = (int)((n + kSeg - 1) / kSeg);
    for (int k0 = 0; k0 <= max_lag; k0 += kMaxLag + 1) {  // lags in chunks of 1024, one pair of launches each
        const int L = ((max_lag - k0) < kMaxLag) ? (max_lag - k0) : kMaxLag;
        const size_t lds = (size_t)((k0 == 0 ? kSeg : 2 * kSeg) + L) * sizeof(double);
        hipLaunchKernelGGL(autocov_partial_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, x_dev, (long long)n, mean, k0, L,
                           (double *)scratch_dev);
        hipLaunchKernelGGL(autocov_final_kernel, dim3((unsigned)((L + 256) / 256)), dim3(256), 0, stream,
                           (const double *)scratch_dev, blocks, L, sums_out_dev + k0);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_negative_part(const double *scores_dev, double *out_dev, size_t n, hipStream_t stream)
{
    if (n > 0) {
        hipLaunchKernelGGL(negative_part_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, scores_dev, out_dev,
                           (long long)n);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

int launch_soft_counts(const double *scores_dev, double center, double scale, double *out_dev, size_t n, hipStream_t stream)
{
    if (n > 0) {
        hipLaunchKernelGGL(soft_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, scores_dev, center, scale,
                           out_dev, (long long)n);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

}  // namespace rocco
