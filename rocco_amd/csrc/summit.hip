// rocco_amd/csrc/summit.hip -- narrowPeak summit offsets of the final peaks, gfx950.
//
// Replaces the per-peak NumPy statements of rocco/rocco.py:838-872 (`_write_narrowpeak_summit_offsets`) over the
// track of rocco/rocco.py:809-835 (`_cpy_narrowpeak_summit_track`): for every peak [start, end) in base pairs the
// loci whose start lies in [start, end) are found by binary search in the locus starts, the first maximum of the
// float32-rounded effect mean among them (NaNs skipped, np.nanargmax) gives the summit locus, and the offset of
// that locus' centre from the peak start, clipped to the peak, is written (-1: empty peak, no locus, or no finite
// value).  Integer and compare work only: results equal the reference's exactly.
// One wavefront per peak: the lanes stride over the peak's loci and keep (value, first index) of their maximum,
// then the wavefront reduces with the same tie rule.
#include "kernels.h"

namespace rocco {

namespace {

constexpr int kLanes = 64;

__device__ __forceinline__ long long lower_bound_ll(const long long *__restrict__ a, long long n, long long key)
{
    // np.searchsorted(a, key, side="left")
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        if (a[mid] < key) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    return lo;
}

__global__ __launch_bounds__(256) void summit_offsets_kernel(const long long *__restrict__ intervals, long long usable,
                                                            const long long *__restrict__ centers,
                                                            const double *__restrict__ effect_mean,
                                                            const long long *__restrict__ peak_start,
                                                            const long long *__restrict__ peak_end, long long n_peaks,
                                                            long long *__restrict__ offsets)
{
    const long long peak = (long long)blockIdx.x * (blockDim.x / kLanes) + threadIdx.x / kLanes;
    const int lane = threadIdx.x % kLanes;
    if (peak >= n_peaks) {
        return;
    }
    const long long start = peak_start[peak], end = peak_end[peak];
    const long long length = end - start;
    long long result = -1;
    if (length > 0 && usable > 0) {
        const long long left = lower_bound_ll(intervals, usable, start), right = lower_bound_ll(intervals, usable, end);
        // (value, index) of the first maximum; NaN is treated as -inf (np.nanargmax), `finite` as np.any(np.isfinite)
        float best = -INFINITY;
        long long best_i = 0x7fffffffffffffffLL;
        bool finite = false;
        for (long long j = left + lane; j < right; j += kLanes) {
            const float v = (float)effect_mean[j];  // the track is stored as float32 (rocco.py:814)
            finite = finite || isfinite(v);
            const float w = isnan(v) ? -INFINITY : v;
            if (w > best || best_i == 0x7fffffffffffffffLL) {
                best = w;
                best_i = j;
            }
        }
        for (int off = kLanes / 2; off > 0; off >>= 1) {
            const float ob = __shfl_down(best, off);
            const long long oi = __shfl_down(best_i, off);
            const bool of = __shfl_down((int)finite, off) != 0;
            finite = finite || of;
            if (oi != 0x7fffffffffffffffLL && (best_i == 0x7fffffffffffffffLL || ob > best || (ob == best && oi < best_i))) {
                best = ob;
                best_i = oi;
            }
        }
        if (lane == 0 && right > left && finite) {
            // rocco.py:819-822: (start of the locus + start of the next) // 2 (>> 1 is floor division)
            const long long centre = (centers != nullptr) ? centers[best_i] : ((intervals[best_i] + intervals[best_i + 1]) >> 1);
            long long off = centre - start;
            const long long hi = (length - 1 > 0) ? (length - 1) : 0;
            off = off < 0 ? 0 : (off > hi ? hi : off);
            result = off;
        }
    }
    if (lane == 0) {
        offsets[peak] = result;
    }
}

}  // namespace

int launch_summit_offsets(const int64_t *intervals_dev, size_t n_intervals, const int64_t *centers_dev,
                          const double *effect_mean_dev, size_t n_mean,
                          const int64_t *peak_start_dev, const int64_t *peak_end_dev, size_t n_peaks,
                          int64_t *offsets_out_dev, hipStream_t stream)
{
    if (n_peaks == 0) {
        return ROCCO_HIP_OK;
    }
    // usable = min(max(len(intervals) - 1, 0), len(effect_mean))  (rocco.py:816); with explicit centres the
    // caller passes the track's own starts and every one of them is usable
    const size_t avail = (centers_dev != nullptr) ? n_intervals : (n_intervals > 0 ? n_intervals - 1 : 0);
    const long long usable = (long long)(avail < n_mean ? avail : n_mean);
    const unsigned per_block = 256 / kLanes;
    hipLaunchKernelGGL(summit_offsets_kernel, dim3((unsigned)((n_peaks + per_block - 1) / per_block)), dim3(256), 0, stream,
                       (const long long *)intervals_dev, usable, (const long long *)centers_dev, effect_mean_dev, (const long long *)peak_start_dev,
                       (const long long *)peak_end_dev, (long long)n_peaks, (long long *)offsets_out_dev);
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
