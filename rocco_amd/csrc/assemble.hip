// rocco_amd/csrc/assemble.hip -- K x m signal matrix from per-track (locus start, value) lists (SURVEY.md section 8 (f)
// item 2), gfx950.
//
// Replaces the NumPy statements at the end of generate_chrom_matrix (rocco/readtracks.py:614-633):
//   common = np.sort(np.unique(np.concatenate(interval_matrix)))            -> radix sort + unique
//   bigWig: np.unique(np.diff(common)).size must be 1                        -> one compare per locus
//   matrix[i, np.searchsorted(common, intervals_i)] = vals_i  (zeros elsewhere) -> binary search + scatter
// A fancy-index assignment with a repeated index keeps the LAST value written; the scatter is therefore done in
// two steps: every (track, locus) cell receives the largest source position that maps to it (atomicMax), then the
// cells gather their value (or zero).  Integer work and copies only: results equal NumPy's exactly.
#include "kernels.h"

#include <hipcub/hipcub.hpp>

namespace rocco {

namespace {

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

__global__ __launch_bounds__(256) void fixed_step_kernel(const long long *__restrict__ common, const unsigned long long *m_ptr,
                                                        int *__restrict__ not_fixed)
{
    const long long m = (long long)*m_ptr;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < m && i >= 1) {
        if (common[i + 1] - common[i] != common[1] - common[0]) {
            atomicOr(not_fixed, 1);
        }
    }
}

__global__ __launch_bounds__(256) void scatter_positions_kernel(const long long *__restrict__ common, long long m,
                                                               const long long *__restrict__ intervals, long long count,
                                                               unsigned *__restrict__ pos_row)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) {
        return;
    }
    const long long key = intervals[j];
    long long lo = 0, hi = m;  // np.searchsorted(common, key), side="left"
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        if (common[mid] < key) {
            lo = mid + 1;
        } else {
            hi = mid;
        }
    }
    if (lo < m) {
        atomicMax(&pos_row[lo], (unsigned)(j + 1));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_values_kernel(const unsigned *__restrict__ pos_row, long long m,
                                                           const double *__restrict__ vals, T *__restrict__ out_row)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m) {
        const unsigned p = pos_row[c];
        out_row[c] = p ? (T)vals[p - 1] : (T)0;
    }
}

}  // namespace

size_t union_scratch_bytes(size_t count)
{
    size_t a = 0, b = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const long long *)nullptr, (long long *)nullptr, (int)count);
    (void)hipcub::DeviceSelect::Unique(nullptr, b, (const long long *)nullptr, (long long *)nullptr,
                                       (unsigned long long *)nullptr, (int)count);
    return align256(count * 8) + align256(a > b ? a : b) + 512;
}

int launch_union_intervals(const int64_t *values_dev, size_t count, int64_t *unique_out_dev, size_t *n_unique_out,
                           int *fixed_step_out, void *scratch_dev, hipStream_t stream)
{
    char *sc = (char *)scratch_dev;
    long long *sorted = (long long *)sc;
    unsigned long long *m_dev = (unsigned long long *)(sc + align256(count * 8));
    int *flag = (int *)(m_dev + 1);
    void *tmp = sc + align256(count * 8) + 256;
    size_t a = 0, b = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const long long *)values_dev, sorted, (int)count);
    (void)hipcub::DeviceSelect::Unique(nullptr, b, sorted, (long long *)unique_out_dev, m_dev, (int)count);
    ROCCO_HIP_TRY(hipMemsetAsync(m_dev, 0, 16, stream));
    ROCCO_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(tmp, a, (const long long *)values_dev, sorted, (int)count, 0, 64, stream));
    ROCCO_HIP_TRY(hipcub::DeviceSelect::Unique(tmp, b, sorted, (long long *)unique_out_dev, m_dev, (int)count, stream));
    hipLaunchKernelGGL(fixed_step_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       (const long long *)unique_out_dev, m_dev, flag);
    ROCCO_HIP_TRY(hipGetLastError());
    unsigned long long host[2] = {0, 0};
    ROCCO_HIP_TRY(hipMemcpyAsync(host, m_dev, 16, hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    *n_unique_out = (size_t)host[0];
    if (fixed_step_out != nullptr) {
        *fixed_step_out = ((int)(host[1] & 0xffffffffULL) == 0) ? 1 : 0;
    }
    return ROCCO_HIP_OK;
}

size_t scatter_scratch_bytes(size_t K, size_t m) { return align256(K * m * sizeof(unsigned)) + 256; }

int launch_scatter_tracks(const int64_t *common_dev, size_t m, const int64_t *intervals_concat_dev,
                          const double *vals_concat_dev, const size_t *offsets_host, size_t K, int out_dtype,
                          void *matrix_out_dev, void *scratch_dev, hipStream_t stream)
{
    unsigned *pos = (unsigned *)scratch_dev;
    ROCCO_HIP_TRY(hipMemsetAsync(pos, 0, K * m * sizeof(unsigned), stream));
    const unsigned blocks_m = (unsigned)((m + 255) / 256);
    for (size_t k = 0; k < K; ++k) {
        const size_t off = offsets_host[k], cnt = offsets_host[k + 1] - off;
        if (cnt > 0) {
            hipLaunchKernelGGL(scatter_positions_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream,
                               (const long long *)common_dev, (long long)m, (const long long *)intervals_concat_dev + off,
                               (long long)cnt, pos + k * m);
        }
        if (out_dtype == 0) {
            hipLaunchKernelGGL(gather_values_kernel<double>, dim3(blocks_m), dim3(256), 0, stream, pos + k * m, (long long)m,
                               vals_concat_dev + off, (double *)matrix_out_dev + k * m);
        } else {
            hipLaunchKernelGGL(gather_values_kernel<float>, dim3(blocks_m), dim3(256), 0, stream, pos + k * m, (long long)m,
                               vals_concat_dev + off, (float *)matrix_out_dev + k * m);
        }
    }
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));  // the scratch buffer is the solver's
    return ROCCO_HIP_OK;
}

}  // namespace rocco

// ---- bigWig dense fill (rocco/readtracks.py:141-186) ----------------------------------------------------------
namespace rocco {

namespace {

enum { kBwNonFinite = 1, kBwBadWidth = 2, kBwVarWidth = 4, kBwMisaligned = 8, kBwDuplicate = 16 };

__global__ __launch_bounds__(256) void bigwig_check_kernel(const long long *__restrict__ starts, const long long *__restrict__ ends,
                                                          const double *__restrict__ vals, long long count,
                                                          int *__restrict__ flags)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) {
        return;
    }
    const long long step = ends[0] - starts[0], offset = starts[0];
    int f = 0;
    if (!isfinite(vals[i])) {
        f |= kBwNonFinite;
    }
    const long long w = ends[i] - starts[i];
    if (w <= 0) {
        f |= kBwBadWidth;
    }
    if (w != step) {
        f |= kBwVarWidth;
    }
    if (step > 0) {
        if ((starts[i] - offset) % step != 0) {
            f |= kBwMisaligned;
        }
        // np.unique(idx).size != idx.size: pyBigWig returns intervals in ascending order, so a repeated or
        // out-of-order start shows up next to its neighbour
        if (i > 0 && starts[i] <= starts[i - 1]) {
            f |= kBwDuplicate;
        }
    }
    if (f != 0) {
        atomicOr(flags, f);
    }
}

__global__ __launch_bounds__(256) void bigwig_scatter_kernel(const long long *__restrict__ starts, const double *__restrict__ vals,
                                                            long long count, long long step, double *__restrict__ full)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        full[(starts[i] - starts[0]) / step] = vals[i];
    }
}

// full = np.round(full * const_scale (if const_scale >= 0), round_digits): np.round multiplies by 10**d, rounds
// half to even (np.rint) and divides by 10**d (d >= 0), or divides, rounds and multiplies (d < 0)
__global__ __launch_bounds__(256) void scale_round_kernel(double *__restrict__ full, long long n, double const_scale,
                                                         int apply_scale, double pow10, int digits)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    double v = full[i];
    if (apply_scale) {
        v = v * const_scale;
    }
    if (digits > 0) {
        v = rint(v * pow10) / pow10;
    } else if (digits == 0) {
        v = rint(v);
    } else {
        v = rint(v / pow10) * pow10;
    }
    full[i] = v;
}

}  // namespace

int launch_bigwig_dense_fill(const int64_t *starts_dev, const int64_t *ends_dev, const double *vals_dev, size_t count,
                             double const_scale, int round_digits, double *full_out_dev, size_t capacity,
                             int64_t *first_start_out, int64_t *step_out, size_t *n_full_out, int *flags_out,
                             void *scratch_dev, hipStream_t stream)
{
    int *flags = (int *)scratch_dev;
    long long ends_host[1], starts_host[2];
    ROCCO_HIP_TRY(hipMemsetAsync(flags, 0, sizeof(int), stream));
    hipLaunchKernelGGL(bigwig_check_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       (const long long *)starts_dev, (const long long *)ends_dev, vals_dev, (long long)count, flags);
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipMemcpyAsync(&starts_host[0], starts_dev, 8, hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync(&starts_host[1], starts_dev + (count - 1), 8, hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync(&ends_host[0], ends_dev, 8, hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync(flags_out, flags, sizeof(int), hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    const long long step = ends_host[0] - starts_host[0];
    *first_start_out = starts_host[0];
    *step_out = step;
    *n_full_out = 0;
    if (*flags_out != 0 || step <= 0) {
        return ROCCO_HIP_OK;  // the caller raises the reference's ValueError for the flag
    }
    const size_t n_full = (size_t)((starts_host[1] - starts_host[0]) / step) + 1;  // np.arange(first, last + step, step)
    *n_full_out = n_full;
    if (full_out_dev == nullptr || capacity < n_full) {
        return ROCCO_HIP_OK;  // size query
    }
    ROCCO_HIP_TRY(hipMemsetAsync(full_out_dev, 0, n_full * sizeof(double), stream));
    hipLaunchKernelGGL(bigwig_scatter_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       (const long long *)starts_dev, vals_dev, (long long)count, step, full_out_dev);
    const int digits = round_digits;
    double pow10 = 1.0;
    for (int d = 0; d < (digits < 0 ? -digits : digits); ++d) {
        pow10 *= 10.0;  // 10**|d| as NumPy's integer power converted to float64 (exact up to 10**22)
    }
    hipLaunchKernelGGL(scale_round_kernel, dim3((unsigned)((n_full + 255) / 256)), dim3(256), 0, stream, full_out_dev,
                       (long long)n_full, const_scale, const_scale >= 0.0 ? 1 : 0, pow10, digits);
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    return ROCCO_HIP_OK;
}

}  // namespace rocco
