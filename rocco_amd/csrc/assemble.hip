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
