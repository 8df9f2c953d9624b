// rocco_amd/csrc/npsum.hip -- sums in NumPy's order, and the budget-null statistics built on them (SURVEY.md section 8
// (f) item 1, the K x n part of the wild-bootstrap budget null), gfx950.
//
// np.sum / np.mean of a contiguous float64 vector is not a left-to-right sum: np.add.reduce walks the vector in
// chunks of the ufunc buffer (8192 elements), sums each chunk with `pairwise_sum` (numpy/_core/src/umath/
// loops_utils.h.src: blocks of at most 128 elements with 8 interleaved accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), larger ranges split at n/2 rounded down to a multiple of 8) and adds the chunk
// sums to a running total that starts at 0.  The statistics of a budget-null draw (rocco/inference.py:676-684) are
// np.mean's of elementwise functions of the bootstrap scores, so they must be summed in exactly this order to
// equal the reference bit for bit.  A full chunk is a perfect binary tree over 64 blocks of 128: one workgroup per
// chunk (the chunk goes through LDS, one lane per block, then a tree over the 64 block sums); the ragged last chunk
// and the running total are a single thread's work (at most 8191 + n/8192 additions).
#include "kernels.h"

namespace rocco {

namespace {

constexpr int kChunk = 8192;  // np.getbufsize()
constexpr int kBlock = 128;   // PW_BLOCKSIZE
constexpr int kPad = kBlock + 1;

enum NpMode { kIdentity = 0, kPositive = 1, kPositiveScaled = 2, kPositiveFlag = 3, kAboveFlag = 4 };

struct NpParams {
    double center, scale, threshold;
};

template <int MODE>
__device__ __forceinline__ double np_value(double x, const NpParams &p)
{
    if (MODE == kIdentity) {
        return x;
    }
    if (MODE == kAboveFlag) {
        return (x > p.threshold) ? 1.0 : 0.0;  // np.mean(scores > null_threshold)
    }
    const double r = x - p.center;          // bootstrap_residual_scores
    const double pos = (r > 0.0) ? r : 0.0;  // np.clip(r, 0.0, None); NaN stays out of scope (finite scores)
    if (MODE == kPositive) {
        return pos;
    }
    if (MODE == kPositiveScaled) {
        return pos / p.scale;
    }
    return (pos > 0.0) ? 1.0 : 0.0;
}

// pairwise_sum of a[0..n) as NumPy does it (sequential; used for the ragged last chunk)
template <int MODE>
__device__ double np_pairwise(const double *__restrict__ a, long long n, const NpParams &p)
{
    if (n < 8) {
        double res = 0.0;
        for (long long i = 0; i < n; ++i) {
            res += np_value<MODE>(a[i], p);
        }
        return res;
    }
    if (n <= kBlock) {
        double r[8];
        for (int j = 0; j < 8; ++j) {
            r[j] = np_value<MODE>(a[j], p);
        }
        long long i;
        for (i = 8; i < n - (n % 8); i += 8) {
            for (int j = 0; j < 8; ++j) {
                r[j] += np_value<MODE>(a[i + j], p);
            }
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) {
            res += np_value<MODE>(a[i], p);
        }
        return res;
    }
    long long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise<MODE>(a, n2, p) + np_pairwise<MODE>(a + n2, n - n2, p);
}

template <int MODE>
__global__ __launch_bounds__(64) void np_chunk_kernel(const double *__restrict__ x, long long full_chunks, NpParams p,
                                                     double *__restrict__ chunk_sums)
{
    __shared__ double tile[64 * kPad];
    __shared__ double sums[64];
    const int lane = threadIdx.x;
    const long long chunk = blockIdx.x;
    if (chunk >= full_chunks) {
        return;
    }
    const double *__restrict__ a = x + chunk * kChunk;
    for (int e = lane; e < kChunk; e += 64) {
        tile[(e / kBlock) * kPad + (e % kBlock)] = np_value<MODE>(a[e], p);
    }
    __syncthreads();
    {
        const double *__restrict__ b = tile + lane * kPad;
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r[j] = b[j];
        }
        for (int i = 8; i < kBlock; i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r[j] += b[i + j];
            }
        }
        sums[lane] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    }
    __syncthreads();
    // 8192 = 2^13: every split is an exact halving, so the tree pairs neighbours level by level
    for (int width = 1; width < 64; width <<= 1) {
        double v = 0.0;
        const bool active = (lane % (2 * width)) == 0;
        if (active) {
            v = sums[lane] + sums[lane + width];
        }
        __syncthreads();
        if (active) {
            sums[lane] = v;
        }
        __syncthreads();
    }
    if (lane == 0) {
        chunk_sums[chunk] = sums[0];
    }
}

template <int MODE>
__global__ void np_total_kernel(const double *__restrict__ x, long long n, long long full_chunks, NpParams p,
                                const double *__restrict__ chunk_sums, double *__restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) {
        return;
    }
    double acc = 0.0;
    for (long long c = 0; c < full_chunks; ++c) {
        acc = acc + chunk_sums[c];
    }
    const long long rest = n - full_chunks * kChunk;
    if (rest > 0) {
        acc = acc + np_pairwise<MODE>(x + full_chunks * kChunk, rest, p);
    }
    *out = acc;
}

template <int MODE>
void launch_mode(const double *x, long long n, NpParams p, double *chunk_sums, double *out, hipStream_t stream)
{
    const long long full = n / kChunk;
    if (full > 0) {
        hipLaunchKernelGGL(np_chunk_kernel<MODE>, dim3((unsigned)full), dim3(64), 0, stream, x, full, p, chunk_sums);
    }
    hipLaunchKernelGGL(np_total_kernel<MODE>, dim3(1), dim3(64), 0, stream, x, n, full, p, chunk_sums, out);
}

__global__ __launch_bounds__(256) void multiply_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                      double *__restrict__ out, long long count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        out[i] = a[i] * b[i];
    }
}

// out[k][i] = matrix[k][i] - max(row[i], 0)   (rocco/inference.py:717-721)
__global__ __launch_bounds__(256) void subtract_positive_row_kernel(const double *__restrict__ matrix,
                                                                   const double *__restrict__ row, long long n,
                                                                   long long count, double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const double r = row[i % n];
        out[i] = matrix[i] - ((r > 0.0) ? r : 0.0);
    }
}

}  // namespace

size_t npsum_scratch_bytes(size_t n) { return (n / kChunk + 1) * sizeof(double) * 5 + 5 * sizeof(double) + 256; }

// sums_out_host[0..n_modes): modes[i] in {0 identity, 1 positive part of (x - center), 2 that / scale,
// 3 (x - center > 0), 4 (x > threshold)}
int launch_numpy_sums(const double *x_dev, size_t n, const int *modes, int n_modes, double center, double scale,
                      double threshold, void *scratch_dev, double *sums_out_host, hipStream_t stream)
{
    if (n_modes < 1 || n_modes > 5) {
        return ROCCO_HIP_EINVAL;
    }
    const NpParams p{center, scale, threshold};
    const size_t per = n / kChunk + 1;
    double *chunk_sums = (double *)scratch_dev;
    double *out_dev = chunk_sums + 5 * per;
    for (int i = 0; i < n_modes; ++i) {
        double *cs = chunk_sums + (size_t)i * per;
        switch (modes[i]) {
        case kIdentity: launch_mode<kIdentity>(x_dev, (long long)n, p, cs, out_dev + i, stream); break;
        case kPositive: launch_mode<kPositive>(x_dev, (long long)n, p, cs, out_dev + i, stream); break;
        case kPositiveScaled: launch_mode<kPositiveScaled>(x_dev, (long long)n, p, cs, out_dev + i, stream); break;
        case kPositiveFlag: launch_mode<kPositiveFlag>(x_dev, (long long)n, p, cs, out_dev + i, stream); break;
        case kAboveFlag: launch_mode<kAboveFlag>(x_dev, (long long)n, p, cs, out_dev + i, stream); break;
        default: return ROCCO_HIP_EINVAL;
        }
    }
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipMemcpyAsync(sums_out_host, out_dev, n_modes * sizeof(double), hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    return ROCCO_HIP_OK;
}

int launch_multiply(const double *a_dev, const double *b_dev, double *out_dev, size_t count, hipStream_t stream)
{
    if (count > 0) {
        hipLaunchKernelGGL(multiply_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, a_dev, b_dev, out_dev,
                           (long long)count);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

int launch_subtract_positive_row(const double *matrix_dev, const double *row_dev, size_t K, size_t n, double *out_dev,
                                 hipStream_t stream)
{
    const size_t count = K * n;
    if (count > 0) {
        hipLaunchKernelGGL(subtract_positive_row_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, matrix_dev,
                           row_dev, (long long)n, (long long)count, out_dev);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

}  // namespace rocco
