// rocco_amd/csrc/synth.hip -- device-resident synthetic K x n signal matrices (benchmark/test input).
//
// Counter-based: element (k, j) is a pure function of (seed, k, j) built from 64-bit integer
// mixing, exact int->double conversions, one IEEE multiply per stage and one IEEE divide, so
// rocco_amd/synth.py regenerates any slice bit-for-bit with NumPy on the host.  Shape of the data
// (SURVEY.md section 8d): exponential-like background with mean ~0.3 rounded to 5 decimals (the
// reference rounds bigWig values the same way, rocco/readtracks.py:186), one enriched region per
// 1500 loci of width 4..40 loci, present in ~80% of samples with amplitude 2..10.
#include "kernels.h"

namespace rocco {

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double synth_value(uint64_t seed, uint64_t k, uint64_t j)
{
    const uint64_t row_key = mix64(seed ^ ((k + 1ULL) * 0xD6E8FEB86659FD93ULL));
    const uint64_t h = mix64(row_key + j);
    // background: piecewise-linear -log2(u) on a 24-bit uniform, scaled to mean ~0.3
    const uint32_t a = (uint32_t)(h >> 40) | 1U;          // 1 .. 2^24-1, odd
    const int e = 32 - __clz(a);                           // a in [2^(e-1), 2^e)
    const double mant = (double)a * __longlong_as_double((long long)(1023 - e) << 52);  // a / 2^e
    const double t = (double)(24 - e) + 2.0 * (1.0 - mant);
    const double bg = t * 0.20794415416798357;             // 0.3 * ln 2
    double units = rint(bg * 100000.0);
    // planted peak of this 1500-locus period
    const uint64_t period = j / 1500ULL;
    const uint64_t hp = mix64(seed ^ (period * 0xA24BAED4963EE407ULL) ^ 0x5851F42D4C957F2DULL);
    const uint64_t start = period * 1500ULL + 300ULL + (hp % 600ULL);
    const uint64_t width = 4ULL + ((hp >> 16) % 37ULL);
    if (j >= start && j < start + width) {
        const uint64_t hs = mix64(hp ^ ((k + 1ULL) * 0x9FB21C651E98DF25ULL));
        if ((hs % 10ULL) < 8ULL) {
            const double u = (double)((hs >> 20) & 0xFFFFFULL) * (1.0 / 1048576.0);  // 20-bit uniform
            const double amp = 2.0 + 8.0 * u;
            units += rint(amp * 100000.0);
        }
    }
    return units / 100000.0;
}

template <typename T>
__global__ __launch_bounds__(256) void synth_kernel(T *__restrict__ m, long long K, long long n,
                                                    long long stride, uint64_t seed)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long k = blockIdx.y;
    if (j < n && k < K) {
        m[k * stride + j] = (T)synth_value(seed, (uint64_t)k, (uint64_t)j);
    }
}

}  // namespace

int launch_synth(void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, uint64_t seed,
                 hipStream_t stream)
{
    if (K == 0 || n == 0) {
        return ROCCO_HIP_OK;
    }
    if (K > 65535) {
        set_last_error("synth: K > 65535 unsupported");
        return ROCCO_HIP_EINVAL;
    }
    const int threads = 256;
    const dim3 grid((unsigned)((n + threads - 1) / threads), (unsigned)K);
    if (dtype == 0) {
        hipLaunchKernelGGL((synth_kernel<double>), grid, dim3(threads), 0, stream, (double *)matrix_dev,
                           (long long)K, (long long)n, (long long)row_stride, seed);
    } else {
        hipLaunchKernelGGL((synth_kernel<float>), grid, dim3(threads), 0, stream, (float *)matrix_dev,
                           (long long)K, (long long)n, (long long)row_stride, seed);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
