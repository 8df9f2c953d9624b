// scripts/ubench/median_gap.hip -- the whole-genome K = 100 median launch after different kinds of pause: how long the
// launch takes, what shader clock it runs at (s_memtime / s_memrealtime) and how its progress is spread over its own
// duration (workgroups retired per tenth of the launch).  For the "+0.5 ms inside a step" question of VERDICT round 2.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -o median_gap median_gap.hip && ./median_gap
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

template <int N>
__device__ __forceinline__ void select_middle(double (&v)[N])
{
#pragma unroll
    for (int p = 1; p < N; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j <= N - 1 - k; j += 2 * k)
#pragma unroll
                for (int i = 0; i <= ((k - 1 < N - j - k - 1) ? (k - 1) : (N - j - k - 1)); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const double a = v[i + j], b = v[i + j + k];
                        v[i + j] = fmin(a, b);
                        v[i + j + k] = fmax(a, b);
                    }
}

__device__ __forceinline__ unsigned xcd_contiguous_block()
{
    const unsigned nblk = gridDim.x;
    const unsigned per = nblk / 8U, rem = nblk % 8U;
    const unsigned xcd = blockIdx.x % 8U, slot = blockIdx.x / 8U;
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

constexpr int K = 100;

struct Stamp {
    unsigned long long real0, real1, clk0, clk1;
};

__global__ __launch_bounds__(256) void med(const double *__restrict__ m, long long n, long long stride, double *__restrict__ out,
                                           Stamp *__restrict__ stamps)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    const long long j0 = (long long)xcd_contiguous_block() * 256;
    const long long j = j0 + threadIdx.x;
    if (j < n) {
        double v[K];
        const double *base = m + j0;
        const unsigned lane = threadIdx.x;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v[k] = base[lane];
            base += stride;
        }
        select_middle<K>(v);
        out[j] = (v[K / 2 - 1] + v[K / 2]) / 2.0;
    }
    if (stamps != nullptr && threadIdx.x == 0) {
        Stamp s = {r0, __builtin_amdgcn_s_memrealtime(), c0, __builtin_amdgcn_s_memtime()};
        stamps[blockIdx.x] = s;
    }
}

__global__ void tiny(double *p) { if (threadIdx.x == 1000) p[0] = 1.0; }

__global__ void fillk(double *m, long long total)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ULL;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ULL;
        x ^= x >> 32;
        m[i] = (double)(x >> 40) * (1.0 / 16777216.0);
    }
}

#define CHECK(x)                                                         \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? std::atoll(argv[1]) : 61765409LL;
    double *m, *out;
    Stamp *stamps;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    CHECK(hipMalloc(&m, sizeof(double) * n * K));
    CHECK(hipMalloc(&out, sizeof(double) * n));
    CHECK(hipMalloc(&stamps, sizeof(Stamp) * blocks));
    hipLaunchKernelGGL(fillk, dim3(65536), dim3(256), 0, 0, m, n * K);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Stamp> host(blocks);
    struct Scenario {
        const char *name;
        int idle_us;
        int tiny_kernels;
    };
    const Scenario scenarios[] = {{"back to back", 0, 0},        {"after 1 ms idle", 1000, 0},    {"after 3 ms idle", 3000, 0},
                                  {"after 10 ms idle", 10000, 0}, {"after 100 ms idle", 100000, 0}, {"after 400 tiny kernels", 0, 400},
                                  {"after 1500 tiny kernels", 0, 1500}, {"back to back again", 0, 0}};
    for (const Scenario &sc : scenarios) {
        float total = 0.f;
        const int reps = 4;
        double clock_ghz = 0;
        std::vector<double> deciles(10, 0.0);
        for (int r = 0; r < reps; ++r) {
            // the launch before the pause: the same kernel (so "back to back" really follows a full-speed stream)
            hipLaunchKernelGGL(med, dim3(blocks), dim3(256), 0, 0, m, n, n, out, (Stamp *)nullptr);
            if (sc.idle_us > 0) {
                CHECK(hipDeviceSynchronize());
                std::this_thread::sleep_for(std::chrono::microseconds(sc.idle_us));
            }
            for (int t = 0; t < sc.tiny_kernels; ++t) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, 0, out);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(med, dim3(blocks), dim3(256), 0, 0, m, n, n, out, stamps);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
            CHECK(hipMemcpy(host.data(), stamps, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            double dclk = 0, dreal = 0;
            for (const Stamp &s : host) {
                t0 = std::min(t0, s.real0);
                t1 = std::max(t1, s.real1);
                dclk += (double)(s.clk1 - s.clk0);
                dreal += (double)(s.real1 - s.real0);
            }
            clock_ghz += dclk / dreal * 0.1;  // s_memrealtime ticks at 100 MHz
            const double span = (double)(t1 - t0);
            for (const Stamp &s : host) {
                int d = (int)(10.0 * (double)(s.real1 - t0) / span);
                deciles[d > 9 ? 9 : d] += 1.0;
            }
        }
        std::printf("%-26s %.3f ms  shader clock %.3f GHz  workgroups retired per tenth of the launch (%% of all):", sc.name, total / reps,
                    clock_ghz / reps);
        for (double d : deciles) std::printf(" %.1f", 100.0 * d / reps / blocks);
        std::printf("\n");
        std::fflush(stdout);
    }
    return 0;
}
