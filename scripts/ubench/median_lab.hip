// scripts/ubench/median_lab.hip -- variants of the K = 100 float64 column-median kernel at whole-genome size
// (one 100 x 61 765 409 matrix, 49.4 GB), timed back to back with HIP events.  What it is for: finding out why the
// selection network's issue time adds to the stream time instead of hiding behind it.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -o median_lab median_lab.hip && ./median_lab [n]
//
// variants (all one lane per column, all 100 values in registers):
//   base      256-thread workgroups, 64-bit per-lane address per row (round 2's kernel)
//   saddr     scalar row base + one 32-bit lane offset: the load phase needs no vector ALU instruction per row
//   saddr64   saddr with 64-thread workgroups (every wavefront scheduled on its own)
//   prio      saddr + s_setprio 3 while loading, 0 while computing
//   loads     saddr loads + a sum, no network  (the stream's own ceiling at the same occupancy)
//   network   the network on values made in registers (the ALU floor)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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
enum Mode { BASE = 0, SADDR = 1, PRIO = 2, LOADS = 3, NETWORK = 4, NT = 5, LOADS_NT = 6 };

template <int MODE, int WG>
__global__ __launch_bounds__(WG) void med(const double *__restrict__ m, long long n, long long stride, double *__restrict__ out)
{
    const long long j = (long long)xcd_contiguous_block() * WG + threadIdx.x;
    if (j >= n) return;
    double v[K];
    if (MODE == BASE) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = m[(long long)k * stride + j];
    } else if (MODE == NETWORK) {
        const double s = m[j];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = s * (double)((k * 37) % 101) + (double)k;
    } else {
        if (MODE == PRIO) __builtin_amdgcn_s_setprio(3);
        // the workgroup's first column as a scalar, the lane's distance from it as 32 bits: every row's address is
        // (scalar base) + (one vector offset)
        const long long j0 = (long long)xcd_contiguous_block() * WG;
        const double *base = m + j0;
        const unsigned lane = threadIdx.x;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v[k] = (MODE == NT || MODE == LOADS_NT) ? __builtin_nontemporal_load(base + lane) : base[lane];
            base += stride;
        }
        if (MODE == PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (MODE == LOADS || MODE == LOADS_NT) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += v[k];
        out[j] = s;
    } else {
        double s = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += v[k];
        select_middle<K>(v);
        const double r = (v[K / 2 - 1] + v[K / 2]) / 2.0;
        out[j] = (s != s) ? s : r;
    }
}

// pure read streams over the same bytes: what the memory system gives a kernel that only reads
template <int VEC, bool NTL>
__global__ __launch_bounds__(256) void stream_sum(const double *__restrict__ m, long long total, double *__restrict__ out)
{
    typedef double vec_t __attribute__((ext_vector_type(VEC)));
    const vec_t *p = (const vec_t *)m;
    const long long nv = total / VEC;
    // every workgroup owns one contiguous slab (XCD-contiguous order), lanes interleaved inside it
    const long long per = (nv + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)xcd_contiguous_block() * per;
    const long long hi = (lo + per < nv) ? lo + per : nv;
    double s = 0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256 * 4) {
        vec_t a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long q = i + 256LL * u;
            if (q < hi) a[u] = NTL ? __builtin_nontemporal_load(p + q) : p[q];
            else for (int e = 0; e < VEC; ++e) a[u][e] = 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            for (int e = 0; e < VEC; ++e) s += a[u][e];
    }
    out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

// the K-row pattern with 16 bytes per lane (two columns per lane), sum per column: does the access width matter?
template <bool NTL>
__global__ __launch_bounds__(256) void rows_sum2(const double *__restrict__ m, long long n, long long stride, double *__restrict__ out)
{
    typedef double vec2 __attribute__((ext_vector_type(2)));
    const long long j0 = (long long)xcd_contiguous_block() * 512;
    const unsigned lane = threadIdx.x;
    if (j0 + 2 * lane + 1 >= n) return;
    const double *base = m + j0;
    vec2 s = {0, 0};
#pragma unroll 20
    for (int k = 0; k < K; ++k) {
        const vec2 *q = (const vec2 *)(base) + lane;  // (rows are 16-byte aligned only when stride is even: the lab's n is made even)
        s += NTL ? __builtin_nontemporal_load(q) : *q;
        base += stride;
    }
    *((vec2 *)(out + j0) + lane) = s;
}

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

__global__ void checksum(const double *a, long long n, unsigned long long *acc)
{
    unsigned long long s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        s += (unsigned long long)__double_as_longlong(a[i]) * (unsigned long long)(2 * i + 1);
    atomicAdd(acc, s);
}

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

struct Variant {
    const char *name;
    int wg;
    void (*launch)(const double *, long long, long long, double *, unsigned, hipStream_t);
    bool median;
};

template <int MODE, int WG>
void go(const double *m, long long n, long long stride, double *out, unsigned blocks, hipStream_t s)
{
    hipLaunchKernelGGL((med<MODE, WG>), dim3(blocks), dim3(WG), 0, s, m, n, stride, out);
}
template <int VEC, bool NTL, int GRID>
void go_stream(const double *m, long long n, long long stride, double *out, unsigned, hipStream_t s)
{
    hipLaunchKernelGGL((stream_sum<VEC, NTL>), dim3(GRID), dim3(256), 0, s, m, n * K, out);
}
template <bool NTL>
void go_rows2(const double *m, long long n, long long stride, double *out, unsigned, hipStream_t s)
{
    hipLaunchKernelGGL((rows_sum2<NTL>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, s, m, n, stride, out);
}

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? std::atoll(argv[1]) : 61765409LL;  // (odd n: rows are 8-byte aligned only, as in real chromosomes;
                                                                        //  the two-column variants need an even n)
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    double *m, *out;
    unsigned long long *acc;
    CHECK(hipMalloc(&m, sizeof(double) * n * K));
    CHECK(hipMalloc(&out, sizeof(double) * n));
    CHECK(hipMalloc(&acc, 8));
    hipLaunchKernelGGL(fillk, dim3(65536), dim3(256), 0, 0, m, n * K);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Variant> variants = {
        {"base", 256, go<BASE, 256>, true},       {"saddr", 256, go<SADDR, 256>, true}, {"saddr64", 64, go<SADDR, 64>, true},
        {"saddr128", 128, go<SADDR, 128>, true},  {"prio", 256, go<PRIO, 256>, true},   {"prio64", 64, go<PRIO, 64>, true},
        {"loads", 256, go<LOADS, 256>, false},    {"network", 256, go<NETWORK, 256>, false},
        {"nt", 256, go<NT, 256>, true},           {"nt64", 64, go<NT, 64>, true},       {"loads_nt", 256, go<LOADS_NT, 256>, false},
        {"seq8B", 256, go_stream<1, false, 8192>, false},   {"seq16B", 256, go_stream<2, false, 8192>, false},
        {"seq16Bnt", 256, go_stream<2, true, 8192>, false}, {"seq16B_2k", 256, go_stream<2, false, 2048>, false},
        {"seq16B_64k", 256, go_stream<2, false, 65536>, false},
    };
    if (n % 2 == 0) {
        variants.push_back({"rows16B", 256, go_rows2<false>, false});
        variants.push_back({"rows16Bnt", 256, go_rows2<true>, false});
    }
    const char *only = argc > 3 ? argv[3] : nullptr;  // comma list of variant names
    const double bytes = (8.0 * K + 8.0) * (double)n;
    unsigned long long want = 0;
    for (int round = 0; round < 2; ++round) {  // two rounds: the order effect (clock, caches) shows as the difference
        for (const Variant &v : variants) {
            if (only != nullptr && std::strstr(only, v.name) == nullptr) continue;
            const unsigned blocks = (unsigned)((n + v.wg - 1) / v.wg);
            v.launch(m, n, n, out, blocks, 0);  // warm-up
            CHECK(hipDeviceSynchronize());
            float best = 1e30f, sum = 0.f;
            for (int r = 0; r < reps; ++r) {
                CHECK(hipEventRecord(e0));
                v.launch(m, n, n, out, blocks, 0);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                sum += ms;
            }
            unsigned long long got = 0;
            if (v.median) {
                CHECK(hipMemset(acc, 0, 8));
                hipLaunchKernelGGL(checksum, dim3(1024), dim3(256), 0, 0, out, n, acc);
                CHECK(hipMemcpy(&got, acc, 8, hipMemcpyDeviceToHost));
                if (want == 0) want = got;
            }
            std::printf("round %d  %-9s wg %3d  mean %.3f ms  best %.3f ms  %.0f GB/s (best)%s\n", round, v.name, v.wg, sum / reps, best,
                        bytes / best / 1e6, v.median ? (got == want ? "  [same medians]" : "  [MEDIANS DIFFER]") : "");
            std::fflush(stdout);
        }
    }
    return 0;
}
