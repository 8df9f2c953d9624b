// micro-benchmark: memory part vs network part of the K=100 median kernel
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__device__ __forceinline__ void select_middle(double (&v)[N]) {
#pragma unroll
  for (int p = 1; p < N; p <<= 1)
#pragma unroll
    for (int k = p; k >= 1; k >>= 1)
#pragma unroll
      for (int j = k % p; j <= N - 1 - k; j += 2 * k)
#pragma unroll
        for (int i = 0; i <= ((k - 1 < N - j - k - 1) ? (k - 1) : (N - j - k - 1)); ++i)
          if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) { double a = v[i + j], b = v[i + j + k]; v[i + j] = fmin(a, b); v[i + j + k] = fmax(a, b); }
}
template <int MODE, int KP>
__global__ __launch_bounds__(256) void med(const double* __restrict__ m, long long n, long long stride, double* __restrict__ out) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double v[KP];
  if (MODE != 1) {
#pragma unroll
    for (int k = 0; k < KP; ++k) v[k] = m[(long long)k * stride + j];
  } else {
    const double s = m[j];
#pragma unroll
    for (int k = 0; k < KP; ++k) v[k] = s * (double)((k * 37) % 101) + (double)k;
  }
  if (MODE == 0) { double s = 0; 
#pragma unroll
    for (int k = 0; k < KP; ++k) s += v[k]; out[j] = s; }
  else { select_middle<KP>(v); out[j] = (v[KP / 2 - 1] + v[KP / 2]) / 2.0; }
}
__global__ void fillk(double* m, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long x = (unsigned long long)i * 0x9E3779B97F4A7C15ULL; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
    m[i] = (double)(x >> 40) * (1.0 / 16777216.0);
  }
}
int main() {
  const long long n = 4979129; const int K = 100;
  double* m; double* out; hipMalloc(&m, sizeof(double) * n * K); hipMalloc(&out, sizeof(double) * n);
  hipMemset(m, 0, sizeof(double) * n * K);
  hipLaunchKernelGGL(fillk, dim3(65536), dim3(256), 0, 0, m, n * K);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"loads+sum only", "network only", "loads+network"};
  for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 3; ++rep) {
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL((med<0, 100>), g, b, 0, 0, m, n, n, out);
    if (mode == 1) hipLaunchKernelGGL((med<1, 100>), g, b, 0, 0, m, n, n, out);
    if (mode == 2) hipLaunchKernelGGL((med<2, 100>), g, b, 0, 0, m, n, n, out);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("%s: %.3f ms (%.0f GB/s of 8K+8 B/locus)\n", names[mode], ms, (8.0 * K + 8) * n / ms / 1e6);
  }
  return 0;
}
