// micro-benchmark: issue rate of f64 min/max/add and 64-bit integer compare-select on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 0.001 + i; b[i] = seed * 0.5 + i * 0.37; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) { double lo = fmin(a[i], b[i]); double hi = fmax(a[i], b[i]); a[i] = lo + 1e-30; b[i] = hi; }
      if (OP == 1) { a[i] = a[i] + b[i]; b[i] = b[i] + 1.0; }
      if (OP == 2) { long long x = __double_as_longlong(a[i]), y = __double_as_longlong(b[i]); bool c = x < y; long long lo = c ? x : y, hi = c ? y : x; a[i] = __longlong_as_double(lo + 1); b[i] = __longlong_as_double(hi); }
      if (OP == 3) { double lo = __builtin_fmin(a[i], b[i]); a[i] = lo; b[i] = b[i] * 1.0000001; }
    }
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, 1024 * 256 * 8 * 8);
  const int iters = 4000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"minmax+add (3 f64 ops/elem)", "add+add (2 f64 ops)", "i64 cmp+4 cndmask+add", "min+mul (2 f64 ops)"};
  for (int op = 0; op < 4; ++op) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (op == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, d, iters, 1.5);
      if (op == 1) hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, d, iters, 1.5);
      if (op == 2) hipLaunchKernelGGL(k<2>, dim3(4096), dim3(256), 0, 0, d, iters, 1.5);
      if (op == 3) hipLaunchKernelGGL(k<3>, dim3(4096), dim3(256), 0, 0, d, iters, 1.5);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double elems = 4096.0 * 256 * iters * 8;
      if (rep == 1) printf("%s: %.3f ms, %.2f T elem-iters/s\n", names[op], ms, elems / ms / 1e9);
    }
  }
  return 0;
}
