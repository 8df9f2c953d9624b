// ad-hoc: shader clock (s_memtime) vs the 100 MHz wall clock while a lone wavefront / a full grid spins
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(double *out, long long *stamps, int iters)
{
    double x = out[threadIdx.x];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = x + 1e-9;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
}
int main()
{
    double *d; long long *s, h[2];
    hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20); hipMalloc(&s, 64);
    for (int grid : {1, 1, 256, 2048, 1}) {
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, 0, d, s, 200000);
        hipDeviceSynchronize();
        hipMemcpy(h, s, 16, hipMemcpyDeviceToHost);
        printf("grid %5d: %lld shader ticks in %lld wall ticks (100 MHz) -> %.0f MHz, %.2f ns per dependent add\n", grid, h[0], h[1],
               100.0 * h[0] / h[1], 10.0 * h[1] / (200000.0 * 16));
    }
    return 0;
}
