// single-wave dependent-chain latency of f64 add / cmp+select on gfx950 (ad-hoc)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain_add(double *out, double a, int iters)
{
    double x = out[threadIdx.x];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = x + a;
    }
    out[threadIdx.x] = x;
}
__global__ void chain_sel(double *out, double a, double b, int iters)
{
    double x = out[threadIdx.x], y = out[threadIdx.x + 64];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double l = y - a, kp = y + b - a, en = x - a + b - a;
            const bool tl = l > x, te = en >= kp;
            x = tl ? l : x;
            y = te ? en : kp;
        }
    }
    out[threadIdx.x] = x + y;
}
int main()
{
    double *d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 100000;
        hipEventRecord(e0); hipLaunchKernelGGL(chain_add, dim3(1), dim3(64), 0, 0, d, 1e-9, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dependent v_add_f64: %.2f ns/op\n", ms * 1e6 / (iters * 16.0));
        hipEventRecord(e0); hipLaunchKernelGGL(chain_sel, dim3(1), dim3(64), 0, 0, d, 1e-9, 0.3, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("viterbi step (3 add chain + cmp + select): %.2f ns/step\n", ms * 1e6 / (iters * 8.0));
    }
    return 0;
}
