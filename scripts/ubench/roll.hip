// ad-hoc: where does the rolling-sum chain of wls.hip spend its time?  (one workgroup, lanes 0..2 active)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kTile = 512, kStream = kTile + 72;
// V=0: adds only (register operands); 1: + LDS operand reads; 2: + LDS result writes; 3: + barrier per tile
template <int V>
__global__ __launch_bounds__(256) void roll(double *out, int tiles, int off)
{
    __shared__ double P[3][kStream], S[3][kTile];
    for (int i = threadIdx.x; i < 3 * kStream; i += blockDim.x) (&P[0][0])[i] = 1e-3 * i;
    __syncthreads();
    const int lane = threadIdx.x;
    double sum = 0.0;
    for (int tile = 0; tile < tiles; ++tile) {
        if (lane < 3) {
            const double *__restrict__ p = P[lane];
            double *__restrict__ s = S[lane];
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[u] = p[u]; b[u] = p[u + off]; }
#pragma unroll 1
            for (int t = 0; t < kTile; t += 16) {
                double a2[8], b2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (V >= 1) { a2[u] = p[t + 8 + u]; b2[u] = p[t + 8 + u + off]; } else { a2[u] = a[u] + 1.0; b2[u] = b[u]; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { if (V >= 2) s[t + u] = sum; sum = (sum - a[u]) + b[u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (V >= 1) { a[u] = p[t + 16 + u]; b[u] = p[t + 16 + u + off]; } else { a[u] = a2[u] + 1.0; b[u] = b2[u]; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { if (V >= 2) s[t + 8 + u] = sum; sum = (sum - a2[u]) + b2[u]; }
            }
        }
        if (V >= 3) __syncthreads();
    }
    if (lane < 3) out[lane] = sum + S[lane][5];
}
template <int V> void run(double *d, const char *what)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(roll<V>, dim3(1), dim3(256), 0, 0, d, tiles, 31); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-40s %.2f ns/step\n", what, ms * 1e6 / (tiles * (double)kTile));
    }
}
int main()
{
    double *d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    run<0>(d, "adds only");
    run<1>(d, "+ LDS operand reads");
    run<2>(d, "+ LDS result writes");
    run<3>(d, "+ barrier per tile");
    return 0;
}
