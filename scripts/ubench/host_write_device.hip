// Ad-hoc: can the host write task descriptors straight into device-resident memory (no copy dispatch)?  Allocates
// fine-grained device memory, writes it from the host, launches a kernel that reads it; also times host write + launch against
// hipMemcpyAsync(H2D from pinned) + launch for a 4 KB descriptor block.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void sum_kernel(const unsigned *in, int n, unsigned *out)
{
    unsigned s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += in[i];
    atomicAdd(out, s);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int n = 1024;
    unsigned *fine = nullptr, *plain = nullptr, *out = nullptr, *pinned = nullptr, *out_h = nullptr;
    hipStream_t s; CK(hipStreamCreate(&s));
    hipError_t e = hipExtMallocWithFlags((void **)&fine, n * 4, hipDeviceMallocFinegrained);
    std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    hipPointerAttribute_t attr; CK(hipPointerGetAttributes(&attr, fine));
    std::printf("memory type %d, device pointer %p, host pointer %p\n", (int)attr.type, attr.devicePointer, attr.hostPointer);
    CK(hipMalloc(&plain, n * 4)); CK(hipMalloc(&out, 4)); CK(hipHostMalloc(&pinned, n * 4)); CK(hipHostMalloc(&out_h, 4));
    // host writes into the fine-grained device allocation
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < n; ++i) fine[i] = (unsigned)(i + rep);
        __builtin_ia32_sfence();
        CK(hipMemsetAsync(out, 0, 4, s));
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, fine, n, out);
        CK(hipMemcpyAsync(out_h, out, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        std::printf("rep %d: kernel saw sum %u, expected %u\n", rep, *out_h, (unsigned)(n * (n - 1) / 2 + rep * n));
    }
    const int reps = 300;
    double t0 = now_us();
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < n; ++i) pinned[i] = (unsigned)(i + r);
        CK(hipMemcpyAsync(plain, pinned, n * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, plain, n, out);
        CK(hipStreamSynchronize(s));
    }
    std::printf("pinned + hipMemcpyAsync + launch + sync: %.1f us per round\n", (now_us() - t0) / reps);
    t0 = now_us();
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < n; ++i) fine[i] = (unsigned)(i + r);
        __builtin_ia32_sfence();
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, fine, n, out);
        CK(hipStreamSynchronize(s));
    }
    std::printf("host writes into device memory + launch + sync: %.1f us per round\n", (now_us() - t0) / reps);
    return 0;
}
