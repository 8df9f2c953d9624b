// Ad-hoc: how long after a kernel's end does the host know?  (a) hipStreamSynchronize, (b) polling a word the kernel's last
// workgroup writes to pinned host memory.  Round trip = launch of a ~100 us kernel -> host notices -> next launch, 200 times.
//   hipcc --offload-arch=gfx950 -O3 -o sync_latency sync_latency.hip && ./sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ void spin_kernel(long long cycles, unsigned *counter, volatile unsigned *flag_host, unsigned seq, double *sink)
{
    const long long t0 = wall_clock64();
    double x = threadIdx.x;
    while (wall_clock64() - t0 < cycles) {
        x = x * 1.0000001 + 1.0;
    }
    if (x == 12345.678) {
        sink[0] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned done = atomicAdd(counter, 1u);
        if (done == gridDim.x - 1) {
            *counter = 0u;
            if (flag_host != nullptr) {
                __threadfence_system();
                *flag_host = seq;
            }
        }
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    unsigned *counter, *flag;
    double *sink;
    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    CK(hipMalloc(&counter, 4));
    CK(hipMemset(counter, 0, 4));
    CK(hipMalloc(&sink, 8));
    CK(hipHostMalloc(&flag, 64, hipHostMallocDefault));
    *flag = 0;
    const long long cycles = 10000;  // 100 MHz wall clock: 100 us
    const int reps = 200, blocks = 512;
    for (int mode = 0; mode < 3; ++mode) {
        // warm
        hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, stream, cycles, counter, (volatile unsigned *)nullptr, 0u, sink);
        CK(hipStreamSynchronize(stream));
        unsigned seq = *flag;
        const double t0 = now_us();
        for (int r = 0; r < reps; ++r) {
            ++seq;
            hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, stream, cycles, counter, (volatile unsigned *)(mode == 0 ? nullptr : flag), seq, sink);
            if (mode == 0) {
                CK(hipStreamSynchronize(stream));
            } else if (mode == 1) {
                while (*(volatile unsigned *)flag != seq) {
                }
            } else {
                while (*(volatile unsigned *)flag != seq) {
                    __builtin_ia32_pause();
                }
            }
        }
        CK(hipStreamSynchronize(stream));
        const double per = (now_us() - t0) / reps;
        std::printf("%s: %.1f us per launch + notice round trip of a ~100 us kernel (overhead %.1f us)\n",
                    mode == 0 ? "hipStreamSynchronize" : (mode == 1 ? "poll pinned word        " : "poll pinned word + pause"), per, per - 100.0);
    }
    return 0;
}
