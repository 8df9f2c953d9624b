// How much of the chain wavefront's LDS / memory traffic overlaps its dependent FP64 chain?  One wavefront alone on a CU
// (plus optional idle partners), batches of 8 loci: 8 x (mul, mul, sub, sub) with operands that are already in registers,
// beside a chosen mix of memory instructions whose results are NOT used by the chain (so only issue / pipe contention and
// the wait at the batch's end are measured).
//   R = LDS reads per batch:  0 none | 1 4 x b128 (lane's own 16 B) | 2 8 x b64 | 3 8 x b128 broadcast (one address)
//                             | 4 = 1 + 3 (values and multipliers, the N1 form) | 5 4 x global_load_dwordx4 (lane's own row)
//   W = writes per batch:     0 none | 1 4 x ds_write_b128 | 2 8 x ds_write_b64 | 3 4 x global_store_dwordx4
//   P = placement:            0 all memory instructions at the batch's start, one lgkmcnt(0)/vmcnt wait at its end
//                             1 one memory instruction behind each locus of the chain's first half (sched_group_barrier)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
using d2 = __attribute__((ext_vector_type(2))) double;

struct Stamps {
    long long cycles, wall;
};

template <int R, int W, int P, int LANES>
__global__ __launch_bounds__(64) void mix(double *out, Stamps *stamps, int batches, const double *seed, double *gmem, long long row_stride)
{
    __shared__ __attribute__((aligned(16))) double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) {
        lds[i] = seed[i & 1023];
    }
    __syncthreads();
    const int lane = threadIdx.x;
    const bool on = lane < LANES;
    double p1 = seed[lane], p2 = seed[lane + 1];
    double v[8], a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[q] = seed[8 + q + lane];
        a[q] = -1.9545 - 1e-6 * q;
        b[q] = 0.9555 + 1e-6 * q;
        asm volatile("" : "+v"(v[q]), "+v"(a[q]), "+v"(b[q]));
    }
    const d2 *lp = reinterpret_cast<const d2 *>(lds) + lane;  // lane's own 16 B
    d2 *wp = reinterpret_cast<d2 *>(lds) + 2048 + lane;
    double *grow = gmem + (long long)lane * row_stride;
    d2 sink = {0.0, 0.0};
    const long long c0 = clock64(), w0 = wall_clock64();
    constexpr int D = (R == 5) ? 3 : 1;  // batches between a load and its use (registers: four sets in rotation)
    d2 rd[4][12];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            rd[s][q] = d2{0.0, 0.0};
        }
    }
    auto batch = [&](int it, d2(&ld)[12], const d2(&use)[12]) {
        const int base = (it & 7) * 64;
        int nr = 0;
        if (R == 1 || R == 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ld[nr++] = lp[base + q * 17 * 4];
            }
        }
        if (R == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                ld[nr].x = lds[(base + q * 17) * 2 + lane];
                ld[nr++].y = 0.0;
            }
        }
        if (R == 3 || R == 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                ld[nr++] = reinterpret_cast<const d2 *>(lds)[3000 + base / 8 + q];
            }
        }
        if (R == 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ld[nr++] = *reinterpret_cast<const d2 *>(grow + (long long)it * 8 + 2 * q);
            }
        }
        // the values loaded D batches ago enter the chain as its inputs (so they must have arrived)
        double vv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            vv[q] = v[q];
        }
        if (R == 1 || R == 4 || R == 5) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vv[2 * q] = use[q].x;
                vv[2 * q + 1] = use[q].y;
            }
        } else if (R == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                vv[q] = use[q].x;
            }
        }
        double r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double aa = a[q], bb = b[q];
            if (R == 3) {
                aa = use[q].x, bb = use[q].y;
            } else if (R == 4) {
                aa = use[4 + q].x, bb = use[4 + q].y;
            }
            const double t1 = aa * p1;
            const double t2 = bb * p2;
            r[q] = vv[q] - t1 - t2;
            p2 = p1;
            p1 = r[q];
        }
        if (W == 1) {
            if (on) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wp[base + q * 17 * 4] = d2{r[2 * q], r[2 * q + 1]};
                }
            }
        }
        if (W == 2) {
            if (on) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    lds[4096 + (base + q * 17) * 2 + lane] = r[q];
                }
            }
        }
        if (W == 3) {
            if (on) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    *reinterpret_cast<d2 *>(grow + 4096 + (long long)it * 8 + 2 * q) = d2{r[2 * q], r[2 * q + 1]};
                }
            }
        }
        if (P == 1) {
            // one memory instruction behind every four VALU instructions while there are any
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100 | 0x200 | 0x020 | 0x040, 1, 0);
            }
        }
        SCHED_FENCE();
    };
#pragma unroll 1
    for (int it = 0; it < batches; it += 4) {
        batch(it, rd[0], rd[(0 + 4 - D) & 3]);
        batch(it + 1, rd[1], rd[(1 + 4 - D) & 3]);
        batch(it + 2, rd[2], rd[(2 + 4 - D) & 3]);
        batch(it + 3, rd[3], rd[(3 + 4 - D) & 3]);
    }
    sink = rd[0][0] + rd[1][1] + rd[2][2] + rd[3][3];
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = p1 + p2 + sink.x + sink.y;
    if (lane == 0) {
        stamps->cycles = c1 - c0;
        stamps->wall = w1 - w0;
    }
}

static double *g_out, *g_seed, *g_mem;
static Stamps *g_stamps;
constexpr long long kRowStride = 1 << 21;  // doubles between the lanes' rows

template <int R, int W, int P, int LANES>
static void run(const char *what)
{
    const int batches = 100000;
    Stamps h{};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((mix<R, W, P, LANES>), dim3(1), dim3(64), 0, 0, g_out, g_stamps, batches, g_seed, g_mem, kRowStride);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, g_stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("R%d W%d P%d lanes %2d  %-78s %7.2f cycles per locus  %7.2f ns\n", R, W, P, LANES, what, h.cycles / (batches * 8.0),
           10.0 * h.wall / (batches * 8.0));
    fflush(stdout);
}

int main()
{
    (void)hipMalloc(&g_out, 1 << 16);
    (void)hipMalloc(&g_stamps, sizeof(Stamps));
    std::vector<double> seed(2048);
    for (int i = 0; i < 2048; ++i) {
        seed[i] = 1e-3 * ((i * 37) % 101) - 0.05;
    }
    (void)hipMalloc(&g_seed, 2048 * 8);
    (void)hipMemcpy(g_seed, seed.data(), 2048 * 8, hipMemcpyHostToDevice);
    (void)hipMalloc(&g_mem, 64 * kRowStride * 8);
    (void)hipMemset(g_mem, 0, 64 * kRowStride * 8);
    run<0, 0, 0, 16>("chain alone");
    run<1, 0, 0, 16>("+ 4 ds_read_b128 (own 16 B)");
    run<2, 0, 0, 16>("+ 8 ds_read_b64");
    run<3, 0, 0, 16>("+ 8 ds_read_b128 broadcast");
    run<4, 0, 0, 16>("+ 4 ds_read_b128 + 8 broadcast b128");
    run<0, 1, 0, 16>("+ 4 ds_write_b128");
    run<0, 2, 0, 16>("+ 8 ds_write_b64");
    run<1, 1, 0, 16>("+ 4 ds_read_b128 + 4 ds_write_b128");
    run<4, 1, 0, 16>("+ 4 + 8 ds_read_b128 + 4 ds_write_b128   [N1's mix]");
    run<1, 1, 1, 16>("+ 4 ds_read_b128 + 4 ds_write_b128, spread over the chain");
    run<4, 1, 1, 16>("+ 4 + 8 ds_read_b128 + 4 ds_write_b128, spread over the chain");
    run<2, 2, 1, 16>("+ 8 ds_read_b64 + 8 ds_write_b64, spread over the chain");
    run<5, 0, 0, 16>("+ 4 global_load_dwordx4 (16 rows)");
    run<5, 0, 0, 64>("+ 4 global_load_dwordx4 (64 rows)");
    run<5, 3, 0, 16>("+ 4 global_load_dwordx4 + 4 global_store_dwordx4 (16 rows)");
    run<5, 3, 0, 64>("+ 4 global_load_dwordx4 + 4 global_store_dwordx4 (64 rows)");
    run<5, 3, 1, 64>("+ 4 global_load_dwordx4 + 4 global_store_dwordx4 (64 rows), spread");
    return 0;
}
