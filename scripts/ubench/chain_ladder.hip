// Where do the two sequential chains of the count path spend their time?  
//
// One workgroup; wavefront 0 runs the chain in LANES lanes exactly as the product kernels' chain wavefronts do, rung by
// rung:   W0 registers only  ->  W1 + LDS operand reads as whittaker_rows_kernel issues them  ->  W2 + LDS result writes
//         ->  W3 + four helper wavefronts staging tiles and a barrier per 64-locus tile
// and the candidate forms:
//         N1 operands as ds_read_b128 pairs, multipliers as one broadcast b128 per locus, results as ds_write_b128 pairs, the
//            next batch's reads and the last batch's writes issued BEHIND an explicit wait and AHEAD of the batch's chain
//         N2 = N1 with the multipliers in scalar registers (s_load from the factor table)
// The same ladder for the rolling sums (two dependent additions per start position): R0 .. R3 as roll.hip, and
//         M1 (leaving, entering) as one b128 per start position, sums written as b128 pairs, waits as N1.
// Prints ns and shader cycles per locus.   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 chain_ladder.hip -o chain_ladder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt lgkmcnt(0) only (vmcnt / expcnt left at their maxima): gfx9 encoding vm[3:0] | exp[6:4] | lgkm[11:8] | vm_hi[15:14]
#define WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

constexpr int kTile = 64;    // loci per tile (whittaker_rows_kernel's)
constexpr int kPitch = 17;   // doubles per locus line of the product layout (2 x 8 rows + 1)
constexpr int kPairPitch = 34;  // doubles per PAIR of loci in the candidate layout: [t / 2][col][2] + 2 of padding

struct Stamps {
    long long cycles, wall;
};

// ---- Whittaker substitution: r = v - c1 * p1 - c2 * p2 (baseline_backend.c:146-151) -------------------------------------------
__device__ __forceinline__ double chain_step(double v, double c1, double c2, double p1, double p2)
{
    const double t1 = c1 * p1;
    const double t2 = c2 * p2;
    return v - t1 - t2;
}

struct ProductTiles {  // as RowTiles of whittaker.hip
    double in[2][kTile * kPitch];
    double out[2][kTile * kPitch];
    double coef[2][2][kTile][2];
};

// V = 0 registers only; 1 + operand reads; 2 + result writes; 3 + helpers and a barrier per tile
template <int V, int LANES>
__global__ __launch_bounds__(320) void whittaker_product(double *out, Stamps *stamps, int tiles, const double *seed)
{
    __shared__ __attribute__((aligned(16))) ProductTiles T;
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 8); i += blockDim.x) {
        reinterpret_cast<double *>(&T)[i] = seed[i & 1023];
    }
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const bool chain_lane = lane < LANES;
    const int col = chain_lane ? (lane % 16) : 0;
    const int parity = col / 8;
    double p1 = 0.0, p2 = 0.0;
    if (V < 3 && wave > 0) {
        return;
    }
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
    }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int k = 0; k < tiles; ++k) {
        if (wave == 0) {
            const double *__restrict__ in = T.in[k & 1];
            double *__restrict__ o = T.out[k & 1];
            const double(*__restrict__ coef)[2] = T.coef[k & 1][parity];
            double v[8], a[8], b[8], v2[8], a2[8], b2[8], r[8];
            auto fetch = [&](double(&vv)[8], double(&aa)[8], double(&bb)[8], int j) {
                const int t0 = 8 * ((j < kTile / 8) ? j : (kTile / 8 - 1));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (V >= 1) {
                        vv[q] = in[(t0 + q) * kPitch + col];
                        aa[q] = coef[t0 + q][0];
                        bb[q] = coef[t0 + q][1];
                    } else {
                        vv[q] = 1e-3 * (double)(j + q);
                        aa[q] = -1.9545;
                        bb[q] = 0.9555;
                    }
                }
            };
            auto run = [&](const double(&vv)[8], const double(&aa)[8], const double(&bb)[8], int j) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    r[q] = chain_step(vv[q], aa[q], bb[q], p1, p2);
                    p2 = p1;
                    p1 = r[q];
                }
                if (V >= 2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (chain_lane) {
                            o[(8 * j + q) * kPitch + col] = r[q];
                        }
                    }
                }
            };
            fetch(v, a, b, 0);
#pragma unroll 1
            for (int j = 0; j < kTile / 8; j += 2) {
                fetch(v2, a2, b2, j + 1);
                run(v, a, b, j);
                fetch(v, a, b, j + 2);
                run(v2, a2, b2, j + 1);
            }
        } else {
            // a helper wavefront's LDS work per tile in the product: two rows of 64 loci stored transposed (4 ds_write_b64),
            // two rows read back for the write-back (4 ds_read_b64), one of them also the multipliers
            double *__restrict__ in = T.in[(k + 1) & 1];
            const double *__restrict__ o = T.out[(k + 1) & 1];
            const int h = wave - 1;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = h + 4 * q;
                in[lane * kPitch + r] = seed[(k + lane) & 1023];
                in[lane * kPitch + 8 + r] = seed[(k + lane + 7) & 1023];
                acc += o[lane * kPitch + r] + o[lane * kPitch + 8 + r];
            }
            if (acc == 12345.678) {
                out[64 + threadIdx.x] = acc;
            }
        }
        if (V >= 3) {
            __syncthreads();
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (wave == 0) {
        out[lane] = p1 + p2 + T.out[0][5];
        if (lane == 0) {
            stamps->cycles = c1 - c0;
            stamps->wall = w1 - w0;
        }
    }
}

// ---- the candidate: pairs of loci per LDS access, explicit waits ------------------------------------------------------------
struct PairTiles {
    double in[2][kTile / 2 * kPairPitch];   // [t / 2][col][2]
    double out[2][kTile / 2 * kPairPitch];
    double coef[2][2][kTile][2];            // as the product's (one b128 per locus, broadcast)
};

// COEF: 0 = multipliers as LDS broadcasts, 1 = in scalar registers from `factor` (l1 then l2, kTile entries per tile)
template <int COEF, int LANES, bool HELPERS>
__global__ __launch_bounds__(320) void whittaker_pairs(double *out, Stamps *stamps, int tiles, const double *seed,
                                                        const double *__restrict__ factor)
{
    __shared__ __attribute__((aligned(16))) PairTiles T;
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 8); i += blockDim.x) {
        reinterpret_cast<double *>(&T)[i] = seed[i & 1023];
    }
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const bool chain_lane = lane < LANES;
    const int col = chain_lane ? (lane % 16) : 0;
    const int parity = COEF ? 0 : (col / 8);
    double p1 = 0.0, p2 = 0.0;
    if (!HELPERS && wave > 0) {
        return;
    }
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
    }
    using d2 = __attribute__((ext_vector_type(2))) double;
    const long long c0 = clock64(), w0 = wall_clock64();
    if (wave == 0) {
        d2 v[4], v2[4], c[8], c2[8], r[4], rp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rp[q] = d2{0.0, 0.0};
        }
        long long locus = 0;  // (for the scalar loads: batch j of tile k reads factor[locus .. locus + 8))
        auto fetch = [&](d2(&vv)[4], d2(&cc)[8], int k, int j, long long at) {
            const d2 *__restrict__ in = reinterpret_cast<const d2 *>(T.in[k & 1]);
            const d2 *__restrict__ coef = reinterpret_cast<const d2 *>(T.coef[k & 1][parity]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vv[q] = in[(4 * j + q) * (kPairPitch / 2) + col];
            }
            if (COEF == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    cc[q] = coef[8 * j + q];
                }
            } else {
                // uniform addresses: scalar loads (the table repeats every 4096 loci so that it stays in the scalar cache)
                const double *__restrict__ f = factor + (at & 4095);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    cc[q] = d2{f[q], f[4096 + q]};
                }
            }
        };
        auto run = [&](const d2(&vv)[4], const d2(&cc)[8]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double a0 = cc[2 * q].x, b0 = cc[2 * q].y, a1 = cc[2 * q + 1].x, b1 = cc[2 * q + 1].y;
                const double r0 = chain_step(vv[q].x, a0, b0, p1, p2);
                const double r1 = chain_step(vv[q].y, a1, b1, r0, p1);
                p2 = r0;
                p1 = r1;
                r[q] = d2{r0, r1};
            }
        };
        auto flush = [&](const d2(&rr)[4], int k, int j) {
            d2 *__restrict__ o = reinterpret_cast<d2 *>(T.out[k & 1]);
            if (chain_lane) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[(4 * j + q) * (kPairPitch / 2) + col] = rr[q];
                }
            }
        };
        fetch(v, c, 0, 0, 0);
        for (int k = 0; k < tiles; ++k) {
#pragma unroll 1
            for (int j = 0; j < kTile / 8; j += 2) {
                // batch j: its operands were fetched a batch ago; behind the wait the next batch's reads and the last batch's
                // writes are issued, then the chain runs on registers alone
                WAIT_LGKM0();
                SCHED_FENCE();
                fetch(v2, c2, k, j + 1, locus + 8);
                flush(rp, k, j);
                SCHED_FENCE();
                run(v, c);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rp[q] = r[q];
                }
                locus += 8;
                SCHED_FENCE();
                WAIT_LGKM0();
                SCHED_FENCE();
                fetch(v, c, (j + 2 < kTile / 8) ? k : (k + 1), (j + 2) & (kTile / 8 - 1), locus + 8);
                flush(rp, k, j + 1);
                SCHED_FENCE();
                run(v2, c2);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rp[q] = r[q];
                }
                locus += 8;
                SCHED_FENCE();
            }
            if (HELPERS) {
                __syncthreads();
            }
        }
    } else {
        for (int k = 0; k < tiles; ++k) {
            double *__restrict__ in = T.in[(k + 1) & 1];
            const double *__restrict__ o = T.out[(k + 1) & 1];
            const int h = wave - 1;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = h + 4 * q;
                in[(lane >> 1) * kPairPitch + 2 * r + (lane & 1)] = seed[(k + lane) & 1023];
                in[(lane >> 1) * kPairPitch + 2 * (8 + r) + (lane & 1)] = seed[(k + lane + 7) & 1023];
                acc += o[(lane >> 1) * kPairPitch + 2 * r + (lane & 1)] + o[(lane >> 1) * kPairPitch + 2 * (8 + r) + (lane & 1)];
            }
            if (acc == 12345.678) {
                out[64 + threadIdx.x] = acc;
            }
            __syncthreads();
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (wave == 0) {
        out[lane] = p1 + p2 + T.out[0][5];
        if (lane == 0) {
            stamps->cycles = c1 - c0;
            stamps->wall = w1 - w0;
        }
    }
}

// ---- rolling sums: sum = (sum - a) + b (wls_backend.c:711-722) ---------------------------------------------------------------
constexpr int kRollTile = 64, kRollPitch = 25;  // G = 8: 24 chains per line + 1

struct RollProduct {
    double S[2][kRollTile][kRollPitch];
    double P[4 * kRollTile][kRollPitch];
};

// V = 0 registers only; 1 + operand reads (two b64 per start position, two batches ahead as the product); 2 + result writes;
// 3 + eight helper wavefronts and a barrier per tile
template <int V, int LANES>
__global__ __launch_bounds__(576) void rolling_product(double *out, Stamps *stamps, int tiles, const double *seed)
{
    __shared__ __attribute__((aligned(16))) RollProduct T;
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 8); i += blockDim.x) {
        reinterpret_cast<double *>(&T)[i] = seed[i & 1023];
    }
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const bool runs = lane < LANES;
    const int col = runs ? (lane % 24) : 0;
    double sum = 0.0;
    if (V < 3 && wave > 0) {
        return;
    }
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
    }
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int k = 0; k < tiles; ++k) {
        if (wave == 0) {
            double(*__restrict__ S)[kRollPitch] = T.S[k & 1];
            const int a0 = (k * kRollTile) & (4 * kRollTile - 1), b0 = (a0 + 31) & (4 * kRollTile - 1);
            double a[4][8], b[4][8];
            auto fetch = [&](int set, int batch) {
                const int t = 8 * ((batch < kRollTile / 8) ? batch : (kRollTile / 8 - 1));
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (V >= 1) {
                        a[set][u] = T.P[a0 + t + u][col];
                        b[set][u] = T.P[(b0 + t + u) & (4 * kRollTile - 1)][col];
                    } else {
                        a[set][u] = 1e-3 * (batch + u);
                        b[set][u] = 1e-3 * (batch + u) + 1e-5;
                    }
                }
            };
            auto run = [&](int set, int batch) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (V >= 2) {
                        S[8 * batch + u][col] = sum;
                    }
                    sum = (sum - a[set][u]) + b[set][u];
                }
            };
            fetch(0, 0);
            fetch(1, 1);
#pragma unroll 1
            for (int j = 0; j < kRollTile / 8; j += 4) {
                fetch(2, j + 2);
                run(0, j);
                fetch(3, j + 3);
                run(1, j + 1);
                fetch(0, j + 4);
                run(2, j + 2);
                fetch(1, j + 5);
                run(3, j + 3);
            }
        } else {
            // a helper lane of the product per tile: three ds_write_b64, five ds_read_b64, three divisions
            const int hl = threadIdx.x - 64, ht = hl & (kRollTile - 1), hr = hl / kRollTile;
            double *__restrict__ line = T.P[((k + 2) * kRollTile + ht) & (4 * kRollTile - 1)];
            const double rv = seed[(k + hl) & 1023];
            line[3 * hr] = rv;
            line[3 * hr + 1] = rv * rv;
            line[3 * hr + 2] = rv * 1.5;
            const double *__restrict__ sums = T.S[(k + 1) & 1][ht] + 3 * hr;
            const double sy = sums[0], ssq = sums[1], slag = sums[2];
            const double mean = sy / 31.0, g0 = ssq - 31.0 * mean * mean, g1 = slag - mean * sy;
            const double beta = g1 / (g0 * 1.03 + 1e-4), var = g0 / 31.0 * (1.0 - beta * beta);
            if (var == 12345.678) {
                out[64 + threadIdx.x] = var;
            }
        }
        if (V >= 3) {
            __syncthreads();
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (wave == 0) {
        out[lane] = sum + T.S[0][5][3];
        if (lane == 0) {
            stamps->cycles = c1 - c0;
            stamps->wall = w1 - w0;
        }
    }
}

// the candidate: line t holds, per chain, the PAIR (leaving(t), entering(t)) -- the helpers store every product twice --
// so a start position is one ds_read_b128; the sums of two start positions leave as one ds_write_b128
constexpr int kRollPairPitch = 50;  // doubles per line: 24 pairs + 2
struct RollPairs {
    double S[2][kRollTile / 2][kRollPairPitch];  // [t / 2][col][2]
    double P[2 * kRollTile][kRollPairPitch];     // [t][col][2] = (leaving, entering)
};

template <int LANES, bool HELPERS>
__global__ __launch_bounds__(576) void rolling_pairs(double *out, Stamps *stamps, int tiles, const double *seed)
{
    __shared__ __attribute__((aligned(16))) RollPairs T;
    for (int i = threadIdx.x; i < (int)(sizeof(T) / 8); i += blockDim.x) {
        reinterpret_cast<double *>(&T)[i] = seed[i & 1023];
    }
    __syncthreads();
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    const bool runs = lane < LANES;
    const int col = runs ? (lane % 24) : 0;
    double sum = 0.0;
    if (!HELPERS && wave > 0) {
        return;
    }
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
    }
    using d2 = __attribute__((ext_vector_type(2))) double;
    const long long c0 = clock64(), w0 = wall_clock64();
    if (wave == 0) {
        d2 ab[8], ab2[8], s[4], sp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sp[q] = d2{0.0, 0.0};
        }
        auto fetch = [&](d2(&x)[8], int k, int j) {
            const d2 *__restrict__ P = reinterpret_cast<const d2 *>(T.P[(k & 1) * kRollTile + 8 * j]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = P[u * (kRollPairPitch / 2) + col];
            }
        };
        auto run = [&](const d2(&x)[8]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double s0 = sum;
                const double s1 = (s0 - x[2 * u].x) + x[2 * u].y;
                sum = (s1 - x[2 * u + 1].x) + x[2 * u + 1].y;
                s[u] = d2{s0, s1};
            }
        };
        auto flush = [&](const d2(&x)[4], int k, int j) {
            d2 *__restrict__ S = reinterpret_cast<d2 *>(T.S[k & 1][4 * j]);
            if (runs) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    S[u * (kRollPairPitch / 2) + col] = x[u];
                }
            }
        };
        fetch(ab, 0, 0);
        for (int k = 0; k < tiles; ++k) {
#pragma unroll 1
            for (int j = 0; j < kRollTile / 8; j += 2) {
                WAIT_LGKM0();
                SCHED_FENCE();
                fetch(ab2, k, j + 1);
                flush(sp, k, j);
                SCHED_FENCE();
                run(ab);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sp[q] = s[q];
                }
                SCHED_FENCE();
                WAIT_LGKM0();
                SCHED_FENCE();
                fetch(ab, (j + 2 < kRollTile / 8) ? k : (k + 1), (j + 2) & (kRollTile / 8 - 1));
                flush(sp, k, j + 1);
                SCHED_FENCE();
                run(ab2);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sp[q] = s[q];
                }
                SCHED_FENCE();
            }
            if (HELPERS) {
                __syncthreads();
            }
        }
    } else {
        for (int k = 0; k < tiles; ++k) {
            const int hl = threadIdx.x - 64, ht = hl & (kRollTile - 1), hr = hl / kRollTile;
            using d2v = __attribute__((ext_vector_type(2))) double;
            d2v *__restrict__ line = reinterpret_cast<d2v *>(T.P[((k + 1) & 1) * kRollTile + ht]);
            const double rv = seed[(k + hl) & 1023];
            // (the product would write the entering halves 31 lines back: same count of stores)
            line[3 * hr] = d2v{rv, rv + 1.0};
            line[3 * hr + 1] = d2v{rv * rv, rv * rv + 1.0};
            line[3 * hr + 2] = d2v{rv * 1.5, rv * 1.5 + 1.0};
            const double *__restrict__ sums = &T.S[(k + 1) & 1][ht >> 1][2 * 3 * hr + (ht & 1)];
            const double sy = sums[0], ssq = sums[2], slag = sums[4];
            const double mean = sy / 31.0, g0 = ssq - 31.0 * mean * mean, g1 = slag - mean * sy;
            const double beta = g1 / (g0 * 1.03 + 1e-4), var = g0 / 31.0 * (1.0 - beta * beta);
            if (var == 12345.678) {
                out[64 + threadIdx.x] = var;
            }
            __syncthreads();
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (wave == 0) {
        out[lane] = sum + T.S[0][5][3];
        if (lane == 0) {
            stamps->cycles = c1 - c0;
            stamps->wall = w1 - w0;
        }
    }
}

// ---- dependent-operation latencies ------------------------------------------------------------------------------------------
template <int OP>
__global__ void dependent(double *out, Stamps *stamps, int iters, double a, double b)
{
    double x = out[threadIdx.x], y = out[threadIdx.x] + 1.0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (OP == 0) {
                x = x + a;  // one dependent addition
            } else if (OP == 1) {
                x = x * a;  // one dependent multiplication
            } else if (OP == 2) {
                x = (x - a) + b;  // the rolling update
            } else {
                const double r = chain_step(b, a, 0.9555, x, y);  // the substitution step
                y = x;
                x = r;
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) {
        stamps->cycles = c1 - c0;
        stamps->wall = w1 - w0;
    }
}

static double *g_out, *g_seed, *g_factor;
static Stamps *g_stamps;

template <typename F>
static void report(const char *what, double loci, F launch)
{
    Stamps h{};
    float best = 1e30f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipMemcpy(&h, g_stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-86s %7.2f ns  %7.2f cycles  (%.0f MHz)\n", what, 10.0 * h.wall / loci, h.cycles / loci, 100.0 * h.cycles / h.wall);
    fflush(stdout);
    (void)best;
}

int main()
{
    hipMalloc(&g_out, 1 << 16);
    hipMemset(g_out, 0, 1 << 16);
    hipMalloc(&g_stamps, sizeof(Stamps));
    std::vector<double> seed(1024), factor(8192 + 16);
    for (int i = 0; i < 1024; ++i) {
        seed[i] = 1e-3 * ((i * 37) % 101) - 0.05;
    }
    for (int i = 0; i < 4096 + 8; ++i) {
        factor[i] = -1.9545 - 1e-9 * i;
        factor[4096 + i] = 0.9555 + 1e-9 * i;
    }
    hipMalloc(&g_seed, 1024 * 8);
    hipMemcpy(g_seed, seed.data(), 1024 * 8, hipMemcpyHostToDevice);
    hipMalloc(&g_factor, factor.size() * 8);
    hipMemcpy(g_factor, factor.data(), factor.size() * 8, hipMemcpyHostToDevice);
    const int iters = 100000, tiles = 20000;
    const double ops = iters * 16.0, loci = (double)tiles * kTile;

    printf("-- dependent FP64 operations, one wavefront (64 lanes) alone on the chip, per operation / step\n");
    report("x = x + a", ops, [&] { hipLaunchKernelGGL(dependent<0>, dim3(1), dim3(64), 0, 0, g_out, g_stamps, iters, 1e-9, 0.3); });
    report("x = x * a", ops, [&] { hipLaunchKernelGGL(dependent<1>, dim3(1), dim3(64), 0, 0, g_out, g_stamps, iters, 1.0000001, 0.3); });
    report("x = (x - a) + b   [rolling update: 2 dependent]", ops,
           [&] { hipLaunchKernelGGL(dependent<2>, dim3(1), dim3(64), 0, 0, g_out, g_stamps, iters, 1e-9, 0.3); });
    report("r = v - c1 p1 - c2 p2   [substitution step: mul, sub, sub dependent + 1 mul beside]", ops,
           [&] { hipLaunchKernelGGL(dependent<3>, dim3(1), dim3(64), 0, 0, g_out, g_stamps, iters, -1.9545, 0.3); });

    printf("-- Whittaker substitution chain, per locus (whittaker_rows_kernel's chain wavefront, 16 chains unless said)\n");
#define WP(V, L, what) report(what, loci, [&] { hipLaunchKernelGGL((whittaker_product<V, L>), dim3(1), dim3(320), 0, 0, g_out, g_stamps, tiles, g_seed); })
    WP(0, 16, "W0 registers only");
    WP(0, 64, "W0 registers only, 64 lanes");
    WP(1, 16, "W1 + LDS operand reads (b64 value + b128 multipliers per locus, next batch fetched before this one runs)");
    WP(2, 16, "W2 + LDS result writes (b64 per locus)");
    WP(3, 16, "W3 + four helper wavefronts staging, barrier per 64 loci  [= the product's loop]");
    WP(3, 64, "W3 with 64 chains");
#define WN(C, L, H, what) report(what, loci, [&] { hipLaunchKernelGGL((whittaker_pairs<C, L, H>), dim3(1), dim3(320), 0, 0, g_out, g_stamps, tiles, g_seed, g_factor); })
    WN(0, 16, false, "N1 pairs: b128 values / results, b128 broadcast multipliers, explicit wait -> fetch -> chain");
    WN(0, 16, true, "N1 + helpers and barrier");
    WN(0, 64, true, "N1 + helpers and barrier, 64 chains");
    WN(1, 16, false, "N2 = N1 with the multipliers in scalar registers");
    WN(1, 16, true, "N2 + helpers and barrier");
    WN(1, 64, true, "N2 + helpers and barrier, 64 chains");

    printf("-- rolling sums chain, per start position (wls_rolling_rows_kernel<8>'s chain wavefront, 24 chains unless said)\n");
#define RP(V, L, what) report(what, loci, [&] { hipLaunchKernelGGL((rolling_product<V, L>), dim3(1), dim3(576), 0, 0, g_out, g_stamps, tiles, g_seed); })
    RP(0, 24, "R0 registers only");
    RP(1, 24, "R1 + LDS operand reads (two b64 per position, two batches ahead)");
    RP(2, 24, "R2 + LDS result writes (b64 per position)");
    RP(3, 24, "R3 + eight helper wavefronts (stores, variances with divisions), barrier per 64 positions  [= the product's loop]");
#define RN(L, H, what) report(what, loci, [&] { hipLaunchKernelGGL((rolling_pairs<L, H>), dim3(1), dim3(576), 0, 0, g_out, g_stamps, tiles, g_seed); })
    RN(24, false, "M1 pairs: one b128 (leaving, entering) per position, sums as b128 pairs, explicit wait -> fetch -> chain");
    RN(24, true, "M1 + helpers and barrier");
    RN(48, true, "M1 + helpers and barrier, 48 chains");
    return 0;
}
