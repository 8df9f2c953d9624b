// rocco_amd/csrc/normal.hip -- the multipliers of the budget null's bootstrap draws on the device (gfx950 only).
//
// What is replaced: rocco/inference.py:546-575 `_generate_dependent_wild_weights`, which the reference calls K times
// per draw of the count-matrix null (inference.py:654-664) and once per draw of the score-track null (1206-1213):
//     innovations = rng.standard_normal(n + taps - 1)            NumPy Generator over PCG64
//     weights     = fftconvolve(innovations, bartlett, "valid")  SciPy
//     weights     = (weights - mean) / std
// On a chromosome-sized matrix that host work is ~100x the draw's device work.
//
// 1. rocco_hip_pcg64_standard_normal_f64 reproduces `Generator.standard_normal` -- NumPy's ziggurat
//    (numpy/random/src/distributions/distributions.c: random_standard_normal) over PCG64 XSL-RR 128/64
//    (numpy/random/src/pcg64/pcg64.h) -- from the generator state the host hands over.  A value consumes ONE raw
//    64-bit draw 98.5 % of the time, two in a wedge (1.47 %), two per turn of the tail loop (0.026 %): the position of
//    every later value in the raw stream depends on what came before, which is why NumPy's loop is sequential.
//    Here every thread owns a chunk of 256 raw draws (its starting state by the LCG's O(log) jump-ahead) and walks it
//    as if a value began at the chunk's first draw.  That assumption is wrong only when the previous chunk's last value
//    spilled over the border, and a walk that starts one or two draws late falls into step with the assumed one at
//    the first draw both visit -- almost always at once.  So: walk and count (outputs, draws spilled into the next
//    chunk); hand every chunk the spill of its predecessor and let the few that assumed wrongly walk again, until no
//    chunk's spill changes (checked on the device: a chunk whose entry still disagrees with its predecessor's spill
//    raises a flag and the call fails -- never met); prefix-sum the output counts; walk once more, writing the values
//    where they belong.  The walks are integer multiplies, table look-ups and one compare; there is no stored raw stream.
//    Output: the values, and how many raw draws they consumed (the host's generator is advanced by that:
//    `bit_generator.advance`), so host and device draws interleave in one stream.
//    Bit for bit NumPy's values, except the tail (|x| > 3.654, 2.6e-4 of the values): NumPy takes log1p from the
//    host's libm, which is not correctly rounded and differs between its FMA / non-FMA builds; the device's log1p may
//    differ from it in the last place (measured: tests/test_gpu_normal.py).  The accept / reject comparisons only see
//    exp and log1p through inequalities that are never decided by the last bit, so the POSITIONS never differ.
// 2. rocco_hip_bartlett_multipliers_f64: the "valid" convolution of every row of innovations with the taps as a direct
//    sum in a fixed order (not SciPy's FFT: equal to ~1e-16 of the values' scale, not bit for bit), then centring and
//    scaling of every row by its mean and standard deviation (fixed-order trees).
#include "common.h"
#include "kernels.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include "ziggurat_tables.h"

#include <hipcub/hipcub.hpp>

namespace rocco {

namespace {

constexpr int kZigChunk = 256;    // raw draws per thread
constexpr int kZigThreads = 256;  // threads per workgroup
constexpr double kZigR = 3.6541528853610087963519472518;
constexpr double kZigInvR = 0.27366123732975827203338247596;

struct U128 {
    unsigned long long hi, lo;
};

__host__ __device__ __forceinline__ U128 mul128(U128 a, U128 b)
{
    U128 r;
#ifdef __HIP_DEVICE_COMPILE__
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
#else
    const unsigned __int128 p = (unsigned __int128)a.lo * b.lo;
    r.lo = (unsigned long long)p;
    r.hi = (unsigned long long)(p >> 64) + a.hi * b.lo + a.lo * b.hi;
#endif
    return r;
}

__host__ __device__ __forceinline__ U128 add128(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + ((r.lo < a.lo) ? 1ull : 0ull);
    return r;
}

__host__ __device__ __forceinline__ U128 pcg_mult() { return U128{0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull}; }

// state after `delta` steps of  s <- s * M + inc  (pcg64.h: pcg_advance_lcg_128)
__host__ __device__ inline U128 pcg_advance(U128 state, U128 inc, unsigned long long delta)
{
    U128 acc_mult{0ull, 1ull}, acc_plus{0ull, 0ull};
    U128 cur_mult = pcg_mult(), cur_plus = inc;
    while (delta > 0ull) {
        if (delta & 1ull) {
            acc_mult = mul128(acc_mult, cur_mult);
            acc_plus = add128(mul128(acc_plus, cur_mult), cur_plus);
        }
        cur_plus = mul128(add128(cur_mult, U128{0ull, 1ull}), cur_plus);
        cur_mult = mul128(cur_mult, cur_mult);
        delta >>= 1;
    }
    return add128(mul128(acc_mult, state), acc_plus);
}

struct Pcg {
    U128 state, inc;
    __device__ __forceinline__ unsigned long long next()
    {
        state = add128(mul128(state, pcg_mult()), inc);
        const unsigned long long x = state.hi ^ state.lo;
        const unsigned rot = (unsigned)(state.hi >> 58);
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }
    __device__ __forceinline__ double next_double() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct ZigTables {
    unsigned long long ki[256];
    double wi[256], fi[256];
};

__device__ __forceinline__ void load_tables(ZigTables &t, const unsigned long long *ki, const unsigned long long *wi,
                                            const unsigned long long *fi)
{
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        t.ki[i] = ki[i];
        t.wi[i] = __longlong_as_double((long long)wi[i]);
        t.fi[i] = __longlong_as_double((long long)fi[i]);
    }
    __syncthreads();
}

// One chunk: values start at raw draw `entry` of the chunk (0 unless the previous chunk spilled) and at every draw a
// value's attempt ends on, until the chunk's end.  EMIT: the values go to out[0 ...], at most `limit` of them; the raw
// draws used up to and including the limit-th value are reported.  Returns the number of values; `spill` = draws taken
// beyond the chunk's end.
// where the tail values of a fill went, for the host to compute them again with ITS log1p (see rocco_hip_pcg64_standard_normal_f64)
struct TailList {
    unsigned long long *count;  // values listed (may exceed capacity: then the list is incomplete)
    long long *index;           // position in the output | sign in bit 62
    double *u;                  // the uniform of the accepted turn's first draw
    long long capacity;
};

template <bool EMIT>
__device__ __forceinline__ int walk_chunk(Pcg g, int entry, const ZigTables &t, double *out, long long limit, int *spill,
                                          int *used_at_limit, const TailList *tails = nullptr, long long first_index = 0)
{
    int pos = entry, n_out = 0;
    while (pos < kZigChunk) {
        unsigned long long r = g.next();
        ++pos;
        const int idx = (int)(r & 0xFFull);
        r >>= 8;
        const bool negative = (r & 1ull) != 0ull;
        const unsigned long long rabs = (r >> 1) & 0x000FFFFFFFFFFFFFull;
        double x = (double)rabs * t.wi[idx];
        x = negative ? -x : x;
        bool have = rabs < t.ki[idx];
        if (!have) {
            if (idx == 0) {
                // tail: NumPy's loop, two draws per turn
                for (;;) {
                    const double u1 = g.next_double();
                    const double xx = -kZigInvR * log1p(-u1);
                    const double yy = -log1p(-g.next_double());
                    pos += 2;
                    if (yy + yy > xx * xx) {
                        const bool minus = ((rabs >> 8) & 1ull) != 0ull;
                        x = minus ? -(kZigR + xx) : kZigR + xx;
                        if (EMIT && tails != nullptr && (long long)n_out < limit) {
                            const unsigned long long slot = atomicAdd(tails->count, 1ull);
                            if ((long long)slot < tails->capacity) {
                                tails->index[slot] = (first_index + n_out) | (minus ? (1LL << 62) : 0LL);
                                tails->u[slot] = u1;
                            }
                        }
                        break;
                    }
                }
                have = true;
            } else {
                const double u = g.next_double();
                ++pos;
                have = (t.fi[idx - 1] - t.fi[idx]) * u + t.fi[idx] < exp(-0.5 * x * x);
            }
        }
        if (have) {
            if (EMIT) {
                if ((long long)n_out < limit) {
                    out[n_out] = x;
                    if ((long long)n_out + 1 == limit) {
                        *used_at_limit = pos;
                    }
                }
            }
            ++n_out;
        }
    }
    *spill = pos - kZigChunk;
    return n_out;
}

struct ZigArgs {
    U128 state, inc;
    long long n_chunks;
    const unsigned long long *ki, *wi, *fi;
    int *entry;        // [n_chunks] raw draws of the chunk that belong to the previous chunk's last value
    int *n_out;        // [n_chunks]
    int *spill_a;      // [n_chunks] double-buffered spills
    int *spill_b;
    unsigned *flag;    // [0]: chunks that walked again in this pass; [1]: a chunk's entry disagrees with its predecessor
};

__global__ __launch_bounds__(kZigThreads) void zig_count_kernel(ZigArgs a)
{
    __shared__ ZigTables t;
    load_tables(t, a.ki, a.wi, a.fi);
    const long long c = (long long)blockIdx.x * kZigThreads + threadIdx.x;
    if (c >= a.n_chunks) {
        return;
    }
    Pcg g{pcg_advance(a.state, a.inc, (unsigned long long)c * kZigChunk), a.inc};
    int spill = 0, unused = 0;
    a.n_out[c] = walk_chunk<false>(g, 0, t, nullptr, 0, &spill, &unused);
    a.entry[c] = 0;
    a.spill_a[c] = spill;
}

// chunks whose predecessor spilled otherwise than they assumed walk again (spills read from `in`, written to `out`)
__global__ __launch_bounds__(kZigThreads) void zig_fix_kernel(ZigArgs a, const int *__restrict__ in, int *__restrict__ out,
                                                              int verify)
{
    __shared__ ZigTables t;
    load_tables(t, a.ki, a.wi, a.fi);
    const long long c = (long long)blockIdx.x * kZigThreads + threadIdx.x;
    if (c >= a.n_chunks) {
        return;
    }
    const int want = (c == 0) ? 0 : in[c - 1];
    if (want == a.entry[c]) {
        if (!verify) {
            out[c] = in[c];
        }
        return;
    }
    if (verify) {
        atomicOr(&a.flag[1], 1u);
        return;
    }
    int spill = 0, unused = 0;
    if (want >= kZigChunk) {
        // (a tail loop that ran across a whole chunk: this chunk starts no value at all)
        a.n_out[c] = 0;
        spill = want - kZigChunk;
    } else {
        Pcg g{pcg_advance(a.state, a.inc, (unsigned long long)c * kZigChunk + (unsigned long long)want), a.inc};
        a.n_out[c] = walk_chunk<false>(g, want, t, nullptr, 0, &spill, &unused);
    }
    a.entry[c] = want;
    out[c] = spill;
    atomicAdd(&a.flag[0], 1u);
}

__global__ __launch_bounds__(kZigThreads) void zig_fill_kernel(ZigArgs a, const long long *__restrict__ offsets, long long count,
                                                               double *__restrict__ values, unsigned long long *consumed, TailList tails)
{
    __shared__ ZigTables t;
    load_tables(t, a.ki, a.wi, a.fi);
    const long long c = (long long)blockIdx.x * kZigThreads + threadIdx.x;
    if (c >= a.n_chunks) {
        return;
    }
    const long long first = offsets[c];
    const int entry = a.entry[c];
    if (first >= count || a.n_out[c] == 0 || entry >= kZigChunk) {
        return;
    }
    Pcg g{pcg_advance(a.state, a.inc, (unsigned long long)c * kZigChunk + (unsigned long long)entry), a.inc};
    int spill = 0, used = -1;
    walk_chunk<true>(g, entry, t, values + first, count - first, &spill, &used, (tails.capacity > 0) ? &tails : nullptr, first);
    if (used >= 0) {
        // this chunk holds the last value asked for: the raw draws up to the end of its attempt
        *consumed = (unsigned long long)c * kZigChunk + (unsigned long long)used;
    }
}

__global__ void scatter_tail_kernel(const long long *__restrict__ index, const double *__restrict__ value, long long n, double *__restrict__ values)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        values[index[i] & ~(1LL << 62)] = value[i];
    }
}

__global__ void widen_counts_kernel(const int *__restrict__ n_out, long long *__restrict__ wide, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        wide[i] = n_out[i];
    }
}

// ---- Bartlett smoothing + standardisation -------------------------------------------------------------------
// Round 5: every thread owns EIGHT CONSECUTIVE outputs and walks the taps eight at a time with its inputs in a sliding
// window of sixteen registers: per eight taps and eight outputs one wavefront issues 4 ds_read_b128 and 64 v_fma_f64 with
// the taps as scalar operands (round 4's kernel: five ds_read_b64 and eight VALU instructions per tap for four outputs --
// LDS-issue-bound at a tenth of the FP64 rate).  The inputs of a tile sit in LDS in chunks of 8 doubles at a pitch of 10:
// thread t reads chunk t + block, 80 bytes from its neighbour's -- conflict-free b128 reads.  Each output is the same sum
// in the same order as before (taps ascending), now with one rounding per term (v_fma_f64) instead of two; the multipliers
// were never the reference's bits (it convolves by FFT: rocco/inference.py:561), the tests hold them to 1e-12 of it.
constexpr int kConvThreads = 256;
constexpr int kConvPerThread = 8;
constexpr int kConvTile = kConvThreads * kConvPerThread;  // outputs per workgroup
constexpr int kConvMaxTapsLds = 2048;
constexpr int kConvPitch = 10;  // doubles per chunk of 8 in LDS

typedef double conv2_t __attribute__((ext_vector_type(2)));

// weights[row][j] = sum_{t = 0}^{n_taps - 1} innovations[row][j + t] * taps[n_taps - 1 - t]   (np.convolve "valid")
// rev: the taps reversed, padded with zeros to `padded` (a multiple of 8) entries
__global__ __launch_bounds__(kConvThreads) void bartlett_conv_kernel(const double *__restrict__ innovations, long long row_stride,
                                                                    long long n, const double *__restrict__ rev, int n_taps, int padded,
                                                                    double *__restrict__ weights)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];  // (kConvThreads + padded / 8) chunks of kConvPitch doubles
    const long long row = blockIdx.y;
    const long long j0 = (long long)blockIdx.x * kConvTile;
    const double *src = innovations + row * row_stride;
    const long long have = (n + n_taps - 1) - j0;  // inputs of this row from j0 on
    const int n_in = kConvTile + padded;
    for (int i = threadIdx.x; i < n_in; i += kConvThreads) {
        lds[(i >> 3) * kConvPitch + (i & 7)] = (i < have) ? src[j0 + i] : 0.0;
    }
    __syncthreads();
    // Three chunks of inputs (this block's two + the one the next block adds) and two blocks of taps (uniform: scalar
    // registers) are in registers at any time, in three sets that take turns: what a block needs was fetched while the block
    // before it ran its 64 FMAs, so no block starts by waiting for LDS or for the scalar cache.
    double acc[kConvPerThread], c0[8], c1[8], c2[8], t0[8], t1[8], t2[8];
    const int blocks = padded / 8;
    auto fetch = [&](int chunk, double(&into)[8]) {
        const conv2_t *at = reinterpret_cast<const conv2_t *>(lds + chunk * kConvPitch);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const conv2_t v = at[q];
            into[2 * q] = v.x;
            into[2 * q + 1] = v.y;
        }
    };
    auto taps_of = [&](int tb, double(&into)[8]) {
        const double *__restrict__ at = rev + 8 * ((tb < blocks) ? tb : (blocks - 1));
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            into[tt] = at[tt];
        }
    };
#pragma unroll
    for (int k = 0; k < kConvPerThread; ++k) {
        acc[k] = 0.0;
    }
    // block tb: inputs lo = chunk (thread + tb), hi = chunk (thread + tb + 1), taps w; fetches chunk (thread + tb + 2) and
    // the taps of block tb + 1 for its successor (the chunk index stays inside the staged tile: see n_in)
    auto block = [&](int tb, const double(&lo)[8], const double(&hi)[8], double(&ahead)[8], const double(&w)[8], double(&w_ahead)[8]) {
        // (an explicit wait for what the block before fetched, BEFORE this block's fetches go out: scalar loads return out
        // of order, so a wait the compiler places behind them is a wait for them too)
        __builtin_amdgcn_s_waitcnt(0xC07F);  // s_waitcnt lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        fetch((int)threadIdx.x + ((tb + 2 <= blocks) ? (tb + 2) : blocks), ahead);
        taps_of(tb + 1, w_ahead);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
#pragma unroll
            for (int k = 0; k < kConvPerThread; ++k) {
                const double x = (tt + k < 8) ? lo[tt + k] : hi[tt + k - 8];
                acc[k] = __builtin_fma(x, w[tt], acc[k]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch((int)threadIdx.x, c0);
    fetch((int)threadIdx.x + 1, c1);
    taps_of(0, t0);
    for (int tb = 0; tb < blocks; tb += 3) {
        block(tb, c0, c1, c2, t0, t1);
        if (tb + 1 < blocks) {
            block(tb + 1, c1, c2, c0, t1, t2);
        }
        if (tb + 2 < blocks) {
            block(tb + 2, c2, c0, c1, t2, t0);
        }
    }
    const long long j = j0 + (long long)threadIdx.x * kConvPerThread;
#pragma unroll
    for (int k = 0; k < kConvPerThread; ++k) {
        if (j + k < n) {
            weights[row * n + j + k] = acc[k];
        }
    }
}

// taps too many for LDS (tracks of a few hundred loci with a bandwidth near their length): straight from memory
__global__ __launch_bounds__(kConvThreads) void bartlett_conv_slow_kernel(const double *__restrict__ innovations, long long row_stride,
                                                                         long long n, const double *__restrict__ taps, int n_taps,
                                                                         double *__restrict__ weights)
{
    const long long row = blockIdx.y;
    const long long j = (long long)blockIdx.x * kConvThreads + threadIdx.x;
    if (j >= n) {
        return;
    }
    const double *src = innovations + row * row_stride + j;
    double acc = 0.0;
    for (int t = 0; t < n_taps; ++t) {
        acc = __builtin_fma(src[t], taps[n_taps - 1 - t], acc);  // (as the tiled kernel: one rounding per term)
    }
    weights[row * n + j] = acc;
}

constexpr int kMomentSegment = 8192;

// partial[row][segment] = sum over the segment of (x - shift)^power, fixed order: 256 lanes x 32 strided terms, then a tree
template <int POWER>
__global__ __launch_bounds__(256) void row_moment_partial_kernel(const double *__restrict__ x, long long n, const double *__restrict__ shift,
                                                                 double *__restrict__ partial, int segments)
{
    __shared__ double red[256];
    const long long row = blockIdx.y;
    const int seg = blockIdx.x;
    const double s = (shift != nullptr) ? shift[row] : 0.0;
    const double *src = x + row * n;
    const long long base = (long long)seg * kMomentSegment;
    double acc = 0.0;
    for (int k = 0; k < kMomentSegment / 256; ++k) {
        const long long j = base + (long long)k * 256 + threadIdx.x;
        if (j < n) {
            const double d = src[j] - s;
            acc += (POWER == 1) ? d : d * d;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[threadIdx.x] += red[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[row * segments + seg] = red[0];
    }
}

// out[row] = (sum of the row's partials, in segment order) / n, optionally its square root
__global__ void row_moment_final_kernel(const double *__restrict__ partial, int segments, long long n, int root, double *__restrict__ out,
                                        long long rows)
{
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) {
        return;
    }
    double acc = 0.0;
    for (int s = 0; s < segments; ++s) {
        acc += partial[row * segments + s];
    }
    acc /= (double)n;
    out[row] = root ? sqrt(acc) : acc;
}

__global__ __launch_bounds__(256) void row_standardise_kernel(double *__restrict__ x, long long n, const double *__restrict__ mean,
                                                              const double *__restrict__ sd, int *__restrict__ degenerate)
{
    const long long row = blockIdx.y;
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    const double s = sd[row];
    if (!(isfinite(s) && s > 1.0e-8)) {  // rocco/inference.py:565: the reference then draws signs instead
        if (j == 0) {
            atomicOr(degenerate, 1);
        }
        return;
    }
    if (j < n) {
        x[row * n + j] = (x[row * n + j] - mean[row]) / s;
    }
}

}  // namespace

}  // namespace rocco

using namespace rocco;

extern "C" {

int rocco_hip_pcg64_standard_normal_f64(rocco_hip_solver *solver, unsigned long long state_hi, unsigned long long state_lo,
                                        unsigned long long inc_hi, unsigned long long inc_lo, size_t count, double *values_dev,
                                        unsigned long long *raw_draws_out, void *stream_)
{
    if (solver == nullptr || (values_dev == nullptr && count > 0) || raw_draws_out == nullptr) {
        return ROCCO_HIP_EINVAL;
    }
    *raw_draws_out = 0ull;
    if (count == 0) {
        return ROCCO_HIP_OK;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    hipStream_t stream = (hipStream_t)stream_;
    // a value takes 1.022 raw draws on average; 3 % + a chunk's worth of slack, and the fill checks that it sufficed
    const long long raws = (long long)((double)count * 1.03) + 16 * kZigChunk;
    const long long n_chunks = (raws + kZigChunk - 1) / kZigChunk;
    size_t off = 0;
    auto carve = [&off](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) / 256 * 256;
        return at;
    };
    const size_t o_tables = carve(3 * 256 * sizeof(unsigned long long));
    const size_t o_entry = carve((size_t)n_chunks * sizeof(int));
    const size_t o_nout = carve((size_t)n_chunks * sizeof(int));
    const size_t o_sa = carve((size_t)n_chunks * sizeof(int));
    const size_t o_sb = carve((size_t)n_chunks * sizeof(int));
    const size_t o_wide = carve((size_t)n_chunks * sizeof(long long));
    const size_t o_offsets = carve((size_t)n_chunks * sizeof(long long));
    const size_t o_flag = carve(4 * sizeof(unsigned));
    const size_t o_consumed = carve(sizeof(unsigned long long));
    // The tail of the ziggurat (|x| > 3.654: one value in 3 800) is -log1p(-u) / r + r, and NumPy takes log1p from the host's
    // libm -- not correctly rounded, and not the same function on every host (glibc picks an FMA build where the CPU has
    // one) -- so no device log1p gives NumPy's bits everywhere.  The fill lists where its tail values went and the uniform
    // each came from; the host computes them again with ITS log1p -- the function the NumPy of this host calls -- and a
    // small kernel puts them in place: the device's normals are then this host's NumPy's, bit for bit (round 4: 130 of
    // 10^8 values one ulp off).  ROCCO_HIP_NORMAL_TAIL=device keeps the device's log1p (one synchronisation less).
    const char *tail_mode = std::getenv("ROCCO_HIP_NORMAL_TAIL");
    const bool host_tails = !(tail_mode != nullptr && std::strcmp(tail_mode, "device") == 0);
    const long long tail_capacity = host_tails ? ((long long)(count / 500) + 65536) : 0;
    const size_t o_tail_count = carve(sizeof(unsigned long long));
    const size_t o_tail_index = carve((size_t)tail_capacity * sizeof(long long));
    const size_t o_tail_u = carve((size_t)tail_capacity * sizeof(double));
    size_t scan_bytes = 0;
    ROCCO_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const long long *)nullptr, (long long *)nullptr, (int)n_chunks, stream));
    const size_t o_scan = carve(scan_bytes + 256);
    int rc;
    if ((rc = solver->dev_params.reserve(off)) != ROCCO_HIP_OK) return rc;
    char *dv = (char *)solver->dev_params.ptr;
    if ((rc = solver->host_stage.reserve(3 * 256 * sizeof(unsigned long long))) != ROCCO_HIP_OK) return rc;
    std::memcpy(solver->host_stage.ptr, kZigguratKi, 256 * sizeof(unsigned long long));
    std::memcpy((char *)solver->host_stage.ptr + 2048, kZigguratWiBits, 256 * sizeof(unsigned long long));
    std::memcpy((char *)solver->host_stage.ptr + 4096, kZigguratFiBits, 256 * sizeof(unsigned long long));
    ROCCO_HIP_TRY(hipMemcpyAsync(dv + o_tables, solver->host_stage.ptr, 3 * 2048, hipMemcpyHostToDevice, stream));
    ROCCO_HIP_TRY(hipMemsetAsync(dv + o_flag, 0, 4 * sizeof(unsigned) + 256 + sizeof(unsigned long long), stream));
    ROCCO_HIP_TRY(hipMemsetAsync(dv + o_tail_count, 0, sizeof(unsigned long long), stream));
    TailList tails;
    tails.count = (unsigned long long *)(dv + o_tail_count);
    tails.index = (long long *)(dv + o_tail_index);
    tails.u = (double *)(dv + o_tail_u);
    tails.capacity = tail_capacity;
    ZigArgs a;
    a.state = U128{state_hi, state_lo};
    a.inc = U128{inc_hi, inc_lo};
    a.n_chunks = n_chunks;
    a.ki = (const unsigned long long *)(dv + o_tables);
    a.wi = a.ki + 256;
    a.fi = a.ki + 512;
    a.entry = (int *)(dv + o_entry);
    a.n_out = (int *)(dv + o_nout);
    a.spill_a = (int *)(dv + o_sa);
    a.spill_b = (int *)(dv + o_sb);
    a.flag = (unsigned *)(dv + o_flag);
    const unsigned blocks = (unsigned)((n_chunks + kZigThreads - 1) / kZigThreads);
    hipLaunchKernelGGL(zig_count_kernel, dim3(blocks), dim3(kZigThreads), 0, stream, a);
    // spills settle after one pass unless a corrected walk ends on another draw than the assumed one (it merges with it
    // at the first draw both visit: almost always at once); three passes, then a check
    int *in = a.spill_a, *out = a.spill_b;
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(zig_fix_kernel, dim3(blocks), dim3(kZigThreads), 0, stream, a, (const int *)in, out, 0);
        std::swap(in, out);
    }
    hipLaunchKernelGGL(zig_fix_kernel, dim3(blocks), dim3(kZigThreads), 0, stream, a, (const int *)in, out, 1);
    hipLaunchKernelGGL(widen_counts_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, stream, (const int *)a.n_out,
                       (long long *)(dv + o_wide), n_chunks);
    ROCCO_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(dv + o_scan, scan_bytes, (const long long *)(dv + o_wide), (long long *)(dv + o_offsets),
                                                   (int)n_chunks, stream));
    hipLaunchKernelGGL(zig_fill_kernel, dim3(blocks), dim3(kZigThreads), 0, stream, a, (const long long *)(dv + o_offsets), (long long)count,
                       values_dev, (unsigned long long *)(dv + o_consumed), tails);
    ROCCO_HIP_TRY(hipGetLastError());
    if ((rc = solver->host_back.reserve(64 + (size_t)tail_capacity * 16)) != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipMemcpyAsync(solver->host_back.ptr, dv + o_flag, 16, hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync((char *)solver->host_back.ptr + 16, dv + o_consumed, 8, hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync((char *)solver->host_back.ptr + 24, dv + o_tail_count, 8, hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    const unsigned *flag = (const unsigned *)solver->host_back.ptr;
    const unsigned long long consumed = *(const unsigned long long *)((const char *)solver->host_back.ptr + 16);
    const unsigned long long n_tail = *(const unsigned long long *)((const char *)solver->host_back.ptr + 24);
    if (host_tails && n_tail > 0ull) {
        if ((long long)n_tail > tail_capacity) {
            set_last_error("rocco_hip_pcg64_standard_normal_f64: more tail values than one in 500 (not a ziggurat stream?)");
            return ROCCO_HIP_EHIP;
        }
        long long *idx_host = (long long *)((char *)solver->host_back.ptr + 64);
        double *val_host = (double *)((char *)solver->host_back.ptr + 64 + (size_t)tail_capacity * 8);
        ROCCO_HIP_TRY(hipMemcpyAsync(idx_host, dv + o_tail_index, (size_t)n_tail * 8, hipMemcpyDeviceToHost, stream));
        ROCCO_HIP_TRY(hipMemcpyAsync(val_host, dv + o_tail_u, (size_t)n_tail * 8, hipMemcpyDeviceToHost, stream));
        ROCCO_HIP_TRY(hipStreamSynchronize(stream));
        for (unsigned long long i = 0; i < n_tail; ++i) {
            // numpy/random/src/distributions/distributions.c, random_standard_normal: the accepted turn's value
            const double xx = -kZigInvR * std::log1p(-val_host[i]);
            val_host[i] = (idx_host[i] & (1LL << 62)) ? -(kZigR + xx) : (kZigR + xx);
        }
        ROCCO_HIP_TRY(hipMemcpyAsync(dv + o_tail_u, val_host, (size_t)n_tail * 8, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(scatter_tail_kernel, dim3((unsigned)((n_tail + 255) / 256)), dim3(256), 0, stream, (const long long *)(dv + o_tail_index),
                           (const double *)(dv + o_tail_u), (long long)n_tail, values_dev);
        ROCCO_HIP_TRY(hipGetLastError());
        ROCCO_HIP_TRY(hipStreamSynchronize(stream));  // (the staging area is the solver's)
    }
    if (flag[1] != 0u) {
        set_last_error("rocco_hip_pcg64_standard_normal_f64: the chunk walks did not settle (a tail loop ran across chunks)");
        return ROCCO_HIP_EHIP;
    }
    if (consumed == 0ull) {
        set_last_error("rocco_hip_pcg64_standard_normal_f64: the raw draws set aside did not yield the values asked for");
        return ROCCO_HIP_EHIP;
    }
    *raw_draws_out = consumed;
    return ROCCO_HIP_OK;
}

int rocco_hip_bartlett_multipliers_f64(rocco_hip_solver *solver, const double *innovations_dev, size_t rows, size_t n,
                                       const double *taps_host, size_t n_taps, double *weights_dev, int *degenerate_out,
                                       void *stream_)
{
    if (solver == nullptr || innovations_dev == nullptr || weights_dev == nullptr || taps_host == nullptr || degenerate_out == nullptr ||
        rows == 0 || n == 0 || n_taps == 0 || n_taps > (size_t)0x7FFFFFFF || rows > 65535) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    hipStream_t stream = (hipStream_t)stream_;
    const int segments = (int)((n + kMomentSegment - 1) / kMomentSegment);
    size_t off = 0;
    auto carve = [&off](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) / 256 * 256;
        return at;
    };
    const size_t padded = (n_taps + 7) / 8 * 8;
    const size_t o_taps = carve(n_taps * sizeof(double));
    const size_t o_rev = carve(padded * sizeof(double));
    const size_t o_part = carve(rows * (size_t)segments * sizeof(double));
    const size_t o_mean = carve(rows * sizeof(double));
    const size_t o_sd = carve(rows * sizeof(double));
    const size_t o_flag = carve(sizeof(int));
    int rc;
    if ((rc = solver->dev_misc.reserve(off)) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_stage.reserve((n_taps + padded) * sizeof(double))) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_back.reserve(64)) != ROCCO_HIP_OK) return rc;
    char *dv = (char *)solver->dev_misc.ptr;
    std::memcpy(solver->host_stage.ptr, taps_host, n_taps * sizeof(double));
    double *rev_host = (double *)solver->host_stage.ptr + n_taps;  // reversed, zeros behind the last one
    for (size_t t = 0; t < padded; ++t) {
        rev_host[t] = (t < n_taps) ? taps_host[n_taps - 1 - t] : 0.0;
    }
    ROCCO_HIP_TRY(hipMemcpyAsync(dv + o_taps, solver->host_stage.ptr, n_taps * sizeof(double), hipMemcpyHostToDevice, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync(dv + o_rev, rev_host, padded * sizeof(double), hipMemcpyHostToDevice, stream));
    ROCCO_HIP_TRY(hipMemsetAsync(dv + o_flag, 0, sizeof(int), stream));
    const long long row_stride = (long long)(n + n_taps - 1);
    if (n_taps <= (size_t)kConvMaxTapsLds) {
        const size_t lds = ((size_t)kConvThreads + padded / 8 + 1) * kConvPitch * sizeof(double);
        hipLaunchKernelGGL(bartlett_conv_kernel, dim3((unsigned)((n + kConvTile - 1) / kConvTile), (unsigned)rows), dim3(kConvThreads), lds,
                           stream, innovations_dev, row_stride, (long long)n, (const double *)(dv + o_rev), (int)n_taps, (int)padded, weights_dev);
    } else {
        hipLaunchKernelGGL(bartlett_conv_slow_kernel, dim3((unsigned)((n + kConvThreads - 1) / kConvThreads), (unsigned)rows),
                           dim3(kConvThreads), 0, stream, innovations_dev, row_stride, (long long)n, (const double *)(dv + o_taps),
                           (int)n_taps, weights_dev);
    }
    double *partial = (double *)(dv + o_part), *mean = (double *)(dv + o_mean), *sd = (double *)(dv + o_sd);
    hipLaunchKernelGGL(row_moment_partial_kernel<1>, dim3((unsigned)segments, (unsigned)rows), dim3(256), 0, stream,
                       (const double *)weights_dev, (long long)n, (const double *)nullptr, partial, segments);
    hipLaunchKernelGGL(row_moment_final_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, stream, (const double *)partial, segments,
                       (long long)n, 0, mean, (long long)rows);
    hipLaunchKernelGGL(row_moment_partial_kernel<2>, dim3((unsigned)segments, (unsigned)rows), dim3(256), 0, stream,
                       (const double *)weights_dev, (long long)n, (const double *)mean, partial, segments);
    hipLaunchKernelGGL(row_moment_final_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, stream, (const double *)partial, segments,
                       (long long)n, 1, sd, (long long)rows);
    hipLaunchKernelGGL(row_standardise_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0, stream, weights_dev,
                       (long long)n, (const double *)mean, (const double *)sd, (int *)(dv + o_flag));
    ROCCO_HIP_TRY(hipGetLastError());
    ROCCO_HIP_TRY(hipMemcpyAsync(solver->host_back.ptr, dv + o_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    *degenerate_out = *(const int *)solver->host_back.ptr;
    return ROCCO_HIP_OK;
}

}  // extern "C"
