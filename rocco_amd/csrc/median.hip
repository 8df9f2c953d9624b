// rocco_amd/csrc/median.hip -- column-wise median over K samples (K x n row-major -> n), gfx950.
//
// Replaces rocco/rocco.py:264-265 (np.median(chrom_matrix, axis=0) inside
// score_central_tendency_chrom) as called from rocco/rocco.py:983-991.
//
// One lane owns one locus: it reads its K samples (row k is a coalesced 512-byte segment per
// wavefront), keeps them in registers, and runs a Batcher merge-exchange network whose compare
// indices are all compile-time constants.  Only the two middle outputs are consumed, so the
// compiler prunes every min/max that cannot reach them (a selection network, not a full sort).
// K is padded to a supported even size with -inf/+inf in equal numbers (one extra +inf for odd
// K), which leaves the middle order statistics unchanged.  HBM-bound: 8K (or 4K) bytes read and
// 8 bytes written per locus.
#include "kernels.h"

#include <limits>

#ifndef ROCCO_MEDIAN_NT
#define ROCCO_MEDIAN_NT 0  // 1: non-temporal loads of the rows of an exact-size column.  Measured on MI355X (same box,
                           // alternating libraries): 1-5 % faster for one isolated launch, 5-10 % SLOWER and noisy in a stream of
                           // launches -- off
#endif

namespace rocco {

namespace {

template <int N>
__device__ __forceinline__ void select_middle(double (&v)[N])
{
    // Batcher's merge exchange for arbitrary N; every index below is a compile-time constant
    // once the loops are fully unrolled.
#pragma unroll
    for (int p = 1; p < N; p <<= 1) {
#pragma unroll
        for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
            for (int j = k % p; j <= N - 1 - k; j += 2 * k) {
#pragma unroll
                for (int i = 0; i <= ((k - 1 < N - j - k - 1) ? (k - 1) : (N - j - k - 1)); ++i) {
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const double a = v[i + j];
                        const double b = v[i + j + k];
                        v[i + j] = fmin(a, b);
                        v[i + j + k] = fmax(a, b);
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ bool is_nan_bits(double x)
{
    // this file is compiled with -fno-honor-nans (so that the network is bare v_min_f64 / v_max_f64);
    // NaN tests therefore go through the bit pattern
    return (__double_as_longlong(x) & 0x7FFFFFFFFFFFFFFFLL) > 0x7FF0000000000000LL;
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Rows are not aligned to
// cache lines (n is arbitrary), so neighbouring workgroups share the line that straddles their
// boundary in every row: give each XCD one contiguous range of loci so that the shared lines meet
// in one L2 instead of being fetched from memory twice.
__device__ __forceinline__ unsigned xcd_contiguous_block()
{
    const unsigned nblk = gridDim.x;
    const unsigned per = nblk / 8U, rem = nblk % 8U;
    const unsigned xcd = blockIdx.x % 8U, slot = blockIdx.x / 8U;
    // XCD x owns per + (x < rem) workgroups, laid out one range after the other
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

// EXACT: K == KP is known at compile time (no padding, unpredicated loads).
// `col0` is the workgroup's first column and `lane` the thread's distance from it: every row is read at (scalar row
// base) + (one 32-bit vector offset), so the load phase costs no vector-ALU instruction per row and no address
// registers -- a wavefront issues its loads while its SIMD partner saturates the FP64 pipe with its network.
template <typename T, int KP, bool EXACT>
__device__ __forceinline__ double median_column(const T *__restrict__ m, int K_runtime, long long n, long long stride,
                                                double *__restrict__ out, long long col0, unsigned lane)
{
    const long long j = col0 + lane;
    if (j >= n) {
        return 0.0;
    }
    const int K = EXACT ? KP : K_runtime;
    const int pad = KP - K;
    const int n_lo = pad / 2;  // -inf entries; the remaining pad entries are +inf
    double v[KP];
    double sum = 0.0;  // NaN in the column <=> NaN sum (or +inf and -inf together: checked below)
    // every load unconditional (a load under a per-row condition is waited for before the next is issued): slots past
    // the matrix re-read its last row (a cache hit) and are replaced by the padding afterwards
    const T *row = m + col0;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (EXACT) {
#if ROCCO_MEDIAN_NT
            v[k] = (double)__builtin_nontemporal_load(row + lane);
#else
            v[k] = (double)row[lane];
#endif
            row += stride;
        } else {
            v[k] = (double)row[lane];
            row += (k + 1 < K) ? stride : 0;
        }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (EXACT || k < K) {
            sum += v[k];
        } else {
            v[k] = (k - K < n_lo) ? -std::numeric_limits<double>::infinity()
                                  : std::numeric_limits<double>::infinity();
        }
    }
    bool has_nan = false;
    if (is_nan_bits(sum)) {  // rare: find out whether a real NaN is present, park NaNs at +inf
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (EXACT || k < K) {
                const bool bad = is_nan_bits(v[k]);
                has_nan |= bad;
                v[k] = bad ? std::numeric_limits<double>::infinity() : v[k];
            }
        }
    }
    select_middle<KP>(v);
    double r;
    if (K & 1) {
        r = v[KP / 2 - 1];
    } else {
        r = (v[KP / 2 - 1] + v[KP / 2]) / 2.0;
    }
    r = has_nan ? __longlong_as_double(0x7FF8000000000000LL) : r;
    out[j] = r;
    return r;
}

template <typename T, int KP, bool EXACT>
__global__ __launch_bounds__(256) void median_kernel(const T *__restrict__ m, int K_runtime, long long n,
                                                     long long stride, double *__restrict__ out)
{
    median_column<T, KP, EXACT>(m, K_runtime, n, stride, out, (long long)xcd_contiguous_block() * 256, threadIdx.x);
}

// Several matrices of the same K and element type in ONE launch (the chromosomes of a rank): a launch starts
// with every workgroup loading and then every workgroup computing -- a transient of about 0.15 ms that a
// launch per chromosome pays 24 times per genome and this one once.
template <typename T, int KP, bool EXACT>
__global__ __launch_bounds__(256) void median_batch_kernel(MedianBatch batch)
{
    const unsigned logical = xcd_contiguous_block();
    // the task of this workgroup: how many tasks begin at or before it (unused slots begin at 0xFFFFFFFF) -- compares
    // on values that arrive with three wide scalar loads, no dependent load per task
    int ti = -1;
#pragma unroll
    for (int i = 0; i < kMedianBatchMax; ++i) {
        ti += (batch.block_begin[i] <= logical) ? 1 : 0;
    }
    const MedianTask &task = batch.tasks[ti];
    const long long col0 = (long long)(logical - task.block_begin) * 256;
    const long long j = col0 + threadIdx.x;
    const double r = median_column<T, KP, EXACT>((const T *)task.matrix, batch.K, task.n, task.stride, task.out, col0, threadIdx.x);
    if (task.partials != nullptr) {
        // min / max / sum |.| of this WAVEFRONT's scores (fmin / fmax skip NaN as the solve's own statistics pass
        // does, the sum carries it): the budgeted solve starts from these instead of reading the scores again.
        // No barrier: a workgroup-wide reduction would hold every wavefront of the workgroup until its slowest.
        const bool valid = j < task.n;
        const bool nan = valid && is_nan_bits(r);
        const double inf = std::numeric_limits<double>::infinity();
        double mn = (valid && !nan) ? r : inf, mx = (valid && !nan) ? r : -inf, ab = (valid && !nan) ? fabs(r) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, off));
            mx = fmax(mx, __shfl_xor(mx, off));
            ab += __shfl_xor(ab, off);
        }
        const bool any_nan = __ballot(nan) != 0ull;
        if ((threadIdx.x & 63) == 0) {
            double *p = task.partials + 3LL * (4LL * (logical - task.block_begin) + (threadIdx.x >> 6));
            p[0] = mn;
            p[1] = mx;
            p[2] = any_nan ? __longlong_as_double(0x7FF8000000000000LL) : ab;
        }
    }
}

// Per-wavefront partials -> [min, max, sum |.|] of every score array, in two steps: kStatsSlices workgroups per array
// reduce a slice each (a single workgroup would crawl through megabytes of partials), one more folds the slices.
constexpr int kStatsSlices = 64;
struct MedianStatsTask {
    const double *partials;
    long long n_partials;
};
struct MedianStatsBatch {
    MedianStatsTask tasks[kMedianBatchMax];
};

__device__ __forceinline__ void stats_block_reduce(double mn, double mx, double ab, bool nan, double *__restrict__ out)
{
    __shared__ double red[4][3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, off));
        mx = fmax(mx, __shfl_xor(mx, off));
        ab += __shfl_xor(ab, off);
    }
    const bool any_nan = __syncthreads_or(nan ? 1 : 0) != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[wave][0] = mn;
        red[wave][1] = mx;
        red[wave][2] = ab;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
        out[1] = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
        const double sum = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
        out[2] = any_nan ? __longlong_as_double(0x7FF8000000000000LL) : sum;
    }
}

// grid (kStatsSlices, tasks): slice s of task t -> slices[(t * kStatsSlices + s) * 3]
__global__ __launch_bounds__(256) void median_stats_slice_kernel(MedianStatsBatch batch, double *__restrict__ slices)
{
    const MedianStatsTask task = batch.tasks[blockIdx.y];
    const long long per = (task.n_partials + kStatsSlices - 1) / kStatsSlices;
    const long long lo = per * blockIdx.x, hi = min(task.n_partials, lo + per);
    const double inf = std::numeric_limits<double>::infinity();
    double mn = inf, mx = -inf, ab = 0.0;
    bool nan = false;
    for (long long b = lo + threadIdx.x; b < hi; b += 256) {
        mn = fmin(mn, task.partials[3 * b + 0]);
        mx = fmax(mx, task.partials[3 * b + 1]);
        const double a = task.partials[3 * b + 2];
        const bool bad = is_nan_bits(a);
        nan |= bad;
        ab += bad ? 0.0 : a;
    }
    stats_block_reduce(mn, mx, ab, nan, slices + 3LL * ((long long)blockIdx.y * kStatsSlices + blockIdx.x));
}

// one workgroup per task: its slices -> out[task * 3]
__global__ __launch_bounds__(256) void median_stats_final_kernel(const double *__restrict__ slices, double *__restrict__ out)
{
    const double inf = std::numeric_limits<double>::infinity();
    double mn = inf, mx = -inf, ab = 0.0;
    bool nan = false;
    if (threadIdx.x < kStatsSlices) {
        const double *p = slices + 3LL * ((long long)blockIdx.x * kStatsSlices + threadIdx.x);
        mn = p[0];
        mx = p[1];
        nan = is_nan_bits(p[2]);
        ab = nan ? 0.0 : p[2];
    }
    stats_block_reduce(mn, mx, ab, nan, out + 3LL * blockIdx.x);
}

// the same statistics from a score array itself (element types / K the network kernel does not take)
__global__ __launch_bounds__(256) void score_stats_partial_kernel(const double *__restrict__ scores, long long n,
                                                                  double *__restrict__ partials)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = j < n;
    const double r = valid ? scores[j] : 0.0;
    const bool nan = valid && is_nan_bits(r);
    const double inf = std::numeric_limits<double>::infinity();
    double mn = (valid && !nan) ? r : inf, mx = (valid && !nan) ? r : -inf, ab = (valid && !nan) ? fabs(r) : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, off));
        mx = fmax(mx, __shfl_xor(mx, off));
        ab += __shfl_xor(ab, off);
    }
    const bool any_nan = __ballot(nan) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        double *p = partials + 3LL * (4LL * blockIdx.x + (threadIdx.x >> 6));
        p[0] = mn;
        p[1] = mx;
        p[2] = any_nan ? __longlong_as_double(0x7FF8000000000000LL) : ab;
    }
}

// 100 < K <= 200: two sorted halves.  The column (padded to 200 entries with -inf / +inf in equal numbers, one extra
// +inf for odd K, which leaves the middle order statistics where they are) is split into its first 100 entries and
// the rest; each half is sorted completely by the 100-input network (registers), the first one parked in LDS; the two
// middle entries of the union of two sorted 100-sequences are  max_i min(a_i, b_{99-i})  and  min_i max(a_i, b_{99-i}).
// About 4 800 min / max per locus against 2 K^2 compares of the rank-counting kernel.
constexpr int kHalf = 100;

template <typename T>
__global__ __launch_bounds__(64) void median_split_kernel(const T *__restrict__ m, int K, long long n, long long stride,
                                                          double *__restrict__ out)
{
    __shared__ double first[kHalf][64];
    const long long j = (long long)xcd_contiguous_block() * blockDim.x + threadIdx.x;
    if (j >= n) {
        return;
    }
    const double inf = std::numeric_limits<double>::infinity();
    const int pad = 2 * kHalf - K;
    const int n_lo = pad / 2;  // -inf entries (the other pad entries are +inf)
    double v[kHalf];
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {  // rows 0..99 (K > 100)
        v[k] = (double)m[(long long)k * stride + j];
        sum += v[k];
    }
    bool has_nan = false;
    if (is_nan_bits(sum)) {
#pragma unroll
        for (int k = 0; k < kHalf; ++k) {
            const bool bad = is_nan_bits(v[k]);
            has_nan |= bad;
            v[k] = bad ? inf : v[k];
        }
    }
    select_middle<kHalf>(v);  // every output is used below: a complete sort
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
        first[k][threadIdx.x] = v[k];
    }
    sum = 0.0;
    // rows 100..K-1, then the padding.  Every load is unconditional (a load under a per-row condition is waited for
    // before the next is issued): rows past the matrix re-read its last row (a cache hit) and are replaced afterwards.
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
        const int row = min(kHalf + k, K - 1);
        v[k] = (double)m[(long long)row * stride + j];
    }
#pragma unroll
    for (int k = 0; k < kHalf; ++k) {
        const int row = kHalf + k;
        if (row < K) {
            sum += v[k];
        } else {
            v[k] = (row - K < n_lo) ? -inf : inf;
        }
    }
    if (is_nan_bits(sum)) {
#pragma unroll
        for (int k = 0; k < kHalf; ++k) {
            const bool bad = (kHalf + k < K) && is_nan_bits(v[k]);
            has_nan |= bad;
            v[k] = bad ? inf : v[k];
        }
    }
    select_middle<kHalf>(v);
    double lower = -inf, upper = inf;
#pragma unroll
    for (int i = 0; i < kHalf; ++i) {
        const double a = first[i][threadIdx.x], b = v[kHalf - 1 - i];
        lower = fmax(lower, fmin(a, b));
        upper = fmin(upper, fmax(a, b));
    }
    const double r = (K & 1) ? lower : (lower + upper) / 2.0;
    out[j] = has_nan ? __longlong_as_double(0x7FF8000000000000LL) : r;
}

// 200 < K <= 1200: P sorted parts of L entries parked in LDS (4 x 64 or 3 x 100 for 64 columns per workgroup; 6 x 100 for
// 32 and 12 x 100 for 16 -- the parts of a whole wavefront's columns would not fit), each lane its own column; then a P-way merge with the parts' heads in registers up to the lower middle rank (P L / 2 steps of
// P - 1 compares and ONE LDS read): the element taken last and the smallest remaining head are the middle pair.
// (A first version searched the ranks by nested binary searches: P^2 log^2 L dependent LDS reads, 5x slower.)
template <typename T, int L, int P, int LANES>
__global__ __launch_bounds__(LANES) void median_parts_kernel(const T *__restrict__ m, int K, long long n, long long stride,
                                                             double *__restrict__ out)
{
    extern __shared__ double parts[];  // [P][L][LANES]
    const long long j = (long long)xcd_contiguous_block() * blockDim.x + threadIdx.x;
    if (j >= n) {
        return;
    }
    const int lane = threadIdx.x;
    const double inf = std::numeric_limits<double>::infinity();
    const int pad = P * L - K;
    const int n_lo = pad / 2;  // -inf entries in front of the last part's padding; the rest +inf
    bool has_nan = false;
#pragma unroll 1
    for (int p = 0; p < P; ++p) {
        double v[L];
        double sum = 0.0;
        if ((p + 1) * L <= K) {
            // a whole part of real rows: unconditional loads, all in flight together (a load under a per-row condition
            // is waited for before the next one is issued)
            const T *__restrict__ src = m + (long long)(p * L) * stride + j;
#pragma unroll
            for (int k = 0; k < L; ++k) {
                v[k] = (double)src[(long long)k * stride];
            }
#pragma unroll
            for (int k = 0; k < L; ++k) {
                sum += v[k];
            }
        } else {
            // the part that holds the padding: rows past the matrix re-read its last row and are replaced afterwards
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const int row = min(p * L + k, K - 1);
                v[k] = (double)m[(long long)row * stride + j];
            }
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const int row = p * L + k;
                if (row < K) {
                    sum += v[k];
                } else {
                    v[k] = (row - K < n_lo) ? -inf : inf;
                }
            }
        }
        if (is_nan_bits(sum)) {
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const bool bad = (p * L + k < K) && is_nan_bits(v[k]);
                has_nan |= bad;
                v[k] = bad ? inf : v[k];
            }
        }
        select_middle<L>(v);  // every output is stored: a complete sort
#pragma unroll
        for (int k = 0; k < L; ++k) {
            parts[((size_t)p * L + k) * LANES + lane] = v[k];
        }
    }
    // (each lane reads only what it wrote: no barrier)
    // P-way merge up to the lower middle rank: the heads of the parts in registers, one LDS read per step
    const int total = P * L;
    const int k1 = total / 2;  // 1-based rank of the lower middle element of the padded column
    int idx[P];
    double head[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        idx[p] = 0;
        head[p] = parts[((size_t)p * L) * LANES + lane];
    }
    double first = -inf;
#pragma unroll 2
    for (int step = 0; step < k1; ++step) {
        double mn = head[0];
        int w = 0;
#pragma unroll
        for (int p = 1; p < P; ++p) {
            const bool less = head[p] < mn;
            mn = less ? head[p] : mn;
            w = less ? p : w;
        }
        first = mn;
        int iw = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            idx[p] += (p == w) ? 1 : 0;
            iw = (p == w) ? idx[p] : iw;
        }
        const double nv = (iw < L) ? parts[((size_t)w * L + min(iw, L - 1)) * LANES + lane] : inf;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            head[p] = (p == w) ? nv : head[p];
        }
    }
    double second = head[0];
#pragma unroll
    for (int p = 1; p < P; ++p) {
        second = fmin(second, head[p]);
    }
    // padded length is even; the padding keeps the middle pair where the K-column has it (odd K: one extra +inf, and
    // the lower middle element is the median)
    const double r = (K & 1) ? first : (first + second) / 2.0;
    out[j] = has_nan ? __longlong_as_double(0x7FF8000000000000LL) : r;
}

// K == 1: copy (rocco.py:254-255; power == 1.0 is the identity)
template <typename T>
__global__ __launch_bounds__(256) void copy_row_kernel(const T *__restrict__ m, long long n,
                                                       double *__restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        out[j] = (double)m[j];
    }
}

// Any K (used above the largest network): rank counting, O(K^2) reads served from L1/L2.
template <typename T>
__global__ __launch_bounds__(256) void median_rank_kernel(const T *__restrict__ m, int K, long long n,
                                                          long long stride, double *__restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) {
        return;
    }
    const int want_lo = (K - 1) / 2;  // 0-based order statistics wanted
    const int want_hi = K / 2;
    double lo = 0.0, hi = 0.0;
    bool has_nan = false;
    for (int a = 0; a < K; ++a) {
        const double x = (double)m[(long long)a * stride + j];
        if (is_nan_bits(x)) {
            has_nan = true;
            continue;
        }
        int less = 0, equal = 0;
        for (int b = 0; b < K; ++b) {
            const double y = (double)m[(long long)b * stride + j];
            less += (y < x);
            equal += (y == x);
        }
        // x occupies sorted positions [less, less + equal)
        if (want_lo >= less && want_lo < less + equal) {
            lo = x;
        }
        if (want_hi >= less && want_hi < less + equal) {
            hi = x;
        }
    }
    out[j] = has_nan ? __longlong_as_double(0x7FF8000000000000LL) : ((K & 1) ? lo : (lo + hi) / 2.0);
}

// The branches of score_central_tendency_chrom the reference's driver does not reach (rocco.py:267-272, 298-299):
// nearest-rank quantile (the rank comes from the host, computed by NumPy's own rule) and the column mean.
template <typename T>
__global__ __launch_bounds__(256) void order_statistic_kernel(const T *__restrict__ m, int K, long long n, long long stride,
                                                             int rank, double *__restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) {
        return;
    }
    double found = 0.0;
    bool has_nan = false;
    for (int a = 0; a < K; ++a) {
        const double x = (double)m[(long long)a * stride + j];
        if (is_nan_bits(x)) {
            has_nan = true;
            continue;
        }
        int less = 0, equal = 0;
        for (int b = 0; b < K; ++b) {
            const double y = (double)m[(long long)b * stride + j];
            less += (y < x);
            equal += (y == x);
        }
        if (rank >= less && rank < less + equal) {  // x occupies the sorted positions [less, less + equal)
            found = x;
        }
    }
    out[j] = has_nan ? __longlong_as_double(0x7FF8000000000000LL) : found;
}

// np.mean(matrix, axis=0): the rows are added one after the other, then one division by K
template <typename T>
__global__ __launch_bounds__(256) void column_mean_kernel(const T *__restrict__ m, int K, long long n, long long stride,
                                                         double *__restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) {
        return;
    }
    double acc = (double)m[j];
    for (int a = 1; a < K; ++a) {
        acc += (double)m[(long long)a * stride + j];
    }
    out[j] = acc / (double)K;
}

// stats.tmean(column, limits=(lo, hi), inclusive=(True, True)) of SciPy 1.15 (rocco/rocco.py:273-297): values outside
// [lo, hi] are replaced by 0.0, np.sum adds the K entries of the (strided) column in NumPy's pairwise order -- fewer
// than 8 one after the other, up to 128 with eight interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
// and the remainder appended, longer ranges split at n/2 rounded down to a multiple of 8 -- and the sum is divided by
// the number of values kept.  lo / hi are order statistics of the column (np.quantile(..., method="nearest") at tprop
// and 1 - tprop: their ranks come from the host, by NumPy's own rounding rule).
template <typename T>
__device__ double trimmed_pairwise(const T *__restrict__ col, long long stride, int n, double lo, double hi)
{
    auto kept = [&](int k) -> double {
        const double v = (double)col[(long long)k * stride];
        return (v < lo || v > hi) ? 0.0 : v;
    };
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) {
            res += kept(i);
        }
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int q = 0; q < 8; ++q) {
            r[q] = kept(q);
        }
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
            for (int q = 0; q < 8; ++q) {
                r[q] += kept(i + q);
            }
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) {
            res += kept(i);
        }
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return trimmed_pairwise(col, stride, n2, lo, hi) + trimmed_pairwise(col + (long long)n2 * stride, stride, n - n2, lo, hi);
}

template <typename T>
__global__ __launch_bounds__(256) void trimmed_mean_kernel(const T *__restrict__ m, int K, long long n, long long stride,
                                                          int rank_lo, int rank_hi, double *__restrict__ out)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) {
        return;
    }
    // the two limits by rank counting (as order_statistic_kernel)
    double lo = 0.0, hi = 0.0;
    bool has_nan = false;
    for (int a = 0; a < K; ++a) {
        const double x = (double)m[(long long)a * stride + j];
        if (is_nan_bits(x)) {
            has_nan = true;
            continue;
        }
        int less = 0, equal = 0;
        for (int b = 0; b < K; ++b) {
            const double y = (double)m[(long long)b * stride + j];
            less += (y < x);
            equal += (y == x);
        }
        if (rank_lo >= less && rank_lo < less + equal) {
            lo = x;
        }
        if (rank_hi >= less && rank_hi < less + equal) {
            hi = x;
        }
    }
    if (has_nan) {  // NaN limits keep every value, and the NaN in the sum makes the mean NaN
        out[j] = __longlong_as_double(0x7FF8000000000000LL);
        return;
    }
    double count = 0.0;
    for (int a = 0; a < K; ++a) {
        const double v = (double)m[(long long)a * stride + j];
        count += (v < lo || v > hi) ? 0.0 : 1.0;
    }
    out[j] = trimmed_pairwise(m + j, stride, K, lo, hi) / count;
}

// np.power(x, p) of rocco/rocco.py:255,304: NumPy's loop squares for p == 2 (exact); every other exponent goes to its
// pow, which differs between NumPy's SVML and libm builds in the last place -- the device's pow stands in for it.
__global__ __launch_bounds__(256) void power_kernel(const double *__restrict__ x, double p, double *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double v = x[i];
        out[i] = (p == 2.0) ? (v * v) : pow(v, p);
    }
}

template <typename T, int KP>
void launch_kp(const T *m, int K, long long n, long long stride, double *out, hipStream_t stream)
{
    const int threads = 256;
    const long long blocks = (n + threads - 1) / threads;
    if (K == KP) {
        hipLaunchKernelGGL((median_kernel<T, KP, true>), dim3((unsigned)blocks), dim3(threads), 0, stream, m,
                           K, n, stride, out);
    } else {
        hipLaunchKernelGGL((median_kernel<T, KP, false>), dim3((unsigned)blocks), dim3(threads), 0, stream, m,
                           K, n, stride, out);
    }
}

template <typename T, int L, int P, int LANES>
int launch_parts(const T *m, int K, long long n, long long stride, double *out, hipStream_t stream)
{
    static bool configured = false;
    const size_t lds = (size_t)P * L * LANES * sizeof(double);
    if (!configured) {
        ROCCO_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(median_parts_kernel<T, L, P, LANES>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL((median_parts_kernel<T, L, P, LANES>), dim3((unsigned)((n + LANES - 1) / LANES)), dim3(LANES), lds, stream, m, K,
                       n, stride, out);
    return ROCCO_HIP_OK;
}

template <typename T>
int dispatch(const T *m, size_t K, size_t n, size_t stride, double *out, hipStream_t stream)
{
    int rc = ROCCO_HIP_OK;
    const long long nn = (long long)n;
    const long long st = (long long)stride;
    const int threads = 256;
    const long long blocks = (nn + threads - 1) / threads;
    if (K == 1) {
        hipLaunchKernelGGL((copy_row_kernel<T>), dim3((unsigned)blocks), dim3(threads), 0, stream, m,
                           nn, out);
    } else if (K <= 2) {
        launch_kp<T, 2>(m, (int)K, nn, st, out, stream);
    } else if (K <= 4) {
        launch_kp<T, 4>(m, (int)K, nn, st, out, stream);
    } else if (K <= 6) {
        launch_kp<T, 6>(m, (int)K, nn, st, out, stream);
    } else if (K <= 8) {
        launch_kp<T, 8>(m, (int)K, nn, st, out, stream);
    } else if (K <= 10) {
        launch_kp<T, 10>(m, (int)K, nn, st, out, stream);
    } else if (K <= 12) {
        launch_kp<T, 12>(m, (int)K, nn, st, out, stream);
    } else if (K <= 16) {
        launch_kp<T, 16>(m, (int)K, nn, st, out, stream);
    } else if (K <= 20) {
        launch_kp<T, 20>(m, (int)K, nn, st, out, stream);
    } else if (K <= 24) {
        launch_kp<T, 24>(m, (int)K, nn, st, out, stream);
    } else if (K <= 32) {
        launch_kp<T, 32>(m, (int)K, nn, st, out, stream);
    } else if (K <= 40) {
        launch_kp<T, 40>(m, (int)K, nn, st, out, stream);
    } else if (K <= 50) {
        launch_kp<T, 50>(m, (int)K, nn, st, out, stream);
    } else if (K <= 64) {
        launch_kp<T, 64>(m, (int)K, nn, st, out, stream);
    } else if (K <= 80) {
        launch_kp<T, 80>(m, (int)K, nn, st, out, stream);
    } else if (K <= 100) {
        launch_kp<T, 100>(m, (int)K, nn, st, out, stream);
    } else if (K <= 2 * (size_t)kHalf) {
        hipLaunchKernelGGL((median_split_kernel<T>), dim3((unsigned)((nn + 63) / 64)), dim3(64), 0, stream, m, (int)K, nn, st, out);
    } else if (K <= 256) {
        if ((rc = launch_parts<T, 64, 4, 64>(m, (int)K, nn, st, out, stream)) != ROCCO_HIP_OK) return rc;
    } else if (K <= 300) {
        if ((rc = launch_parts<T, 100, 3, 64>(m, (int)K, nn, st, out, stream)) != ROCCO_HIP_OK) return rc;
    } else if (K <= 600) {
        // half a wavefront per workgroup (the parts of 64 columns would not fit the LDS): half the lanes idle
        if ((rc = launch_parts<T, 100, 6, 32>(m, (int)K, nn, st, out, stream)) != ROCCO_HIP_OK) return rc;
    } else if (K <= 1200) {
        if ((rc = launch_parts<T, 100, 12, 16>(m, (int)K, nn, st, out, stream)) != ROCCO_HIP_OK) return rc;
    } else {
        hipLaunchKernelGGL((median_rank_kernel<T>), dim3((unsigned)blocks), dim3(threads), 0, stream,
                           m, (int)K, nn, st, out);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

template <typename T, int KP>
void launch_batch_kp(const MedianBatch &batch, unsigned blocks, hipStream_t stream)
{
    if (batch.K == KP) {
        hipLaunchKernelGGL((median_batch_kernel<T, KP, true>), dim3(blocks), dim3(256), 0, stream, batch);
    } else {
        hipLaunchKernelGGL((median_batch_kernel<T, KP, false>), dim3(blocks), dim3(256), 0, stream, batch);
    }
}

template <typename T>
bool dispatch_batch(const MedianBatch &batch, unsigned blocks, hipStream_t stream)
{
    const int K = batch.K;
    if (K < 2 || K > 100) {
        return false;
    }
    if (K <= 2) launch_batch_kp<T, 2>(batch, blocks, stream);
    else if (K <= 4) launch_batch_kp<T, 4>(batch, blocks, stream);
    else if (K <= 6) launch_batch_kp<T, 6>(batch, blocks, stream);
    else if (K <= 8) launch_batch_kp<T, 8>(batch, blocks, stream);
    else if (K <= 10) launch_batch_kp<T, 10>(batch, blocks, stream);
    else if (K <= 12) launch_batch_kp<T, 12>(batch, blocks, stream);
    else if (K <= 16) launch_batch_kp<T, 16>(batch, blocks, stream);
    else if (K <= 20) launch_batch_kp<T, 20>(batch, blocks, stream);
    else if (K <= 24) launch_batch_kp<T, 24>(batch, blocks, stream);
    else if (K <= 32) launch_batch_kp<T, 32>(batch, blocks, stream);
    else if (K <= 40) launch_batch_kp<T, 40>(batch, blocks, stream);
    else if (K <= 50) launch_batch_kp<T, 50>(batch, blocks, stream);
    else if (K <= 64) launch_batch_kp<T, 64>(batch, blocks, stream);
    else if (K <= 80) launch_batch_kp<T, 80>(batch, blocks, stream);
    else launch_batch_kp<T, 100>(batch, blocks, stream);
    return true;
}

}  // namespace

size_t median_partials_count(const size_t *n, size_t count)
{
    size_t blocks = 0;
    for (size_t i = 0; i < count; ++i) {
        blocks += 4 * ((n[i] + 255) / 256);  // one partial per wavefront
    }
    return blocks + (size_t)kMedianBatchMax * 64;  // + the slices of the second reduction step (kStatsSlices per task)
}

int launch_median_batch(const void *const *matrices_dev, int dtype, size_t K, const size_t *n, const size_t *row_strides,
                        double *const *scores_dev, size_t count, hipStream_t stream, double *stats_dev, double *partials_dev)
{
    const bool want_stats = (stats_dev != nullptr && partials_dev != nullptr);
    size_t partials_total = 0;
    for (size_t i = 0; i < count; ++i) {
        partials_total += 4 * ((n[i] + 255) / 256);
    }
    size_t at = 0;
    size_t partial_at = 0;  // workgroups (of 256 loci) before the current matrix
    while (at < count) {
        MedianBatch batch;
        MedianStatsBatch finals;
        int final_index[kMedianBatchMax];
        batch.K = (int)K;
        batch.n_tasks = 0;
        for (int i = 0; i < kMedianBatchMax; ++i) {
            batch.block_begin[i] = 0xFFFFFFFFu;
        }
        unsigned blocks = 0;
        const size_t first = at;
        while (at < count && batch.n_tasks < kMedianBatchMax) {
            if (n[at] > 0) {
                const size_t nb = (n[at] + 255) / 256;
                finals.tasks[batch.n_tasks].partials = want_stats ? partials_dev + 3 * partial_at : nullptr;
                finals.tasks[batch.n_tasks].n_partials = (long long)(4 * nb);
                final_index[batch.n_tasks] = (int)at;
                MedianTask &t = batch.tasks[batch.n_tasks++];
                t.matrix = matrices_dev[at];
                t.out = scores_dev[at];
                t.n = (long long)n[at];
                t.stride = (long long)row_strides[at];
                t.block_begin = blocks;
                batch.block_begin[batch.n_tasks - 1] = blocks;
                t.partials = want_stats ? partials_dev + 3 * partial_at : nullptr;
                blocks += (unsigned)nb;
                partial_at += 4 * nb;
            }
            ++at;
        }
        if (batch.n_tasks == 0) {
            continue;
        }
        const bool done = (dtype == 0) ? dispatch_batch<double>(batch, blocks, stream) : dispatch_batch<float>(batch, blocks, stream);
        if (!done) {  // K outside the network sizes: one launch per matrix, statistics from the scores
            for (int i = 0; i < batch.n_tasks; ++i) {
                const int rc = launch_median(batch.tasks[i].matrix, dtype, K, (size_t)batch.tasks[i].n,
                                             (size_t)batch.tasks[i].stride, batch.tasks[i].out, stream);
                if (rc != ROCCO_HIP_OK) return rc;
                if (want_stats) {
                    hipLaunchKernelGGL(score_stats_partial_kernel, dim3((unsigned)(finals.tasks[i].n_partials / 4)), dim3(256), 0, stream,
                                       (const double *)batch.tasks[i].out, batch.tasks[i].n, batch.tasks[i].partials);
                }
            }
        }
        if (want_stats) {
            // results of this launch's matrices are contiguous in stats_dev only if no empty matrix lies between
            // them: reduce every run of consecutive indices with one launch
            int i = 0;
            while (i < batch.n_tasks) {
                int k = i + 1;
                while (k < batch.n_tasks && final_index[k] == final_index[k - 1] + 1) {
                    ++k;
                }
                MedianStatsBatch part;
                for (int q = i; q < k; ++q) {
                    part.tasks[q - i] = finals.tasks[q];
                }
                // (the slices live behind the partials: 3 * kStatsSlices doubles per task of one launch)
                double *slices = partials_dev + 3 * partials_total;
                hipLaunchKernelGGL(median_stats_slice_kernel, dim3(kStatsSlices, (unsigned)(k - i)), dim3(256), 0, stream, part, slices);
                hipLaunchKernelGGL(median_stats_final_kernel, dim3((unsigned)(k - i)), dim3(256), 0, stream, (const double *)slices,
                                   stats_dev + 3 * (size_t)final_index[i]);
                i = k;
            }
        }
        (void)first;
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

int launch_median(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride,
                  double *scores_dev, hipStream_t stream)
{
    if (n == 0) {
        return ROCCO_HIP_OK;
    }
    if (dtype == 0) {
        return dispatch<double>((const double *)matrix_dev, K, n, row_stride, scores_dev, stream);
    }
    return dispatch<float>((const float *)matrix_dev, K, n, row_stride, scores_dev, stream);
}

int launch_order_statistic(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, int rank,
                           double *scores_dev, hipStream_t stream)
{
    if (n == 0) {
        return ROCCO_HIP_OK;
    }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 0) {
        hipLaunchKernelGGL(order_statistic_kernel<double>, grid, block, 0, stream, (const double *)matrix_dev, (int)K,
                           (long long)n, (long long)row_stride, rank, scores_dev);
    } else {
        hipLaunchKernelGGL(order_statistic_kernel<float>, grid, block, 0, stream, (const float *)matrix_dev, (int)K,
                           (long long)n, (long long)row_stride, rank, scores_dev);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_trimmed_mean(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, int rank_lo, int rank_hi,
                        double *scores_dev, hipStream_t stream)
{
    if (n == 0) {
        return ROCCO_HIP_OK;
    }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 0) {
        hipLaunchKernelGGL(trimmed_mean_kernel<double>, grid, block, 0, stream, (const double *)matrix_dev, (int)K, (long long)n,
                           (long long)row_stride, rank_lo, rank_hi, scores_dev);
    } else {
        hipLaunchKernelGGL(trimmed_mean_kernel<float>, grid, block, 0, stream, (const float *)matrix_dev, (int)K, (long long)n,
                           (long long)row_stride, rank_lo, rank_hi, scores_dev);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

int launch_power(const double *x_dev, double p, double *out_dev, size_t n, hipStream_t stream)
{
    if (n > 0) {
        hipLaunchKernelGGL(power_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x_dev, p, out_dev, (long long)n);
        ROCCO_HIP_TRY(hipGetLastError());
    }
    return ROCCO_HIP_OK;
}

int launch_column_mean(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, double *scores_dev,
                       hipStream_t stream)
{
    if (n == 0) {
        return ROCCO_HIP_OK;
    }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 0) {
        hipLaunchKernelGGL(column_mean_kernel<double>, grid, block, 0, stream, (const double *)matrix_dev, (int)K,
                           (long long)n, (long long)row_stride, scores_dev);
    } else {
        hipLaunchKernelGGL(column_mean_kernel<float>, grid, block, 0, stream, (const float *)matrix_dev, (int)K,
                           (long long)n, (long long)row_stride, scores_dev);
    }
    ROCCO_HIP_TRY(hipGetLastError());
    return ROCCO_HIP_OK;
}

}  // namespace rocco
