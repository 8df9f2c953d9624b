// rocco_amd/csrc/kernels.h -- device task descriptors and launchers shared by the .hip files.
#pragma once

#include "common.h"

namespace rocco {
#if defined(__HIPCC__)
// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wavefront's global loads and stores
// (s_waitcnt vmcnt(0)): a wavefront that has just issued the loads of a tile it needs three trips later then sits out a
// memory latency at every barrier -- measured in round 5 on the rolling-sums launch: 22 ns per locus with the chain AND the
// variances compiled out.  The tiles these kernels hand over live in LDS; what they read from or write to global memory
// is touched by one thread only.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif


// ---- chain_exact.hip ------------------------------------------------------------------------
struct ExactTask {
    const double *scores;         // n
    const double *switch_costs;   // n-1 or nullptr
    double gamma;
    long long n;
    const double *lambdas;        // n_lambda penalties (device)
    int n_lambda;                 // 1..64
    int record_lane;              // lane whose decisions are recorded + backtracked
    double *values_out;           // n_lambda
    long long *counts_out;        // n_lambda
    unsigned long long *decision_words;  // ceil((n-1)/32) words or nullptr
    uint8_t *solution;            // n bytes (used iff decision_words)
};

int launch_chain_exact(const ExactTask *tasks_dev, int n_tasks, hipStream_t stream);

// ---- median.hip -----------------------------------------------------------------------------
int launch_median(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride,
                  double *scores_dev, hipStream_t stream);

// several matrices of one K / element type in one launch (kernel-argument task table: up to kMedianBatchMax per launch)
constexpr int kMedianBatchMax = 48;
struct MedianTask {
    const void *matrix;
    double *out;
    long long n;
    long long stride;
    unsigned block_begin;
    unsigned pad;
    double *partials;  // [wavefront of this task][3] = min, max, sum |.| of its scores (nullptr: not wanted)
};
struct MedianBatch {
    // first workgroup of every task, ascending; unused slots hold 0xFFFFFFFF.  Kept apart from the task records so that
    // a workgroup finds its task with a few wide scalar loads and compares in registers -- not one dependent scalar
    // load per task before its first vector load can issue (24 chromosomes: 0.5 ms per genome launch)
    unsigned block_begin[kMedianBatchMax];
    MedianTask tasks[kMedianBatchMax];
    int n_tasks;
    int K;
};
// `stats_dev` (optional): [count][3] = min, max and sum of absolute values of every score array (what the budgeted
// solve otherwise reads the scores once more for), reduced from per-workgroup partials kept in `partials_dev`
// (3 doubles per 64 loci of every matrix; median_partials_count() tells how many wavefronts that is).
size_t median_partials_count(const size_t *n, size_t count);
int launch_median_batch(const void *const *matrices_dev, int dtype, size_t K, const size_t *n, const size_t *row_strides,
                        double *const *scores_dev, size_t count, hipStream_t stream, double *stats_dev = nullptr,
                        double *partials_dev = nullptr);

int launch_order_statistic(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, int rank,
                           double *scores_dev, hipStream_t stream);
int launch_column_mean(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, double *scores_dev,
                       hipStream_t stream);
int launch_trimmed_mean(const void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, int rank_lo, int rank_hi,
                        double *scores_dev, hipStream_t stream);
int launch_power(const double *x_dev, double p, double *out_dev, size_t n, hipStream_t stream);

// ---- decode.hip -----------------------------------------------------------------------------
// scratch: at least decode_scratch_bytes(n) bytes
size_t decode_scratch_bytes(size_t n);
int launch_decode_runs(const uint8_t *solution_dev, size_t n, int64_t *run_begin_dev,
                       int64_t *run_end_dev, size_t capacity, void *scratch_dev,
                       unsigned long long *n_runs_host_pinned, hipStream_t stream);

// several solutions in three launches and one synchronisation (n <= 1 tasks must be left out by the caller)
constexpr int kDecodeBatchMax = 48;
struct DecodeTask {
    const uint8_t *solution;
    long long n;
    long long tile_begin;  // first tile of this task in the launches (decode_tiles(n) tiles each)
    int64_t *run_begin;    // (table form: the table itself, rows of 3 values)
    int64_t *run_end;      // (table form: unused)
    unsigned long long capacity;  // (table form: rows of the whole table)
    long long unit;        // table form: what the first column of this task's rows holds
};
struct DecodeBatch {
    DecodeTask tasks[kDecodeBatchMax];
    int n_tasks;
    int pad;
};
long long decode_tiles(size_t n);
size_t decode_batch_scratch_bytes(long long total_tiles, int n_tasks);
// totals_host_pinned: 2 * n_tasks values (runs begun, runs ended -- equal) after the synchronisation
int launch_decode_runs_batch(const DecodeBatch &batch, long long total_tiles, void *scratch_dev,
                             unsigned long long *totals_host_pinned, hipStream_t stream);
// the same as ONE table of (unit, begin, end) rows, task after task; `eager_rows` rows are copied to
// `table_host_pinned` in front of the synchronisation, together with the totals (no synchronisation of its own
// when eager_rows covers the table)
int launch_decode_runs_table(const DecodeBatch &batch, long long total_tiles, void *scratch_dev, int64_t *table_dev,
                             unsigned long long *totals_host_pinned, int64_t *table_host_pinned, size_t eager_rows,
                             hipStream_t stream);

// ---- objective.hip --------------------------------------------------------------------------
size_t objective_scratch_bytes(size_t n);
int launch_objective(const uint8_t *solution_dev, const double *scores_dev,
                     const double *switch_costs_dev, double gamma, size_t n, void *scratch_dev,
                     double *objective_host_pinned, hipStream_t stream, bool synchronize = true);

// several problems at once: tasks_dev[t].tile_begin = first tile of task t (objective_tiles(n) tiles each);
// partial_dev: 2 * total_tiles doubles; out_dev: n_tasks doubles (no copy, no synchronisation)
struct ObjectiveTask {
    const uint8_t *solution;
    const double *scores;
    const double *switch_costs;  // n-1 or nullptr
    double gamma;
    long long n;
    long long tile_begin;
};
long long objective_tiles(size_t n);
int launch_objective_batch(const ObjectiveTask *tasks_dev, int n_tasks, long long total_tiles, double *partial_dev,
                           double *out_dev, hipStream_t stream);

// ---- whittaker.hip --------------------------------------------------------------------------
// factor_dev: 6 * factor_cap doubles filled by launch_whittaker_factor for a length factor_cap >= cols and
// the same penalty; scratch: at least whittaker_scratch_bytes(rows, cols) bytes; matrix and output must
// not overlap
size_t whittaker_scratch_bytes(size_t rows, size_t cols);
int launch_whittaker_factor(size_t cap, double penalty_lambda, double *factor_dev, hipStream_t stream,
                            const double *old_factor_dev = nullptr, size_t old_cap = 0);
// the same table walked by two host threads and copied up (whittaker_host.cpp): same bits, a tenth of the time; returns
// with the stream drained
int build_whittaker_factor_on_host(size_t cap, double penalty_lambda, double *factor_dev, hipStream_t stream,
                                   const double *old_factor_dev = nullptr, size_t old_cap = 0);
// one group of rows of one matrix over one SEGMENT of its loci: a workgroup of the row-parallel sweeps (whittaker.hip).
// Tiles are 64 loci, counted from locus 0.  The workgroup writes the tiles [tile_begin, tile_end); a segment that does not
// start where the sweep starts first walks `warm_tiles` tiles ahead of it (below tile_begin forward, above tile_end
// backward) from a zero state, leaves the state it reaches in `spec` and goes on from it; the state at the segment's far
// end goes to `edge`.  spec / edge: 2 x group rows chains x (last value, the one before), see whittaker.hip.
struct WhittakerRowTask {
    const double *src0, *src1;
    double *dst0, *dst1;
    long long n;
    int row0, rows;
    const double *tail;  // the three factor entries per parity that depend on the length
    long long tile_begin, tile_end;
    long long warm_tiles;
    double *spec, *edge;
    // round 5: the two elementwise passes around the sweeps ride on them.  offsets (may be null): one value per row of the
    // matrix, the sweeps' input is  src - offsets[row]  (the pilot offset of rocco/inference.py:330-331); minus (backward
    // only, may be null): the forward sweep's input matrix -- dst0 receives  (minus - offsets[row]) - baseline  instead of
    // the baseline (inference.py:335), and *bad is raised when a baseline is not finite (inference.py:207-208)
    const double *offsets, *minus;
    int *bad;
};
// one matrix of a batch for the seam check behind a sweep (whittaker.hip: whittaker_seam_kernel)
struct WhittakerSeamMatrix {
    const double *src0, *src1;  // as the sweep's tasks
    double *dst0, *dst1;
    long long n;
    int rows;
    int n_seg;                 // segments per row
    int group_rows;            // rows per workgroup (the matrix's rows dealt evenly)
    long long seg_tiles;       // tiles per segment (the last one may be shorter)
    long long task_base;       // the matrix's first task: task = task_base + group * n_seg + segment
    const double *tail;
    const double *offsets, *minus;  // as the sweep's tasks
    int *bad;
};
// several matrices of one penalty (the chromosomes of a genome) in ONE pair of launches (+ one seam check each);
// tasks_host_pinned: room for whittaker_batch_stage_bytes(...), must stay untouched until the stream has passed the copy
size_t whittaker_batch_scratch_bytes(const size_t *rows, const size_t *cols, size_t count);
size_t whittaker_batch_stage_bytes(const size_t *rows, const size_t *cols, size_t count);
int whittaker_group_rows();  // rows per workgroup of the batched sweeps
// how often a segment's warm-up had NOT reached the row's own values at the seam (each such seam was recomputed from
// the true state: results are the sequential sweep's either way); process-wide, for tests and diagnostics
long long whittaker_seam_repairs();
// after the stream of a batch launch has been waited for; returns non-zero when a residual launch met a non-finite baseline
int whittaker_collect_repairs(const void *tasks_host_pinned);
// offsets_dev (may be null, entries may be null): per matrix, one value per row that the sweeps subtract from their input;
// residual != 0: baselines_dev[i] receives (matrix - offsets) - baseline instead of the baseline (must not be the matrix)
int launch_crossfit_whittaker_batch(const double *const *matrices_dev, const size_t *rows, const size_t *cols, size_t count,
                                    double penalty_lambda, const double *factor_dev, size_t factor_cap, double *const *baselines_dev,
                                    void *scratch_dev, void *tasks_host_pinned, hipStream_t stream,
                                    const double *const *offsets_dev = nullptr, int residual = 0);

// ---- wls.hip --------------------------------------------------------------------------------
// scratch: at least wls_scratch_bytes(K, n) bytes; synchronises the stream before returning
int wls_spatial_window(size_t n, int requested);
// up to `group_rows` (1, 2, 4 or 8: wls_rolling_group_rows) consecutive rows of one matrix for the rolling launch: `row` the
// first of them (the next n doubles further), out = window variances of the row's n - window + 1 starts (the next row's
// right behind)
struct WlsRollingTask {
    const double *row;
    long long n;
    int window;
    int rows;
    double *out;
};
constexpr int kWlsRollingGroup = 8;
int wls_rolling_group_rows(size_t total_rows);
void wls_set_rolling_group_min(int rows_per_workgroup);  // (calling thread: at least so many rows per workgroup from now on; 1, 2, 4, 8)
int launch_wls_rolling_batch(const WlsRollingTask *tasks_dev, size_t n_tasks, int group_rows, hipStream_t stream);
int wls_max_window();
size_t wls_scratch_bytes(size_t K, size_t n, int spatial_window = 0, bool own_variances = true);
int launch_score_centered_wls(const double *centered_dev, size_t K, size_t n, double lower_bound_z, double prior_df,
                              double min_effect, int use_min_effect, int spatial_window,
                              double precision_floor_ratio, double *mean_dev, double *raw_var_dev,
                              double *prior_var_dev, double *mod_var_dev, double *se_dev, double *scores_dev,
                              void *scratch_dev, double *df_out, int *window_out, hipStream_t stream, int *flag_host_pinned, const double *vas_given = nullptr,
                              int *sorted_rows_out = nullptr);

int launch_log2_selfcheck(int family, unsigned long long seed, unsigned long long first, size_t count, unsigned long long *out_dev,
                          hipStream_t stream);
// row a2 glue (wls.hip): log2(max(x, 0) + pseudocount), row medians subtracted; out may alias the input
size_t log_scale_scratch_bytes(size_t K, size_t n);
int launch_log_scale_center_rows(const double *counts_dev, size_t K, size_t n, double pseudocount, int apply_log,
                                 double *centered_out_dev, double *row_offsets_out_dev, void *scratch_dev,
                                 hipStream_t stream, int *flag_host_pinned, int center = 1);
// out = log2(max(in, 0) + pseudocount), correctly rounded; *bad_dev |= 1 if a value is not finite
int launch_log_scale(const double *in_dev, double *out_dev, size_t count, double pseudocount, int *bad_dev, hipStream_t stream);
int launch_subtract(const double *a_dev, const double *b_dev, double *out_dev, size_t count, hipStream_t stream, int *bad_dev = nullptr);

// ---- summit.hip -----------------------------------------------------------------------------
int launch_summit_offsets(const int64_t *intervals_dev, size_t n_intervals, const int64_t *centers_dev,
                          const double *effect_mean_dev, size_t n_mean,
                          const int64_t *peak_start_dev, const int64_t *peak_end_dev, size_t n_peaks,
                          int64_t *offsets_out_dev, hipStream_t stream);

// ---- assemble.hip ---------------------------------------------------------------------------
size_t union_scratch_bytes(size_t count);
int launch_union_intervals(const int64_t *values_dev, size_t count, int64_t *unique_out_dev, size_t *n_unique_out,
                           int *fixed_step_out, void *scratch_dev, hipStream_t stream);
size_t scatter_scratch_bytes(size_t K, size_t m);
int launch_scatter_tracks(const int64_t *common_dev, size_t m, const int64_t *intervals_concat_dev,
                          const double *vals_concat_dev, const size_t *offsets_host, size_t K, int out_dtype,
                          void *matrix_out_dev, void *scratch_dev, hipStream_t stream);

// ---- npsum.hip ------------------------------------------------------------------------------
size_t npsum_scratch_bytes(size_t n);
int launch_numpy_sums(const double *x_dev, size_t n, const int *modes, int n_modes, double center, double scale,
                      double threshold, void *scratch_dev, double *sums_out_host, hipStream_t stream);
int launch_multiply(const double *a_dev, const double *b_dev, double *out_dev, size_t count, hipStream_t stream);
int launch_subtract_positive_row(const double *matrix_dev, const double *row_dev, size_t K, size_t n, double *out_dev,
                                 hipStream_t stream);

int launch_bigwig_dense_fill(const int64_t *starts_dev, const int64_t *ends_dev, const double *vals_dev, size_t count,
                             double const_scale, int round_digits, double *full_out_dev, size_t capacity,
                             int64_t *first_start_out, int64_t *step_out, size_t *n_full_out, int *flags_out,
                             void *scratch_dev, hipStream_t stream);

// ---- budget_stats.hip ----------------------------------------------------------------------
struct SortedProbe {
    long long ranks[8];
    double thresholds[8];
    double shift;  // thresholds are compared with (x - shift)
    int n_ranks, n_thresholds;
};
size_t sort_f64_scratch_bytes(size_t n);
int launch_sort_f64(const double *x_dev, size_t n, double *sorted_out_dev, void *scratch_dev, hipStream_t stream);
// values_out_dev: n_ranks doubles; counts_out_dev: 2 * n_thresholds (count of (x - shift) <= t, then < t)
int launch_sorted_probe(const double *sorted_dev, size_t n, const SortedProbe &probe, double *values_out_dev,
                        long long *counts_out_dev, hipStream_t stream);
size_t autocov_scratch_bytes(size_t n, int max_lag);
// sums_out_dev[k] = sum_i (x_i - mean)(x_{i+k} - mean), k = 0..max_lag (<= 1023), in a fixed order
int launch_autocov(const double *x_dev, size_t n, double mean, int max_lag, double *sums_out_dev, void *scratch_dev,
                   hipStream_t stream);
int launch_negative_part(const double *scores_dev, double *out_dev, size_t n, hipStream_t stream);
int launch_soft_counts(const double *scores_dev, double center, double scale, double *out_dev, size_t n, hipStream_t stream);

// ---- peakscore.hip -------------------------------------------------------------------------
// scratch_dev: P * K doubles
int launch_peak_signal(const double *counts_dev, const double *lengths_dev, size_t P, size_t K, double row_scale, double pc,
                       double percentile, double *out_dev, void *scratch_dev, hipStream_t stream);
int launch_ecdf_survival(const double *stat_dev, const int *bin_dev, const double *null_values_dev, const long long *null_offsets_dev,
                         size_t P, double *out_dev, hipStream_t stream);
size_t bh_scratch_bytes(size_t m);
int launch_bh_adjust(const double *pvals_dev, size_t m, double *qvals_out_dev, void *scratch_dev, hipStream_t stream);

// ---- synth.hip ------------------------------------------------------------------------------
int launch_synth(void *matrix_dev, int dtype, size_t K, size_t n, size_t row_stride, uint64_t seed,
                 hipStream_t stream);

}  // namespace rocco
