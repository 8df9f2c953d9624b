/*
 * include/rocco_hip.h -- C ABI of librocco_hip.so, the MI355X (gfx950) implementation of ROCCO's
 * per-chromosome solve path:  K x n signal matrix -> per-locus scores -> budgeted chain solve ->
 * 0/1 vector -> merged runs (BED3 intervals).
 *
 * Convention (modelled on the reference's native backends, rocco/native/wls_backend.h:11-28 and
 * rocco/native/baseline_backend.h:12-23): plain pointers and sizes, caller-owned buffers, `int`
 * status -- 0 ok, -1 out of memory, -2 invalid argument, -3 HIP runtime error (see
 * rocco_hip_last_error).  All `*_dev` pointers are DEVICE pointers (HBM); everything else is host
 * memory.  `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 * synchronous with respect to their scalar outputs (they return after the needed readback) and
 * re-entrant across different solver handles; one handle must not be used from two host threads
 * at once.  There is no CPU fallback anywhere behind this interface.
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef ROCCO_HIP_H
#define ROCCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROCCO_HIP_OK 0
#define ROCCO_HIP_ENOMEM (-1)
#define ROCCO_HIP_EINVAL (-2)
#define ROCCO_HIP_EHIP (-3)

/* How a result was obtained (diagnostic; written to `path_out` / result.path). */
#define ROCCO_HIP_PATH_CERTIFIED 1 /* parallel delta-form kernels, certified equal to the reference */
#define ROCCO_HIP_PATH_EXACT 2     /* sequential exact emulation kernel (same IEEE op order)        */
#define ROCCO_HIP_PATH_TRIVIAL 3   /* decided without a kernel launch (e.g. n == 1)                 */
#define ROCCO_HIP_PATH_SPINE 4     /* parallel kernels + exact spine through the hazard chunks      */

typedef struct rocco_hip_solver rocco_hip_solver;

/* ABI version of this header (major * 1000 + minor). */
int rocco_hip_abi_version(void);

/* Last HIP/runtime error text for this thread ("" if none). */
const char *rocco_hip_last_error(void);

/* Create / destroy a solver handle bound to HIP device `device`.  The handle owns device scratch
 * and pinned host staging that grow on demand. */
int rocco_hip_solver_create(rocco_hip_solver **solver_out, int device);
void rocco_hip_solver_destroy(rocco_hip_solver *solver);

/* Tunables (speculation depth of the lambda search, force the exact kernel, ...).  Unknown keys
 * return ROCCO_HIP_EINVAL.  Keys: "force_exact" (0/1), "spec_depth" (1..6), "active_set" (0/1), "lean" (0/1),
 * "rolling_group_min" (1, 2, 4, 8: least rows per workgroup of the batched rolling launch -- fewer, fuller workgroups for a
 * caller that runs other work beside it; none changes a result). */
int rocco_hip_solver_set(rocco_hip_solver *solver, const char *key, long long value);

/* ---- scoring -------------------------------------------------------------------------------
 * Replaces rocco/rocco.py:243-304 `score_central_tendency_chrom`, median branch (264-265), as
 * reached from rocco/rocco.py:983-991.  `matrix_dev` is row-major [K][row_stride] (row_stride >= n,
 * in elements); dtype 0 = float64, 1 = float32 (the --low_memory matrices, readtracks.py:621).
 * K == 1 copies the row (rocco.py:254-255).  A NaN anywhere in a column gives NaN. */
int rocco_hip_score_median(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K,
                           size_t n, size_t row_stride, double *scores_dev, void *stream);

/* The same for `count` matrices of one K and element type in ONE launch (the chromosomes a rank owns: the
 * call-site loop of rocco/rocco.py:948-991 over chromosomes).  matrices_dev / n / row_strides / scores_dev are
 * host arrays of `count` entries; the results are those of `count` rocco_hip_score_median calls. */
int rocco_hip_score_median_batch(rocco_hip_solver *solver, const void *const *matrices_dev, int dtype, size_t K,
                                 const size_t *n, const size_t *row_strides, double *const *scores_dev, size_t count,
                                 void *stream);

/* ... and, reduced inside the same launch, what the budgeted solve first asks of every score array
 * (rocco/dp.py:110-111 np.min / np.max; the sum of absolute values bounds its running values):
 * stats_dev[3 i .. 3 i + 3) = min, max, sum |.| of scores_dev[i] (device memory; min / max skip NaN scores, the sum
 * carries them).  Every n[i] must be positive.  Copy them to the host and hand them to
 * rocco_hip_solve_budget_batch_stats_f64: the scores are then not read a second time. */
int rocco_hip_score_median_batch_stats(rocco_hip_solver *solver, const void *const *matrices_dev, int dtype, size_t K,
                                       const size_t *n, const size_t *row_strides, double *const *scores_dev,
                                       size_t count, double *stats_dev, void *stream);

/* The other branches of score_central_tendency_chrom (not reached from the reference's driver): the nearest-rank
 * quantile of rocco.py:267-272 -- `rank` (0-based position in the sorted column) is computed by the caller with
 * NumPy's own rule, np.quantile(np.arange(K), q, method="nearest") -- and the column mean of rocco.py:298-299
 * (rows added in order, one division).  Same matrix conventions as rocco_hip_score_median; NaN in a column gives NaN. */
int rocco_hip_score_order_statistic(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K, size_t n,
                                    size_t row_stride, int rank, double *scores_dev, void *stream);
int rocco_hip_score_mean(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K, size_t n,
                         size_t row_stride, double *scores_dev, void *stream);

/* method="tmean" (rocco/rocco.py:273-297): per column the values outside [q_lo, q_hi] (the order statistics of ranks
 * rank_lo <= rank_hi: np.quantile(..., method="nearest") at tprop and 1 - tprop) are dropped and the rest averaged, summed as
 * SciPy 1.15's stats.tmean sums them (dropped values as 0.0, NumPy's pairwise order over the K entries): bit for bit. */
int rocco_hip_score_trimmed_mean(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K, size_t n,
                                 size_t row_stride, int rank_lo, int rank_hi, double *scores_dev, void *stream);

/* np.power(scores, power) (rocco/rocco.py:255, 304): exact for power 2 (NumPy squares); any other exponent through the
 * device's pow -- NumPy's own pow differs between its SVML and libm builds in the last place, so no algorithm matches
 * it on every host (power 1, the driver's value, never comes here). */
int rocco_hip_power_f64(rocco_hip_solver *solver, const double *x_dev, double power, double *out_dev, size_t n, void *stream);

/* ---- chain solve at a fixed selection penalty ----------------------------------------------
 * Replaces rocco/_chain_dp.c:9-213 `solve_penalized_chain` (Python wrapper rocco/dp.py:49-86).
 * `switch_costs_dev` has n-1 entries, or is NULL to use the scalar `gamma` at every boundary
 * (what rocco/dp.py:37-46 materialises).  `solution_dev` (n bytes, 0/1) may be NULL.
 * value/count are the reference's (best_val, best_count) (rocco/_chain_dp.c:194-198); the value is
 * reproduced bit-for-bit only on the exact path, otherwise to ~1e-12 relative. */
int rocco_hip_solve_penalized_chain_f64(rocco_hip_solver *solver, const double *scores_dev,
                                        const double *switch_costs_dev, double gamma, size_t n,
                                        double selection_penalty, uint8_t *solution_dev,
                                        double *value_out, long long *count_out, int *path_out,
                                        void *stream);

/* ---- budgeted solve, one launch sequence for a batch of chromosomes -------------------------
 * Replaces rocco/dp.py:89-164 `calibrate_selection_penalty` (bracket + exactly `max_iter`
 * bisection steps on the selection penalty) for every task in the batch at once.  The bracket
 * seeds are formed as the reference does (rocco/dp.py:110-111: lower = min(s) - sum(c) - 1,
 * upper = max(s) + sum(c) + 1): min / max are computed on the device (exact), `sum_costs` must be
 * supplied by the caller as NumPy's pairwise np.sum(switch_costs) gives it, because its last bits
 * fix the midpoint sequence. */
typedef struct {
    const double *scores_dev;       /* n doubles                                        */
    const double *switch_costs_dev; /* n-1 doubles or NULL (use gamma)                  */
    double gamma;
    size_t n;
    long long target_count;         /* int(floor(n * budget)), rocco/dp.py:197          */
    double sum_costs;               /* np.sum(switch_costs), rocco/dp.py:110-111        */
    int max_iter;                   /* 60 in the reference (rocco/dp.py:93)             */
    uint8_t *solution_dev;          /* n bytes out                                      */
} rocco_hip_budget_task;

typedef struct {
    double selection_penalty; /* the returned `upper` (rocco/dp.py:164)                          */
    double penalized_value;   /* best_value                                                      */
    long long selected_count; /* best_count                                                      */
    int evaluations;          /* chain evaluations the reference would have made (62 normally)   */
    int path;                 /* ROCCO_HIP_PATH_*                                                */
    int passes;               /* device rounds this task took part in                            */
    int zone_iters;           /* bisection steps left at the first uncertain probe (-1: none)    */
    long long n_diff;         /* decisions left open by the final window (-1: no window)         */
    int maps;                 /* binade maps built for this task                                 */
} rocco_hip_budget_result;

int rocco_hip_solve_budget_batch_f64(rocco_hip_solver *solver, size_t n_tasks,
                                     const rocco_hip_budget_task *tasks,
                                     rocco_hip_budget_result *results, void *stream);

/* The same with the score statistics the solve starts from -- np.min(scores), np.max(scores) (rocco/dp.py:110-111)
 * and sum |scores| (bounds the running values) -- handed in by the caller: score_stats_host[3 t .. 3 t + 3) in HOST
 * memory, exactly as rocco_hip_score_median_batch_stats produced them for these very arrays (NULL: computed here,
 * as above; ignored for a batch with cost vectors).  Saves one pass over every score array and one device round. */
int rocco_hip_solve_budget_batch_stats_f64(rocco_hip_solver *solver, size_t n_tasks,
                                           const rocco_hip_budget_task *tasks, const double *score_stats_host,
                                           rocco_hip_budget_result *results, void *stream);

/* ---- delta-form evaluation (the parallel kernels behind the two solves above) ----------------
 * Count-only evaluation of the chain at several penalties in one pass over the scores, with the
 * certification statistics of DESIGN.md section 4, and the joint "window" evaluation over a
 * penalty interval.  These are what rocco_hip_solve_budget_batch_f64 drives; they are exported so
 * that the kernels can be checked bit-for-bit against their sequential definition
 * (oracle/delta_oracle.c) and so that callers can run their own searches.  They stand where the
 * reference calls rocco/_chain_dp.c once per penalty (rocco/dp.py:113-162). */
typedef struct {
    long long count;     /* selected loci under the exact-rule classes                          */
    long long uncertain; /* loci whose class is not certified                                   */
    long long effect;    /* bound on |count(reference) - count| (n + 1 if the model overflowed) */
    long long max_run;   /* longest run without a clear clamp behind a locus with tolerance      */
} rocco_hip_probe_stats;

/* `emap_dev`: per-chunk binade codes (ceil(n / 32) bytes, see rocco_hip_delta_build_map_f64) or NULL. */
int rocco_hip_delta_probe_f64(rocco_hip_solver *solver, const double *scores_dev,
                              const double *switch_costs_dev, double gamma, size_t n,
                              const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                              rocco_hip_probe_stats *stats_out, void *stream);

/* Binade map at penalty lambda_ref: for every 32-locus chunk, the binade of the reference's running
 * value (rocco/_chain_dp.c:118-129 operate on prev0 / prev1 of that magnitude) and whether it keeps
 * a distance > margin from every power of two.  Writes ceil(n / 32) bytes to emap_dev. */
int rocco_hip_delta_build_map_f64(rocco_hip_solver *solver, const double *scores_dev,
                                  const double *switch_costs_dev, double gamma, size_t n,
                                  double lambda_ref, double margin, uint8_t *emap_dev, void *stream);

/* The same codes from the lean kernels the budgeted solve uses for the first map of its compacted problems (lean.hip:
 * lean_map_kernel, lean_mapcode_kernel: one tile kernel + one code kernel instead of six launches); scalar switch cost,
 * n >= 2.  Exported so that the two builders can be compared byte for byte. */
int rocco_hip_delta_build_map_lean_f64(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                                       double lambda_ref, double margin, uint8_t *emap_dev, void *stream);

typedef struct {
    long long count_lo, count_hi; /* selected loci of fill(LO) / fill(HI)                 */
    long long n_diff;             /* loci whose class differs between LO and HI           */
    int diff_adjacent;            /* every difference is one class step                   */
    int overflow;                 /* tolerance model exceeded somewhere                   */
    long long max_run;
    long long diff_locus[16];     /* first differences, ascending (-1 = unused)           */
    double diff_margin_lo[16], diff_margin_hi[16];
    long long diff_run[16];
    int diff_cls_lo[16], diff_cls_hi[16];
} rocco_hip_window_stats;

/* Writes fill(LO) (n bytes) to solution_dev. */
int rocco_hip_delta_window_f64(rocco_hip_solver *solver, const double *scores_dev,
                               const double *switch_costs_dev, double gamma, size_t n,
                               const uint8_t *emap_dev, double lambda_lo, double lambda_hi,
                               uint8_t *solution_dev, rocco_hip_window_stats *stats_out, void *stream);

/* The counts rocco_hip_delta_probe_f64 gives -- what rocco/_chain_dp.c returns as best_count at each penalty, through
 * the rounding model of oracle/delta_oracle.c -- from the lean kernel that carries several penalties per workgroup
 * (lean.hip: lean_model_kernel; what the budgeted solve runs on its compacted problems).  It certifies more
 * conservatively than the full kernels: open_out[i] != 0 says counts_out[i] is NOT certified equal to the reference's
 * (the solve repeats such a penalty with the full kernels); open_out[i] == 0 says it is.  Needs a binade map
 * (rocco_hip_delta_build_map_f64); scalar switch cost only; n >= 2. */
int rocco_hip_delta_model_lean_f64(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                                   const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                                   long long *counts_out, long long *open_out, void *stream);

/* Count-only evaluation in exact arithmetic on the problem's grid q (the "bound" evaluation of DESIGN.md
 * section 4.4: no rounding model; shifted by -/+ eps it brackets what rocco/_chain_dp.c returns as
 * best_count), in `n_rounds` rounds of up to 32 penalties each (round r holds round_sizes[r] of them).  A round
 * whose penalties all lie at or above a penalty of the round before it runs on the loci that one selected
 * (the selected sets are nested in the penalty) -- the counts are the same, the
 * work is not.  lambdas_used_out: the penalties as evaluated (snapped to the grid); level_len_out[r]: loci
 * of the array round r ran on (n = the caller's).  Scalar switch cost only. */
int rocco_hip_delta_bound_rounds_f64(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                                     const double *lambdas, const int *round_sizes, int n_rounds,
                                     double *lambdas_used_out, long long *counts_out, long long *level_len_out,
                                     void *stream);

/* Exact counts (and optionally one exact solution) for up to 64 penalties through the "spine":
 * the parallel kernels record per-chunk state, then one wavefront per chromosome carries the
 * reference's running values exactly -- stepping only through the chunks where the parallel
 * recursion is not provably the reference's own.  `emap_dev` must be a map
 * built by rocco_hip_delta_build_map_f64 at (or near) these penalties.  counts_out[i] is what
 * rocco/_chain_dp.c returns as best_count for lambdas[i]; solution_dev (n bytes) receives the
 * solution of lambdas[solution_index] when solution_index >= 0. */
int rocco_hip_delta_spine_f64(rocco_hip_solver *solver, const double *scores_dev,
                              const double *switch_costs_dev, double gamma, size_t n,
                              const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                              int solution_index, uint8_t *solution_dev, long long *counts_out,
                              void *stream);

/* ---- objective ------------------------------------------------------------------------------
 * Replaces rocco/dp.py:16-34 `objective_value`: -(s . z) + c . |diff z|  (fixed-order tree sum;
 * the reference uses BLAS dot, so agreement is to relative 1e-12, not bitwise). */
int rocco_hip_objective_value_f64(rocco_hip_solver *solver, const uint8_t *solution_dev,
                                  const double *scores_dev, const double *switch_costs_dev,
                                  double gamma, size_t n, double *objective_out, void *stream);

/* ---- run-length decode ------------------------------------------------------------------------
 * Replaces the per-locus Python loop + merge of rocco/rocco.py:180-190 (`chrom_solution_to_bed`)
 * and rocco/rocco.py:74-95 (`_merge_bed_records`) for contiguous fixed-step loci: maximal runs of
 * selected loci among loci 0..n-2 (the last locus is never emitted, rocco.py:180) are written as
 * half-open locus index pairs [run_begin, run_end) in ascending order; the interval in base pairs
 * is (intervals[run_begin], intervals[run_end]).  At most `capacity` runs are written;
 * *n_runs_out is the total found. */
int rocco_hip_decode_runs(rocco_hip_solver *solver, const uint8_t *solution_dev, size_t n,
                          int64_t *run_begin_dev, int64_t *run_end_dev, size_t capacity,
                          size_t *n_runs_out, void *stream);

/* The same for `count` solutions in three launches and ONE synchronisation (the chromosomes a rank owns; the loop
 * over chromosomes of rocco/rocco.py:1176-1196).  Host arrays of `count` entries; n_runs_out[i] may exceed
 * capacities[i] (then only the first capacities[i] runs were written: call again with more room). */
/* The same as ONE table: the runs of solution i, in ascending order, are rows row_offsets_out[i] .. row_offsets_out[i + 1]
 * of table_dev, each row three int64 (units[i], run_begin, run_end) -- what a rank hands to the interval gather
 * (exchange 2) or reads back to write its BED records.  count <= 48.  row_offsets_out has
 * count + 1 entries; when row_offsets_out[count] > capacity_rows only the first capacity_rows rows were written: call
 * again with more room.  table_host_out (may be NULL): receives a pointer to the table in pinned host memory owned by
 * the solver, valid until the next call on this solver (NULL when the table did not fit): the rows that exist are copied
 * there by a kernel in front of the call's one synchronisation (their number is known on the device only; `eager_rows`, the
 * size of an earlier host-sized copy, is ignored). */
int rocco_hip_decode_runs_table(rocco_hip_solver *solver, size_t count, const uint8_t *const *solutions_dev, const size_t *n,
                                const long long *units, int64_t *table_dev, size_t capacity_rows, size_t eager_rows,
                                size_t *row_offsets_out, const int64_t **table_host_out, void *stream);
int rocco_hip_decode_runs_batch(rocco_hip_solver *solver, size_t count, const uint8_t *const *solutions_dev, const size_t *n,
                                int64_t *const *run_begin_dev, int64_t *const *run_end_dev, const size_t *capacities,
                                size_t *n_runs_out, void *stream);

/* ---- row baselines ------------------------------------------------ */
/* Cross-fit Whittaker baseline of every row of a row-major rows x cols matrix.
 * Replaces rocco_crossfit_whittaker_baseline_matrix_f64 (rocco/native/baseline_backend.h:12-23,
 * baseline_backend.c:305-334) as called through rocco/_baseline.c:16-104 from rocco/inference.py:185-209,
 * on device buffers: same arguments, same results bit for bit, cols < 25 -> zeros.  matrix_dev and
 * baseline_out_dev must not overlap.  Returns 0, ROCCO_HIP_ENOMEM (the reference's -1) or EINVAL / EHIP. */
int rocco_hip_crossfit_whittaker_baseline_matrix_f64(rocco_hip_solver *solver, const double *matrix_dev,
                                                     size_t rows, size_t cols, double penalty_lambda,
                                                     double *baseline_out_dev, void *stream);
/* The same for `count` matrices of ONE penalty -- the chromosomes of a genome -- in one pair of launches: every group of
 * 8 rows of every matrix is a workgroup, and the pair lasts as long as the longest row (the loop over chromosomes of
 * rocco/rocco.py:948-1018 around the call of rocco/inference.py:198-206).  Host arrays of `count` entries. */
int rocco_hip_crossfit_whittaker_baseline_batch_f64(rocco_hip_solver *solver, size_t count, const double *const *matrices_dev,
                                                    const size_t *rows, const size_t *cols, double penalty_lambda,
                                                    double *const *baselines_dev, void *stream);
/* The batch form with the two elementwise statements around the fit folded into its sweeps (rocco/inference.py:330-338:
 * `global_centered = log_matrix - offsets; local = baseline(global_centered); centered = global_centered - local`):
 * centered_out_dev[i] = (matrices_dev[i] - row_offsets_dev[i][row]) - baseline of (matrices_dev[i] - row_offsets_dev[i][row]),
 * every operation rounded as the separate statements round it.  row_offsets_dev (may be NULL, entries may be NULL): K_i doubles per
 * matrix.  centered_out_dev[i] must not be matrices_dev[i].  ROCCO_HIP_EINVAL ("Local baseline fit produced non-finite values",
 * inference.py:207-208) when a baseline is not finite. */
int rocco_hip_crossfit_whittaker_residual_batch_f64(rocco_hip_solver *solver, size_t count, const double *const *matrices_dev,
                                                    const double *const *row_offsets_dev, const size_t *rows, const size_t *cols,
                                                    double penalty_lambda, double *const *centered_out_dev, void *stream);

/* The same with the sweeps' scratch (the forward sweep's two parities: twice the matrices' bytes, + records) in a block of
 * the caller's -- a framework that pools device memory keeps one pool that way instead of two that cannot see each other's reserves
 * (round 5: the composed driver's budget null wanted the ~100 GB a K = 100 genome's sweeps had left idle in the solver).
 * scratch_dev: at least rocco_hip_whittaker_batch_scratch_bytes(count, rows, cols) bytes, aligned to 256. */
size_t rocco_hip_whittaker_batch_scratch_bytes(size_t count, const size_t *rows, const size_t *cols);
int rocco_hip_crossfit_whittaker_residual_batch_scratch_f64(rocco_hip_solver *solver, size_t count, const double *const *matrices_dev,
                                                            const double *const *row_offsets_dev, const size_t *rows,
                                                            const size_t *cols, double penalty_lambda, double *const *centered_out_dev,
                                                            void *scratch_dev, size_t scratch_bytes, void *stream);

/* ---- centred-WLS locus scores ------------------------------------
 * Replaces rocco_score_centered_wls_f64 (rocco/native/wls_backend.h:11-28, wls_backend.c:744-947) as
 * called through rocco/_wls.c from rocco/inference.py:231-299 (`_score_centered_wls_matrix`), on device
 * buffers: centered_dev is the row-major K x n matrix of baseline-subtracted tracks; the six outputs are
 * n doubles each; *df_out / *window_out (host, may be NULL) receive the degrees of freedom and the
 * spatial window used.  Same arguments and, on finite input, the same results bit for bit.  Differences:
 * non-finite values are rejected with EINVAL (the reference drops such pairs from the trend fit).  Any spatial
 * window is taken (up to 63 loci the rolling sums run on LDS tiles, above that straight from memory; the
 * reference's caller always passes 31).
 * Returns 0, ROCCO_HIP_ENOMEM (the reference's -1), EINVAL (its -2) or EHIP. */
int rocco_hip_score_centered_wls_f64(rocco_hip_solver *solver, const double *centered_dev, size_t K, size_t n,
                                     double lower_bound_z, double prior_df, double min_effect, int use_min_effect,
                                     int spatial_window, double precision_floor_ratio, double *mean_dev,
                                     double *raw_var_dev, double *prior_var_dev, double *mod_var_dev, double *se_dev,
                                     double *scores_dev, double *df_out, int *window_out, void *stream);
/* The first step of the scoring above -- every row's rolling AR(1) innovation variances (wls_backend.c:610-742) -- for the
 * rows of `count` matrices in ONE launch (the chromosomes of a genome; one workgroup per row, two per compute unit):
 * variances_dev[i] receives K[i] x (n[i] - window_i + 1) doubles, window_i = the spatial window the scoring resolves for
 * n[i] (at most 63 here).  rocco_hip_score_centered_wls_given_variances_f64 is the scoring with that step already done
 * (variances_dev == NULL: the plain call). */
int rocco_hip_wls_rolling_variances_batch_f64(rocco_hip_solver *solver, size_t count, const double *const *centered_dev,
                                              const size_t *K, const size_t *n, int spatial_window, double *const *variances_dev,
                                              void *stream);
int rocco_hip_score_centered_wls_given_variances_f64(rocco_hip_solver *solver, const double *centered_dev, size_t K, size_t n,
                                                     double lower_bound_z, double prior_df, double min_effect, int use_min_effect,
                                                     int spatial_window, double precision_floor_ratio, const double *variances_dev,
                                                     double *mean_dev, double *raw_var_dev, double *prior_var_dev, double *mod_var_dev,
                                                     double *se_dev, double *scores_dev, double *df_out, int *window_out, void *stream);

/* Diagnostic of the log scale above: the device's log2 is a two-stage evaluation (a fast double-double stage accepted when its
 * result survives its error bound on either side, else a full one with error ~2^-100); this runs `count` inputs of a family
 * (0: any positive finite bit pattern, 1: next to 1, 2: the integers first .. first + count - 1, 3: mantissas next to the
 * table's cell boundaries in random binades, 4: uniform in [0.5, 4)) through both and through the full stage alone and
 * returns in *mismatches_out how many results differ (0 expected). */
int rocco_hip_log2_selfcheck(rocco_hip_solver *solver, int family, unsigned long long seed, unsigned long long first, size_t count,
                             unsigned long long *mismatches_out, void *stream);

/* How many rows of this solver's last centred-WLS call fitted their variance trend on the sorted path (two radix sorts of the
 * row's pairs: rows shorter than 4096 loci, rows with a run of equal |value| across a bin boundary, rows with more than 8192
 * values in one cell of the rank finder) instead of the sort-free one.  Diagnostic: the results are the same either way. */
int rocco_hip_wls_sorted_rows(const rocco_hip_solver *solver);

/* ---- count-path glue of score_loci_wls ---------------------------------
 * Replaces the NumPy statements of rocco/inference.py:40-47 (`_log_scale_wls_matrix`) and 330-331 (pilot
 * offset): centered_out[k][i] = log2(max(counts[k][i], 0) + pseudocount) - median_i(log2(...)[k][:]).
 * The row medians are exact order statistics (np.median: mean of the two middle values for even n); the
 * logarithm is CORRECTLY ROUNDED (log2_cr.h; see rocco_hip_log_scale_f64 below).  NumPy's own log2 is an SVML
 * routine or libm's, neither correctly rounded: it is one ulp off on ~0.03 % of integer counts, and what that does
 * downstream is measured in tests/test_gpu_score_loci_wls.py and stated in INTEGRATION.md section 5.  centered_out_dev
 * may alias counts_dev; row_offsets_out_dev (K doubles, may be NULL) receives the medians.  apply_log2 == 0
 * takes the matrix as already log-scaled (only the pilot offset is removed; bit-exact).  Non-finite
 * input -> EINVAL (the reference raises ValueError). */
/* out = log2(max(values, 0) + pseudocount) elementwise (`_log_scale_wls_matrix`, rocco/inference.py:40-47), CORRECTLY
 * ROUNDED.  The reference calls np.log2, which is an AVX-512 SVML routine or libm's log2 depending on the host; neither
 * is correctly rounded (in this image ~0.03 % of integer counts are one ulp off) and they differ from each other, so
 * the correctly rounded value is the host-independent target.  ROCCO_HIP_EINVAL for a non-finite value. */
int rocco_hip_log_scale_f64(rocco_hip_solver *solver, const double *values_dev, size_t count, double pseudocount, double *out_dev,
                            void *stream);

int rocco_hip_log_scale_center_rows_f64(rocco_hip_solver *solver, const double *counts_dev, size_t K, size_t n,
                                        double pseudocount, int apply_log2, double *centered_out_dev,
                                        double *row_offsets_out_dev, void *stream);

/* The same log scale and the same row medians WITHOUT subtracting them: log_out_dev = log2(max(counts, 0) + pseudocount)
 * (may alias counts_dev), row_offsets_out_dev (K doubles, required) = the medians (rocco/inference.py:330).  For callers whose next
 * pass subtracts the offsets on its way (rocco_hip_crossfit_whittaker_residual_batch_f64): one pass over the matrix less. */
int rocco_hip_log_scale_row_offsets_f64(rocco_hip_solver *solver, const double *counts_dev, size_t K, size_t n,
                                        double pseudocount, int apply_log2, double *log_out_dev, double *row_offsets_out_dev,
                                        void *stream);

/* out = a - b, element by element (rocco/inference.py:335 `centered = global_centered - local_baselines`);
 * out_dev may alias a_dev or b_dev. */
int rocco_hip_subtract_f64(rocco_hip_solver *solver, const double *a_dev, const double *b_dev, double *out_dev,
                           size_t count, void *stream);
/* out = a - b with the reference's check on b folded in (rocco/inference.py:207-208, "Local baseline fit produced non-finite
 * values"): ROCCO_HIP_EINVAL when b holds a non-finite value.  Synchronises the stream. */
int rocco_hip_subtract_finite_f64(rocco_hip_solver *solver, const double *a_dev, const double *b_dev, double *out_dev,
                                  size_t count, void *stream);

/* ---- narrowPeak summit offsets ---------------------------------------
 * Replaces the per-peak NumPy statements of rocco/rocco.py:838-872 (`_write_narrowpeak_summit_offsets`) over the
 * summit track of rocco/rocco.py:809-835 (`_cpy_narrowpeak_summit_track`), for the peaks of one chromosome:
 * intervals_dev = the chromosome's locus starts (ascending, n_intervals of them), effect_mean_dev = the WLS mean
 * per locus (n_mean doubles, rounded to float32 inside as the reference stores it), peaks as half-open base-pair
 * intervals.  offsets_out[p] = clip(centre of the first locus with the largest non-NaN mean among the loci
 * starting in [start, end) - start, 0, end - start - 1), or -1 when the peak is empty, holds no locus or no
 * finite value.  centers_dev == NULL: the centre of locus i is (intervals[i] + intervals[i+1]) // 2 and the last
 * entry of intervals only closes the last locus (rocco.py:816-822); otherwise intervals_dev / centers_dev are the
 * `starts` / `centers` arrays of a stored summit track (rocco.py:855-857), n_intervals each.  Exact (integer /
 * compare work). */
int rocco_hip_narrowpeak_summit_offsets(rocco_hip_solver *solver, const int64_t *intervals_dev, size_t n_intervals,
                                        const int64_t *centers_dev, const double *effect_mean_dev, size_t n_mean, const int64_t *peak_start_dev,
                                        const int64_t *peak_end_dev, size_t n_peaks, int64_t *offsets_out_dev,
                                        void *stream);

/* ---- signal matrix assembly --------------------------------------------
 * The NumPy statements at the end of generate_chrom_matrix (rocco/readtracks.py:614-633) on device buffers.
 * rocco_hip_union_intervals: unique_out = np.sort(np.unique(values)) (capacity `count`), *n_unique_out = its
 * length, *fixed_step_out (may be NULL) = 1 iff np.unique(np.diff(unique_out)).size <= 1 (the bigWig check,
 * readtracks.py:615-620).
 * rocco_hip_scatter_tracks: matrix[k, np.searchsorted(common, intervals_k)] = vals_k on a zero matrix (K x m,
 * row-major, out_dtype 0 = float64, 1 = float32 as with low_memory); track k owns the entries
 * offsets_host[k] .. offsets_host[k+1] of the concatenated interval / value arrays; repeated loci inside a
 * track keep the last value, as NumPy's fancy-index assignment does.  Exact. */
int rocco_hip_union_intervals(rocco_hip_solver *solver, const int64_t *values_dev, size_t count,
                              int64_t *unique_out_dev, size_t *n_unique_out, int *fixed_step_out, void *stream);
int rocco_hip_scatter_tracks(rocco_hip_solver *solver, const int64_t *common_dev, size_t m,
                             const int64_t *intervals_concat_dev, const double *vals_concat_dev,
                             const size_t *offsets_host, size_t K, int out_dtype, void *matrix_out_dev, void *stream);

/* ---- the K x n part of the wild-bootstrap budget null -------------------
 * The random multipliers stay with NumPy on the host (rocco/inference.py:540-571 draws them from
 * np.random.default_rng streams); what moves to the device is what the reference does with them per draw
 * (rocco/inference.py:628-685 `_compute_budget_null_draw`): the K x n product, the WLS rescoring
 * (rocco_hip_score_centered_wls_f64) and the four means -- and the residual template they start from (688-722).
 * rocco_hip_numpy_sum_f64: np.sum of a contiguous float64 vector in NumPy's own order (8192-element buffer chunks,
 *   pairwise summation inside a chunk, running total over the chunks), bit for bit.
 * rocco_hip_budget_null_draw_stats_f64: stats_out[0..4) = np.mean(pos), np.mean(pos / null_soft_scale),
 *   np.mean(pos > 0), np.mean(scores > null_threshold) with pos = np.clip(scores - null_center, 0, None)
 *   (inference.py:676-684), summed in NumPy's order.
 * rocco_hip_multiply_f64: out = a * b elementwise (inference.py:665).
 * rocco_hip_subtract_positive_row_f64: out[k][i] = matrix[k][i] - max(row[i], 0) (inference.py:717-721). */
int rocco_hip_numpy_sum_f64(rocco_hip_solver *solver, const double *x_dev, size_t n, double *sum_out, void *stream);
int rocco_hip_budget_null_draw_stats_f64(rocco_hip_solver *solver, const double *scores_dev, size_t n,
                                         double null_center, double null_soft_scale, double null_threshold,
                                         double *stats_out, void *stream);
int rocco_hip_multiply_f64(rocco_hip_solver *solver, const double *a_dev, const double *b_dev, double *out_dev,
                           size_t count, void *stream);
int rocco_hip_subtract_positive_row_f64(rocco_hip_solver *solver, const double *matrix_dev, const double *row_dev,
                                        size_t K, size_t n, double *out_dev, void *stream);

/* ---- sizing a solver's buffers ahead of a count-path batch ------------------------------------------------------
 * The count-path entries grow the solver's scratch on demand (hipMalloc + hipFree: device-wide synchronisations).  A
 * caller that runs several solver handles side by side on streams of their own (rocco_amd.inference.score_loci_wls_batch_device)
 * sizes each handle for the matrices it will see BEFORE the threads start: rows[i] x cols[i], i < count; penalty_lambda > 0
 * also builds (or extends) the device's Whittaker factor for the longest row.  rocco_hip_buffer_growths(): how many times
 * any solver buffer of the process has grown so far (diagnostic). */
int rocco_hip_count_path_reserve(rocco_hip_solver *solver, size_t count, const size_t *rows, const size_t *cols,
                                 double penalty_lambda, void *stream);
/* sweeps_scratch_is_the_callers != 0: the baseline sweeps will get their scratch from the caller
 * (rocco_hip_crossfit_whittaker_residual_batch_scratch_f64), nothing is reserved for them */
int rocco_hip_count_path_reserve_ex(rocco_hip_solver *solver, size_t count, const size_t *rows, const size_t *cols,
                                    double penalty_lambda, int sweeps_scratch_is_the_callers, void *stream);
long long rocco_hip_buffer_growths(void);
/* Diagnostic, process-wide since load: the batched baseline sweeps cut long rows into segments whose workgroups start from
 * a warm-up (csrc/whittaker.hip); a seam whose warm-up had not reached the row's own values is recomputed from the true
 * state -- the results are the sequential sweep's (rocco/native/baseline_backend.c:142-172) either way.  How many seams
 * (row, parity, sweep) were recomputed so far.  ROCCO_HIP_WHITTAKER_SEGMENT_LOCI / ROCCO_HIP_WHITTAKER_WARM_LOCI (read
 * per call) set the segment length (0: rows are never cut) and the warm-up. */
long long rocco_hip_whittaker_seam_repairs(void);
/* device memory the solver's scratch buffers hold now (they are kept between calls) */
long long rocco_hip_solver_device_bytes(const rocco_hip_solver *solver);
/* Diagnostic, process-wide since load: how the last bisection steps of the calibrations (rocco/dp.py:141-162) were
 * sequenced -- out[0] chains of rounding-model rounds queued by the device-side director (csrc/model_chain.h), out[1]
 * certified counts the host took over from them, out[2] counts its replay of the reference's steps was answered from
 * them without device work, out[3] counts it asked for that the chain had not evaluated (answered by regular rounds). */
void rocco_hip_model_chain_counters(long long out[4]);
/* ... and how their last step -- the window that certifies and writes the solution of the final penalty (rocco/dp.py:164, the
 * solve at `upper`) -- was served: out[0] solutions a chain wrote itself at the penalty its bisection ended at
 * (lean_write_solutions_kernel from the class words of that penalty's certified evaluation), out[1] final windows answered
 * from them without device work. */
void rocco_hip_model_chain_written_counters(long long out[2]);

/* ---- the multipliers of the bootstrap draws on the device (missing item 2) ---------------------
 * Replaces rocco/inference.py:546-575 `_generate_dependent_wild_weights` (called per row and draw at 654-664 and per
 * draw at 1206-1213): NumPy's `Generator.standard_normal` over PCG64, SciPy's `fftconvolve(..., "valid")` with the
 * Bartlett taps (inference.py:534-543), centring and scaling to unit variance.
 * rocco_hip_pcg64_standard_normal_f64: `count` values of `np.random.Generator(PCG64).standard_normal` continuing from
 *   the generator state (state, inc: the two 128-bit integers of `bit_generator.state["state"]`, high and low halves);
 *   *raw_draws_out = the 64-bit draws they consumed (`bit_generator.advance(raw_draws)` puts the host's generator
 *   where NumPy's would be).  NumPy's values bit for bit except that tail values (|x| > 3.654; 2.6e-4 of them) may
 *   differ in the last place (NumPy calls the host libm's log1p); positions in the stream never differ.
 * rocco_hip_bartlett_multipliers_f64: innovations_dev = rows x (n + n_taps - 1) row-major; weights_dev = rows x n:
 *   every row's "valid" convolution with the taps (direct sum in a fixed order -- SciPy's FFT result to ~1e-16 of the
 *   scale, not bit for bit), minus its mean, over its standard deviation.  *degenerate_out != 0: a row's deviation is
 *   <= 1e-8 (the reference then draws signs, inference.py:565-569: left to the caller; that row is returned unscaled).
 * Opt-in (`multipliers="device"` in rocco_amd.budget): the estimates agree with the host path to ~1e-12, not bit for bit. */
int rocco_hip_pcg64_standard_normal_f64(rocco_hip_solver *solver, unsigned long long state_hi, unsigned long long state_lo,
                                        unsigned long long inc_hi, unsigned long long inc_lo, size_t count, double *values_dev,
                                        unsigned long long *raw_draws_out, void *stream);
int rocco_hip_bartlett_multipliers_f64(rocco_hip_solver *solver, const double *innovations_dev, size_t rows, size_t n,
                                       const double *taps_host, size_t n_taps, double *weights_dev, int *degenerate_out,
                                       void *stream);

/* ---- score-track budget estimate: the n-long pieces (rocco/inference.py:1151-1421, 446-501; rocco/rocco.py:751-789) ----
 * rocco_hip_sort_f64: ascending sorted copy of a float64 vector (-0.0 before +0.0).  Every np.median / MAD / median
 *   of the positive scores of the estimate is an order statistic of it.
 * rocco_hip_sorted_probe_f64: values at up to 8 ranks of a sorted vector, and for up to 8 thresholds t how many
 *   elements x have (x - shift) <= t and (x - shift) < t (binary search: x - shift is monotone in x).
 * rocco_hip_autocovariance_sums_f64: sums_out[k] = sum_i (x_i - mean)(x_{i+k} - mean), k = 0..max_lag (< n; any number of lags,
 *   1024 per pair of launches), summed in a fixed order.  (The reference takes them from an FFT: agreement to ~1e-13 relative, not bitwise.)
 * rocco_hip_negative_part_f64: out = scores - clip(scores, 0, None) (the residual template of the direct-score null).
 * rocco_hip_soft_counts_f64: out = clip(scores - center, 0, None) / scale (the series whose autocorrelation time is taken). */
int rocco_hip_sort_f64(rocco_hip_solver *solver, const double *x_dev, size_t n, double *sorted_out_dev, void *stream);
int rocco_hip_sorted_probe_f64(rocco_hip_solver *solver, const double *sorted_dev, size_t n, const long long *ranks,
                               size_t n_ranks, double *values_out, double shift, const double *thresholds,
                               size_t n_thresholds, long long *counts_le_out, long long *counts_lt_out, void *stream);
int rocco_hip_autocovariance_sums_f64(rocco_hip_solver *solver, const double *x_dev, size_t n, double mean, int max_lag,
                                      double *sums_out, void *stream);
int rocco_hip_negative_part_f64(rocco_hip_solver *solver, const double *scores_dev, double *out_dev, size_t n, void *stream);
int rocco_hip_soft_counts_f64(rocco_hip_solver *solver, const double *scores_dev, double center, double scale, double *out_dev,
                              size_t n, void *stream);

/* ---- post-hoc peak scoring: the per-peak arithmetic (rocco/scores.py:180-194, 128-141, 560-583) ----
 * rocco_hip_peak_signal_stat_f64: counts_dev is [n_peaks][n_samples] (scaled counts, row-major), lengths_dev the peak
 *   lengths; stat_out[p] = np.percentile(log2(max(counts[p] * row_scale / max(int(length), 1) + pc, pc)), percentile)
 *   (`_peak_signal_stat`; the logarithm correctly rounded, see rocco_hip_log_scale_f64; NaN in a row gives NaN).
 * rocco_hip_ecdf_survival_f64: bin_dev[p] selects the sorted null values null_values_dev[null_offsets_dev[b] ..
 *   null_offsets_dev[b + 1]); pvals_out[p] = (size - searchsorted_left(null, stat[p]) + 1) / (size + 1) (`EmpiricalNull.survival`).
 * rocco_hip_bh_adjust_f64: Benjamini-Hochberg adjusted p-values as scipy.stats.false_discovery_control(ps, method="bh")
 *   computes them for m > 1 (p * (m / rank), running minimum from the largest rank down, clip to [0, 1]). */
int rocco_hip_peak_signal_stat_f64(rocco_hip_solver *solver, const double *counts_dev, const double *lengths_dev, size_t n_peaks,
                                   size_t n_samples, double row_scale, double pc, double percentile, double *stat_out_dev, void *stream);
int rocco_hip_ecdf_survival_f64(rocco_hip_solver *solver, const double *stat_dev, const int *bin_dev, const double *null_values_dev,
                                const long long *null_offsets_dev, size_t n_peaks, double *pvals_out_dev, void *stream);
int rocco_hip_bh_adjust_f64(rocco_hip_solver *solver, const double *pvals_dev, size_t m, double *qvals_out_dev, void *stream);

/* bigWig dense fill: the NumPy statements of get_bigwig_chrom_scores after the file has been read
 * (rocco/readtracks.py:141-186) for one track's intervals in ascending order (as pyBigWig returns them).
 * *flags_out: bit 0 non-finite value (147-150), bit 1 non-positive width (153-156), bit 2 variable width (158-161),
 * bit 3 start off the fixed grid (165-168), bit 4 overlapping / duplicate / out-of-order bins (170-173); nothing is
 * written when a flag is set.  Otherwise *first_start_out, *step_out and *n_full_out describe
 * np.arange(starts[0], starts[-1] + step, step); with full_out_dev != NULL and capacity >= *n_full_out it receives
 * np.round(scatter(vals) * const_scale, round_digits) (no scaling for const_scale < 0; rounding as NumPy does it:
 * multiply by 10**d, round half to even, divide).  Call once with full_out_dev == NULL for the size.  Exact. */
int rocco_hip_bigwig_dense_fill_f64(rocco_hip_solver *solver, const int64_t *starts_dev, const int64_t *ends_dev,
                                    const double *vals_dev, size_t count, double const_scale, int round_digits,
                                    double *full_out_dev, size_t capacity, int64_t *first_start_out, int64_t *step_out,
                                    size_t *n_full_out, int *flags_out, void *stream);

/* ---- synthetic signal matrices (benchmark / test support, device-resident) -------------------
 * Fills a row-major [K][n] matrix with the counter-based synthetic tracks described in
 * DESIGN.md section 7 (5-decimal background + planted peaks with per-sample dropout); the same
 * integer arithmetic is restated in NumPy in rocco_amd/synth.py so any slice can be regenerated
 * on the host bit-for-bit. */
int rocco_hip_synth_matrix(rocco_hip_solver *solver, void *matrix_dev, int dtype, size_t K, size_t n,
                           size_t row_stride, uint64_t seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ROCCO_HIP_H */
