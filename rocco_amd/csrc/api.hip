// rocco_amd/csrc/api.hip -- C ABI entry points of librocco_hip.so (see include/rocco_hip.h).
#include <map>
#include <memory>
#include <mutex>
#include "kernels.h"
#include <atomic>

#include "budget.h"

#include <algorithm>
#include <cmath>
#include <new>

namespace rocco {

static thread_local std::string g_last_error;

void set_last_error(const std::string &msg) { g_last_error = msg; }

// How often a solver buffer had to grow (hipMalloc + hipFree, or the pinned twins): the count path's batch sizes every
// buffer of a pipeline before its worker threads start, so that no thread allocates or frees device memory -- both
// device-wide synchronisations -- while the others' streams are in flight (tests/test_gpu_count_path_batch.py reads this).
static std::atomic<long long> g_buffer_growths{0};

int DeviceBuffer::reserve(size_t want)
{
    if (want <= bytes) {
        return ROCCO_HIP_OK;
    }
    g_buffer_growths.fetch_add(1, std::memory_order_relaxed);
    // small buffers double (few growths); large ones take what is asked for + 1/16 -- a 40 GB scratch must not become 80 --
    // and the old block goes first (nothing is carried over: a scratch buffer's contents end with the call that wrote them)
    size_t grow = bytes ? bytes * 2 : (size_t)1 << 16;
    if (grow < want || want >= ((size_t)256 << 20)) {
        grow = want + ((want >= ((size_t)256 << 20)) ? want / 16 : 0);
    }
    if (ptr != nullptr) {
        (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    void *p = nullptr;
    ROCCO_HIP_TRY(hipMalloc(&p, grow));
    ptr = p;
    bytes = grow;
    return ROCCO_HIP_OK;
}

void DeviceBuffer::release()
{
    if (ptr != nullptr) {
        (void)hipFree(ptr);
    }
    ptr = nullptr;
    bytes = 0;
}

SharedFactor::~SharedFactor() { buf.release(); }

int PinnedBuffer::reserve(size_t want)
{
    if (want <= bytes) {
        return ROCCO_HIP_OK;
    }
    g_buffer_growths.fetch_add(1, std::memory_order_relaxed);
    size_t grow = bytes ? bytes * 2 : (size_t)1 << 14;
    if (grow < want) {
        grow = want;
    }
    void *p = nullptr;
    ROCCO_HIP_TRY(hipHostMalloc(&p, grow, coherent ? (hipHostMallocCoherent | hipHostMallocMapped) : hipHostMallocDefault));
    if (ptr != nullptr) {
        (void)hipHostFree(ptr);
    }
    ptr = p;
    bytes = grow;
    return ROCCO_HIP_OK;
}

void PinnedBuffer::release()
{
    if (ptr != nullptr) {
        (void)hipHostFree(ptr);
    }
    ptr = nullptr;
    bytes = 0;
}

}  // namespace rocco

using namespace rocco;

extern "C" {

int rocco_hip_abi_version(void) { return 1000; }

const char *rocco_hip_last_error(void) { return g_last_error.c_str(); }

int rocco_hip_solver_create(rocco_hip_solver **solver_out, int device)
{
    if (solver_out == nullptr) {
        return ROCCO_HIP_EINVAL;
    }
    *solver_out = nullptr;
    int count = 0;
    ROCCO_HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        set_last_error("rocco_hip_solver_create: no such HIP device");
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(device));
    rocco_hip_solver *s = new (std::nothrow) rocco_hip_solver();
    if (s == nullptr) {
        return ROCCO_HIP_ENOMEM;
    }
    s->device = device;
    *solver_out = s;
    return ROCCO_HIP_OK;
}

void rocco_hip_solver_destroy(rocco_hip_solver *solver)
{
    if (solver == nullptr) {
        return;
    }
    (void)hipSetDevice(solver->device);
    // whatever was queued with this handle's buffers has to be over before they go (a kernel that outlives its scratch
    // faults: "Memory access fault by GPU" at teardown)
    (void)hipDeviceSynchronize();
    solver->dev_tasks.release();
    solver->dev_params.release();
    solver->dev_results.release();
    solver->dev_bits.release();
    solver->dev_misc.release();
    solver->dev_solution.release();
    solver->dev_maps.release();
    solver->dev_frozen.release();
    solver->factor.reset();
    solver->dev_lean_pool.release();
    solver->dev_lean_round.release();
    solver->dev_lean_look.release();
    solver->dev_lean_progress.release();
    solver->dev_lean_desc.release();
    solver->dev_lean_wcap.release();
    solver->dev_chain.release();
    solver->host_chain.release();
    solver->host_follow.release();
    solver->host_objective.release();
    solver->dev_objective.release();
    solver->dev_map.release();
    solver->host_map_stage.release();
    solver->dev_median_partials.release();
    solver->host_lean_stage.release();
    solver->host_lean_back.release();
    solver->host_stage.release();
    solver->host_back.release();
    solver->host_table.release();
    delete solver;
}

int rocco_hip_solver_set(rocco_hip_solver *solver, const char *key, long long value)
{
    if (solver == nullptr || key == nullptr) {
        return ROCCO_HIP_EINVAL;
    }
    const std::string k(key);
    if (k == "force_exact") {
        solver->force_exact = value ? 1 : 0;
    } else if (k == "spec_depth") {
        if (value < 1 || value > 6) {
            return ROCCO_HIP_EINVAL;
        }
        solver->spec_depth = (int)value;
    } else if (k == "active_set") {
        solver->active_set = value ? 1 : 0;
    } else if (k == "lean") {
        solver->lean = value ? 1 : 0;
    } else if (k == "rolling_group_min") {
        if (value != 1 && value != 2 && value != 4 && value != 8) {
            return ROCCO_HIP_EINVAL;
        }
        solver->rolling_group_min = (int)value;
    } else {
        set_last_error("rocco_hip_solver_set: unknown key " + k);
        return ROCCO_HIP_EINVAL;
    }
    return ROCCO_HIP_OK;
}

int rocco_hip_score_median_batch(rocco_hip_solver *solver, const void *const *matrices_dev, int dtype, size_t K,
                                 const size_t *n, const size_t *row_strides, double *const *scores_dev, size_t count,
                                 void *stream)
{
    if (solver == nullptr || K == 0 || (dtype != 0 && dtype != 1) ||
        (count > 0 && (matrices_dev == nullptr || n == nullptr || row_strides == nullptr || scores_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    for (size_t i = 0; i < count; ++i) {
        if ((n[i] > 0 && (matrices_dev[i] == nullptr || scores_dev[i] == nullptr)) || row_strides[i] < n[i]) {
            return ROCCO_HIP_EINVAL;
        }
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_median_batch(matrices_dev, dtype, K, n, row_strides, scores_dev, count, (hipStream_t)stream);
}

int rocco_hip_score_median_batch_stats(rocco_hip_solver *solver, const void *const *matrices_dev, int dtype, size_t K,
                                       const size_t *n, const size_t *row_strides, double *const *scores_dev, size_t count,
                                       double *stats_dev, void *stream)
{
    if (stats_dev == nullptr) {
        return rocco_hip_score_median_batch(solver, matrices_dev, dtype, K, n, row_strides, scores_dev, count, stream);
    }
    if (solver == nullptr || K == 0 || (dtype != 0 && dtype != 1) ||
        (count > 0 && (matrices_dev == nullptr || n == nullptr || row_strides == nullptr || scores_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    for (size_t i = 0; i < count; ++i) {
        if (n[i] == 0 || matrices_dev[i] == nullptr || scores_dev[i] == nullptr || row_strides[i] < n[i]) {
            return ROCCO_HIP_EINVAL;  // (statistics of an empty score array do not exist)
        }
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    const size_t blocks = median_partials_count(n, count);
    int rc;
    if ((rc = solver->dev_median_partials.reserve(3 * blocks * sizeof(double) + 256)) != ROCCO_HIP_OK) return rc;
    return launch_median_batch(matrices_dev, dtype, K, n, row_strides, scores_dev, count, (hipStream_t)stream, stats_dev,
                               (double *)solver->dev_median_partials.ptr);
}

int rocco_hip_score_median(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K,
                           size_t n, size_t row_stride, double *scores_dev, void *stream)
{
    if (solver == nullptr || matrix_dev == nullptr || scores_dev == nullptr || K == 0 ||
        row_stride < n || (dtype != 0 && dtype != 1)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_median(matrix_dev, dtype, K, n, row_stride, scores_dev, (hipStream_t)stream);
}

int rocco_hip_score_order_statistic(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K, size_t n,
                                    size_t row_stride, int rank, double *scores_dev, void *stream)
{
    if (solver == nullptr || (n > 0 && (matrix_dev == nullptr || scores_dev == nullptr)) || K == 0 || row_stride < n ||
        (dtype != 0 && dtype != 1) || rank < 0 || (size_t)rank >= K) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_order_statistic(matrix_dev, dtype, K, n, row_stride, rank, scores_dev, (hipStream_t)stream);
}

int rocco_hip_score_trimmed_mean(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K, size_t n,
                                 size_t row_stride, int rank_lo, int rank_hi, double *scores_dev, void *stream)
{
    if (solver == nullptr || matrix_dev == nullptr || scores_dev == nullptr || K == 0 || row_stride < n ||
        (dtype != 0 && dtype != 1) || rank_lo < 0 || rank_hi < rank_lo || rank_hi >= (int)K) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_trimmed_mean(matrix_dev, dtype, K, n, row_stride, rank_lo, rank_hi, scores_dev, (hipStream_t)stream);
}

int rocco_hip_power_f64(rocco_hip_solver *solver, const double *x_dev, double power, double *out_dev, size_t n, void *stream)
{
    if (solver == nullptr || (n > 0 && (x_dev == nullptr || out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_power(x_dev, power, out_dev, n, (hipStream_t)stream);
}

int rocco_hip_score_mean(rocco_hip_solver *solver, const void *matrix_dev, int dtype, size_t K, size_t n,
                         size_t row_stride, double *scores_dev, void *stream)
{
    if (solver == nullptr || (n > 0 && (matrix_dev == nullptr || scores_dev == nullptr)) || K == 0 || row_stride < n ||
        (dtype != 0 && dtype != 1)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_column_mean(matrix_dev, dtype, K, n, row_stride, scores_dev, (hipStream_t)stream);
}

int rocco_hip_solve_penalized_chain_f64(rocco_hip_solver *solver, const double *scores_dev,
                                        const double *switch_costs_dev, double gamma, size_t n,
                                        double selection_penalty, uint8_t *solution_dev,
                                        double *value_out, long long *count_out, int *path_out,
                                        void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || n == 0 || n >= ((size_t)1 << 31)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return solve_fixed_penalty(solver, scores_dev, switch_costs_dev, gamma, n, selection_penalty,
                               solution_dev, value_out, count_out, path_out, (hipStream_t)stream);
}

int rocco_hip_solve_budget_batch_f64(rocco_hip_solver *solver, size_t n_tasks,
                                     const rocco_hip_budget_task *tasks,
                                     rocco_hip_budget_result *results, void *stream)
{
    if (solver == nullptr || (n_tasks > 0 && (tasks == nullptr || results == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    for (size_t t = 0; t < n_tasks; ++t) {
        if (tasks[t].scores_dev == nullptr || tasks[t].n == 0 || tasks[t].n >= ((size_t)1 << 31) ||
            tasks[t].solution_dev == nullptr || tasks[t].max_iter < 0) {
            return ROCCO_HIP_EINVAL;
        }
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return solve_budget_batch(solver, n_tasks, tasks, results, (hipStream_t)stream);
}

int rocco_hip_solve_budget_batch_stats_f64(rocco_hip_solver *solver, size_t n_tasks, const rocco_hip_budget_task *tasks,
                                           const double *score_stats_host, rocco_hip_budget_result *results, void *stream)
{
    if (solver == nullptr || (n_tasks > 0 && (tasks == nullptr || results == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    for (size_t t = 0; t < n_tasks; ++t) {
        if (tasks[t].scores_dev == nullptr || tasks[t].n == 0 || tasks[t].n >= ((size_t)1 << 31) ||
            tasks[t].solution_dev == nullptr || tasks[t].max_iter < 0) {
            return ROCCO_HIP_EINVAL;
        }
        if (score_stats_host != nullptr && !(score_stats_host[3 * t] <= score_stats_host[3 * t + 1])) {
            return ROCCO_HIP_EINVAL;  // min <= max (also refuses NaN)
        }
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return solve_budget_batch(solver, n_tasks, tasks, results, (hipStream_t)stream, score_stats_host);
}

int rocco_hip_delta_model_lean_f64(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                                   const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas, long long *counts_out,
                                   long long *open_out, void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || emap_dev == nullptr || n < 2 || n >= ((size_t)1 << 31) ||
        (n_lambdas > 0 && (lambdas == nullptr || counts_out == nullptr || open_out == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return delta_model_lean(solver, scores_dev, gamma, n, emap_dev, lambdas, n_lambdas, counts_out, open_out, (hipStream_t)stream);
}

int rocco_hip_delta_probe_f64(rocco_hip_solver *solver, const double *scores_dev,
                              const double *switch_costs_dev, double gamma, size_t n,
                              const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                              rocco_hip_probe_stats *stats_out, void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || n == 0 || n >= ((size_t)1 << 31) ||
        (n_lambdas > 0 && (lambdas == nullptr || stats_out == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return delta_probe(solver, scores_dev, switch_costs_dev, gamma, n, emap_dev, lambdas, n_lambdas,
                       stats_out, (hipStream_t)stream);
}

int rocco_hip_delta_bound_rounds_f64(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                                     const double *lambdas, const int *round_sizes, int n_rounds,
                                     double *lambdas_used_out, long long *counts_out, long long *level_len_out,
                                     void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || n < 2 || n >= ((size_t)1 << 31) || n_rounds < 0 ||
        (n_rounds > 0 && (lambdas == nullptr || round_sizes == nullptr || lambdas_used_out == nullptr ||
                          counts_out == nullptr || level_len_out == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return delta_bound_rounds(solver, scores_dev, gamma, n, lambdas, round_sizes, n_rounds, lambdas_used_out, counts_out,
                              level_len_out, (hipStream_t)stream);
}

int rocco_hip_delta_spine_f64(rocco_hip_solver *solver, const double *scores_dev,
                              const double *switch_costs_dev, double gamma, size_t n,
                              const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                              int solution_index, uint8_t *solution_dev, long long *counts_out,
                              void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || n == 0 || n >= ((size_t)1 << 31) || emap_dev == nullptr ||
        lambdas == nullptr || counts_out == nullptr || n_lambdas == 0 || n_lambdas > 64 ||
        solution_index >= (int)n_lambdas || (solution_index >= 0 && solution_dev == nullptr)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return delta_spine(solver, scores_dev, switch_costs_dev, gamma, n, emap_dev, lambdas, n_lambdas,
                       solution_index, solution_dev, counts_out, (hipStream_t)stream);
}

int rocco_hip_delta_build_map_f64(rocco_hip_solver *solver, const double *scores_dev,
                                  const double *switch_costs_dev, double gamma, size_t n,
                                  double lambda_ref, double margin, uint8_t *emap_dev, void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || n == 0 || n >= ((size_t)1 << 31) || emap_dev == nullptr ||
        !(margin >= 0.0)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return delta_build_map(solver, scores_dev, switch_costs_dev, gamma, n, lambda_ref, margin, emap_dev,
                           (hipStream_t)stream);
}

int rocco_hip_delta_build_map_lean_f64(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                                       double lambda_ref, double margin, uint8_t *emap_dev, void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || n < 2 || n >= ((size_t)1 << 31) || emap_dev == nullptr || !(margin >= 0.0)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return delta_build_map_lean(solver, scores_dev, gamma, n, lambda_ref, margin, emap_dev, (hipStream_t)stream);
}

int rocco_hip_delta_window_f64(rocco_hip_solver *solver, const double *scores_dev,
                               const double *switch_costs_dev, double gamma, size_t n,
                               const uint8_t *emap_dev, double lambda_lo, double lambda_hi,
                               uint8_t *solution_dev, rocco_hip_window_stats *stats_out, void *stream)
{
    if (solver == nullptr || scores_dev == nullptr || n == 0 || n >= ((size_t)1 << 31) ||
        solution_dev == nullptr || stats_out == nullptr || !(lambda_lo <= lambda_hi)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return delta_window(solver, scores_dev, switch_costs_dev, gamma, n, emap_dev, lambda_lo, lambda_hi,
                        solution_dev, stats_out, (hipStream_t)stream);
}

int rocco_hip_objective_value_f64(rocco_hip_solver *solver, const uint8_t *solution_dev,
                                  const double *scores_dev, const double *switch_costs_dev,
                                  double gamma, size_t n, double *objective_out, void *stream)
{
    if (solver == nullptr || objective_out == nullptr || (n > 0 && (solution_dev == nullptr || scores_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc = solver->dev_misc.reserve(objective_scratch_bytes(n));
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    rc = solver->host_back.reserve(64);
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    double *back = (double *)solver->host_back.ptr;
    rc = launch_objective(solution_dev, scores_dev, switch_costs_dev, gamma, n, solver->dev_misc.ptr,
                          back, (hipStream_t)stream);
    if (rc == ROCCO_HIP_OK) {
        *objective_out = *back;
    }
    return rc;
}

int rocco_hip_peak_signal_stat_f64(rocco_hip_solver *solver, const double *counts_dev, const double *lengths_dev, size_t n_peaks,
                                   size_t n_samples, double row_scale, double pc, double percentile, double *stat_out_dev, void *stream)
{
    if (solver == nullptr || n_samples == 0 || n_samples > 4096 || !(percentile >= 0.0 && percentile <= 100.0) ||
        (n_peaks > 0 && (counts_dev == nullptr || lengths_dev == nullptr || stat_out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    const int rc = solver->dev_misc.reserve(n_peaks * n_samples * sizeof(double) + 256);
    if (rc != ROCCO_HIP_OK) return rc;
    return launch_peak_signal(counts_dev, lengths_dev, n_peaks, n_samples, row_scale, pc, percentile, stat_out_dev, solver->dev_misc.ptr,
                              (hipStream_t)stream);
}

int rocco_hip_ecdf_survival_f64(rocco_hip_solver *solver, const double *stat_dev, const int *bin_dev, const double *null_values_dev,
                                const long long *null_offsets_dev, size_t n_peaks, double *pvals_out_dev, void *stream)
{
    if (solver == nullptr || (n_peaks > 0 && (stat_dev == nullptr || bin_dev == nullptr || null_values_dev == nullptr ||
                                              null_offsets_dev == nullptr || pvals_out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_ecdf_survival(stat_dev, bin_dev, null_values_dev, null_offsets_dev, n_peaks, pvals_out_dev, (hipStream_t)stream);
}

int rocco_hip_bh_adjust_f64(rocco_hip_solver *solver, const double *pvals_dev, size_t m, double *qvals_out_dev, void *stream)
{
    if (solver == nullptr || m >= ((size_t)1 << 31) || (m > 0 && (pvals_dev == nullptr || qvals_out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    const int rc = solver->dev_misc.reserve(bh_scratch_bytes(m));
    if (rc != ROCCO_HIP_OK) return rc;
    return launch_bh_adjust(pvals_dev, m, qvals_out_dev, solver->dev_misc.ptr, (hipStream_t)stream);
}

int rocco_hip_sort_f64(rocco_hip_solver *solver, const double *x_dev, size_t n, double *sorted_out_dev, void *stream)
{
    if (solver == nullptr || (n > 0 && (x_dev == nullptr || sorted_out_dev == nullptr)) || n >= ((size_t)1 << 31)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    const int rc = solver->dev_misc.reserve(sort_f64_scratch_bytes(n));
    if (rc != ROCCO_HIP_OK) return rc;
    return launch_sort_f64(x_dev, n, sorted_out_dev, solver->dev_misc.ptr, (hipStream_t)stream);
}

int rocco_hip_sorted_probe_f64(rocco_hip_solver *solver, const double *sorted_dev, size_t n, const long long *ranks,
                               size_t n_ranks, double *values_out, double shift, const double *thresholds,
                               size_t n_thresholds, long long *counts_le_out, long long *counts_lt_out, void *stream)
{
    if (solver == nullptr || sorted_dev == nullptr || n == 0 || n_ranks > 8 || n_thresholds > 8 ||
        (n_ranks > 0 && (ranks == nullptr || values_out == nullptr)) ||
        (n_thresholds > 0 && (thresholds == nullptr || counts_le_out == nullptr || counts_lt_out == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    SortedProbe p;
    p.n_ranks = (int)n_ranks;
    p.n_thresholds = (int)n_thresholds;
    p.shift = shift;
    for (size_t i = 0; i < 8; ++i) {
        p.ranks[i] = (i < n_ranks) ? ranks[i] : 0;
        p.thresholds[i] = (i < n_thresholds) ? thresholds[i] : 0.0;
    }
    int rc = solver->dev_results.reserve(8 * sizeof(double) + 16 * sizeof(long long) + 64);
    if (rc != ROCCO_HIP_OK) return rc;
    rc = solver->host_back.reserve(8 * sizeof(double) + 16 * sizeof(long long) + 64);
    if (rc != ROCCO_HIP_OK) return rc;
    double *d_vals = (double *)solver->dev_results.ptr;
    long long *d_cnts = (long long *)((char *)solver->dev_results.ptr + 64);
    rc = launch_sorted_probe(sorted_dev, n, p, d_vals, d_cnts, (hipStream_t)stream);
    if (rc != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipMemcpyAsync(solver->host_back.ptr, solver->dev_results.ptr, 64 + 16 * sizeof(long long), hipMemcpyDeviceToHost,
                                 (hipStream_t)stream));
    ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    const double *hv = (const double *)solver->host_back.ptr;
    const long long *hc = (const long long *)((const char *)solver->host_back.ptr + 64);
    for (size_t i = 0; i < n_ranks; ++i) values_out[i] = hv[i];
    for (size_t i = 0; i < n_thresholds; ++i) {
        counts_le_out[i] = hc[2 * i];
        counts_lt_out[i] = hc[2 * i + 1];
    }
    return ROCCO_HIP_OK;
}

int rocco_hip_autocovariance_sums_f64(rocco_hip_solver *solver, const double *x_dev, size_t n, double mean, int max_lag,
                                      double *sums_out, void *stream)
{
    if (solver == nullptr || x_dev == nullptr || n == 0 || sums_out == nullptr || max_lag < 0 || (size_t)max_lag >= n) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc = solver->dev_misc.reserve(autocov_scratch_bytes(n, max_lag));
    if (rc != ROCCO_HIP_OK) return rc;
    rc = solver->dev_results.reserve(((size_t)max_lag + 1024) * sizeof(double));
    if (rc != ROCCO_HIP_OK) return rc;
    rc = solver->host_back.reserve(((size_t)max_lag + 1024) * sizeof(double));
    if (rc != ROCCO_HIP_OK) return rc;
    rc = launch_autocov(x_dev, n, mean, max_lag, (double *)solver->dev_results.ptr, solver->dev_misc.ptr, (hipStream_t)stream);
    if (rc != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipMemcpyAsync(solver->host_back.ptr, solver->dev_results.ptr, (size_t)(max_lag + 1) * sizeof(double),
                                 hipMemcpyDeviceToHost, (hipStream_t)stream));
    ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::memcpy(sums_out, solver->host_back.ptr, (size_t)(max_lag + 1) * sizeof(double));
    return ROCCO_HIP_OK;
}

int rocco_hip_negative_part_f64(rocco_hip_solver *solver, const double *scores_dev, double *out_dev, size_t n, void *stream)
{
    if (solver == nullptr || (n > 0 && (scores_dev == nullptr || out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_negative_part(scores_dev, out_dev, n, (hipStream_t)stream);
}

int rocco_hip_soft_counts_f64(rocco_hip_solver *solver, const double *scores_dev, double center, double scale, double *out_dev,
                              size_t n, void *stream)
{
    if (solver == nullptr || (n > 0 && (scores_dev == nullptr || out_dev == nullptr)) || !(scale > 0.0)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_soft_counts(scores_dev, center, scale, out_dev, n, (hipStream_t)stream);
}

int rocco_hip_decode_runs_batch(rocco_hip_solver *solver, size_t count, const uint8_t *const *solutions_dev, const size_t *n,
                                int64_t *const *run_begin_dev, int64_t *const *run_end_dev, const size_t *capacities,
                                size_t *n_runs_out, void *stream)
{
    if (solver == nullptr || (count > 0 && (solutions_dev == nullptr || n == nullptr || run_begin_dev == nullptr ||
                                            run_end_dev == nullptr || capacities == nullptr || n_runs_out == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    size_t at = 0;
    while (at < count) {
        DecodeBatch batch;
        batch.n_tasks = 0;
        batch.pad = 0;
        long long tiles = 0;
        size_t owner[kDecodeBatchMax];
        while (at < count && batch.n_tasks < kDecodeBatchMax) {
            n_runs_out[at] = 0;
            if (n[at] > 1) {
                if (solutions_dev[at] == nullptr || (capacities[at] > 0 && (run_begin_dev[at] == nullptr || run_end_dev[at] == nullptr))) {
                    return ROCCO_HIP_EINVAL;
                }
                DecodeTask &t = batch.tasks[batch.n_tasks];
                t.solution = solutions_dev[at];
                t.n = (long long)n[at];
                t.tile_begin = tiles;
                t.run_begin = run_begin_dev[at];
                t.run_end = run_end_dev[at];
                t.capacity = (unsigned long long)capacities[at];
                owner[batch.n_tasks++] = at;
                tiles += decode_tiles(n[at]);
            }
            ++at;
        }
        if (batch.n_tasks == 0) {
            continue;
        }
        int rc = solver->dev_misc.reserve(decode_batch_scratch_bytes(tiles, batch.n_tasks));
        if (rc != ROCCO_HIP_OK) return rc;
        rc = solver->host_back.reserve((size_t)batch.n_tasks * 16 + 64);
        if (rc != ROCCO_HIP_OK) return rc;
        unsigned long long *back = (unsigned long long *)solver->host_back.ptr;
        rc = launch_decode_runs_batch(batch, tiles, solver->dev_misc.ptr, back, (hipStream_t)stream);
        if (rc != ROCCO_HIP_OK) return rc;
        for (int i = 0; i < batch.n_tasks; ++i) {
            n_runs_out[owner[i]] = (size_t)back[2 * i];
        }
    }
    return ROCCO_HIP_OK;
}

int rocco_hip_decode_runs_table(rocco_hip_solver *solver, size_t count, const uint8_t *const *solutions_dev, const size_t *n,
                                const long long *units, int64_t *table_dev, size_t capacity_rows, size_t eager_rows,
                                size_t *row_offsets_out, const int64_t **table_host_out, void *stream)
{
    if (solver == nullptr || row_offsets_out == nullptr || count > (size_t)kDecodeBatchMax ||
        (count > 0 && (solutions_dev == nullptr || n == nullptr || units == nullptr)) || (capacity_rows > 0 && table_dev == nullptr)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    if (table_host_out != nullptr) {
        *table_host_out = nullptr;
    }
    DecodeBatch batch;
    batch.n_tasks = 0;
    batch.pad = 0;
    long long tiles = 0;
    size_t owner[kDecodeBatchMax];
    for (size_t at = 0; at < count; ++at) {
        row_offsets_out[at] = 0;
        if (n[at] > 1) {
            if (solutions_dev[at] == nullptr) {
                return ROCCO_HIP_EINVAL;
            }
            DecodeTask &t = batch.tasks[batch.n_tasks];
            t.solution = solutions_dev[at];
            t.n = (long long)n[at];
            t.tile_begin = tiles;
            t.run_begin = table_dev;
            t.run_end = nullptr;
            t.capacity = (unsigned long long)capacity_rows;
            t.unit = units[at];
            owner[batch.n_tasks++] = at;
            tiles += decode_tiles(n[at]);
        }
    }
    row_offsets_out[count] = 0;
    if (batch.n_tasks == 0) {
        return ROCCO_HIP_OK;
    }
    int rc = solver->dev_misc.reserve(decode_batch_scratch_bytes(tiles, batch.n_tasks));
    if (rc != ROCCO_HIP_OK) return rc;
    rc = solver->host_back.reserve((size_t)batch.n_tasks * 16 + 64);
    if (rc != ROCCO_HIP_OK) return rc;
    const bool want_host = (table_host_out != nullptr);
    (void)eager_rows;  // (was: how many rows to copy before their number is known; the rows that exist are copied by a kernel now)
    const size_t eager = want_host ? capacity_rows : 0;
    if (want_host) {
        rc = solver->host_table.reserve(std::max<size_t>(capacity_rows, 1) * 3 * sizeof(int64_t));
        if (rc != ROCCO_HIP_OK) return rc;
    }
    unsigned long long *back = (unsigned long long *)solver->host_back.ptr;
    int64_t *host_rows = want_host ? (int64_t *)solver->host_table.ptr : nullptr;
    rc = launch_decode_runs_table(batch, tiles, solver->dev_misc.ptr, table_dev, back, host_rows, eager, (hipStream_t)stream);
    if (rc != ROCCO_HIP_OK) return rc;
    // per-solution row ranges: row_offsets_out[i] .. row_offsets_out[i + 1]
    size_t total = 0;
    int next = 0;
    for (size_t at = 0; at < count; ++at) {
        row_offsets_out[at] = total;
        if (next < batch.n_tasks && owner[next] == at) {
            total += (size_t)back[2 * next];
            ++next;
        }
    }
    row_offsets_out[count] = total;
    if (want_host && total <= capacity_rows) {
        if (total > eager) {  // the eager copy did not cover the table: the rest, and one more synchronisation
            ROCCO_HIP_TRY(hipMemcpyAsync(host_rows + 3 * eager, table_dev + 3 * eager, (total - eager) * 3 * sizeof(int64_t),
                                         hipMemcpyDeviceToHost, (hipStream_t)stream));
            ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        }
        *table_host_out = host_rows;
    }
    return ROCCO_HIP_OK;
}

int rocco_hip_decode_runs(rocco_hip_solver *solver, const uint8_t *solution_dev, size_t n,
                          int64_t *run_begin_dev, int64_t *run_end_dev, size_t capacity,
                          size_t *n_runs_out, void *stream)
{
    if (solver == nullptr || n_runs_out == nullptr || (n > 0 && solution_dev == nullptr) ||
        (capacity > 0 && (run_begin_dev == nullptr || run_end_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc = solver->dev_misc.reserve(decode_scratch_bytes(n));
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    rc = solver->host_back.reserve(64);
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    unsigned long long *back = (unsigned long long *)solver->host_back.ptr;
    rc = launch_decode_runs(solution_dev, n, run_begin_dev, run_end_dev, capacity, solver->dev_misc.ptr,
                            back, (hipStream_t)stream);
    if (rc == ROCCO_HIP_OK) {
        *n_runs_out = (size_t)*back;
    }
    return rc;
}

namespace {

// The LDL^T factor depends on the penalty and (at its last two entries only) on the length: the solver keeps the one of
// the longest row seen with this penalty.  The recurrence is sequential and data-independent: two host threads walk it
// (whittaker_host.cpp, ~10 ns per step against ~190 ns for a GPU lane: 0.1 s instead of 1.05 s for chr1 at 50 bp) and the
// table is copied up.  A longer row with the same penalty extends the factor instead of starting over; a new penalty
// starts a new one.
std::mutex g_factor_mutex;
// per device; on the heap and never destroyed: at process exit the HIP runtime may be gone before static destructors run
// per (device, penalty): a batch that mixes short contigs (a window below 101 loci: another penalty) with chromosomes keeps both
std::map<std::pair<int, double>, std::shared_ptr<rocco::SharedFactor>> &g_factors = *new std::map<std::pair<int, double>, std::shared_ptr<rocco::SharedFactor>>();

// the device's factor covers `cols` loci at this penalty when this returns; solver->factor holds it for the call
int ensure_whittaker_factor(rocco_hip_solver *solver, size_t cols, double penalty_lambda, hipStream_t stream)
{
    if (cols < 25) {
        return ROCCO_HIP_OK;
    }
    std::lock_guard<std::mutex> lock(g_factor_mutex);  // (a second thread waits for the first one's factor rather than building its own)
    std::shared_ptr<rocco::SharedFactor> &current = g_factors[std::make_pair(solver->device, penalty_lambda)];
    if (current && current->cap >= cols && current->lambda == penalty_lambda) {
        solver->factor = current;
        return ROCCO_HIP_OK;
    }
    auto grown = std::make_shared<rocco::SharedFactor>();
    grown->device = solver->device;
    int rc;
    if ((rc = grown->buf.reserve(6 * cols * sizeof(double))) != ROCCO_HIP_OK) {
        return rc;
    }
    const bool extend = current && current->cap > 0 && current->lambda == penalty_lambda;
    // (ROCCO_HIP_FACTOR_ON_DEVICE=1: the recurrence walked by one GPU lane per parity instead of two host threads)
    const char *on_device = std::getenv("ROCCO_HIP_FACTOR_ON_DEVICE");
    rc = (on_device != nullptr && std::atoi(on_device) != 0)
             ? launch_whittaker_factor(cols, penalty_lambda, (double *)grown->buf.ptr, stream,
                                       extend ? (const double *)current->buf.ptr : nullptr, extend ? current->cap : 0)
             : build_whittaker_factor_on_host(cols, penalty_lambda, (double *)grown->buf.ptr, stream,
                                              extend ? (const double *)current->buf.ptr : nullptr, extend ? current->cap : 0);
    if (rc == ROCCO_HIP_OK && hipStreamSynchronize(stream) != hipSuccess) {
        rc = ROCCO_HIP_EHIP;
    }
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    grown->cap = cols;
    grown->lambda = penalty_lambda;
    current = grown;
    solver->factor = grown;
    return ROCCO_HIP_OK;
}

}  // namespace

namespace {

int whittaker_batch(rocco_hip_solver *solver, size_t count, const double *const *matrices_dev, const double *const *offsets_dev,
                    const size_t *rows, const size_t *cols, double penalty_lambda, double *const *baselines_dev, int residual,
                    void *stream, void *scratch_dev = nullptr, size_t scratch_bytes = 0)
{
    if (solver == nullptr || (count > 0 && (matrices_dev == nullptr || rows == nullptr || cols == nullptr || baselines_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    size_t longest = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] * cols[i] > 0 && (matrices_dev[i] == nullptr || baselines_dev[i] == nullptr || (residual != 0 && baselines_dev[i] == matrices_dev[i]))) {
            return ROCCO_HIP_EINVAL;
        }
        if (rows[i] > 0) {
            longest = std::max(longest, cols[i]);
        }
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if (scratch_dev != nullptr) {  // (the caller's block: what the sweeps need between them, see rocco_hip_whittaker_batch_scratch_bytes)
        if (scratch_bytes < whittaker_batch_scratch_bytes(rows, cols, count) || ((uintptr_t)scratch_dev & 255u) != 0) {
            set_last_error("the sweeps' scratch block is too small or not aligned to 256 bytes");
            return ROCCO_HIP_EINVAL;
        }
    } else if ((rc = solver->dev_misc.reserve(whittaker_batch_scratch_bytes(rows, cols, count))) != ROCCO_HIP_OK) {
        return rc;
    }
    if ((rc = solver->host_stage.reserve(whittaker_batch_stage_bytes(rows, cols, count))) != ROCCO_HIP_OK) return rc;
    if ((rc = ensure_whittaker_factor(solver, longest, penalty_lambda, (hipStream_t)stream)) != ROCCO_HIP_OK) return rc;
    const rocco::SharedFactor *factor = (longest >= 25) ? solver->factor.get() : nullptr;
    rc = launch_crossfit_whittaker_batch(matrices_dev, rows, cols, count, penalty_lambda,
                                         factor != nullptr ? (const double *)factor->buf.ptr : nullptr,
                                         factor != nullptr ? factor->cap : 0, baselines_dev,
                                         scratch_dev != nullptr ? scratch_dev : solver->dev_misc.ptr,
                                         solver->host_stage.ptr, (hipStream_t)stream, offsets_dev, residual);
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // the scratch buffers are the solver's
    if (whittaker_collect_repairs(solver->host_stage.ptr) != 0 && residual != 0) {
        set_last_error("Local baseline fit produced non-finite values");
        return ROCCO_HIP_EINVAL;
    }
    return ROCCO_HIP_OK;
}

}  // namespace

int rocco_hip_crossfit_whittaker_baseline_batch_f64(rocco_hip_solver *solver, size_t count, const double *const *matrices_dev,
                                                    const size_t *rows, const size_t *cols, double penalty_lambda,
                                                    double *const *baselines_dev, void *stream)
{
    return whittaker_batch(solver, count, matrices_dev, nullptr, rows, cols, penalty_lambda, baselines_dev, 0, stream);
}

int rocco_hip_crossfit_whittaker_residual_batch_f64(rocco_hip_solver *solver, size_t count, const double *const *matrices_dev,
                                                    const double *const *row_offsets_dev, const size_t *rows, const size_t *cols,
                                                    double penalty_lambda, double *const *centered_out_dev, void *stream)
{
    return whittaker_batch(solver, count, matrices_dev, row_offsets_dev, rows, cols, penalty_lambda, centered_out_dev, 1, stream);
}

size_t rocco_hip_whittaker_batch_scratch_bytes(size_t count, const size_t *rows, const size_t *cols)
{
    return (rows == nullptr || cols == nullptr) ? 0 : whittaker_batch_scratch_bytes(rows, cols, count);
}

int rocco_hip_crossfit_whittaker_residual_batch_scratch_f64(rocco_hip_solver *solver, size_t count, const double *const *matrices_dev,
                                                            const double *const *row_offsets_dev, const size_t *rows,
                                                            const size_t *cols, double penalty_lambda, double *const *centered_out_dev,
                                                            void *scratch_dev, size_t scratch_bytes, void *stream)
{
    if (scratch_dev == nullptr) {
        return ROCCO_HIP_EINVAL;
    }
    return whittaker_batch(solver, count, matrices_dev, row_offsets_dev, rows, cols, penalty_lambda, centered_out_dev, 1, stream,
                           scratch_dev, scratch_bytes);
}

int rocco_hip_crossfit_whittaker_baseline_matrix_f64(rocco_hip_solver *solver, const double *matrix_dev,
                                                     size_t rows, size_t cols, double penalty_lambda,
                                                     double *baseline_out_dev, void *stream)
{
    if (solver == nullptr || ((matrix_dev == nullptr || baseline_out_dev == nullptr) && rows * cols > 0)) {
        return ROCCO_HIP_EINVAL;
    }
    return rocco_hip_crossfit_whittaker_baseline_batch_f64(solver, 1, &matrix_dev, &rows, &cols, penalty_lambda, &baseline_out_dev,
                                                           stream);
}

int rocco_hip_wls_rolling_variances_batch_f64(rocco_hip_solver *solver, size_t count, const double *const *centered_dev,
                                              const size_t *K, const size_t *n, int spatial_window, double *const *variances_dev,
                                              void *stream)
{
    if (solver == nullptr || (count > 0 && (centered_dev == nullptr || K == nullptr || n == nullptr || variances_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    size_t rows = 0;
    for (size_t i = 0; i < count; ++i) {
        const int window = wls_spatial_window(n[i], spatial_window);
        if (window > wls_max_window()) {
            set_last_error("rocco_hip_wls_rolling_variances_batch_f64: spatial windows above 63 loci go through the single-matrix call");
            return ROCCO_HIP_EINVAL;
        }
        if (window > 0 && n[i] >= 4) {
            if (centered_dev[i] == nullptr || variances_dev[i] == nullptr) {
                return ROCCO_HIP_EINVAL;
            }
            rows += K[i];
        }
    }
    if (rows == 0) {
        return ROCCO_HIP_OK;
    }
    // rows per workgroup (one task record each); a solver may ask for at least so many (rocco_hip_solver_set "rolling_group_min")
    wls_set_rolling_group_min(solver->rolling_group_min);
    const size_t group = (size_t)wls_rolling_group_rows(rows);
    int rc;
    if ((rc = solver->dev_tasks.reserve(rows * sizeof(WlsRollingTask))) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_stage.reserve(rows * sizeof(WlsRollingTask))) != ROCCO_HIP_OK) return rc;
    WlsRollingTask *host = (WlsRollingTask *)solver->host_stage.ptr;
    size_t t = 0;
    for (size_t i = 0; i < count; ++i) {
        const int window = wls_spatial_window(n[i], spatial_window);
        if (window <= 0 || n[i] < 4) {
            continue;
        }
        const size_t stride = n[i] - (size_t)window + 1;
        for (size_t k = 0; k < K[i]; k += group, ++t) {
            host[t].row = centered_dev[i] + k * n[i];
            host[t].n = (long long)n[i];
            host[t].window = window;
            host[t].rows = (int)std::min(group, K[i] - k);
            host[t].out = variances_dev[i] + k * stride;
        }
    }
    ROCCO_HIP_TRY(hipMemcpyAsync(solver->dev_tasks.ptr, host, t * sizeof(WlsRollingTask), hipMemcpyHostToDevice, (hipStream_t)stream));
    if ((rc = launch_wls_rolling_batch((const WlsRollingTask *)solver->dev_tasks.ptr, t, (int)group, (hipStream_t)stream)) != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));  // the task table is the solver's
    return ROCCO_HIP_OK;
}

long long rocco_hip_buffer_growths(void) { return g_buffer_growths.load(std::memory_order_relaxed); }

long long rocco_hip_whittaker_seam_repairs(void) { return whittaker_seam_repairs(); }

void rocco_hip_model_chain_counters(long long out[4])
{
    if (out != nullptr) {
        rocco::model_chain_counters(out);
    }
}

void rocco_hip_model_chain_written_counters(long long out[2])
{
    if (out != nullptr) {
        rocco::model_chain_written_counters(out);
    }
}

long long rocco_hip_solver_device_bytes(const rocco_hip_solver *solver)
{
    if (solver == nullptr) {
        return 0;
    }
    const rocco::DeviceBuffer *all[] = {&solver->dev_tasks, &solver->dev_params, &solver->dev_results, &solver->dev_bits, &solver->dev_misc,
                                        &solver->dev_median_partials, &solver->dev_solution, &solver->dev_maps, &solver->dev_frozen,
                                        &solver->dev_lean_pool, &solver->dev_lean_round, &solver->dev_lean_look, &solver->dev_lean_progress, &solver->dev_lean_desc,
                                        &solver->dev_lean_wcap, &solver->dev_chain};
    long long total = 0;
    for (const rocco::DeviceBuffer *b : all) {
        total += (long long)b->bytes;
    }
    return total;
}

int rocco_hip_count_path_reserve(rocco_hip_solver *solver, size_t count, const size_t *rows, const size_t *cols, double penalty_lambda,
                                 void *stream)
{
    return rocco_hip_count_path_reserve_ex(solver, count, rows, cols, penalty_lambda, 0, stream);
}

int rocco_hip_count_path_reserve_ex(rocco_hip_solver *solver, size_t count, const size_t *rows, const size_t *cols, double penalty_lambda,
                                    int sweeps_scratch_is_the_callers, void *stream)
{
    if (solver == nullptr || (count > 0 && (rows == nullptr || cols == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    size_t misc = (sweeps_scratch_is_the_callers != 0) ? 0 : whittaker_batch_scratch_bytes(rows, cols, count);
    size_t stage_single = 0, total_rows = 0, most_rows = 0, longest = 0;
    for (size_t i = 0; i < count; ++i) {
        if (rows[i] == 0 || cols[i] == 0) {
            continue;
        }
        misc = std::max(misc, wls_scratch_bytes(rows[i], cols[i], 31, true));
        misc = std::max(misc, wls_scratch_bytes(rows[i], cols[i], 31, false));
        misc = std::max(misc, log_scale_scratch_bytes(rows[i], cols[i]));
        if (sweeps_scratch_is_the_callers == 0) {
            misc = std::max(misc, whittaker_batch_scratch_bytes(&rows[i], &cols[i], 1));
        }
        stage_single = std::max(stage_single, whittaker_batch_stage_bytes(&rows[i], &cols[i], 1));
        total_rows += rows[i];
        most_rows = std::max(most_rows, rows[i]);
        longest = std::max(longest, cols[i]);
    }
    int rc;
    if ((rc = solver->dev_misc.reserve(misc)) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->dev_tasks.reserve(total_rows * sizeof(WlsRollingTask) + 256)) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_stage.reserve(std::max(std::max(whittaker_batch_stage_bytes(rows, cols, count), stage_single), total_rows * sizeof(WlsRollingTask) + 256))) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_back.reserve(256 + most_rows * sizeof(int))) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->dev_results.reserve(256)) != ROCCO_HIP_OK) return rc;
    if (penalty_lambda > 0.0 && (rc = ensure_whittaker_factor(solver, longest, penalty_lambda, (hipStream_t)stream)) != ROCCO_HIP_OK) return rc;
    return ROCCO_HIP_OK;
}

int rocco_hip_score_centered_wls_f64(rocco_hip_solver *solver, const double *centered_dev, size_t K, size_t n,
                                     double lower_bound_z, double prior_df, double min_effect, int use_min_effect,
                                     int spatial_window, double precision_floor_ratio, double *mean_dev,
                                     double *raw_var_dev, double *prior_var_dev, double *mod_var_dev, double *se_dev,
                                     double *scores_dev, double *df_out, int *window_out, void *stream)
{
    return rocco_hip_score_centered_wls_given_variances_f64(solver, centered_dev, K, n, lower_bound_z, prior_df, min_effect,
                                                            use_min_effect, spatial_window, precision_floor_ratio, nullptr, mean_dev,
                                                            raw_var_dev, prior_var_dev, mod_var_dev, se_dev, scores_dev, df_out,
                                                            window_out, stream);
}

int rocco_hip_score_centered_wls_given_variances_f64(rocco_hip_solver *solver, const double *centered_dev, size_t K, size_t n,
                                                     double lower_bound_z, double prior_df, double min_effect, int use_min_effect,
                                                     int spatial_window, double precision_floor_ratio, const double *variances_dev,
                                                     double *mean_dev, double *raw_var_dev, double *prior_var_dev, double *mod_var_dev,
                                                     double *se_dev, double *scores_dev, double *df_out, int *window_out, void *stream)
{
    // argument checks of rocco_score_centered_wls_f64 (wls_backend.c:770-776)
    if (solver == nullptr || centered_dev == nullptr || mean_dev == nullptr || raw_var_dev == nullptr ||
        prior_var_dev == nullptr || mod_var_dev == nullptr || se_dev == nullptr || scores_dev == nullptr || K == 0 ||
        n == 0 || n > (size_t)0x7fffffff) {
        set_last_error("rocco_hip_score_centered_wls_f64: null buffer, empty matrix or more than 2^31-1 loci");
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(wls_scratch_bytes(K, n, spatial_window, variances_dev == nullptr))) != ROCCO_HIP_OK) {
        return rc;
    }
    if ((rc = solver->host_back.reserve(256 + K * sizeof(int))) != ROCCO_HIP_OK) {  // the non-finite flag + one flag per row
        return rc;
    }
    if (variances_dev != nullptr && wls_spatial_window(n, spatial_window) > wls_max_window()) {
        set_last_error("rocco_hip_score_centered_wls_given_variances_f64: windows above 63 loci compute their own variances");
        return ROCCO_HIP_EINVAL;
    }
    wls_set_rolling_group_min(solver->rolling_group_min);
    return launch_score_centered_wls(centered_dev, K, n, lower_bound_z, prior_df, min_effect, use_min_effect,
                                     spatial_window, precision_floor_ratio, mean_dev, raw_var_dev, prior_var_dev,
                                     mod_var_dev, se_dev, scores_dev, solver->dev_misc.ptr, df_out, window_out,
                                     (hipStream_t)stream, (int *)solver->host_back.ptr, variances_dev, &solver->wls_sorted_rows);
}

int rocco_hip_wls_sorted_rows(const rocco_hip_solver *solver) { return solver != nullptr ? solver->wls_sorted_rows : -1; }

int rocco_hip_log2_selfcheck(rocco_hip_solver *solver, int family, unsigned long long seed, unsigned long long first, size_t count,
                             unsigned long long *mismatches_out, void *stream)
{
    if (solver == nullptr || mismatches_out == nullptr || family < 0 || family > 4 || count > ((size_t)1 << 39)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_results.reserve(256)) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_back.reserve(256)) != ROCCO_HIP_OK) return rc;
    unsigned long long *out = (unsigned long long *)solver->dev_results.ptr;
    ROCCO_HIP_TRY(hipMemsetAsync(out, 0, sizeof(unsigned long long), (hipStream_t)stream));
    if ((rc = launch_log2_selfcheck(family, seed, first, count, out, (hipStream_t)stream)) != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipMemcpyAsync(solver->host_back.ptr, out, sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *mismatches_out = *(unsigned long long *)solver->host_back.ptr;
    return ROCCO_HIP_OK;
}

int rocco_hip_log_scale_f64(rocco_hip_solver *solver, const double *values_dev, size_t count, double pseudocount, double *out_dev,
                            void *stream)
{
    if (solver == nullptr || (count > 0 && (values_dev == nullptr || out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc = solver->dev_results.reserve(256);
    if (rc != ROCCO_HIP_OK) return rc;
    int *bad = (int *)solver->dev_results.ptr;
    ROCCO_HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), (hipStream_t)stream));
    if ((rc = launch_log_scale(values_dev, out_dev, count, pseudocount, bad, (hipStream_t)stream)) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_back.reserve(256)) != ROCCO_HIP_OK) return rc;
    int &bad_host = *(int *)solver->host_back.ptr;  // (pinned: see launch_score_centered_wls)
    bad_host = 0;
    ROCCO_HIP_TRY(hipMemcpyAsync(&bad_host, bad, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (bad_host != 0) {
        set_last_error("`chrom_matrix` contains non-finite values");
        return ROCCO_HIP_EINVAL;
    }
    return ROCCO_HIP_OK;
}

int rocco_hip_log_scale_center_rows_f64(rocco_hip_solver *solver, const double *counts_dev, size_t K, size_t n,
                                        double pseudocount, int apply_log2, double *centered_out_dev,
                                        double *row_offsets_out_dev, void *stream)
{
    if (solver == nullptr || counts_dev == nullptr || centered_out_dev == nullptr || K == 0 || n == 0 ||
        n > (size_t)0x7fffffff) {
        set_last_error("rocco_hip_log_scale_center_rows_f64: null buffer, empty matrix or more than 2^31-1 loci");
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(log_scale_scratch_bytes(K, n))) != ROCCO_HIP_OK) {
        return rc;
    }
    if ((rc = solver->host_back.reserve(256)) != ROCCO_HIP_OK) {
        return rc;
    }
    return launch_log_scale_center_rows(counts_dev, K, n, pseudocount, apply_log2, centered_out_dev, row_offsets_out_dev,
                                        solver->dev_misc.ptr, (hipStream_t)stream, (int *)solver->host_back.ptr);
}

int rocco_hip_log_scale_row_offsets_f64(rocco_hip_solver *solver, const double *counts_dev, size_t K, size_t n,
                                        double pseudocount, int apply_log2, double *log_out_dev, double *row_offsets_out_dev,
                                        void *stream)
{
    if (solver == nullptr || counts_dev == nullptr || log_out_dev == nullptr || row_offsets_out_dev == nullptr || K == 0 || n == 0 ||
        n > (size_t)0x7fffffff) {
        set_last_error("rocco_hip_log_scale_row_offsets_f64: null buffer, empty matrix or more than 2^31-1 loci");
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(log_scale_scratch_bytes(K, n))) != ROCCO_HIP_OK) {
        return rc;
    }
    if ((rc = solver->host_back.reserve(256)) != ROCCO_HIP_OK) {
        return rc;
    }
    return launch_log_scale_center_rows(counts_dev, K, n, pseudocount, apply_log2, log_out_dev, row_offsets_out_dev,
                                        solver->dev_misc.ptr, (hipStream_t)stream, (int *)solver->host_back.ptr, 0);
}

int rocco_hip_subtract_finite_f64(rocco_hip_solver *solver, const double *a_dev, const double *b_dev, double *out_dev,
                                  size_t count, void *stream)
{
    if (solver == nullptr || ((a_dev == nullptr || b_dev == nullptr || out_dev == nullptr) && count > 0)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc = solver->dev_results.reserve(256);
    if (rc != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_back.reserve(256)) != ROCCO_HIP_OK) return rc;
    int *bad = (int *)solver->dev_results.ptr;
    ROCCO_HIP_TRY(hipMemsetAsync(bad, 0, sizeof(int), (hipStream_t)stream));
    if ((rc = launch_subtract(a_dev, b_dev, out_dev, count, (hipStream_t)stream, bad)) != ROCCO_HIP_OK) return rc;
    int &bad_host = *(int *)solver->host_back.ptr;
    bad_host = 0;
    ROCCO_HIP_TRY(hipMemcpyAsync(&bad_host, bad, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ROCCO_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (bad_host != 0) {
        set_last_error("Local baseline fit produced non-finite values");
        return ROCCO_HIP_EINVAL;
    }
    return ROCCO_HIP_OK;
}

int rocco_hip_subtract_f64(rocco_hip_solver *solver, const double *a_dev, const double *b_dev, double *out_dev,
                           size_t count, void *stream)
{
    if (solver == nullptr || ((a_dev == nullptr || b_dev == nullptr || out_dev == nullptr) && count > 0)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_subtract(a_dev, b_dev, out_dev, count, (hipStream_t)stream);
}

int rocco_hip_narrowpeak_summit_offsets(rocco_hip_solver *solver, const int64_t *intervals_dev, size_t n_intervals,
                                        const int64_t *centers_dev, const double *effect_mean_dev, size_t n_mean, const int64_t *peak_start_dev,
                                        const int64_t *peak_end_dev, size_t n_peaks, int64_t *offsets_out_dev,
                                        void *stream)
{
    if (solver == nullptr ||
        (n_peaks > 0 && (peak_start_dev == nullptr || peak_end_dev == nullptr || offsets_out_dev == nullptr)) ||
        (n_intervals > 0 && intervals_dev == nullptr) || (n_mean > 0 && effect_mean_dev == nullptr)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_summit_offsets(intervals_dev, n_intervals, centers_dev, effect_mean_dev, n_mean, peak_start_dev, peak_end_dev,
                                 n_peaks, offsets_out_dev, (hipStream_t)stream);
}

int rocco_hip_union_intervals(rocco_hip_solver *solver, const int64_t *values_dev, size_t count,
                              int64_t *unique_out_dev, size_t *n_unique_out, int *fixed_step_out, void *stream)
{
    if (solver == nullptr || n_unique_out == nullptr || count > (size_t)0x7fffffff ||
        (count > 0 && (values_dev == nullptr || unique_out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    *n_unique_out = 0;
    if (fixed_step_out != nullptr) {
        *fixed_step_out = 1;
    }
    if (count == 0) {
        return ROCCO_HIP_OK;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(union_scratch_bytes(count))) != ROCCO_HIP_OK) {
        return rc;
    }
    return launch_union_intervals(values_dev, count, unique_out_dev, n_unique_out, fixed_step_out, solver->dev_misc.ptr,
                                  (hipStream_t)stream);
}

int rocco_hip_scatter_tracks(rocco_hip_solver *solver, const int64_t *common_dev, size_t m,
                             const int64_t *intervals_concat_dev, const double *vals_concat_dev,
                             const size_t *offsets_host, size_t K, int out_dtype, void *matrix_out_dev, void *stream)
{
    if (solver == nullptr || offsets_host == nullptr || (out_dtype != 0 && out_dtype != 1) ||
        (K * m > 0 && (common_dev == nullptr || matrix_out_dev == nullptr))) {
        return ROCCO_HIP_EINVAL;
    }
    for (size_t k = 0; k < K; ++k) {
        if (offsets_host[k + 1] < offsets_host[k] || offsets_host[k + 1] - offsets_host[k] >= (size_t)0xffffffff) {
            return ROCCO_HIP_EINVAL;
        }
    }
    if (K > 0 && offsets_host[K] > 0 && (intervals_concat_dev == nullptr || vals_concat_dev == nullptr)) {
        return ROCCO_HIP_EINVAL;
    }
    if (K * m == 0) {
        return ROCCO_HIP_OK;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(scatter_scratch_bytes(K, m))) != ROCCO_HIP_OK) {
        return rc;
    }
    return launch_scatter_tracks(common_dev, m, intervals_concat_dev, vals_concat_dev, offsets_host, K, out_dtype,
                                 matrix_out_dev, solver->dev_misc.ptr, (hipStream_t)stream);
}

int rocco_hip_numpy_sum_f64(rocco_hip_solver *solver, const double *x_dev, size_t n, double *sum_out, void *stream)
{
    if (solver == nullptr || sum_out == nullptr || (n > 0 && x_dev == nullptr)) {
        return ROCCO_HIP_EINVAL;
    }
    *sum_out = 0.0;
    if (n == 0) {
        return ROCCO_HIP_OK;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(npsum_scratch_bytes(n))) != ROCCO_HIP_OK) {
        return rc;
    }
    const int mode = 0;
    return launch_numpy_sums(x_dev, n, &mode, 1, 0.0, 1.0, 0.0, solver->dev_misc.ptr, sum_out, (hipStream_t)stream);
}

int rocco_hip_budget_null_draw_stats_f64(rocco_hip_solver *solver, const double *scores_dev, size_t n,
                                         double null_center, double null_soft_scale, double null_threshold,
                                         double *stats_out, void *stream)
{
    if (solver == nullptr || stats_out == nullptr || scores_dev == nullptr || n == 0) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(npsum_scratch_bytes(n))) != ROCCO_HIP_OK) {
        return rc;
    }
    const int modes[4] = {1, 2, 3, 4};
    double sums[4];
    if ((rc = launch_numpy_sums(scores_dev, n, modes, 4, null_center, null_soft_scale, null_threshold,
                                solver->dev_misc.ptr, sums, (hipStream_t)stream)) != ROCCO_HIP_OK) {
        return rc;
    }
    for (int i = 0; i < 4; ++i) {
        stats_out[i] = sums[i] / (double)n;  // np.mean: the sum divided by the count
    }
    return ROCCO_HIP_OK;
}

int rocco_hip_multiply_f64(rocco_hip_solver *solver, const double *a_dev, const double *b_dev, double *out_dev,
                           size_t count, void *stream)
{
    if (solver == nullptr || ((a_dev == nullptr || b_dev == nullptr || out_dev == nullptr) && count > 0)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_multiply(a_dev, b_dev, out_dev, count, (hipStream_t)stream);
}

int rocco_hip_subtract_positive_row_f64(rocco_hip_solver *solver, const double *matrix_dev, const double *row_dev,
                                        size_t K, size_t n, double *out_dev, void *stream)
{
    if (solver == nullptr || ((matrix_dev == nullptr || row_dev == nullptr || out_dev == nullptr) && K * n > 0)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_subtract_positive_row(matrix_dev, row_dev, K, n, out_dev, (hipStream_t)stream);
}

int rocco_hip_bigwig_dense_fill_f64(rocco_hip_solver *solver, const int64_t *starts_dev, const int64_t *ends_dev,
                                    const double *vals_dev, size_t count, double const_scale, int round_digits,
                                    double *full_out_dev, size_t capacity, int64_t *first_start_out, int64_t *step_out,
                                    size_t *n_full_out, int *flags_out, void *stream)
{
    if (solver == nullptr || starts_dev == nullptr || ends_dev == nullptr || vals_dev == nullptr || count == 0 ||
        first_start_out == nullptr || step_out == nullptr || n_full_out == nullptr || flags_out == nullptr ||
        round_digits > 22 || round_digits < -22) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    int rc;
    if ((rc = solver->dev_misc.reserve(256)) != ROCCO_HIP_OK) {
        return rc;
    }
    return launch_bigwig_dense_fill(starts_dev, ends_dev, vals_dev, count, const_scale, round_digits, full_out_dev,
                                    capacity, first_start_out, step_out, n_full_out, flags_out, solver->dev_misc.ptr,
                                    (hipStream_t)stream);
}

int rocco_hip_synth_matrix(rocco_hip_solver *solver, void *matrix_dev, int dtype, size_t K, size_t n,
                           size_t row_stride, uint64_t seed, void *stream)
{
    if (solver == nullptr || matrix_dev == nullptr || row_stride < n || (dtype != 0 && dtype != 1)) {
        return ROCCO_HIP_EINVAL;
    }
    ROCCO_HIP_TRY(hipSetDevice(solver->device)); (void)hipGetLastError();  // (no other library's stale error for this call's launch checks)
    return launch_synth(matrix_dev, dtype, K, n, row_stride, seed, (hipStream_t)stream);
}

}  // extern "C"
