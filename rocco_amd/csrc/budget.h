// rocco_amd/csrc/budget.h -- host orchestration of the chain solves (fixed penalty and budgeted).
#pragma once

#include "kernels.h"

namespace rocco {

int solve_fixed_penalty(rocco_hip_solver *solver, const double *scores_dev,
                        const double *switch_costs_dev, double gamma, size_t n, double lambda,
                        uint8_t *solution_dev, double *value_out, long long *count_out, int *path_out,
                        hipStream_t stream);

// `score_stats_host` (optional): [n_tasks][3] = min, max, sum |.| of every task's scores, already known to the caller
// (rocco_hip_score_median_batch_stats): the statistics pass over the scores is skipped for a batch without cost vectors.
int solve_budget_batch(rocco_hip_solver *solver, size_t n_tasks, const rocco_hip_budget_task *tasks,
                       rocco_hip_budget_result *results, hipStream_t stream, const double *score_stats_host = nullptr);

}  // namespace rocco

namespace rocco {

int delta_probe(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                double gamma, size_t n, const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                rocco_hip_probe_stats *stats_out, hipStream_t stream);

int delta_spine(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                double gamma, size_t n, const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                int solution_index, uint8_t *solution_dev, long long *counts_out, hipStream_t stream);

int delta_build_map(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                    double gamma, size_t n, double lambda_ref, double margin, uint8_t *emap_dev,
                    hipStream_t stream);

int delta_bound_rounds(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                       const double *lambdas, const int *round_sizes, int n_rounds, double *lambdas_used_out,
                       long long *counts_out, long long *level_len_out, hipStream_t stream);

int delta_window(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                 double gamma, size_t n, const uint8_t *emap_dev, double lambda_lo, double lambda_hi,
                 uint8_t *solution_dev, rocco_hip_window_stats *stats_out, hipStream_t stream);

// Test entry of the lean rounding-model evaluation (lean_model_kernel) on any array with a binade map: counts and,
// per penalty, whether the count is NOT certified (the product then asks the full kernels).
int delta_model_lean(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n, const uint8_t *emap_dev,
                     const double *lambdas, size_t n_lambdas, long long *counts_out, long long *open_out, hipStream_t stream);

// Test entry of the binade map built by the lean kernels (lean.h: lean_map_kernel): the codes delta_build_map gives for a
// scalar switch cost, bit for bit.
int delta_build_map_lean(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n, double lambda_ref,
                         double margin, uint8_t *emap_dev, hipStream_t stream);

// process-wide diagnostic (include/rocco_hip.h: rocco_hip_model_chain_counters)
void model_chain_counters(long long out[4]);
void model_chain_written_counters(long long out[2]);

}  // namespace rocco
