// rocco_amd/csrc/budget.hip -- host orchestration: replay of the reference's penalty calibration
// (rocco/dp.py:89-164) over batches of chromosomes, with the device doing every chain evaluation.
//
// The reference evaluates the chain 2 + 60 times strictly one after another.  Here several levels
// of its bisection tree are evaluated speculatively in one device pass (the midpoints are formed
// with the reference's own expression (lower + upper) / 2.0, rocco/dp.py:143, so the visited
// penalties are bit-identical), and all chromosomes of a batch share each pass.
#include "budget.h"

#include <algorithm>
#include <cmath>

namespace rocco {

namespace {

struct EvalRequest {
    size_t task = 0;
    std::vector<double> lambdas;  // 1..64 penalties
    bool record = false;          // write the 0/1 solution for lambdas[0]
    std::vector<double> values;
    std::vector<long long> counts;
};

// Evaluate every request with the exact kernel: one wavefront per request, one lane per penalty.
int exact_evaluate(rocco_hip_solver *solver, const rocco_hip_budget_task *tasks,
                   std::vector<EvalRequest> &reqs, hipStream_t stream)
{
    const size_t R = reqs.size();
    if (R == 0) {
        return ROCCO_HIP_OK;
    }
    size_t words_total = 0;
    for (const EvalRequest &r : reqs) {
        if (r.lambdas.empty() || r.lambdas.size() > 64) {
            return ROCCO_HIP_EINVAL;
        }
        if (r.record) {
            words_total += (tasks[r.task].n + 30) / 32 + 1;
        }
    }
    int rc;
    if ((rc = solver->dev_tasks.reserve(R * sizeof(ExactTask))) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->dev_params.reserve(R * 64 * sizeof(double))) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->dev_results.reserve(R * 64 * (sizeof(double) + sizeof(long long)))) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->dev_bits.reserve(words_total * sizeof(unsigned long long) + 8)) != ROCCO_HIP_OK) return rc;
    const size_t stage_bytes = R * sizeof(ExactTask) + R * 64 * sizeof(double);
    if ((rc = solver->host_stage.reserve(stage_bytes)) != ROCCO_HIP_OK) return rc;
    if ((rc = solver->host_back.reserve(R * 64 * (sizeof(double) + sizeof(long long)))) != ROCCO_HIP_OK) return rc;

    ExactTask *h_tasks = (ExactTask *)solver->host_stage.ptr;
    double *h_lams = (double *)((char *)solver->host_stage.ptr + R * sizeof(ExactTask));
    double *d_lams = (double *)solver->dev_params.ptr;
    double *d_vals = (double *)solver->dev_results.ptr;
    long long *d_cnts = (long long *)((char *)solver->dev_results.ptr + R * 64 * sizeof(double));
    unsigned long long *d_words = (unsigned long long *)solver->dev_bits.ptr;

    size_t word_off = 0;
    for (size_t r = 0; r < R; ++r) {
        const rocco_hip_budget_task &t = tasks[reqs[r].task];
        ExactTask &e = h_tasks[r];
        e.scores = t.scores_dev;
        e.switch_costs = t.switch_costs_dev;
        e.gamma = t.gamma;
        e.n = (long long)t.n;
        e.lambdas = d_lams + r * 64;
        e.n_lambda = (int)reqs[r].lambdas.size();
        e.record_lane = 0;
        e.values_out = d_vals + r * 64;
        e.counts_out = d_cnts + r * 64;
        if (reqs[r].record) {
            e.decision_words = d_words + word_off;
            e.solution = t.solution_dev;
            word_off += (t.n + 30) / 32 + 1;
        } else {
            e.decision_words = nullptr;
            e.solution = nullptr;
        }
        for (size_t l = 0; l < 64; ++l) {
            h_lams[r * 64 + l] = reqs[r].lambdas[l < reqs[r].lambdas.size() ? l : 0];
        }
    }
    ROCCO_HIP_TRY(hipMemcpyAsync(solver->dev_tasks.ptr, h_tasks, R * sizeof(ExactTask),
                                 hipMemcpyHostToDevice, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync(d_lams, h_lams, R * 64 * sizeof(double), hipMemcpyHostToDevice, stream));
    if ((rc = launch_chain_exact((const ExactTask *)solver->dev_tasks.ptr, (int)R, stream)) != ROCCO_HIP_OK) {
        return rc;
    }
    double *b_vals = (double *)solver->host_back.ptr;
    long long *b_cnts = (long long *)((char *)solver->host_back.ptr + R * 64 * sizeof(double));
    ROCCO_HIP_TRY(hipMemcpyAsync(b_vals, d_vals, R * 64 * sizeof(double), hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipMemcpyAsync(b_cnts, d_cnts, R * 64 * sizeof(long long), hipMemcpyDeviceToHost, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    for (size_t r = 0; r < R; ++r) {
        const size_t L = reqs[r].lambdas.size();
        reqs[r].values.assign(b_vals + r * 64, b_vals + r * 64 + L);
        reqs[r].counts.assign(b_cnts + r * 64, b_cnts + r * 64 + L);
    }
    return ROCCO_HIP_OK;
}

// Midpoints of `depth` levels of the reference's bisection tree below (lower, upper), breadth first.
// Node i has children 2i+1 (count <= target: upper = mid) and 2i+2 (count > target: lower = mid).
void build_tree(double lower, double upper, int depth, std::vector<double> &mids)
{
    const size_t nodes = ((size_t)1 << depth) - 1;
    std::vector<double> lo(nodes), hi(nodes);
    mids.assign(nodes, 0.0);
    if (nodes == 0) {
        return;
    }
    lo[0] = lower;
    hi[0] = upper;
    for (size_t i = 0; i < nodes; ++i) {
        const double mid = (lo[i] + hi[i]) / 2.0;  // rocco/dp.py:143
        mids[i] = mid;
        const size_t l = 2 * i + 1, r = 2 * i + 2;
        if (l < nodes) {
            lo[l] = lo[i];
            hi[l] = mid;
        }
        if (r < nodes) {
            lo[r] = mid;
            hi[r] = hi[i];
        }
    }
}

struct Search {
    enum Phase { kAll, kBrackets, kBisect, kFinal, kDone } phase = kBrackets;
    long long target = 0;
    double lower = 0.0, upper = 0.0;
    bool lower_ok = false, upper_ok = false;
    int iters_left = 0;
    double best_lambda = 0.0, best_value = 0.0;
    long long best_count = 0;
    int evals = 0;
    int passes = 0;
    // bookkeeping for the request in flight
    int tree_depth = 0;
    size_t tree_offset = 0;  // index of the first tree node in the request's lambda list
};

}  // namespace

int solve_fixed_penalty(rocco_hip_solver *solver, const double *scores_dev,
                        const double *switch_costs_dev, double gamma, size_t n, double lambda,
                        uint8_t *solution_dev, double *value_out, long long *count_out, int *path_out,
                        hipStream_t stream)
{
    rocco_hip_budget_task t{};
    t.scores_dev = scores_dev;
    t.switch_costs_dev = switch_costs_dev;
    t.gamma = gamma;
    t.n = n;
    t.solution_dev = solution_dev;
    std::vector<EvalRequest> reqs(1);
    reqs[0].task = 0;
    reqs[0].lambdas = {lambda};
    reqs[0].record = (solution_dev != nullptr);
    const int rc = exact_evaluate(solver, &t, reqs, stream);
    if (rc != ROCCO_HIP_OK) {
        return rc;
    }
    if (value_out) *value_out = reqs[0].values[0];
    if (count_out) *count_out = reqs[0].counts[0];
    if (path_out) *path_out = ROCCO_HIP_PATH_EXACT;
    return ROCCO_HIP_OK;
}

int solve_budget_batch(rocco_hip_solver *solver, size_t n_tasks, const rocco_hip_budget_task *tasks,
                       rocco_hip_budget_result *results, hipStream_t stream)
{
    const int kExactDepth = 6;     // 63 speculative penalties per pass
    const int kBracketDepth = 5;   // 2 bracket ends + 31 tree nodes in the first pass
    std::vector<Search> st(n_tasks);
    for (size_t t = 0; t < n_tasks; ++t) {
        Search &s = st[t];
        const long long n = (long long)tasks[t].n;
        s.target = std::max(0LL, std::min(tasks[t].target_count, n));  // rocco/dp.py:101
        s.lower = tasks[t].lower0;
        s.upper = tasks[t].upper0;
        s.iters_left = tasks[t].max_iter;
        s.phase = (s.target == n) ? Search::kAll : Search::kBrackets;  // rocco/dp.py:102-108
    }

    for (;;) {
        std::vector<EvalRequest> reqs;
        std::vector<size_t> owner;
        for (size_t t = 0; t < n_tasks; ++t) {
            Search &s = st[t];
            if (s.phase == Search::kDone) {
                continue;
            }
            EvalRequest r;
            r.task = t;
            if (s.phase == Search::kAll) {
                r.lambdas = {0.0};
                r.record = true;
            } else if (s.phase == Search::kBrackets) {
                r.lambdas = {s.lower, s.upper};
                s.tree_depth = std::min(kBracketDepth, s.iters_left);
                s.tree_offset = 2;
                std::vector<double> mids;
                build_tree(s.lower, s.upper, s.tree_depth, mids);
                r.lambdas.insert(r.lambdas.end(), mids.begin(), mids.end());
            } else if (s.phase == Search::kBisect) {
                s.tree_depth = std::min(kExactDepth, s.iters_left);
                s.tree_offset = 0;
                build_tree(s.lower, s.upper, s.tree_depth, r.lambdas);
            } else {  // kFinal
                r.lambdas = {s.best_lambda};
                r.record = true;
            }
            reqs.push_back(std::move(r));
            owner.push_back(t);
        }
        if (reqs.empty()) {
            break;
        }
        const int rc = exact_evaluate(solver, tasks, reqs, stream);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        for (size_t q = 0; q < reqs.size(); ++q) {
            Search &s = st[owner[q]];
            const EvalRequest &r = reqs[q];
            ++s.passes;
            if (s.phase == Search::kAll) {
                s.best_lambda = 0.0;
                s.best_value = r.values[0];
                s.best_count = r.counts[0];
                s.evals = 1;
                s.phase = Search::kDone;
                continue;
            }
            if (s.phase == Search::kFinal) {
                s.best_value = r.values[0];
                s.best_count = r.counts[0];
                s.phase = Search::kDone;
                continue;
            }
            bool tree_valid = true;
            if (s.phase == Search::kBrackets) {
                if (!s.lower_ok) {  // rocco/dp.py:113-125
                    ++s.evals;
                    if (r.counts[0] <= s.target) {
                        s.lower -= std::max(1.0, std::fabs(s.lower));
                        tree_valid = false;
                    } else {
                        s.lower_ok = true;
                    }
                }
                if (s.lower_ok && !s.upper_ok) {  // rocco/dp.py:127-138
                    ++s.evals;
                    if (r.counts[1] > s.target) {
                        s.upper += std::max(1.0, std::fabs(s.upper));
                        tree_valid = false;
                    } else {
                        s.upper_ok = true;
                        s.best_lambda = r.lambdas[1];
                        s.best_value = r.values[1];
                        s.best_count = r.counts[1];
                    }
                }
                if (!(s.lower_ok && s.upper_ok)) {
                    continue;  // ask again with the expanded bracket
                }
                s.phase = Search::kBisect;
                if (!tree_valid) {
                    continue;  // brackets moved during this pass: the speculative tree is stale
                }
            }
            // walk the evaluated levels (rocco/dp.py:141-162)
            size_t i = 0;
            for (int level = 0; level < s.tree_depth; ++level) {
                const size_t idx = s.tree_offset + i;
                const double mid = r.lambdas[idx];
                ++s.evals;
                if (r.counts[idx] > s.target) {
                    s.lower = mid;
                    i = 2 * i + 2;
                } else {
                    s.upper = mid;
                    s.best_lambda = mid;
                    s.best_value = r.values[idx];
                    s.best_count = r.counts[idx];
                    i = 2 * i + 1;
                }
            }
            s.iters_left -= s.tree_depth;
            if (s.iters_left <= 0) {
                s.phase = Search::kFinal;
            }
        }
    }
    for (size_t t = 0; t < n_tasks; ++t) {
        const Search &s = st[t];
        results[t].selection_penalty = (st[t].target == (long long)tasks[t].n) ? 0.0 : s.upper;
        results[t].penalized_value = s.best_value;
        results[t].selected_count = s.best_count;
        results[t].evaluations = s.evals;
        results[t].path = ROCCO_HIP_PATH_EXACT;
        results[t].passes = s.passes;
    }
    return ROCCO_HIP_OK;
}

}  // namespace rocco
