// tests/host_logic/harness.cpp -- CPU test harness for the product's host-side search logic.
//
// TEST INFRASTRUCTURE: compiles rocco_amd/csrc/search.cpp (the very file that goes into
// librocco_hip.so) against an Evaluator backed by the CPU oracle (oracle/liboracle.so), so the
// certification / replay logic can be checked against the reference without a GPU.
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "../../include/rocco_hip.h"
#include "../../oracle/oracle.h"
#include "../../rocco_amd/csrc/search.h"

using namespace rocco;

namespace {

struct HostProblem {
    const double *scores;
    const double *costs;  // may be null
    double gamma;
    size_t n;
    int qexp;
    double cmax, sabs;
    uint8_t *solution;
    std::vector<uint8_t> emap;  // empty = no map
};

int grid_exponent(double cmax, double smin, double smax)
{
    const double r = cmax + (smax - smin) + 2.0;
    return (int)std::ceil(std::log2(8.0 * r)) - 52;
}

class OracleEvaluator : public Evaluator {
public:
    std::vector<HostProblem> hp;
    long long probe_calls = 0, window_calls = 0, exact_calls = 0, exact_lambdas = 0, map_calls = 0, spine_calls = 0;

    int probe(std::vector<ProbeRequest> &reqs) override
    {
        ++probe_calls;
        for (ProbeRequest &r : reqs) {
            const HostProblem &p = hp[r.problem];
            r.results.resize(r.lambdas.size());
            for (size_t i = 0; i < r.lambdas.size(); ++i) {
                oracle_delta_stats st;
                // bound requests: plain grid-q arithmetic (= the map-less evaluation), count only
                const int rc = oracle_delta_chain_f64(p.scores, p.costs, p.gamma, p.n, r.lambdas[i], p.qexp,
                                                      p.cmax, p.sabs,
                                                      (r.bound || p.emap.empty()) ? nullptr : p.emap.data(), nullptr, &st);
                if (rc != 0) return rc;
                r.results[i].count = st.count;
                r.results[i].uncertain = r.bound ? 0 : st.uncertain;
                r.results[i].effect = r.bound ? 0 : (st.overflow ? (long long)p.n + 1 : st.effect);
                r.results[i].max_run = st.max_run;
            }
        }
        return 0;
    }
    int window(std::vector<WindowRequest> &reqs) override
    {
        ++window_calls;
        for (WindowRequest &r : reqs) {
            const HostProblem &p = hp[r.problem];
            oracle_window_stats st;
            oracle_window_diff diffs[16];
            const int rc = oracle_delta_window_f64(p.scores, p.costs, p.gamma, p.n, r.lambda_lo, r.lambda_hi,
                                                   p.qexp, p.cmax, p.sabs, p.emap.empty() ? nullptr : p.emap.data(), p.solution, &st,
                                                   diffs, 16);
            if (rc != 0) return rc;
            r.result.count_lo = st.count_lo;
            r.result.count_hi = st.count_hi;
            r.result.n_diff = st.n_diff;
            r.result.diff_adjacent = st.diff_adjacent != 0;
            r.result.overflow = st.overflow != 0;
            r.result.max_run = st.max_run;
            r.result.diffs.clear();
            for (long long i = 0; i < st.n_diff && i < 16; ++i) {
                WindowDiff d;
                d.locus = diffs[i].locus;
                d.margin_lo = diffs[i].margin_lo;
                d.margin_hi = diffs[i].margin_hi;
                d.run = diffs[i].run;
                d.cls_lo = diffs[i].cls_lo;
                d.cls_hi = diffs[i].cls_hi;
                r.result.diffs.push_back(d);
            }
        }
        return 0;
    }
    int build_map(std::vector<MapRequest> &reqs) override
    {
        ++map_calls;
        for (MapRequest &r : reqs) {
            HostProblem &p = hp[r.problem];
            p.emap.assign((p.n + ORACLE_CHUNK - 1) / ORACLE_CHUNK, 0);
            const int rc = oracle_binade_map(p.scores, p.costs, p.gamma, p.n, r.lambda_ref, p.qexp, r.margin,
                                             p.emap.data());
            if (rc != 0) return rc;
        }
        return 0;
    }
    int spine(std::vector<SpineRequest> &reqs) override
    {
        ++spine_calls;
        for (SpineRequest &r : reqs) {
            const HostProblem &p = hp[r.problem];
            if (p.emap.empty()) return -2;  // the product only asks for the spine with a map in place
            r.counts.resize(r.lambdas.size());
            for (size_t i = 0; i < r.lambdas.size(); ++i) {
                double v = 0.0;
                long long c = 0;
                const int rc = oracle_solve_penalized_chain_f64(
                    p.scores, p.costs, p.gamma, p.n, r.lambdas[i],
                    ((int)i == r.solution_index) ? p.solution : nullptr, &v, &c);
                if (rc != 0) return rc;
                r.counts[i] = c;
            }
        }
        return 0;
    }
    int exact(std::vector<ExactRequest> &reqs) override
    {
        ++exact_calls;
        for (ExactRequest &r : reqs) {
            const HostProblem &p = hp[r.problem];
            r.results.resize(r.lambdas.size());
            exact_lambdas += (long long)r.lambdas.size();
            for (size_t i = 0; i < r.lambdas.size(); ++i) {
                double v = 0.0;
                long long c = 0;
                const int rc = oracle_solve_penalized_chain_f64(
                    p.scores, p.costs, p.gamma, p.n, r.lambdas[i],
                    (r.write_solution && i == 0) ? p.solution : nullptr, &v, &c);
                if (rc != 0) return rc;
                r.results[i].value = v;
                r.results[i].count = c;
            }
        }
        return 0;
    }
    int penalized_value(size_t problem, double lambda, long long count, double *value_out) override
    {
        const HostProblem &p = hp[problem];
        const double obj = oracle_objective_value_f64(p.solution, p.scores, p.costs, p.gamma, p.n);
        *value_out = -obj - lambda * (double)count;
        return 0;
    }
};

}  // namespace

extern "C" {

// Calibrate one chromosome with the product's search logic on the CPU oracle backend.
// out_i: [path, evaluations, passes, zone_iters, n_diff, probe_calls, window_calls, exact_calls, exact_lambdas]
int hostlogic_calibrate(const double *scores, const double *costs, double gamma, size_t n,
                        long long target, double sum_costs, int max_iter, int spec_depth, int force_exact,
                        uint8_t *solution, double *penalty_out, double *value_out,
                        long long *count_out, long long *out_i)
{
    double smin = scores[0], smax = scores[0];
    for (size_t i = 1; i < n; ++i) {
        smin = std::fmin(smin, scores[i]);
        smax = std::fmax(smax, scores[i]);
    }
    double cmin = gamma, cmax = gamma;
    if (costs != nullptr && n > 1) {
        cmin = cmax = costs[0];
        for (size_t i = 1; i + 1 < n; ++i) {
            cmin = std::fmin(cmin, costs[i]);
            cmax = std::fmax(cmax, costs[i]);
        }
    }
    OracleEvaluator ev;
    HostProblem h{scores, costs, gamma, n, grid_exponent(cmax, smin, smax), cmax,
                  std::fmax(std::fabs(smin), std::fabs(smax)), solution, {}};
    ev.hp.push_back(h);
    ChainProblem p;
    p.n = n;
    p.gamma = gamma;
    p.has_cost_vector = costs != nullptr;
    p.cost_min = cmin;
    p.cost_max = cmax;
    p.score_min = smin;
    p.score_max = smax;
    {
        double sabs_sum = 0.0;
        for (size_t i = 0; i < n; ++i) {
            sabs_sum += std::fabs(scores[i]);
        }
        p.score_abs_sum = sabs_sum;
    }
    p.target_count = target;
    p.sum_costs = sum_costs;
    p.max_iter = max_iter;
    SearchOptions opt;
    opt.spec_depth = spec_depth;
    opt.force_exact = force_exact != 0;
    if (const char *e = std::getenv("ROCCO_HIP_BOUNDS")) opt.use_bounds = std::atoi(e) != 0;
    std::vector<CalibrationResult> res;
    const int rc = calibrate_batch(ev, {p}, opt, res);
    if (rc != 0) return rc;
    *penalty_out = res[0].selection_penalty;
    *value_out = res[0].penalized_value;
    *count_out = res[0].selected_count;
    out_i[0] = res[0].path;
    out_i[1] = res[0].evaluations;
    out_i[2] = res[0].passes;
    out_i[3] = res[0].zone_iters;
    out_i[4] = res[0].n_diff;
    out_i[5] = ev.probe_calls;
    out_i[6] = ev.window_calls;
    out_i[7] = ev.exact_calls;
    out_i[8] = ev.exact_lambdas;
    out_i[9] = res[0].maps;
    out_i[10] = ev.spine_calls;
    return 0;
}

int hostlogic_solve_fixed(const double *scores, const double *costs, double gamma, size_t n, double lambda,
                          uint8_t *solution, double *value_out, long long *count_out,
                          long long *out_i)
{
    double smin = scores[0], smax = scores[0];
    for (size_t i = 1; i < n; ++i) {
        smin = std::fmin(smin, scores[i]);
        smax = std::fmax(smax, scores[i]);
    }
    double cmin = gamma, cmax = gamma;
    if (costs != nullptr && n > 1) {
        cmin = cmax = costs[0];
        for (size_t i = 1; i + 1 < n; ++i) {
            cmin = std::fmin(cmin, costs[i]);
            cmax = std::fmax(cmax, costs[i]);
        }
    }
    OracleEvaluator ev;
    // penalties outside [smin - 1, smax + 1] are legal here: widen the grid range accordingly
    const double lo = std::fmin(smin, lambda), hi = std::fmax(smax, lambda);
    HostProblem h{scores, costs, gamma, n, grid_exponent(cmax, lo, hi), cmax,
                  std::fmax(std::fabs(smin), std::fabs(smax)), solution, {}};
    ev.hp.push_back(h);
    ChainProblem p;
    p.n = n;
    p.gamma = gamma;
    p.has_cost_vector = costs != nullptr;
    p.cost_min = cmin;
    p.cost_max = cmax;
    p.score_min = smin;
    p.score_max = smax;
    SearchOptions opt;
    std::vector<CalibrationResult> res;
    const int rc = solve_fixed_batch(ev, {p}, {lambda}, opt, res);
    if (rc != 0) return rc;
    *value_out = res[0].penalized_value;
    *count_out = res[0].selected_count;
    out_i[0] = res[0].path;
    out_i[4] = res[0].n_diff;
    return 0;
}
}
