// tests/host_logic/harness.cpp -- CPU test harness for the product's host-side search logic.
//
// TEST INFRASTRUCTURE: compiles rocco_amd/csrc/search.cpp (the very file that goes into
// librocco_hip.so) against an Evaluator backed by the CPU oracle (oracle/liboracle.so), so the
// certification / replay logic can be checked against the reference without a GPU.
#include <cmath>
#include <cstdio>
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
    double smin0 = 0.0, smax0 = 0.0;  // the caller's score range
    uint8_t *solution;
    std::vector<uint8_t> emap;  // empty = no map
    // compaction (CompactRequest): the arrays above then point into these
    const double *orig_scores = nullptr;
    size_t orig_n = 0;
    uint8_t *orig_solution = nullptr;
    std::vector<double> c_scores;
    std::vector<long long> c_orig;  // original locus of every compacted locus (-1: separator)
    std::vector<uint8_t> c_solution;
    bool compacted = false;
    bool solution_in_orig = false;  // the exact evaluator wrote the original buffer directly
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
    long long compact_calls = 0;
    int compact_tile = 0;  // > 0: keep trailing COPY runs of every tile of this many loci (what the device kernels do)
    bool allow_compact = true;
    double compact_slack = 0.0;  // compact at lambda_base - slack (the device side may hold a copy built lower)
    bool compact_at_once = false;  // serve compactions through compact_now()
    int round_points = 0;          // > 0: penalties per bound round (what the device side asks for on small levels)
    int bound_points(size_t problem, int default_points, double base_hint) const override
    {
        (void)problem;
        (void)base_hint;
        return round_points > 0 ? round_points : default_points;
    }
    double pilot_noise = -1.0;  // >= 0: answer pilot requests with counts off by up to this relative error
    long long pilot_calls = 0;

    bool can_pilot(size_t problem) const override { return pilot_noise >= 0.0 && hp[problem].costs == nullptr; }

    bool can_compact(size_t problem) const override
    {
        return allow_compact && hp[problem].costs == nullptr && !hp[problem].compacted;
    }

    // Selected set at lambda_base in exact arithmetic on the grid q (the recursion of oracle/delta_oracle.c
    // without a map, i.e. what a bound evaluation computes), optionally widened by the trailing COPY runs of
    // every tile; then the compacted arrays: runs of that set, separated by one locus that no penalty >=
    // lambda_base can select.
    bool compact_now(CompactRequest &req) override
    {
        if (!compact_at_once) {
            return false;
        }
        std::vector<CompactRequest> one{req};
        compact(one);
        req = one[0];
        return true;
    }

    int compact(std::vector<CompactRequest> &reqs) override
    {
        ++compact_calls;
        for (CompactRequest &r : reqs) {
            const double lambda_base = r.lambda_base - compact_slack;
            HostProblem &p = hp[r.problem];
            if (p.costs != nullptr || p.compacted) {
                continue;
            }
            const size_t n = p.n;
            const double magic = std::ldexp(1.5, 52 + p.qexp);
            auto rq = [magic](double x) {
                volatile double t = x + magic;
                return t - magic;
            };
            const double c = rq(p.gamma);
            std::vector<uint8_t> cls(n), act(n);
            double d = 0.0;
            for (size_t j = 0; j < n; ++j) {
                const double a = rq(p.scores[j] - lambda_base);
                d = (j == 0) ? a : std::fmin(std::fmax(d, -c), c) + a;
                if (j + 1 < n) {
                    cls[j] = (d > c) ? 2 : ((d <= -c) ? 0 : 1);
                } else {
                    cls[j] = (d > 0.0) ? 2 : 0;
                }
            }
            uint8_t state = 0;
            for (size_t j = n; j-- > 0;) {
                if (compact_tile > 0 && (j + 1) % (size_t)compact_tile == 0 && j + 1 < n) {
                    state = 1;  // fill value entering a tile from the right: unknown to the device, taken as 1
                }
                if (cls[j] != 1) {
                    state = (uint8_t)(cls[j] == 2);
                }
                act[j] = state;
            }
            const double sep = std::floor(r.lambda_base - 2.0 * p.cmax - 2.0);
            p.c_scores.clear();
            p.c_orig.clear();
            for (size_t j = 0; j < n; ++j) {
                if (act[j]) {
                    if (j > 0 && !act[j - 1] && p.c_scores.empty()) {
                        p.c_scores.push_back(sep);  // leading separator: the first run does not start the chain
                        p.c_orig.push_back(-1);
                    }
                    p.c_scores.push_back(p.scores[j]);
                    p.c_orig.push_back((long long)j);
                    if (j + 1 < n && !act[j + 1]) {
                        p.c_scores.push_back(sep);
                        p.c_orig.push_back(-1);
                    }
                }
            }
            if (p.c_scores.empty()) {
                p.c_scores.push_back(sep);
                p.c_orig.push_back(-1);
            }
            p.orig_scores = p.scores;
            p.orig_n = p.n;
            p.orig_solution = p.solution;
            p.c_solution.assign(p.c_scores.size(), 0);
            p.scores = p.c_scores.data();
            p.n = p.c_scores.size();
            p.solution = p.c_solution.data();
            p.emap.clear();
            // grid and magnitude as the product keeps them (rocco_amd/csrc/budget.hip: adopt_level): the caller's score
            // range widened by the separator, never the narrower range of the compacted copy
            double smin = std::fmin(p.smin0, sep), smax = p.smax0;
            p.qexp = grid_exponent(p.cmax, smin, smax);
            p.sabs = std::fmax(std::fabs(smin), std::fabs(smax));
            p.compacted = true;
            r.done = true;
            r.n_new = p.n;
            r.score_floor = sep;
        }
        return 0;
    }

    // compacted problems: copy the solution back to the caller's buffer (zero outside the compacted runs)
    void scatter(size_t problem)
    {
        HostProblem &p = hp[problem];
        if (!p.compacted || p.solution_in_orig) {
            return;
        }
        std::memset(p.orig_solution, 0, p.orig_n);
        for (size_t i = 0; i < p.n; ++i) {
            if (p.c_orig[i] >= 0) {
                p.orig_solution[p.c_orig[i]] = p.c_solution[i];
            }
        }
    }

    int probe(std::vector<ProbeRequest> &reqs) override
    {
        ++probe_calls;
        for (ProbeRequest &r : reqs) {
            const HostProblem &p = hp[r.problem];
            r.results.resize(r.lambdas.size());
            if (r.pilot) {
                ++pilot_calls;
            }
            for (size_t i = 0; i < r.lambdas.size(); ++i) {
                oracle_delta_stats st;
                // bound requests: plain grid-q arithmetic (= the map-less evaluation), count only
                const int rc = oracle_delta_chain_f64(p.scores, p.costs, p.gamma, p.n, r.lambdas[i], p.qexp,
                                                      p.cmax, p.sabs,
                                                      (r.bound || p.emap.empty()) ? nullptr : p.emap.data(), nullptr, &st);
                if (rc != 0) return rc;
                r.results[i].count = st.count;
                if (r.pilot) {
                    // a deliberately wrong estimate (deterministic in the penalty): the search must not depend on it
                    unsigned long long h = 0;
                    std::memcpy(&h, &r.lambdas[i], sizeof(h));
                    h = (h ^ (h >> 29)) * 0x9E3779B97F4A7C15ULL;
                    const double u = (double)(h >> 11) / 9007199254740992.0;  // [0, 1)
                    r.results[i].count = (long long)((double)st.count * (1.0 + pilot_noise * (2.0 * u - 1.0)));
                }
                r.results[i].uncertain = r.bound ? 0 : st.uncertain;
                r.results[i].effect = r.bound ? 0 : (st.overflow ? (long long)p.n + 1 : st.effect);
                r.results[i].max_run = st.max_run;
                if (std::getenv("ROCCO_HOSTLOGIC_TRACE") != nullptr) {
                    std::fprintf(stderr, "[probe] n=%zu %s lambda=%.17g count=%lld uncertain=%lld effect=%lld\n", p.n,
                                 r.bound ? "bound" : (r.pilot ? "pilot" : "model"), r.lambdas[i],
                                 (long long)r.results[i].count, (long long)r.results[i].uncertain,
                                 (long long)r.results[i].effect);
                }
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
            if (std::getenv("ROCCO_HOSTLOGIC_TRACE") != nullptr) {
                std::fprintf(stderr, "[window] n=%zu [%.17g, %.17g] counts %lld %lld n_diff=%lld overflow=%d\n", p.n, r.lambda_lo,
                             r.lambda_hi, (long long)st.count_lo, (long long)st.count_hi, (long long)st.n_diff, (int)st.overflow);
            }
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
            HostProblem &p = hp[r.problem];
            r.results.resize(r.lambdas.size());
            exact_lambdas += (long long)r.lambdas.size();
            // the exact evaluator is the last resort: always on the caller's own arrays
            const double *sc = p.compacted ? p.orig_scores : p.scores;
            const size_t nn = p.compacted ? p.orig_n : p.n;
            uint8_t *sol = p.compacted ? p.orig_solution : p.solution;
            if (r.write_solution && p.compacted) {
                p.solution_in_orig = true;
            }
            for (size_t i = 0; i < r.lambdas.size(); ++i) {
                double v = 0.0;
                long long c = 0;
                const int rc = oracle_solve_penalized_chain_f64(
                    sc, p.costs, p.gamma, nn, r.lambdas[i],
                    (r.write_solution && i == 0) ? sol : nullptr, &v, &c);
                if (rc != 0) return rc;
                r.results[i].value = v;
                r.results[i].count = c;
            }
        }
        return 0;
    }
    int penalized_value(size_t problem, double lambda, long long count, double *value_out) override
    {
        scatter(problem);
        const HostProblem &p = hp[problem];
        if (p.compacted) {
            const double obj = oracle_objective_value_f64(p.orig_solution, p.orig_scores, p.costs, p.gamma, p.orig_n);
            *value_out = -obj - lambda * (double)count;
            return 0;
        }
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
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_COMPACT")) ev.allow_compact = std::atoi(e) != 0;
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_TILE")) ev.compact_tile = std::atoi(e);
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_PILOT")) ev.pilot_noise = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_SLACK")) ev.compact_slack = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_NOW")) ev.compact_at_once = std::atoi(e) != 0;
    HostProblem h;
    h.scores = scores;
    h.costs = costs;
    h.gamma = gamma;
    h.n = n;
    h.qexp = grid_exponent(cmax, smin, smax);
    h.smin0 = smin;
    h.smax0 = smax;
    h.cmax = cmax;
    h.sabs = std::fmax(std::fabs(smin), std::fabs(smax));
    h.solution = solution;
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
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_PILOT_ROUNDS")) opt.pilot_rounds = std::atoi(e);
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_PILOT_POINTS")) opt.pilot_points = std::atoi(e);
    if (const char *e = std::getenv("ROCCO_HOSTLOGIC_POINTS")) ev.round_points = std::atoi(e);
    std::vector<CalibrationResult> res;
    const int rc = calibrate_batch(ev, {p}, opt, res);
    if (rc != 0) return rc;
    ev.scatter(0);
    out_i[11] = ev.hp[0].compacted ? (long long)ev.hp[0].n : -1;
    out_i[12] = ev.pilot_calls;
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
    HostProblem h;
    h.scores = scores;
    h.costs = costs;
    h.gamma = gamma;
    h.n = n;
    h.qexp = grid_exponent(cmax, lo, hi);
    h.smin0 = smin;
    h.smax0 = smax;
    h.cmax = cmax;
    h.sabs = std::fmax(std::fabs(smin), std::fabs(smax));
    h.solution = solution;
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
