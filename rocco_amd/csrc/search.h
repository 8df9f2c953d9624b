// rocco_amd/csrc/search.h -- host-side replay of the reference's penalty calibration with
// certification.  Pure C++ (no HIP): the device work is behind the Evaluator interface, so the
// same code is compiled into librocco_hip.so (HIP evaluator) and into the CPU test harness
// tests/host_logic/ (evaluator backed by the CPU oracle) where it is checked against the reference.
//
// What is replayed: rocco/dp.py:89-164 (bracket, bracket expansion, exactly max_iter bisection
// steps with midpoint (lower + upper) / 2.0).  What is certified: DESIGN.md section 4.
#pragma once

#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace rocco {

struct ChainProblem {
    size_t n = 0;
    double gamma = 0.0;     // scalar switch cost (ignored by evaluators that hold a cost vector)
    bool has_cost_vector = false;
    double cost_min = 0.0;  // min / max switch cost (== gamma for the scalar case)
    double cost_max = 0.0;
    double score_min = 0.0, score_max = 0.0;
    double score_abs_sum = -1.0;  // sum of |score| (any summation order), < 0: unknown
    long long target_count = 0;
    double sum_costs = 0.0;  // np.sum(switch_costs) as NumPy computes it (rocco/dp.py:110-111)
    int max_iter = 60;
};

struct ProbeResult {
    long long count = 0;      // selected loci under the exact-rule classes
    long long uncertain = 0;  // loci whose class is not certified
    long long effect = 0;     // bound on |count(reference) - count|
    long long max_run = 0;
};

struct WindowDiff {
    long long locus = 0;
    double margin_lo = 0.0, margin_hi = 0.0;  // delta - decision boundary at lambda_lo / lambda_hi
    long long run = 0;
    int cls_lo = 0, cls_hi = 0;
};

struct WindowResult {
    long long count_lo = 0, count_hi = 0;
    long long n_diff = 0;
    bool diff_adjacent = true;
    bool overflow = false;
    long long max_run = 0;
    std::vector<WindowDiff> diffs;
};

struct ExactResult {
    double value = 0.0;
    long long count = 0;
};

// With the rounding-model probes of a bisection: how the search goes on from them while every outcome is certain (the
// reference's steps, rocco/dp.py:141-162, with the outcomes that need no evaluation).  An evaluator that sequences its
// own launches may evaluate the following rounds' penalties ahead and answer the later requests from what it holds:
// a certified count is a fact about the reference at that penalty, whoever chose the penalty.
struct BisectionAhead {
    bool valid = false;
    double lower = 0.0, upper = 0.0;  // the bracket the request's tree starts from
    int iters_left = 0;
    long long target = 0;
    // outcomes known without evaluation (search.cpp: known_count): at or below G more than the target, at or above L at most
    bool G_real = false, L_real = false;
    double G = 0.0, L = 0.0;
    long long cG = 0, cL = 0;
    // analytic_count's terms: n (|lambda| + sabs + cost_max + 1) < 1e15 and cost_min >= 0;  lambda >= none_from selects
    // nothing, lambda <= all_upto everything
    bool cost_ok = false;
    long long n = 0;
    double sabs = 0.0, cost_max = 0.0, none_from = 0.0, all_upto = 0.0;
    int open_depth = 0;   // open levels per round of this request (before the cut at iters_left)
    int depth_floor = 0;  // what the search's own rule asks for at least in the rounds that follow
};

// One batch round of device work.  Indices refer to the problems of the batch.
struct ProbeRequest {
    size_t problem = 0;
    // bound = true: evaluate every lambda in exact arithmetic on the problem's grid q with no rounding
    // model (results carry only `count`); the caller shifts lambda by -/+ epsilon to bracket the
    // reference's count
    bool bound = false;
    // pilot = true (with bound): the counts may be ESTIMATES from a sample of the loci; they only steer where
    // the certified evaluations of the threshold search are placed and never decide anything
    bool pilot = false;
    std::vector<double> lambdas;
    std::vector<ProbeResult> results;  // filled by the evaluator
    BisectionAhead ahead;              // (optional)
};

struct WindowRequest {
    size_t problem = 0;
    double lambda_lo = 0.0, lambda_hi = 0.0;
    WindowResult result;  // filled by the evaluator; the solution buffer of the problem = fill(LO)
};

// Build (or rebuild) the per-chunk binade map of a problem at penalty lambda_ref; `margin` is the
// distance the reference's running value must keep from every power of two for a chunk to use
// the reference's own rounding grid.
struct MapRequest {
    size_t problem = 0;
    double lambda_ref = 0.0;
    double margin = 0.0;
};

// Restrict a problem to the loci that can still be selected at any penalty >= lambda_base (DESIGN.md
// section 4.7): in exact arithmetic on the grid q the selected sets are nested in the penalty, and the
// reference's own selection at a penalty lambda lies inside the exact one at lambda - eps.  The evaluator
// replaces the problem's arrays by the compacted runs (separated by loci no penalty >= lambda_base can
// select); every later request of that problem is answered on the compacted arrays, and the solution
// is scattered back when the calibration ends.
struct CompactRequest {
    size_t problem = 0;
    double lambda_base = 0.0;
    // filled by the evaluator
    bool done = false;
    size_t n_new = 0;           // loci of the compacted problem (separators included)
    double score_floor = 0.0;   // score of the separator loci (<= every other score that matters)
};

struct ExactRequest {
    size_t problem = 0;
    std::vector<double> lambdas;  // <= 64
    bool write_solution = false;  // solution for lambdas[0]
    std::vector<ExactResult> results;
};

// Exact evaluation of a handful of penalties through the "spine" (chain_fast.hip): the reference's
// own counts, and optionally the reference's solution for one of them.  Needs a binade map that is
// valid for these penalties.
struct SpineRequest {
    size_t problem = 0;
    std::vector<double> lambdas;   // <= 64
    int solution_index = -1;       // write the solution of lambdas[solution_index]
    // Last tree of a bisection: lambdas = the tree in heap order followed by the current upper end.
    // An evaluator that supports it walks the tree itself (count > select_target: right child, else
    // this node becomes the answer and the walk goes left), writes that penalty's solution and
    // reports its index in `selected`; otherwise it leaves selected = -1.
    int select_depth = 0;
    bool select_has_upper = true;  // false: lambdas holds the tree only (selected = -1 if no node qualifies)
    long long select_target = 0;
    int selected = -1;
    std::vector<long long> counts; // filled by the evaluator
    std::vector<long long> stepped; // diagnostic: chunks the spine had to step exactly
};

class Evaluator {
public:
    virtual ~Evaluator() = default;
    // delta-form evaluation of every (problem, lambda): count + certification statistics
    virtual int probe(std::vector<ProbeRequest> &reqs) = 0;
    // joint window evaluation, writes fill(LO) into the problem's solution buffer
    virtual int window(std::vector<WindowRequest> &reqs) = 0;
    // per-chunk binade map used by later probe / window calls on that problem
    virtual int build_map(std::vector<MapRequest> &reqs) = 0;
    // Optional: after a map valid for the whole bracket was built, learn which parts of each
    // problem cannot change any more inside [lambda_lo, lambda_hi] so that later rounds can skip
    // them.  Never changes results; the default does nothing.
    virtual int survey(std::vector<WindowRequest> &reqs)
    {
        (void)reqs;
        return 0;
    }
    // One search iteration's device work.  Requests of different kinds never depend on each other
    // inside an iteration (a problem asks for a map OR a probe OR a window OR a spine; a survey may
    // accompany a probe and only prunes later rounds), so an evaluator may run them all in one pass.
    virtual int round(std::vector<MapRequest> &maps, std::vector<WindowRequest> &surveys,
                      std::vector<ProbeRequest> &probes, std::vector<WindowRequest> &windows,
                      std::vector<SpineRequest> &spines)
    {
        int rc;
        if (!maps.empty() && (rc = build_map(maps)) != 0) return rc;
        if (!surveys.empty() && (rc = survey(surveys)) != 0) return rc;
        if (!probes.empty() && (rc = probe(probes)) != 0) return rc;
        if (!windows.empty() && (rc = window(windows)) != 0) return rc;
        if (!spines.empty() && (rc = spine(spines)) != 0) return rc;
        return 0;
    }
    // Compaction (optional): an evaluator that cannot compact leaves `done` false.
    virtual bool can_compact(size_t problem) const
    {
        (void)problem;
        return false;
    }
    virtual int compact(std::vector<CompactRequest> &reqs)
    {
        (void)reqs;
        return 0;
    }
    // Levels of the bisection tree a probe round of this problem should speculate (0: no preference): an evaluator
    // whose probes cost little per penalty asks for deep trees.
    virtual int probe_depth(size_t problem) const
    {
        (void)problem;
        return 0;
    }
    // A compaction that needs no device work (the evaluator already holds a suitable compacted copy): done at once,
    // so that the problem can go on within the same search iteration.  Default: not available.
    virtual bool compact_now(CompactRequest &req)
    {
        (void)req;
        return false;
    }
    // One search iteration including compactions (which touch no other request of the iteration).
    virtual int round_all(std::vector<CompactRequest> &compacts, std::vector<MapRequest> &maps,
                          std::vector<WindowRequest> &surveys, std::vector<ProbeRequest> &probes,
                          std::vector<WindowRequest> &windows, std::vector<SpineRequest> &spines)
    {
        int rc;
        if (!compacts.empty() && (rc = compact(compacts)) != 0) return rc;
        if (maps.empty() && surveys.empty() && probes.empty() && windows.empty() && spines.empty()) return 0;
        return round(maps, surveys, probes, windows, spines);
    }
    // Can the evaluator estimate exact-arithmetic counts of this problem from a sample, much cheaper than it
    // evaluates them?  (ProbeRequest::pilot)
    virtual bool can_pilot(size_t problem) const
    {
        (void)problem;
        return false;
    }
    // Evaluations per threshold-search round of this problem (an evaluator whose exact-arithmetic
    // counts are cheap may ask for more than the default).
    // `base_hint`: the largest evaluated penalty known to select more than the target (NaN: none yet) -- where the
    // evaluator would compact before the next round.
    virtual int bound_points(size_t problem, int default_points, double base_hint) const
    {
        (void)problem;
        (void)base_hint;
        return default_points;
    }
    // Fraction of a problem's loci that rounds inside the surveyed bracket still evaluate.
    virtual double work_fraction(size_t problem) const
    {
        (void)problem;
        return 1.0;
    }
    // exact counts / solution through the spine (requires a map built by build_map)
    virtual int spine(std::vector<SpineRequest> &reqs) = 0;
    // exact emulation of the reference (always correct)
    virtual int exact(std::vector<ExactRequest> &reqs) = 0;
    // penalised value  sum (s - lambda) z - sum c |dz|  of the solution currently in the buffer
    virtual int penalized_value(size_t problem, double lambda, long long count, double *value_out) = 0;
    // the same for several problems at once (an evaluator may run them concurrently)
    virtual int penalized_values(const std::vector<size_t> &which, const std::vector<double> &lambdas,
                                 const std::vector<long long> &counts, std::vector<double> &values)
    {
        values.assign(which.size(), 0.0);
        for (size_t i = 0; i < which.size(); ++i) {
            const int rc = penalized_value(which[i], lambdas[i], counts[i], &values[i]);
            if (rc != 0) return rc;
        }
        return 0;
    }
};

struct CalibrationResult {
    double selection_penalty = 0.0;
    double penalized_value = 0.0;
    long long selected_count = 0;
    int evaluations = 0;   // chain evaluations the reference would have made
    int path = 0;          // ROCCO_HIP_PATH_*
    int passes = 0;        // device rounds this problem took part in
    int zone_iters = 0;    // bisection steps left when the zone window was taken (-1: none)
    int maps = 0;          // binade maps built
    long long n_diff = -1; // window differences (-1: no window evaluated)
};

struct SearchOptions {
    int spec_depth = 2;     // levels of the bisection tree evaluated per probe round (medium rounds)
    // rounds are throughput-bound when they cover many loci (speculating wastes evaluations) and
    // latency-bound when they cover few (speculating deeper saves launches)
    double big_round_loci = 1.0e18;     // above: one level per round
    double small_round_loci = 1.0e6;   // below: spec_depth + 1 levels per round
    double tiny_round_loci = 0.3e6;     // below: spec_depth + 3 levels per round
    double map_rebuild_ratio = 0.8;    // rebuild a bracket map when its margin would shrink below this ratio
    bool exact_penalty = true;         // settle a lone open decision through the spine (penalty bit-exact)
    int search_points = 2;             // evaluations per round of the threshold search (equally spaced)
    bool search_interpolate = false;   // ... plus one at the log-linear estimate of the crossing (does not pay)
    double search_gate = 0.015;        // the threshold search stops when (loci that can still change) <= gate * workgroups
    bool use_bounds = true;            // decide early bisection steps with shifted exact-arithmetic counts
    double survey_gate = 0.5;          // survey a bracket when (loci that can still change) <= gate * workgroups
    int exact_depth = 6;    // same for the exact kernel (63 lanes)
    bool force_exact = false;
    bool use_spine = true;  // finish undecided endgames through the exact spine (else the exact kernel)
    bool use_compaction = true;  // after the threshold search: continue on the loci that can still be selected
    int pilot_rounds = 2;        // sampled estimates that place the first certified evaluations (0: none)
    int pilot_points = 32;
    // multiples of the target at which the first certified round evaluates (where the pilot's estimate crosses them)
    std::vector<double> pilot_levels = {2.2, 1.35, 1.08, 0.8};
    // A problem whose threshold search has ended asks for its binade map (six small launches and the tolerance cap).
    // Problems of one batch end their search in different rounds; every round is shared and lasts as long as its
    // kernels, so a problem that waits for the others loses nothing -- the batch ends with its slowest member either
    // way -- and the maps of all of them take ONE set of launches instead of one per straggler.
    bool align_maps = true;
    // The same for the window that ends a calibration whose every bisection step is decided (it certifies the final
    // penalty's solution and writes it): while another problem of the batch still asks for probes or a map, it waits --
    // one set of window launches for the batch instead of one per group of finishers.
    bool align_windows = true;
};

// What an evaluator found out about a problem before the calibration starts (chain.hip: the threshold search run as one
// chain of device launches).  Every entry of `evals` is a FACT -- the exact-arithmetic count of the problem at a penalty
// on the problem's grid, DESIGN.md section 4.4 -- however the penalty was chosen; calibrate_batch derives its certified
// thresholds from them with its own epsilon, exactly as it does from the evaluations it asks for itself.
struct Presearch {
    std::vector<std::pair<double, long long>> evals;  // (penalty, count_q(penalty)), in the order they were evaluated
    int rounds = 0;     // device rounds behind them (diagnostic: CalibrationResult::passes)
    bool done = false;  // the evaluator's search ended by the stop rule: nothing more to gain from exact arithmetic
};

// Calibrate every problem of the batch; solutions are left in the evaluator's solution buffers.
int calibrate_batch(Evaluator &ev, const std::vector<ChainProblem> &problems,
                    const SearchOptions &opt, std::vector<CalibrationResult> &results,
                    const std::vector<Presearch> *presearch = nullptr);

// Fixed-penalty solve (rocco/dp.py:49-86) with certification, exact fallback.
int solve_fixed_batch(Evaluator &ev, const std::vector<ChainProblem> &problems,
                      const std::vector<double> &lambdas, const SearchOptions &opt,
                      std::vector<CalibrationResult> &results);

// Is the fast path applicable to this problem at all (cost not degenerate, sizes sane)?
bool fast_path_applicable(const ChainProblem &p);

// Penalties whose outcome is known without evaluation (rocco/_chain_dp.c semantics):
// lambda >= score_max + 1 selects nothing, lambda <= score_min - 1 selects everything.
bool analytic_count(const ChainProblem &p, double lambda, long long *count_out);

}  // namespace rocco
