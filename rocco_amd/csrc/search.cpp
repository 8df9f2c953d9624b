// rocco_amd/csrc/search.cpp -- see search.h.  Pure host logic, no HIP.
#include "search.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>

#include "../../include/rocco_hip.h"

namespace rocco {

namespace {

// Midpoints of `depth` levels of the reference's bisection tree below (lower, upper), breadth
// first.  Node i has children 2i+1 (count <= target: upper = mid) and 2i+2 (count > target:
// lower = mid).  The midpoint expression is the reference's own (rocco/dp.py:143).
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
        const double mid = (lo[i] + hi[i]) / 2.0;
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

enum class Outcome { kGreater, kLessEqual, kUncertain };

Outcome classify(const ProbeResult &r, long long target)
{
    if (r.count - r.effect > target) {
        return Outcome::kGreater;
    }
    if (r.count + r.effect <= target) {
        return Outcome::kLessEqual;
    }
    return Outcome::kUncertain;
}

struct State {
    enum Phase {
        kAll,           // target == n: single solve at penalty 0 (rocco/dp.py:102-108)
        kLowerBracket,  // rocco/dp.py:113-125
        kUpperBracket,  // rocco/dp.py:127-138
        kBisect,        // rocco/dp.py:141-162
        kNeedMap,       // an uncertain probe was met and the binade map is missing / too coarse
        kZone,          // an uncertain probe was met: joint window over the current bracket
        kFinalExact,    // exact solve at the final penalty (writes the solution)
        kFinalSpine,    // spine solve at the final penalty (writes the solution)
        kDone
    } phase = kLowerBracket;
    long long target = 0;
    double lower = 0.0, upper = 0.0;
    int iters_left = 0;
    bool use_exact = false;
    bool use_spine = false;  // decisions left open by the window: finish through the exact spine
    bool has_map = false;
    double map_lo = 0.0, map_hi = 0.0;  // penalties the current binade map is valid for
    bool point_pending = false;         // the map is a point map at the next midpoint: probe it alone
    double req_ref = 0.0, req_lo = 0.0, req_hi = 0.0, req_margin = 0.0;  // map being requested
    double map_margin = 0.0;                                              // margin of the map in use
    double survey_width = -1.0;                                           // bracket width at the last survey
    // Threshold search (exact arithmetic on the grid q).  count_q is monotone in the
    // penalty and brackets the reference:  count_q(x) > target  =>  the reference selects more than the
    // target at every penalty <= x - eps;  count_q(x) <= target  =>  at most the target at every penalty
    // >= x + eps.  G / L are the best such certified thresholds; every bracket or bisection step of the
    // reference whose penalty lies outside (G, L) is then decided on the host for free.
    bool searching = false;
    // pilot: estimated counts from a sample of the loci place the first certified evaluations
    int pilot_left = 0;
    double pg = 0.0, pl = 0.0;
    std::vector<std::pair<double, double>> pilot_evals;  // (x, estimated count)
    bool pilot_round = false;
    bool pilot_hint = false;  // the next certified round takes its points from pilot_evals
    bool want_compact = false;  // the search has ended: restrict the problem to the loci that can still be selected
    bool compacted = false;
    bool bound_round = false;  // the probe request in flight belongs to the threshold search
    double eps = 0.0;
    double G = 0.0, L = 0.0;
    long long cG = 0, cL = 0;   // counts behind the thresholds (cG > target >= cL)
    bool G_real = false, L_real = false;  // thresholds come from an evaluation (not from the analytic range)
    int search_rounds = 0;
    long long open_before = 0x7FFFFFFFFFFFFFFFLL;  // cG - cL after the previous round
    std::vector<std::pair<double, long long>> evals;  // every (x, count_q(x)) evaluated so far
    long long lower_count = 0;   // selected loci at `lower` (bounds the count anywhere in the bracket)
    long long upper_count = -1;  // selected loci at `upper` (-1: not evaluated yet)
    Phase after_map = kBisect;
    CalibrationResult out;
    // request in flight
    int tree_depth = 0;
    std::vector<double> tree;
    std::vector<int> tree_slot;  // index into the request's penalty list, or -1 if decided analytically
    // probe rounds: the reference's bisection tree below (lower, upper), grown only where the outcome is open
    struct TreeNode {
        double mid;
        int slot;    // index into the request's penalty list, or -1: outcome known without device work
        int le, gt;  // next node after "count <= target" / "count > target" (-1: the round ends there)
    };
    std::vector<TreeNode> nodes;
    struct TreeItem {
        double lo, hi;
        int open, steps, parent;
        bool gt;
    };
    std::vector<TreeItem> queue;  // build_open_tree's work list
};

// Outcome of the reference at `lambda` known without device work: analytic, or outside the certified
// thresholds of the search.  `c` is only compared with the target.
bool known_count(const ChainProblem &p, const State &s, double lambda, long long *c)
{
    if (analytic_count(p, lambda, c)) {
        return true;
    }
    if (s.G_real && lambda <= s.G) {
        *c = s.cG;
        return true;
    }
    if (s.L_real && lambda >= s.L) {
        *c = s.cL;
        return true;
    }
    return false;
}

// The reference's next bisection steps as a tree: a step whose outcome is known (analytically or from the certified
// thresholds) costs nothing and has one successor; an open one is evaluated and has two.  `open_depth` open steps per
// path at most, `max_steps` steps in all (the iterations the reference has left).
void build_open_tree(const ChainProblem &p, State &s, int open_depth, int max_steps, std::vector<State::TreeNode> &nodes,
                     std::vector<double> &lambdas)
{
    using Item = State::TreeItem;
    nodes.clear();
    std::vector<Item> &queue = s.queue;  // (kept between rounds: no allocation on the way)
    queue.clear();
    queue.push_back({s.lower, s.upper, 0, 0, -1, false});
    for (size_t at = 0; at < queue.size() && nodes.size() < 8192; ++at) {
        const Item it = queue[at];
        if (it.steps >= max_steps || it.open >= open_depth) {
            continue;
        }
        const double mid = (it.lo + it.hi) / 2.0;  // rocco/dp.py:143
        long long c = 0;
        const bool known = known_count(p, s, mid, &c);
        State::TreeNode node;
        node.mid = mid;
        node.slot = known ? -1 : (int)lambdas.size();
        node.le = node.gt = -1;
        const int idx = (int)nodes.size();
        nodes.push_back(node);
        if (!known) {
            lambdas.push_back(mid);
        }
        if (it.parent >= 0) {
            (it.gt ? nodes[(size_t)it.parent].gt : nodes[(size_t)it.parent].le) = idx;
        }
        if (!known || c > s.target) {
            queue.push_back({mid, it.hi, it.open + (known ? 0 : 1), it.steps + 1, idx, true});
        }
        if (!known || c <= s.target) {
            queue.push_back({it.lo, mid, it.open + (known ? 0 : 1), it.steps + 1, idx, false});
        }
    }
}

// Lowest penalty the device can still be asked about: below the certified threshold G every outcome is
// known ("more than the target"), so maps, surveys and windows only need to hold from here to `upper`.
double eff_lower(const State &s)
{
    return (s.G_real && s.G > s.lower) ? s.G : s.lower;
}

// Highest penalty the device is still asked about while bisecting (at and above L every outcome is
// known).  Only the final window must reach up to `upper` itself.
double eff_upper(const State &s)
{
    return (s.L_real && s.L < s.upper) ? s.L : s.upper;
}

// Counts bounding the reference's count at the ends of the current bracket, from the evaluations of
// the search (count_q is non-increasing and brackets the reference's count within eps).
void bracket_counts_from_evals(const ChainProblem &p, State &s)
{
    long long up = (long long)p.n, lo = -1;
    for (const auto &e : s.evals) {
        if (e.first <= s.lower - s.eps && e.second < up) {
            up = e.second;  // count(lower) <= count_q(lower - eps) <= count_q(x) for x <= lower - eps
        }
        if (e.first >= s.upper + s.eps && e.second > lo) {
            lo = e.second;  // count(upper) >= count_q(upper + eps) >= count_q(x) for x >= upper + eps
        }
    }
    s.lower_count = up;
    s.upper_count = (lo < 0) ? 0 : lo;
}

// Advance through bracket / bisection steps whose outcome is known without device work.
void advance_analytic(const ChainProblem &p, State &s)
{
    long long c = 0;
    for (;;) {
        if (s.phase == State::kLowerBracket) {
            if (!known_count(p, s, s.lower, &c)) {
                return;
            }
            ++s.out.evaluations;
            if (c <= s.target) {
                s.lower -= std::max(1.0, std::fabs(s.lower));
            } else {
                s.phase = State::kUpperBracket;
            }
        } else if (s.phase == State::kUpperBracket) {
            if (!known_count(p, s, s.upper, &c)) {
                return;
            }
            ++s.out.evaluations;
            if (c > s.target) {
                s.upper += std::max(1.0, std::fabs(s.upper));
            } else {
                s.phase = State::kBisect;
            }
        } else if (s.phase == State::kBisect) {
            if (s.iters_left <= 0) {
                return;
            }
            const double mid = (s.lower + s.upper) / 2.0;
            if (!known_count(p, s, mid, &c)) {
                return;
            }
            ++s.out.evaluations;
            --s.iters_left;
            if (c > s.target) {
                s.lower = mid;
            } else {
                s.upper = mid;
            }
        } else {
            return;
        }
    }
}

// Replay the remaining bisection steps when count > target  <=>  mid < critical.
void replay_with_critical(const ChainProblem &p, State &s, double critical)
{
    while (s.iters_left > 0) {
        const double mid = (s.lower + s.upper) / 2.0;
        long long c = 0;
        const bool greater = known_count(p, s, mid, &c) ? (c > s.target) : (mid < critical);
        if (greater) {
            s.lower = mid;
        } else {
            s.upper = mid;
        }
        ++s.out.evaluations;
        --s.iters_left;
    }
}

// How far the reference's intermediates can sit from its stay-off value (the distance a clean chunk of the binade
// map keeps from every power of two): c + |s - lambda| for the state-1 value, plus |s| because the reference adds the
// score before it subtracts the penalty (rocco/_chain_dp.c:120,125,127-128).
double model_reach(const ChainProblem &p, double lambda_lo, double lambda_hi)
{
    const double sabs = std::max(std::fabs(p.score_min), std::fabs(p.score_max));
    return p.cost_max + (std::max(p.score_max, lambda_hi) - std::min(p.score_min, lambda_lo)) + sabs + 2.0;
}

// Decide which binade map to build for the current bracket: one valid for the whole bracket when
// the running values cannot move much inside it, otherwise a point map at the next midpoint.
void plan_map(const ChainProblem &p, State &s, bool force_bracket)
{
    const double lo = eff_lower(s);
    const double hi = force_bracket ? s.upper : eff_upper(s);  // (forced: for the final window)
    const double width = hi - lo;
    const double reach = model_reach(p, p.score_min, p.score_max);
    const double drift = 2.0 * width * (double)s.lower_count;
    if (force_bracket || drift <= 4.0 * reach) {
        s.req_ref = (lo + hi) / 2.0;
        s.req_lo = lo;
        s.req_hi = hi;
        s.req_margin = reach + drift + 2.0;
        s.point_pending = false;
    } else {
        s.req_ref = s.req_lo = s.req_hi = (s.lower + s.upper) / 2.0;
        s.req_margin = reach + 2.0;
        s.point_pending = true;
    }
    s.phase = State::kNeedMap;
}

}  // namespace

// Shift that makes exact arithmetic on the grid q bracket the reference: every
// step of the reference's recursion, seen through delta = prev1 - prev0, differs from the exact step by
// at most w = 4 hb + q and its comparisons have slack 9 hb + 2 q, hb = 2^(e+2-53) with 2^(e+1) above every
// running value (same constants and the same bound on the running values as the hazard mode of
// oracle/delta_oracle.c, with sum |s| in place of the sum of the positive parts).  With eps >= w + slack, a
// multiple of q:  count_q(lambda + eps) <= count_reference(lambda) <= count_q(lambda - eps).
static int bound_grid_exponent(const ChainProblem &p)
{
    const double r = std::max(p.cost_max, 0.0) + (p.score_max - p.score_min) + 2.0;
    return (int)std::ceil(std::log2(8.0 * r)) - 52;
}

// nearest multiple of q = 2^qexp (bound evaluations run on scores rounded to that grid once per tile)
static double snap_to_grid(double x, int qexp)
{
    return std::ldexp(std::nearbyint(std::ldexp(x, -qexp)), qexp);
}

static bool bound_epsilon(const ChainProblem &p, double lambda, double *eps_out)
{
    // sum_j max(0, s_j - lambda) <= sum |s_j| + n * max(0, -lambda)
    const double sabs = std::max(std::fabs(p.score_min), std::fabs(p.score_max));
    const double sum_abs = (p.score_abs_sum >= 0.0) ? (p.score_abs_sum * (1.0 + 1e-9) + 1.0) : ((double)p.n * sabs);
    const double pos = sum_abs + (double)p.n * std::max(0.0, -lambda);
    const double pb = 2.0 * (pos + (double)p.n * 0.0625 + std::max(p.cost_max, 0.0) + sabs + std::fabs(lambda) + 1.0);
    if (!(pb > 0.0) || !std::isfinite(pb) || !(p.cost_min >= 0.0)) {
        return false;
    }
    const int e = std::ilogb(pb);
    const double r = std::max(p.cost_max, 0.0) + (p.score_max - p.score_min) + 2.0;
    const int qexp = (int)std::ceil(std::log2(8.0 * r)) - 52;
    const double hb = std::ldexp(1.0, e + 2 - 53);
    const double q = std::ldexp(1.0, qexp);
    if (!(hb >= q)) {
        return false;
    }
    *eps_out = 16.0 * hb + 4.0 * q;
    return true;
}

bool analytic_count(const ChainProblem &p, double lambda, long long *count_out)
{
    // every partial sum of the reference stays far below 2^50 so that a margin of 1 in every
    // decision dwarfs the rounding of its running values
    const double mag = (double)p.n * (std::fabs(lambda) + std::max(std::fabs(p.score_min), std::fabs(p.score_max)) + p.cost_max + 1.0);
    if (!(mag < 1.0e15) || !(p.cost_min >= 0.0)) {
        return false;
    }
    if (lambda >= p.score_max + 1.0) {
        *count_out = 0;
        return true;
    }
    if (lambda <= p.score_min - 1.0) {
        *count_out = (long long)p.n;
        return true;
    }
    return false;
}

bool fast_path_applicable(const ChainProblem &p)
{
    if (!(p.cost_min >= 1.0e-3) || !(p.cost_max <= 1.0e6)) {
        return false;
    }
    if (!std::isfinite(p.score_min) || !std::isfinite(p.score_max)) {
        return false;
    }
    if (!(p.score_max - p.score_min <= 1.0e9) || !(std::max(std::fabs(p.score_min), std::fabs(p.score_max)) <= 1.0e12)) {
        return false;
    }
    return p.n >= 1;
}

int calibrate_batch(Evaluator &ev, const std::vector<ChainProblem> &problems_in,
                    const SearchOptions &opt, std::vector<CalibrationResult> &results,
                    const std::vector<Presearch> *presearch)
{
    // (a compaction replaces a problem's arrays: its length and score floor change on the way)
    struct Tick {
        static double now()
        {
            timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
        }
    };
    const bool timing = std::getenv("ROCCO_SEARCH_TIMING") != nullptr;
    const bool search_debug = std::getenv("ROCCO_SEARCH_DEBUG") != nullptr;
    const double tick0 = Tick::now();
    int tick_iter = 0;
    std::vector<ChainProblem> problems(problems_in);
    const size_t B = problems.size();
    std::vector<State> st(B);
    for (size_t b = 0; b < B; ++b) {
        const ChainProblem &p = problems[b];
        State &s = st[b];
        s.target = std::max(0LL, std::min(p.target_count, (long long)p.n));  // rocco/dp.py:101
        s.lower = p.score_min - p.sum_costs - 1.0;                            // rocco/dp.py:110
        s.upper = p.score_max + p.sum_costs + 1.0;                            // rocco/dp.py:111
        s.iters_left = p.max_iter;
        s.use_exact = opt.force_exact || !fast_path_applicable(p);
        s.phase = (s.target == (long long)p.n) ? State::kAll : State::kLowerBracket;
        s.out.zone_iters = -1;
        s.lower_count = (long long)p.n;
        s.nodes.reserve(32);  // (here, while the device still works on what was queued before: not between two rounds)
        s.queue.reserve(64);
        if (opt.use_bounds && !s.use_exact && s.phase != State::kAll && bound_epsilon(p, p.score_min - 1.0, &s.eps)) {
            // penalties outside [s_min - 1, s_max + 1] are decided analytically: the search starts there
            s.searching = true;
            s.G = p.score_min - 1.0;
            s.L = p.score_max + 1.0;
            s.cG = (long long)p.n;
            s.cL = 0;
            if (opt.pilot_rounds > 0 && s.target > 0 && ev.can_pilot(b)) {
                s.pilot_left = opt.pilot_rounds;
                s.pg = s.G;
                s.pl = s.L;
            }
            if (presearch != nullptr && b < presearch->size() && !(*presearch)[b].evals.empty()) {
                // exact-arithmetic counts the evaluator already holds: the thresholds they certify (same rule as for a
                // bound round of this search, below), no pilot
                const Presearch &ps = (*presearch)[b];
                s.pilot_left = 0;
                for (const auto &e : ps.evals) {
                    const double x = e.first;
                    const long long c = e.second;
                    if (!(x > p.score_min - 1.0 && x < p.score_max + 1.0)) {
                        continue;  // (outside the interval the grid was sized for: not a fact this search may use)
                    }
                    s.evals.emplace_back(x, c);
                    if (c > s.target) {
                        if (!s.G_real || x - s.eps > s.G) {
                            s.G = x - s.eps;
                            s.cG = c;
                            s.G_real = true;
                        }
                    } else if (!s.L_real || x + s.eps < s.L) {
                        s.L = x + s.eps;
                        s.cL = c;
                        s.L_real = true;
                    }
                }
                s.search_rounds = ps.rounds;
                s.out.passes += ps.rounds;
                s.open_before = (s.G_real && s.L_real) ? (s.cG - s.cL) : (long long)p.n;
                if (search_debug) {
                    std::fprintf(stderr, "[search] problem %zu presearch (%zu counts, %d rounds%s): G %.17g (%lld) L %.17g (%lld) width %.3g eps %.3g\n",
                                 b, ps.evals.size(), ps.rounds, ps.done ? ", ended" : "", s.G, s.cG, s.L, s.cL, s.L - s.G, s.eps);
                }
                if (ps.done || s.L - s.G <= 8.0 * s.eps) {
                    s.searching = false;
                    s.want_compact = opt.use_compaction && s.G_real && !s.compacted && ev.can_compact(b);
                    advance_analytic(p, s);
                    bracket_counts_from_evals(p, s);
                }
            }
        }
    }

    if (timing) std::fprintf(stderr, "[search timing] setup %.1f us\n", Tick::now() - tick0);
    // (the request lists live across the iterations: their storage is taken once, here, not between two rounds)
    std::vector<ProbeRequest> probes;
    std::vector<WindowRequest> windows;
    std::vector<ExactRequest> exacts;
    std::vector<MapRequest> maps;
    std::vector<SpineRequest> spines;
    std::vector<size_t> probe_owner, window_owner, exact_owner, map_owner, spine_owner, compact_owner;
    std::vector<WindowRequest> surveys;
    std::vector<CompactRequest> compacts;
    std::vector<size_t> first_maps;  // positions in `maps` of maps asked for right after a threshold search
    std::vector<size_t> final_windows;  // positions in `windows` of windows that only certify and write a decided penalty
    probes.reserve(B);
    windows.reserve(B);
    maps.reserve(B);
    probe_owner.reserve(B);
    window_owner.reserve(B);
    map_owner.reserve(B);
    for (;;) {
        const double tick_a = Tick::now();
        probes.clear();
        windows.clear();
        exacts.clear();
        maps.clear();
        spines.clear();
        probe_owner.clear();
        window_owner.clear();
        exact_owner.clear();
        map_owner.clear();
        spine_owner.clear();
        compact_owner.clear();
        surveys.clear();
        compacts.clear();
        first_maps.clear();
        final_windows.clear();

        // speculation depth of this iteration's probe rounds, from the loci they will cover
        double round_loci = 0.0;
        bool all_compacted = true;
        for (size_t b = 0; b < B; ++b) {
            if (st[b].phase != State::kDone && !st[b].use_exact) {
                round_loci += (double)problems[b].n * ev.work_fraction(b);
                all_compacted = all_compacted && st[b].compacted;
            }
        }
        // rounds over compacted problems are launch-bound (about 110 us whatever they evaluate) until their
        // (workgroup, penalty) pairs fill the device a few times over: speculate as deep as that allows
        int deep = opt.spec_depth;
        int deep_floor = opt.spec_depth;  // (what the rule below asks for before the evaluator's own wish)
        if (all_compacted) {
            double blocks = 0.0;
            for (size_t b = 0; b < B; ++b) {
                if (st[b].phase != State::kDone && !st[b].use_exact) {
                    blocks += (double)(problems[b].n / 8192 + 1);
                }
            }
            deep = opt.spec_depth + 3;
            while (deep > opt.spec_depth && blocks * (double)((1 << deep) - 1) > 1600.0) {
                --deep;
            }
            deep_floor = deep;
            // (an evaluator with cheap probes says how deep it wants them)
            int wanted = 64;
            for (size_t b = 0; b < B; ++b) {
                if (st[b].phase != State::kDone && !st[b].use_exact) {
                    wanted = std::min(wanted, ev.probe_depth(b));
                }
            }
            if (wanted > 0 && wanted < 64) {
                deep = std::max(deep, wanted);
            }
        }
        const int spec_depth =
            all_compacted ? deep :
            (round_loci > opt.big_round_loci)
                ? 1
                : ((round_loci < opt.tiny_round_loci)
                       ? opt.spec_depth + 3
                       : ((round_loci < opt.small_round_loci) ? opt.spec_depth + 1 : opt.spec_depth));

        const double tick_w = Tick::now();
        double tick_tree = 0.0;
        long long tick_nodes = 0;
        if (timing) std::fprintf(stderr, "[search timing]   depth rule %.1f us\n", tick_w - tick_a);
        for (size_t b = 0; b < B; ++b) {
            const ChainProblem &p = problems[b];
            State &s = st[b];
            if (b + 1 < B) {
                // (the evaluator's launch calls between two iterations leave little of this in the near caches)
                const char *next = reinterpret_cast<const char *>(&st[b + 1]);
                for (size_t off = 0; off < sizeof(State); off += 64) {
                    __builtin_prefetch(next + off);
                }
                __builtin_prefetch(&problems[b + 1]);
                __builtin_prefetch(st[b + 1].nodes.data());
                __builtin_prefetch(st[b + 1].queue.data());
            }
            if (s.phase == State::kDone) {
                continue;
            }
            s.bound_round = false;
            s.pilot_round = false;
            if (s.searching && s.pilot_left > 0) {
                // estimated counts at equally spaced penalties inside the pilot's own interval
                ProbeRequest r;
                r.problem = b;
                r.bound = true;
                r.pilot = true;
                const int qexp = bound_grid_exponent(p);
                const int pts = std::max(2, std::min(64, opt.pilot_points));
                for (int k = 1; k <= pts; ++k) {
                    const double x = snap_to_grid(s.pg + (s.pl - s.pg) * (double)k / (double)(pts + 1), qexp);
                    if (x > s.pg && x < s.pl && (r.lambdas.empty() || x > r.lambdas.back())) {
                        r.lambdas.push_back(x);
                    }
                }
                if (!r.lambdas.empty()) {
                    s.pilot_round = true;
                    probes.push_back(std::move(r));
                    probe_owner.push_back(b);
                    continue;
                }
                s.pilot_left = 0;
                s.pilot_hint = !s.pilot_evals.empty();
            }
            if (s.searching) {
                // threshold search: a few exact-arithmetic counts inside (G, L); every one of them moves
                // G or L (quartiles, plus the log-linear estimate of the crossing once both ends are real)
                const double width = s.L - s.G;
                // evaluations per round: a pass over many loci is dominated by the evaluations themselves
                // (two per round are cheapest per bit), a small one by its fixed launch + sync cost
                const int points_default = (round_loci > 30.0e6) ? opt.search_points
                                                                 : ((round_loci > 12.0e6) ? opt.search_points + 1 : opt.search_points + 3);
                const double base_hint = s.G_real ? (s.G + s.eps) : std::nan("");
                const int points = std::max(1, std::min(64, ev.bound_points(b, points_default, base_hint)));
                std::vector<double> fr;
                for (int k = 1; k <= points; ++k) {
                    fr.push_back((double)k / (double)(points + 1));
                }
                if (opt.search_interpolate && s.G_real && s.L_real && s.cG > s.cL && s.cL > 0) {
                    const double lg = std::log((double)s.cG), ll = std::log((double)s.cL);
                    const double lt = std::log((double)std::max(1LL, s.target));
                    fr.push_back(std::min(0.98, std::max(0.02, (lg - lt) / (lg - ll))));
                }
                ProbeRequest r;
                r.problem = b;
                r.bound = true;
                const int qexp = bound_grid_exponent(p);
                if (s.pilot_hint) {
                    // first certified round: penalties at which the pilot's estimate crosses a few multiples of
                    // the target (log-linear between its samples) -- two that should select more, one close to
                    // the target, one that should select less (four: one register batch of the evaluation kernel)
                    s.pilot_hint = false;
                    std::sort(s.pilot_evals.begin(), s.pilot_evals.end());
                    for (double mult : opt.pilot_levels) {
                        const double want = mult * (double)s.target;
                        double x = 0.0;
                        bool found = false;
                        for (size_t i = 0; i + 1 < s.pilot_evals.size(); ++i) {
                            const double c0 = s.pilot_evals[i].second, c1 = s.pilot_evals[i + 1].second;
                            if (c0 >= want && c1 < want) {
                                const double l0 = std::log(std::max(c0, 0.5)), l1 = std::log(std::max(c1, 0.5));
                                const double f = (l0 > l1) ? (l0 - std::log(want)) / (l0 - l1) : 0.5;
                                x = s.pilot_evals[i].first + f * (s.pilot_evals[i + 1].first - s.pilot_evals[i].first);
                                found = true;
                                break;
                            }
                        }
                        if (found) {
                            x = snap_to_grid(x, qexp);
                            if (x - s.eps > s.G && x + s.eps < s.L &&
                                std::find(r.lambdas.begin(), r.lambdas.end(), x) == r.lambdas.end()) {
                                r.lambdas.push_back(x);
                            }
                        }
                    }
                    std::sort(r.lambdas.begin(), r.lambdas.end());
                }
                if (r.lambdas.empty()) {
                    for (double f : fr) {
                        const double x = snap_to_grid(s.G + f * width, qexp);
                        if (x - s.eps > s.G && x + s.eps < s.L) {
                            r.lambdas.push_back(x);
                        }
                    }
                }
                if (!r.lambdas.empty()) {
                    s.bound_round = true;
                    probes.push_back(std::move(r));
                    probe_owner.push_back(b);
                    continue;
                }
                s.searching = false;  // (G, L) is as narrow as eps allows
                s.want_compact = opt.use_compaction && s.G_real && !s.compacted && ev.can_compact(b);
                advance_analytic(p, s);
                bracket_counts_from_evals(p, s);
            }
            if (s.want_compact) {
                // every penalty still to be asked about lies at or above G = (evaluated point) - eps, and the
                // reference's selection there lies inside the exact one at G - eps
                CompactRequest r;
                r.problem = b;
                r.lambda_base = s.G - s.eps;
                if (!ev.compact_now(r)) {
                    compacts.push_back(r);
                    compact_owner.push_back(b);
                    continue;
                }
                // the evaluator already held a compacted copy built at or below that penalty: go on at once
                s.want_compact = false;
                if (r.done) {
                    s.compacted = true;
                    ChainProblem &pm = problems[b];
                    pm.n = r.n_new;
                    pm.score_min = std::min(pm.score_min, r.score_floor);
                    s.lower_count = std::min(s.lower_count, (long long)pm.n);
                }
            }
            advance_analytic(p, s);
            if (s.phase == State::kBisect && s.iters_left <= 0) {
                // every step was decided with certainty: the final penalty is `upper`
                if (s.use_exact) {
                    s.phase = State::kFinalExact;
                } else if (s.use_spine) {
                    s.phase = State::kFinalSpine;
                } else {
                    s.phase = State::kZone;  // degenerate zone [upper, upper]: certify + materialise
                    s.lower = s.upper;
                }
            }
            s.tree.clear();
            s.tree_depth = 0;
            switch (s.phase) {
            case State::kAll:
                if (s.use_exact) {
                    ExactRequest r;
                    r.problem = b;
                    r.lambdas = {0.0};
                    r.write_solution = true;
                    exacts.push_back(r);
                    exact_owner.push_back(b);
                } else {
                    WindowRequest r;
                    r.problem = b;
                    r.lambda_lo = r.lambda_hi = 0.0;
                    windows.push_back(r);
                    window_owner.push_back(b);
                }
                break;
            case State::kLowerBracket:
            case State::kUpperBracket: {
                const double lam = (s.phase == State::kLowerBracket) ? s.lower : s.upper;
                if (s.use_exact) {
                    ExactRequest r;
                    r.problem = b;
                    r.lambdas = {lam};
                    exacts.push_back(r);
                    exact_owner.push_back(b);
                } else {
                    ProbeRequest r;
                    r.problem = b;
                    r.lambdas = {lam};
                    probes.push_back(r);
                    probe_owner.push_back(b);
                }
                break;
            }
            case State::kBisect: {
                if (!s.use_exact && s.use_spine) {
                    s.tree_depth = std::min(opt.exact_depth, s.iters_left);
                    build_tree(s.lower, s.upper, s.tree_depth, s.tree);
                    SpineRequest r;
                    r.problem = b;
                    r.lambdas = s.tree;
                    for (double &lam : r.lambdas) {
                        // a node below the certified threshold selects more than the target, and so does
                        // the threshold itself: ask about that one (it lies inside the surveyed interval)
                        lam = std::max(lam, std::min(eff_lower(s), s.upper));
                    }
                    if (s.tree_depth == s.iters_left && s.tree.size() < 64) {
                        // the last tree: let the evaluator also pick and materialise the answer.  The
                        // current upper end joins it unless it lies above the surveyed interval (the
                        // round would have to evaluate every block again): then, should no node of the
                        // tree qualify, a separate pass settles the upper end.
                        r.select_depth = s.tree_depth;
                        r.select_target = s.target;
                        r.select_has_upper = (s.upper <= eff_upper(s));
                        if (r.select_has_upper) {
                            r.lambdas.push_back(s.upper);
                        }
                    }
                    spines.push_back(r);
                    spine_owner.push_back(b);
                    break;
                }
                if (!s.use_exact && !s.has_map && s.search_rounds > 0) {
                    // the threshold search left a narrow bracket: the rounding model starts with a map
                    plan_map(p, s, false);
                    s.after_map = State::kBisect;
                    MapRequest r;
                    r.problem = b;
                    r.lambda_ref = s.req_ref;
                    r.margin = s.req_margin;
                    first_maps.push_back(maps.size());
                    maps.push_back(r);
                    map_owner.push_back(b);
                    break;
                }
                if (!s.use_exact && s.has_map) {
                    const double width = eff_upper(s) - eff_lower(s);
                    const double mid = (s.lower + s.upper) / 2.0;
                    const bool inside = (s.map_lo <= eff_lower(s) && eff_upper(s) <= s.map_hi);
                    const bool point_ok = s.point_pending && s.map_lo == mid && s.map_hi == mid;
                    // a map built for a much wider bracket carries a larger hazard margin than needed:
                    // rebuild it once that margin would shrink materially
                    const double reach = model_reach(p, p.score_min, p.score_max);
                    const double margin_now = reach + 2.0 * width * (double)s.lower_count + 2.0;
                    const bool stale = inside && (s.map_hi - s.map_lo) > 16.0 * width &&
                                       margin_now < opt.map_rebuild_ratio * s.map_margin;
                    if ((!inside && !point_ok) || stale) {
                        // the map does not cover this bracket, or the bracket shrank a lot since it was
                        // built (its hazard margin can shrink too)
                        plan_map(p, s, false);
                        s.after_map = State::kBisect;
                        MapRequest r;
                        r.problem = b;
                        r.lambda_ref = s.req_ref;
                        r.margin = s.req_margin;
                        maps.push_back(r);
                        map_owner.push_back(b);
                        break;
                    }
                }
                if (!s.use_exact && s.has_map && !s.point_pending && s.map_lo < s.map_hi && s.upper_count >= 0 &&
                    ev.probe_depth(b) == 0 &&  // (settled blocks only pay off for the evaluators that can skip them)
                    (s.survey_width < 0.0 || s.survey_width > 16.0 * (eff_upper(s) - eff_lower(s)))) {
                    // the map covers the whole bracket: when few loci can still change inside it, let the
                    // evaluator find the settled parts so that later rounds skip them
                    const long long diffs = s.lower_count - s.upper_count;
                    const long long blocks = (long long)(p.n / 8192) + 1;
                    if ((double)diffs <= opt.survey_gate * (double)blocks) {
                        WindowRequest w;
                        w.problem = b;
                        w.lambda_lo = eff_lower(s);
                        w.lambda_hi = eff_upper(s);
                        surveys.push_back(w);
                        s.survey_width = eff_upper(s) - eff_lower(s);
                    }
                }
                s.tree_depth = std::min(s.use_exact ? opt.exact_depth : spec_depth, s.iters_left);
                if (!s.use_exact && s.has_map && s.point_pending) {
                    s.tree_depth = std::min(1, s.iters_left);
                }
                if (s.use_exact) {
                    build_tree(s.lower, s.upper, s.tree_depth, s.tree);
                    ExactRequest r;
                    r.problem = b;
                    r.lambdas = s.tree;
                    exacts.push_back(r);
                    exact_owner.push_back(b);
                } else {
                    ProbeRequest r;
                    r.problem = b;
                    const double tick_t = Tick::now();
                    build_open_tree(p, s, s.tree_depth, s.iters_left, s.nodes, r.lambdas);
                    tick_tree += Tick::now() - tick_t;
                    tick_nodes += (long long)s.nodes.size();
                    if (all_compacted && s.has_map && !s.point_pending) {
                        BisectionAhead &a = r.ahead;
                        a.valid = true;
                        a.lower = s.lower;
                        a.upper = s.upper;
                        a.iters_left = s.iters_left;
                        a.target = s.target;
                        a.G_real = s.G_real;
                        a.L_real = s.L_real;
                        a.G = s.G;
                        a.L = s.L;
                        a.cG = s.cG;
                        a.cL = s.cL;
                        a.cost_ok = (p.cost_min >= 0.0);
                        a.n = (long long)p.n;
                        a.sabs = std::max(std::fabs(p.score_min), std::fabs(p.score_max));
                        a.cost_max = p.cost_max;
                        a.none_from = p.score_max + 1.0;
                        a.all_upto = p.score_min - 1.0;
                        a.open_depth = spec_depth;
                        a.depth_floor = deep_floor;
                    }
                    probes.push_back(std::move(r));
                    probe_owner.push_back(b);
                }
                break;
            }
            case State::kNeedMap: {
                MapRequest r;
                r.problem = b;
                r.lambda_ref = s.req_ref;
                r.margin = s.req_margin;
                maps.push_back(r);
                map_owner.push_back(b);
                break;
            }
            case State::kZone: {
                WindowRequest r;
                r.problem = b;
                r.lambda_lo = std::min(eff_lower(s), s.upper);  // below it every outcome is known already
                r.lambda_hi = s.upper;
                if (s.iters_left <= 0 && !s.use_exact) {
                    final_windows.push_back(windows.size());
                }
                windows.push_back(r);
                window_owner.push_back(b);
                break;
            }
            case State::kFinalSpine: {
                SpineRequest r;
                r.problem = b;
                r.lambdas = {s.upper};
                r.solution_index = 0;
                spines.push_back(r);
                spine_owner.push_back(b);
                break;
            }
            case State::kFinalExact: {
                ExactRequest r;
                r.problem = b;
                r.lambdas = {s.upper};
                r.write_solution = true;
                exacts.push_back(r);
                exact_owner.push_back(b);
                break;
            }
            case State::kDone:
                break;
            }
        }
        if (opt.align_maps && !first_maps.empty()) {
            // while another problem of the batch is still in its threshold search (or its pilot), the first maps wait:
            // the problems that ask for them sit this round out and ask again in the next one (plan_map has no effect
            // that the next call does not overwrite)
            bool searching_elsewhere = false;
            for (size_t b = 0; b < B; ++b) {
                searching_elsewhere = searching_elsewhere || (st[b].phase != State::kDone && (st[b].bound_round || st[b].pilot_round));
            }
            if (searching_elsewhere) {
                for (size_t k = first_maps.size(); k-- > 0;) {
                    maps.erase(maps.begin() + (long)first_maps[k]);
                    map_owner.erase(map_owner.begin() + (long)first_maps[k]);
                }
            }
        }
        if (opt.align_windows && !final_windows.empty() && final_windows.size() < B &&
            (!probes.empty() || !maps.empty() || !compacts.empty() || !spines.empty() || windows.size() > final_windows.size())) {
            // other problems of the batch are still at work: the finished ones' last windows sit this round out (their
            // state does not change; they ask again in the next one)
            for (size_t k = final_windows.size(); k-- > 0;) {
                windows.erase(windows.begin() + (long)final_windows[k]);
                window_owner.erase(window_owner.begin() + (long)final_windows[k]);
            }
        }
        if (probes.empty() && windows.empty() && exacts.empty() && maps.empty() && spines.empty() && compacts.empty()) {
            break;
        }
        int rc;
        const double tick_b = Tick::now();
        if (timing) std::fprintf(stderr, "[search timing]   trees %.1f us, %lld nodes\n", tick_tree, tick_nodes);
        if ((rc = ev.round_all(compacts, maps, surveys, probes, windows, spines)) != ROCCO_HIP_OK) {
            return rc;
        }
        if (timing) std::fprintf(stderr, "[search timing] iteration %d: planning %.1f us, round %.1f us\n", tick_iter++, tick_b - tick_a, Tick::now() - tick_b);
        for (size_t q = 0; q < compacts.size(); ++q) {
            const size_t b = compact_owner[q];
            State &s = st[b];
            ++s.out.passes;
            s.want_compact = false;
            if (compacts[q].done) {
                s.compacted = true;
                ChainProblem &p = problems[b];
                p.n = compacts[q].n_new;
                p.score_min = std::min(p.score_min, compacts[q].score_floor);
                s.lower_count = std::min(s.lower_count, (long long)p.n);
            }
        }
        for (size_t q = 0; q < maps.size(); ++q) {
            State &s = st[map_owner[q]];
            ++s.out.passes;
            ++s.out.maps;
            s.has_map = true;
            s.map_lo = s.req_lo;
            s.map_hi = s.req_hi;
            s.map_margin = s.req_margin;
            s.phase = s.after_map;
        }
        if (!exacts.empty() && (rc = ev.exact(exacts)) != ROCCO_HIP_OK) {
            return rc;
        }

        // ---- consume spine results (exact counts) ----
        for (size_t q = 0; q < spines.size(); ++q) {
            State &s = st[spine_owner[q]];
            const SpineRequest &r = spines[q];
            ++s.out.passes;
            if (s.phase == State::kFinalSpine) {
                s.out.selection_penalty = s.upper;
                s.out.selected_count = r.counts[0];
                s.out.path = ROCCO_HIP_PATH_SPINE;
                s.phase = State::kDone;
                continue;
            }
            size_t i = 0;
            // index of the final penalty when this was the last tree (-1: the current upper end, not evaluated)
            long long answer = (r.select_depth > 0 && !r.select_has_upper) ? -1 : (long long)r.lambdas.size() - 1;
            for (int level = 0; level < s.tree_depth; ++level) {
                ++s.out.evaluations;
                --s.iters_left;
                if (r.counts[i] > s.target) {
                    s.lower = s.tree[i];
                    i = 2 * i + 2;
                } else {
                    s.upper = s.tree[i];
                    answer = (long long)i;
                    i = 2 * i + 1;
                }
            }
            if (s.iters_left <= 0) {
                if (r.select_depth > 0 && answer >= 0 && r.selected == (int)answer) {
                    // the evaluator walked the same path and wrote that solution already
                    s.out.selection_penalty = s.upper;
                    s.out.selected_count = r.counts[(size_t)answer];
                    s.out.path = ROCCO_HIP_PATH_SPINE;
                    s.phase = State::kDone;
                } else {
                    s.phase = State::kFinalSpine;
                }
            }
        }

        // ---- consume probe results ----
        for (size_t q = 0; q < probes.size(); ++q) {
            State &s = st[probe_owner[q]];
            const ProbeRequest &r = probes[q];
            const ChainProblem &p = problems[probe_owner[q]];
            ++s.out.passes;
            if (s.pilot_round) {
                for (size_t i = 0; i < r.lambdas.size(); ++i) {
                    const double x = r.lambdas[i];
                    const double c = (double)r.results[i].count;
                    s.pilot_evals.emplace_back(x, c);
                    if (c > (double)s.target) {
                        s.pg = std::max(s.pg, x);
                    } else {
                        s.pl = std::min(s.pl, x);
                    }
                }
                if (--s.pilot_left <= 0 || !(s.pl - s.pg > 64.0 * s.eps)) {
                    s.pilot_left = 0;
                    s.pilot_hint = true;
                }
                if (search_debug) {
                    std::fprintf(stderr, "[pilot] problem %zu: crossing estimated in (%.17g, %.17g)\n", probe_owner[q], s.pg, s.pl);
                }
                continue;
            }
            if (s.bound_round) {
                ++s.search_rounds;
                for (size_t i = 0; i < r.lambdas.size(); ++i) {
                    const double x = r.lambdas[i];
                    const long long c = r.results[i].count;
                    s.evals.emplace_back(x, c);
                    if (c > s.target) {
                        if (!s.G_real || x - s.eps > s.G) {
                            s.G = x - s.eps;
                            s.cG = c;
                            s.G_real = true;
                        }
                    } else if (!s.L_real || x + s.eps < s.L) {
                        s.L = x + s.eps;
                        s.cL = c;
                        s.L_real = true;
                    }
                }
                if (search_debug) {
                    std::fprintf(stderr, "[search] problem %zu round %d: G %.17g (%lld) L %.17g (%lld) width %.3g eps %.3g\n",
                                 probe_owner[q], s.search_rounds, s.G, s.cG, s.L, s.cL, s.L - s.G, s.eps);
                }
                const long long blocks = (long long)(p.n / 8192) + 1;
                // few loci can still change between the thresholds; or their number stopped falling (the
                // count jumps by a whole run of loci at one penalty: narrowing further settles nothing)
                const long long open_now = (s.G_real && s.L_real) ? (s.cG - s.cL) : (long long)p.n;
                const bool stalled = open_now >= s.open_before && (double)open_now <= opt.survey_gate * (double)blocks;
                s.open_before = open_now;
                const bool few_left = stalled || (double)open_now <= opt.search_gate * (double)blocks;
                if (few_left || s.L - s.G <= 8.0 * s.eps || s.search_rounds >= 48) {
                    // few loci can still change between the thresholds: from here on the rounding model
                    // with its frozen blocks (rounds that skip the settled parts) is cheaper
                    s.searching = false;
                    s.want_compact = opt.use_compaction && s.G_real && !s.compacted && ev.can_compact(probe_owner[q]);
                    advance_analytic(p, s);
                    bracket_counts_from_evals(p, s);
                }
                continue;
            }
            if (s.phase == State::kLowerBracket || s.phase == State::kUpperBracket) {
                const Outcome o = classify(r.results[0], s.target);
                if (o == Outcome::kUncertain) {
                    s.use_exact = true;  // redo this bracket step exactly
                    continue;
                }
                ++s.out.evaluations;
                if (s.phase == State::kLowerBracket) {
                    if (o == Outcome::kLessEqual) {
                        s.lower -= std::max(1.0, std::fabs(s.lower));
                    } else {
                        s.phase = State::kUpperBracket;
                    }
                } else {
                    if (o == Outcome::kGreater) {
                        s.upper += std::max(1.0, std::fabs(s.upper));
                    } else {
                        s.phase = State::kBisect;
                    }
                }
                continue;
            }
            // kBisect: walk the tree while the outcomes are certain
            int i = s.nodes.empty() ? -1 : 0;
            while (i >= 0) {
                const State::TreeNode node = s.nodes[(size_t)i];
                Outcome o;
                long long analytic = 0;
                if (node.slot < 0) {
                    known_count(p, s, node.mid, &analytic);
                    o = (analytic > s.target) ? Outcome::kGreater : Outcome::kLessEqual;
                } else {
                    o = classify(r.results[(size_t)node.slot], s.target);
                }
                if (o == Outcome::kUncertain) {
                    // the uncertain node is the midpoint of the current bracket
                    const double mid = node.mid;
                    const bool is_point_here = s.has_map && s.map_lo == mid && s.map_hi == mid;
                    const bool covers = s.has_map && s.map_lo <= eff_lower(s) && s.upper <= s.map_hi;
                    if (!s.has_map) {
                        plan_map(p, s, false);  // sharpen the rounding model, then ask again
                        s.after_map = State::kBisect;
                    } else if (!is_point_here) {
                        // retry this midpoint alone with a map built exactly at it
                        s.req_ref = s.req_lo = s.req_hi = mid;
                        s.req_margin = model_reach(p, p.score_min, p.score_max) + 2.0;
                        s.point_pending = true;
                        s.phase = State::kNeedMap;
                        s.after_map = State::kBisect;
                    } else if (!covers) {
                        plan_map(p, s, true);  // the window needs a map valid across the bracket
                        s.after_map = State::kZone;
                        s.out.zone_iters = s.iters_left;
                    } else {
                        s.phase = State::kZone;
                        s.out.zone_iters = s.iters_left;
                    }
                    break;
                }
                s.point_pending = false;
                ++s.out.evaluations;
                --s.iters_left;
                if (o == Outcome::kGreater) {
                    s.lower = node.mid;
                    if (node.slot >= 0) {
                        s.lower_count = r.results[(size_t)node.slot].count + r.results[(size_t)node.slot].effect;
                    }
                    i = node.gt;
                } else {
                    s.upper = node.mid;
                    if (node.slot >= 0) {
                        s.upper_count = std::max(0LL, r.results[(size_t)node.slot].count - r.results[(size_t)node.slot].effect);
                    }
                    i = node.le;
                }
            }
        }

        // ---- consume window results ----
        for (size_t q = 0; q < windows.size(); ++q) {
            State &s = st[window_owner[q]];
            const ChainProblem &p = problems[window_owner[q]];
            const WindowResult &w = windows[q].result;
            ++s.out.passes;
            s.out.n_diff = w.n_diff;
            if (s.phase == State::kAll) {
                if (w.n_diff == 0 && !w.overflow) {
                    s.out.selected_count = w.count_lo;
                    s.out.selection_penalty = 0.0;
                    s.out.evaluations = 1;
                    s.out.path = ROCCO_HIP_PATH_CERTIFIED;
                    s.phase = State::kDone;
                } else if (!s.has_map) {
                    s.req_ref = s.req_lo = s.req_hi = 0.0;  // map at the penalty being solved
                    s.req_margin = model_reach(p, p.score_min, p.score_max) + 2.0;
                    s.phase = State::kNeedMap;
                    s.after_map = State::kAll;
                } else if (opt.use_spine) {
                    s.use_spine = true;
                    s.upper = 0.0;
                    s.out.evaluations = 1;
                    s.phase = State::kFinalSpine;
                } else {
                    s.use_exact = true;
                }
                continue;
            }
            // kZone
            bool certified = false;
            if (!w.overflow && w.n_diff == 0 && w.count_lo <= s.target) {
                // every remaining probe selects count_lo <= target loci: upper = mid each step
                replay_with_critical(p, s, -INFINITY);
                certified = true;
            } else if (!(opt.use_spine && opt.exact_penalty && s.has_map && s.map_lo <= eff_lower(s) &&
                         s.upper <= s.map_hi) &&
                       !w.overflow && w.n_diff == 1 && w.diff_adjacent && w.count_lo <= s.target &&
                       w.count_hi > s.target && !w.diffs.empty()) {
                // (only when the exact spine cannot take over: it returns the reference's penalty bit
                // for bit, this rule to within the interpolation of the crossing)
                // exactly one decision separates the two candidates; the reference keeps the one
                // within budget.  The reported penalty follows the crossing of that decision.
                const WindowDiff &d = w.diffs[0];
                double critical = s.upper;
                const double span = d.margin_lo - d.margin_hi;
                if (span > 0.0 && d.margin_lo > 0.0 && d.margin_hi <= 0.0) {
                    critical = s.lower + (d.margin_lo / span) * (s.upper - s.lower);
                }
                replay_with_critical(p, s, critical);
                certified = true;
            }
            if (certified) {
                s.out.selected_count = w.count_lo;
                s.out.selection_penalty = s.upper;
                s.out.path = ROCCO_HIP_PATH_CERTIFIED;
                s.phase = State::kDone;
            } else if (opt.use_spine && s.has_map && s.map_lo <= eff_lower(s) && s.upper <= s.map_hi) {
                // not separable by the window: finish the reference's own steps through the exact spine
                s.use_spine = true;
                s.phase = (s.iters_left > 0) ? State::kBisect : State::kFinalSpine;
            } else {
                s.use_exact = true;
                s.phase = (s.iters_left > 0) ? State::kBisect : State::kFinalExact;
            }
        }

        // ---- consume exact results ----
        for (size_t q = 0; q < exacts.size(); ++q) {
            State &s = st[exact_owner[q]];
            const ExactRequest &r = exacts[q];
            ++s.out.passes;
            if (s.phase == State::kAll) {
                s.out.selection_penalty = 0.0;
                s.out.penalized_value = r.results[0].value;
                s.out.selected_count = r.results[0].count;
                s.out.evaluations = 1;
                s.out.path = ROCCO_HIP_PATH_EXACT;
                s.phase = State::kDone;
            } else if (s.phase == State::kFinalExact) {
                s.out.selection_penalty = s.upper;
                s.out.penalized_value = r.results[0].value;
                s.out.selected_count = r.results[0].count;
                s.out.path = ROCCO_HIP_PATH_EXACT;
                s.phase = State::kDone;
            } else if (s.phase == State::kLowerBracket) {
                ++s.out.evaluations;
                if (r.results[0].count <= s.target) {
                    s.lower -= std::max(1.0, std::fabs(s.lower));
                } else {
                    s.phase = State::kUpperBracket;
                }
            } else if (s.phase == State::kUpperBracket) {
                ++s.out.evaluations;
                if (r.results[0].count > s.target) {
                    s.upper += std::max(1.0, std::fabs(s.upper));
                } else {
                    s.phase = State::kBisect;
                }
            } else if (s.phase == State::kBisect) {
                size_t i = 0;
                for (int level = 0; level < s.tree_depth; ++level) {
                    ++s.out.evaluations;
                    --s.iters_left;
                    if (r.results[i].count > s.target) {
                        s.lower = r.lambdas[i];
                        i = 2 * i + 2;
                    } else {
                        s.upper = r.lambdas[i];
                        i = 2 * i + 1;
                    }
                }
                if (s.iters_left <= 0) {
                    s.phase = State::kFinalExact;
                }
            }
        }
    }

    results.resize(B);
    std::vector<size_t> which;
    std::vector<double> lams, values;
    std::vector<long long> counts;
    for (size_t b = 0; b < B; ++b) {
        const State &s = st[b];
        if (s.out.path == ROCCO_HIP_PATH_CERTIFIED || s.out.path == ROCCO_HIP_PATH_SPINE) {
            which.push_back(b);
            lams.push_back(s.out.selection_penalty);
            counts.push_back(s.out.selected_count);
        }
    }
    if (!which.empty()) {
        const int rc = ev.penalized_values(which, lams, counts, values);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        for (size_t i = 0; i < which.size(); ++i) {
            st[which[i]].out.penalized_value = values[i];
        }
    }
    for (size_t b = 0; b < B; ++b) {
        results[b] = st[b].out;
    }
    return ROCCO_HIP_OK;
}

int solve_fixed_batch(Evaluator &ev, const std::vector<ChainProblem> &problems,
                      const std::vector<double> &lambdas, const SearchOptions &opt,
                      std::vector<CalibrationResult> &results)
{
    const size_t B = problems.size();
    results.assign(B, CalibrationResult());
    std::vector<char> need_exact(B, 0), need_spine(B, 0), pending(B, 0);
    for (size_t b = 0; b < B; ++b) {
        results[b].selection_penalty = lambdas[b];
        results[b].evaluations = 1;
        results[b].zone_iters = -1;
        if (opt.force_exact || !fast_path_applicable(problems[b])) {
            need_exact[b] = 1;
        } else {
            pending[b] = 1;
        }
    }
    int rc;
    // attempt 0: global rounding bound; attempt 1: with a binade map at the penalty itself
    for (int attempt = 0; attempt < 2; ++attempt) {
        std::vector<WindowRequest> windows;
        std::vector<size_t> owner;
        std::vector<MapRequest> maps;
        for (size_t b = 0; b < B; ++b) {
            if (!pending[b]) {
                continue;
            }
            if (attempt == 1) {
                MapRequest m;
                m.problem = b;
                m.lambda_ref = lambdas[b];
                const ChainProblem &p = problems[b];
                m.margin = model_reach(p, lambdas[b], lambdas[b]) + 2.0;
                maps.push_back(m);
            }
            WindowRequest r;
            r.problem = b;
            r.lambda_lo = r.lambda_hi = lambdas[b];
            windows.push_back(r);
            owner.push_back(b);
        }
        if (windows.empty()) {
            break;
        }
        if (!maps.empty() && (rc = ev.build_map(maps)) != ROCCO_HIP_OK) {
            return rc;
        }
        if ((rc = ev.window(windows)) != ROCCO_HIP_OK) {
            return rc;
        }
        for (size_t q = 0; q < windows.size(); ++q) {
            const size_t b = owner[q];
            const WindowResult &w = windows[q].result;
            results[b].passes += 1 + attempt;
            results[b].maps = attempt;
            results[b].n_diff = w.n_diff;
            if (w.n_diff == 0 && !w.overflow) {
                results[b].selected_count = w.count_lo;
                results[b].path = ROCCO_HIP_PATH_CERTIFIED;
                pending[b] = 0;
                if ((rc = ev.penalized_value(b, lambdas[b], w.count_lo, &results[b].penalized_value)) != ROCCO_HIP_OK) {
                    return rc;
                }
            } else if (attempt == 1) {
                pending[b] = 0;
                if (opt.use_spine) {
                    need_spine[b] = 1;
                } else {
                    need_exact[b] = 1;
                }
            }
        }
    }
    std::vector<SpineRequest> spines;
    std::vector<size_t> spine_owner;
    for (size_t b = 0; b < B; ++b) {
        if (need_spine[b]) {
            SpineRequest r;
            r.problem = b;
            r.lambdas = {lambdas[b]};
            r.solution_index = 0;
            spines.push_back(r);
            spine_owner.push_back(b);
        }
    }
    if (!spines.empty()) {
        if ((rc = ev.spine(spines)) != ROCCO_HIP_OK) {
            return rc;
        }
        for (size_t q = 0; q < spines.size(); ++q) {
            const size_t b = spine_owner[q];
            results[b].selected_count = spines[q].counts[0];
            results[b].path = ROCCO_HIP_PATH_SPINE;
            ++results[b].passes;
            if ((rc = ev.penalized_value(b, lambdas[b], results[b].selected_count, &results[b].penalized_value)) != ROCCO_HIP_OK) {
                return rc;
            }
        }
    }
    std::vector<ExactRequest> exacts;
    std::vector<size_t> exact_owner;
    for (size_t b = 0; b < B; ++b) {
        if (need_exact[b]) {
            ExactRequest r;
            r.problem = b;
            r.lambdas = {lambdas[b]};
            r.write_solution = true;
            exacts.push_back(r);
            exact_owner.push_back(b);
        }
    }
    if (!exacts.empty()) {
        if ((rc = ev.exact(exacts)) != ROCCO_HIP_OK) {
            return rc;
        }
        for (size_t q = 0; q < exacts.size(); ++q) {
            const size_t b = exact_owner[q];
            results[b].penalized_value = exacts[q].results[0].value;
            results[b].selected_count = exacts[q].results[0].count;
            results[b].path = ROCCO_HIP_PATH_EXACT;
            ++results[b].passes;
        }
    }
    return ROCCO_HIP_OK;
}

}  // namespace rocco
