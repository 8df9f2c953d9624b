// rocco_amd/csrc/budget.hip -- device-side Evaluator for the host search logic (search.cpp) and the
// two solve entry points built on it.
//
// search.cpp replays the reference's calibration (rocco/dp.py:89-164) and asks for three kinds of
// device work; this file turns each request batch into launches:
//   probe  -> one fast round (chain_fast.hip) over every (chromosome, penalty) of the batch
//   window -> one fast round in window mode (also materialises fill(LO) into the solution buffers)
//   exact  -> the sequential emulation kernel (chain_exact.hip), one wavefront per chromosome
#include "budget.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "chain.h"
#include "model_chain.h"
#include "chain_fast.h"
#include "lean.h"
#include "search.h"

namespace rocco {

namespace {

struct DevProblem {
    const double *scores = nullptr;
    const double *costs = nullptr;
    double gamma = 0.0;
    size_t n = 0;
    uint8_t *solution = nullptr;
    int qexp = 0;
    double cmax = 0.0, sabs = 0.0;
    uint8_t *emap = nullptr;  // device binade map (owned by the solver's map buffer), null = none
    // frozen blocks (active-set rounds): device summaries + host copy of the flags
    FrozenArrays frz = {};
    bool frz_valid = false;
    double frz_lo = 0.0, frz_hi = 0.0;  // the (latest) surveyed bracket
    std::vector<uint8_t> frz_flags;
    std::vector<int> active_blocks;
    double smin = 0.0, smax = 0.0;
    // compaction (lean.h): once `compacted`, scores / n / solution above are the compacted problem's and
    // these are the caller's
    bool compacted = false;
    bool solution_in_orig = false;  // the exact evaluator wrote the caller's buffer directly
    bool scattered = false;
    const double *orig_scores = nullptr;
    size_t orig_n = 0;
    uint8_t *orig_solution = nullptr;
    const int *lean_orig = nullptr;  // original locus of every compacted locus (-1: separator)
    int map_version = 0;             // bumped whenever `emap` is (re)built
    int wcap_version = -1;           // map version the tolerance cap of the lean model kernel was computed for
};

// Levels of one problem (lean.h): level 0 = the caller's array; every deeper level holds the loci selected
// at its base penalty in exact arithmetic (plus separators) and serves every evaluation at or above it.
struct LeanLevel {
    const double *s = nullptr;
    const int *orig = nullptr;
    long long m = 0;
    double base = -INFINITY;
    double sep = 0.0;       // score of the level's separator loci
    bool has_eval = false;  // pts / child_len / bits / tile_off describe the latest evaluation on this level
    std::vector<double> pts;
    std::vector<long long> child_len;
    unsigned *bits = nullptr;
    unsigned *tile_off = nullptr;
    int cap_points = 0;
    size_t pool_mark = 0;  // the problem's pool offset before this level was created
};

struct LeanState {
    std::vector<LeanLevel> levels;
    size_t pool_begin = 0, pool_end = 0, pool_at = 0;
};

struct LeanReq {
    size_t problem = 0;
    std::vector<double> lambdas;
    ProbeRequest *probe = nullptr;
    CompactRequest *comp = nullptr;  // final compaction at lambdas[0]
    bool pilot = false;              // estimates from a sample of the tiles of level 0
    bool model = false;              // rounding-model counts on the compacted problem (lean_model_kernel)
    double pilot_scale = 1.0;
    int result_begin = 0;
    // final compaction
    double *out_s = nullptr;
    int *out_orig = nullptr;
    long long capacity = 0;
    double sep = 0.0;
    size_t mark = 0;
};

// true when -lambda lies exactly half-way between two points of a grid u = 2^(e-52) that a clean
// chunk can use (u >= q = 2^qexp)
bool lambda_ties_some_grid(double lambda, int qexp)
{
    if (lambda == 0.0) {
        return false;
    }
    for (int e = qexp + 52; e <= 62; ++e) {
        const double magic_u = std::ldexp(1.5, e);
        volatile double t = -lambda + magic_u;
        const double r = t - magic_u;
        if (std::fabs(-lambda - r) == std::ldexp(1.0, e - 53)) {
            return true;
        }
    }
    return false;
}

int grid_exponent(double cmax, double lo, double hi)
{
    const double r = cmax + (hi - lo) + 2.0;
    return (int)std::ceil(std::log2(8.0 * r)) - 52;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

std::atomic<long long> g_model_chain_counters[6];

class HipEvaluator : public Evaluator {
public:
    HipEvaluator(rocco_hip_solver *solver, hipStream_t stream) : solver_(solver), stream_(stream) {}

    // Environment switches are read once per evaluator (= per solve call: tests and A/B scripts change them between calls);
    // getenv walks the whole environment, and the search asks some of these per chromosome and round.
    mutable std::vector<std::pair<const char *, const char *>> env_cache_;
    const char *env(const char *name) const
    {
        for (const auto &e : env_cache_) {
            if (e.first == name || std::strcmp(e.first, name) == 0) {
                return e.second;
            }
        }
        const char *value = std::getenv(name);
        env_cache_.emplace_back(name, value);
        return value;
    }

    std::vector<DevProblem> probs;

    struct RoundTask {
        size_t problem = 0;
        bool window = false;
        bool map = false;
        bool record = false;
        bool survey = false;
        bool bound = false;
        bool use_frozen = false;
        int solution_index = -1;
        int select_depth = 0;
        bool select_has_upper = true;
        long long select_target = 0;
        double margin = 0.0;
        std::vector<double> lambdas;
        ProbeRequest *probe = nullptr;
        WindowRequest *win = nullptr;
        SpineRequest *spine = nullptr;
    };

    void add_probe_tasks(std::vector<ProbeRequest> &reqs, std::vector<RoundTask> &tasks)
    {
        for (ProbeRequest &r : reqs) {
            if (lean_takes(*this, r)) {
                continue;  // answered by the lean evaluation of this iteration (lean_submit)
            }
            r.results.assign(r.lambdas.size(), ProbeResult());
            if (r.lambdas.empty()) {
                continue;
            }
            RoundTask t;
            t.problem = r.problem;
            t.window = false;
            t.bound = r.bound;
            t.lambdas = r.lambdas;
            t.probe = &r;
            tasks.push_back(t);
        }
    }

    int probe(std::vector<ProbeRequest> &reqs) override
    {
        std::vector<CompactRequest> none;
        int rc;
        if ((rc = lean_submit(none, reqs)) != ROCCO_HIP_OK) return rc;
        std::vector<RoundTask> tasks;
        add_probe_tasks(reqs, tasks);
        if (tasks.empty()) {
            if (lean_wait_needed()) {
                ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
            }
        } else if ((rc = run_round(tasks)) != ROCCO_HIP_OK) {
            return rc;
        }
        if ((rc = lean_consume()) != ROCCO_HIP_OK) return rc;
        return model_fallback();
    }

    std::vector<char> window_answered_;  // (round_all: windows answered from a solution a chain wrote -- no task for them)

    void add_window_tasks(std::vector<WindowRequest> &reqs, std::vector<RoundTask> &tasks)
    {
        for (size_t i = 0; i < reqs.size(); ++i) {
            WindowRequest &r = reqs[i];
            if (i < window_answered_.size() && window_answered_[i]) {
                continue;
            }
            RoundTask t;
            t.problem = r.problem;
            t.window = true;
            t.lambdas = {r.lambda_lo, r.lambda_hi};
            t.win = &r;
            tasks.push_back(t);
        }
    }

    int window(std::vector<WindowRequest> &reqs) override
    {
        window_answered_.clear();
        std::vector<RoundTask> tasks;
        add_window_tasks(reqs, tasks);
        return run_round(tasks);
    }

    int add_survey_tasks(std::vector<WindowRequest> &reqs, std::vector<RoundTask> &tasks)
    {
        if (solver_->active_set == 0 || reqs.empty()) {
            return ROCCO_HIP_OK;
        }
        int rc;
        if (!frozen_allocated_) {
            size_t total_blocks = 0;
            for (const DevProblem &p : probs) {
                total_blocks += align_up((p.n + kFastBlockLoci - 1) / kFastBlockLoci, 32);
            }
            const size_t per_block = 8 * 8 + 4 * 4 + 3;  // doubles, ints/unsigneds, bytes
            if ((rc = solver_->dev_frozen.reserve(total_blocks * per_block + 4096)) != ROCCO_HIP_OK) return rc;
            char *base = (char *)solver_->dev_frozen.ptr;
            double *dbl = (double *)base;
            int *i32 = (int *)(base + total_blocks * 64);
            uint8_t *u8 = (uint8_t *)(base + total_blocks * 80);
            size_t off = 0;
            for (DevProblem &p : probs) {
                const size_t nb = align_up((p.n + kFastBlockLoci - 1) / kFastBlockLoci, 32);
                p.frz.B = dbl + off;
                p.frz.gain_lo = dbl + total_blocks + off;
                p.frz.gain_hi = dbl + 2 * total_blocks + off;
                p.frz.lam_lo = dbl + 3 * total_blocks + off;
                p.frz.lam_hi = dbl + 4 * total_blocks + off;
                p.frz.gx_lo = dbl + 5 * total_blocks + off;
                p.frz.mg = dbl + 6 * total_blocks + off;
                p.frz.cprev = dbl + 7 * total_blocks + off;
                p.frz.m = i32 + off;
                p.frz.lc = i32 + total_blocks + off;
                p.frz.pend = (unsigned *)(i32 + 2 * total_blocks + off);
                p.frz.base = (unsigned *)(i32 + 3 * total_blocks + off);
                p.frz.flag = u8 + off;
                p.frz.e = (int8_t *)(u8 + total_blocks + off);
                p.frz.fv = u8 + 2 * total_blocks + off;
                off += nb;
            }
            frozen_allocated_ = true;
        }
        for (WindowRequest &r : reqs) {
            if (probs[r.problem].emap == nullptr) {
                continue;
            }
            RoundTask t;
            t.problem = r.problem;
            t.window = true;
            t.survey = true;
            t.lambdas = {r.lambda_lo, r.lambda_hi};
            t.win = &r;
            tasks.push_back(t);
        }
        return ROCCO_HIP_OK;
    }

    int survey(std::vector<WindowRequest> &reqs) override
    {
        std::vector<RoundTask> tasks;
        const int rc = add_survey_tasks(reqs, tasks);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        return run_round(tasks);
    }

    double work_fraction(size_t problem) const override
    {
        const DevProblem &p = probs[problem];
        if (!p.frz_valid || solver_->active_set == 0 || p.frz_flags.empty()) {
            return 1.0;
        }
        return (double)p.active_blocks.size() / (double)p.frz_flags.size();
    }

    int add_map_tasks(std::vector<MapRequest> &reqs, std::vector<RoundTask> &tasks)
    {
        if (reqs.empty()) {
            return ROCCO_HIP_OK;
        }
        // carve (once) a map region per problem out of the solver's map buffer
        if (!maps_allocated_) {
            size_t total = 0;
            for (const DevProblem &p : probs) {
                total += align_up((p.n + kChunk - 1) / kChunk, 256);
            }
            int rc;
            if ((rc = solver_->dev_maps.reserve(total + 256)) != ROCCO_HIP_OK) return rc;
            size_t off = 0;
            map_ptrs_.resize(probs.size());
            for (size_t b = 0; b < probs.size(); ++b) {
                map_ptrs_[b] = (uint8_t *)solver_->dev_maps.ptr + off;
                off += align_up((probs[b].n + kChunk - 1) / kChunk, 256);
            }
            maps_allocated_ = true;
        }
        // the first map of a compacted problem: by the lean kernels (lean.h: lean_map_kernel), queued here and not waited for
        std::vector<MapRequest *> lean_maps;
        for (MapRequest &r : reqs) {
            if (lean_map_takes(r.problem)) {
                lean_maps.push_back(&r);
                continue;
            }
            RoundTask t;
            t.problem = r.problem;
            t.map = true;
            t.margin = r.margin;
            t.lambdas = {r.lambda_ref};
            tasks.push_back(t);
        }
        return lean_map_enqueue(lean_maps);
    }

    bool force_lean_map_ = false;  // test entry: any array without switch-cost vector and without a map
    long long lean_maps_built = 0;
    const unsigned *lean_map_error_ = nullptr;  // pinned word a lean map round reports to (checked at the next waits)

    bool lean_map_takes(size_t problem) const
    {
        const char *flag = env("ROCCO_HIP_LEAN_MAP");
        const bool enabled = (flag != nullptr) ? std::atoi(flag) != 0 : true;
        const DevProblem &p = probs[problem];
        return enabled && solver_->lean != 0 && p.costs == nullptr && p.emap == nullptr && p.n >= 2 &&
               (force_lean_map_ || (lean_ready_ && p.compacted && p.lean_orig != nullptr));
    }

    int lean_map_check()
    {
        if (lean_map_error_ != nullptr && *lean_map_error_ != 0u) {
            solver_->lean_look_dirty = 1;
            set_last_error("lean map: a tile waited for its predecessor beyond the spin limit");
            return ROCCO_HIP_EHIP;
        }
        return ROCCO_HIP_OK;
    }

    int lean_map_enqueue(const std::vector<MapRequest *> &reqs)
    {
        if (reqs.empty()) {
            return ROCCO_HIP_OK;
        }
        int rc;
        if ((rc = lean_prepare()) != ROCCO_HIP_OK) return rc;
        const size_t T = reqs.size();
        std::vector<LeanTask> tasks(T);
        std::vector<double> points(T);
        std::vector<LeanMapCodeTask> codes(T);
        int units = 0;
        for (size_t i = 0; i < T; ++i) {
            const MapRequest &r = *reqs[i];
            const DevProblem &p = probs[r.problem];
            LeanTask &t = tasks[i];
            t.s = p.scores;
            t.m = (long long)p.n;
            t.c_raw = p.gamma;
            t.magic = std::ldexp(1.5, 52 + p.qexp);
            t.big = std::ldexp(1.0, 50 + p.qexp);
            t.n_tiles = (int)((p.n + kLeanTile - 1) / kLeanTile);
            t.n_points = 1;
            t.n_groups = 1;
            t.unit_begin = units;
            t.point_begin = (int)i;
            t.rec_begin = units;
            t.bits_begin = 0;
            t.off_begin = 0;
            t.result_begin = (int)i;
            t.tile_stride = 1;
            t.independent = 0;
            t.store = 0;
            t.emap = nullptr;
            t.wcap = nullptr;
            t.clean_chunks = nullptr;
            t.cmax = t.sabs = 0.0;
            t.qexp = p.qexp;
            t.batch = 1;
            points[i] = r.lambda_ref;
            units += t.n_tiles;
        }
        const size_t b_tasks = align_up(T * sizeof(LeanTask), 256);
        const size_t b_points = align_up(T * sizeof(double), 256);
        const size_t b_codes = align_up(T * sizeof(LeanMapCodeTask), 256);
        const size_t up_bytes = b_tasks + b_points + b_codes;
        const size_t b_results = align_up(T * sizeof(LeanResult), 256);
        const size_t b_gc = align_up((size_t)units * kLeanThreads * sizeof(double), 256);
        const size_t b_gb = align_up((size_t)units * sizeof(double), 256);
        if ((rc = solver_->dev_map.reserve(up_bytes + b_results + b_gc + b_gb + 256)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_map_stage.reserve(up_bytes + 256)) != ROCCO_HIP_OK) return rc;
        char *dv = (char *)solver_->dev_map.ptr;
        char *h = (char *)solver_->host_map_stage.ptr;
        double *gain_chunk = (double *)(dv + up_bytes + b_results);
        double *gain_block = (double *)(dv + up_bytes + b_results + b_gc);
        for (size_t i = 0; i < T; ++i) {
            LeanMapCodeTask &c = codes[i];
            c.gain_chunk = gain_chunk + (size_t)tasks[i].rec_begin * kLeanThreads;
            c.gain_block = gain_block + tasks[i].rec_begin;
            c.emap = map_ptrs_[reqs[i]->problem];
            c.m = tasks[i].m;
            c.margin = reqs[i]->margin;
            c.n_tiles = tasks[i].n_tiles;
            c.block_begin = tasks[i].unit_begin;
        }
        std::memcpy(h, tasks.data(), T * sizeof(LeanTask));
        std::memcpy(h + b_tasks, points.data(), T * sizeof(double));
        std::memcpy(h + b_tasks + b_points, codes.data(), T * sizeof(LeanMapCodeTask));
        unsigned *error_host = (unsigned *)(h + up_bytes);
        *error_host = 0u;
        lean_map_error_ = error_host;
        ROCCO_HIP_TRY(hipMemcpyAsync(dv, h, up_bytes, hipMemcpyHostToDevice, stream_));
        // round scratch as the lean rounds keep it (lean_enqueue)
        const size_t b_look = align_up(256 + (size_t)units * 4 * sizeof(unsigned long long), 256);
        {
            const void *old_ptr = solver_->dev_lean_look.ptr;
            const size_t old_bytes = solver_->dev_lean_look.bytes;
            if ((rc = solver_->dev_lean_look.reserve(b_look)) != ROCCO_HIP_OK) return rc;
            if (solver_->lean_look_dirty != 0 || solver_->dev_lean_look.ptr != old_ptr || solver_->dev_lean_look.bytes != old_bytes) {
                ROCCO_HIP_TRY(hipMemsetAsync(solver_->dev_lean_look.ptr, 0xFF, solver_->dev_lean_look.bytes, stream_));
                ROCCO_HIP_TRY(hipMemsetAsync((char *)solver_->dev_lean_look.ptr + 128, 0, 4, stream_));
                solver_->lean_look_dirty = 0;
            }
        }
        if ((rc = solver_->dev_lean_round.reserve(align_up((size_t)units * sizeof(LeanTileRec), 256) + 256)) != ROCCO_HIP_OK) return rc;
        char *look = (char *)solver_->dev_lean_look.ptr;
        LeanLaunch L;
        L.tasks = (const LeanTask *)dv;
        L.n_tasks = (int)T;
        L.n_units = units;
        L.points = (const double *)(dv + b_tasks);
        L.ticket = (unsigned *)look;
        L.look = (unsigned long long *)(look + 256);
        L.recs = (LeanTileRec *)solver_->dev_lean_round.ptr;
        L.bits = (unsigned *)solver_->dev_lean_pool.ptr;
        L.tile_off = (unsigned *)solver_->dev_lean_pool.ptr;
        L.results = (LeanResult *)(dv + up_bytes);
        L.error = (unsigned *)(look + 128);
        L.error_out = error_host;
        L.self_reset = 1;
        L.pad = 0;
        L.ctl = nullptr;
        LeanMapOut out;
        out.gain_chunk = gain_chunk;
        out.gain_block = gain_block;
        solver_->lean_look_dirty = 1;
        if ((rc = launch_lean_map(L, out, stream_)) != ROCCO_HIP_OK) return rc;
        if ((rc = launch_lean_finish(L, (int)T, stream_)) != ROCCO_HIP_OK) return rc;
        solver_->lean_look_dirty = 0;
        if ((rc = launch_lean_mapcode((const LeanMapCodeTask *)(dv + b_tasks + b_points), (int)T, units, stream_)) != ROCCO_HIP_OK) return rc;
        lean_maps_built += (long long)T;
        if (env("ROCCO_HIP_DEBUG") != nullptr) {
            std::fprintf(stderr, "[lean map] %zu problems, %d tiles\n", T, units);
        }
        return ROCCO_HIP_OK;
    }

    void adopt_maps(std::vector<MapRequest> &reqs)
    {
        for (MapRequest &r : reqs) {
            probs[r.problem].emap = map_ptrs_[r.problem];
            ++probs[r.problem].map_version;
        }
    }

    int build_map(std::vector<MapRequest> &reqs) override
    {
        std::vector<RoundTask> tasks;
        int rc = add_map_tasks(reqs, tasks);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        rc = run_round(tasks);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        adopt_maps(reqs);
        return ROCCO_HIP_OK;
    }

    int add_spine_tasks(std::vector<SpineRequest> &reqs, std::vector<RoundTask> &tasks)
    {
        for (SpineRequest &r : reqs) {
            if (r.lambdas.empty() || r.lambdas.size() > 64 || probs[r.problem].emap == nullptr) {
                return ROCCO_HIP_EINVAL;
            }
            RoundTask t;
            t.problem = r.problem;
            t.record = true;
            t.solution_index = r.solution_index;
            t.select_depth = r.select_depth;
            t.select_has_upper = r.select_has_upper;
            t.select_target = r.select_target;
            t.lambdas = r.lambdas;
            t.spine = &r;
            tasks.push_back(t);
        }
        return ROCCO_HIP_OK;
    }

    int spine(std::vector<SpineRequest> &reqs) override
    {
        std::vector<RoundTask> tasks;
        const int rc = add_spine_tasks(reqs, tasks);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        return run_round(tasks);
    }

    // every request of one search iteration in a single device pass
    int round(std::vector<MapRequest> &maps, std::vector<WindowRequest> &surveys, std::vector<ProbeRequest> &probes,
              std::vector<WindowRequest> &windows, std::vector<SpineRequest> &spines) override
    {
        std::vector<RoundTask> tasks;
        int rc;
        if ((rc = add_map_tasks(maps, tasks)) != ROCCO_HIP_OK) return rc;
        if ((rc = add_survey_tasks(surveys, tasks)) != ROCCO_HIP_OK) return rc;
        add_probe_tasks(probes, tasks);
        add_window_tasks(windows, tasks);
        if ((rc = add_spine_tasks(spines, tasks)) != ROCCO_HIP_OK) return rc;
        if ((rc = run_round(tasks)) != ROCCO_HIP_OK) return rc;
        adopt_maps(maps);
        return ROCCO_HIP_OK;
    }

    int exact(std::vector<ExactRequest> &reqs) override
    {
        const size_t R = reqs.size();
        if (R == 0) {
            return ROCCO_HIP_OK;
        }
        size_t words_total = 0;
        for (const ExactRequest &r : reqs) {
            if (r.lambdas.empty() || r.lambdas.size() > 64) {
                return ROCCO_HIP_EINVAL;
            }
            if (r.write_solution) {
                const DevProblem &pw = probs[r.problem];
                words_total += ((pw.compacted ? pw.orig_n : pw.n) + 30) / 32 + 1;
            }
        }
        int rc;
        if ((rc = solver_->dev_tasks.reserve(R * sizeof(ExactTask))) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->dev_params.reserve(R * 64 * sizeof(double))) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->dev_results.reserve(R * 64 * (sizeof(double) + sizeof(long long)))) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->dev_bits.reserve(words_total * sizeof(unsigned long long) + 8)) != ROCCO_HIP_OK) return rc;
        if ((rc = stage_wait()) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_stage.reserve(R * sizeof(ExactTask) + R * 64 * sizeof(double))) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_back.reserve(R * 64 * (sizeof(double) + sizeof(long long)))) != ROCCO_HIP_OK) return rc;

        ExactTask *h_tasks = (ExactTask *)solver_->host_stage.ptr;
        double *h_lams = (double *)((char *)solver_->host_stage.ptr + R * sizeof(ExactTask));
        double *d_lams = (double *)solver_->dev_params.ptr;
        double *d_vals = (double *)solver_->dev_results.ptr;
        long long *d_cnts = (long long *)((char *)solver_->dev_results.ptr + R * 64 * sizeof(double));
        unsigned long long *d_words = (unsigned long long *)solver_->dev_bits.ptr;
        size_t word_off = 0;
        for (size_t r = 0; r < R; ++r) {
            DevProblem &pc = probs[reqs[r].problem];
            // the exact evaluator is the last resort: always on the caller's own arrays
            DevProblem p = pc;
            if (pc.compacted) {
                p.scores = pc.orig_scores;
                p.n = pc.orig_n;
                p.solution = pc.orig_solution;
                if (reqs[r].write_solution) {
                    pc.solution_in_orig = true;
                }
            }
            ExactTask &e = h_tasks[r];
            e.scores = p.scores;
            e.switch_costs = p.costs;
            e.gamma = p.gamma;
            e.n = (long long)p.n;
            e.lambdas = d_lams + r * 64;
            e.n_lambda = (int)reqs[r].lambdas.size();
            e.record_lane = 0;
            e.values_out = d_vals + r * 64;
            e.counts_out = d_cnts + r * 64;
            if (reqs[r].write_solution) {
                e.decision_words = d_words + word_off;
                e.solution = p.solution;
                word_off += (p.n + 30) / 32 + 1;
            } else {
                e.decision_words = nullptr;
                e.solution = nullptr;
            }
            for (size_t l = 0; l < 64; ++l) {
                h_lams[r * 64 + l] = reqs[r].lambdas[l < reqs[r].lambdas.size() ? l : 0];
            }
        }
        ROCCO_HIP_TRY(hipMemcpyAsync(solver_->dev_tasks.ptr, h_tasks, R * sizeof(ExactTask), hipMemcpyHostToDevice, stream_));
        ROCCO_HIP_TRY(hipMemcpyAsync(d_lams, h_lams, R * 64 * sizeof(double), hipMemcpyHostToDevice, stream_));
        if ((rc = launch_chain_exact((const ExactTask *)solver_->dev_tasks.ptr, (int)R, stream_)) != ROCCO_HIP_OK) {
            return rc;
        }
        double *b_vals = (double *)solver_->host_back.ptr;
        long long *b_cnts = (long long *)((char *)solver_->host_back.ptr + R * 64 * sizeof(double));
        ROCCO_HIP_TRY(hipMemcpyAsync(b_vals, d_vals, R * 64 * sizeof(double), hipMemcpyDeviceToHost, stream_));
        ROCCO_HIP_TRY(hipMemcpyAsync(b_cnts, d_cnts, R * 64 * sizeof(long long), hipMemcpyDeviceToHost, stream_));
        ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
        for (size_t r = 0; r < R; ++r) {
            const size_t Lr = reqs[r].lambdas.size();
            reqs[r].results.resize(Lr);
            for (size_t l = 0; l < Lr; ++l) {
                reqs[r].results[l].value = b_vals[r * 64 + l];
                reqs[r].results[l].count = b_cnts[r * 64 + l];
            }
        }
        return ROCCO_HIP_OK;
    }

    // compacted problems: copy the solution back into the caller's buffer (zero outside the kept loci)
    int scatter_solution(size_t problem)
    {
        DevProblem &p = probs[problem];
        if (!p.compacted || p.solution_in_orig || p.scattered) {
            return ROCCO_HIP_OK;
        }
        ROCCO_HIP_TRY(hipMemsetAsync(p.orig_solution, 0, p.orig_n, stream_));
        const int rc = launch_lean_scatter(p.solution, p.lean_orig, (long long)p.n, p.orig_solution, stream_);
        p.scattered = true;
        return rc;
    }

    // every compacted problem of the batch in two launches (zero the callers' buffers, scatter the kept loci)
    // `only`: just these problems (nullptr: every compacted problem not scattered yet)
    int scatter_all(const std::vector<size_t> *only = nullptr)
    {
        std::vector<LeanScatterTask> tasks;
        int zero_blocks = 0, scatter_blocks = 0;
        for (size_t b = 0; b < probs.size(); ++b) {
            DevProblem &p = probs[b];
            if (!p.compacted || p.solution_in_orig || p.scattered) {
                continue;
            }
            if (only != nullptr && std::find(only->begin(), only->end(), b) == only->end()) {
                continue;
            }
            LeanScatterTask t;
            t.level_solution = p.solution;
            t.orig = p.lean_orig;
            t.m = (long long)p.n;
            t.full = p.orig_solution;
            t.n = (long long)p.orig_n;
            t.zero_begin = zero_blocks;
            t.scatter_begin = scatter_blocks;
            zero_blocks += (int)((p.orig_n + 16383) / 16384);
            scatter_blocks += (int)((p.n + 255) / 256);
            tasks.push_back(t);
            p.scattered = true;
        }
        if (tasks.empty()) {
            return ROCCO_HIP_OK;
        }
        const size_t bytes = tasks.size() * sizeof(LeanScatterTask);
        int rc;
        if ((rc = solver_->dev_lean_desc.reserve(bytes + 256)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_lean_stage.reserve(bytes + 256)) != ROCCO_HIP_OK) return rc;
        std::memcpy(solver_->host_lean_stage.ptr, tasks.data(), bytes);
        ROCCO_HIP_TRY(hipMemcpyAsync(solver_->dev_lean_desc.ptr, solver_->host_lean_stage.ptr, bytes, hipMemcpyHostToDevice, stream_));
        return launch_lean_scatter_batch((const LeanScatterTask *)solver_->dev_lean_desc.ptr, (int)tasks.size(), zero_blocks,
                                         scatter_blocks, stream_);
    }

    // the caller's view of a problem (what the objective is evaluated on)
    DevProblem caller_view(size_t problem) const
    {
        DevProblem p = probs[problem];
        if (p.compacted) {
            p.scores = p.orig_scores;
            p.n = p.orig_n;
            p.solution = p.orig_solution;
        }
        return p;
    }

    int penalized_value(size_t problem, double lambda, long long count, double *value_out) override
    {
        int rc;
        if ((rc = scatter_solution(problem)) != ROCCO_HIP_OK) return rc;
        const DevProblem p = caller_view(problem);
        if ((rc = solver_->dev_misc.reserve(objective_scratch_bytes(p.n))) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_back.reserve(64)) != ROCCO_HIP_OK) return rc;
        double *back = (double *)solver_->host_back.ptr;
        rc = launch_objective(p.solution, p.scores, p.costs, p.gamma, p.n, solver_->dev_misc.ptr, back, stream_);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        *value_out = -(*back) - lambda * (double)count;
        return ROCCO_HIP_OK;
    }

    // Objective sums fetched ahead (prefetch_objectives): problem -> sum of the solution that a window wrote, valid until
    // another round touches the problem's solution.
    std::vector<char> objective_ready_;
    std::vector<double> objective_sum_;
    std::vector<size_t> objective_pending_;  // problems whose sums are on their way to pinned memory
    long long objective_prefetches = 0, objective_prefetch_hits = 0;

    void objective_invalidate(size_t problem)
    {
        if (problem < objective_ready_.size() && objective_ready_[problem]) {
            objective_ready_[problem] = 0;
        }
        probs[problem].scattered = false;  // (a solution scattered ahead is scattered again once it is final)
        if (problem < written_.size()) {
            written_[problem].valid = false;  // (a solution a chain wrote is about to be rewritten)
        }
    }

    // Behind a round of windows that each write the solution of ONE penalty (the calibration's last step when every
    // bisection step was decided), before that round is waited for: scatter those solutions and reduce their objectives,
    // so that the one wait brings both.  Undone per problem by objective_collect when its window did not certify.
    int prefetch_objectives(const std::vector<size_t> &which)
    {
        for (size_t b : objective_pending_) {
            probs[b].scattered = false;  // (sums queued earlier and never collected: their scatter does not count either)
        }
        objective_pending_.clear();
        objective_behind_chain_ = false;
        const size_t W = which.size();
        if (W == 0) {
            return ROCCO_HIP_OK;
        }
        int rc;
        // Everything here goes through buffers of its own (dev_objective, host_objective): behind a chain these launches sit
        // in the stream for a while, and the rounds' staging buffers may be written again meanwhile.
        std::vector<LeanScatterTask> scatters;
        int zero_blocks = 0, scatter_blocks = 0;
        for (size_t b : which) {
            DevProblem &p = probs[b];
            if (!p.compacted || p.solution_in_orig || p.scattered) {
                continue;
            }
            LeanScatterTask t;
            t.level_solution = p.solution;
            t.orig = p.lean_orig;
            t.m = (long long)p.n;
            t.full = p.orig_solution;
            t.n = (long long)p.orig_n;
            t.zero_begin = zero_blocks;
            t.scatter_begin = scatter_blocks;
            zero_blocks += (int)((p.orig_n + 16383) / 16384);
            scatter_blocks += (int)((p.n + 255) / 256);
            scatters.push_back(t);
            p.scattered = true;
        }
        std::vector<ObjectiveTask> tasks(W);
        long long tiles = 0;
        for (size_t i = 0; i < W; ++i) {
            const DevProblem &p = probs[which[i]];
            const bool on_level = p.compacted && !p.solution_in_orig;
            tasks[i].solution = on_level ? p.solution : (p.compacted ? p.orig_solution : p.solution);
            tasks[i].scores = on_level ? p.scores : (p.compacted ? p.orig_scores : p.scores);
            tasks[i].switch_costs = p.costs;
            tasks[i].gamma = p.gamma;
            tasks[i].n = (long long)(on_level ? p.n : (p.compacted ? p.orig_n : p.n));
            tasks[i].tile_begin = tiles;
            tiles += objective_tiles((size_t)tasks[i].n);
        }
        const size_t b_scatter = align_up(scatters.size() * sizeof(LeanScatterTask), 256);
        const size_t b_tasks = align_up(W * sizeof(ObjectiveTask), 256);
        const size_t b_part = align_up((size_t)(2 * tiles + 2) * sizeof(double), 256);
        const size_t b_out = align_up(W * sizeof(double), 256);
        if ((rc = solver_->dev_objective.reserve(b_scatter + b_tasks + b_part + b_out)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_objective.reserve(b_scatter + b_tasks + b_out)) != ROCCO_HIP_OK) return rc;
        char *dv = (char *)solver_->dev_objective.ptr;
        char *ho = (char *)solver_->host_objective.ptr;
        // The descriptors of an earlier prefetch may still be waiting in the stream for their copy to the device (behind a
        // chain the host runs ahead of the stream: it resumes when the last director publishes `finished`, before that
        // prefetch's copy has run): the pinned block is written again only once that copy has been done.
        if (objective_staged_pending_) {
            objective_staged_pending_ = false;
            ROCCO_HIP_TRY(hipEventSynchronize(objective_staged_event_));
        }
        if (!scatters.empty()) std::memcpy(ho, scatters.data(), scatters.size() * sizeof(LeanScatterTask));
        std::memcpy(ho + b_scatter, tasks.data(), W * sizeof(ObjectiveTask));
        ROCCO_HIP_TRY(hipMemcpyAsync(dv, ho, b_scatter + b_tasks, hipMemcpyHostToDevice, stream_));
        if (objective_staged_event_ == nullptr) {
            ROCCO_HIP_TRY(hipEventCreateWithFlags(&objective_staged_event_, hipEventDisableTiming));
        }
        ROCCO_HIP_TRY(hipEventRecord(objective_staged_event_, stream_));
        objective_staged_pending_ = true;
        if (!scatters.empty()) {
            if ((rc = launch_lean_scatter_batch((const LeanScatterTask *)dv, (int)scatters.size(), zero_blocks, scatter_blocks, stream_)) != ROCCO_HIP_OK) return rc;
        }
        if ((rc = launch_objective_batch((const ObjectiveTask *)(dv + b_scatter), (int)W, tiles, (double *)(dv + b_scatter + b_tasks),
                                         (double *)(dv + b_scatter + b_tasks + b_part), stream_)) != ROCCO_HIP_OK) {
            return rc;
        }
        ROCCO_HIP_TRY(hipMemcpyAsync(ho + b_scatter + b_tasks, dv + b_scatter + b_tasks + b_part, W * sizeof(double), hipMemcpyDeviceToHost, stream_));
        objective_back_ = (const double *)(ho + b_scatter + b_tasks);
        objective_pending_ = which;
        ++objective_prefetches;
        return ROCCO_HIP_OK;
    }
    const double *objective_back_ = nullptr;
    bool objective_behind_chain_ = false;  // the pending sums were queued behind a chain: valid for the problems it wrote
    hipEvent_t objective_staged_event_ = nullptr;  // behind the last prefetch's copy of its descriptors to the device
    bool objective_staged_pending_ = false;

    // after the wait: keep the sums of the problems whose window certified its solution; the others are scattered again later
    void objective_collect(const std::vector<char> &certified)
    {
        if (objective_ready_.size() < probs.size()) {
            objective_ready_.resize(probs.size(), 0);
            objective_sum_.resize(probs.size(), 0.0);
        }
        const double *back = objective_back_;
        for (size_t i = 0; i < objective_pending_.size(); ++i) {
            const size_t b = objective_pending_[i];
            if (certified[i]) {
                objective_ready_[b] = 1;
                objective_sum_[b] = back[i];
            } else {
                probs[b].scattered = false;
            }
        }
        objective_pending_.clear();
    }

    int penalized_values(const std::vector<size_t> &which, const std::vector<double> &lambdas,
                         const std::vector<long long> &counts, std::vector<double> &values) override
    {
        if (objective_behind_chain_ && !objective_pending_.empty()) {
            // sums queued behind a chain of rounding-model rounds: they are there once the stream has drained, and they
            // count for the problems whose solution the chain wrote and nothing has touched since
            int rcw;
            if ((rcw = model_chain_drain()) != ROCCO_HIP_OK) return rcw;
            ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
            std::vector<char> good(objective_pending_.size(), 0);
            for (size_t i = 0; i < objective_pending_.size(); ++i) {
                const size_t b = objective_pending_[i];
                good[i] = (b < written_.size() && written_[b].valid && probs[b].scattered) ? 1 : 0;
            }
            objective_collect(good);
            objective_behind_chain_ = false;
        }
        {
            // every sum fetched ahead behind the final windows: nothing left to do on the device
            bool all = !which.empty();
            for (size_t b : which) {
                all = all && b < objective_ready_.size() && objective_ready_[b] != 0;
            }
            if (all) {
                int rc0;
                if ((rc0 = scatter_all()) != ROCCO_HIP_OK) return rc0;  // (problems that are not in `which`, if any)
                values.assign(which.size(), 0.0);
                for (size_t i = 0; i < which.size(); ++i) {
                    values[i] = -objective_sum_[which[i]] - lambdas[i] * (double)counts[i];
                }
                objective_prefetch_hits += (long long)which.size();
                return ROCCO_HIP_OK;
            }
        }
        // every objective in two launches, on the arrays the solution lives in (a compacted problem's kept loci
        // carry the same scores and the same transitions as the caller's array), one synchronisation; the
        // compacted solutions are scattered to the callers' buffers on the way
        const size_t W = which.size();
        values.assign(W, 0.0);
        int rc;
        if ((rc = scatter_all()) != ROCCO_HIP_OK) return rc;
        if (W == 0) {
            return ROCCO_HIP_OK;
        }
        std::vector<ObjectiveTask> tasks(W);
        long long tiles = 0;
        for (size_t i = 0; i < W; ++i) {
            const DevProblem &p = probs[which[i]];
            const bool on_level = p.compacted && !p.solution_in_orig;
            tasks[i].solution = on_level ? p.solution : (p.compacted ? p.orig_solution : p.solution);
            tasks[i].scores = on_level ? p.scores : (p.compacted ? p.orig_scores : p.scores);
            tasks[i].switch_costs = p.costs;
            tasks[i].gamma = p.gamma;
            tasks[i].n = (long long)(on_level ? p.n : (p.compacted ? p.orig_n : p.n));
            tasks[i].tile_begin = tiles;
            tiles += objective_tiles((size_t)tasks[i].n);
        }
        const size_t b_tasks = align_up(W * sizeof(ObjectiveTask), 256);
        const size_t b_part = align_up((size_t)(2 * tiles + 2) * sizeof(double), 256);
        const size_t b_out = align_up(W * sizeof(double), 256);
        if ((rc = solver_->dev_misc.reserve(b_tasks + b_part + b_out)) != ROCCO_HIP_OK) return rc;
        if ((rc = stage_wait()) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_stage.reserve(b_tasks)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_back.reserve(b_out)) != ROCCO_HIP_OK) return rc;
        char *dv = (char *)solver_->dev_misc.ptr;
        std::memcpy(solver_->host_stage.ptr, tasks.data(), W * sizeof(ObjectiveTask));
        ROCCO_HIP_TRY(hipMemcpyAsync(dv, solver_->host_stage.ptr, W * sizeof(ObjectiveTask), hipMemcpyHostToDevice, stream_));
        if ((rc = launch_objective_batch((const ObjectiveTask *)dv, (int)W, tiles, (double *)(dv + b_tasks),
                                         (double *)(dv + b_tasks + b_part), stream_)) != ROCCO_HIP_OK) {
            return rc;
        }
        ROCCO_HIP_TRY(hipMemcpyAsync(solver_->host_back.ptr, dv + b_tasks + b_part, W * sizeof(double), hipMemcpyDeviceToHost, stream_));
        ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
        const double *back = (const double *)solver_->host_back.ptr;
        for (size_t i = 0; i < W; ++i) {
            values[i] = -back[i] - lambdas[i] * (double)counts[i];
        }
        return ROCCO_HIP_OK;
    }

    // ---- lean evaluation (lean.h): threshold-search rounds on compacted levels, final compaction ----
    std::vector<LeanState> lean_;
    bool lean_ready_ = false;
    std::vector<LeanReq> lean_inflight_;
    int lean_rounds = 0;
    long long lean_units = 0;

    bool lean_eligible(size_t problem) const
    {
        const DevProblem &p = probs[problem];
        return solver_->lean != 0 && p.costs == nullptr && !p.compacted && p.n >= 2;
    }

    bool can_compact(size_t problem) const override { return lean_eligible(problem); }

    // rounding-model probes of a compacted problem with a map in place go through lean_model_kernel
    bool model_eligible(size_t problem) const
    {
        const char *flag = env("ROCCO_HIP_LEAN_MODEL");  // (read per call: tests switch it within one process)
        const bool enabled = flag == nullptr || std::atoi(flag) != 0;
        const DevProblem &p = probs[problem];
        return enabled && !force_full_ && solver_->lean != 0 && lean_ready_ && ((p.compacted && p.lean_orig != nullptr) || model_any_) &&
               p.costs == nullptr && p.emap != nullptr && p.n >= 2;
    }
    bool force_full_ = false;  // (set while penalties the model kernel could not certify are repeated by the full kernels)
    bool model_any_ = false;   // test entry: any problem with a map, and the flags are reported instead of settled

    // Levels of the bisection tree a rounding-model round evaluates (2^depth - 1 penalties, four per workgroup): as deep
    // as keeps the round of ALL compacted problems within one wave of workgroups -- a round costs its latency (one
    // wave: ~40 us, plus ~45 us of turn-around) until it spills into further waves, and fourteen open levels take
    // five rounds of three levels (the genome's 214 tiles: 2.55 ms per calibration) against three of five (2.68 ms).
    int probe_depth(size_t problem) const override
    {
        if (!model_eligible(problem)) {
            return 0;
        }
        if (const char *e = env("ROCCO_HIP_MODEL_DEPTH")) {
            return std::max(1, std::min(6, std::atoi(e)));
        }
        // (the problems still asking: those of the last rounding-model round; before the first one, every compacted problem)
        long long tiles = model_tiles_last_round_;
        if (tiles <= 0) {
            for (size_t b = 0; b < probs.size(); ++b) {
                if (probs[b].compacted && probs[b].costs == nullptr) {
                    tiles += (long long)((probs[b].n + kLeanTile - 1) / kLeanTile);
                }
            }
        }
        int depth = 3;
        while (depth < 6 && tiles * ((((1LL << (depth + 1)) - 1) + kLeanModelBatch - 1) / kLeanModelBatch) <= 512) {
            ++depth;
        }
        return depth;
    }

    long long model_tiles_last_round_ = 0;

    static constexpr long long kPilotMinTiles = 128;

    bool can_pilot(size_t problem) const override
    {
        return lean_eligible(problem) && (long long)((probs[problem].n + kLeanTile - 1) / kLeanTile) >= kPilotMinTiles;
    }

    int bound_points(size_t problem, int default_points, double base_hint) const override
    {
        if (!lean_eligible(problem)) {
            return default_points;
        }
        long long m = (long long)probs[problem].n;
        bool deep = false;
        if (lean_ready_ && problem < lean_.size() && !lean_[problem].levels.empty()) {
            const LeanLevel &lv = lean_[problem].levels.back();
            if (lean_[problem].levels.size() > 1) {
                m = lv.m;
                deep = true;
            }
            // the next round compacts at `base_hint` first if that halves the level: it then runs on the child
            if (lv.has_eval && base_hint == base_hint) {
                for (size_t i = 0; i < lv.pts.size(); ++i) {
                    if (lv.pts[i] == base_hint && lv.child_len[i] >= 1 && 2 * lv.child_len[i] <= lv.m) {
                        m = lv.child_len[i];
                        deep = true;
                    }
                }
            }
        }
        // a round on a small level costs its launch latency whatever it evaluates (one tile and 8 penalties per
        // workgroup: up to 256 of them run at once), a pass over a long one its evaluations
        const long long small = env("ROCCO_HIP_POINTS_SMALL") ? std::atoll(env("ROCCO_HIP_POINTS_SMALL")) : 256000;
        int pts = (m > 8000000) ? 3 : ((m > 2000000) ? 4 : ((m > 4 * small) ? 8 : ((m > 2 * small) ? 16 : ((m > small) ? 32 : 64))));
        if (!deep) {
            pts = std::min(pts, 8);
        }
        return pts;
    }

    int lean_prepare()
    {
        if (lean_ready_) {
            return ROCCO_HIP_OK;
        }
        lean_.assign(probs.size(), LeanState());
        size_t total = 0;
        for (size_t b = 0; b < probs.size(); ++b) {
            lean_[b].pool_begin = total;
            total += align_up(40 * probs[b].n + ((size_t)1 << 19), 256);
            lean_[b].pool_end = total;
            lean_[b].pool_at = lean_[b].pool_begin;
        }
        int rc = solver_->dev_lean_pool.reserve(total + 256);
        if (rc != ROCCO_HIP_OK) return rc;
        // per problem: the tolerance cap of the rounding-model evaluation and the counters it is summed from (zero
        // between uses: lean_wcap_sum_kernel clears what it read)
        const size_t wcap_bytes = align_up(probs.size() * sizeof(double), 256) + probs.size() * 512 * sizeof(unsigned);
        if ((rc = solver_->dev_lean_wcap.reserve(wcap_bytes + 256)) != ROCCO_HIP_OK) return rc;
        ROCCO_HIP_TRY(hipMemsetAsync(solver_->dev_lean_wcap.ptr, 0, wcap_bytes, stream_));
        lean_ready_ = true;
        return ROCCO_HIP_OK;
    }

    void *lean_alloc(LeanState &ls, size_t bytes)
    {
        const size_t at = align_up(ls.pool_at, 256);
        if (at + bytes > ls.pool_end) {
            return nullptr;
        }
        ls.pool_at = at + bytes;
        return (char *)solver_->dev_lean_pool.ptr + at;
    }

    static double separator_score(double base, double gamma) { return std::floor(base - 2.0 * gamma - 2.0); }

    // ---- chained rounding-model rounds (model_chain.h) ----
    // certified counts the evaluator holds per problem: (penalty, the reference's count there)
    std::vector<std::vector<std::pair<double, long long>>> facts_;
    std::vector<std::pair<size_t, double>> open_facts_;  // (problem, penalty) the chains evaluated without certifying the count
    bool model_chain_inflight_ = false;  // the requests of lean_inflight_ started a chain (lean_consume answers them from its facts)
    bool model_chain_running_ = false;   // a chain is on the stream and may still publish rounds
    int model_chain_ingested_ = 0;       // rounds of it whose facts are in facts_
    std::vector<size_t> model_chain_problems_;
    const ModelChainReport *model_chain_report_ = nullptr;
    const int *model_chain_np_ = nullptr;
    const ModelChainFact *model_chain_facts_ = nullptr;
    const ModelChainFinal *model_chain_finals_ = nullptr;
    // solutions a chain wrote at its end: problem -> (penalty, selected loci); valid until a round touches the problem
    struct Written {
        bool valid = false;
        double penalty = 0.0;
        long long count = 0;
    };
    std::vector<Written> written_;
    long long model_chain_written = 0, model_chain_windows_answered = 0;
    long long model_chains = 0, model_chain_facts = 0, model_chain_hits = 0, model_chain_misses = 0, model_chain_stopped = 0;
    long long model_tiles_answered_ = 0;
    double t_mchain_submit_ = 0.0, t_mchain_wait_ = 0.0;

    // host-side event log (ROCCO_HIP_TIMING=2): label, microseconds
    std::vector<std::pair<const char *, double>> marks_;
    bool marks_on_ = false;
    void mark(const char *label)
    {
        if (marks_on_) {
            marks_.emplace_back(label, now_us());
        }
    }

    // A round that builds binade maps hands nothing back to the host: it is queued and NOT waited for -- whatever the
    // search asks next (the rounding-model rounds) queues behind it while it runs.  Only the pinned staging buffer its
    // descriptors were copied from must not be written again before that copy has happened: an event marks it.
    hipEvent_t staged_event_ = nullptr;
    bool staged_pending_ = false;
    ~HipEvaluator() override
    {
        // An evaluator that goes away with work still queued (a call that ends on an error with a chain running, or with
        // sums on their way): the queued kernels write into the caller's buffers and into the solver's pinned memory, which
        // the next call clears -- so the stream is drained first.
        if (model_chain_running_ || !objective_pending_.empty() || objective_staged_pending_ || staged_pending_) {
            (void)hipStreamSynchronize(stream_);
        }
        if (staged_event_ != nullptr) {
            (void)hipEventDestroy(staged_event_);
        }
        if (objective_staged_event_ != nullptr) {
            (void)hipEventDestroy(objective_staged_event_);
        }
    }
    int stage_wait()
    {
        if (staged_pending_) {
            staged_pending_ = false;
            ROCCO_HIP_TRY(hipEventSynchronize(staged_event_));
        }
        return ROCCO_HIP_OK;
    }
    int stage_mark()
    {
        if (staged_event_ == nullptr) {
            ROCCO_HIP_TRY(hipEventCreateWithFlags(&staged_event_, hipEventDisableTiming));
        }
        ROCCO_HIP_TRY(hipEventRecord(staged_event_, stream_));
        staged_pending_ = true;
        return ROCCO_HIP_OK;
    }

    // Does the iteration's lean work need the stream to drain before it is read?  Not when nothing was queued (every
    // request was answered from the facts), and not for a chain: its rounds are followed through pinned memory.
    bool lean_wait_needed() const { return !lean_inflight_.empty() && !model_chain_inflight_; }

    bool fact_lookup(size_t problem, double lambda, long long *count) const
    {
        if (problem >= facts_.size()) {
            return false;
        }
        for (const auto &f : facts_[problem]) {
            if (f.first == lambda) {
                *count = f.second;
                return true;
            }
        }
        return false;
    }

    bool facts_answer(const ProbeRequest &q, std::vector<ProbeResult> &res) const
    {
        res.assign(q.lambdas.size(), ProbeResult());
        for (size_t i = 0; i < q.lambdas.size(); ++i) {
            if (!fact_lookup(q.problem, q.lambdas[i], &res[i].count)) {
                return false;
            }
        }
        return true;
    }

    bool in_running_chain(size_t problem) const
    {
        return model_chain_running_ &&
               std::find(model_chain_problems_.begin(), model_chain_problems_.end(), problem) != model_chain_problems_.end();
    }

    // Take over the rounds the running chain has published since the last call; `wait`: until at least one more round is
    // there or the chain has ended.  The facts sit in host-coherent pinned memory, written by the director kernels.
    int model_chain_ingest(bool wait, bool *got)
    {
        *got = false;
        if (!model_chain_running_) {
            return ROCCO_HIP_OK;
        }
        const double t0 = now_us();
        const size_t B = model_chain_problems_.size();
        const ModelChainReport *rep = model_chain_report_;
        long long spins = 0;
        for (;;) {
            const int finished = __atomic_load_n(&rep->finished, __ATOMIC_ACQUIRE);
            const int published = __atomic_load_n(&rep->published, __ATOMIC_ACQUIRE);
            if (published > model_chain_ingested_) {
                for (int r = model_chain_ingested_; r < published; ++r) {
                    for (size_t i = 0; i < B; ++i) {
                        const int np = model_chain_np_[(size_t)r * B + i];
                        const ModelChainFact *f = model_chain_facts_ + ((size_t)r * B + i) * kLeanMaxPoints;
                        for (int k = 0; k < np && k < kLeanMaxPoints; ++k) {
                            if (f[k].flags == 0) {
                                facts_[model_chain_problems_[i]].emplace_back(f[k].penalty, f[k].count);
                                ++model_chain_facts;
                            } else {
                                open_facts_.emplace_back(model_chain_problems_[i], f[k].penalty);  // evaluated, not certified
                            }
                        }
                    }
                }
                model_chain_ingested_ = published;
                *got = true;
                mark("chain round ingested");
            }
            if (finished != 0 && model_chain_ingested_ >= __atomic_load_n(&rep->published, __ATOMIC_ACQUIRE)) {
                model_chain_running_ = false;
                model_chain_stopped += rep->stopped;
                if (written_.size() < probs.size()) {
                    written_.resize(probs.size());
                }
                for (size_t i = 0; i < B; ++i) {
                    const ModelChainFinal &fin = model_chain_finals_[i];
                    Written &w = written_[model_chain_problems_[i]];
                    w.valid = fin.written != 0 && fin.iters_left == 0;
                    w.penalty = fin.upper;
                    w.count = fin.count;
                    if (w.valid) {
                        ++model_chain_written;
                        if (!objective_behind_chain_) {
                            probs[model_chain_problems_[i]].scattered = false;  // (no scatter queued behind the write: it is due)
                        }
                    }
                }
                if (env("ROCCO_HIP_DEBUG") != nullptr) {
                    std::fprintf(stderr, "[model chain] ended: %d rounds asked something, %d problems stopped at an open outcome\n",
                                 rep->rounds_run, rep->stopped);
                }
                if (rep->error & 1u) {
                    solver_->lean_look_dirty = 1;
                    set_last_error("lean evaluation: a tile waited for its predecessor beyond the spin limit");
                    return ROCCO_HIP_EHIP;
                }
                break;
            }
            if (*got || !wait) {
                break;
            }
            if (spins > 200000) {
                std::this_thread::yield();  // (a chain queued behind long work of another kind: do not burn the core)
            }
            if ((++spins & 1023) == 0) {
                const hipError_t q = hipStreamQuery(stream_);
                if (q != hipErrorNotReady) {
                    // the stream has drained (whatever was going to be published is there) or failed
                    if (q != hipSuccess || __atomic_load_n(&rep->finished, __ATOMIC_ACQUIRE) == 0) {
                        model_chain_running_ = false;
                        set_last_error(q != hipSuccess ? "model chain: the stream reports an error" : "model chain: the stream ended without the chain's report");
                        return ROCCO_HIP_EHIP;
                    }
                }
            }
        }
        t_mchain_wait_ += now_us() - t0;
        return ROCCO_HIP_OK;
    }

    int model_chain_drain()
    {
        int rc = ROCCO_HIP_OK;
        bool got = false;
        while (model_chain_running_ && rc == ROCCO_HIP_OK) {
            rc = model_chain_ingest(true, &got);
        }
        return rc;
    }

    // every request of the round is a rounding-model probe that says how the bisection goes on: run the rounds ahead
    bool model_chain_wanted(const std::vector<LeanReq> &reqs) const
    {
        const char *flag = env("ROCCO_HIP_MODEL_CHAIN");
        if ((flag != nullptr && std::atoi(flag) == 0) || model_any_ || reqs.empty() || reqs.size() > (size_t)kModelChainMaxProblems) {
            return false;
        }
        int rounds = 0;
        for (const LeanReq &r : reqs) {
            if (!r.model || r.probe == nullptr || !r.probe->ahead.valid || r.probe->ahead.open_depth < 1) {
                return false;
            }
            const BisectionAhead &a = r.probe->ahead;
            rounds = std::max(rounds, (a.iters_left + a.open_depth - 1) / a.open_depth);
        }
        return rounds >= 2;
    }

    int model_chain_enqueue(std::vector<LeanReq> &reqs)
    {
        int rc;
        if ((rc = model_chain_drain()) != ROCCO_HIP_OK) return rc;
        const double t0 = now_us();
        mark("model chain: enqueue begins");
        const size_t B = reqs.size();
        ++lean_rounds;
        ++model_chains;
        std::vector<ModelChainWalk> walk(B);
        std::vector<LeanTask> tasks(B);
        std::vector<LeanWcapTask> wcap_tasks;
        int wcap_blocks = 0;
        long long tiles = 0;
        int rounds = 0, depth0 = 0, depth_floor = 0;
        model_chain_problems_.assign(B, 0);
        for (size_t i = 0; i < B; ++i) {
            LeanReq &r = reqs[i];
            DevProblem &p = probs[r.problem];
            const BisectionAhead &a = r.probe->ahead;
            model_chain_problems_[i] = r.problem;
            double *wcap = (double *)solver_->dev_lean_wcap.ptr + r.problem;
            unsigned *counters = (unsigned *)((char *)solver_->dev_lean_wcap.ptr + align_up(probs.size() * sizeof(double), 256)) + 512 * r.problem;
            if (p.wcap_version != p.map_version) {
                LeanWcapTask wt;
                wt.emap = p.emap;
                wt.s = p.scores;
                wt.m = (long long)p.n;
                wt.qexp = p.qexp;
                wt.e_floor = std::ilogb(2.0 * p.cmax + 2.0 * p.sabs + (p.sabs + 2.0) + 2.0);  // (|penalty| <= sabs + 2)
                wt.counters = counters;
                wt.clean_chunks = counters + 384;
                wt.wcap = wcap;
                wt.block_begin = wcap_blocks;
                wt.pad = 0;
                wcap_blocks += (int)((p.n + kLeanTile - 1) / kLeanTile);
                wcap_tasks.push_back(wt);
                p.wcap_version = p.map_version;
            }
            LeanTask &t = tasks[i];
            t.s = p.scores;
            t.m = (long long)p.n;
            t.c_raw = p.gamma;
            t.magic = std::ldexp(1.5, 52 + p.qexp);
            t.big = std::ldexp(1.0, 50 + p.qexp);
            t.n_tiles = (int)((p.n + kLeanTile - 1) / kLeanTile);
            t.n_points = 0;
            t.n_groups = 0;
            t.unit_begin = 0;
            t.point_begin = (int)i * kLeanMaxPoints;
            t.rec_begin = 0;
            t.bits_begin = 0;
            t.off_begin = 0;
            t.result_begin = (int)i * kLeanMaxPoints;
            t.tile_stride = 1;
            t.independent = 0;
            t.store = 0;
            t.emap = p.emap;
            t.wcap = wcap;
            t.clean_chunks = counters + 384;
            t.cmax = p.cmax;
            t.sabs = p.sabs;
            t.qexp = p.qexp;
            t.batch = kLeanModelBatch;
            tiles += t.n_tiles;
            ModelChainWalk &w = walk[i];
            w.lower = a.lower;
            w.upper = a.upper;
            w.G = a.G;
            w.L = a.L;
            w.sabs = a.sabs;
            w.cost_max = a.cost_max;
            w.none_from = a.none_from;
            w.all_upto = a.all_upto;
            w.target = a.target;
            w.cG = a.cG;
            w.cL = a.cL;
            w.n = a.n;
            w.iters_left = a.iters_left;
            w.G_real = a.G_real ? 1 : 0;
            w.L_real = a.L_real ? 1 : 0;
            w.cost_ok = a.cost_ok ? 1 : 0;
            w.n_tiles = t.n_tiles;
            w.pad[0] = w.pad[1] = w.pad[2] = 0;
            w.solution = p.solution;
            w.m = (long long)p.n;
            depth0 = std::max(depth0, a.open_depth);
            depth_floor = std::max(depth_floor, a.depth_floor);
            rounds = std::max(rounds, (a.iters_left + a.open_depth - 1) / a.open_depth);
        }
        if (const char *e = std::getenv("ROCCO_HIP_MODEL_CHAIN_ROUNDS")) {
            rounds = std::atoi(e);
        }
        rounds = std::max(1, std::min(kModelChainMaxRounds, rounds));
        model_tiles_last_round_ = tiles + model_tiles_answered_;

        // device: [tasks][walk][wcap tasks] (uploaded) [state][points][results][ctl][globals]
        const size_t b_tasks = align_up(B * sizeof(LeanTask), 256);
        const size_t b_walk = align_up(B * sizeof(ModelChainWalk), 256);
        const size_t b_wcap = align_up(wcap_tasks.size() * sizeof(LeanWcapTask), 256);
        const size_t up_bytes = b_tasks + b_walk + b_wcap;
        const size_t b_state = align_up(B * sizeof(ModelChainState), 256);
        const size_t b_points = align_up(B * kLeanMaxPoints * sizeof(double), 256);
        const size_t b_results = align_up(B * kLeanMaxPoints * sizeof(LeanResult), 256);
        // what writes the final solutions at the chain's end (lean.h: LeanTask::store == 2): two 256-word planes and one
        // entering value per (tile, penalty) pair of every round; ROCCO_HIP_CHAIN_WRITE=0: nothing is kept, the final windows
        // run as before
        const char *write_env = std::getenv("ROCCO_HIP_CHAIN_WRITE");
        long long cap_pairs = (write_env != nullptr && std::atoi(write_env) == 0) ? 0 : std::max<long long>(tiles * 7, 2048);
        if ((size_t)rounds * (size_t)cap_pairs * 2 * 256 * sizeof(unsigned) > ((size_t)1 << 30)) {
            cap_pairs = 0;  // (more than 1 GiB of class words: not worth it, the windows run)
        }
        const size_t b_bits = align_up((size_t)rounds * (size_t)cap_pairs * 2 * 256 * sizeof(unsigned), 256);
        const size_t b_enter = align_up((size_t)rounds * (size_t)cap_pairs * sizeof(unsigned), 256);
        const size_t b_writes = align_up(B * sizeof(LeanWriteTask), 256);
        const size_t dev_bytes = up_bytes + b_state + b_points + b_results + 512 + b_writes + 256 + b_enter + b_bits;
        // host-coherent: [report][n_points per round and problem][finals][facts]
        const size_t b_np = align_up((size_t)rounds * B * sizeof(int), 256);
        const size_t b_finals = align_up(B * sizeof(ModelChainFinal), 256);
        const size_t follow_bytes = 256 + b_np + b_finals + (size_t)rounds * B * kLeanMaxPoints * sizeof(ModelChainFact);
        if ((rc = solver_->dev_chain.reserve(dev_bytes + 256)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_chain.reserve(up_bytes + 256)) != ROCCO_HIP_OK) return rc;
        solver_->host_follow.coherent = true;
        if ((rc = solver_->host_follow.reserve(follow_bytes + 256)) != ROCCO_HIP_OK) return rc;
        char *dv = (char *)solver_->dev_chain.ptr;
        char *h = (char *)solver_->host_chain.ptr;
        char *f = (char *)solver_->host_follow.ptr;
        std::memcpy(h, tasks.data(), B * sizeof(LeanTask));
        std::memcpy(h + b_tasks, walk.data(), B * sizeof(ModelChainWalk));
        if (!wcap_tasks.empty()) std::memcpy(h + b_tasks + b_walk, wcap_tasks.data(), wcap_tasks.size() * sizeof(LeanWcapTask));
        mark("model chain: tables built");
        ROCCO_HIP_TRY(hipMemcpyAsync(dv, h, up_bytes, hipMemcpyHostToDevice, stream_));
        if (!wcap_tasks.empty()) {
            if ((rc = launch_lean_wcap((const LeanWcapTask *)(dv + b_tasks + b_walk), (int)wcap_tasks.size(), wcap_blocks, stream_)) != ROCCO_HIP_OK) return rc;
        }
        ModelChainArgs A;
        A.n_problems = (int)B;
        A.depth0 = depth0;
        A.depth_floor = depth_floor;
        A.depth_fixed = env("ROCCO_HIP_MODEL_DEPTH") ? std::max(1, std::min(6, std::atoi(env("ROCCO_HIP_MODEL_DEPTH")))) : 0;
        A.adapt_batch = (env("ROCCO_HIP_LEAN_BATCH") == nullptr || std::atoi(env("ROCCO_HIP_LEAN_BATCH")) != 0) ? 1 : 0;
        A.cap_pairs = (int)cap_pairs;
        A.tasks = (LeanTask *)dv;
        A.walk = (const ModelChainWalk *)(dv + b_tasks);
        A.state = (ModelChainState *)(dv + up_bytes);
        A.points = (double *)(dv + up_bytes + b_state);
        A.results = (LeanResult *)(dv + up_bytes + b_state + b_points);
        A.ctl = (LeanRoundCtl *)(dv + up_bytes + b_state + b_points + b_results);
        A.globals = (int *)(dv + up_bytes + b_state + b_points + b_results + 256);
        A.report = (ModelChainReport *)f;
        A.n_points_out = (int *)(f + 256);
        A.finals = (ModelChainFinal *)(f + 256 + b_np);
        A.facts = (ModelChainFact *)(f + 256 + b_np + b_finals);
        {
            char *tail = dv + up_bytes + b_state + b_points + b_results + 512;
            A.writes = (LeanWriteTask *)tail;
            A.n_writes = (int *)(tail + b_writes);
            A.entering = (unsigned *)(tail + b_writes + 256);
            A.bits = (unsigned *)(tail + b_writes + 256 + b_enter);
        }
        model_chain_finals_ = A.finals;
        std::memset(f, 0, 256);
        model_chain_report_ = A.report;
        model_chain_np_ = A.n_points_out;
        model_chain_facts_ = A.facts;
        if (facts_.size() < probs.size()) {
            facts_.resize(probs.size());
        }

        // round scratch as the regular rounds keep it (lean_enqueue): sized for the deepest round a chain may plan
        const long long max_recs = tiles * (long long)(kLeanMaxPoints - 1);
        const size_t b_look = align_up(256 + (size_t)max_recs * 4 * sizeof(unsigned long long), 256);
        {
            const void *old_ptr = solver_->dev_lean_look.ptr;
            const size_t old_bytes = solver_->dev_lean_look.bytes;
            if ((rc = solver_->dev_lean_look.reserve(b_look)) != ROCCO_HIP_OK) return rc;
            if (solver_->lean_look_dirty != 0 || solver_->dev_lean_look.ptr != old_ptr || solver_->dev_lean_look.bytes != old_bytes) {
                ROCCO_HIP_TRY(hipMemsetAsync(solver_->dev_lean_look.ptr, 0xFF, solver_->dev_lean_look.bytes, stream_));
                ROCCO_HIP_TRY(hipMemsetAsync((char *)solver_->dev_lean_look.ptr + 128, 0, 4, stream_));
                solver_->lean_look_dirty = 0;
            }
        }
        if ((rc = solver_->dev_lean_round.reserve(align_up((size_t)max_recs * sizeof(LeanTileRec), 256) + 256)) != ROCCO_HIP_OK) return rc;
        char *look = (char *)solver_->dev_lean_look.ptr;
        LeanLaunch L;
        L.tasks = A.tasks;
        L.n_tasks = 0;
        L.n_units = 0;
        L.points = A.points;
        L.ticket = (unsigned *)look;
        L.look = (unsigned long long *)(look + 256);
        L.recs = (LeanTileRec *)solver_->dev_lean_round.ptr;
        L.bits = A.bits;          // (rounding-model tasks keep nothing in the level pool: the chain's own word arrays)
        L.tile_off = A.entering;
        L.results = A.results;
        L.error = (unsigned *)(look + 128);
        L.error_out = nullptr;
        L.self_reset = 1;
        L.pad = 0;
        L.ctl = A.ctl;
        LeanLaunch M = L;
        M.ticket = (unsigned *)look + 2;
        const int grid_eval = (int)std::max(1LL, std::min(512LL, tiles * 16));
        const int grid_finish = (int)std::max(1LL, std::min(1536LL, (long long)B * (kLeanMaxPoints - 1)));
        solver_->lean_look_dirty = 1;
        // (round 5: evaluation and finish of a round as one launch, as the threshold search's rounds)
        const int fused_mode = env("ROCCO_HIP_CHAIN_FUSED") == nullptr ? 0 : std::atoi(env("ROCCO_HIP_CHAIN_FUSED"));
        const bool fused = fused_mode == 1 || fused_mode == 3;
        A.reset = LeanRoundReset{nullptr, 0, nullptr, nullptr};
        if (fused) {
            const size_t words = (size_t)kLeanProgressPairs + (size_t)B * (size_t)kLeanMaxPoints;
            if ((rc = solver_->dev_lean_progress.reserve(words * sizeof(unsigned))) != ROCCO_HIP_OK) return rc;
            A.reset = LeanRoundReset{(unsigned *)solver_->dev_lean_progress.ptr, (int)words, L.ticket, L.error};
        }
        for (int r = 0; r < rounds; ++r) {
            if ((rc = launch_model_chain_director(A, r, 0, stream_)) != ROCCO_HIP_OK) return rc;
            if (fused) {
                if ((rc = launch_lean_round_chain(M, nullptr, A.reset.progress, grid_eval, 1, stream_)) != ROCCO_HIP_OK) return rc;
                continue;
            }
            if ((rc = launch_lean_model_chain(M, grid_eval, stream_)) != ROCCO_HIP_OK) return rc;
            if ((rc = launch_lean_finish_chain(L, grid_finish, stream_)) != ROCCO_HIP_OK) return rc;
        }
        if ((rc = launch_model_chain_director(A, rounds, 1, stream_)) != ROCCO_HIP_OK) return rc;
        if (cap_pairs > 0) {
            // the solutions of the bisections that ended, at the penalties they ended at (the last director names them)
            if ((rc = launch_lean_write_solutions(A.writes, A.n_writes, (int)std::max(1LL, std::min(256LL, tiles)), stream_)) != ROCCO_HIP_OK) return rc;
            // ... and, behind them, their scatter into the callers' buffers and their objective sums (for every problem of the
            // chain: what was not written is scattered and summed again when its window has run)
            const char *pf = env("ROCCO_HIP_PREFETCH_OBJECTIVE");
            if (pf == nullptr || std::atoi(pf) != 0) {
                if ((rc = prefetch_objectives(model_chain_problems_)) != ROCCO_HIP_OK) return rc;
                objective_behind_chain_ = true;
            }
        }
        ROCCO_HIP_TRY(hipGetLastError());
        solver_->lean_look_dirty = 0;
        lean_inflight_ = reqs;
        model_chain_inflight_ = true;
        model_chain_running_ = true;
        model_chain_ingested_ = 0;
        t_mchain_submit_ += now_us() - t0;
        mark("model chain: queued");
        if (env("ROCCO_HIP_DEBUG") != nullptr) {
            std::fprintf(stderr, "[model chain] %zu problems, %lld tiles, %d rounds queued (depth %d, floor %d)\n", B, tiles, rounds, depth0,
                         depth_floor);
        }
        return ROCCO_HIP_OK;
    }

    // the requests that started the chain: answered as soon as its first round is published
    int model_chain_consume()
    {
        model_chain_inflight_ = false;
        int rc;
        bool got = false;
        while (model_chain_running_ && model_chain_ingested_ < 1) {
            if ((rc = model_chain_ingest(true, &got)) != ROCCO_HIP_OK) return rc;
        }
        for (LeanReq &r : lean_inflight_) {
            DevProblem &p = probs[r.problem];
            r.probe->results.assign(r.lambdas.size(), ProbeResult());
            for (size_t i = 0; i < r.lambdas.size(); ++i) {
                long long c = 0;
                if (fact_lookup(r.problem, r.lambdas[i], &c)) {
                    r.probe->results[i].count = c;
                    ++model_chain_hits;
                } else {
                    // not among the facts (the model left it open, or the director asked something else): the full kernels decide
                    r.probe->results[i].uncertain = 1;
                    r.probe->results[i].effect = (long long)p.n + 1;
                    model_open_.emplace_back(r.probe, i);
                    const bool evaluated = std::find(open_facts_.begin(), open_facts_.end(), std::make_pair(r.problem, r.lambdas[i])) != open_facts_.end();
                    model_chain_misses += evaluated ? 0 : 1;
                }
            }
            lean_model_points += (long long)r.lambdas.size();
        }
        lean_inflight_.clear();
        return ROCCO_HIP_OK;
    }

    // Queue one round of lean work on the stream: compactions decided by the previous round, the evaluation of
    // every request, the layout of its compactions, the final compactions.  lean_consume() after the stream
    // has been synchronised.
    int lean_enqueue(std::vector<LeanReq> &reqs)
    {
        const double le0 = now_us();
        lean_inflight_.clear();
        if (reqs.empty()) {
            return ROCCO_HIP_OK;
        }
        int rc;
        if ((rc = lean_prepare()) != ROCCO_HIP_OK) return rc;
        if (model_chain_wanted(reqs)) {
            return model_chain_enqueue(reqs);
        }
        ++lean_rounds;
        std::vector<LeanCompactTask> pre, post;
        std::vector<LeanTask> tasks, model_tasks;
        long long model_tiles = 0;
        std::vector<LeanWcapTask> wcap_tasks;
        int model_units = 0, wcap_blocks = 0;
        std::vector<double> points;
        std::vector<size_t> post_req;
        int units = 0, recs = 0, results = 0, pre_blocks = 0, post_blocks = 0;
        const unsigned *pool_words = (const unsigned *)solver_->dev_lean_pool.ptr;
        for (LeanReq &r : reqs) {
            DevProblem &p = probs[r.problem];
            LeanState &ls = lean_[r.problem];
            if (r.model) {
                // the compacted problem itself, chunk modes from its binade map; its own launch (other kernel)
                double *wcap = (double *)solver_->dev_lean_wcap.ptr + r.problem;
                if (p.wcap_version != p.map_version) {
                    LeanWcapTask wt;
                    wt.emap = p.emap;
                    wt.s = p.scores;
                    wt.m = (long long)p.n;
                    wt.qexp = p.qexp;
                    wt.e_floor = std::ilogb(2.0 * p.cmax + 2.0 * p.sabs + (p.sabs + 2.0) + 2.0);  // (|penalty| <= sabs + 2)
                    wt.counters = (unsigned *)((char *)solver_->dev_lean_wcap.ptr + align_up(probs.size() * sizeof(double), 256)) + 512 * r.problem;
                    wt.clean_chunks = wt.counters + 384;
                    wt.wcap = wcap;
                    wt.block_begin = wcap_blocks;
                    wt.pad = 0;
                    wcap_blocks += (int)((p.n + kLeanTile - 1) / kLeanTile);
                    wcap_tasks.push_back(wt);
                    p.wcap_version = p.map_version;
                }
                const int np = (int)r.lambdas.size();
                const int nt = (int)((p.n + kLeanTile - 1) / kLeanTile);
                model_tiles += nt;
                LeanTask t;
                t.s = p.scores;
                t.m = (long long)p.n;
                t.c_raw = p.gamma;
                t.magic = std::ldexp(1.5, 52 + p.qexp);
                t.big = std::ldexp(1.0, 50 + p.qexp);
                t.n_tiles = nt;
                t.n_points = np;
                t.n_groups = (np + kLeanModelBatch - 1) / kLeanModelBatch;
                t.unit_begin = model_units;
                t.point_begin = (int)points.size();
                t.rec_begin = recs;
                t.bits_begin = 0;
                t.off_begin = 0;
                t.result_begin = results;
                t.tile_stride = 1;
                t.independent = 0;
                t.store = 0;
                t.emap = p.emap;
                t.wcap = wcap;
                t.clean_chunks = (const unsigned *)((char *)solver_->dev_lean_wcap.ptr + align_up(probs.size() * sizeof(double), 256)) + 512 * r.problem + 384;
                t.cmax = p.cmax;
                t.sabs = p.sabs;
                t.qexp = p.qexp;
                t.batch = kLeanModelBatch;
                r.result_begin = results;
                model_units += nt * t.n_groups;
                recs += nt * np;
                results += np;
                points.insert(points.end(), r.lambdas.begin(), r.lambdas.end());
                model_tasks.push_back(t);
                continue;
            }
            if (ls.levels.empty()) {
                LeanLevel l0;
                l0.s = p.scores;
                l0.m = (long long)p.n;
                l0.pool_mark = ls.pool_at;
                ls.levels.push_back(l0);
            }
            if (r.pilot) {
                // every stride-th tile of the caller's array, each as a chain of its own; nothing is kept
                const long long all_tiles = (long long)((p.n + kLeanTile - 1) / kLeanTile);
                const long long pilot_tiles = env("ROCCO_HIP_PILOT_TILES") ? std::max(2, std::atoi(env("ROCCO_HIP_PILOT_TILES"))) : 16;
                const int stride = (int)std::max(4LL, all_tiles / pilot_tiles);  // about 16 tiles per chromosome
                const int nt = (int)((all_tiles + stride - 1) / stride);
                long long sampled = 0;
                for (int k = 0; k < nt; ++k) {
                    const long long at = (long long)k * stride * kLeanTile;
                    sampled += std::min((long long)kLeanTile, (long long)p.n - at);
                }
                r.pilot_scale = (double)p.n / (double)std::max(1LL, sampled);
                const int np = (int)r.lambdas.size();
                LeanTask t;
                t.s = p.scores;
                t.m = (long long)p.n;
                t.c_raw = p.gamma;
                t.magic = std::ldexp(1.5, 52 + p.qexp);
                t.big = std::ldexp(1.0, 50 + p.qexp);
                t.n_tiles = nt;
                t.n_points = np;
                t.n_groups = (np + kLeanBatch - 1) / kLeanBatch;
                t.unit_begin = units;
                t.point_begin = (int)points.size();
                t.rec_begin = recs;
                t.bits_begin = 0;
                t.off_begin = 0;
                t.result_begin = results;
                t.tile_stride = stride;
                t.independent = 1;
                t.store = 0;
                t.emap = nullptr;
                t.wcap = nullptr;
                t.clean_chunks = nullptr;
                t.cmax = t.sabs = 0.0;
                t.qexp = p.qexp;
                t.batch = kLeanBatch;
                r.result_begin = results;
                units += nt * t.n_groups;
                recs += nt * np;
                results += np;
                points.insert(points.end(), r.lambdas.begin(), r.lambdas.end());
                tasks.push_back(t);
                continue;
            }
            const double lam_min = *std::min_element(r.lambdas.begin(), r.lambdas.end());
            while (ls.levels.size() > 1 && ls.levels.back().base > lam_min) {
                ls.pool_at = ls.levels.back().pool_mark;
                ls.levels.pop_back();
            }
            // a deeper level from the latest evaluation of this one?
            {
                LeanLevel &lv = ls.levels.back();
                int best = -1;
                if (lv.has_eval) {
                    for (size_t i = 0; i < lv.pts.size(); ++i) {
                        if (lv.pts[i] <= lam_min && (best < 0 || lv.pts[i] > lv.pts[(size_t)best])) {
                            best = (int)i;
                        }
                    }
                }
                if (best >= 0 && lv.child_len[(size_t)best] >= 1 && 2 * lv.child_len[(size_t)best] <= lv.m) {
                    const long long cl = lv.child_len[(size_t)best];
                    const size_t mark = ls.pool_at;
                    double *cs = (double *)lean_alloc(ls, (size_t)cl * sizeof(double));
                    int *co = (int *)lean_alloc(ls, (size_t)cl * sizeof(int));
                    if (cs != nullptr && co != nullptr) {
                        const int nt = (int)((lv.m + kLeanTile - 1) / kLeanTile);
                        LeanCompactTask ct;
                        ct.s = lv.s;
                        ct.orig = lv.orig;
                        ct.m = lv.m;
                        ct.n_tiles = nt;
                        ct.block_begin = pre_blocks;
                        ct.bits = lv.bits + (size_t)best * (size_t)nt * kLeanThreads;
                        ct.tile_off = lv.tile_off + (size_t)best * (size_t)nt;
                        ct.sep = separator_score(lv.pts[(size_t)best], p.gamma);
                        ct.out_s = cs;
                        ct.out_orig = co;
                        ct.capacity = cl;
                        pre.push_back(ct);
                        pre_blocks += nt;
                        LeanLevel child;
                        child.s = cs;
                        child.orig = co;
                        child.m = cl;
                        child.base = lv.pts[(size_t)best];
                        child.sep = ct.sep;
                        child.pool_mark = mark;
                        ls.levels.push_back(child);
                    } else {
                        ls.pool_at = mark;
                    }
                }
            }
            LeanLevel &lv = ls.levels.back();
            const int np = (int)r.lambdas.size();
            const int nt = (int)((lv.m + kLeanTile - 1) / kLeanTile);
            if (lv.cap_points < np) {
                // storage of this level's evaluations (level 0 never takes more than 8 penalties a round)
                const int cap = std::max(np, (ls.levels.size() > 1) ? kLeanMaxPoints : 8);
                lv.bits = (unsigned *)lean_alloc(ls, (size_t)cap * (size_t)nt * kLeanThreads * sizeof(unsigned));
                lv.tile_off = (unsigned *)lean_alloc(ls, (size_t)cap * (size_t)nt * sizeof(unsigned));
                if (lv.bits == nullptr || lv.tile_off == nullptr) {
                    set_last_error("lean evaluation: level pool exhausted");
                    return ROCCO_HIP_ENOMEM;
                }
                lv.cap_points = cap;
            }
            lv.has_eval = false;
            LeanTask t;
            t.s = lv.s;
            t.m = lv.m;
            t.c_raw = p.gamma;
            t.magic = std::ldexp(1.5, 52 + p.qexp);
            t.big = std::ldexp(1.0, 50 + p.qexp);
            t.n_tiles = nt;
            t.n_points = np;
            t.n_groups = (np + kLeanBatch - 1) / kLeanBatch;
            t.unit_begin = units;
            t.point_begin = (int)points.size();
            t.rec_begin = recs;
            t.bits_begin = (long long)(lv.bits - pool_words);
            t.off_begin = (long long)(lv.tile_off - pool_words);
            t.result_begin = results;
            t.tile_stride = 1;
            t.independent = 0;
            t.store = 1;
            t.emap = nullptr;
            t.wcap = nullptr;
            t.clean_chunks = nullptr;
            t.cmax = t.sabs = 0.0;
            t.qexp = p.qexp;
            t.batch = kLeanBatch;
            r.result_begin = results;
            units += nt * t.n_groups;
            recs += nt * np;
            results += np;
            points.insert(points.end(), r.lambdas.begin(), r.lambdas.end());
            tasks.push_back(t);
            if (r.comp != nullptr) {
                // final compaction at lambdas[0]: its length is only known after the round
                r.mark = ls.pool_at;
                r.capacity = lv.m + 2;
                r.out_s = (double *)lean_alloc(ls, (size_t)r.capacity * sizeof(double));
                r.out_orig = (int *)lean_alloc(ls, (size_t)r.capacity * sizeof(int));
                r.sep = separator_score(r.lambdas[0], p.gamma);
                if (r.out_s != nullptr && r.out_orig != nullptr) {
                    LeanCompactTask ct;
                    ct.s = lv.s;
                    ct.orig = lv.orig;
                    ct.m = lv.m;
                    ct.n_tiles = nt;
                    ct.block_begin = post_blocks;
                    ct.bits = lv.bits;
                    ct.tile_off = lv.tile_off;
                    ct.sep = r.sep;
                    ct.out_s = r.out_s;
                    ct.out_orig = r.out_orig;
                    ct.capacity = r.capacity;
                    post.push_back(ct);
                    post_blocks += nt;
                } else {
                    ls.pool_at = r.mark;
                    r.out_s = nullptr;
                }
            }
        }
        // Penalties per workgroup: a workgroup's time grows with what it carries (about 6 us + 4 us per penalty), a
        // round's with the number of waves of workgroups the device needs (512 at a time).  While the whole round fits
        // at once, carry less per workgroup.
        const bool adapt = env("ROCCO_HIP_LEAN_BATCH") == nullptr || std::atoi(env("ROCCO_HIP_LEAN_BATCH")) != 0;
        auto rebatch = [](std::vector<LeanTask> &ts, int full, int &total_units) {
            for (int b = 2; b < full; b *= 2) {
                long long u = 0;
                for (const LeanTask &t : ts) {
                    u += (long long)t.n_tiles * ((t.n_points + b - 1) / b);
                }
                if (u <= 512) {
                    int at = 0;
                    for (LeanTask &t : ts) {
                        t.batch = b;
                        t.n_groups = (t.n_points + b - 1) / b;
                        t.unit_begin = at;
                        at += t.n_tiles * t.n_groups;
                    }
                    total_units = at;
                    return;
                }
            }
        };
        if (adapt) {
            rebatch(tasks, kLeanBatch, units);
            rebatch(model_tasks, kLeanModelBatch, model_units);
        }
        lean_units += units + model_units;
        if (model_tiles + model_tiles_answered_ > 0) {
            model_tiles_last_round_ = model_tiles + model_tiles_answered_;
        }
        const int n_bound_tasks = (int)tasks.size();
        tasks.insert(tasks.end(), model_tasks.begin(), model_tasks.end());

        // descriptors: [pre][tasks (bound, then rounding-model)][points][post]
        const size_t b_pre = align_up(pre.size() * sizeof(LeanCompactTask), 256);
        const size_t b_tasks = align_up(tasks.size() * sizeof(LeanTask), 256);
        const size_t b_points = align_up(points.size() * sizeof(double), 256);
        const size_t b_post = align_up(post.size() * sizeof(LeanCompactTask), 256);
        const size_t b_wcap = align_up(wcap_tasks.size() * sizeof(LeanWcapTask), 256);
        const size_t desc = b_pre + b_tasks + b_points + b_post + b_wcap;
        if ((rc = solver_->dev_lean_desc.reserve(desc + 256)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_lean_stage.reserve(desc + 256)) != ROCCO_HIP_OK) return rc;
        char *h = (char *)solver_->host_lean_stage.ptr;
        char *d = (char *)solver_->dev_lean_desc.ptr;
        if (!pre.empty()) std::memcpy(h, pre.data(), pre.size() * sizeof(LeanCompactTask));
        std::memcpy(h + b_pre, tasks.data(), tasks.size() * sizeof(LeanTask));
        std::memcpy(h + b_pre + b_tasks, points.data(), points.size() * sizeof(double));
        if (!post.empty()) std::memcpy(h + b_pre + b_tasks + b_points, post.data(), post.size() * sizeof(LeanCompactTask));
        if (!wcap_tasks.empty()) std::memcpy(h + b_pre + b_tasks + b_points + b_post, wcap_tasks.data(), wcap_tasks.size() * sizeof(LeanWcapTask));
        const double le1 = now_us();
        ROCCO_HIP_TRY(hipMemcpyAsync(d, h, desc, hipMemcpyHostToDevice, stream_));
        const double le2 = now_us();
        t_lean_cpu_ += le1 - le0;
        t_lean_h2d_ += le2 - le1;
        struct Launches {
            double &acc;
            double t0;
            ~Launches() { acc += now_us() - t0; }
        } launches{t_lean_launch_, le2};
        if (!wcap_tasks.empty()) {
            if ((rc = launch_lean_wcap((const LeanWcapTask *)(d + b_pre + b_tasks + b_points + b_post), (int)wcap_tasks.size(), wcap_blocks,
                                       stream_)) != ROCCO_HIP_OK) return rc;
        }

        // Round scratch.  Tickets, the error word and the hand-off granules live in a buffer of their own that every
        // round leaves as it found it (lean_finish_kernel restores what the round used: tickets and granules all-ones,
        // error zero), so no fill is issued per round; it is initialised when it grows or after a failed round.  The
        // records are plain scratch.  Results and the error word are written straight into pinned host memory by the
        // finish kernel.
        const char *fills = env("ROCCO_HIP_LEAN_FILLS");
        const bool self_reset = fills == nullptr || std::atoi(fills) == 0;
        const size_t b_look = align_up(256 + (size_t)recs * 4 * sizeof(unsigned long long), 256);
        const size_t b_recs = align_up((size_t)recs * sizeof(LeanTileRec), 256);
        const size_t b_res = align_up((size_t)results * sizeof(LeanResult) + 64, 256);
        {
            const void *old_ptr = solver_->dev_lean_look.ptr;
            const size_t old_bytes = solver_->dev_lean_look.bytes;
            if ((rc = solver_->dev_lean_look.reserve(b_look)) != ROCCO_HIP_OK) return rc;
            if (!self_reset || solver_->lean_look_dirty != 0 || solver_->dev_lean_look.ptr != old_ptr || solver_->dev_lean_look.bytes != old_bytes) {
                ROCCO_HIP_TRY(hipMemsetAsync(solver_->dev_lean_look.ptr, 0xFF, solver_->dev_lean_look.bytes, stream_));
                ROCCO_HIP_TRY(hipMemsetAsync((char *)solver_->dev_lean_look.ptr + 128, 0, 4, stream_));
                solver_->lean_look_dirty = 0;
            }
        }
        if ((rc = solver_->dev_lean_round.reserve(b_recs + 256)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_lean_back.reserve(b_res)) != ROCCO_HIP_OK) return rc;
        char *look = (char *)solver_->dev_lean_look.ptr;
        unsigned *error = (unsigned *)(look + 128);
        LeanResult *results_host = (LeanResult *)solver_->host_lean_back.ptr;
        unsigned *error_host = (unsigned *)((char *)solver_->host_lean_back.ptr + (size_t)results * sizeof(LeanResult));
        error_host[0] = 0u;
        error_host[1] = 0u;  // second slot: the word of a compaction launched BEHIND the finish kernel
        if (!pre.empty()) {
            if ((rc = launch_lean_compact((const LeanCompactTask *)d, (int)pre.size(), pre_blocks, error, stream_)) != ROCCO_HIP_OK) return rc;
        }
        LeanLaunch L;
        L.tasks = (const LeanTask *)(d + b_pre);
        L.n_tasks = n_bound_tasks;
        L.n_units = units;
        L.points = (const double *)(d + b_pre + b_tasks);
        L.ticket = (unsigned *)look;
        L.look = (unsigned long long *)(look + 256);
        L.recs = (LeanTileRec *)solver_->dev_lean_round.ptr;
        L.bits = (unsigned *)solver_->dev_lean_pool.ptr;
        L.tile_off = (unsigned *)solver_->dev_lean_pool.ptr;
        L.results = results_host;
        L.error = error;
        L.error_out = error_host;
        L.self_reset = 1;
        L.pad = 0;
        L.ctl = nullptr;
        solver_->lean_look_dirty = 1;  // until the finish kernel that restores the scratch is in the stream
        if ((rc = launch_lean_eval(L, stream_)) != ROCCO_HIP_OK) return rc;
        if (!model_tasks.empty()) {
            LeanLaunch M = L;
            M.tasks = L.tasks + n_bound_tasks;
            M.n_tasks = (int)model_tasks.size();
            M.n_units = model_units;
            M.ticket = (unsigned *)look + 2;
            if ((rc = launch_lean_model(M, stream_)) != ROCCO_HIP_OK) return rc;
        }
        L.n_tasks = (int)tasks.size();  // the finish launch closes every task's fill
        if ((rc = launch_lean_finish(L, results, stream_)) != ROCCO_HIP_OK) return rc;
        solver_->lean_look_dirty = 0;
        if (!post.empty()) {
            // (rare since levels are adopted: a final compaction behind the finish kernel reports through a copy)
            if ((rc = launch_lean_compact((const LeanCompactTask *)(d + b_pre + b_tasks + b_points), (int)post.size(), post_blocks,
                                          error, stream_)) != ROCCO_HIP_OK) return rc;
            // its own pinned slot: the finish kernel has already written the evaluation's word to slot 0 (and cleared the
            // device word), and a spin-limit failure recorded there must survive this copy
            ROCCO_HIP_TRY(hipMemcpyAsync(error_host + 1, error, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
            ROCCO_HIP_TRY(hipMemsetAsync(error, 0, sizeof(unsigned), stream_));
        }
        lean_result_count_ = results;
        lean_inflight_ = reqs;
        if (env("ROCCO_HIP_DEBUG") != nullptr) {
            std::fprintf(stderr, "[lean round %d] %zu tasks (%zu rounding-model), %d + %d workgroups, %zu compactions before, %zu after\n",
                         lean_rounds, tasks.size(), model_tasks.size(), units, model_units, pre.size(), post.size());
        }
        return ROCCO_HIP_OK;
    }

    int lean_consume()
    {
        if (lean_inflight_.empty()) {
            return ROCCO_HIP_OK;
        }
        {
            const int rcm = lean_map_check();
            if (rcm != ROCCO_HIP_OK) return rcm;
        }
        if (model_chain_inflight_) {
            return model_chain_consume();
        }
        const LeanResult *res = (const LeanResult *)solver_->host_lean_back.ptr;
        const unsigned *error_words = (const unsigned *)((const char *)solver_->host_lean_back.ptr + (size_t)lean_result_count_ * sizeof(LeanResult));
        const unsigned error = error_words[0] | error_words[1];
        if (error & 1u) {
            solver_->lean_look_dirty = 1;  // tiles that gave up left granules behind
            set_last_error("lean evaluation: a tile waited for its predecessor beyond the spin limit");
            return ROCCO_HIP_EHIP;
        }
        for (LeanReq &r : lean_inflight_) {
            DevProblem &p = probs[r.problem];
            LeanState &ls = lean_[r.problem];
            if (r.pilot) {
                r.probe->results.assign(r.lambdas.size(), ProbeResult());
                for (size_t i = 0; i < r.lambdas.size(); ++i) {
                    r.probe->results[i].count = (long long)std::llround((double)res[r.result_begin + (int)i].count * r.pilot_scale);
                }
                continue;
            }
            if (r.model) {
                r.probe->results.assign(r.lambdas.size(), ProbeResult());
                size_t open = 0;
                long long why = 0;
                for (size_t i = 0; i < r.lambdas.size(); ++i) {
                    const LeanResult &lr = res[r.result_begin + (int)i];
                    r.probe->results[i].count = lr.count;
                    why |= lr.flags;
                    if (lr.flags != 0) {
                        // not certified here: the full kernels decide (model_fallback)
                        r.probe->results[i].uncertain = lr.flags;  // (reason mask, see lean_model_kernel)
                        r.probe->results[i].effect = (long long)p.n + 1;
                        model_open_.emplace_back(r.probe, i);
                        ++open;
                    }
                }
                lean_model_points += (long long)r.lambdas.size();
                lean_model_open += (long long)open;
                if (env("ROCCO_HIP_DEBUG") != nullptr) {
                    std::fprintf(stderr, "[lean model] problem %zu (n=%zu): %zu penalties, first %.17g -> count %lld, %zu not certified (reasons %lld)\n",
                                 r.problem, p.n, r.lambdas.size(), r.lambdas[0], res[r.result_begin].count, open, why);
                }
                continue;
            }
            LeanLevel &lv = ls.levels.back();
            const size_t np = r.lambdas.size();
            lv.pts = r.lambdas;
            lv.child_len.resize(np);
            for (size_t i = 0; i < np; ++i) {
                lv.child_len[i] = res[r.result_begin + (int)i].child_len;
            }
            lv.has_eval = true;
            if (r.probe != nullptr) {
                r.probe->results.assign(np, ProbeResult());
                for (size_t i = 0; i < np; ++i) {
                    r.probe->results[i].count = res[r.result_begin + (int)i].count;
                }
            }
            if (env("ROCCO_HIP_DEBUG") != nullptr) {
                std::fprintf(stderr, "[lean] problem %zu level %zu (m=%lld, base %.17g): %zu penalties, first %.17g -> count %lld, child %lld\n",
                             r.problem, ls.levels.size() - 1, lv.m, lv.base, np, r.lambdas[0], res[r.result_begin].count,
                             res[r.result_begin].child_len);
            }
            if (r.comp != nullptr) {
                const long long cl = res[r.result_begin].child_len;
                const bool fits = r.out_s != nullptr && cl >= 1 && cl <= r.capacity && !(error & 2u);
                // worth it only when the compacted problem is much smaller than the caller's
                if (fits && 10 * cl <= 6 * (long long)p.n && adopt_level(r.problem, r.out_s, r.out_orig, cl, r.sep)) {
                    r.comp->done = true;
                    r.comp->n_new = (size_t)cl;
                    r.comp->score_floor = r.sep;
                }
                if (!r.comp->done && env("ROCCO_HIP_DEBUG") != nullptr) {
                    std::fprintf(stderr, "[lean] problem %zu: final compaction declined (child %lld of %zu)\n", r.problem, cl, p.n);
                }
            }
        }
        lean_inflight_.clear();
        return ROCCO_HIP_OK;
    }

    int lean_result_count_ = 0;
    std::vector<std::pair<ProbeRequest *, size_t>> model_open_;  // penalties the model kernel left open
    long long lean_model_points = 0, lean_model_open = 0;

    // penalties the lean model kernel could not certify: once more through the full kernels (rare)
    int model_fallback()
    {
        if (model_open_.empty() || model_any_) {
            model_open_.clear();
            return ROCCO_HIP_OK;
        }
        std::vector<ProbeRequest> again;
        std::vector<std::vector<std::pair<ProbeRequest *, size_t>>> back;
        for (const auto &o : model_open_) {
            size_t k = 0;
            while (k < again.size() && back[k][0].first != o.first) {
                ++k;
            }
            if (k == again.size()) {
                ProbeRequest q;
                q.problem = o.first->problem;
                again.push_back(q);
                back.emplace_back();
            }
            again[k].lambdas.push_back(o.first->lambdas[o.second]);
            back[k].push_back(o);
        }
        model_open_.clear();
        force_full_ = true;
        std::vector<RoundTask> tasks;
        add_probe_tasks(again, tasks);
        const int rc = run_round(tasks);
        force_full_ = false;
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
        for (size_t k = 0; k < again.size(); ++k) {
            for (size_t i = 0; i < back[k].size(); ++i) {
                back[k][i].first->results[back[k][i].second] = again[k].results[i];
            }
        }
        return ROCCO_HIP_OK;
    }

    // switch a problem over to one of its compacted levels
    bool adopt_level(size_t problem, const double *level_s, const int *level_orig, long long m, double sep)
    {
        DevProblem &p = probs[problem];
        LeanState &ls = lean_[problem];
        uint8_t *sol = (uint8_t *)lean_alloc(ls, (size_t)m);
        if (sol == nullptr) {
            return false;
        }
        p.orig_scores = p.scores;
        p.orig_n = p.n;
        p.orig_solution = p.solution;
        p.compacted = true;
        p.scores = level_s;
        p.n = (size_t)m;
        p.solution = sol;
        p.lean_orig = level_orig;
        p.emap = nullptr;
        p.frz_valid = false;
        p.smin = std::min(p.smin, sep);
        p.sabs = std::max(p.sabs, std::fabs(sep));
        p.qexp = grid_exponent(std::max(p.cmax, 0.0), p.smin, p.smax);
        return true;
    }

    // the deepest level built at or below the requested penalty serves as the compacted problem: no device work
    bool compact_now(CompactRequest &req) override
    {
        req.done = false;
        if (!lean_ready_ || !lean_eligible(req.problem) || req.problem >= lean_.size()) {
            return false;
        }
        LeanState &ls = lean_[req.problem];
        for (size_t k = ls.levels.size(); k-- > 1;) {
            const LeanLevel &lv = ls.levels[k];
            if (lv.base <= req.lambda_base) {
                if (10 * lv.m > 6 * (long long)probs[req.problem].n) {
                    return false;  // not worth it: let the regular request evaluate and compact at the penalty itself
                }
                if (!adopt_level(req.problem, lv.s, lv.orig, lv.m, lv.sep)) {
                    return false;
                }
                req.done = true;
                req.n_new = (size_t)lv.m;
                req.score_floor = lv.sep;
                if (env("ROCCO_HIP_DEBUG") != nullptr) {
                    std::fprintf(stderr, "[lean] problem %zu: level %zu (m=%lld, base %.17g) adopted for penalties >= %.17g\n",
                                 req.problem, k, lv.m, lv.base, req.lambda_base);
                }
                return true;
            }
        }
        return false;
    }

    int compact(std::vector<CompactRequest> &reqs) override
    {
        std::vector<ProbeRequest> none;
        int rc;
        if ((rc = lean_submit(reqs, none)) != ROCCO_HIP_OK) return rc;
        ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
        return lean_consume();
    }

    // queue the lean part of an iteration; returns the probes that stay with the general kernels
    int lean_submit(std::vector<CompactRequest> &compacts, std::vector<ProbeRequest> &probes)
    {
        std::vector<LeanReq> reqs;
        for (CompactRequest &c : compacts) {
            c.done = false;
            if (!lean_eligible(c.problem)) {
                continue;
            }
            LeanReq r;
            r.problem = c.problem;
            r.lambdas = {c.lambda_base};
            r.comp = &c;
            reqs.push_back(r);
        }
        model_tiles_answered_ = 0;
        for (ProbeRequest &q : probes) {
            if (lean_takes(*this, q)) {
                if (!q.bound && !model_any_ && q.problem < facts_.size() && (!facts_[q.problem].empty() || in_running_chain(q.problem))) {
                    // rounding-model counts the evaluator already holds, or that the running chain is about to publish
                    // (model_chain.h): no device work
                    std::vector<ProbeResult> res;
                    bool all = facts_answer(q, res);
                    while (!all && in_running_chain(q.problem)) {
                        bool got = false;
                        const int rc = model_chain_ingest(true, &got);
                        if (rc != ROCCO_HIP_OK) return rc;
                        all = facts_answer(q, res);
                    }
                    if (all) {
                        q.results = res;
                        model_chain_hits += (long long)q.lambdas.size();
                        lean_model_points += (long long)q.lambdas.size();
                        model_tiles_answered_ += (long long)((probs[q.problem].n + kLeanTile - 1) / kLeanTile);
                        continue;
                    }
                }
                LeanReq r;
                r.problem = q.problem;
                r.lambdas = q.lambdas;
                r.probe = &q;
                r.model = !q.bound;
                r.pilot = q.bound && q.pilot && can_pilot(q.problem);
                reqs.push_back(r);
            }
        }
        if (reqs.empty() && model_tiles_answered_ > 0) {
            model_tiles_last_round_ = model_tiles_answered_;
        }
        return lean_enqueue(reqs);
    }

    static bool lean_takes(const HipEvaluator &ev, const ProbeRequest &q)
    {
        if (q.lambdas.empty() || q.lambdas.size() > (size_t)kLeanMaxPoints) {
            return false;
        }
        return q.bound ? ev.lean_eligible(q.problem) : (!q.pilot && ev.model_eligible(q.problem));
    }

    int round_all(std::vector<CompactRequest> &compacts, std::vector<MapRequest> &maps, std::vector<WindowRequest> &surveys,
                  std::vector<ProbeRequest> &probes, std::vector<WindowRequest> &windows,
                  std::vector<SpineRequest> &spines) override
    {
        int rc;
        const double tr0 = now_us();
        struct Total {
            double &acc;
            double t0;
            ~Total() { acc += now_us() - t0; }
        } total{t_round_, tr0};
        mark("round begins");
        if ((rc = lean_submit(compacts, probes)) != ROCCO_HIP_OK) return rc;
        mark("round: lean part submitted");
        // a window that only certifies and writes the solution of ONE penalty, asked of a problem whose chain of
        // rounding-model rounds ended at that very penalty: the chain's last evaluation there was certified class by class
        // and lean_write_solutions_kernel has turned it into the solution bytes (model_chain.h) -- nothing left to run
        window_answered_.assign(windows.size(), 0);
        std::vector<size_t> answered_problems;
        if (!windows.empty()) {
            bool any_in_chain = false;
            for (const WindowRequest &w : windows) {
                any_in_chain = any_in_chain || in_running_chain(w.problem);
            }
            if (any_in_chain && (rc = model_chain_drain()) != ROCCO_HIP_OK) return rc;
            for (size_t i = 0; i < windows.size(); ++i) {
                WindowRequest &w = windows[i];
                if (w.problem < written_.size() && written_[w.problem].valid && w.lambda_lo == w.lambda_hi &&
                    w.lambda_lo == written_[w.problem].penalty) {
                    w.result = WindowResult();
                    w.result.count_lo = w.result.count_hi = written_[w.problem].count;
                    w.result.n_diff = 0;
                    window_answered_[i] = 1;
                    answered_problems.push_back(w.problem);
                    ++model_chain_windows_answered;
                }
            }
        }
        std::vector<RoundTask> tasks;
        if ((rc = add_map_tasks(maps, tasks)) != ROCCO_HIP_OK) return rc;
        if ((rc = add_survey_tasks(surveys, tasks)) != ROCCO_HIP_OK) return rc;
        add_probe_tasks(probes, tasks);
        add_window_tasks(windows, tasks);
        if ((rc = add_spine_tasks(spines, tasks)) != ROCCO_HIP_OK) return rc;
        if (tasks.empty()) {
            if (lean_wait_needed()) {
                const double ts0 = now_us();
                ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
                t_wait_ += now_us() - ts0;
            }
        } else {
            // windows that each write the solution of one penalty and nothing else: the calibration's last round when every
            // step was decided -- their objectives ride behind them
            std::vector<size_t> final_problems;
            const char *pf = env("ROCCO_HIP_PREFETCH_OBJECTIVE");
            const size_t open_windows = windows.size() - answered_problems.size();
            if ((pf == nullptr || std::atoi(pf) != 0) && open_windows > 0 && tasks.size() == open_windows && lean_inflight_.empty()) {
                for (size_t i = 0; i < windows.size(); ++i) {
                    if (!window_answered_[i] && windows[i].lambda_lo == windows[i].lambda_hi) {
                        final_problems.push_back(windows[i].problem);
                    }
                }
                if (final_problems.size() != open_windows) {
                    final_problems.clear();
                }
            }
            // (a round of maps alone is not waited for unless lean results of this iteration are read below)
            if ((rc = run_round(tasks, !lean_wait_needed(), final_problems.empty() ? nullptr : &final_problems)) != ROCCO_HIP_OK) {
                return rc;
            }
            if (!final_problems.empty()) {
                std::vector<char> certified;
                for (size_t i = 0; i < windows.size(); ++i) {
                    if (!window_answered_[i]) {
                        certified.push_back((!windows[i].result.overflow && windows[i].result.n_diff == 0) ? 1 : 0);
                    }
                }
                objective_collect(certified);
            }
        }
        const double tc0 = now_us();
        mark("round: device work waited for");
        if ((rc = lean_consume()) != ROCCO_HIP_OK) return rc;
        if ((rc = model_fallback()) != ROCCO_HIP_OK) return rc;
        adopt_maps(maps);
        t_consume_ += now_us() - tc0;
        mark("round ends");
        ++rounds_all;
        return ROCCO_HIP_OK;
    }

    // smin, smax, cmin, cmax of every problem (one stats pass over the batch)
    int compute_stats(std::vector<double> &out)
    {
        const size_t B = probs.size();
        out.assign(5 * B, 0.0);
        if (B == 0) {
            return ROCCO_HIP_OK;
        }
        std::vector<int2> blockmap;
        for (size_t b = 0; b < B; ++b) {
            const int nb = (int)((probs[b].n + kFastBlockLoci - 1) / kFastBlockLoci);
            for (int k = 0; k < nb; ++k) {
                blockmap.push_back(make_int2((int)b, k));
            }
        }
        const size_t nbt = blockmap.size();
        const size_t bytes_tasks = align_up(B * sizeof(StatsTask), 256);
        const size_t bytes_map = align_up(nbt * sizeof(int2), 256);
        const size_t bytes_part = align_up(nbt * 5 * sizeof(double), 256);
        const size_t bytes_out = align_up(B * 5 * sizeof(double), 256);
        int rc;
        if ((rc = solver_->dev_misc.reserve(bytes_tasks + bytes_map + bytes_part + bytes_out)) != ROCCO_HIP_OK) return rc;
        if ((rc = stage_wait()) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_stage.reserve(bytes_tasks + bytes_map)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_back.reserve(bytes_out)) != ROCCO_HIP_OK) return rc;
        char *h = (char *)solver_->host_stage.ptr;
        StatsTask *ht = (StatsTask *)h;
        for (size_t b = 0; b < B; ++b) {
            ht[b].scores = probs[b].scores;
            ht[b].switch_costs = probs[b].costs;
            ht[b].n = (long long)probs[b].n;
        }
        std::memcpy(h + bytes_tasks, blockmap.data(), nbt * sizeof(int2));
        char *dv = (char *)solver_->dev_misc.ptr;
        ROCCO_HIP_TRY(hipMemcpyAsync(dv, h, bytes_tasks + bytes_map, hipMemcpyHostToDevice, stream_));
        double *d_part = (double *)(dv + bytes_tasks + bytes_map);
        double *d_out = (double *)(dv + bytes_tasks + bytes_map + bytes_part);
        if ((rc = launch_stats((const StatsTask *)dv, (int)B, (const int2 *)(dv + bytes_tasks), (int)nbt, d_part,
                               d_out, stream_)) != ROCCO_HIP_OK) {
            return rc;
        }
        ROCCO_HIP_TRY(hipMemcpyAsync(solver_->host_back.ptr, d_out, B * 5 * sizeof(double), hipMemcpyDeviceToHost, stream_));
        ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
        std::memcpy(out.data(), solver_->host_back.ptr, B * 5 * sizeof(double));
        return ROCCO_HIP_OK;
    }

    // ---- the threshold search as one chain of launches (chain.h) ----
    // Queues the statistics pass and `rounds` rounds of [director, compaction, evaluation, finish], waits ONCE, and turns
    // the report into (a) the statistics prepare() needs, (b) the levels of every problem as lean_enqueue would have left
    // them, (c) the certified counts for calibrate_batch.  `ran` false: nothing was queued (the caller takes the
    // statistics pass of its own).
    int chain_rounds_run = 0;
    int chain_search(const std::vector<ChainProblem> &problems, const SearchOptions &opt, std::vector<double> &stats_out,
                     std::vector<Presearch> &pre, bool &ran)
    {
        ran = false;
        const double tc0 = now_us();
        const size_t B = probs.size();
        const char *flag = std::getenv("ROCCO_HIP_CHAIN");
        if ((flag != nullptr && std::atoi(flag) == 0) || solver_->lean == 0 || B == 0 || B > (size_t)kChainMaxProblems ||
            !opt.use_bounds || opt.force_exact || !opt.use_compaction) {
            return ROCCO_HIP_OK;
        }
        bool any = false;
        long long tiles0 = 0;
        for (size_t b = 0; b < B; ++b) {
            any = any || lean_eligible(b);
            tiles0 += (long long)((probs[b].n + kLeanTile - 1) / kLeanTile);
            if (probs[b].n >= ((size_t)1 << 31)) {
                return ROCCO_HIP_OK;
            }
        }
        if (!any) {
            return ROCCO_HIP_OK;
        }
        // A chained round costs its director and three launch boundaries (~45 us), about what a host-sequenced round costs
        // in turn-around; what the chain gains is in the rounds' shape (three short pilot rounds of eight penalties, two
        // penalties on the pass over every locus), which pays on a genome (7 555 tiles: 2.80 -> 2.60 ms per calibration) and not
        // on a rank's shard of one (973 tiles at N = 8: 0.94 -> 1.07 ms).  Below `min_tiles` the host sequences the rounds.
        const long long min_tiles = std::getenv("ROCCO_HIP_CHAIN_MIN_TILES") ? std::atoll(std::getenv("ROCCO_HIP_CHAIN_MIN_TILES")) : 4096;
        if (tiles0 < min_tiles && flag == nullptr) {
            return ROCCO_HIP_OK;
        }
        int rc;
        if ((rc = lean_prepare()) != ROCCO_HIP_OK) return rc;
        ChainTuning tune;
        tune.pilot_rounds = std::getenv("ROCCO_HIP_CHAIN_PILOT_ROUNDS") ? std::atoi(std::getenv("ROCCO_HIP_CHAIN_PILOT_ROUNDS")) : 6;
        tune.pilot_points = std::getenv("ROCCO_HIP_CHAIN_PILOT_POINTS") ? std::atoi(std::getenv("ROCCO_HIP_CHAIN_PILOT_POINTS")) : 8;
        if (opt.pilot_rounds <= 0) {
            tune.pilot_rounds = 0;
        }
        tune.wgs = std::getenv("ROCCO_HIP_CHAIN_WGS") ? std::max(64, std::atoi(std::getenv("ROCCO_HIP_CHAIN_WGS"))) : 512;
        tune.pilot_wgs = std::getenv("ROCCO_HIP_CHAIN_PILOT_WGS") ? std::max(64, std::atoi(std::getenv("ROCCO_HIP_CHAIN_PILOT_WGS"))) : 512;
        tune.pilot_tiles = std::getenv("ROCCO_HIP_CHAIN_PILOT_TILES") ? std::max(2, std::atoi(std::getenv("ROCCO_HIP_CHAIN_PILOT_TILES"))) : 16;
        tune.big_points = std::getenv("ROCCO_HIP_CHAIN_BIG_POINTS") ? std::atoi(std::getenv("ROCCO_HIP_CHAIN_BIG_POINTS")) : 2;
        tune.search_gate = opt.search_gate;
        tune.survey_gate = opt.survey_gate;
        tune.interpolate = std::getenv("ROCCO_HIP_CHAIN_INTERP") ? std::atoi(std::getenv("ROCCO_HIP_CHAIN_INTERP")) : 0;
        tune.spread = std::getenv("ROCCO_HIP_CHAIN_SPREAD") ? std::atof(std::getenv("ROCCO_HIP_CHAIN_SPREAD")) : 0.005;
        if (!(tune.spread > 0.0 && tune.spread < 0.5)) {
            tune.spread = 0.005;
        }
        tune.soft_mult = std::getenv("ROCCO_HIP_CHAIN_SOFT") ? std::atof(std::getenv("ROCCO_HIP_CHAIN_SOFT")) : 0.8;
        {
            std::vector<double> mults = {2.2, 1.2};
            if (const char *e = std::getenv("ROCCO_HIP_CHAIN_LEVELS")) {
                mults.clear();
                for (const char *q = e; *q != '\0';) {
                    char *end = nullptr;
                    const double v = std::strtod(q, &end);
                    if (end == q) break;
                    if (v > 0.0) mults.push_back(v);
                    q = (*end == ',') ? end + 1 : end;
                }
            }
            tune.n_mults = (int)std::min(mults.size(), (size_t)kChainMaxMults);
            for (int k = 0; k < kChainMaxMults; ++k) {
                tune.mults[k] = (k < tune.n_mults) ? mults[(size_t)k] : 0.0;
            }
        }
        const int R = std::getenv("ROCCO_HIP_CHAIN_ROUNDS") ? std::max(1, std::atoi(std::getenv("ROCCO_HIP_CHAIN_ROUNDS"))) : 12;

        // ---- statistics pass: descriptors as in compute_stats ----
        std::vector<int2> blockmap;
        for (size_t b = 0; b < B; ++b) {
            const int nb = (int)((probs[b].n + kFastBlockLoci - 1) / kFastBlockLoci);
            for (int k = 0; k < nb; ++k) {
                blockmap.push_back(make_int2((int)b, k));
            }
        }
        const size_t nbt = blockmap.size();
        // ---- one device buffer: [ctl][inputs][stats tasks][blockmap] (uploaded) [probs][stats] (downloaded) [rest] ----
        size_t off = 0;
        auto carve = [&off](size_t bytes) {
            const size_t at = off;
            off += align_up(bytes, 256);
            return at;
        };
        const size_t o_ctl = carve(sizeof(LeanRoundCtl));
        const size_t o_in = carve(B * sizeof(ChainInput));
        const size_t o_stasks = carve(B * sizeof(StatsTask));
        const size_t o_bmap = carve(nbt * sizeof(int2));
        const size_t up_bytes = off;
        const size_t o_probs = carve(B * sizeof(ChainProb));
        const size_t o_stats = carve(B * 5 * sizeof(double));
        const size_t o_ctl_back = carve(sizeof(LeanRoundCtl));
        const size_t down_bytes = off - o_probs;
        const size_t o_hot = carve(B * sizeof(ChainHot));
        const size_t o_pilot = carve(B * sizeof(ChainPilot));
        const size_t o_evals = carve(B * sizeof(ChainEvals));
        const size_t o_tasks = carve(B * sizeof(LeanTask));
        const size_t o_points = carve(B * kLeanMaxPoints * sizeof(double));
        const size_t o_results = carve(B * kLeanMaxPoints * sizeof(LeanResult));
        const size_t o_pre = carve(B * sizeof(LeanCompactTask));
        const size_t o_part = carve(nbt * 5 * sizeof(double));
        const bool want_trace = std::getenv("ROCCO_HIP_CHAIN_TRACE") != nullptr;
        const size_t o_trace = carve((size_t)(R + 2) * 8 * sizeof(long long));
        if ((rc = solver_->dev_chain.reserve(off + 256)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_chain.reserve(std::max(up_bytes, down_bytes) + 256)) != ROCCO_HIP_OK) return rc;
        char *dv = (char *)solver_->dev_chain.ptr;
        char *h = (char *)solver_->host_chain.ptr;
        std::memset(h, 0, o_in);
        ChainInput *hin = (ChainInput *)(h + o_in);
        StatsTask *hst = (StatsTask *)(h + o_stasks);
        for (size_t b = 0; b < B; ++b) {
            const DevProblem &p = probs[b];
            hin[b].scores = p.scores;
            hin[b].n = (long long)p.n;
            hin[b].gamma = p.gamma;
            hin[b].target = std::max(0LL, std::min(problems[b].target_count, (long long)p.n));
            hin[b].pool_begin = lean_[b].pool_begin;
            hin[b].pool_end = lean_[b].pool_end;
            hin[b].allowed = lean_eligible(b) ? 1 : 0;
            hin[b].can_pilot = can_pilot(b) ? 1 : 0;
            hst[b].scores = p.scores;
            hst[b].switch_costs = p.costs;
            hst[b].n = (long long)p.n;
        }
        std::memcpy(h + o_bmap, blockmap.data(), nbt * sizeof(int2));
        ROCCO_HIP_TRY(hipMemcpyAsync(dv, h, up_bytes, hipMemcpyHostToDevice, stream_));
        if ((rc = launch_stats((const StatsTask *)(dv + o_stasks), (int)B, (const int2 *)(dv + o_bmap), (int)nbt, (double *)(dv + o_part),
                               (double *)(dv + o_stats), stream_)) != ROCCO_HIP_OK) {
            return rc;
        }

        // ---- round scratch (as lean_enqueue keeps it) ----
        const long long rec_cap = 8 * (tiles0 + (long long)tune.wgs) + 64 * (long long)B;
        const size_t b_look = align_up(256 + (size_t)rec_cap * 4 * sizeof(unsigned long long), 256);
        {
            const void *old_ptr = solver_->dev_lean_look.ptr;
            const size_t old_bytes = solver_->dev_lean_look.bytes;
            if ((rc = solver_->dev_lean_look.reserve(b_look)) != ROCCO_HIP_OK) return rc;
            if (solver_->lean_look_dirty != 0 || solver_->dev_lean_look.ptr != old_ptr || solver_->dev_lean_look.bytes != old_bytes) {
                ROCCO_HIP_TRY(hipMemsetAsync(solver_->dev_lean_look.ptr, 0xFF, solver_->dev_lean_look.bytes, stream_));
                ROCCO_HIP_TRY(hipMemsetAsync((char *)solver_->dev_lean_look.ptr + 128, 0, 4, stream_));
                solver_->lean_look_dirty = 0;
            }
        }
        if ((rc = solver_->dev_lean_round.reserve(align_up((size_t)rec_cap * sizeof(LeanTileRec), 256) + 256)) != ROCCO_HIP_OK) return rc;
        char *look = (char *)solver_->dev_lean_look.ptr;

        ChainArgs A;
        A.n_problems = (int)B;
        A.rec_capacity = (int)std::min<long long>(rec_cap, 0x7FFFFFFF);
        A.inputs = (const ChainInput *)(dv + o_in);
        A.probs = (ChainProb *)(dv + o_probs);
        A.hot = (ChainHot *)(dv + o_hot);
        A.pilot = (ChainPilot *)(dv + o_pilot);
        A.evals = (ChainEvals *)(dv + o_evals);
        A.stats = (const double *)(dv + o_stats);
        A.ctl = (LeanRoundCtl *)(dv + o_ctl);
        A.tasks = (LeanTask *)(dv + o_tasks);
        A.points = (double *)(dv + o_points);
        A.results = (LeanResult *)(dv + o_results);
        A.pre = (LeanCompactTask *)(dv + o_pre);
        A.pool = (char *)solver_->dev_lean_pool.ptr;
        A.trace = want_trace ? (long long *)(dv + o_trace) : nullptr;
        A.tune = tune;
        // the report comes through host-coherent memory as soon as the searches have ended (chain.h: ChainArgs::follow);
        // ROCCO_HIP_CHAIN_FOLLOW=0: through a copy at the end of the stream
        const bool follow = std::getenv("ROCCO_HIP_CHAIN_FOLLOW") == nullptr || std::atoi(std::getenv("ROCCO_HIP_CHAIN_FOLLOW")) != 0;
        A.follow = nullptr;
        A.follow_words = (int)((o_ctl_back - o_probs) / 8);
        A.follow_ctl_word = A.follow_words;
        if (follow) {
            solver_->host_follow.coherent = true;
            if ((rc = solver_->host_follow.reserve(256 + down_bytes + 256)) != ROCCO_HIP_OK) return rc;
            A.follow = (unsigned long long *)solver_->host_follow.ptr;
            std::memset(solver_->host_follow.ptr, 0, 256);
        }

        LeanLaunch L;
        L.tasks = A.tasks;
        L.n_tasks = 0;
        L.n_units = 0;
        L.points = A.points;
        L.ticket = (unsigned *)look;
        L.look = (unsigned long long *)(look + 256);
        L.recs = (LeanTileRec *)solver_->dev_lean_round.ptr;
        L.bits = (unsigned *)solver_->dev_lean_pool.ptr;
        L.tile_off = (unsigned *)solver_->dev_lean_pool.ptr;
        L.results = A.results;
        L.error = (unsigned *)(look + 128);
        L.error_out = nullptr;
        L.self_reset = 1;
        L.pad = 0;
        L.ctl = A.ctl;
        const int eval_grid = (int)std::min<long long>(512, std::max<long long>(1, tiles0 * 8));
        const int compact_grid = (int)std::min<long long>(1024, std::max<long long>(1, tiles0));
        const int finish_grid = (int)std::min<size_t>(2048, B * (size_t)kLeanMaxPoints);
        solver_->lean_look_dirty = 1;  // until every finish launch that restores the scratch is in the stream
        // Round 5, measured and NOT adopted: a round's compaction, evaluation and finish as ONE launch
        // (lean.h: LeanRoundReset; the director in front of the next round restores what the finish launch restored).  Whole
        // genome, same box, interleaved: three launches 2.71-2.87 ms, one launch 3.29-3.45 (the pairs of a round are finished
        // by the few workgroups that held the last tiles instead of 1 536 at once, and the compactions run two to a CU behind
        // fences), compactions + evaluation as one and the finish apart 3.11-3.18; the rounding-model rounds as one launch
        // 2.71-2.90 against 2.71-2.87.  ROCCO_HIP_CHAIN_FUSED: 0 (default) neither chain, 1 both, 2 the threshold search only,
        // 3 the rounding-model rounds only, 4 the threshold search's compactions + evaluation.
        const int fused_mode = env("ROCCO_HIP_CHAIN_FUSED") == nullptr ? 0 : std::atoi(env("ROCCO_HIP_CHAIN_FUSED"));
        const bool fused = fused_mode == 1 || fused_mode == 2, half_fused = fused_mode == 4;  // (4: compactions + evaluation, finish apart)
        A.reset = LeanRoundReset{nullptr, 0, nullptr, nullptr};
        if (fused || half_fused) {
            const size_t words = (size_t)kLeanProgressPairs + B * (size_t)kLeanMaxPoints;
            if ((rc = solver_->dev_lean_progress.reserve(words * sizeof(unsigned))) != ROCCO_HIP_OK) return rc;
            A.reset = LeanRoundReset{(unsigned *)solver_->dev_lean_progress.ptr, half_fused ? kLeanProgressPairs : (int)words,
                                     half_fused ? nullptr : L.ticket, L.error};
        }
        for (int r = 0; r < R; ++r) {
            if ((rc = launch_chain_director(A, r, 0, stream_)) != ROCCO_HIP_OK) return rc;
            if (fused) {
                if ((rc = launch_lean_round_chain(L, A.pre, A.reset.progress, eval_grid, 0, stream_)) != ROCCO_HIP_OK) return rc;
                continue;
            }
            if (half_fused) {
                if ((rc = launch_lean_round_chain(L, A.pre, A.reset.progress, eval_grid, 0, stream_, 0)) != ROCCO_HIP_OK) return rc;
                if ((rc = launch_lean_finish_chain(L, finish_grid, stream_)) != ROCCO_HIP_OK) return rc;
                continue;
            }
            if ((rc = launch_lean_compact_chain(A.pre, A.ctl, compact_grid, stream_)) != ROCCO_HIP_OK) return rc;
            if ((rc = launch_lean_eval_chain(L, eval_grid, stream_)) != ROCCO_HIP_OK) return rc;
            if ((rc = launch_lean_finish_chain(L, finish_grid, stream_)) != ROCCO_HIP_OK) return rc;
        }
        if ((rc = launch_chain_director(A, R, 1, stream_)) != ROCCO_HIP_OK) return rc;
        ROCCO_HIP_TRY(hipGetLastError());
        const double ts0 = now_us();
        if (follow) {
            const unsigned long long *flag = A.follow;
            for (long long spins = 1;; ++spins) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != 0ull) {
                    break;
                }
                if (spins > 200000) {
                    std::this_thread::yield();
                }
                if ((spins & 1023) == 0) {
                    const hipError_t q = hipStreamQuery(stream_);
                    if (q != hipErrorNotReady) {
                        if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != 0ull) {
                            break;
                        }
                        set_last_error(q != hipSuccess ? "chained search: the stream reports an error" : "chained search: the stream ended without the chain's report");
                        return ROCCO_HIP_EHIP;
                    }
                }
            }
            h = (char *)solver_->host_follow.ptr + 256;
        } else {
            ROCCO_HIP_TRY(hipMemcpyAsync(dv + o_ctl_back, dv + o_ctl, sizeof(LeanRoundCtl), hipMemcpyDeviceToDevice, stream_));
            ROCCO_HIP_TRY(hipMemcpyAsync(h, dv + o_probs, down_bytes, hipMemcpyDeviceToHost, stream_));
            ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
        }
        const double ts1 = now_us();
        mark("threshold chain waited for");
        t_wait_ += ts1 - ts0;
        t_chain_submit_ = ts0 - tc0;
        t_chain_wait_ = ts1 - ts0;
        struct Parse {
            double &acc;
            double t0;
            ~Parse() { acc = now_us() - t0; }
        } parse{t_chain_parse_, ts1};
        solver_->lean_look_dirty = 0;
        const ChainProb *rep = (const ChainProb *)h;
        const double *hstats = (const double *)(h + (o_stats - o_probs));
        const LeanRoundCtl *ctl = (const LeanRoundCtl *)(h + (o_ctl_back - o_probs));
        if (ctl->error != 0u) {
            solver_->lean_look_dirty = 1;
            set_last_error((ctl->error & 1u) ? "chained search: a tile waited for its predecessor beyond the spin limit"
                                             : "chained search: a compaction overflowed its level");
            return ROCCO_HIP_EHIP;
        }
        ran = true;
        chain_rounds_run = ctl->round;
        if (want_trace) {
            std::vector<long long> tr((size_t)(R + 1) * 8);
            ROCCO_HIP_TRY(hipMemcpy(tr.data(), dv + o_trace, tr.size() * sizeof(long long), hipMemcpyDeviceToHost));
            for (int r = 0; r <= R; ++r) {
                const long long *t = tr.data() + 8 * r;
                std::fprintf(stderr, "[chain] director %d: state in %.2f us, results + compaction %.2f, groups %.2f, penalties %.2f, offsets %.2f, descriptors + state out %.2f; since the chain began %.2f\n",
                             r, 0.01 * (double)(t[1] - t[0]), 0.01 * (double)(t[2] - t[1]), 0.0, 0.01 * (double)(t[3] - t[2]),
                             0.01 * (double)(t[4] - t[3]), 0.01 * (double)(t[5] - t[4]), 0.01 * (double)(t[0] - tr[0]));
            }
        }
        stats_out.assign(hstats, hstats + 5 * B);
        pre.assign(B, Presearch());
        const bool debug = env("ROCCO_HIP_DEBUG") != nullptr;
        const bool chain_debug = std::getenv("ROCCO_HIP_CHAIN_DEBUG") != nullptr;
        bool grid_ok = true;
        for (size_t b = 0; b < B; ++b) {
            const ChainProb &r = rep[b];
            if (r.searching == 0) {
                continue;
            }
            // the grid the device evaluated on must be the one the rest of the solve will use
            const double cmax = (probs[b].costs != nullptr && probs[b].n > 1) ? hstats[5 * b + 3] : probs[b].gamma;
            if (r.qexp != grid_exponent(std::max(cmax, 0.0), hstats[5 * b + 0], hstats[5 * b + 1])) {
                grid_ok = false;
            }
        }
        if (!grid_ok) {
            // (a score range within rounding of a power of two: the host's log2 and the device's exact logarithm disagree;
            // nothing of the chain is used)
            if (debug) {
                std::fprintf(stderr, "[chain] grid exponent differs from the host's: the chain's work is discarded\n");
            }
            pre.assign(B, Presearch());
            return ROCCO_HIP_OK;
        }
        for (size_t b = 0; b < B; ++b) {
            const ChainProb &r = rep[b];
            if (r.searching == 0 || r.n_evals == 0) {
                continue;
            }
            Presearch &ps = pre[b];
            for (int i = 0; i < r.n_above; ++i) {
                ps.evals.emplace_back(r.above_x[i], r.above_c[i]);
            }
            for (int i = 0; i < r.n_below; ++i) {
                ps.evals.emplace_back(r.below_x[i], r.below_c[i]);
            }
            ps.rounds = r.rounds + r.pilots;
            ps.done = (r.done == 1);
            LeanState &ls = lean_[b];
            ls.levels.clear();
            for (int k = 0; k < r.n_levels; ++k) {
                LeanLevel lv;
                lv.s = r.levels[k].s;
                lv.orig = r.levels[k].orig;
                lv.m = r.levels[k].m;
                lv.base = r.levels[k].base;
                lv.sep = r.levels[k].sep;
                lv.has_eval = false;
                lv.bits = r.levels[k].bits;
                lv.tile_off = r.levels[k].tile_off;
                lv.cap_points = r.levels[k].cap_points;
                lv.pool_mark = (size_t)r.levels[k].pool_mark;
                ls.levels.push_back(lv);
            }
            ls.pool_at = (size_t)r.pool_at;
            if (chain_debug) {
                std::vector<ChainEvals> all(1);
                ROCCO_HIP_TRY(hipMemcpy(all.data(), dv + o_evals + b * sizeof(ChainEvals), sizeof(ChainEvals), hipMemcpyDeviceToHost));
                for (int i = 0; i < r.n_evals; ++i) {
                    std::fprintf(stderr, "[chain] problem %zu count(%.17g) = %lld%s\n", b, all[0].x[i], all[0].c[i],
                                 all[0].c[i] > std::max(0LL, std::min(problems[b].target_count, (long long)probs[b].n)) ? "  >" : "");
                }
            }
            if (debug) {
                std::fprintf(stderr, "[chain] problem %zu (n=%zu): %d pilot + %d certified rounds, %d counts, %d levels (deepest m=%lld, base %.17g)%s\n",
                             b, probs[b].n, r.pilots, r.rounds, r.n_evals, r.n_levels, r.levels[r.n_levels - 1].m,
                             r.levels[r.n_levels - 1].base, r.done == 1 ? ", ended" : (r.done == 2 ? ", given up" : ", out of rounds"));
            }
        }
        return ROCCO_HIP_OK;
    }

    int rounds = 0;
    long long blocks_launched = 0;

private:

public:
    double t_prep_ = 0.0, t_launch_ = 0.0, t_wait_ = 0.0, t_consume_ = 0.0, t_round_ = 0.0;
    double t_lean_cpu_ = 0.0, t_lean_h2d_ = 0.0, t_lean_launch_ = 0.0;
    double t_chain_submit_ = 0.0, t_chain_wait_ = 0.0, t_chain_parse_ = 0.0;
    int rounds_all = 0;
    static double now_us()
    {
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

private:
    // `may_defer`: the caller reads nothing behind this round that needs the stream drained (see stage_mark);
    // `prefetch`: problems whose objective sums are queued behind the round's kernels, before the wait (prefetch_objectives)
    int run_round(std::vector<RoundTask> &rt, bool may_defer = false, const std::vector<size_t> *prefetch = nullptr)
    {
        const size_t T = rt.size();
        if (T == 0) {
            return ROCCO_HIP_OK;
        }
        ++rounds;
        for (size_t t = 0; t < T; ++t) {
            objective_invalidate(rt[t].problem);  // (the round may rewrite the problem's solution)
        }
        {
            const int rcw = stage_wait();
            if (rcw != ROCCO_HIP_OK) return rcw;
        }
        const double tt0 = now_us();
        std::vector<FastTask> tasks(T);
        std::vector<FastChain> chains;
        std::vector<FastSlot> slots;
        std::vector<int2> blockmap, blockmap_all;
        std::vector<std::vector<int>> launched(T);
        std::vector<char> launched_all(T, 0);
        std::vector<long long> skip_off(T, -1);
        std::vector<uint8_t> skip_bytes;
        long long chain_chunks = 0, chain_blocks = 0, slot_chunks = 0, slot_blocks = 0, rec_entries = 0;
        long long rec_groups = 0;
        bool any_record = false, any_select = false;
        std::vector<int> solution_slot(T, -1);
        bool any_costs = false, any_plain = false, any_window = false, any_map = false;
        for (size_t t = 0; t < T; ++t) {
            const DevProblem &p = probs[rt[t].problem];
            FastTask &ft = tasks[t];
            const long long nchunks = (long long)((p.n + kChunk - 1) / kChunk);
            const int nblocks = (int)((p.n + kFastBlockLoci - 1) / kFastBlockLoci);
            ft.scores = p.scores;
            ft.switch_costs = p.costs;
            ft.gamma = p.gamma;
            ft.n = (long long)p.n;
            ft.qexp = p.qexp;
            ft.emap = p.emap;
            ft.emap_out = rt[t].map ? map_ptrs_[rt[t].problem] : nullptr;
            ft.map_margin = rt[t].margin;
            ft.magic = std::ldexp(1.5, 52 + p.qexp);
            ft.big = std::ldexp(1.0, 50 + p.qexp);
            ft.qstep = std::ldexp(1.0, p.qexp);
            ft.cmax = p.cmax;
            ft.sabs = p.sabs;
            ft.n_blocks = nblocks;
            ft.solution = p.solution;
            ft.frz = FrozenArrays{};
            ft.frz_out = FrozenArrays{};
            // rounds inside the surveyed bracket skip the frozen blocks (the final zone windows and
            // the spine always run over everything: they materialise the solution)
            bool use_frozen = p.frz_valid && !no_frozen_ && solver_->active_set != 0 && !rt[t].bound;
            if (use_frozen) {
                for (double lam : rt[t].lambdas) {
                    if (!(lam >= p.frz_lo && lam <= p.frz_hi) || lambda_ties_some_grid(lam, p.qexp)) {
                        use_frozen = false;
                    }
                }
            }
            rt[t].use_frozen = use_frozen;
            launched[t].clear();
            if (use_frozen) {
                ft.frz = p.frz;
                // the flags are read from a copy uploaded with the descriptors: a survey of the same
                // problem may be rewriting the live ones in this very round
                skip_off[t] = (long long)skip_bytes.size();
                std::vector<uint8_t> skip(p.frz_flags);
                if (rt[t].record) {
                    // the spine may still be stepping when it leaves an active block: also evaluate
                    // the block after each active one (it resynchronises there)
                    for (int k : p.active_blocks) {
                        if (k + 1 < nblocks) {
                            skip[k + 1] = 0;
                        }
                    }
                }
                for (int k = 0; k < nblocks; ++k) {
                    if (!skip[k]) {
                        launched[t].push_back(k);
                    }
                }
                skip.resize(align_up((size_t)nblocks, 64), 0);
                skip_bytes.insert(skip_bytes.end(), skip.begin(), skip.end());
            } else {
                launched_all[t] = 1;  // every block: the list stays empty
            }
            if (rt[t].survey) {
                ft.frz_out = p.frz;
            }
            if (env("ROCCO_HIP_DEBUG") != nullptr) {
                const char *kind = rt[t].record ? "spine" : (rt[t].survey ? "survey" : (rt[t].window ? "window" : (rt[t].map ? "map" : "probe")));
                std::fprintf(stderr, "[round %d] %s problem %zu: %zu lambdas (first %.17g, margin %.3g), %zu / %d blocks%s\n",
                             rounds, kind, rt[t].problem, rt[t].lambdas.size(), rt[t].lambdas[0], rt[t].margin,
                             launched_all[t] ? (size_t)nblocks : launched[t].size(), nblocks, p.emap ? "" : " [no map]");
            }
            ft.slot_begin = (int)slots.size();
            any_costs = any_costs || (p.costs != nullptr);
            any_plain = any_plain || (p.costs == nullptr);
            any_map = any_map || rt[t].map;
            if (rt[t].window) {
                any_window = true;
                FastSlot s;
                s.task = (int)t;
                s.mode = kModeWindow;
                s.chain_a = (int)chains.size();
                s.chain_b = s.chain_a + 1;
                s.chunk_off = slot_chunks;
                s.block_off = slot_blocks;
                slot_chunks += nchunks;
                slot_blocks += nblocks;
                for (int k = 0; k < 2; ++k) {
                    FastChain c;
                    c.task = (int)t;
                    c.lambda = rt[t].lambdas[k];
                    c.chunk_off = chain_chunks;
                    c.block_off = chain_blocks;
                    chain_chunks += nchunks;
                    chain_blocks += nblocks;
                    chains.push_back(c);
                }
                slots.push_back(s);
            } else {
                for (double lam : rt[t].lambdas) {
                    FastSlot s;
                    s.task = (int)t;
                    s.mode = rt[t].map ? kModeMap
                                       : (rt[t].record ? kModeRecord : (rt[t].bound ? kModeBound : kModeProbe));
                    s.chain_a = s.chain_b = (int)chains.size();
                    s.chunk_off = slot_chunks;
                    s.block_off = slot_blocks;
                    slot_chunks += nchunks;
                    slot_blocks += nblocks;
                    FastChain c;
                    c.task = (int)t;
                    c.lambda = lam;
                    c.chunk_off = chain_chunks;
                    c.block_off = chain_blocks;
                    chain_chunks += nchunks;
                    chain_blocks += nblocks;
                    chains.push_back(c);
                    slots.push_back(s);
                }
            }
            ft.slot_count = (int)slots.size() - ft.slot_begin;
            ft.rec_off = 0;
            ft.rec_goff = 0;
            ft.pre_round = rt[t].bound ? 1 : 0;
            ft.sel_depth = rt[t].record ? rt[t].select_depth : 0;
            ft.sel_has_upper = rt[t].select_has_upper ? 1 : 0;
            ft.sel_target = rt[t].select_target;
            any_select = any_select || (ft.sel_depth > 0);
            if (rt[t].record) {
                any_record = true;
                ft.rec_off = rec_entries;
                rec_entries += nchunks * ft.slot_count;
                ft.rec_goff = rec_groups;
                rec_groups += ((nchunks + 31) / 32) * ft.slot_count;
                if (rt[t].solution_index >= 0) {
                    solution_slot[t] = ft.slot_begin + rt[t].solution_index;
                }
            }
            const size_t at_all = blockmap_all.size();
            blockmap_all.resize(at_all + (size_t)nblocks);
            for (int k = 0; k < nblocks; ++k) {
                blockmap_all[at_all + (size_t)k] = make_int2((int)t, k);
            }
            if (launched_all[t]) {
                blockmap.insert(blockmap.end(), blockmap_all.begin() + (long)at_all, blockmap_all.end());
            } else {
                for (int k : launched[t]) {
                    blockmap.push_back(make_int2((int)t, k));
                }
            }
        }
        const size_t C = chains.size(), S = slots.size(), NB = blockmap.size(), NBA = blockmap_all.size();
        blocks_launched += (long long)NB;

        // ---- descriptor upload ----
        const size_t b_tasks = align_up(T * sizeof(FastTask), 256);
        const size_t b_chains = align_up(C * sizeof(FastChain), 256);
        const size_t b_slots = align_up(S * sizeof(FastSlot), 256);
        const size_t b_map = align_up(NB * sizeof(int2), 256);
        const size_t b_mapall = align_up(NBA * sizeof(int2), 256);
        const size_t b_skip = align_up(skip_bytes.size() + 1, 256);
        const size_t desc_bytes = b_tasks + b_chains + b_slots + b_map + b_mapall + b_skip;
        int rc;
        if ((rc = solver_->dev_tasks.reserve(desc_bytes)) != ROCCO_HIP_OK) return rc;
        if ((rc = solver_->host_stage.reserve(desc_bytes)) != ROCCO_HIP_OK) return rc;
        char *h = (char *)solver_->host_stage.ptr;
        char *dd = (char *)solver_->dev_tasks.ptr;
        const double tt1 = now_us();
        t_prep_ += tt1 - tt0;
        for (size_t t = 0; t < T; ++t) {
            if (skip_off[t] >= 0) {  // record rounds: "not evaluated this round" instead of "frozen"
                tasks[t].frz.flag = (uint8_t *)(dd + b_tasks + b_chains + b_slots + b_map + b_mapall + skip_off[t]);
            }
        }
        std::memcpy(h, tasks.data(), T * sizeof(FastTask));
        std::memcpy(h + b_tasks, chains.data(), C * sizeof(FastChain));
        std::memcpy(h + b_tasks + b_chains, slots.data(), S * sizeof(FastSlot));
        std::memcpy(h + b_tasks + b_chains + b_slots, blockmap.data(), NB * sizeof(int2));
        std::memcpy(h + b_tasks + b_chains + b_slots + b_map, blockmap_all.data(), NBA * sizeof(int2));
        if (!skip_bytes.empty()) {
            std::memcpy(h + b_tasks + b_chains + b_slots + b_map + b_mapall, skip_bytes.data(), skip_bytes.size());
        }
        ROCCO_HIP_TRY(hipMemcpyAsync(dd, h, desc_bytes, hipMemcpyHostToDevice, stream_));

        // ---- scratch carve ----
        size_t off = 0;
        auto carve = [&off](size_t bytes) {
            const size_t at = off;
            off += align_up(bytes, 256);
            return at;
        };
        const size_t o_agg_a = carve((size_t)chain_chunks * 8), o_agg_lo = carve((size_t)chain_chunks * 8),
                     o_agg_hi = carve((size_t)chain_chunks * 8), o_pstar = carve((size_t)chain_chunks);
        const size_t o_blk_a = carve((size_t)chain_blocks * 8), o_blk_lo = carve((size_t)chain_blocks * 8),
                     o_blk_hi = carve((size_t)chain_blocks * 8), o_din = carve((size_t)chain_blocks * 8);
        const size_t o_lcc = carve((size_t)slot_chunks);
        const size_t o_wc = carve((size_t)slot_chunks * 8), o_gc = carve(any_map ? (size_t)slot_chunks * 8 : 8);
        const size_t o_lcb = carve((size_t)slot_blocks * 4), o_lcin = carve((size_t)slot_blocks * 4);
        const size_t o_wb = carve((size_t)slot_blocks * 8), o_winb = carve((size_t)slot_blocks * 8);
        const size_t o_gb = carve((size_t)slot_blocks * 8), o_ginb = carve((size_t)slot_blocks * 8);
        const size_t o_fvlo = carve((size_t)slot_blocks), o_fvhi = carve((size_t)slot_blocks);
        const size_t o_plo = carve((size_t)slot_blocks * 4), o_phi = carve((size_t)slot_blocks * 4);
        const size_t o_blo = carve((size_t)slot_blocks * 4), o_bhi = carve((size_t)slot_blocks * 4);
        const size_t o_rin = carve((size_t)slot_blocks);
        const size_t o_rdin = carve((size_t)rec_entries * 8 + 8), o_rgain = carve((size_t)rec_entries * 8 + 8);
        const size_t o_rd = carve((size_t)rec_entries * 4 + 8), o_rv = carve((size_t)rec_entries * 4 + 8);
        const size_t o_rf = carve((size_t)rec_entries + 8);
        const size_t o_rgs = carve((size_t)rec_groups * 8 + 8), o_rgo = carve((size_t)rec_groups + 8);
        const size_t o_ss = carve(T * sizeof(int));
        const size_t o_res = carve(S * sizeof(FastSlotResult));
        if ((rc = solver_->dev_params.reserve(off)) != ROCCO_HIP_OK) return rc;
        char *sc = (char *)solver_->dev_params.ptr;

        FastLaunch L;
        L.tasks = (const FastTask *)dd;
        L.chains = (const FastChain *)(dd + b_tasks);
        L.slots = (const FastSlot *)(dd + b_tasks + b_chains);
        L.blockmap = (const int2 *)(dd + b_tasks + b_chains + b_slots);
        L.n_tasks = (int)T;
        L.n_chains = (int)C;
        L.n_slots = (int)S;
        L.blockmap_all = (const int2 *)(dd + b_tasks + b_chains + b_slots + b_map);
        L.n_blocks_total = (int)NB;
        L.n_blocks_all = (int)NBA;
        {
            // few blocks with many penalties each (spine rounds): spread the slots over workgroups
            int max_slots = 1;
            for (size_t t = 0; t < T; ++t) {
                max_slots = std::max(max_slots, tasks[t].slot_count);
            }
            int groups = 1;
            while (groups < max_slots && (size_t)(2 * groups) * NB <= 2048) {
                groups *= 2;
            }
            L.slot_groups = std::min(groups, max_slots);
        }
        L.any_costs = any_costs;
        L.any_plain = any_plain;
        L.any_window = any_window;
        L.any_map = any_map;
        L.buf.agg_a = (double *)(sc + o_agg_a);
        L.buf.agg_lo = (double *)(sc + o_agg_lo);
        L.buf.agg_hi = (double *)(sc + o_agg_hi);
        L.buf.pstar = (uint8_t *)(sc + o_pstar);
        L.buf.blk_a = (double *)(sc + o_blk_a);
        L.buf.blk_lo = (double *)(sc + o_blk_lo);
        L.buf.blk_hi = (double *)(sc + o_blk_hi);
        L.buf.din = (double *)(sc + o_din);
        L.buf.lc_chunk = (int8_t *)(sc + o_lcc);
        L.buf.w_chunk = (double *)(sc + o_wc);
        L.buf.gain_chunk = (double *)(sc + o_gc);
        L.buf.w_block = (double *)(sc + o_wb);
        L.buf.win_block = (double *)(sc + o_winb);
        L.buf.gain_block = (double *)(sc + o_gb);
        L.buf.gainin_block = (double *)(sc + o_ginb);
        L.buf.lc_block = (int *)(sc + o_lcb);
        L.buf.lcin_block = (int *)(sc + o_lcin);
        L.buf.bfv_lo = (uint8_t *)(sc + o_fvlo);
        L.buf.bfv_hi = (uint8_t *)(sc + o_fvhi);
        L.buf.bpend_lo = (unsigned *)(sc + o_plo);
        L.buf.bpend_hi = (unsigned *)(sc + o_phi);
        L.buf.bbase_lo = (unsigned *)(sc + o_blo);
        L.buf.bbase_hi = (unsigned *)(sc + o_bhi);
        L.buf.rin_lo = (uint8_t *)(sc + o_rin);
        L.buf.rec_din = (double *)(sc + o_rdin);
        L.buf.rec_gain = (double *)(sc + o_rgain);
        L.buf.rec_d = (unsigned *)(sc + o_rd);
        L.buf.rec_v = (unsigned *)(sc + o_rv);
        L.buf.rec_flags = (uint8_t *)(sc + o_rf);
        L.buf.rec_gsum = (double *)(sc + o_rgs);
        L.buf.rec_gok = (uint8_t *)(sc + o_rgo);
        L.buf.results = (FastSlotResult *)(sc + o_res);
        ROCCO_HIP_TRY(hipMemsetAsync(L.buf.results, 0, S * sizeof(FastSlotResult), stream_));
        if ((rc = launch_fast_round(L, stream_)) != ROCCO_HIP_OK) {
            return rc;
        }
        {
            bool only_maps = true;
            for (size_t t = 0; t < T; ++t) {
                only_maps = only_maps && rt[t].map;
            }
            const char *defer = env("ROCCO_HIP_DEFER_MAPS");
            if (may_defer && only_maps && (defer == nullptr || std::atoi(defer) != 0)) {
                // nothing of a map round is read on the host (adopt_maps only takes the pointers over)
                t_launch_ += now_us() - tt1;
                return stage_mark();
            }
        }
        if (any_record) {
            if ((rc = solver_->host_back.reserve(T * sizeof(int) + S * sizeof(FastSlotResult))) != ROCCO_HIP_OK) return rc;
            int *h_ss = (int *)((char *)solver_->host_back.ptr + S * sizeof(FastSlotResult));
            std::memcpy(h_ss, solution_slot.data(), T * sizeof(int));
            ROCCO_HIP_TRY(hipMemcpyAsync(sc + o_ss, h_ss, T * sizeof(int), hipMemcpyHostToDevice, stream_));
            if ((rc = launch_spine(L, (int *)(sc + o_ss), any_select, stream_)) != ROCCO_HIP_OK) {
                return rc;
            }
        }
        size_t flag_bytes = 0;
        for (size_t t = 0; t < T; ++t) {
            if (rt[t].survey) {
                flag_bytes += align_up((size_t)tasks[t].n_blocks, 64);
            }
        }
        if ((rc = solver_->host_back.reserve(S * sizeof(FastSlotResult) + flag_bytes + 64)) != ROCCO_HIP_OK) return rc;
        ROCCO_HIP_TRY(hipMemcpyAsync(solver_->host_back.ptr, L.buf.results, S * sizeof(FastSlotResult),
                                     hipMemcpyDeviceToHost, stream_));
        {
            char *hf = (char *)solver_->host_back.ptr + S * sizeof(FastSlotResult);
            for (size_t t = 0; t < T; ++t) {
                if (rt[t].survey) {
                    ROCCO_HIP_TRY(hipMemcpyAsync(hf, probs[rt[t].problem].frz.flag, (size_t)tasks[t].n_blocks,
                                                 hipMemcpyDeviceToHost, stream_));
                    hf += align_up((size_t)tasks[t].n_blocks, 64);
                }
            }
        }
        if (prefetch != nullptr && (rc = prefetch_objectives(*prefetch)) != ROCCO_HIP_OK) {
            return rc;
        }
        const double tt2 = now_us();
        t_launch_ += tt2 - tt1;
        ROCCO_HIP_TRY(hipStreamSynchronize(stream_));
        const double tt3 = now_us();
        t_wait_ += tt3 - tt2;
        const FastSlotResult *hr = (const FastSlotResult *)solver_->host_back.ptr;
        for (size_t t = 0; t < T; ++t) {
            if (!(rt[t].record && rt[t].use_frozen)) {
                continue;
            }
            for (int k = 0; k < tasks[t].slot_count; ++k) {
                if (hr[tasks[t].slot_begin + k].overflow) {
                    // a lane was still stepping when it reached a block that was not evaluated
                    if (env("ROCCO_HIP_DEBUG") != nullptr) {
                        std::fprintf(stderr, "[spine] problem %zu: repeated in full\n", rt[t].problem);
                    }
                    no_frozen_ = true;
                    const int rc2 = run_round(rt);
                    no_frozen_ = false;
                    return rc2;
                }
            }
        }
        {
            const char *hf = (const char *)solver_->host_back.ptr + S * sizeof(FastSlotResult);
            for (size_t t = 0; t < T; ++t) {
                if (!rt[t].survey) {
                    continue;
                }
                DevProblem &p = probs[rt[t].problem];
                const int nb = tasks[t].n_blocks;
                if (!rt[t].use_frozen) {
                    p.frz_flags.assign(nb, 0);
                }
                // a block surveyed this round carries its new verdict; skipped (frozen) blocks stay frozen
                std::vector<int> active;
                const std::vector<int> *surveyed = (rt[t].use_frozen && !launched_all[t]) ? &launched[t] : nullptr;
                if (surveyed == nullptr) {
                    for (int k = 0; k < nb; ++k) {
                        p.frz_flags[k] = (uint8_t)hf[k];
                    }
                } else {
                    for (int k : *surveyed) {
                        p.frz_flags[k] = (uint8_t)hf[k];
                    }
                }
                for (int k = 0; k < nb; ++k) {
                    if (!p.frz_flags[k]) {
                        active.push_back(k);
                    }
                }
                p.active_blocks.swap(active);
                if (env("ROCCO_HIP_DEBUG") != nullptr) {
                    std::fprintf(stderr, "[survey] problem %zu: bracket [%.17g, %.17g] width %.3g active %zu / %d blocks\n",
                                 rt[t].problem, rt[t].lambdas[0], rt[t].lambdas[1],
                                 rt[t].lambdas[1] - rt[t].lambdas[0], p.active_blocks.size(), nb);
                }
                p.frz_valid = true;
                p.frz_lo = rt[t].lambdas[0];
                p.frz_hi = rt[t].lambdas[1];
                hf += align_up((size_t)nb, 64);
            }
        }

        for (size_t t = 0; t < T; ++t) {
            const FastTask &ft = tasks[t];
            const DevProblem &p = probs[rt[t].problem];
            if (rt[t].map) {
                continue;
            }
            if (rt[t].record) {
                rt[t].spine->counts.resize(rt[t].lambdas.size());
                rt[t].spine->stepped.resize(rt[t].lambdas.size());
                for (int k = 0; k < ft.slot_count; ++k) {
                    rt[t].spine->counts[k] = hr[ft.slot_begin + k].count_lo;
                    rt[t].spine->stepped[k] = hr[ft.slot_begin + k].uncertain;
                }
                rt[t].spine->selected = (ft.sel_depth > 0) ? hr[ft.slot_begin].e_global : -1;
                if (env("ROCCO_HIP_DEBUG") != nullptr) {
                    std::fprintf(stderr, "[spine] problem %zu: lane 0 stepped %lld chunks of %lld\n", rt[t].problem,
                                 (long long)hr[ft.slot_begin].uncertain, (long long)((p.n + kChunk - 1) / kChunk));
                    std::fprintf(stderr, "[spine] prof (100 MHz ticks): fetch %lld loop %lld (steps %lld) slow groups %lld\n",
                                 hr[ft.slot_begin].p16, hr[ft.slot_begin].npos, hr[ft.slot_begin].max_run,
                                 hr[ft.slot_begin].n_diff);
                }
                continue;
            }
            if (rt[t].window) {
                const FastSlotResult &r = hr[ft.slot_begin];
                WindowResult &w = rt[t].win->result;
                w.count_lo = r.count_lo;
                w.count_hi = r.count_hi;
                w.n_diff = r.n_diff;
                w.diff_adjacent = (r.nonadjacent == 0);
                w.overflow = (r.overflow != 0);
                w.max_run = r.max_run;
                w.diffs.clear();
                const long long listed = std::min<long long>(r.n_diff, kMaxDiffs);
                for (long long k = 0; k < listed; ++k) {
                    WindowDiff d;
                    d.locus = r.diffs[k].locus;
                    d.margin_lo = r.diffs[k].margin_lo;
                    d.margin_hi = r.diffs[k].margin_hi;
                    d.run = r.diffs[k].run;
                    d.cls_lo = r.diffs[k].cls_lo;
                    d.cls_hi = r.diffs[k].cls_hi;
                    w.diffs.push_back(d);
                }
                std::sort(w.diffs.begin(), w.diffs.end(),
                          [](const WindowDiff &a, const WindowDiff &b) { return a.locus < b.locus; });
            } else {
                for (int k = 0; k < ft.slot_count; ++k) {
                    const FastSlotResult &r = hr[ft.slot_begin + k];
                    ProbeResult &o = rt[t].probe->results[k];
                    o.count = r.count_lo;
                    o.uncertain = rt[t].bound ? 0 : r.uncertain;
                    o.effect = rt[t].bound ? 0 : (r.overflow ? (long long)p.n + 1 : r.effect);
                    o.max_run = r.max_run;
                }
            }
        }
        return ROCCO_HIP_OK;
    }

    rocco_hip_solver *solver_;
    hipStream_t stream_;
    bool maps_allocated_ = false;
    bool frozen_allocated_ = false;
    bool no_frozen_ = false;
    std::vector<uint8_t *> map_ptrs_;
};

// Fill ChainProblem / DevProblem statistics from one stats pass.
int prepare(HipEvaluator &ev, std::vector<ChainProblem> &problems, const std::vector<double> *fixed_lambdas,
            const double *score_stats_host = nullptr, const std::vector<double> *stats5 = nullptr)
{
    std::vector<double> stats;
    if (stats5 != nullptr && stats5->size() == 5 * problems.size()) {
        stats = *stats5;  // (the chained search ran the statistics pass)
    }
    bool given = (score_stats_host != nullptr) && stats.empty();
    for (size_t b = 0; given && b < problems.size(); ++b) {
        given = (ev.probs[b].costs == nullptr || ev.probs[b].n <= 1);  // cost vectors need their own extremes
    }
    if (given) {
        stats.assign(5 * problems.size(), 0.0);
        for (size_t b = 0; b < problems.size(); ++b) {
            stats[5 * b + 0] = score_stats_host[3 * b + 0];
            stats[5 * b + 1] = score_stats_host[3 * b + 1];
            stats[5 * b + 4] = score_stats_host[3 * b + 2];
        }
    } else if (stats.empty()) {
        const int rc = ev.compute_stats(stats);
        if (rc != ROCCO_HIP_OK) {
            return rc;
        }
    }
    for (size_t b = 0; b < problems.size(); ++b) {
        ChainProblem &p = problems[b];
        DevProblem &d = ev.probs[b];
        p.score_min = stats[5 * b + 0];
        p.score_max = stats[5 * b + 1];
        p.score_abs_sum = stats[5 * b + 4];
        if (d.costs != nullptr && d.n > 1) {
            p.cost_min = stats[5 * b + 2];
            p.cost_max = stats[5 * b + 3];
            p.has_cost_vector = true;
        } else {
            p.cost_min = p.cost_max = d.gamma;
            p.has_cost_vector = false;
        }
        double lo = p.score_min, hi = p.score_max;
        if (fixed_lambdas != nullptr) {
            lo = std::min(lo, (*fixed_lambdas)[b]);
            hi = std::max(hi, (*fixed_lambdas)[b]);
        }
        d.cmax = p.cost_max;
        d.smin = p.score_min;
        d.smax = p.score_max;
        d.sabs = std::max(std::fabs(p.score_min), std::fabs(p.score_max));
        d.qexp = (std::isfinite(lo) && std::isfinite(hi) && std::isfinite(p.cost_max))
                     ? grid_exponent(std::max(p.cost_max, 0.0), lo, hi)
                     : 0;
    }
    return ROCCO_HIP_OK;
}

}  // namespace

void model_chain_counters(long long out[4])
{
    for (int k = 0; k < 4; ++k) {
        out[k] = g_model_chain_counters[k].load(std::memory_order_relaxed);
    }
}

void model_chain_written_counters(long long out[2])
{
    out[0] = g_model_chain_counters[4].load(std::memory_order_relaxed);
    out[1] = g_model_chain_counters[5].load(std::memory_order_relaxed);
}

int solve_fixed_penalty(rocco_hip_solver *solver, const double *scores_dev,
                        const double *switch_costs_dev, double gamma, size_t n, double lambda,
                        uint8_t *solution_dev, double *value_out, long long *count_out, int *path_out,
                        hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = (n > 1) ? switch_costs_dev : nullptr;
    d.gamma = gamma;
    d.n = n;
    int rc;
    if (solution_dev == nullptr) {  // the fast path always materialises; give it scratch
        if ((rc = solver->dev_solution.reserve(n)) != ROCCO_HIP_OK) return rc;
        d.solution = (uint8_t *)solver->dev_solution.ptr;
    } else {
        d.solution = solution_dev;
    }
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    std::vector<double> lambdas = {lambda};
    if ((rc = prepare(ev, problems, &lambdas)) != ROCCO_HIP_OK) return rc;
    SearchOptions opt;
    opt.force_exact = solver->force_exact != 0;
    std::vector<CalibrationResult> res;
    if ((rc = solve_fixed_batch(ev, problems, lambdas, opt, res)) != ROCCO_HIP_OK) return rc;
    if (value_out) *value_out = res[0].penalized_value;
    if (count_out) *count_out = res[0].selected_count;
    if (path_out) *path_out = res[0].path;
    return ROCCO_HIP_OK;
}

int delta_build_map(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                    double gamma, size_t n, double lambda_ref, double margin, uint8_t *emap_dev,
                    hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = (n > 1) ? switch_costs_dev : nullptr;
    d.gamma = gamma;
    d.n = n;
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    int rc;
    if ((rc = prepare(ev, problems, nullptr)) != ROCCO_HIP_OK) return rc;
    std::vector<MapRequest> reqs(1);
    reqs[0].problem = 0;
    reqs[0].lambda_ref = lambda_ref;
    reqs[0].margin = margin;
    if ((rc = ev.build_map(reqs)) != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipMemcpyAsync(emap_dev, ev.probs[0].emap, (n + kChunk - 1) / kChunk, hipMemcpyDeviceToDevice, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    return ROCCO_HIP_OK;
}

int delta_build_map_lean(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n, double lambda_ref,
                         double margin, uint8_t *emap_dev, hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = nullptr;
    d.gamma = gamma;
    d.n = n;
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    int rc;
    if ((rc = prepare(ev, problems, nullptr)) != ROCCO_HIP_OK) return rc;
    ev.force_lean_map_ = true;
    if (!ev.lean_map_takes(0)) {
        set_last_error("lean map: not available for this array (lean evaluation off, or fewer than two loci)");
        return ROCCO_HIP_EINVAL;
    }
    std::vector<MapRequest> reqs(1);
    reqs[0].problem = 0;
    reqs[0].lambda_ref = lambda_ref;
    reqs[0].margin = margin;
    if ((rc = ev.build_map(reqs)) != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipMemcpyAsync(emap_dev, ev.probs[0].emap, (n + kChunk - 1) / kChunk, hipMemcpyDeviceToDevice, stream));
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    return ev.lean_map_check();
}

int delta_spine(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                double gamma, size_t n, const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                int solution_index, uint8_t *solution_dev, long long *counts_out, hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = (n > 1) ? switch_costs_dev : nullptr;
    d.gamma = gamma;
    d.n = n;
    d.solution = solution_dev;
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    int rc;
    if ((rc = prepare(ev, problems, nullptr)) != ROCCO_HIP_OK) return rc;
    ev.probs[0].emap = const_cast<uint8_t *>(emap_dev);
    std::vector<SpineRequest> reqs(1);
    reqs[0].problem = 0;
    reqs[0].lambdas.assign(lambdas, lambdas + n_lambdas);
    reqs[0].solution_index = solution_index;
    if ((rc = ev.spine(reqs)) != ROCCO_HIP_OK) return rc;
    for (size_t i = 0; i < n_lambdas; ++i) {
        counts_out[i] = reqs[0].counts[i];
    }
    if (getenv("ROCCO_HIP_DEBUG") != nullptr) {
        fprintf(stderr, "[rocco_hip] spine: n=%zu chunks=%zu stepped[0]=%lld\n", n, (n + 31) / 32, reqs[0].stepped[0]);
    }
    return ROCCO_HIP_OK;
}

int delta_probe(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                double gamma, size_t n, const uint8_t *emap_dev, const double *lambdas, size_t n_lambdas,
                rocco_hip_probe_stats *stats_out, hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = (n > 1) ? switch_costs_dev : nullptr;
    d.gamma = gamma;
    d.n = n;
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    int rc;
    if ((rc = prepare(ev, problems, nullptr)) != ROCCO_HIP_OK) return rc;
    ev.probs[0].emap = const_cast<uint8_t *>(emap_dev);
    std::vector<ProbeRequest> reqs(1);
    reqs[0].problem = 0;
    reqs[0].lambdas.assign(lambdas, lambdas + n_lambdas);
    if ((rc = ev.probe(reqs)) != ROCCO_HIP_OK) return rc;
    for (size_t i = 0; i < n_lambdas; ++i) {
        stats_out[i].count = reqs[0].results[i].count;
        stats_out[i].uncertain = reqs[0].results[i].uncertain;
        stats_out[i].effect = reqs[0].results[i].effect;
        stats_out[i].max_run = reqs[0].results[i].max_run;
    }
    return ROCCO_HIP_OK;
}

int delta_model_lean(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n, const uint8_t *emap_dev,
                     const double *lambdas, size_t n_lambdas, long long *counts_out, long long *open_out, hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = nullptr;
    d.gamma = gamma;
    d.n = n;
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    int rc;
    if ((rc = prepare(ev, problems, nullptr)) != ROCCO_HIP_OK) return rc;
    ev.probs[0].emap = const_cast<uint8_t *>(emap_dev);
    ev.probs[0].map_version = 1;
    if ((rc = ev.lean_prepare()) != ROCCO_HIP_OK) return rc;
    ev.model_any_ = true;
    if (!ev.model_eligible(0)) {
        set_last_error("rocco_hip_delta_model_lean_f64: the lean evaluation is switched off for this solver");
        return ROCCO_HIP_EINVAL;
    }
    for (size_t at = 0; at < n_lambdas; at += (size_t)kLeanMaxPoints) {
        std::vector<ProbeRequest> reqs(1);
        reqs[0].problem = 0;
        const size_t k = std::min((size_t)kLeanMaxPoints, n_lambdas - at);
        reqs[0].lambdas.assign(lambdas + at, lambdas + at + k);
        if ((rc = ev.probe(reqs)) != ROCCO_HIP_OK) return rc;
        for (size_t i = 0; i < k; ++i) {
            counts_out[at + i] = reqs[0].results[i].count;
            open_out[at + i] = reqs[0].results[i].uncertain;
        }
    }
    return ROCCO_HIP_OK;
}

int delta_bound_rounds(rocco_hip_solver *solver, const double *scores_dev, double gamma, size_t n,
                       const double *lambdas, const int *round_sizes, int n_rounds, double *lambdas_used_out,
                       long long *counts_out, long long *level_len_out, hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = nullptr;
    d.gamma = gamma;
    d.n = n;
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    int rc;
    if ((rc = prepare(ev, problems, nullptr)) != ROCCO_HIP_OK) return rc;
    if (!ev.lean_eligible(0)) {
        set_last_error("rocco_hip_delta_bound_rounds_f64: the lean evaluation is switched off for this solver");
        return ROCCO_HIP_EINVAL;
    }
    const int qexp = ev.probs[0].qexp;
    size_t at = 0;
    for (int r = 0; r < n_rounds; ++r) {
        std::vector<ProbeRequest> reqs(1);
        reqs[0].problem = 0;
        reqs[0].bound = true;
        for (int i = 0; i < round_sizes[r]; ++i) {
            const double x = std::ldexp(std::nearbyint(std::ldexp(lambdas[at + (size_t)i], -qexp)), qexp);
            reqs[0].lambdas.push_back(x);
            lambdas_used_out[at + (size_t)i] = x;
        }
        if (reqs[0].lambdas.empty() || reqs[0].lambdas.size() > (size_t)kLeanMaxPoints) {
            return ROCCO_HIP_EINVAL;
        }
        if ((rc = ev.probe(reqs)) != ROCCO_HIP_OK) return rc;
        for (int i = 0; i < round_sizes[r]; ++i) {
            counts_out[at + (size_t)i] = reqs[0].results[(size_t)i].count;
        }
        level_len_out[r] = ev.lean_[0].levels.back().m;
        at += (size_t)round_sizes[r];
    }
    return ROCCO_HIP_OK;
}

int delta_window(rocco_hip_solver *solver, const double *scores_dev, const double *switch_costs_dev,
                 double gamma, size_t n, const uint8_t *emap_dev, double lambda_lo, double lambda_hi,
                 uint8_t *solution_dev, rocco_hip_window_stats *stats_out, hipStream_t stream)
{
    HipEvaluator ev(solver, stream);
    DevProblem d;
    d.scores = scores_dev;
    d.costs = (n > 1) ? switch_costs_dev : nullptr;
    d.gamma = gamma;
    d.n = n;
    d.solution = solution_dev;
    ev.probs.push_back(d);
    std::vector<ChainProblem> problems(1);
    problems[0].n = n;
    problems[0].gamma = gamma;
    int rc;
    if ((rc = prepare(ev, problems, nullptr)) != ROCCO_HIP_OK) return rc;
    ev.probs[0].emap = const_cast<uint8_t *>(emap_dev);
    std::vector<WindowRequest> reqs(1);
    reqs[0].problem = 0;
    reqs[0].lambda_lo = lambda_lo;
    reqs[0].lambda_hi = lambda_hi;
    if ((rc = ev.window(reqs)) != ROCCO_HIP_OK) return rc;
    const WindowResult &w = reqs[0].result;
    stats_out->count_lo = w.count_lo;
    stats_out->count_hi = w.count_hi;
    stats_out->n_diff = w.n_diff;
    stats_out->diff_adjacent = w.diff_adjacent ? 1 : 0;
    stats_out->overflow = w.overflow ? 1 : 0;
    stats_out->max_run = w.max_run;
    for (size_t k = 0; k < 16; ++k) {
        if (k < w.diffs.size()) {
            stats_out->diff_locus[k] = w.diffs[k].locus;
            stats_out->diff_margin_lo[k] = w.diffs[k].margin_lo;
            stats_out->diff_margin_hi[k] = w.diffs[k].margin_hi;
            stats_out->diff_run[k] = w.diffs[k].run;
            stats_out->diff_cls_lo[k] = w.diffs[k].cls_lo;
            stats_out->diff_cls_hi[k] = w.diffs[k].cls_hi;
        } else {
            stats_out->diff_locus[k] = -1;
            stats_out->diff_margin_lo[k] = stats_out->diff_margin_hi[k] = 0.0;
            stats_out->diff_run[k] = 0;
            stats_out->diff_cls_lo[k] = stats_out->diff_cls_hi[k] = 0;
        }
    }
    return ROCCO_HIP_OK;
}

int solve_budget_batch(rocco_hip_solver *solver, size_t n_tasks, const rocco_hip_budget_task *tasks,
                       rocco_hip_budget_result *results, hipStream_t stream, const double *score_stats_host)
{
    HipEvaluator ev(solver, stream);
    std::vector<ChainProblem> problems(n_tasks);
    for (size_t t = 0; t < n_tasks; ++t) {
        DevProblem d;
        d.scores = tasks[t].scores_dev;
        d.costs = (tasks[t].n > 1) ? tasks[t].switch_costs_dev : nullptr;
        d.gamma = tasks[t].gamma;
        d.n = tasks[t].n;
        d.solution = tasks[t].solution_dev;
        ev.probs.push_back(d);
        problems[t].n = tasks[t].n;
        problems[t].gamma = tasks[t].gamma;
        problems[t].target_count = tasks[t].target_count;
        problems[t].sum_costs = tasks[t].sum_costs;
        problems[t].max_iter = tasks[t].max_iter;
    }
    int rc;
    SearchOptions opt;
    opt.force_exact = solver->force_exact != 0;
    opt.spec_depth = solver->spec_depth;
    if (const char *e = std::getenv("ROCCO_HIP_BIG_ROUND")) opt.big_round_loci = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HIP_SMALL_ROUND")) opt.small_round_loci = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HIP_SURVEY_GATE")) opt.survey_gate = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HIP_TINY_ROUND")) opt.tiny_round_loci = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HIP_MAP_REBUILD")) opt.map_rebuild_ratio = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HIP_BOUNDS")) opt.use_bounds = std::atoi(e) != 0;
    if (const char *e = std::getenv("ROCCO_HIP_SEARCH_GATE")) opt.search_gate = std::atof(e);
    if (const char *e = std::getenv("ROCCO_HIP_SEARCH_POINTS")) opt.search_points = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("ROCCO_HIP_SEARCH_INTERP")) opt.search_interpolate = std::atoi(e) != 0;
    std::vector<CalibrationResult> res;
    const double t_solve0 = HipEvaluator::now_us();
    ev.marks_on_ = std::getenv("ROCCO_HIP_TIMING") != nullptr && std::atoi(std::getenv("ROCCO_HIP_TIMING")) >= 2;
    ev.mark("solve begins");
    if (const char *e = std::getenv("ROCCO_HIP_COMPACT")) opt.use_compaction = std::atoi(e) != 0;
    if (const char *e = std::getenv("ROCCO_HIP_PILOT_ROUNDS")) opt.pilot_rounds = std::atoi(e);
    if (const char *e = std::getenv("ROCCO_HIP_PILOT_POINTS")) opt.pilot_points = std::atoi(e);
    if (const char *e = std::getenv("ROCCO_HIP_ALIGN_MAPS")) opt.align_maps = std::atoi(e) != 0;
    if (const char *e = std::getenv("ROCCO_HIP_ALIGN_WINDOWS")) opt.align_windows = std::atoi(e) != 0;
    if (const char *e = std::getenv("ROCCO_HIP_PILOT_LEVELS")) {  // comma list, e.g. "2.2" or "2.2,1.0"
        opt.pilot_levels.clear();
        for (const char *q = e; *q != '\0';) {
            char *end = nullptr;
            const double v = std::strtod(q, &end);
            if (end == q) break;
            if (v > 0.0) opt.pilot_levels.push_back(v);
            q = (*end == ',') ? end + 1 : end;
        }
    }
    struct LeanOverride {  // ROCCO_HIP_LEAN overrides the solver's setting for this call only
        rocco_hip_solver *solver;
        int saved;
        ~LeanOverride() { solver->lean = saved; }
    } lean_override{solver, solver->lean};
    if (const char *e = std::getenv("ROCCO_HIP_LEAN")) solver->lean = std::atoi(e) != 0;
    // the threshold search of every eligible problem as one chain of launches behind the statistics pass (chain.h); what
    // it leaves open the host-sequenced rounds of calibrate_batch finish
    std::vector<double> chain_stats;
    std::vector<Presearch> presearch;
    bool chained = false;
    if (score_stats_host == nullptr) {
        if ((rc = ev.chain_search(problems, opt, chain_stats, presearch, chained)) != ROCCO_HIP_OK) return rc;
    }
    const double tp0 = HipEvaluator::now_us();
    ev.mark("threshold chain parsed");
    if ((rc = prepare(ev, problems, nullptr, score_stats_host, chained ? &chain_stats : nullptr)) != ROCCO_HIP_OK) return rc;
    const double tp1 = HipEvaluator::now_us();
    if ((rc = calibrate_batch(ev, problems, opt, res, chained ? &presearch : nullptr)) != ROCCO_HIP_OK) return rc;
    const double tp2 = HipEvaluator::now_us();
    ev.mark("calibration ends");
    if ((rc = ev.scatter_all()) != ROCCO_HIP_OK) return rc;
    ROCCO_HIP_TRY(hipStreamSynchronize(stream));
    const double t_prepare = tp1 - tp0, t_calibrate = tp2 - tp1, t_scatter = HipEvaluator::now_us() - tp2;
    g_model_chain_counters[0].fetch_add(ev.model_chains, std::memory_order_relaxed);
    g_model_chain_counters[1].fetch_add(ev.model_chain_facts, std::memory_order_relaxed);
    g_model_chain_counters[2].fetch_add(ev.model_chain_hits, std::memory_order_relaxed);
    g_model_chain_counters[3].fetch_add(ev.model_chain_misses, std::memory_order_relaxed);
    g_model_chain_counters[4].fetch_add(ev.model_chain_written, std::memory_order_relaxed);
    g_model_chain_counters[5].fetch_add(ev.model_chain_windows_answered, std::memory_order_relaxed);
    if (std::getenv("ROCCO_HIP_DEBUG") != nullptr || std::getenv("ROCCO_HIP_TIMING") != nullptr) {
        const double total = HipEvaluator::now_us() - t_solve0;
        std::fprintf(stderr, "[host] solve %.0f us: %d rounds (%d with rounding-model kernels): in the rounds %.0f us = waiting for the device %.0f "
                             "+ reading results %.0f + preparing and submitting %.0f; search logic and the rest %.0f us\n",
                     total, ev.rounds_all, ev.rounds, ev.t_round_, ev.t_wait_, ev.t_consume_,
                     ev.t_round_ - ev.t_wait_ - ev.t_consume_, total - ev.t_round_);
        std::fprintf(stderr, "[host] lean rounds: building the descriptors %.0f us, their upload call %.0f, the launch calls %.0f; general rounds: "
                             "preparing %.0f, launching %.0f\n",
                     ev.t_lean_cpu_, ev.t_lean_h2d_, ev.t_lean_launch_, ev.t_prep_, ev.t_launch_);
        std::fprintf(stderr, "[host] chained search: queued in %.0f us, waited for %.0f, report read in %.0f; statistics -> problems %.0f; "
                             "calibration after it %.0f; scatter and the last wait %.0f\n",
                     ev.t_chain_submit_, ev.t_chain_wait_, ev.t_chain_parse_, t_prepare, t_calibrate, t_scatter);
        std::fprintf(stderr, "[host] chained rounding-model rounds: %lld chains queued in %.0f us, their rounds waited for %.0f us; %lld counts taken "
                             "over, %lld requests answered from them, %lld not\n",
                     ev.model_chains, ev.t_mchain_submit_, ev.t_mchain_wait_, ev.model_chain_facts, ev.model_chain_hits, ev.model_chain_misses);
        std::fprintf(stderr, "[host] objectives fetched behind final windows: %lld rounds, %lld sums used; solutions written by the chains: %lld, "
                             "final windows answered from them: %lld\n",
                     ev.objective_prefetches, ev.objective_prefetch_hits, ev.model_chain_written, ev.model_chain_windows_answered);
        ev.mark("solve ends");
        for (size_t k = 0; k < ev.marks_.size(); ++k) {
            std::fprintf(stderr, "[mark] %8.1f  +%6.1f  %s\n", ev.marks_[k].second - t_solve0,
                         k ? ev.marks_[k].second - ev.marks_[k - 1].second : 0.0, ev.marks_[k].first);
        }
    }
    for (size_t t = 0; t < n_tasks; ++t) {
        results[t].selection_penalty = res[t].selection_penalty;
        results[t].penalized_value = res[t].penalized_value;
        results[t].selected_count = res[t].selected_count;
        results[t].evaluations = res[t].evaluations;
        results[t].path = res[t].path;
        results[t].passes = res[t].passes;
        results[t].zone_iters = res[t].zone_iters;
        results[t].n_diff = res[t].n_diff;
        results[t].maps = res[t].maps;
    }
    return ROCCO_HIP_OK;
}

}  // namespace rocco
