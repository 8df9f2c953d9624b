// rocco_amd/csrc/chain.h -- the threshold search of the penalty calibration (DESIGN.md sections 4.4.2, 4.7) as ONE
// chain of launches that the host queues once and waits for once.
//
// What is replaced: the host-sequenced rounds of search.cpp's threshold search (rocco/dp.py:113-162 is what they
// stand in for: the first ~40 of the reference's 62 chain evaluations).  A round used to be
//     host plans -> upload -> [compact] eval finish -> download -> host waits -> host reads
// and a calibration took seven or eight of them one after the other, 40-60 us of turn-around each.  Here a small
// "director" kernel does the planning between the rounds on the device: it reads the counts the finish kernel left,
// moves the thresholds, decides the next compaction and writes the next round's task descriptors; the evaluation,
// finish and compaction launches of every round are queued in advance with fixed grids and read their sizes from the
// device (lean.h: LeanRoundCtl).
//
// What stays on the host: certification.  The director only CHOOSES penalties -- every (penalty, count) pair it
// records is the exact-arithmetic count of the problem at that penalty whatever made it pick that penalty -- and the
// host folds the recorded pairs into its own thresholds with its own epsilon when the chain has ended (search.cpp:
// calibrate_batch takes them as `Presearch`).  The statistics pass rides in front of the chain, so the host learns the
// score range only together with the records; it recomputes the grid exponent and discards the chain's work if the
// device used another one.
#pragma once

#include "lean.h"

namespace rocco {

constexpr int kChainMaxProblems = 128;  // one director wavefront slot and one LDS state block per problem
constexpr int kChainMaxLevels = 8;
constexpr int kChainMaxEvals = 384;     // certified (penalty, count) records kept per problem
constexpr int kChainMaxPilot = 192;     // sampled estimates kept per problem
constexpr int kChainMaxMults = 4;
constexpr int kChainKeep = 4;           // certified counts reported per side of the target

// what the host knows of a problem before the chain starts (uploaded)
struct ChainInput {
    const double *scores;
    long long n;
    double gamma;
    long long target;                         // selected loci wanted, clamped to [0, n] (rocco/dp.py:101)
    unsigned long long pool_begin, pool_end;  // this problem's share of the level pool (byte offsets)
    int allowed;                              // the lean evaluation may serve this problem at all
    int can_pilot;
};

struct ChainLevelReport {
    const double *s;
    const int *orig;
    long long m;
    double base, sep;
    unsigned long long pool_mark;
    unsigned *bits;
    unsigned *tile_off;
    int cap_points;
    int pad;
};

// per problem: what the director reports (one D2H copy at the end)
struct ChainProb {
    double smin, smax, sabs_sum;
    double eps;
    int qexp;
    int searching;  // the director ran the threshold search of this problem
    int done;       // 1: the search reached its stop rule; 2: the director gave up (the host goes on from the records)
    int rounds;     // certified rounds run
    int pilots;     // pilot rounds run
    int n_evals;
    int n_levels;
    int pad;
    unsigned long long pool_at;
    ChainLevelReport levels[kChainMaxLevels];
    // The certified counts the host needs: the tightest ones on either side of the target.  Every round's penalties lie
    // between the thresholds of the round before, so the counts above the target arrive with ascending penalties and the
    // others with descending ones: the last few recorded on each side are the tightest.  (All of them, in order of
    // evaluation: ChainEvals, read back for diagnostics only.)
    int n_above, n_below;
    double above_x[kChainKeep], below_x[kChainKeep];
    long long above_c[kChainKeep], below_c[kChainKeep];
};

struct ChainEvals {
    double x[kChainMaxEvals];
    long long c[kChainMaxEvals];
};

// The director's working state of a problem: a block of scalars that travels global memory -> LDS -> global memory as a
// whole at every director launch (one coalesced round trip for the whole batch instead of a chain of dependent loads).
struct ChainHot {
    // the problem
    const double *scores;
    long long n;
    double gamma;
    long long target;
    unsigned long long pool_end, pool_at;
    double smin, smax, sabs_sum, eps;
    // pilot
    double pg, pl, pilot_scale;
    double est_pg, est_pl;  // the pilot's estimated counts at pg and pl
    // thresholds
    double G, L;
    long long cG, cL;
    long long open_before;
    // reach above G while no evaluation has certified an upper threshold: first to where the pilot saw `soft_count` loci,
    // then four times as far each time that fell short
    double soft_hi, soft_span, pilot_res, soft_count;
    // the deepest level
    const double *lv_s;
    const int *lv_orig;
    long long lv_m;
    unsigned *lv_bits;
    unsigned *lv_tile_off;
    int lv_cap;
    int n_levels;
    int qexp, searching, done, rounds, pilots, n_evals;
    int phase;  // 0 not searched, 1 pilot, 2 search, 3 ended
    int pilot_left, pilot_hint, n_pilot;
    int G_real, L_real;
    int kind;  // task in flight: 0 none, 1 pilot, 2 certified
    int np;    // its penalties
    int have_soft;
    int can_pilot;
    int pilot_ready;  // the pilot's estimate is smooth enough (or its rounds are used up): waits for the batch's other pilots
    int pad;
};
static_assert(sizeof(ChainHot) % 8 == 0, "ChainHot is moved as 8-byte words");

// sampled estimates of a problem, sorted by penalty (device only)
struct ChainPilot {
    double x[kChainMaxPilot], c[kChainMaxPilot];
};

struct ChainTuning {
    int pilot_rounds;
    int pilot_points;
    int n_mults;
    int pilot_wgs;      // workgroups a pilot round should fill
    int pilot_tiles;    // tiles of a chromosome a pilot round samples (about: every (tiles / pilot_tiles)-th one, at least every 4th)
    int wgs;            // workgroups a round on compacted levels should fill
    double mults[kChainMaxMults];  // multiples of the target at which the first certified round evaluates
    double search_gate, survey_gate;
    int big_points;     // penalties of a round over a long level 0 once the pilot's hints are used up
    int interpolate;    // every other penalty near the linear estimate of the crossing (else: all equally spaced)
    double spread;      // ... the innermost pair this fraction of the outermost one's distance
    double soft_mult;   // multiple of the target up to whose estimated penalty the first round after the hints reaches
};

struct ChainArgs {
    int n_problems;
    int rec_capacity;       // tile records (and granule quadruples) the round scratch holds
    const ChainInput *inputs;
    ChainProb *probs;
    ChainHot *hot;
    ChainPilot *pilot;
    ChainEvals *evals;
    const double *stats;    // [n_problems][5] smin, smax, cmin, cmax, sum |s| (stats_final_kernel)
    LeanRoundCtl *ctl;
    LeanTask *tasks;        // [n_problems]
    double *points;         // [n_problems][kLeanMaxPoints]
    LeanResult *results;    // [n_problems][kLeanMaxPoints]
    LeanCompactTask *pre;   // [n_problems]
    char *pool;             // the solver's level pool
    long long *trace;       // nullptr, or [rounds + 1][8] timestamps of the director's phases (100 MHz clock; diagnostics)
    // Host-coherent pinned memory (or nullptr): the director that finds every search ended -- or the last one -- copies the
    // report there ([32 words: word 0 = 1 when complete][follow_words words from `probs` on, the statistics behind them]
    // [the round sizes at follow_ctl_word]) so that the host need not wait for the launches still queued behind it.
    unsigned long long *follow;
    int follow_words;
    int follow_ctl_word;
    ChainTuning tune;
    LeanRoundReset reset;   // rounds as one launch each: what their finish launch used to restore (lean.h)
};

// plan round `round` (consuming the results of round - 1); `last`: consume only
int launch_chain_director(const ChainArgs &A, int round, int last, hipStream_t stream);

}  // namespace rocco
