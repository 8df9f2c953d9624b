// rocco_amd/csrc/model_chain.h -- the rounding-model rounds of the penalty calibration (the reference's last bisection
// steps, rocco/dp.py:141-162, three to six of them per round through lean_model_kernel) as ONE chain of launches.
//
// What is replaced: a host turn-around per round.  After the threshold search (chain.h) the host used to walk the
// reference's bisection three steps at a time: plan the tree of the next midpoints -> upload -> model kernel -> finish
// -> wait -> read -> walk.  Seven such rounds ended a genome's calibration, 30-40 us of turn-around each.  Here a
// director kernel does the walking and the planning between the rounds on the device; the model and finish launches of
// every round are queued in advance with fixed grids and read their sizes from the device (lean.h: LeanRoundCtl).
//
// What stays on the host: the bisection itself.  The director only CHOOSES the penalties that are evaluated.  Every
// (penalty, count) pair whose count lean_model_kernel certified equal to the reference's is a FACT; the director
// writes them into pinned host memory round by round, the host follows while the chain is still running, and its own
// replay (search.cpp) goes on asking for the counts it needs -- the evaluator answers from the facts without device
// work when it holds them (waiting for the round that brings them if need be), and with a regular round when it does not
// (the director stopped at an outcome the model could not certify, or -- never observed -- walked another way than
// the host).  The director mirrors search.cpp (known_count, build_open_tree, the walk of a probe round) so that it
// asks exactly what the host will ask.
#pragma once

#include "lean.h"

namespace rocco {

constexpr int kModelChainMaxProblems = 128;
constexpr int kModelChainMaxDepth = 6;  // 63 penalties per problem and round at most (kLeanMaxPoints = 64)
constexpr int kModelChainMaxRounds = 16;

// per problem, uploaded once: the state of the host's bisection when the chain starts (search.h: BisectionAhead)
struct ModelChainWalk {
    double lower, upper;
    double G, L;
    double sabs, cost_max, none_from, all_upto;
    long long target, cG, cL, n;
    int iters_left;
    int G_real, L_real, cost_ok;
    int n_tiles;
    int pad[3];
    uint8_t *solution;   // the problem's (level's) solution bytes: written at the chain's end when the bisection has ended
    long long m;         // its loci
};

// per problem, on the device between the rounds
struct ModelChainState {
    double lower, upper;
    unsigned long long mask;   // open nodes of the round in flight, heap order (bit h: node h was asked)
    int iters_left;
    int active;                // 0: ended (no step left, or an outcome that was not certified)
    int n_points;
    int stopped;               // ended at an outcome the model did not certify
    // where the evaluation that set `upper` left what writes its solution (lean.h: LeanTask::store == 2): word offsets of
    // its [tile][lane] planes for either entering value, offset of its per-tile entering values; upper_known = 0: `upper`
    // was not set by an evaluation of this chain (or that round kept nothing)
    long long upper_w1, upper_w0, upper_zin;
    long long upper_count;
    int upper_known;
    int rec_begin;             // of the round in flight (the director's own prefix of tiles x penalties)
    int stored;                // the round in flight keeps solution words
    int pad;
};

// per problem, when the chain has ended (host-coherent memory, behind the report)
struct ModelChainFinal {
    double lower, upper;
    long long count;     // selected loci at `upper` (certified equal to the reference's)
    int iters_left;
    int written;         // 1: the solution of `upper` is in the problem's solution bytes (lean_write_solutions_kernel behind the chain)
};

// one evaluated penalty, written straight into pinned host memory by the director: slot [(round * problems + problem) * 64 + k]
struct ModelChainFact {
    double penalty;
    long long count;
    long long flags;  // 0: the count is certified equal to the reference's
};

// header of the report (pinned, host-coherent memory).  The host follows the chain while it runs: `published` rounds have
// their facts (and their n_points words) in place.
struct ModelChainReport {
    int published;
    int finished;        // the last director has run: nothing more will be published
    unsigned error;      // lean.h: LeanRoundCtl::error
    int stopped;         // problems that ended at an outcome the model did not certify
    int rounds_run;      // rounds that evaluated something
    int pad[3];
};

struct ModelChainArgs {
    int n_problems;
    int depth0;          // open levels of round 0 (the host's own request)
    int depth_floor;     // from the second round on: max(depth_floor, the evaluator's rule over the tiles of the round before)
    int depth_fixed;     // > 0: the evaluator's rule is overridden (ROCCO_HIP_MODEL_DEPTH)
    int adapt_batch;     // two penalties per workgroup while the round fits 512 workgroups that way
    int cap_pairs;       // (tile, penalty) pairs per round whose solution words fit the scratch (0: none are kept)
    const ModelChainWalk *walk;
    ModelChainState *state;
    LeanTask *tasks;       // [n_problems], uploaded complete; the director sets n_points, n_groups, unit_begin, rec_begin, batch
    double *points;        // [n_problems * 64]
    LeanResult *results;   // [n_problems * 64] (device memory: the director reads them)
    LeanRoundCtl *ctl;
    int *globals;          // device words of the director: [0] depth of the round in flight, [1] rounds that asked, [2] stopped
    ModelChainReport *report;  // pinned
    int *n_points_out;         // pinned [round * n_problems + problem]
    ModelChainFact *facts;     // pinned
    ModelChainFinal *finals;   // pinned [n_problems]
    // solution words: `bits` [round][cap_pairs][2][256] words, `entering` [round][cap_pairs] (the launches' LeanLaunch::bits /
    // tile_off point at them); the write tasks of the chain's end and their number
    unsigned *bits;
    unsigned *entering;
    LeanWriteTask *writes;
    int *n_writes;
    LeanRoundReset reset;  // rounds as one launch each: what their finish launch used to restore (lean.h)
};

// round: 0 .. n_rounds; the call with last = 1 only reads the last round's results
int launch_model_chain_director(const ModelChainArgs &A, int round, int last, hipStream_t stream);
// lean_model_kernel for a round whose sizes are on the device (L.ctl), fixed grid
int launch_lean_model_chain(const LeanLaunch &L, int grid, hipStream_t stream);

}  // namespace rocco
