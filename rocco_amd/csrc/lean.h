// rocco_amd/csrc/lean.h -- count-only evaluation of the chain solve in exact arithmetic, and compaction
// of a problem to the loci that can still be selected.
//
// What it computes: for a list of penalties x (multiples of the problem's grid q) the number of loci the
// delta-form recursion of oracle/delta_oracle.c selects when every input is rounded to the grid q and no
// rounding model is applied ("bound" evaluation):
//     delta_0 = a_0,  delta_j = clamp(delta_{j-1}, -c, c) + a_j,  a_j = rn_q(s_j) - x,  c = rn_q(gamma)
//     class_j = ONE if delta_j > c, ZERO if delta_j <= -c, else COPY (last locus: ONE iff delta > 0), backward fill.
// This is the reference's forward pass and backtrack (rocco/_chain_dp.c:115-186) in exact arithmetic.  The
// host search (search.cpp) shifts x by -/+ eps to bracket the reference's own count.
//
// Levels: level 0 is the caller's score array.  Because the selected sets are nested in x (minimal
// minimisers of a parametric cut), every evaluation at x >= b only needs the loci selected at b: a deeper
// level holds those loci, the runs separated by one "separator" locus whose score no penalty >= b can
// select, so that a plain chain over the level reproduces every delta of the kept loci exactly.
#pragma once

#include "common.h"

namespace rocco {

constexpr int kLeanChunk = 32;                                // loci per lane
constexpr int kLeanThreads = 256;                             // lanes per workgroup
constexpr int kLeanTile = kLeanChunk * kLeanThreads;          // 8192 loci per workgroup
constexpr int kLeanBatch = 8;                                 // penalties one workgroup carries in registers
constexpr int kLeanMaxPoints = 64;                            // penalties per task and round

// per (tile, penalty): what the finish kernel needs to close the backward fill and to place the tile's
// loci in a compacted array
struct LeanTileRec {
    unsigned base;   // selected loci of the tile if the fill value entering from the right is 0
    unsigned tail;   // loci at the tile's end that copy that value
    unsigned cells;  // loci + separators the tile contributes to a compaction (the tile's last run end excluded)
    unsigned flags;  // bit 0: every locus of the tile copies; bit 1: value leaving to the left (if not bit 0);
                     // bit 2: first locus kept; bit 3: last locus kept; bit 4 (model tasks): a class of the tile is
                     // not certified
};

struct LeanTask {
    const double *s;      // level array (raw scores, separators included)
    long long m;          // its length
    double c_raw;         // switch cost (gamma)
    double magic;         // 1.5 * 2^(52 + qexp): (x + magic) - magic rounds x to the grid q
    double big;           // 2^(50 + qexp)
    int n_tiles;
    int n_points;
    int n_groups;         // workgroups per tile (each takes `batch` penalties)
    int unit_begin;       // first ticket of this task (tickets are tile-major: tile * n_groups + group)
    int point_begin;      // offset of the task's penalties in the round's point list
    int rec_begin;        // offset of the task's records: [(point) * n_tiles + tile]
    long long bits_begin; // offset (in words) of the task's kept-locus words: [(point) * n_tiles * 256 + tile * 256 + lane]
    long long off_begin;  // offset of the task's tile offsets: [(point) * n_tiles + tile]
    int result_begin;     // offset of the task's results: [point]
    int tile_stride;      // 1; a pilot task evaluates every tile_stride-th tile of the array ...
    int independent;      // ... each as a chain of its own (estimates from a sample; nothing is stored)
    int store;            // 1: keep the kept-locus words and tile offsets of every penalty (compaction candidates);
                          // 2: keep what writes a SOLUTION later (lean_write_solutions_kernel): per lane the selected-locus word
                          //    for either value the fill may carry into the tile from the right ([which][penalty][tile][lane],
                          //    which = 1 first), and per tile that value itself in place of the offset (finish kernel)
    // rounding-model tasks (lean_model_kernel): the reference's own arithmetic per chunk of 32 loci, as
    // oracle/delta_oracle.c defines it from the binade map; penalties are arbitrary doubles
    const uint8_t *emap;  // binade code of every chunk (nullptr: a bound task)
    const double *wcap;   // upper bound on the tolerance any locus can inherit (sum of the hazard chunks' weights)
    const unsigned *clean_chunks;  // [128] clean chunks per binade code exponent (a penalty tying on that grid turns them hazard)
    double cmax, sabs;    // largest switch cost, largest |score| (floor of the hazard chunks' exponent)
    int qexp;
    int batch;            // penalties per workgroup of this task (bound: 8, 4 or 2; rounding model: 4 or 2): fewer when the
                          // whole round still fits the device at once -- a workgroup's time grows with what it carries
};

struct LeanResult {
    long long count;      // selected loci
    long long child_len;  // length of the level a compaction at this penalty would produce
    long long flags;      // model tasks: nonzero = the count is not certified equal to the reference's
};

// Sizes of a round whose shape is decided on the device (chain.hip): its launches are queued before the sizes exist, with
// fixed grids, and read them here when they run.
struct LeanRoundCtl {
    int n_tasks;       // tasks of the evaluation launch
    int n_units;       // its tickets (workgroups of the fixed grid keep taking tickets until they run out)
    int n_pairs;       // (task, penalty) pairs of the finish launch
    int n_pre_tasks;   // compactions in front of the evaluation
    int n_pre_blocks;  // ... and their workgroup slots
    unsigned error;    // error words of every round so far (bit 0: a tile gave up waiting, bit 1: a compaction overflowed)
    int round;         // rounds the director has planned
    int all_done;      // nothing left to plan: the remaining launches of the chain find every size zero
};

struct LeanLaunch {
    const LeanTask *tasks;
    int n_tasks;
    int n_units;                // workgroups of the evaluation launch
    const double *points;       // penalties of every task
    unsigned *ticket;           // one word, 0xFFFFFFFF before the launch
    unsigned long long *look;   // [(rec index) * 4 + {lo, hi, a, out}] granules, all-ones before the launch
    LeanTileRec *recs;
    unsigned *bits;
    unsigned *tile_off;
    LeanResult *results;        // (may be pinned host memory: written once per task and penalty by the finish kernel)
    unsigned *error;            // device word, zero before the launch; set when a bounded wait gave up
    unsigned *error_out;        // where the finish kernel copies it for the host (next to the results), or nullptr
    int self_reset;             // the finish kernel restores tickets, granules and the error word for the next round
    int pad;
    LeanRoundCtl *ctl;          // nullptr: n_tasks / n_units above hold; else the sizes are read from the device (chain.hip)
};

// A chained round as ONE launch (round 5: lean_round_chain_kernel): compaction, evaluation and finish used to be three
// launches behind the director's; now the workgroups of one launch take the compactions' blocks as tickets, then (once every
// block has been written -- each is held by a RUNNING workgroup, so the wait ends) the evaluation's tickets, and the
// workgroup that completes the last tile of a (task, penalty) pair finishes that pair.  `progress`: words that are zero
// before the launch -- [0] compaction tickets taken, [1] compaction blocks done, [kLeanProgressPairs + result index] tiles of
// a pair done.  What the finish launch did once per round (tickets back to all-ones, the error word to the round sizes)
// cannot be done while workgroups still take tickets: the NEXT director kernel does it (lean_round_reset).
constexpr int kLeanProgressPairs = 8;
struct LeanRoundReset {
    unsigned *progress;   // nullptr: the rounds are three launches each (the finish kernel resets)
    int progress_words;
    unsigned *tickets;    // LeanLaunch::ticket of the rounds (words 0 and 2 are ticket counters)
    unsigned *error;      // LeanLaunch::error
};
#if defined(__HIPCC__)
// by every thread of a director kernel, first thing (before a barrier of its own)
__device__ __forceinline__ void lean_round_reset(const LeanRoundReset &R, LeanRoundCtl *ctl)
{
    if (R.progress == nullptr) {
        return;
    }
    for (int i = threadIdx.x; i < R.progress_words; i += blockDim.x) {
        R.progress[i] = 0u;
    }
    if (threadIdx.x == 0 && R.tickets != nullptr) {
        const unsigned e = *R.error;
        if (e != 0u) {
            atomicOr(&ctl->error, e);
            *R.error = 0u;
        }
        R.tickets[0] = 0xFFFFFFFFu;
        R.tickets[2] = 0xFFFFFFFFu;
    }
}
#endif

// compaction of one task at one of its evaluated penalties
struct LeanCompactTask {
    const double *s;          // parent level
    const int *orig;          // parent's original-locus map (nullptr: level 0, the identity)
    long long m;
    int n_tiles;
    int block_begin;          // first workgroup of this task in the launch
    const unsigned *bits;     // kept-locus words of the chosen penalty [tile * 256 + lane]
    const unsigned *tile_off; // offsets of the chosen penalty [tile]
    double sep;               // separator score
    double *out_s;
    int *out_orig;
    long long capacity;       // cells available at out_s / out_orig
};

// one evaluated penalty of a task turned into the level's 0/1 solution (model_chain.hip names them when a bisection ends)
struct LeanWriteTask {
    const unsigned *word1;   // [tile][lane] selected-locus words if the value entering the tile is 1 ...
    const unsigned *word0;   // ... and if it is 0
    const unsigned *entering;  // [tile] the value that does enter (finish kernel)
    uint8_t *solution;       // the level's solution bytes
    long long m;
    int n_tiles;
    int block_begin;         // first workgroup of this task in the launch (one per tile)
};
// tasks_dev[0 .. *n_tasks_dev): fixed grid, the workgroups take the tiles of every task in turn
int launch_lean_write_solutions(const LeanWriteTask *tasks_dev, const int *n_tasks_dev, int grid, hipStream_t stream);

// ---- the binade map of a level by lean kernels -------------------------------------------------
// What chain_fast.hip's map round computes (K1 ... K6: the running stay-off value p0 at both ends of every chunk of 32
// loci at one penalty, in exact arithmetic on the grid q, and its binade code) with the lean evaluation's tile kernel: the
// same per-locus operations in the same order and the same reduction trees, so the same codes bit for bit
// (tests/test_gpu_lean_map.py compares the bytes).  Tasks are LeanTasks with one penalty each (any double);
// lean_map_kernel leaves per-chunk and per-tile gains, lean_finish_kernel restores the round scratch, lean_mapcode_kernel
// writes the codes.
struct LeanMapOut {
    double *gain_chunk;  // [(task.rec_begin + tile) * 256 + lane]
    double *gain_block;  // [task.rec_begin + tile]
};
struct LeanMapCodeTask {
    const double *gain_chunk;  // the task's own [tile * 256 + lane]
    const double *gain_block;  // ... [tile]
    uint8_t *emap;
    long long m;
    double margin;
    int n_tiles;
    int block_begin;
};
int launch_lean_map(const LeanLaunch &L, const LeanMapOut &out, hipStream_t stream);
int launch_lean_mapcode(const LeanMapCodeTask *tasks_dev, int n_tasks, int n_blocks, hipStream_t stream);

int launch_lean_eval(const LeanLaunch &L, hipStream_t stream);
// the launches of a chained round (L.ctl != nullptr): fixed grids, sizes read on the device
int launch_lean_eval_chain(const LeanLaunch &L, int grid, hipStream_t stream);
int launch_lean_finish_chain(const LeanLaunch &L, int grid, hipStream_t stream);
int launch_lean_compact_chain(const LeanCompactTask *tasks_dev, LeanRoundCtl *ctl, int grid, hipStream_t stream);
// the three of them as one launch (`pre` may be nullptr: a round without compactions; `model`: rounding-model tasks)
// `finish` = 0: compactions and evaluation only, the pairs finished by launch_lean_finish_chain behind it (which then also
// restores the tickets: LeanRoundReset::tickets stays nullptr)
int launch_lean_round_chain(const LeanLaunch &L, const LeanCompactTask *pre, unsigned *progress, int grid, int model, hipStream_t stream,
                            int finish = 1);
// the same for rounding-model tasks (kLeanModelBatch penalties per workgroup)
constexpr int kLeanModelBatch = 4;
int launch_lean_model(const LeanLaunch &L, hipStream_t stream);
// wcap[0] = sum over the hazard chunks of `emap` of 32 * (4 hb + q), + u for every score of a clean chunk that rounds
// as an exact tie, + the largest hazard base (9 hb + 2 q), with the hazard exponent floored at e_floor (that of the
// largest penalty magnitude to come): what a locus can inherit at most.  `counters`: 384 words, zero before the
// launch and again after it; `clean_chunks`: 128 words out, the clean chunks per exponent.
struct LeanWcapTask {
    const uint8_t *emap;
    const double *s;
    long long m;
    int qexp;
    int e_floor;
    unsigned *counters;
    unsigned *clean_chunks;
    double *wcap;
    int block_begin;  // first workgroup of this task (one per 8192 loci)
    int pad;
};
int launch_lean_wcap(const LeanWcapTask *tasks_dev, int n_tasks, int n_blocks, hipStream_t stream);
int launch_lean_finish(const LeanLaunch &L, int n_pairs, hipStream_t stream);
int launch_lean_compact(const LeanCompactTask *tasks_dev, int n_tasks, int n_blocks, unsigned *error_dev, hipStream_t stream);
// every compacted problem of a batch in two launches: zero the callers' solution buffers, then scatter
struct LeanScatterTask {
    const uint8_t *level_solution;
    const int *orig;
    long long m;          // loci of the level
    uint8_t *full;
    long long n;          // loci of the caller's array
    int zero_begin;       // first workgroup of this task in the zero launch (16 KiB per workgroup)
    int scatter_begin;    // ... and in the scatter launch (256 loci of the level per workgroup)
};
int launch_lean_scatter_batch(const LeanScatterTask *tasks_dev, int n_tasks, int zero_blocks, int scatter_blocks,
                              hipStream_t stream);
// solution_full[orig[i]] = solution_level[i] for every kept locus (the caller zeroes solution_full first)
int launch_lean_scatter(const uint8_t *solution_level, const int *orig, long long m, uint8_t *solution_full, hipStream_t stream);

}  // namespace rocco
