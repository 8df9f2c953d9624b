// rocco_amd/csrc/chain_fast.h -- device descriptors of the parallel delta-form ("fast path") kernels.
//
// SPEC: identical, bit for bit, to the sequential definition in oracle/delta_oracle.c (DESIGN.md
// section 4): arithmetic grid q = 2^qexp, chunks of 32 loci carrying a binade code (clean chunks use
// the reference's own rounding grid u = 2^(e-52), hazard chunks the grid q with a per-step
// tolerance weight), constant clear-clamp guard 2^-16, tolerance tau_j = weight accumulated since
// the last provable clear clamp.
#pragma once

#include "common.h"

namespace rocco {

constexpr int kChunk = 32;                             // loci per lane (== ORACLE_CHUNK)
constexpr int kFastThreads = 256;                      // lanes per workgroup
constexpr int kFastBlockLoci = kChunk * kFastThreads;  // 8192 loci per workgroup
constexpr double kGuard = 0x1p-16;                     // == ORACLE_GUARD
constexpr int kMapBias = 64;                           // == ORACLE_MAP_BIAS
constexpr int kMapNone = 0xFF;                         // == ORACLE_MAP_NONE
constexpr int kMaxDiffs = 16;

// kModeBound: exact arithmetic on the grid q with no rounding model at all (like a map round, but it
// counts): evaluated at lambda -/+ epsilon it brackets the reference's count at lambda
enum FastMode { kModeProbe = 0, kModeWindow = 1, kModeMap = 2, kModeRecord = 3, kModeBound = 4 };

// Per-workgroup summaries of "frozen" blocks (valid for every penalty of a surveyed bracket): all
// chunks clean with one binade exponent and no rounding tie, identical classes at both ends of the
// bracket, constant block function.  Then, exactly,  delta_out(lambda) = B + m * rn_u(-lambda).
struct FrozenArrays {
    uint8_t *flag;       // 1 = frozen
    double *B;           // delta at the block's last locus minus m * rn_u(-lambda)
    int *m;              // steps since the last clamp before the block's last locus
    int8_t *e;           // binade exponent of the block's chunks
    double *gain_lo, *gain_hi;  // gain of the block at the bracket ends (maps: interpolated)
    // exact spine: gain without the block's first step = gx_lo + mg * (rn_u(-lambda) - rn_u(-lam_lo)),
    // and the first step's cost on the block's grid
    double *gx_lo, *mg, *cprev;
    int *lc;             // last jointly clear clamp (global locus index, -1 none)
    uint8_t *fv;         // fill summary of the (identical) classes
    unsigned *pend, *base;
    double *lam_lo, *lam_hi;  // bracket ends the gains were taken at
};

struct FastTask {
    const double *scores;
    const double *switch_costs;  // n-1 or nullptr
    double gamma;
    long long n;
    int qexp;
    double magic;  // 1.5 * 2^(52 + qexp)
    double big;    // 2^(50 + qexp): saturation bound of the shift component
    double qstep;  // 2^qexp
    double cmax, sabs;
    int slot_begin, slot_count;
    int n_blocks;
    long long rec_off;   // record slots: offset of this task in the [chunk][slot] record arrays
    long long rec_goff;  // ... and in the [group of 32 chunks][slot] summary arrays
    int pre_round;          // every slot of the task is a bound slot: the staged tile holds rn_q(score)
    int sel_has_upper;      // record slots: the last slot is the current upper end (else: the tree only)
    int sel_depth;          // record slots: > 0 = slots are a bisection tree (heap order) [+ the current upper end];
    long long sel_target;   //   the spine launch walks it against this target and materialises the answer
    uint8_t *solution;    // n bytes (window slots write fill(LO) here)
    const uint8_t *emap;  // binade code per chunk, or nullptr (every chunk = hazard, global exponent)
    uint8_t *emap_out;    // map slots write the new codes here
    double map_margin;
    FrozenArrays frz;      // frz.flag == nullptr: no frozen blocks in use this round
    FrozenArrays frz_out;  // frz_out.flag != nullptr: the window slot of this task is a survey
};

// One delta chain = (task, penalty).  A probe / map slot owns one chain, a window slot two
// (chain_a = lambda_lo, the larger delta; chain_b = lambda_hi).
struct FastChain {
    int task;
    double lambda;
    long long chunk_off;  // offset of this chain in the per-chunk chain arrays
    long long block_off;  // offset in the per-block chain arrays
};

struct FastSlot {
    int task;
    int mode;
    int chain_a, chain_b;
    long long chunk_off;  // per-chunk slot arrays
    long long block_off;  // per-block slot arrays
};

struct FastDiff {
    long long locus;
    double margin_lo, margin_hi;
    long long run;
    int cls_lo, cls_hi;
};

struct FastSlotResult {
    long long count_lo, count_hi;  // probe: count_lo == the exact-rule count
    long long uncertain, effect;
    long long max_run;
    long long n_diff;
    long long p16, npos;
    int e_global;
    int overflow;
    int nonadjacent;
    int pad;
    FastDiff diffs[kMaxDiffs];
};

struct FastBuffers {
    // per (chain, chunk)
    double *agg_a, *agg_lo, *agg_hi;
    uint8_t *pstar;
    // per (chain, block)
    double *blk_a, *blk_lo, *blk_hi, *din;
    // per (slot, chunk)
    int8_t *lc_chunk;
    double *w_chunk;     // tolerance weight after the last clear clamp (or of the whole chunk)
    double *gain_chunk;  // map slots: sum of max(0, delta - c) over the chunk's steps
    // per (slot, block)
    int *lc_block, *lcin_block;
    double *w_block, *win_block;
    double *gain_block, *gainin_block;
    uint8_t *bfv_lo, *bfv_hi;
    unsigned *bpend_lo, *bpend_hi, *bbase_lo, *bbase_hi;
    uint8_t *rin_lo;
    // record slots (exact spine): per (chunk, slot of the task), slot fastest
    double *rec_din, *rec_gain;
    unsigned *rec_d, *rec_v;
    uint8_t *rec_flags;  // bit 0: the parallel recursion is exact in this chunk (clean, no rounding tie)
    double *rec_gsum;    // per (group of 32 chunks, slot): sum of the chunks' gains
    uint8_t *rec_gok;    // ... and whether every chunk of the group is exact
    // per slot
    FastSlotResult *results;
};

struct FastLaunch {
    const FastTask *tasks;
    const FastChain *chains;
    const FastSlot *slots;
    const int2 *blockmap;      // global workgroup -> (task, local block): the blocks evaluated this round
    const int2 *blockmap_all;  // every block of every task (tail patches)
    int n_tasks, n_chains, n_slots, n_blocks_total, n_blocks_all;
    int slot_groups;  // K1 / K3 / class fill: workgroups per block, each taking every slot_groups-th slot
    bool any_costs, any_plain;
    bool any_window, any_map;
    FastBuffers buf;
};

int launch_fast_round(const FastLaunch &L, hipStream_t stream);

// Exact spine over the record arrays of a round (chain_spine.hip): afterwards rec_d / rec_v hold the
// reference's exact classes for every slot; then counts (and one solution per task) are rebuilt.
int launch_spine(const FastLaunch &L, int *solution_slot_dev, bool any_select, hipStream_t stream);

// min / max of scores (and of switch costs) and sum |score| per task:
// out[5 * t + {0,1,2,3,4}] = smin, smax, cmin, cmax, sum_abs
struct StatsTask {
    const double *scores;
    const double *switch_costs;
    long long n;
};
int launch_stats(const StatsTask *tasks_dev, int n_tasks, const int2 *blockmap_dev, int n_blocks_total,
                 double *partials_dev, double *out_dev, hipStream_t stream);

}  // namespace rocco
