// rocco_amd/csrc/common.h -- shared host-side plumbing for librocco_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <memory>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rocco_hip.h"

namespace rocco {

void set_last_error(const std::string &msg);

#define ROCCO_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (void)hipGetLastError(); /* (the runtime keeps the error for the thread's next hipGetLastError(): a later */ \
                                     /* call's launch check must not find this one -- round 5, a retried allocation) */ \
            ::rocco::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));      \
            return (_e == hipErrorOutOfMemory) ? ROCCO_HIP_ENOMEM : ROCCO_HIP_EHIP;          \
        }                                                                                    \
    } while (0)

// Growable device / pinned-host buffers owned by a solver handle.
struct DeviceBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t want);
    void release();
};

struct PinnedBuffer {
    void *ptr = nullptr;
    size_t bytes = 0;
    bool coherent = false;  // host-coherent (fine-grained): the host may read what a kernel wrote while the stream is still busy
    int reserve(size_t want);
    void release();
};

// Whittaker LDL^T factor of the longest row seen on a device: read-only once built, so every solver of the device shares it
// (the host threads of the count-path batch run one solver each); a longer or differently penalised one replaces it as a
// whole and the old one lives until the last call using it has returned.
struct SharedFactor {
    DeviceBuffer buf;
    double lambda = 0.0;
    size_t cap = 0;
    int device = 0;
    ~SharedFactor();
};

}  // namespace rocco

struct rocco_hip_solver {
    int device = 0;
    // tunables
    int force_exact = 0;
    int spec_depth = 2;
    int active_set = 1;  // skip blocks that a survey proved settled for the whole bracket
    int lean = 1;        // threshold search on compacted levels + layer-3 rounds on the compacted problem (lean.hip)
    int rolling_group_min = 1;  // least rows per workgroup of the batched rolling launch (1, 2, 4, 8): a caller that runs other
                                // work beside the launch wants fewer, fuller workgroups (a workgroup holds a CU's registers)
    // scratch
    rocco::DeviceBuffer dev_tasks;    // kernel task descriptors
    rocco::DeviceBuffer dev_params;   // per-launch lambda lists etc.
    rocco::DeviceBuffer dev_results;  // per-launch counters / values
    rocco::DeviceBuffer dev_bits;     // exact-path decision bits
    rocco::DeviceBuffer dev_misc;     // decode / reduction scratch
    rocco::DeviceBuffer dev_median_partials;  // per-workgroup score statistics of a median launch
    rocco::DeviceBuffer dev_solution; // solution scratch when the caller wants counts only
    rocco::DeviceBuffer dev_maps;     // per-chunk binade maps of the problems being solved
    rocco::DeviceBuffer dev_frozen;   // per-block frozen summaries of the problems being solved
    std::shared_ptr<rocco::SharedFactor> factor;  // the device's Whittaker LDL^T factor this solver last used (whittaker.hip)
    rocco::DeviceBuffer dev_lean_pool;   // compacted levels of the problems being solved (lean.hip)
    rocco::DeviceBuffer dev_lean_round;  // per-round scratch of the lean evaluation (tile records)
    rocco::DeviceBuffer dev_lean_look;   // its tickets, error word and hand-off granules (restored by every round)
    rocco::DeviceBuffer dev_lean_progress;  // progress counters of chained rounds that are one launch each (lean.h: LeanRoundReset)
    int lean_look_dirty = 1;             // ... unless a round failed: then the next one initialises it again
    rocco::DeviceBuffer dev_lean_desc;   // its descriptors
    rocco::DeviceBuffer dev_lean_wcap;   // per problem: tolerance cap of the rounding-model evaluation
    rocco::DeviceBuffer dev_chain;       // state, descriptors and results of the device-sequenced threshold search (chain.hip)
    rocco::PinnedBuffer host_chain;      // ... its inputs going up and its report coming back
    rocco::DeviceBuffer dev_map;         // binade maps built by the lean kernels: descriptors, gains
    rocco::PinnedBuffer host_map_stage;  // ... their descriptors going up, their error word coming back
    rocco::DeviceBuffer dev_objective;   // ... their descriptors, partial sums and sums on the device
    rocco::PinnedBuffer host_objective;  // objective sums fetched behind the final windows: descriptors up, sums back
    rocco::PinnedBuffer host_follow;     // host-coherent: what the chained rounding-model rounds publish round by round (model_chain.h)
    rocco::PinnedBuffer host_lean_stage; // ... their pinned staging
    rocco::PinnedBuffer host_lean_back;  // ... and the readback of its results
    rocco::PinnedBuffer host_stage;   // pinned staging for uploads
    rocco::PinnedBuffer host_back;    // pinned staging for readbacks
    int wls_sorted_rows = 0;          // rows of the last centred-WLS call whose trend fit sorted its pairs (wls.hip)
    rocco::PinnedBuffer host_table;   // the interval table of the last rocco_hip_decode_runs_table call
};
