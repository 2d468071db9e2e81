// gangfit_device.h — shared declarations between the HIP kernels (gangfit_kernels.hip) and the C-ABI host
// layer (gangfit_api.cpp).  gfx950 / CDNA4 only: wave64, no compatibility paths.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gangfit.h"

namespace gangfit {

// The node table as the kernels see it.  "Slot" space: slots [0, n_x) are the executor priority order
// (executorNodePriorityOrder, already permuted so that a wave scanning slots b..b+63 issues three coalesced
// 512-byte loads); slots [n_x, n_slots-1) hold driver candidates that are not executor candidates; the last slot
// is a sentinel (available = -2^62) that every unknown node name maps to.  SoA, int64, one array per dimension.
struct NodeTable {
    int64_t* cpu;  // milli-cores   [n_slots]
    int64_t* mem;  // bytes         [n_slots]
    int64_t* gpu;  // devices       [n_slots]
    const uint32_t* slot_node;  // [n_slots] slot -> caller's node index (what is written to exec_nodes)
    const uint32_t* dslot;      // [n_d]     position in driverNodePriorityOrder -> slot
    const uint32_t* node_slot;  // [n_nodes] caller's node index -> slot
    // Chunk-maxima index: cmax[d * n_chunks + c] = max over slots [64c, 64c+64) of dimension d, taken on the SNAPSHOT.
    // Upper bounds stay valid while a FIFO chain subtracts, so "cmax < request in some dimension" proves that no slot
    // of the chunk can host the request: the scans skip such chunks without loading them.  The reference's priority
    // order (least free memory first) makes this skip the whole front of the order for large executors.
    const int64_t* cmax;        // [3][n_chunks]
    uint32_t n_chunks;          // ceil(n_slots / 64)
    uint32_t n_x;
    uint32_t n_d;
    uint32_t n_slots;
    uint32_t n_nodes;
    uint32_t d_identity;  // 1 when dslot[i] == i for all i (driver order is a prefix of the executor order)
    uint32_t x_skip;      // leading executor-order slots with a negative component (dead for every app: cap == 0)
    uint32_t d_skip;      // leading driver-order positions that sit on dead slots / unknown nodes
};

// Kernel-visible counters used by tests/bench to report visited bytes honestly (SURVEY.md section 8d
// "early-exit note").  One uint64 pair per launch, accumulated with a single atomic per app.
struct ScanStats {
    unsigned long long exec_slots_visited;    // executor-order slots whose capacity was evaluated
    unsigned long long driver_slots_visited;  // driver-order positions whose fit was evaluated
    unsigned long long fifo_shader_cycles;    // s_memtime delta over the last FIFO chain kernel (shader clock)
    unsigned long long fifo_realtime_ticks;   // s_memrealtime delta over the same span (constant 100 MHz)
    unsigned long long fifo_phase_cycles[6];  // wave-0 shader cycles: stage | driver scan | executor scan | slow path | commit | steps
};

// Launchers (defined in gangfit_kernels.hip).  scratch: 2 * total_k uint32 (DistributeEvenly survivor lists).
hipError_t launch_fit_independent(gf_algo algo, const NodeTable& table, uint32_t n_apps, const gf_app* d_apps,
                                  gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                                  uint64_t scratch_half, ScanStats* d_stats, hipStream_t stream);

// FIFO chain: one workgroup of n_waves (1, 4 or 16) wavefronts; the first lds_slots slots of the working table are
// kept in LDS (24 bytes per slot + fifo_fixed_lds_bytes).  Followed by the slot->node translation kernel.
size_t fifo_fixed_lds_bytes(int n_waves);
hipError_t launch_fit_fifo_chain(gf_algo algo, int n_waves, const NodeTable& table, uint32_t lds_slots,
                                 uint32_t n_apps, const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes,
                                 uint32_t* d_scratch, uint64_t scratch_half, int32_t* d_chain_failed_at,
                                 ScanStats* d_stats, hipStream_t stream);

// Device self-test of the wave primitives (DPP scan, exact clamped division) against plain reference code.
// Writes the number of mismatching lanes/cases to *d_mismatch.
hipError_t launch_selftest(uint64_t seed, uint32_t n_cases, uint32_t* d_mismatch, hipStream_t stream);

}  // namespace gangfit
