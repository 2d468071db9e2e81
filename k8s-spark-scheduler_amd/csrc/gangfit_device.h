// gangfit_device.h — shared declarations between the HIP kernels (gangfit_kernels.hip) and the C-ABI host
// layer (gangfit_api*.cpp, gangfit_ctx.h).  gfx950 / CDNA4 only: wave64, no compatibility paths.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gangfit.h"

namespace gangfit {

// The node table as the kernels see it, in "slot" space.  SoA, int64, one array per dimension; a wave scanning slots
// b..b+63 issues three coalesced 512-byte loads.  Two layouts (gf_orders_set picks one):
//   merged  — driverNodePriorityOrder and executorNodePriorityOrder are subsequences of one common order (they both
//             derive from getNodeNamesInPriorityOrder, internal/sort/nodesorting.go:41-64, so this is the production
//             shape): slots [0, n_x) are that common order, n_d == n_x, driver position == slot (d_identity), and the
//             per-chunk bit masks say which slots are driver / executor candidates.
//   general — the two orders disagree (different label-priority re-sorts for drivers and executors): slots [0, n_x)
//             are the executor order, driver-only nodes follow, dslot[] maps driver positions to slots.
// The last slot is a sentinel (available = -2^62) that unknown node names map to.
struct NodeTable {
    int64_t* cpu;  // milli-cores   [n_slots]
    int64_t* mem;  // bytes         [n_slots]
    int64_t* gpu;  // devices       [n_slots]
    const uint32_t* slot_node;  // [n_slots] slot -> caller's node index (what is written to exec_nodes)
    const uint32_t* dslot;      // [n_d]     position in driverNodePriorityOrder -> slot (general layout only)
    const uint32_t* node_slot;  // [n_nodes] caller's node index -> slot
    // Chunk-maxima index: cmax[d * n_chunks + c] = max over slots [64c, 64c+64) of dimension d, taken on the SNAPSHOT.
    // Upper bounds stay valid while a FIFO chain subtracts, so "cmax < request in some dimension" proves that no slot
    // of the chunk can host the request: the scans skip such chunks without loading them.  The reference's priority
    // order (least free memory first) makes this skip the whole front of the order for large executors.
    const int64_t* cmax;        // [3][n_chunks]
    const uint64_t* xmask;      // [n_chunks] bit l of xmask[c]: slot 64c+l is an executor candidate
    const uint64_t* dmask;      // [n_chunks] bit l of dmask[c]: slot 64c+l is a driver candidate (merged layout)
    uint32_t n_chunks;          // ceil(n_slots / 64)
    uint32_t n_x;               // slots scanned by the executor packers
    uint32_t n_d;               // driver candidates (positions)
    uint32_t n_slots;
    uint32_t n_nodes;
    uint32_t d_identity;        // 1: driver position == slot (merged layout)
    // The SNAPSHOT in the scaled int32 domain (value = scaled value * nunit[dimension], |scaled value| < 2^30), slot order; nullptr
    // when the snapshot has no such form.  Read by the minimal-fragmentation independent batch (gangfit_minfrag.inc), whose every
    // pass evaluates every candidate slot: a capacity there is a multiply-high instead of a 64-bit quotient.
    const int32_t* ncpu = nullptr;
    const int32_t* nmem = nullptr;
    const int32_t* ngpu = nullptr;
    int64_t nunit[3] = {1, 1, 1};
};

// Sparse-dimension view of the executor order for the independent batch (merged layout): the executor candidates that have
// at least one free gpu, in priority order, as a compact table of their own ("sub-slots").  gpu nodes are a minority of a
// cluster, so a gang whose executors need a gpu collects them from a few slots of MANY 64-slot chunks of the full order —
// a chain of dependent round trips that makes it the slowest wavefront of a launch.  Every node left out has capacity 0 for
// such a request (available gpu <= 0 < request), so scanning the compact table instead gives the same placements.
struct SparseTable {
    const int64_t* cpu;  // [n_sub + pad] values of the sub-slots (the padding slots never fit)
    const int64_t* mem;
    const int64_t* gpu;
    const uint32_t* slot_node;    // sub-slot -> caller's node index
    const int64_t* cmax;          // [3][n_chunks] chunk maxima of the compact table
    const uint64_t* xmask;        // [n_chunks] all sub-slots are executor candidates
    const uint32_t* sub_of_slot;  // [n_slots of the full table] slot -> sub-slot, GF_NO_NODE when the slot is not in the view
    uint32_t n_x;                 // sub-slots (0 = no sparse view: the kernel uses the full table)
    uint32_t n_chunks;
    // the zone packers' one-launch kernel (fit_zoned_fused_kernel) packs a zone's gangs of gpu executors from the view too:
    const uint64_t* zmask;        // [n_zones of the evaluation list][n_chunks] the sub-slots whose node lies in the zone
    const uint32_t* slot_of_sub;  // [n_sub + pad] sub-slot -> slot of the full table (its placements are slot ids)
};

// Zone view of the two candidate orders for the single-AZ packers (LIB/binpack/single_az.go:23-72): zone zi of the
// evaluation list keeps only its own nodes of driverNodePriorityOrder / executorNodePriorityOrder, order preserved —
// i.e. the same slot table scanned through per-zone candidate bit masks.  The evaluation list is driverZonesInOrder
// (zones by first appearance in the driver order) restricted to zones that own at least one executor candidate (:36-41).
struct ZoneTable {
    const uint64_t* xmask;  // [n_zones][stride] executor-candidate bits of each zone (indexed like NodeTable::xmask)
    const uint64_t* dmask;  // [n_zones][stride] driver-candidate bits: by slot (merged layout) or by driver position (general)
    uint32_t n_zones;
    uint32_t stride;        // 64-bit words per zone row
};

// Node tables the efficiency kernels read (LIB/binpack/efficiency.go:66-156): AvailableResources and
// SchedulableResources, SoA int64, indexed by slot (hot path) or by the caller's node index (full per-node map).
struct EffTables {
    const int64_t* avail[3];
    const int64_t* sched[3];
};

// Narrow (scaled int32) domain of the FIFO chain — see gangfit_fifo_common.inc.
struct NApp {  // 64 bytes, produced by the chain prologue (prepare_app): requests divided by the table's units
    int32_t drv[3];
    int32_t k;
    int32_t exe[3];  // >= 0
    uint32_t flags;
    uint32_t mag[3];  // division by exe as a multiplication (narrow_magic, gangfit_fifo_common.inc): 2 * ceil(2^(30 + l) / exe), 0 when exe == 0
    uint32_t msh;     // the three post-shifts l = ceil(log2(exe)), 8 bits each
    uint64_t exec_off;
    uint64_t pad;
};
static_assert(sizeof(NApp) == 64, "NApp must be 64 bytes");

// d_dst = d_src * factor[dimension] for the 3 * n_slots table values and the 3 * n_chunks chunk maxima; sentinel values
// (<= INT32_MIN / 2: empty slots) are kept.
hipError_t launch_narrow_rescale(const int32_t* d_src, int32_t* d_dst, uint32_t n_slots, const int32_t* d_cmax_src,
                                 int32_t* d_cmax_dst, uint32_t n_chunks, const int32_t factor[3], hipStream_t stream);

struct NarrowTable {  // value = scaled value * unit[dimension]
    int32_t* cpu;  // [n_slots] working copy, scaled
    int32_t* mem;
    int32_t* gpu;
    const int32_t* cmax;  // [3][n_chunks] scaled chunk maxima of the snapshot
    int64_t unit[3];
};


// Kernel-visible counters used by tests/bench to report visited bytes honestly (SURVEY.md section 8d
// "early-exit note").  One uint64 pair per launch, accumulated with a single atomic per app.
struct ScanStats {
    unsigned long long exec_slots_visited;    // executor-order slots whose capacity was evaluated
    unsigned long long driver_slots_visited;  // driver-order positions whose fit was evaluated
    unsigned long long fifo_shader_cycles;    // s_memtime delta over the last FIFO chain kernel (shader clock)
    unsigned long long fifo_realtime_ticks;   // s_memrealtime delta over the same span (constant 100 MHz)
    unsigned long long fifo_phase_cycles[6];  // wave-0 shader cycles: stage | driver scan | executor scan | slow path | commit | steps
    // the solo chain's rare endings (gangfit_fifo_solo.inc, instrumented variant only), by return code: [1] a request without
    // a shape id, [2] the capacity bound, [3] no driver candidate, [4] the gang does not fit behind its first driver
    unsigned long long fifo_rare_count[5];
    unsigned long long fifo_rare_cycles[5];   // shader cycles inside solo_rare_app, same index
    unsigned long long fifo_hw_id;            // HW_ID of the chain's controlling wavefront (CU, SE, ...) | XCC_ID << 32
};

// Launchers (defined in gangfit_kernels.hip).  scratch: 2 * total_k uint32 (DistributeEvenly survivor lists).
// d_feasible != nullptr: feasibility only (gf_fit_feasible) — n_apps bytes (padded to a multiple of four; device-mapped pinned
// memory or a device buffer) receive HasCapacity, d_results is not written, no counters.  d_feasible_sync: ceil(n_apps / 4)
// words of device memory, zero between launches (the kernel's collecting workgroup clears them again).
hipError_t launch_fit_independent(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, uint32_t n_apps,
                                  const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                                  uint64_t scratch_half, ScanStats* d_stats, hipStream_t stream, uint8_t* d_feasible = nullptr,
                                  uint32_t* d_feasible_sync = nullptr);

// ---- the resident worker of the independent batch (gangfit_worker.inc; host side: gangfit_api_worker.cpp)
#ifndef GF_WORKER_RING
#define GF_WORKER_RING 64  // (a power of two; every translation unit must see the same value)
#endif
constexpr uint32_t kWorkerRing = GF_WORKER_RING;
constexpr uint32_t kWorkerInline = 16;  // tickets a launch of the worker carries in its arguments (at most one per set)
constexpr int kWorkerWaves = 16;             // wavefronts per workgroup of the worker kernel (one workgroup fills a CU)
constexpr uint32_t kWorkerCountStride = 64;  // words between two tickets' counters: one 256-byte line (one memory channel) each  // tickets in flight at most (host side waits for ticket t - kWorkerRing before it posts t)

// A ticket: six self-validating 8-byte words.  word[0] = ticket number + 1; words 1..5 carry worker_tag(ticket) in their
// top sixteen bits (device and pinned-host addresses, the placement count and n_apps | flags << 32 all stay below 2^48): a
// reader that finds the right tag in every word has the whole ticket without any ordering between the words.
struct WorkerTicket {  // 64 bytes
    unsigned long long word[8];  // [0] seq, [1] apps, [2] results, [3] exec_nodes, [4] exec_len, [5] n_apps | flags << 32
};
static_assert(sizeof(WorkerTicket) == 64, "one cache line per ticket");
__host__ __device__ inline unsigned long long worker_tag(unsigned long long ticket) { return ((ticket + 1ull) & 0x7FFFull) | 0x8000ull; }

struct WorkerHostCtl {  // pinned host memory, device-mapped
    unsigned long long posted;  // host -> device: tickets posted so far (the doorbell)
    unsigned long long stop;    // host -> device: 1 = leave now; N + 2 = leave once ticket N - 1 has been relayed (N tickets in all)
    unsigned long long pad0[6];
    unsigned long long consumed;  // device -> host: tickets the leader has relayed (valid once state == 2)
    unsigned long long state;     // device -> host: 1 = the leader runs, 2 = it has left
    unsigned long long pad1[6];
    unsigned long long done[kWorkerRing];  // device -> host: done[t % ring] = t + 1 when ticket t is complete
    WorkerTicket ring[kWorkerRing];        // host -> device
};

struct WorkerDevCtl {  // fine-grained device memory
    unsigned long long quit;  // = generation of the launch whose leader has left (never reset: the next launch has another one)
    unsigned long long pad0[7];
    uint32_t count[kWorkerRing * kWorkerCountStride];  // [slot * stride]: applications of the slot's ticket finished so far
    WorkerTicket ring[kWorkerRing];
};

struct WorkerArgs {
    WorkerHostCtl* host;
    WorkerDevCtl* dev;
    unsigned long long generation;    // of this launch (1, 2, ...): the value the leader writes into the quit word when it leaves
    unsigned long long first_ticket;  // tickets below this one were served by an earlier launch
    unsigned long long idle_ticks;    // the leader leaves after this long without a new ticket (100 MHz)
    unsigned long long leave_after;   // != 0: nothing behind ticket leave_after - 1 will be posted to this launch (GF_WORKER_LEAVE_AFTER)
    uint32_t* scratch;                // [kWorkerRing][3 * scratch_stride]: private placements, two survivor lists
    unsigned long long scratch_stride;
    uint32_t sets;
    uint32_t blocks_per_set;
    ScanStats* stats;  // nullable: slots visited, summed over every decision of the launch (gf_scan_stats)
    // The first tickets of the launch travel with it: tickets first_ticket .. first_ticket + n_inline - 1 as their six tagged
    // words, exactly as in the ring.  A set finds its first ticket in the kernel's argument segment — which every wavefront
    // reads at its start anyway — instead of waiting for the leader's look at the doorbell and its relay over the host link
    // (two round trips, ~4 us of a 60 us window of twenty tickets).  The leader relays them all the same (the ring is what
    // later rounds and the completion accounting index).
    uint32_t n_inline;
    uint32_t pad_inline;
    unsigned long long inline_words[kWorkerInline][6];
};

hipError_t worker_blocks_per_cu(gf_algo algo, int* out);
// One launch: 1 + sets * blocks_per_set workgroups of sixteen wavefronts.
hipError_t launch_fit_worker(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, const WorkerArgs& args,
                             hipStream_t stream);

// FIFO chain (fitEarlierDrivers + final pack) of the plain packers.  One workgroup walks the chain; two kernels:
//   solo (gangfit_fifo_solo.inc) — merged layout, scaled int32 table in LDS, ONE wavefront walking the chain: the fast
//                                  path; when a request of the batch has no scaled form it returns at once
//   v2   (gangfit_kernels.hip)   — any layout, wide (int64) table: the fallback (guarded the other way when both run)
// followed by expand_translate_kernel (run heads -> placement list, slot ids -> node indices).
struct FifoPlan {
    bool narrow;                // launch the solo kernel (merged layout and the table has a narrow form)
    bool wide;                  // launch the wide kernel (alone, or guarded by *d_wide_needed behind the solo kernel)
    uint32_t lds_slots_v2;      // table slots each kernel keeps in LDS
    uint32_t lds_slots_solo;
};
// Checkpoints of a chain (incremental Filter chains, gf_fit_batch): before it stages the application with ABSOLUTE index
// i << shift (i >= 1) the solo kernel dumps its narrow working table (3 * n_slots int32, the layout of NarrowTable) to
// base + (i - 1) * 3 * n_slots.  a_base = absolute index of the first application of this launch (a resumed chain is
// launched on the tail of the queue with the restored table); base == nullptr: no checkpoints.
struct ChainCkpt {
    int32_t* base;
    uint32_t a_base;
    uint32_t shift;
    size_t stride;                          // int32 words between two checkpoints (>= 3 * n_slots; chain_ckpt_stride)
    const unsigned long long* resume_mask;  // solo kernel with a global table tail: the cumulative dirty-chunk mask of the
                                            // checkpoint this launch resumes from (nullptr: from the snapshot)
    // The TIP (LDS chain kernels with the whole table in LDS; nullptr: none): when the chain ends — at the application it aborts at, else at
    // the driver being filtered, behind which nothing is committed — the table it holds is the table BEFORE that application.
    // The epilogue leaves it here (3 * n_slots int32, the layout of a checkpoint), and the next Filter of a queue that agrees with
    // this one up to there resumes from it: the Filter of driver j + 1 evaluates drivers j and j + 1 instead of everything since
    // the last multiple of 2^shift (gangfit_api_fit.cpp, chain_plan).
    int32_t* tip;
};
// Checkpoint layout: cpu | mem | gpu (n_slots int32 each), then — 8-byte aligned — two masks of one bit per 64-slot chunk:
//   cumulative  the chunk differs from the snapshot at this checkpoint,
//   delta       the chunk was touched by a commit since the PREVIOUS checkpoint of the chain.
// Written by the solo kernel when the table has a global tail (the DELTA format): only the delta chunks are dumped — a dump
// costs what the last 2^shift applications touched, not what the chain has touched so far — and a chain that resumes from
// checkpoint i starts from the snapshot with, for every chunk of cumulative(i), the copy in the LATEST checkpoint j <= i whose
// delta names it (chain_prologue_kernel).  Kernels that dump whole tables leave the masks unused.
inline size_t chain_ckpt_mask_words(uint32_t n_chunks) { return 2 * (size_t)((n_chunks + 63u) / 64u); }  // uint32 words of ONE mask
inline size_t chain_ckpt_stride(uint32_t n_slots, uint32_t n_chunks) {
    return 3 * (size_t)n_slots + (n_slots & 1u) + 2 * chain_ckpt_mask_words(n_chunks);
}
// What a chain launcher folds into its first and its last kernel, so that a Filter's chain is three launches instead of
// nine runtime calls (each costs ~7 us of host time alone, several times that when eight threads enqueue eight chains):
//   first kernel (chain_prologue_kernel) — reads the app records from wherever the caller has them (device-mapped pinned
//       host memory, or the device), writes the device copy and the scaled records, presets the run-head slice, copies up
//       to two tables (working table <- snapshot / checkpoint) and lays the dirty chunks of a checkpoint over the first;
//   last kernel (the translate / expand step) — also writes the final results, placements and the abort index to
//       device-mapped host buffers (posted writes, visible when the kernel has completed).
// Everything is optional; a default-constructed ChainIo means "records are in d_apps, nothing to copy, no host outputs".
struct ChainIo {
    const gf_app* apps_src = nullptr;  // device-visible source of the n_apps records (nullptr: d_apps holds them already)
    const uint32_t* copy_src[2] = {nullptr, nullptr};
    uint32_t* copy_dst[2] = {nullptr, nullptr};
    size_t copy_words[2] = {0, 0};
    const int32_t* overlay = nullptr;  // checkpoint 1 of a series in the delta format (chain_ckpt_stride): checkpoints
                                       // 1 .. overlay_count are laid over overlay_dst after the copies, the latest delta of a
                                       // chunk winning
    size_t overlay_stride = 0;         // int32 words between two checkpoints
    uint32_t overlay_count = 0;        // the checkpoint the chain resumes from (1-based)
    int32_t* overlay_dst = nullptr;
    uint32_t overlay_slots = 0, overlay_chunks = 0;
    int32_t* wide_clear = nullptr;     // the flag word the NEXT chain will use (zeroed here: the words alternate)
    gf_result* h_results = nullptr;    // [n_apps] of this launch
    uint32_t* h_exec = nullptr;        // indexed by the records' absolute exec_off
    int32_t* h_failed = nullptr;
};
size_t fifo_v2_lds_bytes(uint32_t lds_slots, uint32_t n_chunks);
size_t fifo_solo_lds_bytes(uint32_t lds_slots, uint32_t n_chunks);
// heads_lo: first entry of d_scratch this launch may write run heads to (the placements of a resumed chain start there)
hipError_t launch_fit_fifo(gf_algo algo, const FifoPlan& plan, const NodeTable& table, const NarrowTable& ntable,
                           uint32_t n_apps, const gf_app* d_apps, NApp* d_napps, int32_t* d_wide_needed,
                           gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch, uint64_t scratch_half,
                           uint64_t heads_lo, int32_t* d_chain_failed_at, const ChainCkpt& ckpt, const ChainIo& io,
                           ScanStats* d_stats, hipStream_t stream);

// Zone-aware packers on an independent batch: one wave per (app, zone) runs SparkBinPack on the zone's view, one wave
// per decision averages the packing efficiencies of [driver] ++ executors in slice order (chooseBestResult,
// single_az.go:75-97), one wave per app keeps the first zone with the strictly highest average and writes the final
// result.  az_aware: apps without a single-zone fit keep the plain tightly-pack answer already in d_results.
struct ZoneBuffers {
    gf_result* zres;      // [n_apps * n_zones]
    uint32_t* zexec;      // [n_zones][zexec_stride] placements as SLOT ids
    uint64_t zexec_stride;
    double* zavg;         // [n_apps * n_zones][4]
    uint32_t* cnt;        // [n_cnt_waves][n_slots] zero between launches: per-wave executor multiplicity scratch
    uint32_t n_cnt_waves;
    double* avg_out;      // [n_apps][4] AvgPackingEfficiency of the chosen result (zeros when infeasible); nullable
};
hipError_t launch_fit_zoned(int inner_algo, bool az_aware, bool reserve_execs, const NodeTable& table,
                            const ZoneTable& zones, const EffTables& eff, const ZoneBuffers& buf, uint32_t n_apps,
                            const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                            uint64_t scratch_half, hipStream_t stream);

// The same as ONE launch (fit_zoned_fused_kernel): a workgroup per application, a wavefront per candidate view, the choice
// through LDS, the winner's placements written as node indices to d_exec_nodes / d_results — which may be device-mapped host
// memory.  d_zexec: (n_zones + 1) rows of zexec_stride uint32 (the candidates' placements as slot ids).  At most 64 views.
hipError_t launch_fit_zoned_fused(int inner_algo, bool az_aware, const NodeTable& table, const SparseTable& gpu_view, const ZoneTable& zones,
                                  const int64_t* d_sched, uint32_t* d_zexec, uint64_t zexec_stride, uint32_t n_apps,
                                  const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                                  uint64_t scratch_half, hipStream_t stream, uint8_t* d_feasible = nullptr,
                                  uint32_t* d_feasible_sync = nullptr,  // (feasibility only: as launch_fit_independent)
                                  bool eff_nonneg = false);  // feasibility only: no node's efficiency can be negative (gf_ctx::eff_nonneg)

// FIFO chain for any packer (zone-aware ones included): one workgroup, one wavefront per candidate view of the current
// app (each zone, plus the plain pack for az-aware), working table in global memory.  buf.zexec needs
// (n_zones + 1) rows, buf.cnt at least 16 rows.  Writes final results / placements as node indices.
hipError_t launch_fit_fifo_generic(int inner_algo, bool zoned, bool az_aware, bool reserve_execs, const NodeTable& table,
                                   const ZoneTable& zones, const int64_t* d_sched, const ZoneBuffers& buf,
                                   uint32_t n_apps, const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes,
                                   uint32_t* d_scratch, uint64_t scratch_half, int32_t* d_chain_failed_at,
                                   const int32_t* d_run_if, hipStream_t stream);

// LDS-resident chain for the zone-aware tightly-pack packers on the merged layout with a narrow table
// (gangfit_fifo_zoned.inc).  Returns at once when a request of the batch has no scaled form (*d_wide_needed != 0 after
// its own prepare step): the caller then launches launch_fit_fifo_generic with d_run_if = d_wide_needed.
// d_spill: 2 * 16 rows of spill_stride uint32 (run-list tails beyond the LDS capacity).
// n_cand = candidate views = zones (+ 1 for az-aware: the plain pack); the workgroup's size follows from it.
size_t fifo_zoned_lds_bytes(uint32_t lds_slots, uint32_t n_chunks, uint32_t n_zones, uint32_t n_cand, uint32_t n_shapes);
hipError_t launch_fit_fifo_zoned_lds(bool az_aware, const NodeTable& table, const NarrowTable& ntable, const ZoneTable& zones,
                                     const int64_t* d_sched, uint32_t lds_slots, uint32_t n_shapes, uint32_t n_apps,
                                     const gf_app* d_apps, NApp* d_napps, int32_t* d_wide_needed, gf_result* d_results,
                                     uint32_t* d_exec_nodes, uint32_t* d_spill, uint64_t spill_stride,
                                     int32_t* d_chain_failed_at, const ChainCkpt& ckpt, const ChainIo& io, ScanStats* d_stats,
                                     hipStream_t stream);

// LDS-resident, block-cooperative chain for minimal-fragmentation, plain (zoned = false) or single-AZ
// (gangfit_fifo_minfrag.inc); same contract as launch_fit_fifo_zoned_lds.
size_t fifo_minfrag_lds_bytes(uint32_t lds_slots, uint32_t n_chunks, uint32_t n_zones, uint32_t n_shapes);
// int32 words of the capacity histograms (one per candidate view and executor shape; gangfit_fifo_minfrag.inc)
size_t fifo_minfrag_hist_words(uint32_t n_zones, uint32_t n_shapes);
hipError_t launch_fit_fifo_minfrag_lds(bool zoned, const NodeTable& table, const NarrowTable& ntable, const ZoneTable& zones,
                                       const int64_t* d_sched, uint32_t lds_slots,
                                       uint32_t n_shapes /* shape ids per role */, uint32_t n_idx /* ... with index rows in LDS */,
                                       uint32_t n_apps,
                                       const gf_app* d_apps, NApp* d_napps, int32_t* d_wide_needed, gf_result* d_results,
                                       uint32_t* d_exec_nodes, uint32_t* d_spill, uint64_t spill_stride,
                                       int32_t* d_chain_failed_at, int32_t* d_capmat /* n_shapes x n_slots, nullable */,
                                       int32_t* d_hist /* fifo_minfrag_hist_words, nullable: no histogram path */,
                                       const ChainCkpt& ckpt, const ChainIo& io, ScanStats* d_stats /* nullable */,
                                       hipStream_t stream);

// ComputeAvgPackingEfficiency over [driver] ++ executors of n_apps finished results whose placements are NODE indices
// (efficiency.go:114-156); d_avg_out: n_apps x 4 doubles {CPU, Memory, GPU, Max}.
hipError_t launch_avg_efficiency(bool reserve_execs, const NodeTable& table, const EffTables& eff, uint32_t* d_cnt,
                                 uint32_t n_cnt_waves, uint32_t n_apps, const gf_app* d_apps,
                                 const gf_result* d_results, const uint32_t* d_exec_nodes, double* d_avg_out,
                                 hipStream_t stream);

// ComputePackingEfficiencies (efficiency.go:66-103) for ONE result: per-node {cpu, memory, gpu} efficiencies in the
// caller's node-index space (eff tables indexed by node), d_reserved: n_nodes x 3 int64 scratch (zeroed here).
hipError_t launch_node_efficiencies(bool reserve_execs, const EffTables& eff_by_node, uint32_t n_nodes, int32_t k,
                                    const gf_app* d_app, const gf_result* d_result, const uint32_t* d_exec_nodes,
                                    int64_t* d_reserved, double* d_eff_out, hipStream_t stream);

// Node-range sharding of an independent batch (gangfit_shard.inc): a GPU owns one or more ranges
// [c_lo, c_hi) of 64-slot chunks of the merged slot order — one per shard it hosts (one shard per device on a real box; a
// repeated device id, the one-GPU stand-in, makes one device host several: its kernels then run ONE launch per step with a
// grid row per hosted shard, one app upload, one gathered table and one placement buffer per device).
constexpr uint32_t kMaxGroupDevices = 16;
struct ShardRange {
    uint32_t c_lo, c_hi;
    uint32_t shard, n_shards;
    uint32_t g_lo, g_hi;  // the range's sub-slots of the sparse gpu view (SparseTable; g_lo == g_hi: none / no view)
};
struct ShardSet {  // the shards one device hosts (grid row q of a step's launch handles entry q)
    uint32_t n, n_shards;
    uint32_t c_lo[kMaxGroupDevices], c_hi[kMaxGroupDevices], shard[kMaxGroupDevices];
    uint32_t g_lo[kMaxGroupDevices], g_hi[kMaxGroupDevices];
};
inline ShardSet shard_set_of(const ShardRange& r) {
    ShardSet s{};
    s.n = 1;
    s.n_shards = r.n_shards;
    s.c_lo[0] = r.c_lo;
    s.c_hi[0] = r.c_hi;
    s.shard[0] = r.shard;
    s.g_lo[0] = r.g_lo;
    s.g_hi[0] = r.g_hi;
    return s;
}
// Exchanges of the in-process multi-device context: up to 16 peer pointers by value.  The producing kernels write their
// 16-byte records straight into row `shard` of EVERY device's gathered table (posted stores over xGMI; dsts.n == 0: into the
// local array d_out instead, row = grid row — the one-process-per-GPU path and the RCCL exchange gather them afterwards).
struct PeerPtrs {
    void* p[kMaxGroupDevices];
    uint32_t n;
};
// gpu_view (n_x == 0: none): gangs whose executors need a gpu are summed / emitted from the range's part of the compact table
hipError_t launch_shard_partials(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, const ShardSet& set,
                                 uint32_t n_apps, const gf_app* d_apps, gf_shard_partial* d_out, const PeerPtrs& dsts,
                                 hipStream_t stream);
hipError_t launch_shard_drivers(const NodeTable& table, const ShardSet& set, uint32_t n_apps, const gf_app* d_apps,
                                const gf_shard_partial* d_all_partials, gf_shard_driver* d_out, const PeerPtrs& dsts,
                                hipStream_t stream);
// (zeroes d_exec2 first; every hosted shard writes its slice of the ONE buffer, row 0 also the results)
hipError_t launch_shard_emit(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, const ShardSet& set,
                             uint32_t n_apps, const gf_app* d_apps, const gf_shard_partial* d_all_partials,
                             const gf_shard_driver* d_all_drivers, gf_result* d_results, uint32_t* d_exec2, uint64_t half,
                             hipStream_t stream);
hipError_t launch_shard_finish(gf_algo algo, uint32_t n_shards, uint32_t n_apps, const gf_app* d_apps,
                               const gf_shard_partial* d_all_partials, const gf_shard_driver* d_all_drivers,
                               const gf_result* d_results, uint32_t* d_exec2, uint64_t half, hipStream_t stream);
// d_dst[i] += sum over srcs of src[i], n uint32 entries (only the first device finishes a batch: the all-reduce is a reduce)
hipError_t launch_shard_reduce_pull(const PeerPtrs& srcs, uint32_t* d_dst, size_t n, hipStream_t stream);

// Placing one executor per request (gangfit_executor.inc): first fit or the minimal-fragmentation choice.
// d_reserved: 3 x n_nodes int64 by node index (row-major, nullable); d_hosts: per request a bit set over node indices.
hipError_t launch_executor_fit(bool minimal_fragmentation, const NodeTable& table, const int64_t* d_reserved, uint32_t n_req,
                               const int64_t* d_exe, const uint32_t* d_hosts, uint32_t hosts_stride, const uint32_t* d_node_zone,
                               const uint32_t* d_req_zone, uint32_t* d_node_out, hipStream_t stream);

// findNodes of the failover reconciler (gangfit_findnodes.inc): n_req requests against the executor order; chained = one
// wavefront walks them in order on `table` (the mutable working copy) and subtracts each request's `reserved` map.
// d_adds_out: n_req x n_nodes (zeroed by the caller) or nullptr.
hipError_t launch_find_nodes(bool chained, const NodeTable& table, uint32_t n_req, const int64_t* d_exe, const int32_t* d_k,
                             const uint64_t* d_exec_off, gf_find_result* d_results, uint32_t* d_exec_nodes,
                             uint32_t* d_adds_out, hipStream_t stream);

// Snapshot construction on the device (gangfit_snapshot.hip): reservation replay, available / schedulable columns,
// zone order, node priority order.  All pointers are device buffers owned by the host layer; columns are SoA
// (cpu | memory | gpu, n_nodes each).  On return (stream order) d_avail / d_sched hold the columns and d_perm_b the
// node indices in priority order.
struct SnapshotBuild {
    uint32_t n_nodes, n_res, n_zones;
    bool usage_resident = false;  // d_usage holds the sums already (gf_usage_apply): neither zeroed nor scattered into
    const int64_t* d_alloc;      // 3 * n_nodes
    const int64_t* d_overhead;   // 3 * n_nodes or nullptr
    const uint32_t* d_res_node;  // n_res
    const int64_t* d_res_req;    // 3 * n_res (cpu | memory | gpu)
    const uint32_t* d_zone;      // n_nodes
    const uint32_t* d_name_rank; // n_nodes, a permutation
    int64_t* d_usage;            // 3 * n_nodes
    int64_t* d_avail;            // 3 * n_nodes
    int64_t* d_sched;            // 3 * n_nodes
    int64_t* d_zone_sum;         // 3 * n_zones (memory, cpu interleaved | population)
    uint32_t* d_zone_order;      // n_zones
    uint32_t* d_zone_rank;       // n_zones
    uint32_t* d_perm_a;          // n_nodes
    uint32_t* d_perm_b;          // n_nodes
    int64_t* d_keys_a;           // n_nodes
    int64_t* d_keys_b;           // n_nodes
    int64_t* d_keys_c;           // n_nodes
    uint32_t* d_perm_c;          // n_nodes
    uint32_t* d_sort_work;       // snapshot_sort_work_words() uint32 (8-byte aligned): count tables, barrier, scalars
    int sort_fault = 0;          // fault injection (tests): 1 = one workgroup of the sort never arrives at the grid barrier and
                                 // the others give up after 2^12 instead of 2^24 probes -> the error word is set
    // what launch_snapshot_finalize accumulates into, cleared by the build's one clearing launch (nullptr: no finalize follows):
    uint32_t* d_zfirst = nullptr;  // n_zones words, set to all ones
    uint32_t* d_zhasx = nullptr;   // the range d_zhasx | d_zeval | d_scalars[0 .. 16), set to zero ...
    size_t zhasx_to_scalars_words = 0;  // ... this many words
};
// usage[node] += sign * entry for n_entries reservation entries (columns cpu | memory | gpu of d_req); entries on nodes
// >= n_nodes are ignored.
// d_negative (nullable): after a removal (sign < 0) bit 0 is set when a touched node's sum went below zero.
hipError_t launch_usage_apply(uint32_t n_entries, uint32_t n_nodes, const uint32_t* d_node, const int64_t* d_req, int sign,
                              int64_t* d_usage, uint32_t* d_negative, hipStream_t stream);
// The slot tables of the merged layout built from the device-resident snapshot columns (what gf_orders_set builds on the
// host): every node gets the slot of its position in the priority order.
struct SnapshotFinalize {
    uint32_t n_nodes, n_slots, n_chunks, n_zones;
    const int64_t* d_avail;     // 3 * n_nodes, by node
    const int64_t* d_sched;     // 3 * n_nodes, by node
    const uint32_t* d_perm;     // n_nodes: priority order -> node
    const uint32_t* d_zone;     // n_nodes
    const uint32_t* d_flags;    // n_nodes: GF_NODE_*
    int64_t* d_snap;            // 3 * n_slots
    int64_t* d_sched_slot;      // 3 * n_slots
    uint32_t* d_slot_node;      // n_slots
    uint32_t* d_node_slot;      // n_nodes
    uint32_t* d_dslot;          // n_slots (identity)
    uint64_t* d_masks;          // 2 * n_chunks: executor | driver candidate bits
    int64_t* d_cmax;            // 3 * n_chunks
    int64_t* d_node_tab;        // 6 * n_nodes
    unsigned long long* d_gcd_part;  // 6 * n_chunks: chunk gcds, then chunk maxima of the magnitudes
    long long* d_units;         // 3
    uint32_t* d_zfirst;         // n_zones
    uint32_t* d_zhasx;          // n_zones
    uint32_t* d_zeval;          // n_zones: zone -> index in the evaluation list, GF_NO_NODE = not evaluated
    uint32_t* d_scalars;        // 16 words: [0] = zones in the evaluation list, [1] = no narrow form, [2] = negative schedulable
                                // value, [3] = the sort's error word, [4 .. 16) = d_units as pairs of words (one read-back)
    uint32_t* h_out = nullptr;  // nullable: the device's address of sixteen pinned host words, ZEROED BY THE HOST before the launch, that
                                // receive d_scalars as the kernels produce it (no read-back copy on the stream)
    uint64_t* d_zmasks;         // 2 * n_zones * n_chunks: executor rows, then (from row n_zones) driver rows
    int32_t* d_nsnap;           // 3 * n_slots
    int32_t* d_ncmax;           // 3 * n_chunks
};
hipError_t launch_snapshot_finalize(const SnapshotFinalize& f, const uint32_t* d_sort_error, hipStream_t stream);
struct CopyOut {  // three ranges of 32-bit words, any of them empty
    const uint32_t* src[3];
    uint32_t* dst[3];
    size_t words[3];
};
hipError_t launch_copy_out(const CopyOut& c, hipStream_t stream);
hipError_t launch_stream_copy(const void* src, void* dst, size_t bytes, hipStream_t stream);
hipError_t launch_stream_read(const void* src, size_t bytes, uint32_t* sink, hipStream_t stream);
hipError_t launch_empty(uint32_t* sink, hipStream_t stream);
size_t snapshot_sort_work_words();     // uint32 words of SnapshotBuild::d_sort_work
uint32_t snapshot_sort_error_word();   // index of the word that is non-zero when the sort's grid barrier gave up
hipError_t launch_snapshot_build(const SnapshotBuild& b, hipStream_t stream);

// Device self-test of the wave primitives (DPP scan, exact clamped division) against plain reference code.
// Writes the number of mismatching lanes/cases to *d_mismatch.
hipError_t launch_selftest(uint64_t seed, uint32_t n_cases, uint32_t* d_mismatch, hipStream_t stream);

}  // namespace gangfit
