// gangfit_ctx.h — internal to libgangfit's host layer (the translation units behind include/gangfit.h): the context object, the
// buffer helpers, the entry-point macros and the few functions one translation unit offers to the others.  Not installed, not
// part of the ABI.
//   gangfit_api.cpp           context life cycle, options, probes, recorded sequences, timers / counters / self-test
//   gangfit_api_snapshot.cpp  gf_snapshot_set / zones / orders, the resident cluster + usage, gf_snapshot_build*
//   gangfit_api_fit.cpp       the launches of every packer, the incremental chain cache, gf_fit_batch*, single executors,
//                             findNodes, efficiencies
//   gangfit_api_worker.cpp    the resident worker of the independent batch (gf_worker_*)
//   gangfit_api_group.cpp     node-range sharding: the gf_shard_* steps and the multi-device context (peer stores or RCCL)
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "gangfit.h"
#include "gangfit_device.h"

using gangfit::NodeTable;
using gangfit::ScanStats;

namespace gfapi {


constexpr int64_t kSentinelAvail = -(INT64_C(1) << 62);  // "node is not in nodesSchedulingMetadata"

// Completion waits.  hipStreamSynchronize / hipEventSynchronize park the calling thread and pay an interrupt + wake-up
// (tens of microseconds) per call — more than a whole 1 000-application batch takes on the device, and a visible part of
// every Filter.  The entry points of this library are short blocking calls, so they poll instead (hipStreamQuery /
// hipEventQuery, sub-microsecond per probe) and only fall back to the blocking wait when the device takes long (50 ms) or
// when GANGFIT_WAIT=block asks for it (a host that cannot spare the core for the duration of a call).
inline bool wait_blocking() {
    static const bool block = [] {
        const char* e = std::getenv("GANGFIT_WAIT");
        return e != nullptr && std::strcmp(e, "block") == 0;
    }();
    return block;
}
template <class Query, class Block>
inline hipError_t poll_then_block(Query query, Block block) {
    if (wait_blocking()) return block();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0;; ++i) {
        const hipError_t e = query();
        if (e != hipErrorNotReady) return e;
        __builtin_ia32_pause();
        if ((i & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            (void)hipGetLastError();
            return block();
        }
    }
}
inline hipError_t gf_wait_stream(hipStream_t st) {
    const hipError_t e = poll_then_block([st] { return hipStreamQuery(st); }, [st] { return hipStreamSynchronize(st); });
    if (e == hipSuccess) (void)hipGetLastError();  // hipErrorNotReady of the probes is not an error
    return e;
}
inline hipError_t gf_wait_event(hipEvent_t ev) {
    const hipError_t e = poll_then_block([ev] { return hipEventQuery(ev); }, [ev] { return hipEventSynchronize(ev); });
    if (e == hipSuccess) (void)hipGetLastError();
    return e;
}

template <typename T>
struct DeviceBuf {
    T* ptr = nullptr;
    size_t cap = 0;  // elements
    bool fine = false;  // fine-grained (device-coherent) memory: buffers other devices store into / read from
    bool borrowed = false;  // a view's alias of its parent's buffer (gf_ctx_view): never grown, never freed here
    void alias(const DeviceBuf& o) {
        if (!borrowed) release();
        ptr = o.ptr;
        cap = o.cap;
        borrowed = true;
    }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (borrowed) return hipErrorInvalidValue;
        size_t want = cap ? cap : 256;
        while (want < n) want *= 2;
        T* fresh = nullptr;
        hipError_t e = fine ? hipExtMallocWithFlags(reinterpret_cast<void**>(&fresh), want * sizeof(T), hipDeviceMallocFinegrained)
                            : hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(T));
        if (e != hipSuccess) return e;
        if (ptr) (void)hipFree(ptr);
        ptr = fresh;
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (ptr && !borrowed) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        borrowed = false;
    }
};

template <typename T>
struct PinnedBuf {
    T* ptr = nullptr;
    T* dev = nullptr;  // the device's address of the same memory (nullptr: not mapped); looked up once per allocation
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        size_t want = cap ? cap : 256;
        while (want < n) want *= 2;
        T* fresh = nullptr;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (ptr) (void)hipHostFree(ptr);
        ptr = fresh;
        cap = want;
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, fresh, 0) == hipSuccess) {
            dev = static_cast<T*>(d);
        } else {
            (void)hipGetLastError();
            dev = nullptr;
        }
        return hipSuccess;
    }
    void release() {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        dev = nullptr;
        cap = 0;
    }
};

// The collective library, bound at run time (a host without librccl still loads libgangfit): the in-process exchange of a
// multi-device context can run on RCCL (ncclCommInitAll: one communicator per device of THIS process, collectives grouped
// per step) instead of the peer stores of gangfit_shard.inc.  Only the handful of entry points used; constants as in rccl.h.
struct Rccl {
    typedef void* comm_t;
    void* lib = nullptr;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    static constexpr int kChar = 0, kUint32 = 3, kSum = 0;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        Reduce = reinterpret_cast<decltype(Reduce)>(dlsym(lib, "ncclReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (CommInitAll && CommDestroy && AllGather && Reduce && GroupStart && GroupEnd) return true;
        dlclose(lib);
        lib = nullptr;
        return false;
    }
};

inline Rccl& rccl() {
    static Rccl r;
    return r;
}

}  // namespace gfapi

using gfapi::DeviceBuf;
using gfapi::PinnedBuf;

namespace gfapi {
// The submitting threads of a multi-device context (gangfit_api_group.cpp): a sharded batch is ~6 runtime calls per device and
// three points where every device's stream must wait for every other device's event.  One host thread issuing all of it is
// ~50 serial runtime calls per batch on an 8-GPU box; here device d's calls are issued by thread d (the caller's thread is
// device 0's), all of them meeting at host-side barriers between the steps — an event must have been RECORDED before another
// stream can be told to wait for it.  Threads park on a condition variable between batches after a short spin.
struct GroupPool {
    explicit GroupPool(uint32_t n_devices);
    ~GroupPool();
    void run(const std::function<void(uint32_t)>& job);  // job(d) on every device's thread; returns when all have returned
    void barrier();                                      // called from inside job by EVERY device, the same number of times
    uint32_t parties;
private:
    void worker(uint32_t d);
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<uint64_t> generation{0};
    std::atomic<uint32_t> remaining{0};
    std::atomic<uint32_t> bar_count{0}, bar_gen{0};
    const std::function<void(uint32_t)>* job = nullptr;
    bool quit = false;
};
}  // namespace gfapi

struct gf_ctx {
    std::recursive_mutex mu;  // recursive: gf_snapshot_build installs its result through the public setters
    std::mutex seq_m;         // gf_ctx_lock / gf_ctx_unlock: a flag, not a held mutex, so any thread may release it
    std::condition_variable seq_cv;
    bool seq_held = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // The resident worker of the independent batch (gf_worker_*; gangfit_worker.inc).
    struct Worker {
        bool allocated = false;
        bool running = false;       // a launch is (or may still be) on the device
        int algo = -1;
        uint64_t epoch = 0;         // snapshot the launch's table arguments belong to
        hipStream_t stream = nullptr;
        gangfit::WorkerHostCtl* h = nullptr;  // pinned, coherent, device-mapped
        gangfit::WorkerHostCtl* h_dev = nullptr;
        gangfit::WorkerDevCtl* d = nullptr;   // device memory
        DeviceBuf<uint32_t> scratch;
        uint64_t scratch_stride = 0;
        uint64_t posted = 0;          // tickets posted so far (the host's copy of the doorbell)
        uint64_t completed_upto = 0;  // every ticket below this one is known complete
        uint32_t sets = 0;            // option "worker_sets": 0 = chosen per launch (worker_launch)
        uint32_t blocks_per_set = 0;  // option "worker_blocks_per_set" (x 16 wavefronts): 0 = chosen per launch
        uint32_t cur_sets = 0, cur_blocks_per_set = 0;  // what the last launch ran with (gf_worker_geometry)
        uint32_t hint_apps = 1000, hint_per_wave = 3;   // what the caller that launches expects: ticket size, applications per wavefront
        uint32_t idle_us = 200;
        uint64_t leave_after = 0;     // the next launch serves tickets below this one and leaves (GF_WORKER_LEAVE_AFTER), 0 = resident
        uint64_t launches = 0;
        int per_cu[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // resident workgroups of the worker kernel per CU, by packer (asked of the runtime once)
        // HIP events on the worker's stream around its launch: how long the last finished launch stayed on the device and how
        // many tickets it served (gf_worker_kernel_time: the per-ticket kernel time of bench.py's roofline)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        uint64_t launch_first = 0;     // first ticket of the launch on the device (or of the last one)
        float last_ms = 0.0f;          // duration of the last finished launch
        bool elapsed_pending = false;  // ... still to be read from the events (worker_elapsed)
        uint64_t last_tickets = 0;     // tickets it relayed
        // staging of gf_worker_fit: one pinned (coherent, device-mapped) slice per ring slot
        void* stage = nullptr;
        void* stage_dev = nullptr;
        size_t stage_apps = 0, stage_k = 0;  // capacity per slot
    } worker;
    hipStream_t timer_stream = nullptr;
    std::string err;
    gf_device_info info{};

    // host copy of the snapshot (node-index order)
    uint32_t n_nodes = 0;
    std::vector<int64_t> avail[3];
    std::vector<int64_t> sched[3];
    bool have_snapshot = false, have_sched = false, have_orders = false;

    // slot-ordered device tables
    uint32_t n_x = 0, n_d = 0, n_slots = 0;
    DeviceBuf<int64_t> d_snap;   // 3 * n_slots: cpu | mem | gpu of the snapshot
    DeviceBuf<int64_t> d_work;   // working copy mutated by FIFO chains
    DeviceBuf<uint32_t> d_slot_node, d_dslot, d_node_slot;
    DeviceBuf<int64_t> d_cmax;   // chunk-maxima index, 3 * n_chunks
    PinnedBuf<int64_t> h_cmax;
    uint32_t n_chunks = 0;
    std::vector<uint32_t> h_node_slot;  // kept for gf_residual_get
    PinnedBuf<int64_t> h_table;
    PinnedBuf<uint32_t> h_index;
    bool work_valid = false;
    bool d_identity = false;
    bool merged = false;       // slot space is the merged order (see NodeTable)
    uint32_t shard = 0, n_shards = 1;  // node-range sharding (gf_shard_set)
    DeviceBuf<uint64_t> d_masks;  // xmask | dmask, n_chunks each
    // sparse gpu view of the executor order (gangfit::SparseTable): compact table | node ids + slot map | maxima | masks
    DeviceBuf<int64_t> d_gtab, d_gcmax;
    DeviceBuf<uint32_t> d_gidx;   // slot_node of the sub-slots (n_gpad), then sub_of_slot (n_slots)
    DeviceBuf<uint64_t> d_gmask;
    PinnedBuf<int64_t> h_gtab;
    PinnedBuf<uint32_t> h_gidx;
    uint32_t n_g = 0, n_gpad = 0;  // sub-slots, padded to whole chunks; 0 = no view
    std::vector<uint32_t> g_prefix;  // [chunk of the full order] sub-slots before it: where a node-range shard's part of the view begins
    bool sparse_gpu = true;        // option "sparse_gpu" = 0 disables the view
    bool zero_copy = true;         // option "zero_copy" = 0: gf_fit_batch always stages through device buffers
    bool feasible_announce = true; // option "feasible_announce" = 0: gf_fit_feasible waits for the stream instead of watching its answers arrive
    bool zoned_fused = true;       // option "zoned_fused" = 0: independent batches of the zone-aware packers take the four-kernel path
    double call_phase_us[5] = {0, 0, 0, 0, 0};  // last gf_fit_batch on the zero-copy path: stage | launch | wait | copy out | total
    PinnedBuf<uint64_t> h_masks;
    DeviceBuf<gangfit::NApp> d_napps;       // FIFO chain: app records in the narrow domain (chain_prologue_kernel)
    DeviceBuf<int32_t> d_wide_needed;       // two words used alternately: set by the chain prologue when a request has no narrow
                                            // form; each prologue zeroes the word the NEXT chain will use (wide_flag)
    uint32_t wide_seq = 0;                  // chains launched: parity picks the word
    bool wide_dirty = false;                // a launch failed half way: both words are cleared before the next chain
    struct HostIo {  // set by gf_fit_batch around launch(): where the first / last kernel of a chain may read and write directly
        bool active = false;
        uint32_t n_apps = 0;           // records of the whole queue in h_apps
        const gf_app* apps = nullptr;  // device addresses of the pinned h_apps / h_results / h_exec / h_failed
        gf_result* results = nullptr;
        uint32_t* exec = nullptr;
        int32_t* failed = nullptr;
        bool apps_done = false;  // a kernel of this launch writes (or a copy wrote) the records to d_apps
        bool out_done = false;   // the last kernel of this launch writes the answers to the host buffers
    } hio;
    DeviceBuf<int32_t> d_capmat;            // minimal-fragmentation chain: capacity per (request shape, slot)
    bool fifo_minfrag_matrix = true;        // option "minfrag_matrix" = 0 recomputes capacities in every pass
    DeviceBuf<int32_t> d_mfhist;            // ... and the capacity histograms per (candidate view, request shape)
    bool fifo_minfrag_hist = true;          // option "minfrag_hist" = 0: block-cooperative passes instead of the histogram path
    // narrow (scaled int32) form of the table: value = scaled * unit[dim]; exists when every |value / unit| < 2^30
    bool narrow_ok = false;
    int64_t unit[3] = {1, 1, 1};
    int64_t nmax[3] = {0, 0, 0};  // largest |scaled value| per dimension: how far the units may still be refined per batch
    DeviceBuf<int32_t> d_nsnap, d_nwork, d_ncmax, d_ncmax_w;
    PinnedBuf<int32_t> h_ntable;
    bool fifo_generic = false;  // option "fifo_generic": chains run on the wide / generic global-memory kernels only
    bool force_general_layout = false;  // option "force_general_layout": gf_orders_set never merges the two orders
    uint32_t lds_budget = 0;   // bytes of LDS one workgroup may use

    // zone views + efficiency tables (single-AZ packers, LIB/binpack/single_az.go; efficiency.go)
    std::vector<uint32_t> zone;        // per node; empty = one zone
    DeviceBuf<int64_t> d_sched;        // 3 * n_slots SchedulableResources in slot order (0 on empty slots)
    DeviceBuf<int64_t> d_node_tab;     // 6 * n_nodes: avail cpu|mem|gpu, sched cpu|mem|gpu by node index
    DeviceBuf<uint64_t> d_zmasks;      // [2][n_zones][zstride]: executor masks, then driver masks
    PinnedBuf<uint64_t> h_zmasks;
    uint32_t n_zones = 0, zstride = 0;
    uint32_t zd_row0 = 0;              // row of d_zmasks where the driver masks start (n_zones, or the zone count of a device build)
    bool host_stale = false;           // the host mirrors (avail / sched / h_node_slot) still sit on the device (gf_snapshot_build)
    // every node's available quantities lie at or below its schedulable ones in the snapshot on the device (gf_snapshot_set checked the
    // host arrays; a snapshot built on the device leaves it false): then every term of an average packing efficiency is >= 0
    bool eff_nonneg = false;
    bool snapshot_finalize_on_device = true;  // option "snapshot_finalize_host" = 1 builds the slot tables through gf_orders_set
    int sort_fault = 0;                       // option "sort_fault" (tests): the priority sort's grid barrier cannot complete
    DeviceBuf<gf_result> d_zres;
    DeviceBuf<uint32_t> d_zexec;
    DeviceBuf<double> d_zavg, d_avg;
    DeviceBuf<uint32_t> d_cnt;         // [cnt_rows][cnt_slots], all-zero between launches
    uint32_t cnt_rows = 0, cnt_slots = 0;
    DeviceBuf<int64_t> d_reserved;
    DeviceBuf<double> d_eff;
    PinnedBuf<double> h_avg;

    // gf_cluster_set: the static columns of gf_snapshot_build, resident
    DeviceBuf<int64_t> d_cl_i64;   // allocatable (3n) | overhead (3n)
    DeviceBuf<int64_t> d_cl_usage;  // resident UsageForNodes sums (3n), maintained by gf_usage_apply
    DeviceBuf<int64_t> d_delta_i64; // one gf_usage_apply call's entries
    DeviceBuf<uint32_t> d_delta_u32;
    __int128 usage_total[3] = {0, 0, 0};  // sum of everything applied: bounds every node's sum
    DeviceBuf<uint32_t> d_cl_u32;  // zone | name_rank | node_flags (n each)
    std::vector<uint32_t> cl_flags, cl_zone;  // host copies (candidate lists, ctx->zone); cl_flags = the flags of the last build
    std::vector<uint32_t> cl_default_flags;   // the flags of gf_cluster_set: what node_flags == NULL selects
    bool d_flags_default = true;              // the device column holds cl_default_flags (not a request's candidate flags)
    bool usage_ok = true;                     // false after a failed update: the resident sums are unknown until gf_usage_reset
    uint64_t cluster_gen = 0, usage_gen = 0;  // bumped by gf_cluster_set / gf_usage_reset + gf_usage_apply (gf_generation)
    DeviceBuf<uint32_t> d_flag32;             // one device word for yes / no answers of small kernels
    DeviceBuf<uint32_t> d_sortwork;           // count tables, grid barrier and scalars of the priority sort (gangfit_snapshot.hip)
    uint32_t cl_n = 0, cl_zones = 1;
    bool cl_over = false, have_cluster = false;
    int64_t cl_max_over[3] = {0, 0, 0};

    // gf_snapshot_build
    DeviceBuf<int64_t> d_bi64;   // alloc | overhead | usage | avail | sched (3n each) | keys_a | keys_b (n each) | res_req (3r) | zone_sum
    DeviceBuf<uint32_t> d_bu32;  // zone | name_rank | perm_a | perm_b (n each) | res_node (r) | zone_order | zone_rank
    PinnedBuf<int64_t> h_bcols;  // avail | sched (3n each)
    PinnedBuf<uint32_t> h_border;

    // single-executor requests (gf_executor_fit)
    DeviceBuf<int64_t> d_xexe, d_xreserved;
    DeviceBuf<uint32_t> d_xhosts, d_xout, d_xnzone, d_xqzone;

    // ---- multi-device context (gf_init with n_dev > 1): this object only routes; one sub-context per device id does the
    //      work and owns shard `shard` of `n_shards` of the priority order.  The g_* members live in the sub-contexts.
    std::vector<gf_ctx*> group;               // one sub-context per DEVICE (per listed id with GANGFIT_TEST_GROUP_SPLIT=1)
    std::vector<uint32_t> my_shards;          // (in a sub-context) the shards — ranges of the priority order — this device scans
    uint32_t g_total_shards = 0;              // (in the routing object) the ids gf_init was given = shards of the order
    gfapi::GroupPool* g_pool = nullptr;       // (in the routing object) one submitting thread per device but the first
    DeviceBuf<gf_shard_partial> g_part_loc, g_part_all;  // this shard's records | [n_shards][n_apps] gathered
    DeviceBuf<gf_shard_driver> g_drv_loc, g_drv_all;
    DeviceBuf<uint32_t> g_exec2;                         // 2 * half: placements (node + 1) | capacities
    hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};    // behind partials+push | drivers+push | emit
    // ... and these in the routing object
    uint64_t g_verified_epoch = 0;  // snapshot epoch whose first sharded batch agreed with the first device's own answer
    bool g_verify = true;           // option "group_verify"
    bool g_shard_off = false;       // a sharded batch disagreed: every batch is served by the first device from then on
    int g_fault = 0;                // option "group_fault" (tests): 1 = the placement reduction is skipped, 2 = zeroed capacity sums
    std::vector<void*> g_comms;     // option "group_exchange" = 1: one RCCL communicator per sub-context (ncclCommInitAll)
    std::vector<int> g_devices;     // the device ids gf_init was given

    // findNodes requests (gf_find_nodes)
    DeviceBuf<int32_t> d_fk;
    DeviceBuf<uint64_t> d_foff;
    DeviceBuf<gf_find_result> d_fres;
    DeviceBuf<uint32_t> d_fadds;
    PinnedBuf<uint64_t> h_foff;

    // batch buffers
    DeviceBuf<gf_app> d_apps;
    DeviceBuf<gf_result> d_results;
    DeviceBuf<uint32_t> d_exec, d_scratch;
    // gf_fit_feasible's own placement / scratch / per-view buffers: the call may return while its kernel is still storing
    // placements (the answers announce themselves), so nothing another entry point launches — on a caller's stream, or with a
    // null-stream copy — may share them
    DeviceBuf<uint32_t> d_feas_exec, d_feas_scratch, d_feas_zexec;
    DeviceBuf<int32_t> d_failed;
    DeviceBuf<ScanStats> d_stats;
    PinnedBuf<gf_app> h_apps;
    PinnedBuf<gf_result> h_results;
    PinnedBuf<uint32_t> h_exec;
    PinnedBuf<int32_t> h_failed;
    PinnedBuf<uint8_t> h_feasible;  // gf_fit_feasible: one HasCapacity byte per application, written by the kernel
    bool feasible_sync_dirty = false;     // a call failed: the collection words are cleared again before the next launch
    DeviceBuf<uint32_t> d_feasible_sync;  // ... the bytes as the wavefronts leave them in device memory (zero between launches)
    bool stats_on = false;

    // ---- views (gf_ctx_view): contexts that fit on THIS context's installed snapshot with buffers and a stream of their own.
    //      A view aliases the read-only tables of the snapshot; installs on the parent wait for the views' calls in flight
    //      (views_mu: shared by a view's call, exclusive by an install), and a view re-aliases when the epoch has moved on.
    gf_ctx* view_of = nullptr;
    uint64_t view_epoch = 0;     // parent snap_epoch the aliases were taken at
    std::shared_mutex views_mu;  // (in the parent)
    int install_depth = 0;       // (in the parent, under mu) nested installs take views_mu once
    int n_views = 0;             // (in the parent, under mu) live views
    std::vector<gf_ctx*> views;  // (in the parent, under mu) the live views: an install waits for their streams

    // ---- incremental FIFO chains (gf_fit_batch, GF_MODE_FIFO_CHAIN).  The reference replays every earlier driver on every
    //      Filter (resource.go:309-328); with an unchanged snapshot driver j + 1's chain is driver j's chain plus one
    //      application.  The chain kernels therefore dump their working table every 2^shift applications (ChainCkpt), the
    //      host keeps the last chain's records, results and placements, and the next chain resumes from the last checkpoint
    //      inside the longest common prefix of the two queues.  Anything that installs a snapshot, zones or orders bumps
    //      snap_epoch and with it drops the cache.  Results are those of a full replay bit for bit: a checkpoint IS the
    //      table a replay would hold at that application.
    uint64_t snap_epoch = 1;
    bool chain_cache_on = true;  // GANGFIT_CHAIN_CACHE=0 / option "chain_cache" = 0: every chain replays from the snapshot
    struct ChainCache {
        bool valid = false;
        uint64_t epoch = 0;
        int algo = -1;
        int64_t unit[3] = {0, 0, 0};  // narrow units the checkpoints are scaled in
        uint32_t shift = 5;
        uint32_t n_apps = 0;
        uint32_t n_ckpt = 0;          // checkpoints 1 .. n_ckpt hold the table before application i << shift
        int32_t failed_at = -1;
        std::vector<gf_app> apps;     // the queue of the last chain (with exec_off)
        std::vector<gf_result> results;
        std::vector<uint32_t> exec;
        DeviceBuf<int32_t> d_ckpt;    // [n][slot_words]
        DeviceBuf<int32_t> d_tip;     // [slot_words] the table before application tip_at of the cached chain (ChainCkpt::tip)
        bool tip_valid = false;       // ... written by the cached chain (LDS chain kernel, whole table in LDS)
        uint32_t tip_at = 0;
        size_t slot_words = 0;        // chain_ckpt_stride of the snapshot the buffer was laid out for
        bool dirty_format = false;    // the checkpoints are DELTAS (the chunks touched since the previous checkpoint + a cumulative
                                      // and a delta mask; the solo kernel on a table with a global tail): restored by a kernel
                                      // that lays checkpoints 1 .. i over the snapshot instead of a copy
    } chain;
    uint64_t chain_stat[4] = {0, 0, 0, 0};  // chains | resumed chains | applications evaluated | applications skipped
    struct PlannedUnits {  // what chain_plan found for the call in progress: narrow_begin does not scan the queue again
        bool valid = false;
        int64_t eff[3] = {0, 0, 0};
        int32_t factor[3] = {1, 1, 1};
    } planned_units;
};

namespace gfapi {

int fail(gf_ctx* ctx, int code, const char* fmt, ...);

// An install on a context that has views: exclusive against the views' calls in flight (taken once per outermost install;
// ctx->mu is held, so the depth counter needs no further protection).
void worker_quiesce(gf_ctx* ctx);
struct InstallGuard {
    gf_ctx* c;
    explicit InstallGuard(gf_ctx* ctx) : c(ctx) {
        if (c->install_depth++ == 0) {
            worker_quiesce(c);  // the resident worker reads the installed tables: it leaves before they change
            c->views_mu.lock();
            // a view's asynchronous entry points (gf_fit_batch_dev, recorded graphs) return with kernels still reading the
            // aliased tables, which an install overwrites in place: wait for every view's stream, not only for its calls
            for (gf_ctx* v : c->views)
                if (v->stream != nullptr && hipSetDevice(v->device) == hipSuccess) (void)gf_wait_stream(v->stream);
        }
    }
    ~InstallGuard() {
        if (--c->install_depth == 0) c->views_mu.unlock();
    }
    InstallGuard(const InstallGuard&) = delete;
    InstallGuard& operator=(const InstallGuard&) = delete;
};

int view_refresh(gf_ctx* v);

// At the top of every entry point that READS the installed snapshot (after ctx->mu): a view holds its parent's views_mu
// shared for the whole call and re-aliases the parent's tables when a new snapshot has been installed since.
#define GF_VIEW_ENTER(ctx)                                                               \
    std::shared_lock<std::shared_mutex> view_lock__;                                     \
    if ((ctx)->view_of != nullptr) {                                                     \
        view_lock__ = std::shared_lock<std::shared_mutex>((ctx)->view_of->views_mu);     \
        if (const int vrc__ = view_refresh(ctx); vrc__ != GF_OK) return vrc__;           \
    }
#define GF_NOT_ON_A_VIEW(ctx) \
    if ((ctx)->view_of != nullptr) return fail((ctx), GF_ERR_STATE, "a view fits on its parent's snapshot: it does not install one")

#define GF_HIP(ctx, call)                                                                                    \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess) return fail((ctx), GF_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// Entry points that have no multi-device form run on the first sub-context of a group (gf_init with n_dev > 1).
#define GF_DELEGATE(ctx, expr)                                   \
    do {                                                         \
        if ((ctx) != nullptr && !(ctx)->group.empty()) {         \
            gf_ctx* const group__ = (ctx);                       \
            (ctx) = group__->group[0];                           \
            const int rc__ = (expr);                             \
            if (rc__ != GF_OK) group__->err = (ctx)->err;        \
            return rc__;                                         \
        }                                                        \
    } while (0)
// The same call on every sub-context (snapshot / zones / orders are replicated: each device scans only its range).
#define GF_EACH(ctx, expr)                                                 \
    do {                                                                   \
        if ((ctx) != nullptr && !(ctx)->group.empty()) {                   \
            gf_ctx* const group__ = (ctx);                                 \
            std::lock_guard<std::recursive_mutex> glock__(group__->mu);    \
            for (gf_ctx* sub__ : group__->group) {                         \
                (ctx) = sub__;                                             \
                const int rc__ = (expr);                                   \
                if (rc__ != GF_OK) {                                       \
                    group__->err = sub__->err;                             \
                    return rc__;                                           \
                }                                                          \
            }                                                              \
            return GF_OK;                                                  \
        }                                                                  \
    } while (0)

// ---- what the translation units offer each other
// gangfit_api_group.cpp
int group_fit_batch(gf_ctx* g, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* apps, gf_result* results,
                    uint32_t* exec_nodes, uint64_t exec_nodes_cap, int32_t* chain_failed_at);
// gangfit_api_fit.cpp
NodeTable make_table(gf_ctx* ctx, int64_t* base);
gangfit::SparseTable make_sparse(gf_ctx* ctx);
gangfit::EffTables slot_eff_tables(gf_ctx* ctx, const int64_t* avail_base);
bool reserves_executors(gf_algo algo);
bool is_zone_algo(gf_algo algo);
int ensure_cnt(gf_ctx* ctx, uint64_t n_decisions, hipStream_t stream);
// gangfit_api_snapshot.cpp
int materialize_host(gf_ctx* ctx);
// gangfit_api_worker.cpp: worker_quiesce (declared above, next to InstallGuard)

}  // namespace gfapi
