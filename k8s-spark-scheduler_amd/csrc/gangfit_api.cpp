// gangfit_api.cpp — the C ABI of libgangfit (include/gangfit.h): context, snapshot/order staging, launches.
//
// Host side only: builds the slot-ordered node table the kernels scan (gangfit_device.h), moves app records and
// results through pinned staging buffers and serialises callers per context.  No CPU fallback lives here: when the
// device path cannot serve a call the function returns < 0 and the caller (the Go shim) decides what to do.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
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

namespace {

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
Rccl& rccl() {
    static Rccl r;
    return r;
}

}  // namespace

struct gf_ctx {
    std::recursive_mutex mu;  // recursive: gf_snapshot_build installs its result through the public setters
    std::mutex seq_m;         // gf_ctx_lock / gf_ctx_unlock: a flag, not a held mutex, so any thread may release it
    std::condition_variable seq_cv;
    bool seq_held = false;
    int device = 0;
    hipStream_t stream = nullptr;
    bool stream_borrowed = false;  // a shard of a multi-device context on a device an earlier shard is on: it uses that one's stream
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
        uint32_t sets = 3;
        uint32_t blocks_per_set = 64;  // x 16 wavefronts
        uint32_t idle_us = 200;
        uint64_t launches = 0;
        // HIP events on the worker's stream around its launch: how long the last finished launch stayed on the device and how
        // many tickets it served (gf_worker_kernel_time: the per-ticket kernel time of bench.py's roofline)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        uint64_t launch_first = 0;     // first ticket of the launch on the device (or of the last one)
        float last_ms = 0.0f;          // duration of the last finished launch
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
    bool sparse_gpu = true;        // option "sparse_gpu" = 0 disables the view
    bool zero_copy = true;         // option "zero_copy" = 0: gf_fit_batch always stages through device buffers
    // a lone blocking independent batch announces its own completion in pinned memory (gangfit::IndHostOut): the caller polls a
    // word instead of waiting for the stream
    bool host_flag = true;         // option "host_flag" = 0: gf_fit_batch waits for the stream as before
    DeviceBuf<uint32_t> d_ind_done;             // arrival counters, all zero between launches
    PinnedBuf<unsigned long long> h_ind_flag;   // [0] = sequence number of the last batch that announced itself
    uint64_t ind_seq = 0;
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
    DeviceBuf<uint32_t> d_xhosts, d_xout;

    // ---- multi-device context (gf_init with n_dev > 1): this object only routes; one sub-context per device id does the
    //      work and owns shard `shard` of `n_shards` of the priority order.  The g_* members live in the sub-contexts.
    std::vector<gf_ctx*> group;
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
    DeviceBuf<int32_t> d_failed;
    DeviceBuf<ScanStats> d_stats;
    PinnedBuf<gf_app> h_apps;
    PinnedBuf<gf_result> h_results;
    PinnedBuf<uint32_t> h_exec;
    PinnedBuf<int32_t> h_failed;
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

namespace {

int fail(gf_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

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

int group_fit_batch(gf_ctx* g, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* apps, gf_result* results,
                    uint32_t* exec_nodes, uint64_t exec_nodes_cap, int32_t* chain_failed_at);

NodeTable make_table(gf_ctx* ctx, int64_t* base) {
    NodeTable t;
    t.cpu = base;
    t.mem = base + ctx->n_slots;
    t.gpu = base + 2 * (size_t)ctx->n_slots;
    t.slot_node = ctx->d_slot_node.ptr;
    t.dslot = ctx->d_dslot.ptr;
    t.node_slot = ctx->d_node_slot.ptr;
    t.cmax = ctx->d_cmax.ptr;
    t.n_chunks = ctx->n_chunks;
    t.n_x = ctx->n_x;
    t.n_d = ctx->n_d;
    t.n_slots = ctx->n_slots;
    t.n_nodes = ctx->n_nodes;
    t.d_identity = ctx->d_identity ? 1u : 0u;
    t.xmask = ctx->d_masks.ptr;
    t.dmask = ctx->d_masks.ptr + ctx->n_chunks;
    return t;
}

gangfit::SparseTable make_sparse(gf_ctx* ctx) {
    gangfit::SparseTable g{};
    if (ctx->n_g == 0) return g;
    g.cpu = ctx->d_gtab.ptr;
    g.mem = g.cpu + ctx->n_gpad;
    g.gpu = g.mem + ctx->n_gpad;
    g.slot_node = ctx->d_gidx.ptr;
    g.sub_of_slot = ctx->d_gidx.ptr + ctx->n_gpad;
    g.cmax = ctx->d_gcmax.ptr;
    g.xmask = ctx->d_gmask.ptr;
    g.n_x = ctx->n_g;
    g.n_chunks = ctx->n_gpad / 64;
    return g;
}

gangfit::EffTables slot_eff_tables(gf_ctx* ctx, const int64_t* avail_base) {
    gangfit::EffTables e;
    for (int j = 0; j < 3; ++j) {
        e.avail[j] = avail_base + (size_t)j * ctx->n_slots;
        e.sched[j] = ctx->d_sched.ptr + (size_t)j * ctx->n_slots;
    }
    return e;
}

// minimalFragmentation never records its placements in `reserved` (minimal_fragmentation.go:59-91)
bool reserves_executors(gf_algo algo) {
    return algo != GF_ALGO_MINIMAL_FRAGMENTATION && algo != GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION;
}
bool is_zone_algo(gf_algo algo) {
    return algo == GF_ALGO_AZ_AWARE_TIGHTLY_PACK || algo == GF_ALGO_SINGLE_AZ_TIGHTLY_PACK ||
           algo == GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION;
}

// Rows of the per-wave multiplicity scratch: enough waves to fill the chip, bounded to 256 MiB.
int ensure_cnt(gf_ctx* ctx, uint64_t n_decisions, hipStream_t stream) {
    uint64_t rows = n_decisions < 1024 ? n_decisions : 1024;
    const uint64_t cap = (UINT64_C(256) << 20) / (4 * (uint64_t)ctx->n_slots);
    if (rows > cap) rows = cap;
    if (rows < 1) rows = 1;
    if (rows <= ctx->cnt_rows && ctx->cnt_slots == ctx->n_slots) return GF_OK;
    if (rows < ctx->cnt_rows) rows = ctx->cnt_rows;
    GF_HIP(ctx, gf_wait_stream(stream));
    GF_HIP(ctx, ctx->d_cnt.reserve(rows * ctx->n_slots));
    GF_HIP(ctx, hipMemsetAsync(ctx->d_cnt.ptr, 0, rows * ctx->n_slots * sizeof(uint32_t), stream));
    ctx->cnt_rows = (uint32_t)rows;
    ctx->cnt_slots = ctx->n_slots;
    return GF_OK;
}


// The narrow (scaled int32) working table of one FIFO chain.  The table's units are the gcds of its own columns; a batch
// whose requests are finer than that (a 2 GiB driver on a cluster whose free memory happens to be a multiple of 4 GiB)
// would have no scaled form and fall to the wide kernels.  When the host sees the batch (h_apps; gf_fit_batch) the units are
// therefore refined to gcd(table unit, every request of the batch) and the working copy is multiplied up by the ratio —
// as long as every scaled magnitude stays below 2^30; comparisons, subtractions and floor divisions are invariant under a
// common factor, so the chain is bit-identical.  Device-resident batches (gf_fit_batch_dev) keep the table's units.
// *proven (nullable): every request of the batch is a multiple of the resulting units and fits the narrow range, i.e. the
// narrow kernel will not hand the batch to its wide twin (what prepare_app tests on the device).
void narrow_units(const gf_ctx* ctx, const gf_app* h_apps, uint32_t n_apps, int64_t eff[3], int32_t factor[3], bool* proven) {
    for (int j = 0; j < 3; ++j) {
        eff[j] = ctx->unit[j];
        factor[j] = 1;
    }
    if (proven) *proven = false;
    if (h_apps == nullptr) return;
    for (uint32_t a = 0; a < n_apps; ++a)
        for (int j = 0; j < 3; ++j)
            for (const int64_t v : {h_apps[a].drv[j], h_apps[a].exe[j]})
                if (v > 0 && v % eff[j] != 0) {
                    int64_t x = eff[j], y = v;
                    while (y) {
                        const int64_t t = x % y;
                        x = y;
                        y = t;
                    }
                    eff[j] = x;
                }
    bool ok = true;
    for (int j = 0; j < 3; ++j) {
        const int64_t f = ctx->unit[j] / eff[j];
        const int64_t room = ctx->nmax[j] > 0 ? ((INT64_C(1) << 30) - 1) / ctx->nmax[j] : (INT64_C(1) << 30) - 1;
        ok = ok && f <= room;
        factor[j] = ok ? (int32_t)f : 1;
    }
    if (!ok)
        for (int j = 0; j < 3; ++j) {
            eff[j] = ctx->unit[j];
            factor[j] = 1;
        }
    if (proven) {
        bool all = true;
        for (uint32_t a = 0; a < n_apps && all; ++a)
            for (int j = 0; j < 3; ++j)
                for (const int64_t v : {h_apps[a].drv[j], h_apps[a].exe[j]})
                    all = all && v >= 0 && v % eff[j] == 0 && v / eff[j] < (INT64_C(1) << 30);
        *proven = all;
    }
}

// restore (nullable): a checkpoint of an earlier chain in the SAME units — the working copy starts from it instead of the
// snapshot (incremental chains).
// The copies themselves are left to the chain's first kernel (io).
int narrow_begin(gf_ctx* ctx, const gf_app* h_apps, uint32_t n_apps, hipStream_t stream, gangfit::NarrowTable* nt,
                 gangfit::ChainIo* io, const int32_t* restore = nullptr, bool restore_dirty_chunks = false) {
    int64_t eff[3];
    int32_t factor[3];
    if (ctx->planned_units.valid) {
        for (int j = 0; j < 3; ++j) {
            eff[j] = ctx->planned_units.eff[j];
            factor[j] = ctx->planned_units.factor[j];
        }
    } else {
        narrow_units(ctx, h_apps, n_apps, eff, factor, nullptr);
    }
    nt->cpu = ctx->d_nwork.ptr;
    nt->mem = nt->cpu + ctx->n_slots;
    nt->gpu = nt->mem + ctx->n_slots;
    for (int j = 0; j < 3; ++j) nt->unit[j] = eff[j];
    const size_t table_bytes = 3 * (size_t)ctx->n_slots * sizeof(int32_t);
    const bool whole = restore != nullptr && !restore_dirty_chunks;  // the checkpoint is the whole table
    const int32_t* src = nullptr;  // what the working copy starts from, when a plain copy makes it
    if (factor[0] == 1 && factor[1] == 1 && factor[2] == 1) {
        src = whole ? restore : ctx->d_nsnap.ptr;
        nt->cmax = ctx->d_ncmax.ptr;
    } else {
        GF_HIP(ctx, ctx->d_ncmax_w.reserve(3 * (size_t)ctx->n_chunks));
        GF_HIP(ctx, gangfit::launch_narrow_rescale(ctx->d_nsnap.ptr, ctx->d_nwork.ptr, ctx->n_slots, ctx->d_ncmax.ptr,
                                                   ctx->d_ncmax_w.ptr, ctx->n_chunks, factor, stream));
        if (whole) src = restore;
        nt->cmax = ctx->d_ncmax_w.ptr;
    }
    if (src != nullptr) {
        io->copy_src[0] = reinterpret_cast<const uint32_t*>(src);
        io->copy_dst[0] = reinterpret_cast<uint32_t*>(ctx->d_nwork.ptr);
        io->copy_words[0] = table_bytes / sizeof(uint32_t);
    }
    // ... or only the chunks that differ from the snapshot (in the chain's units), laid over it
    if (restore != nullptr && restore_dirty_chunks) {
        io->overlay = ctx->chain.d_ckpt.ptr;  // checkpoints 1 .. count, the latest delta of a chunk wins (delta format)
        io->overlay_stride = ctx->chain.slot_words;
        io->overlay_count = (uint32_t)((size_t)(restore - ctx->chain.d_ckpt.ptr) / ctx->chain.slot_words) + 1u;
        io->overlay_dst = ctx->d_nwork.ptr;
        io->overlay_slots = ctx->n_slots;
        io->overlay_chunks = ctx->n_chunks;
    }
    return GF_OK;
}

// How one FIFO chain of gf_fit_batch uses the chain cache (decided by chain_plan before the launch).
struct ChainRun {
    uint32_t a_begin = 0;        // first application this launch evaluates (a multiple of 1 << shift); 0 = from the snapshot
    bool record = false;         // dump checkpoints into ctx->chain.d_ckpt
    bool narrow_proven = false;  // every request has a scaled form (checked on the host): the wide twin is not launched
    uint32_t common = 0;         // leading applications identical to the cached queue's (>= a_begin): the cache keeps them
};

// The checkpoint arguments of a chain kernel and the table it starts from.
gangfit::ChainCkpt chain_ckpt_args(gf_ctx* ctx, const ChainRun* run, const int32_t** restore) {
    gangfit::ChainCkpt ck{nullptr, run ? run->a_begin : 0u, ctx->chain.shift, ctx->chain.slot_words, nullptr};
    *restore = nullptr;
    if (run != nullptr && (run->record || run->a_begin > 0)) {
        ck.base = ctx->chain.d_ckpt.ptr;
        if (run->a_begin > 0) {
            *restore = ck.base + (size_t)((run->a_begin >> ck.shift) - 1u) * ctx->chain.slot_words;
            if (ctx->chain.dirty_format)
                ck.resume_mask = reinterpret_cast<const unsigned long long*>(*restore + 3 * (size_t)ctx->n_slots + (ctx->n_slots & 1u));
        }
    }
    return ck;
}

// The flag word of the chain being launched ("a request has no scaled form") and the ChainIo that goes with a launch on
// the records [a0, n_apps): the records come from the pinned host buffer when gf_fit_batch offered it, the answers go to
// the host buffers when the translate step is the last kernel to write them (answers_final).
int32_t* wide_flag(gf_ctx* ctx) { return ctx->d_wide_needed.ptr + (ctx->wide_seq & 1u); }
int chain_io_begin(gf_ctx* ctx, uint32_t a0, bool answers_final, hipStream_t stream, gangfit::ChainIo* io) {
    if (ctx->wide_dirty) {
        GF_HIP(ctx, hipMemsetAsync(ctx->d_wide_needed.ptr, 0, 2 * sizeof(int32_t), stream));
        ctx->wide_dirty = false;
    }
    io->wide_clear = ctx->d_wide_needed.ptr + ((ctx->wide_seq & 1u) ^ 1u);
    gf_ctx::HostIo& h = ctx->hio;
    if (h.active && !h.apps_done) io->apps_src = h.apps + a0;
    if (h.active && answers_final) {
        io->h_results = h.results + a0;
        io->h_exec = h.exec;
        io->h_failed = h.failed;
    }
    ctx->wide_dirty = true;  // until chain_io_end: a launch that fails half way leaves the flag words in an unknown state
    return GF_OK;
}
void chain_io_end(gf_ctx* ctx, const gangfit::ChainIo& io) {
    ctx->wide_dirty = false;
    ++ctx->wide_seq;
    if (io.apps_src != nullptr) ctx->hio.apps_done = true;
    if (io.h_results != nullptr) ctx->hio.out_done = true;
}
// Launch paths whose first kernel does not take the records from the host: an ordinary copy, once per gf_fit_batch.
int apps_to_device(gf_ctx* ctx, hipStream_t stream) {
    gf_ctx::HostIo& h = ctx->hio;
    if (!h.active || h.apps_done) return GF_OK;
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_apps.ptr, ctx->h_apps.ptr, (size_t)h.n_apps * sizeof(gf_app), hipMemcpyHostToDevice, stream));
    h.apps_done = true;
    return GF_OK;
}

// Table slots the solo chain kernel keeps in LDS (whole 64-slot chunk blocks of 784 bytes next to its fixed tables).
uint32_t solo_lds_slots(const gf_ctx* ctx) {
    const size_t fixed = gangfit::fifo_solo_lds_bytes(0, ctx->n_chunks);
    const size_t per_chunk = gangfit::fifo_solo_lds_bytes(64, ctx->n_chunks) - fixed;
    const size_t fit = ctx->lds_budget > fixed ? (ctx->lds_budget - fixed) / per_chunk : 0;
    const size_t whole = (ctx->n_slots + 63u) / 64u;
    return (uint32_t)((fit < whole ? fit : whole) * 64u);
}

// Geometry of the LDS-resident chains of the zone-aware tightly-pack packers (gangfit_fifo_zoned.inc) and of the
// minimal-fragmentation packers (gangfit_fifo_minfrag.inc); false = the generic global-memory chain serves.
bool zoned_lds_geometry(const gf_ctx* ctx, bool az_aware, uint32_t* n_shapes, uint32_t* lds_slots) {
    const uint32_t nz = ctx->n_zones;
    if (!(ctx->merged && ctx->narrow_ok && !ctx->fifo_generic) || nz + (az_aware ? 1u : 0u) > 16) return false;
    // as many shape-index rows as LDS allows next to the masks (64 down to 4), then as much of the table as fits
    uint32_t ns = 64;
    const uint32_t n_cand = nz + (az_aware ? 1u : 0u);
    while (ns > 4 && gangfit::fifo_zoned_lds_bytes(64, ctx->n_chunks, nz, n_cand, ns) > ctx->lds_budget) ns /= 2;
    const size_t fixed = gangfit::fifo_zoned_lds_bytes(0, ctx->n_chunks, nz, n_cand, ns);
    if (ctx->lds_budget <= fixed + 12 * 64) return false;
    uint32_t slots = (uint32_t)((ctx->lds_budget - fixed) / 12);
    *lds_slots = slots >= ctx->n_slots ? ctx->n_slots : slots / 64 * 64;
    *n_shapes = ns;
    return true;
}
bool minfrag_lds_geometry(const gf_ctx* ctx, bool zoned, uint32_t* n_idx, uint32_t* lds_slots) {
    const uint32_t nz = ctx->n_zones;
    if (!(ctx->merged && ctx->narrow_ok && !ctx->fifo_generic) || (zoned && (nz == 0 || nz > 16))) return false;
    const uint32_t zviews = zoned ? nz : 0u;
    // 64 shape ids per role (rows of the capacity matrix, histograms); as many of them as LDS allows next to the masks also
    // get chunk-index rows (64 down to 0 — the histogram path does without), then as much of the table as fits
    uint32_t ni = 64;
    while (ni > 0 && gangfit::fifo_minfrag_lds_bytes(64, ctx->n_chunks, zviews, ni) > ctx->lds_budget) ni /= 2;
    const size_t fixed = gangfit::fifo_minfrag_lds_bytes(0, ctx->n_chunks, zviews, ni);
    if (ctx->lds_budget <= fixed + 12 * 64) return false;
    uint32_t slots = (uint32_t)((ctx->lds_budget - fixed) / 12);
    *lds_slots = slots >= ctx->n_slots ? ctx->n_slots : slots / 64 * 64;
    *n_idx = ni;
    return true;
}

// The LDS-resident minimal-fragmentation chain when the layout is merged, the table has a narrow form and the tables fit;
// *run_if is then set to the flag the generic kernel must test (it only runs when a request had no scaled form) and
// *served to true.  d_apps / d_results: the arrays of the whole queue (a resumed chain is launched on their tail).
int try_minfrag_lds(gf_ctx* ctx, bool zoned, const gangfit::ZoneTable& zt, uint32_t n_apps, const gf_app* h_apps,
                    const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes, uint64_t half, int32_t* d_failed,
                    hipStream_t stream, const ChainRun* run, const int32_t** run_if, bool* served) {
    *run_if = nullptr;
    *served = false;
    uint32_t n_idx = 0, lds_slots = 0;
    if (!minfrag_lds_geometry(ctx, zoned, &n_idx, &lds_slots)) return GF_OK;
    const uint32_t nz = ctx->n_zones;
    const uint32_t zviews = zoned ? nz : 0u;
    const uint32_t n_shapes = 64;
    GF_HIP(ctx, ctx->d_napps.reserve(n_apps));
    GF_HIP(ctx, ctx->d_zexec.reserve(32 * half));
    gangfit::NarrowTable nt{};
    const int32_t* restore = nullptr;
    const gangfit::ChainCkpt ck = chain_ckpt_args(ctx, run, &restore);
    gangfit::ChainIo io;
    if (const int irc = chain_io_begin(ctx, ck.a_base, run != nullptr && run->narrow_proven, stream, &io); irc != GF_OK) return irc;
    if (const int nrc = narrow_begin(ctx, h_apps, n_apps, stream, &nt, &io, restore); nrc != GF_OK) return nrc;
    // capacity matrix: one int32 per (request shape, slot); skipped (capacities recomputed per pass) beyond 1 GiB
    int32_t* capmat = nullptr;
    if ((uint64_t)n_shapes * ctx->n_slots * sizeof(int32_t) <= (UINT64_C(1) << 30) && ctx->fifo_minfrag_matrix) {
        GF_HIP(ctx, ctx->d_capmat.reserve((size_t)n_shapes * ctx->n_slots + 2048));  // rows are read 2048 slots at a time
        capmat = ctx->d_capmat.ptr;
    }
    int32_t* hist = nullptr;
    if (capmat != nullptr && ctx->fifo_minfrag_hist) {
        GF_HIP(ctx, ctx->d_mfhist.reserve(gangfit::fifo_minfrag_hist_words(zviews, n_shapes)));
        hist = ctx->d_mfhist.ptr;
    }
    const uint32_t a0 = ck.a_base;
    GF_HIP(ctx, gangfit::launch_fit_fifo_minfrag_lds(zoned, make_table(ctx, ctx->d_work.ptr), nt, zt, ctx->d_sched.ptr, lds_slots,
                                                     n_shapes, n_idx, n_apps - a0, d_apps + a0, ctx->d_napps.ptr + a0,
                                                     wide_flag(ctx), d_results + a0, d_exec_nodes, ctx->d_zexec.ptr, half,
                                                     d_failed, capmat, hist, ck, io, ctx->stats_on ? ctx->d_stats.ptr : nullptr, stream));
    *run_if = wide_flag(ctx);
    chain_io_end(ctx, io);
    *served = true;
    return GF_OK;
}

int launch_zoned(gf_ctx* ctx, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* h_apps, const gf_app* d_apps,
                 gf_result* d_results,
                 uint32_t* d_exec_nodes, uint64_t exec_nodes_len, int32_t* d_failed, hipStream_t stream, const ChainRun* run) {
    if (!ctx->have_sched)
        return fail(ctx, GF_ERR_STATE, "zone-aware packers compare packing efficiencies: gf_snapshot_set needs the schedulable columns");
    const int inner = algo == GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION ? GF_ALGO_MINIMAL_FRAGMENTATION : GF_ALGO_TIGHTLY_PACK;
    const uint64_t half = exec_nodes_len + 1;
    const uint32_t nz = ctx->n_zones;
    const uint64_t n_dec = (uint64_t)n_apps * (nz ? nz : 1);
    GF_HIP(ctx, ctx->d_zres.reserve(n_dec));
    GF_HIP(ctx, ctx->d_zexec.reserve(((uint64_t)nz + 1) * half));
    GF_HIP(ctx, ctx->d_zavg.reserve(4 * n_dec));
    GF_HIP(ctx, ctx->d_avg.reserve(4 * (size_t)n_apps));
    int rc = ensure_cnt(ctx, n_dec < 16 ? 16 : n_dec, stream);
    if (rc != GF_OK) return rc;
    gangfit::ZoneTable zt{ctx->d_zmasks.ptr, ctx->d_zmasks.ptr + (size_t)ctx->zd_row0 * ctx->zstride, nz, ctx->zstride};
    gangfit::ZoneBuffers zb{ctx->d_zres.ptr, ctx->d_zexec.ptr, half, ctx->d_zavg.ptr, ctx->d_cnt.ptr, ctx->cnt_rows,
                            ctx->d_avg.ptr};
    if (mode == GF_MODE_FIFO_CHAIN) {
        if (nz + 1 > 64) return fail(ctx, GF_ERR_UNSUPPORTED, "more than 63 zones in a FIFO chain");
        if (ctx->cnt_rows < 16) return fail(ctx, GF_ERR_HIP, "multiplicity scratch too small");
        const bool proven = run != nullptr && run->narrow_proven;  // the LDS chain serves for certain: no generic twin
        // every chain starts from the snapshot: availableNodesSchedulingMetadata is rebuilt per request (resource.go:303);
        // the LDS chains rewrite every real slot of the wide working table in their epilogue
        if (!proven) {
            if (const int arc = apps_to_device(ctx, stream); arc != GF_OK) return arc;  // the generic kernel reads d_apps
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_work.ptr, ctx->d_snap.ptr, 3 * (size_t)ctx->n_slots * sizeof(int64_t),
                                       hipMemcpyDeviceToDevice, stream));
        }
        ctx->work_valid = true;
        const bool az_aware = algo == GF_ALGO_AZ_AWARE_TIGHTLY_PACK;
        const int32_t* run_if = nullptr;
        bool served = false;
        // fast path: tightly-pack family, merged layout, narrow table, every candidate view gets its own wavefront
        uint32_t n_shapes = 0, lds_slots = 0;
        if (inner == GF_ALGO_TIGHTLY_PACK && zoned_lds_geometry(ctx, az_aware, &n_shapes, &lds_slots)) {
            GF_HIP(ctx, ctx->d_napps.reserve(n_apps));
            GF_HIP(ctx, ctx->d_zexec.reserve(32 * half));
            gangfit::NarrowTable nt{};
            const int32_t* restore = nullptr;
            const gangfit::ChainCkpt ck = chain_ckpt_args(ctx, run, &restore);
            gangfit::ChainIo io;
            if (const int irc = chain_io_begin(ctx, ck.a_base, proven, stream, &io); irc != GF_OK) return irc;
            if (const int nrc = narrow_begin(ctx, h_apps, n_apps, stream, &nt, &io, restore); nrc != GF_OK) return nrc;
            const uint32_t a0 = ck.a_base;
            GF_HIP(ctx, gangfit::launch_fit_fifo_zoned_lds(az_aware, make_table(ctx, ctx->d_work.ptr), nt, zt, ctx->d_sched.ptr,
                                                           lds_slots, n_shapes, n_apps - a0, d_apps + a0, ctx->d_napps.ptr + a0,
                                                           wide_flag(ctx), d_results + a0, d_exec_nodes,
                                                           ctx->d_zexec.ptr, half, d_failed, ck, io,
                                                           ctx->stats_on ? ctx->d_stats.ptr : nullptr, stream));
            run_if = wide_flag(ctx);  // the generic kernel below only runs when a request had no scaled form
            chain_io_end(ctx, io);
            zb.zexec = ctx->d_zexec.ptr;
            served = true;
        }
        if (inner == GF_ALGO_MINIMAL_FRAGMENTATION) {
            const int rc2 = try_minfrag_lds(ctx, true, zt, n_apps, h_apps, d_apps, d_results, d_exec_nodes, half, d_failed, stream,
                                            run, &run_if, &served);
            if (rc2 != GF_OK) return rc2;
            if (run_if) zb.zexec = ctx->d_zexec.ptr;
        }
        if (served && proven) return GF_OK;
        if (proven) return fail(ctx, GF_ERR_HIP, "chain plan and launch disagree about the LDS chain");
        GF_HIP(ctx, gangfit::launch_fit_fifo_generic(inner, true, az_aware,
                                                     reserves_executors(algo), make_table(ctx, ctx->d_work.ptr), zt,
                                                     ctx->d_sched.ptr, zb, n_apps, d_apps, d_results, d_exec_nodes,
                                                     ctx->d_scratch.ptr, half, d_failed, run_if, stream));
        return GF_OK;
    }
    if (const int arc = apps_to_device(ctx, stream); arc != GF_OK) return arc;
    GF_HIP(ctx, gangfit::launch_fit_zoned(inner, algo == GF_ALGO_AZ_AWARE_TIGHTLY_PACK,
                                          reserves_executors(algo), make_table(ctx, ctx->d_snap.ptr), zt,
                                          slot_eff_tables(ctx, ctx->d_snap.ptr), zb, n_apps, d_apps, d_results,
                                          d_exec_nodes, ctx->d_scratch.ptr, half, stream));
    return GF_OK;
}

// Which chains resume: every packer, when its LDS-resident chain kernel serves (merged layout, narrow table, the kernel's
// tables fit) and every request has a scaled form.  Returns false when the chain cache is not used for this call (run stays {0, false, false}).
bool chain_plan(gf_ctx* ctx, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* h_apps, ChainRun* run) {
    *run = ChainRun{};
    gf_ctx::ChainCache& C = ctx->chain;
    if (mode != GF_MODE_FIFO_CHAIN || !ctx->chain_cache_on || ctx->stats_on || !ctx->have_orders) return false;
    if (!(ctx->merged && ctx->narrow_ok) || ctx->fifo_generic) return false;
    bool solo = false, table_in_lds = false;
    {  // the LDS-resident chain kernel of this packer must be the one that serves (they dump and restore the checkpoints)
        uint32_t g0 = 0, g1 = 0;
        bool lds_chain = false;
        switch (algo) {
        case GF_ALGO_TIGHTLY_PACK:
        case GF_ALGO_DISTRIBUTE_EVENLY:
            lds_chain = solo = true;
            g1 = solo_lds_slots(ctx);
            break;
        case GF_ALGO_SINGLE_AZ_TIGHTLY_PACK: lds_chain = ctx->have_sched && zoned_lds_geometry(ctx, false, &g0, &g1); break;
        case GF_ALGO_AZ_AWARE_TIGHTLY_PACK: lds_chain = ctx->have_sched && zoned_lds_geometry(ctx, true, &g0, &g1); break;
        case GF_ALGO_MINIMAL_FRAGMENTATION: lds_chain = minfrag_lds_geometry(ctx, false, &g0, &g1); break;
        case GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION: lds_chain = ctx->have_sched && minfrag_lds_geometry(ctx, true, &g0, &g1); break;
        default: break;
        }
        if (!lds_chain) return false;
        table_in_lds = g1 >= ctx->n_slots;
    }
    // The narrow units of the queue and the proof that every request has a scaled form.  A scan of the whole queue is twelve
    // 64-bit divisions per application — more host time than a resumed chain takes on the device —, so a queue that shares a
    // prefix with the cached one is only scanned behind it: the cached units divide the prefix by construction, and when they
    // divide the new applications too they ARE a valid set of units for this queue (any common divisor keeps the chain exact;
    // the checkpoints are scaled in them).  Otherwise: the full scan, and the chain replays.
    int64_t eff[3];
    int32_t factor[3];
    bool proven = false;
    uint32_t common = 0;  // applications this queue shares with the cached one, from the front (the last of either excluded)
    bool units_from_cache = false;
    if (C.valid && C.epoch == ctx->snap_epoch && C.algo == (int)algo && C.n_apps > 0) {
        const uint32_t lim = (n_apps < C.n_apps ? n_apps : C.n_apps) - 1;
        while (common < lim && std::memcmp(&h_apps[common], &C.apps[common], sizeof(gf_app)) == 0) ++common;
        bool ok = common > 0;
        for (uint32_t i = common; i < n_apps && ok; ++i) {
            ok = h_apps[i].k >= 0 && h_apps[i].k <= GF_MAX_K;
            for (int j = 0; j < 3 && ok; ++j)
                for (const int64_t v : {h_apps[i].drv[j], h_apps[i].exe[j]})
                    ok = ok && v >= 0 && v % C.unit[j] == 0 && v / C.unit[j] < (INT64_C(1) << 30);
        }
        if (ok) {
            units_from_cache = proven = true;
            for (int j = 0; j < 3; ++j) {
                eff[j] = C.unit[j];
                factor[j] = (int32_t)(ctx->unit[j] / C.unit[j]);  // (the cached chain passed the range check with these)
            }
        }
    }
    if (!units_from_cache) narrow_units(ctx, h_apps, n_apps, eff, factor, &proven);
    if (!proven) return false;
    // checkpoint interval: 32 applications while a dump is cheap — the whole table from LDS, or (solo kernel, table with a
    // global tail) only the chunks that differ from the snapshot; 128 where a dump copies a table that lives in global memory
    // (the zone-aware and minimal-fragmentation chains beyond their LDS front); wider when 128 dumps would not fit 2 GiB
    const size_t slot_words = gangfit::chain_ckpt_stride(ctx->n_slots, ctx->n_chunks);
    const bool dirty_format = solo && !table_in_lds;
    uint32_t shift = (table_in_lds || solo) ? 5 : 7;
    while (shift < 12 && (size_t)(4096u >> shift) * slot_words * sizeof(int32_t) > (UINT64_C(2) << 30)) ++shift;
    const size_t n_ck = (size_t)((n_apps - 1) >> shift);
    if (n_ck * slot_words * sizeof(int32_t) > (UINT64_C(4) << 30)) return false;
    uint32_t a_begin = 0;
    const bool same = C.valid && C.epoch == ctx->snap_epoch && C.algo == (int)algo && C.shift == shift && C.dirty_format == dirty_format &&
                      C.slot_words == slot_words && C.unit[0] == eff[0] && C.unit[1] == eff[1] && C.unit[2] == eff[2];
    if (same) {
        // longest common prefix of the two queues, the last application of either excluded (nothing is committed behind
        // the driver being filtered: its table is not a state of the longer chain)
        uint32_t c = common >> shift;
        if (c > C.n_ckpt) c = C.n_ckpt;
        a_begin = c << shift;
    }
    // the checkpoint buffer keeps what it holds when it grows
    if (n_ck * slot_words > C.d_ckpt.cap) {
        size_t want = C.d_ckpt.cap ? C.d_ckpt.cap : 32 * slot_words;
        while (want < n_ck * slot_words) want *= 2;
        int32_t* fresh = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(int32_t)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        const size_t keep = (size_t)(a_begin >> shift) * slot_words;
        if (keep && hipMemcpy(fresh, C.d_ckpt.ptr, keep * sizeof(int32_t), hipMemcpyDeviceToDevice) != hipSuccess) {
            (void)hipFree(fresh);
            return false;
        }
        if (C.d_ckpt.ptr) {
            (void)gf_wait_stream(ctx->stream);
            (void)hipFree(C.d_ckpt.ptr);
        }
        C.d_ckpt.ptr = fresh;
        C.d_ckpt.cap = want;
    }
    if (!same) C.valid = false;
    C.shift = shift;
    C.slot_words = slot_words;
    C.dirty_format = dirty_format;
    for (int j = 0; j < 3; ++j) C.unit[j] = eff[j];
    run->a_begin = a_begin;
    run->common = same ? common : 0;
    run->record = true;
    run->narrow_proven = true;
    ctx->planned_units.valid = true;
    for (int j = 0; j < 3; ++j) {
        ctx->planned_units.eff[j] = eff[j];
        ctx->planned_units.factor[j] = factor[j];
    }
    return true;
}

// The chain that just ran becomes the cached one (h_results / h_exec hold the complete answer, prefix included).
void chain_commit(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, uint64_t total_k, int32_t failed_at, const ChainRun& run) {
    gf_ctx::ChainCache& C = ctx->chain;
    // what the cached queue already holds stays: records up to the common prefix, answers up to the first application evaluated
    const uint32_t keep_apps = C.valid ? (run.common < n_apps ? run.common : n_apps) : 0u;
    const uint32_t keep_res = C.valid ? run.a_begin : 0u;
    const uint64_t keep_exec = keep_res > 0 ? ctx->h_apps.ptr[keep_res].exec_off : 0;
    C.apps.resize(n_apps);
    std::memcpy(C.apps.data() + keep_apps, ctx->h_apps.ptr + keep_apps, (size_t)(n_apps - keep_apps) * sizeof(gf_app));
    C.results.resize(n_apps);
    std::memcpy(C.results.data() + keep_res, ctx->h_results.ptr + keep_res, (size_t)(n_apps - keep_res) * sizeof(gf_result));
    C.exec.resize(total_k);
    if (total_k > keep_exec)
        std::memcpy(C.exec.data() + keep_exec, ctx->h_exec.ptr + keep_exec, (size_t)(total_k - keep_exec) * sizeof(uint32_t));
    C.n_apps = n_apps;
    C.failed_at = failed_at;
    C.algo = (int)algo;
    C.epoch = ctx->snap_epoch;
    // the chain reached application `last` (the one it aborted at, else the filtered driver): dumps exist up to there
    const uint32_t last = failed_at >= 0 ? (uint32_t)failed_at : n_apps - 1;
    C.n_ckpt = last >> C.shift;
    C.valid = true;
    ctx->chain_stat[0] += 1;
    ctx->chain_stat[1] += run.a_begin > 0 ? 1 : 0;
    ctx->chain_stat[2] += (failed_at >= 0 ? (uint32_t)failed_at + 1 : n_apps) - run.a_begin;
    ctx->chain_stat[3] += run.a_begin;
}

// h_apps: the same records on the host when the caller has them (gf_fit_batch), nullptr for device-resident batches.
// run (nullable): gf_fit_batch's plan for a FIFO chain; d_apps / d_results are always the arrays of the WHOLE queue.
int launch(gf_ctx* ctx, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* h_apps, const gf_app* d_apps,
           gf_result* d_results, uint32_t* d_exec_nodes, uint64_t exec_nodes_len, int32_t* d_failed, hipStream_t stream,
           const ChainRun* run = nullptr, const gangfit::IndHostOut* host_out = nullptr) {
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede a fit");
    const uint64_t half = exec_nodes_len + 1;
    GF_HIP(ctx, ctx->d_scratch.reserve(2 * half));
    if (is_zone_algo(algo)) {
        if (mode != GF_MODE_INDEPENDENT && mode != GF_MODE_FIFO_CHAIN)
            return fail(ctx, GF_ERR_UNSUPPORTED, "unknown gf_mode %d", (int)mode);
        return launch_zoned(ctx, mode, algo, n_apps, h_apps, d_apps, d_results, d_exec_nodes, exec_nodes_len, d_failed, stream, run);
    }
    if (algo != GF_ALGO_TIGHTLY_PACK && algo != GF_ALGO_DISTRIBUTE_EVENLY && algo != GF_ALGO_MINIMAL_FRAGMENTATION)
        return fail(ctx, GF_ERR_UNSUPPORTED, "gf_algo %d is not served by the device path", (int)algo);
    if (algo == GF_ALGO_MINIMAL_FRAGMENTATION && mode == GF_MODE_FIFO_CHAIN) {
        // the LDS chain; else (and as its guarded twin) the generic chain kernel: one candidate view, one wavefront, against
        // the working table in global memory
        const bool proven = run != nullptr && run->narrow_proven;
        GF_HIP(ctx, ctx->d_zexec.reserve(half));
        if (!proven) {
            if (const int arc = apps_to_device(ctx, stream); arc != GF_OK) return arc;  // the generic kernel reads d_apps
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_work.ptr, ctx->d_snap.ptr, 3 * (size_t)ctx->n_slots * sizeof(int64_t),
                                       hipMemcpyDeviceToDevice, stream));
        }
        ctx->work_valid = true;
        gangfit::ZoneTable zt{nullptr, nullptr, 0, 0};
        const int32_t* run_if = nullptr;
        bool served = false;
        const int rc2 = try_minfrag_lds(ctx, false, zt, n_apps, h_apps, d_apps, d_results, d_exec_nodes, half, d_failed, stream, run,
                                        &run_if, &served);
        if (rc2 != GF_OK) return rc2;
        if (served && proven) return GF_OK;
        if (proven) return fail(ctx, GF_ERR_HIP, "chain plan and launch disagree about the LDS chain");
        gangfit::ZoneBuffers zb{nullptr, ctx->d_zexec.ptr, half, nullptr, nullptr, 0, nullptr};
        GF_HIP(ctx, gangfit::launch_fit_fifo_generic(GF_ALGO_MINIMAL_FRAGMENTATION, false, false, false,
                                                     make_table(ctx, ctx->d_work.ptr), zt, nullptr, zb, n_apps, d_apps,
                                                     d_results, d_exec_nodes, ctx->d_scratch.ptr, half, d_failed, run_if,
                                                     stream));
        return GF_OK;
    }
    ScanStats* stats = ctx->stats_on ? ctx->d_stats.ptr : nullptr;
    if (mode == GF_MODE_INDEPENDENT) {
        if (const int arc = apps_to_device(ctx, stream); arc != GF_OK) return arc;
        GF_HIP(ctx, gangfit::launch_fit_independent(algo, make_table(ctx, ctx->d_snap.ptr), make_sparse(ctx), n_apps, d_apps,
                                                    d_results, d_exec_nodes, ctx->d_scratch.ptr, half, stats, stream, host_out));
    } else if (mode == GF_MODE_FIFO_CHAIN) {
        gangfit::FifoPlan plan{};
        plan.narrow = ctx->merged && ctx->narrow_ok && !ctx->fifo_generic;
        plan.wide = !(plan.narrow && run != nullptr && run->narrow_proven);
        const uint32_t a_begin = (plan.narrow && run != nullptr) ? run->a_begin : 0u;
        // every chain starts from the snapshot: availableNodesSchedulingMetadata is rebuilt per request (resource.go:303).
        // The solo kernel rewrites every real slot of the wide working table in its epilogue: the copy is only needed by the
        // wide kernel.  Like the narrow table's, the copy is made by the chain's first kernel (ChainIo).
        gangfit::ChainIo io;
        if (const int irc = chain_io_begin(ctx, a_begin, true, stream, &io); irc != GF_OK) return irc;
        if (plan.wide) {
            io.copy_src[1] = reinterpret_cast<const uint32_t*>(ctx->d_snap.ptr);
            io.copy_dst[1] = reinterpret_cast<uint32_t*>(ctx->d_work.ptr);
            io.copy_words[1] = 3 * (size_t)ctx->n_slots * (sizeof(int64_t) / sizeof(uint32_t));
        }
        ctx->work_valid = true;
        // as much of the table front as fits next to each kernel's fixed LDS needs stays in LDS for the whole chain
        auto front = [&](size_t fixed, size_t per_slot, uint32_t round) {
            uint32_t n = ctx->lds_budget > fixed ? (uint32_t)((ctx->lds_budget - fixed) / per_slot) : 0;
            const uint32_t whole = (ctx->n_slots + round - 1) / round * round;  // the whole table, padded to full steps
            if (n >= whole) return whole;
            return n / round * round;
        };
        plan.lds_slots_v2 = front(gangfit::fifo_v2_lds_bytes(0, ctx->n_chunks), 24, 64);
        if (plan.lds_slots_v2 > ctx->n_slots) plan.lds_slots_v2 = ctx->n_slots;
        plan.lds_slots_solo = solo_lds_slots(ctx);
        gangfit::NarrowTable nt{};
        gangfit::ChainCkpt ck{nullptr, 0u, ctx->chain.shift};
        if (plan.narrow) {
            GF_HIP(ctx, ctx->d_napps.reserve(n_apps));
            const int32_t* restore = nullptr;
            ck = chain_ckpt_args(ctx, run, &restore);
            if (const int nrc = narrow_begin(ctx, h_apps, n_apps, stream, &nt, &io, restore, ctx->chain.dirty_format); nrc != GF_OK)
                return nrc;
        }
        // a resumed chain is launched on the tail of the queue: exec_off is absolute, so offset pointers are all it takes
        const uint64_t heads_lo = a_begin > 0 ? h_apps[a_begin].exec_off : 0;
        GF_HIP(ctx, gangfit::launch_fit_fifo(algo, plan, make_table(ctx, ctx->d_work.ptr), nt, n_apps - a_begin, d_apps + a_begin,
                                             ctx->d_napps.ptr + a_begin, wide_flag(ctx), d_results + a_begin,
                                             d_exec_nodes, ctx->d_scratch.ptr, half, heads_lo, d_failed, ck, io, stats, stream));
        chain_io_end(ctx, io);
    } else {
        return fail(ctx, GF_ERR_UNSUPPORTED, "unknown gf_mode %d", (int)mode);
    }
    return GF_OK;
}

// Point a view at the snapshot its parent holds now (the caller holds the parent's views_mu shared: no install is running).
int view_refresh(gf_ctx* v) {
    const gf_ctx* p = v->view_of;
    if (v->view_epoch == p->snap_epoch) return GF_OK;
    GF_HIP(v, hipSetDevice(v->device));
    GF_HIP(v, gf_wait_stream(v->stream));
    v->n_nodes = p->n_nodes;
    v->have_snapshot = p->have_snapshot;
    v->have_sched = p->have_sched;
    v->have_orders = p->have_orders;
    v->n_x = p->n_x;
    v->n_d = p->n_d;
    v->n_slots = p->n_slots;
    v->n_chunks = p->n_chunks;
    v->d_identity = p->d_identity;
    v->merged = p->merged;
    v->n_g = p->n_g;
    v->n_gpad = p->n_gpad;
    v->narrow_ok = p->narrow_ok;
    for (int j = 0; j < 3; ++j) {
        v->unit[j] = p->unit[j];
        v->nmax[j] = p->nmax[j];
    }
    v->n_zones = p->n_zones;
    v->zstride = p->zstride;
    v->zd_row0 = p->zd_row0;
    v->d_snap.alias(p->d_snap);
    v->d_slot_node.alias(p->d_slot_node);
    v->d_dslot.alias(p->d_dslot);
    v->d_node_slot.alias(p->d_node_slot);
    v->d_cmax.alias(p->d_cmax);
    v->d_masks.alias(p->d_masks);
    v->d_gtab.alias(p->d_gtab);
    v->d_gcmax.alias(p->d_gcmax);
    v->d_gidx.alias(p->d_gidx);
    v->d_gmask.alias(p->d_gmask);
    v->d_nsnap.alias(p->d_nsnap);
    v->d_ncmax.alias(p->d_ncmax);
    v->d_sched.alias(p->d_sched);
    v->d_node_tab.alias(p->d_node_tab);
    v->d_zmasks.alias(p->d_zmasks);
    // the working copies are the view's own
    if (v->have_orders) {
        GF_HIP(v, v->d_work.reserve(3 * (size_t)v->n_slots));
        if (v->narrow_ok) GF_HIP(v, v->d_nwork.reserve(3 * (size_t)v->n_slots));
    }
    v->work_valid = false;
    v->host_stale = true;  // host mirrors (residuals, efficiencies) are fetched from the aliased device tables when asked for
    v->cnt_slots = 0;      // the multiplicity scratch is sized by n_slots
    v->cnt_rows = 0;
    ++v->snap_epoch;       // drops the view's chain cache and recorded graphs
    v->view_epoch = p->snap_epoch;
    return GF_OK;
}

}  // namespace

extern "C" {

int gf_version(void) { return GF_VERSION; }

int gf_init(const int* device_ids, int n_dev, gf_ctx** out) {
    if (!out) return GF_ERR_INVALID;
    *out = nullptr;
    if (n_dev > 1) {
        // One context over several devices: sub-context i owns range i of n_dev of the priority order.  A device id may
        // repeat (several shards on one GPU: how the path is exercised on a one-GPU box).
        if (!device_ids || n_dev > (int)gangfit::kMaxGroupDevices) return GF_ERR_INVALID;
        gf_ctx* g = new (std::nothrow) gf_ctx();
        if (!g) return GF_ERR_HIP;
        g->device = device_ids[0];
        for (int i = 0; i < n_dev; ++i) {
            gf_ctx* sub = nullptr;
            const int rc = gf_init(&device_ids[i], 1, &sub);
            if (rc != GF_OK) {
                gf_destroy(g);
                return rc;
            }
            sub->shard = (uint32_t)i;
            sub->n_shards = (uint32_t)n_dev;
            // what the other devices store into / read from lives in fine-grained memory: a posted peer store must not depend
            // on what a kernel boundary does to this device's caches
            sub->g_part_all.fine = sub->g_drv_all.fine = sub->g_exec2.fine = true;
            // Shards on ONE device (a repeated id: how the path runs on a one-GPU box) share a stream: separate streams buy them
            // nothing there, and with sixteen hardware queues every cross-stream event wait of the exchanges is a real
            // cross-queue barrier (eight shards: 1.0 ms per headline batch with eight streams, 0.35 with the runtime's four
            // queues).  Shards on different devices keep their own.
            for (gf_ctx* earlier : g->group)
                if (earlier->device == sub->device) {
                    (void)hipSetDevice(sub->device);
                    (void)hipStreamDestroy(sub->stream);
                    sub->stream = earlier->stream;
                    sub->stream_borrowed = true;
                    break;
                }
            g->group.push_back(sub);
            g->g_devices.push_back(device_ids[i]);
            bool ok = hipSetDevice(sub->device) == hipSuccess;
            for (hipEvent_t& e : sub->g_ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
            if (!ok) {
                gf_destroy(g);
                return GF_ERR_HIP;
            }
        }
        bool peers = std::getenv("GANGFIT_TEST_NO_PEER") == nullptr;  // (fault injection of host_test: "no device can reach another")
        for (int i = 0; i < n_dev && peers; ++i)  // every shard's kernels write into / read from every other shard's buffers
            for (int j = 0; j < n_dev && peers; ++j) {
                if (device_ids[i] == device_ids[j]) continue;
                int can = 0;
                if (hipSetDevice(device_ids[i]) != hipSuccess ||
                    hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) != hipSuccess || !can) {
                    peers = false;
                    break;
                }
                const hipError_t e = hipDeviceEnablePeerAccess(device_ids[j], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) peers = false;
                (void)hipGetLastError();
            }
        if (!peers) {
            // The devices cannot reach each other's memory: serve everything from the first device instead of refusing the
            // whole context (the Go host would otherwise run every Filter on the CPU).  gf_shard_count says so.
            gf_destroy(g);
            const int rc = gf_init(&device_ids[0], 1, out);
            if (rc == GF_OK) (*out)->err = "peer access between the requested devices is unavailable: serving from the first device only";
            return rc;
        }
        g->info = g->group[0]->info;
        *out = g;
        return GF_OK;
    }
    if (n_dev != 1 && !(n_dev == 0 && device_ids == nullptr)) return GF_ERR_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GF_ERR_NO_DEVICE;
    const int dev = device_ids ? device_ids[0] : 0;
    if (dev < 0 || dev >= count) return GF_ERR_NO_DEVICE;
    gf_ctx* ctx = new (std::nothrow) gf_ctx();
    if (!ctx) return GF_ERR_HIP;
    ctx->device = dev;
    hipDeviceProp_t prop;
    if (hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        delete ctx;
        return GF_ERR_NO_DEVICE;
    }
    std::snprintf(ctx->info.name, sizeof ctx->info.name, "%s", prop.name);
    std::snprintf(ctx->info.arch, sizeof ctx->info.arch, "%s", prop.gcnArchName);
    ctx->info.compute_units = prop.multiProcessorCount;
    ctx->info.lds_bytes_per_cu = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    ctx->info.wavefront_size = prop.warpSize;
    ctx->info.clock_khz = prop.clockRate;
    ctx->info.hbm_bytes = (int64_t)prop.totalGlobalMem;
    ctx->lds_budget = (uint32_t)prop.maxSharedMemoryPerMultiProcessor;
    // the one switch of the data path a deployment may want (GANGFIT_WAIT is the other environment variable, see above):
    // every chain replays from the snapshot, as the reference does
    if (const char* z = std::getenv("GANGFIT_CHAIN_CACHE")) ctx->chain_cache_on = std::strcmp(z, "0") != 0;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 || prop.warpSize != 64) {
        delete ctx;
        return GF_ERR_NO_DEVICE;  // the kernels are gfx950 / wave64 only
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess ||
        ctx->d_stats.reserve(1) != hipSuccess || ctx->d_failed.reserve(1) != hipSuccess ||
        ctx->d_wide_needed.reserve(2) != hipSuccess ||
        ctx->h_failed.reserve(1) != hipSuccess ||
        hipMemset(ctx->d_wide_needed.ptr, 0, 2 * sizeof(int32_t)) != hipSuccess ||
        hipMemset(ctx->d_stats.ptr, 0, sizeof(ScanStats)) != hipSuccess) {
        gf_destroy(ctx);
        return GF_ERR_HIP;
    }
    *out = ctx;
    return GF_OK;
}

int gf_ctx_view(gf_ctx* parent, gf_ctx** out) {
    if (!parent || !out) return GF_ERR_INVALID;
    *out = nullptr;
    if (!parent->group.empty()) return fail(parent, GF_ERR_UNSUPPORTED, "views of a multi-device context are not served");
    if (parent->view_of != nullptr) parent = parent->view_of;  // a view of a view is a view of the same parent
    gf_ctx* v = nullptr;
    const int rc = gf_init(&parent->device, 1, &v);
    if (rc != GF_OK) return rc;
    {
        std::lock_guard<std::recursive_mutex> lock(parent->mu);
        ++parent->n_views;
        parent->views.push_back(v);
        v->lds_budget = parent->lds_budget;
        v->fifo_generic = parent->fifo_generic;
        v->fifo_minfrag_matrix = parent->fifo_minfrag_matrix;
        v->fifo_minfrag_hist = parent->fifo_minfrag_hist;
        v->chain_cache_on = parent->chain_cache_on;
        v->zero_copy = parent->zero_copy;
    }
    v->view_of = parent;
    // The HIP runtime multiplexes streams over a few hardware queues (four unless GPU_MAX_HW_QUEUES says otherwise, read when
    // the runtime initialises), and two FIFO chains whose streams share a queue run one after the other: eight views took 3.1x
    // one chain's time with four queues, 1.1x with sixteen (host_test gpu, TestConcurrentViews).  That variable belongs to the
    // deployment ("Deployment"), not to a library loaded into somebody else's process: say so, once per view.
    {
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");
        const long nq = q ? std::strtol(q, nullptr, 10) : 0;
        if (nq < 8)
            v->err = "note: GPU_MAX_HW_QUEUES is " + std::string(q ? q : "unset (the HIP runtime's default is 4 hardware queues)") +
                     ": chains of concurrent views may run one after the other; set GPU_MAX_HW_QUEUES=16 in the extender's "
                     "environment before the process starts";
    }
    *out = v;
    return GF_OK;
}

void gf_destroy(gf_ctx* ctx) {
    if (!ctx) return;
    if (!ctx->group.empty()) {
        for (void* c : ctx->g_comms)
            if (c) (void)rccl().CommDestroy(c);
        ctx->g_comms.clear();
        for (auto it = ctx->group.rbegin(); it != ctx->group.rend(); ++it) gf_destroy(*it);  // (borrowers of a stream before its owner)
        ctx->group.clear();
        (void)hipSetDevice(ctx->device);
        ctx->h_apps.release();
        ctx->h_results.release();
        ctx->h_exec.release();
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)gf_wait_stream(ctx->stream);
    // the resident worker reads the tables released below as kernel arguments: it serves what was posted and leaves first
    // (hipFree would otherwise wait — implicitly, and for up to worker_idle_us — for a kernel that is still reading them)
    if (ctx->worker.allocated) worker_quiesce(ctx);
    if (ctx->view_of != nullptr) {
        std::lock_guard<std::recursive_mutex> plock(ctx->view_of->mu);
        --ctx->view_of->n_views;
        auto& vs = ctx->view_of->views;
        vs.erase(std::remove(vs.begin(), vs.end(), ctx), vs.end());
    }
    ctx->d_snap.release();
    ctx->d_work.release();
    ctx->d_slot_node.release();
    ctx->d_dslot.release();
    ctx->d_node_slot.release();
    ctx->d_cmax.release();
    ctx->d_masks.release();
    ctx->h_masks.release();
    ctx->d_gtab.release();
    ctx->d_gcmax.release();
    ctx->d_gidx.release();
    ctx->d_gmask.release();
    ctx->h_gtab.release();
    ctx->h_gidx.release();
    ctx->d_napps.release();
    ctx->chain.d_ckpt.release();
    ctx->d_flag32.release();
    ctx->d_ind_done.release();
    ctx->h_ind_flag.release();
    ctx->d_sortwork.release();
    ctx->d_wide_needed.release();
    ctx->d_capmat.release();
    ctx->d_mfhist.release();
    ctx->d_nsnap.release();
    ctx->d_nwork.release();
    ctx->d_ncmax.release();
    ctx->d_ncmax_w.release();
    ctx->h_ntable.release();
    ctx->h_cmax.release();
    ctx->d_sched.release();
    ctx->d_node_tab.release();
    ctx->d_zmasks.release();
    ctx->h_zmasks.release();
    ctx->d_zres.release();
    ctx->d_zexec.release();
    ctx->d_zavg.release();
    ctx->d_avg.release();
    ctx->d_cnt.release();
    ctx->d_reserved.release();
    ctx->d_eff.release();
    ctx->h_avg.release();
    ctx->d_cl_i64.release();
    ctx->d_cl_u32.release();
    ctx->d_cl_usage.release();
    ctx->d_delta_i64.release();
    ctx->d_delta_u32.release();
    ctx->d_bi64.release();
    ctx->d_bu32.release();
    ctx->h_bcols.release();
    ctx->h_border.release();
    ctx->d_xexe.release();
    ctx->d_xreserved.release();
    ctx->d_xhosts.release();
    ctx->d_xout.release();
    ctx->g_part_loc.release();
    ctx->g_part_all.release();
    ctx->g_drv_loc.release();
    ctx->g_drv_all.release();
    ctx->g_exec2.release();
    for (hipEvent_t& e : ctx->g_ev)
        if (e) (void)hipEventDestroy(e);
    ctx->d_fk.release();
    ctx->d_foff.release();
    ctx->d_fres.release();
    ctx->d_fadds.release();
    ctx->h_foff.release();
    ctx->d_apps.release();
    ctx->d_results.release();
    ctx->d_exec.release();
    ctx->d_scratch.release();
    ctx->d_failed.release();
    ctx->d_stats.release();
    ctx->h_table.release();
    ctx->h_index.release();
    ctx->h_apps.release();
    ctx->h_results.release();
    ctx->h_exec.release();
    ctx->h_failed.release();
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->worker.allocated) {
        ctx->worker.scratch.release();
        if (ctx->worker.stage) (void)hipHostFree(ctx->worker.stage);
        if (ctx->worker.d) (void)hipFree(ctx->worker.d);
        if (ctx->worker.h) (void)hipHostFree(ctx->worker.h);
        if (ctx->worker.ev0) (void)hipEventDestroy(ctx->worker.ev0);
        if (ctx->worker.ev1) (void)hipEventDestroy(ctx->worker.ev1);
        if (ctx->worker.stream) (void)hipStreamDestroy(ctx->worker.stream);
    }
    if (ctx->stream && !ctx->stream_borrowed) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void gf_ctx_lock(gf_ctx* ctx) {
    if (!ctx) return;
    std::unique_lock<std::mutex> l(ctx->seq_m);
    ctx->seq_cv.wait(l, [ctx] { return !ctx->seq_held; });
    ctx->seq_held = true;
}

void gf_ctx_unlock(gf_ctx* ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> l(ctx->seq_m);
        ctx->seq_held = false;
    }
    ctx->seq_cv.notify_one();
}

const char* gf_last_error(gf_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gf_set_option(gf_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return GF_ERR_INVALID;
    if (!ctx->group.empty()) {
        std::lock_guard<std::recursive_mutex> glock(ctx->mu);
        const std::string gk(key);
        if (gk == "group_verify") {
            ctx->g_verify = value != 0;
            return GF_OK;
        }
        if (gk == "group_fault") {
            ctx->g_fault = (int)value;
            ctx->g_verified_epoch = 0;
            return GF_OK;
        }
        if (gk == "group_shard_off") {  // read-back for tests: 1 sets, 0 clears (and re-arms the self-check)
            ctx->g_shard_off = value != 0;
            ctx->g_verified_epoch = 0;
            return GF_OK;
        }
        if (gk == "group_exchange") {
            for (void* c : ctx->g_comms)
                if (c) (void)rccl().CommDestroy(c);
            ctx->g_comms.clear();
            ctx->g_verified_epoch = 0;  // the other exchange proves itself on its first batch
            if (value == 0) return GF_OK;
            if (!rccl().load()) return fail(ctx, GF_ERR_UNSUPPORTED, "librccl.so cannot be loaded");
            std::vector<void*> comms(ctx->group.size(), nullptr);
            const int rc = rccl().CommInitAll(comms.data(), (int)comms.size(), ctx->g_devices.data());
            if (rc != 0)  // e.g. a device id that repeats: RCCL wants one rank per physical device
                return fail(ctx, GF_ERR_UNSUPPORTED, "ncclCommInitAll failed: %s",
                            rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
            ctx->g_comms = comms;
            return GF_OK;
        }
        for (gf_ctx* sub : ctx->group)
            if (const int rc = gf_set_option(sub, key, value); rc != GF_OK) {
                ctx->err = sub->err;
                return rc;
            }
        return GF_OK;
    }
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const std::string k(key);
    if (k == "lds_budget") {
        if (value < 0 || value > (int64_t)ctx->info.lds_bytes_per_cu) return fail(ctx, GF_ERR_INVALID, "lds_budget outside [0, %d]", ctx->info.lds_bytes_per_cu);
        ctx->lds_budget = (uint32_t)value;
    } else if (k == "fifo_generic") {
        ctx->fifo_generic = value != 0;
    } else if (k == "minfrag_matrix") {
        ctx->fifo_minfrag_matrix = value != 0;
    } else if (k == "minfrag_hist") {
        ctx->fifo_minfrag_hist = value != 0;
    } else if (k == "sparse_gpu") {
        ctx->sparse_gpu = value != 0;
    } else if (k == "zero_copy") {
        ctx->zero_copy = value != 0;
    } else if (k == "host_flag") {
        ctx->host_flag = value != 0;
    } else if (k == "snapshot_finalize_host") {
        ctx->snapshot_finalize_on_device = value == 0;
    } else if (k == "sort_fault") {
        ctx->sort_fault = value != 0 ? 1 : 0;
    } else if (k == "force_general_layout") {
        ctx->force_general_layout = value != 0;
    } else if (k == "chain_cache") {
        ctx->chain_cache_on = value != 0;
    } else if (k == "worker_sets" || k == "worker_blocks_per_set" || k == "worker_idle_us") {
        worker_quiesce(ctx);
        if (k == "worker_sets") {
            if (value < 1 || value > 16) return fail(ctx, GF_ERR_INVALID, "worker_sets outside [1, 16]");
            ctx->worker.sets = (uint32_t)value;
        } else if (k == "worker_blocks_per_set") {
            if (value < 1 || value > 1024) return fail(ctx, GF_ERR_INVALID, "worker_blocks_per_set outside [1, 1024]");
            ctx->worker.blocks_per_set = (uint32_t)value;
        } else {
            if (value < 10 || value > 1000000) return fail(ctx, GF_ERR_INVALID, "worker_idle_us outside [10, 10^6]");
            ctx->worker.idle_us = (uint32_t)value;
        }
    } else if (k == "rccl_selftest") {
        // the run-time binding of the collective library, exercised with a one-rank communicator on this device: an
        // all-gather and a reduction of `value` words must reproduce their input
        if (!rccl().load()) return fail(ctx, GF_ERR_UNSUPPORTED, "librccl.so cannot be loaded");
        if (value <= 0 || value > (1 << 20)) return fail(ctx, GF_ERR_INVALID, "rccl_selftest wants a word count in (0, 2^20]");
        GF_HIP(ctx, hipSetDevice(ctx->device));
        void* comm = nullptr;
        int rc = rccl().CommInitAll(&comm, 1, &ctx->device);
        if (rc != 0) return fail(ctx, GF_ERR_UNSUPPORTED, "ncclCommInitAll failed: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
        const size_t n = (size_t)value;
        std::vector<uint32_t> h(n), back(2 * n, 0u);
        for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u + 7u);
        uint32_t* d = nullptr;
        bool ok = hipMalloc(reinterpret_cast<void**>(&d), 3 * n * sizeof(uint32_t)) == hipSuccess &&
                  hipMemcpy(d, h.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && rccl().AllGather(d, d + n, n * sizeof(uint32_t), Rccl::kChar, comm, ctx->stream) == 0 &&
             rccl().Reduce(d, d + 2 * n, n, Rccl::kUint32, Rccl::kSum, 0, comm, ctx->stream) == 0 &&
             gf_wait_stream(ctx->stream) == hipSuccess &&
             hipMemcpy(back.data(), d + n, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
        if (d) (void)hipFree(d);
        (void)rccl().CommDestroy(comm);
        for (size_t i = 0; i < n && ok; ++i) ok = back[i] == h[i] && back[n + i] == h[i];
        if (!ok) return fail(ctx, GF_ERR_HIP, "the one-rank all-gather / reduce did not reproduce its input");
        return GF_OK;
    } else {
        return fail(ctx, GF_ERR_INVALID, "unknown option '%s'", key);
    }
    ctx->chain.valid = false;
    return GF_OK;
}

int gf_shard_count(gf_ctx* ctx) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (ctx->group.empty() || ctx->g_shard_off) return 1;
    return (int)ctx->group.size();
}

int gf_generation(gf_ctx* ctx, uint64_t out[3]) {
    GF_DELEGATE(ctx, gf_generation(ctx, out));
    if (!ctx || !out) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    out[0] = ctx->snap_epoch;
    out[1] = ctx->cluster_gen;
    out[2] = ctx->usage_gen;
    return GF_OK;
}

int gf_chain_cache_stats(gf_ctx* ctx, int reset, uint64_t out[4]) {
    GF_DELEGATE(ctx, gf_chain_cache_stats(ctx, reset, out));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (out)
        for (int i = 0; i < 4; ++i) out[i] = ctx->chain_stat[i];
    if (reset)
        for (uint64_t& v : ctx->chain_stat) v = 0;
    return GF_OK;
}

int gf_hbm_probe(gf_ctx* ctx, uint64_t bytes, uint32_t iters, double* read_gb_per_s, double* copy_gb_per_s) {
    GF_DELEGATE(ctx, gf_hbm_probe(ctx, bytes, iters, read_gb_per_s, copy_gb_per_s));
    if (!ctx || bytes < 16 || iters == 0) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_HIP(ctx, hipSetDevice(ctx->device));
    bytes &= ~UINT64_C(15);
    void *src = nullptr, *dst = nullptr;
    GF_HIP(ctx, hipMalloc(&src, bytes));
    if (copy_gb_per_s && hipMalloc(&dst, bytes) != hipSuccess) {
        (void)hipFree(src);
        return fail(ctx, GF_ERR_HIP, "hipMalloc of the probe buffer failed");
    }
    int rc = GF_OK;
    float ms_read = 0.0f, ms_copy = 0.0f;
    uint32_t* sink = reinterpret_cast<uint32_t*>(ctx->d_stats.ptr);  // never written (see stream_read_kernel)
    do {
        if (hipMemsetAsync(src, 1, bytes, ctx->stream) != hipSuccess) { rc = GF_ERR_HIP; break; }
        if (read_gb_per_s) {
            if (gangfit::launch_stream_read(src, bytes, sink, ctx->stream) != hipSuccess) { rc = GF_ERR_HIP; break; }  // warm-up
            if (hipEventRecord(ctx->ev_begin, ctx->stream) != hipSuccess) { rc = GF_ERR_HIP; break; }
            for (uint32_t i = 0; i < iters && rc == GF_OK; ++i)
                if (gangfit::launch_stream_read(src, bytes, sink, ctx->stream) != hipSuccess) rc = GF_ERR_HIP;
            if (rc != GF_OK) break;
            if (hipEventRecord(ctx->ev_end, ctx->stream) != hipSuccess || gf_wait_event(ctx->ev_end) != hipSuccess ||
                hipEventElapsedTime(&ms_read, ctx->ev_begin, ctx->ev_end) != hipSuccess) { rc = GF_ERR_HIP; break; }
        }
        if (copy_gb_per_s) {
            if (gangfit::launch_stream_copy(src, dst, bytes, ctx->stream) != hipSuccess) { rc = GF_ERR_HIP; break; }  // warm-up
            if (hipEventRecord(ctx->ev_begin, ctx->stream) != hipSuccess) { rc = GF_ERR_HIP; break; }
            for (uint32_t i = 0; i < iters && rc == GF_OK; ++i)
                if (gangfit::launch_stream_copy(i & 1 ? dst : src, i & 1 ? src : dst, bytes, ctx->stream) != hipSuccess) rc = GF_ERR_HIP;
            if (rc != GF_OK) break;
            if (hipEventRecord(ctx->ev_end, ctx->stream) != hipSuccess || gf_wait_event(ctx->ev_end) != hipSuccess ||
                hipEventElapsedTime(&ms_copy, ctx->ev_begin, ctx->ev_end) != hipSuccess)
                rc = GF_ERR_HIP;
        }
    } while (false);
    (void)gf_wait_stream(ctx->stream);
    (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (rc != GF_OK) return fail(ctx, rc, "bandwidth probe failed");
    if (read_gb_per_s) *read_gb_per_s = ms_read > 0.0f ? (double)bytes * iters / ((double)ms_read * 1e-3) / 1e9 : 0.0;
    if (copy_gb_per_s) *copy_gb_per_s = ms_copy > 0.0f ? 2.0 * (double)bytes * iters / ((double)ms_copy * 1e-3) / 1e9 : 0.0;
    return GF_OK;
}

int gf_launch_floor(gf_ctx* ctx, void* stream, uint32_t iters, float* us_per_launch) {
    GF_DELEGATE(ctx, gf_launch_floor(ctx, stream, iters, us_per_launch));
    if (!ctx || !us_per_launch || iters == 0) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    for (int i = 0; i < 8; ++i) GF_HIP(ctx, gangfit::launch_empty(nullptr, st));
    GF_HIP(ctx, gf_wait_stream(st));
    GF_HIP(ctx, hipEventRecord(ctx->ev_begin, st));
    for (uint32_t i = 0; i < iters; ++i) GF_HIP(ctx, gangfit::launch_empty(nullptr, st));
    GF_HIP(ctx, hipEventRecord(ctx->ev_end, st));
    GF_HIP(ctx, gf_wait_event(ctx->ev_end));
    float ms = 0.0f;
    GF_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    *us_per_launch = ms * 1e3f / (float)iters;
    return GF_OK;
}

int gf_device_info_get(gf_ctx* ctx, gf_device_info* out) {
    if (!ctx || !out) return GF_ERR_INVALID;
    *out = ctx->info;
    return GF_OK;
}

namespace {
// After a device-side gf_snapshot_build the host mirrors of the snapshot are fetched only when something asks for them.
int materialize_host(gf_ctx* ctx) {
    if (!ctx->host_stale) return GF_OK;
    const size_t N = ctx->n_nodes;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, ctx->h_bcols.reserve(6 * N + 1));
    GF_HIP(ctx, ctx->h_border.reserve(N + 1));
    if (N) {
        GF_HIP(ctx, hipMemcpyAsync(ctx->h_bcols.ptr, ctx->d_node_tab.ptr, 6 * N * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        GF_HIP(ctx, hipMemcpyAsync(ctx->h_border.ptr, ctx->d_node_slot.ptr, N * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    GF_HIP(ctx, gf_wait_stream(ctx->stream));
    for (int j = 0; j < 3; ++j) {
        ctx->avail[j].assign(ctx->h_bcols.ptr + (size_t)j * N, ctx->h_bcols.ptr + (size_t)(j + 1) * N);
        ctx->sched[j].assign(ctx->h_bcols.ptr + (size_t)(3 + j) * N, ctx->h_bcols.ptr + (size_t)(4 + j) * N);
    }
    ctx->h_node_slot.assign(ctx->h_border.ptr, ctx->h_border.ptr + N);
    ctx->host_stale = false;
    return GF_OK;
}
}  // namespace

int gf_snapshot_set(gf_ctx* ctx, uint32_t n_nodes, const int64_t* avail_cpu_milli, const int64_t* avail_mem_bytes,
                    const int64_t* avail_gpu, const int64_t* sched_cpu_milli, const int64_t* sched_mem_bytes,
                    const int64_t* sched_gpu) {
    GF_EACH(ctx, gf_snapshot_set(ctx, n_nodes, avail_cpu_milli, avail_mem_bytes, avail_gpu, sched_cpu_milli, sched_mem_bytes, sched_gpu));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    InstallGuard install_guard(ctx);
    if (n_nodes > 0 && (!avail_cpu_milli || !avail_mem_bytes || !avail_gpu))
        return fail(ctx, GF_ERR_INVALID, "available arrays must not be NULL");
    if (n_nodes >= GF_NO_NODE) return fail(ctx, GF_ERR_INVALID, "too many nodes");
    const int64_t* av[3] = {avail_cpu_milli, avail_mem_bytes, avail_gpu};
    const int64_t* sc[3] = {sched_cpu_milli, sched_mem_bytes, sched_gpu};
    for (int j = 0; j < 3; ++j)
        for (uint32_t n = 0; n < n_nodes; ++n)
            if (av[j][n] >= GF_MAX_ABS_QUANTITY || av[j][n] <= -GF_MAX_ABS_QUANTITY)
                return fail(ctx, GF_ERR_INVALID, "available[%d][%u] outside (-2^62, 2^62)", j, n);
    ctx->have_sched = sc[0] && sc[1] && sc[2];
    if (ctx->have_sched)
        for (int j = 0; j < 3; ++j)
            for (uint32_t n = 0; n < n_nodes; ++n)
                if (sc[j][n] < 0 || sc[j][n] >= GF_MAX_ABS_QUANTITY)
                    return fail(ctx, GF_ERR_INVALID, "schedulable[%d][%u] outside [0, 2^62)", j, n);
    ctx->zone.clear();
    ctx->host_stale = false;
    for (int j = 0; j < 3; ++j) {
        ctx->avail[j].assign(av[j], av[j] + n_nodes);
        if (ctx->have_sched)
            ctx->sched[j].assign(sc[j], sc[j] + n_nodes);
        else
            ctx->sched[j].clear();
    }
    ctx->n_nodes = n_nodes;
    ctx->have_snapshot = true;
    ctx->have_orders = false;
    ctx->work_valid = false;
    ++ctx->snap_epoch;  // drops the chain cache
    // node-indexed copy for the per-node efficiency map (gf_packing_efficiencies)
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, gf_wait_stream(ctx->stream));
    GF_HIP(ctx, ctx->d_node_tab.reserve(6 * (size_t)n_nodes + 1));
    for (int j = 0; j < 3 && n_nodes; ++j) {
        GF_HIP(ctx, hipMemcpy(ctx->d_node_tab.ptr + (size_t)j * n_nodes, av[j], (size_t)n_nodes * sizeof(int64_t),
                              hipMemcpyHostToDevice));
        if (ctx->have_sched)
            GF_HIP(ctx, hipMemcpy(ctx->d_node_tab.ptr + (size_t)(3 + j) * n_nodes, sc[j],
                                  (size_t)n_nodes * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    return GF_OK;
}

int gf_zones_set(gf_ctx* ctx, const uint32_t* zone_of_node) {
    GF_EACH(ctx, gf_zones_set(ctx, zone_of_node));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    InstallGuard install_guard(ctx);
    if (!ctx->have_snapshot) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set must precede gf_zones_set");
    if (ctx->n_nodes > 0 && !zone_of_node) return fail(ctx, GF_ERR_INVALID, "zone array must not be NULL");
    ctx->zone.assign(zone_of_node, zone_of_node + ctx->n_nodes);
    ctx->have_orders = false;  // the zone views are built by gf_orders_set
    ++ctx->snap_epoch;
    return GF_OK;
}

int gf_orders_set(gf_ctx* ctx, const uint32_t* driver_order, uint32_t n_d, const uint32_t* exec_order, uint32_t n_x) {
    GF_EACH(ctx, gf_orders_set(ctx, driver_order, n_d, exec_order, n_x));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    InstallGuard install_guard(ctx);
    if (!ctx->have_snapshot) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set must precede gf_orders_set");
    if (int mrc = materialize_host(ctx); mrc != GF_OK) return mrc;
    if ((n_d > 0 && !driver_order) || (n_x > 0 && !exec_order))
        return fail(ctx, GF_ERR_INVALID, "order arrays must not be NULL");
    GF_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n_nodes = ctx->n_nodes;
    std::vector<uint32_t>& node_slot = ctx->h_node_slot;
    node_slot.assign(n_nodes, GF_NO_NODE);
    // ---- positions of the known nodes in the two orders.  Unknown names (index >= n_nodes) never host anything
    //      (binpack.go:68, pack_tightly.go:51, distribute_evenly.go:59) and a repeated driver candidate can only repeat
    //      the failure of its first occurrence, so both are dropped from the slot space without changing any result.
    std::vector<uint32_t> xpos(n_nodes, GF_NO_NODE), dpos(n_nodes, GF_NO_NODE);
    std::vector<uint32_t> xs, ds;
    xs.reserve(n_x);
    ds.reserve(n_d);
    for (uint32_t i = 0; i < n_x; ++i) {
        const uint32_t n = exec_order[i];
        if (n >= n_nodes) continue;
        if (xpos[n] != GF_NO_NODE)
            return fail(ctx, GF_ERR_INVALID, "node %u appears twice in the executor priority order", n);
        xpos[n] = (uint32_t)xs.size();
        xs.push_back(n);
    }
    bool d_has_unknown_or_dup = false;
    for (uint32_t i = 0; i < n_d; ++i) {
        const uint32_t n = driver_order[i];
        if (n >= n_nodes || dpos[n] != GF_NO_NODE) {
            d_has_unknown_or_dup = true;
            continue;
        }
        dpos[n] = (uint32_t)ds.size();
        ds.push_back(n);
    }
    (void)d_has_unknown_or_dup;
    // ---- merged layout: one order that has both (cleaned) orders as subsequences, if it exists
    std::vector<uint32_t> merged;
    std::vector<uint8_t> mflags;  // bit 0: executor candidate, bit 1: driver candidate
    bool mergeable = !ctx->force_general_layout;
    if (mergeable) {
        merged.reserve(xs.size() + ds.size());
        size_t i = 0, j = 0;
        while (i < ds.size() || j < xs.size()) {
            if (i < ds.size() && j < xs.size() && ds[i] == xs[j]) {
                merged.push_back(ds[i]);
                mflags.push_back(3);
                ++i;
                ++j;
            } else if (i < ds.size() && xpos[ds[i]] == GF_NO_NODE) {
                merged.push_back(ds[i++]);
                mflags.push_back(2);
            } else if (j < xs.size() && dpos[xs[j]] == GF_NO_NODE) {
                merged.push_back(xs[j++]);
                mflags.push_back(1);
            } else {  // two nodes present in both orders, in opposite relative order
                mergeable = false;
                break;
            }
        }
    }
    uint32_t n_slots, n_x_slots, n_d_pos;
    if (mergeable) {
        n_x_slots = n_d_pos = (uint32_t)merged.size();
        const uint64_t n_slots64 = (uint64_t)merged.size() + 1;
        if (n_slots64 >= GF_NO_NODE) return fail(ctx, GF_ERR_INVALID, "order vectors too long");
        n_slots = (uint32_t)n_slots64;
        for (uint32_t sl = 0; sl < merged.size(); ++sl) node_slot[merged[sl]] = sl;
    } else {
        // general layout: executor order (with its unknown names, which stay empty slots), then driver-only nodes
        for (uint32_t i = 0; i < n_x; ++i)
            if (exec_order[i] < n_nodes) node_slot[exec_order[i]] = i;
        uint32_t extra = 0;
        for (uint32_t i = 0; i < n_d; ++i) {
            const uint32_t n = driver_order[i];
            if (n < n_nodes && node_slot[n] == GF_NO_NODE) node_slot[n] = n_x + extra++;
        }
        const uint64_t n_slots64 = (uint64_t)n_x + extra + 1;
        if (n_slots64 >= GF_NO_NODE) return fail(ctx, GF_ERR_INVALID, "order vectors too long");
        n_slots = (uint32_t)n_slots64;
        n_x_slots = n_x;
        n_d_pos = n_d;
    }
    const uint32_t sentinel = n_slots - 1;
    const uint32_t n_chunks = (n_slots + 63) / 64;

    GF_HIP(ctx, ctx->h_table.reserve(3 * (size_t)n_slots));
    GF_HIP(ctx, ctx->h_index.reserve((size_t)n_slots + n_d_pos + n_nodes + 1));
    GF_HIP(ctx, ctx->h_masks.reserve(2 * (size_t)n_chunks));
    int64_t* tcpu = ctx->h_table.ptr;
    int64_t* tmem = tcpu + n_slots;
    int64_t* tgpu = tmem + n_slots;
    uint32_t* slot_node = ctx->h_index.ptr;
    uint32_t* dslot = slot_node + n_slots;
    uint32_t* nslot = dslot + n_d_pos;
    uint64_t* xmask = ctx->h_masks.ptr;
    uint64_t* dmask = xmask + n_chunks;
    for (uint32_t s = 0; s < n_slots; ++s) {
        tcpu[s] = tmem[s] = tgpu[s] = kSentinelAvail;
        slot_node[s] = GF_NO_NODE;
    }
    for (uint32_t c = 0; c < n_chunks; ++c) xmask[c] = dmask[c] = 0;
    for (uint32_t n = 0; n < n_nodes; ++n) {
        const uint32_t s = node_slot[n];
        nslot[n] = s;
        if (s == GF_NO_NODE) continue;
        slot_node[s] = n;
        tcpu[s] = ctx->avail[0][n];
        tmem[s] = ctx->avail[1][n];
        tgpu[s] = ctx->avail[2][n];
    }
    bool identity = true;
    if (mergeable) {
        for (uint32_t s = 0; s < merged.size(); ++s) {
            dslot[s] = s;
            if (mflags[s] & 1) xmask[s >> 6] |= 1ull << (s & 63);
            if (mflags[s] & 2) dmask[s >> 6] |= 1ull << (s & 63);
        }
    } else {
        for (uint32_t i = 0; i < n_d; ++i) {
            const uint32_t n = driver_order[i];
            dslot[i] = n < n_nodes ? node_slot[n] : sentinel;
        }
        identity = false;
        for (uint32_t i = 0; i < n_x; ++i)
            if (exec_order[i] < n_nodes) xmask[i >> 6] |= 1ull << (i & 63);
        for (uint32_t c = 0; c < n_chunks; ++c) dmask[c] = ~0ull;  // not consulted: positions go through dslot[]
    }
    ctx->d_identity = identity;

    // chunk-maxima index over all slots (see NodeTable::cmax)
    GF_HIP(ctx, ctx->h_cmax.reserve(3 * (size_t)n_chunks));
    {
        const int64_t* cols[3] = {tcpu, tmem, tgpu};
        for (int j = 0; j < 3; ++j)
            for (uint32_t c = 0; c < n_chunks; ++c) {
                int64_t m = INT64_MIN;
                const uint32_t hi = (c + 1) * 64 < n_slots ? (c + 1) * 64 : n_slots;
                for (uint32_t s2 = c * 64; s2 < hi; ++s2) m = cols[j][s2] > m ? cols[j][s2] : m;
                ctx->h_cmax.ptr[(size_t)j * n_chunks + c] = m;
            }
    }
    // narrow form: unit[j] = gcd of dimension j over the real slots; scaled magnitudes must stay below 2^30
    {
        const int64_t* cols[3] = {tcpu, tmem, tgpu};
        bool ok = true;
        for (int j = 0; j < 3; ++j) {
            uint64_t g = 0;
            for (uint32_t s2 = 0; s2 + 1 < n_slots; ++s2) {
                if (slot_node[s2] == GF_NO_NODE) continue;
                uint64_t v = (uint64_t)(cols[j][s2] < 0 ? -cols[j][s2] : cols[j][s2]);
                while (v) {  // Euclid
                    const uint64_t t = g % v;
                    g = v;
                    v = t;
                }
                if (g == 1) break;
            }
            ctx->unit[j] = g ? (int64_t)g : 1;
        }
        GF_HIP(ctx, ctx->h_ntable.reserve(3 * (size_t)n_slots + 3 * (size_t)n_chunks));
        int32_t* nt = ctx->h_ntable.ptr;
        int32_t* ncm = nt + 3 * (size_t)n_slots;
        for (int j = 0; j < 3 && ok; ++j) {
            ctx->nmax[j] = 0;
            for (uint32_t c = 0; c < n_chunks; ++c) ncm[(size_t)j * n_chunks + c] = INT32_MIN;
            for (uint32_t s2 = 0; s2 < n_slots; ++s2) {
                int32_t v32 = INT32_MIN / 2;  // sentinel / empty slot: never fits, never hosts
                if (s2 + 1 < n_slots && slot_node[s2] != GF_NO_NODE) {
                    const int64_t q = cols[j][s2] / ctx->unit[j];
                    if (q >= (INT64_C(1) << 30) || q <= -(INT64_C(1) << 30)) {
                        ok = false;
                        break;
                    }
                    v32 = (int32_t)q;
                    const int64_t mag = q < 0 ? -q : q;
                    if (mag > ctx->nmax[j]) ctx->nmax[j] = mag;
                }
                nt[(size_t)j * n_slots + s2] = v32;
                int32_t& m = ncm[(size_t)j * n_chunks + (s2 >> 6)];
                m = v32 > m ? v32 : m;
            }
        }
        ctx->narrow_ok = ok;
    }
    GF_HIP(ctx, gf_wait_stream(ctx->stream));  // nothing in flight may still read the old tables
    if (ctx->narrow_ok) {
        GF_HIP(ctx, ctx->d_nsnap.reserve(3 * (size_t)n_slots));
        GF_HIP(ctx, ctx->d_nwork.reserve(3 * (size_t)n_slots));
        GF_HIP(ctx, ctx->d_ncmax.reserve(3 * (size_t)n_chunks));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_nsnap.ptr, ctx->h_ntable.ptr, 3 * (size_t)n_slots * sizeof(int32_t),
                                   hipMemcpyHostToDevice, ctx->stream));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_ncmax.ptr, ctx->h_ntable.ptr + 3 * (size_t)n_slots,
                                   3 * (size_t)n_chunks * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    }
    GF_HIP(ctx, ctx->d_cmax.reserve(3 * (size_t)n_chunks));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_cmax.ptr, ctx->h_cmax.ptr, 3 * (size_t)n_chunks * sizeof(int64_t),
                               hipMemcpyHostToDevice, ctx->stream));
    ctx->n_chunks = n_chunks;
    GF_HIP(ctx, ctx->d_masks.reserve(2 * (size_t)n_chunks));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_masks.ptr, ctx->h_masks.ptr, 2 * (size_t)n_chunks * sizeof(uint64_t),
                               hipMemcpyHostToDevice, ctx->stream));
    GF_HIP(ctx, ctx->d_snap.reserve(3 * (size_t)n_slots));
    GF_HIP(ctx, ctx->d_work.reserve(3 * (size_t)n_slots));
    GF_HIP(ctx, ctx->d_slot_node.reserve(n_slots));
    GF_HIP(ctx, ctx->d_dslot.reserve(n_d_pos + 1));
    GF_HIP(ctx, ctx->d_node_slot.reserve(n_nodes + 1));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_snap.ptr, tcpu, 3 * (size_t)n_slots * sizeof(int64_t), hipMemcpyHostToDevice,
                               ctx->stream));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_slot_node.ptr, slot_node, (size_t)n_slots * sizeof(uint32_t),
                               hipMemcpyHostToDevice, ctx->stream));
    if (n_d_pos)
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_dslot.ptr, dslot, (size_t)n_d_pos * sizeof(uint32_t), hipMemcpyHostToDevice,
                                   ctx->stream));
    if (n_nodes)
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_node_slot.ptr, nslot, (size_t)n_nodes * sizeof(uint32_t),
                                   hipMemcpyHostToDevice, ctx->stream));
    // ---- sparse gpu view (gangfit::SparseTable): the executor candidates with a free gpu as a compact table of their own,
    //      when they are a minority of the order (merged layout only: the independent kernel's fast path)
    ctx->n_g = ctx->n_gpad = 0;
    if (mergeable && ctx->sparse_gpu) {
        uint32_t n_g = 0;
        for (uint32_t s2 = 0; s2 < merged.size(); ++s2)
            if ((mflags[s2] & 1) && tgpu[s2] > 0) ++n_g;
        if (n_g > 0 && (uint64_t)n_g * 4 <= merged.size()) {
            const uint32_t n_gpad = (n_g + 63u) / 64u * 64u, gch = n_gpad / 64u;
            GF_HIP(ctx, ctx->h_gtab.reserve(3 * (size_t)n_gpad + 3 * (size_t)gch));
            GF_HIP(ctx, ctx->h_gidx.reserve((size_t)n_gpad + n_slots));
            int64_t* g0 = ctx->h_gtab.ptr;
            int64_t* gmax = g0 + 3 * (size_t)n_gpad;
            uint32_t* gnode = ctx->h_gidx.ptr;
            uint32_t* gsub = gnode + n_gpad;
            for (uint32_t i = 0; i < 3 * n_gpad; ++i) g0[i] = kSentinelAvail;
            for (uint32_t i = 0; i < n_gpad; ++i) gnode[i] = GF_NO_NODE;
            for (uint32_t s2 = 0; s2 < n_slots; ++s2) gsub[s2] = GF_NO_NODE;
            uint32_t k = 0;
            for (uint32_t s2 = 0; s2 < merged.size(); ++s2)
                if ((mflags[s2] & 1) && tgpu[s2] > 0) {
                    g0[k] = tcpu[s2];
                    g0[n_gpad + k] = tmem[s2];
                    g0[2 * (size_t)n_gpad + k] = tgpu[s2];
                    gnode[k] = slot_node[s2];
                    gsub[s2] = k++;
                }
            for (int j = 0; j < 3; ++j)
                for (uint32_t c = 0; c < gch; ++c) {
                    int64_t m = INT64_MIN;
                    for (uint32_t i = c * 64; i < (c + 1) * 64; ++i) m = g0[(size_t)j * n_gpad + i] > m ? g0[(size_t)j * n_gpad + i] : m;
                    gmax[(size_t)j * gch + c] = m;
                }
            GF_HIP(ctx, ctx->d_gtab.reserve(3 * (size_t)n_gpad));
            GF_HIP(ctx, ctx->d_gcmax.reserve(3 * (size_t)gch));
            GF_HIP(ctx, ctx->d_gidx.reserve((size_t)n_gpad + n_slots));
            GF_HIP(ctx, ctx->d_gmask.reserve(gch));
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gtab.ptr, g0, 3 * (size_t)n_gpad * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gcmax.ptr, gmax, 3 * (size_t)gch * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gidx.ptr, gnode, ((size_t)n_gpad + n_slots) * sizeof(uint32_t), hipMemcpyHostToDevice,
                                       ctx->stream));
            std::vector<uint64_t> gm(gch, 0);
            for (uint32_t i = 0; i < n_g; ++i) gm[i >> 6] |= 1ull << (i & 63);
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gmask.ptr, gm.data(), gch * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
            GF_HIP(ctx, gf_wait_stream(ctx->stream));  // gm is a local
            ctx->n_g = n_g;
            ctx->n_gpad = n_gpad;
        }
    }
    // ---- SchedulableResources in slot order (efficiencies); empty slots read 0
    GF_HIP(ctx, ctx->d_sched.reserve(3 * (size_t)n_slots));
    if (ctx->have_sched) {
        // h_table is free again only after the snapshot copy above has completed
        GF_HIP(ctx, gf_wait_stream(ctx->stream));
        for (int j = 0; j < 3; ++j)
            for (uint32_t s2 = 0; s2 < n_slots; ++s2)
                ctx->h_table.ptr[(size_t)j * n_slots + s2] = slot_node[s2] == GF_NO_NODE ? 0 : ctx->sched[j][slot_node[s2]];
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_sched.ptr, ctx->h_table.ptr, 3 * (size_t)n_slots * sizeof(int64_t),
                                   hipMemcpyHostToDevice, ctx->stream));
    } else {
        GF_HIP(ctx, hipMemsetAsync(ctx->d_sched.ptr, 0, 3 * (size_t)n_slots * sizeof(int64_t), ctx->stream));
    }
    // ---- zone views (single_az.go:23-72): evaluation list = zones in order of first appearance in the driver order
    //      that own at least one executor candidate; per zone, candidate masks over the same slot table
    {
        auto zone_of = [&](uint32_t n) { return ctx->zone.empty() ? 0u : ctx->zone[n]; };
        std::vector<uint32_t> zlist;
        for (uint32_t n : ds) {
            const uint32_t z = zone_of(n);
            bool seen = false;
            for (uint32_t q : zlist) seen = seen || q == z;
            if (!seen) zlist.push_back(z);
        }
        std::vector<uint32_t> eval;
        for (uint32_t z : zlist) {
            bool has_x = false;
            for (uint32_t n : xs)
                if (zone_of(n) == z) {
                    has_x = true;
                    break;
                }
            if (has_x) eval.push_back(z);
        }
        const uint32_t d_words = (n_d_pos + 63) / 64;
        const uint32_t zstride = n_chunks > d_words ? n_chunks : d_words;
        const uint32_t nz = (uint32_t)eval.size();
        GF_HIP(ctx, ctx->h_zmasks.reserve(2 * (size_t)nz * zstride + 1));
        GF_HIP(ctx, ctx->d_zmasks.reserve(2 * (size_t)nz * zstride + 1));
        uint64_t* zx = ctx->h_zmasks.ptr;
        uint64_t* zd = zx + (size_t)nz * zstride;
        for (size_t i = 0; i < 2 * (size_t)nz * zstride; ++i) zx[i] = 0;
        for (uint32_t zi = 0; zi < nz; ++zi) {
            const uint32_t z = eval[zi];
            uint64_t* rx = zx + (size_t)zi * zstride;
            uint64_t* rd = zd + (size_t)zi * zstride;
            if (mergeable) {
                for (uint32_t s2 = 0; s2 < merged.size(); ++s2) {
                    if (zone_of(merged[s2]) != z) continue;
                    if (mflags[s2] & 1) rx[s2 >> 6] |= 1ull << (s2 & 63);
                    if (mflags[s2] & 2) rd[s2 >> 6] |= 1ull << (s2 & 63);
                }
            } else {
                for (uint32_t i = 0; i < n_x; ++i)
                    if (exec_order[i] < n_nodes && zone_of(exec_order[i]) == z) rx[i >> 6] |= 1ull << (i & 63);
                for (uint32_t i = 0; i < n_d; ++i)  // by driver POSITION (Orders::dpos_mask)
                    if (driver_order[i] < n_nodes && zone_of(driver_order[i]) == z) rd[i >> 6] |= 1ull << (i & 63);
            }
        }
        if (nz)
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_zmasks.ptr, zx, 2 * (size_t)nz * zstride * sizeof(uint64_t),
                                       hipMemcpyHostToDevice, ctx->stream));
        ctx->n_zones = nz;
        ctx->zstride = zstride;
        ctx->zd_row0 = nz;
    }
    GF_HIP(ctx, gf_wait_stream(ctx->stream));
    ctx->n_x = n_x_slots;
    ctx->n_d = n_d_pos;
    ctx->n_slots = n_slots;
    ctx->merged = mergeable;
    ctx->have_orders = true;
    ctx->work_valid = false;
    ++ctx->snap_epoch;
    return GF_OK;
}

int gf_fit_batch(gf_ctx* ctx, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* apps, gf_result* results,
                 uint32_t* exec_nodes, uint64_t exec_nodes_cap, int32_t* chain_failed_at) {
    if (ctx != nullptr && !ctx->group.empty())
        return group_fit_batch(ctx, mode, algo, n_apps, apps, results, exec_nodes, exec_nodes_cap, chain_failed_at);
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (n_apps > 0 && (!apps || !results)) return fail(ctx, GF_ERR_INVALID, "apps/results must not be NULL");
    if (chain_failed_at) *chain_failed_at = -1;
    if (n_apps == 0) return GF_OK;
    const auto t_entry = std::chrono::steady_clock::now();
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, ctx->h_apps.reserve(n_apps));
    uint64_t total_k = 0;
    for (uint32_t a = 0; a < n_apps; ++a) {
        const gf_app& in = apps[a];
        if (in.k < 0 || in.k > GF_MAX_K) return fail(ctx, GF_ERR_INVALID, "apps[%u].k = %d outside [0, %d]", a, in.k, GF_MAX_K);
        for (int j = 0; j < 3; ++j)
            if (in.drv[j] < 0 || in.drv[j] >= GF_MAX_ABS_QUANTITY || in.exe[j] < 0 || in.exe[j] >= GF_MAX_ABS_QUANTITY)
                return fail(ctx, GF_ERR_INVALID, "apps[%u] request outside [0, 2^62)", a);
        gf_app& o = ctx->h_apps.ptr[a];
        o = in;
        o.exec_off = total_k;
        total_k += (uint64_t)in.k;
    }
    if (total_k > exec_nodes_cap || (total_k > 0 && !exec_nodes))
        return fail(ctx, GF_ERR_CAPACITY, "exec_nodes holds %llu entries, %llu needed",
                    (unsigned long long)exec_nodes_cap, (unsigned long long)total_k);
    GF_HIP(ctx, ctx->d_apps.reserve(n_apps));
    GF_HIP(ctx, ctx->d_results.reserve(n_apps));
    GF_HIP(ctx, ctx->d_exec.reserve(total_k + 1));
    GF_HIP(ctx, ctx->h_results.reserve(n_apps));
    GF_HIP(ctx, ctx->h_exec.reserve(total_k + 1));
    hipStream_t st = ctx->stream;
    // Small independent batches of the plain packers skip the three staging copies: the kernel reads the app records from
    // the pinned staging buffer and writes results and placements straight into pinned host memory (posted PCIe writes,
    // visible when the kernel has completed).  A copy engine round trip costs more than the whole kernel at these sizes.
    if (ctx->zero_copy && mode == GF_MODE_INDEPENDENT && !is_zone_algo(algo) && ctx->have_orders &&
        (uint64_t)n_apps * sizeof(gf_app) + total_k * sizeof(uint32_t) <= (UINT64_C(4) << 20)) {
        void *da = ctx->h_apps.dev, *dr = ctx->h_results.dev, *de = ctx->h_exec.dev;
        if (da != nullptr && dr != nullptr && de != nullptr) {
            using clk = std::chrono::steady_clock;
            const auto t_staged = clk::now();
            // the launch announces its own completion in pinned memory (IndHostOut): what a 5 us kernel otherwise waits longest
            // for is the kernel-end release, the completion signal and the runtime's query
            gangfit::IndHostOut ho{};
            bool flagged = ctx->host_flag && !wait_blocking();
            if (flagged) {
                if (ctx->d_ind_done.ptr == nullptr) {
                    const size_t words = (size_t)(gangfit::kIndDoneCounters + 1) * gangfit::kIndDoneStride;
                    GF_HIP(ctx, ctx->d_ind_done.reserve(words));
                    GF_HIP(ctx, hipMemsetAsync(ctx->d_ind_done.ptr, 0, words * sizeof(uint32_t), st));
                    GF_HIP(ctx, ctx->h_ind_flag.reserve(8));
                    ctx->h_ind_flag.ptr[0] = 0;
                }
                flagged = ctx->h_ind_flag.dev != nullptr;
            }
            if (flagged) {
                ho.h_results = static_cast<gf_result*>(dr);
                ho.h_exec = static_cast<uint32_t*>(de);
                ho.counters = ctx->d_ind_done.ptr;
                ho.flag = ctx->h_ind_flag.dev;
                ho.seq = ++ctx->ind_seq;
            }
            const int rc0 = flagged ? launch(ctx, mode, algo, n_apps, ctx->h_apps.ptr, static_cast<const gf_app*>(da),
                                             ctx->d_results.ptr, ctx->d_exec.ptr, total_k, ctx->d_failed.ptr, st, nullptr, &ho)
                                    : launch(ctx, mode, algo, n_apps, ctx->h_apps.ptr, static_cast<const gf_app*>(da),
                                             static_cast<gf_result*>(dr), static_cast<uint32_t*>(de), total_k, ctx->d_failed.ptr, st);
            if (rc0 != GF_OK) return rc0;
            const auto t_launched = clk::now();
            bool seen = false;
            if (flagged) {
                const unsigned long long* f = ctx->h_ind_flag.ptr;
                for (uint32_t spins = 0;; ++spins) {
                    if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == ho.seq) {
                        seen = true;
                        break;
                    }
                    // (a launch that faults never writes the word: the stream wait below reports it)
                    if ((spins & 0x3FFu) == 0x3FFu && clk::now() - t_launched > std::chrono::milliseconds(5)) break;
                }
            }
            if (!seen) GF_HIP(ctx, gf_wait_stream(st));
            const auto t_done = clk::now();
            std::memcpy(results, ctx->h_results.ptr, (size_t)n_apps * sizeof(gf_result));
            if (total_k) std::memcpy(exec_nodes, ctx->h_exec.ptr, (size_t)total_k * sizeof(uint32_t));
            const auto t_out = clk::now();
            auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            ctx->call_phase_us[0] = us(t_entry, t_staged);
            ctx->call_phase_us[1] = us(t_staged, t_launched);
            ctx->call_phase_us[2] = us(t_launched, t_done);
            ctx->call_phase_us[3] = us(t_done, t_out);
            ctx->call_phase_us[4] = us(t_entry, t_out);
            return GF_OK;
        }
    }
    // ---- FIFO chains of the plain packers on the solo kernel: resume from the last chain's checkpoints where the queues agree
    ChainRun run;
    ctx->planned_units.valid = false;
    const bool use_cache = chain_plan(ctx, mode, algo, n_apps, ctx->h_apps.ptr, &run);
    const uint32_t a0 = run.a_begin;
    const uint64_t k0 = a0 > 0 ? ctx->h_apps.ptr[a0].exec_off : 0;  // placements of the skipped prefix
    // The answers travel to the pinned host buffers by posted writes of a kernel when the buffers are device-mapped: three
    // copy-engine transfers behind the last kernel are three hand-overs between the compute queue and a copy engine — a
    // visible part of a resumed chain, and what keeps chains on different streams from overlapping.  A FIFO chain goes
    // further: its first kernel reads the records from the pinned buffer and its last one writes the answers there
    // (gf_ctx::HostIo), which makes a Filter three launches and no copy.
    void *da = ctx->h_apps.dev, *dr = ctx->h_results.dev, *de = ctx->h_exec.dev, *df = ctx->h_failed.dev;
    const bool mapped = ctx->zero_copy && dr != nullptr && de != nullptr && df != nullptr;
    gf_ctx::HostIo& hio = ctx->hio;
    hio = gf_ctx::HostIo{};
    if (mapped && mode == GF_MODE_FIFO_CHAIN && da != nullptr) {
        hio.active = true;
        hio.n_apps = n_apps;
        hio.apps = static_cast<const gf_app*>(da);
        hio.results = static_cast<gf_result*>(dr);
        hio.exec = static_cast<uint32_t*>(de);
        hio.failed = static_cast<int32_t*>(df);
    } else {
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_apps.ptr + a0, ctx->h_apps.ptr + a0, (size_t)(n_apps - a0) * sizeof(gf_app),
                                   hipMemcpyHostToDevice, st));
    }
    const int rc = launch(ctx, mode, algo, n_apps, ctx->h_apps.ptr, ctx->d_apps.ptr, ctx->d_results.ptr, ctx->d_exec.ptr,
                          total_k, ctx->d_failed.ptr, st, use_cache ? &run : nullptr);
    const bool answers_sent = hio.active && hio.out_done;
    hio.active = false;
    ctx->planned_units.valid = false;
    if (rc != GF_OK) {
        ctx->chain.valid = false;
        return rc;
    }
    if (answers_sent) {
        // the chain's last kernel wrote results, placements and the abort index to the host buffers
    } else if (mapped) {
        gangfit::CopyOut co{};
        co.src[0] = reinterpret_cast<const uint32_t*>(ctx->d_results.ptr + a0);
        co.dst[0] = reinterpret_cast<uint32_t*>(static_cast<gf_result*>(dr) + a0);
        co.words[0] = (size_t)(n_apps - a0) * (sizeof(gf_result) / 4);
        co.src[1] = ctx->d_exec.ptr + k0;
        co.dst[1] = static_cast<uint32_t*>(de) + k0;
        co.words[1] = (size_t)(total_k - k0);
        co.src[2] = reinterpret_cast<const uint32_t*>(ctx->d_failed.ptr);
        co.dst[2] = static_cast<uint32_t*>(df);
        co.words[2] = mode == GF_MODE_FIFO_CHAIN ? 1 : 0;
        GF_HIP(ctx, gangfit::launch_copy_out(co, st));
    } else {
        (void)hipGetLastError();
        GF_HIP(ctx, hipMemcpyAsync(ctx->h_results.ptr + a0, ctx->d_results.ptr + a0, (size_t)(n_apps - a0) * sizeof(gf_result),
                                   hipMemcpyDeviceToHost, st));
        if (total_k > k0)
            GF_HIP(ctx, hipMemcpyAsync(ctx->h_exec.ptr + k0, ctx->d_exec.ptr + k0, (size_t)(total_k - k0) * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, st));
        if (mode == GF_MODE_FIFO_CHAIN)
            GF_HIP(ctx, hipMemcpyAsync(ctx->h_failed.ptr, ctx->d_failed.ptr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    const hipError_t we = gf_wait_stream(st);
    if (we != hipSuccess) {
        ctx->chain.valid = false;
        return fail(ctx, GF_ERR_HIP, "waiting for the batch failed: %s", hipGetErrorString(we));
    }
    int32_t failed_at = mode == GF_MODE_FIFO_CHAIN ? ctx->h_failed.ptr[0] : -1;
    if (a0 > 0) {  // the prefix the chain did not replay comes from the cache, straight to the caller; the kernel counted from a0
        std::memcpy(results, ctx->chain.results.data(), (size_t)a0 * sizeof(gf_result));
        if (k0) std::memcpy(exec_nodes, ctx->chain.exec.data(), (size_t)k0 * sizeof(uint32_t));
        if (failed_at >= 0) failed_at += (int32_t)a0;
    }
    if (use_cache) chain_commit(ctx, algo, n_apps, total_k, failed_at, run);
    std::memcpy(results + a0, ctx->h_results.ptr + a0, (size_t)(n_apps - a0) * sizeof(gf_result));
    if (total_k > k0) std::memcpy(exec_nodes + k0, ctx->h_exec.ptr + k0, (size_t)(total_k - k0) * sizeof(uint32_t));
    if (mode == GF_MODE_FIFO_CHAIN && chain_failed_at) *chain_failed_at = failed_at;
    return GF_OK;
}

int gf_fit_batch_dev(gf_ctx* ctx, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* d_apps,
                     gf_result* d_results, uint32_t* d_exec_nodes, uint64_t exec_nodes_len, int32_t* d_chain_failed_at,
                     void* stream) {
    GF_DELEGATE(ctx, gf_fit_batch_dev(ctx, mode, algo, n_apps, d_apps, d_results, d_exec_nodes, exec_nodes_len, d_chain_failed_at, stream));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)  // launch() grows buffers and flips state flags
    if (n_apps > 0 && (!d_apps || !d_results)) return fail(ctx, GF_ERR_INVALID, "device apps/results must not be NULL");
    if (mode == GF_MODE_FIFO_CHAIN && !d_chain_failed_at) d_chain_failed_at = ctx->d_failed.ptr;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    return launch(ctx, mode, algo, n_apps, nullptr, d_apps, d_results, d_exec_nodes, exec_nodes_len, d_chain_failed_at, st);
}

// ------------------------------------------------------------------------------------------------ resident worker
namespace {
constexpr uint32_t kRing = gangfit::kWorkerRing;

inline uint64_t host_load(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void host_store(unsigned long long* p, uint64_t v) { __atomic_store_n(p, (unsigned long long)v, __ATOMIC_RELEASE); }

int worker_alloc(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (w.allocated) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    void* hp = nullptr;
    GF_HIP(ctx, hipHostMalloc(&hp, sizeof(gangfit::WorkerHostCtl), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(hp, 0, sizeof(gangfit::WorkerHostCtl));
    void* hd = nullptr;
    if (hipHostGetDevicePointer(&hd, hp, 0) != hipSuccess) {
        (void)hipHostFree(hp);
        return fail(ctx, GF_ERR_HIP, "the worker's control block cannot be mapped to the device");
    }
    void* dp = nullptr;
    // (ordinary device memory: the relaxed agent-scope loads of the pollers are served by their XCD's L2 — 32 workgroups
    //  probing one line cost one miss per update and XCD; in fine-grained memory every probe of every workgroup went to the
    //  one memory channel that holds the line)
    if (hipMalloc(&dp, sizeof(gangfit::WorkerDevCtl)) != hipSuccess ||
        hipMemset(dp, 0, sizeof(gangfit::WorkerDevCtl)) != hipSuccess) {
        if (dp) (void)hipFree(dp);
        (void)hipHostFree(hp);
        return fail(ctx, GF_ERR_HIP, "the worker's device control block cannot be allocated");
    }
    // its own non-blocking stream.  (A stream with a CU mask — to keep compute units free for FIFO chains — was measured first:
    // its first window cost 10 ms and, depending on the context, every ticket 3.8 instead of 2.4 us.  The free CUs come from the
    // worker's shape instead: workgroups of sixteen wavefronts that fill a CU's registers, fewer of them than the device has CUs.)
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipFree(dp);
        (void)hipHostFree(hp);
        return fail(ctx, GF_ERR_HIP, "the worker's stream cannot be created");
    }
    w.h = static_cast<gangfit::WorkerHostCtl*>(hp);
    w.h_dev = static_cast<gangfit::WorkerHostCtl*>(hd);
    w.d = static_cast<gangfit::WorkerDevCtl*>(dp);
    w.stream = st;
    (void)hipEventCreate(&w.ev0);
    (void)hipEventCreate(&w.ev1);
    w.allocated = true;
    return GF_OK;
}

void worker_advance(gf_ctx::Worker& w);
int worker_launch(gf_ctx* ctx, gf_algo algo, uint64_t first_ticket);

// The launch has left the device (its stream is idle): duration between the two events around it, tickets it relayed.
void worker_finished(gf_ctx::Worker& w) {
    float ms = 0.0f;
    if (w.ev0 && w.ev1 && hipEventElapsedTime(&ms, w.ev0, w.ev1) == hipSuccess) {
        const uint64_t consumed = host_load(&w.h->consumed);
        w.last_ms = ms;
        w.last_tickets = consumed > w.launch_first ? consumed - w.launch_first : 0;
    }
    (void)hipGetLastError();
}

// Makes the launch on the device (if any) leave once it has relayed and served every ticket posted so far, and waits for that.
// The leader may have idled out (or been stopped by worker_wait_ticket's 5 s limit) just as the last tickets were posted: it
// then left with consumed < posted.  Those tickets are re-driven here, on the still-installed snapshot, before the caller —
// usually an install — may go on; a context whose tickets cannot be served any more forgets them instead of refusing every
// later call.
int worker_join(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (!w.running) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    for (int attempt = 0;; ++attempt) {
        host_store(&w.h->stop, w.posted + 2);  // "leave after ticket posted - 1" (gangfit_worker.inc)
        const hipError_t e = gf_wait_stream(w.stream);
        host_store(&w.h->stop, 0);
        w.running = false;
        if (e != hipSuccess) return fail(ctx, GF_ERR_HIP, "the worker did not leave the device: %s", hipGetErrorString(e));
        worker_finished(w);
        worker_advance(w);
        if (w.completed_upto == w.posted) return GF_OK;
        const uint64_t consumed = host_load(&w.h->consumed);
        if (attempt < 4 && w.algo >= 0 && consumed < w.posted && w.epoch == ctx->snap_epoch) {
            if (const int rc = worker_launch(ctx, (gf_algo)w.algo, consumed); rc != GF_OK) return rc;
            continue;
        }
        const uint64_t lost_lo = w.completed_upto, lost_hi = w.posted;
        w.completed_upto = w.posted;  // forget them: the ring is usable again (their callers were told, or never will wait)
        host_store(&w.h->consumed, w.posted);
        return fail(ctx, GF_ERR_HIP, "the worker left with tickets %llu .. %llu unserved (relayed %llu)", (unsigned long long)lost_lo,
                    (unsigned long long)lost_hi, (unsigned long long)consumed);
    }
}

// (Re)launches the worker for tickets >= first_ticket on the installed snapshot.
int worker_launch(gf_ctx* ctx, gf_algo algo, uint64_t first_ticket) {
    gf_ctx::Worker& w = ctx->worker;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    host_store(&w.h->state, 0);
    host_store(&w.h->stop, 0);
    gangfit::WorkerArgs a{};
    a.host = w.h_dev;
    a.dev = w.d;
    a.generation = w.launches + 1;
    a.first_ticket = first_ticket;
    a.idle_ticks = (unsigned long long)w.idle_us * 100ull;  // wall_clock64 ticks at 100 MHz
    a.scratch = w.scratch.ptr;
    a.scratch_stride = w.scratch_stride;
    // every workgroup must be resident at once (a group that waits for a CU would leave its tickets unserved while the others
    // spin), and sixteen CUs stay free for FIFO chains (a chain needs a whole CU: sixteen wavefronts, the LDS): a workgroup of
    // the worker fills a CU's registers, so it has a CU to itself and the count of workgroups is the count of CUs taken
    uint32_t sets = w.sets;
    {
        const uint32_t cus = (uint32_t)ctx->info.compute_units;
        int per_cu = 0;
        GF_HIP(ctx, gangfit::worker_blocks_per_cu(algo, &per_cu));
        if (per_cu < 1) return fail(ctx, GF_ERR_HIP, "the worker kernel does not fit a CU");
        const uint32_t room = cus > 32 ? cus - 16u : cus;  // (per_cu is 1 for the tightly-pack instance; never count on more)
        while (sets > 1 && 1u + sets * w.blocks_per_set > room) --sets;
        if (1u + sets * w.blocks_per_set > room) return fail(ctx, GF_ERR_INVALID, "worker_blocks_per_set does not fit the device");
    }
    a.sets = sets;
    a.blocks_per_set = w.blocks_per_set;
    a.stats = ctx->stats_on ? ctx->d_stats.ptr : nullptr;
    w.launch_first = first_ticket;
    if (w.ev0) (void)hipEventRecord(w.ev0, w.stream);
    GF_HIP(ctx, gangfit::launch_fit_worker(algo, make_table(ctx, ctx->d_snap.ptr), make_sparse(ctx), a, w.stream));
    if (w.ev1) (void)hipEventRecord(w.ev1, w.stream);
    w.running = true;
    w.algo = (int)algo;
    w.epoch = ctx->snap_epoch;
    ++w.launches;
    return GF_OK;
}

void worker_advance(gf_ctx::Worker& w) {
    while (w.completed_upto < w.posted && host_load(&w.h->done[w.completed_upto % kRing]) == w.completed_upto + 1) ++w.completed_upto;
}

// The leader leaves when no ticket has arrived for a while — possibly just as one was posted.  When it has left: the old
// launch is joined (its wavefronts work off everything it relayed first) and, if tickets were posted that it did not relay,
// the worker is launched again from the first of them.
int worker_revive(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (w.running) {
        if (host_load(&w.h->state) != 2) return GF_OK;
        GF_HIP(ctx, hipSetDevice(ctx->device));
        GF_HIP(ctx, gf_wait_stream(w.stream));
        w.running = false;
        worker_finished(w);
    }
    // not on the device: whatever was posted behind the last ticket the leader relayed needs a launch
    const uint64_t consumed = host_load(&w.h->consumed);
    if (w.algo >= 0 && consumed < w.posted) {
        if (w.epoch != ctx->snap_epoch) return fail(ctx, GF_ERR_STATE, "the snapshot changed under a posted ticket");
        return worker_launch(ctx, (gf_algo)w.algo, consumed);
    }
    return GF_OK;
}

// Waits for ticket t (t < posted).
int worker_wait_ticket(gf_ctx* ctx, uint64_t t) {
    gf_ctx::Worker& w = ctx->worker;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    for (;;) {
        if (t < w.completed_upto || host_load(&w.h->done[t % kRing]) == t + 1) return GF_OK;
        if (wait_blocking() && (spins & 0x7u) == 0x7u)  // GANGFIT_WAIT=block: the host cannot spare the core for the wait
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        if ((++spins & 0x3Fu) == 0) {
            if (const int rc = worker_revive(ctx); rc != GF_OK) return rc;
            if (!w.running && host_load(&w.h->done[t % kRing]) != t + 1)
                return fail(ctx, GF_ERR_STATE, "ticket %llu was never served (posted %llu, doorbell %llu, relayed %llu, complete below %llu, "
                            "completion word %llu, launches %llu)", (unsigned long long)t, (unsigned long long)w.posted,
                            (unsigned long long)host_load(&w.h->posted), (unsigned long long)host_load(&w.h->consumed),
                            (unsigned long long)w.completed_upto, (unsigned long long)host_load(&w.h->done[t % kRing]),
                            (unsigned long long)w.launches);
            if ((spins & 0xFFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                host_store(&w.h->stop, 1);
                return fail(ctx, GF_ERR_HIP, "the worker did not complete ticket %llu within 5 s", (unsigned long long)t);
            }
        }
    }
}

int worker_drain(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    worker_advance(w);
    for (uint64_t t = w.completed_upto; t < w.posted; ++t)
        if (const int rc = worker_wait_ticket(ctx, t); rc != GF_OK) return rc;
    worker_advance(w);
    return GF_OK;
}

// need_launch (nullable): instead of launching, report that a launch for tickets >= posted is needed — the caller posts its
// tickets first, so that the leader finds them at its first look (gf_worker_submit_dev).
int worker_prepare(gf_ctx* ctx, gf_algo algo, uint64_t max_total_k, bool* need_launch = nullptr) {
    gf_ctx::Worker& w = ctx->worker;
    if (!ctx->group.empty() || ctx->view_of != nullptr)
        return fail(ctx, GF_ERR_UNSUPPORTED, "the resident worker serves plain contexts (no views, one device)");
    if (algo != GF_ALGO_TIGHTLY_PACK && algo != GF_ALGO_DISTRIBUTE_EVENLY && algo != GF_ALGO_MINIMAL_FRAGMENTATION)
        return fail(ctx, GF_ERR_UNSUPPORTED, "the resident worker serves the plain packers");
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede a fit");
    if (const int rc = worker_alloc(ctx); rc != GF_OK) return rc;
    if (const int rc = worker_revive(ctx); rc != GF_OK) return rc;
    const bool grow = max_total_k + 1 > w.scratch_stride;
    if (w.running && (w.algo != (int)algo || w.epoch != ctx->snap_epoch || grow)) {
        // another packer, another snapshot or a larger scratch: everything posted is served first, then the worker leaves
        if (const int rc = worker_join(ctx); rc != GF_OK) return rc;
    }
    if (grow) {
        if (const int rc = worker_drain(ctx); rc != GF_OK) return rc;
        uint64_t stride = w.scratch_stride ? w.scratch_stride : 1024;
        while (stride < max_total_k + 1) stride *= 2;
        GF_HIP(ctx, hipSetDevice(ctx->device));
        GF_HIP(ctx, w.scratch.reserve((size_t)kRing * 3 * stride));
        w.scratch_stride = stride;
    }
    // not running: every ticket posted so far was relayed and served (worker_revive re-drives the ones that were not)
    if (need_launch) *need_launch = !w.running;
    if (!w.running && !need_launch) return worker_launch(ctx, algo, w.posted);
    return GF_OK;
}

// Posts one ticket (the caller has made room in the ring).
void worker_post(gf_ctx::Worker& w, uint32_t n_apps, const gf_app* apps, gf_result* results, uint32_t* exec_nodes, uint64_t exec_len,
                 bool host_out) {
    gangfit::WorkerTicket& tk = w.h->ring[w.posted % kRing];
    const unsigned long long tag = gangfit::worker_tag(w.posted) << 48;
    tk.word[1] = (unsigned long long)reinterpret_cast<uintptr_t>(apps) | tag;
    tk.word[2] = (unsigned long long)reinterpret_cast<uintptr_t>(results) | tag;
    tk.word[3] = (unsigned long long)reinterpret_cast<uintptr_t>(exec_nodes) | tag;
    tk.word[4] = (unsigned long long)exec_len | tag;
    tk.word[5] = (unsigned long long)n_apps | ((unsigned long long)(host_out ? 1u : 0u) << 32) | tag;
    tk.word[0] = w.posted + 1;
    ++w.posted;
}
void worker_quiesce(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (!w.allocated) return;
    if (worker_revive(ctx) != GF_OK) return;  // (it may have left for lack of work just as tickets were posted)
    if (w.running)
        (void)worker_join(ctx);
    else
        (void)worker_drain(ctx);
}
}  // namespace

int gf_worker_submit_dev(gf_ctx* ctx, gf_algo algo, uint32_t n_batches, const gf_worker_batch* batches, uint64_t* first_ticket) {
    if (!ctx || (n_batches > 0 && !batches)) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    uint64_t max_k = 0;
    for (uint32_t i = 0; i < n_batches; ++i) {
        if (batches[i].n_apps == 0 || !batches[i].d_apps || !batches[i].d_results)
            return fail(ctx, GF_ERR_INVALID, "batch %u: empty, or apps / results NULL", i);
        if (batches[i].exec_nodes_len > max_k) max_k = batches[i].exec_nodes_len;
    }
    bool need_launch = false;
    if (const int rc = worker_prepare(ctx, algo, max_k, &need_launch); rc != GF_OK) return rc;
    gf_ctx::Worker& w = ctx->worker;
    const uint64_t first = w.posted;
    if (first_ticket) *first_ticket = first;
    for (uint32_t i = 0; i < n_batches; ++i) {
        if (w.posted - w.completed_upto >= kRing) {  // the slot of ticket `posted` is free once ticket posted - ring is done
            host_store(&w.h->posted, w.posted);      // (ring the doorbell for what has been written so far)
            if (need_launch) {
                need_launch = false;
                if (const int rc = worker_launch(ctx, algo, first); rc != GF_OK) return rc;
            }
            if (const int rc = worker_wait_ticket(ctx, w.posted - kRing); rc != GF_OK) return rc;
            worker_advance(w);
        }
        const gf_worker_batch& b = batches[i];
        worker_post(w, b.n_apps, b.d_apps, b.d_results, b.d_exec_nodes, b.exec_nodes_len, (b.flags & GF_WORKER_HOST_OUTPUTS) != 0);
    }
    host_store(&w.h->posted, w.posted);  // the doorbell: one word for the whole group
    if (need_launch) return worker_launch(ctx, algo, first);
    return GF_OK;
}

int gf_worker_wait(gf_ctx* ctx, uint64_t first_ticket, uint32_t n_tickets) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    gf_ctx::Worker& w = ctx->worker;
    if (!w.allocated || first_ticket + n_tickets > w.posted) return fail(ctx, GF_ERR_INVALID, "tickets that were never posted");
    for (uint64_t t = first_ticket; t < first_ticket + n_tickets; ++t)
        if (const int rc = worker_wait_ticket(ctx, t); rc != GF_OK) return rc;
    worker_advance(w);
    return GF_OK;
}

int gf_worker_fit(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* apps, gf_result* results, uint32_t* exec_nodes,
                  uint64_t exec_nodes_cap) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (n_apps > 0 && (!apps || !results)) return fail(ctx, GF_ERR_INVALID, "apps/results must not be NULL");
    if (n_apps == 0) return GF_OK;
    uint64_t total_k = 0;
    for (uint32_t a = 0; a < n_apps; ++a) {
        const gf_app& in = apps[a];
        if (in.k < 0 || in.k > GF_MAX_K) return fail(ctx, GF_ERR_INVALID, "apps[%u].k = %d outside [0, %d]", a, in.k, GF_MAX_K);
        for (int j = 0; j < 3; ++j)
            if (in.drv[j] < 0 || in.drv[j] >= GF_MAX_ABS_QUANTITY || in.exe[j] < 0 || in.exe[j] >= GF_MAX_ABS_QUANTITY)
                return fail(ctx, GF_ERR_INVALID, "apps[%u] request outside [0, 2^62)", a);
        total_k += (uint64_t)in.k;
    }
    if (total_k > exec_nodes_cap || (total_k > 0 && !exec_nodes))
        return fail(ctx, GF_ERR_CAPACITY, "exec_nodes holds %llu entries, %llu needed", (unsigned long long)exec_nodes_cap,
                    (unsigned long long)total_k);
    if (const int rc = worker_prepare(ctx, algo, total_k); rc != GF_OK) return rc;
    gf_ctx::Worker& w = ctx->worker;
    // one pinned slice per ring slot: records in, results and placements out — the device reads and writes them in place
    if (n_apps > w.stage_apps || total_k + 1 > w.stage_k) {
        if (const int rc = worker_drain(ctx); rc != GF_OK) return rc;
        size_t na = w.stage_apps ? w.stage_apps : 1024, nk = w.stage_k ? w.stage_k : 16384;
        while (na < n_apps) na *= 2;
        while (nk < total_k + 1) nk *= 2;
        const size_t slice = na * (sizeof(gf_app) + sizeof(gf_result)) + nk * sizeof(uint32_t);
        void *hp = nullptr, *hd = nullptr;
        GF_HIP(ctx, hipSetDevice(ctx->device));
        GF_HIP(ctx, hipHostMalloc(&hp, slice * kRing, hipHostMallocMapped | hipHostMallocCoherent));
        if (hipHostGetDevicePointer(&hd, hp, 0) != hipSuccess) {
            (void)hipHostFree(hp);
            return fail(ctx, GF_ERR_HIP, "the worker's staging cannot be mapped to the device");
        }
        if (w.stage) (void)hipHostFree(w.stage);
        w.stage = hp;
        w.stage_dev = hd;
        w.stage_apps = na;
        w.stage_k = nk;
    }
    if (w.posted - w.completed_upto >= kRing) {
        if (const int rc = worker_wait_ticket(ctx, w.posted - kRing); rc != GF_OK) return rc;
        worker_advance(w);
    }
    const size_t slice = w.stage_apps * (sizeof(gf_app) + sizeof(gf_result)) + w.stage_k * sizeof(uint32_t);
    const size_t off = (size_t)(w.posted % kRing) * slice;
    char* hb = static_cast<char*>(w.stage) + off;
    char* db = static_cast<char*>(w.stage_dev) + off;
    gf_app* h_apps = reinterpret_cast<gf_app*>(hb);
    gf_result* h_res = reinterpret_cast<gf_result*>(hb + w.stage_apps * sizeof(gf_app));
    uint32_t* h_exec = reinterpret_cast<uint32_t*>(hb + w.stage_apps * (sizeof(gf_app) + sizeof(gf_result)));
    uint64_t k_off = 0;
    for (uint32_t a = 0; a < n_apps; ++a) {
        h_apps[a] = apps[a];
        h_apps[a].exec_off = k_off;
        k_off += (uint64_t)apps[a].k;
    }
    const uint64_t ticket = w.posted;
    worker_post(w, n_apps, reinterpret_cast<const gf_app*>(db), reinterpret_cast<gf_result*>(db + w.stage_apps * sizeof(gf_app)),
                reinterpret_cast<uint32_t*>(db + w.stage_apps * (sizeof(gf_app) + sizeof(gf_result))), total_k, true);
    host_store(&w.h->posted, w.posted);
    if (const int rc = worker_wait_ticket(ctx, ticket); rc != GF_OK) return rc;
    worker_advance(w);
    std::memcpy(results, h_res, (size_t)n_apps * sizeof(gf_result));
    if (total_k) std::memcpy(exec_nodes, h_exec, (size_t)total_k * sizeof(uint32_t));
    return GF_OK;
}

int gf_worker_stop(gf_ctx* ctx) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    worker_quiesce(ctx);
    return GF_OK;
}

int gf_worker_stats(gf_ctx* ctx, uint64_t out[4]) {
    if (!ctx || !out) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const gf_ctx::Worker& w = ctx->worker;
    out[0] = w.posted;
    out[1] = w.completed_upto;
    out[2] = w.launches;
    out[3] = (w.allocated && w.running && host_load(&w.h->state) != 2) ? 1 : 0;
    return GF_OK;
}

int gf_call_phases(gf_ctx* ctx, double out_us[5]) {
    GF_DELEGATE(ctx, gf_call_phases(ctx, out_us));
    if (!ctx || !out_us) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    for (int i = 0; i < 5; ++i) out_us[i] = ctx->call_phase_us[i];
    return GF_OK;
}

int gf_worker_kernel_time(gf_ctx* ctx, float* ms, uint64_t* tickets) {
    if (!ctx || !ms || !tickets) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const gf_ctx::Worker& w = ctx->worker;
    if (!w.allocated || w.launches == 0 || (w.running && w.launches == 1))
        return fail(ctx, GF_ERR_STATE, "no launch of the worker has finished yet (gf_worker_stop first)");
    *ms = w.last_ms;
    *tickets = w.last_tickets;
    return GF_OK;
}

int gf_graph_begin(gf_ctx* ctx, void* stream) {
    GF_DELEGATE(ctx, gf_graph_begin(ctx, stream));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    GF_HIP(ctx, gf_wait_stream(st));
    GF_HIP(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    return GF_OK;
}

namespace {
// A recorded sequence names device buffers by address: it is only replayable while the snapshot / orders it was recorded
// on are the installed ones (their buffers may be reallocated by the next install).
struct RecordedGraph {
    hipGraphExec_t exec = nullptr;
    uint64_t epoch = 0;
};
}  // namespace

int gf_graph_end(gf_ctx* ctx, void* stream, void** graph_out) {
    GF_DELEGATE(ctx, gf_graph_end(ctx, stream, graph_out));
    if (!ctx || !graph_out) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    *graph_out = nullptr;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    hipGraph_t graph = nullptr;
    GF_HIP(ctx, hipStreamEndCapture(st, &graph));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(ctx, GF_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    RecordedGraph* rg = new (std::nothrow) RecordedGraph();
    if (!rg) {
        (void)hipGraphExecDestroy(exec);
        return fail(ctx, GF_ERR_HIP, "out of memory");
    }
    rg->exec = exec;
    rg->epoch = ctx->snap_epoch;
    *graph_out = rg;
    return GF_OK;
}

int gf_graph_launch(gf_ctx* ctx, void* graph, void* stream) {
    GF_DELEGATE(ctx, gf_graph_launch(ctx, graph, stream));
    if (!ctx || !graph) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const RecordedGraph* rg = static_cast<const RecordedGraph*>(graph);
    if (rg->epoch != ctx->snap_epoch)
        return fail(ctx, GF_ERR_STATE, "the snapshot / orders changed since the sequence was recorded: record it again");
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    GF_HIP(ctx, hipGraphLaunch(rg->exec, st));
    return GF_OK;
}

void gf_graph_destroy(gf_ctx* ctx, void* graph) {
    (void)ctx;
    if (!graph) return;
    RecordedGraph* rg = static_cast<RecordedGraph*>(graph);
    if (rg->exec) (void)hipGraphExecDestroy(rg->exec);
    delete rg;
}

int gf_spark_binpack(gf_ctx* ctx, gf_algo algo, const gf_app* app, gf_result* result, uint32_t* exec_nodes,
                     uint64_t exec_nodes_cap) {
    return gf_fit_batch(ctx, GF_MODE_INDEPENDENT, algo, 1, app, result, exec_nodes, exec_nodes_cap, nullptr);
}

int gf_cluster_set(gf_ctx* ctx, uint32_t n_nodes, const int64_t* alloc_cpu_milli, const int64_t* alloc_mem_bytes,
                   const int64_t* alloc_gpu, const int64_t* over_cpu_milli, const int64_t* over_mem_bytes,
                   const int64_t* over_gpu, const uint32_t* node_flags, const uint32_t* zone_of_node, uint32_t n_zones,
                   const uint32_t* name_rank) {
    GF_EACH(ctx, gf_cluster_set(ctx, n_nodes, alloc_cpu_milli, alloc_mem_bytes, alloc_gpu, over_cpu_milli, over_mem_bytes,
                                over_gpu, node_flags, zone_of_node, n_zones, name_rank));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    ctx->have_cluster = false;
    const uint32_t n = n_nodes;
    if (n >= GF_NO_NODE) return fail(ctx, GF_ERR_INVALID, "too many nodes");
    if (n > 0 && (!alloc_cpu_milli || !alloc_mem_bytes || !alloc_gpu || !node_flags || !name_rank))
        return fail(ctx, GF_ERR_INVALID, "allocatable / node_flags / name_rank must not be NULL");
    const bool with_over = over_cpu_milli || over_mem_bytes || over_gpu;
    if (with_over && !(over_cpu_milli && over_mem_bytes && over_gpu))
        return fail(ctx, GF_ERR_INVALID, "overhead columns must be all NULL or all set");
    if (zone_of_node == nullptr) n_zones = 1;
    if (n_zones == 0 || n_zones > 4096) return fail(ctx, GF_ERR_INVALID, "n_zones = %u outside [1, 4096]", n_zones);
    {  // name_rank must be a permutation: it seeds the stable sort with the name order (nodesorting.go:92)
        std::vector<uint8_t> seen(n, 0);
        for (uint32_t i = 0; i < n; ++i) {
            if (name_rank[i] >= n || seen[name_rank[i]]) return fail(ctx, GF_ERR_INVALID, "name_rank is not a permutation");
            seen[name_rank[i]] = 1;
        }
        if (zone_of_node)
            for (uint32_t i = 0; i < n; ++i)
                if (zone_of_node[i] >= n_zones) return fail(ctx, GF_ERR_INVALID, "zone_of_node[%u] >= n_zones", i);
    }
    const int64_t* cols[3] = {alloc_cpu_milli, alloc_mem_bytes, alloc_gpu};
    const int64_t* ocols[3] = {over_cpu_milli, over_mem_bytes, over_gpu};
    const int64_t lim = GF_MAX_ABS_QUANTITY >> 1;
    for (int j = 0; j < 3; ++j) {
        ctx->cl_max_over[j] = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (cols[j][i] < 0 || cols[j][i] >= GF_MAX_ABS_QUANTITY || (with_over && (ocols[j][i] < 0 || ocols[j][i] >= lim)))
                return fail(ctx, GF_ERR_INVALID, "allocatable / overhead value out of range at node %u", i);
            if (with_over && ocols[j][i] > ctx->cl_max_over[j]) ctx->cl_max_over[j] = ocols[j][i];
        }
    }
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t N = n;
    GF_HIP(ctx, gf_wait_stream(st));  // nothing in flight may still read the columns that are about to be replaced
    GF_HIP(ctx, ctx->d_cl_i64.reserve(6 * N + 1));
    GF_HIP(ctx, ctx->d_cl_u32.reserve(3 * N + 1));
    for (int j = 0; j < 3 && N; ++j) {
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_cl_i64.ptr + j * N, cols[j], N * sizeof(int64_t), hipMemcpyHostToDevice, st));
        if (with_over)
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_cl_i64.ptr + (3 + j) * N, ocols[j], N * sizeof(int64_t), hipMemcpyHostToDevice, st));
    }
    if (N) {
        if (zone_of_node)
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_cl_u32.ptr, zone_of_node, N * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        else
            GF_HIP(ctx, hipMemsetAsync(ctx->d_cl_u32.ptr, 0, N * sizeof(uint32_t), st));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_cl_u32.ptr + N, name_rank, N * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_cl_u32.ptr + 2 * N, node_flags, N * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    GF_HIP(ctx, ctx->d_cl_usage.reserve(3 * N + 1));
    GF_HIP(ctx, hipMemsetAsync(ctx->d_cl_usage.ptr, 0, (3 * N + 1) * sizeof(int64_t), st));  // a new node set: no usage yet
    for (int j = 0; j < 3; ++j) ctx->usage_total[j] = 0;
    GF_HIP(ctx, gf_wait_stream(st));  // the caller's arrays are free again
    ctx->cl_flags.assign(node_flags, node_flags + n);
    ctx->cl_default_flags = ctx->cl_flags;
    ctx->d_flags_default = true;
    ctx->usage_ok = true;
    ++ctx->cluster_gen;
    ++ctx->usage_gen;
    if (zone_of_node)
        ctx->cl_zone.assign(zone_of_node, zone_of_node + n);
    else
        ctx->cl_zone.clear();
    ctx->cl_n = n;
    ctx->cl_zones = n_zones;
    ctx->cl_over = with_over;
    ctx->have_cluster = true;
    return GF_OK;
}

int gf_snapshot_build(gf_ctx* ctx, uint32_t n_nodes, const int64_t* alloc_cpu_milli, const int64_t* alloc_mem_bytes,
                      const int64_t* alloc_gpu, const int64_t* over_cpu_milli, const int64_t* over_mem_bytes,
                      const int64_t* over_gpu, uint32_t n_res, const uint32_t* res_node, const int64_t* res_cpu_milli,
                      const int64_t* res_mem_bytes, const int64_t* res_gpu, const uint32_t* node_flags,
                      const uint32_t* zone_of_node, uint32_t n_zones, const uint32_t* name_rank,
                      const uint32_t* driver_label_rank, const uint32_t* exec_label_rank, uint32_t* driver_order_out,
                      uint32_t* n_d_out, uint32_t* exec_order_out, uint32_t* n_x_out) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);  // cluster + build are one sequence
    const int rc = gf_cluster_set(ctx, n_nodes, alloc_cpu_milli, alloc_mem_bytes, alloc_gpu, over_cpu_milli, over_mem_bytes,
                                  over_gpu, node_flags, zone_of_node, n_zones, name_rank);
    if (rc != GF_OK) return rc;
    return gf_snapshot_build_resident(ctx, n_res, res_node, res_cpu_milli, res_mem_bytes, res_gpu, nullptr, driver_label_rank,
                                      exec_label_rank, driver_order_out, n_d_out, exec_order_out, n_x_out);
}

int gf_usage_reset(gf_ctx* ctx) {
    GF_EACH(ctx, gf_usage_reset(ctx));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    if (!ctx->have_cluster) return fail(ctx, GF_ERR_STATE, "gf_cluster_set must precede gf_usage_reset");
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, hipMemsetAsync(ctx->d_cl_usage.ptr, 0, (3 * (size_t)ctx->cl_n + 1) * sizeof(int64_t), ctx->stream));
    for (int j = 0; j < 3; ++j) ctx->usage_total[j] = 0;
    ctx->usage_ok = true;
    ++ctx->usage_gen;
    return GF_OK;
}

int gf_usage_apply(gf_ctx* ctx, uint32_t n_entries, const uint32_t* res_node, const int64_t* res_cpu_milli,
                   const int64_t* res_mem_bytes, const int64_t* res_gpu, int sign) {
    if (ctx != nullptr && !ctx->group.empty()) {
        // every device keeps the same sums; an update that reaches some devices and fails on another leaves them apart:
        // the resident usage is then unusable everywhere until gf_usage_reset
        gf_ctx* const g = ctx;
        std::lock_guard<std::recursive_mutex> glock(g->mu);
        for (size_t i = 0; i < g->group.size(); ++i) {
            const int rc = gf_usage_apply(g->group[i], n_entries, res_node, res_cpu_milli, res_mem_bytes, res_gpu, sign);
            if (rc != GF_OK) {
                g->err = g->group[i]->err;
                if (i > 0)
                    for (gf_ctx* sub : g->group) sub->usage_ok = false;
                return rc;
            }
        }
        return GF_OK;
    }
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    if (!ctx->have_cluster) return fail(ctx, GF_ERR_STATE, "gf_cluster_set must precede gf_usage_apply");
    if (!ctx->usage_ok) return fail(ctx, GF_ERR_STATE, "an earlier update failed half way: gf_usage_reset must rebuild the resident usage");
    if (sign != 1 && sign != -1) return fail(ctx, GF_ERR_INVALID, "sign must be +1 or -1");
    if (n_entries == 0) return GF_OK;
    if (!res_node || !res_cpu_milli || !res_mem_bytes || !res_gpu) return fail(ctx, GF_ERR_INVALID, "entry columns must not be NULL");
    const int64_t* rcols[3] = {res_cpu_milli, res_mem_bytes, res_gpu};
    const int64_t lim = GF_MAX_ABS_QUANTITY >> 1;
    __int128 total[3];
    for (int j = 0; j < 3; ++j) {
        __int128 sum = 0;
        for (uint32_t i = 0; i < n_entries; ++i) {
            if (rcols[j][i] < 0 || rcols[j][i] >= lim) return fail(ctx, GF_ERR_INVALID, "entry %u out of range", i);
            if (res_node[i] < ctx->cl_n) sum += rcols[j][i];
        }
        total[j] = ctx->usage_total[j] + (sign > 0 ? sum : -sum);
        // every node's sum lies between 0 and the sum of everything applied: that (plus the overhead) must stay below 2^62
        if (total[j] < 0) return fail(ctx, GF_ERR_INVALID, "more usage removed than was ever added (dimension %d)", j);
        if (total[j] + (__int128)ctx->cl_max_over[j] >= (__int128)GF_MAX_ABS_QUANTITY)
            return fail(ctx, GF_ERR_INVALID, "the resident usage can sum past 2^62: not representable");
    }
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t R = n_entries;
    GF_HIP(ctx, gf_wait_stream(st));  // an earlier update may still read the staging buffers that are about to grow
    GF_HIP(ctx, ctx->d_delta_i64.reserve(3 * R));
    GF_HIP(ctx, ctx->d_delta_u32.reserve(R));
    for (int j = 0; j < 3; ++j)
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_delta_i64.ptr + j * R, rcols[j], R * sizeof(int64_t), hipMemcpyHostToDevice, st));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_delta_u32.ptr, res_node, R * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GF_HIP(ctx, ctx->d_flag32.reserve(1));
    if (sign < 0) GF_HIP(ctx, hipMemsetAsync(ctx->d_flag32.ptr, 0, sizeof(uint32_t), st));
    ++ctx->usage_gen;
    ctx->usage_ok = false;  // until the update is known to have been applied in full
    GF_HIP(ctx, gangfit::launch_usage_apply(n_entries, ctx->cl_n, ctx->d_delta_u32.ptr, ctx->d_delta_i64.ptr, sign,
                                            ctx->d_cl_usage.ptr, ctx->d_flag32.ptr, st));
    if (sign < 0)
        GF_HIP(ctx, hipMemcpyAsync(ctx->h_failed.ptr, ctx->d_flag32.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GF_HIP(ctx, gf_wait_stream(st));  // the caller's arrays are free again
    if (sign < 0 && ctx->h_failed.ptr[0] != 0) {
        // an entry was removed from a node that never carried it: the node's sum went negative (the snapshot would report
        // available > allocatable).  Put the update back and refuse it.
        GF_HIP(ctx, gangfit::launch_usage_apply(n_entries, ctx->cl_n, ctx->d_delta_u32.ptr, ctx->d_delta_i64.ptr, +1,
                                                ctx->d_cl_usage.ptr, nullptr, st));
        GF_HIP(ctx, gf_wait_stream(st));
        ctx->usage_ok = true;
        return fail(ctx, GF_ERR_INVALID, "an entry was removed from a node that never carried it (a node's usage went negative)");
    }
    ctx->usage_ok = true;
    for (int j = 0; j < 3; ++j) ctx->usage_total[j] = total[j];
    return GF_OK;
}

int gf_snapshot_build_resident(gf_ctx* ctx, uint32_t n_res, const uint32_t* res_node, const int64_t* res_cpu_milli,
                               const int64_t* res_mem_bytes, const int64_t* res_gpu, const uint32_t* node_flags,
                               const uint32_t* driver_label_rank, const uint32_t* exec_label_rank,
                               uint32_t* driver_order_out, uint32_t* n_d_out, uint32_t* exec_order_out, uint32_t* n_x_out) {
    if (ctx != nullptr && !ctx->group.empty()) {  // the caller's order lists come from the first device only
        gf_ctx* const g = ctx;
        std::lock_guard<std::recursive_mutex> glock(g->mu);
        for (size_t i = 0; i < g->group.size(); ++i) {
            const bool first = i == 0;
            const int rc = gf_snapshot_build_resident(g->group[i], n_res, res_node, res_cpu_milli, res_mem_bytes, res_gpu, node_flags,
                                                      driver_label_rank, exec_label_rank, first ? driver_order_out : nullptr,
                                                      first ? n_d_out : nullptr, first ? exec_order_out : nullptr,
                                                      first ? n_x_out : nullptr);
            if (rc != GF_OK) {
                g->err = g->group[i]->err;
                return rc;
            }
        }
        return GF_OK;
    }
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    InstallGuard install_guard(ctx);
    if (!ctx->have_cluster) return fail(ctx, GF_ERR_STATE, "gf_cluster_set must precede gf_snapshot_build_resident");
    const uint32_t n = ctx->cl_n;
    const uint32_t n_zones = ctx->cl_zones;
    const bool with_over = ctx->cl_over;
    const bool usage_resident = n_res == GF_RESIDENT_USAGE;  // the sums gf_usage_apply maintains: no entry travels
    if (usage_resident) n_res = 0;
    if (n_res > 0 && (!res_node || !res_cpu_milli || !res_mem_bytes || !res_gpu))
        return fail(ctx, GF_ERR_INVALID, "reservation columns must not be NULL");
    if (usage_resident && !ctx->usage_ok)
        return fail(ctx, GF_ERR_STATE, "the resident usage is unknown (a failed update): gf_usage_reset must rebuild it");
    // this request's candidate flags; NULL = the flags of gf_cluster_set (not those of the previous request)
    if (node_flags)
        ctx->cl_flags.assign(node_flags, node_flags + n);
    else
        ctx->cl_flags = ctx->cl_default_flags;
    const uint32_t* const flags_upload = node_flags ? node_flags : (ctx->d_flags_default ? nullptr : ctx->cl_default_flags.data());
    const uint32_t* const zone_of_node = ctx->cl_zone.empty() ? nullptr : ctx->cl_zone.data();
    const uint32_t* const flags_host = ctx->cl_flags.data();
    const int64_t* rcols[3] = {res_cpu_milli, res_mem_bytes, res_gpu};
    const int64_t lim = GF_MAX_ABS_QUANTITY >> 1;
    int64_t max_res[3] = {0, 0, 0};
    for (int j = 0; j < 3; ++j)
        for (uint32_t i = 0; i < n_res; ++i) {
            if (rcols[j][i] < 0 || rcols[j][i] >= lim) return fail(ctx, GF_ERR_INVALID, "reservation %u out of range", i);
            if (rcols[j][i] > max_res[j]) max_res[j] = rcols[j][i];
        }
    if ((uint64_t)n_res >= (1ull << 32) - 1) return fail(ctx, GF_ERR_INVALID, "too many reservations");
    {  // the per-node sums (usage + overhead) must stay below 2^62: the device accumulates in 64 bits and would wrap silently.
        // Coarse bound first (every entry on one node); only when that fails, the real per-node entry counts.
        auto fits = [&](uint64_t count) {
            for (int j = 0; j < 3; ++j)
                if ((unsigned __int128)count * (uint64_t)max_res[j] + (uint64_t)ctx->cl_max_over[j] >= (unsigned __int128)GF_MAX_ABS_QUANTITY)
                    return false;
            return true;
        };
        if (!fits(n_res)) {
            std::vector<uint32_t> cnt(n, 0);
            uint32_t most = 0;
            for (uint32_t i = 0; i < n_res; ++i)
                if (res_node[i] < n && ++cnt[res_node[i]] > most) most = cnt[res_node[i]];
            if (!fits(most))
                return fail(ctx, GF_ERR_INVALID, "the reservations of one node (%u entries) can sum past 2^62: not representable", most);
        }
    }
    if (n == 0) {
        int rc = gf_snapshot_set(ctx, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc != GF_OK) return rc;
        if (n_d_out) *n_d_out = 0;
        if (n_x_out) *n_x_out = 0;
        return gf_orders_set(ctx, nullptr, 0, nullptr, 0);
    }
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // ---- device buffers (the static columns live in the resident cluster buffers)
    const size_t N = n, R = n_res, Z = n_zones;
    const size_t NCH = (N + 1 + 63) / 64;  // chunks of the slot space (nodes + sentinel)
    GF_HIP(ctx, gf_wait_stream(st));  // nothing in flight may still read buffers that are about to grow
    GF_HIP(ctx, ctx->d_bi64.reserve(9 * N + 3 * N + 3 * R + 3 * Z + 6 * NCH + 16));
    GF_HIP(ctx, ctx->d_bu32.reserve(3 * N + R + 5 * Z + 16));
    int64_t* d_alloc = ctx->d_cl_i64.ptr;
    int64_t* d_over = d_alloc + 3 * N;
    int64_t* d_usage = ctx->d_bi64.ptr;
    int64_t* d_avail = d_usage + 3 * N;
    int64_t* d_sched = d_avail + 3 * N;
    int64_t* d_keys_a = d_sched + 3 * N;
    int64_t* d_keys_b = d_keys_a + N;
    int64_t* d_keys_c = d_keys_b + N;
    int64_t* d_res_req = d_keys_c + N;
    int64_t* d_zone_sum = d_res_req + 3 * R;
    uint32_t* d_zone = ctx->d_cl_u32.ptr;
    uint32_t* d_name_rank = d_zone + N;
    uint32_t* d_flags = d_name_rank + N;
    uint32_t* d_perm_a = ctx->d_bu32.ptr;
    uint32_t* d_perm_b = d_perm_a + N;
    uint32_t* d_perm_c = d_perm_b + N;
    uint32_t* d_res_node = d_perm_c + N;
    uint32_t* d_zone_order = d_res_node + R;
    uint32_t* d_zone_rank = d_zone_order + Z;
    uint32_t* d_zfirst = d_zone_rank + Z;
    uint32_t* d_zhasx = d_zfirst + Z;
    uint32_t* d_zeval = d_zhasx + Z;
    uint32_t* d_scalars = d_zeval + Z;  // 4
    unsigned long long* d_gcd_part = reinterpret_cast<unsigned long long*>(d_zone_sum + 3 * Z);
    long long* d_units = reinterpret_cast<long long*>(d_gcd_part + 6 * NCH);  // gcd partials | magnitude partials | units
    for (int j = 0; j < 3 && R; ++j)
        GF_HIP(ctx, hipMemcpyAsync(d_res_req + j * R, rcols[j], R * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (R) GF_HIP(ctx, hipMemcpyAsync(d_res_node, res_node, R * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (flags_upload) {
        GF_HIP(ctx, hipMemcpyAsync(d_flags, flags_upload, N * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        ctx->d_flags_default = node_flags == nullptr;
    }
    gangfit::SnapshotBuild b{};
    b.n_nodes = n;
    b.n_res = n_res;
    b.n_zones = n_zones;
    b.d_alloc = d_alloc;
    b.d_overhead = with_over ? d_over : nullptr;
    b.d_res_node = d_res_node;
    b.d_res_req = d_res_req;
    b.d_zone = d_zone;
    b.d_name_rank = d_name_rank;
    b.d_usage = usage_resident ? ctx->d_cl_usage.ptr : d_usage;
    b.usage_resident = usage_resident;
    b.d_avail = d_avail;
    b.d_sched = d_sched;
    b.d_zone_sum = d_zone_sum;
    b.d_zone_order = d_zone_order;
    b.d_zone_rank = d_zone_rank;
    b.d_perm_a = d_perm_a;
    b.d_perm_b = d_perm_b;
    b.d_keys_a = d_keys_a;
    b.d_keys_b = d_keys_b;
    b.d_keys_c = d_keys_c;
    b.d_perm_c = d_perm_c;
    b.sort_fault = ctx->sort_fault;
    GF_HIP(ctx, ctx->d_sortwork.reserve(gangfit::snapshot_sort_work_words()));
    b.d_sort_work = ctx->d_sortwork.ptr;
    GF_HIP(ctx, gangfit::launch_snapshot_build(b, st));
    if (ctx->snapshot_finalize_on_device && !driver_label_rank && !exec_label_rank) {
        // ---- the slot tables on the device too: nothing of size O(n_nodes) returns to the host unless the caller asks
        //      for the orders.  (Label re-sorts can break the merged layout: those go through gf_orders_set below.)
        const uint32_t n_slots = n + 1, n_chunks = (uint32_t)NCH;
        GF_HIP(ctx, ctx->d_snap.reserve(3 * (size_t)n_slots));
        GF_HIP(ctx, ctx->d_work.reserve(3 * (size_t)n_slots));
        GF_HIP(ctx, ctx->d_sched.reserve(3 * (size_t)n_slots));
        GF_HIP(ctx, ctx->d_slot_node.reserve(n_slots));
        GF_HIP(ctx, ctx->d_dslot.reserve((size_t)n_slots + 1));
        GF_HIP(ctx, ctx->d_node_slot.reserve(N + 1));
        GF_HIP(ctx, ctx->d_cmax.reserve(3 * (size_t)n_chunks));
        GF_HIP(ctx, ctx->d_masks.reserve(2 * (size_t)n_chunks));
        GF_HIP(ctx, ctx->d_node_tab.reserve(6 * N + 1));
        GF_HIP(ctx, ctx->d_zmasks.reserve(2 * Z * (size_t)n_chunks + 1));
        GF_HIP(ctx, ctx->d_nsnap.reserve(3 * (size_t)n_slots));
        GF_HIP(ctx, ctx->d_nwork.reserve(3 * (size_t)n_slots));
        GF_HIP(ctx, ctx->d_ncmax.reserve(3 * (size_t)n_chunks));
        gangfit::SnapshotFinalize f{};
        f.n_nodes = n;
        f.n_slots = n_slots;
        f.n_chunks = n_chunks;
        f.n_zones = n_zones;
        f.d_avail = d_avail;
        f.d_sched = d_sched;
        f.d_perm = d_perm_b;
        f.d_zone = d_zone;
        f.d_flags = d_flags;
        f.d_snap = ctx->d_snap.ptr;
        f.d_sched_slot = ctx->d_sched.ptr;
        f.d_slot_node = ctx->d_slot_node.ptr;
        f.d_node_slot = ctx->d_node_slot.ptr;
        f.d_dslot = ctx->d_dslot.ptr;
        f.d_masks = ctx->d_masks.ptr;
        f.d_cmax = ctx->d_cmax.ptr;
        f.d_node_tab = ctx->d_node_tab.ptr;
        f.d_gcd_part = d_gcd_part;
        f.d_units = d_units;
        f.d_zfirst = d_zfirst;
        f.d_zhasx = d_zhasx;
        f.d_zeval = d_zeval;
        f.d_scalars = d_scalars;
        f.d_zmasks = ctx->d_zmasks.ptr;
        f.d_nsnap = ctx->d_nsnap.ptr;
        f.d_ncmax = ctx->d_ncmax.ptr;
        GF_HIP(ctx, gangfit::launch_snapshot_finalize(f, st));
        GF_HIP(ctx, ctx->h_bcols.reserve(6 * N + 8));
        GF_HIP(ctx, ctx->h_border.reserve(N + 8));
        long long* h_units = reinterpret_cast<long long*>(ctx->h_bcols.ptr);  // 3 units, then the 3 largest scaled magnitudes
        uint32_t* h_scalars = ctx->h_border.ptr;
        GF_HIP(ctx, hipMemcpyAsync(h_units, d_units, 6 * sizeof(long long), hipMemcpyDeviceToHost, st));
        GF_HIP(ctx, hipMemcpyAsync(h_scalars, d_scalars, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        GF_HIP(ctx, hipMemcpyAsync(h_scalars + 4, ctx->d_sortwork.ptr + gangfit::snapshot_sort_error_word(), sizeof(uint32_t),
                                   hipMemcpyDeviceToHost, st));
        GF_HIP(ctx, gf_wait_stream(st));
        if (h_scalars[4] != 0) return fail(ctx, GF_ERR_HIP, "the priority sort's grid barrier gave up (device oversubscribed?)");
        const uint32_t nz = h_scalars[0];
        for (int j = 0; j < 3; ++j) {
            ctx->unit[j] = (int64_t)h_units[j];
            ctx->nmax[j] = (int64_t)h_units[3 + j];
        }
        ctx->narrow_ok = h_scalars[1] == 0;
        ctx->have_sched = h_scalars[2] == 0;  // a negative schedulable value (overhead above allocatable) disables the efficiencies
        ctx->n_nodes = n;
        ctx->have_snapshot = true;
        ctx->zone.clear();
        if (zone_of_node) ctx->zone.assign(zone_of_node, zone_of_node + N);
        ctx->n_x = ctx->n_d = n;
        ctx->n_g = ctx->n_gpad = 0;  // the sparse gpu view is built by gf_orders_set only; the full order serves here
        ctx->n_slots = n_slots;
        ctx->n_chunks = n_chunks;
        ctx->d_identity = true;
        ctx->merged = true;
        ctx->n_zones = nz;
        ctx->zstride = n_chunks;
        ctx->zd_row0 = n_zones;
        ctx->have_orders = true;
        ctx->work_valid = false;
        ++ctx->snap_epoch;
        ctx->host_stale = true;
        if (driver_order_out || exec_order_out || n_d_out || n_x_out) {  // the two lists, for callers that want them
            GF_HIP(ctx, hipMemcpyAsync(ctx->h_border.ptr, d_perm_b, N * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            GF_HIP(ctx, gf_wait_stream(st));
            uint32_t nd = 0, nx = 0;
            for (size_t i = 0; i < N; ++i) {
                const uint32_t node = ctx->h_border.ptr[i];
                const uint32_t fl = flags_host[node];
                if (fl & GF_NODE_DRIVER_CANDIDATE) {
                    if (driver_order_out) driver_order_out[nd] = node;
                    ++nd;
                }
                if (!(fl & GF_NODE_UNSCHEDULABLE) && (fl & GF_NODE_READY)) {
                    if (exec_order_out) exec_order_out[nx] = node;
                    ++nx;
                }
            }
            if (n_d_out) *n_d_out = nd;
            if (n_x_out) *n_x_out = nx;
        }
        return GF_OK;
    }
    GF_HIP(ctx, ctx->h_bcols.reserve(6 * N));
    GF_HIP(ctx, ctx->h_border.reserve(N + 8));
    GF_HIP(ctx, hipMemcpyAsync(ctx->h_bcols.ptr, d_avail, 6 * N * sizeof(int64_t), hipMemcpyDeviceToHost, st));  // avail | sched
    GF_HIP(ctx, hipMemcpyAsync(ctx->h_border.ptr, d_perm_b, N * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GF_HIP(ctx, hipMemcpyAsync(ctx->h_failed.ptr, ctx->d_sortwork.ptr + gangfit::snapshot_sort_error_word(), sizeof(uint32_t),
                               hipMemcpyDeviceToHost, st));
    GF_HIP(ctx, gf_wait_stream(st));
    if (ctx->h_failed.ptr[0] != 0) return fail(ctx, GF_ERR_HIP, "the priority sort's grid barrier gave up (device oversubscribed?)");
    // ---- the two candidate lists (nodesorting.go:47-63) and the optional stable label re-sorts (:161-199)
    const int64_t* h_avail = ctx->h_bcols.ptr;
    const int64_t* h_sched = ctx->h_bcols.ptr + 3 * N;
    bool sched_ok = true;
    for (size_t i = 0; i < 3 * N && sched_ok; ++i) sched_ok = h_sched[i] >= 0;
    std::vector<uint32_t> D, X;
    D.reserve(N);
    X.reserve(N);
    for (size_t i = 0; i < N; ++i) {
        const uint32_t node = ctx->h_border.ptr[i];
        const uint32_t f = flags_host[node];
        if (f & GF_NODE_DRIVER_CANDIDATE) D.push_back(node);
        if (!(f & GF_NODE_UNSCHEDULABLE) && (f & GF_NODE_READY)) X.push_back(node);
    }
    auto by_rank = [](std::vector<uint32_t>& v, const uint32_t* rank) {
        std::stable_sort(v.begin(), v.end(), [rank](uint32_t a, uint32_t b) { return rank[a] < rank[b]; });
    };
    if (driver_label_rank) by_rank(D, driver_label_rank);
    if (exec_label_rank) by_rank(X, exec_label_rank);
    int rc = gf_snapshot_set(ctx, n, h_avail, h_avail + N, h_avail + 2 * N, sched_ok ? h_sched : nullptr,
                             sched_ok ? h_sched + N : nullptr, sched_ok ? h_sched + 2 * N : nullptr);
    if (rc != GF_OK) return rc;
    if (zone_of_node && (rc = gf_zones_set(ctx, zone_of_node)) != GF_OK) return rc;
    if ((rc = gf_orders_set(ctx, D.data(), (uint32_t)D.size(), X.data(), (uint32_t)X.size())) != GF_OK) return rc;
    if (n_d_out) *n_d_out = (uint32_t)D.size();
    if (n_x_out) *n_x_out = (uint32_t)X.size();
    if (driver_order_out) std::memcpy(driver_order_out, D.data(), D.size() * sizeof(uint32_t));
    if (exec_order_out) std::memcpy(exec_order_out, X.data(), X.size() * sizeof(uint32_t));
    return GF_OK;
}

int gf_snapshot_get(gf_ctx* ctx, int64_t* avail_out, int64_t* sched_out) {
    GF_DELEGATE(ctx, gf_snapshot_get(ctx, avail_out, sched_out));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (!ctx->have_snapshot) return fail(ctx, GF_ERR_STATE, "no snapshot");
    if (int mrc = materialize_host(ctx); mrc != GF_OK) return mrc;
    for (uint32_t i = 0; i < ctx->n_nodes; ++i)
        for (int j = 0; j < 3; ++j) {
            if (avail_out) avail_out[3 * (size_t)i + j] = ctx->avail[j][i];
            if (sched_out) sched_out[3 * (size_t)i + j] = ctx->have_sched ? ctx->sched[j][i] : 0;
        }
    return GF_OK;
}

int gf_executor_fit(gf_ctx* ctx, int minimal_fragmentation, uint32_t n_req, const int64_t* exe, const int64_t* reserved,
                    const uint32_t* hosts_app, uint32_t* node_out) {
    GF_DELEGATE(ctx, gf_executor_fit(ctx, minimal_fragmentation, n_req, exe, reserved, hosts_app, node_out));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (n_req == 0) return GF_OK;
    if (!exe || !node_out) return fail(ctx, GF_ERR_INVALID, "exe/node_out must not be NULL");
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede gf_executor_fit");
    for (size_t i = 0; i < 3 * (size_t)n_req; ++i)
        if (exe[i] < 0 || exe[i] >= GF_MAX_ABS_QUANTITY) return fail(ctx, GF_ERR_INVALID, "executor request outside [0, 2^62)");
    const uint32_t n = ctx->n_nodes;
    if (reserved)
        for (size_t i = 0; i < 3 * (size_t)n; ++i)
            if (reserved[i] < 0 || reserved[i] >= GF_MAX_ABS_QUANTITY)
                return fail(ctx, GF_ERR_INVALID, "reserved[%zu] outside [0, 2^62)", i);
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t words = (n + 31) / 32;
    GF_HIP(ctx, ctx->d_xexe.reserve(3 * (size_t)n_req));
    GF_HIP(ctx, ctx->d_xout.reserve(n_req));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_xexe.ptr, exe, 3 * (size_t)n_req * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (reserved) {
        GF_HIP(ctx, ctx->d_xreserved.reserve(3 * (size_t)n + 1));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_xreserved.ptr, reserved, 3 * (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    }
    const bool with_hosts = minimal_fragmentation && hosts_app && words > 0;
    if (with_hosts) {
        GF_HIP(ctx, ctx->d_xhosts.reserve((size_t)n_req * words));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_xhosts.ptr, hosts_app, (size_t)n_req * words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    GF_HIP(ctx, gangfit::launch_executor_fit(minimal_fragmentation != 0, make_table(ctx, ctx->d_snap.ptr),
                                             reserved ? ctx->d_xreserved.ptr : nullptr, n_req, ctx->d_xexe.ptr,
                                             with_hosts ? ctx->d_xhosts.ptr : nullptr, words, ctx->d_xout.ptr, st));
    GF_HIP(ctx, hipMemcpyAsync(node_out, ctx->d_xout.ptr, (size_t)n_req * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GF_HIP(ctx, gf_wait_stream(st));
    return GF_OK;
}

int gf_find_nodes(gf_ctx* ctx, int chained, uint32_t n_req, const int64_t* exe, const int32_t* k, gf_find_result* results,
                  uint32_t* exec_nodes, uint64_t exec_nodes_cap, uint32_t* reserved_adds) {
    GF_DELEGATE(ctx, gf_find_nodes(ctx, chained, n_req, exe, k, results, exec_nodes, exec_nodes_cap, reserved_adds));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (n_req == 0) return GF_OK;
    if (!exe || !k || !results) return fail(ctx, GF_ERR_INVALID, "exe/k/results must not be NULL");
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede gf_find_nodes");
    for (size_t i = 0; i < 3 * (size_t)n_req; ++i)
        if (exe[i] < 0 || exe[i] >= GF_MAX_ABS_QUANTITY) return fail(ctx, GF_ERR_INVALID, "executor request outside [0, 2^62)");
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, ctx->h_foff.reserve(n_req));
    uint64_t total_k = 0;
    for (uint32_t q = 0; q < n_req; ++q) {
        if (k[q] < 0 || k[q] > GF_MAX_K) return fail(ctx, GF_ERR_INVALID, "k[%u] = %d outside [0, %d]", q, k[q], GF_MAX_K);
        ctx->h_foff.ptr[q] = total_k;
        total_k += (uint64_t)k[q];
    }
    if (total_k > exec_nodes_cap || (total_k > 0 && !exec_nodes))
        return fail(ctx, GF_ERR_CAPACITY, "exec_nodes holds %llu entries, %llu needed", (unsigned long long)exec_nodes_cap,
                    (unsigned long long)total_k);
    const uint32_t n = ctx->n_nodes;
    hipStream_t st = ctx->stream;
    GF_HIP(ctx, ctx->d_xexe.reserve(3 * (size_t)n_req));
    GF_HIP(ctx, ctx->d_fk.reserve(n_req));
    GF_HIP(ctx, ctx->d_foff.reserve(n_req));
    GF_HIP(ctx, ctx->d_fres.reserve(n_req));
    GF_HIP(ctx, ctx->d_exec.reserve(total_k + 1));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_xexe.ptr, exe, 3 * (size_t)n_req * sizeof(int64_t), hipMemcpyHostToDevice, st));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_fk.ptr, k, (size_t)n_req * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_foff.ptr, ctx->h_foff.ptr, (size_t)n_req * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    uint32_t* d_adds = nullptr;
    if (reserved_adds && n > 0) {
        GF_HIP(ctx, ctx->d_fadds.reserve((size_t)n_req * n));
        GF_HIP(ctx, hipMemsetAsync(ctx->d_fadds.ptr, 0, (size_t)n_req * n * sizeof(uint32_t), st));
        d_adds = ctx->d_fadds.ptr;
    }
    if (chained) {  // every reconcile starts from the snapshot (availableResourcesPerInstanceGroup, failover.go:286-322)
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_work.ptr, ctx->d_snap.ptr, 3 * (size_t)ctx->n_slots * sizeof(int64_t),
                                   hipMemcpyDeviceToDevice, st));
        ctx->work_valid = true;
    }
    GF_HIP(ctx, gangfit::launch_find_nodes(chained != 0, make_table(ctx, chained ? ctx->d_work.ptr : ctx->d_snap.ptr), n_req,
                                           ctx->d_xexe.ptr, ctx->d_fk.ptr, ctx->d_foff.ptr, ctx->d_fres.ptr, ctx->d_exec.ptr,
                                           d_adds, st));
    GF_HIP(ctx, hipMemcpyAsync(results, ctx->d_fres.ptr, (size_t)n_req * sizeof(gf_find_result), hipMemcpyDeviceToHost, st));
    if (total_k)
        GF_HIP(ctx, hipMemcpyAsync(exec_nodes, ctx->d_exec.ptr, (size_t)total_k * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (d_adds)
        GF_HIP(ctx, hipMemcpyAsync(reserved_adds, d_adds, (size_t)n_req * n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GF_HIP(ctx, gf_wait_stream(st));
    return GF_OK;
}

// ---- node-range sharding (gangfit_shard.inc)
namespace {
int shard_ready(gf_ctx* ctx, gf_algo algo, gangfit::ShardRange* r) {
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede a sharded fit");
    if (!ctx->merged)
        return fail(ctx, GF_ERR_UNSUPPORTED, "node-range sharding needs the merged slot layout (driver and executor "
                                             "orders must be subsequences of one priority order)");
    if (algo != GF_ALGO_TIGHTLY_PACK && algo != GF_ALGO_DISTRIBUTE_EVENLY)
        return fail(ctx, GF_ERR_UNSUPPORTED, "node-range sharding serves tightly-pack and distribute-evenly only");
    const uint64_t xc = ((uint64_t)ctx->n_x + 63) / 64;  // chunks of the merged order (the sentinel slot hosts nothing)
    r->c_lo = (uint32_t)(xc * ctx->shard / ctx->n_shards);
    r->c_hi = (uint32_t)(xc * (ctx->shard + 1) / ctx->n_shards);
    r->shard = ctx->shard;
    r->n_shards = ctx->n_shards;
    return GF_OK;
}
}  // namespace

int gf_shard_set(gf_ctx* ctx, uint32_t shard, uint32_t n_shards) {
    if (ctx != nullptr && !ctx->group.empty())
        return fail(ctx, GF_ERR_UNSUPPORTED, "a multi-device context runs the shard steps and their exchanges itself (gf_fit_batch)");
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_NOT_ON_A_VIEW(ctx);
    if (n_shards == 0 || shard >= n_shards || n_shards > 1024)
        return fail(ctx, GF_ERR_INVALID, "shard %u of %u", shard, n_shards);
    ctx->shard = shard;
    ctx->n_shards = n_shards;
    return GF_OK;
}

int gf_shard_partials_dev(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* d_apps, gf_shard_partial* d_out,
                          void* stream) {
    if (ctx != nullptr && !ctx->group.empty())
        return fail(ctx, GF_ERR_UNSUPPORTED, "a multi-device context runs the shard steps and their exchanges itself (gf_fit_batch)");
    if (!ctx) return GF_ERR_INVALID;
    if (n_apps > 0 && (!d_apps || !d_out)) return fail(ctx, GF_ERR_INVALID, "device pointers must not be NULL");
    gangfit::ShardRange r{};
    const int rc = shard_ready(ctx, algo, &r);
    if (rc != GF_OK) return rc;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    GF_HIP(ctx, gangfit::launch_shard_partials(algo, make_table(ctx, ctx->d_snap.ptr), r, n_apps, d_apps, d_out, st));
    return GF_OK;
}

int gf_shard_drivers_dev(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* d_apps,
                         const gf_shard_partial* d_all_partials, gf_shard_driver* d_out, void* stream) {
    if (ctx != nullptr && !ctx->group.empty())
        return fail(ctx, GF_ERR_UNSUPPORTED, "a multi-device context runs the shard steps and their exchanges itself (gf_fit_batch)");
    if (!ctx) return GF_ERR_INVALID;
    if (n_apps > 0 && (!d_apps || !d_all_partials || !d_out))
        return fail(ctx, GF_ERR_INVALID, "device pointers must not be NULL");
    gangfit::ShardRange r{};
    const int rc = shard_ready(ctx, algo, &r);
    if (rc != GF_OK) return rc;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    GF_HIP(ctx, gangfit::launch_shard_drivers(make_table(ctx, ctx->d_snap.ptr), r, n_apps, d_apps, d_all_partials, d_out, st));
    return GF_OK;
}

int gf_shard_emit_dev(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* d_apps,
                      const gf_shard_partial* d_all_partials, const gf_shard_driver* d_all_drivers, gf_result* d_results,
                      uint32_t* d_exec2, uint64_t half, void* stream) {
    if (ctx != nullptr && !ctx->group.empty())
        return fail(ctx, GF_ERR_UNSUPPORTED, "a multi-device context runs the shard steps and their exchanges itself (gf_fit_batch)");
    if (!ctx) return GF_ERR_INVALID;
    if (n_apps > 0 && (!d_apps || !d_all_partials || !d_all_drivers || !d_results || !d_exec2 || half == 0))
        return fail(ctx, GF_ERR_INVALID, "device pointers must not be NULL");
    gangfit::ShardRange r{};
    const int rc = shard_ready(ctx, algo, &r);
    if (rc != GF_OK) return rc;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    GF_HIP(ctx, gangfit::launch_shard_emit(algo, make_table(ctx, ctx->d_snap.ptr), r, n_apps, d_apps, d_all_partials,
                                           d_all_drivers, d_results, d_exec2, half, st));
    return GF_OK;
}

int gf_shard_finish_dev(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* d_apps,
                        const gf_shard_partial* d_all_partials, const gf_shard_driver* d_all_drivers,
                        const gf_result* d_results, uint32_t* d_exec2, uint64_t half, void* stream) {
    if (ctx != nullptr && !ctx->group.empty())
        return fail(ctx, GF_ERR_UNSUPPORTED, "a multi-device context runs the shard steps and their exchanges itself (gf_fit_batch)");
    if (!ctx) return GF_ERR_INVALID;
    if (n_apps > 0 && (!d_apps || !d_all_partials || !d_all_drivers || !d_results || !d_exec2 || half == 0))
        return fail(ctx, GF_ERR_INVALID, "device pointers must not be NULL");
    gangfit::ShardRange r{};
    const int rc = shard_ready(ctx, algo, &r);
    if (rc != GF_OK) return rc;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    GF_HIP(ctx, gangfit::launch_shard_finish(algo, ctx->n_shards, n_apps, d_apps, d_all_partials, d_all_drivers,
                                             d_results, d_exec2, half, st));
    return GF_OK;
}

int gf_avg_packing_efficiency(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* apps, const gf_result* results,
                              const uint32_t* exec_nodes, uint64_t exec_nodes_len, gf_avg_efficiency* out) {
    GF_DELEGATE(ctx, gf_avg_packing_efficiency(ctx, algo, n_apps, apps, results, exec_nodes, exec_nodes_len, out));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (n_apps > 0 && (!apps || !results || !out)) return fail(ctx, GF_ERR_INVALID, "apps/results/out must not be NULL");
    if (n_apps == 0) return GF_OK;
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede");
    if (!ctx->have_sched) return fail(ctx, GF_ERR_STATE, "efficiencies need the schedulable columns of gf_snapshot_set");
    if (int mrc = materialize_host(ctx); mrc != GF_OK) return mrc;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    // validate the lists on the host: every placed node must own a slot (it came out of one of the two orders)
    uint64_t total_k = 0;
    GF_HIP(ctx, ctx->h_apps.reserve(n_apps));
    for (uint32_t a = 0; a < n_apps; ++a) {
        gf_app& o = ctx->h_apps.ptr[a];
        o = apps[a];
        if (o.k < 0 || o.k > GF_MAX_K) return fail(ctx, GF_ERR_INVALID, "apps[%u].k out of range", a);
        o.exec_off = total_k;
        if (results[a].has_capacity) {
            const uint32_t d = results[a].driver_node;
            if (d >= ctx->n_nodes || ctx->h_node_slot[d] == GF_NO_NODE)
                return fail(ctx, GF_ERR_INVALID, "results[%u].driver_node is not a candidate node", a);
            if (total_k + (uint64_t)o.k > exec_nodes_len || (o.k > 0 && !exec_nodes))
                return fail(ctx, GF_ERR_CAPACITY, "exec_nodes too short");
            for (int32_t i = 0; i < o.k; ++i) {
                const uint32_t n = exec_nodes[total_k + i];
                if (n >= ctx->n_nodes || ctx->h_node_slot[n] == GF_NO_NODE)
                    return fail(ctx, GF_ERR_INVALID, "exec_nodes[%llu] is not a candidate node",
                                (unsigned long long)(total_k + i));
            }
        }
        total_k += (uint64_t)o.k;
    }
    hipStream_t st = ctx->stream;
    GF_HIP(ctx, ctx->d_apps.reserve(n_apps));
    GF_HIP(ctx, ctx->d_results.reserve(n_apps));
    GF_HIP(ctx, ctx->d_exec.reserve(total_k + 1));
    GF_HIP(ctx, ctx->d_avg.reserve(4 * (size_t)n_apps));
    GF_HIP(ctx, ctx->h_avg.reserve(4 * (size_t)n_apps));
    int rc = ensure_cnt(ctx, n_apps, st);
    if (rc != GF_OK) return rc;
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_apps.ptr, ctx->h_apps.ptr, (size_t)n_apps * sizeof(gf_app), hipMemcpyHostToDevice, st));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_results.ptr, results, (size_t)n_apps * sizeof(gf_result), hipMemcpyHostToDevice, st));
    if (total_k && exec_nodes)
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_exec.ptr, exec_nodes, (size_t)(total_k <= exec_nodes_len ? total_k : exec_nodes_len) * sizeof(uint32_t),
                                   hipMemcpyHostToDevice, st));
    GF_HIP(ctx, gangfit::launch_avg_efficiency(reserves_executors(algo), make_table(ctx, ctx->d_snap.ptr),
                                               slot_eff_tables(ctx, ctx->d_snap.ptr), ctx->d_cnt.ptr, ctx->cnt_rows,
                                               n_apps, ctx->d_apps.ptr, ctx->d_results.ptr, ctx->d_exec.ptr,
                                               ctx->d_avg.ptr, st));
    GF_HIP(ctx, hipMemcpyAsync(ctx->h_avg.ptr, ctx->d_avg.ptr, 4 * (size_t)n_apps * sizeof(double), hipMemcpyDeviceToHost, st));
    GF_HIP(ctx, gf_wait_stream(st));
    static_assert(sizeof(gf_avg_efficiency) == 4 * sizeof(double), "gf_avg_efficiency layout");
    std::memcpy(out, ctx->h_avg.ptr, 4 * (size_t)n_apps * sizeof(double));
    return GF_OK;
}

int gf_packing_efficiencies(gf_ctx* ctx, gf_algo algo, const gf_app* app, const gf_result* result,
                            const uint32_t* exec_nodes, double* eff_out) {
    GF_DELEGATE(ctx, gf_packing_efficiencies(ctx, algo, app, result, exec_nodes, eff_out));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (!app || !result || !eff_out) return fail(ctx, GF_ERR_INVALID, "app/result/eff_out must not be NULL");
    if (!ctx->have_snapshot || !ctx->have_sched)
        return fail(ctx, GF_ERR_STATE, "efficiencies need gf_snapshot_set with the schedulable columns");
    if (app->k < 0 || app->k > GF_MAX_K || (result->has_capacity && app->k > 0 && !exec_nodes))
        return fail(ctx, GF_ERR_INVALID, "bad k / exec_nodes");
    const uint32_t n = ctx->n_nodes;
    if (n == 0) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    GF_HIP(ctx, ctx->d_apps.reserve(1));
    GF_HIP(ctx, ctx->d_results.reserve(1));
    GF_HIP(ctx, ctx->d_exec.reserve((size_t)app->k + 1));
    GF_HIP(ctx, ctx->d_reserved.reserve(3 * (size_t)n));
    GF_HIP(ctx, ctx->d_eff.reserve(3 * (size_t)n));
    GF_HIP(ctx, ctx->h_apps.reserve(1));
    ctx->h_apps.ptr[0] = *app;
    ctx->h_apps.ptr[0].exec_off = 0;
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_apps.ptr, ctx->h_apps.ptr, sizeof(gf_app), hipMemcpyHostToDevice, st));
    GF_HIP(ctx, hipMemcpyAsync(ctx->d_results.ptr, result, sizeof(gf_result), hipMemcpyHostToDevice, st));
    if (result->has_capacity && app->k > 0)
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_exec.ptr, exec_nodes, (size_t)app->k * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    gangfit::EffTables e;
    for (int j = 0; j < 3; ++j) {
        e.avail[j] = ctx->d_node_tab.ptr + (size_t)j * n;
        e.sched[j] = ctx->d_node_tab.ptr + (size_t)(3 + j) * n;
    }
    GF_HIP(ctx, gangfit::launch_node_efficiencies(reserves_executors(algo), e, n, app->k, ctx->d_apps.ptr,
                                                  ctx->d_results.ptr, ctx->d_exec.ptr, ctx->d_reserved.ptr,
                                                  ctx->d_eff.ptr, st));
    GF_HIP(ctx, hipMemcpyAsync(eff_out, ctx->d_eff.ptr, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    GF_HIP(ctx, gf_wait_stream(st));
    return GF_OK;
}

int gf_residual_get(gf_ctx* ctx, int64_t* avail_out) {
    GF_DELEGATE(ctx, gf_residual_get(ctx, avail_out));
    if (!ctx || !avail_out) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (!ctx->have_orders || !ctx->work_valid) return fail(ctx, GF_ERR_STATE, "no FIFO chain has run on the current orders");
    if (int mrc = materialize_host(ctx); mrc != GF_OK) return mrc;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, ctx->h_table.reserve(3 * (size_t)ctx->n_slots));
    GF_HIP(ctx, hipMemcpyAsync(ctx->h_table.ptr, ctx->d_work.ptr, 3 * (size_t)ctx->n_slots * sizeof(int64_t),
                               hipMemcpyDeviceToHost, ctx->stream));
    GF_HIP(ctx, gf_wait_stream(ctx->stream));
    const int64_t* t = ctx->h_table.ptr;
    for (uint32_t n = 0; n < ctx->n_nodes; ++n) {
        const uint32_t s = ctx->h_node_slot[n];
        for (int j = 0; j < 3; ++j)
            avail_out[3 * (size_t)n + j] = (s == GF_NO_NODE) ? ctx->avail[j][n] : t[(size_t)j * ctx->n_slots + s];
    }
    return GF_OK;
}

int gf_timer_begin(gf_ctx* ctx, void* stream) {
    GF_DELEGATE(ctx, gf_timer_begin(ctx, stream));
    if (!ctx) return GF_ERR_INVALID;
    ctx->timer_stream = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    GF_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->timer_stream));
    return GF_OK;
}

int gf_timer_end(gf_ctx* ctx, float* elapsed_ms) {
    GF_DELEGATE(ctx, gf_timer_end(ctx, elapsed_ms));
    if (!ctx || !elapsed_ms) return GF_ERR_INVALID;
    GF_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->timer_stream ? ctx->timer_stream : ctx->stream));
    GF_HIP(ctx, gf_wait_event(ctx->ev_end));
    GF_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
    return GF_OK;
}

int gf_scan_stats(gf_ctx* ctx, int enable, int reset, uint64_t out[10]) {
    GF_DELEGATE(ctx, gf_scan_stats(ctx, enable, reset, out));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, gf_wait_stream(ctx->stream));
    if (out) {
        ScanStats s;
        GF_HIP(ctx, hipMemcpy(&s, ctx->d_stats.ptr, sizeof s, hipMemcpyDeviceToHost));
        out[0] = s.exec_slots_visited;
        out[1] = s.driver_slots_visited;
        out[2] = s.fifo_shader_cycles;
        out[3] = s.fifo_realtime_ticks;
        for (int i = 0; i < 6; ++i) out[4 + i] = s.fifo_phase_cycles[i];
    }
    if (reset) {  // on the context's stream: a null-stream memset is not ordered against a non-blocking stream
        GF_HIP(ctx, hipMemsetAsync(ctx->d_stats.ptr, 0, sizeof(ScanStats), ctx->stream));
        GF_HIP(ctx, gf_wait_stream(ctx->stream));
    }
    ctx->stats_on = enable != 0;
    return GF_OK;
}

int gf_selftest(gf_ctx* ctx, uint64_t seed, uint32_t n_cases, uint32_t* mismatches) {
    GF_DELEGATE(ctx, gf_selftest(ctx, seed, n_cases, mismatches));
    if (!ctx || !mismatches) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_HIP(ctx, hipSetDevice(ctx->device));
    DeviceBuf<uint32_t> d;
    GF_HIP(ctx, d.reserve(1));
    hipError_t e = hipMemsetAsync(d.ptr, 0, sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) e = gangfit::launch_selftest(seed, n_cases, d.ptr, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches, d.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = gf_wait_stream(ctx->stream);
    d.release();
    if (e != hipSuccess) return fail(ctx, GF_ERR_HIP, "selftest failed: %s", hipGetErrorString(e));
    return GF_OK;
}

}  // extern "C"

namespace {

// gf_fit_batch on a multi-device context.  Independent batches of the two plain packers are node-range sharded across the
// sub-contexts (the four steps of gangfit_shard.inc with the three exchanges done by peer access,
// see shard_push_kernel / shard_reduce_pull_kernel); everything else — FIFO chains (each commit must be visible to the next
// scan), the zone-aware and minimal-fragmentation packers, orders that do not merge — runs on the first device.
int group_fit_batch(gf_ctx* g, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app* apps, gf_result* results,
                    uint32_t* exec_nodes, uint64_t exec_nodes_cap, int32_t* chain_failed_at) {
    std::lock_guard<std::recursive_mutex> glock(g->mu);
    gf_ctx* const first = g->group[0];
    bool sharded = mode == GF_MODE_INDEPENDENT && (algo == GF_ALGO_TIGHTLY_PACK || algo == GF_ALGO_DISTRIBUTE_EVENLY) && n_apps > 0 &&
                   !g->g_shard_off;
    for (gf_ctx* s : g->group) sharded = sharded && s->have_orders && s->merged;
    if (!sharded) {
        const int rc = gf_fit_batch(first, mode, algo, n_apps, apps, results, exec_nodes, exec_nodes_cap, chain_failed_at);
        if (rc != GF_OK) g->err = first->err;
        return rc;
    }
    if (!apps || !results) return fail(g, GF_ERR_INVALID, "apps/results must not be NULL");
    if (chain_failed_at) *chain_failed_at = -1;
    const uint32_t S = (uint32_t)g->group.size();
    GF_HIP(g, hipSetDevice(first->device));
    GF_HIP(g, g->h_apps.reserve(n_apps));
    uint64_t total_k = 0;
    for (uint32_t a = 0; a < n_apps; ++a) {
        const gf_app& in = apps[a];
        if (in.k < 0 || in.k > GF_MAX_K) return fail(g, GF_ERR_INVALID, "apps[%u].k = %d outside [0, %d]", a, in.k, GF_MAX_K);
        for (int j = 0; j < 3; ++j)
            if (in.drv[j] < 0 || in.drv[j] >= GF_MAX_ABS_QUANTITY || in.exe[j] < 0 || in.exe[j] >= GF_MAX_ABS_QUANTITY)
                return fail(g, GF_ERR_INVALID, "apps[%u] request outside [0, 2^62)", a);
        gf_app& o = g->h_apps.ptr[a];
        o = in;
        o.exec_off = total_k;
        total_k += (uint64_t)in.k;
    }
    if (total_k > exec_nodes_cap || (total_k > 0 && !exec_nodes))
        return fail(g, GF_ERR_CAPACITY, "exec_nodes holds %llu entries, %llu needed", (unsigned long long)exec_nodes_cap,
                    (unsigned long long)total_k);
    const uint64_t half = total_k + 1;
    GF_HIP(g, g->h_results.reserve(n_apps));
    GF_HIP(g, g->h_exec.reserve(total_k + 1));
    // ---- buffers and the app table on every device
    gangfit::ShardRange range[gangfit::kMaxGroupDevices];
    gangfit::PeerPtrs part_all{}, drv_all{}, exec_others{};
    for (uint32_t s = 0; s < S; ++s) {
        gf_ctx* c = g->group[s];
        GF_HIP(g, hipSetDevice(c->device));
        if (const int rc = shard_ready(c, algo, &range[s]); rc != GF_OK) {
            g->err = c->err;
            return rc;
        }
        GF_HIP(g, c->d_apps.reserve(n_apps));
        GF_HIP(g, c->d_results.reserve(n_apps));
        GF_HIP(g, c->g_part_loc.reserve(n_apps));
        GF_HIP(g, c->g_drv_loc.reserve(n_apps));
        GF_HIP(g, c->g_part_all.reserve((size_t)S * n_apps));
        GF_HIP(g, c->g_drv_all.reserve((size_t)S * n_apps));
        GF_HIP(g, c->g_exec2.reserve(2 * half));
        GF_HIP(g, hipMemcpyAsync(c->d_apps.ptr, g->h_apps.ptr, (size_t)n_apps * sizeof(gf_app), hipMemcpyHostToDevice, c->stream));
        part_all.p[s] = c->g_part_all.ptr;
        drv_all.p[s] = c->g_drv_all.ptr;
        if (s > 0) exec_others.p[exec_others.n++] = c->g_exec2.ptr;
    }
    part_all.n = drv_all.n = S;
    const bool use_rccl = g->g_comms.size() == S;
    bool several_streams = false;  // (every shard on one device: one stream, nothing to order with events)
    for (uint32_t s = 1; s < S; ++s) several_streams = several_streams || g->group[s]->stream != first->stream;
    // RCCL exchange: every device's collective is enqueued on its own stream inside one group call; the library orders the
    // streams against each other, so the event fan-out of the peer-store path is not needed
    auto rccl_all_gather = [&](auto loc, auto all, size_t bytes_each) -> int {
        if (rccl().GroupStart() != 0) return -1;
        int bad = 0;
        for (uint32_t s2 = 0; s2 < S; ++s2) {
            gf_ctx* c = g->group[s2];
            if (hipSetDevice(c->device) != hipSuccess) bad = 1;
            bad |= rccl().AllGather(loc(c), all(c), bytes_each, Rccl::kChar, g->g_comms[s2], c->stream);
        }
        return rccl().GroupEnd() | bad;
    };
    auto everyone_waits = [&](int which) -> hipError_t {  // stream t continues only behind event `which` of every other shard
        // (S (S - 1) stream waits; joining the events on one stream first — 2 S + 1 calls — measured slower with eight shards on
        //  one device: the extra hop costs more than the calls it saves)
        for (uint32_t t = 0; t < S; ++t) {
            hipError_t e = hipSetDevice(g->group[t]->device);
            for (uint32_t s = 0; s < S && e == hipSuccess; ++s)
                if (s != t && g->group[s]->stream != g->group[t]->stream)  // (shards on one device share a stream: already ordered)
                    e = hipStreamWaitEvent(g->group[t]->stream, g->group[s]->g_ev[which], 0);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    // ---- step 1: per-range capacity sums, gathered everywhere
    for (uint32_t s = 0; s < S; ++s) {
        gf_ctx* c = g->group[s];
        GF_HIP(g, hipSetDevice(c->device));
        GF_HIP(g, gangfit::launch_shard_partials(algo, make_table(c, c->d_snap.ptr), range[s], n_apps, c->d_apps.ptr, c->g_part_loc.ptr, c->stream));
        if (g->g_fault == 2 && s > 0)  // fault injection: this shard's capacity sums arrive as zeros
            GF_HIP(g, hipMemsetAsync(c->g_part_loc.ptr, 0, (size_t)n_apps * sizeof(gf_shard_partial), c->stream));
        if (use_rccl) continue;
        GF_HIP(g, gangfit::launch_shard_push(c->g_part_loc.ptr, part_all, (size_t)s * n_apps * sizeof(gf_shard_partial),
                                             (size_t)n_apps * sizeof(gf_shard_partial), c->stream));
        if (several_streams) GF_HIP(g, hipEventRecord(c->g_ev[0], c->stream));
    }
    if (use_rccl) {
        if (rccl_all_gather([](gf_ctx* c) { return (const void*)c->g_part_loc.ptr; }, [](gf_ctx* c) { return (void*)c->g_part_all.ptr; },
                            (size_t)n_apps * sizeof(gf_shard_partial)) != 0)
            return fail(g, GF_ERR_HIP, "ncclAllGather of the capacity sums failed");
    } else {
        if (several_streams) GF_HIP(g, everyone_waits(0));
    }
    // ---- step 2: first feasible driver of each range, gathered everywhere
    for (uint32_t s = 0; s < S; ++s) {
        gf_ctx* c = g->group[s];
        GF_HIP(g, hipSetDevice(c->device));
        GF_HIP(g, gangfit::launch_shard_drivers(make_table(c, c->d_snap.ptr), range[s], n_apps, c->d_apps.ptr, c->g_part_all.ptr, c->g_drv_loc.ptr, c->stream));
        if (use_rccl) continue;
        GF_HIP(g, gangfit::launch_shard_push(c->g_drv_loc.ptr, drv_all, (size_t)s * n_apps * sizeof(gf_shard_driver),
                                             (size_t)n_apps * sizeof(gf_shard_driver), c->stream));
        if (several_streams) GF_HIP(g, hipEventRecord(c->g_ev[1], c->stream));
    }
    if (use_rccl) {
        if (rccl_all_gather([](gf_ctx* c) { return (const void*)c->g_drv_loc.ptr; }, [](gf_ctx* c) { return (void*)c->g_drv_all.ptr; },
                            (size_t)n_apps * sizeof(gf_shard_driver)) != 0)
            return fail(g, GF_ERR_HIP, "ncclAllGather of the driver records failed");
    } else {
        if (several_streams) GF_HIP(g, everyone_waits(1));
    }
    // ---- step 3: every shard emits its slice of the placements
    for (uint32_t s = 0; s < S; ++s) {
        gf_ctx* c = g->group[s];
        GF_HIP(g, hipSetDevice(c->device));
        GF_HIP(g, gangfit::launch_shard_emit(algo, make_table(c, c->d_snap.ptr), range[s], n_apps, c->d_apps.ptr, c->g_part_all.ptr,
                                             c->g_drv_all.ptr, c->d_results.ptr, c->g_exec2.ptr, half, c->stream));
        if (several_streams) GF_HIP(g, hipEventRecord(c->g_ev[2], c->stream));
    }
    // ---- step 4 on the first device only: sum of the slices (each entry written by exactly one shard), finish, D2H
    if (use_rccl) {  // the reduction north_star names: sum of the placement slices onto the first device, over xGMI
        if (rccl().GroupStart() != 0) return fail(g, GF_ERR_HIP, "ncclGroupStart failed");
        int bad = 0;
        for (uint32_t s = 0; s < S; ++s) {
            gf_ctx* c = g->group[s];
            GF_HIP(g, hipSetDevice(c->device));
            bad |= rccl().Reduce(c->g_exec2.ptr, first->g_exec2.ptr, (size_t)(2 * half), Rccl::kUint32, Rccl::kSum, 0, g->g_comms[s], c->stream);
        }
        if ((rccl().GroupEnd() | bad) != 0) return fail(g, GF_ERR_HIP, "ncclReduce of the placements failed");
        GF_HIP(g, hipSetDevice(first->device));
    } else {
        GF_HIP(g, hipSetDevice(first->device));
        for (uint32_t s = 1; s < S; ++s)
            if (g->group[s]->stream != first->stream) GF_HIP(g, hipStreamWaitEvent(first->stream, g->group[s]->g_ev[2], 0));
        if (g->g_fault != 1)  // fault injection: the other shards' placement slices never arrive
            GF_HIP(g, gangfit::launch_shard_reduce_pull(exec_others, first->g_exec2.ptr, (size_t)(2 * half), first->stream));
    }
    GF_HIP(g, gangfit::launch_shard_finish(algo, S, n_apps, first->d_apps.ptr, first->g_part_all.ptr, first->g_drv_all.ptr,
                                           first->d_results.ptr, first->g_exec2.ptr, half, first->stream));
    GF_HIP(g, hipMemcpyAsync(g->h_results.ptr, first->d_results.ptr, (size_t)n_apps * sizeof(gf_result), hipMemcpyDeviceToHost, first->stream));
    if (total_k)
        GF_HIP(g, hipMemcpyAsync(g->h_exec.ptr, first->g_exec2.ptr, (size_t)total_k * sizeof(uint32_t), hipMemcpyDeviceToHost, first->stream));
    GF_HIP(g, gf_wait_stream(first->stream));
    std::memcpy(results, g->h_results.ptr, (size_t)n_apps * sizeof(gf_result));
    if (total_k) std::memcpy(exec_nodes, g->h_exec.ptr, (size_t)total_k * sizeof(uint32_t));
    // ---- self-check: the first sharded batch on every newly installed snapshot is also answered by the first device alone.
    //      A wrong exchange (peer stores that did not land, a collective that reduced something else) must not decide a
    //      Filter: on a mismatch the context stops sharding, says why, and serves the first device's answer.
    if (g->g_verify && first->snap_epoch != g->g_verified_epoch) {
        std::vector<gf_result> ref_res(n_apps);
        std::vector<uint32_t> ref_exec((size_t)total_k + 1);
        const int rc = gf_fit_batch(first, mode, algo, n_apps, apps, ref_res.data(), ref_exec.data(), total_k, nullptr);
        if (rc != GF_OK) {
            g->err = first->err;
            return rc;
        }
        bool same = std::memcmp(ref_res.data(), results, (size_t)n_apps * sizeof(gf_result)) == 0;
        for (uint32_t a = 0; a < n_apps && same; ++a)
            if (ref_res[a].has_capacity)
                same = std::memcmp(ref_exec.data() + g->h_apps.ptr[a].exec_off, exec_nodes + g->h_apps.ptr[a].exec_off,
                                   (size_t)ref_res[a].exec_len * sizeof(uint32_t)) == 0;
        if (same) {
            g->g_verified_epoch = first->snap_epoch;
        } else {
            g->g_shard_off = true;
            g->err = "the node-range sharded batch disagreed with the first device's own answer: sharding is off for this context";
            std::memcpy(results, ref_res.data(), (size_t)n_apps * sizeof(gf_result));
            if (total_k) std::memcpy(exec_nodes, ref_exec.data(), (size_t)total_k * sizeof(uint32_t));
        }
    }
    return GF_OK;
}

}  // namespace
