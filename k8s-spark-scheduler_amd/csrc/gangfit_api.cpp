// gangfit_api.cpp — the C ABI of libgangfit (include/gangfit.h): context life cycle, options, probes, recorded launch
// sequences, timers / counters / self-test.  The other entry points live in gangfit_api_{snapshot,fit,worker,group}.cpp
// (map: gangfit_ctx.h).
//
// Host side only: builds the slot-ordered node table the kernels scan (gangfit_device.h), moves app records and
// results through pinned staging buffers and serialises callers per context.  No CPU fallback lives here: when the
// device path cannot serve a call the function returns < 0 and the caller (the Go shim) decides what to do.
#include "gangfit_ctx.h"

using namespace gfapi;

namespace gfapi {

int fail(gf_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
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
    v->eff_nonneg = p->eff_nonneg;
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

}  // namespace gfapi

extern "C" {

int gf_version(void) { return GF_VERSION; }

int gf_init(const int* device_ids, int n_dev, gf_ctx** out) {
    if (!out) return GF_ERR_INVALID;
    *out = nullptr;
    if (n_dev > 1) {
        // One context over several devices: the priority order is cut into n_dev ranges (shards), id i names the device that
        // scans range i.  One sub-context per DEVICE: a device whose id repeats (how the path is exercised on a one-GPU box)
        // hosts several shards in one sub-context — one snapshot copy, one stream, one launch per step with a grid row per
        // shard.  GANGFIT_TEST_GROUP_SPLIT=1 (tests) gives every listed id a sub-context, a stream and a submitting thread
        // of its own instead: a repeated id then exercises what distinct devices exercise — events between streams, the
        // host-side barriers of the submitting threads, pushes into several gathered tables, the pull of the placements.
        if (!device_ids || n_dev > (int)gangfit::kMaxGroupDevices) return GF_ERR_INVALID;
        const bool split = std::getenv("GANGFIT_TEST_GROUP_SPLIT") != nullptr;
        gf_ctx* g = new (std::nothrow) gf_ctx();
        if (!g) return GF_ERR_HIP;
        g->device = device_ids[0];
        g->g_total_shards = (uint32_t)n_dev;
        for (int i = 0; i < n_dev; ++i) {
            gf_ctx* host = nullptr;
            if (!split)
                for (gf_ctx* earlier : g->group)
                    if (earlier->device == device_ids[i]) host = earlier;
            if (host == nullptr) {
                const int rc = gf_init(&device_ids[i], 1, &host);
                if (rc != GF_OK) {
                    gf_destroy(g);
                    return rc;
                }
                host->n_shards = (uint32_t)n_dev;
                // what the other devices store into / read from lives in fine-grained memory: a posted peer store must not
                // depend on what a kernel boundary does to this device's caches
                host->g_part_all.fine = host->g_drv_all.fine = host->g_exec2.fine = true;
                g->group.push_back(host);
                g->g_devices.push_back(device_ids[i]);
                bool ok = hipSetDevice(host->device) == hipSuccess;
                for (hipEvent_t& e : host->g_ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
                if (!ok) {
                    gf_destroy(g);
                    return GF_ERR_HIP;
                }
            }
            host->my_shards.push_back((uint32_t)i);
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
        if (g->group.size() > 1) g->g_pool = new (std::nothrow) gfapi::GroupPool((uint32_t)g->group.size());
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
        v->feasible_announce = parent->feasible_announce;
        v->zoned_fused = parent->zoned_fused;
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
        delete ctx->g_pool;
        ctx->g_pool = nullptr;
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
    ctx->chain.d_tip.release();
    ctx->d_flag32.release();
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
    ctx->d_xnzone.release();
    ctx->d_xqzone.release();
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
    ctx->d_feas_exec.release();
    ctx->d_feas_scratch.release();
    ctx->d_feas_zexec.release();
    ctx->d_failed.release();
    ctx->d_stats.release();
    ctx->h_table.release();
    ctx->h_index.release();
    ctx->h_apps.release();
    ctx->h_results.release();
    ctx->h_exec.release();
    ctx->h_failed.release();
    ctx->h_feasible.release();
    ctx->d_feasible_sync.release();
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
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
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
            if (ctx->g_total_shards != ctx->group.size())
                return fail(ctx, GF_ERR_UNSUPPORTED, "a device hosts several shards: the RCCL exchange wants one rank per physical device");
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
    } else if (k == "feasible_announce") {
        ctx->feasible_announce = value != 0;
    } else if (k == "zoned_fused") {
        ctx->zoned_fused = value != 0;
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
            if (value < 0 || value > 16) return fail(ctx, GF_ERR_INVALID, "worker_sets outside [0, 16]");
            ctx->worker.sets = (uint32_t)value;
        } else if (k == "worker_blocks_per_set") {
            if (value < 0 || value > 1024) return fail(ctx, GF_ERR_INVALID, "worker_blocks_per_set outside [0, 1024]");
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
    return (int)ctx->g_total_shards;
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

int gf_call_phases(gf_ctx* ctx, double out_us[5]) {
    GF_DELEGATE(ctx, gf_call_phases(ctx, out_us));
    if (!ctx || !out_us) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    for (int i = 0; i < 5; ++i) out_us[i] = ctx->call_phase_us[i];
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

int gf_chain_profile(gf_ctx* ctx, uint64_t out[12]) {
    GF_DELEGATE(ctx, gf_chain_profile(ctx, out));
    if (!ctx || !out) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_HIP(ctx, hipSetDevice(ctx->device));
    GF_HIP(ctx, gf_wait_stream(ctx->stream));
    ScanStats s;
    GF_HIP(ctx, hipMemcpy(&s, ctx->d_stats.ptr, sizeof s, hipMemcpyDeviceToHost));
    for (int i = 0; i < 5; ++i) {
        out[i] = s.fifo_rare_count[i];
        out[5 + i] = s.fifo_rare_cycles[i];
    }
    out[10] = s.fifo_hw_id & 0xFFFFFFFFull;
    out[11] = s.fifo_hw_id >> 32;
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
