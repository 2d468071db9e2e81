// gangfit_api_group.cpp — node-range sharding: the gf_shard_* steps for one process per GPU, and gf_fit_batch on a multi-device
// context (one gf_ctx over several devices: exchanges by peer stores over xGMI or by RCCL).
#include "gangfit_ctx.h"

using namespace gfapi;

namespace gfapi {

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
    // the range's sub-slots of the sparse gpu view (gangs of gpu executors are summed and emitted from it: gangfit_shard.inc)
    r->g_lo = r->g_hi = 0;
    if (ctx->n_g != 0 && !ctx->g_prefix.empty()) {
        const size_t last = ctx->g_prefix.size() - 1;
        r->g_lo = ctx->g_prefix[r->c_lo < last ? r->c_lo : last];
        r->g_hi = ctx->g_prefix[r->c_hi < last ? r->c_hi : last];
    }
    return GF_OK;
}
// (a context without the prefix table — a view — takes the full order)
static gangfit::SparseTable shard_sparse(gf_ctx* ctx) {
    return (ctx->n_g != 0 && !ctx->g_prefix.empty()) ? make_sparse(ctx) : gangfit::SparseTable{};
}

// ---- the submitting threads (GroupPool, gangfit_ctx.h)
GroupPool::GroupPool(uint32_t n_devices) : parties(n_devices) {
    for (uint32_t d = 1; d < n_devices; ++d) threads.emplace_back([this, d] { worker(d); });
}
GroupPool::~GroupPool() {
    {
        std::lock_guard<std::mutex> l(m);
        quit = true;
        generation.fetch_add(1, std::memory_order_release);
    }
    cv.notify_all();
    for (std::thread& t : threads) t.join();
}
void GroupPool::worker(uint32_t d) {
    uint64_t seen = 0;
    for (;;) {
        // a short spin (batches that follow each other find the thread awake), then the condition variable
        const auto t0 = std::chrono::steady_clock::now();
        while (generation.load(std::memory_order_acquire) == seen) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return generation.load(std::memory_order_acquire) != seen; });
                break;
            }
        }
        seen = generation.load(std::memory_order_acquire);
        if (quit) return;
        (*job)(d);
        remaining.fetch_sub(1, std::memory_order_acq_rel);
    }
}
void GroupPool::run(const std::function<void(uint32_t)>& j) {
    job = &j;
    remaining.store(parties - 1, std::memory_order_release);
    {
        std::lock_guard<std::mutex> l(m);
        generation.fetch_add(1, std::memory_order_release);
    }
    cv.notify_all();
    j(0);
    while (remaining.load(std::memory_order_acquire) != 0) std::this_thread::yield();
}
void GroupPool::barrier() {
    const uint32_t gen = bar_gen.load(std::memory_order_acquire);
    if (bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == parties) {
        bar_count.store(0, std::memory_order_relaxed);
        bar_gen.fetch_add(1, std::memory_order_release);
    } else {
        uint32_t spins = 0;
        while (bar_gen.load(std::memory_order_acquire) == gen)
            if ((++spins & 0xFFFu) == 0) std::this_thread::yield();
    }
}

// gf_fit_batch on a multi-device context.  Independent batches of the two plain packers are node-range sharded across the
// sub-contexts (the four steps of gangfit_shard.inc with the three exchanges done by peer access — the
// producing kernels write straight into every device's gathered table, shard_reduce_pull_kernel collects the placements — or
// by RCCL); everything else — FIFO chains (each commit must be visible to the next scan), the zone-aware and
// minimal-fragmentation packers, orders that do not merge — runs on the first device.
//
// One sub-context = one DEVICE and the shards it hosts: per step ONE launch per device (a grid row per hosted shard), one
// upload of the records, one gathered table and one placement buffer per device.  With several devices device d's calls are
// issued by submitting thread d (GroupPool); between the steps the threads meet at a host barrier, because stream t may only
// be told to wait for event s once event s has been recorded.
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
    const uint32_t D = (uint32_t)g->group.size();  // devices (sub-contexts)
    const uint32_t S = g->g_total_shards;          // shards of the priority order
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
    // what the reduce of the placement buffers carries: its second half (the capacities of pass 1's nodes) is distribute-evenly's
    const size_t reduce_words = (size_t)(algo == GF_ALGO_DISTRIBUTE_EVENLY ? 2 * half : half);
    GF_HIP(g, g->h_results.reserve(n_apps));
    GF_HIP(g, g->h_exec.reserve(total_k + 1));
    // ---- buffers on every device (growth only: no-ops from the second batch of a size on) and what each device's kernels
    //      must know about the others: every gathered table, every placement buffer
    gangfit::ShardSet set[gangfit::kMaxGroupDevices];
    gangfit::PeerPtrs part_all{}, drv_all{}, exec_others{};
    for (uint32_t d = 0; d < D; ++d) {
        gf_ctx* c = g->group[d];
        GF_HIP(g, hipSetDevice(c->device));
        set[d] = gangfit::ShardSet{};
        set[d].n_shards = S;
        for (uint32_t sh : c->my_shards) {
            c->shard = sh;
            c->n_shards = S;
            gangfit::ShardRange r{};
            if (const int rc = shard_ready(c, algo, &r); rc != GF_OK) {
                g->err = c->err;
                return rc;
            }
            const uint32_t q = set[d].n++;
            set[d].c_lo[q] = r.c_lo;
            set[d].c_hi[q] = r.c_hi;
            set[d].shard[q] = sh;
            set[d].g_lo[q] = r.g_lo;
            set[d].g_hi[q] = r.g_hi;
        }
        GF_HIP(g, c->d_apps.reserve(n_apps));
        GF_HIP(g, c->d_results.reserve(n_apps));
        GF_HIP(g, c->g_part_loc.reserve((size_t)c->my_shards.size() * n_apps));
        GF_HIP(g, c->g_drv_loc.reserve((size_t)c->my_shards.size() * n_apps));
        GF_HIP(g, c->g_part_all.reserve((size_t)S * n_apps));
        GF_HIP(g, c->g_drv_all.reserve((size_t)S * n_apps));
        GF_HIP(g, c->g_exec2.reserve(2 * half));
        part_all.p[d] = c->g_part_all.ptr;
        drv_all.p[d] = c->g_drv_all.ptr;
        if (d > 0) exec_others.p[exec_others.n++] = c->g_exec2.ptr;
    }
    part_all.n = drv_all.n = D;
    const bool use_rccl = g->g_comms.size() == D && D == S;
    const gangfit::PeerPtrs no_peers{};
    std::vector<int> rc_of(D, GF_OK);
    auto hip_ok = [&](uint32_t d, hipError_t e, const char* what) {
        if (e == hipSuccess || rc_of[d] != GF_OK) return e == hipSuccess;
        rc_of[d] = fail(g->group[d], GF_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
        return false;
    };
#define GF_STEP(d, call) hip_ok((d), (call), #call)
    // the steps of ONE device, each issued on that device's stream by whichever thread runs them
    auto step_partials = [&](uint32_t d) {
        gf_ctx* c = g->group[d];
        if (!GF_STEP(d, hipSetDevice(c->device))) return;
        if (!GF_STEP(d, hipMemcpyAsync(c->d_apps.ptr, g->h_apps.ptr, (size_t)n_apps * sizeof(gf_app), hipMemcpyHostToDevice, c->stream))) return;
        if (!GF_STEP(d, gangfit::launch_shard_partials(algo, make_table(c, c->d_snap.ptr), shard_sparse(c), set[d], n_apps, c->d_apps.ptr, c->g_part_loc.ptr,
                                                       use_rccl ? no_peers : part_all, c->stream)))
            return;
        if (g->g_fault == 2 && d > 0 && !use_rccl) {  // fault injection: this device's capacity sums arrive as zeros everywhere
            for (uint32_t t = 0; t < D; ++t)
                for (uint32_t sh : c->my_shards)
                    (void)GF_STEP(d, hipMemsetAsync(static_cast<gf_shard_partial*>(part_all.p[t]) + (size_t)sh * n_apps, 0,
                                                   (size_t)n_apps * sizeof(gf_shard_partial), c->stream));
        } else if (g->g_fault == 2 && d > 0) {
            (void)GF_STEP(d, hipMemsetAsync(c->g_part_loc.ptr, 0, (size_t)n_apps * sizeof(gf_shard_partial), c->stream));
        }
        if (D > 1 && !use_rccl) (void)GF_STEP(d, hipEventRecord(c->g_ev[0], c->stream));
    };
    auto wait_others = [&](uint32_t d, int which) {  // this device's stream continues only behind event `which` of every other
        gf_ctx* c = g->group[d];
        for (uint32_t s2 = 0; s2 < D; ++s2)
            if (s2 != d && !GF_STEP(d, hipStreamWaitEvent(c->stream, g->group[s2]->g_ev[which], 0))) return;
    };
    auto step_drivers = [&](uint32_t d) {
        gf_ctx* c = g->group[d];
        if (rc_of[d] != GF_OK || !GF_STEP(d, hipSetDevice(c->device))) return;
        if (D > 1 && !use_rccl) wait_others(d, 0);
        if (!GF_STEP(d, gangfit::launch_shard_drivers(make_table(c, c->d_snap.ptr), set[d], n_apps, c->d_apps.ptr, c->g_part_all.ptr,
                                                      c->g_drv_loc.ptr, use_rccl ? no_peers : drv_all, c->stream)))
            return;
        if (D > 1 && !use_rccl) (void)GF_STEP(d, hipEventRecord(c->g_ev[1], c->stream));
    };
    auto step_emit = [&](uint32_t d) {
        gf_ctx* c = g->group[d];
        if (rc_of[d] != GF_OK || !GF_STEP(d, hipSetDevice(c->device))) return;
        if (D > 1 && !use_rccl) wait_others(d, 1);
        if (!GF_STEP(d, gangfit::launch_shard_emit(algo, make_table(c, c->d_snap.ptr), shard_sparse(c), set[d], n_apps, c->d_apps.ptr, c->g_part_all.ptr,
                                                   c->g_drv_all.ptr, c->d_results.ptr, c->g_exec2.ptr, half, c->stream)))
            return;
        if (D > 1) (void)GF_STEP(d, hipEventRecord(c->g_ev[2], c->stream));
    };
    auto step_finish = [&]() {  // the first device only: sum of the slices (each entry written by exactly one shard), finish, D2H
        if (rc_of[0] != GF_OK || !GF_STEP(0, hipSetDevice(first->device))) return;
        if (!use_rccl) {
            for (uint32_t s2 = 1; s2 < D; ++s2)
                if (!GF_STEP(0, hipStreamWaitEvent(first->stream, g->group[s2]->g_ev[2], 0))) return;
            if (g->g_fault != 1 && D > 1)  // fault injection: the other devices' placement slices never arrive
                if (!GF_STEP(0, gangfit::launch_shard_reduce_pull(exec_others, first->g_exec2.ptr, reduce_words, first->stream))) return;
        }
        if (!GF_STEP(0, gangfit::launch_shard_finish(algo, S, n_apps, first->d_apps.ptr, first->g_part_all.ptr, first->g_drv_all.ptr,
                                                     first->d_results.ptr, first->g_exec2.ptr, half, first->stream)))
            return;
        if (!GF_STEP(0, hipMemcpyAsync(g->h_results.ptr, first->d_results.ptr, (size_t)n_apps * sizeof(gf_result), hipMemcpyDeviceToHost, first->stream))) return;
        if (total_k && !GF_STEP(0, hipMemcpyAsync(g->h_exec.ptr, first->g_exec2.ptr, (size_t)total_k * sizeof(uint32_t), hipMemcpyDeviceToHost, first->stream))) return;
        (void)GF_STEP(0, gf_wait_stream(first->stream));
    };
    // An error must not return while the other devices' kernels, their peer stores into every gathered table, or pending
    // reads of the pinned records are still running: the next batch reuses all of those buffers.  Every stream is waited
    // for first; when even that fails the context stops sharding and serves from its first device.
    auto drain_all = [&]() {
        bool ok = true;
        for (uint32_t d = 0; d < D; ++d) {
            gf_ctx* c = g->group[d];
            if (hipSetDevice(c->device) != hipSuccess || gf_wait_stream(c->stream) != hipSuccess) ok = false;
        }
        (void)hipGetLastError();
        (void)hipSetDevice(first->device);
        if (!ok) g->g_shard_off = true;
    };
    auto fail_drained = [&](const char* what) {
        drain_all();
        return fail(g, GF_ERR_HIP, "%s", what);
    };
    if (use_rccl) {
        // RCCL exchange (one shard per device): every device's collective is enqueued on its own stream inside one group call
        // issued by THIS thread; the library orders the streams against each other
        auto rccl_all_gather = [&](auto loc, auto all, size_t bytes_each) -> int {
            if (rccl().GroupStart() != 0) return -1;
            int bad = 0;
            for (uint32_t s2 = 0; s2 < D; ++s2) {
                gf_ctx* c = g->group[s2];
                if (hipSetDevice(c->device) != hipSuccess) bad = 1;
                bad |= rccl().AllGather(loc(c), all(c), bytes_each, Rccl::kChar, g->g_comms[s2], c->stream);
            }
            return rccl().GroupEnd() | bad;
        };
        for (uint32_t d = 0; d < D; ++d) step_partials(d);
        if (rccl_all_gather([](gf_ctx* c) { return (const void*)c->g_part_loc.ptr; }, [](gf_ctx* c) { return (void*)c->g_part_all.ptr; },
                            (size_t)n_apps * sizeof(gf_shard_partial)) != 0)
            return fail_drained("ncclAllGather of the capacity sums failed");
        for (uint32_t d = 0; d < D; ++d) step_drivers(d);
        if (rccl_all_gather([](gf_ctx* c) { return (const void*)c->g_drv_loc.ptr; }, [](gf_ctx* c) { return (void*)c->g_drv_all.ptr; },
                            (size_t)n_apps * sizeof(gf_shard_driver)) != 0)
            return fail_drained("ncclAllGather of the driver records failed");
        for (uint32_t d = 0; d < D; ++d) step_emit(d);
        // the reduction north_star names: sum of the placement slices onto the first device, over xGMI
        if (rccl().GroupStart() != 0) return fail_drained("ncclGroupStart failed");
        int bad = 0;
        for (uint32_t d = 0; d < D; ++d) {
            gf_ctx* c = g->group[d];
            if (hipSetDevice(c->device) != hipSuccess) bad = 1;
            bad |= rccl().Reduce(c->g_exec2.ptr, first->g_exec2.ptr, reduce_words, Rccl::kUint32, Rccl::kSum, 0, g->g_comms[d], c->stream);
        }
        if ((rccl().GroupEnd() | bad) != 0) return fail_drained("ncclReduce of the placements failed");
        step_finish();
    } else if (D == 1 || g->g_pool == nullptr) {
        for (uint32_t d = 0; d < D; ++d) step_partials(d);
        for (uint32_t d = 0; d < D; ++d) step_drivers(d);
        for (uint32_t d = 0; d < D; ++d) step_emit(d);
        step_finish();
    } else {
        // one submitting thread per device; a host barrier behind every step that records an event another device waits for
        GroupPool& pool = *g->g_pool;
        const std::function<void(uint32_t)> job = [&](uint32_t d) {
            step_partials(d);
            pool.barrier();
            step_drivers(d);
            pool.barrier();
            step_emit(d);
            pool.barrier();
            if (d == 0) step_finish();
        };
        pool.run(job);
    }
#undef GF_STEP
    for (uint32_t d = 0; d < D; ++d)
        if (rc_of[d] != GF_OK) {
            drain_all();  // (step_finish returned without waiting for anything when an earlier step had failed)
            g->err = g->group[d]->err;
            return rc_of[d];
        }
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

}  // namespace gfapi

extern "C" {

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
    GF_HIP(ctx, gangfit::launch_shard_partials(algo, make_table(ctx, ctx->d_snap.ptr), shard_sparse(ctx), gangfit::shard_set_of(r), n_apps, d_apps, d_out,
                                               gangfit::PeerPtrs{}, st));
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
    GF_HIP(ctx, gangfit::launch_shard_drivers(make_table(ctx, ctx->d_snap.ptr), gangfit::shard_set_of(r), n_apps, d_apps, d_all_partials,
                                              d_out, gangfit::PeerPtrs{}, st));
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
    GF_HIP(ctx, gangfit::launch_shard_emit(algo, make_table(ctx, ctx->d_snap.ptr), shard_sparse(ctx), gangfit::shard_set_of(r), n_apps, d_apps, d_all_partials,
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

}  // extern "C"
