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
    return GF_OK;
}

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

}  // extern "C"
