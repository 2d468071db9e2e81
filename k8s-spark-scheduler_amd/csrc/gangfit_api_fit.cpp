// gangfit_api_fit.cpp — the decision side of the C ABI: table views, the launches of every packer (independent batches, FIFO chains on the LDS
// kernels and their fallbacks), the incremental chain cache, gf_fit_batch / gf_fit_batch_dev / gf_spark_binpack, single
// executors, findNodes and the packing efficiencies.
#include "gangfit_ctx.h"

using namespace gfapi;

namespace gfapi {

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
    // the snapshot's scaled int32 twin travels with the SNAPSHOT only (a chain's working copy has its own, in the chain's units)
    if (ctx->narrow_ok && base == ctx->d_snap.ptr && ctx->d_nsnap.ptr != nullptr) {
        t.ncpu = ctx->d_nsnap.ptr;
        t.nmem = t.ncpu + ctx->n_slots;
        t.ngpu = t.nmem + ctx->n_slots;
        for (int j = 0; j < 3; ++j) t.nunit[j] = ctx->unit[j] > 0 ? ctx->unit[j] : 1;
    }
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
    g.zmask = g.xmask + g.n_chunks;
    g.slot_of_sub = g.sub_of_slot + ctx->n_slots;
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
    bool from_tip = false;       // the launch starts from the cached chain's tip (a_begin = its tip_at, any value) instead of a checkpoint
    bool write_tip = false;      // the launch leaves its own tip in ctx->chain.d_tip
};

// The checkpoint arguments of a chain kernel and the table it starts from.
gangfit::ChainCkpt chain_ckpt_args(gf_ctx* ctx, const ChainRun* run, const int32_t** restore) {
    gangfit::ChainCkpt ck{nullptr, run ? run->a_begin : 0u, ctx->chain.shift, ctx->chain.slot_words, nullptr, nullptr};
    *restore = nullptr;
    if (run != nullptr && (run->record || run->a_begin > 0)) {
        ck.base = ctx->chain.d_ckpt.ptr;
        if (run->write_tip) ck.tip = ctx->chain.d_tip.ptr;
        if (run->a_begin > 0 && run->from_tip) {
            *restore = ctx->chain.d_tip.ptr;
        } else if (run->a_begin > 0) {
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
    if (mode == GF_MODE_INDEPENDENT && ctx->zoned_fused && nz + 1 <= 64) {
        // one launch: a workgroup per application decides every candidate view, chooses and writes the final answer
        // (fit_zoned_fused_kernel) — d_apps / d_results / d_exec_nodes may be device-mapped host memory (gf_fit_batch)
        GF_HIP(ctx, ctx->d_zexec.reserve(((uint64_t)nz + 1) * half));
        gangfit::ZoneTable zt{ctx->d_zmasks.ptr, ctx->d_zmasks.ptr + (size_t)ctx->zd_row0 * ctx->zstride, nz, ctx->zstride};
        if (const int arc = apps_to_device(ctx, stream); arc != GF_OK) return arc;
        GF_HIP(ctx, gangfit::launch_fit_zoned_fused(inner, algo == GF_ALGO_AZ_AWARE_TIGHTLY_PACK, make_table(ctx, ctx->d_snap.ptr), make_sparse(ctx), zt,
                                                    ctx->d_sched.ptr, ctx->d_zexec.ptr, half, n_apps, d_apps, d_results,
                                                    d_exec_nodes, ctx->d_scratch.ptr, half, stream));
        return GF_OK;
    }
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
    // The cached chain's TIP (whole table in LDS): the table before the application that chain ended at.  A queue
    // that agrees with the cached one up to there — the Filter of the next driver in creation order, the same Filter again —
    // resumes from it and evaluates one or two applications instead of everything since the last checkpoint.  Not when this
    // chain would cross a checkpoint boundary: it then starts from the checkpoint, so that the boundary's dump is made (the
    // kernel dumps where a refill of its staging buffer falls, every 32 applications from ITS first one).
    const bool tip_possible = table_in_lds;  // (every LDS chain kernel leaves a tip when its whole table sits in LDS)
    bool from_tip = false;
    if (same && tip_possible && C.tip_valid && C.d_tip.ptr != nullptr && C.tip_at > a_begin && C.tip_at <= common &&
        ((n_apps - 1) >> shift) == (C.tip_at >> shift)) {
        a_begin = C.tip_at;
        from_tip = true;
    }
    if (tip_possible && slot_words > C.d_tip.cap) {
        if (C.d_tip.ptr) (void)gf_wait_stream(ctx->stream);
        if (C.d_tip.reserve(slot_words) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        C.tip_valid = false;
        if (from_tip) {  // (cannot happen: a valid tip lives in a buffer of this size)
            from_tip = false;
            a_begin = 0;
        }
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
    run->from_tip = from_tip;
    run->write_tip = tip_possible;
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
    C.tip_valid = run.write_tip;  // the epilogue left the table before application `last`
    C.tip_at = last;
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
           const ChainRun* run = nullptr) {
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
                                                    d_results, d_exec_nodes, ctx->d_scratch.ptr, half, stats, stream));
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

}  // namespace gfapi

extern "C" {

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
    if (ctx->zero_copy && mode == GF_MODE_INDEPENDENT && ctx->have_orders &&
        (!is_zone_algo(algo) || (ctx->zoned_fused && ctx->have_sched && ctx->n_zones + 1 <= 64)) &&
        (uint64_t)n_apps * sizeof(gf_app) + total_k * sizeof(uint32_t) <= (UINT64_C(4) << 20)) {
        void *da = ctx->h_apps.dev, *dr = ctx->h_results.dev, *de = ctx->h_exec.dev;
        if (da != nullptr && dr != nullptr && de != nullptr) {
            using clk = std::chrono::steady_clock;
            const auto t_staged = clk::now();
            const int rc0 = launch(ctx, mode, algo, n_apps, ctx->h_apps.ptr, static_cast<const gf_app*>(da),
                                   static_cast<gf_result*>(dr), static_cast<uint32_t*>(de), total_k, ctx->d_failed.ptr, st);
            if (rc0 != GF_OK) return rc0;
            const auto t_launched = clk::now();
            GF_HIP(ctx, gf_wait_stream(st));
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

int gf_fit_feasible(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* apps, uint8_t* has_capacity) {
    if (!ctx) return GF_ERR_INVALID;
    if (n_apps > 0 && (!apps || !has_capacity)) return fail(ctx, GF_ERR_INVALID, "apps/has_capacity must not be NULL");
    if (n_apps == 0) return GF_OK;
    const bool plain = algo == GF_ALGO_TIGHTLY_PACK || algo == GF_ALGO_DISTRIBUTE_EVENLY || algo == GF_ALGO_MINIMAL_FRAGMENTATION;
    // the zone-aware packers as ONE launch (fit_zoned_fused_kernel: a workgroup per application decides every candidate view
    // and chooses): served here too; their four-kernel route is not
    const bool fused_zoned = is_zone_algo(algo) && ctx->zoned_fused && ctx->have_sched && ctx->n_zones + 1 <= 64;
    if (!ctx->group.empty() || !(plain || fused_zoned)) {
        // a multi-device context, or a route without a feasibility-only kernel: the full batch, of which only HasCapacity is
        // handed on
        uint64_t total_k = 0;
        for (uint32_t a = 0; a < n_apps; ++a) {  // validated HERE: the sizes below come from it (gf_fit_batch checks the rest)
            if (apps[a].k < 0 || apps[a].k > GF_MAX_K)
                return fail(ctx, GF_ERR_INVALID, "apps[%u].k = %d outside [0, %d]", a, apps[a].k, GF_MAX_K);
            total_k += (uint64_t)apps[a].k;
        }
        std::vector<gf_result> res;
        std::vector<uint32_t> exec;
        try {  // no exception crosses the C ABI
            res.resize(n_apps);
            exec.resize((size_t)total_k + 1);
        } catch (const std::exception&) {
            return fail(ctx, GF_ERR_CAPACITY, "gf_fit_feasible: no host memory for %u results + %llu placements", n_apps,
                        (unsigned long long)total_k);
        }
        const int rc = gf_fit_batch(ctx, GF_MODE_INDEPENDENT, algo, n_apps, apps, res.data(), exec.data(), total_k, nullptr);
        if (rc != GF_OK) return rc;
        for (uint32_t a = 0; a < n_apps; ++a) has_capacity[a] = res[a].has_capacity ? 1 : 0;
        return GF_OK;
    }
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (n_apps >= 0x80000000u) return fail(ctx, GF_ERR_INVALID, "n_apps = %u", n_apps);
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede a fit");
    using clk = std::chrono::steady_clock;
    const auto t_entry = clk::now();
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
        o.exec_off = total_k;  // the placements are still made (same decision code): into device memory, where they stay
        total_k += (uint64_t)in.k;
    }
    const uint64_t half = total_k + 1;
    GF_HIP(ctx, ctx->d_feas_exec.reserve(total_k + 1));  // private to this entry point: see gf_ctx::d_feas_exec
    GF_HIP(ctx, ctx->d_feas_scratch.reserve(2 * half));
    GF_HIP(ctx, ctx->h_feasible.reserve((size_t)n_apps + 4));  // the kernel writes whole words
    hipStream_t st = ctx->stream;
    {  // the collection words start at zero; the kernel's collecting workgroup leaves them at zero again
        const uint32_t* before = ctx->d_feasible_sync.ptr;
        GF_HIP(ctx, ctx->d_feasible_sync.reserve(((size_t)n_apps + 3) / 4 + 4));
        if (ctx->d_feasible_sync.ptr != before || ctx->feasible_sync_dirty) {
            GF_HIP(ctx, hipMemsetAsync(ctx->d_feasible_sync.ptr, 0, ctx->d_feasible_sync.cap * sizeof(uint32_t), st));
            ctx->feasible_sync_dirty = false;
        }
    }
    const gf_app* d_apps = ctx->h_apps.dev;
    uint8_t* d_feas = ctx->h_feasible.dev;
    const bool mapped = ctx->zero_copy && d_apps != nullptr && d_feas != nullptr &&
                        (uint64_t)n_apps * sizeof(gf_app) <= (UINT64_C(4) << 20);
    DeviceBuf<uint8_t> staged;  // (hosts without mapped pinned memory, or very large batches: two copies around the kernel)
    if (!mapped) {
        GF_HIP(ctx, ctx->d_apps.reserve(n_apps));
        GF_HIP(ctx, staged.reserve((size_t)n_apps + 4));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_apps.ptr, ctx->h_apps.ptr, (size_t)n_apps * sizeof(gf_app), hipMemcpyHostToDevice, st));
        d_apps = ctx->d_apps.ptr;
        d_feas = staged.ptr;
    }
    // Mapped staging: the answers announce themselves.  Every byte of the pinned array is preset to "not yet"; the kernel's
    // collecting workgroup writes the whole array with system-scope (written-through) stores, and the caller watches the bytes
    // arrive instead of waiting for the kernel-end write-back and the stream's completion signal (6-8 us of a 12-15 us wait:
    // profiles/r5h_feasible_call.txt).  A byte that has arrived is final, so no ordering between them is needed; when all have
    // arrived every wavefront has read its record and made its decision — what the kernel still owes (placement stores to device
    // memory, its end) goes to buffers only this entry point uses (d_feas_*), and the next gf_fit_feasible is ordered behind it
    // on the context's stream: an entry point that launches on a caller's stream, or copies on the null stream, shares nothing
    // with the tail of this kernel.  gf_fit_batch cannot do the same:
    // 13 000 placement words written through over the host link cost more than the signal they replace (round 4: 30.3 us
    // against 23.3).
    constexpr uint8_t kNotYet = 0xFF;
    // (GANGFIT_WAIT=block: the host cannot spare a core for the duration of a call — no spinning on the answers either)
    const bool announce = mapped && ctx->feasible_announce && !wait_blocking();
    if (announce) std::memset(ctx->h_feasible.ptr, kNotYet, n_apps);
    const auto t_staged = clk::now();
    hipError_t e;
    if (fused_zoned) {
        const uint32_t nz = ctx->n_zones;
        const int inner = algo == GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION ? GF_ALGO_MINIMAL_FRAGMENTATION : GF_ALGO_TIGHTLY_PACK;
        e = ctx->d_feas_zexec.reserve(((uint64_t)nz + 1) * half);
        gangfit::ZoneTable zt{ctx->d_zmasks.ptr, ctx->d_zmasks.ptr + (size_t)ctx->zd_row0 * ctx->zstride, nz, ctx->zstride};
        if (e == hipSuccess)
            e = gangfit::launch_fit_zoned_fused(inner, algo == GF_ALGO_AZ_AWARE_TIGHTLY_PACK, make_table(ctx, ctx->d_snap.ptr), make_sparse(ctx), zt,
                                                ctx->d_sched.ptr, ctx->d_feas_zexec.ptr, half, n_apps, d_apps, nullptr, nullptr,
                                                ctx->d_feas_scratch.ptr, half, st, d_feas, ctx->d_feasible_sync.ptr, ctx->eff_nonneg);
    } else {
        e = gangfit::launch_fit_independent(algo, make_table(ctx, ctx->d_snap.ptr), make_sparse(ctx), n_apps, d_apps, nullptr,
                                            ctx->d_feas_exec.ptr, ctx->d_feas_scratch.ptr, half, nullptr, st, d_feas,
                                            ctx->d_feasible_sync.ptr);
    }
    if (e == hipSuccess && !mapped)
        e = hipMemcpyAsync(ctx->h_feasible.ptr, staged.ptr, n_apps, hipMemcpyDeviceToHost, st);
    const auto t_launched = clk::now();
    bool arrived = false;
    if (e == hipSuccess && announce) {
        const volatile uint8_t* f = ctx->h_feasible.ptr;
        uint32_t next = 0;  // everything before it has arrived
        for (uint32_t spin = 0;; ++spin) {
            while (next < n_apps && f[next] != kNotYet) ++next;
            if (next == n_apps) {
                arrived = true;
                break;
            }
            __builtin_ia32_pause();
            // a kernel that faulted never writes: after 2 ms the stream is asked (it reports the error, or completes)
            if ((spin & 255u) == 255u && clk::now() - t_launched > std::chrono::milliseconds(2)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (e == hipSuccess && !arrived) e = gf_wait_stream(st);
    if (e != hipSuccess) ctx->feasible_sync_dirty = true;  // the collection may have stopped anywhere
    const auto t_done = clk::now();
    staged.release();
    if (e != hipSuccess) return fail(ctx, GF_ERR_HIP, "gf_fit_feasible failed: %s", hipGetErrorString(e));
    std::memcpy(has_capacity, ctx->h_feasible.ptr, n_apps);
    const auto t_out = clk::now();
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    ctx->call_phase_us[0] = us(t_entry, t_staged);  // gf_call_phases: stage | launch | wait | copy-out | total
    ctx->call_phase_us[1] = us(t_staged, t_launched);
    ctx->call_phase_us[2] = us(t_launched, t_done);
    ctx->call_phase_us[3] = us(t_done, t_out);
    ctx->call_phase_us[4] = us(t_entry, t_out);
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


int gf_spark_binpack(gf_ctx* ctx, gf_algo algo, const gf_app* app, gf_result* result, uint32_t* exec_nodes,
                     uint64_t exec_nodes_cap) {
    return gf_fit_batch(ctx, GF_MODE_INDEPENDENT, algo, 1, app, result, exec_nodes, exec_nodes_cap, nullptr);
}

int gf_executor_fit(gf_ctx* ctx, int minimal_fragmentation, uint32_t n_req, const int64_t* exe, const int64_t* reserved,
                    const uint32_t* hosts_app, uint32_t* node_out) {
    return gf_executor_fit_zoned(ctx, minimal_fragmentation, n_req, exe, reserved, hosts_app, nullptr, nullptr, node_out);
}

int gf_executor_fit_zoned(gf_ctx* ctx, int minimal_fragmentation, uint32_t n_req, const int64_t* exe, const int64_t* reserved,
                          const uint32_t* hosts_app, const uint32_t* node_zone, const uint32_t* req_zone, uint32_t* node_out) {
    GF_DELEGATE(ctx, gf_executor_fit_zoned(ctx, minimal_fragmentation, n_req, exe, reserved, hosts_app, node_zone, req_zone, node_out));
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    GF_VIEW_ENTER(ctx)
    if (n_req == 0) return GF_OK;
    if (!exe || !node_out) return fail(ctx, GF_ERR_INVALID, "exe/node_out must not be NULL");
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede gf_executor_fit");
    if ((node_zone == nullptr) != (req_zone == nullptr))
        return fail(ctx, GF_ERR_INVALID, "node_zone and req_zone come together (both NULL: no zone step)");
    for (size_t i = 0; i < 3 * (size_t)n_req; ++i)
        if (exe[i] < 0 || exe[i] >= GF_MAX_ABS_QUANTITY) return fail(ctx, GF_ERR_INVALID, "executor request outside [0, 2^62)");
    const uint32_t n = ctx->n_nodes;
    if (node_zone)
        for (uint32_t i = 0; i < n; ++i)
            if (node_zone[i] == GF_ANY_ZONE) return fail(ctx, GF_ERR_INVALID, "node_zone[%u] is GF_ANY_ZONE: a node has a zone", i);
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
    if (node_zone) {
        GF_HIP(ctx, ctx->d_xnzone.reserve((size_t)n + 1));
        GF_HIP(ctx, ctx->d_xqzone.reserve(n_req));
        if (n) GF_HIP(ctx, hipMemcpyAsync(ctx->d_xnzone.ptr, node_zone, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        GF_HIP(ctx, hipMemcpyAsync(ctx->d_xqzone.ptr, req_zone, (size_t)n_req * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
    GF_HIP(ctx, gangfit::launch_executor_fit(minimal_fragmentation != 0, make_table(ctx, ctx->d_snap.ptr),
                                             reserved ? ctx->d_xreserved.ptr : nullptr, n_req, ctx->d_xexe.ptr,
                                             with_hosts ? ctx->d_xhosts.ptr : nullptr, words, node_zone ? ctx->d_xnzone.ptr : nullptr,
                                             node_zone ? ctx->d_xqzone.ptr : nullptr, ctx->d_xout.ptr, st));
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

}  // extern "C"
