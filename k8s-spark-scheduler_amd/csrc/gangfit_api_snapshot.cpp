// gangfit_api_snapshot.cpp — the snapshot side of the C ABI: gf_snapshot_set / gf_zones_set / gf_orders_set (host-built slot tables), the resident
// cluster columns and usage sums, gf_snapshot_build* (reservation replay + metadata + priority sort + slot tables on the device).
#include "gangfit_ctx.h"

using namespace gfapi;

namespace gfapi {

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

}  // namespace gfapi

extern "C" {

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
    ctx->eff_nonneg = ctx->have_sched;  // (fit_zoned_fused_kernel's feasibility instantiation: when no efficiency can be negative)
    for (int j = 0; j < 3 && ctx->eff_nonneg; ++j)
        for (uint32_t n = 0; n < n_nodes; ++n)
            if (av[j][n] > sc[j][n]) {
                ctx->eff_nonneg = false;
                break;
            }
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
    ctx->g_prefix.clear();
    if (mergeable && ctx->sparse_gpu) {
        uint32_t n_g = 0;
        for (uint32_t s2 = 0; s2 < merged.size(); ++s2)
            if ((mflags[s2] & 1) && tgpu[s2] > 0) ++n_g;
        if (n_g > 0 && (uint64_t)n_g * 4 <= merged.size()) {
            const uint32_t n_gpad = (n_g + 63u) / 64u * 64u, gch = n_gpad / 64u;
            GF_HIP(ctx, ctx->h_gtab.reserve(3 * (size_t)n_gpad + 3 * (size_t)gch));
            GF_HIP(ctx, ctx->h_gidx.reserve(2 * (size_t)n_gpad + n_slots));
            int64_t* g0 = ctx->h_gtab.ptr;
            int64_t* gmax = g0 + 3 * (size_t)n_gpad;
            uint32_t* gnode = ctx->h_gidx.ptr;
            uint32_t* gsub = gnode + n_gpad;
            uint32_t* gslot = gsub + n_slots;  // sub-slot -> slot (SparseTable::slot_of_sub; the padding names the sentinel slot)
            for (uint32_t i = 0; i < 3 * n_gpad; ++i) g0[i] = kSentinelAvail;
            for (uint32_t i = 0; i < n_gpad; ++i) gnode[i] = GF_NO_NODE;
            for (uint32_t i = 0; i < n_gpad; ++i) gslot[i] = n_slots - 1u;
            for (uint32_t s2 = 0; s2 < n_slots; ++s2) gsub[s2] = GF_NO_NODE;
            uint32_t k = 0;
            ctx->g_prefix.assign((size_t)n_slots / 64u + 2u, n_g);
            for (uint32_t s2 = 0; s2 < merged.size(); ++s2) {
                if ((s2 & 63u) == 0u) ctx->g_prefix[s2 >> 6] = k;
                if ((mflags[s2] & 1) && tgpu[s2] > 0) {
                    g0[k] = tcpu[s2];
                    g0[n_gpad + k] = tmem[s2];
                    g0[2 * (size_t)n_gpad + k] = tgpu[s2];
                    gnode[k] = slot_node[s2];
                    gslot[k] = s2;
                    gsub[s2] = k++;
                }
            }
            for (int j = 0; j < 3; ++j)
                for (uint32_t c = 0; c < gch; ++c) {
                    int64_t m = INT64_MIN;
                    for (uint32_t i = c * 64; i < (c + 1) * 64; ++i) m = g0[(size_t)j * n_gpad + i] > m ? g0[(size_t)j * n_gpad + i] : m;
                    gmax[(size_t)j * gch + c] = m;
                }
            GF_HIP(ctx, ctx->d_gtab.reserve(3 * (size_t)n_gpad));
            GF_HIP(ctx, ctx->d_gcmax.reserve(3 * (size_t)gch));
            GF_HIP(ctx, ctx->d_gidx.reserve(2 * (size_t)n_gpad + n_slots));
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gtab.ptr, g0, 3 * (size_t)n_gpad * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gcmax.ptr, gmax, 3 * (size_t)gch * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gidx.ptr, gnode, (2 * (size_t)n_gpad + n_slots) * sizeof(uint32_t), hipMemcpyHostToDevice,
                                       ctx->stream));
            // (the candidate words of the view — all sub-slots, then one row per zone of the evaluation list — follow the zone views below)
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
        if (ctx->n_g != 0) {  // SparseTable::xmask (row 0: every sub-slot) and ::zmask (row 1 + zi: the sub-slots of zone eval[zi])
            const uint32_t gch = ctx->n_gpad / 64u;
            const uint32_t* gnode = ctx->h_gidx.ptr;
            std::vector<uint64_t> gm((size_t)gch * (1u + nz), 0);
            for (uint32_t i = 0; i < ctx->n_g; ++i) {
                gm[i >> 6] |= 1ull << (i & 63);
                const uint32_t z = zone_of(gnode[i]);
                for (uint32_t zi = 0; zi < nz; ++zi)
                    if (eval[zi] == z) gm[(size_t)gch * (1u + zi) + (i >> 6)] |= 1ull << (i & 63);
            }
            GF_HIP(ctx, ctx->d_gmask.reserve(gm.size()));
            GF_HIP(ctx, hipMemcpyAsync(ctx->d_gmask.ptr, gm.data(), gm.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
            GF_HIP(ctx, gf_wait_stream(ctx->stream));  // gm is a local
        }
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
    b.d_zfirst = d_zfirst;  // the finalize step's accumulators start clean with everything else (one clearing launch)
    b.d_zhasx = d_zhasx;
    b.zhasx_to_scalars_words = 2 * Z + 16;  // d_zhasx | d_zeval | d_scalars
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
        GF_HIP(ctx, ctx->h_bcols.reserve(6 * N + 8));
        GF_HIP(ctx, ctx->h_border.reserve(N + 16));
        // everything the host needs back is one range of sixteen words (SnapshotFinalize::d_scalars): the kernels write it into
        // pinned memory as they produce it (no copy on the stream), or ONE copy where that memory is not mapped to the device
        uint32_t* h_scalars = ctx->h_border.ptr;
        if (ctx->h_border.dev != nullptr) {
            std::memset(h_scalars, 0, 16 * sizeof(uint32_t));
            f.h_out = ctx->h_border.dev;
        }
        GF_HIP(ctx, gangfit::launch_snapshot_finalize(f, ctx->d_sortwork.ptr + gangfit::snapshot_sort_error_word(), st));
        if (f.h_out == nullptr) GF_HIP(ctx, hipMemcpyAsync(h_scalars, d_scalars, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        GF_HIP(ctx, gf_wait_stream(st));
        if (h_scalars[3] != 0) return fail(ctx, GF_ERR_HIP, "the priority sort's grid barrier gave up (device oversubscribed?)");
        const uint32_t nz = h_scalars[0];
        for (int j = 0; j < 3; ++j) {  // 3 units, then the 3 largest scaled magnitudes, as pairs of words
            ctx->unit[j] = (int64_t)((uint64_t)h_scalars[4 + 2 * j] | ((uint64_t)h_scalars[5 + 2 * j] << 32));
            ctx->nmax[j] = (int64_t)((uint64_t)h_scalars[10 + 2 * j] | ((uint64_t)h_scalars[11 + 2 * j] << 32));
        }
        ctx->narrow_ok = h_scalars[1] == 0;
        ctx->have_sched = h_scalars[2] == 0;  // a negative schedulable value (overhead above allocatable) disables the efficiencies
        ctx->n_nodes = n;
        ctx->have_snapshot = true;
        ctx->zone.clear();
        if (zone_of_node) ctx->zone.assign(zone_of_node, zone_of_node + N);
        ctx->n_x = ctx->n_d = n;
        ctx->n_g = ctx->n_gpad = 0;  // the sparse gpu view is built by gf_orders_set only; the full order serves here
        ctx->g_prefix.clear();
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
        ctx->eff_nonneg = false;
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

}  // extern "C"
