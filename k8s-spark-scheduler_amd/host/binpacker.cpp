#include "binpacker.hpp"

namespace gangfit::host {

bool flatten(const NodeGroupSchedulingMetadata& metadata, const std::vector<std::string>& driverOrder,
             const std::vector<std::string>& executorOrder, FlatSnapshot* out, std::string* err) {
    FlatSnapshot s;
    const size_t n = metadata.size();
    s.names.reserve(n);
    for (int j = 0; j < 3; ++j) {
        s.avail[j].reserve(n);
        s.sched[j].reserve(n);
    }
    std::map<std::string, uint32_t> zone_ids;
    for (const auto& [name, m] : metadata) {
        int64_t a[3], sc[3] = {0, 0, 0};
        if (!m.AvailableResources.canonical(a)) {
            if (err) *err = "available resources of node " + name + " are not exactly representable (cpu milli / bytes / gpus)";
            return false;
        }
        if (!m.SchedulableResources.canonical(sc) || sc[0] < 0 || sc[1] < 0 || sc[2] < 0) {
            s.sched_ok = false;
            sc[0] = sc[1] = sc[2] = 0;
        }
        s.index[name] = (uint32_t)s.names.size();
        s.names.push_back(name);
        for (int j = 0; j < 3; ++j) {
            s.avail[j].push_back(a[j]);
            s.sched[j].push_back(sc[j]);
        }
        auto z = zone_ids.emplace(m.ZoneLabel, (uint32_t)zone_ids.size());
        s.zone.push_back(z.first->second);
    }
    // a name that is not a metadata key keeps a distinct out-of-range index: it never hosts anything
    // (LIB/binpack/binpack.go:68, pack_tightly.go:51, distribute_evenly.go:59)
    uint32_t unknown = (uint32_t)n;
    auto to_index = [&](const std::vector<std::string>& order, std::vector<uint32_t>& dst) {
        dst.reserve(order.size());
        for (const std::string& name : order) {
            auto it = s.index.find(name);
            dst.push_back(it == s.index.end() ? unknown++ : it->second);
        }
    };
    to_index(driverOrder, s.driver_order);
    to_index(executorOrder, s.exec_order);
    *out = std::move(s);
    return true;
}

bool upload(gf_ctx* ctx, const FlatSnapshot& s, std::string* err) {
    auto bad = [&](const char* what) {
        if (err) *err = std::string(what) + ": " + gf_last_error(ctx);
        return false;
    };
    const uint32_t n = (uint32_t)s.names.size();
    if (gf_snapshot_set(ctx, n, s.avail[0].data(), s.avail[1].data(), s.avail[2].data(),
                        s.sched_ok ? s.sched[0].data() : nullptr, s.sched_ok ? s.sched[1].data() : nullptr,
                        s.sched_ok ? s.sched[2].data() : nullptr) != GF_OK)
        return bad("gf_snapshot_set");
    if (gf_zones_set(ctx, s.zone.data()) != GF_OK) return bad("gf_zones_set");
    if (gf_orders_set(ctx, s.driver_order.data(), (uint32_t)s.driver_order.size(), s.exec_order.data(),
                      (uint32_t)s.exec_order.size()) != GF_OK)
        return bad("gf_orders_set");
    return true;
}

PackingResult Binpacker::BinpackFunc(const Resources& driverResources, const Resources& executorResources,
                                     int executorCount, const std::vector<std::string>& driverNodePriorityOrder,
                                     const std::vector<std::string>& executorNodePriorityOrder,
                                     const NodeGroupSchedulingMetadata& metadata) const {
    PackingResult r;  // EmptyPackingResult (binpack.go:33-40)
    auto not_served = [&](const std::string& why) {
        r.served = false;
        r.error = why;
        return r;
    };
    gf_app app{};
    if (!driverResources.canonical(app.drv) || !executorResources.canonical(app.exe))
        return not_served("application resources are not exactly representable");
    for (int j = 0; j < 3; ++j)
        if (app.drv[j] < 0 || app.exe[j] < 0) return not_served("negative application resources");
    if (executorCount < 0 || executorCount > GF_MAX_K) return not_served("executor count out of range");
    app.k = executorCount;
    FlatSnapshot snap;
    std::string err;
    if (!flatten(metadata, driverNodePriorityOrder, executorNodePriorityOrder, &snap, &err)) return not_served(err);
    CtxSequence seq(ctx);  // snapshot + decision (+ efficiencies) are one sequence on the shared context
    if (!upload(ctx, snap, &err)) return not_served(err);
    gf_result res{};
    std::vector<uint32_t> exec((size_t)executorCount + 1);
    if (gf_spark_binpack(ctx, Algo, &app, &res, exec.data(), (uint64_t)executorCount) != GF_OK)
        return not_served(std::string("gf_spark_binpack: ") + gf_last_error(ctx));
    r.HasCapacity = res.has_capacity != 0;
    if (!r.HasCapacity) return r;
    r.DriverNode = snap.names[res.driver_node];
    r.ExecutorNodes.reserve(res.exec_len);
    for (uint32_t i = 0; i < res.exec_len; ++i) r.ExecutorNodes.push_back(snap.names[exec[i]]);
    if (with_efficiencies && snap.sched_ok) {  // ComputePackingEfficiencies (binpack.go:77), one entry per metadata key
        std::vector<double> eff(3 * snap.names.size());
        app.exec_off = 0;
        if (gf_packing_efficiencies(ctx, Algo, &app, &res, exec.data(), eff.data()) == GF_OK)
            for (size_t n = 0; n < snap.names.size(); ++n)
                r.PackingEfficiencies[snap.names[n]] = {snap.names[n], eff[3 * n], eff[3 * n + 1], eff[3 * n + 2]};
    }
    return r;
}

Binpacker SelectBinpacker(const std::string& name, gf_ctx* ctx) {
    struct E {
        const char* name;
        gf_algo algo;
        bool single_az;
    };
    static const E table[] = {{"tightly-pack", GF_ALGO_TIGHTLY_PACK, false},
                              {"distribute-evenly", GF_ALGO_DISTRIBUTE_EVENLY, false},
                              {"az-aware-tightly-pack", GF_ALGO_AZ_AWARE_TIGHTLY_PACK, false},
                              {"single-az-tightly-pack", GF_ALGO_SINGLE_AZ_TIGHTLY_PACK, true},
                              {"single-az-minimal-fragmentation", GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION, true}};
    for (const E& e : table)
        if (name == e.name) return {e.name, e.algo, e.single_az, ctx};
    return {"distribute-evenly", GF_ALGO_DISTRIBUTE_EVENLY, false, ctx};
}

}  // namespace gangfit::host
