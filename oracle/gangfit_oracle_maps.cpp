// gangfit_oracle_maps.cpp — CPU ORACLE, reference-SHAPED variant (test infrastructure, NOT product code).
//
// The same decisions as oracle/gangfit_oracle.c, written with the reference's data structures so that its COST looks like
// the Go code's: node names are strings, nodesSchedulingMetadata / reserved / PackingEfficiencies / the usage map are hash
// maps keyed by them (Go: map[string]*...), every driver candidate gets a fresh `reserved` map sized for N entries
// (binpack.go:72), every successful pack builds the per-node efficiency map over ALL nodes (binpack.go:77 ->
// efficiency.go:66-103) — which fitEarlierDrivers then throws away for each replayed driver.  Quantities stay canonical
// int64 triples (a resource.Quantity compare / add is costlier than that: still conservative).
// Used by bench.py's cpu_baseline legs next to the dense-array port, and checked against it by tests/test_oracle.py.
// tightly-pack and distribute-evenly only (the packers of BASELINE.json's north star).
//
// Paths are relative to /root/reference; LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg.
#include <cstdint>
#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

#include "gangfit_oracle.h"

namespace {

struct Res {
    int64_t v[3] = {0, 0, 0};
    bool GreaterThan(const Res& o) const { return v[0] > o.v[0] || v[1] > o.v[1] || v[2] > o.v[2]; }  // resources.go:239-241
    void Add(const Res& o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; }
    void Sub(const Res& o) { v[0] -= o.v[0]; v[1] -= o.v[1]; v[2] -= o.v[2]; }
};
struct Meta {  // resources.NodeSchedulingMetadata
    Res available, schedulable;
};
struct Eff {   // binpack.PackingEfficiency
    double cpu, mem, gpu;
};
using Metadata = std::unordered_map<std::string, Meta>;
using NodeGroupResources = std::unordered_map<std::string, Res>;

// tightlyPackExecutors, LIB/binpack/pack_tightly.go:34-63
bool tightlyPackExecutors(const Res& exe, int k, const std::vector<std::string>& order, const Metadata& meta,
                          NodeGroupResources& reserved, std::vector<std::string>& nodes) {
    nodes.clear();
    if (k == 0) return true;
    for (const std::string& n : order) {
        auto r = reserved.find(n);
        if (r == reserved.end()) r = reserved.emplace(n, Res{}).first;
        for (;;) {
            r->second.Add(exe);
            auto m = meta.find(n);
            if (m == meta.end() || r->second.GreaterThan(m->second.available)) {
                r->second.Sub(exe);
                break;
            }
            nodes.push_back(n);
            if ((int)nodes.size() == k) return true;
        }
    }
    return false;
}

// distributeExecutorsEvenly, LIB/binpack/distribute_evenly.go:34-73
bool distributeExecutorsEvenly(const Res& exe, int k, const std::vector<std::string>& order, const Metadata& meta,
                               NodeGroupResources& reserved, std::vector<std::string>& nodes) {
    nodes.clear();
    std::unordered_map<std::string, bool> availableNodes;
    availableNodes.reserve(order.size());
    for (const std::string& n : order) availableNodes[n] = true;
    if (k == 0) return true;
    while (!availableNodes.empty()) {
        for (const std::string& n : order) {
            if (availableNodes.find(n) == availableNodes.end()) continue;
            auto r = reserved.find(n);
            if (r == reserved.end()) r = reserved.emplace(n, Res{}).first;
            r->second.Add(exe);
            auto m = meta.find(n);
            if (m == meta.end() || r->second.GreaterThan(m->second.available)) {
                availableNodes.erase(n);
                r->second.Sub(exe);
            } else {
                nodes.push_back(n);
                if ((int)nodes.size() == k) return true;
            }
        }
    }
    return false;
}

int64_t value_of_milli(int64_t milli) {  // Quantity.Value(): ceil to whole units, away from zero (quantity.go:732-734)
    return milli >= 0 ? (milli + 999) / 1000 : -((-milli + 999) / 1000);
}

// ComputePackingEfficiencies, LIB/binpack/efficiency.go:66-103: one entry per metadata key
void computePackingEfficiencies(const Metadata& meta, const NodeGroupResources& reserved,
                                std::unordered_map<std::string, Eff>& out) {
    out.clear();
    out.reserve(meta.size());
    for (const auto& [name, m] : meta) {
        Res r;
        if (auto it = reserved.find(name); it != reserved.end()) r = it->second;
        const int64_t s0 = value_of_milli(m.schedulable.v[0]);
        Eff e;
        e.cpu = (double)value_of_milli(m.schedulable.v[0] - m.available.v[0] + r.v[0]) / (double)(s0 == 0 ? 1 : s0);
        e.mem = (double)(m.schedulable.v[1] - m.available.v[1] + r.v[1]) / (double)(m.schedulable.v[1] == 0 ? 1 : m.schedulable.v[1]);
        e.gpu = m.schedulable.v[2] == 0 ? 0.0 : (double)(m.schedulable.v[2] - m.available.v[2] + r.v[2]) / (double)m.schedulable.v[2];
        out.emplace(name, e);
    }
}

struct PackingResult {
    std::string DriverNode;
    std::vector<std::string> ExecutorNodes;
    std::unordered_map<std::string, Eff> PackingEfficiencies;
    bool HasCapacity = false;
};

// SparkBinPack, LIB/binpack/binpack.go:60-87
void SparkBinPack(int algo, const Res& drv, const Res& exe, int k, const std::vector<std::string>& driverOrder,
                  const std::vector<std::string>& executorOrder, const Metadata& meta, PackingResult& out) {
    out.HasCapacity = false;
    out.DriverNode.clear();
    out.ExecutorNodes.clear();
    for (const std::string& d : driverOrder) {
        auto m = meta.find(d);
        if (m == meta.end() || drv.GreaterThan(m->second.available)) continue;   // :68-71
        NodeGroupResources reserved;
        reserved.reserve(meta.size());                                            // make(NodeGroupResources, len(metadata)) :72
        reserved[d] = drv;                                                        // :73
        const bool ok = algo == GO_ALGO_TIGHTLY_PACK ? tightlyPackExecutors(exe, k, executorOrder, meta, reserved, out.ExecutorNodes)
                                                     : distributeExecutorsEvenly(exe, k, executorOrder, meta, reserved, out.ExecutorNodes);
        if (ok) {
            out.DriverNode = d;
            out.HasCapacity = true;
            computePackingEfficiencies(meta, reserved, out.PackingEfficiencies);  // :77
            return;
        }
    }
    out.ExecutorNodes.clear();
}

std::string node_name(uint32_t i) {
    char buf[24];
    std::snprintf(buf, sizeof buf, "node-%07u", i);
    return buf;
}

}  // namespace

extern "C" {

// fitEarlierDrivers + the final pack (resource.go:224-262, 309-328) — or, with chain == 0, n_apps independent packs
// against the same snapshot — on string-keyed maps.  Same results as go_fit_fifo_chain / go_fit_independent; avail is
// updated like availableNodesSchedulingMetadata when chain != 0.  Returns failed_at (chain) or -1.
int32_t go_fit_maps(int algo, int chain, int64_t* avail, const int64_t* sched, uint32_t n_nodes, const go_app* apps,
                    uint32_t n_apps, const uint32_t* driver_order, uint32_t n_d, const uint32_t* exec_order, uint32_t n_x,
                    go_result* results, const uint64_t* exec_off, uint32_t* exec_out) {
    Metadata meta;
    meta.reserve(n_nodes);
    std::vector<std::string> names(n_nodes);
    std::unordered_map<std::string, uint32_t> index;
    index.reserve(n_nodes);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        names[i] = node_name(i);
        index[names[i]] = i;
        Meta m;
        for (int j = 0; j < 3; ++j) {
            m.available.v[j] = avail[3 * (size_t)i + j];
            m.schedulable.v[j] = sched ? sched[3 * (size_t)i + j] : INT64_MAX >> 2;
        }
        meta.emplace(names[i], m);
    }
    auto to_names = [&](const uint32_t* order, uint32_t n) {
        std::vector<std::string> out;
        out.reserve(n);
        for (uint32_t i = 0; i < n; ++i) out.push_back(order[i] < n_nodes ? names[order[i]] : "ghost-" + std::to_string(order[i]));
        return out;
    };
    const std::vector<std::string> D = to_names(driver_order, n_d), X = to_names(exec_order, n_x);
    int32_t failed_at = -1;
    PackingResult r;
    for (uint32_t a = 0; a < n_apps; ++a) {
        results[a].has_capacity = 0;
        results[a].driver_node = GO_NO_NODE;
        results[a].exec_len = 0;
        results[a].evaluated = 0;
    }
    for (uint32_t a = 0; a < n_apps; ++a) {
        Res drv, exe;
        for (int j = 0; j < 3; ++j) {
            drv.v[j] = apps[a].drv[j];
            exe.v[j] = apps[a].exe[j];
        }
        SparkBinPack(algo, drv, exe, apps[a].k, D, X, meta, r);
        results[a].evaluated = 1;
        results[a].has_capacity = r.HasCapacity ? 1 : 0;
        if (r.HasCapacity) {
            results[a].driver_node = index[r.DriverNode];
            results[a].exec_len = (uint32_t)r.ExecutorNodes.size();
            for (size_t i = 0; i < r.ExecutorNodes.size(); ++i) exec_out[exec_off[a] + i] = index[r.ExecutorNodes[i]];
        }
        if (!chain || a + 1 == n_apps) continue;
        if (!r.HasCapacity) {
            if (apps[a].flags & GO_APP_SKIPPABLE) continue;   // resource.go:244-248
            failed_at = (int32_t)a;                           // :249-251
            break;
        }
        // sparkResourceUsage (sparkpods.go:139-146) + SubtractUsageIfExists (resources.go:129-135)
        NodeGroupResources usage;
        usage[r.DriverNode] = drv;
        for (const std::string& n : r.ExecutorNodes) usage[n] = exe;
        for (const auto& [n, u] : usage)
            if (auto m = meta.find(n); m != meta.end()) m->second.available.Sub(u);
    }
    if (chain)
        for (uint32_t i = 0; i < n_nodes; ++i)
            for (int j = 0; j < 3; ++j) avail[3 * (size_t)i + j] = meta[names[i]].available.v[j];
    return failed_at;
}

}  // extern "C"
