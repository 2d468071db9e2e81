#include "failover.hpp"

#include <map>

#include "binpacker.hpp"

namespace gangfit::host {

namespace {

// executorResources x count, exactly (the reference reaches it by repeated Add)
Resources times(const int64_t exe[3], uint32_t count) {
    Resources r;
    r.CPU = Quantity::FromNano((i128)exe[0] * 1000000 * count);
    r.Memory = Quantity::FromNano((i128)exe[1] * 1000000000 * count);
    r.NvidiaGPU = Quantity::FromNano((i128)exe[2] * 1000000000 * count);
    return r;
}

std::vector<FindNodesResult> run(gf_ctx* ctx, bool chained, const std::vector<FindNodesRequest>& requests,
                                 const NodeGroupResources& available, const std::vector<Node>& orderedNodes) {
    std::vector<FindNodesResult> out(requests.size());
    auto not_served = [&](const std::string& why) {
        for (FindNodesResult& r : out) {
            r.served = false;
            r.error = why;
        }
        return out;
    };
    if (requests.empty()) return out;
    // flatten: node index = position in orderedNodes (each node once, :297-314); availableResources[n.Name] exists for every
    // ordered node (both derive from schedulableNodes)
    const uint32_t n = (uint32_t)orderedNodes.size();
    std::vector<int64_t> cols[3];
    std::vector<uint32_t> order(n);
    for (int j = 0; j < 3; ++j) cols[j].resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        auto it = available.find(orderedNodes[i].Name);
        if (it == available.end()) return not_served("node " + orderedNodes[i].Name + " has no availableResources entry");
        int64_t v[3];
        if (!it->second.canonical(v)) return not_served("available resources of node " + orderedNodes[i].Name + " are not exactly representable");
        for (int j = 0; j < 3; ++j) cols[j][i] = v[j];
        order[i] = i;
    }
    std::vector<int64_t> exe(3 * requests.size());
    std::vector<int32_t> k(requests.size());
    uint64_t total = 0;
    for (size_t q = 0; q < requests.size(); ++q) {
        if (!requests[q].executorResources.canonical(&exe[3 * q]) || exe[3 * q] < 0 || exe[3 * q + 1] < 0 || exe[3 * q + 2] < 0)
            return not_served("executor resources are not exactly representable");
        if (requests[q].executorCount <= 0 || requests[q].executorCount > GF_MAX_K) return not_served("executor count out of range");
        k[q] = requests[q].executorCount;
        total += (uint64_t)k[q];
    }
    CtxSequence seq(ctx);
    if (gf_snapshot_set(ctx, n, cols[0].data(), cols[1].data(), cols[2].data(), nullptr, nullptr, nullptr) != GF_OK ||
        gf_orders_set(ctx, nullptr, 0, order.data(), n) != GF_OK)
        return not_served(std::string("snapshot upload: ") + gf_last_error(ctx));
    std::vector<gf_find_result> res(requests.size());
    std::vector<uint32_t> nodes(total + 1);
    if (gf_find_nodes(ctx, chained ? 1 : 0, (uint32_t)requests.size(), exe.data(), k.data(), res.data(), nodes.data(), total,
                      nullptr) != GF_OK)
        return not_served(std::string("gf_find_nodes: ") + gf_last_error(ctx));
    uint64_t off = 0;
    for (size_t q = 0; q < requests.size(); ++q) {
        FindNodesResult& r = out[q];
        std::map<uint32_t, uint32_t> mult;
        for (uint32_t i = 0; i < res[q].placed; ++i) {
            const uint32_t node = nodes[off + i];
            r.executorNodeNames.push_back(orderedNodes[node].Name);
            ++mult[node];
        }
        // the `reserved` map, rebuilt as include/gangfit.h documents: every node up to last_node was reached and carries
        // one failing add on top of its placements, except last_node itself when the count was reached there
        if (res[q].last_node != GF_NO_NODE)
            for (uint32_t node = 0; node <= res[q].last_node && node < n; ++node) {
                uint32_t adds = (mult.count(node) ? mult[node] : 0u) + 1u;
                if (node == res[q].last_node && res[q].placed == (uint32_t)k[q]) --adds;
                r.reserved[orderedNodes[node].Name] = times(&exe[3 * q], adds);
            }
        off += (uint64_t)k[q];
    }
    return out;
}

}  // namespace

FindNodesResult findNodes(gf_ctx* ctx, int executorCount, const Resources& executorResources,
                          const NodeGroupResources& availableResources, const std::vector<Node>& orderedNodes) {
    return run(ctx, false, {{executorCount, executorResources}}, availableResources, orderedNodes)[0];
}

std::vector<FindNodesResult> findNodesForStaleApplications(gf_ctx* ctx, const std::vector<FindNodesRequest>& requests,
                                                           NodeGroupResources* availableResources,
                                                           const std::vector<Node>& orderedNodes) {
    std::vector<FindNodesResult> out = run(ctx, true, requests, *availableResources, orderedNodes);
    for (const FindNodesResult& r : out) {
        if (!r.served) break;
        for (const auto& [node, res] : r.reserved) (*availableResources)[node].Sub(res);  // NodeGroupResources.Sub (:159)
    }
    return out;
}

}  // namespace gangfit::host
