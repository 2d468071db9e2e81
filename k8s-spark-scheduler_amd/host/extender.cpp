#include "extender.hpp"

namespace gangfit::host {

std::string executorReservationName(int i) { return "executor-" + std::to_string(i + 1); }

ResourceReservation newResourceReservation(const std::string& driverNode, const std::vector<std::string>& executorNodes,
                                           const Pod& driver, const Resources& d, const Resources& e) {
    ResourceReservation rr;
    auto list = [](const Resources& r) {
        return ResourceList{{kResourceCPU, r.CPU}, {kResourceMemory, r.Memory}, {kResourceNvidiaGPU, r.NvidiaGPU}};
    };
    rr.Reservations["driver"] = {driverNode, list(d)};
    for (size_t i = 0; i < executorNodes.size(); ++i)
        rr.Reservations[executorReservationName((int)i)] = {executorNodes[i], list(e)};
    auto app = driver.labels.find(common::SparkAppIDLabel);
    rr.Name = app == driver.labels.end() ? "" : app->second;
    rr.AppIDLabel = rr.Name;
    rr.Namespace = driver.Namespace;
    rr.OwnerPodName = driver.Name;
    rr.Pods["driver"] = driver.Name;
    return rr;
}

bool SparkSchedulerExtender::shouldSkipDriverFifo(const Pod& pod, const std::string& instanceGroup) const {
    int64_t age = fifo_.DefaultEnforceAfterPodAgeNanos;
    if (auto it = fifo_.EnforceAfterPodAgeByInstanceGroup.find(instanceGroup); it != fifo_.EnforceAfterPodAgeByInstanceGroup.end())
        age = it->second;
    return pod.CreationTimestampNanos + age > nowNanos;
}

SelectNodeResult SparkSchedulerExtender::selectDriverNode(const std::string& instanceGroup, const Pod& driver,
                                                          const std::vector<std::string>& nodeNames,
                                                          const std::vector<Node>& availableNodes) {
    SelectNodeResult out;
    auto app_label = driver.labels.find(common::SparkAppIDLabel);
    const std::string app_id = app_label == driver.labels.end() ? "" : app_label->second;
    // an application that already holds a reservation keeps its driver node (resource.go:278-291)
    for (const ResourceReservation& rr : reservations)
        if (rr.Name == app_id && rr.Namespace == driver.Namespace) {
            auto d = rr.Reservations.find("driver");
            out.node = d == rr.Reservations.end() ? "" : d->second.Node;
            out.outcome = outcome::success;
            return out;
        }
    // snapshot (resource.go:300-303)
    NodeGroupResources usage = UsageForNodes(reservations);
    for (const auto& [n, r] : softReservationUsage) usage[n].Add(r);
    NodeGroupSchedulingMetadata metadata = NodeSchedulingMetadataForNodes(availableNodes, usage, overhead);
    auto [driverNodeNames, executorNodeNames] = sorter_.PotentialNodes(metadata, nodeNames);
    std::string err;
    auto resources = sparkResources(driver, &err);
    if (!resources) {
        out.outcome = outcome::failureInternal;
        out.error = "failed to get spark resources: " + err;
        return out;
    }
    // FIFO replay + final pack as one chain (resource.go:309-328)
    std::vector<gf_app> apps;
    if (isFIFO_) {
        for (const Pod* p : filterToEarliestAndSort(driver, pods)) {
            auto r = sparkResources(*p, nullptr);
            if (!r) continue;  // "failed to get driver resources, skipping driver" (resource.go:231-237)
            gf_app a{};
            if (!r->DriverResources.canonical(a.drv) || !r->ExecutorResources.canonical(a.exe) || r->MinExecutorCount < 0 ||
                r->MinExecutorCount > GF_MAX_K) {
                out.served = false;
                out.error = "earlier driver " + p->Name + " is not exactly representable";
                return out;
            }
            a.k = r->MinExecutorCount;
            a.flags = shouldSkipDriverFifo(*p, instanceGroup) ? GF_APP_SKIPPABLE : 0u;
            apps.push_back(a);
        }
    }
    gf_app cur{};
    if (!resources->DriverResources.canonical(cur.drv) || !resources->ExecutorResources.canonical(cur.exe) ||
        resources->MinExecutorCount < 0 || resources->MinExecutorCount > GF_MAX_K) {
        out.served = false;
        out.error = "application resources are not exactly representable";
        return out;
    }
    cur.k = resources->MinExecutorCount;
    apps.push_back(cur);
    FlatSnapshot snap;
    if (!flatten(metadata, driverNodeNames, executorNodeNames, &snap, &err) || !upload(binpacker_.ctx, snap, &err)) {
        out.served = false;
        out.error = err;
        return out;
    }
    uint64_t total_k = 0;
    for (const gf_app& a : apps) total_k += (uint64_t)a.k;
    std::vector<gf_result> results(apps.size());
    std::vector<uint32_t> exec(total_k + 1);
    int32_t failed_at = -1;
    if (gf_fit_batch(binpacker_.ctx, GF_MODE_FIFO_CHAIN, binpacker_.Algo, (uint32_t)apps.size(), apps.data(), results.data(),
                     exec.data(), total_k, &failed_at) != GF_OK) {
        out.served = false;
        out.error = std::string("gf_fit_batch: ") + gf_last_error(binpacker_.ctx);
        return out;
    }
    if (failed_at >= 0) {  // resource.go:315-318
        out.outcome = outcome::failureEarlierDriver;
        out.error = "earlier drivers do not fit to the cluster";
        return out;
    }
    const gf_result& last = results.back();
    if (!last.has_capacity) {  // resource.go:346-349
        out.outcome = outcome::failureFit;
        out.error = "application does not fit to the cluster";
        return out;
    }
    std::vector<std::string> executorNodes;
    const uint64_t off = total_k - (uint64_t)cur.k;
    for (uint32_t i = 0; i < last.exec_len; ++i) executorNodes.push_back(snap.names[exec[off + i]]);
    out.node = snap.names[last.driver_node];
    out.outcome = outcome::success;
    out.created = newResourceReservation(out.node, executorNodes, driver, resources->DriverResources,
                                         resources->ExecutorResources);
    return out;
}

SelectNodeResult SparkSchedulerExtender::rescheduleExecutor(const Pod& driver, const std::vector<std::string>& nodeNames,
                                                            const std::vector<Node>& availableNodes,
                                                            const std::set<std::string>& nodesHostingApp,
                                                            bool isExtraExecutor) {
    SelectNodeResult out;
    std::string err;
    auto resources = sparkResources(driver, &err);
    if (!resources) {
        out.outcome = outcome::failureInternal;
        out.error = err;
        return out;
    }
    int64_t exe[3];
    if (!resources->ExecutorResources.canonical(exe) || exe[0] < 0 || exe[1] < 0 || exe[2] < 0) {
        out.served = false;
        out.error = "executor resources are not exactly representable";
        return out;
    }
    NodeGroupResources usage = UsageForNodes(reservations);  // GetReservedResources
    for (const auto& [n, r] : softReservationUsage) usage[n].Add(r);
    std::set<std::string> had_usage;
    for (const auto& kv : usage) had_usage.insert(kv.first);
    NodeGroupSchedulingMetadata metadata = NodeSchedulingMetadataForNodes(availableNodes, usage, overhead);
    auto [driverOrder, executorNodeNames] = sorter_.PotentialNodes(metadata, nodeNames);
    (void)driverOrder;
    FlatSnapshot snap;
    if (!flatten(metadata, {}, executorNodeNames, &snap, &err) || !upload(binpacker_.ctx, snap, &err)) {
        out.served = false;
        out.error = err;
        return out;
    }
    const bool minfrag = binpacker_.Name == "single-az-minimal-fragmentation";
    // what this path subtracts on top of the snapshot: `usage.Add(overhead)` (:642) counts the overhead a second time
    // for nodes NodeSchedulingMetadataForNodes already updated in place; GetNodeCapacities (:682) is handed the
    // overhead map as reservedResources
    std::vector<int64_t> reserved(3 * snap.names.size(), 0);
    bool any_reserved = false;
    for (size_t i = 0; i < snap.names.size(); ++i) {
        auto o = overhead.find(snap.names[i]);
        if (o == overhead.end()) continue;
        if (!minfrag && !had_usage.count(snap.names[i])) continue;
        int64_t v[3];
        if (!o->second.canonical(v) || v[0] < 0 || v[1] < 0 || v[2] < 0) {
            out.served = false;
            out.error = "overhead of node " + snap.names[i] + " is not exactly representable";
            return out;
        }
        for (int j = 0; j < 3; ++j) reserved[3 * i + j] = v[j];
        any_reserved = true;
    }
    std::vector<uint32_t> hosts((snap.names.size() + 31) / 32, 0u);
    for (const std::string& n : nodesHostingApp)
        if (auto it = snap.index.find(n); it != snap.index.end()) hosts[it->second >> 5] |= 1u << (it->second & 31);
    uint32_t node = GF_NO_NODE;
    if (gf_executor_fit(binpacker_.ctx, minfrag ? 1 : 0, 1, exe, any_reserved ? reserved.data() : nullptr,
                        minfrag ? hosts.data() : nullptr, &node) != GF_OK) {
        out.served = false;
        out.error = std::string("gf_executor_fit: ") + gf_last_error(binpacker_.ctx);
        return out;
    }
    if (node == GF_NO_NODE) {
        out.outcome = outcome::failureFit;
        out.error = "not enough capacity to reschedule the executor";
        return out;
    }
    out.node = snap.names[node];
    out.outcome = isExtraExecutor ? outcome::successScheduledExtraExecutor : outcome::successRescheduled;
    return out;
}

bool SparkSchedulerExtender::DoesPodExceedClusterCapacity(const Pod& driver, const std::vector<Node>& availableNodes,
                                                          const NodeGroupResources& nonSchedulableOverhead, bool* served,
                                                          std::string* err) {
    if (served) *served = true;
    NodeGroupResources usage;  // empty cluster
    NodeGroupSchedulingMetadata metadata = NodeSchedulingMetadataForNodes(availableNodes, usage, nonSchedulableOverhead);
    std::vector<std::string> names;
    for (const Node& n : availableNodes) names.push_back(n.Name);
    std::string e;
    auto resources = sparkResources(driver, &e);
    if (!resources) {
        if (err) *err = e;
        return false;  // the reference returns (false, err)
    }
    Binpacker bp = binpacker_;
    bp.with_efficiencies = false;
    PackingResult r = bp.BinpackFunc(resources->DriverResources, resources->ExecutorResources, resources->MinExecutorCount,
                                     names, names, metadata);
    if (!r.served) {
        if (served) *served = false;
        if (err) *err = r.error;
        return false;
    }
    return !r.HasCapacity;
}

}  // namespace gangfit::host

namespace gangfit::host {

std::vector<std::pair<std::string, bool>> SparkSchedulerExtender::scanForUnschedulablePods(
    const std::vector<Pod>& allPods, int64_t timeoutNanos, const std::vector<Node>& availableNodes,
    const NodeGroupResources& nonSchedulableOverhead, bool* served, std::string* err) {
    std::vector<std::pair<std::string, bool>> out;
    if (served) *served = true;
    if (timeoutNanos <= 0) timeoutNanos = 600ll * 1000000000;  // 10 minutes (unschedulablepods.go:62-64)
    std::vector<const Pod*> stale;
    std::vector<gf_app> apps;
    for (const Pod& pod : allPods) {
        auto role = pod.labels.find(common::SparkRoleLabel);
        if (pod.SchedulerName != common::SparkSchedulerName || !pod.NodeName.empty() || pod.Deleting ||
            role == pod.labels.end() || role->second != common::Driver || pod.CreationTimestampNanos + timeoutNanos >= nowNanos)
            continue;
        std::string e;
        auto r = sparkResources(pod, &e);
        if (!r) {  // "failed to check if pod was unschedulable": the reference returns from the scan
            if (err) *err = e;
            break;
        }
        gf_app a{};
        if (!r->DriverResources.canonical(a.drv) || !r->ExecutorResources.canonical(a.exe) || r->MinExecutorCount < 0 ||
            r->MinExecutorCount > GF_MAX_K) {
            if (served) *served = false;
            if (err) *err = "pod " + pod.Name + " is not exactly representable";
            return {};
        }
        a.k = r->MinExecutorCount;
        stale.push_back(&pod);
        apps.push_back(a);
    }
    if (apps.empty()) return out;
    NodeGroupResources usage;  // zeroUsage: the cluster as if nothing ran on it
    NodeGroupSchedulingMetadata metadata = NodeSchedulingMetadataForNodes(availableNodes, usage, nonSchedulableOverhead);
    std::vector<std::string> names;
    for (const Node& n : availableNodes) names.push_back(n.Name);
    FlatSnapshot snap;
    std::string e;
    if (!flatten(metadata, names, names, &snap, &e) || !upload(binpacker_.ctx, snap, &e)) {
        if (served) *served = false;
        if (err) *err = e;
        return {};
    }
    uint64_t total_k = 0;
    for (const gf_app& a : apps) total_k += (uint64_t)a.k;
    std::vector<gf_result> results(apps.size());
    std::vector<uint32_t> exec(total_k + 1);
    if (gf_fit_batch(binpacker_.ctx, GF_MODE_INDEPENDENT, binpacker_.Algo, (uint32_t)apps.size(), apps.data(), results.data(),
                     exec.data(), total_k, nullptr) != GF_OK) {
        if (served) *served = false;
        if (err) *err = std::string("gf_fit_batch: ") + gf_last_error(binpacker_.ctx);
        return {};
    }
    for (size_t i = 0; i < stale.size(); ++i) out.emplace_back(stale[i]->Name, results[i].has_capacity == 0);
    return out;
}

}  // namespace gangfit::host
