#include "extender.hpp"

#include <cstring>
#include <algorithm>
#include <atomic>

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
    // (this route parses every pod afresh and never touches parsed_apps_: the cache and its prune belong to the flat route,
    //  under flat_mu_)
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
    CtxSequence seq(binpacker_.ctx);
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
    CtxSequence seq(binpacker_.ctx);
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

bool filterNodesToZone(const std::vector<Node>& initialNodes, const std::string& zone, std::vector<Node>* out, std::string* err) {
    out->clear();
    for (const Node& node : initialNodes) {
        auto z = node.labels.find(kLabelTopologyZone);
        if (z == node.labels.end()) {  // resource.go:466-468
            if (err) *err = "Could not read zone label from node, unable to make scheduling decisions based on AZ";
            return false;
        }
        if (z->second == zone) out->push_back(node);
    }
    return true;
}

std::pair<std::string, bool> SparkSchedulerExtender::getCommonZoneForExecutorsApplication(const Pod& executor,
                                                                                         const std::vector<Pod>& allPods,
                                                                                         std::string* err) const {
    if (err) err->clear();
    auto label = executor.labels.find(common::SparkAppIDLabel);
    if (label == executor.labels.end()) {  // :494-497
        if (err) *err = "Executor does not have a Spark app id label, could not create label selector";
        return {"", false};
    }
    std::set<std::string> azs;
    for (const Pod& pod : allPods) {  // podLister.Pods(namespace).List(selector spark-app-id == label), :548-555
        if (pod.Namespace != executor.Namespace) continue;
        auto l = pod.labels.find(common::SparkAppIDLabel);
        if (l == pod.labels.end() || l->second != label->second) continue;
        if (pod.Phase != "Running") continue;  // filterToRunningPods, :535-545: other phases are not assigned to a node
        const Node* node = nullptr;            // getAzsOfPods, :521-533
        for (const Node& n : nodes)
            if (n.Name == pod.NodeName) {
                node = &n;
                break;
            }
        if (node == nullptr) {
            if (err) *err = "node \"" + pod.NodeName + "\" not found";  // the lister's NotFound error
            return {"", false};
        }
        auto z = node->labels.find(kLabelTopologyZone);
        if (z == node->labels.end()) {
            if (err) *err = "Could not read zone label from node, unable to make scheduling decisions based on AZ";
            return {"", false};
        }
        azs.insert(z->second);
    }
    if (azs.size() > 1) return {"", false};  // :511-513
    if (azs.empty()) {                        // :514-516
        if (err) *err = "Application has no scheduled pods, can't make scheduling decisions based on AZ";
        return {"", false};
    }
    return {*azs.begin(), true};
}

SelectNodeResult SparkSchedulerExtender::rescheduleExecutor(const Pod& executor, const Pod& driver, const std::vector<Pod>& allPods,
                                                            const std::vector<std::string>& nodeNames,
                                                            const std::set<std::string>& nodesHostingApp, bool isExtraExecutor) {
    SelectNodeResult out;
    std::string err;
    // sparkResources(driver) comes first in the reference (:599-602): its failure is failure-internal whatever the zones say
    if (!sparkResources(driver, &err)) {
        out.outcome = outcome::failureInternal;
        out.error = err;
        return out;
    }
    std::vector<Node> availableNodes;  // getNodes (:448-460): lister order of nodeNames, unknown names skipped
    for (const std::string& name : nodeNames)
        for (const Node& n : nodes)
            if (n.Name == name) {
                availableNodes.push_back(n);
                break;
            }
    std::vector<std::string> names = nodeNames;
    if (binpacker_.IsSingleAz && shouldScheduleDynamicallyAllocatedExecutorsInSameAZ) {  // :608
        auto [zone, allPodsInSameAz] = getCommonZoneForExecutorsApplication(executor, allPods, &err);
        if (!err.empty()) {  // :611-613: ("", "", err)
            out.outcome = "";
            out.error = err;
            return out;
        }
        if (allPodsInSameAz) {  // :614-627
            std::vector<Node> inZone;
            if (!filterNodesToZone(availableNodes, zone, &inZone, &err)) {
                out.outcome = outcome::failureInternal;
                out.error = err;
                return out;
            }
            availableNodes = std::move(inZone);
            names.clear();
            for (const Node& n : availableNodes) names.push_back(n.Name);
        }
    }
    return rescheduleExecutor(driver, names, availableNodes, nodesHostingApp, isExtraExecutor);
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
    CtxSequence seq(binpacker_.ctx);
    if (!flatten(metadata, names, names, &snap, &e) || !upload(binpacker_.ctx, snap, &e)) {
        if (served) *served = false;
        if (err) *err = e;
        return {};
    }
    // DoesPodExceedClusterCapacity reads HasCapacity and nothing else (unschedulablepods.go:165): the feasibility-only batch
    std::vector<uint8_t> fits(apps.size());
    if (gf_fit_feasible(binpacker_.ctx, binpacker_.Algo, (uint32_t)apps.size(), apps.data(), fits.data()) != GF_OK) {
        if (served) *served = false;
        if (err) *err = std::string("gf_fit_feasible: ") + gf_last_error(binpacker_.ctx);
        return {};
    }
    for (size_t i = 0; i < stale.size(); ++i) out.emplace_back(stale[i]->Name, fits[i] == 0);
    return out;
}

}  // namespace gangfit::host

namespace gangfit::host {

bool FlatCluster::Build(const std::vector<Node>& nodes, FlatCluster* out, std::string* err) {
    static std::atomic<uint64_t> next_version{1};
    FlatCluster c;
    c.version = next_version.fetch_add(1);
    const size_t n = nodes.size();
    std::map<std::string, uint32_t> zone_ids;  // label order
    for (const Node& nd : nodes) {
        auto z = nd.labels.find(kLabelZoneFailureDomain);
        zone_ids.emplace(z == nd.labels.end() ? kZoneLabelPlaceholder : z->second, 0u);
    }
    uint32_t zi = 0;
    for (auto& kv : zone_ids) {
        kv.second = zi++;
        c.zone_labels.push_back(kv.first);
    }
    std::vector<std::pair<std::string, uint32_t>> by_name;
    by_name.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const Node& nd = nodes[i];
        Resources a = Resources::Zero();
        if (auto it = nd.Allocatable.find(kResourceCPU); it != nd.Allocatable.end()) a.CPU = it->second;
        if (auto it = nd.Allocatable.find(kResourceMemory); it != nd.Allocatable.end()) a.Memory = it->second;
        if (auto it = nd.Allocatable.find(kResourceNvidiaGPU); it != nd.Allocatable.end()) a.NvidiaGPU = it->second;
        int64_t v[3];
        if (!a.canonical(v) || v[0] < 0 || v[1] < 0 || v[2] < 0) {
            if (err) *err = "allocatable of node " + nd.Name + " is not exactly representable";
            return false;
        }
        for (int j = 0; j < 3; ++j) c.alloc[j].push_back(v[j]);
        c.names.push_back(nd.Name);
        if (!c.index.emplace(nd.Name, (uint32_t)i).second) {
            if (err) *err = "duplicate node name " + nd.Name;
            return false;
        }
        auto z = nd.labels.find(kLabelZoneFailureDomain);
        c.zone.push_back(zone_ids.at(z == nd.labels.end() ? kZoneLabelPlaceholder : z->second));
        c.base_flags.push_back((nd.Unschedulable ? GF_NODE_UNSCHEDULABLE : 0u) | (nd.Ready ? GF_NODE_READY : 0u));
        by_name.emplace_back(nd.Name, (uint32_t)i);
    }
    std::sort(by_name.begin(), by_name.end());
    c.name_rank.assign(n, 0);
    for (size_t r = 0; r < n; ++r) c.name_rank[by_name[r].second] = (uint32_t)r;
    *out = std::move(c);
    return true;
}

bool FlatReservations::Build(const std::vector<ResourceReservation>& reservations,
                             const NodeGroupResources& softReservationUsage, const FlatCluster& cluster,
                             FlatReservations* out, std::string* err) {
    static std::atomic<uint64_t> next_version{1};
    FlatReservations f;
    f.version = next_version.fetch_add(1);
    auto push = [&](const std::string& node, const Resources& r) {
        auto it = cluster.index.find(node);
        if (it == cluster.index.end()) return true;  // usage of a node outside this instance group is never read
        int64_t v[3];
        if (!r.canonical(v) || v[0] < 0 || v[1] < 0 || v[2] < 0) return false;
        f.node.push_back(it->second);
        for (int j = 0; j < 3; ++j) f.req[j].push_back(v[j]);
        return true;
    };
    for (const ResourceReservation& rr : reservations)
        for (const auto& [name, res] : rr.Reservations) {
            Resources r = Resources::Zero();
            if (auto it = res.Resources.find(kResourceCPU); it != res.Resources.end()) r.CPU = it->second;
            if (auto it = res.Resources.find(kResourceMemory); it != res.Resources.end()) r.Memory = it->second;
            if (auto it = res.Resources.find(kResourceNvidiaGPU); it != res.Resources.end()) r.NvidiaGPU = it->second;
            if (!push(res.Node, r)) {
                if (err) *err = "a reservation of " + rr.Name + " is not exactly representable";
                return false;
            }
        }
    for (const auto& [node, r] : softReservationUsage)
        if (!push(node, r)) {
            if (err) *err = "soft reservations on " + node + " are not exactly representable";
            return false;
        }
    *out = std::move(f);
    return true;
}

SelectNodeResult SparkSchedulerExtender::selectDriverNodeFlat(const std::string& instanceGroup, const Pod& driver,
                                                              const std::vector<std::string>& nodeNames,
                                                              const FlatCluster& cluster, const FlatReservations* flat) {
    SelectNodeResult out;
    auto not_served = [&](const std::string& why) {
        out.served = false;
        out.error = why;
        return out;
    };
    auto app_label = driver.labels.find(common::SparkAppIDLabel);
    const std::string app_id = app_label == driver.labels.end() ? "" : app_label->second;
    for (const ResourceReservation& rr : reservations)
        if (rr.Name == app_id && rr.Namespace == driver.Namespace) {
            auto d = rr.Reservations.find("driver");
            out.node = d == rr.Reservations.end() ? "" : d->second.Node;
            out.outcome = outcome::success;
            return out;
        }
    const uint32_t n = (uint32_t)cluster.names.size();
    // ---- the per-request columns: one entry per reservation (UsageForNodes' input), overhead, request flags
    FlatReservations local;
    if (flat == nullptr) {
        std::string ferr;
        if (!FlatReservations::Build(reservations, softReservationUsage, cluster, &local, &ferr)) return not_served(ferr);
        flat = &local;
    }
    const std::vector<uint32_t>& rnode = flat->node;
    const std::vector<int64_t>* rreq = flat->req;
    std::vector<int64_t> over[3];
    if (!overhead.empty()) {
        for (int j = 0; j < 3; ++j) over[j].assign(n, 0);
        for (const auto& [node, r] : overhead) {
            auto it = cluster.index.find(node);
            if (it == cluster.index.end()) continue;
            int64_t v[3];
            if (!r.canonical(v) || v[0] < 0 || v[1] < 0 || v[2] < 0) return not_served("overhead of " + node + " is not exactly representable");
            for (int j = 0; j < 3; ++j) over[j][it->second] = v[j];
        }
    }
    // ---- from here on this Filter reads and writes the extender's caches and its record of what sits on the device
    std::lock_guard<std::mutex> flat_lock(*flat_mu_);
    ++flat_calls_;
    // the request's NodeNames as candidate flags: the same list for every Filter of an instance group until the node set
    // changes, so the 10 000 map lookups are done once per (cluster version, list)
    // (eight bytes per multiply: a byte-at-a-time hash of 100 000 names is a 1.5 ms dependent chain)
    uint64_t names_hash = 1469598103934665603ull;
    for (const std::string& name : nodeNames) {
        const char* p = name.data();
        size_t left = name.size();
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)left;
        for (; left >= 8; left -= 8, p += 8) {
            uint64_t w;
            std::memcpy(&w, p, 8);
            h = (h ^ w) * 0xFF51AFD7ED558CCDull;
            h ^= h >> 29;
        }
        if (left) {
            uint64_t w = 0;
            std::memcpy(&w, p, left);
            h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
            h ^= h >> 31;
        }
        names_hash = (names_hash ^ h) * 1099511628211ull;  // order-sensitive across names; h of each name is independent work
    }
    // (a hash hit is confirmed on the list itself: comparing 10 000 short strings costs a fraction of the 10 000 map lookups
    //  it saves, and a collision would silently change the node the Filter returns)
    if (cluster.version == 0 || flags_cluster_ != cluster.version || flags_hash_ != names_hash || flags_names_ != nodeNames) {
        flags_cache_ = cluster.base_flags;
        for (const std::string& name : nodeNames)
            if (auto it = cluster.index.find(name); it != cluster.index.end()) flags_cache_[it->second] |= GF_NODE_DRIVER_CANDIDATE;
        flags_cluster_ = cluster.version;
        flags_hash_ = names_hash;
        flags_names_ = nodeNames;
    }
    const std::vector<uint32_t>& flags = flags_cache_;
    // ---- the applications: earlier drivers in creation order, then this one
    std::string err;
    auto resources = sparkResources(driver, &err);
    if (!resources) {
        out.outcome = outcome::failureInternal;
        out.error = "failed to get spark resources: " + err;
        return out;
    }
    std::vector<gf_app> apps;
    if (isFIFO_)
        for (const Pod* p : filterToEarliestAndSort(driver, pods)) {
            // annotations -> canonical requests, once per (pod, resourceVersion); the skip flag depends on the clock
            ParsedApp fresh;
            ParsedApp* pa = &fresh;
            bool cached = false;
            if (p->ResourceVersion != 0) {
                pa = &parsed_apps_[p->UID.empty() ? p->Namespace + "/" + p->Name : p->UID];
                cached = pa->version == p->ResourceVersion;
            }
            pa->seen = flat_calls_;
            if (!cached) {
                pa->version = p->ResourceVersion;
                pa->app = gf_app{};
                auto r = sparkResources(*p, nullptr);
                pa->ok = r.has_value();
                pa->representable = pa->ok && r->DriverResources.canonical(pa->app.drv) && r->ExecutorResources.canonical(pa->app.exe) &&
                                    r->MinExecutorCount >= 0 && r->MinExecutorCount <= GF_MAX_K;
                if (pa->representable) pa->app.k = r->MinExecutorCount;
            }
            if (!pa->ok) continue;
            if (!pa->representable) return not_served("earlier driver " + p->Name + " is not exactly representable");
            gf_app a = pa->app;
            a.flags = shouldSkipDriverFifo(*p, instanceGroup) ? GF_APP_SKIPPABLE : 0u;
            apps.push_back(a);
        }
    // pods come and go: entries no request has used lately are dropped once the map outgrows the pods it is asked about
    if (parsed_apps_.size() > 2 * pods.size() + 64)
        for (auto it = parsed_apps_.begin(); it != parsed_apps_.end();) it = it->second.seen == flat_calls_ ? std::next(it) : parsed_apps_.erase(it);
    gf_app cur{};
    if (!resources->DriverResources.canonical(cur.drv) || !resources->ExecutorResources.canonical(cur.exe) ||
        resources->MinExecutorCount < 0 || resources->MinExecutorCount > GF_MAX_K)
        return not_served("application resources are not exactly representable");
    cur.k = resources->MinExecutorCount;
    apps.push_back(cur);
    // ---- snapshot + orders on the device, then the chain
    gf_ctx* ctx = binpacker_.ctx;
    CtxSequence seq(ctx);
    if (overhead.empty()) {
        // the node-side columns change only when the node set does: they stay on the device (gf_cluster_set), a Filter moves
        // its reservation entries and its own candidate flags only.  Another user of the context (the UnschedulablePodMarker
        // shares it) may have replaced the resident cluster or usage in between: the context's generations say so.
        uint64_t gen[3] = {0, 0, 0};
        (void)gf_generation(ctx, gen);
        if (resident_cluster_ != cluster.version || cluster.version == 0 || gen[1] != seen_cluster_gen_) {
            if (gf_cluster_set(ctx, n, cluster.alloc[0].data(), cluster.alloc[1].data(), cluster.alloc[2].data(), nullptr, nullptr,
                               nullptr, cluster.base_flags.data(), cluster.zone.data(), (uint32_t)cluster.zone_labels.size(),
                               cluster.name_rank.data()) != GF_OK)
                return not_served(std::string("gf_cluster_set: ") + gf_last_error(ctx));
            resident_cluster_ = cluster.version;
            resident_usage_ = 0;  // gf_cluster_set zeroed the resident usage
            (void)gf_generation(ctx, gen);
            seen_cluster_gen_ = gen[1];
            seen_usage_gen_ = gen[2];
        }
        // a caller that keeps its flattened reservations (flat->version != 0) gets the usage sums kept on the device too: they
        // are sent when the list changes, and a Filter between two changes moves no reservation at all.  (A host that tracks
        // its ResourceReservation events would send only the K + 1 entries of the object that changed: gf_usage_apply.)
        const bool keep_usage = flat != &local && flat->version != 0;
        if (keep_usage && (resident_usage_ != flat->version || gen[2] != seen_usage_gen_)) {
            resident_usage_ = 0;
            if (gf_usage_reset(ctx) != GF_OK ||
                gf_usage_apply(ctx, (uint32_t)rnode.size(), rnode.data(), rreq[0].data(), rreq[1].data(), rreq[2].data(), +1) != GF_OK)
                return not_served(std::string("gf_usage_apply: ") + gf_last_error(ctx));
            resident_usage_ = flat->version;
            (void)gf_generation(ctx, gen);
            seen_usage_gen_ = gen[2];
        }
        // nothing changed since this extender's last Filter: the installed snapshot IS this request's snapshot
        const bool same_snapshot = keep_usage && built_epoch_ != 0 && gen[0] == built_epoch_ && built_cluster_ == cluster.version &&
                                   built_usage_ == flat->version && built_flags_ == flags;
        if (!same_snapshot) {
            built_epoch_ = 0;
            const int brc = keep_usage
                                ? gf_snapshot_build_resident(ctx, GF_RESIDENT_USAGE, nullptr, nullptr, nullptr, nullptr, flags.data(),
                                                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)
                                : gf_snapshot_build_resident(ctx, (uint32_t)rnode.size(), rnode.data(), rreq[0].data(),
                                                             rreq[1].data(), rreq[2].data(), flags.data(), nullptr, nullptr, nullptr,
                                                             nullptr, nullptr, nullptr);
            if (brc != GF_OK) {
                resident_cluster_ = 0;
                resident_usage_ = 0;
                return not_served(std::string("gf_snapshot_build_resident: ") + gf_last_error(ctx));
            }
            if (keep_usage && gf_generation(ctx, gen) == GF_OK) {
                built_epoch_ = gen[0];
                built_cluster_ = cluster.version;
                built_usage_ = flat->version;
                built_flags_ = flags;
            }
        }
    } else if (gf_snapshot_build(ctx, n, cluster.alloc[0].data(), cluster.alloc[1].data(), cluster.alloc[2].data(), over[0].data(),
                                 over[1].data(), over[2].data(), (uint32_t)rnode.size(), rnode.data(), rreq[0].data(),
                                 rreq[1].data(), rreq[2].data(), flags.data(), cluster.zone.data(),
                                 (uint32_t)cluster.zone_labels.size(), cluster.name_rank.data(), nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr) != GF_OK) {
        resident_cluster_ = 0;
        return not_served(std::string("gf_snapshot_build: ") + gf_last_error(ctx));
    } else {
        resident_cluster_ = 0;  // gf_snapshot_build replaced the resident cluster (with this request's overhead)
        built_epoch_ = 0;
    }
    uint64_t total_k = 0;
    for (const gf_app& a : apps) total_k += (uint64_t)a.k;
    std::vector<gf_result> results(apps.size());
    std::vector<uint32_t> exec(total_k + 1);
    int32_t failed_at = -1;
    if (gf_fit_batch(ctx, GF_MODE_FIFO_CHAIN, binpacker_.Algo, (uint32_t)apps.size(), apps.data(), results.data(), exec.data(),
                     total_k, &failed_at) != GF_OK)
        return not_served(std::string("gf_fit_batch: ") + gf_last_error(ctx));
    if (failed_at >= 0) {
        out.outcome = outcome::failureEarlierDriver;
        out.error = "earlier drivers do not fit to the cluster";
        return out;
    }
    const gf_result& last = results.back();
    if (!last.has_capacity) {
        out.outcome = outcome::failureFit;
        out.error = "application does not fit to the cluster";
        return out;
    }
    std::vector<std::string> executorNodes;
    const uint64_t off = total_k - (uint64_t)cur.k;
    for (uint32_t i = 0; i < last.exec_len; ++i) executorNodes.push_back(cluster.names[exec[off + i]]);
    out.node = cluster.names[last.driver_node];
    out.outcome = outcome::success;
    out.created = newResourceReservation(out.node, executorNodes, driver, resources->DriverResources,
                                         resources->ExecutorResources);
    return out;
}

}  // namespace gangfit::host
