#include "resources.hpp"

namespace gangfit::host {

namespace {
Quantity get(const ResourceList& l, const char* key) {
    auto it = l.find(key);
    return it == l.end() ? Quantity() : it->second;  // a missing map entry is the zero Quantity in Go
}
Resources from_list(const ResourceList& l) {  // getResourcesFromResourceList, resources.go:168-174
    return {get(l, kResourceCPU), get(l, kResourceMemory), get(l, kResourceNvidiaGPU)};
}
Resources subtract_from_list(const ResourceList& l, const Resources& r) {  // resources.go:137-148
    Resources out = from_list(l);
    out.Sub(r);
    return out;
}
}  // namespace

void NodeGroupSchedulingMetadata::SubtractUsageIfExists(const NodeGroupResources& used) {
    for (const auto& [name, r] : used) {
        auto it = find(name);
        if (it != end()) it->second.AvailableResources.Sub(r);
    }
}

NodeGroupResources UsageForNodes(const std::vector<ResourceReservation>& reservations) {
    NodeGroupResources res;
    for (const auto& rr : reservations)
        for (const auto& [name, reservation] : rr.Reservations) res[reservation.Node].Add(from_list(reservation.Resources));
    return res;
}

NodeGroupSchedulingMetadata NodeSchedulingMetadataForNodes(const std::vector<Node>& nodes, NodeGroupResources& currentUsage,
                                                           const NodeGroupResources& overheadUsage) {
    NodeGroupSchedulingMetadata out;
    for (const Node& node : nodes) {
        Resources overhead = Resources::Zero();
        if (auto it = overheadUsage.find(node.Name); it != overheadUsage.end()) overhead = it->second;
        Resources usage_local = Resources::Zero();
        Resources* usage = &usage_local;
        if (auto it = currentUsage.find(node.Name); it != currentUsage.end()) usage = &it->second;
        usage->Add(overhead);  // in place when the node has an entry (resources.go:76)
        NodeSchedulingMetadata m;
        m.AvailableResources = subtract_from_list(node.Allocatable, *usage);
        m.SchedulableResources = subtract_from_list(node.Allocatable, overhead);
        m.CreationTimestamp = node.CreationTimestamp;
        auto z = node.labels.find(kLabelZoneFailureDomain);
        m.ZoneLabel = z == node.labels.end() ? kZoneLabelPlaceholder : z->second;
        m.AllLabels = node.labels;
        m.Unschedulable = node.Unschedulable;
        m.Ready = node.Ready;
        out[node.Name] = std::move(m);
    }
    return out;
}

}  // namespace gangfit::host
