#include "nodesorting.hpp"

#include <algorithm>
#include <map>
#include <set>

namespace gangfit::host {

bool resourcesLessThan(const Resources& left, const Resources& right) {
    const int m = left.Memory.Cmp(right.Memory);
    if (m != 0) return m == -1;
    return left.CPU.Cmp(right.CPU) == -1;
}

bool scheduleContextLessThan(const ScheduleContext& left, const ScheduleContext& right) {
    if (left.azPriority != right.azPriority) return left.azPriority < right.azPriority;
    if (!left.nodeResources.Eq(right.nodeResources)) return resourcesLessThan(left.nodeResources, right.nodeResources);
    return left.nodeName < right.nodeName;
}

std::vector<std::string> getNodeNamesInPriorityOrder(const NodeGroupSchedulingMetadata& metadata) {
    // free resources per zone (getAvailableResourcesByAZ, :124-134); std::map iterates zones by label
    std::map<std::string, Resources> by_az;
    for (const auto& [name, m] : metadata) by_az[m.ZoneLabel].Add(m.AvailableResources);
    std::vector<std::string> az_labels;
    for (const auto& kv : by_az) az_labels.push_back(kv.first);
    std::stable_sort(az_labels.begin(), az_labels.end(), [&](const std::string& a, const std::string& b) {
        return resourcesLessThan(by_az.at(a), by_az.at(b));  // :102-104; ties stay in label order
    });
    std::map<std::string, int> az_priority;
    for (size_t i = 0; i < az_labels.size(); ++i) az_priority[az_labels[i]] = (int)i;
    std::vector<ScheduleContext> ctxs;
    ctxs.reserve(metadata.size());
    for (const auto& [name, m] : metadata) ctxs.push_back({az_priority.at(m.ZoneLabel), m.AvailableResources, name});
    // scheduleContextLessThan (:84-93) is not a strict weak order when two nodes tie on memory and cpu but differ in
    // gpu count (it answers "not less" both ways for them, yet orders each against a third node by name), so the
    // reference's sort.Slice output is unspecified there.  Sorting by the total order (az, memory, cpu, name) agrees
    // with scheduleContextLessThan wherever that is consistent and is deterministic elsewhere.
    std::sort(ctxs.begin(), ctxs.end(), [](const ScheduleContext& l, const ScheduleContext& r) {
        if (l.azPriority != r.azPriority) return l.azPriority < r.azPriority;
        const int m = l.nodeResources.Memory.Cmp(r.nodeResources.Memory);
        if (m != 0) return m < 0;
        const int c = l.nodeResources.CPU.Cmp(r.nodeResources.CPU);
        if (c != 0) return c < 0;
        return l.nodeName < r.nodeName;
    });
    std::vector<std::string> names;
    names.reserve(ctxs.size());
    for (auto& c : ctxs) names.push_back(std::move(c.nodeName));
    return names;
}

void sortNodesByLabelPriority(std::vector<std::string>& nodeNames, const NodeGroupSchedulingMetadata& metadata,
                              const LabelPriorityOrder& order) {
    std::map<std::string, int> ranks;
    for (size_t i = 0; i < order.DescendingPriorityValues.size(); ++i) ranks[order.DescendingPriorityValues[i]] = (int)i;
    auto rank_of = [&](const std::string& node, int* rank) {  // extractRank, :183-190
        auto m = metadata.find(node);
        if (m == metadata.end()) return false;
        auto l = m->second.AllLabels.find(order.Name);
        if (l == m->second.AllLabels.end()) return false;
        auto r = ranks.find(l->second);
        if (r == ranks.end()) return false;
        *rank = r->second;
        return true;
    };
    std::stable_sort(nodeNames.begin(), nodeNames.end(), [&](const std::string& a, const std::string& b) {
        int ra = 0, rb = 0;
        if (!rank_of(a, &ra)) return false;  // :171-174
        if (!rank_of(b, &rb)) return true;   // :175-178
        return ra < rb;
    });
}

std::pair<std::vector<std::string>, std::vector<std::string>> NodeSorter::PotentialNodes(
    const NodeGroupSchedulingMetadata& metadata, const std::vector<std::string>& nodeNames) const {
    const std::vector<std::string> in_order = getNodeNamesInPriorityOrder(metadata);
    const std::set<std::string> requested(nodeNames.begin(), nodeNames.end());
    std::vector<std::string> drivers, executors;
    for (const std::string& name : in_order) {
        if (requested.count(name)) drivers.push_back(name);  // :52-54
        const NodeSchedulingMetadata& m = metadata.at(name);
        if (!m.Unschedulable && m.Ready) executors.push_back(name);  // :55-57
    }
    if (driver_) sortNodesByLabelPriority(drivers, metadata, *driver_);
    if (executor_) sortNodesByLabelPriority(executors, metadata, *executor_);
    return {drivers, executors};
}

}  // namespace gangfit::host
