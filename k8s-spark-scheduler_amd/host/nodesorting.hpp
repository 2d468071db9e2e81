// nodesorting.hpp — NodeSorter.PotentialNodes: the producer of driverNodePriorityOrder / executorNodePriorityOrder.
//
//   NodeSorter, NewNodeSorter, PotentialNodes      internal/sort/nodesorting.go:25-64
//   resourcesLessThan, scheduleContextLessThan     internal/sort/nodesorting.go:73-93
//   getNodeNamesInPriorityOrder                    internal/sort/nodesorting.go:95-122
//   createLabelLessThanFunction / stable re-sort   internal/sort/nodesorting.go:161-199
//   config.LabelPriorityOrder                      config/config.go
// The reference sorts with sort.Slice (unstable) over Go-map iteration order, so where its comparator ties — two
// zones with equal free memory and cpu, two nodes of one zone with equal memory and cpu but different gpu counts —
// its output order is unspecified.  This implementation resolves those ties by zone label resp. node name, which is
// one of the orders the reference can produce.
#pragma once

#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "resources.hpp"

namespace gangfit::host {

struct LabelPriorityOrder {
    std::string Name;
    std::vector<std::string> DescendingPriorityValues;
};

bool resourcesLessThan(const Resources& left, const Resources& right);  // memory, then cpu, ascending

struct ScheduleContext {
    int azPriority;
    Resources nodeResources;
    std::string nodeName;
};
bool scheduleContextLessThan(const ScheduleContext& left, const ScheduleContext& right);

std::vector<std::string> getNodeNamesInPriorityOrder(const NodeGroupSchedulingMetadata& metadata);

// sort.SliceStable by label rank (nodes whose label value is not ranked go last, keeping their relative order)
void sortNodesByLabelPriority(std::vector<std::string>& nodeNames, const NodeGroupSchedulingMetadata& metadata,
                              const LabelPriorityOrder& order);

class NodeSorter {
public:
    NodeSorter() = default;
    NodeSorter(std::optional<LabelPriorityOrder> driver, std::optional<LabelPriorityOrder> executor)
        : driver_(std::move(driver)), executor_(std::move(executor)) {}
    // -> {driverNodes, executorNodes}
    std::pair<std::vector<std::string>, std::vector<std::string>> PotentialNodes(
        const NodeGroupSchedulingMetadata& metadata, const std::vector<std::string>& nodeNames) const;

private:
    std::optional<LabelPriorityOrder> driver_, executor_;
};

}  // namespace gangfit::host
