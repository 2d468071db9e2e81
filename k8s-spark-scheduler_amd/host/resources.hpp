// resources.hpp — the scheduling data model of the gang-fit path (host side, C++ mirror of the Go types).
//
// LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg in the reference.
//   Resources, Zero, Add/Sub/GreaterThan/Eq/Copy            LIB/resources/resources.go:151-246
//   NodeSchedulingMetadata, NodeGroupSchedulingMetadata      LIB/resources/resources.go:102-166
//   UsageForNodes                                            LIB/resources/resources.go:31-43
//   NodeSchedulingMetadataForNodes, subtractFromResourceList LIB/resources/resources.go:61-100, 137-148
//   SubtractUsageIfExists                                    LIB/resources/resources.go:129-135
// Only the fields of v1.Node / v1.Pod / ResourceReservation that this path reads are modelled; the k8s API objects
// themselves stay in the Go host.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "quantity.hpp"

namespace gangfit::host {

constexpr const char* kResourceCPU = "cpu";                 // corev1.ResourceCPU
constexpr const char* kResourceMemory = "memory";           // corev1.ResourceMemory
constexpr const char* kResourceNvidiaGPU = "nvidia.com/gpu";  // v1beta2.ResourceNvidiaGPU
constexpr const char* kLabelZoneFailureDomain = "failure-domain.beta.kubernetes.io/zone";  // corev1.LabelZoneFailureDomain
constexpr const char* kZoneLabelPlaceholder = "default";    // resources.go:26
// corev1.LabelTopologyZone: the label the EXECUTOR path's zone filter reads (resource.go:466, :526) — not the one the bin-pack
// snapshot reads
constexpr const char* kLabelTopologyZone = "topology.kubernetes.io/zone";

struct Resources {
    Quantity CPU, Memory, NvidiaGPU;
    static Resources Zero() { return {}; }
    static Resources Create(int64_t cpu, int64_t memory, int64_t gpus) {  // CreateResources, resources.go:243-250
        return {Quantity::FromInt(cpu), Quantity::FromInt(memory), Quantity::FromInt(gpus)};
    }
    void Add(const Resources& o) {
        CPU.Add(o.CPU);
        Memory.Add(o.Memory);
        NvidiaGPU.Add(o.NvidiaGPU);
    }
    void Sub(const Resources& o) {
        CPU.Sub(o.CPU);
        Memory.Sub(o.Memory);
        NvidiaGPU.Sub(o.NvidiaGPU);
    }
    bool GreaterThan(const Resources& o) const {  // ANY component greater, resources.go:239-241
        return CPU.Cmp(o.CPU) > 0 || Memory.Cmp(o.Memory) > 0 || NvidiaGPU.Cmp(o.NvidiaGPU) > 0;
    }
    bool Eq(const Resources& o) const {
        return CPU.Cmp(o.CPU) == 0 && Memory.Cmp(o.Memory) == 0 && NvidiaGPU.Cmp(o.NvidiaGPU) == 0;
    }
    // canonical int64 triple {cpu milli, memory bytes, gpu devices}; false if any component is not exactly representable
    bool canonical(int64_t out[3]) const {
        return CPU.canonical_milli(&out[0]) && Memory.canonical_units(&out[1]) && NvidiaGPU.canonical_units(&out[2]);
    }
};

using ResourceList = std::map<std::string, Quantity>;  // corev1.ResourceList / v1beta2.ResourceList
using Labels = std::map<std::string, std::string>;

struct Node {  // the fields of corev1.Node read by resources.go:61-100
    std::string Name;
    Labels labels;
    ResourceList Allocatable;
    bool Unschedulable = false;
    bool Ready = false;                 // condition NodeReady == True
    int64_t CreationTimestamp = 0;      // unix seconds
};

struct Reservation {  // v1beta2.Reservation
    std::string Node;
    ResourceList Resources;
};
struct ResourceReservation {  // v1beta2.ResourceReservation (spec.reservations + status.pods + identifying metadata)
    std::string Name, Namespace, OwnerPodName, AppIDLabel;
    std::map<std::string, Reservation> Reservations;  // "driver", "executor-1", ...
    std::map<std::string, std::string> Pods;          // reservation name -> pod name
};

struct NodeSchedulingMetadata {  // resources.go:158-166
    Resources AvailableResources, SchedulableResources;
    int64_t CreationTimestamp = 0;
    std::string ZoneLabel;
    Labels AllLabels;
    bool Unschedulable = false;
    bool Ready = false;
};

using NodeGroupResources = std::map<std::string, Resources>;
struct NodeGroupSchedulingMetadata : std::map<std::string, NodeSchedulingMetadata> {
    void SubtractUsageIfExists(const NodeGroupResources& used);  // resources.go:129-135
};

NodeGroupResources UsageForNodes(const std::vector<ResourceReservation>& reservations);  // resources.go:31-43
// NOTE the reference mutates currentUsage[node] in place (`currentUsageForNode.Add(overhead)`, resources.go:72-76 —
// SURVEY.md quirk 5); so does this function, through the non-const reference.
NodeGroupSchedulingMetadata NodeSchedulingMetadataForNodes(const std::vector<Node>& nodes, NodeGroupResources& currentUsage,
                                                           const NodeGroupResources& overheadUsage);

}  // namespace gangfit::host
