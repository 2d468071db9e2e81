// sparkpods.hpp — what the driver-Filter path reads from pods (host side, C++ mirror of the Go code).
//
//   annotation / label names                 internal/common/constants.go:17-51
//   sparkResources                           internal/extender/sparkpods.go:73-137
//   filterToEarliestAndSort                  internal/extender/sparkpods.go:54-71   (FIFO predecessor set)
//   sparkResourceUsage                       internal/extender/sparkpods.go:139-146 (map overwrite quirk)
//   types.SparkApplicationResources          internal/types/types.go:22-27
#pragma once

#include <optional>
#include <string>
#include <vector>

#include "resources.hpp"

namespace gangfit::host {

namespace common {
constexpr const char* SparkSchedulerName = "spark-scheduler";
constexpr const char* SparkRoleLabel = "spark-role";
constexpr const char* SparkAppIDLabel = "spark-app-id";
constexpr const char* Driver = "driver";
constexpr const char* Executor = "executor";
constexpr const char* DriverCPU = "spark-driver-cpu";
constexpr const char* DriverMemory = "spark-driver-mem";
constexpr const char* DriverNvidiaGPUs = "spark-driver-nvidia.com/gpu";
constexpr const char* ExecutorCPU = "spark-executor-cpu";
constexpr const char* ExecutorMemory = "spark-executor-mem";
constexpr const char* ExecutorNvidiaGPUs = "spark-executor-nvidia.com/gpu";
constexpr const char* DynamicAllocationEnabled = "spark-dynamic-allocation-enabled";
constexpr const char* ExecutorCount = "spark-executor-count";
constexpr const char* DAMinExecutorCount = "spark-dynamic-allocation-min-executor-count";
constexpr const char* DAMaxExecutorCount = "spark-dynamic-allocation-max-executor-count";
}  // namespace common

struct Pod {  // the fields of corev1.Pod this path reads
    std::string Name, Namespace, UID;
    Labels labels;
    std::map<std::string, std::string> Annotations;
    int64_t CreationTimestampNanos = 0;  // metav1.Time
    std::string NodeName;                // Spec.NodeName
    std::string Phase;                   // Status.Phase ("Pending", "Running", ...): filterToRunningPods, resource.go:535-545
    std::string SchedulerName;           // Spec.SchedulerName
    bool Deleting = false;               // DeletionTimestamp != nil
    std::string InstanceGroup;           // value the pod's node affinity / selector requires for the instance-group label
                                         // (internal.MatchPodInstanceGroup compares exactly this, internal/podspec.go)
    uint64_t ResourceVersion = 0;        // metadata.resourceVersion as a number; 0 = unknown.  Annotations are immutable per
                                         // version: selectDriverNodeFlat keeps the parsed requests of a (UID, version) pair
};

struct SparkApplicationResources {
    Resources DriverResources, ExecutorResources;
    int MinExecutorCount = 0, MaxExecutorCount = 0;
};

// Returns std::nullopt and fills *err with the reference's message on failure.
std::optional<SparkApplicationResources> sparkResources(const Pod& pod, std::string* err);

std::vector<const Pod*> filterToEarliestAndSort(const Pod& driver, const std::vector<Pod>& allDrivers);

NodeGroupResources sparkResourceUsage(const Resources& driverResources, const Resources& executorResources,
                                      const std::string& driverNode, const std::vector<std::string>& executorNodes);

}  // namespace gangfit::host
