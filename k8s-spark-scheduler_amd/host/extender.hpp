// extender.hpp — the driver Filter decision around the gang-fit kernels (host side, C++ mirror of the Go code).
//
//   SparkSchedulerExtender.selectDriverNode          internal/extender/resource.go:272-370
//   fitEarlierDrivers / shouldSkipDriverFifo         internal/extender/resource.go:224-270
//   outcome strings                                  internal/extender/resource.go:46-55
//   newResourceReservation / executorReservationName internal/extender/resourcereservations.go:491-533
//   UnschedulablePodMarker.DoesPodExceedClusterCapacity  internal/extender/unschedulablepods.go:132-166
// The reference calls BinpackFunc once per earlier driver and once for the driver being filtered
// (resource.go:238, :321); here the whole FIFO replay + final pack is ONE GF_MODE_FIFO_CHAIN call (the "L-batch"
// insertion level of SURVEY.md section 8b).  Everything else — listers, caches, demands, metrics, the API write of the
// reservation — stays in the Go host and is represented by plain inputs / outputs.
#pragma once

#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "binpacker.hpp"
#include "nodesorting.hpp"
#include "sparkpods.hpp"

namespace gangfit::host {

namespace outcome {
constexpr const char* failureUnbound = "failure-unbound";
constexpr const char* failureInternal = "failure-internal";
constexpr const char* failureFit = "failure-fit";
constexpr const char* failureEarlierDriver = "failure-earlier-driver";
constexpr const char* failureNonSparkPod = "failure-non-spark-pod";
constexpr const char* success = "success";
constexpr const char* successRescheduled = "success-rescheduled";
constexpr const char* successScheduledExtraExecutor = "success-scheduled-extra-executor";
}  // namespace outcome

struct FifoConfig {  // config.FifoConfig
    int64_t DefaultEnforceAfterPodAgeNanos = 0;
    std::map<std::string, int64_t> EnforceAfterPodAgeByInstanceGroup;
};

// filterNodesToZone (resource.go:462-478): the nodes whose topology.kubernetes.io/zone label equals `zone`, in order; false
// (with the reference's message) when a node has no such label.
bool filterNodesToZone(const std::vector<Node>& initialNodes, const std::string& zone, std::vector<Node>* out, std::string* err);

ResourceReservation newResourceReservation(const std::string& driverNode, const std::vector<std::string>& executorNodes,
                                           const Pod& driver, const Resources& driverResources,
                                           const Resources& executorResources);
std::string executorReservationName(int i);  // "executor-<i+1>"

struct SelectNodeResult {
    std::string node;     // empty on failure
    std::string outcome;  // one of outcome::*
    std::string error;    // the reference's error text (or why the device could not serve the request)
    bool served = true;   // false: the accelerator could not serve the request -> the Go host must run its own path
    std::optional<ResourceReservation> created;  // what CreateReservations would persist
};

// The node side of the cluster in the flat form gf_snapshot_build consumes.  Built when the node set changes (an informer
// event in the Go host), reused by every Filter: names in lexicographic order give the name ranks, zone ids follow the
// label order (the tie-break documented at the C ABI).
struct FlatCluster {
    std::vector<std::string> names;                  // node index -> name (the order of `nodes` at Build time)
    std::unordered_map<std::string, uint32_t> index;
    std::vector<uint32_t> name_rank, zone, base_flags;  // base_flags: GF_NODE_UNSCHEDULABLE / GF_NODE_READY
    std::vector<std::string> zone_labels;            // zone id -> label
    std::vector<int64_t> alloc[3];
    uint64_t version = 0;  // unique per Build: lets a caller keep the columns resident on the device (gf_cluster_set)
    static bool Build(const std::vector<Node>& nodes, FlatCluster* out, std::string* err);
};

// UsageForNodes' input in flat form: one (node index, cpu milli, memory bytes, gpus) entry per reservation of every
// ResourceReservation plus the soft reservations (GetReservedResources, resourcereservations.go:258-263).  A host keeps it
// next to its ResourceReservation cache and updates it on the same events; the replay itself (the sum per node) happens on
// the device at every Filter.
struct FlatReservations {
    std::vector<uint32_t> node;
    std::vector<int64_t> req[3];
    uint64_t version = 0;  // unique per Build: lets the extender keep the usage sums of this entry list resident on the device
                           // (gf_usage_apply) for as long as the list does not change
    static bool Build(const std::vector<ResourceReservation>& reservations, const NodeGroupResources& softReservationUsage,
                      const FlatCluster& cluster, FlatReservations* out, std::string* err);
};

class SparkSchedulerExtender {
public:
    SparkSchedulerExtender(Binpacker binpacker, NodeSorter sorter, bool isFIFO, FifoConfig fifo)
        : binpacker_(std::move(binpacker)), sorter_(std::move(sorter)), isFIFO_(isFIFO), fifo_(std::move(fifo)) {}

    // Cluster state the listers / caches of the Go host would provide.
    std::vector<Node> nodes;                         // nodeLister
    std::vector<Pod> pods;                           // podLister (drivers)
    std::vector<ResourceReservation> reservations;   // resourceReservationManager
    NodeGroupResources softReservationUsage;         // added by GetReservedResources (resourcereservations.go:258-263)
    NodeGroupResources overhead;                     // overheadComputer.GetOverhead
    int64_t nowNanos = 0;                            // time.Now()
    // install.ShouldScheduleDynamicallyAllocatedExecutorsInSameAZ (config/config.go:33; the reference's test harness sets it,
    // extendertest/extender_test_utils.go:135): with a single-AZ packer an executor only goes to the zone its application's
    // running pods are in (resource.go:606-632)
    bool shouldScheduleDynamicallyAllocatedExecutorsInSameAZ = false;

    // availableNodes = nodes whose labels satisfy the driver's required node affinity; the caller passes the predicate's
    // result because affinity matching is k8s API bookkeeping (resource.go:292-298).
    SelectNodeResult selectDriverNode(const std::string& instanceGroup, const Pod& driver,
                                      const std::vector<std::string>& nodeNames, const std::vector<Node>& availableNodes);

    // The same decision with the snapshot built on the device (gf_snapshot_build): the reservation replay, the
    // available / schedulable columns and NodeSorter.PotentialNodes never touch a string-keyed map.  `cluster` must
    // describe exactly the nodes the driver's affinity matches.  Label-priority re-sorts are not configured on this path.
    // `flat` (nullable) = the reservations already in flat form; when null they are flattened from `reservations` here.
    SelectNodeResult selectDriverNodeFlat(const std::string& instanceGroup, const Pod& driver,
                                          const std::vector<std::string>& nodeNames, const FlatCluster& cluster,
                                          const FlatReservations* flat = nullptr);
    // The next selectDriverNodeFlat rebuilds the snapshot even if nothing changed (host_bench times both).
    void forgetInstalledSnapshot() { built_epoch_ = 0; }

    // unschedulablepods.go:132-166: does the application fit an EMPTY cluster (usage = 0, the given overhead)?
    // nodes are used in lister order for both candidate lists.
    bool DoesPodExceedClusterCapacity(const Pod& driver, const std::vector<Node>& availableNodes,
                                      const NodeGroupResources& nonSchedulableOverhead, bool* served, std::string* err);

    // rescheduleExecutor (resource.go:594-673) for an executor whose driver is `driver`: availableNodes = getNodes(nodeNames)
    // (already narrowed to one zone by the caller when single-AZ dynamic allocation applies, :606-633);
    // nodesHostingApp = getNodesWithExecutorsBelongingToSameApp (only read by single-az-minimal-fragmentation).
    SelectNodeResult rescheduleExecutor(const Pod& driver, const std::vector<std::string>& nodeNames,
                                        const std::vector<Node>& availableNodes,
                                        const std::set<std::string>& nodesHostingApp, bool isExtraExecutor);

    // The whole of rescheduleExecutor (resource.go:594-673) including its zone step: `executor` is the pod being filtered,
    // `allPods` what the pod lister holds (the application's pods are selected by namespace + spark-app-id label,
    // getSparkApplicationPodsForExecutor :548-555); availableNodes = getNodes(nodeNames) comes from `nodes` in nodeNames order
    // (:448-460, unknown names skipped).  With a single-AZ packer and shouldScheduleDynamicallyAllocatedExecutorsInSameAZ the
    // candidates are narrowed to the one zone (label topology.kubernetes.io/zone, quirk 7) the application's Running pods
    // are in — BEFORE the snapshot and the sort, like the reference — and an application spread over several zones is
    // scheduled anywhere (:628-630).  Errors are the reference's: (outcome "", error) from getCommonZone..., failure-internal
    // from filterNodesToZone.
    SelectNodeResult rescheduleExecutor(const Pod& executor, const Pod& driver, const std::vector<Pod>& allPods,
                                        const std::vector<std::string>& nodeNames, const std::set<std::string>& nodesHostingApp,
                                        bool isExtraExecutor);
    // getCommonZoneForExecutorsApplication (resource.go:493-519): {zone, all pods in one zone}; *err set = the reference's error
    std::pair<std::string, bool> getCommonZoneForExecutorsApplication(const Pod& executor, const std::vector<Pod>& allPods,
                                                                      std::string* err) const;

    // scanForUnschedulablePods (unschedulablepods.go:93-129) as ONE independent batch: every pending driver of this
    // scheduler that is older than timeoutNanos is checked against the EMPTY cluster (the per-pod BinpackFunc calls of the
    // reference batched into one launch).  Returns {pod name, exceedsCapacity} in listing order; like the reference the scan
    // stops at the first pod whose resources cannot be parsed (*err says why).  availableNodes = the nodes the drivers'
    // node affinity matches (one instance group per call).
    std::vector<std::pair<std::string, bool>> scanForUnschedulablePods(const std::vector<Pod>& allPods, int64_t timeoutNanos,
                                                                       const std::vector<Node>& availableNodes,
                                                                       const NodeGroupResources& nonSchedulableOverhead,
                                                                       bool* served, std::string* err);

    bool shouldSkipDriverFifo(const Pod& pod, const std::string& instanceGroup) const;

private:
    uint64_t resident_cluster_ = 0;  // FlatCluster::version whose static columns sit on the device (selectDriverNodeFlat)
    uint64_t resident_usage_ = 0;    // FlatReservations::version whose usage sums sit on the device (for resident_cluster_)
    // what the context held after this extender's last call (gf_generation): another user of the same gf_ctx that replaces the
    // cluster, the usage or the snapshot moves these on, and the next Filter sends its own state again
    uint64_t seen_cluster_gen_ = 0, seen_usage_gen_ = 0;
    // the snapshot the last Filter installed: while cluster, usage and candidate flags are the same and nobody installed
    // another one (snapshot epoch), the next Filter neither rebuilds nor re-sorts — and its chain resumes from the previous
    // chain's checkpoints (include/gangfit.h, "Incremental FIFO chains")
    uint64_t built_epoch_ = 0, built_cluster_ = 0, built_usage_ = 0;
    std::vector<uint32_t> built_flags_;
    // The reference parses the nine annotations of every earlier driver on every Filter (sparkpods.go:73-137 from
    // resource.go:231) and matches the request's NodeNames against the node set again.  Both are pure functions of things
    // that carry a version: the flat route keeps the canonical requests per (pod UID, resourceVersion) and the candidate
    // flags per (cluster version, NodeNames).
    // Predicate requests run on per-request threads (cmd/endpoints.go:29-37): everything below is read and written under
    // flat_mu_, which a flat Filter takes before it touches the caches and holds until it returns (always before the
    // context's sequence lock, never the other way round).
    struct ParsedApp {
        uint64_t version = 0;
        uint64_t seen = 0;  // flat_calls_ of the last Filter that used the entry: what a prune keeps
        bool ok = false, representable = false;
        gf_app app{};
    };
    std::shared_ptr<std::mutex> flat_mu_ = std::make_shared<std::mutex>();  // (the extender stays copyable: copies share it)
    uint64_t flat_calls_ = 0;
    std::unordered_map<std::string, ParsedApp> parsed_apps_;  // pruned to the pods of the current request when it outgrows them
    uint64_t flags_cluster_ = 0, flags_hash_ = 0;
    std::vector<std::string> flags_names_;  // the NodeNames the cached flags were computed from (a hash hit is confirmed on them)
    std::vector<uint32_t> flags_cache_;
    Binpacker binpacker_;
    NodeSorter sorter_;
    bool isFIFO_;
    FifoConfig fifo_;
};

}  // namespace gangfit::host
