// failover.hpp — the one compute step of the failover reconciler, served by libgangfit (host side, C++ mirror).
//
//   findNodes                                    internal/extender/failover.go:412-436
//   constructResourceReservation (call site)     internal/extender/failover.go:350-390 (:368)
//   r.availableResources[ig].Sub(reserved)       internal/extender/failover.go:159
//   availableResourcesPerInstanceGroup           internal/extender/failover.go:286-322
// The reconciler itself (listing pods, patching / creating reservations, demands, soft reservations) is k8s API
// bookkeeping and stays in the Go host; what is mirrored here is where it asks "which nodes still hold k executors of this
// size, walking the nodes in order" — including the reference's over-add: the `reserved[n].Add(exe)` that fails the
// comparison is not taken back (:424-427), and the caller subtracts the whole map (:159).
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "gangfit.h"
#include "resources.hpp"

namespace gangfit::host {

struct FindNodesResult {
    std::vector<std::string> executorNodeNames;  // may hold fewer than executorCount names (:369-372 only logs)
    NodeGroupResources reserved;                 // what the reference returns as reservedResources
    bool served = true;                          // false: not evaluated on the device (see `error`)
    std::string error;
};

struct FindNodesRequest {
    int executorCount;            // MinExecutorCount - len(executors), > 0 (:366-367)
    Resources executorResources;
};

// findNodes(executorCount, executorResources, availableResources, orderedNodes): one request.
FindNodesResult findNodes(gf_ctx* ctx, int executorCount, const Resources& executorResources,
                          const NodeGroupResources& availableResources, const std::vector<Node>& orderedNodes);

// The reconciler's loop over the stale applications of ONE instance group as a single chained device call: request i sees
// availableResources after the `Sub` of requests 0..i-1.  *availableResources is updated like r.availableResources[ig].
std::vector<FindNodesResult> findNodesForStaleApplications(gf_ctx* ctx, const std::vector<FindNodesRequest>& requests,
                                                           NodeGroupResources* availableResources,
                                                           const std::vector<Node>& orderedNodes);

}  // namespace gangfit::host
