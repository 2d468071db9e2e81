// binpacker.hpp — the plug-in seam of the reference, served by libgangfit (host side, C++ mirror of the Go interface).
//
//   binpack.PackingResult, SparkBinPackFunction    LIB/binpack/binpack.go:25-48
//   Binpacker, binpackFunctions, SelectBinpacker   internal/binpacker/binpack.go:37-58
//   PackingEfficiency                              LIB/binpack/efficiency.go:25-64
// BinpackFunc has the argument list of SparkBinPackFunction (string node names, Quantity resources, the scheduling
// metadata map); it flattens them to the C ABI of include/gangfit.h and runs the decision on the GPU.  When the inputs
// are not exactly representable in the canonical int64 units, or the device path fails, `served` is false and the
// caller (in the Go host: the cgo shim) must use the Go packer — there is no CPU packer in this library.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "gangfit.h"
#include "resources.hpp"

namespace gangfit::host {

struct PackingEfficiency {
    std::string NodeName;
    double CPU = 0, Memory = 0, GPU = 0;
};

struct PackingResult {
    std::string DriverNode;
    std::vector<std::string> ExecutorNodes;
    std::map<std::string, PackingEfficiency> PackingEfficiencies;
    bool HasCapacity = false;
    bool served = true;      // false: not evaluated on the device (see `error`)
    std::string error;
};

// Flattened snapshot shared by the single-decision and the batched entry points.
struct FlatSnapshot {
    std::vector<std::string> names;            // node index -> name (metadata keys, sorted)
    std::map<std::string, uint32_t> index;
    std::vector<int64_t> avail[3], sched[3];
    std::vector<uint32_t> zone;
    std::vector<uint32_t> driver_order, exec_order;
    bool sched_ok = true;                      // schedulable columns representable (efficiencies / single-AZ packers)
};
// false (with *err) when an available quantity is not exactly representable.
bool flatten(const NodeGroupSchedulingMetadata& metadata, const std::vector<std::string>& driverOrder,
             const std::vector<std::string>& executorOrder, FlatSnapshot* out, std::string* err);
bool upload(gf_ctx* ctx, const FlatSnapshot& snap, std::string* err);

// Holds the context's sequence lock (gf_ctx_lock) for one "install a snapshot, then decide on it" sequence: Predicate and
// the UnschedulablePodMarker call from different threads (cmd/server.go:230) and must not see each other's snapshot.
struct CtxSequence {
    explicit CtxSequence(gf_ctx* c) : ctx(c) { gf_ctx_lock(ctx); }
    ~CtxSequence() { gf_ctx_unlock(ctx); }
    CtxSequence(const CtxSequence&) = delete;
    CtxSequence& operator=(const CtxSequence&) = delete;
    gf_ctx* ctx;
};

struct Binpacker {
    std::string Name;
    gf_algo Algo;
    bool IsSingleAz;
    gf_ctx* ctx;  // not owned
    bool with_efficiencies = true;  // fill PackingResult.PackingEfficiencies (one more device call per decision)

    PackingResult BinpackFunc(const Resources& driverResources, const Resources& executorResources, int executorCount,
                              const std::vector<std::string>& driverNodePriorityOrder,
                              const std::vector<std::string>& executorNodePriorityOrder,
                              const NodeGroupSchedulingMetadata& nodesSchedulingMetadata) const;
};

// Unknown names select distribute-evenly (internal/binpacker/binpack.go:52-58).
Binpacker SelectBinpacker(const std::string& name, gf_ctx* ctx);

}  // namespace gangfit::host
