#include "sparkpods.hpp"

#include <algorithm>

namespace gangfit::host {

namespace {
// strconv.ParseBool
bool parse_bool(const std::string& s, bool* out) {
    static const char* t[] = {"1", "t", "T", "TRUE", "true", "True"};
    static const char* f[] = {"0", "f", "F", "FALSE", "false", "False"};
    for (const char* v : t)
        if (s == v) return *out = true, true;
    for (const char* v : f)
        if (s == v) return *out = false, true;
    return false;
}
}  // namespace

std::optional<SparkApplicationResources> sparkResources(const Pod& pod, std::string* err) {
    using namespace common;
    auto fail = [&](const std::string& m) {
        if (err) *err = m;
        return std::nullopt;
    };
    std::map<std::string, Quantity> parsed;
    bool da = false;
    if (auto it = pod.Annotations.find(DynamicAllocationEnabled); it != pod.Annotations.end())
        if (!parse_bool(it->second, &da)) return fail("annotation DynamicAllocationEnabled could not be parsed as a boolean");
    for (const char* a : {DriverCPU, DriverMemory, DriverNvidiaGPUs, ExecutorCPU, ExecutorMemory, ExecutorNvidiaGPUs,
                          ExecutorCount, DAMinExecutorCount, DAMaxExecutorCount}) {
        const std::string key = a;
        auto it = pod.Annotations.find(key);
        if (it == pod.Annotations.end()) {
            if (key == DriverNvidiaGPUs || key == ExecutorNvidiaGPUs) continue;  // optional: a missing one is the zero Quantity
            if (!da && key == ExecutorCount)
                return fail("annotation ExecutorCount is required when DynamicAllocationEnabled is false");
            if (da && (key == DAMinExecutorCount || key == DAMaxExecutorCount))
                return fail("annotation " + key + " is required when DynamicAllocationEnabled is true");
            if (key == ExecutorCount || key == DAMinExecutorCount || key == DAMaxExecutorCount) continue;
            return fail("annotation " + key + " is missing from driver");
        }
        Quantity q;
        if (!Quantity::Parse(it->second, &q))
            return fail("annotation " + key + " does not have a parseable value " + it->second);
        parsed[key] = q;
    }
    SparkApplicationResources r;
    if (da) {
        r.MinExecutorCount = (int)parsed[DAMinExecutorCount].Value();
        r.MaxExecutorCount = (int)parsed[DAMaxExecutorCount].Value();
    } else {
        r.MinExecutorCount = r.MaxExecutorCount = (int)parsed[ExecutorCount].Value();
    }
    r.DriverResources = {parsed[DriverCPU], parsed[DriverMemory], parsed[DriverNvidiaGPUs]};
    r.ExecutorResources = {parsed[ExecutorCPU], parsed[ExecutorMemory], parsed[ExecutorNvidiaGPUs]};
    return r;
}

std::vector<const Pod*> filterToEarliestAndSort(const Pod& driver, const std::vector<Pod>& allDrivers) {
    std::vector<const Pod*> earlier;
    for (const Pod& p : allDrivers)
        if (p.NodeName.empty() && p.SchedulerName == driver.SchedulerName && p.InstanceGroup == driver.InstanceGroup &&
            p.CreationTimestampNanos < driver.CreationTimestampNanos && !p.Deleting)
            earlier.push_back(&p);
    // sort.Slice by creation time; equal timestamps are unordered in the reference, listing order is kept here
    std::stable_sort(earlier.begin(), earlier.end(),
                     [](const Pod* a, const Pod* b) { return a->CreationTimestampNanos < b->CreationTimestampNanos; });
    return earlier;
}

NodeGroupResources sparkResourceUsage(const Resources& driverResources, const Resources& executorResources,
                                      const std::string& driverNode, const std::vector<std::string>& executorNodes) {
    NodeGroupResources res;
    res[driverNode] = driverResources;
    for (const std::string& n : executorNodes) res[n] = executorResources;  // overwrites: multiplicity is lost
    return res;
}

}  // namespace gangfit::host
