// host_test.cpp — tests of the C++ host mirror, written after the reference's own Go tests so that they read alike:
//   TestSparkResources, TestIsEarliest                          internal/extender/sparkpods_test.go:40-223
//   TestResourcesSorting, TestScheduleContextSorting,
//   TestAZAwareNodeSorting(+IfZoneLabelIsMissing), TestLabelPrioritySorting   internal/sort/nodesorting_test.go:27-250
//   TestScheduler, TestUnschedulablePodMarker, TestSchedulerFailsToScheduleWhenNotEnoughNvidiaGPUs
//                                                               internal/extender/resource_test.go:27-71, unschedulablepods_test.go:24-80
//   findNodes (no reference test exists: checked against the literal loop of failover.go:412-436 on the host types)
// `host_test cpu` needs no GPU (parsing, sorting, snapshot, reservations); `host_test gpu` drives the device through
// the C ABI exactly like the Go shim would.  Exit code 0 = all passed.
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <set>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "extender.hpp"
#include "failover.hpp"

using namespace gangfit::host;

static int g_failed = 0, g_checked = 0;
#define CHECK(cond)                                                            \
    do {                                                                       \
        ++g_checked;                                                           \
        if (!(cond)) {                                                         \
            ++g_failed;                                                        \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);        \
        }                                                                      \
    } while (0)

static const int64_t Mi = 1024 * 1024, Gi = 1024 * Mi;

// ------------------------------------------------------------------------------------------------ quantity
static void TestParseQuantity() {
    struct C {
        const char* s;
        bool ok;
        int64_t value, milli;
    };
    const C cases[] = {
        {"0", true, 0, 0},          {"1", true, 1, 1000},          {"+1", true, 1, 1000},     {"-1", true, -1, -1000},
        {"500m", true, 1, 500},     {"1500m", true, 2, 1500},      {"0.5", true, 1, 500},     {"1.5", true, 2, 1500},
        {"100u", true, 1, 1},       {"1n", true, 1, 1},            {"1k", true, 1000, 1000000}, {"1Ki", true, 1024, 1024000},
        {"2432Mi", true, 2432 * Mi, 2432 * Mi * 1000}, {"6758Mi", true, 6758 * Mi, 6758 * Mi * 1000},
        {"8Gi", true, 8 * Gi, 8 * Gi * 1000},          {"1.5Gi", true, 3 * Gi / 2, 3 * Gi / 2 * 1000},
        {"1e3", true, 1000, 1000000}, {"1E3", true, 1000, 1000000}, {"12e-1", true, 2, 1200},  {"1.", true, 1, 1000},
        {"007", true, 7, 7000},     {".5", true, 1, 500},          {"1.G", true, 1000000000, 1000000000000},
        {"", false, 0, 0},          {"abc", false, 0, 0},          {"1x", false, 0, 0},       {"1 ", false, 0, 0},
        {"Gi", false, 0, 0},        {"1Gii", false, 0, 0},         {"1e", false, 0, 0},       {"--1", false, 0, 0},
        {"1mi", false, 0, 0},       {"1KI", false, 0, 0},
    };
    for (const C& c : cases) {
        Quantity q;
        const bool ok = Quantity::Parse(c.s, &q);
        CHECK(ok == c.ok);
        if (ok && c.ok) {
            if (q.Value() != c.value || q.MilliValue() != c.milli) std::printf("   case \"%s\": value %lld milli %lld\n", c.s, (long long)q.Value(), (long long)q.MilliValue());
            CHECK(q.Value() == c.value);
            CHECK(q.MilliValue() == c.milli);
        } else if (ok != c.ok) {
            std::printf("   case \"%s\"\n", c.s);
        }
    }
    // finer than nano rounds up to one nano (quantity.go:343-350), canonical forms are exact-or-nothing
    Quantity q;
    CHECK(Quantity::Parse("0.0000000001", &q) && q.nano() == 1);
    int64_t v = 0;
    CHECK(Quantity::Parse("250m", &q) && q.canonical_milli(&v) && v == 250 && !q.canonical_units(&v));
    CHECK(Quantity::Parse("1500u", &q) && !q.canonical_milli(&v));
    CHECK(Quantity::Parse("4Gi", &q) && q.canonical_units(&v) && v == 4 * Gi);
    Quantity a = Quantity::FromInt(3), b = Quantity::FromMilli(2500);
    CHECK(a.Cmp(b) == 1 && b.Cmp(a) == -1 && a.Cmp(a) == 0);
    a.Sub(b);
    CHECK(a.MilliValue() == 500);
    a.Neg();
    CHECK(a.MilliValue() == -500 && a.Value() == -1);
}

// ------------------------------------------------------------------------------------------------ sparkpods_test.go
static bool SameResources(const Resources& a, const Resources& b) { return a.Eq(b); }

static void TestSparkResources() {
    using namespace common;
    {  // "parses static allocation pod annotations into resources"
        Pod pod;
        pod.Annotations = {{DriverCPU, "1"},        {DriverMemory, "2432Mi"},   {DriverNvidiaGPUs, "1"}, {ExecutorCPU, "2"},
                           {ExecutorMemory, "6758Mi"}, {ExecutorNvidiaGPUs, "1"}, {ExecutorCount, "2"}};
        std::string err;
        auto r = sparkResources(pod, &err);
        CHECK(r.has_value());
        CHECK(SameResources(r->DriverResources, Resources::Create(1, 2432 * Mi, 1)));
        CHECK(SameResources(r->ExecutorResources, Resources::Create(2, 6758 * Mi, 1)));
        CHECK(r->MinExecutorCount == 2 && r->MaxExecutorCount == 2);
    }
    {  // "parses dynamic allocation pod annotations into resources"
        Pod pod;
        pod.Annotations = {{DriverCPU, "1"},          {DriverMemory, "2432Mi"},         {DriverNvidiaGPUs, "1"},
                           {ExecutorCPU, "2"},        {ExecutorMemory, "6758Mi"},       {ExecutorNvidiaGPUs, "1"},
                           {DynamicAllocationEnabled, "true"}, {DAMinExecutorCount, "2"}, {DAMaxExecutorCount, "5"}};
        auto r = sparkResources(pod, nullptr);
        CHECK(r.has_value());
        CHECK(SameResources(r->DriverResources, Resources::Create(1, 2432 * Mi, 1)));
        CHECK(SameResources(r->ExecutorResources, Resources::Create(2, 6758 * Mi, 1)));
        CHECK(r->MinExecutorCount == 2 && r->MaxExecutorCount == 5);
    }
    {  // "... when no gpu annotation is present"
        Pod pod;
        pod.Annotations = {{DriverCPU, "1"}, {DriverMemory, "2432Mi"}, {ExecutorCPU, "2"}, {ExecutorMemory, "6758Mi"},
                           {ExecutorCount, "2"}};
        auto r = sparkResources(pod, nullptr);
        CHECK(r.has_value());
        CHECK(SameResources(r->DriverResources, Resources::Create(1, 2432 * Mi, 0)));
        CHECK(SameResources(r->ExecutorResources, Resources::Create(2, 6758 * Mi, 0)));
    }
    // error branches (sparkpods.go:78-103)
    auto error_of = [](std::map<std::string, std::string> ann) {
        Pod pod;
        pod.Annotations = std::move(ann);
        std::string err;
        auto r = sparkResources(pod, &err);
        return r.has_value() ? std::string("<ok>") : err;
    };
    CHECK(error_of({{DriverCPU, "1"}, {DriverMemory, "1"}, {ExecutorCPU, "1"}, {ExecutorMemory, "1"}}) ==
          "annotation ExecutorCount is required when DynamicAllocationEnabled is false");
    CHECK(error_of({{DriverCPU, "1"}, {DriverMemory, "1"}, {ExecutorCPU, "1"}, {ExecutorMemory, "1"},
                    {DynamicAllocationEnabled, "true"}, {DAMinExecutorCount, "1"}}) ==
          "annotation spark-dynamic-allocation-max-executor-count is required when DynamicAllocationEnabled is true");
    CHECK(error_of({{DriverMemory, "1"}, {ExecutorCPU, "1"}, {ExecutorMemory, "1"}, {ExecutorCount, "1"}}) ==
          "annotation spark-driver-cpu is missing from driver");
    CHECK(error_of({{DriverCPU, "one"}, {DriverMemory, "1"}, {ExecutorCPU, "1"}, {ExecutorMemory, "1"}, {ExecutorCount, "1"}}) ==
          "annotation spark-driver-cpu does not have a parseable value one");
    CHECK(error_of({{DynamicAllocationEnabled, "maybe"}}) == "annotation DynamicAllocationEnabled could not be parsed as a boolean");
}

static Pod createPod(int64_t seconds, const char* uid, const char* instanceGroup) {
    Pod p;
    p.UID = uid;
    p.CreationTimestampNanos = seconds * 1000000000;
    p.InstanceGroup = instanceGroup;
    return p;
}

static void TestIsEarliest() {
    struct T {
        Pod pod;
        std::vector<Pod> pods;
        std::vector<std::string> result;
    };
    const char* g = "instance-group-foobar";
    std::vector<T> tests = {
        {createPod(100, "1", g), {createPod(101, "3", g), createPod(150, "2", g), createPod(100, "1", g)}, {}},
        {createPod(100, "1", g), {createPod(101, "2", g)}, {}},
        {createPod(100, "1", g), {createPod(101, "3", g), createPod(99, "2", g), createPod(100, "1", g)}, {"2"}},
        {createPod(100, "1", g), {createPod(99, "3", g), createPod(101, "2", g)}, {"3"}},
    };
    for (const T& t : tests) {
        std::vector<std::string> uids;
        for (const Pod* p : filterToEarliestAndSort(t.pod, t.pods)) uids.push_back(p->UID);
        CHECK(uids == t.result);
    }
    // other instance groups, assigned pods, other schedulers and pods being deleted are not predecessors; order is creation time
    Pod me = createPod(100, "me", g);
    me.SchedulerName = common::SparkSchedulerName;
    std::vector<Pod> all = {createPod(50, "other-group", "x"), createPod(60, "assigned", g), createPod(70, "b", g),
                            createPod(65, "a", g), createPod(66, "deleting", g), createPod(67, "default-scheduler", g)};
    for (Pod& p : all) p.SchedulerName = common::SparkSchedulerName;
    all[1].NodeName = "node1";
    all[4].Deleting = true;
    all[5].SchedulerName = "default-scheduler";
    std::vector<std::string> uids;
    for (const Pod* p : filterToEarliestAndSort(me, all)) uids.push_back(p->UID);
    CHECK((uids == std::vector<std::string>{"a", "b"}));
}

// ------------------------------------------------------------------------------------------------ nodesorting_test.go
static Resources CpuMem(int64_t cpu, int64_t mem) { return Resources::Create(cpu, mem, 0); }

static void TestResourcesSorting() {
    Resources node = CpuMem(1, 1), freeMemory = CpuMem(1, 2), freeCPU = CpuMem(2, 1);
    CHECK(!resourcesLessThan(freeMemory, node) && resourcesLessThan(node, freeMemory));
    CHECK(!resourcesLessThan(freeCPU, node) && resourcesLessThan(node, freeCPU));
}

static void TestScheduleContextSorting() {
    Resources less = CpuMem(1, 1), more = CpuMem(1, 2);
    ScheduleContext base1{0, less, "base1"}, base2{0, less, "base2"}, lowerAzPriority{1, less, "lower"},
        moreNodeResources{0, more, "more"};
    CHECK(!scheduleContextLessThan(lowerAzPriority, base1) && scheduleContextLessThan(base1, lowerAzPriority));
    CHECK(!scheduleContextLessThan(moreNodeResources, base1) && scheduleContextLessThan(base1, moreNodeResources));
    CHECK(!scheduleContextLessThan(base2, base1) && scheduleContextLessThan(base1, base2));
}

static NodeSchedulingMetadata Meta(int64_t cpu, int64_t mem, const char* zone, bool ready = false) {
    NodeSchedulingMetadata m;
    m.AvailableResources = CpuMem(cpu, mem);
    m.ZoneLabel = zone;
    m.Ready = ready;
    return m;
}

static void TestAZAwareNodeSorting() {
    NodeGroupSchedulingMetadata md;
    md["zone1Node1"] = Meta(1, 1, "zone1");
    md["zone1Node2"] = Meta(1, 2, "zone1");
    md["zone1Node3"] = Meta(2, 1, "zone1");
    md["zone2Node1"] = Meta(1, 1, "zone2");
    CHECK((getNodeNamesInPriorityOrder(md) == std::vector<std::string>{"zone2Node1", "zone1Node1", "zone1Node3", "zone1Node2"}));
}

static void TestAZAwareNodeSortingWorksIfZoneLabelIsMissing() {
    NodeGroupSchedulingMetadata md;
    md["node1"] = Meta(2, 1, "", true);
    md["node2"] = Meta(2, 2, "", true);
    md["node3"] = Meta(1, 1, "", true);
    CHECK((getNodeNamesInPriorityOrder(md) == std::vector<std::string>{"node3", "node1", "node2"}));
}

static void TestLabelPrioritySorting() {
    auto labelled = [](std::vector<std::pair<const char*, const char*>> v) {
        NodeGroupSchedulingMetadata md;
        for (auto& [node, value] : v) {
            NodeSchedulingMetadata m;
            if (value) m.AllLabels["test-label"] = value;
            md[node] = m;
        }
        return md;
    };
    struct T {
        LabelPriorityOrder order;
        NodeGroupSchedulingMetadata md;
        std::vector<std::string> nodeNames, expected;
    };
    std::vector<T> tests = {
        {{"test-label", {"best", "good"}}, labelled({{"node1", "worst"}, {"node2", "good"}, {"node3", "best"}}),
         {"node1", "node3", "node2"}, {"node3", "node2", "node1"}},
        {{"test-label", {"best", "good"}}, labelled({{"node1", nullptr}, {"node2", "good"}, {"node3", "best"}}),
         {"node2", "node3", "node1"}, {"node3", "node2", "node1"}},
        {{"test-label", {"best", "better", "good"}}, labelled({{"node1", "better"}, {"node2", "good"}, {"node3", "best"}}),
         {"node1", "node2", "node3"}, {"node3", "node1", "node2"}},
    };
    for (T& t : tests) {
        sortNodesByLabelPriority(t.nodeNames, t.md, t.order);
        CHECK(t.nodeNames == t.expected);
    }
}

static void TestPotentialNodes() {
    // nodesorting.go:41-64: drivers = requested names in priority order; executors = schedulable && ready nodes
    NodeGroupSchedulingMetadata md;
    md["a"] = Meta(4, 4, "z", true);
    md["b"] = Meta(1, 1, "z", true);
    md["c"] = Meta(2, 2, "z", false);  // not ready: driver candidate only
    md["d"] = Meta(3, 3, "z", true);
    md["d"].Unschedulable = true;
    md["a"].AllLabels["pool"] = "cheap";
    md["b"].AllLabels["pool"] = "costly";
    NodeSorter plain;
    auto [drivers, executors] = plain.PotentialNodes(md, {"a", "c", "d", "not-a-node"});
    CHECK((drivers == std::vector<std::string>{"c", "d", "a"}));
    CHECK((executors == std::vector<std::string>{"b", "a"}));
    NodeSorter labelled(std::nullopt, LabelPriorityOrder{"pool", {"cheap", "costly"}});
    auto [d2, e2] = labelled.PotentialNodes(md, {"a", "b"});
    CHECK((d2 == std::vector<std::string>{"b", "a"}));
    CHECK((e2 == std::vector<std::string>{"a", "b"}));
}

// ------------------------------------------------------------------------------------------------ snapshot & reservations
static Node NewNode(const char* name, const char* zone) {  // extendertest.NewNode (extender_test_utils.go:239-271)
    Node n;
    n.Name = name;
    n.labels = {{"resource_channel", "batch-medium-priority"},
                {"com.palantir.rubix/instance-group", "batch-medium-priority"},
                {"test", "something"},
                {"topology.kubernetes.io/zone", zone}};  // NOT the label the bin-pack snapshot reads
    n.Allocatable = {{kResourceCPU, Quantity::FromInt(8)}, {kResourceMemory, Quantity::FromInt(8 * Gi)},
                     {kResourceNvidiaGPU, Quantity::FromInt(1)}};
    n.Ready = true;
    return n;
}

static Pod Driver(const char* app, std::map<std::string, std::string> annotations, int64_t created_s = 0) {
    Pod p;
    p.Name = std::string(app) + "-spark-driver";
    p.Namespace = "namespace";
    p.labels = {{common::SparkRoleLabel, common::Driver}, {common::SparkAppIDLabel, app}};
    p.Annotations = std::move(annotations);
    p.SchedulerName = common::SparkSchedulerName;
    p.InstanceGroup = "batch-medium-priority";
    p.CreationTimestampNanos = created_s * 1000000000;
    return p;
}
static std::map<std::string, std::string> StaticAnnotations(int numExecutors, const char* driverMem = "1",
                                                            const char* driverCPU = "1", const char* executorMem = "1",
                                                            const char* executorCPU = "1", bool executorGpu = false) {
    std::map<std::string, std::string> a = {{"spark-driver-cpu", driverCPU},     {"spark-driver-mem", driverMem},
                                            {"spark-driver-nvidia.com/gpu", "1"}, {"spark-executor-cpu", executorCPU},
                                            {"spark-executor-mem", executorMem},
                                            {"spark-executor-count", std::to_string(numExecutors)}};
    if (executorGpu) a["spark-executor-nvidia.com/gpu"] = "1";
    return a;
}

static void TestSnapshotAndReservations() {
    std::vector<Node> nodes = {NewNode("node1", "zone1"), NewNode("node2", "zone1")};
    nodes[1].labels[kLabelZoneFailureDomain] = "us-east-1a";
    Pod driver = Driver("app", StaticAnnotations(3));
    auto res = sparkResources(driver, nullptr);
    ResourceReservation rr = newResourceReservation("node1", {"node1", "node2", "node1"}, driver, res->DriverResources,
                                                    res->ExecutorResources);
    CHECK(rr.Name == "app" && rr.Namespace == "namespace" && rr.Pods.at("driver") == "app-spark-driver");
    CHECK(rr.Reservations.size() == 4 && rr.Reservations.at("driver").Node == "node1");
    CHECK(rr.Reservations.at("executor-1").Node == "node1" && rr.Reservations.at("executor-2").Node == "node2" &&
          rr.Reservations.at("executor-3").Node == "node1");  // names follow ExecutorNodes order (:501-502)
    CHECK(rr.Reservations.at("driver").Resources.at(kResourceNvidiaGPU).Value() == 1);
    // UsageForNodes: every reservation counts in full
    NodeGroupResources usage = UsageForNodes({rr});
    CHECK(usage.at("node1").CPU.Value() == 3 && usage.at("node2").CPU.Value() == 1 && usage.at("node1").NvidiaGPU.Value() == 1);
    NodeGroupResources overhead;
    overhead["node1"] = Resources{Quantity::FromMilli(500), Quantity::FromInt(Gi), Quantity()};
    NodeGroupSchedulingMetadata md = NodeSchedulingMetadataForNodes(nodes, usage, overhead);
    CHECK(md.at("node1").AvailableResources.CPU.MilliValue() == 8000 - 3000 - 500);
    CHECK(md.at("node1").SchedulableResources.CPU.MilliValue() == 7500);
    CHECK(md.at("node1").AvailableResources.Memory.Value() == 8 * Gi - 3 - Gi);
    CHECK(md.at("node2").AvailableResources.CPU.MilliValue() == 7000 && md.at("node2").SchedulableResources.Memory.Value() == 8 * Gi);
    CHECK(md.at("node1").ZoneLabel == "default" && md.at("node2").ZoneLabel == "us-east-1a");
    CHECK(md.at("node1").Ready && !md.at("node1").Unschedulable);
    CHECK(usage.at("node1").CPU.MilliValue() == 3500);  // the in-place Add of the reference (quirk 5)
    // sparkResourceUsage: multiplicity lost, the driver entry overwritten by an executor on the same node
    NodeGroupResources u = sparkResourceUsage(Resources::Create(2, 2, 0), Resources::Create(3, 3, 0), "n1", {"n1", "n1", "n2"});
    CHECK(u.size() == 2 && u.at("n1").CPU.Value() == 3 && u.at("n2").CPU.Value() == 3);
    md.SubtractUsageIfExists({{"node2", Resources::Create(1, 1, 0)}, {"ghost", Resources::Create(1, 1, 0)}});
    CHECK(md.at("node2").AvailableResources.CPU.MilliValue() == 6000 && md.count("ghost") == 0);
}

// ------------------------------------------------------------------------------------------------ device-backed tests
static gf_ctx* g_ctx = nullptr;

static SparkSchedulerExtender NewTestExtender(const char* binpackAlgo, std::vector<Node> nodes, bool fifo = true) {
    SparkSchedulerExtender e(SelectBinpacker(binpackAlgo, g_ctx), NodeSorter(), fifo, FifoConfig{});
    e.nodes = std::move(nodes);
    e.nowNanos = 1000ll * 1000000000;
    return e;
}

static void TestScheduler() {  // resource_test.go:27-71 (the driver half; executors bind to the reservation in the Go host)
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone1");
    auto ext = NewTestExtender("single-az-tightly-pack", {node1, node2});
    Pod driver = Driver("2-executor-app", StaticAnnotations(2));
    SelectNodeResult r = ext.selectDriverNode("batch-medium-priority", driver, {"node1", "node2"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::success) && r.node == "node1");
    CHECK(r.created.has_value());
    if (r.created) {
        CHECK(r.created->Reservations.at("driver").Node == "node1");
        CHECK(r.created->Reservations.at("executor-1").Node == "node1" && r.created->Reservations.at("executor-2").Node == "node1");
        CHECK(r.created->Reservations.size() == 3);
        ext.reservations.push_back(*r.created);
    }
    // a second Filter for the same driver returns the reserved node (resource.go:278-291)
    r = ext.selectDriverNode("batch-medium-priority", driver, {"node2"}, ext.nodes);
    CHECK(r.outcome == std::string(outcome::success) && r.node == "node1" && !r.created.has_value());
    // with the first app's reservations in place a 13-executor app does not fit (16 cpu - 3 = 13 < 14)
    Pod big = Driver("big-app", StaticAnnotations(13), 10);
    r = ext.selectDriverNode("batch-medium-priority", big, {"node1", "node2"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::failureFit) && r.node.empty());
}

static void TestUnschedulablePodMarker() {  // unschedulablepods_test.go:24-53
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone1");
    auto ext = NewTestExtender("single-az-tightly-pack", {node1, node2});
    bool served = false;
    std::string err;
    CHECK(!ext.DoesPodExceedClusterCapacity(Driver("2-executor-app", StaticAnnotations(2)), ext.nodes, {}, &served, &err) && served);
    CHECK(ext.DoesPodExceedClusterCapacity(Driver("100-executor-app", StaticAnnotations(100)), ext.nodes, {}, &served, &err) && served);
}

static void TestUnschedulablePodScanBatched() {  // scanForUnschedulablePods (unschedulablepods.go:93-129) in one launch
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone1");
    auto ext = NewTestExtender("single-az-tightly-pack", {node1, node2});
    ext.nowNanos = 10000ll * 1000000000;
    Pod fits = Driver("2-executor-app", StaticAnnotations(2), 1), too_big = Driver("100-executor-app", StaticAnnotations(100), 2),
        young = Driver("young-app", StaticAnnotations(100), 9999), gpus = Driver("gpu-app", StaticAnnotations(2, "1", "1", "1", "1", true), 3);
    Pod bound = Driver("bound-app", StaticAnnotations(100), 4);
    bound.NodeName = "node1";
    Pod executor = Driver("exec", StaticAnnotations(100), 5);
    executor.labels[common::SparkRoleLabel] = common::Executor;
    bool served = false;
    std::string err;
    auto r = ext.scanForUnschedulablePods({fits, too_big, young, gpus, bound, executor}, 600ll * 1000000000, ext.nodes, {}, &served, &err);
    CHECK(served && r.size() == 3);
    if (r.size() == 3) {
        CHECK(r[0].first == "2-executor-app-spark-driver" && !r[0].second);
        CHECK(r[1].first == "100-executor-app-spark-driver" && r[1].second);
        CHECK(r[2].first == "gpu-app-spark-driver" && r[2].second);
    }
}

static void TestSchedulerFailsToScheduleWhenNotEnoughNvidiaGPUs() {  // unschedulablepods_test.go:55-80
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone1");
    auto ext = NewTestExtender("single-az-tightly-pack", {node1, node2});
    bool served = false;
    CHECK(ext.DoesPodExceedClusterCapacity(Driver("gpu-app", StaticAnnotations(2, "1", "1", "1", "1", true)), ext.nodes, {}, &served, nullptr) && served);
    SelectNodeResult r = ext.selectDriverNode("batch-medium-priority", Driver("gpu-app", StaticAnnotations(2, "1", "1", "1", "1", true)),
                                              {"node1", "node2"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::failureFit));
}

static void TestFifoAndBinpackers() {
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone1");
    // an earlier driver that can never fit blocks the queue (resource.go:244-253) ...
    auto ext = NewTestExtender("tightly-pack", {node1, node2});
    Pod hog = Driver("hog", StaticAnnotations(100), 1), small = Driver("small", StaticAnnotations(1), 5);
    ext.pods = {hog, small};
    SelectNodeResult r = ext.selectDriverNode("batch-medium-priority", small, {"node1", "node2"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::failureEarlierDriver));
    // ... unless it is still younger than the enforce-after age (shouldSkipDriverFifo, :264-270)
    FifoConfig cfg;
    cfg.EnforceAfterPodAgeByInstanceGroup["batch-medium-priority"] = 3600ll * 1000000000;
    SparkSchedulerExtender young(SelectBinpacker("tightly-pack", g_ctx), NodeSorter(), true, cfg);
    young.nodes = ext.nodes;
    young.pods = ext.pods;
    young.nowNanos = 1000ll * 1000000000;
    r = young.selectDriverNode("batch-medium-priority", small, {"node1", "node2"}, young.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::success));
    // an earlier driver that fits takes its share first: 2 x (8 cpu): earlier app 1 + 6, then 1 + 8 no longer fits
    auto ext2 = NewTestExtender("tightly-pack", {node1, node2});
    Pod first = Driver("first", StaticAnnotations(6), 1), second = Driver("second", StaticAnnotations(8), 2);
    ext2.pods = {first, second};
    r = ext2.selectDriverNode("batch-medium-priority", second, {"node1", "node2"}, ext2.nodes);
    // quirk 1: the replay subtracts ONE executor per distinct node (and no driver: it shares node1 with executors), so
    // node1 8-1=7 and node2 8 remain -> 1 + 8 fits where a full accounting would not
    CHECK(r.served && r.outcome == std::string(outcome::success));
    // registry names (internal/binpacker/binpack.go:43-58); unknown names select distribute-evenly
    CHECK(SelectBinpacker("nope", g_ctx).Name == "distribute-evenly" && !SelectBinpacker("nope", g_ctx).IsSingleAz);
    CHECK(SelectBinpacker("single-az-minimal-fragmentation", g_ctx).IsSingleAz);
    // the SparkBinPackFunction shape with string names: distribute-evenly round-robins
    NodeGroupResources none;
    NodeGroupSchedulingMetadata md = NodeSchedulingMetadataForNodes({node1, node2}, none, {});
    Binpacker even = SelectBinpacker("distribute-evenly", g_ctx);
    PackingResult p = even.BinpackFunc(Resources::Create(1, 1, 1), Resources::Create(1, 1, 0), 3, {"node1", "node2"},
                                       {"node1", "node2", "ghost"}, md);
    CHECK(p.served && p.HasCapacity && p.DriverNode == "node1");
    CHECK((p.ExecutorNodes == std::vector<std::string>{"node1", "node2", "node1"}));
    CHECK(p.PackingEfficiencies.size() == 2 && p.PackingEfficiencies.at("node1").CPU == 3.0 / 8.0 &&
          p.PackingEfficiencies.at("node2").CPU == 1.0 / 8.0 && p.PackingEfficiencies.at("node1").GPU == 1.0);
    // a quantity that is not exactly representable is refused, never rounded
    Resources odd = Resources::Create(1, 1, 0);
    Quantity::Parse("1500u", &odd.CPU);
    p = even.BinpackFunc(odd, Resources::Create(1, 1, 0), 1, {"node1"}, {"node1"}, md);
    CHECK(!p.served && !p.HasCapacity);
    for (const char* name : {"tightly-pack", "az-aware-tightly-pack", "single-az-tightly-pack", "single-az-minimal-fragmentation"}) {
        p = SelectBinpacker(name, g_ctx).BinpackFunc(Resources::Create(1, 1, 1), Resources::Create(1, 1, 0), 2,
                                                     {"node1", "node2"}, {"node1", "node2"}, md);
        CHECK(p.served && p.HasCapacity && p.DriverNode == "node1" && (p.ExecutorNodes == std::vector<std::string>{"node1", "node1"}));
    }
}

static std::map<std::string, std::string> DynamicAnnotations(int minExecutors, int maxExecutors, const char* driverMem,
                                                             const char* driverCPU, const char* executorMem,
                                                             const char* executorCPU) {
    return {{"spark-driver-cpu", driverCPU},
            {"spark-driver-mem", driverMem},
            {"spark-driver-nvidia.com/gpu", "1"},
            {"spark-executor-cpu", executorCPU},
            {"spark-executor-mem", executorMem},
            {"spark-dynamic-allocation-enabled", "true"},
            {"spark-dynamic-allocation-min-executor-count", std::to_string(minExecutors)},
            {"spark-dynamic-allocation-max-executor-count", std::to_string(maxExecutors)}};
}

static void TestMinimalFragmentationEdgeCase() {  // resource_test.go:127-170
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone1");
    auto ext = NewTestExtender("single-az-minimal-fragmentation", {node1, node2});
    Pod staticDriver = Driver("static-app", StaticAnnotations(0, "4", "1", "1", "1"));
    Pod dynDriver = Driver("dyn-app", DynamicAnnotations(0, 1, "1", "4", "1", "3"), 1);
    // "schedule a driver on each node"
    SelectNodeResult r = ext.selectDriverNode("batch-medium-priority", staticDriver, {"node1"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::success) && r.node == "node1");
    if (r.created) ext.reservations.push_back(*r.created);
    r = ext.selectDriverNode("batch-medium-priority", dynDriver, {"node2"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::success) && r.node == "node2");
    if (r.created) ext.reservations.push_back(*r.created);
    // "This pod should be scheduled on node2 has it has the smallest capacity"
    r = ext.rescheduleExecutor(dynDriver, {"node1", "node2"}, ext.nodes, {}, true);
    CHECK(r.served && r.outcome == std::string(outcome::successScheduledExtraExecutor) && r.node == "node2");
    // the first-fit packers take the first node of the order instead (node1: less free memory)
    auto tight = NewTestExtender("single-az-tightly-pack", {node1, node2});
    tight.reservations = ext.reservations;
    r = tight.rescheduleExecutor(dynDriver, {"node1", "node2"}, tight.nodes, {}, false);
    CHECK(r.served && r.outcome == std::string(outcome::successRescheduled) && r.node == "node1");
    // an executor already running on node1 attracts the next one (resource_test.go:73-125)
    r = ext.rescheduleExecutor(dynDriver, {"node1", "node2"}, ext.nodes, {"node1"}, true);
    CHECK(r.served && r.node == "node1");
    // quirk 5: with overhead on a node that also carries reservations the first-fit loop sees the overhead twice
    tight.overhead["node1"] = Resources{Quantity::FromInt(2), Quantity(), Quantity()};
    // node1: 8 - 1 (static driver) - 2 - 2 = 3 cpu < 3? no: exactly 3 -> still fits; one more milli-core of overhead does not
    r = tight.rescheduleExecutor(dynDriver, {"node1", "node2"}, tight.nodes, {}, false);
    CHECK(r.served && r.node == "node1");
    tight.overhead["node1"] = Resources{Quantity::FromMilli(2001), Quantity(), Quantity()};
    r = tight.rescheduleExecutor(dynDriver, {"node1", "node2"}, tight.nodes, {}, false);
    CHECK(r.served && r.node == "node2");  // 8 - 1 - 2*2.001 < 3 although allocatable - usage - overhead = 4.999 >= 3
}

// TestDynamicAllocationScheduling, case "schedules an executor only in the same AZ as the original application"
// (resource_test.go:262-292), through the mirror's whole rescheduleExecutor: NO hand-filtered order — the zone step
// (getCommonZoneForExecutorsApplication + filterNodesToZone, resource.go:493-553, :606-632) narrows the candidates itself.
static void TestDynamicAllocationSameAZ() {
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone2");
    auto ext = NewTestExtender("single-az-tightly-pack", {node1, node2});
    ext.shouldScheduleDynamicallyAllocatedExecutorsInSameAZ = true;  // extendertest/extender_test_utils.go:135
    auto Executor = [](const char* app, int i) {
        Pod p;
        p.Name = std::string(app) + "-spark-exec-" + std::to_string(i);
        p.Namespace = "namespace";
        p.labels = {{common::SparkRoleLabel, common::Executor}, {common::SparkAppIDLabel, app}};
        p.SchedulerName = common::SparkSchedulerName;
        p.Phase = "Pending";
        return p;
    };
    Pod staticDriver = Driver("static-allocation-app", StaticAnnotations(1));
    Pod dynDriver = Driver("dynamic-allocation-app", DynamicAnnotations(0, 2, "1", "1", "1", "1"), 1);
    std::vector<Pod> all = {staticDriver, Executor("static-allocation-app", 0), dynDriver, Executor("dynamic-allocation-app", 0),
                            Executor("dynamic-allocation-app", 1)};
    for (Pod& p : all) p.Phase = "Pending";
    auto bind = [&](size_t i, const std::string& node) {  // what Harness.Schedule does on success (extender_test_utils.go:180-190)
        all[i].NodeName = node;
        all[i].Phase = "Running";
    };
    // "We first schedule a statically allocated application to zone1 to make it more desirable as there is less space"
    SelectNodeResult r = ext.selectDriverNode("batch-medium-priority", all[0], {"node1"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::success) && r.node == "node1" && r.created.has_value());
    if (r.created) ext.reservations.push_back(*r.created);
    bind(0, "node1");
    bind(1, "node1");  // the static executor binds to its reservation on node1 (resource.go:382-424: Go-host bookkeeping)
    r = ext.selectDriverNode("batch-medium-priority", all[2], {"node2"}, ext.nodes);
    CHECK(r.served && r.outcome == std::string(outcome::success) && r.node == "node2" && r.created.has_value());
    if (r.created) ext.reservations.push_back(*r.created);
    bind(2, "node2");
    // Before any soft reservation exists node1 sorts first (less free memory: 8 GiB - 2 against 8 GiB - 1).  The same request
    // with the flag off, or with a packer that is not single-AZ, is scheduled anywhere — node1:
    ext.shouldScheduleDynamicallyAllocatedExecutorsInSameAZ = false;
    r = ext.rescheduleExecutor(all[3], all[2], all, {"node1", "node2"}, {}, true);
    CHECK(r.served && r.node == "node1");
    ext.shouldScheduleDynamicallyAllocatedExecutorsInSameAZ = true;
    auto plain = NewTestExtender("tightly-pack", {node1, node2});
    plain.shouldScheduleDynamicallyAllocatedExecutorsInSameAZ = true;
    plain.reservations = ext.reservations;
    r = plain.rescheduleExecutor(all[3], all[2], all, {"node1", "node2"}, {}, true);
    CHECK(r.served && r.node == "node1");
    // ... and so is an application whose running pods span two zones (:628-630)
    std::vector<Pod> spread = all;
    spread[3].NodeName = "node1";
    spread[3].Phase = "Running";
    r = ext.rescheduleExecutor(all[4], all[2], spread, {"node1", "node2"}, {}, true);
    CHECK(r.served && r.node == "node1");
    // executor-0 and executor-1 of the dynamic application, both nodes offered: the application lives in zone2 — "node2" both
    // times (expectedPodToNodeSoftReservationsMap, :289-292)
    for (size_t i : {size_t(3), size_t(4)}) {
        r = ext.rescheduleExecutor(all[i], all[2], all, {"node1", "node2"}, {}, true);
        CHECK(r.served && r.outcome == std::string(outcome::successScheduledExtraExecutor) && r.node == "node2");
        bind(i, r.node);
        ext.softReservationUsage["node2"].Add(Resources::Create(1, 1, 0));  // the soft reservation AddReservationForPod records
    }
    // no running pod: the reference's error, no outcome (:514-516, :611-613)
    std::vector<Pod> none = all;
    for (Pod& p : none) p.Phase = "Pending";
    r = ext.rescheduleExecutor(all[4], all[2], none, {"node1", "node2"}, {}, true);
    CHECK(r.node.empty() && r.outcome.empty() &&
          r.error == "Application has no scheduled pods, can't make scheduling decisions based on AZ");
    // an executor without the app id label (:494-497)
    Pod bare = all[4];
    bare.labels.erase(common::SparkAppIDLabel);
    r = ext.rescheduleExecutor(bare, all[2], all, {"node1", "node2"}, {}, true);
    CHECK(r.node.empty() && r.error == "Executor does not have a Spark app id label, could not create label selector");
    // a candidate node without the topology label: failure-internal from filterNodesToZone (:466-468, :620-623)
    Node bareNode = NewNode("node3", "zone2");
    bareNode.labels.erase(kLabelTopologyZone);
    ext.nodes.push_back(bareNode);
    r = ext.rescheduleExecutor(all[4], all[2], all, {"node1", "node2", "node3"}, {}, true);
    CHECK(r.node.empty() && r.outcome == std::string(outcome::failureInternal) &&
          r.error == "Could not read zone label from node, unable to make scheduling decisions based on AZ");
    ext.nodes.pop_back();
    // the zone is full: failure-fit (the reference then creates a demand for that zone, :664-668 — Go-host bookkeeping)
    ext.softReservationUsage["node2"].Add(Resources::Create(8, 1, 0));
    r = ext.rescheduleExecutor(all[4], all[2], all, {"node1", "node2"}, {}, true);
    CHECK(r.served && r.node.empty() && r.outcome == std::string(outcome::failureFit));
}

// gf_snapshot_build (reservation replay + metadata + priority orders on the device) against the string-keyed host mirror
// of the same reference functions (UsageForNodes, NodeSchedulingMetadataForNodes, NodeSorter.PotentialNodes), which the
// reference's sort tests pin.
static void TestDeviceSnapshotBuildAgainstHostMirror() {
    const int n = 300;
    uint64_t rng = 0x5EED;
    auto next = [&]() {
        rng += 0x9E3779B97F4A7C15ull;
        uint64_t z = rng;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    const char* zones[] = {"az-a", "az-b", "az-c"};  // ids in label order
    std::vector<Node> nodes;
    std::vector<std::string> requested;
    LabelPriorityOrder pool{"pool", {"spot", "on-demand"}};
    for (int i = 0; i < n; ++i) {
        Node nd;
        nd.Name = "node-" + std::to_string(next() % 100000) + "-" + std::to_string(i);
        nd.labels[kLabelZoneFailureDomain] = zones[next() % 3];
        const uint64_t p = next() % 3;
        if (p < 2) nd.labels["pool"] = p == 0 ? "spot" : "on-demand";
        nd.Allocatable = {{kResourceCPU, Quantity::FromInt(8 + 8 * (int64_t)(next() % 3))},
                          {kResourceMemory, Quantity::FromInt((int64_t)(16 + 16 * (next() % 4)) * Gi)},
                          {kResourceNvidiaGPU, Quantity::FromInt(next() % 10 == 0 ? 4 : 0)}};
        nd.Unschedulable = next() % 20 == 0;
        nd.Ready = next() % 20 != 0;
        if (next() % 5 != 0) requested.push_back(nd.Name);
        nodes.push_back(nd);
    }
    std::vector<ResourceReservation> rrs;
    for (int r = 0; r < 120; ++r) {
        ResourceReservation rr;
        rr.Name = "app-" + std::to_string(r);
        const int k = 1 + (int)(next() % 9);
        for (int e = 0; e <= k; ++e) {
            Reservation res;
            res.Node = nodes[next() % n].Name;
            res.Resources = {{kResourceCPU, Quantity::FromMilli(500 * (int64_t)(1 + next() % 8))},
                             {kResourceMemory, Quantity::FromInt((int64_t)(1 + next() % 8) * Gi)},
                             {kResourceNvidiaGPU, Quantity::FromInt(next() % 30 == 0 ? 1 : 0)}};
            rr.Reservations[e == 0 ? "driver" : executorReservationName(e - 1)] = res;
        }
        rrs.push_back(rr);
    }
    NodeGroupResources overhead;
    for (int i = 0; i < n; i += 3) overhead[nodes[i].Name] = Resources{Quantity::FromMilli(250), Quantity::FromInt(Gi / 2), Quantity()};
    // ---- host mirror
    NodeGroupResources usage = UsageForNodes(rrs);
    NodeGroupSchedulingMetadata md = NodeSchedulingMetadataForNodes(nodes, usage, overhead);
    NodeSorter sorter(std::nullopt, pool);
    auto [wantD, wantX] = sorter.PotentialNodes(md, requested);
    // ---- flat columns for the device
    std::vector<std::string> sorted_names;
    for (const Node& nd : nodes) sorted_names.push_back(nd.Name);
    std::sort(sorted_names.begin(), sorted_names.end());
    std::map<std::string, uint32_t> index, rank;
    for (int i = 0; i < n; ++i) index[nodes[i].Name] = (uint32_t)i;
    for (int i = 0; i < n; ++i) rank[sorted_names[i]] = (uint32_t)i;
    std::set<std::string> req(requested.begin(), requested.end());
    std::vector<int64_t> alloc[3], over[3], rreq[3];
    std::vector<uint32_t> flags, zone, name_rank, exec_label, rnode;
    for (const Node& nd : nodes) {
        Resources a{nd.Allocatable.at(kResourceCPU), nd.Allocatable.at(kResourceMemory), nd.Allocatable.at(kResourceNvidiaGPU)};
        int64_t v[3], o[3] = {0, 0, 0};
        a.canonical(v);
        if (overhead.count(nd.Name)) overhead.at(nd.Name).canonical(o);
        for (int j = 0; j < 3; ++j) {
            alloc[j].push_back(v[j]);
            over[j].push_back(o[j]);
        }
        flags.push_back((nd.Unschedulable ? GF_NODE_UNSCHEDULABLE : 0u) | (nd.Ready ? GF_NODE_READY : 0u) |
                        (req.count(nd.Name) ? GF_NODE_DRIVER_CANDIDATE : 0u));
        const std::string& z = nd.labels.at(kLabelZoneFailureDomain);
        zone.push_back(z == "az-a" ? 0u : (z == "az-b" ? 1u : 2u));
        name_rank.push_back(rank.at(nd.Name));
        auto l = nd.labels.find("pool");
        exec_label.push_back(l == nd.labels.end() ? 0xFFFFFFFFu : (l->second == "spot" ? 0u : 1u));
    }
    for (const auto& rr : rrs)
        for (const auto& [name, res] : rr.Reservations) {
            rnode.push_back(index.at(res.Node));
            Resources r{res.Resources.at(kResourceCPU), res.Resources.at(kResourceMemory), res.Resources.at(kResourceNvidiaGPU)};
            int64_t v[3];
            r.canonical(v);
            for (int j = 0; j < 3; ++j) rreq[j].push_back(v[j]);
        }
    std::vector<uint32_t> D(n), X(n);
    uint32_t nd = 0, nx = 0;
    const int rc = gf_snapshot_build(g_ctx, n, alloc[0].data(), alloc[1].data(), alloc[2].data(), over[0].data(), over[1].data(),
                                     over[2].data(), (uint32_t)rnode.size(), rnode.data(), rreq[0].data(), rreq[1].data(),
                                     rreq[2].data(), flags.data(), zone.data(), 3, name_rank.data(), nullptr, exec_label.data(),
                                     D.data(), &nd, X.data(), &nx);
    CHECK(rc == GF_OK);
    if (rc != GF_OK) {
        std::printf("   %s\n", gf_last_error(g_ctx));
        return;
    }
    std::vector<std::string> gotD, gotX;
    for (uint32_t i = 0; i < nd; ++i) gotD.push_back(nodes[D[i]].Name);
    for (uint32_t i = 0; i < nx; ++i) gotX.push_back(nodes[X[i]].Name);
    CHECK(gotD == wantD);
    CHECK(gotX == wantX);
    std::vector<int64_t> avail(3 * n), sched(3 * n);
    CHECK(gf_snapshot_get(g_ctx, avail.data(), sched.data()) == GF_OK);
    bool same = true;
    for (int i = 0; i < n; ++i) {
        int64_t a[3], s2[3];
        md.at(nodes[i].Name).AvailableResources.canonical(a);
        md.at(nodes[i].Name).SchedulableResources.canonical(s2);
        for (int j = 0; j < 3; ++j) same = same && avail[3 * i + j] == a[j] && sched[3 * i + j] == s2[j];
    }
    CHECK(same);
    // and a decision on the device-built snapshot equals the one through the string interface
    Binpacker bp = SelectBinpacker("single-az-tightly-pack", g_ctx);
    gf_app app{};
    Resources::Create(1, 2 * Gi, 0).canonical(app.drv);
    Resources::Create(2, 4 * Gi, 0).canonical(app.exe);
    app.k = 40;
    gf_result res{};
    std::vector<uint32_t> exec(41);
    CHECK(gf_spark_binpack(g_ctx, bp.Algo, &app, &res, exec.data(), 40) == GF_OK);
    PackingResult want = bp.BinpackFunc(Resources::Create(1, 2 * Gi, 0), Resources::Create(2, 4 * Gi, 0), 40, wantD, wantX, md);
    CHECK(want.served && want.HasCapacity == (res.has_capacity != 0));
    if (want.HasCapacity && res.has_capacity) {
        CHECK(nodes[res.driver_node].Name == want.DriverNode);
        bool same_exec = true;
        for (int i = 0; i < 40; ++i) same_exec = same_exec && nodes[exec[i]].Name == want.ExecutorNodes[i];
        CHECK(same_exec);
    }
}


// Consecutive Filters on an unchanged cluster (selectDriverNodeFlat with the host's flattened reservations): the snapshot is
// not rebuilt and the FIFO chain resumes from the previous chain's checkpoints (include/gangfit.h, "Incremental FIFO chains").
// Every answer must be the one the string-keyed route gives — which installs its own snapshot per call and replays the whole
// chain, like the reference (resource.go:309-328).
static void TestIncrementalFilters() {
    const int n = 500, n_pending = 120;
    uint64_t rng = 0xF1F0;
    auto next = [&]() {
        rng += 0x9E3779B97F4A7C15ull;
        uint64_t z = rng;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (const char* packer : {"tightly-pack", "single-az-tightly-pack", "single-az-minimal-fragmentation"}) {
        SparkSchedulerExtender ext(SelectBinpacker(packer, g_ctx), NodeSorter(), true, FifoConfig{});
        const char* zones[] = {"az-a", "az-b", "az-c"};
        std::vector<std::string> names;
        for (int i = 0; i < n; ++i) {
            Node nd;
            nd.Name = "n" + std::to_string(next() % 100000) + "-" + std::to_string(i);
            nd.labels[kLabelZoneFailureDomain] = zones[next() % 3];
            nd.Allocatable = {{kResourceCPU, Quantity::FromInt(16 + 16 * (int64_t)(next() % 3))},
                              {kResourceMemory, Quantity::FromInt((int64_t)(64 + 64 * (next() % 3)) * Gi)},
                              {kResourceNvidiaGPU, Quantity::FromInt(0)}};
            nd.Ready = true;
            names.push_back(nd.Name);
            ext.nodes.push_back(nd);
        }
        for (int r = 0; r < 60; ++r) {
            ResourceReservation rr;
            rr.Name = "running-" + std::to_string(r);
            rr.Namespace = "namespace";
            const int k = 1 + (int)(next() % 12);
            for (int e = 0; e <= k; ++e) {
                Reservation res;
                res.Node = ext.nodes[next() % n].Name;
                res.Resources = {{kResourceCPU, Quantity::FromInt(1 + (int64_t)(next() % 2))},
                                 {kResourceMemory, Quantity::FromInt((int64_t)(2 + next() % 6) * Gi)},
                                 {kResourceNvidiaGPU, Quantity::FromInt(0)}};
                rr.Reservations[e == 0 ? "driver" : executorReservationName(e - 1)] = res;
            }
            ext.reservations.push_back(rr);
        }
        const char* ecpu[] = {"1", "2", "4"};
        const char* emem[] = {"4Gi", "8Gi", "16Gi"};
        for (int p = 0; p < n_pending; ++p) {
            const int k = 1 + (int)(next() % 30);
            const char* em = emem[next() % 3];
            const char* ec = ecpu[next() % 3];
            Pod pod = Driver(("pending-" + std::to_string(p)).c_str(), StaticAnnotations(k, "2Gi", "1", em, ec), p + 1);
            pod.Annotations.erase("spark-driver-nvidia.com/gpu");
            pod.UID = "uid-" + std::to_string(p);
            pod.ResourceVersion = 100 + (uint64_t)p;  // the flat route parses each (UID, version) once
            pod.Annotations.erase("spark-executor-nvidia.com/gpu");
            ext.pods.push_back(pod);
        }
        ext.nowNanos = (int64_t)(n_pending + 10) * 1000000000;
        FlatCluster cluster;
        FlatReservations flat;
        std::string err;
        CHECK(FlatCluster::Build(ext.nodes, &cluster, &err));
        CHECK(FlatReservations::Build(ext.reservations, ext.softReservationUsage, cluster, &flat, &err));
        std::vector<SelectNodeResult> want;
        for (int j = 60; j < n_pending; ++j) want.push_back(ext.selectDriverNode("batch-medium-priority", ext.pods[(size_t)j], names, ext.nodes));
        uint64_t st0[4], st1[4];
        gf_chain_cache_stats(g_ctx, 1, st0);
        bool all_same = true;
        for (int round = 0; round < 2; ++round)
            for (int j = 60; j < n_pending; ++j) {
                if (round == 1 && j == 90) {  // another user of the context in between: its own cluster and snapshot
                    auto other = NewTestExtender(packer, {NewNode("node1", "zone1"), NewNode("node2", "zone1")});
                    FlatCluster oc;
                    CHECK(FlatCluster::Build(other.nodes, &oc, &err));
                    Pod small = Driver("small", StaticAnnotations(1), 5);
                    other.pods = {small};
                    SelectNodeResult o = other.selectDriverNodeFlat("batch-medium-priority", small, {"node1", "node2"}, oc);
                    CHECK(o.served && o.outcome == std::string(outcome::success));
                }
                const SelectNodeResult got = ext.selectDriverNodeFlat("batch-medium-priority", ext.pods[(size_t)j], names, cluster, &flat);
                const SelectNodeResult& w = want[(size_t)(j - 60)];
                bool same = got.served && w.served && got.outcome == w.outcome && got.node == w.node &&
                            got.created.has_value() == w.created.has_value();
                if (same && got.created)
                    for (const auto& [name, res] : w.created->Reservations)
                        same = same && got.created->Reservations.count(name) && got.created->Reservations.at(name).Node == res.Node;
                all_same = all_same && same;
            }
        CHECK(all_same);
        gf_chain_cache_stats(g_ctx, 0, st1);
        CHECK(st1[0] == 2u * (n_pending - 60));
        CHECK(st1[1] >= 2u * (n_pending - 60) - 4u);  // every Filter but the first of a round (and the one behind the intruder) resumes
        // a pod whose annotations change arrives with a new resourceVersion: its cached requests are parsed again
        ext.pods[70].Annotations["spark-executor-count"] = "57";
        ext.pods[70].ResourceVersion += 1000;
        const Pod& lastd = ext.pods[(size_t)n_pending - 1];
        const SelectNodeResult w2 = ext.selectDriverNode("batch-medium-priority", lastd, names, ext.nodes);
        const SelectNodeResult g2 = ext.selectDriverNodeFlat("batch-medium-priority", lastd, names, cluster, &flat);
        bool same2 = g2.served && w2.served && g2.outcome == w2.outcome && g2.node == w2.node && g2.created.has_value() == w2.created.has_value();
        if (same2 && g2.created)
            for (const auto& [name, res] : w2.created->Reservations)
                same2 = same2 && g2.created->Reservations.count(name) && g2.created->Reservations.at(name).Node == res.Node;
        CHECK(same2);
    }
}

// ------------------------------------------------------------------------------------------------ failover: findNodes
// The loop of internal/extender/failover.go:412-436 on the host mirror's own types (Quantity arithmetic, string-keyed maps):
// what the device-backed findNodes must reproduce, over-add included.
static std::pair<std::vector<std::string>, NodeGroupResources> literalFindNodes(int executorCount, const Resources& executorResources,
                                                                                 const NodeGroupResources& availableResources,
                                                                                 const std::vector<Node>& orderedNodes) {
    std::vector<std::string> executorNodeNames;
    NodeGroupResources reserved;
    for (const Node& n : orderedNodes) {
        if (!reserved.count(n.Name)) reserved[n.Name] = Resources::Zero();
        for (;;) {
            reserved[n.Name].Add(executorResources);
            if (reserved[n.Name].GreaterThan(availableResources.at(n.Name))) break;
            executorNodeNames.push_back(n.Name);
            if ((int)executorNodeNames.size() == executorCount) return {executorNodeNames, reserved};
        }
    }
    return {executorNodeNames, reserved};
}

static bool sameReserved(const NodeGroupResources& a, const NodeGroupResources& b) {
    if (a.size() != b.size()) return false;
    for (const auto& [k, v] : a) {
        auto it = b.find(k);
        if (it == b.end() || !it->second.Eq(v)) return false;
    }
    return true;
}

static void TestFindNodes() {
    std::vector<Node> ordered;
    NodeGroupResources available;
    const int64_t cpu[] = {4, 0, 9, 2, 16, 1, 7};
    const int64_t memGi[] = {8, 4, 2, 64, 16, 1, 7};
    const int64_t gpu[] = {0, 0, 1, 0, 2, 0, 0};
    for (int i = 0; i < 7; ++i) {
        Node n;
        n.Name = "node-" + std::to_string(i);
        n.Ready = true;
        ordered.push_back(n);
        available[n.Name] = Resources::Create(cpu[i], memGi[i] * Gi, gpu[i]);
    }
    available["node-5"].CPU.Sub(Quantity::FromInt(3));  // an overcommitted node: the first add already exceeds
    struct Rq {
        int count;
        Resources exe;
    };
    const Rq rqs[] = {{5, Resources::Create(1, 2 * Gi, 0)},  {40, Resources::Create(2, 1 * Gi, 0)}, {2, Resources::Create(1, 1 * Gi, 1)},
                      {3, Resources::Create(32, 1 * Gi, 0)}, {4, Resources::Create(0, 0, 0)},        {1, Resources::Create(4, 8 * Gi, 0)}};
    for (const Rq& rq : rqs) {
        auto want = literalFindNodes(rq.count, rq.exe, available, ordered);
        FindNodesResult got = findNodes(g_ctx, rq.count, rq.exe, available, ordered);
        CHECK(got.served);
        CHECK(got.executorNodeNames == want.first);
        CHECK(sameReserved(got.reserved, want.second));
    }
    // the reconcile loop over several stale applications: each sees availableResources after the previous `Sub` (:159)
    std::vector<FindNodesRequest> chain;
    for (const Rq& rq : rqs) chain.push_back({rq.count, rq.exe});
    NodeGroupResources avail_dev = available, avail_ref = available;
    std::vector<FindNodesResult> got = findNodesForStaleApplications(g_ctx, chain, &avail_dev, ordered);
    for (size_t q = 0; q < chain.size(); ++q) {
        auto want = literalFindNodes(chain[q].executorCount, chain[q].executorResources, avail_ref, ordered);
        for (const auto& [node, r] : want.second) avail_ref[node].Sub(r);
        CHECK(got[q].served);
        CHECK(got[q].executorNodeNames == want.first);
        CHECK(sameReserved(got[q].reserved, want.second));
    }
    CHECK(sameReserved(avail_dev, avail_ref));
    // a quantity the canonical units cannot hold exactly must be refused, not rounded
    Resources odd = Resources::Create(1, 1 * Gi, 0);
    odd.CPU = Quantity::FromNano(1500);  // 1.5 micro-cores
    CHECK(!findNodes(g_ctx, 1, odd, available, ordered).served);
}

// ------------------------------------------------------------------------------------------------ two threads, one context
// cmd/server.go:230 starts the UnschedulablePodMarker next to the HTTP server: Predicate (a FIFO Filter) and the marker's
// scan (an independent batch on a DIFFERENT snapshot: the empty cluster) reach the binpacker concurrently.  Both go through
// one gf_ctx here; every answer must equal the single-threaded one.
static void TestTwoThreadsOneContext() {
    Node node1 = NewNode("node1", "zone1"), node2 = NewNode("node2", "zone1");
    auto filter = NewTestExtender("tightly-pack", {node1, node2});
    Pod first = Driver("first", StaticAnnotations(6), 1), second = Driver("second", StaticAnnotations(8), 2);
    filter.pods = {first, second};
    std::vector<Node> big;
    for (int i = 0; i < 150; ++i) big.push_back(NewNode(("big-" + std::to_string(i)).c_str(), i % 3 == 0 ? "zone1" : "zone2"));
    auto marker = NewTestExtender("tightly-pack", big);
    marker.nowNanos = 10000ll * 1000000000;
    std::vector<Pod> pending;
    for (int i = 0; i < 40; ++i) pending.push_back(Driver(("app-" + std::to_string(i)).c_str(), StaticAnnotations(1 + 37 * i), 1 + i));
    bool served = true;
    std::string err;
    const SelectNodeResult want_filter = filter.selectDriverNode("batch-medium-priority", second, {"node1", "node2"}, filter.nodes);
    const auto want_scan = marker.scanForUnschedulablePods(pending, 600ll * 1000000000, marker.nodes, {}, &served, &err);
    CHECK(want_filter.served && want_filter.outcome == std::string(outcome::success));
    CHECK(served && want_scan.size() == pending.size());
    int n_exceed = 0;
    for (const auto& kv : want_scan) n_exceed += kv.second ? 1 : 0;
    CHECK(n_exceed > 0 && n_exceed < (int)want_scan.size());  // both answers occur: 150 nodes x 8 cpu hold 1 + 37 i up to i = 32
    std::atomic<int> bad_filter{0}, bad_scan{0};
    const int iters = 200;
    std::thread t1([&] {
        for (int i = 0; i < iters; ++i) {
            SelectNodeResult r = filter.selectDriverNode("batch-medium-priority", second, {"node1", "node2"}, filter.nodes);
            bool same = r.served && r.outcome == want_filter.outcome && r.node == want_filter.node && r.created.has_value();
            if (same)
                for (const auto& [name, res] : want_filter.created->Reservations)
                    same = same && r.created->Reservations.count(name) && r.created->Reservations.at(name).Node == res.Node;
            if (!same) ++bad_filter;
        }
    });
    std::thread t2([&] {
        for (int i = 0; i < iters; ++i) {
            bool sv = true;
            std::string e;
            auto r = marker.scanForUnschedulablePods(pending, 600ll * 1000000000, marker.nodes, {}, &sv, &e);
            if (!sv || r != want_scan) ++bad_scan;
        }
    });
    t1.join();
    t2.join();
    CHECK(bad_filter.load() == 0);
    CHECK(bad_scan.load() == 0);
}

// ------------------------------------------------------------------------------------------------ views: concurrent chains
// gf_ctx_view (include/gangfit.h): eight views of one context run eight 999 + 1 FIFO chains (different heads of one queue) at
// the same time.  Each chain is one workgroup, so eight of them occupy eight compute units: they must finish in about the time
// of one, every view's results and residual table must be the ones it gets when it runs alone (no cross-talk), and the parent
// keeps answering an independent batch in between.
static void TestConcurrentViews() {
    const uint32_t n = 10000, n_apps = 1000, n_views = 8;
    uint64_t rng = 0xC0FFEE;
    auto next = [&]() {
        rng += 0x9E3779B97F4A7C15ull;
        uint64_t z = rng;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    std::vector<int64_t> cpu(n), mem(n), gpu(n, 0);
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) {
        cpu[i] = (int64_t)(4 + next() % 60) * 1000;
        mem[i] = (int64_t)(8 + next() % 248) * Gi;
        order[i] = i;
    }
    gf_ctx* parent = nullptr;
    CHECK(gf_init(nullptr, 0, &parent) == GF_OK);
    if (!parent) return;
    CHECK(gf_set_option(parent, "chain_cache", 0) == GF_OK);  // every chain replays: what is timed is eight full chains
    CHECK(gf_snapshot_set(parent, n, cpu.data(), mem.data(), gpu.data(), nullptr, nullptr, nullptr) == GF_OK);
    CHECK(gf_orders_set(parent, order.data(), n, order.data(), n) == GF_OK);
    std::vector<gf_app> queue(n_apps);
    uint64_t total_k = 0;
    for (gf_app& a : queue) {
        a = gf_app{};
        a.drv[0] = 1000 * (int64_t)(1 + next() % 3);
        a.drv[1] = (int64_t)(2 << (next() % 3)) * Gi;
        a.exe[0] = 1000 * (int64_t)(1 << (next() % 4));
        a.exe[1] = (int64_t)(4 << (next() % 4)) * Gi;
        a.k = 1 + (int32_t)(next() % 24);
        a.flags = GF_APP_SKIPPABLE;
        total_k += (uint64_t)a.k;
    }
    struct Run {
        gf_ctx* v = nullptr;
        std::vector<gf_app> apps;
        std::vector<gf_result> res, want_res;
        std::vector<uint32_t> exec, want_exec;
        std::vector<int64_t> resid, want_resid;
        int32_t failed = -1;
    };
    std::vector<Run> runs(n_views);
    bool ok = true;
    for (uint32_t i = 0; i < n_views; ++i) {
        Run& r = runs[i];
        ok = ok && gf_ctx_view(parent, &r.v) == GF_OK;
        r.apps.assign(queue.begin() + i, queue.end());
        r.apps.insert(r.apps.end(), queue.begin(), queue.begin() + i);  // another head
        uint64_t off = 0;
        for (gf_app& a : r.apps) {  // (gf_fit_batch fills exec_off in its own copy; the comparison below needs it here)
            a.exec_off = off;
            off += (uint64_t)a.k;
        }
        r.res.resize(n_apps);
        r.want_res.resize(n_apps);
        r.exec.resize(total_k + 1);
        r.want_exec.resize(total_k + 1);
        r.resid.resize(3 * (size_t)n);
        r.want_resid.resize(3 * (size_t)n);
    }
    CHECK(ok);
    if (!ok) return;
    auto chain = [&](Run& r, std::vector<gf_result>& res, std::vector<uint32_t>& exec) {
        return gf_fit_batch(r.v, GF_MODE_FIFO_CHAIN, GF_ALGO_TIGHTLY_PACK, n_apps, r.apps.data(), res.data(), exec.data(), total_k, &r.failed);
    };
    for (Run& r : runs) {  // alone: the answers to reproduce (and warm buffers)
        ok = ok && chain(r, r.want_res, r.want_exec) == GF_OK && gf_residual_get(r.v, r.want_resid.data()) == GF_OK;
    }
    CHECK(ok);
    using Clock = std::chrono::steady_clock;
    const int reps = 10;
    auto t0 = Clock::now();
    for (int i = 0; i < reps; ++i) ok = ok && chain(runs[0], runs[0].res, runs[0].exec) == GF_OK;
    const double one_ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count() / reps;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    t0 = Clock::now();
    for (uint32_t i = 0; i < n_views; ++i)
        th.emplace_back([&, i] {
            Run& r = runs[i];
            for (int it = 0; it < reps; ++it) {
                if (chain(r, r.res, r.exec) != GF_OK || gf_residual_get(r.v, r.resid.data()) != GF_OK) ++bad;
                if (std::memcmp(r.res.data(), r.want_res.data(), n_apps * sizeof(gf_result)) != 0 || r.resid != r.want_resid) ++bad;
                for (uint32_t a = 0; a < n_apps; ++a)
                    if (r.want_res[a].has_capacity &&
                        std::memcmp(&r.exec[r.apps[a].exec_off], &r.want_exec[r.apps[a].exec_off], r.want_res[a].exec_len * 4u) != 0) {
                        ++bad;
                        break;
                    }
            }
        });
    // the parent answers an independent batch while the views walk their chains
    std::vector<gf_result> pres(n_apps);
    std::vector<uint32_t> pexec(total_k + 1);
    bool parent_ok = true;
    for (int it = 0; it < 20; ++it)
        parent_ok = parent_ok && gf_fit_batch(parent, GF_MODE_INDEPENDENT, GF_ALGO_TIGHTLY_PACK, n_apps, queue.data(), pres.data(),
                                              pexec.data(), total_k, nullptr) == GF_OK;
    for (std::thread& t : th) t.join();
    const double eight_ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count() / reps;
    CHECK(parent_ok);
    CHECK(bad.load() == 0);
    std::printf("   one chain %.3f ms; eight concurrent chains (+ residual reads) %.3f ms per round = %.2fx\n", one_ms, eight_ms,
                eight_ms / one_ms);
    // (each round also fetches the 240 KB residual table; the chains themselves overlap: 1.1x on an otherwise idle GPU.  The
    //  bound here only separates "concurrent" from "one after the other" (8x): this test shares the box with whatever else
    //  the suite is running, the measured ratio is printed above and recorded by host_bench)
    CHECK(eight_ms < 4.0 * one_ms);
    // the residuals of different heads differ: the views really worked on tables of their own
    CHECK(runs[0].want_resid != runs[1].want_resid);
    // installs are refused on a view; a new snapshot on the parent is what the views see next
    CHECK(gf_orders_set(runs[0].v, order.data(), n, order.data(), n) == GF_ERR_STATE);
    for (uint32_t i = 0; i < 200; ++i) cpu[i] = 0;
    CHECK(gf_snapshot_set(parent, n, cpu.data(), mem.data(), gpu.data(), nullptr, nullptr, nullptr) == GF_OK);
    CHECK(gf_orders_set(parent, order.data(), n, order.data(), n) == GF_OK);
    CHECK(chain(runs[0], runs[0].res, runs[0].exec) == GF_OK);
    CHECK(std::memcmp(runs[0].res.data(), runs[0].want_res.data(), n_apps * sizeof(gf_result)) != 0);
    std::vector<int64_t> snap(3 * (size_t)n);
    CHECK(gf_snapshot_get(runs[0].v, snap.data(), nullptr) == GF_OK && snap[0] == 0 && snap[3 * 199] == 0 && snap[3 * 200] == cpu[200]);
    for (Run& r : runs) gf_destroy(r.v);
    gf_destroy(parent);
}

// ------------------------------------------------------------------------------------------------ one context, several devices
// gf_init with n_dev > 1 (include/gangfit.h): the Go shim reaches node-range sharding by passing more device ids, nothing
// else changes.  Driven here through gangfit.h only (no torch, no Python): the sharded batch must equal the one-device one.
// The resident worker of the independent batch through the C ABI (gf_worker_fit): what a launch answers, batch after batch,
// across an install (which makes the worker leave) and after it has left for lack of work.
static void TestResidentWorker() {
    const uint32_t n = 4000, n_apps = 600;
    uint64_t rng = 0xFEEDBEEF;
    auto next = [&]() {
        rng += 0x9E3779B97F4A7C15ull;
        uint64_t z = rng;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    gf_ctx* ctx = nullptr;
    CHECK(gf_init(nullptr, 0, &ctx) == GF_OK);
    if (!ctx) return;
    CHECK(gf_set_option(ctx, "worker_idle_us", 100) == GF_OK);
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    for (int round = 0; round < 2; ++round) {  // two snapshots: the second install finds a worker on the device
        std::vector<int64_t> cpu(n), mem(n), gpu(n, 0);
        for (uint32_t i = 0; i < n; ++i) {
            cpu[i] = (int64_t)(4 + next() % 60) * 1000;
            mem[i] = (int64_t)(8 + next() % 248) * Gi;
        }
        CHECK(gf_snapshot_set(ctx, n, cpu.data(), mem.data(), gpu.data(), nullptr, nullptr, nullptr) == GF_OK);
        CHECK(gf_orders_set(ctx, order.data(), n, order.data(), n) == GF_OK);
        for (gf_algo algo : {GF_ALGO_TIGHTLY_PACK, GF_ALGO_DISTRIBUTE_EVENLY, GF_ALGO_MINIMAL_FRAGMENTATION}) {
            for (int batch = 0; batch < 4; ++batch) {
                const uint32_t na = batch == 3 ? 1u : n_apps - 37u * (uint32_t)batch;
                std::vector<gf_app> apps(na);
                uint64_t total_k = 0;
                for (gf_app& a : apps) {
                    a = gf_app{};
                    a.drv[0] = 1000 * (int64_t)(1 + next() % 3);
                    a.drv[1] = (int64_t)(2 << (next() % 3)) * Gi;
                    a.exe[0] = 1000 * (int64_t)(1 << (next() % 4));
                    a.exe[1] = (int64_t)(4 << (next() % 4)) * Gi;
                    a.k = (int32_t)(next() % 24);
                    total_k += (uint64_t)a.k;
                }
                std::vector<gf_result> want(na), got(na);
                std::vector<uint32_t> want_x(total_k + 1), got_x(total_k + 1);
                CHECK(gf_fit_batch(ctx, GF_MODE_INDEPENDENT, algo, na, apps.data(), want.data(), want_x.data(), total_k, nullptr) == GF_OK);
                CHECK(gf_worker_fit(ctx, algo, na, apps.data(), got.data(), got_x.data(), total_k) == GF_OK);
                CHECK(std::memcmp(want.data(), got.data(), na * sizeof(gf_result)) == 0);
                CHECK(std::memcmp(want_x.data(), got_x.data(), total_k * sizeof(uint32_t)) == 0);
                if (batch == 1) std::this_thread::sleep_for(std::chrono::milliseconds(5));  // the worker leaves; the next batch brings it back
            }
        }
    }
    uint64_t st[4] = {0, 0, 0, 0};
    CHECK(gf_worker_stats(ctx, st) == GF_OK);
    CHECK(st[0] == 24 && st[1] == 24 && st[2] >= 8);  // 24 tickets; at least one launch per (snapshot, packer) + the idle exits
    CHECK(gf_worker_stop(ctx) == GF_OK);
    CHECK(gf_worker_stats(ctx, st) == GF_OK && st[3] == 0);
    gf_app bad{};
    bad.k = 1;
    gf_result r{};
    uint32_t x[2];
    CHECK(gf_worker_fit(ctx, GF_ALGO_SINGLE_AZ_TIGHTLY_PACK, 1, &bad, &r, x, 1) != GF_OK);  // zone packers are not served
    gf_destroy(ctx);
}

static void TestMultiDeviceContext() {
    const uint32_t n = 3000, n_apps = 500;
    std::vector<int64_t> col[3];
    std::vector<uint32_t> order(n);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&x]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    for (uint32_t i = 0; i < n; ++i) {
        col[0].push_back((int64_t)(rnd() % 64) * 500 - 1000);
        col[1].push_back((int64_t)(rnd() % 128) * (Gi / 2));
        col[2].push_back(rnd() % 10 == 0 ? (int64_t)(rnd() % 8) : 0);
        order[i] = i;
    }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return col[1][a] != col[1][b] ? col[1][a] < col[1][b] : (col[0][a] != col[0][b] ? col[0][a] < col[0][b] : a < b);
    });
    std::vector<gf_app> apps(n_apps);
    uint64_t total_k = 0;
    for (gf_app& a : apps) {
        a = gf_app{};
        a.drv[0] = 1000 * (1 + (int64_t)(rnd() % 3));
        a.drv[1] = Gi * (1 + (int64_t)(rnd() % 4));
        a.exe[0] = 500 * (1 + (int64_t)(rnd() % 8));
        a.exe[1] = Gi * (1 + (int64_t)(rnd() % 16));
        a.exe[2] = rnd() % 20 == 0 ? 1 : 0;
        a.k = (int32_t)(1 + rnd() % 90);
        if (rnd() % 9 == 0) a.k = 4000;  // gangs the cluster cannot host: both answers must occur
        total_k += (uint64_t)a.k;
    }
    auto run = [&](gf_ctx* c, gf_algo algo, std::vector<gf_result>* res, std::vector<uint32_t>* exec) {
        res->assign(n_apps, gf_result{});
        exec->assign(total_k + 1, 0u);
        return gf_snapshot_set(c, n, col[0].data(), col[1].data(), col[2].data(), nullptr, nullptr, nullptr) == GF_OK &&
               gf_orders_set(c, order.data(), n, order.data(), n) == GF_OK &&
               gf_fit_batch(c, GF_MODE_INDEPENDENT, algo, n_apps, apps.data(), res->data(), exec->data(), total_k, nullptr) == GF_OK;
    };
    // ---- a box nobody has tried the context on: (a) devices that cannot reach each other's memory degrade the context to its
    //      first device; (b) a wrong exchange (1: the placement pull never runs, 2: the other devices' capacity sums arrive as
    //      zeros) is caught by the self-check of the first sharded batch — right answers, sharding off, the context says why
    auto same_as_one_device = [&](gf_ctx* g, gf_algo algo) {
        std::vector<gf_result> r1, rg;
        std::vector<uint32_t> e1, eg;
        if (!run(g_ctx, algo, &r1, &e1) || !run(g, algo, &rg, &eg)) return false;
        bool same = true;
        uint64_t off = 0;
        for (uint32_t a = 0; a < n_apps; ++a) {
            same = same && r1[a].has_capacity == rg[a].has_capacity && r1[a].driver_node == rg[a].driver_node;
            if (r1[a].has_capacity)
                for (uint32_t i = 0; i < r1[a].exec_len; ++i) same = same && e1[off + i] == eg[off + i];
            off += (uint64_t)apps[a].k;
        }
        return same;
    };
    {
        (void)setenv("GANGFIT_TEST_NO_PEER", "1", 1);
        const int ids[3] = {0, 0, 0};
        gf_ctx* g = nullptr;
        CHECK(gf_init(ids, 3, &g) == GF_OK);
        (void)unsetenv("GANGFIT_TEST_NO_PEER");
        if (g) {
            CHECK(gf_shard_count(g) == 1);
            CHECK(std::string(gf_last_error(g)).find("peer access") != std::string::npos);
            CHECK(same_as_one_device(g, GF_ALGO_TIGHTLY_PACK));
            gf_destroy(g);
        }
    }
    (void)setenv("GANGFIT_TEST_GROUP_SPLIT", "1", 1);
    for (int fault : {1, 2}) {
        const int ids[4] = {0, 0, 0, 0};
        gf_ctx* g = nullptr;
        CHECK(gf_init(ids, 4, &g) == GF_OK);
        if (!g) continue;
        CHECK(same_as_one_device(g, GF_ALGO_TIGHTLY_PACK));
        CHECK(gf_shard_count(g) == 4);
        CHECK(gf_set_option(g, "group_fault", fault) == GF_OK);
        CHECK(same_as_one_device(g, GF_ALGO_TIGHTLY_PACK));  // the self-check serves the first device's answer
        CHECK(gf_shard_count(g) == 1);
        CHECK(std::string(gf_last_error(g)).find("disagreed") != std::string::npos);
        gf_destroy(g);
    }
    (void)unsetenv("GANGFIT_TEST_GROUP_SPLIT");
    // every shard on device 0: the one-GPU form of the eight-GPU context — first with the shards of the device in ONE sub-context
    // (a launch per step, a grid row per shard), then with every listed id in a sub-context, stream and submitting thread of its
    // own (GANGFIT_TEST_GROUP_SPLIT: events between streams, host barriers, pushes into several tables, the placement pull)
    for (int pass = 0; pass < 2; ++pass)
    for (int n_dev : {2, 5, 8}) {
        // gf_init reads the switch: set (or cleared) right before it
        if (pass == 1)
            (void)setenv("GANGFIT_TEST_GROUP_SPLIT", "1", 1);
        else
            (void)unsetenv("GANGFIT_TEST_GROUP_SPLIT");
        std::vector<int> ids((size_t)n_dev, 0);
        gf_ctx* g = nullptr;
        CHECK(gf_init(ids.data(), n_dev, &g) == GF_OK);
        if (!g) continue;
        for (gf_algo algo : {GF_ALGO_TIGHTLY_PACK, GF_ALGO_DISTRIBUTE_EVENLY}) {
            std::vector<gf_result> r1, rg;
            std::vector<uint32_t> e1, eg;
            CHECK(run(g_ctx, algo, &r1, &e1));
            CHECK(run(g, algo, &rg, &eg));
            bool same = true;
            int feasible = 0;
            uint64_t off = 0;
            for (uint32_t a = 0; a < n_apps; ++a) {
                same = same && r1[a].has_capacity == rg[a].has_capacity && r1[a].driver_node == rg[a].driver_node &&
                       r1[a].exec_len == rg[a].exec_len;
                if (r1[a].has_capacity) {
                    ++feasible;
                    for (uint32_t i = 0; i < r1[a].exec_len; ++i) same = same && e1[off + i] == eg[off + i];
                }
                off += (uint64_t)apps[a].k;
            }
            CHECK(same);
            CHECK(feasible > 0 && feasible < (int)n_apps);
        }
        // a FIFO chain on the group runs on its first device and still answers
        std::vector<gf_result> rc(n_apps);
        std::vector<uint32_t> ec(total_k + 1);
        int32_t failed = 0;
        CHECK(gf_fit_batch(g, GF_MODE_FIFO_CHAIN, GF_ALGO_TIGHTLY_PACK, n_apps, apps.data(), rc.data(), ec.data(), total_k, &failed) == GF_OK);
        gf_destroy(g);
    }
    (void)unsetenv("GANGFIT_TEST_GROUP_SPLIT");
}

int main(int argc, char** argv) {
    // the deployment's part ("Deployment"): the library never changes the environment itself
    (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    if (mode == "cpu" || mode == "all") {
        TestParseQuantity();
        TestSparkResources();
        TestIsEarliest();
        TestResourcesSorting();
        TestScheduleContextSorting();
        TestAZAwareNodeSorting();
        TestAZAwareNodeSortingWorksIfZoneLabelIsMissing();
        TestLabelPrioritySorting();
        TestPotentialNodes();
        TestSnapshotAndReservations();
    }
    if (mode == "gpu" || mode == "all") {
        if (gf_init(nullptr, 0, &g_ctx) != GF_OK) {
            std::printf("FAIL gf_init: no gfx950 device (there is no CPU fallback)\n");
            return 2;
        }
        TestScheduler();
        TestUnschedulablePodMarker();
        TestUnschedulablePodScanBatched();
        TestSchedulerFailsToScheduleWhenNotEnoughNvidiaGPUs();
        TestFifoAndBinpackers();
        TestMinimalFragmentationEdgeCase();
        TestDynamicAllocationSameAZ();
        TestDeviceSnapshotBuildAgainstHostMirror();
        TestIncrementalFilters();
        TestFindNodes();
        TestTwoThreadsOneContext();
        TestConcurrentViews();
        TestResidentWorker();
        TestMultiDeviceContext();
        gf_destroy(g_ctx);
    }
    std::printf("%s: %d checks, %d failed\n", mode.c_str(), g_checked, g_failed);
    return g_failed == 0 ? 0 : 1;
}
