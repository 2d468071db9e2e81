// host_bench.cpp — the driver Filter end to end through the C++ mirror of the reference's interface (string node names,
// Quantity-valued annotations, ResourceReservation objects): BASELINE's "p99 Filter latency at 10k nodes x 1k pending
// apps" measured over the whole span the reference times (internal/extender/resource.go:142-154 `schedule.time`), not
// just the kernel.  Two routes:
//   map   selectDriverNode      — UsageForNodes / NodeSchedulingMetadataForNodes / PotentialNodes on string-keyed maps
//                                 like the Go host, then ONE FIFO-chain call
//   flat  selectDriverNodeFlat  — flat columns -> gf_snapshot_build (replay + metadata + sort on the device) -> the chain
// usage: host_bench [n_nodes] [n_pending] [n_reservations] [packer]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "extender.hpp"

using namespace gangfit::host;
using Clock = std::chrono::steady_clock;

static uint64_t g_rng = 0x5EED0010;
static uint64_t next() {
    g_rng += 0x9E3779B97F4A7C15ull;
    uint64_t z = g_rng;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static double pct(std::vector<double> v, double q) {
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, (size_t)(q * (v.size() - 1) + 0.5))];
}

int main(int argc, char** argv) {
    // the deployment's part ("Deployment"): the library never changes the environment itself
    (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
    const int n_nodes = argc > 1 ? std::atoi(argv[1]) : 10000;
    const int n_pending = argc > 2 ? std::atoi(argv[2]) : 1000;
    const int n_rr = argc > 3 ? std::atoi(argv[3]) : 2000;
    const std::string packer = argc > 4 ? argv[4] : "tightly-pack";
    const int64_t Gi = 1024ll * 1024 * 1024;
    gf_ctx* ctx = nullptr;
    if (gf_init(nullptr, 0, &ctx) != GF_OK) {
        std::printf("no gfx950 device (there is no CPU fallback)\n");
        return 2;
    }
    const Binpacker chosen = SelectBinpacker(packer, ctx);
    const gf_algo ext_algo = chosen.Algo;
    SparkSchedulerExtender ext(chosen, NodeSorter(), true, FifoConfig{});
    const char* zones[] = {"az-a", "az-b", "az-c"};
    const int cpus[] = {16, 32, 64, 96}, mems[] = {64, 128, 256, 384};
    std::vector<std::string> nodeNames;
    for (int i = 0; i < n_nodes; ++i) {
        Node nd;
        char buf[32];
        std::snprintf(buf, sizeof buf, "node-%06d", (int)(next() % 1000000));
        nd.Name = std::string(buf) + "-" + std::to_string(i);
        nd.labels[kLabelZoneFailureDomain] = zones[next() % 3];
        const int shape = (int)(next() % 4);
        nd.Allocatable = {{kResourceCPU, Quantity::FromInt(cpus[shape])}, {kResourceMemory, Quantity::FromInt(mems[shape] * Gi)},
                          {kResourceNvidiaGPU, Quantity::FromInt(0)}};
        nd.Ready = true;
        nodeNames.push_back(nd.Name);
        ext.nodes.push_back(std::move(nd));
    }
    for (int r = 0; r < n_rr; ++r) {  // running applications: K + 1 reservations each
        ResourceReservation rr;
        rr.Name = "running-" + std::to_string(r);
        rr.Namespace = "namespace";
        const int k = 1 + (int)(next() % 24);
        for (int e = 0; e <= k; ++e) {
            Reservation res;
            res.Node = ext.nodes[next() % n_nodes].Name;
            res.Resources = {{kResourceCPU, Quantity::FromInt(1 + (int64_t)(next() % 2))},
                             {kResourceMemory, Quantity::FromInt((int64_t)(2 + next() % 6) * Gi)},
                             {kResourceNvidiaGPU, Quantity::FromInt(0)}};
            rr.Reservations[e == 0 ? "driver" : executorReservationName(e - 1)] = std::move(res);
        }
        ext.reservations.push_back(std::move(rr));
    }
    const char* dcpu[] = {"1", "2", "4"};
    const char* dmem[] = {"2Gi", "4Gi", "8Gi"};
    const char* ecpu[] = {"1", "2", "4", "8"};
    const char* emem[] = {"4Gi", "8Gi", "16Gi", "32Gi"};
    for (int p = 0; p < n_pending; ++p) {
        Pod pod;
        pod.Name = "pending-" + std::to_string(p) + "-spark-driver";
        pod.Namespace = "namespace";
        pod.labels = {{common::SparkRoleLabel, common::Driver}, {common::SparkAppIDLabel, "pending-" + std::to_string(p)}};
        int k = 1;
        while (k < 512 && next() % 12 != 0) ++k;
        pod.Annotations = {{common::DriverCPU, dcpu[next() % 3]},   {common::DriverMemory, dmem[next() % 3]},
                           {common::ExecutorCPU, ecpu[next() % 4]}, {common::ExecutorMemory, emem[next() % 4]},
                           {common::ExecutorCount, std::to_string(k)}};
        pod.SchedulerName = common::SparkSchedulerName;
        pod.InstanceGroup = "batch-medium-priority";
        pod.CreationTimestampNanos = (int64_t)(p + 1) * 1000000000;
        pod.UID = "uid-" + std::to_string(p);
        pod.ResourceVersion = 1000 + (uint64_t)p;  // what an informer hands over: the flat route parses each version once
        ext.pods.push_back(std::move(pod));
    }
    ext.nowNanos = (int64_t)(n_pending + 10) * 1000000000;
    const Pod& driver = ext.pods.back();  // the youngest: all the others are earlier drivers
    FlatCluster cluster;
    std::string err;
    auto t0 = Clock::now();
    if (!FlatCluster::Build(ext.nodes, &cluster, &err)) {
        std::printf("FlatCluster::Build: %s\n", err.c_str());
        return 1;
    }
    const double build_ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    FlatReservations flat;
    t0 = Clock::now();
    if (!FlatReservations::Build(ext.reservations, ext.softReservationUsage, cluster, &flat, &err)) {
        std::printf("FlatReservations::Build: %s\n", err.c_str());
        return 1;
    }
    const double flat_rr_ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();

    SelectNodeResult a = ext.selectDriverNode("batch-medium-priority", driver, nodeNames, ext.nodes);
    SelectNodeResult b = ext.selectDriverNodeFlat("batch-medium-priority", driver, nodeNames, cluster);
    bool same = a.served && b.served && a.outcome == b.outcome && a.node == b.node && a.created.has_value() == b.created.has_value();
    if (same && a.created)
        for (const auto& [name, res] : a.created->Reservations)
            same = same && b.created->Reservations.count(name) && b.created->Reservations.at(name).Node == res.Node;
    std::printf("routes agree: %s (outcome %s, node %s, %zu reservations)%s%s\n", same ? "yes" : "NO", b.outcome.c_str(),
                b.node.c_str(), b.created ? b.created->Reservations.size() : 0, b.served ? "" : "  flat not served: ",
                b.served ? "" : b.error.c_str());
    std::vector<double> map_ms, flat_ms, cached_ms;
    for (int i = 0; i < 8; ++i) {
        t0 = Clock::now();
        ext.selectDriverNode("batch-medium-priority", driver, nodeNames, ext.nodes);
        map_ms.push_back(std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    }
    for (int i = 0; i < 40; ++i) {
        t0 = Clock::now();
        ext.selectDriverNodeFlat("batch-medium-priority", driver, nodeNames, cluster);
        flat_ms.push_back(std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    }
    // the reservations kept in flat form next to the host's reservation cache (cluster columns and usage sums resident):
    //   cached     every Filter rebuilds the snapshot and replays the whole chain (what round 2 measured)
    //   unchanged  nothing changed since the previous Filter: no rebuild, the chain still replays (chain cache off)
    //   order      ... and the chain resumes: Filters for the last 64 drivers in creation order, cyclically
    //   retry      ... the same driver again
    std::vector<double> unchanged_ms, order_ms, retry_ms;
    for (int i = 0; i < 40; ++i) {
        ext.forgetInstalledSnapshot();
        t0 = Clock::now();
        ext.selectDriverNodeFlat("batch-medium-priority", driver, nodeNames, cluster, &flat);
        cached_ms.push_back(std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    }
    gf_set_option(ctx, "chain_cache", 0);
    for (int i = 0; i < 40; ++i) {
        t0 = Clock::now();
        ext.selectDriverNodeFlat("batch-medium-priority", driver, nodeNames, cluster, &flat);
        unchanged_ms.push_back(std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    }
    gf_set_option(ctx, "chain_cache", 1);
    const int span = n_pending < 64 ? n_pending : 64;
    for (int i = 0; i < 3 * span + 4; ++i) {
        const Pod& d = ext.pods[(size_t)(n_pending - span + i % span)];
        t0 = Clock::now();
        ext.selectDriverNodeFlat("batch-medium-priority", d, nodeNames, cluster, &flat);
        if (i >= 4) order_ms.push_back(std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    }
    for (int i = 0; i < 44; ++i) {
        t0 = Clock::now();
        SelectNodeResult r = ext.selectDriverNodeFlat("batch-medium-priority", driver, nodeNames, cluster, &flat);
        if (i >= 4) retry_ms.push_back(std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
        if (i == 43) {
            bool again = r.served && r.outcome == b.outcome && r.node == b.node && r.created.has_value() == b.created.has_value();
            if (again && r.created)
                for (const auto& [name, res] : b.created->Reservations)
                    again = again && r.created->Reservations.count(name) && r.created->Reservations.at(name).Node == res.Node;
            same = same && again;  // the resumed chain answers what the replayed one answered
        }
    }
    std::printf("{\"nodes\": %d, \"pending_drivers\": %d, \"resource_reservations\": %d, \"packer\": \"%s\", "
                "\"filter_map_route_ms\": {\"p50\": %.3f, \"p99\": %.3f}, \"filter_flat_route_ms\": {\"p50\": %.3f, \"p99\": %.3f}, "
                "\"filter_flat_route_cached_reservations_ms\": {\"p50\": %.3f, \"p99\": %.3f}, "
                "\"filter_unchanged_snapshot_replayed_chain_ms\": {\"p50\": %.3f, \"p99\": %.3f}, "
                "\"filter_unchanged_snapshot_creation_order_heads_ms\": {\"p50\": %.3f, \"p99\": %.3f}, "
                "\"filter_unchanged_snapshot_same_head_ms\": {\"p50\": %.3f, \"p99\": %.3f}, \"reservation_entries\": %zu, "
                "\"flat_cluster_build_ms\": %.3f, \"flat_reservations_build_ms\": %.3f, \"routes_agree\": %s}\n",
                n_nodes, n_pending, n_rr, packer.c_str(), pct(map_ms, 0.5), pct(map_ms, 0.99), pct(flat_ms, 0.5),
                pct(flat_ms, 0.99), pct(cached_ms, 0.5), pct(cached_ms, 0.99), pct(unchanged_ms, 0.5), pct(unchanged_ms, 0.99),
                pct(order_ms, 0.5), pct(order_ms, 0.99), pct(retry_ms, 0.5), pct(retry_ms, 0.99), flat.node.size(), build_ms, flat_rr_ms,
                same ? "true" : "false");
    // ---- eight Filters' chains at once: eight views of the context (gf_ctx_view) walk the pending queue from eight different
    //      heads on the snapshot the last Filter installed, one thread each — against one of them alone
    {
        std::vector<gf_app> q;
        for (const Pod& p : ext.pods) {
            auto r = sparkResources(p, nullptr);
            gf_app a{};
            if (!r || !r->DriverResources.canonical(a.drv) || !r->ExecutorResources.canonical(a.exe)) continue;
            a.k = r->MinExecutorCount;
            a.flags = GF_APP_SKIPPABLE;
            q.push_back(a);
        }
        uint64_t total_k = 0;
        for (const gf_app& a : q) total_k += (uint64_t)a.k;
        const int n_views = 8, reps = 10;
        gf_set_option(ctx, "chain_cache", 0);  // full replays: the chains themselves are what is timed
        std::vector<gf_ctx*> views(n_views, nullptr);
        std::vector<std::vector<gf_app>> qs(n_views);
        bool ok = true;
        for (int i = 0; i < n_views; ++i) {
            ok = ok && gf_ctx_view(ctx, &views[i]) == GF_OK;
            qs[i].assign(q.begin() + i, q.end());
            qs[i].insert(qs[i].end(), q.begin(), q.begin() + i);
        }
        const gf_algo algo = ext_algo;
        auto chain = [&](int i) {
            std::vector<gf_result> res(q.size());
            std::vector<uint32_t> exec(total_k + 1);
            int32_t failed = -1;
            return gf_fit_batch(views[i], GF_MODE_FIFO_CHAIN, algo, (uint32_t)q.size(), qs[i].data(), res.data(), exec.data(), total_k, &failed);
        };
        for (int i = 0; i < n_views && ok; ++i) ok = chain(i) == GF_OK;
        t0 = Clock::now();
        for (int r = 0; r < reps && ok; ++r) ok = chain(0) == GF_OK;
        const double one_ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count() / reps;
        std::atomic<int> bad{0};
        std::vector<std::thread> th;
        t0 = Clock::now();
        for (int i = 0; i < n_views; ++i)
            th.emplace_back([&, i] {
                for (int r = 0; r < reps; ++r)
                    if (chain(i) != GF_OK) ++bad;
            });
        for (std::thread& t : th) t.join();
        const double eight_ms = std::chrono::duration<double, std::milli>(Clock::now() - t0).count() / reps;
        std::printf("{\"concurrent_views\": {\"views\": %d, \"chain_apps\": %zu, \"one_chain_ms\": %.3f, \"eight_concurrent_chains_ms\": %.3f, "
                    "\"ratio\": %.2f, \"ok\": %s}}\n", n_views, q.size(), one_ms, eight_ms, eight_ms / one_ms, ok && bad.load() == 0 ? "true" : "false");
        for (gf_ctx* v : views)
            if (v) gf_destroy(v);
    }
    gf_destroy(ctx);
    return same ? 0 : 1;
}
