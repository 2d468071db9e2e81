"""gf_snapshot_build latency (host entry point) at 10k and 100k nodes; run on the MI355X box."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
ctx = gangfit.Context(0, options={"chain_cache": 0})
for n_nodes, n_rr in ((10000, 2000), (100000, 20000)):
    rng = np.random.default_rng(n_nodes)
    shape = rng.integers(0, 4, size=n_nodes)
    alloc = np.stack([np.array([16, 32, 64, 96])[shape] * 1000, np.array([64, 128, 256, 384])[shape] * wl.GIB, np.zeros(n_nodes, dtype=np.int64)], axis=1).astype(np.int64)
    ks = rng.integers(2, 26, size=n_rr)
    rnode = rng.integers(0, n_nodes, size=int(ks.sum())).astype(np.uint32)
    rreq = np.stack([rng.choice([1000, 2000, 4000], size=len(rnode)), rng.choice([4, 8, 16], size=len(rnode)) * wl.GIB, np.zeros(len(rnode), dtype=np.int64)], axis=1).astype(np.int64)
    flags = np.full(n_nodes, 6, dtype=np.uint32)
    ranks = rng.permutation(n_nodes).astype(np.uint32)
    zone = rng.integers(0, 3, size=n_nodes).astype(np.uint32)
    f = lambda: ctx.build_snapshot(alloc, flags, ranks, res_node=rnode, res_req=rreq, zone=zone, n_zones=3, want_orders=False)
    for _ in range(5): f()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(n_nodes, "nodes: gf_snapshot_build p50 %.3f ms p99 %.3f ms" % (ts[50], ts[98]))
    ctx.set_cluster(alloc, flags, ranks, zone=zone, n_zones=3)
    g = lambda: ctx.build_snapshot_resident(res_node=rnode, res_req=rreq, want_orders=False)
    for _ in range(5): g()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); g(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(n_nodes, "nodes: resident cluster + reservations only p50 %.3f ms p99 %.3f ms" % (ts[50], ts[98]))
    ctx.usage_reset()
    rcols = [np.ascontiguousarray(rreq[:, j]) for j in range(3)]
    ctx.usage_apply(rnode, res_cols=rcols, sign=+1)
    h = lambda: ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
    for _ in range(5): h()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); h(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(n_nodes, "nodes: resident cluster + resident usage p50 %.3f ms p99 %.3f ms" % (ts[50], ts[98]))
