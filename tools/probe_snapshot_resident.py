"""gf_snapshot_build_resident on a resident cluster + resident usage (what a Filter pays when the snapshot changed), N builds.
   python tools/probe_snapshot_resident.py <n_nodes> [builds]      (run on the MI355X box; under rocprofv3 for the per-kernel split)"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
builds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_rr = n_nodes // 5
ctx = gangfit.Context(0, options={"chain_cache": 0})
rng = np.random.default_rng(n_nodes)
shape = rng.integers(0, 4, size=n_nodes)
alloc = np.stack([np.array([16, 32, 64, 96])[shape] * 1000, np.array([64, 128, 256, 384])[shape] * wl.GIB, np.zeros(n_nodes, dtype=np.int64)], axis=1).astype(np.int64)
ks = rng.integers(2, 26, size=n_rr)
rnode = rng.integers(0, n_nodes, size=int(ks.sum())).astype(np.uint32)
rreq = np.stack([rng.choice([1000, 2000, 4000], size=len(rnode)), rng.choice([4, 8, 16], size=len(rnode)) * wl.GIB, np.zeros(len(rnode), dtype=np.int64)], axis=1).astype(np.int64)
flags = np.full(n_nodes, 6, dtype=np.uint32)
ranks = rng.permutation(n_nodes).astype(np.uint32)
zone = rng.integers(0, 3, size=n_nodes).astype(np.uint32)
ctx.set_cluster(alloc, flags, ranks, zone=zone, n_zones=3)
ctx.usage_reset()
ctx.usage_apply(rnode, res_cols=[np.ascontiguousarray(rreq[:, j]) for j in range(3)], sign=+1)
h = lambda: ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
for _ in range(5):
    h()
ts = []
for _ in range(builds):
    t0 = time.perf_counter(); h(); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(n_nodes, "nodes: resident cluster + resident usage p50 %.3f ms p99 %.3f ms" % (ts[len(ts) // 2], ts[int(len(ts) * 0.99) - 1]))
