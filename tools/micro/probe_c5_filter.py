"""The bench's config-5 Filter (100 000 nodes, 270 k reservation entries): build variants + chain, p50 over 40 calls."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w5 = wl.config(5)
n5 = len(w5.snapshot.avail)
rng = np.random.default_rng(5)
ks = rng.integers(2, 26, size=20000)
rnode = rng.integers(0, n5, size=int(ks.sum())).astype(np.uint32)
rreq = np.stack([rng.choice([1000, 2000, 4000], size=len(rnode)), rng.choice([4, 8, 16], size=len(rnode)) * wl.GIB,
                 np.zeros(len(rnode), dtype=np.int64)], axis=1).astype(np.int64)
flags5 = np.full(n5, 2 | 4, dtype=np.uint32)
ranks5 = np.arange(n5, dtype=np.uint32)
alloc5 = w5.snapshot.sched + 0
q5 = gangfit.make_apps(w5.drv, w5.exe, w5.k, w5.flags)
ctx = gangfit.Context(0, options={"chain_cache": 0})
rcols = [np.ascontiguousarray(rreq[:, j]) for j in range(3)]
def p50(f, n=40):
    for _ in range(3): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); return round(ts[len(ts) // 2], 3), round(ts[int(len(ts) * 0.99)], 3)
print("build (all columns per call) + chain", p50(lambda: (ctx.build_snapshot(alloc5, flags5, ranks5, res_node=rnode, res_req=rreq, want_orders=False), ctx.fit_batch(1, 0, q5))))
ctx.set_cluster(alloc5, flags5, ranks5)
print("resident cluster, row-major reservations + chain", p50(lambda: (ctx.build_snapshot_resident(res_node=rnode, res_req=rreq, want_orders=False), ctx.fit_batch(1, 0, q5))))
print("resident cluster, reservation columns + chain", p50(lambda: (ctx.build_snapshot_resident(res_node=rnode, res_cols=rcols, want_orders=False), ctx.fit_batch(1, 0, q5))))
print("chain alone", p50(lambda: ctx.fit_batch(1, 0, q5)))
ctx.usage_reset(); ctx.usage_apply(rnode, res_cols=rcols, sign=+1)
dn, dc = rnode[:14], [c[:14] for c in rcols]
print("resident cluster + resident usage (2 delta calls) + chain", p50(lambda: (ctx.usage_apply(dn, res_cols=dc, sign=-1), ctx.usage_apply(dn, res_cols=dc, sign=+1), ctx.build_snapshot_resident(resident_usage=True, want_orders=False), ctx.fit_batch(1, 0, q5))))
print("build from resident usage alone", p50(lambda: (ctx.build_snapshot_resident(resident_usage=True, want_orders=False), ctx.synchronize() if hasattr(ctx, "synchronize") else ctx.snapshot)))
