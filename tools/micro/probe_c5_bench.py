"""The bench's config-5 chain (100 000 nodes built on the device from 270 k reservation entries): time + phase profile."""
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
ctx.build_snapshot(alloc5, flags5, ranks5, res_node=rnode, res_req=rreq, want_orders=False)
for n_apps in (1, 100, 1000):
    apps = q5[:n_apps]
    ctx.fit_batch(1, 0, apps)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.fit_batch(1, 0, apps); ts.append((time.perf_counter() - t0) * 1e3)
    print(n_apps, "apps", round(min(ts), 3), "ms")
ctx.scan_stats(enable=True, reset=True)
r = ctx.fit_batch(1, 0, q5)
x, d = ctx.scan_stats(enable=False)
cyc, ticks = ctx.last_fifo_clock
print("feasible", int(r.results["has_capacity"].sum()), "failed_at", r.failed_at, "exec slots visited/app", x / 1000, "driver slots/app", d / 1000,
      "phases(cycles/app: stage|driver|exec|slow|commit|visits)", [p // 1000 for p in ctx.last_fifo_phases], "kernel ms", ticks / 1e5)
avail, _ = ctx.snapshot()
print("free cpu quantiles", np.quantile(avail[:, 0], [0, 0.01, 0.5, 0.99]), "free mem GiB quantiles", np.quantile(avail[:, 1], [0, 0.01, 0.5]) / wl.GIB)
