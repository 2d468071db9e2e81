import sys, time, os
import numpy as np
REPO = "/root/repo"
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
for n_nodes in (20000, 40000, 100000):
    w = wl.headline(n_nodes, 1000)
    s = w.snapshot
    ctx = gangfit.Context(0, options={"chain_cache": 0})
    ctx.set_snapshot(s.avail, s.sched)
    zone = (wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(3)).astype(np.uint32)
    ctx.set_zones(zone)
    order = wl.reference_node_order(s.avail, zone)
    ctx.set_orders(order, order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    for algo in (5,):
        ctx.fit_batch(1, algo, apps)
        ctx.scan_stats(enable=True, reset=True)
        r = ctx.fit_batch(1, algo, apps)
        ctx.scan_stats(enable=False)
        print("nodes", n_nodes, "algo", algo, "feasible", int(r.results["has_capacity"].sum()), [p // 1000 for p in ctx.last_fifo_phases])
    ctx.close()
