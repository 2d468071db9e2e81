import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
for n_nodes, nz in ((10000, 1), (10000, 3), (100000, 3)):
    w = wl.headline(n_nodes, 1000)
    s = w.snapshot
    ctx = gangfit.Context(0)
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_zones((wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(nz)).astype(np.uint32))
    ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    for algo in (2, 5):
        ctx.fit_batch(1, algo, apps)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
        print("nodes", n_nodes, "zones", nz, "algo", algo, "fifo ms", min(ts))
    ctx.close()
