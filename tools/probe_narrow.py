"""Phase profile of the plain FIFO chain (narrow kernel): wave-0 shader cycles per phase, per app."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
for n_nodes in (10000, 100000):
    w = wl.headline(n_nodes, 1000)
    s = w.snapshot
    ctx = gangfit.Context(0)
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    for algo in (0, 1):
        ctx.fit_batch(1, algo, apps)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
        ctx.scan_stats(enable=True, reset=True)
        ctx.fit_batch(1, algo, apps)
        xv, dv = ctx.scan_stats(enable=False)
        cyc, ticks = ctx.last_fifo_clock
        print("nodes", n_nodes, "algo", algo, "ms", min(ts), "kernel cycles/app", cyc // 1000, "MHz", cyc / max(ticks, 1) * 100,
              "phases", [p // 1000 for p in ctx.last_fifo_phases], "visited x/d per app", xv // 1000, dv // 1000)
    ctx.close()
