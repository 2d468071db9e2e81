"""Fixed cost of a chain call (prologue + epilogue + launches + copies): chains of 1, 100 and 1 000 apps at 10 k / 100 k nodes."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
for n_nodes in (10000, 100000):
    w = wl.config(5, n_nodes=n_nodes)
    s = w.snapshot
    ctx = gangfit.Context(0, options={"chain_cache": 0})
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    out = []
    for n_apps in (1, 100, 1000):
        apps = gangfit.make_apps(w.drv[:n_apps], w.exe[:n_apps], w.k[:n_apps], w.flags[:n_apps])
        ctx.fit_batch(1, 0, apps)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); ctx.fit_batch(1, 0, apps); ts.append((time.perf_counter() - t0) * 1e3)
        out.append(f"{n_apps} apps {min(ts):.3f} ms")
    print(n_nodes, "nodes:", "  ".join(out))
    ctx.close()
