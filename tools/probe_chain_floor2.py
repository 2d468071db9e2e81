import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
for name, k in (("headline", w.k), ("K=0", np.zeros_like(w.k))):
    apps = gangfit.make_apps(w.drv, w.exe, k, np.ones(len(k), dtype=np.uint32))
    ctx.fit_batch(1, 0, apps)
    ctx.scan_stats(enable=True, reset=True)
    ctx.fit_batch(1, 0, apps)
    ctx.scan_stats(enable=False)
    cyc, ticks = ctx.last_fifo_clock
    print(name, "cycles/app", cyc // 1000, "phases stage|driver|scan|slow|commit|steps", [p / 1000 for p in ctx.last_fifo_phases])
