"""Wavefront timeline of one independent-batch launch (headline workload): when the first wavefront starts, when the last
one ends and how long a wavefront lives on average, from s_memrealtime (10 ns ticks) — next to the HIP-event duration."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
with gangfit.Context(0) as ctx:
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    for algo in (0, 1):
        for _ in range(3):
            ctx.fit_batch(0, algo, apps)
        ctx.timer_begin(); ctx.fit_batch(0, algo, apps); ms = ctx.timer_end()
        for rep in range(3):
            ctx.scan_stats(enable=True, reset=True)
            ctx.fit_batch(0, algo, apps)
            ctx.scan_stats(enable=False)
            ph = ctx.last_fifo_phases
            span = ph[1] - ((~ph[2]) & 0xFFFFFFFFFFFFFFFF)
            print("algo", algo, "host-entry stream ms", round(ms, 4), "first start -> last end us", span / 100.0,
                  "mean wavefront life us", ph[0] / len(apps) / 100.0)
