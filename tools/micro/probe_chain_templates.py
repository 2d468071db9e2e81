"""How much of the plain chain's time is the price of MANY request shapes (stale index bits, cold rows)?  The headline chain
with its 41 shapes, with 4 templates, and with one template (the index is then exact: a commit re-tests the app's own shapes)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
rng = np.random.default_rng(1)
for name, nt in (("41 shapes (headline)", 0), ("4 templates", 4), ("1 template", 1)):
    drv, exe = w.drv, w.exe
    if nt:
        t = rng.integers(0, nt, size=len(w.k)) * 37 % len(w.k)
        drv, exe = w.drv[t], w.exe[t]
    apps = gangfit.make_apps(drv, exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    ctx.fit_batch(1, 0, apps)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); r = ctx.fit_batch(1, 0, apps); ts.append((time.perf_counter() - t0) * 1e3)
    ctx.scan_stats(enable=True, reset=True)
    ctx.fit_batch(1, 0, apps)
    x, d = ctx.scan_stats(enable=False)
    print(f"{name:22s} {min(ts):.3f} ms  feasible {int(r.results['has_capacity'].sum())}  exec slots/app {x / 1000:.0f} driver slots/app {d / 1000:.0f} "
          f"phases(cycles/app stage|driver|exec|slow|commit|visits) {[p // 1000 for p in ctx.last_fifo_phases]}")
