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
def run(name, drv, exe, k):
    apps = gangfit.make_apps(drv, exe, k, np.ones(len(k), dtype=np.uint32))
    ctx.fit_batch(1, 0, apps)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); o = ctx.fit_batch(1, 0, apps); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:40s} {min(ts):7.3f} ms  feasible {o.results['has_capacity'].mean():.2f}")
run("headline", w.drv, w.exe, w.k)
run("K = 0 (driver only)", w.drv, w.exe, np.zeros_like(w.k))
run("K = 1", w.drv, w.exe, np.ones_like(w.k))
one = np.tile(w.exe[:1], (1000, 1)); oned = np.tile(w.drv[:1], (1000, 1))
run("one shape, headline K", oned, one, w.k)
run("one shape, K = 1", oned, one, np.ones_like(w.k))
small = one.copy(); small[:, 0] = 1000; small[:, 1] = 1 << 30; small[:, 2] = 0
run("tiny executors (1 cpu, 1 GiB), K = 1", oned, small, np.ones_like(w.k))
run("tiny executors, headline K", oned, small, w.k)
