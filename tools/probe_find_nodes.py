"""findNodes of the failover reconciler (failover.go:412-436) for 200 stale applications in a row on the headline cluster (chained: one
wavefront, the working table mutated between requests), host entry point.  Run on the MI355X box."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
ctx = gangfit.Context(0)
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
for chained in (True, False):
    f = lambda: ctx.find_nodes(w.exe[:200], w.k[:200], chained=chained, want_adds=False)
    for _ in range(3):
        f()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"findNodes, 200 requests, chained={chained}: p50 {ts[10]:.3f} ms  p99 {ts[-1]:.3f} ms")
