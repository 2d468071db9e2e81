"""Where does a config-5 Filter (100 000 nodes, 999 earlier drivers + 1) spend its time?  Run on the MI355X box."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl

def t(f, n=5):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]

for n_nodes in (10000, 100000):
    w = wl.config(5, n_nodes=n_nodes)
    s = w.snapshot
    ctx = gangfit.Context(0, options={"chain_cache": 0})
    print(n_nodes, "set_snapshot ms", t(lambda: ctx.set_snapshot(s.avail, s.sched)))
    print(n_nodes, "set_orders   ms", t(lambda: ctx.set_orders(s.driver_order, s.exec_order)))
    t0 = time.perf_counter(); o = wl.reference_node_order(s.avail); print(n_nodes, "numpy lexsort ms", (time.perf_counter() - t0) * 1e3)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    for algo in (0, 1, 4):
        if algo == 4:
            ctx.set_snapshot(s.avail, s.sched); ctx.set_zones(np.zeros(n_nodes, dtype=np.uint32)); ctx.set_orders(s.driver_order, s.exec_order)
        print(n_nodes, "fifo chain algo", algo, "ms", t(lambda: ctx.fit_batch(1, algo, apps), 3))
        print(n_nodes, "independent algo", algo, "ms", t(lambda: ctx.fit_batch(0, algo, apps), 3))
    ctx.close()
