import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
IND, TIGHT = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_ALGO_TIGHTLY_PACK
for trial in range(6):
    ctx = gangfit.Context(0, options={"worker_idle_us": 500000})
    w = wl.headline(5000, 300, seed=0xFEED)
    s = w.snapshot
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    if trial % 2 == 0:
        ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, TIGHT, apps)
    a = ctx.worker_fit(TIGHT, apps)
    st1 = ctx.worker_stats(); g = ctx.worker_geometry()
    t0 = time.perf_counter()
    b = ctx.fit_batch(IND, TIGHT, apps)
    dt = time.perf_counter() - t0
    st2 = ctx.worker_stats()
    time.sleep(0.01)
    st3 = ctx.worker_stats()
    print(trial, f"fit_batch took {dt*1e3:.2f} ms", "after worker_fit", st1, g, "after fit_batch", st2, "10 ms later", st3, flush=True)
    ctx.close()
