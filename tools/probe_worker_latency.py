import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl, _native as N
TIGHT = gangfit.GF_ALGO_TIGHTLY_PACK
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
ctx = gangfit.Context(0)
ctx.set_snapshot(s.avail, s.sched); ctx.set_orders(s.driver_order, s.exec_order)
happs, htotal = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
for n in (1000, 100, 1):
    a = np.ascontiguousarray(happs[:n]); tk = int(a["k"].sum())
    res = np.zeros(n, dtype=N.RESULT_DTYPE); ex = np.zeros(tk + 1, dtype=np.uint32)
    lib, h = ctx._lib, ctx._h
    pa, pr, pe = N.ptr(a), N.ptr(res), N.ptr(ex)
    for name, fn in (("worker", lambda: lib.gf_worker_fit(h, TIGHT, n, pa, pr, pe, tk)), ("launch", lambda: lib.gf_fit_batch(h, 0, TIGHT, n, pa, pr, pe, tk, None))):
        for _ in range(20): fn()
        st0 = ctx.worker_stats()
        ts = []
        for _ in range(300):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(n, name, "p50 %.1f us p90 %.1f" % (ts[150] * 1e6, ts[270] * 1e6), "launches", ctx.worker_stats()["launches"] - st0["launches"], flush=True)
ctx.close()
