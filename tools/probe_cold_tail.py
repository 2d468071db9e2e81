import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl, _native as N
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
apps0 = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
for opts in ({}, {"zero_copy": 0}):
    ctx = gangfit.Context(0, options=opts)
    ctx.set_snapshot(s.avail, s.sched); ctx.set_orders(s.driver_order, s.exec_order)
    lib, h = ctx._lib, ctx._h
    n = len(apps0); tk = int(apps0["k"].sum())
    res = np.zeros(n, dtype=N.RESULT_DTYPE); ex = np.zeros(tk + 1, dtype=np.uint32)
    import ctypes as C
    failed = C.c_int32(-1)
    rolled = [np.ascontiguousarray(np.roll(apps0, -i)) for i in range(400)]
    ts = []
    for i in range(400):
        pa = N.ptr(rolled[i])
        t0 = time.perf_counter()
        rc = lib.gf_fit_batch(h, 1, 0, n, pa, N.ptr(res), N.ptr(ex), tk, C.byref(failed))
        ts.append(time.perf_counter() - t0)
    a = np.array(ts[5:]) * 1e3
    print(opts, "p50 %.3f p90 %.3f p99 %.3f max %.3f; >3ms: %d at %s" % (np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max(), (a > 3).sum(), np.nonzero(a > 3)[0][:10]), flush=True)
    ctx.close()
