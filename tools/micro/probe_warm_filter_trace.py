"""200 warm Filters (creation-order heads on the unchanged snapshot: every chain resumes from the previous one's checkpoints) —
run under `rocprofv3 --kernel-trace --stats` to see what a resumed chain is on the device."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import ctypes as C
import gangfit
from gangfit import _native as N
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
ctx = gangfit.Context(0)
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
apps, total = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
res = np.zeros(len(apps), dtype=N.RESULT_DTYPE)
ex = np.zeros(total + 1, dtype=np.uint32)
failed = C.c_int32(0)
lib, h = ctx._lib, ctx._h
ts = []
for rep in range(3):
    for j in range(744, 1000):
        q = apps[: j + 1]
        t0 = time.perf_counter()
        rc = lib.gf_fit_batch(h, 1, 0, len(q), N.ptr(q), N.ptr(res), N.ptr(ex), total, C.byref(failed))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
ts = np.array(ts[256:]) * 1e6
print(f"warm Filter: p50 {np.median(ts):.1f} us p99 {np.percentile(ts, 99):.1f} us over {len(ts)} calls; cache {ctx.chain_cache_stats()}")
