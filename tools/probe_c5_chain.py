"""Config 5 (100 000 nodes): the chain with and without the chain cache recording checkpoints, and resumed.  Run on the MI355X box."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl

w = wl.config(5)
s = w.snapshot
apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
for algo, name in ((0, "tightly-pack"), (4, "single-az-tightly-pack"), (2, "minimal-fragmentation")):
    for cache in (0, 1):
        ctx = gangfit.Context(0, options={"chain_cache": cache})
        ctx.set_snapshot(s.avail, s.sched)
        if algo == 4:
            ctx.set_zones((np.arange(len(s.avail)) % 3).astype(np.uint32))
        ctx.set_orders(s.driver_order, s.exec_order)
        cold, warm = [], []
        for i in range(6):
            q = np.roll(apps, -i)
            t0 = time.perf_counter(); ctx.fit_batch(1, algo, q); cold.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter(); ctx.fit_batch(1, algo, q); warm.append((time.perf_counter() - t0) * 1e3)
        print(f"{name}: chain_cache={cache} cold {min(cold[1:]):.3f} ms, same queue again {min(warm[1:]):.3f} ms, stats {ctx.chain_cache_stats()}")
        ctx.close()
