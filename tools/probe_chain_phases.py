"""Phase profile (wave 0 shader cycles per application) of the zone-aware and minimal-fragmentation FIFO chains."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
zone3 = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
for nz in (1, 3):
    ctx = gangfit.Context(0, options={"chain_cache": 0})
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_zones(zone3 if nz == 3 else np.zeros(len(s.avail), dtype=np.uint32))
    ctx.set_orders(s.driver_order, s.exec_order)
    for algo, name in ((4, "single-az-tightly-pack"), (2, "minimal-fragmentation"), (5, "single-az-minimal-fragmentation")):
        ctx.fit_batch(1, algo, apps)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); o = ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
        ctx.scan_stats(enable=True, reset=True)
        ctx.fit_batch(1, algo, apps)
        ctx.scan_stats(enable=False, reset=True)
        cyc, ticks = ctx.last_fifo_clock
        ph = ctx.last_fifo_phases
        n = len(apps)
        print(f"zones {nz} {name:34s} {min(ts):7.3f} ms  cycles/app {cyc / n:7.0f}  phases/app " + " ".join(f"{p / n:7.0f}" for p in ph),
              "feasible", int(o.results['has_capacity'].sum()))
    ctx.close()
