"""Phase profile of the LDS-resident zone-aware FIFO chain (wave 0 shader cycles per phase)."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = wl.headline(n_nodes, 1000)
s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
zone = (wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(nz)).astype(np.uint32)
ctx.set_zones(zone)
if len(sys.argv) > 3 and sys.argv[3] == "azmajor":  # the reference's own order: zones are contiguous ranges (nodesorting.go:82-122)
    zo = wl.reference_node_order(s.avail, zone)
    ctx.set_orders(zo, zo)
else:
    ctx.set_orders(s.driver_order, s.exec_order)
apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
for algo in (4, 3):
    ctx.fit_batch(1, algo, apps)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
    ctx.scan_stats(enable=True, reset=True)
    ctx.fit_batch(1, algo, apps)
    ctx.scan_stats(enable=False)
    cyc, ticks = ctx.last_fifo_clock
    print("nodes", n_nodes, "zones", nz, "algo", algo, "ms", min(ts), "shader MHz", cyc / max(ticks, 1) * 100,
          "phases(cycles/app)", [p // 1000 for p in ctx.last_fifo_phases])
