"""Latency and phase profile (wave 0 shader cycles per app) of the minimal-fragmentation chains.
    python tools/probe_minfrag.py [azmajor]"""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
azmajor = len(sys.argv) > 1 and sys.argv[1] == "azmajor"
for n_nodes, nz in ((10000, 1), (10000, 3), (100000, 3)):
    w = wl.headline(n_nodes, 1000)
    s = w.snapshot
    ctx = gangfit.Context(0, options={"chain_cache": 0})
    ctx.set_snapshot(s.avail, s.sched)
    zone = (wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(nz)).astype(np.uint32)
    ctx.set_zones(zone)
    order = wl.reference_node_order(s.avail, zone) if azmajor else s.exec_order
    ctx.set_orders(order, order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    for algo in (2, 5):
        ctx.fit_batch(1, algo, apps)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
        ctx.scan_stats(enable=True, reset=True)
        ctx.fit_batch(1, algo, apps)
        xvis, _ = ctx.scan_stats(enable=False)
        print("nodes", n_nodes, "zones", nz, "azmajor" if azmajor else "interleaved", "algo", algo, "fifo ms", round(min(ts), 3),
              "phases(cycles/app: stage+fill | decide | efficiency | barrier | select+commit | patch+barrier)",
              [p // 1000 for p in ctx.last_fifo_phases], "walk batches of 2048 slots per app (all views)", round(xvis / 2048 / 1000, 2))
    ctx.close()
