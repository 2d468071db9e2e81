"""Cold 1 000-application chains at 100 000 nodes x 3 zones (the reference's AZ-major order): minimal-fragmentation,
single-AZ minimal-fragmentation, single-az-tightly-pack — the size at which the LDS chain kernels sit next to the per-view masks
with a few dozen bytes to spare (tests/test_gpu_fullsize.py guards it).  Prints algo id and milliseconds per chain."""
import sys, time, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
n_nodes, nz = 100000, 3
w = wl.headline(n_nodes, 1000)
s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
zone = (wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(nz)).astype(np.uint32)
ctx.set_zones(zone)
order = wl.reference_node_order(s.avail, zone)
ctx.set_orders(order, order)
apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
for algo in (2, 5, 4):
    ctx.fit_batch(1, algo, apps)
    t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); print(algo, (time.perf_counter() - t0) * 1e3, flush=True)
