"""Config 5 (100 000 nodes): cold chains over rotated heads, and the FIRST chain on a just-installed snapshot (cold caches)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w = wl.config(5)
s = w.snapshot
apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
ref = ctx.fit_batch(1, 0, apps)
cold, first = [], []
for i in range(40):
    q = np.roll(apps, -i)
    t0 = time.perf_counter(); ctx.fit_batch(1, 0, q); cold.append((time.perf_counter() - t0) * 1e3)
for i in range(12):
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    q = np.roll(apps, -i)
    t0 = time.perf_counter(); ctx.fit_batch(1, 0, q); first.append((time.perf_counter() - t0) * 1e3)
again = ctx.fit_batch(1, 0, apps)
print(f"lib {os.environ.get('GANGFIT_LIB', 'default')}: cold chain p50 {np.median(cold):.3f} p99 {np.percentile(cold, 99):.3f} ms; first chain on a fresh snapshot p50 {np.median(first):.3f} ms; "
      f"answers stable {bool(np.array_equal(again.results, ref.results) and np.array_equal(again.exec_nodes, ref.exec_nodes))} crc {int(np.bitwise_xor.reduce(ref.exec_nodes.astype(np.uint64) * np.arange(1, len(ref.exec_nodes) + 1, dtype=np.uint64))) & 0xFFFFFFFF:08x}")
