"""Chain latencies (host entry, best of 7) for every packer + the headline kernel time, for build-variant comparisons."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
zone3 = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
ctx = gangfit.Context(0, options={"chain_cache": 0})  # every chain replays: these probes time the kernels
ctx.set_snapshot(s.avail, s.sched)
ctx.set_zones(zone3)
ctx.set_orders(s.driver_order, s.exec_order)
out = []
for algo, name, reps in ((0, "tight", 7), (1, "even", 7), (4, "saz-tight", 4), (2, "minfrag", 3), (5, "saz-minfrag", 2)):
    ctx.fit_batch(1, algo, apps)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
    out.append(f"{name} {min(ts):.3f}")
# the reference's own order is AZ-major (nodesorting.go:82-122): every zone is a contiguous range of the priority order
zo = wl.reference_node_order(s.avail, zone3)
ctx.set_orders(zo, zo)
for algo, name, reps in ((3, "azmajor:az-aware", 4), (4, "azmajor:saz-tight", 4), (5, "azmajor:saz-minfrag", 2)):
    ctx.fit_batch(1, algo, apps)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
    out.append(f"{name} {min(ts):.3f}")
ctx.set_orders(s.driver_order, s.exec_order)
a2, tk = gangfit.with_offsets(apps)
import torch
dev = torch.device("cuda:0")
d_apps = torch.from_numpy(a2.view(np.uint8).copy()).to(dev); d_res = torch.zeros(len(a2) * 16, dtype=torch.uint8, device=dev); d_exec = torch.zeros(tk + 1, dtype=torch.int32, device=dev)
f = lambda: ctx.fit_batch_dev(0, 0, len(a2), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), tk)
for _ in range(50): f()
torch.cuda.synchronize(); ctx.timer_begin()
for _ in range(1000): f()
ms = ctx.timer_end()
out.append(f"headline_us {ms:.3f}")
print("  ".join(out))
