"""minimal-fragmentation, independent batch at the headline size: passes over the order per application (scan counters) and the
batch's time by gang size — one launch of only the applications with K <= kmax.  Run on the MI355X box."""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
dev = torch.device("cuda", 0)
ctx = gangfit.Context(0)
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
IND, MF = gangfit.GF_MODE_INDEPENDENT, 2
print("K quantiles (50/90/99/max):", [int(np.quantile(w.k, q)) for q in (0.5, 0.9, 0.99, 1.0)])
for kmax in (8, 16, 32, 64, 128, 100000):
    sel = np.nonzero(w.k <= kmax)[0]
    apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv[sel], w.exe[sel], w.k[sel], w.flags[sel]))
    d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
    d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
    f = lambda: ctx.fit_batch_dev(IND, MF, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=0)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ctx.scan_stats(enable=True, reset=True)
    f(); torch.cuda.synchronize()
    xvis, dvis = ctx.scan_stats(enable=False, reset=True)
    print(f"K <= {kmax:6d}: {len(apps):5d} applications, batch {np.median(ts):8.1f} us, passes over the executor order per application {xvis / len(apps) / len(s.exec_order):5.2f}")
