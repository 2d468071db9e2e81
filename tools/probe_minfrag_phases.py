"""Where a minimal-fragmentation decision of the independent batch spends its cycles (a -DGF_MF_PROBE build through GANGFIT_LIB)."""
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
for kmax in (8, 100000):
    sel = np.nonzero(w.k <= kmax)[0]
    apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv[sel], w.exe[sel], w.k[sel], w.flags[sel]))
    d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
    d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
    f = lambda: ctx.fit_batch_dev(gangfit.GF_MODE_INDEPENDENT, 2, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=0)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e6
    ctx.scan_stats(enable=True, reset=True)
    f(); torch.cuda.synchronize()
    ctx.scan_stats(enable=False, reset=False)
    ph = ctx.last_fifo_phases
    n = len(apps)
    print(f"K <= {kmax}: {n} applications, batch {dt:.1f} us; cycles per application: whole decision {ph[0] / n:.0f} | scaled requests {ph[1] / n:.0f} | pass 1 {ph[2] / n:.0f} | level plan {ph[3] / n:.0f} | pass 2 {ph[4] / n:.0f} | histogram form used by wave {[(ph[5] >> (16 * i)) & 0xFFFF for i in range(4)]} of {n}")
