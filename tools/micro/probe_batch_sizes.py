"""Kernel time of the independent batch at several sizes (HIP events, 300 launches each) for build-variant comparisons."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl
dev = torch.device("cuda:0")
out = []
for name, w in (("headline", wl.headline(10000, 1000)), ("config3", wl.config(3)), ("config4", wl.config(4)), ("congested", wl.headline(10000, 1000, congested=True))):
    s = w.snapshot
    ctx = gangfit.Context(0, options={"chain_cache": 0})
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    for algo in (0, 1):
        a2, tk = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
        d_apps = torch.from_numpy(a2.view(np.uint8).copy()).to(dev); d_res = torch.zeros(len(a2) * 16, dtype=torch.uint8, device=dev); d_exec = torch.zeros(tk + 1, dtype=torch.int32, device=dev)
        f = lambda: ctx.fit_batch_dev(0, algo, len(a2), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), tk)
        for _ in range(30): f()
        torch.cuda.synchronize(); ctx.timer_begin()
        for _ in range(300): f()
        ms = ctx.timer_end()
        out.append(f"{name}/{'tight' if algo == 0 else 'even'} {ms / 300 * 1e3:.2f}us")
    ctx.close()
print("  ".join(out))
