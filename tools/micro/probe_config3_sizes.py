"""Config 3 by batch size: is the second round of wavefronts what a 10 000-application launch pays for?  (fit_independent_kernel
sits at seven wavefronts per SIMD: 7 168 resident of 10 000.)  us per launch, tightly-pack / distribute-evenly."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl
dev = torch.device("cuda:0")
w = wl.config(3)
s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
for n in (1000, 2000, 4096, 6144, 7168, 8192, 9216, 10000):
    out = []
    for algo in (0, 1):
        a2, tk = gangfit.with_offsets(gangfit.make_apps(w.drv[:n], w.exe[:n], w.k[:n], w.flags[:n]))
        d_apps = torch.from_numpy(a2.view(np.uint8).copy()).to(dev)
        d_res = torch.zeros(len(a2) * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(tk + 1, dtype=torch.int32, device=dev)
        f = lambda: ctx.fit_batch_dev(0, algo, len(a2), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), tk)
        for _ in range(30):
            f()
        torch.cuda.synchronize()
        ctx.timer_begin()
        for _ in range(300):
            f()
        ms = ctx.timer_end()
        out.append(f"{'tight' if algo == 0 else 'even'} {ms / 300 * 1e3:6.2f} us = {n / (ms / 300 * 1e-3) / 1e9:5.3f} G/s")
    print(f"{n:6d} applications  ", "   ".join(out), flush=True)
