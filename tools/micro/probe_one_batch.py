"""N launches of ONE independent batch (for rocprofv3 --pmc runs that must not mix workloads):
    python tools/micro/probe_one_batch.py <headline|config3|config4|congested> <algo> [launches]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl
name, algo = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
w = {"headline": lambda: wl.headline(10000, 1000), "config3": lambda: wl.config(3), "config4": lambda: wl.config(4),
     "congested": lambda: wl.headline(10000, 1000, congested=True)}[name]()
dev = torch.device("cuda:0")
s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
a2, tk = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
d_apps = torch.from_numpy(a2.view(np.uint8).copy()).to(dev); d_res = torch.zeros(len(a2) * 16, dtype=torch.uint8, device=dev); d_exec = torch.zeros(tk + 1, dtype=torch.int32, device=dev)
for _ in range(n):
    ctx.fit_batch_dev(0, algo, len(a2), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), tk)
torch.cuda.synchronize()
print(name, algo, n, "launches of", len(a2), "apps, total_k", tk)
