import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch, gangfit
from gangfit import workloads as wl
TIGHT = gangfit.GF_ALGO_TIGHTLY_PACK
dev = torch.device("cuda:0")
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
outs = [(torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev), torch.zeros(total_k + 1, dtype=torch.int32, device=dev)) for _ in range(8)]
torch.cuda.synchronize()
junk = []
for trial in range(10):
    ctx = gangfit.Context(0, options={"worker_sets": 3})
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    K = 2000
    arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), outs[i % 8][0].data_ptr(), outs[i % 8][1].data_ptr(), total_k) for i in range(K)])
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = ctx.worker_submit_prepared(TIGHT, arr)
        ctx.worker_stop()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    print(trial, " ".join("%.2f" % t for t in ts), flush=True)
    ctx.close()
    junk.append(torch.zeros(int(np.random.default_rng(trial).integers(1, 64)) << 16, dtype=torch.uint8, device=dev))  # shift later allocations
