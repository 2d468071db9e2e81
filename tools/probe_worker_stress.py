import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch, gangfit
from gangfit import workloads as wl
IND, TIGHT = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_ALGO_TIGHTLY_PACK
dev = torch.device("cuda:0")
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
outs = [(torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev), torch.zeros(total_k + 1, dtype=torch.int32, device=dev)) for _ in range(8)]
for sets, idle in ((2, 200), (2, 10), (3, 200), (3, 10)):
    ctx = gangfit.Context(0, options={"worker_sets": sets, "worker_idle_us": idle})
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    K = 3000
    arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), outs[i % 8][0].data_ptr(), outs[i % 8][1].data_ptr(), total_k) for i in range(K)])
    try:
        for rep in range(3):
            t0 = time.perf_counter()
            first = ctx.worker_submit_prepared(TIGHT, arr)
            ctx.worker_wait(first, K)
            print(sets, idle, rep, "ok %.2f us/batch" % ((time.perf_counter() - t0) / K * 1e6), ctx.worker_stats(), flush=True)
    except Exception as e:
        print(sets, idle, "FAIL", e, ctx.worker_stats(), flush=True)
    ctx.close()
