"""How fast can tickets be SUPPLIED to the resident worker?  Streams of K tickets of n applications each: with n small the wavefronts
have nearly nothing to do and the time per ticket is the supply chain's (host posting, the leader's relay over the host link, the
completion words back).  Run on the MI355X box."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl

TIGHT = gangfit.GF_ALGO_TIGHTLY_PACK
dev = torch.device("cuda:0")
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
ctx = gangfit.Context(0)
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
for n in (1, 16, 336, 1000):
    apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv[:n], w.exe[:n], w.k[:n], w.flags[:n]))
    d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
    outs = [(torch.zeros(n * 16, dtype=torch.uint8, device=dev), torch.zeros(total_k + 1, dtype=torch.int32, device=dev)) for _ in range(8)]
    for K in (2000,):
        arr = ctx.worker_batches([(n, d_apps.data_ptr(), outs[i % 8][0].data_ptr(), outs[i % 8][1].data_ptr(), total_k) for i in range(K)], leave_after=True)
        ts = []
        for rep in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.worker_submit_prepared(TIGHT, arr)
            ctx.worker_stop()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{n:5d} applications per ticket, K = {K}: {ts[len(ts) // 2] / K * 1e6:6.2f} us per ticket  (geometry {ctx.worker_geometry()})", flush=True)
ctx.close()
