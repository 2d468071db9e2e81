"""Which applications make a launch slow?  One launch per distinct executor request of the headline workload (64 copies
of the same application, K fixed), tightly-pack, HIP events + visited slots."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl
K = int(os.environ.get("K", "60"))
w = wl.headline(10000, 1000)
s = w.snapshot
dev = torch.device("cuda:0")
shapes = np.unique(w.exe, axis=0)
with gangfit.Context(0, options={"chain_cache": 0}) as ctx:
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    stream = torch.cuda.current_stream().cuda_stream
    for e in shapes:
        n = 64
        drv = np.tile(w.drv[0], (n, 1)); exe = np.tile(e, (n, 1)); k = np.full(n, K, dtype=np.int32)
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(drv, exe, k, np.ones(n, dtype=np.uint32)))
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        for _ in range(5):
            ctx.fit_batch_dev(0, 0, n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
        torch.cuda.synchronize()
        ctx.timer_begin(stream)
        for _ in range(50):
            ctx.fit_batch_dev(0, 0, n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
        ms = ctx.timer_end()
        ctx.scan_stats(enable=True, reset=True)
        ctx.fit_batch_dev(0, 0, n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
        torch.cuda.synchronize()
        xv, dv = ctx.scan_stats(enable=False, reset=True)
        res = d_res.cpu().numpy().view(gangfit._native.RESULT_DTYPE)
        print(f"exe {tuple(int(v) for v in e)}: {ms * 1000 / 50:6.2f} us, feasible {int(res['has_capacity'][0])}, "
              f"exec slots / app {xv / n:8.1f}, driver slots / app {dv / n:7.1f}")
