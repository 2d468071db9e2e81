"""Kernel time of the independent batch against the number of applications (= wavefronts) per launch, device-resident
entry point, HIP events over 200 back-to-back launches: separates the fixed cost of a launch from the per-wavefront work."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 4000)
if os.environ.get("K"):
    w.k[:] = int(os.environ["K"])  # every gang the same size
s = w.snapshot
dev = torch.device("cuda:0")
with gangfit.Context(0, options={"chain_cache": 0}) as ctx:
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    stream = torch.cuda.current_stream().cuda_stream
    for n in (1, 4, 64, 256, 1000, 2000, 4000):
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv[:n], w.exe[:n], w.k[:n], np.ones(n, dtype=np.uint32)))
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        for algo in (0, 1):
            for _ in range(20):
                ctx.fit_batch_dev(0, algo, n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
            torch.cuda.synchronize()
            ctx.timer_begin(stream)
            for _ in range(200):
                ctx.fit_batch_dev(0, algo, n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
            ms = ctx.timer_end()
            print("apps", n, "algo", algo, "us per launch", round(ms * 1000 / 200, 3))
