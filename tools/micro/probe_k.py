"""Independent tightly-pack launch (10k nodes x 1000 apps) against the executor-count distribution: is the launch bound by
its slowest wavefront (the largest gang visits the most chunks, one dependent round trip each) or by the common case?"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch
import gangfit
from gangfit import workloads as wl
w = wl.headline(10000, 1000)
s = w.snapshot
dev = torch.device("cuda:0")
with gangfit.Context(0, options={"chain_cache": 0}) as ctx:
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    stream = torch.cuda.current_stream().cuda_stream
    print("K distribution: mean", w.k.mean(), "p99", np.percentile(w.k, 99), "max", w.k.max())
    for name, k in (("as is", w.k), ("all 1", np.ones_like(w.k)), ("all 12", np.full_like(w.k, 12)),
                    ("clamped to 40", np.minimum(w.k, 40)), ("all 100", np.full_like(w.k, 100))):
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, k, np.ones(len(k), dtype=np.uint32)))
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        for algo in (0, 1):
            for _ in range(20):
                ctx.fit_batch_dev(0, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
            torch.cuda.synchronize()
            ctx.timer_begin(stream)
            for _ in range(300):
                ctx.fit_batch_dev(0, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
            ms = ctx.timer_end()
            ctx.scan_stats(enable=True, reset=True)
            ctx.fit_batch_dev(0, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
            torch.cuda.synchronize()
            xv, dv = ctx.scan_stats(enable=False, reset=True)
            print(f"K {name:14s} algo {algo}: {ms * 1000 / 300:6.2f} us per launch, exec slots visited per app {xv / len(apps):7.1f}, driver {dv / len(apps):6.1f}")
