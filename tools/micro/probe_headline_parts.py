"""What bounds one headline launch now?  Kernel time (HIP events, 400 launches) of the headline batch and of subsets of it."""
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
    print("launch floor us", ctx.launch_floor(stream, 400))
    def run(name, sel, algo=0, kover=None):
        drv, exe, k = w.drv[sel], w.exe[sel], (w.k[sel] if kover is None else np.full(int(np.sum(sel)) if sel.dtype == bool else len(sel), kover, dtype=np.int32))
        n = len(k)
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(drv, exe, k, np.ones(n, dtype=np.uint32)))
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        f = lambda: ctx.fit_batch_dev(0, algo, n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=stream)
        for _ in range(20): f()
        torch.cuda.synchronize()
        ctx.timer_begin(stream)
        for _ in range(400): f()
        ms = ctx.timer_end()
        ctx.scan_stats(enable=True, reset=True); f(); torch.cuda.synchronize()
        xv, dv = ctx.scan_stats(enable=False, reset=True)
        print(f"{name:44s} n={n:5d} {ms * 1000 / 400:6.2f} us   exec slots/app {xv / n:7.1f}  driver slots/app {dv / n:6.1f}  maxK {int(k.max()) if n else 0}")
    allm = np.ones(1000, dtype=bool)
    gpu = w.exe[:, 2] > 0
    run("headline", allm)
    run("headline distribute-evenly", allm, algo=1)
    run("no gpu gangs", ~gpu)
    run("gpu gangs only", gpu)
    run("K <= 12", w.k <= 12)
    run("K > 40", w.k > 40)
    run("K > 40, no gpu", (w.k > 40) & ~gpu)
    run("one app", np.arange(1))
    run("64 apps", np.arange(64))
    run("all, K = 1", allm, kover=1)
    big = w.exe[:, 0] >= 8000
    run("8-core executors", big)
    run("32 GiB executors", w.exe[:, 1] >= 32 * wl.GIB)
