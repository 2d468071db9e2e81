"""Where the time of a throughput-sized independent batch goes: config 3 (10 000 x 10 000) with the gangs cut down.
    full / K<=8 / K=1 / K=0 (driver search only) / no-gpu requests, tightly-pack and distribute-evenly; us per launch."""
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
def run(label, drv, exe, k):
    out = []
    for algo in (0, 1):
        a2, tk = gangfit.with_offsets(gangfit.make_apps(drv, exe, k, w.flags))
        d_apps = torch.from_numpy(a2.view(np.uint8).copy()).to(dev); d_res = torch.zeros(len(a2) * 16, dtype=torch.uint8, device=dev); d_exec = torch.zeros(tk + 1, dtype=torch.int32, device=dev)
        f = lambda: ctx.fit_batch_dev(0, algo, len(a2), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), tk)
        for _ in range(30): f()
        torch.cuda.synchronize(); ctx.timer_begin()
        for _ in range(300): f()
        ms = ctx.timer_end()
        res = np.frombuffer(d_res.cpu().numpy().tobytes(), dtype=gangfit.RESULT_DTYPE) if hasattr(gangfit, "RESULT_DTYPE") else None
        out.append(f"{'tight' if algo == 0 else 'even'} {ms / 300 * 1e3:.2f}us")
    print(f"{label:28s}", "  ".join(out), f" mean K {k.mean():.1f}")
k = w.k
run("full", w.drv, w.exe, k)
run("K<=8", w.drv, w.exe, np.minimum(k, 8).astype(np.int32))
run("K=1", w.drv, w.exe, np.minimum(k, 1).astype(np.int32))
run("K=0", w.drv, w.exe, np.zeros_like(k))
e = w.exe.copy(); e[:, 2] = 0
run("no gpu request", w.drv, e, k)
run("1000 apps", w.drv[:1000], w.exe[:1000], k[:1000]) if False else None
