"""Resident worker: how many sets of wavefronts, how many workgroups each?  A set with fewer than 63 workgroups (16 wavefronts =
16 applications each) gives some wavefronts two or more applications of a ticket, one after the other — the ticket takes longer,
but more tickets are in flight on the same 240 CUs.  Window = what bench.py times: K tickets posted, worker launched, served,
stopped, device synchronised.      python tools/probe_worker_sets.py [sets:blocks ...]        (run on the MI355X box)"""
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
apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
NOUT = 8
outs = [(torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev), torch.zeros(total_k + 1, dtype=torch.int32, device=dev)) for _ in range(NOUT)]
configs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(3, 64), (7, 32), (11, 21), (14, 16), (5, 47)]


def med(x):
    x = sorted(x)
    return x[len(x) // 2]


ref = None
for sets, bps in configs:
    opts = {"worker_sets": sets, "worker_blocks_per_set": bps}
    ctx = gangfit.Context(0, options=opts)
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    line = f"sets {sets:2d} x {bps:2d} workgroups ({1 + sets * bps:3d} CUs)"
    for K in (20, 200, 2000):
        arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), outs[i % NOUT][0].data_ptr(), outs[i % NOUT][1].data_ptr(), total_k) for i in range(K)], leave_after=os.environ.get("PROBE_LEAVE_AFTER", "1") != "0")
        walls, kern = [], []
        for rep in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.worker_submit_prepared(TIGHT, arr)
            ctx.worker_stop()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if rep >= 2:
                walls.append(t1 - t0)
                ms, n = ctx.worker_kernel_time()
                if n == K:
                    kern.append(ms / n * 1e3)
        line += f" | K {K:4d}: {med(walls) * 1e6:8.1f} us = {med(walls) / K * 1e6:5.2f} us/ticket {len(apps) * K / med(walls) / 1e6:6.1f} M/s, kernel {med(kern) if kern else float('nan'):5.2f} us/ticket"
    res = (outs[0][0].cpu().numpy().tobytes(), outs[0][1].cpu().numpy().tobytes())
    if ref is None:  # the launch path's answer (fit_independent_kernel) on the same batch
        l_res, l_exec = torch.zeros_like(outs[0][0]), torch.zeros_like(outs[0][1])
        ctx.fit_batch_dev(gangfit.GF_MODE_INDEPENDENT, TIGHT, len(apps), d_apps.data_ptr(), l_res.data_ptr(), l_exec.data_ptr(), total_k, stream=0)
        torch.cuda.synchronize()
        ref = (l_res.cpu().numpy().tobytes(), l_exec.cpu().numpy().tobytes())
    print(line + f" | answers == launch path: {res == ref}", flush=True)
    ctx.close()
