"""Independent batches on several VIEWS of one context (gf_ctx_view: own stream, own scratch) at once: K headline batches per
window split over B views, each view's share recorded as one graph; a window = B graph launches + a synchronise.  A 1 000-application
batch is ~1 000 wavefronts on 7 000 wavefront slots and latency-bound, so batches on different hardware queues should overlap.
Measured (profiles/r4_parallel_views.txt): at the driver's K=20 a window is B graph launches + a synchronise and the
resident worker wins (3.5 us/step); in 200-step windows 4 views reach 1.8 us/step.  ONE recording with B forked branches (event
fork/join inside the capture) was also tried: the runtime's graph executor puts its own cross-queue markers between the branches
and it is slower than a plain one-stream graph (6.1 us/step at 2-4 branches) -- that entry point was not kept.
Run on the MI355X box:  python tools/probe_parallel_graphs.py"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import gangfit
from gangfit import workloads as wl

dev = torch.device("cuda:0")
w = wl.headline(10000, 1000, seed=0x5EED0010)
s = w.snapshot
apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
ctx = gangfit.Context(0)
ctx.set_snapshot(s.avail, s.sched)
ctx.set_orders(s.driver_order, s.exec_order)
ref_r = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
ref_e = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
ctx.fit_batch_dev(0, 0, len(apps), d_apps.data_ptr(), ref_r.data_ptr(), ref_e.data_ptr(), total_k)
torch.cuda.synchronize()


def med(x):
    x = sorted(x)
    return x[len(x) // 2]


for B in (1, 2, 3, 4, 6, 8):
    views = [ctx] + [ctx.view() for _ in range(B - 1)]
    outs = [(torch.zeros_like(ref_r), torch.zeros_like(ref_e)) for _ in range(B)]
    for K in (20, 200):
        graphs = []
        for b, v in enumerate(views):
            n = K // B + (1 if b < K % B else 0)
            f = lambda v=v, b=b: v.fit_batch_dev(0, 0, len(apps), d_apps.data_ptr(), outs[b][0].data_ptr(), outs[b][1].data_ptr(), total_k)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            v.graph_begin(0)
            for _ in range(n):
                f()
            graphs.append(v.graph_end(0))
        walls = []
        for rep in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for v, g in zip(views, graphs):
                v.graph_launch(g, 0)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        same = all(bool(torch.equal(o[0], ref_r)) and bool(torch.equal(o[1], ref_e)) for o in outs)
        wm = med(walls[2:])
        print(f"branches {B} K {K}: window {wm * 1e6:7.1f} us = {wm / K * 1e6:5.2f} us/batch = {len(apps) * K / wm / 1e6:6.1f} M decisions/s  same={same}", flush=True)
        for v, g in zip(views, graphs):
            v.graph_destroy(g)
    for v in views[1:]:
        v.close()
ctx.close()
