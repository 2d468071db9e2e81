"""Where an independent batch's time goes (default: the one-launch zone kernel, fit_zoned_fused_kernel; any packer's name as an argument;
10 000 nodes x 1 000 applications, three zones, AZ-major order): the same batch with parts taken away — no gpu requests, one executor per gang, no executors, subsets
of the batch — device time between HIP events, with the plain packer beside it.  Run on the MI355X box."""
import os
import sys

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch  # noqa: E402

import gangfit  # noqa: E402
from gangfit import workloads as wl  # noqa: E402

n_nodes, n_apps, nz = 10000, 1000, 3
IND = gangfit.GF_MODE_INDEPENDENT
w = wl.headline(n_nodes, n_apps)
s = w.snapshot
zone = (wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(nz)).astype(np.uint32)
zorder = wl.reference_node_order(s.avail, zone)
if os.environ.get("PROBE_ORDER") == "scattered":  # every zone has candidates in every 64-chunk group of the order (no extender orders nodes so)
    zorder = s.exec_order
dev = torch.device("cuda", 0)


def variants():
    drv, exe, k, fl = w.drv.copy(), w.exe.copy(), w.k.copy(), w.flags.copy()
    yield "the batch", drv, exe, k, fl
    e2 = exe.copy()
    e2[:, 2] = 0
    yield "no gpu requests", drv, e2, k, fl
    yield "one executor per gang", drv, exe, np.minimum(k, 1), fl
    yield "no executors", drv, exe, np.zeros_like(k), fl
    g = exe[:, 2] > 0
    yield f"only the {int(g.sum())} gangs of gpu executors", drv[g], exe[g], k[g], fl[g]
    yield f"only the {int((~g).sum())} others", drv[~g], exe[~g], k[~g], fl[~g]
    big = k > 40
    yield f"only the {int(big.sum())} gangs of more than 40", drv[big], exe[big], k[big], fl[big]
    yield "one application", drv[:1], exe[:1], k[:1], fl[:1]
    for lo, hi in ((0, 250), (250, 500), (500, 750), (750, 1000)):
        yield f"applications {lo}..{hi}", drv[lo:hi], exe[lo:hi], k[lo:hi], fl[lo:hi]
    order = np.argsort(k, kind="stable")
    for lo, hi in ((0, 500), (500, 900), (900, 1000)):
        sel = order[lo:hi]
        yield f"K rank {lo}..{hi} (K {int(k[sel].min())}..{int(k[sel].max())})", drv[sel], exe[sel], k[sel], fl[sel]


ALL = (("tightly-pack", 0, False), ("single-az-tightly-pack", 4, True), ("az-aware-tightly-pack", 3, True),
       ("minimal-fragmentation", 2, False), ("single-az-minimal-fragmentation", 5, True), ("distribute-evenly", 1, False))
want = sys.argv[1:] or ["tightly-pack", "single-az-tightly-pack", "az-aware-tightly-pack"]
for name, algo, zoned in [x for x in ALL if x[0] in want]:
    ctx = gangfit.Context(0)
    ctx.set_snapshot(s.avail, s.sched)
    if zoned:
        ctx.set_zones(zone)
    ctx.set_orders(zorder, zorder)
    for what, drv, exe, k, fl in variants():
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(drv, exe, k, fl))
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        ks = []
        for i in range(40):
            ctx.timer_begin(0)
            ctx.fit_batch_dev(IND, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k)
            ms = ctx.timer_end()
            if i >= 10:
                ks.append(ms * 1e3)
        ks.sort()
        feas = int(d_res.cpu().numpy().view(gangfit._native.RESULT_DTYPE)["has_capacity"].sum())
        print(f"{name:24s} {what:40s} {len(apps):5d} applications  device {ks[len(ks) // 2]:7.1f} us (min {ks[0]:6.1f})  feasible {feas}", flush=True)
    ctx.close()
