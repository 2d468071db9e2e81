"""Independent batches of the zone-aware packers at the headline size (10 000 nodes x 1 000 applications, three zones in the
reference's AZ-major order): the four-kernel path (option zoned_fused = 0) against the one-launch path (fit_zoned_fused_kernel),
host entry (blocking gf_fit_batch, raw symbol, preallocated buffers) and device-resident (gf_fit_batch_dev between HIP events),
with the plain packers beside them.  Answers of the two paths are compared.  Run on the MI355X box."""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import torch  # noqa: E402,F401  (device buffers)

import gangfit  # noqa: E402
from gangfit import _native as N  # noqa: E402
from gangfit import workloads as wl  # noqa: E402

n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_apps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 3
IND = gangfit.GF_MODE_INDEPENDENT
w = wl.headline(n_nodes, n_apps)
s = w.snapshot
zone = (wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(nz)).astype(np.uint32)
zorder = wl.reference_node_order(s.avail, zone)
apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags))
dev = torch.device("cuda", 0)
d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)

ALGOS = [("tightly-pack", 0, False), ("minimal-fragmentation", 2, False), ("single-az-tightly-pack", 4, True),
         ("az-aware-tightly-pack", 3, True), ("single-az-minimal-fragmentation", 5, True)]
print(f"# {n_nodes} nodes x {n_apps} applications, {nz} zones (AZ-major order); us per batch: host entry p50 | device-resident kernel(s)")
for name, algo, zoned in ALGOS:
    answers = {}
    for fused in ((0, 1) if zoned else (1,)):
        ctx = gangfit.Context(0, options={"zoned_fused": fused})
        ctx.set_snapshot(s.avail, s.sched)
        if zoned:
            ctx.set_zones(zone)
            ctx.set_orders(zorder, zorder)
        else:
            ctx.set_orders(s.driver_order, s.exec_order)
        hres = np.zeros(len(apps), dtype=N.RESULT_DTYPE)
        hexec = np.zeros(total_k + 1, dtype=np.uint32)
        lib, h = ctx._lib, ctx._h
        pa, pr, pe = N.ptr(apps), N.ptr(hres), N.ptr(hexec)

        def host_batch():
            rc = lib.gf_fit_batch(h, IND, algo, len(apps), pa, pr, pe, total_k, None)
            if rc != 0:
                raise RuntimeError(f"gf_fit_batch: {rc} {ctx.last_error()}")

        for _ in range(10):
            host_batch()
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            host_batch()
            ts.append((time.perf_counter() - t0) * 1e6)
        ts.sort()
        answers[fused] = (hres.copy(), hexec.copy())

        def dev_batch():
            ctx.fit_batch_dev(IND, algo, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k)

        for _ in range(5):
            dev_batch()
        ks = []
        for _ in range(30):
            ctx.timer_begin(0)
            dev_batch()
            ks.append(ctx.timer_end() * 1e3)
        ks.sort()
        same_dev = bool(np.array_equal(d_res.cpu().numpy().view(N.RESULT_DTYPE), hres))
        print(f"{name:34s} fused={fused}  host p50 {ts[50]:8.1f}  p99 {ts[98]:8.1f}   device {ks[15]:8.1f}   "
              f"decisions/s host {len(apps) / ts[50] * 1e6:12.0f}   feasible {int(hres['has_capacity'].sum())}  dev==host {same_dev}")
        ctx.close()
    if zoned:
        same = np.array_equal(answers[0][0], answers[1][0])
        for a in np.nonzero(answers[0][0]["has_capacity"])[0]:  # (the placements of an application that does not fit are not written)
            o, k = int(apps["exec_off"][a]), int(apps["k"][a])
            same = same and np.array_equal(answers[0][1][o:o + k], answers[1][1][o:o + k])
        print(f"{'':34s} one-launch answers == four-kernel answers: {same}")
