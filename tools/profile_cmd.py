#!/usr/bin/env python
"""One kernel family per process, for the counter passes of tools/profile_round.sh (rocprofv3 names kernels without their template
arguments, so every profiled command launches ONE instantiation of the kernel it is about):
    profile_cmd.py batch <algo>   12 device-resident independent batches of a zone-aware packer at the headline size
                                  (10 000 nodes x 1 000 applications, 3 zones, AZ-major order): fit_zoned_fused_kernel
    profile_cmd.py chain <algo>   8 cold FIFO chains (999 + 1, rotated heads, checkpoints off) of the packer: the zone-aware
                                  packers run fit_fifo_zoned_lds_kernel, the minimal-fragmentation ones fit_fifo_minfrag_lds_kernel
algo = single-az-tightly-pack | az-aware-tightly-pack | minimal-fragmentation | single-az-minimal-fragmentation | tightly-pack
No torch (GANGFIT_NO_TORCH=1): the process is the library, numpy and the workload generator.  Prints one JSON line."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("GANGFIT_NO_TORCH", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import numpy as np  # noqa: E402

import gangfit  # noqa: E402
from gangfit import workloads as wl  # noqa: E402

ALGO = {"tightly-pack": 0, "distribute-evenly": 1, "minimal-fragmentation": 2, "az-aware-tightly-pack": 3, "single-az-tightly-pack": 4,
        "single-az-minimal-fragmentation": 5}
what, name = sys.argv[1], sys.argv[2]
algo = ALGO[name]
zoned = algo in (3, 4, 5)
w = wl.headline(10000, 1000)
s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched)
if zoned:
    zone = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    order = wl.reference_node_order(s.avail, zone)
    ctx.set_zones(zone)
    ctx.set_orders(order, order)
else:
    ctx.set_orders(s.driver_order, s.exec_order)
apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
ts = []
if what == "batch":
    for _ in range(12):
        t0 = time.perf_counter()
        ctx.fit_batch(gangfit.GF_MODE_INDEPENDENT, algo, apps)
        ts.append((time.perf_counter() - t0) * 1e3)
else:
    for i in range(8):
        q = np.roll(apps, -i)
        t0 = time.perf_counter()
        ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, q)
        ts.append((time.perf_counter() - t0) * 1e3)
ctx.close()
print(json.dumps({"what": what, "algo": name, "calls": len(ts), "host_ms_min": min(ts), "host_ms_median": sorted(ts)[len(ts) // 2]}))
