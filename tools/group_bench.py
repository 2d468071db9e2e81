#!/usr/bin/env python
"""Time gf_fit_batch on ONE multi-device context (gf_init with n_dev > 1: node-range sharding inside the library, exchanges
by peer access) — host entry point, incl. H2D of the app records and D2H of the results.  Prints one JSON line.
Used by bench.py: with a repeated device id on a one-GPU box (cost of the path itself), with distinct ids from rank 0 of a
multi-GPU run (the other ranks wait at a barrier), as a subprocess so that a failure cannot take the bench line down."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def run(devices, config, steps, warmup, algo=0):
    import gangfit
    from gangfit import workloads as wl

    w = wl.config(4) if config == "config4" else wl.headline(10000, 1000)
    s = w.snapshot
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    out = {"devices": devices, "nodes": len(s.avail), "apps": len(apps)}
    ref = None
    for name, devs in (("one_device", [devices[0]]), ("group", devices)):
        with gangfit.Context(devices=devs) as c:
            c.set_snapshot(s.avail, s.sched)
            c.set_orders(s.driver_order, s.exec_order)
            for _ in range(warmup):
                r = c.fit_batch(gangfit.GF_MODE_INDEPENDENT, algo, apps)
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter()
                r = c.fit_batch(gangfit.GF_MODE_INDEPENDENT, algo, apps)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            med = ts[len(ts) // 2]
            out[name] = {"ms_per_batch_p50": med * 1e3, "decisions_per_s": len(apps) / med}
            if ref is None:
                ref = r
            else:
                out["results_equal_one_device"] = bool(np.array_equal(r.results, ref.results) and
                                                       np.array_equal(r.exec_nodes, ref.exec_nodes))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--config", default="headline", choices=["headline", "config4"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(run([int(x) for x in a.devices.split(",")], a.config, a.steps, a.warmup)))
