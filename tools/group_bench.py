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
    out = {"devices": devices, "nodes": len(s.avail), "apps": len(apps),
           "sub_contexts": "one per DEVICE (the shards of a repeated id share one: a launch per step, a grid row per shard)"
           if not os.environ.get("GANGFIT_TEST_GROUP_SPLIT") else "one per listed id (GANGFIT_TEST_GROUP_SPLIT)"}
    ref = None

    def timed(c):
        for _ in range(warmup):
            r = c.fit_batch(gangfit.GF_MODE_INDEPENDENT, algo, apps)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            r = c.fit_batch(gangfit.GF_MODE_INDEPENDENT, algo, apps)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        return r, {"ms_per_batch_p50": med * 1e3, "decisions_per_s": len(apps) / med}

    with gangfit.Context(devices=[devices[0]]) as c:
        c.set_snapshot(s.avail, s.sched)
        c.set_orders(s.driver_order, s.exec_order)
        ref, out["one_device"] = timed(c)
    with gangfit.Context(devices=devices) as c:
        c.set_snapshot(s.avail, s.sched)
        c.set_orders(s.driver_order, s.exec_order)
        # exchange 0 = peer stores over xGMI (the default: the messages are KB-sized, a posted store costs the kernel it rides on);
        # exchange 1 = RCCL (two grouped ncclAllGather + one ncclReduce per batch) — only with one rank per physical device
        exchanges = [(0, "group")] + ([(1, "group_rccl")] if len(set(devices)) == len(devices) and len(devices) > 1 else [])
        for ex, name in exchanges:
            try:
                c.set_option("group_exchange", ex)
                r, out[name] = timed(c)
                out[name]["exchange"] = "peer stores" if ex == 0 else "rccl"
                out[name]["shard_count"] = c.shard_count()  # 1 = degraded to the first device, or switched off by the self-check
                out[name]["self_check"] = c.last_error() or "the first sharded batch agreed with the first device's own answer"
                out[name]["results_equal_one_device"] = bool(np.array_equal(r.results, ref.results) and
                                                            np.array_equal(r.exec_nodes, ref.exec_nodes))
                out[name]["vs_one_device"] = out[name]["ms_per_batch_p50"] / out["one_device"]["ms_per_batch_p50"]
            except Exception as e:
                out[name] = {"error": f"{type(e).__name__}: {e}"}
        out["results_equal_one_device"] = all(v.get("results_equal_one_device", False) for k, v in out.items()
                                              if k.startswith("group") and isinstance(v, dict))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--config", default="headline", choices=["headline", "config4"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(run([int(x) for x in a.devices.split(",")], a.config, a.steps, a.warmup)))
