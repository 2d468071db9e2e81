#!/usr/bin/env python
"""FIFO-chain Filter latency for the headline size (10k nodes, 999 earlier drivers + 1): host entry point end to end
(H2D + kernel + D2H) and kernel-only (HIP events).  GANGFIT_FIFO_WAVES / GANGFIT_LDS_BUDGET select the variant."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit  # noqa: E402
from gangfit import workloads as wl  # noqa: E402


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))]


def main():
    n_nodes = int(os.environ.get("NODES", "10000"))
    n_apps = int(os.environ.get("APPS", "1000"))
    calls = int(os.environ.get("CALLS", "40"))
    out = {"fifo_waves": os.environ.get("GANGFIT_FIFO_WAVES", "default"),
           "lds_budget": os.environ.get("GANGFIT_LDS_BUDGET", "default"), "nodes": n_nodes, "apps": n_apps}
    with gangfit.Context(0) as ctx:
        for congested in (False, True):
            w = wl.headline(n_nodes, n_apps, congested=congested)
            s = w.snapshot
            ctx.set_snapshot(s.avail, s.sched)
            ctx.set_orders(s.driver_order, s.exec_order)
            for algo, name in ((gangfit.GF_ALGO_TIGHTLY_PACK, "tight"), (gangfit.GF_ALGO_DISTRIBUTE_EVENLY, "even")):
                apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
                e2e, kern = [], []
                for i in range(calls + 3):
                    rolled = np.roll(apps, -i)
                    t0 = time.perf_counter()
                    ctx.timer_begin()
                    ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, rolled)
                    dt = time.perf_counter() - t0
                    k_ms = ctx.timer_end()
                    if i >= 3:
                        e2e.append(dt * 1e3)
                        kern.append(k_ms)
                ctx.scan_stats(enable=True, reset=True)
                ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, apps)
                xv, dv = ctx.scan_stats(enable=False, reset=True)
                cyc, ticks = ctx.last_fifo_clock
                key = ("congested" if congested else "nominal") + "_" + name
                out[key] = {"e2e_p50_ms": round(pct(e2e, 0.5), 4), "e2e_p99_ms": round(pct(e2e, 0.99), 4),
                            "stream_p50_ms": round(pct(kern, 0.5), 4), "exec_slots": xv, "driver_slots": dv,
                            "kernel_cycles": cyc, "kernel_us": ticks / 100.0,
                            "sclk_mhz": round(cyc / max(ticks, 1) * 100.0, 1), "phase_cycles": ctx.last_fifo_phases}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
