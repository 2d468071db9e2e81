"""What a RESUMED chain costs when there is nothing left to evaluate (the same queue again): the fixed part of a Filter on an
unchanged snapshot, by table size.  Host time per call, and the chain kernel's own cycles split into the chain loop's phases and
the rest (prologue + epilogue + checkpoint traffic).       (run on the MI355X box)"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl

for name, w in (("10 000 nodes (headline)", wl.headline(10000, 1000)), ("30 000 nodes", wl.headline(30000, 1000)), ("100 000 nodes (config 5)", wl.config(5))):
    s = w.snapshot
    ctx = gangfit.Context(0)
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    for algo, an in ((0, "tightly-pack"),):
        ctx.fit_batch(1, algo, apps)
        cold = []
        for i in range(5):
            q = np.roll(apps, -i - 1)
            t0 = time.perf_counter(); ctx.fit_batch(1, algo, q); cold.append((time.perf_counter() - t0) * 1e3)
        ctx.fit_batch(1, algo, apps)
        warm = []
        for _ in range(30):
            t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); warm.append((time.perf_counter() - t0) * 1e3)
        ctx.scan_stats(enable=True, reset=True)
        ctx.fit_batch(1, algo, apps)
        ctx.scan_stats(enable=False)
        cyc, ticks = ctx.last_fifo_clock
        ph = ctx.last_fifo_phases
        st = ctx.chain_cache_stats()
        cp = ctx.chain_profile()
        print(f"      prologue {cp['prologue_cycles']} cycles, chain {cp['chain_end_cycles'] - cp['prologue_cycles']}, epilogue {cyc - cp['chain_end_cycles']}"
              f"   (GF_SOLO_PROLOGUE_MARKS build: tables in LDS at {cp['rare_cycles']['unindexed']}, shape ids at {cp['rare_cycles']['bound']}, dominators at {cp['rare_cycles']['no_driver']}, index at {cp['rare_cycles']['short']})")
        print(f"{name:26s} {an}: cold {np.median(cold):.3f} ms, same queue again {np.median(warm):.3f} ms (p99 {np.percentile(warm, 99):.3f}); "
              f"instrumented resumed kernel {ticks / 100.0:.1f} us = {cyc} cycles, of which the chain loop's phases {sum(ph[:5])} ; cache {st}", flush=True)
    ctx.close()
