"""The lone blocking call of UnschedulablePodMarker (unschedulablepods.go:93-166): gf_fit_batch (results + placements back over the
host link) against gf_fit_feasible (one HasCapacity byte per application), 1 000 stale pending drivers on 10 000 nodes, every packer.
Median of 300 calls, one after the other; host-clock phases of the last call (gf_call_phases).  Run on the MI355X box."""
import os, sys, time
import ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import _native as N
from gangfit import workloads as wl


def lat(fn, calls=300):
    for _ in range(20):
        fn()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6, ts[int(len(ts) * 0.99)] * 1e6


for n_apps in (1000, 100, 10, 1):
    w = wl.headline(10000, 1000)
    s = w.snapshot
    ctx = gangfit.Context(0)
    ctx.set_snapshot(s.avail, s.sched)
    zone = (wl.splitmix64(0xA3, 10000, 9) % np.uint64(3)).astype(np.uint32)
    ctx.set_zones(zone)
    order = wl.reference_node_order(s.avail, zone)
    ctx.set_orders(order, order)
    apps, total = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k, w.flags)[:n_apps])
    res = np.zeros(len(apps), dtype=N.RESULT_DTYPE)
    ex = np.zeros(total + 1, dtype=np.uint32)
    fe = np.zeros(len(apps), dtype=np.uint8)
    lib, h = ctx._lib, ctx._h
    pa, pr, pe, pf = N.ptr(apps), N.ptr(res), N.ptr(ex), N.ptr(fe)
    for algo, name in ((0, "tightly-pack"), (1, "distribute-evenly"), (3, "single-az-tightly-pack")):
        def full():
            assert lib.gf_fit_batch(h, 0, algo, len(apps), pa, pr, pe, total, None) == 0
        def feas():
            assert lib.gf_fit_feasible(h, algo, len(apps), pa, pf) == 0
        p50, p99 = lat(full)
        ph_full = ctx.call_phases()
        q50, q99 = lat(feas)
        ph_feas = ctx.call_phases() if algo < 3 else None
        ctx.set_option("feasible_announce", 0)  # the same call waiting for the stream's completion signal
        r50, r99 = lat(feas)
        ph_wait = ctx.call_phases() if algo < 3 else None
        ctx.set_option("feasible_announce", 1)
        same = bool(np.array_equal(fe.astype(bool), res["has_capacity"].astype(bool)))
        print(f"{n_apps:5d} apps {name:24s} gf_fit_batch p50 {p50:6.1f} p99 {p99:6.1f} us   gf_fit_feasible p50 {q50:6.1f} p99 {q99:6.1f} us   same answers {same}")
        print(f"        gf_fit_feasible waiting for the stream p50 {r50:6.1f} p99 {r99:6.1f} us")
        print("        phases us  full:", {k: round(v, 1) for k, v in ph_full.items()}, " feasible:", {k: round(v, 1) for k, v in ph_feas.items()} if ph_feas else None,
              " feasible, stream wait:", {k: round(v, 1) for k, v in ph_wait.items()} if ph_wait else None)
    ctx.close()
