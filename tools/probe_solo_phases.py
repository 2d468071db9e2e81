"""Phase profile of the one-wavefront FIFO chain (fit_fifo_solo_kernel<PROF>): shader cycles per application by phase and the
number of chunk visits, on the headline queue and on config 5.  Run on the MI355X box."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit
from gangfit import workloads as wl

def prof(name, w, algo=0):
    s = w.snapshot
    ctx = gangfit.Context(0, options={"chain_cache": 0})  # every chain replays: these probes time the kernels
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    ctx.fit_batch(1, algo, apps)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.fit_batch(1, algo, apps); ts.append((time.perf_counter() - t0) * 1e3)
    ctx.scan_stats(enable=True, reset=True)
    ctx.fit_batch(1, algo, apps)
    xv, dv = ctx.scan_stats(enable=False, reset=True)
    cyc, ticks = ctx.last_fifo_clock
    ph = ctx.last_fifo_phases
    n = len(apps)
    print(f"{name}: {min(ts):.3f} ms  kernel cycles/app {cyc / n:.0f} clock {cyc / max(ticks, 1) * 100:.0f} MHz  "
          f"phases/app stage {ph[0] / n:.0f} drv {ph[1] / n:.0f} exec {ph[2] / n:.0f} slow {ph[3] / n:.0f} commit {ph[4] / n:.0f}  "
          f"chunk visits/app {ph[5] / n:.2f}  xvis/app {xv / n:.0f} dvis/app {dv / n:.0f}")
    ctx.close()

prof("headline tight", wl.headline(10000, 1000), 0)
prof("headline even", wl.headline(10000, 1000), 1)
prof("congested tight", wl.headline(10000, 1000, congested=True), 0)
prof("config5 100k tight", wl.config(5), 0)
w = wl.headline(10000, 1000)
w.k = np.ones_like(w.k)
prof("headline K=1", w, 0)
w = wl.headline(10000, 1000)
w.k = np.zeros_like(w.k)
prof("headline K=0", w, 0)
