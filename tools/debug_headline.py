import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd"), os.path.join(REPO, "tests")]
import numpy as np
import gangfit
from gangfit import workloads as wl
from oracle import binding as ob
algo = int(sys.argv[1]); congested = sys.argv[2] == "1"; n_apps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
w = wl.headline(congested=congested); s = w.snapshot
ctx = gangfit.Context(0, options={"chain_cache": 0})
ctx.set_snapshot(s.avail, s.sched); ctx.set_orders(s.driver_order, s.exec_order)
apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))[:n_apps]
oapps = ob.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))[:n_apps]
gpu = ctx.fit_batch(1, algo, apps)
ref = ob.fit_fifo_chain(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=True)
pos = {int(nd): i for i, nd in enumerate(s.exec_order)}
nbad = 0
for i in range(len(apps)):
    g, r = gpu.placement(i), ref.placement(i)
    same = g[0] == r[0] and (not g[0] or (g[1] == r[1] and np.array_equal(g[2], r[2])))
    if not same:
        nbad += 1
        if nbad <= 3:
            print("app", i, "k", apps["k"][i], "drv", apps["drv"][i], "exe", apps["exe"][i])
            print("  gpu", g[0], pos.get(g[1]), [pos.get(int(x)) for x in g[2][:16]])
            print("  ref", r[0], pos.get(r[1]), [pos.get(int(x)) for x in r[2][:16]])
print("bad apps", nbad, "residual equal", np.array_equal(ctx.residual(), ref.avail_after))
