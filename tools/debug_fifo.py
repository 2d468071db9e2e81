import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd"), os.path.join(REPO, "tests")]
import numpy as np
import gangfit
from oracle import binding as ob
import test_gpu_parity as T

n, algo, layout = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(77 * (algo + 1) + n + 7 * len(layout))
ctx = gangfit.Context(0, options={"chain_cache": 0})
for rep in range(3):
    a = 120
    avail, D, X, drv, exe, k = T._random_problem(rng, n, a, tight_cluster=(rep == 2), layout=layout)
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 40).astype(np.int32)
    flags = (rng.random(a) < (0.9 if rep else 1.0)).astype(np.uint32)
    ctx.set_snapshot(avail)
    ctx.set_orders(D, X)
    apps = gangfit.make_apps(drv, exe, k, flags)
    gpu = ctx.fit_batch(1, algo, apps)
    ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe, k, flags), D, X)
    pos = {int(nd): i for i, nd in enumerate(X)}
    print("rep", rep, "failed_at", gpu.failed_at, ref.failed_at)
    for i in range(a):
        g, r = gpu.placement(i), ref.placement(i)
        same = g[0] == r[0] and (not g[0] or (g[1] == r[1] and np.array_equal(g[2], r[2])))
        if not same:
            print(" app", i, "k", k[i], "drv", drv[i], "exe", exe[i], "flags", flags[i])
            print("  gpu", g[0], g[1], pos.get(g[1]), [pos.get(int(x)) for x in g[2][:12]])
            print("  ref", r[0], r[1], pos.get(r[1]), [pos.get(int(x)) for x in r[2][:12]])
            for q in sorted(set([pos.get(g[1], 0), pos.get(r[1], 0)])):
                print("  slot", q, "node", X[q], "avail(snapshot)", avail[X[q]], "ref residual", ref.avail_after[X[q]])
            break
    res_eq = np.array_equal(ctx.residual(), ref.avail_after)
    print(" residual equal:", res_eq)
    if not res_eq:
        for m in range(1, a):
            sub = apps[: m + 1].copy()
            sub[m]["k"] = 0
            sub[m]["drv"] = 0
            osub = ob.make_apps(drv[: m + 1], exe[: m + 1], k[: m + 1], flags[: m + 1])
            osub[m]["k"] = 0
            osub[m]["drv"] = 0
            g = ctx.fit_batch(1, algo, sub)
            r = ob.fit_fifo_chain(algo, avail, osub, D, X)
            gr = ctx.residual()
            if not np.array_equal(gr, r.avail_after):
                bad = np.nonzero((gr != r.avail_after).any(axis=1))[0]
                print(" first divergence after app", m - 1, "k", k[m - 1], "drv", drv[m - 1], "exe", exe[m - 1])
                gp, rp = g.placement(m - 1), r.placement(m - 1)
                print("  gpu", gp[0], pos.get(gp[1]), [pos.get(int(x)) for x in gp[2]])
                print("  ref", rp[0], pos.get(rp[1]), [pos.get(int(x)) for x in rp[2]])
                for nd in bad[:6]:
                    print("  node", nd, "slot", pos.get(int(nd)), "gpu", gr[nd], "ref", r.avail_after[nd], "snap", avail[nd])
                break
        break
