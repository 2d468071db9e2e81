"""Randomised parity stress (run on the MI355X box): many seeds x packers x layouts x chain kernels against the oracle.
    python tools/stress_parity.py [seconds]
Prints the first mismatch with its seed and exits 1, or a summary and exits 0."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd"), os.path.join(REPO, "tests")]
import numpy as np
import gangfit
from oracle import binding as ob
from test_gpu_parity import _random_problem
from test_gpu_zones import _zoned_problem

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ALGOS = [0, 1, 2, 3, 4, 5]
t_end = time.time() + budget
n_cases = 0
ctxs = {}
for name, env in (("default", {}), ("generic", {"GANGFIT_FIFO_ZONED": "generic", "GANGFIT_FIFO_KERNEL": "v2"}),
                  ("small-lds", {"GANGFIT_LDS_BUDGET": "50000"}), ("block-cooperative", {"GANGFIT_FIFO_SOLO": "0"}),
                  ("plain-paths", {"GANGFIT_SPARSE_GPU": "0", "GANGFIT_ZEROCOPY": "0", "GANGFIT_WAIT": "block", "GANGFIT_MINFRAG_HIST": "0"})):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ctxs[name] = gangfit.Context(0)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def same(gpu, ref, fifo):
    if not np.array_equal(gpu.results["has_capacity"], ref.results["has_capacity"]):
        return "has_capacity"
    if not np.array_equal(gpu.results["driver_node"], ref.results["driver_node"]):
        return "driver_node"
    if not np.array_equal(gpu.results["evaluated"], ref.results["evaluated"]):
        return "evaluated"
    for a in np.nonzero(ref.results["has_capacity"])[0]:
        if not np.array_equal(gpu.placement(int(a))[2], ref.placement(int(a))[2]):
            return f"placement of app {a}"
    if fifo and gpu.failed_at != ref.failed_at:
        return "failed_at"
    return None


seed = 0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 3, 63, 64, 65, 200, 1000, 2500, 7000]))
    a = int(rng.integers(1, 160))
    layout = str(rng.choice(["general", "merged", "identical"]))
    tight = bool(rng.integers(0, 2))
    nz = int(rng.integers(1, 6))
    avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, tight, layout, nz)
    if rng.random() < 0.3:  # coarse units so that the narrow domain applies with non-trivial gcds
        avail[:, 1] *= 1 << 20
        sched[:, 1] *= 1 << 20
        drv[:, 1] *= 1 << 20
        exe[:, 1] *= 1 << 20
    if rng.random() < 0.4:  # gpu nodes a minority: the sparse gpu view of the independent batch (most executors need a gpu)
        frac = float(rng.choice([0.03, 0.1, 0.2]))
        avail[:, 2] = np.where(rng.random(n) < frac, rng.integers(1, 9, size=n), rng.integers(-1, 1, size=n))
        sched[:, 2] = np.maximum(avail[:, 2], 0) + rng.integers(0, 3, size=n)
        exe[:, 2] = np.where(rng.random(a) < 0.7, rng.integers(1, 4, size=a), 0)
    if rng.random() < 0.3:  # requests finer than the table's gcd units: the per-batch unit refinement of the int32 chains
        f = int(rng.choice([2, 4, 6, 8]))
        avail[:, 1] *= f
        sched[:, 1] *= f
        avail[:, 0] *= 2
        sched[:, 0] *= 2
    if rng.random() < 0.5:  # a handful of templates: few distinct request shapes, runs of equal shapes (the indexed chains)
        t = rng.integers(0, min(a, int(rng.integers(1, 8))), size=a)
        drv, exe = drv[t], exe[t]
    kcap = int(rng.choice([5, 40, 300, 3000]))
    k = np.minimum(k, kcap).astype(np.int32)
    flags = (rng.random(a) < 0.85).astype(np.uint32)
    apps = gangfit.make_apps(drv, exe, k, flags)
    oapps = ob.make_apps(drv, exe, k, flags)
    for cname, ctx in ctxs.items():
        ctx.set_snapshot(avail, sched)
        ctx.set_zones(zone)
        ctx.set_orders(D, X)
        for algo in ALGOS:
            gpu = ctx.fit_batch(0, algo, apps)
            ref = ob.fit_independent(algo, avail, oapps, D, X, sched=sched, zone=zone)
            bad = same(gpu, ref, False)
            if bad is None and algo in (3, 4, 5, 0, 1):
                avg = ctx.avg_packing_efficiency(algo, apps, gpu)
                if ref.avg_eff is not None and not np.array_equal(avg.view(np.uint64), np.asarray(ref.avg_eff).view(np.uint64)):
                    bad = "avg efficiency bits"
            if bad is None:
                exe1 = np.maximum(exe, 1) if rng.random() < 0.7 else exe
                fapps = gangfit.make_apps(drv, exe1, k, flags)
                gpu = ctx.fit_batch(1, algo, fapps)
                ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe1, k, flags), D, X, sched=sched, zone=zone)
                bad = same(gpu, ref, True)
                if bad is None and not np.array_equal(ctx.residual(), ref.avail_after):
                    bad = "fifo residual"
                if bad:
                    bad = "FIFO " + bad
            if bad is None and algo == 0 and cname == "default":  # findNodes, chained, on the same table
                Xk = X[X < n]
                fk = np.clip(k, 1, 50).astype(np.int32)
                placed, last, off, nodes, adds = ctx.find_nodes(exe, fk, chained=True)
                want = ob.find_nodes(avail, exe, fk, Xk, chained=True)
                if not (np.array_equal(placed, want.placed) and np.array_equal(adds, want.adds) and
                        np.array_equal(ctx.residual(), want.avail_after)):
                    bad = "findNodes"
            n_cases += 1
            if bad:
                print(f"MISMATCH seed={seed} ctx={cname} algo={algo} n={n} a={a} layout={layout} tight={tight} nz={nz} kcap={kcap}: {bad}")
                sys.exit(1)
print(f"stress ok: {n_cases} (context, packer) cases over {seed} seeds in {budget:.0f} s")
