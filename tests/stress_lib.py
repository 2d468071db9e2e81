"""Randomised parity stress of the HIP path (through the C ABI) against the literal oracle: clusters, layouts (general /
merged / identical orders), zones, request templates, unit structures and gang sizes drawn at random; all six packers, both
batch shapes, the averages chooseBestResult compares (bit for bit), the residual tables, findNodes, and — since the chain
cache — a SECOND chain per case on a queue that grows, shrinks or diverges from the first (resumed from checkpoints where the
LDS chains serve), in five contexts that drive the fast paths and their fallbacks.  Used by tests/test_gpu_stress.py (a few
hundred seeds under `-m gpu`) and by tools/stress_parity.py (as long as you like)."""
import numpy as np

import gangfit
from oracle import binding as ob
from test_gpu_parity import _random_problem  # noqa: F401  (re-exported for the tool)
from test_gpu_zones import _zoned_problem

CONTEXTS = (("default", {}), ("generic", {"fifo_generic": 1}), ("small-lds", {"lds_budget": 50000}),
            ("plain-paths", {"sparse_gpu": 0, "zero_copy": 0, "minfrag_hist": 0}), ("no-chain-cache", {"chain_cache": 0}))
ALGOS = (0, 1, 2, 3, 4, 5)


def make_contexts():
    return {name: gangfit.Context(0, options=opts) for name, opts in CONTEXTS}


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


def one_seed(ctxs, seed):
    """Returns (cases run, None) or (cases, description of the first mismatch)."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 3, 63, 64, 65, 200, 1000, 2500, 7000]))
    a = int(rng.integers(1, 160))
    layout = str(rng.choice(["general", "merged", "identical"]))
    tight = bool(rng.integers(0, 2))
    nz = int(rng.integers(1, 6))
    avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, tight, layout, nz)
    if rng.random() < 0.3:  # coarse units so that the narrow domain applies with non-trivial gcds
        for arr in (avail, sched, drv, exe):
            arr[:, 1] *= 1 << 20
    if rng.random() < 0.4:  # gpu nodes a minority: the sparse gpu view of the independent batch
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
    where = f"seed={seed} n={n} a={a} layout={layout} tight={tight} nz={nz} kcap={kcap}"
    cases = 0
    ind_refs = {}  # the literal oracle's answer for the independent batch, once per packer: the contexts share the inputs (its
    #                driver retry loop is O(|D| N) per gang that does not fit — seed 61038: 44 s per pass over the six packers)
    for cname, ctx in ctxs.items():
        ctx.set_snapshot(avail, sched)
        ctx.set_zones(zone)
        ctx.set_orders(D, X)
        for algo in ALGOS:
            gpu = ctx.fit_batch(0, algo, apps)
            if algo not in ind_refs:
                ind_refs[algo] = ob.fit_independent(algo, avail, oapps, D, X, sched=sched, zone=zone)
            ref = ind_refs[algo]
            bad = same(gpu, ref, False)
            if bad is None and algo in (0, 1, 2) and cname == "default":  # the same batch as a ticket of the resident worker
                wk = ctx.worker_fit(algo, apps)
                bad = same(wk, ref, False)
                if bad:
                    bad = "resident worker: " + bad
            if bad is None:  # the feasibility-only call (gf_fit_feasible: what UnschedulablePodMarker reads), every packer
                fits = ctx.fit_feasible(algo, apps)
                if not np.array_equal(fits, np.asarray(ref.results["has_capacity"]).astype(bool)):
                    bad = "gf_fit_feasible differs from HasCapacity"
            if bad is None and algo in (3, 4, 5, 0, 1):
                avg = ctx.avg_packing_efficiency(algo, apps, gpu)
                if ref.avg_eff is not None and not np.array_equal(avg.view(np.uint64), np.asarray(ref.avg_eff).view(np.uint64)):
                    bad = "avg efficiency bits"
            if bad is None:
                exe1 = np.maximum(exe, 1) if rng.random() < 0.7 else exe
                for step in range(2):  # the second chain shares a prefix with the first: resumed where the LDS chains serve
                    if step == 0:
                        m = a
                        d2, e2, k2, f2 = drv, exe1, k, flags
                    else:
                        how = int(rng.integers(0, 3))
                        if how == 0:      # the queue shrank (a driver was scheduled / deleted)
                            m = int(rng.integers(1, a + 1))
                            d2, e2, k2, f2 = drv[:m], exe1[:m], k[:m], flags[:m]
                        elif how == 1:    # one application in the middle is another one now
                            m = a
                            d2 = drv.copy()
                            d2[int(rng.integers(0, a))] = drv[int(rng.integers(0, a))]
                            e2, k2, f2 = exe1, k, flags
                        else:             # the same queue again (kube-scheduler retries the pod)
                            m = a
                            d2, e2, k2, f2 = drv, exe1, k, flags
                    gpu = ctx.fit_batch(1, algo, gangfit.make_apps(d2, e2, k2, f2))
                    ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(d2, e2, k2, f2), D, X, sched=sched, zone=zone)
                    bad = same(gpu, ref, True)
                    if bad is None and not np.array_equal(ctx.residual(), ref.avail_after):
                        bad = "fifo residual"
                    if bad:
                        bad = f"FIFO step {step}: " + bad
                        break
            if bad is None and algo == 0 and cname == "default":  # findNodes, chained, on the same table
                Xk = X[X < n]
                fk = np.clip(k, 1, 50).astype(np.int32)
                placed, last, off, nodes, adds = ctx.find_nodes(exe, fk, chained=True)
                want = ob.find_nodes(avail, exe, fk, Xk, chained=True)
                if not (np.array_equal(placed, want.placed) and np.array_equal(adds, want.adds) and
                        np.array_equal(ctx.residual(), want.avail_after)):
                    bad = "findNodes"
            cases += 1
            if bad:
                return cases, f"MISMATCH {where} ctx={cname} algo={algo}: {bad}"
    return cases, None
