#!/usr/bin/env python
"""Generates tests/golden/gangfit_golden_v1.json — small seeded problems with the answers of the LITERAL C oracle
(oracle/gangfit_oracle.c, the line-by-line restatement of the reference's loops) for every registered packer and both
batch shapes.  The reference itself is Go and cannot be built in this image (no Go toolchain), so these vectors pin the
restatement, not the Go binary: they guard the oracle against regressions and give the HIP path a fixed target that
does not depend on the oracle being built on the GPU box.  The reference's own test vectors (T1/T3/T4, doc-comment
examples) live in tests/kats.py.

    python tests/golden/make_golden.py        # rewrites the JSON next to this script
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import binding as ob  # noqa: E402

ALGOS = {"tightly-pack": 0, "distribute-evenly": 1, "minimal-fragmentation": 2, "az-aware-tightly-pack": 3,
         "single-az-tightly-pack": 4, "single-az-minimal-fragmentation": 5}


def problem(seed, n, a, tight):
    rng = np.random.default_rng(seed)
    hi = 30 if tight else 600
    avail = rng.integers(-2, hi, size=(n, 3)).astype(np.int64)
    avail[:, 2] = rng.integers(-1, 6, size=n)
    avail[:, 0] *= 250
    sched = np.maximum(avail, 0) + rng.integers(0, 400, size=(n, 3)) * np.array([250, 1, 1])
    sched[rng.random(n) < 0.1] = 0
    zone = rng.integers(0, 3, size=n).astype(np.uint32)
    base = rng.permutation(n)
    X = base[rng.random(n) < 0.85]
    D = base[rng.random(n) < 0.7]
    if len(X) == 0:
        X = base[:1]
    if len(D) == 0:
        D = base[-1:]
    X = np.insert(X, len(X) // 2, n + 7)  # a name that is not in the metadata
    drv = rng.integers(0, 9, size=(a, 3)).astype(np.int64)
    exe = rng.integers(0, 6, size=(a, 3)).astype(np.int64)
    drv[:, 0] *= 250
    exe[:, 0] *= 250
    exe[rng.random(a) < 0.5, 2] = 0
    k = rng.integers(0, 2 * n, size=a).astype(np.int32)
    k[~exe.any(axis=1)] = np.minimum(k[~exe.any(axis=1)], 40)
    flags = (rng.random(a) < 0.85).astype(np.uint32)
    return dict(avail=avail, sched=sched, zone=zone, D=D.astype(np.uint32), X=X.astype(np.uint32), drv=drv, exe=exe, k=k,
                flags=flags)


def answers_for(p, case):
    """Literal-oracle answers of one problem for every packer, both batch shapes, plus the findNodes chain."""
    a = len(p["k"])
    apps = ob.make_apps(p["drv"], p["exe"], p["k"], p["flags"])
    kf = np.asarray(p.get("fifo_k", np.minimum(p["k"], 25)), dtype=np.int32)
    fexe = np.asarray(p.get("fifo_exe", np.maximum(p["exe"], 1)), dtype=np.int64)
    case["answers"] = {}
    for name, algo in ALGOS.items():
        ind = ob.fit_independent(algo, p["avail"], apps, p["D"], p["X"], sched=p["sched"], zone=p["zone"])
        fapps = ob.make_apps(p["drv"], fexe, kf, p["flags"])
        fifo = ob.fit_fifo_chain(algo, p["avail"], fapps, p["D"], p["X"], sched=p["sched"], zone=p["zone"])
        case["answers"][name] = {
            "independent": {"has_capacity": ind.results["has_capacity"].tolist(),
                            "driver_node": ind.results["driver_node"].tolist(),
                            "exec_nodes": [ind.placement(i)[2].tolist() for i in range(a)]},
            "fifo": {"k": kf.tolist(), "exe": fexe.tolist(), "failed_at": int(fifo.failed_at),
                     "has_capacity": fifo.results["has_capacity"].tolist(),
                     "evaluated": fifo.results["evaluated"].tolist(),
                     "driver_node": fifo.results["driver_node"].tolist(),
                     "exec_nodes": [fifo.placement(i)[2].tolist() for i in range(a)],
                     "avail_after": fifo.avail_after.tolist()},
        }
    # findNodes (failover.go:412-436) over the known nodes of the executor order, requests chained like the reconciler's loop
    order = np.asarray([x for x in p["X"] if x < len(p["avail"])], dtype=np.uint32)
    fk = np.maximum(np.minimum(p["k"], 12), 1).astype(np.int32)
    fn = ob.find_nodes(p["avail"], p["exe"], fk, order, chained=True)
    case["find_nodes"] = {"order": order.tolist(), "k": fk.tolist(), "exe": np.asarray(p["exe"]).tolist(),
                          "placed": fn.placed.tolist(), "exec_nodes": [fn.placement(q).tolist() for q in range(len(fk))],
                          "adds": fn.adds.tolist(), "avail_after": fn.avail_after.tolist()}


def targeted_cases():
    """Hand-built problems that isolate what the random ones only hit by chance (cpu in milli, memory / gpu in units)."""
    out = []
    # (1) distribute-evenly over several passes with nodes dropping out pass by pass (distribute_evenly.go:49-71):
    #     capacities for exe (250 m, 1, 0) are 1, 5, 3, 0 (overcommitted) and 9
    avail = np.array([[250, 9, 0], [1250, 9, 0], [750, 3, 0], [-250, 9, 0], [2250, 9, 1]], dtype=np.int64)
    n = len(avail)
    ks = [4, 7, 14, 18, 19, 3, 9]
    out.append(dict(name="distribute-evenly multi-pass", avail=avail, sched=np.maximum(avail, 0) * 2 + 1,
                    zone=np.array([0, 1, 0, 1, 0], dtype=np.uint32), D=np.array([3, 1, 4, 0], dtype=np.uint32),
                    X=np.array([0, 1, 2, 3, 4], dtype=np.uint32),
                    drv=np.array([[250, 1, 0]] * 5 + [[1000, 2, 0], [0, 0, 1]], dtype=np.int64),
                    exe=np.array([[250, 1, 0]] * 7, dtype=np.int64), k=np.array(ks, dtype=np.int32),
                    flags=np.array([1, 1, 0, 1, 0, 1, 1], dtype=np.uint32), fifo_k=np.array([2, 3, 2, 1, 30, 2, 1], dtype=np.int32),
                    fifo_exe=np.array([[250, 1, 0]] * 7, dtype=np.int64)))
    # (2) the sparkResourceUsage quirk (sparkpods.go:139-146) where the driver shares its node with executors, at the 63 | 64
    #     boundary of the 64-slot chunks the kernels scan: the first 63 nodes are full, node 63 is the last lane of chunk 0,
    #     node 64 the first lane of chunk 1 (absent in the 64-node variant: the table ends at the boundary)
    for n in (64, 65, 130):
        avail = np.zeros((n, 3), dtype=np.int64)
        avail[63:] = [4000, 8, 0]
        if n > 100:
            avail[100] = [4000, 8, 2]
        order = np.arange(n, dtype=np.uint32)
        drv = np.array([[1000, 1, 0], [1000, 1, 0], [2000, 2, 0], [500, 0, 0], [1000, 1, 0], [250, 1, 0], [1000, 1, 1], [0, 0, 0]], dtype=np.int64)
        exe = np.array([[1000, 2, 0], [1000, 2, 0], [1000, 1, 0], [250, 1, 0], [4000, 8, 0], [500, 1, 0], [500, 1, 1], [250, 1, 0]], dtype=np.int64)
        k = np.array([5, 0, 2, 3, 1, 6, 2, 9], dtype=np.int32)  # K = 0: the driver's request is subtracted (no executor overwrites it)
        out.append(dict(name=f"fifo usage quirk at the chunk boundary, {n} nodes", avail=avail, sched=np.maximum(avail, 0) + 1000,
                        zone=(order % 2).astype(np.uint32), D=order, X=order, drv=drv, exe=exe, k=k,
                        flags=np.array([0, 0, 1, 0, 1, 0, 1, 0], dtype=np.uint32), fifo_k=k, fifo_exe=exe))
    return out


def main_v2():
    cases = []
    for p in targeted_cases():
        case = {key: np.asarray(v).tolist() for key, v in p.items() if key != "name"}
        case.update(name=p["name"], seed=0, n_nodes=len(p["avail"]))
        answers_for(p, case)
        cases.append(case)
    for seed, n, a, tight in [(11, 9, 10, True), (12, 64, 20, True), (13, 65, 20, False), (14, 200, 16, True)]:
        p = problem(seed, n, a, tight)
        case = {key: np.asarray(v).tolist() for key, v in p.items()}
        case.update(name=f"random seed {seed}", seed=seed, n_nodes=n)
        answers_for(p, case)
        cases.append(case)
    out = os.path.join(HERE, "gangfit_golden_v2.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (main_v2)", "oracle": "oracle/gangfit_oracle.c (literal loops)",
                   "units": "cpu milli-cores, memory and gpu in whole units; node i is named n%05d, zone z is named z%d, an order "
                            "entry >= n_nodes is a name that is not a key of the metadata map",
                   "algos": ALGOS, "cases": cases}, f, separators=(",", ":"))
    print(out, os.path.getsize(out), "bytes")


def az_major_order(avail, zone):
    """getNodeNamesInPriorityOrder (internal/sort/nodesorting.go:82-122): AZ priority (zones by summed free memory, then cpu,
    ascending), then free memory, free cpu, name."""
    n = len(avail)
    zs = sorted(set(int(z) for z in zone), key=lambda z: (int(avail[zone == z, 1].sum()), int(avail[zone == z, 0].sum()), z))
    prio = {z: r for r, z in enumerate(zs)}
    return np.array(sorted(range(n), key=lambda i: (prio[int(zone[i])], int(avail[i, 1]), int(avail[i, 0]), i)), dtype=np.uint32)


def round3_cases():
    """What round 2's chain kernels added paths for: the reference's AZ-major priority order with a handful of request
    templates (capacity rows, histograms and first positions reused and patched over a whole chain), minimal-fragmentation
    gangs that walk several capacity levels and end on a partially drained one, tiny executor requests (capacities in the
    hundreds) next to ordinary ones, drivers that share a node with their executors."""
    out = []
    rng = np.random.default_rng(20260923)
    # (1) three zones, AZ-major order, six templates, 36 applications
    n = 210
    alloc = np.array([[16000, 64, 0], [32000, 128, 0], [64000, 256, 8]], dtype=np.int64)[rng.integers(0, 3, size=n)]
    used = (rng.random((n, 3)) * 0.9 * alloc).astype(np.int64)
    used[:, 0] = used[:, 0] // 250 * 250
    avail = alloc - used
    avail[rng.random(n) < 0.03] = [-500, -1, 0]
    zone = rng.integers(0, 3, size=n).astype(np.uint32)
    order = az_major_order(avail, zone)
    tmpl_d = np.array([[1000, 2, 0], [2000, 4, 0], [500, 1, 0], [1000, 8, 0], [4000, 4, 0], [1000, 2, 1]], dtype=np.int64)
    tmpl_x = np.array([[1000, 4, 0], [2000, 8, 0], [4000, 16, 0], [500, 2, 0], [8000, 8, 0], [1000, 4, 1]], dtype=np.int64)
    t = rng.integers(0, 6, size=36)
    k = np.minimum(1 + rng.geometric(1 / 9.0, size=36), 60).astype(np.int32)
    out.append(dict(name="AZ-major order, three zones, six templates", avail=avail, sched=alloc.copy(), zone=zone, D=order, X=order,
                    drv=tmpl_d[t], exe=tmpl_x[t], k=k, flags=(rng.random(36) < 0.9).astype(np.uint32), fifo_k=k, fifo_exe=tmpl_x[t]))
    # (2) minimal-fragmentation level walks: capacities for exe (1000 m, 1, 0) are small and repeated (1, 2, 3, 5, 8), the
    #     gangs need several levels, leave remainders that a lower level can or cannot take, and hit capacity == K exactly
    caps = np.array([1, 2, 3, 5, 8, 2, 3, 1, 5, 2, 3, 3, 1, 8, 2, 5, 1, 1, 2, 3] * 4, dtype=np.int64)
    n = len(caps)
    avail = np.stack([caps * 1000 + 300, caps + 0, np.zeros(n, dtype=np.int64)], axis=1)
    order = np.arange(n, dtype=np.uint32)
    ks = np.array([8, 9, 11, 13, 16, 4, 21, 6, 30, 7, 2, 17, 40, 3, 5, 26], dtype=np.int32)
    a = len(ks)
    out.append(dict(name="minimal-fragmentation level walks", avail=avail, sched=avail + [1000, 1, 0], zone=(order // 27).astype(np.uint32),
                    D=order, X=order, drv=np.array([[300, 0, 0]] * a, dtype=np.int64), exe=np.array([[1000, 1, 0]] * a, dtype=np.int64),
                    k=ks, flags=np.ones(a, dtype=np.uint32), fifo_k=ks, fifo_exe=np.array([[1000, 1, 0]] * a, dtype=np.int64)))
    # (3) tiny requests (hundreds of executors per node) between ordinary ones, two zones in AZ-major order
    n = 96
    avail = np.stack([rng.integers(4, 64, size=n) * 1000, rng.integers(8, 256, size=n), np.zeros(n, dtype=np.int64)], axis=1).astype(np.int64)
    zone = rng.integers(0, 2, size=n).astype(np.uint32)
    order = az_major_order(avail, zone)
    exe = np.array([[100, 1, 0], [2000, 8, 0], [50, 1, 0], [1000, 4, 0], [100, 1, 0], [4000, 16, 0]] * 3, dtype=np.int64)
    a = len(exe)
    k = np.array([300, 12, 500, 7, 90, 3] * 3, dtype=np.int32)
    out.append(dict(name="tiny and ordinary requests, two zones", avail=avail, sched=avail + [8000, 32, 0], zone=zone, D=order, X=order,
                    drv=np.array([[1000, 2, 0]] * a, dtype=np.int64), exe=exe, k=k, flags=np.ones(a, dtype=np.uint32), fifo_k=k,
                    fifo_exe=exe))
    return out


def main_v3():
    cases = []
    for p in round3_cases():
        case = {key: np.asarray(v).tolist() for key, v in p.items() if key != "name"}
        case.update(name=p["name"], seed=0, n_nodes=len(p["avail"]))
        answers_for(p, case)
        cases.append(case)
    out = os.path.join(HERE, "gangfit_golden_v3.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (main_v3)", "oracle": "oracle/gangfit_oracle.c (literal loops)",
                   "units": "cpu milli-cores, memory and gpu in whole units; node i is named n%05d, zone z is named z%d",
                   "algos": ALGOS, "cases": cases}, f, separators=(",", ":"))
    print(out, os.path.getsize(out), "bytes")


def main():
    cases = []
    for seed, n, a, tight in [(1, 6, 12, True), (2, 70, 24, True), (3, 130, 24, False), (4, 64, 16, True)]:
        p = problem(seed, n, a, tight)
        case = {key: np.asarray(v).tolist() for key, v in p.items()}
        case.update(seed=seed, n_nodes=n, answers={})
        apps = ob.make_apps(p["drv"], p["exe"], p["k"], p["flags"])
        for name, algo in ALGOS.items():
            ind = ob.fit_independent(algo, p["avail"], apps, p["D"], p["X"], sched=p["sched"], zone=p["zone"])
            kf = np.minimum(p["k"], 25).astype(np.int32)
            fapps = ob.make_apps(p["drv"], np.maximum(p["exe"], 1), kf, p["flags"])
            fifo = ob.fit_fifo_chain(algo, p["avail"], fapps, p["D"], p["X"], sched=p["sched"], zone=p["zone"])
            case["answers"][name] = {
                "independent": {"has_capacity": ind.results["has_capacity"].tolist(),
                                "driver_node": ind.results["driver_node"].tolist(),
                                "exec_nodes": [ind.placement(i)[2].tolist() for i in range(a)]},
                "fifo": {"k": kf.tolist(), "failed_at": int(fifo.failed_at),
                         "has_capacity": fifo.results["has_capacity"].tolist(),
                         "evaluated": fifo.results["evaluated"].tolist(),
                         "driver_node": fifo.results["driver_node"].tolist(),
                         "exec_nodes": [fifo.placement(i)[2].tolist() for i in range(a)],
                         "avail_after": fifo.avail_after.tolist()},
            }
        cases.append(case)
    out = os.path.join(HERE, "gangfit_golden_v1.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "oracle": "oracle/gangfit_oracle.c (literal loops)",
                   "algos": ALGOS, "cases": cases}, f, separators=(",", ":"))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
    main_v2()
    main_v3()
