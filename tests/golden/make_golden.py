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
