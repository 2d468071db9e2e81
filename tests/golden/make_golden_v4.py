#!/usr/bin/env python
"""Generates tests/golden/gangfit_golden_v4.json — the rows the packer fixtures (v1-v3) do not cover:

  snapshot     UsageForNodes / NodeSchedulingMetadataForNodes / NodeSorter.PotentialNodes on flat columns
               (LIB/resources/resources.go:31-100, internal/sort/nodesorting.go:41-122) -> usage, available, schedulable, D, X
  executor     rescheduleExecutor's first-fit loop with the doubled overhead (internal/extender/resource.go:640-662, SURVEY.md
               quirk 5) and rescheduleExecutorWithMinimalFragmentation (:675-703)
  efficiency   ComputeAvgPackingEfficiency over [driver] ++ executors in slice order (LIB/binpack/efficiency.go:114-156), as the
               bit patterns of the four float64 values

Answers come from this repository's restatements (oracle/pysnapshot.py, oracle/gangfit_oracle.c): like v1-v3 they pin the
restatement until integration/go/golden_snapshot_test.go has been run with -update on a machine with a Go toolchain.

Node i is named n%05d of name_rank[i] (so the name order is the rank order); zone z is labelled z%d.

    python tests/golden/make_golden_v4.py
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
from oracle import pysnapshot as ps  # noqa: E402

GIB = 1 << 30


def cluster(seed, n, n_rr, n_zones, gpu_ties=False):
    rng = np.random.default_rng(seed)
    shape = rng.integers(0, 3, size=n)
    alloc = np.stack([np.array([16, 32, 64])[shape] * 1000, np.array([64, 128, 256])[shape] * GIB,
                      np.where(rng.random(n) < 0.2, 4, 0)], axis=1).astype(np.int64)
    overhead = np.stack([rng.integers(0, 4, size=n) * 250, rng.integers(0, 8, size=n) * (GIB // 4), np.zeros(n, dtype=np.int64)],
                        axis=1).astype(np.int64)
    ks = rng.integers(1, 9, size=n_rr)
    res_node = rng.integers(0, n, size=int(ks.sum())).astype(np.uint32)
    res_req = np.stack([rng.choice([1000, 2000, 4000], size=len(res_node)), rng.choice([4, 8, 16], size=len(res_node)) * GIB,
                        (rng.random(len(res_node)) < 0.05).astype(np.int64)], axis=1).astype(np.int64)
    flags = (np.where(rng.random(n) < 0.05, ps.UNSCHEDULABLE, 0) | np.where(rng.random(n) < 0.95, ps.READY, 0) |
             np.where(rng.random(n) < 0.8, ps.DRIVER_CANDIDATE, 0)).astype(np.uint32)
    name_rank = rng.permutation(n).astype(np.uint32)
    zone = rng.integers(0, n_zones, size=n).astype(np.uint32)
    if gpu_ties:
        # two nodes with equal free memory and cpu but different free gpus: resources.Eq is false, resourcesLessThan is false both
        # ways (nodesorting.go:74-93) — the comparator calls them equal WITHOUT consulting the name, and sort.Slice is not
        # stable: their relative order is unspecified in the reference (the device and the restatement use the name order)
        alloc[1] = alloc[0]
        overhead[1] = overhead[0]
        alloc[1, 2] = alloc[0, 2] + 4
        res_node = res_node[(res_node != 0) & (res_node != 1)]
        res_req = res_req[: len(res_node)]
        zone[1] = zone[0]
    return dict(alloc=alloc, overhead=overhead, res_node=res_node, res_req=res_req, node_flags=flags, name_rank=name_rank,
                zone=zone, n_zones=n_zones)


def main():
    cases = []
    for seed, n, n_rr, nz, ties in [(41, 12, 6, 2, False), (42, 64, 30, 3, False), (43, 65, 40, 1, False), (44, 200, 90, 3, True)]:
        c = cluster(seed, n, n_rr, nz, ties)
        avail, sched, D, X = ps.build(**c)
        usage = np.zeros_like(c["alloc"])
        np.add.at(usage, c["res_node"].astype(np.int64), c["res_req"])
        # zones with equal (memory, cpu) sums would leave the AZ order to an unstable sort: the generator avoids them
        zs = [(int(avail[c["zone"] == z, 1].sum()), int(avail[c["zone"] == z, 0].sum())) for z in range(nz)]
        assert len(set(zs)) == nz, "zone sums tie: pick another seed"
        case = {k: np.asarray(v).tolist() for k, v in c.items()}
        case.update(name=f"cluster seed {seed}" + (" with a memory/cpu tie that differs in gpu" if ties else ""), n_nodes=n)
        case["snapshot"] = {"usage": usage.tolist(), "avail": avail.tolist(), "sched": sched.tolist(), "D": D.tolist(), "X": X.tolist(),
                            "unspecified": ("nodes 0 and 1 compare equal without their names (equal memory and cpu, different gpu): "
                                            "any order of the two is a legal output of the reference" if ties else "")}
        # ---- the executor path on the same snapshot: nodes that carry reservations lose their overhead a second time (quirk 5)
        rng = np.random.default_rng(seed + 1000)
        has_usage = np.zeros(n, dtype=bool)
        has_usage[c["res_node"][c["res_node"] < n]] = True
        doubled = np.where(has_usage[:, None], c["overhead"], 0)  # what the first-fit loop subtracts on top of `available`
        exe = np.stack([rng.choice([500, 1000, 4000, 16000], size=10), rng.choice([1, 4, 16, 64], size=10) * GIB,
                        (rng.random(10) < 0.2).astype(np.int64)], axis=1).astype(np.int64)
        hosts = (rng.random((10, n)) < 0.15)
        first = [ob.executor_fit(avail, e, X, reserved=doubled) for e in exe]
        minfrag = [ob.executor_fit(avail, e, X, reserved=c["overhead"], minimal_fragmentation=True, hosts=hosts[i].astype(np.uint8))
                   for i, e in enumerate(exe)]
        case["executor"] = {"exe": exe.tolist(), "hosts": [np.nonzero(h)[0].tolist() for h in hosts],
                            "first_fit_extra_reserved": doubled.tolist(), "first_fit": [int(v) for v in first],
                            "minimal_fragmentation": [int(v) for v in minfrag],
                            "note": "4294967295 = not enough capacity (failure-fit); first_fit compares exe against available - "
                                    "overhead for nodes present in the usage map (resource.go:640-643 adds the overhead to a usage map "
                                    "NodeSchedulingMetadataForNodes already mutated); minimal_fragmentation passes the overhead map as "
                                    "reservedResources of GetNodeCapacities (:682)"}
        # ---- average packing efficiencies of tightly-pack results on the same snapshot, as float64 bit patterns
        a = 12
        drv = np.stack([rng.choice([1000, 2000], size=a), rng.choice([2, 4], size=a) * GIB, np.zeros(a, dtype=np.int64)], axis=1).astype(np.int64)
        xe = np.stack([rng.choice([1000, 2000, 4000], size=a), rng.choice([4, 8, 16], size=a) * GIB, np.zeros(a, dtype=np.int64)], axis=1).astype(np.int64)
        k = rng.integers(0, 20, size=a).astype(np.int32)
        ok = (sched >= 0).all()
        if ok:
            out = ob.fit_independent(0, avail, ob.make_apps(drv, xe, k), D, X, sched=sched, zone=c["zone"])
            case["efficiency"] = {"packer": "tightly-pack", "drv": drv.tolist(), "exe": xe.tolist(), "k": k.tolist(),
                                  "has_capacity": out.results["has_capacity"].tolist(),
                                  "driver_node": out.results["driver_node"].tolist(),
                                  "exec_nodes": [out.placement(i)[2].tolist() for i in range(a)],
                                  "avg_bits": [[f"{int(b):016x}" for b in row] for row in out.avg_eff.view(np.uint64)],
                                  "note": "ComputeAvgPackingEfficiency over [driver] ++ executors, duplicates counted, summed in slice "
                                          "order; CPU, Memory, GPU, Max as IEEE-754 bit patterns (zeros when infeasible)"}
        cases.append(case)
    out = os.path.join(HERE, "gangfit_golden_v4.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden_v4.py", "oracle": "oracle/pysnapshot.py + oracle/gangfit_oracle.c",
                   "units": "cpu milli-cores, memory bytes, gpu devices; node i is named n%05d of name_rank[i]; zone z is labelled z%d",
                   "flags": {"unschedulable": 1, "ready": 2, "driver_candidate": 4}, "cases": cases}, f, separators=(",", ":"))
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
