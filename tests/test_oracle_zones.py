"""CPU tests of the zone-aware packers of the oracle (single-az-tightly-pack, az-aware-tightly-pack,
single-az-minimal-fragmentation: LIB/binpack/single_az.go, az_aware_pack_tightly.go) and of the efficiency averages
they compare (LIB/binpack/efficiency.go).  Three restatements must agree: C literal, C closed form, pure Python."""
import numpy as np
import pytest

import kats
from oracle import binding as ob
from oracle import pyoracle as po

GIB = kats.GIB
SAZ, AZA, SAZ_MF = ob.ALGO_SINGLE_AZ_TIGHTLY_PACK, ob.ALGO_AZ_AWARE_TIGHTLY_PACK, ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION


def _py(algo, avail, sched, zone, drv, exe, k, D, X):
    names = lambda ids: [f"n{i}" for i in ids]  # noqa: E731
    av = {f"n{i}": list(r) for i, r in enumerate(avail)}
    sc = {f"n{i}": list(r) for i, r in enumerate(sched)}
    zn = {f"n{i}": int(z) for i, z in enumerate(zone)}
    if algo == ob.ALGO_MINIMAL_FRAGMENTATION:
        r = po.minimal_fragmentation_pack(drv, exe, k, names(D), names(X), av)
    else:
        fn = {SAZ: po.single_az_tightly_pack, AZA: po.az_aware_tightly_pack, SAZ_MF: po.single_az_minimal_fragmentation}[algo]
        r = fn(drv, exe, k, names(D), names(X), av, sc, zn)
    return r.has_capacity, r.driver_node, r.executor_nodes


@pytest.mark.parametrize("case", kats.REFERENCE_PINNED, ids=[c["name"] for c in kats.REFERENCE_PINNED])
@pytest.mark.parametrize("closed", [False, True])
def test_reference_tests_through_the_packer_they_actually_select(case, closed):
    """T1/T3/T4 run `single-az-tightly-pack` (extendertest harness: binpacker name at extender_test_utils.go:96-101 via
    SingleAzTightlyPack) on two nodes of ONE zone: the wrapper must give exactly the plain tightly-pack answers."""
    sched = case["avail"]  # empty cluster: schedulable == available
    ok, d, ex = ob.spark_binpack(SAZ, case["avail"], case["drv"], case["exe"], case["k"], case["D"], case["X"],
                                 closed_form=closed, sched=sched, zone=[0] * len(sched))
    assert ok == case["feasible"]
    if ok:
        assert d == case["driver"] and ex.tolist() == case["execs"]
    else:
        assert d == ob.NO_NODE and len(ex) == 0


def test_single_az_prefers_the_zone_with_the_higher_average_efficiency():
    # hand-derived from single_az.go:75-97.  zone 0 = {n0, n1} (big, empty), zone 1 = {n2} (small, half used).
    sched = [[16000, 64 * GIB, 0], [16000, 64 * GIB, 0], [8000, 16 * GIB, 0]]
    avail = [[16000, 64 * GIB, 0], [16000, 64 * GIB, 0], [4000, 8 * GIB, 0]]
    zone = [0, 0, 1]
    D = X = [0, 1, 2]
    drv, exe, k = [1000, GIB, 0], [1000, GIB, 0], 2
    # zone 0: driver n0, execs [n0, n0]: every listed node is n0 with cpu (0+1+2)/16, mem 3/64 -> Max avg = 3/16
    # zone 1: driver n2, execs [n2, n2]: cpu (4+3)/8, mem (8+3)/16 -> Max avg = 7/8  => zone 1 wins
    for closed in (False, True):
        out = ob.fit_independent(SAZ, avail, ob.make_apps([drv], [exe], [k]), D, X, closed, sched=sched, zone=zone)
        ok, d, ex = out.placement(0)
        assert ok and d == 2 and ex.tolist() == [2, 2]
        assert out.avg_eff[0].tolist() == [7 / 8, 11 / 16, 1.0, 7 / 8]
    assert _py(SAZ, avail, sched, zone, drv, exe, k, D, X) == (True, "n2", ["n2", "n2"])
    # ties keep the FIRST zone in driver order (strict `<`, single_az.go:90): make both zones identical
    sched2 = [[8000, 16 * GIB, 0], [8000, 16 * GIB, 0]]
    avail2 = [[4000, 8 * GIB, 0], [4000, 8 * GIB, 0]]
    ok, d, ex = ob.spark_binpack(SAZ, avail2, drv, exe, k, [1, 0], [0, 1], sched=sched2, zone=[5, 9])
    assert ok and d == 1 and ex.tolist() == [1, 1]  # zone of n1 (9) comes first in the DRIVER order


def test_single_az_fails_when_no_single_zone_fits_and_az_aware_falls_back():
    sched = [[4000, 8 * GIB, 0], [4000, 8 * GIB, 0]]
    avail = [[2000, 8 * GIB, 0], [2000, 8 * GIB, 0]]
    drv, exe, k = [1000, GIB, 0], [1000, GIB, 0], 2  # 3 cpus needed, 2 per zone, 4 in total
    zone = [0, 1]
    for closed in (False, True):
        ok, d, ex = ob.spark_binpack(SAZ, avail, drv, exe, k, [0, 1], [0, 1], closed, sched=sched, zone=zone)
        assert not ok and d == ob.NO_NODE and len(ex) == 0
        ok, d, ex = ob.spark_binpack(AZA, avail, drv, exe, k, [0, 1], [0, 1], closed, sched=sched, zone=zone)
        assert ok and d == 0 and ex.tolist() == [0, 1]  # az_aware_pack_tightly.go:33-37: plain TightlyPack
    assert _py(SAZ, avail, sched, zone, drv, exe, k, [0, 1], [0, 1]) == (False, "", [])
    assert _py(AZA, avail, sched, zone, drv, exe, k, [0, 1], [0, 1]) == (True, "n0", ["n0", "n1"])


def test_zone_without_executor_candidates_is_skipped_and_zero_efficiency_never_wins():
    sched = [[4000, 8 * GIB, 0], [4000, 8 * GIB, 0]]
    avail = [[4000, 8 * GIB, 0], [4000, 8 * GIB, 0]]
    # zone 7 (n0) has a driver candidate but no executor candidate -> skipped (single_az.go:38-41); zone 3 (n1) works
    ok, d, ex = ob.spark_binpack(SAZ, avail, [1000, GIB, 0], [1000, GIB, 0], 1, [0, 1], [1], sched=sched, zone=[7, 3])
    assert ok and d == 1 and ex.tolist() == [1]
    # a feasible result whose average Max is 0.0 is NOT better than WorstAvgPackingEfficiency (strict <, :80, :90):
    # zero-size app on an unused cluster -> EmptyPackingResult even though SparkBinPack succeeded
    for closed in (False, True):
        ok, d, ex = ob.spark_binpack(SAZ, avail, [0, 0, 0], [0, 0, 0], 2, [0, 1], [0, 1], closed, sched=sched,
                                     zone=[0, 0])
        assert not ok and d == ob.NO_NODE
    assert _py(SAZ, avail, sched, [0, 0], [0, 0, 0], [0, 0, 0], 2, [0, 1], [0, 1]) == (False, "", [])


def test_minimal_fragmentation_efficiency_only_counts_the_driver():
    # minimalFragmentation never writes into `reserved` (minimal_fragmentation.go:59-91): chooseBestResult sees the
    # driver request only.  zone 0: two nodes; zone 1: one node.
    sched = [[8000, 8 * GIB, 0], [8000, 8 * GIB, 0]]
    avail = [[8000, 8 * GIB, 0], [4000, 4 * GIB, 0]]
    drv, exe, k = [1000, GIB, 0], [1000, GIB, 0], 2
    out = ob.fit_independent(SAZ_MF, avail, ob.make_apps([drv], [exe], [k]), [0, 1], [0, 1], sched=sched, zone=[0, 1])
    ok, d, ex = out.placement(0)
    # zone 0: driver n0 execs [n0,n0]; eff(n0) = 1/8 (driver only) -> avg Max 1/8.  zone 1: driver n1, execs [n1,n1];
    # eff(n1) = (4+1)/8 -> 5/8 => zone 1
    assert ok and d == 1 and ex.tolist() == [1, 1]
    assert out.avg_eff[0][3] == 5 / 8
    want = ob.avg_packing_efficiency_list(avail, sched, drv, exe, 1, [1, 1], reserved_includes_executors=False)
    assert out.avg_eff[0].tolist() == want.tolist()
    assert ob.avg_packing_efficiency_list(avail, sched, drv, exe, 1, [1, 1])[3] == 7 / 8  # tightly-pack would see 7/8


def _random_zoned(rng, n_nodes, n_zones):
    sched = np.stack([rng.integers(1, 12, n_nodes) * 1000, rng.integers(1, 12, n_nodes) * GIB,
                      rng.integers(0, 3, n_nodes)], axis=1).astype(np.int64)
    used = (sched * rng.uniform(0, 1.1, size=(n_nodes, 3))).astype(np.int64)
    used[:, 0] = used[:, 0] // 250 * 250  # non-integral cores: exercises Value() rounding
    avail = sched - used
    zone = rng.integers(0, n_zones, n_nodes).astype(np.uint32)
    perm = rng.permutation(n_nodes + 1)  # one name outside the metadata
    D = perm[: rng.integers(1, n_nodes + 2)].astype(np.uint32)
    X = rng.permutation(n_nodes + 1)[: rng.integers(0, n_nodes + 2)].astype(np.uint32)
    drv = [int(rng.integers(0, 3)) * 500, int(rng.integers(0, 3)) * GIB, int(rng.integers(0, 2))]
    exe = [int(rng.integers(0, 4)) * 500, int(rng.integers(0, 3)) * GIB, int(rng.integers(0, 2))]
    return avail, sched, zone, D, X, drv, exe, int(rng.integers(0, 9))


@pytest.mark.parametrize("algo", [SAZ, AZA, SAZ_MF, ob.ALGO_MINIMAL_FRAGMENTATION])
def test_three_restatements_agree_on_random_zoned_cases(algo):
    rng = np.random.default_rng(1234 + algo)
    feasible = 0
    for _ in range(400):
        n = int(rng.integers(1, 9))
        avail, sched, zone, D, X, drv, exe, k = _random_zoned(rng, n, int(rng.integers(1, 4)))
        if algo in (SAZ_MF, ob.ALGO_MINIMAL_FRAGMENTATION) and rng.random() < 0.5:
            k = int(rng.integers(0, 40))  # gangs that need several capacity levels
        apps = ob.make_apps([drv], [exe], [k])
        lit = ob.fit_independent(algo, avail, apps, D, X, False, sched=sched, zone=zone)
        clo = ob.fit_independent(algo, avail, apps, D, X, True, sched=sched, zone=zone)
        assert np.array_equal(lit.results, clo.results)
        assert np.array_equal(lit.placement(0)[2], clo.placement(0)[2])  # entries of infeasible apps are undefined
        assert np.array_equal(lit.avg_eff, clo.avg_eff)  # bit-identical doubles
        ok, d, ex = lit.placement(0)
        pok, pd, pex = _py(algo, avail, sched, zone, drv, exe, k, D.tolist(), X.tolist())
        assert ok == pok
        if ok:
            feasible += 1
            assert pd == f"n{d}" and pex == [f"n{i}" for i in ex]
    assert feasible > 40


def test_fifo_chain_with_single_az_packer_commits_the_chosen_zone():
    rng = np.random.default_rng(99)
    for _ in range(60):
        n = int(rng.integers(2, 10))
        avail, sched, zone, D, X, _, _, _ = _random_zoned(rng, n, 2)
        a = 5
        drv = np.stack([rng.integers(0, 3, a) * 500, rng.integers(0, 2, a) * GIB, np.zeros(a, dtype=np.int64)], axis=1)
        exe = np.stack([rng.integers(1, 3, a) * 500, rng.integers(0, 2, a) * GIB, np.zeros(a, dtype=np.int64)], axis=1)
        apps = ob.make_apps(drv, exe, rng.integers(0, 5, a), np.ones(a, dtype=np.uint32))
        lit = ob.fit_fifo_chain(SAZ, avail, apps, D, X, False, sched=sched, zone=zone)
        clo = ob.fit_fifo_chain(SAZ, avail, apps, D, X, True, sched=sched, zone=zone)
        assert np.array_equal(lit.results, clo.results)
        assert all(np.array_equal(lit.placement(i)[2], clo.placement(i)[2]) for i in range(a))
        assert np.array_equal(lit.avail_after, clo.avail_after)
        # replay by hand: every feasible earlier app subtracts per sparkResourceUsage (sparkpods.go:139-146)
        work = np.array(avail, dtype=np.int64)
        for i in range(a):
            one = ob.fit_independent(SAZ, work, apps[i:i + 1], D, X, sched=sched, zone=zone)
            ok, d, ex = one.placement(0)
            assert (ok, d, ex.tolist()) == (bool(lit.results[i]["has_capacity"]), int(lit.results[i]["driver_node"]),
                                            lit.placement(i)[2].tolist())
            if ok and i + 1 < a:
                for node in set(ex.tolist()):
                    work[node] -= exe[i]
                if d not in set(ex.tolist()):
                    work[d] -= drv[i]
        assert np.array_equal(work, lit.avail_after)
