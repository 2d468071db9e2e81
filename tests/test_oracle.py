"""CPU tests of the oracle itself (no GPU): known-answer cases, the two C restatements against each other and
against the independent pure-Python restatement, FIFO replay semantics, capacity and efficiency helpers."""
import numpy as np
import pytest

import kats
from oracle import binding as ob
from oracle import pyoracle as po

ALGOS = {0: "tightly-pack", 1: "distribute-evenly"}


def _run_c(case, closed_form):
    return ob.spark_binpack(case["algo"], case["avail"], case["drv"], case["exe"], case["k"], case["D"], case["X"],
                            closed_form=closed_form)


def _run_py(case):
    avail = {f"n{i}": list(r) for i, r in enumerate(case["avail"])}
    fn = po.select_binpacker(ALGOS[case["algo"]])
    res = fn(case["drv"], case["exe"], case["k"], [f"n{i}" for i in case["D"]], [f"n{i}" for i in case["X"]], avail)
    return res.has_capacity, res.driver_node, res.executor_nodes


@pytest.mark.parametrize("case", kats.ALL, ids=[c["name"] for c in kats.ALL])
def test_known_answers_c_literal(case):
    ok, driver, execs = _run_c(case, closed_form=False)
    assert ok == case["feasible"]
    if ok:
        assert driver == case["driver"]
        assert execs.tolist() == case["execs"]
    else:  # EmptyPackingResult: DriverNode "", ExecutorNodes []
        assert driver == ob.NO_NODE and len(execs) == 0


@pytest.mark.parametrize("case", kats.ALL, ids=[c["name"] for c in kats.ALL])
def test_known_answers_c_closed_form(case):
    ok, driver, execs = _run_c(case, closed_form=True)
    assert ok == case["feasible"]
    if ok:
        assert driver == case["driver"] and execs.tolist() == case["execs"]


@pytest.mark.parametrize("case", kats.ALL, ids=[c["name"] for c in kats.ALL])
def test_known_answers_python_literal(case):
    ok, driver, execs = _run_py(case)
    assert ok == case["feasible"]
    if ok:
        assert driver == f"n{case['driver']}"
        assert execs == [f"n{i}" for i in case["execs"]]
    else:
        assert driver == "" and execs == []


def _random_case(rng, n_nodes, small=True):
    hi = 12 if small else 64
    avail = rng.integers(-2, hi, size=(n_nodes, 3)).astype(np.int64)
    avail[:, 2] = rng.integers(-1, 4, size=n_nodes)
    perm = rng.permutation(n_nodes + 2)  # two names that are not in the metadata
    n_x = int(rng.integers(0, n_nodes + 3))
    X = perm[:n_x]
    D = rng.permutation(n_nodes + 2)[: int(rng.integers(0, n_nodes + 3))]
    drv = rng.integers(0, 4, size=3)
    exe = rng.integers(0, 4, size=3)
    if rng.random() < 0.5:
        exe[2] = 0
    k = int(rng.integers(0, 3 * n_nodes + 2))
    return avail, D.astype(np.uint32), X.astype(np.uint32), drv.astype(np.int64), exe.astype(np.int64), k


@pytest.mark.parametrize("algo", [0, 1])
def test_three_restatements_agree_on_random_small_cases(algo):
    rng = np.random.default_rng(1234 + algo)
    for _ in range(400):
        n = int(rng.integers(1, 9))
        avail, D, X, drv, exe, k = _random_case(rng, n)
        if not exe.any() and k > 0:
            k = min(k, 6)
        lit = ob.spark_binpack(algo, avail, drv, exe, k, D, X, closed_form=False)
        clo = ob.spark_binpack(algo, avail, drv, exe, k, D, X, closed_form=True)
        av = {f"n{i}": [int(v) for v in avail[i]] for i in range(n)}
        fn = po.select_binpacker(ALGOS[algo])
        pr = fn([int(v) for v in drv], [int(v) for v in exe], k, [f"n{i}" for i in D], [f"n{i}" for i in X], av)
        assert lit[0] == clo[0] == pr.has_capacity
        if lit[0]:
            assert lit[1] == clo[1] and f"n{lit[1]}" == pr.driver_node
            assert lit[2].tolist() == clo[2].tolist()
            assert [f"n{i}" for i in lit[2]] == pr.executor_nodes


@pytest.mark.parametrize("algo", [0, 1])
def test_literal_and_closed_form_agree_on_batches(algo):
    rng = np.random.default_rng(99 + algo)
    for n in (1, 63, 64, 65, 200):
        avail = rng.integers(-3, 40, size=(n, 3)).astype(np.int64)
        X = rng.permutation(n).astype(np.uint32)
        D = rng.permutation(n).astype(np.uint32)[: max(1, n // 2)]
        a = 64
        apps = ob.make_apps(rng.integers(0, 6, size=(a, 3)), rng.integers(0, 5, size=(a, 3)),
                            rng.integers(0, 4 * n + 4, size=a))
        zero_exe = ~apps["exe"].any(axis=1)
        apps["k"][zero_exe] = np.minimum(apps["k"][zero_exe], 50)
        lit = ob.fit_independent(algo, avail, apps, D, X, closed_form=False)
        clo = ob.fit_independent(algo, avail, apps, D, X, closed_form=True)
        assert np.array_equal(lit.results, clo.results)
        for i in range(a):
            assert np.array_equal(lit.placement(i)[2], clo.placement(i)[2])


@pytest.mark.parametrize("algo", [0, 1])
def test_literal_and_closed_form_agree_at_the_headline_size(algo):
    """10 000 nodes x 1 000 applications (the size the metric is quoted on), independent batch and FIFO chain: the literal loops
    and the closed form give the same results, placements and residual table (the GPU tests hold the kernels to both)."""
    from gangfit import workloads as wl
    w = wl.headline()
    s = w.snapshot
    apps = ob.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    lit = ob.fit_independent(algo, s.avail, apps, s.driver_order, s.exec_order, closed_form=False)
    clo = ob.fit_independent(algo, s.avail, apps, s.driver_order, s.exec_order, closed_form=True)
    assert np.array_equal(lit.results, clo.results) and np.array_equal(lit.exec_nodes, clo.exec_nodes)
    lit = ob.fit_fifo_chain(algo, s.avail, apps, s.driver_order, s.exec_order, closed_form=False)
    clo = ob.fit_fifo_chain(algo, s.avail, apps, s.driver_order, s.exec_order, closed_form=True)
    assert lit.failed_at == clo.failed_at == -1
    assert np.array_equal(lit.results, clo.results) and np.array_equal(lit.exec_nodes, clo.exec_nodes)
    assert np.array_equal(lit.avail_after, clo.avail_after)


def test_fifo_quirk_k7():
    c = kats.FIFO_K7
    apps = ob.make_apps([a["drv"] for a in c["apps"]], [a["exe"] for a in c["apps"]], [a["k"] for a in c["apps"]],
                        [1 if a["skippable"] else 0 for a in c["apps"]])
    for closed in (False, True):
        out = ob.fit_fifo_chain(0, c["avail"], apps, c["D"], c["X"], closed_form=closed)
        ok, driver, execs = out.placement(0)
        assert ok and driver == c["first"]["driver"] and execs.tolist() == c["first"]["execs"]
        assert out.avail_after.tolist() == c["residual"]
        assert out.failed_at == -1
        assert out.placement(1)[0] == c["last"]["feasible"]
    # same chain through the dict-based Python restatement
    avail = {f"n{i}": list(r) for i, r in enumerate(c["avail"])}
    papps = [(a["drv"], a["exe"], a["k"], a["skippable"]) for a in c["apps"]]
    res, failed = po.fit_earlier_drivers_then_pack(po.tightly_pack, papps, ["n0", "n1"], ["n0", "n1"], avail)
    assert failed == -1 and res[0].executor_nodes == ["n0", "n0", "n1"]
    assert [avail["n0"], avail["n1"]] == c["residual"]
    assert not res[1].has_capacity


@pytest.mark.parametrize("algo", [0, 1, 2])
def test_fifo_chain_c_vs_python(algo):
    rng = np.random.default_rng(7 + algo)
    for _ in range(60):
        n = int(rng.integers(2, 10))
        avail = rng.integers(0, 30, size=(n, 3)).astype(np.int64)
        X = rng.permutation(n).astype(np.uint32)
        D = rng.permutation(n).astype(np.uint32)
        a = int(rng.integers(1, 8))
        drv = rng.integers(0, 5, size=(a, 3))
        exe = rng.integers(1, 5, size=(a, 3))
        k = rng.integers(0, 12, size=a)
        flags = (rng.random(a) < 0.4).astype(np.uint32)
        apps = ob.make_apps(drv, exe, k, flags)
        out = ob.fit_fifo_chain(algo, avail, apps, D, X)
        av = {f"n{i}": [int(v) for v in avail[i]] for i in range(n)}
        papps = [([int(v) for v in drv[i]], [int(v) for v in exe[i]], int(k[i]), bool(flags[i])) for i in range(a)]
        binpack = po.minimal_fragmentation_pack if algo == 2 else po.select_binpacker(ALGOS[algo])  # (2: not a registry name)
        res, failed = po.fit_earlier_drivers_then_pack(binpack, papps, [f"n{i}" for i in D],
                                                       [f"n{i}" for i in X], av)
        assert failed == out.failed_at
        for i in range(a):
            if res[i] is None:
                assert out.results[i]["evaluated"] == 0
                continue
            ok, driver, execs = out.placement(i)
            assert ok == res[i].has_capacity
            if ok:
                assert f"n{driver}" == res[i].driver_node
                assert [f"n{j}" for j in execs] == res[i].executor_nodes
        assert out.avail_after.tolist() == [av[f"n{i}"] for i in range(n)]


def test_fifo_abort_and_skip_semantics():
    # earlier app 0 does not fit and is NOT skippable -> chain aborts at 0, nothing else is evaluated
    avail = [[4, 4, 0]]
    apps = ob.make_apps([[1, 1, 0], [1, 1, 0]], [[9, 9, 0], [1, 1, 0]], [1, 1], [0, 0])
    out = ob.fit_fifo_chain(0, avail, apps, [0], [0])
    assert out.failed_at == 0 and out.results[1]["evaluated"] == 0
    # skippable: ignored, the current driver is packed against the untouched snapshot
    apps["flags"][0] = ob.APP_SKIPPABLE
    out = ob.fit_fifo_chain(0, avail, apps, [0], [0])
    assert out.failed_at == -1 and out.placement(1)[0] and out.avail_after.tolist() == avail
    # the current (last) app is never a "failure-earlier-driver" and never subtracted
    apps = ob.make_apps([[1, 1, 0]], [[9, 9, 0]], [1])
    out = ob.fit_fifo_chain(0, avail, apps, [0], [0])
    assert out.failed_at == -1 and not out.placement(0)[0]


def test_node_capacity_doc_example():
    # LIB/capacity/capacity.go:32 doc comment: required 4, available 14, reserved 1 -> 3
    assert ob.node_capacity([14, 14, 14], [1, 1, 1], [4, 4, 4]) == 3
    assert ob.node_capacity([14, 14, 14], [15, 0, 0], [4, 4, 4]) == 0  # reserved > available
    assert ob.node_capacity([14, 14, 14], [0, 0, 0], [0, 0, 0]) == np.iinfo(np.int64).max  # math.MaxInt


def test_minimal_fragmentation_doc_examples():
    # LIB/binpack/minimal_fragmentation.go:43-58 worked examples: capacities [1,1,3,5,5,8,9,10] style; here a compact
    # instance: caps a:1 b:2 c:4 d:5, executors of size 1, driver elsewhere.
    avail = [[1, 99, 0], [2, 99, 0], [4, 99, 0], [5, 99, 0], [9, 9, 0]]
    D, X = [4], [0, 1, 2, 3]
    drv, exe = [1, 1, 0], [1, 1, 0]
    # k=4: a single node with capacity >= 4 exists -> first such node in capacity order (c)
    ok, d, ex = ob.spark_binpack(ob.ALGO_MINIMAL_FRAGMENTATION, avail, drv, exe, 4, D, X)
    assert ok and d == 4 and ex.tolist() == [2, 2, 2, 2]
    # k=7: no single node; drain the largest (d:5) then the remaining 2 fit exactly on b (cap 2)
    ok, d, ex = ob.spark_binpack(ob.ALGO_MINIMAL_FRAGMENTATION, avail, drv, exe, 7, D, X)
    assert ok and ex.tolist() == [3, 3, 3, 3, 3, 1, 1]
    # k=13 > 12 total
    ok, _, _ = ob.spark_binpack(ob.ALGO_MINIMAL_FRAGMENTATION, avail, drv, exe, 13, D, X)
    assert not ok


def test_minimal_fragmentation_reference_doc_comment():
    """The reference's own worked examples (minimal_fragmentation.go:43-58): nodePriorityOrder [a..f], capacities
    1, 1, 3, 5, 5, 17.  Four of the five hold for the code as written; for executorCount = 19 the comment says
    [f x 17, a, b] but internalMinimalFragmentation (:101-110) puts the remaining 2 on c (first node of the
    capacity-sorted list with capacity >= 2).  Parity is with the code."""
    a, b, c, d, e, f = range(6)
    avail = [[x, 99, 0] for x in (1, 1, 3, 5, 5, 17)] + [[1, 1, 0]]
    want = {11: [d] * 5 + [e] * 5 + [a], 6: [d] * 5 + [a], 15: [d] * 5 + [e] * 5 + [c] * 3 + [a, b], 17: [f] * 17,
            19: [f] * 17 + [c, c]}
    for k, execs in want.items():
        ok, drv_node, ex = ob.spark_binpack(ob.ALGO_MINIMAL_FRAGMENTATION, avail, [1, 1, 0], [1, 1, 0], k, [6],
                                            [a, b, c, d, e, f])
        assert ok and drv_node == 6 and ex.tolist() == execs, k


def test_packing_efficiency_matches_formula():
    # efficiency.go:79-103: (schedulable - available + reserved).Value() / schedulable.Value(); Value() rounds cpu
    # milli away from zero to whole cores.
    sched = [[8000, 8 * kats.GIB, 1], [8000, 8 * kats.GIB, 0]]
    avail = [[6500, 6 * kats.GIB, 1], [8000, 8 * kats.GIB, 0]]
    eff, avg = ob.packing_efficiency(avail, sched, [500, kats.GIB, 1], [1000, kats.GIB, 0], 0, [0, 1])
    # node0: used cpu = 1500 + 500 + 1000 = 3000m -> 3 cores / 8; mem (2+1+1)/8; gpu (0+1)/1
    assert eff[0].tolist() == [3 / 8, 4 / 8, 1.0]
    # node1: cpu 1000m/8000m -> 1/8, mem 1/8, no gpu on the node -> 0
    assert eff[1].tolist() == [1 / 8, 1 / 8, 0.0]
    assert avg.tolist() == [(3 / 8 + 1 / 8) / 2, (4 / 8 + 1 / 8) / 2, 1.0, (1.0 + 1 / 8) / 2]
    # rounding away from zero: 2001 milli -> 3 cores
    eff, _ = ob.packing_efficiency([[5999, 0, 0]], [[8000, 8, 0]], [0, 0, 0], [0, 0, 0], 0, [])
    assert eff[0][0] == 3 / 8


def test_executor_first_fit():
    avail = [[1, 1, 0], [5, 5, 0], [9, 9, 1]]
    assert ob.executor_first_fit(avail, [2, 2, 0], [0, 1, 2]) == 1
    assert ob.executor_first_fit(avail, [2, 2, 1], [0, 1, 2]) == 2
    assert ob.executor_first_fit(avail, [20, 2, 0], [0, 1, 2]) == ob.NO_NODE


@pytest.mark.parametrize("algo", [0, 1])
def test_map_shaped_restatement_agrees_with_the_dense_one(algo):
    """oracle/gangfit_oracle_maps.cpp (string-keyed maps: the reference's data-structure shape, used as a CPU baseline) must
    give the dense-array oracle's answers: independent batch and FIFO chain, unknown names and skippable drivers included."""
    from test_gpu_parity import _random_problem

    rng = np.random.default_rng(50 + algo)
    for n, layout, tight in ((7, "general", True), (64, "merged", True), (150, "merged", False), (65, "identical", True)):
        avail, D, X, drv, exe, k = _random_problem(rng, n, 40, tight, layout)
        k = np.minimum(k, 60).astype(np.int32)
        flags = (rng.random(len(k)) < 0.8).astype(np.uint32)
        apps = ob.make_apps(drv, np.maximum(exe, 1), k, flags)
        sched = np.abs(avail) + 3
        want = ob.fit_independent(algo, avail, apps, D, X)
        got = ob.fit_maps(algo, avail, apps, D, X, chain=False, sched=sched)
        assert np.array_equal(got.results, want.results)
        for a in np.nonzero(want.results["has_capacity"])[0]:
            assert np.array_equal(got.placement(int(a))[2], want.placement(int(a))[2])
        want = ob.fit_fifo_chain(algo, avail, apps, D, X)
        got = ob.fit_maps(algo, avail, apps, D, X, chain=True, sched=sched)
        assert got.failed_at == want.failed_at and np.array_equal(got.results, want.results)
        assert np.array_equal(got.avail_after, want.avail_after)
