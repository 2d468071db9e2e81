"""GPU parity of minimalFragmentation (LIB/binpack/minimal_fragmentation.go:59-137) and of the registered packer built on
it, `single-az-minimal-fragmentation` (single_az_minimal_fragmentation.go:20), against the literal CPU oracle:
independent batches, FIFO chains, all three slot layouts.  `python -m pytest tests -m gpu`."""
import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob
from test_gpu_parity import _assert_same, _gpu_apps, _random_problem
from test_gpu_zones import _bits, _setup, _zoned_problem

pytestmark = pytest.mark.gpu

IND, FIFO = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_MODE_FIFO_CHAIN
MF, SAZMF = gangfit.GF_ALGO_MINIMAL_FRAGMENTATION, gangfit.GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION
GIB = 1 << 30


def _caps_cluster(caps, extra_driver_node=True):
    """Nodes whose capacity for a (1 cpu, 1 B) executor is caps[i]; one more roomy node hosts the driver."""
    avail = [[c, 99, 0] for c in caps]
    if extra_driver_node:
        avail.append([1, 1, 0])
    return avail


def test_doc_comment_examples(gf_ctx):
    """The worked examples of the reference's doc comment (minimal_fragmentation.go:43-58): nodePriorityOrder
    [a..f] with capacities 1, 1, 3, 5, 5, 17."""
    a, b, c, d, e, f = range(6)
    avail = _caps_cluster([1, 1, 3, 5, 5, 17])
    gf_ctx.set_snapshot(avail)
    gf_ctx.set_orders([6], [a, b, c, d, e, f])
    drv, exe = [1, 1, 0], [1, 1, 0]
    want = {
        11: [d] * 5 + [e] * 5 + [a],
        6: [d] * 5 + [a],
        15: [d] * 5 + [e] * 5 + [c] * 3 + [a, b],
        17: [f] * 17,
        # the doc comment says [f x 17, a, b]; the CODE it documents (:101-110) places the remaining 2 executors on the
        # first node of the capacity-sorted list with capacity >= 2, which is c — parity is with the code
        19: [f] * 17 + [c, c],
    }
    for k, execs in want.items():
        ok, drv_node, ex = gf_ctx.spark_binpack(MF, drv, exe, k)
        assert ok and drv_node == 6 and ex.tolist() == execs, (k, ex.tolist())
        # and the oracle agrees with the doc comment too
        rok, _, rex = ob.spark_binpack(ob.ALGO_MINIMAL_FRAGMENTATION, avail, drv, exe, k, [6], [a, b, c, d, e, f])
        assert rok and rex.tolist() == execs
    assert not gf_ctx.spark_binpack(MF, drv, exe, 33)[0]  # 32 in total


def test_edge_cases(gf_ctx):
    drv, exe = [1, 1, 0], [1, 1, 0]
    # smallest sufficient capacity wins, ties by priority order (stable sort)
    avail = _caps_cluster([7, 6, 6, 9])
    gf_ctx.set_snapshot(avail)
    gf_ctx.set_orders([4], [0, 1, 2, 3])
    assert gf_ctx.spark_binpack(MF, drv, exe, 5)[2].tolist() == [1] * 5
    # K == 0
    ok, d, ex = gf_ctx.spark_binpack(MF, drv, exe, 0)
    assert ok and d == 4 and len(ex) == 0
    # executor that requests nothing: capacity math.MaxInt, (K + MaxInt) / 2 wraps in Go -> the subset attempt is empty
    gf_ctx.set_snapshot([[5, 5, 0], [1, 1, 0], [-1, 5, 0]])
    gf_ctx.set_orders([0], [2, 1, 0])
    ok, d, ex = gf_ctx.spark_binpack(MF, drv, [0, 0, 0], 3)
    rok, rd, rex = ob.spark_binpack(ob.ALGO_MINIMAL_FRAGMENTATION, [[5, 5, 0], [1, 1, 0], [-1, 5, 0]], drv, [0, 0, 0],
                                    3, [0], [2, 1, 0])
    assert (ok, d, ex.tolist()) == (rok, rd, rex.tolist()) == (True, 0, [1, 1, 1])
    # the driver's own node loses capacity to the driver reservation
    gf_ctx.set_snapshot([[4, 4, 0], [3, 9, 0]])
    gf_ctx.set_orders([0, 1], [0, 1])
    ok, d, ex = gf_ctx.spark_binpack(MF, drv, exe, 3)
    assert ok and d == 0 and ex.tolist() == [0, 0, 0]  # caps (3, 3): first of the smallest sufficient
    # huge capacities (1-byte executors against GiB nodes, like the reference's test pods)
    gf_ctx.set_snapshot([[8000, 8 * GIB, 1], [8000, 8 * GIB + 5, 1]])
    gf_ctx.set_orders([0, 1], [1, 0])
    ok, d, ex = gf_ctx.spark_binpack(MF, [1000, 1, 1], [0, 1, 0], 4)
    assert ok and d == 0 and ex.tolist() == [0] * 4  # caps: node1 8Gi+5, node0 8Gi-1 -> node0 is the smaller one


@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129, 500, 1000])
def test_independent_batch_random(gf_ctx, n, layout):
    rng = np.random.default_rng(4242 + n + 7 * len(layout))
    for tight_cluster in (True, False):
        a = 200
        avail, D, X, drv, exe, k = _random_problem(rng, n, a, tight_cluster, layout)
        if not tight_cluster:  # gangs that need several capacity levels
            k = np.minimum(k, rng.integers(0, 30000, size=a)).astype(np.int32)
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders(D, X)
        apps = _gpu_apps(drv, exe, k)
        gpu = gf_ctx.fit_batch(IND, MF, apps)
        ref = ob.fit_independent(ob.ALGO_MINIMAL_FRAGMENTATION, avail, ob.make_apps(drv, exe, k), D, X)
        _assert_same(gpu, ref, apps)
        if n >= 63 and layout != "merged":
            assert ref.results["has_capacity"].any()


@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("n", [3, 64, 200, 1500])
def test_fifo_chain_random(gf_ctx, n, layout):
    rng = np.random.default_rng(99 + n + 7 * len(layout))
    for rep in range(3):
        a = 100
        avail, D, X, drv, exe, k = _random_problem(rng, n, a, tight_cluster=(rep == 2), layout=layout)
        exe = np.maximum(exe, 1)
        k = np.minimum(k, 60).astype(np.int32)
        flags = (rng.random(a) < (0.9 if rep else 1.0)).astype(np.uint32)
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders(D, X)
        apps = _gpu_apps(drv, exe, k, flags)
        gpu = gf_ctx.fit_batch(FIFO, MF, apps)
        ref = ob.fit_fifo_chain(ob.ALGO_MINIMAL_FRAGMENTATION, avail, ob.make_apps(drv, exe, k, flags), D, X)
        assert gpu.failed_at == ref.failed_at
        _assert_same(gpu, ref, apps)
        assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("n", [2, 65, 300, 1000])
def test_single_az_minimal_fragmentation_random(gf_ctx, n, layout):
    rng = np.random.default_rng(555 + n + 5 * len(layout))
    for tight_cluster in (True, False):
        for n_zones in (1, 3):
            a = 110
            avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, tight_cluster, layout, n_zones)
            _setup(gf_ctx, avail, sched, zone, D, X)
            apps = gangfit.make_apps(drv, exe, k)
            gpu = gf_ctx.fit_batch(IND, SAZMF, apps)
            ref = ob.fit_independent(ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION, avail, ob.make_apps(drv, exe, k), D, X,
                                     sched=sched, zone=zone)
            _assert_same(gpu, ref, apps)
            # chooseBestResult compared these averages (driver-only `reserved`), bit for bit
            avg = gf_ctx.avg_packing_efficiency(SAZMF, apps, gpu)
            assert np.array_equal(_bits(avg), _bits(ref.avg_eff))
            # FIFO chain with the same packer
            flags = (rng.random(a) < 0.9).astype(np.uint32)
            k2 = np.minimum(k, 50).astype(np.int32)
            exe2 = np.maximum(exe, 1)
            apps2 = gangfit.make_apps(drv, exe2, k2, flags)
            gpu = gf_ctx.fit_batch(FIFO, SAZMF, apps2)
            ref = ob.fit_fifo_chain(ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION, avail, ob.make_apps(drv, exe2, k2, flags), D,
                                    X, sched=sched, zone=zone)
            assert gpu.failed_at == ref.failed_at
            _assert_same(gpu, ref, apps2)
            assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.parametrize("cap_hi", [3, 12, 60, 255, 300])
@pytest.mark.parametrize("n", [5, 64, 65, 700, 3000])
def test_histogram_form_level_walks(gf_ctx, n, cap_hi):
    """The independent batch in its histogram form (gangfit_minfrag.inc: wave_minfrag_hist — scaled int32 table, capacities below
    256 counted per value, the level walk planned on the counts and emitted by ONE pass): clusters of small whole capacities, many
    nodes per level, gangs from one executor up to most of the cluster — single-node endings, complete and partial drains, the
    "first undrained node of the last level" ending, the subset rule; cap_hi 255 / 300 put capacities at and beyond the last bin
    (300: the walk on the wide table takes over).  Merged layout (D = X = every node), with and without zones."""
    rng = np.random.default_rng(31337 + 17 * n + cap_hi)
    caps = rng.integers(0, cap_hi + 1, size=n)
    caps[rng.random(n) < 0.05] = -1  # overcommitted nodes
    avail = np.stack([caps, np.full(n, 1000), rng.integers(0, 2, size=n)], axis=1).astype(np.int64)
    order = rng.permutation(n).astype(np.uint32)
    a = 150
    total = int(np.maximum(caps, 0).sum())
    k = rng.integers(1, max(2, total), size=a)
    k[: a // 3] = rng.integers(1, max(2, min(total, 3 * cap_hi)), size=a // 3)  # the gangs one or two nodes take
    k = np.minimum(k, 100000).astype(np.int32)
    drv = np.stack([rng.integers(0, 3, size=a), rng.integers(0, 50, size=a), np.zeros(a, dtype=np.int64)], axis=1).astype(np.int64)
    exe = np.stack([rng.integers(1, 3, size=a), rng.integers(0, 9, size=a), np.zeros(a, dtype=np.int64)], axis=1).astype(np.int64)
    gf_ctx.set_snapshot(avail)
    gf_ctx.set_orders(order, order)
    apps = _gpu_apps(drv, exe, k)
    gpu = gf_ctx.fit_batch(IND, MF, apps)
    ref = ob.fit_independent(ob.ALGO_MINIMAL_FRAGMENTATION, avail, ob.make_apps(drv, exe, k), order, order)
    _assert_same(gpu, ref, apps)
    if total > 0:
        assert ref.results["has_capacity"].any()
    # the same cluster in three zones through the registered packer (one wavefront and one histogram per candidate zone)
    zone = rng.integers(0, 3, size=n).astype(np.uint32)
    sched = np.maximum(avail, 1) + 5
    _setup(gf_ctx, avail, sched, zone, order, order)
    k3 = np.minimum(k, max(1, total // 4)).astype(np.int32)
    apps3 = gangfit.make_apps(drv, exe, k3)
    gpu = gf_ctx.fit_batch(IND, SAZMF, apps3)
    ref = ob.fit_independent(ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION, avail, ob.make_apps(drv, exe, k3), order, order, sched=sched,
                             zone=zone)
    _assert_same(gpu, ref, apps3)


def test_headline_size(gf_ctx):
    """10 000 nodes x 1 000 apps (3 zones for the single-AZ wrapper) against the literal oracle."""
    w = wl.headline(10000, 1000)
    s = w.snapshot
    zone = (wl.splitmix64(0xA4, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    _setup(gf_ctx, s.avail, s.sched, zone, s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k)
    oapps = ob.make_apps(w.drv, w.exe, w.k)
    for algo, oalgo in ((MF, ob.ALGO_MINIMAL_FRAGMENTATION), (SAZMF, ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION)):
        gpu = gf_ctx.fit_batch(IND, algo, apps)
        ref = ob.fit_independent(oalgo, s.avail, oapps, s.driver_order, s.exec_order, sched=s.sched, zone=zone)
        _assert_same(gpu, ref, apps)
        assert ref.results["has_capacity"].mean() > 0.5


@pytest.mark.parametrize("env", [{}, {"fifo_generic": 1}, {"lds_budget": 60000},
                                 {"minfrag_matrix": 0}, {"minfrag_hist": 0}, {"lds_budget": 30000}],
                         ids=["lds-chain", "generic-chain", "lds-chain-global-tail", "lds-chain-no-capacity-matrix",
                              "lds-chain-block-passes", "lds-chain-few-index-rows"])
@pytest.mark.parametrize("algo", [MF, SAZMF])
def test_fifo_chain_kernel_variants(algo, env):
    """The block-cooperative LDS chain (gangfit_fifo_minfrag.inc), the generic global-memory chain and the hybrid
    LDS/global table on the same problems: gangs that need one node, several capacity levels, and more than 64 nodes
    (spilled run lists); a request without a scaled form forces the wide fallback."""
    ctx = gangfit.Context(0, options=env)
    oalgo = ob.ALGO_MINIMAL_FRAGMENTATION if algo == MF else ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION
    rng = np.random.default_rng(77 + algo)
    try:
        for rep in range(5):
            n, a = (3000, 60) if rep < 2 else (900, 120)
            avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, rep % 2 == 0, "merged", 1 + rep % 3)
            exe = np.maximum(exe, 1)
            k = np.minimum(k, 600 if rep < 2 else 40).astype(np.int32)
            flags = (rng.random(a) < 0.9).astype(np.uint32)
            if rep == 4:
                drv[3, 0] += 1  # not a multiple of the table's cpu unit
            _setup(ctx, avail, sched, zone, D, X)
            apps = gangfit.make_apps(drv, exe, k, flags)
            gpu = ctx.fit_batch(FIFO, algo, apps)
            ref = ob.fit_fifo_chain(oalgo, avail, ob.make_apps(drv, exe, k, flags), D, X, sched=sched, zone=zone)
            assert gpu.failed_at == ref.failed_at
            _assert_same(gpu, ref, apps)
            assert np.array_equal(ctx.residual(), ref.avail_after)
    finally:
        ctx.close()


@pytest.mark.parametrize("env", [{}, {"minfrag_hist": 0}], ids=["histograms", "block-passes"])
@pytest.mark.parametrize("algo", [MF, SAZMF])
def test_fifo_chain_histogram_path(algo, env):
    """What the histogram path of gangfit_fifo_minfrag.inc has to get right: the AZ-major priority order of the reference
    (zones are contiguous ranges), a handful of templates (rows and histograms reused and patched across hundreds of commits),
    tiny executor requests next to them (capacities of 256 and more: those shapes keep the block-cooperative passes, in the
    same chain), drivers that share a node with their executors, level walks over many levels and gangs of several hundred."""
    ctx = gangfit.Context(0, options=env)
    oalgo = ob.ALGO_MINIMAL_FRAGMENTATION if algo == MF else ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION
    rng = np.random.default_rng(4242 + algo)
    try:
        for rep in range(4):
            n, a = (6000, 400) if rep < 2 else (1200, 300)
            w = wl.headline(n, a, seed=0x77 + rep)
            s = w.snapshot
            zone = (wl.splitmix64(0xB1 + rep, n, 3) % np.uint64(3 if algo == SAZMF else 1)).astype(np.uint32)
            order = wl.reference_node_order(s.avail, zone if algo == SAZMF else None)
            drv, exe, k = w.drv.copy(), w.exe.copy(), w.k.copy()
            t = rng.integers(0, 9, size=a)  # nine templates
            drv, exe = drv[t], exe[t]
            if rep % 2 == 1:
                tiny = rng.random(a) < 0.1
                exe[tiny] = np.array([100, 256 << 20, 0])  # hundreds of executors per node
                k = np.where(rng.random(a) < 0.15, rng.integers(100, 500, size=a), k).astype(np.int32)
            if rep >= 2:  # a drained cluster: many small capacities, gangs spread over many levels
                k = np.minimum(k * 6, 400).astype(np.int32)
            flags = (rng.random(a) < 0.9).astype(np.uint32)
            _setup(ctx, s.avail, s.sched, zone, order, order)
            apps = gangfit.make_apps(drv, exe, k, flags)
            gpu = ctx.fit_batch(FIFO, algo, apps)
            ref = ob.fit_fifo_chain(oalgo, s.avail, ob.make_apps(drv, exe, k, flags), order, order, sched=s.sched, zone=zone)
            assert gpu.failed_at == ref.failed_at
            _assert_same(gpu, ref, apps)
            assert np.array_equal(ctx.residual(), ref.avail_after)
            assert ref.results["has_capacity"].sum() > 20
    finally:
        ctx.close()
