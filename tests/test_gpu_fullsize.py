"""Parity at BASELINE.json's full sizes for configs 4 and 5, through the C ABI:

  C4  50 000 nodes x 10 000 pending apps, dynamic allocation: the gang is MinExecutorCount (quirk 6), then (max - min)
      single-executor first fits for 10 % of the apps (rescheduleExecutor's loop, resource.go:658-662); independent batch
      on one context, and node-range sharded over 8 shards (gf_shard_*).
  C5  100 000 nodes, 20 000 ResourceReservations (K + 1 entries each) replayed by gf_snapshot_build, then the FIFO chain of
      999 earlier drivers + 1 with 5 % of them skippable (resource.go:224-262, 264-270), residuals included.

The LITERAL oracle is affordable at these sizes (its loops are as lazy as the reference's), so everything is compared
bit for bit with it; the closed form is cross-checked on a slice.  Size-independent properties are asserted on top:
placements never exceed what the oracle's residual arithmetic allows, and the residual equals snapshot minus the replayed
usage (sparkResourceUsage quirk) recomputed in numpy from the GPU's own placements."""
import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob
from oracle import pysnapshot as ps
from test_gpu_parity import _assert_same

pytestmark = pytest.mark.gpu
GIB = 1 << 30
IND, FIFO = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_MODE_FIFO_CHAIN


@pytest.fixture(scope="module")
def c4():
    return wl.config(4)


@pytest.mark.parametrize("algo", [0, 1])
def test_config4_independent_50k_x_10k(gf_ctx, c4, algo):
    s = c4.snapshot
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(c4.drv, c4.exe, c4.k)
    oapps = ob.make_apps(c4.drv, c4.exe, c4.k)
    gpu = gf_ctx.fit_batch(IND, algo, apps)
    ref = ob.fit_independent(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=False)
    _assert_same(gpu, ref, apps)
    closed = ob.fit_independent(algo, s.avail, oapps[:512], s.driver_order, s.exec_order, closed_form=True)
    assert np.array_equal(closed.results, ref.results[:512])
    # a harder variant of the same size: the cluster nearly full, so that many gangs need the driver fallback / do not fit
    wc = wl.Workload("C4 congested", wl.make_snapshot(50000, 0x5EED0004, 0.93, 1.0), c4.drv, c4.exe, c4.k, c4.k_max, c4.flags)
    sc = wc.snapshot
    gf_ctx.set_snapshot(sc.avail, sc.sched)
    gf_ctx.set_orders(sc.driver_order, sc.exec_order)
    # an infeasible decision costs O(|D| * N) = 2.5e9 steps in the literal loop: closed form on 2 000 apps, literal on the
    # first infeasible one and its feasible predecessor only
    gpu = gf_ctx.fit_batch(IND, algo, apps)
    ref = ob.fit_independent(algo, sc.avail, oapps[:2000], sc.driver_order, sc.exec_order, closed_form=True)
    assert 0.05 < ref.results["has_capacity"].mean() < 0.95
    assert np.array_equal(gpu.results[:2000], ref.results)
    for a in np.nonzero(ref.results["has_capacity"])[0]:
        assert np.array_equal(gpu.placement(int(a))[2], ref.placement(int(a))[2])
    bad = int(np.argmin(ref.results["has_capacity"]))
    lo = max(bad - 1, 0)
    lit = ob.fit_independent(algo, sc.avail, oapps[lo:bad + 1], sc.driver_order, sc.exec_order, closed_form=False)
    assert np.array_equal(lit.results, ref.results[lo:bad + 1])


def test_config4_extra_executor_first_fits(gf_ctx, c4):
    """(max - min) single executors for every tenth app: ~32 000 independent first fits over 50 000 nodes."""
    s = c4.snapshot
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    extra = np.repeat(c4.exe[::10], np.maximum(c4.k_max[::10] - c4.k[::10], 0), axis=0)
    assert len(extra) > 20000
    got = gf_ctx.executor_fit(extra)
    uniq, inv = np.unique(extra, axis=0, return_inverse=True)
    want = np.array([ob.executor_fit(s.avail, e, s.exec_order) for e in uniq], dtype=np.uint32)[inv.reshape(-1)]
    assert np.array_equal(got, want)
    # property: the chosen node fits, and no earlier node of the order does
    pos = np.empty(len(s.avail), dtype=np.int64)
    pos[s.exec_order] = np.arange(len(s.exec_order))
    fits = lambda e: (s.avail[s.exec_order] >= e).all(axis=1)
    for e, node in zip(uniq, np.array([ob.executor_fit(s.avail, e, s.exec_order) for e in uniq])):
        f = fits(e)
        assert (node == 0xFFFFFFFF and not f.any()) or int(np.argmax(f)) == pos[node]


@pytest.mark.parametrize("algo", [0, 1])
def test_config4_eight_node_range_shards(c4, algo):
    from test_gpu_sharded import _run

    s = c4.snapshot
    apps = gangfit.make_apps(c4.drv, c4.exe, c4.k)
    ref = ob.fit_independent(algo, s.avail, ob.make_apps(c4.drv, c4.exe, c4.k), s.driver_order, s.exec_order, closed_form=False)
    outs = _run(8, algo, s.avail, s.driver_order, s.exec_order, apps)
    _assert_same(outs[0], ref, apps)
    _assert_same(outs[5], ref, apps)


def _c5_cluster():
    """Config 5's snapshot inputs: 100 000 nodes of the C2 shapes, 20 000 ResourceReservations of K + 1 entries."""
    n = 100000
    rng = np.random.default_rng(0x5EED0005)
    shape = rng.integers(0, 4, size=n)
    alloc = np.stack([np.array([16, 32, 64, 96])[shape] * 1000, np.array([64, 128, 256, 384])[shape] * GIB,
                      np.where(rng.random(n) < 0.1, 8, 0)], axis=1).astype(np.int64)
    ks = rng.integers(2, 26, size=20000)
    # reservations land on a third of the cluster so that the front of the priority order is really used up
    res_node = rng.integers(0, n // 3, size=int(ks.sum())).astype(np.uint32)
    res_req = np.stack([rng.choice([1000, 2000, 4000, 8000], size=len(res_node)), rng.choice([4, 8, 16, 32], size=len(res_node)) * GIB,
                        (rng.random(len(res_node)) < 0.02).astype(np.int64)], axis=1).astype(np.int64)
    flags = (np.where(rng.random(n) < 0.02, ps.UNSCHEDULABLE, 0) | np.where(rng.random(n) < 0.98, ps.READY, 0) |
             np.where(rng.random(n) < 0.9, ps.DRIVER_CANDIDATE, 0)).astype(np.uint32)
    return dict(alloc=alloc, node_flags=flags, name_rank=rng.permutation(n).astype(np.uint32), res_node=res_node, res_req=res_req,
                zone=rng.integers(0, 3, size=n).astype(np.uint32), n_zones=3)


def _replayed_usage(n_nodes, apps, results, exec_off, exec_nodes):
    """sparkResourceUsage + SubtractUsageIfExists (sparkpods.go:139-146) recomputed in numpy from the given placements:
    ONE executor request per distinct executor node, the driver request only if its node hosts no executor; the last app
    (the driver being filtered) is not subtracted."""
    usage = np.zeros((n_nodes, 3), dtype=np.int64)
    for a in range(len(apps) - 1):
        r = results[a]
        if not r["has_capacity"]:
            continue
        nodes = np.unique(exec_nodes[int(exec_off[a]):int(exec_off[a]) + int(r["exec_len"])])
        usage[nodes] += apps["exe"][a]
        if int(r["driver_node"]) not in set(nodes.tolist()):
            usage[int(r["driver_node"])] += apps["drv"][a]
    return usage


@pytest.mark.parametrize("algo", [0, 1, 4])
def test_config5_replay_then_fifo_999_plus_1(gf_ctx, algo):
    c = _c5_cluster()
    w5 = wl.config(5)
    D, X = gf_ctx.build_snapshot(**c)
    avail, sched, rD, rX = ps.build(**c)
    got_avail, got_sched = gf_ctx.snapshot()
    assert np.array_equal(got_avail, avail) and np.array_equal(got_sched, sched)
    assert np.array_equal(D, rD) and np.array_equal(X, rX)
    assert int(w5.flags.sum()) >= 30  # 5 % skippable
    zone = c["zone"]
    for variant in ("as specified", "with unfit earlier drivers"):
        drv, exe, k, flags = w5.drv.copy(), w5.exe.copy(), w5.k.copy(), w5.flags.copy()
        if variant != "as specified":
            # gangs no cluster of this shape can host: skippable ones are ignored (resource.go:244-248), the first
            # non-skippable one fails the request with "failure-earlier-driver" (:249-251) and nothing behind it is evaluated
            for a, skippable in ((100, 1), (400, 1), (401, 1), (800, 0), (900, 1)):
                exe[a] = [64000, 300 * GIB, 8]
                k[a] = 700
                flags[a] = skippable
        apps = gangfit.make_apps(drv, exe, k, flags)
        oapps = ob.make_apps(drv, exe, k, flags)
        gpu = gf_ctx.fit_batch(FIFO, algo, apps)
        # an unfit gang costs the literal loop O(|D| * N) = 1e10 steps here: the variant with unfit drivers uses the closed
        # form (tests/test_oracle*.py require literal == closed form everywhere)
        ref = ob.fit_fifo_chain(algo, avail, oapps, rD, rX, sched=sched, zone=zone if algo == 4 else None,
                                closed_form=(variant != "as specified"))
        assert gpu.failed_at == ref.failed_at == (-1 if variant == "as specified" else 800)
        assert np.array_equal(gpu.results, ref.results), variant
        for a in np.nonzero(ref.results["has_capacity"])[0]:
            assert np.array_equal(gpu.placement(int(a))[2], ref.placement(int(a))[2]), (variant, int(a))
        residual = gf_ctx.residual()
        assert np.array_equal(residual, ref.avail_after)
        # size-independent property: residual == snapshot - usage replayed from the GPU's own placements
        evaluated = gpu.results.copy()
        if gpu.failed_at >= 0:
            evaluated["has_capacity"][gpu.failed_at:] = 0
        usage = _replayed_usage(len(avail), apps, evaluated, gpu.exec_off, gpu.exec_nodes)
        assert np.array_equal(residual, avail - usage)
        if variant != "as specified":
            assert not gpu.results["evaluated"][801:].any() and gpu.results["evaluated"][:801].all()
            # the plain packers can still host the first such gangs on the gpu nodes the replay left empty; the one that
            # aborts the chain cannot be hosted by any packer
            assert not gpu.results["has_capacity"][800] and not gpu.results["has_capacity"][401]


def test_100k_nodes_three_zones_chains_stay_on_the_lds_kernels():
    """The LDS-resident chain kernels of the zone-aware and minimal-fragmentation packers fit next to the per-view masks of a
    100 000-node table with a few dozen bytes of LDS to spare (MfShared, gangfit_fifo_minfrag.inc).  A kilobyte more and the
    queue silently takes the generic kernel: the same answers, 16 ms -> 1.3 s.  Nothing else notices, so this does: every
    packer's cold 1 000-application chain at that size stays far below what the generic kernel needs, and the
    single-AZ minimal-fragmentation chain agrees with the oracle on a prefix."""
    import time
    n_nodes, nz = 100000, 3
    w = wl.headline(n_nodes, 1000)
    s = w.snapshot
    zone = (wl.splitmix64(0xA3, n_nodes, 9) % np.uint64(nz)).astype(np.uint32)
    order = wl.reference_node_order(s.avail, zone)
    flags = np.ones(len(w.k), dtype=np.uint32)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, flags)
    with gangfit.Context(0, options={"chain_cache": 0}) as ctx:
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_zones(zone)
        ctx.set_orders(order, order)
        for algo in (0, 1, 2, 3, 4, 5):
            ctx.fit_batch(FIFO, algo, apps)
            t0 = time.perf_counter()
            gpu = ctx.fit_batch(FIFO, algo, apps)
            ms = (time.perf_counter() - t0) * 1e3
            assert ms < 150.0, (algo, ms)
            if algo == 5:
                n = 40
                ref = ob.fit_fifo_chain(ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION, s.avail, ob.make_apps(w.drv[:n], w.exe[:n], w.k[:n], flags[:n]),
                                        order, order, sched=s.sched, zone=zone)
                assert np.array_equal(gpu.results[:n - 1], ref.results[:n - 1])
