"""One gf_ctx over several devices (gf_init with n_dev > 1, include/gangfit.h): an independent batch of a plain packer is
node-range sharded INSIDE the library — four device steps per shard, exchanges by peer access — and must return exactly
what one device returns.  The one-GPU box exercises it with a repeated device id (N shards on cuda:0); everything else a
multi-device context is asked for (FIFO chains, zone-aware packers, single executors, findNodes) runs on its first device."""
import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob
from test_gpu_parity import _assert_same, _random_problem

pytestmark = pytest.mark.gpu
IND, FIFO = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_MODE_FIFO_CHAIN


@pytest.fixture(params=["one-sub-context-per-device", "one-per-listed-id"])
def split(request, monkeypatch):
    """A repeated device id is how a one-GPU box exercises the multi-device context.  By default the shards of ONE device live in
    one sub-context (one launch per step, a grid row per shard); GANGFIT_TEST_GROUP_SPLIT=1 gives every listed id a sub-context,
    a stream and a submitting thread of its own — events between streams, the host barriers of the submitting threads, pushes
    into several gathered tables, the pull of the placements: what distinct devices exercise."""
    if request.param == "one-per-listed-id":
        monkeypatch.setenv("GANGFIT_TEST_GROUP_SPLIT", "1")
    else:
        monkeypatch.delenv("GANGFIT_TEST_GROUP_SPLIT", raising=False)
    return request.param == "one-per-listed-id"


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("n_dev", [2, 3, 8, 16])
@pytest.mark.parametrize("n", [5, 64, 130, 1000])
def test_group_matches_oracle(algo, n_dev, n, split):
    rng = np.random.default_rng(4321 + 17 * n_dev + algo + n)
    with gangfit.Context(devices=[0] * n_dev) as g:
        assert g.shard_count() == n_dev
        for layout in ("merged", "identical"):
            for tight_cluster in (True, False):
                avail, D, X, drv, exe, k = _random_problem(rng, n, 150, tight_cluster, layout)
                g.set_snapshot(avail)
                g.set_orders(D, X)
                apps = gangfit.make_apps(drv, exe, k)
                ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
                _assert_same(g.fit_batch(IND, algo, apps), ref, apps)


def _device_count():
    import torch

    return torch.cuda.device_count()


@pytest.mark.parametrize("exchange", ["peer_stores", "rccl"])
@pytest.mark.parametrize("algo", [0, 1])
def test_group_over_distinct_devices(algo, exchange):
    """The same context over DISTINCT device ids — the first box with two GPUs exercises hipDeviceEnablePeerAccess, the posted peer
    stores into every device's gathered table and the pull of the placements (csrc/gangfit_api_group.cpp), and, exchange = rccl,
    ncclCommInitAll over more than one rank.  One-GPU boxes skip it (what they can run is the repeated-id form above).  A topology
    without peer access degrades to the first device (shard_count 1): still the oracle's answers, and the test says which it saw."""
    n_dev = _device_count()
    if n_dev < 2:
        pytest.skip(f"{n_dev} visible GPU(s): distinct device ids need two")
    ids = list(range(min(n_dev, 8)))
    rng = np.random.default_rng(777 + algo)
    with gangfit.Context(devices=ids) as g:
        sharded = g.shard_count() == len(ids)
        assert sharded or g.shard_count() == 1  # degraded: no peer access between some pair (gf_last_error says which)
        if exchange == "rccl":
            try:
                g.set_option("group_exchange", 1)
            except gangfit.GangfitError as e:
                pytest.skip(f"RCCL exchange unavailable here: {e}")
        for n in (130, 5000):
            for layout in ("merged", "identical"):
                avail, D, X, drv, exe, k = _random_problem(rng, n, 200, n == 130, layout)
                g.set_snapshot(avail)
                g.set_orders(D, X)
                apps = gangfit.make_apps(drv, exe, k)
                ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
                _assert_same(g.fit_batch(IND, algo, apps), ref, apps)
                _assert_same(g.fit_batch(IND, algo, apps), ref, apps)  # the second batch on a snapshot runs without the self-check
        assert g.shard_count() in (1, len(ids)), "the self-check of a sharded batch failed on distinct devices"
        if sharded:
            assert g.shard_count() == len(ids), "the context stopped sharding: its first sharded batch disagreed with one device"


def test_group_general_layout_and_other_modes_run_on_the_first_device():
    rng = np.random.default_rng(99)
    avail, D, X, drv, exe, k = _random_problem(rng, 300, 60, True, "general")
    sched = np.abs(avail) + 5
    zone = rng.integers(0, 3, size=len(avail)).astype(np.uint32)
    apps = gangfit.make_apps(drv, exe, k, np.ones(len(k), dtype=np.uint32))
    oapps = ob.make_apps(drv, exe, k, np.ones(len(k), dtype=np.uint32))
    with gangfit.Context(devices=[0, 0, 0]) as g:
        g.set_snapshot(avail, sched)
        g.set_zones(zone)
        g.set_orders(D, X)
        for algo in (0, 1, 2, 4):
            ref = ob.fit_independent(algo, avail, oapps, D, X, sched=sched, zone=zone)
            _assert_same(g.fit_batch(IND, algo, apps), ref, apps)
            ref = ob.fit_fifo_chain(algo, avail, oapps, D, X, sched=sched, zone=zone)
            out = g.fit_batch(FIFO, algo, apps)
            assert out.failed_at == ref.failed_at and np.array_equal(out.results, ref.results)
            assert np.array_equal(g.residual(), ref.avail_after)
        Xk = X[X < len(avail)]
        assert g.executor_fit(exe[:8]).tolist() == [ob.executor_fit(avail, e, X) for e in exe[:8]]
        placed, _, _, _, adds = g.find_nodes(exe[:5], k[:5], chained=True)
        want = ob.find_nodes(avail, exe[:5], k[:5], Xk)
        assert np.array_equal(placed, want.placed) and np.array_equal(adds, want.adds)
        assert g.device_info()["arch"].startswith("gfx950") and g.selftest(3, 32) == 0
        with pytest.raises(gangfit.GangfitError):  # the shard steps of a group are not the caller's to drive
            g._check(g._lib.gf_shard_set(g._h, 0, 2))


def test_group_device_built_snapshot_and_headline_size(split):
    w = wl.headline(10000, 1000)
    s = w.snapshot
    apps = gangfit.make_apps(w.drv, w.exe, w.k)
    oapps = ob.make_apps(w.drv, w.exe, w.k)
    with gangfit.Context(devices=[0] * 8) as g:
        g.set_snapshot(s.avail, s.sched)
        g.set_orders(s.driver_order, s.exec_order)
        for algo in (0, 1):
            ref = ob.fit_independent(algo, s.avail, oapps, s.driver_order, s.exec_order, closed_form=False)
            _assert_same(g.fit_batch(IND, algo, apps), ref, apps)
        # a snapshot built on the device installs itself on every sub-context
        rng = np.random.default_rng(12)
        n = 5000
        alloc = np.stack([rng.choice([16, 32, 64], size=n) * 1000, rng.choice([64, 128, 256], size=n) * (1 << 30),
                          np.zeros(n, dtype=np.int64)], axis=1).astype(np.int64)
        rnode = rng.integers(0, n, size=30000).astype(np.uint32)
        rreq = np.stack([rng.choice([1000, 2000, 4000], size=30000), rng.choice([4, 8, 16], size=30000) * (1 << 30),
                         np.zeros(30000, dtype=np.int64)], axis=1).astype(np.int64)
        D, X = g.build_snapshot(alloc, np.full(n, 6, dtype=np.uint32), rng.permutation(n).astype(np.uint32), res_node=rnode,
                                res_req=rreq)
        avail, _ = g.snapshot()
        ref = ob.fit_independent(0, avail, oapps, D, X, closed_form=True)
        out = g.fit_batch(IND, 0, apps)
        assert 0 < ref.results["has_capacity"].sum()
        _assert_same(out, ref, apps)
        # the resident flow on a multi-device context: the cluster columns and the usage deltas go to every device
        ranks = rng.permutation(n).astype(np.uint32)
        g.set_cluster(alloc, np.full(n, 6, dtype=np.uint32), ranks)
        g.usage_apply(rnode[:20000], rreq[:20000], +1)
        g.usage_apply(rnode[20000:], rreq[20000:], +1)
        g.usage_apply(rnode[5000:9000], rreq[5000:9000], -1)
        D2, X2 = g.build_snapshot_resident(resident_usage=True)
        keep = np.r_[0:5000, 9000:30000]
        with gangfit.Context(0) as one:
            D1, X1 = one.build_snapshot(alloc, np.full(n, 6, dtype=np.uint32), ranks, res_node=rnode[keep], res_req=rreq[keep])
            avail1, _ = one.snapshot()
        assert np.array_equal(D2, D1) and np.array_equal(X2, X1) and np.array_equal(g.snapshot()[0], avail1)
        ref = ob.fit_independent(1, avail1, oapps, D1, X1, closed_form=True)
        _assert_same(g.fit_batch(IND, 1, apps), ref, apps)


def test_group_argument_errors():
    lib = gangfit._native.load()
    import ctypes as C

    h = C.c_void_p()
    ids = (C.c_int * 17)(*([0] * 17))
    assert lib.gf_init(ids, 17, C.byref(h)) == gangfit._native.GF_ERR_INVALID  # more than 16 devices
    ids = (C.c_int * 2)(0, 999)
    assert lib.gf_init(ids, 2, C.byref(h)) == gangfit._native.GF_ERR_NO_DEVICE


@pytest.mark.parametrize("fault", [1, 2])
def test_a_wrong_exchange_is_caught_and_sharding_switched_off(fault, monkeypatch):
    """Self-check of a multi-device context: the first sharded batch on a newly installed snapshot is also answered by the
    first device alone.  With an exchange made to fail (option "group_fault": 1 = the placement reduction never runs, 2 = the
    other shards' capacity sums arrive as zeros) the answers must still be the right ones, the context must say so and stop
    sharding; without the fault it keeps sharding.  (Every listed id its own sub-context: with the four shards in ONE
    sub-context there is no exchange between devices to break.)"""
    monkeypatch.setenv("GANGFIT_TEST_GROUP_SPLIT", "1")
    rng = np.random.default_rng(700 + fault)
    avail, D, X, drv, exe, k = _random_problem(rng, 1000, 150, False, "merged")
    apps = gangfit.make_apps(drv, exe, k)
    ref = ob.fit_independent(0, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
    with gangfit.Context(devices=[0] * 4) as g:
        g.set_snapshot(avail)
        g.set_orders(D, X)
        _assert_same(g.fit_batch(IND, 0, apps), ref, apps)
        assert g.shard_count() == 4
        g.set_option("group_fault", fault)  # re-arms the self-check
        _assert_same(g.fit_batch(IND, 0, apps), ref, apps)
        assert g.shard_count() == 1 and "disagreed" in g.last_error()
        _assert_same(g.fit_batch(IND, 0, apps), ref, apps)  # served by the first device from now on
        g.set_option("group_fault", 0)
        g.set_option("group_shard_off", 0)
        _assert_same(g.fit_batch(IND, 0, apps), ref, apps)
        assert g.shard_count() == 4
        # the check runs once per installed snapshot: with it switched off a faulty exchange would go unnoticed
        g.set_option("group_verify", 0)
        g.set_option("group_fault", fault)
        out = g.fit_batch(IND, 0, apps)
        assert not (np.array_equal(out.results, ref.results) and all(
            np.array_equal(out.placement(int(a))[2], ref.placement(int(a))[2]) for a in np.nonzero(ref.results["has_capacity"])[0]))


def test_no_peer_access_degrades_to_the_first_device(monkeypatch):
    """Devices that cannot reach each other's memory: gf_init returns a working single-device context (GF_OK), not an error
    for the whole context — the host would otherwise fall back to the CPU for every Filter."""
    monkeypatch.setenv("GANGFIT_TEST_NO_PEER", "1")
    rng = np.random.default_rng(31)
    avail, D, X, drv, exe, k = _random_problem(rng, 500, 80, False, "merged")
    with gangfit.Context(devices=[0, 0, 0]) as g:
        assert g.shard_count() == 1 and "peer access" in g.last_error()
        g.set_snapshot(avail)
        g.set_orders(D, X)
        apps = gangfit.make_apps(drv, exe, k)
        _assert_same(g.fit_batch(IND, 1, apps), ob.fit_independent(1, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True), apps)


def test_many_batches_keep_the_submitting_threads_in_step(monkeypatch):
    """Two hundred sharded batches of changing size back to back on one context (every listed id its own sub-context and
    thread): the host barriers between the steps and the event waits between the streams must hold for every one of them."""
    monkeypatch.setenv("GANGFIT_TEST_GROUP_SPLIT", "1")
    rng = np.random.default_rng(77)
    avail, D, X, drv, exe, k = _random_problem(rng, 1500, 400, False, "merged")
    with gangfit.Context(devices=[0] * 5) as g:
        g.set_snapshot(avail)
        g.set_orders(D, X)
        g.set_option("group_verify", 0)  # the answers are compared here, every time
        for i in range(200):
            n = int(rng.integers(1, 400))
            lo = int(rng.integers(0, 400 - n + 1))
            algo = i & 1
            apps = gangfit.make_apps(drv[lo:lo + n], exe[lo:lo + n], k[lo:lo + n])
            ref = ob.fit_independent(algo, avail, ob.make_apps(drv[lo:lo + n], exe[lo:lo + n], k[lo:lo + n]), D, X, closed_form=True)
            _assert_same(g.fit_batch(IND, algo, apps), ref, apps)


def test_rccl_binding_and_exchange_selection():
    """The collective library is bound at run time.  One GPU cannot host two ranks of one communicator, so: (a) the binding
    itself is exercised with a one-rank communicator (all-gather + reduce must reproduce their input); (b) asking a context
    whose device id repeats for the RCCL exchange is refused (GF_ERR_UNSUPPORTED) and leaves the peer-store exchange in place."""
    with gangfit.Context(0) as one:
        one.set_option("rccl_selftest", 4096)
    rng = np.random.default_rng(32)
    avail, D, X, drv, exe, k = _random_problem(rng, 400, 60, False, "merged")
    apps = gangfit.make_apps(drv, exe, k)
    ref = ob.fit_independent(0, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
    with gangfit.Context(devices=[0, 0]) as g:
        with pytest.raises(gangfit.GangfitError) as e:
            g.set_option("group_exchange", 1)
        assert e.value.code == gangfit._native.GF_ERR_UNSUPPORTED
        g.set_snapshot(avail)
        g.set_orders(D, X)
        _assert_same(g.fit_batch(IND, 0, apps), ref, apps)
        assert g.shard_count() == 2
