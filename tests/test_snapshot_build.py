"""Snapshot construction on the device (gf_snapshot_build: ResourceReservation replay, available / schedulable,
priority orders) against the numpy restatement in oracle/pysnapshot.py, and the decisions made on the built snapshot
against the oracle's decisions on the restated one."""
import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob
from oracle import pysnapshot as ps

GIB = 1 << 30


def _cluster(seed, n, n_rr, n_zones, with_overhead=True, labels=False):
    rng = np.random.default_rng(seed)
    shape = rng.integers(0, 4, size=n)
    alloc = np.stack([np.array([16, 32, 64, 96])[shape] * 1000, np.array([64, 128, 256, 384])[shape] * GIB,
                      np.where(rng.random(n) < 0.1, 8, 0)], axis=1).astype(np.int64)
    overhead = None
    if with_overhead:
        overhead = np.stack([rng.integers(0, 8, size=n) * 250, rng.integers(0, 16, size=n) * (GIB // 4),
                             np.zeros(n, dtype=np.int64)], axis=1).astype(np.int64)
    # reservations: K+1 entries per ResourceReservation, concentrated so that some nodes are overcommitted
    ks = rng.integers(1, 25, size=n_rr)
    res_node = rng.integers(0, n + 3, size=int(ks.sum())).astype(np.uint32)  # a few entries point outside the listed nodes
    res_req = np.stack([rng.choice([1000, 2000, 4000], size=len(res_node)), rng.choice([4, 8, 16], size=len(res_node)) * GIB,
                        (rng.random(len(res_node)) < 0.02).astype(np.int64)], axis=1).astype(np.int64)
    flags = (np.where(rng.random(n) < 0.05, ps.UNSCHEDULABLE, 0) | np.where(rng.random(n) < 0.95, ps.READY, 0) |
             np.where(rng.random(n) < 0.8, ps.DRIVER_CANDIDATE, 0)).astype(np.uint32)
    name_rank = rng.permutation(n).astype(np.uint32)
    zone = rng.integers(0, n_zones, size=n).astype(np.uint32)
    dl = el = None
    if labels:
        dl = rng.choice([0, 1, ps.UNRANKED], size=n).astype(np.uint32)
        el = rng.choice([0, 1, 2, ps.UNRANKED], size=n).astype(np.uint32)
    return dict(alloc=alloc, node_flags=flags, name_rank=name_rank, overhead=overhead, res_node=res_node, res_req=res_req,
                zone=zone, n_zones=n_zones, driver_label_rank=dl, exec_label_rank=el)


def test_numpy_restatement_small_known_answer():
    """nodesorting_test.go:98-152 (TestAZAwareNodeSorting) through the flat interface: zone2 has less free memory."""
    # names: zone1Node1, zone1Node2, zone1Node3, zone2Node1 -> ranks 0..3; zone ids in label order
    alloc = [[1, 1, 0], [1, 2, 0], [2, 1, 0], [1, 1, 0]]
    _, _, D, X = ps.build(alloc, [ps.READY | ps.DRIVER_CANDIDATE] * 4, [0, 1, 2, 3], zone=[0, 0, 0, 1], n_zones=2)
    assert D.tolist() == [3, 0, 2, 1] and X.tolist() == [3, 0, 2, 1]
    # usage and overhead: available = allocatable - (usage + overhead), schedulable = allocatable - overhead
    avail, sched, D, X = ps.build([[8000, 8 * GIB, 1], [8000, 8 * GIB, 1]], [ps.READY | ps.DRIVER_CANDIDATE, ps.READY], [0, 1],
                                  overhead=[[500, GIB, 0], [0, 0, 0]], res_node=[0, 0, 1, 7], res_req=[[1000, 1, 1], [1000, 1, 0],
                                                                                                      [1000, 1, 0], [9, 9, 9]])
    assert avail.tolist() == [[5500, 7 * GIB - 2, 0], [7000, 8 * GIB - 1, 1]]
    assert sched.tolist() == [[7500, 7 * GIB, 1], [8000, 8 * GIB, 1]]
    assert D.tolist() == [0] and X.tolist() == [0, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("labels", [False, True])
@pytest.mark.parametrize("n,n_rr,n_zones", [(1, 0, 1), (5, 3, 2), (64, 40, 1), (1000, 300, 3), (20000, 4000, 4)])
def test_device_build_matches_restatement(gf_ctx, n, n_rr, n_zones, labels):
    c = _cluster(100 + n + n_zones, n, n_rr, n_zones, with_overhead=(n % 2 == 0), labels=labels)
    D, X = gf_ctx.build_snapshot(**c)
    avail, sched, rD, rX = ps.build(**c)
    got_avail, got_sched = gf_ctx.snapshot()
    assert np.array_equal(got_avail, avail)
    assert np.array_equal(got_sched, sched)
    assert np.array_equal(D, rD) and np.array_equal(X, rX)
    # the decisions on the built snapshot: every registered packer family, independent and FIFO
    w = wl.config(2, n_nodes=16, n_apps=min(64, 4 * n))
    apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    oapps = ob.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
    for algo in (0, 1, 4):
        gpu = gf_ctx.fit_batch(gangfit.GF_MODE_INDEPENDENT, algo, apps)
        ref = ob.fit_independent(algo, avail, oapps, rD, rX, sched=sched, zone=c["zone"])
        assert np.array_equal(gpu.results, ref.results)
        gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, apps)
        ref = ob.fit_fifo_chain(algo, avail, oapps, rD, rX, sched=sched, zone=c["zone"])
        assert gpu.failed_at == ref.failed_at and np.array_equal(gpu.results, ref.results)
        assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.gpu
def test_config5_size_and_argument_errors(gf_ctx):
    """100 000 nodes, 20 000 ResourceReservations (K+1 entries each) — BASELINE config 5's snapshot."""
    c = _cluster(5, 100000, 20000, 3)
    D, X = gf_ctx.build_snapshot(**c)
    avail, sched, rD, rX = ps.build(**c)
    got_avail, got_sched = gf_ctx.snapshot()
    assert np.array_equal(got_avail, avail) and np.array_equal(got_sched, sched)
    assert np.array_equal(D, rD) and np.array_equal(X, rX)
    assert (avail < 0).any()  # the replay overcommits some nodes, as the workload intends
    bad = dict(c)
    bad["name_rank"] = np.zeros(100000, dtype=np.uint32)
    with pytest.raises(gangfit.GangfitError):
        gf_ctx.build_snapshot(**bad)
    bad = dict(c)
    bad["zone"] = np.full(100000, 7, dtype=np.uint32)
    with pytest.raises(gangfit.GangfitError):
        gf_ctx.build_snapshot(**bad)


@pytest.mark.gpu
def test_sums_that_could_wrap_are_refused(gf_ctx):
    """The replay accumulates in 64 bits on the device: a node whose reservations can sum past 2^62 must be refused (the shim
    then falls back to Go), never wrapped.  Large values spread over many nodes are fine."""
    n = 8
    alloc = np.tile(np.array([[64000, 256 * GIB, 0]], dtype=np.int64), (n, 1))
    flags = np.full(n, ps.READY | ps.DRIVER_CANDIDATE, dtype=np.uint32)
    ranks = np.arange(n, dtype=np.uint32)
    big = np.int64(1) << 58
    req = np.tile(np.array([[1000, big, 0]], dtype=np.int64), (20, 1))
    with pytest.raises(gangfit.GangfitError) as e:  # 20 x 2^58 on one node > 2^62
        gf_ctx.build_snapshot(alloc, flags, ranks, res_node=np.zeros(20, dtype=np.uint32), res_req=req)
    assert e.value.code == gangfit._native.GF_ERR_INVALID
    node = (np.arange(20) % n).astype(np.uint32)  # at most 3 per node: 3 x 2^58 < 2^62
    D, X = gf_ctx.build_snapshot(alloc, flags, ranks, res_node=node, res_req=req)
    avail, _, _, _ = ps.build(alloc, flags, ranks, res_node=node, res_req=req)
    got, _ = gf_ctx.snapshot()
    assert np.array_equal(got, avail)


@pytest.mark.gpu
@pytest.mark.parametrize("labels", [False, True])
def test_resident_cluster_then_reservations_only(gf_ctx, labels):
    """gf_cluster_set once, then several gf_snapshot_build_resident calls with different reservations and different candidate
    flags: each must equal gf_snapshot_build / the restatement on the same inputs."""
    n = 3000
    c = _cluster(77, n, 500, 3, with_overhead=True, labels=labels)
    gf_ctx.set_cluster(c["alloc"], c["node_flags"], c["name_rank"], overhead=c["overhead"], zone=c["zone"], n_zones=c["n_zones"])
    rng = np.random.default_rng(5)
    for rep in range(3):
        m = int(rng.integers(0, 9000))
        res_node = rng.integers(0, n + 2, size=m).astype(np.uint32)
        res_req = np.stack([rng.choice([500, 1000, 4000], size=m), rng.choice([1, 4, 16], size=m) * GIB, (rng.random(m) < 0.05).astype(np.int64)],
                           axis=1).astype(np.int64)
        flags = c["node_flags"].copy()
        if rep:  # this request's NodeNames: other driver candidates
            flags = (flags & ~np.uint32(ps.DRIVER_CANDIDATE)) | np.where(rng.random(n) < 0.5, ps.DRIVER_CANDIDATE, 0).astype(np.uint32)
        D, X = gf_ctx.build_snapshot_resident(res_node=res_node, res_req=res_req, node_flags=flags if rep else None,
                                              driver_label_rank=c["driver_label_rank"], exec_label_rank=c["exec_label_rank"])
        cc = dict(c, res_node=res_node, res_req=res_req, node_flags=flags)
        avail, sched, rD, rX = ps.build(**cc)
        got_avail, got_sched = gf_ctx.snapshot()
        assert np.array_equal(got_avail, avail) and np.array_equal(got_sched, sched)
        assert np.array_equal(D, rD) and np.array_equal(X, rX)
        w = wl.config(2, n_nodes=16, n_apps=48)
        apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
        oapps = ob.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
        gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, 4, apps)
        ref = ob.fit_fifo_chain(4, avail, oapps, rD, rX, sched=sched, zone=c["zone"])
        assert gpu.failed_at == ref.failed_at and np.array_equal(gpu.results, ref.results)
    with gangfit.Context(0) as fresh:
        with pytest.raises(gangfit.GangfitError) as e:  # no cluster yet
            fresh._cluster_n = 1
            fresh.build_snapshot_resident()
        assert e.value.code == gangfit._native.GF_ERR_STATE


@pytest.mark.gpu
def test_resident_usage_with_deltas(gf_ctx):
    """gf_usage_apply keeps the UsageForNodes sums on the device: after every batch of added / removed entries the snapshot
    built from the resident sums (n_res = GF_RESIDENT_USAGE, no entry travels) equals the replay of the live entry list —
    values, priority orders and a chain on top of it."""
    n = 2500
    c = _cluster(91, n, 300, 3, with_overhead=True, labels=False)
    gf_ctx.set_cluster(c["alloc"], c["node_flags"], c["name_rank"], overhead=c["overhead"], zone=c["zone"], n_zones=c["n_zones"])
    rng = np.random.default_rng(17)

    def entries(m):
        node = rng.integers(0, n + 3, size=m).astype(np.uint32)  # a few on nodes outside the cluster: ignored
        req = np.stack([rng.choice([500, 1000, 4000], size=m), rng.choice([1, 4, 16], size=m) * GIB,
                        (rng.random(m) < 0.05).astype(np.int64)], axis=1).astype(np.int64)
        return node, req

    live_node, live_req = np.zeros(0, dtype=np.uint32), np.zeros((0, 3), dtype=np.int64)
    for rep in range(5):
        if rep == 3:  # start over
            gf_ctx.usage_reset()
            live_node, live_req = live_node[:0], live_req[:0]
        an, ar = entries(int(rng.integers(1, 4000)))
        gf_ctx.usage_apply(an, ar, +1)
        live_node, live_req = np.concatenate([live_node, an]), np.concatenate([live_req, ar])
        if rep % 2 == 1:  # some reservations go away (in another order than they came)
            gone = rng.random(len(live_node)) < 0.4
            idx = rng.permutation(np.nonzero(gone)[0])
            gf_ctx.usage_apply(live_node[idx], res_cols=[np.ascontiguousarray(live_req[idx, j]) for j in range(3)], sign=-1)
            live_node, live_req = live_node[~gone], live_req[~gone]
        D, X = gf_ctx.build_snapshot_resident(resident_usage=True)
        avail, sched, rD, rX = ps.build(**dict(c, res_node=live_node, res_req=live_req))
        got_avail, got_sched = gf_ctx.snapshot()
        assert np.array_equal(got_avail, avail) and np.array_equal(got_sched, sched)
        assert np.array_equal(D, rD) and np.array_equal(X, rX)
        # the explicit entry list on the same context gives the same snapshot and leaves the resident sums alone
        D2, X2 = gf_ctx.build_snapshot_resident(res_node=live_node, res_req=live_req)
        assert np.array_equal(D2, rD) and np.array_equal(gf_ctx.snapshot()[0], avail)
        gf_ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
        assert np.array_equal(gf_ctx.snapshot()[0], avail)
        w = wl.config(2, n_nodes=16, n_apps=40)
        apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
        gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, 0, apps)
        ref = ob.fit_fifo_chain(0, avail, ob.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32)), rD, rX)
        assert gpu.failed_at == ref.failed_at and np.array_equal(gpu.results, ref.results)
    with pytest.raises(gangfit.GangfitError) as e:  # more removed than was ever added
        big = np.array([[1 << 40, 1 << 50, 9]], dtype=np.int64)
        gf_ctx.usage_apply(np.array([0], dtype=np.uint32), big, -1)
    assert e.value.code == gangfit._native.GF_ERR_INVALID
    with pytest.raises(gangfit.GangfitError):
        gf_ctx.usage_apply(np.array([0], dtype=np.uint32), np.array([[1, 1, 1]], dtype=np.int64), 2)


@pytest.mark.gpu
def test_null_flags_select_the_cluster_defaults_again(gf_ctx):
    """gf_snapshot_build_resident(node_flags = NULL) means the flags of gf_cluster_set — also right after a request that
    brought its own candidate flags (the request's flags must not stick)."""
    n = 1500
    c = _cluster(123, n, 200, 2, with_overhead=False, labels=False)
    gf_ctx.set_cluster(c["alloc"], c["node_flags"], c["name_rank"], zone=c["zone"], n_zones=c["n_zones"])
    rng = np.random.default_rng(9)
    other = (c["node_flags"] & ~np.uint32(ps.DRIVER_CANDIDATE)) | np.where(rng.random(n) < 0.3, ps.DRIVER_CANDIDATE, 0).astype(np.uint32)
    D0, X0 = gf_ctx.build_snapshot_resident(res_node=c["res_node"], res_req=c["res_req"])
    D1, X1 = gf_ctx.build_snapshot_resident(res_node=c["res_node"], res_req=c["res_req"], node_flags=other)
    D2, X2 = gf_ctx.build_snapshot_resident(res_node=c["res_node"], res_req=c["res_req"])
    _, _, rD, rX = ps.build(**c)
    _, _, oD, oX = ps.build(**dict(c, node_flags=other))
    assert np.array_equal(D0, rD) and np.array_equal(D1, oD) and not np.array_equal(D1, D0)
    assert np.array_equal(D2, rD) and np.array_equal(X2, rX)


@pytest.mark.gpu
def test_removing_what_was_never_added_is_refused(gf_ctx):
    """gf_usage_apply(sign = -1) of an entry its node does not carry would drive that node's sum negative (available above
    allocatable) although the global total stays positive: GF_ERR_INVALID, and the sums stay as they were."""
    n = 800
    c = _cluster(321, n, 100, 1, with_overhead=False, labels=False)
    gf_ctx.set_cluster(c["alloc"], c["node_flags"], c["name_rank"], zone=c["zone"], n_zones=c["n_zones"])
    g0 = gf_ctx.generation()
    node = np.array([1, 1, 2, 5], dtype=np.uint32)
    req = np.array([[1000, GIB, 0]] * 4, dtype=np.int64)
    gf_ctx.usage_apply(node, req, +1)
    assert gf_ctx.generation()[2] > g0[2] and gf_ctx.generation()[1] == g0[1]
    gf_ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
    before = gf_ctx.snapshot()[0].copy()
    with pytest.raises(gangfit.GangfitError) as e:
        gf_ctx.usage_apply(np.array([7], dtype=np.uint32), req[:1], -1)  # node 7 never got anything
    assert e.value.code == gangfit._native.GF_ERR_INVALID
    gf_ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
    assert np.array_equal(gf_ctx.snapshot()[0], before)
    gf_ctx.usage_apply(np.array([1], dtype=np.uint32), req[:1], -1)  # a real removal still works
    gf_ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
    after = gf_ctx.snapshot()[0]
    assert after[1, 0] == before[1, 0] + 1000 and after[1, 1] == before[1, 1] + GIB


@pytest.mark.gpu
def test_recorded_sequence_is_refused_after_an_install():
    """gf_graph_launch replays device addresses: after gf_snapshot_set / gf_orders_set (which may reallocate them) the
    recording is stale and must be refused (GF_ERR_STATE), not replayed."""
    import torch

    w = wl.config(2, n_nodes=500, n_apps=64)
    s = w.snapshot
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k))
        dev = torch.device("cuda", 0)
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        d_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()  # torch fills on ITS stream; the context's stream does not wait for it

        def step():
            ctx.fit_batch_dev(0, 0, len(apps), d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k, stream=0)

        step()
        torch.cuda.synchronize()
        ctx.graph_begin(0)
        step()
        g = ctx.graph_end(0)
        ctx.graph_launch(g, 0)
        torch.cuda.synchronize()
        first = d_res.cpu().numpy().copy()
        ctx.set_snapshot(s.avail // 2, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        with pytest.raises(gangfit.GangfitError) as e:
            ctx.graph_launch(g, 0)
        assert e.value.code == gangfit._native.GF_ERR_STATE
        ctx.graph_destroy(g)
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        step()
        torch.cuda.synchronize()
        assert np.array_equal(d_res.cpu().numpy(), first)


@pytest.mark.gpu
@pytest.mark.parametrize("spread", ["two-keys", "three-keys", "all-equal", "ties-by-name"])
def test_priority_sort_key_groups(gf_ctx, spread):
    """The hand-written priority sort (gangfit_snapshot.hip, priority_sort_kernel) packs (zone rank, free memory, free cpu)
    into one 64-bit key when the ranges allow; quantities spread over the whole representable range need two or three keys
    sorted one after the other.  Same order as the restatement in every case (nodesorting.go:95-122: zone rank, memory, cpu,
    name — all ascending, negatives included)."""
    n = 5000
    rng = np.random.default_rng(len(spread))
    c = _cluster(900 + len(spread), n, 0, 3, with_overhead=False, labels=False)
    alloc = c["alloc"].copy()
    if spread == "two-keys":      # memory needs ~61 bits, cpu ~20: cpu alone, then (zone, memory)
        alloc[:, 1] = rng.integers(0, 1 << 61, size=n) | 1
        alloc[:, 0] = rng.integers(0, 1 << 20, size=n)
    elif spread == "three-keys":  # memory and cpu need ~61 bits each, and 4 000 zones 12 more
        alloc[:, 1] = rng.integers(0, 1 << 61, size=n) | 1
        alloc[:, 0] = rng.integers(0, 1 << 61, size=n) | 1
        c["zone"] = rng.integers(0, 4000, size=n).astype(np.uint32)
        c["n_zones"] = 4000
    elif spread == "all-equal":   # no varying bit at all: the order is the name order
        alloc[:] = alloc[0]
        c["zone"] = np.zeros(n, dtype=np.uint32)
        c["n_zones"] = 1
    else:                         # a handful of distinct values: long runs decided by the name rank
        alloc[:, 1] = rng.integers(0, 3, size=n) * GIB
        alloc[:, 0] = rng.integers(0, 2, size=n) * 1000
    c["alloc"] = alloc
    c["res_node"], c["res_req"] = np.zeros(0, dtype=np.uint32), np.zeros((0, 3), dtype=np.int64)
    D, X = gf_ctx.build_snapshot(**c)
    avail, sched, rD, rX = ps.build(**c)
    assert np.array_equal(gf_ctx.snapshot()[0], avail)
    assert np.array_equal(D, rD) and np.array_equal(X, rX)


@pytest.mark.gpu
@pytest.mark.parametrize("want_orders", [True, False])
def test_sort_barrier_timeout_is_reported_not_installed(gf_ctx, want_orders):
    """The priority sort's grid barrier is an ordinary launch whose sixty-four workgroups are ASSUMED resident together; when they
    are not (a device saturated by somebody's endless kernel) the barrier gives up and flags an error word.  `sort_fault`
    provokes exactly that (one workgroup never arrives): the build must fail loudly on both paths that read the word — the
    host-finalized one and the device-finalized one —, install nothing, and leave the context usable."""
    c = _cluster(4242, 3000, 500, 3)
    D0, X0 = gf_ctx.build_snapshot(**c)                    # a good snapshot first: it must survive the failed build
    before = gf_ctx.snapshot()[0].copy()
    gf_ctx.set_option("sort_fault", 1)
    c2 = _cluster(4243, 3000, 500, 3)
    with pytest.raises(gangfit.GangfitError) as e:
        gf_ctx.build_snapshot(want_orders=want_orders, **c2)
    assert "grid barrier gave up" in str(e.value)
    gf_ctx.set_option("sort_fault", 0)
    D, X = gf_ctx.build_snapshot(**c)                      # ... and the very next build on the same context is fine
    avail, sched, rD, rX = ps.build(**c)
    assert np.array_equal(D, rD) and np.array_equal(X, rX) and np.array_equal(D, D0) and np.array_equal(X, X0)
    assert np.array_equal(gf_ctx.snapshot()[0], avail) and np.array_equal(before, avail)
