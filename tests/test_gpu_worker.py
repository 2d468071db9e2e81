"""The resident worker of the independent batch (gf_worker_*): same answers as the launch path, tickets in flight, leaving and
coming back, installs under a resident worker."""
import time

import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl

pytestmark = pytest.mark.gpu

TIGHT, EVEN, MINFRAG = gangfit.GF_ALGO_TIGHTLY_PACK, gangfit.GF_ALGO_DISTRIBUTE_EVENLY, gangfit.GF_ALGO_MINIMAL_FRAGMENTATION
IND = gangfit.GF_MODE_INDEPENDENT


def _install(ctx, w):
    s = w.snapshot
    ctx.set_snapshot(s.avail, s.sched)
    ctx.set_orders(s.driver_order, s.exec_order)


def _same(a, b):
    return np.array_equal(a.results, b.results) and np.array_equal(a.exec_nodes, b.exec_nodes)


@pytest.mark.parametrize("algo", [TIGHT, EVEN, MINFRAG])
def test_worker_fit_answers_what_a_launch_answers(algo):
    ctx = gangfit.Context(0)
    try:
        w = wl.headline(3000, 700, seed=0xB0B + algo)
        _install(ctx, w)
        apps = gangfit.make_apps(w.drv, w.exe, w.k)
        want = ctx.fit_batch(IND, algo, apps)
        for _ in range(3):
            assert _same(ctx.worker_fit(algo, apps), want)
        # other batches on the same resident worker, small and ragged ones included
        rng = np.random.default_rng(algo)
        for n in (1, 3, 64, 257, 700):
            pick = rng.permutation(len(apps))[:n]
            sub = apps[pick]
            assert _same(ctx.worker_fit(algo, sub), ctx.fit_batch(IND, algo, sub))
        st = ctx.worker_stats()
        assert st["posted"] == st["complete"] == 8
        # more applications than a set has wavefronts (3 x 1024 + a ragged rest): every wavefront takes several per ticket
        big = np.concatenate([apps] * 5)[:3333]
        assert _same(ctx.worker_fit(algo, big), ctx.fit_batch(IND, algo, big))
    finally:
        ctx.close()


def test_worker_leaves_when_idle_and_comes_back():
    ctx = gangfit.Context(0, options={"worker_idle_us": 50})
    try:
        w = wl.headline(2000, 300, seed=77)
        _install(ctx, w)
        apps = gangfit.make_apps(w.drv, w.exe, w.k)
        want = ctx.fit_batch(IND, TIGHT, apps)
        assert _same(ctx.worker_fit(TIGHT, apps), want)
        time.sleep(0.05)  # a thousand idle periods
        assert not ctx.worker_stats()["resident"]
        assert _same(ctx.worker_fit(TIGHT, apps), want)
        assert ctx.worker_stats()["launches"] >= 2
        ctx.worker_stop()
        assert not ctx.worker_stats()["resident"]
        assert _same(ctx.worker_fit(TIGHT, apps), want)
    finally:
        ctx.close()


def test_an_install_makes_the_worker_leave_and_the_next_batch_sees_the_new_snapshot():
    ctx = gangfit.Context(0, options={"worker_idle_us": 100000})
    try:
        w1, w2 = wl.headline(2000, 300, seed=5), wl.headline(2500, 300, seed=6)
        apps = gangfit.make_apps(w1.drv, w1.exe, w1.k)
        _install(ctx, w1)
        a1 = ctx.worker_fit(TIGHT, apps)
        assert ctx.worker_stats()["resident"]
        assert _same(a1, ctx.fit_batch(IND, TIGHT, apps))
        _install(ctx, w2)  # serves what was posted, then the worker leaves
        assert not ctx.worker_stats()["resident"]
        a2 = ctx.worker_fit(TIGHT, apps)
        assert _same(a2, ctx.fit_batch(IND, TIGHT, apps))
        assert not _same(a1, a2)
        # another packer on the same snapshot: the worker is relaunched for it
        assert _same(ctx.worker_fit(EVEN, apps), ctx.fit_batch(IND, EVEN, apps))
    finally:
        ctx.close()


@pytest.mark.parametrize("algo,sets", [(TIGHT, 4), (EVEN, 3), (TIGHT, 1)])
def test_tickets_in_flight_device_resident(algo, sets):
    """More tickets than the ring holds, different queues, own output arrays: every ticket answers what a launch answers."""
    import torch

    ctx = gangfit.Context(0, options={"worker_sets": sets})
    try:
        w = wl.headline(3000, 400, seed=0x71C + sets)
        _install(ctx, w)
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(sets)
        n_batches = 150
        batches, outs, want = [], [], []
        for b in range(n_batches):
            n = int(rng.integers(1, 401))
            pick = rng.permutation(400)[:n]
            apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv[pick], w.exe[pick], w.k[pick]))
            d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
            d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
            d_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()  # torch fills on ITS stream; the context's stream does not wait for it
            ctx.fit_batch_dev(IND, algo, n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k)
            torch.cuda.synchronize()
            want.append((d_res.clone(), d_exec.clone()))
            d_res.zero_()
            d_exec.zero_()
            batches.append((n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k))
            outs.append((d_apps, d_res, d_exec))
        torch.cuda.synchronize()
        first = ctx.worker_submit_dev(algo, batches)
        ctx.worker_wait(first, n_batches)
        ctx.worker_stop()
        torch.cuda.synchronize()
        bad = []
        for b, ((_, d_res, d_exec), (w_res, w_exec)) in enumerate(zip(outs, want)):
            if not (torch.equal(d_res, w_res) and torch.equal(d_exec, w_exec)):
                r, wr = d_res.cpu().numpy().view(np.uint32).reshape(-1, 4), w_res.cpu().numpy().view(np.uint32).reshape(-1, 4)
                rows = np.nonzero((r != wr).any(axis=1))[0]
                ex = np.nonzero(d_exec.cpu().numpy() != w_exec.cpu().numpy())[0]
                bad.append((b, len(r), rows[:6].tolist(), r[rows[:3]].tolist(), wr[rows[:3]].tolist(), len(ex), ex[:8].tolist()))
        assert not bad, bad[:6]
        st = ctx.worker_stats()
        assert st["posted"] == st["complete"] == n_batches
    finally:
        ctx.close()


@pytest.mark.parametrize("n_batches", [1, 20, 150])
def test_a_bounded_stream_tells_the_worker_where_it_ends(n_batches):
    """GF_WORKER_LEAVE_AFTER on the last batch of a submit that launches the worker: it serves what was posted and leaves the device
    by itself (no stop word, no idle period); the same flag on a submit that finds the worker resident is ignored.  Every ticket
    answers what a launch answers — more tickets than the ring holds included (the host then posts the rest while the worker runs:
    the launch was told the end of what was posted when it started, and the rest reaches a second launch)."""
    import torch

    ctx = gangfit.Context(0, options={"worker_idle_us": 500000})  # an idle period this long would fail the timing below
    try:
        w = wl.headline(3000, 400, seed=0x1EA7)
        _install(ctx, w)
        dev = torch.device("cuda:0")
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k))
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        want_res = torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev)
        want_exec = torch.zeros(total_k + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.fit_batch_dev(IND, TIGHT, len(apps), d_apps.data_ptr(), want_res.data_ptr(), want_exec.data_ptr(), total_k)
        torch.cuda.synchronize()
        outs = [(torch.zeros_like(want_res), torch.zeros_like(want_exec)) for _ in range(n_batches)]
        torch.cuda.synchronize()
        arr = ctx.worker_batches([(len(apps), d_apps.data_ptr(), r.data_ptr(), e.data_ptr(), total_k) for r, e in outs], leave_after=True)
        for rep in range(3):
            for r, e in outs:
                r.zero_()
                e.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            first = ctx.worker_submit_prepared(TIGHT, arr)
            ctx.worker_wait(first, n_batches)
            ctx.worker_stop()
            torch.cuda.synchronize()
            assert time.perf_counter() - t0 < 0.25, "the worker waited to be told (or for its idle period) instead of leaving"
            assert all(torch.equal(r, want_res) and torch.equal(e, want_exec) for r, e in outs)
            st = ctx.worker_stats()
            assert st["posted"] == st["complete"] == (rep + 1) * n_batches and not st["resident"]
    finally:
        ctx.close()


def test_worker_refusals():
    ctx = gangfit.Context(0)
    try:
        with pytest.raises(gangfit.GangfitError):  # no snapshot yet
            ctx.worker_fit(TIGHT, gangfit.make_apps([[1, 1, 0]], [[1, 1, 0]], [1]))
        w = wl.headline(500, 10, seed=1)
        _install(ctx, w)
        with pytest.raises(gangfit.GangfitError):  # zone packers are not served by the worker
            ctx.worker_fit(gangfit.GF_ALGO_SINGLE_AZ_TIGHTLY_PACK, gangfit.make_apps(w.drv, w.exe, w.k))
        v = ctx.view()
        with pytest.raises(gangfit.GangfitError):
            v.worker_fit(TIGHT, gangfit.make_apps(w.drv, w.exe, w.k))
        v.close()
    finally:
        ctx.close()


def test_record_array_rewritten_while_the_worker_is_resident():
    """The contract of gf_worker_submit_dev: a device array whose content is replaced by a copy that has COMPLETED before the
    submit is read with its new content — same address, worker resident all along (its L2s were never invalidated by a
    kernel boundary)."""
    import torch

    ctx = gangfit.Context(0, options={"worker_idle_us": 200000})
    try:
        w = wl.headline(3000, 400, seed=0xD0D0)
        _install(ctx, w)
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(9)
        n = 400
        queues = []
        for _ in range(6):
            pick = rng.permutation(400)
            apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv[pick], w.exe[pick], w.k[pick]))
            queues.append((apps, total_k, ctx.fit_batch(IND, TIGHT, apps)))
        max_k = max(q[1] for q in queues)
        d_apps = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        d_exec = torch.zeros(max_k + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()  # torch's default stream would wait for the resident worker (its stream is a blocking one)
        for apps, total_k, want in queues:
            host = torch.from_numpy(apps.view(np.uint8).copy()).pin_memory()
            with torch.cuda.stream(side):
                d_apps.copy_(host, non_blocking=True)
            side.synchronize()
            first = ctx.worker_submit_dev(TIGHT, [(n, d_apps.data_ptr(), d_res.data_ptr(), d_exec.data_ptr(), total_k)])
            ctx.worker_wait(first, 1)
            with torch.cuda.stream(side):
                got_res = d_res.cpu().numpy().view(gangfit._native.RESULT_DTYPE)
                got_exec = d_exec.cpu().numpy().view(np.uint32)[:total_k]
            assert np.array_equal(got_res, want.results) and np.array_equal(got_exec, want.exec_nodes)
        st = ctx.worker_stats()
        assert st["launches"] == 1 and st["resident"]
        ctx.worker_stop()
    finally:
        ctx.close()


def test_a_fifo_chain_starts_next_to_a_resident_worker():
    """The worker's workgroups each fill a CU and there are fewer of them than CUs: a FIFO chain (which needs a whole CU)
    does not have to wait for the worker to leave."""
    ctx = gangfit.Context(0, options={"worker_idle_us": 500000})
    try:
        w = wl.headline(5000, 300, seed=0xFEED)
        _install(ctx, w)
        apps = gangfit.make_apps(w.drv, w.exe, w.k, np.ones(len(w.k), dtype=np.uint32))
        want = ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, TIGHT, apps)
        ctx.set_option("chain_cache", 0)  # the chain below replays
        assert _same(ctx.worker_fit(TIGHT, apps), ctx.fit_batch(IND, TIGHT, apps))
        assert ctx.worker_stats()["resident"]
        t0 = time.perf_counter()
        got = ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, TIGHT, apps)
        dt = time.perf_counter() - t0
        assert ctx.worker_stats()["resident"]  # ... and it is still there
        assert _same(got, want) and got.failed_at == want.failed_at
        assert dt < 0.1, dt  # (half a second would be the worker's idle period)
        ctx.worker_stop()
    finally:
        ctx.close()


def test_geometry_follows_the_first_ticket_of_a_launch():
    """The worker's geometry is chosen per launch (gf_worker_geometry): a stream of tickets gets three applications per
    wavefront and as many sets as fit next to sixteen free CUs; one blocking ticket one application per wavefront; the options
    override; the answers are the launch path's whatever the geometry."""
    import torch

    ctx = gangfit.Context(0)
    try:
        w = wl.headline(4000, 1000, seed=0x6E0)
        _install(ctx, w)
        assert ctx.worker_geometry() == (0, 0)  # no launch yet
        cus = ctx.device_info()["compute_units"]
        room = cus - 16 if cus > 32 else cus
        apps, total_k = gangfit.with_offsets(gangfit.make_apps(w.drv, w.exe, w.k))
        want = ctx.fit_batch(IND, TIGHT, apps)
        dev = torch.device("cuda:0")
        d_apps = torch.from_numpy(apps.view(np.uint8).copy()).to(dev)
        outs = [(torch.zeros(len(apps) * 16, dtype=torch.uint8, device=dev), torch.zeros(total_k + 1, dtype=torch.int32, device=dev))
                for _ in range(4)]
        torch.cuda.synchronize()

        def stream(n_apps, n_tickets=12):
            first = ctx.worker_submit_dev(TIGHT, [(n_apps, d_apps.data_ptr(), outs[i % 4][0].data_ptr(), outs[i % 4][1].data_ptr(), total_k)
                                                  for i in range(n_tickets)])
            ctx.worker_wait(first, n_tickets)
            g = ctx.worker_geometry()
            ctx.worker_stop()
            torch.cuda.synchronize()
            return g

        sets, bps = stream(1000)
        assert bps == (1000 + 47) // 48 == 21 and sets == min(16, (room - 1) // 21)  # 11 x 21 on 256 CUs
        res = outs[0][0].cpu().numpy().view(gangfit._native.RESULT_DTYPE)
        assert np.array_equal(res, want.results) and np.array_equal(outs[0][1].cpu().numpy()[:total_k].view(np.uint32), want.exec_nodes)
        sets, bps = stream(100)
        assert bps == 3 and sets == 16
        assert _same(ctx.worker_fit(TIGHT, apps), want)  # one blocking ticket: an application per wavefront
        sets, bps = ctx.worker_geometry()
        assert bps == (1000 + 15) // 16 == 63 and sets == (room - 1) // 63
        ctx.set_option("worker_sets", 2)
        ctx.set_option("worker_blocks_per_set", 40)
        assert stream(1000) == (2, 40)
        assert np.array_equal(outs[0][0].cpu().numpy().view(gangfit._native.RESULT_DTYPE), want.results)
    finally:
        ctx.close()
