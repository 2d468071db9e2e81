"""gf_ctx_view (include/gangfit.h): N contexts with their own stream and working tables fit on ONE installed snapshot —
concurrent FIFO chains of different queues on the compute units of one GPU (Predicate and the UnschedulablePodMarker run
concurrently in the reference, cmd/server.go:230).  Every view's answers and residuals must equal the oracle's for ITS queue
(no cross-talk), an install on the parent must be seen by the views, and the installing entry points are refused on a view."""
import threading
import time

import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob
from test_gpu_parity import _assert_same, _random_problem

pytestmark = pytest.mark.gpu
IND, FIFO = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_MODE_FIFO_CHAIN


def test_views_share_the_snapshot_and_nothing_else():
    rng = np.random.default_rng(2026)
    avail, D, X, drv, exe, k = _random_problem(rng, 1500, 160, False, "merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 30).astype(np.int32)
    flags = np.ones(len(k), dtype=np.uint32)
    with gangfit.Context(0) as parent:
        parent.set_snapshot(avail)
        parent.set_orders(D, X)
        views = [parent.view() for _ in range(6)]
        try:
            queues = [np.roll(np.arange(len(k)), -17 * i) for i in range(len(views))]
            refs = [ob.fit_fifo_chain(i % 2, avail, ob.make_apps(drv[q], exe[q], k[q], flags), D, X) for i, q in enumerate(queues)]
            outs, resid, errs = [None] * len(views), [None] * len(views), []

            def run(i):
                try:
                    q = queues[i]
                    for _ in range(3):  # repeated: the second and third chain resume from the view's own checkpoints
                        outs[i] = views[i].fit_batch(FIFO, i % 2, gangfit.make_apps(drv[q], exe[q], k[q], flags))
                    resid[i] = views[i].residual()
                except Exception as e:  # pragma: no cover
                    errs.append(e)

            th = [threading.Thread(target=run, args=(i,)) for i in range(len(views))]
            for t in th:
                t.start()
            ind = parent.fit_batch(IND, 0, gangfit.make_apps(drv, exe, k))  # the parent keeps serving meanwhile
            for t in th:
                t.join()
            assert not errs, errs
            _assert_same(ind, ob.fit_independent(0, avail, ob.make_apps(drv, exe, k), D, X), gangfit.make_apps(drv, exe, k))
            for i, q in enumerate(queues):
                assert outs[i].failed_at == refs[i].failed_at
                _assert_same(outs[i], refs[i], gangfit.make_apps(drv[q], exe[q], k[q], flags))
                assert np.array_equal(resid[i], refs[i].avail_after), i
                assert views[i].chain_cache_stats()[1] == 2
            # an install on the parent is what every view fits on next (and their chain caches start over)
            avail2 = avail.copy()
            avail2[D[D < len(avail)][:60]] //= 3
            parent.set_snapshot(avail2)
            parent.set_orders(D, X)
            q = queues[1]
            out = views[1].fit_batch(FIFO, 0, gangfit.make_apps(drv[q], exe[q], k[q], flags))
            ref = ob.fit_fifo_chain(0, avail2, ob.make_apps(drv[q], exe[q], k[q], flags), D, X)
            _assert_same(out, ref, gangfit.make_apps(drv[q], exe[q], k[q], flags))
            assert np.array_equal(views[1].residual(), ref.avail_after)
            assert np.array_equal(views[1].snapshot()[0], avail2)
            with pytest.raises(gangfit.GangfitError) as e:
                views[0]._check(views[0]._lib.gf_orders_set(views[0]._h, None, 0, None, 0))
            assert e.value.code == gangfit._native.GF_ERR_STATE
        finally:
            for v in views:
                v.close()


def test_eight_concurrent_headline_chains():
    """Eight 999 + 1 chains of the headline queue (different heads) on eight views at once: each is ONE workgroup, so they
    occupy eight compute units and finish in about the time of one — and every one equals the oracle's replay."""
    w = wl.headline(10000, 1000)
    s = w.snapshot
    flags = np.ones(len(w.k), dtype=np.uint32)
    with gangfit.Context(0, options={"chain_cache": 0}) as parent:
        parent.set_snapshot(s.avail, s.sched)
        parent.set_orders(s.driver_order, s.exec_order)
        views = [parent.view() for _ in range(8)]
        try:
            qs = [gangfit.make_apps(np.roll(w.drv, -i, axis=0), np.roll(w.exe, -i, axis=0), np.roll(w.k, -i), flags) for i in range(8)]
            for v, q in zip(views, qs):
                v.fit_batch(FIFO, 0, q)  # buffers grown, kernels loaded
            t0 = time.perf_counter()
            for _ in range(5):
                views[0].fit_batch(FIFO, 0, qs[0])
            one = (time.perf_counter() - t0) / 5
            outs = [None] * 8

            def run(i):
                for _ in range(5):
                    outs[i] = views[i].fit_batch(FIFO, 0, qs[i])

            th = [threading.Thread(target=run, args=(i,)) for i in range(8)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            eight = (time.perf_counter() - t0) / 5
            for i in range(8):
                oq = ob.make_apps(np.roll(w.drv, -i, axis=0), np.roll(w.exe, -i, axis=0), np.roll(w.k, -i), flags)
                ref = ob.fit_fifo_chain(0, s.avail, oq, s.driver_order, s.exec_order, closed_form=True)
                _assert_same(outs[i], ref, qs[i])
            print(f"one chain {one * 1e3:.2f} ms, eight concurrent chains {eight * 1e3:.2f} ms")
            assert eight < 5.0 * one  # concurrent, not one after the other (8x); host_bench records the ratio from C++ threads (1.1x)
        finally:
            for v in views:
                v.close()
