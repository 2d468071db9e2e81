"""A lone blocking independent batch (gf_fit_batch / gf_spark_binpack on the zero-copy path) announces its own completion in
pinned host memory — results and placements leave the kernel as write-through stores, the last wavefront writes the call's
sequence number, the caller polls that word instead of waiting for the stream (gangfit::IndHostOut).  Same answers as the
stream-wait path and as the oracle, for every plain packer, for batches of one application, of a ragged last workgroup, of more
workgroups than arrival counters, many calls in a row (the counters reset themselves), and interleaved with device-resident
launches and FIFO chains on the same stream."""
import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob

pytestmark = pytest.mark.gpu
IND, FIFO = gangfit.GF_MODE_INDEPENDENT, gangfit.GF_MODE_FIFO_CHAIN


def _same(a, b):
    return np.array_equal(a.results, b.results) and np.array_equal(a.exec_nodes, b.exec_nodes)


@pytest.mark.parametrize("algo", [0, 1, 2])
@pytest.mark.parametrize("n_apps", [1, 2, 3, 5, 63, 64, 65, 257, 1000])
def test_flagged_completion_equals_stream_wait_and_oracle(algo, n_apps):
    w = wl.config(2, n_nodes=700, n_apps=n_apps)
    s = w.snapshot
    flags = np.ones(len(w.k), dtype=np.uint32)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, flags)
    ref = ob.fit_independent(algo, s.avail, ob.make_apps(w.drv, w.exe, w.k, flags), s.driver_order, s.exec_order)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        flagged = [ctx.fit_batch(IND, algo, apps) for _ in range(5)]   # the counters reset themselves between launches
        ph = ctx.call_phases()
        assert ph["total"] > 0 and abs(ph["stage"] + ph["launch"] + ph["wait"] + ph["copy_out"] - ph["total"]) < 1.0
        ctx.set_option("host_flag", 0)
        waited = ctx.fit_batch(IND, algo, apps)
        ctx.set_option("host_flag", 1)
        again = ctx.fit_batch(IND, algo, apps)
        assert np.array_equal(waited.results, ref.results)
        for a in np.nonzero(ref.results["has_capacity"])[0]:
            assert np.array_equal(waited.placement(int(a))[2], ref.placement(int(a))[2])
        for f in flagged + [again]:
            assert _same(f, waited)


def test_flagged_batches_between_chains_and_device_launches():
    import torch

    w = wl.headline(3000, 400)
    s = w.snapshot
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    one = apps[:1].copy()
    dev = torch.device("cuda:0")
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        ctx.set_option("host_flag", 0)
        want = ctx.fit_batch(IND, 0, apps)
        want1 = ctx.fit_batch(IND, 0, one)
        chain = ctx.fit_batch(FIFO, 0, apps)
        ctx.set_option("host_flag", 1)
        a2, tk = gangfit.with_offsets(apps)
        d_a = torch.from_numpy(a2.view(np.uint8).copy()).to(dev)
        d_r = torch.zeros(len(a2) * 16, dtype=torch.uint8, device=dev)
        d_e = torch.zeros(tk + 1, dtype=torch.int32, device=dev)
        for i in range(40):
            if i % 4 == 1:  # kernels queued on the same stream in front of the flagged launch
                for _ in range(3):
                    ctx.fit_batch_dev(IND, 0, len(a2), d_a.data_ptr(), d_r.data_ptr(), d_e.data_ptr(), tk)
            if i % 4 == 2:
                c = ctx.fit_batch(FIFO, 0, apps)
                assert _same(c, chain) and c.failed_at == chain.failed_at
            got = ctx.fit_batch(IND, 0, one if i % 3 == 0 else apps)
            assert _same(got, want1 if i % 3 == 0 else want)
        torch.cuda.synchronize()
        assert np.array_equal(d_r.cpu().numpy().view(gangfit._native.RESULT_DTYPE), want.results)
