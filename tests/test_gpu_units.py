"""The scaled-int32 ("narrow") FIFO chain when the batch's requests are FINER than the table's own units: free memory that
happens to be a multiple of 4 GiB everywhere, drivers that ask for 2 GiB (or 1 GiB + 512 MiB).  gf_fit_batch refines the units
to gcd(table, requests) per batch instead of falling to the wide kernels; results and residuals must not change.  Also the
case where refinement is impossible (scaled magnitudes would pass 2^30): the wide kernels must answer, identically."""
import numpy as np
import pytest

import gangfit
from oracle import binding as ob

pytestmark = pytest.mark.gpu
GIB = 1 << 30
FIFO, IND = gangfit.GF_MODE_FIFO_CHAIN, gangfit.GF_MODE_INDEPENDENT


def _cluster(rng, n, mem_unit, cpu_unit=1000):
    avail = np.stack([rng.integers(-1, 64, size=n) * cpu_unit, rng.integers(-1, 96, size=n) * mem_unit,
                      np.where(rng.random(n) < 0.15, rng.integers(0, 8, size=n), 0)], axis=1).astype(np.int64)
    sched = np.maximum(avail, 0) + np.array([8 * cpu_unit, 16 * mem_unit, 1])
    order = np.lexsort((np.arange(n), avail[:, 0], avail[:, 1])).astype(np.uint32)
    zone = rng.integers(0, 3, size=n).astype(np.uint32)
    return avail, sched, order, zone


def _apps(rng, a, mem_quanta, cpu_quanta):
    drv = np.stack([rng.choice(cpu_quanta, size=a), rng.choice(mem_quanta, size=a), np.zeros(a, dtype=np.int64)], axis=1).astype(np.int64)
    exe = np.stack([rng.choice(cpu_quanta, size=a) * rng.integers(1, 4, size=a), rng.choice(mem_quanta, size=a) * rng.integers(1, 6, size=a),
                    (rng.random(a) < 0.1).astype(np.int64)], axis=1).astype(np.int64)
    k = rng.integers(0, 30, size=a).astype(np.int32)
    flags = (rng.random(a) < 0.9).astype(np.uint32)
    return drv, exe, k, flags


@pytest.mark.parametrize("algo", [0, 1, 2, 4, 5])
@pytest.mark.parametrize("n", [70, 1000, 12000])
def test_requests_finer_than_the_table_units(gf_ctx, algo, n):
    rng = np.random.default_rng(7 * n + algo)
    avail, sched, order, zone = _cluster(rng, n, 4 * GIB)
    gf_ctx.set_snapshot(avail, sched)
    gf_ctx.set_zones(zone)
    gf_ctx.set_orders(order, order)
    for mem_quanta, cpu_quanta in (([2 * GIB, 4 * GIB, 6 * GIB], [500, 1000, 2500]),   # half of the table's units
                                   ([GIB + GIB // 2, 3 * GIB], [250, 750]),             # a quarter / an eighth
                                   ([4 * GIB, 8 * GIB], [1000, 2000])):                 # the table's units themselves
        drv, exe, k, flags = _apps(rng, 160, mem_quanta, cpu_quanta)
        apps = gangfit.make_apps(drv, exe, k, flags)
        oapps = ob.make_apps(drv, exe, k, flags)
        ref = ob.fit_fifo_chain(algo, avail, oapps, order, order, sched=sched, zone=zone)
        out = gf_ctx.fit_batch(FIFO, algo, apps)
        assert out.failed_at == ref.failed_at and np.array_equal(out.results, ref.results)
        for a in np.nonzero(ref.results["has_capacity"])[0]:
            assert np.array_equal(out.placement(int(a))[2], ref.placement(int(a))[2])
        assert np.array_equal(gf_ctx.residual(), ref.avail_after)


def test_refinement_impossible_falls_back_to_the_wide_kernels(gf_ctx):
    """Free memory in multiples of 2^40 bytes up to 2^69... is out of range; here: multiples of 2^30 up to 2^59 (scaled 2^29),
    requests of one byte — refining the unit to 1 would need 2^59 per value."""
    rng = np.random.default_rng(3)
    n = 300
    avail = np.stack([rng.integers(0, 64, size=n) * 1000, rng.integers(1, 1 << 29, size=n).astype(np.int64) << 30,
                      np.zeros(n, dtype=np.int64)], axis=1).astype(np.int64)
    order = np.arange(n, dtype=np.uint32)
    gf_ctx.set_snapshot(avail)
    gf_ctx.set_orders(order, order)
    drv = np.tile(np.array([[1000, 1, 0]], dtype=np.int64), (40, 1))
    exe = np.tile(np.array([[500, 3, 0]], dtype=np.int64), (40, 1))
    k = rng.integers(1, 20, size=40).astype(np.int32)
    ref = ob.fit_fifo_chain(0, avail, ob.make_apps(drv, exe, k), order, order)
    out = gf_ctx.fit_batch(FIFO, 0, gangfit.make_apps(drv, exe, k))
    assert np.array_equal(out.results, ref.results) and np.array_equal(gf_ctx.residual(), ref.avail_after)


def test_device_built_snapshot_with_fine_requests(gf_ctx):
    """gf_snapshot_build computes units and the largest scaled magnitudes on the device; the refinement must work there too."""
    rng = np.random.default_rng(11)
    n = 5000
    alloc = np.stack([rng.choice([16, 32, 64], size=n) * 1000, rng.choice([64, 128, 256], size=n) * GIB,
                      np.zeros(n, dtype=np.int64)], axis=1).astype(np.int64)
    rnode = rng.integers(0, n, size=40000).astype(np.uint32)
    rreq = np.stack([rng.choice([1000, 2000, 4000], size=40000), rng.choice([4, 8, 16], size=40000) * GIB,
                     np.zeros(40000, dtype=np.int64)], axis=1).astype(np.int64)
    D, X = gf_ctx.build_snapshot(alloc, np.full(n, 6, dtype=np.uint32), rng.permutation(n).astype(np.uint32), res_node=rnode,
                                 res_req=rreq)
    avail, sched = gf_ctx.snapshot()
    drv, exe, k, flags = _apps(rng, 300, [2 * GIB, 4 * GIB, 8 * GIB], [1000, 2000, 4000])
    for algo in (0, 1, 4):
        ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe, k, flags), D, X, sched=sched)
        out = gf_ctx.fit_batch(FIFO, algo, gangfit.make_apps(drv, exe, k, flags))
        assert out.failed_at == ref.failed_at and np.array_equal(out.results, ref.results)
        assert np.array_equal(gf_ctx.residual(), ref.avail_after)
