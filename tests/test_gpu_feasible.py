"""gf_fit_feasible: the feasibility-only independent batch (what UnschedulablePodMarker reads: `!packingResult.HasCapacity`,
internal/extender/unschedulablepods.go:132-166).  Same decision code as gf_fit_batch(GF_MODE_INDEPENDENT); only one byte per
application crosses the host link."""
import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob

pytestmark = pytest.mark.gpu
IND = gangfit.GF_MODE_INDEPENDENT


def _congested(n_nodes, n_apps, seed):
    w = wl.headline(n_nodes, n_apps, seed=seed, congested=True)  # about half of the gangs do not fit
    return w, w.snapshot


@pytest.mark.parametrize("algo", [0, 1, 2])
@pytest.mark.parametrize("n_nodes,n_apps", [(1, 1), (64, 5), (300, 96), (3000, 700)])
def test_feasible_equals_the_oracle_and_the_full_batch(gf_ctx, algo, n_nodes, n_apps):
    w, s = _congested(n_nodes, n_apps, 0xFEA5 + algo + n_nodes)
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    fits = gf_ctx.fit_feasible(algo, apps)
    full = gf_ctx.fit_batch(IND, algo, apps)
    ref = ob.fit_independent(algo, s.avail, ob.make_apps(w.drv, w.exe, w.k, w.flags), s.driver_order, s.exec_order)
    assert np.array_equal(fits, full.results["has_capacity"].astype(bool))
    assert np.array_equal(fits, ref.results["has_capacity"].astype(bool))
    if n_apps >= 96:
        assert fits.any() and not fits.all()


@pytest.mark.parametrize("algo", [3, 4, 5])
def test_feasible_zone_aware_packers(gf_ctx, algo):
    w, s = _congested(900, 150, 0xFEA6 + algo)
    zone = (wl.splitmix64(0xA7, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    order = wl.reference_node_order(s.avail, zone)
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_zones(zone)
    gf_ctx.set_orders(order, order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    fits = gf_ctx.fit_feasible(algo, apps)
    full = gf_ctx.fit_batch(IND, algo, apps)
    assert np.array_equal(fits, full.results["has_capacity"].astype(bool))
    assert fits.any()


@pytest.mark.parametrize("algo", [3, 4, 5])
def test_feasible_zone_aware_without_the_averages(gf_ctx, algo):
    """fit_zoned_fused_kernel's feasibility instantiation skips a zone's average efficiency when it is certainly above 0 (no node's
    available quantity above its schedulable one — checked by gf_snapshot_set — and a driver that asks for cpu or memory).  The
    cases around that short cut, each against the full batch (which always computes the averages, single_az.go:75-97):
    drivers with and without cpu / memory requests, gangs of nothing at all on nodes nobody uses (every efficiency is exactly 0:
    chooseBestResult's strict '<' turns the application down although it fits), and a snapshot with one node whose available
    memory exceeds its schedulable memory (the short cut must switch itself off)."""
    w, s = _congested(1200, 200, 0xFEA9 + algo)
    zone = (wl.splitmix64(0xA7, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    order = wl.reference_node_order(s.avail, zone)
    drv, exe, k = w.drv.copy(), w.exe.copy(), w.k.copy()
    drv[0::5] = 0                  # drivers that ask for nothing
    drv[1::5, 0] = 0               # ... for memory only
    drv[2::5, 1] = 0               # ... for cpu only
    exe[0::10] = 0
    k[0::10] = 3                   # gangs of nothing at all
    apps = gangfit.make_apps(drv, exe, k, w.flags)
    verdicts = []
    for variant in ("as generated", "empty nodes", "one node above its schedulable memory"):
        avail, sched = s.avail.copy(), s.sched.copy()
        if variant == "empty nodes":
            avail = sched.copy()   # nothing is used anywhere: a gang of nothing has efficiency 0 everywhere
        elif variant == "one node above its schedulable memory":
            avail[order[0], 1] = sched[order[0], 1] + 1
        gf_ctx.set_snapshot(avail, sched)
        gf_ctx.set_zones(zone)
        gf_ctx.set_orders(order, order)
        fits = gf_ctx.fit_feasible(algo, apps)
        full = gf_ctx.fit_batch(IND, algo, apps)
        assert np.array_equal(fits, full.results["has_capacity"].astype(bool)), variant
        verdicts.append(fits)
    # the quirk is there: on empty nodes the gangs of nothing with a driver of nothing are turned down
    nothing = (~drv.any(axis=1)) & (~exe.any(axis=1))
    assert nothing.any() and verdicts[1][~nothing].any()
    if algo != 3:  # (az-aware-tightly-pack falls back to the plain placement when no zone is chosen: az_aware_pack_tightly.go:33-37)
        assert not verdicts[1][nothing].any()


def test_feasible_rejects_what_the_batch_rejects(gf_ctx):
    w, s = _congested(64, 4, 3)
    gf_ctx.set_snapshot(s.avail, s.sched)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    apps["k"][2] = -1
    with pytest.raises(gangfit.GangfitError) as e:
        gf_ctx.fit_feasible(0, apps)
    assert e.value.code == gangfit._native.GF_ERR_INVALID
    assert len(gf_ctx.fit_feasible(0, apps[:0])) == 0


def test_feasible_waiting_for_the_stream(gf_ctx):
    """feasible_announce = 0: the call waits for the stream instead of watching the bytes arrive; same answers, also when the
    two kinds of call alternate on one context (the announcing call returns before its kernel has ended)."""
    w, s = _congested(2000, 500, 17)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    with gangfit.Context(0) as c:
        c.set_snapshot(s.avail, s.sched)
        c.set_orders(s.driver_order, s.exec_order)
        full = c.fit_batch(IND, 0, apps)
        want = full.results["has_capacity"].astype(bool)
        for rnd in range(6):
            c.set_option("feasible_announce", rnd & 1)
            assert np.array_equal(c.fit_feasible(0, apps), want)
            assert np.array_equal(c.fit_feasible(1, apps[: 100 + rnd]), c.fit_batch(IND, 1, apps[: 100 + rnd]).results["has_capacity"].astype(bool))
            again = c.fit_batch(IND, 0, apps)  # right behind an announcing call: placements of the full batch unchanged
            assert np.array_equal(again.results, full.results) and np.array_equal(again.exec_nodes, full.exec_nodes)


def test_feasible_without_mapped_staging(gf_ctx):
    """zero_copy = 0: records and answers travel by copies around the kernel; same answers."""
    w, s = _congested(500, 120, 11)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, w.flags)
    with gangfit.Context(0, options={"zero_copy": 0}) as c:
        c.set_snapshot(s.avail, s.sched)
        c.set_orders(s.driver_order, s.exec_order)
        a = c.fit_feasible(0, apps)
        assert np.array_equal(a, c.fit_batch(IND, 0, apps).results["has_capacity"].astype(bool))
