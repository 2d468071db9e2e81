"""GPU parity of the zone-aware packers (single-az-tightly-pack, az-aware-tightly-pack) and of the packing efficiencies
(float64, BIT-identical: chooseBestResult compares them) against the CPU oracle.  `python -m pytest tests -m gpu`."""
import numpy as np
import pytest

import gangfit
import kats
from gangfit import workloads as wl
from oracle import binding as ob
from test_gpu_parity import _assert_same, _random_problem

pytestmark = pytest.mark.gpu

IND = gangfit.GF_MODE_INDEPENDENT
TIGHT, EVEN = gangfit.GF_ALGO_TIGHTLY_PACK, gangfit.GF_ALGO_DISTRIBUTE_EVENLY
SAZ, AZA = gangfit.GF_ALGO_SINGLE_AZ_TIGHTLY_PACK, gangfit.GF_ALGO_AZ_AWARE_TIGHTLY_PACK
O_ALGO = {SAZ: ob.ALGO_SINGLE_AZ_TIGHTLY_PACK, AZA: ob.ALGO_AZ_AWARE_TIGHTLY_PACK, TIGHT: 0, EVEN: 1}
GIB = kats.GIB


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _setup(ctx, avail, sched, zone, D, X):
    ctx.set_snapshot(avail, sched)
    if zone is not None:
        ctx.set_zones(zone)
    ctx.set_orders(D, X)


@pytest.mark.parametrize("case", kats.REFERENCE_PINNED, ids=[c["name"] for c in kats.REFERENCE_PINNED])
def test_reference_tests_through_single_az_tightly_pack(gf_ctx, case):
    """T1/T3/T4 of the reference select `single-az-tightly-pack` (one zone "default")."""
    avail = np.array(case["avail"], dtype=np.int64)
    _setup(gf_ctx, avail, avail, None, case["D"], case["X"])
    ok, d, ex = gf_ctx.spark_binpack(SAZ, case["drv"], case["exe"], case["k"])
    assert ok == case["feasible"]
    if ok:
        assert d == case["driver"] and ex.tolist() == case["execs"]
    else:
        assert d == gangfit.GF_NO_NODE and len(ex) == 0


def test_zone_choice_and_tie_break(gf_ctx):
    sched = [[16000, 64 * GIB, 0], [16000, 64 * GIB, 0], [8000, 16 * GIB, 0]]
    avail = [[16000, 64 * GIB, 0], [16000, 64 * GIB, 0], [4000, 8 * GIB, 0]]
    _setup(gf_ctx, avail, sched, [0, 0, 1], [0, 1, 2], [0, 1, 2])
    ok, d, ex = gf_ctx.spark_binpack(SAZ, [1000, GIB, 0], [1000, GIB, 0], 2)
    assert ok and d == 2 and ex.tolist() == [2, 2]  # zone 1: average Max 7/8 beats zone 0's 3/16
    sched2 = [[8000, 16 * GIB, 0], [8000, 16 * GIB, 0]]
    avail2 = [[4000, 8 * GIB, 0], [4000, 8 * GIB, 0]]
    _setup(gf_ctx, avail2, sched2, [5, 9], [1, 0], [0, 1])
    ok, d, ex = gf_ctx.spark_binpack(SAZ, [1000, GIB, 0], [1000, GIB, 0], 2)
    assert ok and d == 1 and ex.tolist() == [1, 1]  # tie: first zone of the DRIVER order
    # no single zone fits; az-aware falls back to plain tightly-pack
    sched3 = [[4000, 8 * GIB, 0], [4000, 8 * GIB, 0]]
    avail3 = [[2000, 8 * GIB, 0], [2000, 8 * GIB, 0]]
    _setup(gf_ctx, avail3, sched3, [0, 1], [0, 1], [0, 1])
    assert not gf_ctx.spark_binpack(SAZ, [1000, GIB, 0], [1000, GIB, 0], 2)[0]
    ok, d, ex = gf_ctx.spark_binpack(AZA, [1000, GIB, 0], [1000, GIB, 0], 2)
    assert ok and d == 0 and ex.tolist() == [0, 1]
    # a feasible zone result with average 0.0 is not better than WorstAvgPackingEfficiency -> empty result
    full = [[4000, 8 * GIB, 0], [4000, 8 * GIB, 0]]
    _setup(gf_ctx, full, full, [0, 0], [0, 1], [0, 1])
    assert not gf_ctx.spark_binpack(SAZ, [0, 0, 0], [0, 0, 0], 2)[0]
    # zones need the schedulable columns
    gf_ctx.set_snapshot(full)
    gf_ctx.set_orders([0, 1], [0, 1])
    with pytest.raises(gangfit.GangfitError):
        gf_ctx.spark_binpack(SAZ, [1, 1, 0], [1, 1, 0], 1)


def _zoned_problem(rng, n, a, tight_cluster, layout, n_zones):
    avail, D, X, drv, exe, k = _random_problem(rng, n, a, tight_cluster, layout)
    extra = rng.integers(0, 50 if tight_cluster else 5000, size=(n, 3)).astype(np.int64)
    sched = np.maximum(avail, 0) + extra  # some nodes with sched == 0 in a dimension, some unused, some overcommitted
    sched[rng.random(n) < 0.1] = 0
    sched[:, 0] *= 250  # cpu in quarter cores: Value() rounding away from zero matters
    avail = avail.copy()
    avail[:, 0] *= 250
    drv = drv.copy()
    exe = exe.copy()
    drv[:, 0] *= 250
    exe[:, 0] *= 250
    zone = rng.integers(0, n_zones, size=n).astype(np.uint32) * 7 + 3  # ids need not be dense
    return avail, sched, zone, D, X, drv, exe, k


@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("algo", [SAZ, AZA])
@pytest.mark.parametrize("n", [1, 2, 64, 65, 200, 1000])
def test_zoned_independent_batch_random(gf_ctx, algo, n, layout):
    rng = np.random.default_rng(31 * algo + n + 5 * len(layout))
    seen_feasible = False
    for tight_cluster in (True, False):
        for n_zones in (1, 3):
            a = 130
            avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, tight_cluster, layout, n_zones)
            _setup(gf_ctx, avail, sched, zone, D, X)
            apps = gangfit.make_apps(drv, exe, k)
            gpu = gf_ctx.fit_batch(IND, algo, apps)
            ref = ob.fit_independent(O_ALGO[algo], avail, ob.make_apps(drv, exe, k), D, X, closed_form=True,
                                     sched=sched, zone=zone)
            _assert_same(gpu, ref, apps)
            # the averages chooseBestResult compared, bit for bit
            avg = gf_ctx.avg_packing_efficiency(algo, apps, gpu)
            assert np.array_equal(_bits(avg), _bits(ref.avg_eff))
            seen_feasible = seen_feasible or bool(ref.results["has_capacity"].any())
    if n >= 64:
        assert seen_feasible


@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_avg_efficiency_of_plain_packers_bit_exact(gf_ctx, algo):
    rng = np.random.default_rng(77 + algo)
    for layout in ("merged", "general"):
        avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, 700, 200, False, layout, 1)
        k = np.minimum(k, 400)
        _setup(gf_ctx, avail, sched, None, D, X)
        apps = gangfit.make_apps(drv, exe, k)
        gpu = gf_ctx.fit_batch(IND, algo, apps)
        ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True, sched=sched)
        _assert_same(gpu, ref, apps)
        avg = gf_ctx.avg_packing_efficiency(algo, apps, gpu)
        assert np.array_equal(_bits(avg), _bits(ref.avg_eff))
        assert ref.results["has_capacity"].any()
        # minimal-fragmentation flavour of the same lists: `reserved` holds the driver only
        a0 = int(np.nonzero(ref.results["has_capacity"])[0][0])
        _, d, ex = ref.placement(a0)
        want = ob.avg_packing_efficiency_list(avail, sched, drv[a0], exe[a0], d, ex, reserved_includes_executors=False)
        got = gf_ctx.avg_packing_efficiency(gangfit.GF_ALGO_MINIMAL_FRAGMENTATION, apps[a0:a0 + 1],
                                            gangfit.BatchOut(gpu.results[a0:a0 + 1], np.zeros(1, dtype=np.uint64), ex))
        assert np.array_equal(_bits(got[0]), _bits(want))


def test_per_node_efficiency_map_bit_exact(gf_ctx):
    rng = np.random.default_rng(5)
    avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, 900, 40, False, "merged", 1)
    big = np.int64(1) << 61  # huge quantities: int64 -> float64 conversions must round like the reference's
    sched[:20, 1] += big
    avail[:20, 1] += big - rng.integers(0, 1 << 40, size=20)
    _setup(gf_ctx, avail, sched, None, D, X)
    ref = ob.fit_independent(0, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
    checked = 0
    for a in np.nonzero(ref.results["has_capacity"])[0][:6]:
        _, d, ex = ref.placement(int(a))
        want, _ = ob.packing_efficiency(avail, sched, drv[a], exe[a], d, ex)
        got = gf_ctx.packing_efficiencies(TIGHT, drv[a], exe[a], d, ex)
        assert np.array_equal(_bits(got), _bits(want))
        checked += 1
    assert checked > 0


def test_headline_size_single_az(gf_ctx):
    """10 000 nodes x 1 000 apps, 3 zones, against the closed-form oracle."""
    w = wl.headline(10000, 1000)
    s = w.snapshot
    zone = (wl.splitmix64(0xA2, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    _setup(gf_ctx, s.avail, s.sched, zone, s.driver_order, s.exec_order)
    apps = gangfit.make_apps(w.drv, w.exe, w.k)
    for algo in (SAZ, AZA):
        gpu = gf_ctx.fit_batch(IND, algo, apps)
        ref = ob.fit_independent(O_ALGO[algo], s.avail, ob.make_apps(w.drv, w.exe, w.k), s.driver_order,
                                 s.exec_order, closed_form=True, sched=s.sched, zone=zone)
        _assert_same(gpu, ref, apps)
        assert np.array_equal(_bits(gf_ctx.avg_packing_efficiency(algo, apps, gpu)), _bits(ref.avg_eff))
        assert ref.results["has_capacity"].mean() > 0.5


@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("algo", [SAZ, AZA])
@pytest.mark.parametrize("n", [3, 64, 300, 1500])
def test_zoned_fifo_chain_random(gf_ctx, algo, n, layout):
    """fitEarlierDrivers with a zone-aware packer: every earlier driver commits the zone chooseBestResult picked."""
    rng = np.random.default_rng(13 * algo + n + 3 * len(layout))
    for rep in range(3):
        a = 90
        avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, rep == 2, layout, 1 + rep)
        exe = np.maximum(exe, 1)
        k = np.minimum(k, 40).astype(np.int32)
        flags = (rng.random(a) < (0.9 if rep else 1.0)).astype(np.uint32)
        _setup(gf_ctx, avail, sched, zone, D, X)
        apps = gangfit.make_apps(drv, exe, k, flags)
        gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, apps)
        ref = ob.fit_fifo_chain(O_ALGO[algo], avail, ob.make_apps(drv, exe, k, flags), D, X, closed_form=True,
                                sched=sched, zone=zone)
        assert gpu.failed_at == ref.failed_at
        _assert_same(gpu, ref, apps)
        assert np.array_equal(gf_ctx.residual(), ref.avail_after)


def test_zoned_fifo_chain_headline_shape(gf_ctx):
    """C5-shaped chain (999 earlier drivers + 1) at 10 000 nodes, 3 zones, single-az-tightly-pack."""
    w = wl.headline(10000, 1000)
    s = w.snapshot
    zone = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    _setup(gf_ctx, s.avail, s.sched, zone, s.driver_order, s.exec_order)
    flags = np.ones(len(w.k), dtype=np.uint32)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, flags)
    gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, SAZ, apps)
    ref = ob.fit_fifo_chain(O_ALGO[SAZ], s.avail, ob.make_apps(w.drv, w.exe, w.k, flags), s.driver_order, s.exec_order,
                            closed_form=True, sched=s.sched, zone=zone)
    assert gpu.failed_at == ref.failed_at
    _assert_same(gpu, ref, apps)
    assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.parametrize("env", [{}, {"fifo_generic": 1}, {"lds_budget": 48000}],
                         ids=["lds-chain", "generic-chain", "lds-chain-global-tail"])
@pytest.mark.parametrize("algo", [SAZ, AZA])
def test_zoned_fifo_chain_kernel_variants(algo, env):
    """The LDS-resident chain of the zone-aware tightly-pack packers (gangfit_fifo_zoned.inc), its global-memory
    fallback, and the hybrid LDS/global table, with gangs large enough to spill the in-LDS run lists (> 64 nodes)."""
    ctx = gangfit.Context(0, options=env)
    rng = np.random.default_rng(2024 + algo)
    try:
        for rep in range(4):
            n, a = (3000, 80) if rep < 2 else (700, 150)
            avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, rep % 2 == 0, "merged", 1 + rep)
            exe = np.maximum(exe, 1)
            k = np.minimum(k, 400 if rep < 2 else 30).astype(np.int32)
            flags = (rng.random(a) < 0.9).astype(np.uint32)
            if rep == 3:  # a request that has no scaled form (cpu not a multiple of the table's 250 m unit): wide fallback
                drv[5, 0] += 1
            _setup(ctx, avail, sched, zone, D, X)
            apps = gangfit.make_apps(drv, exe, k, flags)
            gpu = ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, apps)
            ref = ob.fit_fifo_chain(O_ALGO[algo], avail, ob.make_apps(drv, exe, k, flags), D, X, closed_form=True,
                                    sched=sched, zone=zone)
            assert gpu.failed_at == ref.failed_at
            _assert_same(gpu, ref, apps)
            assert np.array_equal(ctx.residual(), ref.avail_after)
    finally:
        ctx.close()


@pytest.mark.parametrize("algo", [SAZ, AZA])
def test_zoned_fifo_chain_az_major_order(gf_ctx, algo):
    """The order the reference's extender actually produces: AZ-major (nodesorting.go:82-122) — every zone a contiguous range of
    the priority lists — at the headline size, 1 000 applications, against the literal oracle incl. residuals."""
    w = wl.headline(10000, 1000)
    s = w.snapshot
    zone = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    order = wl.reference_node_order(s.avail, zone)
    _setup(gf_ctx, s.avail, s.sched, zone, order, order)
    flags = (np.arange(len(w.k)) % 11 != 0).astype(np.uint32)
    apps = gangfit.make_apps(w.drv, w.exe, w.k, flags)
    gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, apps)
    ref = ob.fit_fifo_chain(O_ALGO[algo], s.avail, ob.make_apps(w.drv, w.exe, w.k, flags), order, order, sched=s.sched, zone=zone)
    assert gpu.failed_at == ref.failed_at
    _assert_same(gpu, ref, apps)
    assert np.array_equal(gf_ctx.residual(), ref.avail_after)
    assert ref.results["has_capacity"].sum() > 900


@pytest.mark.parametrize("nz", [15, 16, 17])
def test_zoned_fifo_chains_many_zones(gf_ctx, nz):
    """Sixteen candidate views are what one workgroup holds: with 15 a wavefront is left to expand the winner's placement / to
    patch the min-frag tables, with 16 none is (the winner does it itself), 17 candidates (az-aware with 16 zones, any packer
    with 17) run the generic chain."""
    from test_gpu_minfrag import SAZMF
    for seed in range(2):
        rng = np.random.default_rng(1000 + 7 * nz + seed)
        n, a = (700, 90) if seed == 0 else (2500, 70)
        avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, a, seed == 0, "merged", nz)
        exe = np.maximum(exe, 1)
        k = np.minimum(k, 30 if seed == 0 else 400).astype(np.int32)
        flags = (rng.random(a) < 0.9).astype(np.uint32)
        _setup(gf_ctx, avail, sched, zone, D, X)
        apps = gangfit.make_apps(drv, exe, k, flags)
        for algo, oalgo in ((AZA, O_ALGO[AZA]), (SAZ, O_ALGO[SAZ]), (SAZMF, ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION)):
            gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, apps)
            ref = ob.fit_fifo_chain(oalgo, avail, ob.make_apps(drv, exe, k, flags), D, X, sched=sched, zone=zone)
            assert gpu.failed_at == ref.failed_at
            _assert_same(gpu, ref, apps)
            assert np.array_equal(gf_ctx.residual(), ref.avail_after)


@pytest.mark.parametrize("nz", [2, 3, 5])
def test_zoned_fifo_chain_equal_zones_take_the_exact_sums(gf_ctx, nz):
    """Zones made of the same nodes in the same relative order: their candidates' averages are EQUAL, the chain kernels'
    (tree sum, bound) pairs cannot separate them and the reference's slice-order sums decide (the first zone of the driver
    order wins, single_az.go:75-97).  Second cluster: one zone's schedulable memory differs by one byte — a gap of ~1e-11
    relative, far above the bound, decided without the exact sums.  Third: quantities available beyond the schedulable ones
    (negative terms: no bound)."""
    from test_gpu_minfrag import SAZMF
    rng = np.random.default_rng(1000 + nz)
    per = 70
    for variant in ("equal", "one_byte", "negative"):
        base_sched = np.stack([rng.choice([16000, 32000, 64000], per), rng.choice([64, 128, 256], per) * GIB, np.zeros(per, dtype=np.int64)], axis=1).astype(np.int64)
        used = (rng.random((per, 3)) * 0.8 * base_sched).astype(np.int64)
        base_avail = base_sched - used
        n = per * nz
        avail = np.repeat(base_avail, nz, axis=0)
        sched = np.repeat(base_sched, nz, axis=0)
        zone = (np.arange(n) % nz).astype(np.uint32) + 11
        if variant == "one_byte":
            sched[zone == 11 + (nz - 1), 1] += 1
        if variant == "negative":
            sched[::7, 0] = avail[::7, 0] - 1000  # more available than schedulable
        order = np.arange(n)
        a = 90
        drv = np.stack([rng.choice([1000, 2000], a), rng.choice([2, 4], a) * GIB, np.zeros(a, dtype=np.int64)], axis=1).astype(np.int64)
        exe = np.stack([rng.choice([1000, 4000, 8000], a), rng.choice([4, 16], a) * GIB, np.zeros(a, dtype=np.int64)], axis=1).astype(np.int64)
        k = rng.integers(0, 9, a).astype(np.int32)
        flags = np.ones(a, dtype=np.uint32)
        _setup(gf_ctx, avail, sched, zone, order, order)
        apps = gangfit.make_apps(drv, exe, k, flags)
        for algo, oalgo in ((SAZ, O_ALGO[SAZ]), (AZA, O_ALGO[AZA]), (SAZMF, ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION)):
            for length in (1, 2, a):
                gpu = gf_ctx.fit_batch(gangfit.GF_MODE_FIFO_CHAIN, algo, apps[:length])
                ref = ob.fit_fifo_chain(oalgo, avail, ob.make_apps(drv[:length], exe[:length], k[:length], flags[:length]), order, order,
                                        sched=sched, zone=zone)
                assert gpu.failed_at == ref.failed_at
                _assert_same(gpu, ref, apps[:length])
                assert np.array_equal(gf_ctx.residual(), ref.avail_after)
            if variant == "equal":
                assert ref.results["has_capacity"].any()


@pytest.mark.parametrize("algo", [SAZ, AZA])
@pytest.mark.parametrize("n_zones", [1, 3, 5])
@pytest.mark.parametrize("n", [70, 700, 3000])
def test_zone_views_of_the_compact_gpu_table(algo, n_zones, n):
    """One-launch zone kernel on clusters whose gpu nodes are a minority (the compact gpu view exists: a zone's gangs of gpu executors
    are packed from the zone's sub-slots of it, placements as slots of the full table) — gangs that fit in one zone, in none, drivers
    on gpu nodes and elsewhere, against the oracle and against the same context without the view; the averages bit for bit."""
    rng = np.random.default_rng(7001 + 13 * algo + n_zones + n)
    for layout in ("merged", "identical"):
        for tight_cluster in (True, False):
            avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, n, 140, tight_cluster, layout, n_zones)
            frac = float(rng.choice([0.05, 0.12, 0.2]))
            has = rng.random(n) < frac
            avail[:, 2] = np.where(has, rng.integers(1, 9, size=n), rng.integers(-1, 1, size=n))
            sched[:, 2] = np.maximum(avail[:, 2], 0) + rng.integers(0, 3, size=n)
            exe[:, 2] = np.where(rng.random(len(exe)) < 0.7, rng.integers(1, 4, size=len(exe)), 0)
            drv[:, 2] = np.where(rng.random(len(drv)) < 0.3, 1, 0)
            k = np.where(rng.random(len(k)) < 0.7, np.minimum(k, rng.integers(0, 30, size=len(k))), k).astype(np.int32)
            apps = gangfit.make_apps(drv, exe, k)
            ref = ob.fit_independent(O_ALGO[algo], avail, ob.make_apps(drv, exe, k), D, X, closed_form=True, sched=sched, zone=zone)
            for opts in ({}, {"sparse_gpu": 0}):
                with gangfit.Context(0, options=opts) as ctx:
                    _setup(ctx, avail, sched, zone, D, X)
                    gpu = ctx.fit_batch(IND, algo, apps)
                    _assert_same(gpu, ref, apps)
                    assert np.array_equal(_bits(ctx.avg_packing_efficiency(algo, apps, gpu)), _bits(ref.avg_eff))
                    fits = ctx.fit_feasible(algo, apps)
                    assert np.array_equal(fits, np.asarray(ref.results["has_capacity"]).astype(bool))


@pytest.mark.parametrize("algo", [SAZ, AZA])
def test_long_gangs_take_the_run_averages(gf_ctx, algo):
    """Gangs of 1 .. 700 executors on nodes that take 1 .. 40 of them: placements of a few runs and of more than 63 (the entry-wise
    averages), more than 512 executors (the same), the driver's node inside the list and outside — the choice between the zones
    is made on float64 sums that must come out bit for bit."""
    rng = np.random.default_rng(9100 + algo)
    n, a = 2000, 160
    sched = np.zeros((n, 3), dtype=np.int64)
    sched[:, 0] = rng.integers(8, 65, size=n) * 1000
    sched[:, 1] = rng.integers(16, 257, size=n) * GIB
    used = rng.random((n, 2)) * 0.6
    avail = sched.copy()
    avail[:, 0] -= (used[:, 0] * sched[:, 0]).astype(np.int64) // 250 * 250
    avail[:, 1] -= (used[:, 1] * sched[:, 1]).astype(np.int64)
    zone = rng.integers(0, 3, size=n).astype(np.uint32)
    order = wl.reference_node_order(avail, zone)
    drv = np.zeros((a, 3), dtype=np.int64)
    exe = np.zeros((a, 3), dtype=np.int64)
    drv[:, 0] = rng.integers(1, 5, size=a) * 500
    drv[:, 1] = rng.integers(1, 9, size=a) * GIB
    exe[:, 0] = rng.integers(1, 9, size=a) * 250
    exe[:, 1] = rng.integers(1, 17, size=a) * (GIB // 2)
    k = rng.choice([1, 2, 7, 40, 64, 65, 130, 300, 511, 512, 513, 700], size=a).astype(np.int32)
    _setup(gf_ctx, avail, sched, zone, order, order)
    apps = gangfit.make_apps(drv, exe, k)
    gpu = gf_ctx.fit_batch(IND, algo, apps)
    ref = ob.fit_independent(O_ALGO[algo], avail, ob.make_apps(drv, exe, k), order, order, closed_form=True, sched=sched, zone=zone)
    _assert_same(gpu, ref, apps)
    assert np.array_equal(_bits(gf_ctx.avg_packing_efficiency(algo, apps, gpu)), _bits(ref.avg_eff))
    assert ref.results["has_capacity"].mean() > 0.5
