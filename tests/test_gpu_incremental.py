"""Incremental FIFO chains (gf_fit_batch resumes from the checkpoints of the previous chain when the queues share a prefix,
include/gangfit.h "Incremental FIFO chains"): every answer must be the answer of a FULL replay by the literal oracle — results,
placements, chain_failed_at and the residual table — whatever the cache did.  The reference replays every earlier driver
on every Filter (internal/extender/resource.go:309-328); parity of FIFO replay itself is unpinned by reference tests (see
oracle/gangfit_oracle.c header), these tests pin the resumed chain to the replayed one."""
import numpy as np
import pytest

import gangfit
from gangfit import workloads as wl
from oracle import binding as ob
from test_gpu_parity import _assert_same, _random_problem

pytestmark = pytest.mark.gpu

TIGHT, EVEN = gangfit.GF_ALGO_TIGHTLY_PACK, gangfit.GF_ALGO_DISTRIBUTE_EVENLY
FIFO = gangfit.GF_MODE_FIFO_CHAIN


def _check(ctx, algo, avail, D, X, drv, exe, k, flags, closed_form=False):
    apps = gangfit.make_apps(drv, exe, k, flags)
    gpu = ctx.fit_batch(FIFO, algo, apps)
    ref = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe, k, flags), D, X, closed_form=closed_form)
    assert gpu.failed_at == ref.failed_at
    _assert_same(gpu, ref, apps)
    assert np.array_equal(ctx.residual(), ref.avail_after)
    return ref


@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_growing_queue_resumes_and_matches_full_replay(algo):
    """Creation-order heads on an unchanged snapshot: driver j + 1's chain is driver j's chain plus one application.  Every
    chain length from 1 to 140 (all staging / checkpoint boundaries: 31 | 32 | 33, 63 | 64 | 65, ...)."""
    rng = np.random.default_rng(31 + algo)
    avail, D, X, drv, exe, k = _random_problem(rng, 900, 140, tight_cluster=False, layout="merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 25).astype(np.int32)
    k[::7] = 0
    flags = np.ones(140, dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(avail)
        ctx.set_orders(D, X)
        ctx.chain_cache_stats(reset=True)
        for n in range(1, 141):
            _check(ctx, algo, avail, D, X, drv[:n], exe[:n], k[:n], flags[:n])
        chains, resumed, evaluated, skipped = ctx.chain_cache_stats()
        assert chains == 140
        # The chain of n applications follows the chain of n - 1, which left its TIP: the table before its last application,
        # n - 2.  It resumes there (two applications evaluated) unless it crosses a checkpoint boundary — (n - 1) % 32 == 0 —: it
        # then starts from the last checkpoint, so that the boundary's dump is made (n = 33: no checkpoint yet, a full replay).

        def first_evaluated(n):
            if n >= 3 and (n - 1) % 32 != 0:
                return n - 2
            return ((n - 2) // 32) * 32 if n >= 2 else 0

        assert resumed == sum(first_evaluated(n) > 0 for n in range(1, 141)) == 137
        assert skipped == sum(first_evaluated(n) for n in range(1, 141))
        assert evaluated + skipped == sum(range(1, 141))


@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_headline_creation_order_heads(algo):
    """10 000 nodes, the headline queue: Filters for drivers 900 .. 999 in creation order (the regime in which a thousand
    drivers are pending at all), every one against the oracle's full replay; then the rotated heads of bench.py (a cold
    chain each) in between to show that a miss is a plain replay."""
    w = wl.headline(10000, 1000)
    s = w.snapshot
    flags = np.ones(len(w.k), dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        ctx.chain_cache_stats(reset=True)
        for n in list(range(900, 1001, 7)) + [1000, 1000, 999]:
            _check(ctx, algo, s.avail, s.driver_order, s.exec_order, w.drv[:n], w.exe[:n], w.k[:n], flags[:n], closed_form=True)
        chains, resumed, evaluated, skipped = ctx.chain_cache_stats()
        assert resumed == chains - 1 and skipped > 5 * evaluated
        for r in (1, 2):
            _check(ctx, algo, s.avail, s.driver_order, s.exec_order, np.roll(w.drv, -r, axis=0), np.roll(w.exe, -r, axis=0),
                   np.roll(w.k, -r), flags, closed_form=True)
        assert ctx.chain_cache_stats()[1] == resumed  # rotated queues share no prefix


def test_divergence_in_the_middle_and_shrinking_queue():
    rng = np.random.default_rng(5)
    avail, D, X, drv, exe, k = _random_problem(rng, 2000, 200, tight_cluster=False, layout="merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 30).astype(np.int32)
    flags = np.ones(200, dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(avail)
        ctx.set_orders(D, X)
        _check(ctx, TIGHT, avail, D, X, drv, exe, k, flags)
        ctx.chain_cache_stats(reset=True)
        for pos in (150, 70, 64, 63, 32, 31, 0, 199, 198):
            drv = drv.copy()
            drv[pos] = (drv[pos] + 1) % 9  # another driver request at `pos`: everything from the checkpoint before it replays
            _check(ctx, TIGHT, avail, D, X, drv, exe, k, flags)
        st = ctx.chain_cache_stats()
        assert st[0] == 9 and st[1] == 7  # pos 31 and 0 leave no checkpoint in the common prefix
        # (pos 199 is the driver being filtered: the queues agree up to it and the previous chain's tip — the table before
        #  application 199 — serves; at pos 198 the tip is behind the common prefix and checkpoint 6 does)
        assert st[3] == 128 + 64 + 64 + 32 + 32 + 199 + 192
        for n in (120, 64, 65, 33, 200):  # a driver was scheduled / deleted: shorter queues, then the long one again
            _check(ctx, TIGHT, avail, D, X, drv[:n], exe[:n], k[:n], flags[:n])


@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_the_tip_serves_the_next_filter_and_the_same_filter_again(algo):
    """The Filter of driver j + 1 after the Filter of driver j evaluates two applications (the tip: the table before the last
    application of the previous chain), the same Filter again one; a chain that aborts leaves its tip at the application it
    aborted at; a queue that diverges before the tip falls back to the checkpoints.  Every answer against the full replay."""
    rng = np.random.default_rng(77 + algo)
    avail, D, X, drv, exe, k = _random_problem(rng, 1500, 120, tight_cluster=False, layout="merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 25).astype(np.int32)
    flags = np.ones(120, dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(avail)
        ctx.set_orders(D, X)

        def run(n, d=drv, e=exe, kk=k, f=flags):
            ctx.chain_cache_stats(reset=True)
            ref = _check(ctx, algo, avail, D, X, d[:n], e[:n], kk[:n], f[:n])
            _, resumed, evaluated, skipped = ctx.chain_cache_stats()
            return ref, evaluated, skipped

        assert run(70)[1:] == (70, 0)
        assert run(71)[1:] == (2, 69)     # from the tip of the chain of 70: applications 69 and 70
        assert run(71)[1:] == (1, 70)     # the same Filter again: the filtered driver only
        assert run(72)[1:] == (2, 70)
        assert run(60)[1:] == (60 - 32, 32)  # a shorter queue: the tip (71) lies behind it, checkpoint 1 serves
        assert run(61)[1:] == (2, 59)
        d2 = drv.copy()
        d2[40] = (d2[40] + 1) % 9
        assert run(62, d=d2)[1:] == (62 - 32, 32)  # diverges at 40, before the tip (60): checkpoint 1
        # crossing a checkpoint boundary: the chain of 97 applications ends at application 96 = 3 * 32 — it starts from the last
        # checkpoint so that the dump of checkpoint 3 is made, and the next one uses the tip again
        assert run(96, d=d2)[1:] == (96 - 32, 32)  # (checkpoint 2 did not exist yet: the chain of 62 ended before application 64)
        assert run(97, d=d2)[1:] == (97 - 64, 64)
        assert run(98, d=d2)[1:] == (2, 96)
        # an aborting chain: its tip is the table before the application it aborted at
        f2 = flags.copy()
        e2 = exe.copy()
        f2[105] = 0
        e2[105] = (10 ** 6, 10 ** 6, 0)
        ref, evaluated, skipped = run(110, d=d2, e=e2, f=f2)
        assert ref.failed_at == 105 and skipped == 97 and evaluated == 105 - 97 + 1  # (from the tip of the chain of 98)
        ref, evaluated, skipped = run(111, d=d2, e=e2, f=f2)
        assert ref.failed_at == 105 and (evaluated, skipped) == (1, 105)  # resumes AT the aborting application, aborts again
        f2[105] = 1  # skippable after all: the record changed at 105, the tip (the table before it) still serves
        ref, evaluated, skipped = run(111, d=d2, e=e2, f=f2)
        assert ref.failed_at == -1 and skipped == 105


def test_aborting_chain_resumes_and_aborts_again():
    """A non-skippable earlier driver that does not fit ends every later Filter with failure-earlier-driver
    (resource.go:249-251): the resumed chain must abort at the same application and report the rest as not evaluated."""
    rng = np.random.default_rng(8)
    avail, D, X, drv, exe, k = _random_problem(rng, 700, 150, tight_cluster=False, layout="merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 20).astype(np.int32)
    flags = np.ones(150, dtype=np.uint32)
    flags[100] = 0
    exe[100] = (10 ** 6, 10 ** 6, 0)  # nothing hosts this executor
    k[100] = 3
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(avail)
        ctx.set_orders(D, X)
        for n in (90, 100, 101, 102, 130, 150):
            ref = _check(ctx, TIGHT, avail, D, X, drv[:n], exe[:n], k[:n], flags[:n])
            assert ref.failed_at == (100 if n > 101 else -1)
        flags[100] = 1  # the driver is young enough to be skipped after all (resource.go:264-270): the chain goes on
        _check(ctx, TIGHT, avail, D, X, drv, exe, k, flags)


def test_every_install_drops_the_cache():
    rng = np.random.default_rng(13)
    avail, D, X, drv, exe, k = _random_problem(rng, 1200, 100, tight_cluster=False, layout="merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 20).astype(np.int32)
    flags = np.ones(100, dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(avail)
        ctx.set_orders(D, X)
        _check(ctx, TIGHT, avail, D, X, drv, exe, k, flags)
        _check(ctx, TIGHT, avail, D, X, drv, exe, k, flags)
        assert ctx.chain_cache_stats(reset=True)[1] == 1
        avail2 = avail.copy()
        avail2[D[D < len(avail)][:40]] //= 2  # a reservation landed on the front of the order
        ctx.set_snapshot(avail2)
        ctx.set_orders(D, X)
        _check(ctx, TIGHT, avail2, D, X, drv, exe, k, flags)  # same queue, new snapshot: replay
        assert ctx.chain_cache_stats(reset=True)[1] == 0
        _check(ctx, EVEN, avail2, D, X, drv, exe, k, flags)   # same queue, another packer: replay
        assert ctx.chain_cache_stats(reset=True)[1] == 0
        X2 = X[::-1].copy()
        ctx.set_orders(X2, X2)
        _check(ctx, EVEN, avail2, X2, X2, drv, exe, k, flags)  # new orders: replay
        _check(ctx, EVEN, avail2, X2, X2, drv, exe, k, flags)
        assert ctx.chain_cache_stats(reset=True)[1] == 1
        ctx.set_option("chain_cache", 0)
        _check(ctx, EVEN, avail2, X2, X2, drv, exe, k, flags)
        assert ctx.chain_cache_stats()[0] == 0


def test_finer_request_changes_the_units_and_replays():
    """The checkpoints are scaled in the chain's units (gcd of the table's units and every request of the queue): a new
    application with a finer request changes them, and the chain replays in the new units."""
    rng = np.random.default_rng(21)
    avail, D, X, drv, exe, k = _random_problem(rng, 800, 90, tight_cluster=False, layout="merged")
    avail = avail * 8
    drv, exe = drv * 8, np.maximum(exe, 1) * 8
    k = np.minimum(k, 20).astype(np.int32)
    flags = np.ones(90, dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(avail)
        ctx.set_orders(D, X)
        _check(ctx, TIGHT, avail, D, X, drv[:80], exe[:80], k[:80], flags[:80])
        ctx.chain_cache_stats(reset=True)
        _check(ctx, TIGHT, avail, D, X, drv[:85], exe[:85], k[:85], flags[:85])
        assert ctx.chain_cache_stats(reset=True)[1] == 1
        drv[87] = (4, 8, 0)  # half the unit
        _check(ctx, TIGHT, avail, D, X, drv, exe, k, flags)
        assert ctx.chain_cache_stats(reset=True)[1] == 0
        _check(ctx, TIGHT, avail, D, X, drv, exe, k, flags)
        assert ctx.chain_cache_stats(reset=True)[1] == 1


@pytest.mark.parametrize("algo", [TIGHT, EVEN])
def test_resume_with_a_global_table_tail(algo):
    """Tables larger than the LDS front: the checkpoint is LDS blocks + the global tail (24 000 nodes; and a small LDS
    budget on 3 000 nodes so that most of the table is tail)."""
    w = wl.headline(24000, 300)
    s = w.snapshot
    flags = np.ones(len(w.k), dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_orders(s.driver_order, s.exec_order)
        for n in (200, 201, 260, 300, 299):
            _check(ctx, algo, s.avail, s.driver_order, s.exec_order, w.drv[:n], w.exe[:n], w.k[:n], flags[:n], closed_form=True)
        assert ctx.chain_cache_stats()[1] == 4
    rng = np.random.default_rng(77 + algo)
    avail, D, X, drv, exe, k = _random_problem(rng, 3000, 160, tight_cluster=False, layout="merged")
    exe = np.maximum(exe, 1)
    k = np.minimum(k, 40).astype(np.int32)
    flags = (rng.random(160) < 0.97).astype(np.uint32)
    with gangfit.Context(0, options={"lds_budget": 24000}) as ctx:
        ctx.set_snapshot(avail)
        ctx.set_orders(D, X)
        for n in (100, 101, 130, 160, 129):
            _check(ctx, algo, avail, D, X, drv[:n], exe[:n], k[:n], flags[:n])


SAZ, AZA = gangfit.GF_ALGO_SINGLE_AZ_TIGHTLY_PACK, gangfit.GF_ALGO_AZ_AWARE_TIGHTLY_PACK
MF, SAZMF = gangfit.GF_ALGO_MINIMAL_FRAGMENTATION, gangfit.GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION
O_ALGO = {SAZ: ob.ALGO_SINGLE_AZ_TIGHTLY_PACK, AZA: ob.ALGO_AZ_AWARE_TIGHTLY_PACK, MF: ob.ALGO_MINIMAL_FRAGMENTATION,
          SAZMF: ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION}


def _zcheck(ctx, algo, avail, sched, zone, D, X, drv, exe, k, flags):
    apps = gangfit.make_apps(drv, exe, k, flags)
    gpu = ctx.fit_batch(FIFO, algo, apps)
    ref = ob.fit_fifo_chain(O_ALGO[algo], avail, ob.make_apps(drv, exe, k, flags), D, X, closed_form=True, sched=sched, zone=zone)
    assert gpu.failed_at == ref.failed_at
    _assert_same(gpu, ref, apps)
    assert np.array_equal(ctx.residual(), ref.avail_after)
    return ref


@pytest.mark.parametrize("options", [{}, {"lds_budget": 60000}], ids=["lds-resident", "global-tail"])
@pytest.mark.parametrize("algo", [SAZ, AZA, MF, SAZMF])
def test_zone_aware_and_minfrag_chains_resume(algo, options):
    """The chains of the zone-aware and minimal-fragmentation packers (gangfit_fifo_zoned.inc, gangfit_fifo_minfrag.inc)
    dump the same checkpoints: growing queues, a divergence in the middle, an aborting driver, a shrinking queue — every
    answer against the oracle's full replay (efficiency comparisons of chooseBestResult included: they decide the zone)."""
    from test_gpu_zones import _zoned_problem

    rng = np.random.default_rng(1000 + algo)
    n_apps = 150
    avail, sched, zone, D, X, drv, exe, k = _zoned_problem(rng, 3000, n_apps, False, "merged", 3)
    exe = np.maximum(exe, 1)
    exe[:, 0] = np.maximum(exe[:, 0], 250)
    k = np.minimum(k, 25).astype(np.int32)
    t = rng.integers(0, 6, size=n_apps)  # a handful of templates, like a real queue
    drv, exe = drv[t], exe[t]
    flags = np.ones(n_apps, dtype=np.uint32)
    with gangfit.Context(0, options=options) as ctx:
        ctx.set_snapshot(avail, sched)
        ctx.set_zones(zone)
        ctx.set_orders(D, X)
        ctx.chain_cache_stats(reset=True)
        for n in (40, 41, 64, 65, 66, 97, 130, 150, 150, 149, 96):
            _zcheck(ctx, algo, avail, sched, zone, D, X, drv[:n], exe[:n], k[:n], flags[:n])
        # checkpoints every 32 applications while the table sits in LDS, every 128 where a dump copies a table in global memory
        # (which of the two applies follows from the kernel's LDS geometry: the counters tell)
        lengths = (40, 41, 64, 65, 66, 97, 130, 150, 150, 149, 96)

        def expect(shift, tip):
            skipped = resumed = 0
            for prev, n in zip(lengths, lengths[1:]):
                n_ckpt = (prev - 1) >> shift  # what the previous chain left behind
                common = min(prev, n) - 1     # the common prefix ends before either queue's last application
                a0 = min(common >> shift, n_ckpt) << shift
                tip_at = prev - 1             # ... and its tip (whole table in LDS): the table before its last application
                if tip and a0 < tip_at <= common and ((n - 1) >> shift) == (tip_at >> shift):
                    a0 = tip_at
                skipped += a0
                resumed += a0 > 0
            return resumed, skipped

        st = ctx.chain_cache_stats(reset=True)
        assert st[0] == 11
        assert expect(5, False) == (10, 32 + 32 + 32 + 64 + 64 + 96 + 128 + 128 + 128 + 64)
        assert expect(5, True) == (10, 39 + 40 + 32 + 64 + 64 + 96 + 129 + 149 + 128 + 64)
        if not options:  # the whole table in LDS: checkpoints every 32 applications and the tip
            shift = 5
            assert (st[1], st[3]) == expect(5, True)
        else:            # a smaller LDS budget: whether the table still fits (tip), and which interval applies where it does not,
            #              follows from the kernel's LDS geometry — the counters tell
            got = (st[1], st[3])
            shift, tip = next(((sh, tp) for sh, tp in ((5, True), (5, False), (7, False)) if got == expect(sh, tp)), (None, None))
            assert shift is not None, got
        drv2 = drv.copy()
        drv2[70, 1] += 1
        _zcheck(ctx, algo, avail, sched, zone, D, X, drv2, exe, k, flags)
        assert ctx.chain_cache_stats(reset=True)[3] == (64 if shift == 5 else 0)
        flags2 = flags.copy()
        flags2[100] = 0
        exe2 = exe.copy()
        exe2[100] = (250 * 10 ** 6, 10 ** 6, 0)  # nothing hosts this executor: failure-earlier-driver at 100
        for n in (120, 150):
            ref = _zcheck(ctx, algo, avail, sched, zone, D, X, drv2[:n], exe2[:n], k[:n], flags2[:n])
            assert ref.failed_at == 100
        assert ctx.chain_cache_stats(reset=True)[3] == (96 + 96 if shift == 5 else 0)


def test_headline_single_az_creation_order_heads():
    """single-az-tightly-pack — what production and every reference test select — on the headline cluster with three zones in
    the reference's AZ-major order: consecutive Filters resume, results equal the full replay."""
    w = wl.headline(10000, 1000)
    s = w.snapshot
    zone3 = (wl.splitmix64(0xA3, len(s.avail), 9) % np.uint64(3)).astype(np.uint32)
    order = wl.reference_node_order(s.avail, zone3)
    flags = np.ones(len(w.k), dtype=np.uint32)
    with gangfit.Context(0) as ctx:
        ctx.set_snapshot(s.avail, s.sched)
        ctx.set_zones(zone3)
        ctx.set_orders(order, order)
        ctx.chain_cache_stats(reset=True)
        for n in (960, 961, 975, 1000, 1000):
            _zcheck(ctx, SAZ, s.avail, s.sched, zone3, order, order, w.drv[:n], w.exe[:n], w.k[:n], flags[:n])
        st = ctx.chain_cache_stats()
        assert st[1] == 4 and st[3] == 928 + 960 + 960 + 999  # (the same Filter again resumes from the chain's tip, application 999)
