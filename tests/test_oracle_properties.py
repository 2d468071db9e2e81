"""Properties of the reference's packers that follow from SURVEY.md section 8's array restatement, checked on the C oracle with
generated clusters (hypothesis): independent of HOW the oracle computes its answers — only capacities by the closed form
(capacity.go:36-75) and the definitions of the three packers are used.  CPU only; the GPU parity tests compare the kernels with this
oracle, so what holds for it holds for them."""
import os

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import binding as ob

INF = 1 << 40
# The same examples on every run (a parity suite must not be a lottery); GANGFIT_PROPERTY_EXAMPLES=<n> draws n fresh random ones
# per property instead — how the properties were hunted with before they were committed (20 000 examples each).
_HUNT = int(os.environ.get("GANGFIT_PROPERTY_EXAMPLES", "0"))


def _settings(n):
    return settings(max_examples=_HUNT or n, deadline=None, derandomize=not _HUNT)


def _cap(avail_row, base, exe):
    """cap(n, base): min over dims of 0 if avail - base < 0, +inf if exe == 0, floor((avail - base) / exe) otherwise."""
    out = INF
    for a, b, e in zip(avail_row, base, exe):
        if a - b < 0:
            return 0
        if e != 0:
            out = min(out, (a - b) // e)
    return out


def _fits(drv, avail_row):  # NOT (drv > avail): no component greater (resources.go:239-241)
    return all(d <= a for d, a in zip(drv, avail_row))


cluster = st.integers(1, 9).flatmap(lambda n: st.tuples(
    st.lists(st.tuples(st.integers(-2, 12), st.integers(-2, 12), st.integers(0, 3)), min_size=n, max_size=n),
    st.permutations(list(range(n))), st.permutations(list(range(n))), st.integers(0, n), st.integers(0, n)))
request = st.tuples(st.tuples(st.integers(0, 4), st.integers(0, 4), st.integers(0, 1)),
                    st.tuples(st.integers(0, 4), st.integers(0, 4), st.integers(0, 1)), st.integers(0, 14))


def _check_common(algo, avail, D, X, drv, exe, k):
    ok, d, ex = ob.spark_binpack(algo, avail, drv, exe, k, D, X)
    ex = [int(v) for v in ex]

    def total(dn):
        return sum(min(_cap(avail[n], drv if n == dn else (0, 0, 0), exe), k) for n in X)

    fitting = [dn for dn in D if _fits(drv, avail[dn])]
    feasible = [dn for dn in fitting if total(dn) >= k]
    assert ok == bool(feasible)                      # binpack.go:60-87 with the O(N) predicate of SURVEY section 8
    if not ok:
        return None
    assert d == feasible[0]                          # the FIRST driver candidate whose pack succeeds
    assert len(ex) == k
    counts = {n: ex.count(n) for n in set(ex)}
    for n, c in counts.items():
        assert n in X and c <= _cap(avail[n], drv if n == d else (0, 0, 0), exe)
    return d, ex


@_settings(300)
@given(cluster, request)
def test_tightly_pack_is_the_prefix_of_the_run_length_sequence(cl, rq):
    avail, dperm, xperm, nd, nx = cl
    D, X = list(dperm[:nd]), list(xperm[:nx])
    drv, exe, k = rq
    got = _check_common(ob.ALGO_TIGHTLY_PACK, avail, D, X, drv, exe, k)
    if got is None:
        return
    d, ex = got
    want = []
    for n in X:                                      # pack_tightly.go:45-61
        want += [n] * min(_cap(avail[n], drv if n == d else (0, 0, 0), exe), k - len(want))
    assert ex == want[:k]


@_settings(300)
@given(cluster, request)
def test_distribute_evenly_is_round_robin_over_the_nodes_with_room(cl, rq):
    avail, dperm, xperm, nd, nx = cl
    D, X = list(dperm[:nd]), list(xperm[:nx])
    drv, exe, k = rq
    got = _check_common(ob.ALGO_DISTRIBUTE_EVENLY, avail, D, X, drv, exe, k)
    if got is None:
        return
    d, ex = got
    caps = {n: _cap(avail[n], drv if n == d else (0, 0, 0), exe) for n in X}
    want, r = [], 1
    while len(want) < k:                             # distribute_evenly.go:49-71: pass r visits the nodes with capacity >= r
        row = [n for n in X if caps[n] >= r]
        assert row                                   # feasible: some node always has room
        want += row
        r += 1
    assert ex == want[:k]


@_settings(300)
@given(cluster, request)
def test_minimal_fragmentation_places_k_within_the_capacities(cl, rq):
    avail, dperm, xperm, nd, nx = cl
    D, X = list(dperm[:nd]), list(xperm[:nx])
    drv, exe, k = rq
    got = _check_common(ob.ALGO_MINIMAL_FRAGMENTATION, avail, D, X, drv, exe, k)
    if got is None or k == 0:
        return
    d, ex = got
    caps = {n: _cap(avail[n], drv if n == d else (0, 0, 0), exe) for n in X}
    # a node that could take the whole gang exists => the gang sits on ONE node, and no smaller sufficient node was passed over
    # unless the "avoid mostly empty nodes" rule chose among the smaller ones (minimal_fragmentation.go:64-91, 103-110)
    if any(c >= k for c in caps.values()) and len(set(ex)) == 1:
        assert caps[ex[0]] >= k
    # runs are contiguous: every node occupies one run of the placement list (internalMinimalFragmentation appends whole runs)
    seen, prev = set(), None
    for n in ex:
        if n != prev:
            assert n not in seen
            seen.add(n)
            prev = n


chain = st.lists(st.tuples(request, st.booleans()), min_size=1, max_size=6)


@_settings(200)
@given(cluster, chain, st.sampled_from([ob.ALGO_TIGHTLY_PACK, ob.ALGO_DISTRIBUTE_EVENLY, ob.ALGO_MINIMAL_FRAGMENTATION]))
def test_fifo_replay_is_the_single_decision_applied_in_order_with_the_map_quirk(cl, apps, algo):
    """fitEarlierDrivers + the final pack (resource.go:224-262, 309-328) rebuilt from ONE-decision calls: every earlier driver is
    packed against what its predecessors left; a feasible one subtracts ONE executor request per DISTINCT executor node and the
    driver request only where no executor landed (sparkResourceUsage builds a map: sparkpods.go:139-146); one that does not fit
    is skipped when flagged and aborts the request otherwise; the last application is packed and nothing is subtracted."""
    avail, dperm, xperm, nd, nx = cl
    D, X = list(dperm[:nd]), list(xperm[:nx])
    drv = [list(a[0][0]) for a in apps]
    exe = [list(a[0][1]) for a in apps]
    k = [a[0][2] for a in apps]
    flags = [ob.APP_SKIPPABLE if a[1] else 0 for a in apps]
    out = ob.fit_fifo_chain(algo, avail, ob.make_apps(drv, exe, k, flags), D, X)
    table = [list(r) for r in avail]
    failed = -1
    for i in range(len(apps)):
        ok, d, ex = ob.spark_binpack(algo, table, drv[i], exe[i], k[i], D, X)
        got_ok, got_d, got_ex = out.placement(i)
        assert out.results[i]["evaluated"] != 0
        assert (ok, d if ok else ob.NO_NODE) == (got_ok, got_d)
        if ok:
            assert [int(v) for v in ex] == [int(v) for v in got_ex]
        last = i == len(apps) - 1
        if last:
            break
        if not ok:
            if flags[i]:
                continue
            failed = i
            break
        hosts = set(int(v) for v in ex)
        for n in hosts:
            for j in range(3):
                table[n][j] -= exe[i][j]
        if d not in hosts:
            for j in range(3):
                table[d][j] -= drv[i][j]
    assert out.failed_at == failed
    if failed >= 0:
        assert all(out.results[j]["evaluated"] == 0 for j in range(failed + 1, len(apps)))
    assert out.avail_after.tolist() == table


zoned_cluster = st.integers(1, 8).flatmap(lambda n: st.tuples(
    st.lists(st.tuples(st.integers(0, 12), st.integers(0, 12), st.integers(0, 2)), min_size=n, max_size=n),          # used
    st.lists(st.tuples(st.integers(1, 12), st.integers(1, 12), st.integers(0, 2)), min_size=n, max_size=n),          # schedulable
    st.lists(st.integers(0, 2), min_size=n, max_size=n),                                                              # zone
    st.permutations(list(range(n))), st.permutations(list(range(n))), st.integers(0, n), st.integers(0, n)))


@_settings(250)
@given(zoned_cluster, request,
       st.sampled_from([(ob.ALGO_SINGLE_AZ_TIGHTLY_PACK, ob.ALGO_TIGHTLY_PACK, True),
                        (ob.ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION, ob.ALGO_MINIMAL_FRAGMENTATION, False),
                        (ob.ALGO_AZ_AWARE_TIGHTLY_PACK, ob.ALGO_TIGHTLY_PACK, True)]))
def test_zone_wrappers_are_the_inner_packer_per_zone_and_the_strictly_best_average(cl, rq, algos):
    """getSingleAZSparkBinFunction + chooseBestResult (single_az.go:23-97) rebuilt from the INNER packer run on every zone's
    sub-orders and the averages of ComputeAvgPackingEfficiency: zones in order of first appearance in the driver order, zones
    without executor candidates skipped, the first feasible zone with the strictly highest average Max wins (from 0.0);
    az-aware-tightly-pack falls back to the plain pack (az_aware_pack_tightly.go:33-37)."""
    used, sched, zone, dperm, xperm, nd, nx = cl
    wrapper, inner, reserves_execs = algos
    avail = [[s - u for s, u in zip(srow, urow)] for srow, urow in zip(sched, used)]
    D, X = list(dperm[:nd]), list(xperm[:nx])
    drv, exe, k = rq
    ok, d, ex = ob.spark_binpack(wrapper, avail, drv, exe, k, D, X, sched=sched, zone=zone)
    zones = []
    for n in D:
        if zone[n] not in zones:
            zones.append(zone[n])
    best, best_max = None, 0.0
    for z in zones:
        Dz, Xz = [n for n in D if zone[n] == z], [n for n in X if zone[n] == z]
        if not Xz:
            continue
        zok, zd, zex = ob.spark_binpack(inner, avail, drv, exe, k, Dz, Xz)
        if not zok:
            continue
        avg = ob.avg_packing_efficiency_list(avail, sched, drv, exe, zd, [int(v) for v in zex], reserved_includes_executors=reserves_execs)
        if best_max < avg[3]:
            best, best_max = (zd, [int(v) for v in zex]), float(avg[3])
    if best is None and wrapper == ob.ALGO_AZ_AWARE_TIGHTLY_PACK:
        pok, pd, pex = ob.spark_binpack(inner, avail, drv, exe, k, D, X)
        best = (pd, [int(v) for v in pex]) if pok else None
    assert ok == (best is not None)
    if ok:
        assert (d, [int(v) for v in ex]) == best
