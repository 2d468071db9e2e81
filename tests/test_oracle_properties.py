"""Properties of the reference's packers that follow from SURVEY.md section 8's array restatement, checked on the C oracle with
generated clusters (hypothesis): independent of HOW the oracle computes its answers — only capacities by the closed form
(capacity.go:36-75) and the definitions of the three packers are used.  CPU only; the GPU parity tests compare the kernels with this
oracle, so what holds for it holds for them."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import binding as ob

INF = 1 << 40


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


@settings(max_examples=300, deadline=None)
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


@settings(max_examples=300, deadline=None)
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


@settings(max_examples=300, deadline=None)
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
