"""findNodes of the failover reconciler (internal/extender/failover.go:412-436, call site :368, subtraction :159).

No reference test calls findNodes or the reconciler (internal/extender has resource_test.go, sparkpods_test.go and
unschedulablepods_test.go only): PARITY UNPINNED — three restatements (literal C, closed-form C, pure-Python dicts)
must agree, hand-derived KATs pin the over-add quirk, and the HIP path (gf_find_nodes) is compared with the literal one."""
import numpy as np
import pytest

from oracle import binding as ob
from oracle import pyoracle as po
from test_gpu_parity import _random_problem

NO = 0xFFFFFFFF

# hand-derived from failover.go:418-435 (2-D shown, gpu = 0); order = [n0, n1, n2]
#   (avail, exe, k) -> (placements, adds per node)
KATS = [
    # n0 holds 4 (5th add fails and STAYS: adds 5), count reached on n1 after one add (no over-add there), n2 never reached
    ("partial fill then stop", [[4, 4, 0], [2, 2, 0], [8, 8, 0]], [1, 1, 0], 5, [0, 0, 0, 0, 1], [5, 1, 0]),
    # nothing fits anywhere: every node is visited and charged one failing add; result is empty (partial results are legal)
    ("nothing fits", [[1, 1, 0], [0, 5, 0], [2, 0, 0]], [3, 3, 0], 2, [], [1, 1, 1]),
    # cluster too small: all nodes exhausted -> each keeps placed + 1
    ("cluster too small", [[2, 2, 0], [3, 3, 0], [1, 9, 0]], [1, 1, 0], 9, [0, 0, 1, 1, 1, 2], [3, 4, 2]),
    # exact fill of n0 reaches the count on n0: returns before the failing add
    ("count reached exactly at capacity", [[3, 3, 0], [9, 9, 0], [9, 9, 0]], [1, 1, 0], 3, [0, 0, 0], [3, 0, 0]),
    # negative availability: the first add already exceeds -> one add, nothing placed there
    ("overcommitted node", [[-1, 5, 0], [2, 2, 0], [0, 0, 0]], [1, 1, 0], 2, [1, 1], [1, 2, 0]),
    # zero-size executor: adding zero never exceeds a non-negative node -> all k on the first such node
    ("zero-size executor", [[-1, 0, 0], [0, 0, 0], [5, 5, 0]], [0, 0, 0], 4, [1, 1, 1, 1], [1, 4, 0]),
    # gpu is the binding dimension
    ("gpu bound", [[8, 8, 1], [8, 8, 0], [8, 8, 2]], [1, 1, 1], 3, [0, 2, 2], [2, 1, 2]),
]


@pytest.mark.parametrize("name,avail,exe,k,want,adds", KATS, ids=[c[0] for c in KATS])
def test_hand_derived_kats(name, avail, exe, k, want, adds):
    for closed in (False, True):
        r = ob.find_nodes(avail, [exe], [k], [0, 1, 2], closed_form=closed)
        assert r.placement(0).tolist() == want and int(r.placed[0]) == len(want), (name, closed)
        assert r.adds[0].tolist() == adds, (name, closed)
        # availableResources.Sub(reserved): over-adds included (an exhausted node goes negative by up to one executor)
        assert r.avail_after.tolist() == [[a[j] - adds[n] * exe[j] for j in range(3)] for n, a in enumerate(avail)]
    names, reserved = po.find_nodes(k, exe, {str(n): list(a) for n, a in enumerate(avail)}, ["0", "1", "2"])
    assert [int(n) for n in names] == want
    assert [reserved.get(str(n), [0, 0, 0]) for n in range(3)] == [[a * e for e in exe] for a in adds]


def _requests(rng, n_req, tight):
    exe = np.stack([rng.choice([0, 1, 2, 3, 5], size=n_req), rng.choice([0, 1, 2, 4], size=n_req),
                    rng.choice([0, 0, 0, 1], size=n_req)], axis=1).astype(np.int64)
    k = rng.integers(0, 40 if tight else 12, size=n_req).astype(np.int32)
    return exe, k


@pytest.mark.parametrize("n", [1, 3, 17, 64, 65, 130])
def test_three_restatements_agree(n):
    rng = np.random.default_rng(100 + n)
    for tight in (True, False):
        avail, _, X, _, _, _ = _random_problem(rng, n, 4, tight, "merged")
        X = X[X < n]
        exe, k = _requests(rng, 9, tight)
        for chained in (True, False):
            lit = ob.find_nodes(avail, exe, k, X, closed_form=False, chained=chained)
            clo = ob.find_nodes(avail, exe, k, X, closed_form=True, chained=chained)
            assert np.array_equal(lit.placed, clo.placed) and np.array_equal(lit.exec_nodes, clo.exec_nodes)
            assert np.array_equal(lit.adds, clo.adds) and np.array_equal(lit.avail_after, clo.avail_after)
        lit = ob.find_nodes(avail, exe, k, X, chained=True)
        table = {str(i): [int(v) for v in a] for i, a in enumerate(avail)}
        got = po.find_nodes_chain([(int(kk), [int(v) for v in e]) for kk, e in zip(k, exe)], table, [str(int(x)) for x in X])
        for q, (names, reserved) in enumerate(got):
            assert [int(s) for s in names] == lit.placement(q).tolist()
            for node in range(n):
                want = [int(lit.adds[q, node]) * int(e) for e in exe[q]]
                assert reserved.get(str(node), [0, 0, 0]) == want
        assert [table[str(i)] for i in range(n)] == lit.avail_after.tolist()


def _check_rebuild(placed, last, k, placement, adds_row, X):
    """The documented reconstruction of `reserved` from (placed, last_node, placements) — what the Go shim does."""
    want = np.zeros_like(adds_row)
    if last != NO:
        mult = np.bincount(placement, minlength=len(adds_row))
        for n in X:
            want[n] = mult[n] + 1
            if n == last:
                if placed == k:
                    want[n] = mult[n]
                break
    assert np.array_equal(want, adds_row)


@pytest.mark.gpu
def test_gpu_kats(gf_ctx):
    for name, avail, exe, k, want, adds in KATS:
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders([0, 1, 2], [0, 1, 2])
        for chained in (False, True):
            placed, last, off, nodes, got_adds = gf_ctx.find_nodes([exe], [k], chained=chained)
            assert nodes[: placed[0]].tolist() == want and got_adds[0].tolist() == adds, (name, chained)
        assert gf_ctx.residual().tolist() == [[a[j] - adds[n] * exe[j] for j in range(3)] for n, a in enumerate(avail)]


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 700, 3000])
def test_gpu_matches_oracle(gf_ctx, n, layout):
    rng = np.random.default_rng(31 + n + 5 * len(layout))
    for tight in (True, False):
        avail, D, X, _, _, _ = _random_problem(rng, n, 4, tight, layout)
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders(D, X)
        Xk = X[X < n]
        exe, k = _requests(rng, 33, tight)
        if n >= 700:
            k = (k.astype(np.int64) * 40).astype(np.int32)
        for chained in (False, True):
            want = ob.find_nodes(avail, exe, k, Xk, chained=chained)
            placed, last, off, nodes, adds = gf_ctx.find_nodes(exe, k, chained=chained)
            assert np.array_equal(placed, want.placed)
            assert np.array_equal(adds, want.adds)
            for q in range(len(k)):
                o = int(off[q])
                assert np.array_equal(nodes[o:o + int(placed[q])], want.placement(q)), (q, chained)
                _check_rebuild(int(placed[q]), int(last[q]), int(k[q]), want.placement(q), want.adds[q], Xk)
            if chained:
                assert np.array_equal(gf_ctx.residual(), want.avail_after)


@pytest.mark.gpu
def test_gpu_headline_size_chain(gf_ctx):
    """10 000 nodes, 200 stale applications reconciled in a row (closed-form oracle; literal on a prefix)."""
    from gangfit import workloads as wl

    w = wl.headline(10000, 200)
    s = w.snapshot
    gf_ctx.set_snapshot(s.avail)
    gf_ctx.set_orders(s.driver_order, s.exec_order)
    want = ob.find_nodes(s.avail, w.exe, w.k, s.exec_order, closed_form=True)
    lit = ob.find_nodes(s.avail, w.exe[:24], w.k[:24], s.exec_order, closed_form=False)
    placed, last, off, nodes, adds = gf_ctx.find_nodes(w.exe, w.k, chained=True)
    assert np.array_equal(placed, want.placed) and np.array_equal(adds, want.adds)
    for q in range(len(w.k)):
        o = int(off[q])
        assert np.array_equal(nodes[o:o + int(placed[q])], want.placement(q)), q
    assert np.array_equal(gf_ctx.residual(), want.avail_after)
    assert np.array_equal(lit.adds, want.adds[:24])
