"""The sparse gpu view of the independent batch (gangfit::SparseTable, csrc/gangfit_kernels.hip::wave_decide): when the executor
candidates with a free gpu are a minority of the order, gangs whose executors need a gpu are packed from a compact table of
those nodes instead of the full order.  Every node left out has capacity 0 for such a request, so the placements must not
change.  Clusters here have 3-20 % gpu nodes; requests cover: driver on a gpu node that also hosts executors, driver on a node
outside the view, gangs that only fit with another driver candidate (the decision is redone on the full order), gangs that do
not fit at all, distribute-evenly over several passes, K = 0."""
import numpy as np
import pytest

import gangfit
from oracle import binding as ob
from test_gpu_parity import _assert_same

pytestmark = pytest.mark.gpu
IND = gangfit.GF_MODE_INDEPENDENT


def _problem(rng, n, a, gpu_frac, tight, layout):
    hi = 30 if tight else 400
    avail = rng.integers(-2, hi, size=(n, 3)).astype(np.int64)
    avail[:, 2] = np.where(rng.random(n) < gpu_frac, rng.integers(1, 9, size=n), rng.integers(-1, 1, size=n))
    base = rng.permutation(n)
    if layout == "identical":
        X = base.astype(np.uint32)
        D = X.copy()
    else:
        X = base[rng.random(n) < 0.85].astype(np.uint32)
        D = base[rng.random(n) < 0.6].astype(np.uint32)
        if len(X) == 0:
            X = base[:1].astype(np.uint32)
        if len(D) == 0:
            D = base[-1:].astype(np.uint32)
    drv = rng.integers(0, 9, size=(a, 3)).astype(np.int64)
    drv[:, 2] = (rng.random(a) < 0.3) * rng.integers(0, 3, size=a)
    exe = rng.integers(0, 6, size=(a, 3)).astype(np.int64)
    exe[:, 2] = np.where(rng.random(a) < 0.7, rng.integers(1, 4, size=a), 0)  # most executors need a gpu
    k = rng.integers(0, 60, size=a).astype(np.int32)
    k[rng.random(a) < 0.1] = 0
    return avail, D, X, drv, exe, k


@pytest.mark.parametrize("algo", [0, 1, 2])
@pytest.mark.parametrize("layout", ["merged", "identical"])
@pytest.mark.parametrize("n,gpu_frac", [(40, 0.2), (64, 0.1), (300, 0.2), (1000, 0.05), (5000, 0.03), (5000, 0.2)])
def test_gpu_requests_through_the_sparse_view(gf_ctx, algo, layout, n, gpu_frac):
    rng = np.random.default_rng(17 * n + int(100 * gpu_frac) + algo + len(layout))
    for tight in (True, False):
        avail, D, X, drv, exe, k = _problem(rng, n, 200, gpu_frac, tight, layout)
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders(D, X)
        apps = gangfit.make_apps(drv, exe, k)
        ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=True)
        _assert_same(gf_ctx.fit_batch(IND, algo, apps), ref, apps)
        if n <= 300:
            lit = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, X, closed_form=False)
            assert np.array_equal(lit.results, ref.results)


def test_driver_conflicts_on_gpu_nodes(gf_ctx):
    """The first fitting driver candidate is the only gpu node with room, so reserving the driver there starves the gang:
    SparkBinPack moves on to the next candidate (binpack.go:67-85) — the sparse path must hand such gangs to the full order."""
    n = 200
    avail = np.tile(np.array([[8000, 64, 0]], dtype=np.int64), (n, 1))
    avail[5] = [8000, 64, 4]      # the one gpu node, early in the order
    avail[150] = [9000, 64, 0]
    order = np.arange(n, dtype=np.uint32)
    gf_ctx.set_snapshot(avail)
    gf_ctx.set_orders(order[::-1].copy()[:n] if False else np.array([5, 150] + [i for i in range(n) if i not in (5, 150)], dtype=np.uint32),
                      order)
    D = np.array([5, 150] + [i for i in range(n) if i not in (5, 150)], dtype=np.uint32)
    # driver takes 6000 m cpu: on node 5 that leaves room for one executor only; K = 3 needs the driver elsewhere
    drv = np.array([[6000, 1, 0], [1000, 1, 0], [6000, 1, 0], [1000, 1, 1]], dtype=np.int64)
    exe = np.array([[2000, 1, 1], [2000, 1, 1], [2000, 1, 1], [1000, 1, 1]], dtype=np.int64)
    k = np.array([3, 3, 5, 3], dtype=np.int32)
    apps = gangfit.make_apps(drv, exe, k)
    for algo in (0, 1):
        ref = ob.fit_independent(algo, avail, ob.make_apps(drv, exe, k), D, order, closed_form=False)
        out = gf_ctx.fit_batch(IND, algo, apps)
        _assert_same(out, ref, apps)
    assert ref.results["has_capacity"].tolist() == [1, 1, 0, 1]
    assert int(ref.results["driver_node"][0]) == 150  # moved off the gpu node
