"""Placing single executors (rescheduleExecutor, internal/extender/resource.go:594-703): oracle pinned by the reference's
TestMinimalFragmentation / TestMinimalFragmentationEdgeCase (resource_test.go:73-170), GPU parity through
gf_executor_fit."""
import numpy as np
import pytest

from oracle import binding as ob
from test_gpu_parity import _random_problem

GIB = 1 << 30
NO = 0xFFFFFFFF


def test_reference_minimal_fragmentation_edge_case():
    """resource_test.go:127-170: a 4-byte-memory driver on node1 (1 cpu) and a 4-cpu driver on node2 (1 byte); the extra
    executor (3 cpu, 1 byte) must go to node2 — it sorts second (more free memory) but has the smaller capacity."""
    avail = [[8000 - 1000, 8 * GIB - 4, 0], [8000 - 4000, 8 * GIB - 1, 0]]
    order = [0, 1]  # free memory ascending: node1 first
    exe = [3000, 1, 0]
    assert ob.node_capacity(avail[0], [0, 0, 0], exe) == 2 and ob.node_capacity(avail[1], [0, 0, 0], exe) == 1
    assert ob.executor_fit(avail, exe, order, minimal_fragmentation=True) == 1
    assert ob.executor_fit(avail, exe, order) == 0  # the plain first-fit loop would have said node1


def test_reference_minimal_fragmentation_attracts_to_hosting_node():
    """resource_test.go:73-125: node1 carries the 3 static pods, node2 the dynamic driver and exec-1; exec-2 must follow
    exec-1 to node2 although node1 comes first in the order."""
    avail = [[8000 - 3000, 8 * GIB - 3, 0], [8000 - 2000, 8 * GIB - 2, 0]]
    assert ob.executor_fit(avail, [1000, 1, 0], [0, 1], minimal_fragmentation=True, hosts=[0, 1]) == 1
    # without the hint the smaller capacity wins (node1: 5 < node2: 6)
    assert ob.executor_fit(avail, [1000, 1, 0], [0, 1], minimal_fragmentation=True) == 0


# TestDynamicAllocationScheduling (resource_test.go:172-372, single-az-tightly-pack): where the soft reservation of an
# executor above the minimum lands.  Restated at the first-fit boundary; node shape 8 cpu / 8 GiB / 1 gpu, pods 1 cpu / 1 B
# (driver + 1 gpu).  (avail per node, executor order = free memory ascending then name, expected node)
T2_CASES = [
    # :197-208 driver + executor-0 on node1 -> node1 sorts first and still fits: "node1"
    ("soft reservation over min executor count", [[6000, 8 * GIB - 2, 0], [8000, 8 * GIB, 1]], [0, 1], 0),
    # :209-225 driver and executor-0 went to node2 -> "soft reservations are created on full nodes first": "node2"
    ("full nodes first", [[8000, 8 * GIB, 1], [6000, 8 * GIB - 2, 0]], [1, 0], 1),
    # :226-243 two soft reservations in a row stay on node1
    ("second extra executor", [[5000, 8 * GIB - 3, 0], [8000, 8 * GIB, 1]], [0, 1], 0),
    # :262-292 the application lives in zone2 (node2): only zone2's nodes are candidates although node1 has less free memory
    ("same AZ as the application", [[6000, 8 * GIB - 2, 0], [7000, 8 * GIB - 1, 0]], [1], 1),
]


@pytest.mark.parametrize("name,avail,order,want", T2_CASES, ids=[c[0] for c in T2_CASES])
def test_reference_dynamic_allocation_soft_reservation_nodes(name, avail, order, want):
    assert ob.executor_fit(avail, [1000, 1, 0], order) == want


@pytest.mark.gpu
def test_gpu_reference_dynamic_allocation_cases(gf_ctx):
    for name, avail, order, want in T2_CASES:
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders([0], order)
        assert gf_ctx.executor_fit([[1000, 1, 0]]).tolist() == [want], name


# resource_test.go:262-292 WITHOUT the hand-filtered order of T2_CASES[3]: both nodes stay in the executor order (node1 first:
# less free memory), the zone step — on the device here, in host/extender.cpp for the whole mirror (host_test
# TestDynamicAllocationSameAZ) — keeps the extra executors in zone2.
T2_ZONE = dict(avail=[[6000, 8 * GIB - 2, 0], [7000, 8 * GIB - 1, 0]], order=[0, 1], node_zone=[0, 1], app_zone=1)


def test_reference_same_az_case_is_the_filtered_order_for_the_oracle():
    """filterNodesToZone (resource.go:462-478) happens before the sort: at the first-fit boundary the oracle sees the order
    restricted to the zone.  Without the filter the reference's loop would answer node1."""
    c = T2_ZONE
    in_zone = [n for n in c["order"] if c["node_zone"][n] == c["app_zone"]]
    assert ob.executor_fit(c["avail"], [1000, 1, 0], in_zone) == 1
    assert ob.executor_fit(c["avail"], [1000, 1, 0], c["order"]) == 0


@pytest.mark.gpu
def test_gpu_reference_same_az_case_zone_step_on_the_device(gf_ctx):
    c = T2_ZONE
    gf_ctx.set_snapshot(c["avail"])
    gf_ctx.set_orders([0], c["order"])
    exe = [[1000, 1, 0]] * 3
    got = gf_ctx.executor_fit(exe, node_zone=c["node_zone"], req_zone=[c["app_zone"], 0, 0xFFFFFFFF])
    assert got.tolist() == [1, 0, 0]  # zone2 -> node2 (the reference's expectation); zone1 -> node1; anywhere -> node1
    assert gf_ctx.executor_fit(exe[:1], node_zone=c["node_zone"], req_zone=[7]).tolist() == [NO]  # an empty zone: failure-fit


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 65, 700, 3000])
def test_gpu_zone_step_matches_the_oracle_on_the_filtered_order(gf_ctx, n):
    """gf_executor_fit_zoned against the oracle on the order restricted to the request's zone (both variants, with the
    reserved map and the hosting hint): what filterNodesToZone leaves of a sorted order."""
    rng = np.random.default_rng(77 + n)
    avail, D, X, drv, exe, k = _random_problem(rng, n, 60, True, "merged")
    gf_ctx.set_snapshot(avail)
    gf_ctx.set_orders(D, X)
    node_zone = rng.integers(0, 4, size=n).astype(np.uint32)
    req_zone = rng.integers(0, 5, size=len(exe)).astype(np.uint32)  # zone 4 has no node
    req_zone[rng.random(len(exe)) < 0.2] = 0xFFFFFFFF
    reserved = rng.integers(0, 4, size=(n, 3)).astype(np.int64) * (rng.random((n, 1)) < 0.3)
    hosts = rng.random((len(exe), n)) < 0.05
    for res in (None, reserved):
        def order_of(q):
            return X if req_zone[q] == 0xFFFFFFFF else [x for x in X if x < n and node_zone[x] == req_zone[q]]  # (X holds an unknown name)
        got = gf_ctx.executor_fit(exe, reserved=res, node_zone=node_zone, req_zone=req_zone)
        assert got.tolist() == [ob.executor_fit(avail, e, order_of(q), reserved=res) for q, e in enumerate(exe)]
        got = gf_ctx.executor_fit(exe, reserved=res, minimal_fragmentation=True, hosts=hosts, node_zone=node_zone, req_zone=req_zone)
        assert got.tolist() == [ob.executor_fit(avail, e, order_of(q), reserved=res, minimal_fragmentation=True, hosts=hosts[q])
                                for q, e in enumerate(exe)]


def test_oracle_edge_cases():
    avail = [[1, 1, 0], [5, 5, 0], [9, 9, 1]]
    assert ob.executor_fit(avail, [2, 2, 0], [0, 1, 2]) == 1
    assert ob.executor_fit(avail, [2, 2, 0], [0, 1, 2], reserved=[[0, 0, 0], [4, 0, 0], [0, 0, 0]]) == 2  # quirk 5
    assert ob.executor_fit(avail, [20, 2, 0], [0, 1, 2]) == NO
    assert ob.executor_fit(avail, [2, 2, 0], [0, 1, 2], minimal_fragmentation=True) == 1  # caps 0, 2, 4
    assert ob.executor_fit(avail, [2, 2, 0], [2, 1, 0], minimal_fragmentation=True, hosts=[1, 0, 1]) == 2
    assert ob.executor_fit(avail, [0, 0, 0], [0, 1, 2], minimal_fragmentation=True) == 0  # all math.MaxInt: first stays
    assert ob.executor_fit(avail, [2, 2, 2], [0, 1, 2], minimal_fragmentation=True) == NO


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["general", "merged", "identical"])
@pytest.mark.parametrize("n", [1, 64, 65, 700, 3000])
def test_gpu_matches_oracle(gf_ctx, n, layout):
    rng = np.random.default_rng(8 + n + 3 * len(layout))
    for tight in (True, False):
        avail, D, X, drv, exe, k = _random_problem(rng, n, 120, tight, layout)
        gf_ctx.set_snapshot(avail)
        gf_ctx.set_orders(D, X)
        reserved = rng.integers(0, 4, size=(n, 3)).astype(np.int64) * (rng.random((n, 1)) < 0.3)
        hosts = rng.random((len(exe), n)) < 0.05
        for res in (None, reserved):
            got = gf_ctx.executor_fit(exe, reserved=res)
            want = [ob.executor_fit(avail, e, X, reserved=res) for e in exe]
            assert got.tolist() == want
            for h in (None, hosts):
                got = gf_ctx.executor_fit(exe, reserved=res, minimal_fragmentation=True, hosts=h)
                want = [ob.executor_fit(avail, e, X, reserved=res, minimal_fragmentation=True,
                                        hosts=None if h is None else h[i]) for i, e in enumerate(exe)]
                assert got.tolist() == want


@pytest.mark.gpu
def test_gpu_reference_pinned_cases(gf_ctx):
    gf_ctx.set_snapshot([[7000, 8 * GIB - 4, 0], [4000, 8 * GIB - 1, 0]])
    gf_ctx.set_orders([0, 1], [0, 1])
    assert gf_ctx.executor_fit([[3000, 1, 0]], minimal_fragmentation=True).tolist() == [1]
    assert gf_ctx.executor_fit([[3000, 1, 0]]).tolist() == [0]
    gf_ctx.set_snapshot([[5000, 8 * GIB - 3, 0], [6000, 8 * GIB - 2, 0]])
    gf_ctx.set_orders([0, 1], [0, 1])
    assert gf_ctx.executor_fit([[1000, 1, 0]], minimal_fragmentation=True, hosts=[[False, True]]).tolist() == [1]
    assert gf_ctx.executor_fit([[9000, 1, 0]], minimal_fragmentation=True).tolist() == [NO]
