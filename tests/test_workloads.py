"""The synthetic workloads' priority order against the reference's own sort tests (internal/sort/nodesorting_test.go) and
against the device's snapshot construction."""
import numpy as np

from gangfit import workloads as wl


def _order_names(names, avail, zone=None):
    # reference_node_order breaks ties by node INDEX where the reference compares node NAMES: list the nodes by name
    idx = sorted(range(len(names)), key=lambda i: names[i])
    a = np.array([avail[i] for i in idx], dtype=np.int64)
    z = None if zone is None else np.array([zone[i] for i in idx])
    return [names[idx[i]] for i in wl.reference_node_order(a, z)]


def test_az_aware_node_sorting():
    """TestAZAwareNodeSorting (nodesorting_test.go:98-141): zone2 holds less than zone1, so its node comes first; inside zone1
    memory decides before cpu.  Columns are (cpu, memory, gpu)."""
    names = ["zone1Node1", "zone1Node2", "zone1Node3", "zone2Node1"]
    avail = [(1, 1, 0), (1, 2, 0), (2, 1, 0), (1, 1, 0)]
    zone = [0, 0, 0, 1]
    assert _order_names(names, avail, zone) == ["zone2Node1", "zone1Node1", "zone1Node3", "zone1Node2"]


def test_node_sorting_without_zone_label():
    """TestAZAwareNodeSortingWorksIfZoneLabelIsMissing (:143-182): one (default) zone."""
    names = ["node1", "node2", "node3"]
    avail = [(2, 1, 0), (2, 2, 0), (1, 1, 0)]
    assert _order_names(names, avail) == ["node3", "node1", "node2"]
    assert _order_names(names, avail, [0, 0, 0]) == ["node3", "node1", "node2"]


def test_az_major_order_is_contiguous_per_zone():
    w = wl.headline(3000, 10)
    zone = (wl.splitmix64(0xA3, 3000, 9) % np.uint64(3)).astype(np.uint32)
    order = wl.reference_node_order(w.snapshot.avail, zone)
    assert sorted(order.tolist()) == list(range(3000))
    z = zone[order]
    assert (np.diff(z) != 0).sum() == 2  # three contiguous ranges
    for zz in range(3):  # inside a zone: the single-zone rule
        sel = order[z == zz]
        sub = w.snapshot.avail[sel]
        keys = list(zip(sub[:, 1].tolist(), sub[:, 0].tolist(), sel.tolist()))
        assert keys == sorted(keys)
    # zones in ascending order of their summed free memory (then cpu)
    sums = [(int(w.snapshot.avail[zone == zz, 1].sum()), int(w.snapshot.avail[zone == zz, 0].sum())) for zz in z[np.r_[True, np.diff(z) != 0]]]
    assert sums == sorted(sums)
