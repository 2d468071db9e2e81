"""CPU ORACLE (test infrastructure) — numpy restatement of the snapshot construction that precedes the gang-fit decision:

    UsageForNodes                    LIB/resources/resources.go:31-43      (reservation replay)
    NodeSchedulingMetadataForNodes   LIB/resources/resources.go:61-100
    NodeSorter.PotentialNodes        internal/sort/nodesorting.go:41-122, 161-199

on the flat columns of include/gangfit.h's gf_snapshot_build.  Where the reference's unstable sorts leave ties open
(zones with equal free memory and cpu) the zone-id order decides, as documented at the C ABI.  Pinned indirectly: the C++
host mirror implements the same functions on string-keyed maps, is pinned by the reference's sort tests
(internal/sort/nodesorting_test.go), and host/tests/host_test.cpp checks the device path against it.
"""
import numpy as np

UNSCHEDULABLE, READY, DRIVER_CANDIDATE = 1, 2, 4
UNRANKED = 0xFFFFFFFF


def build(alloc, node_flags, name_rank, overhead=None, res_node=None, res_req=None, zone=None, n_zones=1,
          driver_label_rank=None, exec_label_rank=None):
    alloc = np.asarray(alloc, dtype=np.int64).reshape(-1, 3)
    n = len(alloc)
    over = np.zeros_like(alloc) if overhead is None else np.asarray(overhead, dtype=np.int64).reshape(-1, 3)
    usage = np.zeros_like(alloc)
    if res_node is not None and len(res_node):
        rn = np.asarray(res_node, dtype=np.int64)
        rr = np.asarray(res_req, dtype=np.int64).reshape(-1, 3)
        keep = rn < n
        np.add.at(usage, rn[keep], rr[keep])
    avail = alloc - (usage + over)
    sched = alloc - over
    z = np.zeros(n, dtype=np.int64) if zone is None else np.asarray(zone, dtype=np.int64)
    zmem = np.zeros(n_zones, dtype=np.int64)
    zcpu = np.zeros(n_zones, dtype=np.int64)
    np.add.at(zmem, z, avail[:, 1])
    np.add.at(zcpu, z, avail[:, 0])
    zorder = np.lexsort((np.arange(n_zones), zcpu, zmem))  # memory, then cpu, ties by zone id
    zrank = np.empty(n_zones, dtype=np.int64)
    zrank[zorder] = np.arange(n_zones)
    order = np.lexsort((np.asarray(name_rank, dtype=np.int64), avail[:, 0], avail[:, 1], zrank[z]))
    flags = np.asarray(node_flags, dtype=np.int64)
    D = [int(i) for i in order if flags[i] & DRIVER_CANDIDATE]
    X = [int(i) for i in order if not (flags[i] & UNSCHEDULABLE) and (flags[i] & READY)]
    if driver_label_rank is not None:
        D = sorted(D, key=lambda i: int(driver_label_rank[i]))  # Python's sort is stable
    if exec_label_rank is not None:
        X = sorted(X, key=lambda i: int(exec_label_rank[i]))
    return avail, sched, np.asarray(D, dtype=np.uint32), np.asarray(X, dtype=np.uint32)
