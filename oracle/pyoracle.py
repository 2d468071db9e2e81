"""CPU ORACLE, second opinion (test infrastructure, NOT product code).

A deliberately naive pure-Python restatement of the reference's gang bin-packing, written with the SAME data
structures as the Go code — dicts keyed by node-name strings, per-candidate reserved maps, add-then-compare
loops — so that it can be read side by side with the reference.  It is independent of oracle/gangfit_oracle.c
(different language, different data structures); tests require both to agree.  Only for small cases.

Citations are relative to /root/reference; LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg.
Quantities are canonical ints (cpu milli, memory bytes, gpu count) held in 3-tuples/lists.

minimalFragmentation (LIB/binpack/minimal_fragmentation.go, LIB/capacity/capacity.go) is restated here with the reference's own
shape too — a stably sorted list of (node, capacity) pairs, sort.Search, slices — so that the C oracle's index arithmetic and the
device's histogram form are both held to something that reads like the Go source.

Parity status: DistributeEvenly and FIFO replay are PARITY UNPINNED (no reference test selects them, no Go
toolchain in this image); TightlyPack feasibility is pinned by the reference tests listed in
tests/test_oracle_reference_kats.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

Res = List[int]  # [cpu_milli, mem_bytes, gpu]


def greater_than(a: Sequence[int], b: Sequence[int]) -> bool:
    """Resources.GreaterThan — LIB/resources/resources.go:239-241 (ANY component greater)."""
    return a[0] > b[0] or a[1] > b[1] or a[2] > b[2]


def _add(a: Res, b: Sequence[int]) -> None:
    for j in range(3):
        a[j] += b[j]


def _sub(a: Res, b: Sequence[int]) -> None:
    for j in range(3):
        a[j] -= b[j]


@dataclass
class PackingResult:
    """LIB/binpack/binpack.go:25-30."""

    driver_node: str = ""
    executor_nodes: List[str] = field(default_factory=list)
    has_capacity: bool = False
    reserved: Dict[str, Res] = field(default_factory=dict)  # the `reserved` map efficiencies are computed from


Packer = Callable[[Sequence[int], int, Sequence[str], Dict[str, Res], Dict[str, Res]], Tuple[Optional[List[str]], bool]]


def tightly_pack_executors(exe, count, order, avail, reserved):
    """tightlyPackExecutors — LIB/binpack/pack_tightly.go:34-63."""
    nodes: List[str] = []
    if count == 0:
        return nodes, True
    for n in order:
        if n not in reserved:
            reserved[n] = [0, 0, 0]
        while True:
            _add(reserved[n], exe)
            if n not in avail or greater_than(reserved[n], avail[n]):
                _sub(reserved[n], exe)
                break
            nodes.append(n)
            if len(nodes) == count:
                return nodes, True
    return None, False


def distribute_executors_evenly(exe, count, order, avail, reserved):
    """distributeExecutorsEvenly — LIB/binpack/distribute_evenly.go:34-73."""
    available = {name: True for name in order}
    nodes: List[str] = []
    if count == 0:
        return nodes, True
    while len(available) > 0:
        for n in order:
            if n not in available:
                continue
            if n not in reserved:
                reserved[n] = [0, 0, 0]
            _add(reserved[n], exe)
            if n not in avail or greater_than(reserved[n], avail[n]):
                del available[n]
                _sub(reserved[n], exe)
            else:
                nodes.append(n)
                if len(nodes) == count:
                    return nodes, True
    return None, False


MAX_INT = (1 << 63) - 1  # math.MaxInt on a 64-bit Go build


def _go_int(v: int) -> int:
    """Go int arithmetic wraps (two's complement, 64 bits)."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def capacity_against_single_dimension(available: int, reserved: int, required: int) -> int:
    """getCapacityAgainstSingleDimension — LIB/capacity/capacity.go:36-56."""
    available, reserved, required = int(available), int(reserved), int(required)  # (callers may hand over numpy scalars)
    if reserved > available:
        return 0
    if required == 0:
        return MAX_INT
    return (available - reserved) // required  # inf.RoundFloor of a non-negative quotient


def get_node_capacity(available: Sequence[int], reserved: Sequence[int], single_executor: Sequence[int]) -> int:
    """GetNodeCapacity — LIB/capacity/capacity.go:59-75."""
    return min(capacity_against_single_dimension(available[j], reserved[j], single_executor[j]) for j in range(3))


def get_node_capacities(order, avail, reserved, single_executor):
    """GetNodeCapacities + FilterOutNodesWithoutCapacity — LIB/capacity/capacity.go:78-113: [(name, capacity)], capacity > 0,
    in nodePriorityOrder; names outside the metadata are left out."""
    out = []
    for n in order:
        if n in avail:
            c = get_node_capacity(avail[n], reserved.get(n, [0, 0, 0]), single_executor)
            if c > 0:
                out.append((n, c))
    return out


def _search(n: int, pred) -> int:
    """sort.Search: the smallest index in [0, n) for which pred is true (n when there is none)."""
    lo, hi = 0, n
    while lo < hi:
        mid = (lo + hi) // 2
        if pred(mid):
            hi = mid
        else:
            lo = mid + 1
    return lo


def internal_minimal_fragmentation(count: int, caps):
    """internalMinimalFragmentation — LIB/binpack/minimal_fragmentation.go:93-137; caps sorted by capacity (stable)."""
    caps = list(caps)
    nodes: List[str] = []
    while len(caps) > 0:
        position = _search(len(caps), lambda i: caps[i][1] >= count)
        if position != len(caps):
            return nodes + [caps[position][0]] * count, True
        max_capacity = caps[-1][1]
        first_max = _search(len(caps), lambda i: caps[i][1] >= max_capacity)
        cur = first_max
        while count >= max_capacity and cur < len(caps):
            nodes += [caps[cur][0]] * max_capacity
            count -= max_capacity
            cur += 1
        if count == 0:
            return nodes, True
        caps = caps[:first_max] + caps[cur:]
    return None, False


def minimal_fragmentation(exe, count, order, avail, reserved):
    """minimalFragmentation — LIB/binpack/minimal_fragmentation.go:59-91 (never writes into `reserved`)."""
    count = int(count)
    if count == 0:
        return [], True
    caps = get_node_capacities(order, avail, reserved, exe)
    if len(caps) == 0:
        return None, False
    caps.sort(key=lambda nc: nc[1])  # sort.SliceStable: Python's sort is stable
    max_capacity = caps[-1][1]
    if count < max_capacity:
        total = _go_int(count + max_capacity)  # (executorCount + maxCapacity) wraps when maxCapacity is math.MaxInt
        target = total // 2 if total >= 0 else -((-total) // 2)  # Go's integer division truncates towards zero
        first = _search(len(caps), lambda i: caps[i][1] >= target)
        nodes, ok = internal_minimal_fragmentation(count, caps[:first])
        if ok:
            return nodes, ok
    return internal_minimal_fragmentation(count, caps)


def spark_binpack(drv, exe, count, driver_order, exec_order, avail, packer: Packer) -> PackingResult:
    """SparkBinPack — LIB/binpack/binpack.go:60-87."""
    for d in driver_order:
        if d not in avail or greater_than(drv, avail[d]):
            continue
        reserved: Dict[str, Res] = {d: list(drv)}
        nodes, ok = packer(exe, count, exec_order, avail, reserved)
        if ok:
            return PackingResult(d, list(nodes), True, reserved)
    return PackingResult()


def tightly_pack(drv, exe, count, driver_order, exec_order, avail) -> PackingResult:
    return spark_binpack(drv, exe, count, driver_order, exec_order, avail, tightly_pack_executors)


def distribute_evenly(drv, exe, count, driver_order, exec_order, avail) -> PackingResult:
    return spark_binpack(drv, exe, count, driver_order, exec_order, avail, distribute_executors_evenly)


def _value(v: int, unit: int) -> int:
    """Quantity.Value(): rounded away from zero to whole units (K8S apimachinery quantity.go:731-734, math.go:169-199)."""
    if unit == 1:
        return v
    q, r = divmod(abs(v), unit)
    q += 1 if r else 0
    return q if v >= 0 else -q


_UNITS = (1000, 1, 1)  # canonical cpu is milli-cores; memory bytes and gpu devices are whole units already


def compute_packing_efficiency(name, avail, sched, reserved):
    """computePackingEfficiency — LIB/binpack/efficiency.go:79-103. Returns (cpu, memory, gpu) as float64."""
    eff = []
    for j in range(3):
        used = sched[name][j] - avail[name][j] + (reserved[name][j] if name in reserved else 0)
        denom = _value(sched[name][j], _UNITS[j]) or 1  # normalizeResource, :105-110
        eff.append(float(_value(used, _UNITS[j])) / float(denom))
    if _value(sched[name][2], 1) == 0:
        eff[2] = 0.0
    return tuple(eff)


def compute_avg_packing_efficiency(sched, effs):
    """ComputeAvgPackingEfficiency — efficiency.go:114-156. effs: list of (name, (cpu, mem, gpu)) in slice order."""
    if not effs:
        return (0.0, 0.0, 0.0, 0.0)
    cpu = mem = gpu = mx = 0.0
    with_gpu = 0
    for name, e in effs:
        cpu += e[0]
        mem += e[1]
        if _value(sched[name][2], 1) != 0:
            gpu += e[2]
            with_gpu += 1
        mx += max(e[2], max(e[0], e[1]))
    length = float(max(len(effs), 1))
    return (cpu / length, mem / length, 1.0 if with_gpu == 0 else gpu / with_gpu, mx / length)


def group_nodes_by_zone(names, zone):
    """groupNodesByZone — LIB/binpack/single_az.go:57-72 (zone: dict name -> label for metadata keys only)."""
    zones_in_order: List[str] = []
    by_zone: Dict[str, List[str]] = {}
    for n in names:
        if n not in zone:
            continue
        z = zone[n]
        if z not in by_zone:
            zones_in_order.append(z)
            by_zone[z] = []
        by_zone[z].append(n)
    return zones_in_order, by_zone


def single_az(packer: Packer):
    """getSingleAZSparkBinFunction + chooseBestResult — LIB/binpack/single_az.go:23-55, 75-97."""

    def fn(drv, exe, count, driver_order, exec_order, avail, sched, zone) -> PackingResult:
        zones, d_by_zone = group_nodes_by_zone(driver_order, zone)
        _, x_by_zone = group_nodes_by_zone(exec_order, zone)
        results = []
        for z in zones:
            if z not in x_by_zone:
                continue
            r = spark_binpack(drv, exe, count, d_by_zone[z], x_by_zone[z], avail, packer)
            if r.has_capacity:
                results.append(r)
        best, best_max = PackingResult(), 0.0
        for r in results:
            names = [r.driver_node] + r.executor_nodes
            avg = compute_avg_packing_efficiency(
                sched, [(n, compute_packing_efficiency(n, avail, sched, r.reserved)) for n in names])
            if best_max < avg[3]:
                best, best_max = r, avg[3]
        return best

    return fn


single_az_tightly_pack = single_az(tightly_pack_executors)
single_az_minimal_fragmentation = single_az(minimal_fragmentation)  # LIB/binpack/single_az_minimal_fragmentation.go:20


def minimal_fragmentation_pack(drv, exe, count, driver_order, exec_order, avail) -> PackingResult:
    """MinimalFragmentation — LIB/binpack/minimal_fragmentation.go:27-34."""
    return spark_binpack(drv, exe, count, driver_order, exec_order, avail, minimal_fragmentation)


def az_aware_tightly_pack(drv, exe, count, driver_order, exec_order, avail, sched, zone) -> PackingResult:
    """AzAwareTightlyPack — LIB/binpack/az_aware_pack_tightly.go:27-38."""
    r = single_az_tightly_pack(drv, exe, count, driver_order, exec_order, avail, sched, zone)
    if r.has_capacity:
        return r
    return tightly_pack(drv, exe, count, driver_order, exec_order, avail)


BINPACK_FUNCTIONS = {  # internal/binpacker/binpack.go:43-49 (the two north-star entries)
    "tightly-pack": tightly_pack,
    "distribute-evenly": distribute_evenly,
}


def select_binpacker(name: str):
    """SelectBinpacker — internal/binpacker/binpack.go:52-58: unknown names fall back to distribute-evenly."""
    return BINPACK_FUNCTIONS.get(name, distribute_evenly)


def spark_resource_usage(drv, exe, driver_node, executor_nodes) -> Dict[str, Sequence[int]]:
    """sparkResourceUsage — internal/extender/sparkpods.go:139-146 (map overwrite quirk)."""
    res: Dict[str, Sequence[int]] = {driver_node: drv}
    for n in executor_nodes:
        res[n] = exe
    return res


def subtract_usage_if_exists(avail: Dict[str, Res], usage) -> None:
    """SubtractUsageIfExists — LIB/resources/resources.go:129-135."""
    for name, used in usage.items():
        if name in avail:
            _sub(avail[name], used)


def fit_earlier_drivers_then_pack(binpack, apps, driver_order, exec_order, avail):
    """fitEarlierDrivers + final pack — internal/extender/resource.go:224-262, 309-328.

    apps: list of (drv, exe, count, skippable); apps[:-1] are earlier drivers, apps[-1] the current one.
    Mutates avail.  Returns (results, failed_at) where failed_at is the index of the earlier driver that caused
    "failure-earlier-driver", or -1.
    """
    results: List[Optional[PackingResult]] = [None] * len(apps)
    for i, (drv, exe, count, skippable) in enumerate(apps[:-1]):
        res = binpack(drv, exe, count, driver_order, exec_order, avail)
        results[i] = res
        if not res.has_capacity:
            if skippable:
                continue
            return results, i
        subtract_usage_if_exists(avail, spark_resource_usage(drv, exe, res.driver_node, res.executor_nodes))
    drv, exe, count, _ = apps[-1]
    results[-1] = binpack(drv, exe, count, driver_order, exec_order, avail)
    return results, -1


def find_nodes(executor_count: int, exe: Sequence[int], available: Dict[str, Res], ordered_nodes: Sequence[str]):
    """findNodes — internal/extender/failover.go:412-436, with the reference's dict-of-lists shapes.  Returns
    (executorNodeNames, reserved); the add that fails the comparison stays in `reserved` (no Sub before the break)."""
    names: List[str] = []
    reserved: Dict[str, Res] = {}
    for n in ordered_nodes:
        if n not in reserved:
            reserved[n] = [0, 0, 0]
        while True:
            _add(reserved[n], exe)
            if greater_than(reserved[n], available[n]):
                break
            names.append(n)
            if len(names) == executor_count:
                return names, reserved
    return names, reserved


def find_nodes_chain(requests, available: Dict[str, Res], ordered_nodes: Sequence[str]):
    """The reconciler's loop over stale applications of one instance group (failover.go:132-160): findNodes, then
    availableResources.Sub(reservedResources) (:159; NodeGroupResources.Sub, LIB/resources/resources.go:119-126).
    requests: [(executor_count, exe)].  Mutates `available`; returns [(names, reserved)]."""
    out = []
    for count, exe in requests:
        names, reserved = find_nodes(count, exe, available, ordered_nodes) if count > 0 else ([], {})
        for n, r in reserved.items():
            if n not in available:
                available[n] = [0, 0, 0]
            _sub(available[n], r)
        out.append((names, reserved))
    return out
