"""CPU ORACLE, second opinion (test infrastructure, NOT product code).

A deliberately naive pure-Python restatement of the reference's gang bin-packing, written with the SAME data
structures as the Go code — dicts keyed by node-name strings, per-candidate reserved maps, add-then-compare
loops — so that it can be read side by side with the reference.  It is independent of oracle/gangfit_oracle.c
(different language, different data structures); tests require both to agree.  Only for small cases.

Citations are relative to /root/reference; LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg.
Quantities are canonical ints (cpu milli, memory bytes, gpu count) held in 3-tuples/lists.

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
