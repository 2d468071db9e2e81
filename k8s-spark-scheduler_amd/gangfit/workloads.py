"""Synthetic snapshots and pending-app tables for the BASELINE.json configs.

Deterministic: splitmix64 counter streams, seed = 0x5EED0000 + config number.  Pure numpy; no GPU, no oracle.
All quantities are canonical int64: cpu milli-cores, memory bytes, gpu devices.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

GIB = 1 << 30
MIB = 1 << 20
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n values of the splitmix64 sequence started at `seed` (+ a per-stream offset), vectorised."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        z = base + _GAMMA * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform01(seed: int, n: int, stream: int) -> np.ndarray:
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _choice(seed: int, n: int, stream: int, values) -> np.ndarray:
    idx = (splitmix64(seed, n, stream) % np.uint64(len(values))).astype(np.int64)
    return np.asarray(values, dtype=np.int64)[idx], idx


@dataclass
class Snapshot:
    avail: np.ndarray  # (N, 3) int64 AvailableResources
    sched: np.ndarray  # (N, 3) int64 SchedulableResources (= allocatable here: zero overhead)
    driver_order: np.ndarray  # uint32
    exec_order: np.ndarray  # uint32


@dataclass
class Workload:
    name: str
    snapshot: Snapshot
    drv: np.ndarray  # (A, 3)
    exe: np.ndarray  # (A, 3)
    k: np.ndarray  # (A,) int32 gang size (MinExecutorCount)
    k_max: np.ndarray  # (A,) int32 MaxExecutorCount (dynamic allocation), == k for static apps
    flags: np.ndarray  # (A,) uint32


def reference_node_order(avail: np.ndarray, zone: Optional[np.ndarray] = None) -> np.ndarray:
    """Priority order of getNodeNamesInPriorityOrder (internal/sort/nodesorting.go:74-122): AZ priority first (zones ranked
    by their summed free resources, memory then cpu, ascending — :98-104; ties keep the zone id order here, the reference's
    sort.Slice leaves them unspecified), then free memory ascending, then free cpu ascending, then name — the node index
    stands for the name.  One zone (zone=None): the order of the resources alone."""
    idx = np.arange(len(avail))
    if zone is None:
        return np.lexsort((idx, avail[:, 0], avail[:, 1])).astype(np.uint32)
    zone = np.asarray(zone).astype(np.int64)
    nz = int(zone.max()) + 1 if len(zone) else 0
    zsum = np.zeros((nz, 2), dtype=object)
    for z in range(nz):  # python ints: the sums of 10^5 byte counts stay exact
        sel = zone == z
        zsum[z, 0] = int(avail[sel, 1].astype(object).sum()) if sel.any() else 0
        zsum[z, 1] = int(avail[sel, 0].astype(object).sum()) if sel.any() else 0
    present = [z for z in range(nz) if (zone == z).any()]
    ranked = sorted(present, key=lambda z: (zsum[z, 0], zsum[z, 1], z))
    prio = np.zeros(nz, dtype=np.int64)
    for r, z in enumerate(ranked):
        prio[z] = r
    return np.lexsort((idx, avail[:, 0], avail[:, 1], prio[zone])).astype(np.uint32)


def make_snapshot(n_nodes: int, seed: int, used_lo: float = 0.0, used_hi: float = 0.9) -> Snapshot:
    """Cluster of SURVEY.md 8d/C2: paired (cpu, mem) shapes, 10 % gpu nodes, used fraction ~U[lo, hi] per dim
    quantised to 100 m / 256 MiB, 1 % of nodes overcommitted (negative availability)."""
    shapes_cpu = np.array([16, 32, 64, 96], dtype=np.int64) * 1000
    shapes_mem = np.array([64, 128, 256, 384], dtype=np.int64) * GIB
    _, sidx = _choice(seed, n_nodes, 1, [0, 1, 2, 3])
    alloc_cpu = shapes_cpu[sidx]
    alloc_mem = shapes_mem[sidx]
    alloc_gpu = np.where(_uniform01(seed, n_nodes, 2) < 0.10, 8, 0).astype(np.int64)
    span = used_hi - used_lo
    f_cpu = used_lo + span * _uniform01(seed, n_nodes, 3)
    f_mem = used_lo + span * _uniform01(seed, n_nodes, 4)
    f_gpu = used_lo + span * _uniform01(seed, n_nodes, 5)
    used_cpu = (np.floor(f_cpu * alloc_cpu / 100.0).astype(np.int64)) * 100
    used_mem = (np.floor(f_mem * alloc_mem / (256 * MIB)).astype(np.int64)) * (256 * MIB)
    used_gpu = np.floor(f_gpu * alloc_gpu).astype(np.int64)
    over = _uniform01(seed, n_nodes, 6) < 0.01
    used_cpu = np.where(over, alloc_cpu + 500, used_cpu)
    used_mem = np.where(over, alloc_mem + GIB, used_mem)
    sched = np.stack([alloc_cpu, alloc_mem, alloc_gpu], axis=1)
    avail = np.stack([alloc_cpu - used_cpu, alloc_mem - used_mem, alloc_gpu - used_gpu], axis=1)
    order = reference_node_order(avail)
    return Snapshot(np.ascontiguousarray(avail), np.ascontiguousarray(sched), order, order.copy())


def make_apps(n_apps: int, seed: int, k_cap: int = 512, dynamic: bool = False, skippable_frac: float = 0.0):
    """Pending Spark applications of SURVEY.md 8d/C2 (+ C4's min/max ranges when dynamic)."""
    drv_cpu, _ = _choice(seed, n_apps, 11, [1000, 2000, 4000])
    drv_mem, _ = _choice(seed, n_apps, 12, [2 * GIB, 4 * GIB, 8 * GIB])
    exe_cpu, _ = _choice(seed, n_apps, 13, [1000, 2000, 4000, 8000])
    exe_mem, _ = _choice(seed, n_apps, 14, [4 * GIB, 8 * GIB, 16 * GIB, 32 * GIB])
    exe_gpu = np.where(_uniform01(seed, n_apps, 15) < 0.05, 1, 0).astype(np.int64)
    # K ~ min(1 + Geometric(p = 1/12), k_cap): inverse-CDF sampling
    u = _uniform01(seed, n_apps, 16)
    geom = np.floor(np.log1p(-u) / np.log1p(-1.0 / 12.0)).astype(np.int64)
    k = np.minimum(1 + geom, k_cap).astype(np.int32)
    k_max = k.copy()
    if dynamic:
        k_max = (k + (splitmix64(seed, n_apps, 17) % np.uint64(65)).astype(np.int32)).astype(np.int32)
    flags = np.where(_uniform01(seed, n_apps, 18) < skippable_frac, 1, 0).astype(np.uint32)
    drv = np.stack([drv_cpu, drv_mem, np.zeros(n_apps, dtype=np.int64)], axis=1)
    exe = np.stack([exe_cpu, exe_mem, exe_gpu], axis=1)
    return drv, exe, k, k_max, flags


def config(number: int, n_nodes: Optional[int] = None, n_apps: Optional[int] = None) -> Workload:
    """BASELINE.json configs by number (1-based like SURVEY.md C1..C5); sizes can be overridden for tests."""
    seed = 0x5EED0000 + number
    if number == 1:  # 16 nodes, one app, K = 8 (CPU plumbing case)
        n = n_nodes or 16
        alloc = np.tile(np.array([32000, 128 * GIB, 0], dtype=np.int64), (n, 1))
        f = _uniform01(seed, n, 1) * 0.75
        used_cpu = (np.floor(f * 32000 / 250).astype(np.int64)) * 250
        used_mem = (np.floor(f * 128).astype(np.int64)) * GIB
        avail = alloc - np.stack([used_cpu, used_mem, np.zeros(n, dtype=np.int64)], axis=1)
        order = reference_node_order(avail)
        snap = Snapshot(avail, alloc, order, order.copy())
        drv = np.array([[1000, 4 * GIB, 0]], dtype=np.int64)
        exe = np.array([[2000, 8 * GIB, 0]], dtype=np.int64)
        k = np.array([8], dtype=np.int32)
        return Workload("C1 tightly-pack 16 nodes", snap, drv, exe, k, k.copy(), np.zeros(1, dtype=np.uint32))
    sizes = {2: (1000, 1000), 3: (10000, 10000), 4: (50000, 10000), 5: (100000, 1000)}
    if number not in sizes:
        raise ValueError(f"unknown config {number}")
    n, a = sizes[number]
    n, a = n_nodes or n, n_apps or a
    snap = make_snapshot(n, seed)
    drv, exe, k, k_max, flags = make_apps(a, seed, dynamic=(number == 4), skippable_frac=0.05 if number == 5 else 0.0)
    names = {2: "C2 batched first-fit", 3: "C3 tightly-pack vs distribute-evenly", 4: "C4 dynamic allocation",
             5: "C5 FIFO chain"}
    return Workload(f"{names[number]} {n} nodes x {a} apps", snap, drv, exe, k, k_max, flags)


def headline(n_nodes: int = 10000, n_apps: int = 1000, seed: int = 0x5EED0010, congested: bool = False) -> Workload:
    """The size BASELINE.json's metric is quoted on: 10k nodes x 1k pending apps (C2's distributions).
    congested=True draws node usage from U[0.95, 1.0]: about half of the gangs do not fit and need full scans of the
    executor order plus the driver-candidate fallback — the regime in which a 1k-deep pending queue actually occurs
    (and where the reference's retry loop costs O(|D| * N) per decision)."""
    snap = make_snapshot(n_nodes, seed, 0.95, 1.0) if congested else make_snapshot(n_nodes, seed)
    drv, exe, k, k_max, flags = make_apps(n_apps, seed)
    tag = "congested" if congested else "nominal"
    return Workload(f"headline {n_nodes} nodes x {n_apps} pending apps ({tag})", snap, drv, exe, k, k_max, flags)


def algorithmic_bytes(n_x: int, k: np.ndarray, dims: int = 3, pre_permuted: bool = True) -> int:
    """B(N_x, K, Dm) = N_x * (Dm*8 [+4]) + 72 + (16 + 4K) per decision, summed over the batch."""
    per_node = dims * 8 + (0 if pre_permuted else 4)
    k = np.asarray(k, dtype=np.int64)
    return int(len(k) * (n_x * per_node + 72 + 16) + 4 * int(k.sum()))
