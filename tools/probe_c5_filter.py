"""BASELINE config 5 (100 000 nodes, 270 k reservation entries resident as usage sums): ONE Filter = two gf_usage_apply calls
(one application's entries out, another's in) + gf_snapshot_build_resident(GF_RESIDENT_USAGE) + the 999 + 1 chain on the fresh
snapshot.  Phase split with a device synchronise behind every phase (so a phase's kernels are charged to it), next to the
unsplit Filter and to the chain alone on an unchanged snapshot.  Run on the MI355X box:
    python tools/probe_c5_filter.py [calls]            (prints one JSON line)
    rocprofv3 --kernel-trace --stats -- python tools/probe_c5_filter.py 30      (which kernels a Filter launches)
"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import gangfit
from gangfit import workloads as wl


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))]


def run(calls=100, ctx=None):
    own = ctx is None
    if own:
        ctx = gangfit.Context(0)
    w5 = wl.config(5)
    n5 = len(w5.snapshot.avail)
    rng = np.random.default_rng(5)
    ks = rng.integers(2, 26, size=20000)
    rnode = rng.integers(0, n5, size=int(ks.sum())).astype(np.uint32)
    rreq = np.stack([rng.choice([1000, 2000, 4000], size=len(rnode)), rng.choice([4, 8, 16], size=len(rnode)) * wl.GIB,
                     np.zeros(len(rnode), dtype=np.int64)], axis=1).astype(np.int64)
    flags5 = np.full(n5, 2 | 4, dtype=np.uint32)
    ranks5 = np.arange(n5, dtype=np.uint32)
    alloc5 = w5.snapshot.sched + 0
    q5 = gangfit.make_apps(w5.drv, w5.exe, w5.k, w5.flags)
    ctx.set_cluster(alloc5, flags5, ranks5)
    rcols5 = [np.ascontiguousarray(rreq[:, j]) for j in range(3)]
    ctx.usage_reset()
    ctx.usage_apply(rnode, res_cols=rcols5, sign=+1)
    starts5 = np.concatenate([[0], np.cumsum(ks)])
    FIFO, TIGHT = gangfit.GF_MODE_FIFO_CHAIN, gangfit.GF_ALGO_TIGHTLY_PACK
    sync = torch.cuda.synchronize
    rolled = [np.roll(q5, -i) for i in range(calls + 3)]

    def delta(i):
        j = i % len(ks)
        sl = slice(int(starts5[j]), int(starts5[j + 1]))
        return rnode[sl], [c[sl] for c in rcols5]

    unsplit, split = [], {"usage_apply_x2": [], "snapshot_build_resident": [], "chain_first_on_fresh_epoch": [], "total": []}
    for i in range(calls + 3):  # the unsplit Filter (what bench.py reports as filter_resident_usage)
        dn, dc = delta(i)
        t0 = time.perf_counter()
        ctx.usage_apply(dn, res_cols=dc, sign=-1)
        ctx.usage_apply(dn, res_cols=dc, sign=+1)
        ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
        ctx.fit_batch(FIFO, TIGHT, rolled[i])
        if i >= 3:
            unsplit.append((time.perf_counter() - t0) * 1e3)
    for i in range(calls + 3):  # the same Filter, a synchronise behind every phase
        dn, dc = delta(i)
        sync()
        t0 = time.perf_counter()
        ctx.usage_apply(dn, res_cols=dc, sign=-1)
        ctx.usage_apply(dn, res_cols=dc, sign=+1)
        sync()
        t1 = time.perf_counter()
        ctx.build_snapshot_resident(resident_usage=True, want_orders=False)
        sync()
        t2 = time.perf_counter()
        ctx.fit_batch(FIFO, TIGHT, rolled[i])
        t3 = time.perf_counter()
        if i >= 3:
            split["usage_apply_x2"].append((t1 - t0) * 1e3)
            split["snapshot_build_resident"].append((t2 - t1) * 1e3)
            split["chain_first_on_fresh_epoch"].append((t3 - t2) * 1e3)
            split["total"].append((t3 - t0) * 1e3)
    chain_only = []
    for i in range(min(calls, 60) + 2):  # the chain alone: the snapshot does not change, the queue does (every chain replays)
        t0 = time.perf_counter()
        ctx.fit_batch(FIFO, TIGHT, rolled[i])
        if i >= 2:
            chain_only.append((time.perf_counter() - t0) * 1e3)
    # ... and with the chain cache off: no checkpoint is dumped while the chain runs
    ctx.set_option("chain_cache", 0)
    nock = []
    for i in range(min(calls, 60) + 2):  # (the same rotations as the loop above: the chain's time depends on the head)
        t0 = time.perf_counter()
        ctx.fit_batch(FIFO, TIGHT, rolled[i])
        if i >= 2:
            nock.append((time.perf_counter() - t0) * 1e3)
    ctx.set_option("chain_cache", 1)
    out = {"nodes": n5, "calls": len(unsplit),
           "filter_resident_usage_p50_ms": pct(unsplit, 0.5), "filter_resident_usage_p99_ms": pct(unsplit, 0.99),
           "phases_p50_ms": {k: pct(v, 0.5) for k, v in split.items()}, "phases_p99_ms": {k: pct(v, 0.99) for k, v in split.items()},
           "chain_only_unchanged_snapshot_p50_ms": pct(chain_only, 0.5), "chain_only_p99_ms": pct(chain_only, 0.99),
           "chain_only_no_checkpoints_p50_ms": pct(nock, 0.5),
           "note": "phases: a device synchronise behind each, so `total` exceeds the unsplit Filter by the syncs; the first chain on "
                   "a fresh epoch re-derives the narrow units / rescales the working table and dumps dirty-chunk checkpoints"}
    if own:
        ctx.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 100)))
