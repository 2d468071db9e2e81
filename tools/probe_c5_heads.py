"""BASELINE config 5 (100 000 nodes, 20 000 reservations replayed, FIFO 999 + 1): why is the chain's p99 1.6 x its p50?
For each rotated head of the bench's config-5 loop: the chain's latency (host entry, checkpoints on as shipped, and off), where it
stopped, what it visited (in-kernel counters of the instrumented variant: slots, cycles by phase) and where its placements
landed in the priority order (inside / beyond the LDS front of the solo kernel).  The heads are then split into the fast and
the slow cluster and every quantity is averaged per cluster.  Run on the MI355X box; prints a table, writes JSON.
    python tools/probe_c5_heads.py [n_heads] [out.json]"""
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
import gangfit  # noqa: E402
from gangfit import workloads as wl  # noqa: E402

n_heads = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out_path = sys.argv[2] if len(sys.argv) > 2 else None
FIFO, TIGHT = gangfit.GF_MODE_FIFO_CHAIN, gangfit.GF_ALGO_TIGHTLY_PACK

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
rcols5 = [np.ascontiguousarray(rreq[:, j]) for j in range(3)]

ctx = gangfit.Context(0)
ctx.set_cluster(alloc5, flags5, ranks5)
D5, X5 = ctx.build_snapshot_resident(res_node=rnode, res_cols=rcols5)
pos_of_node = np.full(n5 + 1, n5, dtype=np.int64)
pos_of_node[X5] = np.arange(len(X5))
info = ctx.device_info()


def one(i, cache):
    q = np.roll(q5, -i)
    t0 = time.perf_counter()
    o = ctx.fit_batch(FIFO, TIGHT, q)
    return (time.perf_counter() - t0) * 1e3, o, q


rows = []
for cache in (1, 0):
    ctx.set_option("chain_cache", cache)
    for i in range(3):
        one(i, cache)
    for i in range(n_heads):
        ms, o, q = one(i + 3, cache)
        if cache == 1:
            ev = o.results["evaluated"].astype(bool)
            feas = o.results["has_capacity"].astype(bool)
            # (an application's placements start at its own exec_off: gathered per feasible application)
            pos_list = []
            for a in np.nonzero(feas)[0]:
                pos_list.append(pos_of_node[o.placement(int(a))[2]])
            pos = np.concatenate(pos_list) if pos_list else np.zeros(0, dtype=np.int64)
            kk = q["k"].astype(np.int64)
            rows.append({"head": i + 3, "ms_cache_on": ms, "failed_at": int(o.failed_at), "evaluated": int(ev.sum()),
                         "feasible": int(feas.sum()), "placements": int(len(pos)),
                         "max_position": int(pos.max()) if len(pos) else -1,
                         "placements_beyond_8k": int((pos >= 8192).sum()), "placements_beyond_10k": int((pos >= 10240).sum()),
                         "chunks_beyond_10k": int(len(np.unique(pos[pos >= 10240] >> 6))),
                         "apps_reaching_beyond_10k": int(sum(1 for p in pos_list if len(p) and p.max() >= 10240)),
                         "infeasible_evaluated": int((ev & ~feas).sum()),
                         "sum_k_infeasible": int(kk[ev & ~feas].sum()),
                         "first_app_k": int(kk[0]), "last_app_k": int(kk[-1])})
        else:
            rows[i]["ms_cache_off"] = ms
# the instrumented variant: slots visited, cycles by phase, per head (checkpoints off: gf_scan_stats turns the cache off anyway)
ctx.set_option("chain_cache", 0)
for i in range(n_heads):
    q = np.roll(q5, -(i + 3))
    ctx.scan_stats(enable=True, reset=True)
    ctx.fit_batch(FIFO, TIGHT, q)
    xv, dv = ctx.scan_stats(enable=False, reset=False)
    cyc, ticks = ctx.last_fifo_clock
    ph = ctx.last_fifo_phases
    prof = ctx.chain_profile()
    rows[i].update({"exec_slots_visited": xv, "driver_slots_visited": dv, "kernel_ms_instrumented": ticks / 1e5,
                    "phase_cycles": {"stage": ph[0], "driver_scan": ph[1], "executor_scan": ph[2], "slow_path": ph[3], "commit": ph[4],
                                     "spare": ph[5]},
                    "rare_count": prof["rare_count"], "rare_cycles": prof["rare_cycles"], "hw_id": prof["hw_id"], "xcc_id": prof["xcc_id"]})
# is a head slow every time?  the first 24 heads three times each, instrumented, back to back
repeats = []
for i in range(min(24, n_heads)):
    q = np.roll(q5, -(i + 3))
    trio = []
    for _ in range(3):
        ctx.scan_stats(enable=True, reset=True)
        ctx.fit_batch(FIFO, TIGHT, q)
        ctx.scan_stats(enable=False, reset=False)
        pr = ctx.chain_profile()
        trio.append({"kernel_ms": ctx.last_fifo_clock[1] / 1e5, "short": pr["rare_count"]["short"], "short_cycles": pr["rare_cycles"]["short"],
                     "xcc": pr["xcc_id"] & 0xF, "hw_id": pr["hw_id"]})
    repeats.append({"head": i + 3, "runs": trio})
ctx.close()

lat = np.array([r["ms_cache_off"] for r in rows])
med = float(np.median(lat))
slow = lat > 1.2 * med
keys = ["ms_cache_on", "ms_cache_off", "failed_at", "evaluated", "feasible", "placements", "max_position", "placements_beyond_8k",
        "placements_beyond_10k", "chunks_beyond_10k", "apps_reaching_beyond_10k", "infeasible_evaluated", "sum_k_infeasible",
        "exec_slots_visited", "driver_slots_visited", "kernel_ms_instrumented"]
print(f"# config 5 chain, {n_heads} rotated heads, {info['name']}; median {med:.3f} ms (checkpoints off); slow = above 1.2 x median: {int(slow.sum())} heads")
print(f"{'quantity':32s} {'fast mean':>14s} {'slow mean':>14s} {'corr with ms':>13s}")
summary = {}
for k in keys:
    v = np.array([float(r[k]) for r in rows])
    f, s_ = (v[~slow].mean() if (~slow).any() else float('nan')), (v[slow].mean() if slow.any() else float('nan'))
    c = float(np.corrcoef(v, lat)[0, 1]) if v.std() > 0 else 0.0
    summary[k] = {"fast_mean": f, "slow_mean": s_, "corr_with_ms": c}
    print(f"{k:32s} {f:14.3f} {s_:14.3f} {c:13.3f}")
for pk in ("stage", "driver_scan", "executor_scan", "slow_path", "commit", "spare"):
    v = np.array([float(r["phase_cycles"][pk]) for r in rows])
    f, s_ = (v[~slow].mean() if (~slow).any() else float('nan')), (v[slow].mean() if slow.any() else float('nan'))
    c = float(np.corrcoef(v, lat)[0, 1]) if v.std() > 0 else 0.0
    summary["cycles_" + pk] = {"fast_mean": f, "slow_mean": s_, "corr_with_ms": c}
    print(f"{'cycles ' + pk:32s} {f:14.0f} {s_:14.0f} {c:13.3f}")
for code in ("unindexed", "bound", "no_driver", "short"):
    n_ = np.array([float(r["rare_count"][code]) for r in rows])
    cy = np.array([float(r["rare_cycles"][code]) for r in rows])
    ki = np.array([float(r["kernel_ms_instrumented"]) for r in rows])
    sl = ki > 1.2 * np.median(ki)
    print(f"rare ending {code:10s}: count fast {n_[~sl].mean():7.2f} slow {n_[sl].mean():7.2f}   cycles fast {cy[~sl].mean():12.0f} slow {cy[sl].mean():12.0f}"
          f"   (clusters by the instrumented kernel's own time: {int(sl.sum())} slow heads)")
ki = np.array([float(r["kernel_ms_instrumented"]) for r in rows])
xcc = np.array([r["xcc_id"] & 0xF for r in rows])
print("# instrumented kernel ms by XCC_ID[3:0]:", {int(x): (int((xcc == x).sum()), round(float(ki[xcc == x].mean()), 3), round(float(ki[xcc == x].max()), 3)) for x in np.unique(xcc)})
print("# the first heads three times each (kernel ms / 'short' endings / XCC):")
for r in repeats:
    print("  head", r["head"], " | ".join(f"{t['kernel_ms']:.3f} ms short={t['short']} ({t['short_cycles']} cyc) xcc={t['xcc']}" for t in r["runs"]))
print("# quartiles (cache off):", [round(float(np.percentile(lat, p)), 3) for p in (0, 25, 50, 75, 100)])
print("# quartiles (cache on) :", [round(float(np.percentile([r['ms_cache_on'] for r in rows], p)), 3) for p in (0, 25, 50, 75, 100)])
if out_path:
    with open(out_path, "w") as f:
        json.dump({"device": info, "n_heads": n_heads, "median_ms": med, "summary": summary, "heads": rows, "repeats": repeats}, f, indent=1)
