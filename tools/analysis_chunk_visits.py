"""Which 64-slot chunks of the executor priority order does a tightly-pack FIFO chain touch?  (CPU model of the chain's
placements: first fitting driver, executors front to back with the lazy early exit, sparkpods.go:139-146 commit.)
   python tools/analysis_chunk_visits.py [config]      -> histogram of touched chunks, share inside a prefix of N chunks"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
from gangfit import workloads as wl
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
w = wl.config(cfg) if cfg else wl.headline()
s = w.snapshot
order = np.asarray(s.exec_order)
avail = s.avail[order].astype(np.int64).copy()          # slot-ordered table
n = len(order)
touch = np.zeros((n + 63) // 64, dtype=np.int64)       # chunks where an executor or a driver landed
first_exec = []
napps = len(w.k)
for a in range(napps - 1):
    drv, exe, k = w.drv[a].astype(np.int64), w.exe[a].astype(np.int64), int(w.k[a])
    fits_d = np.all(avail >= drv, axis=1)
    if not fits_d.any():
        continue
    p0 = int(np.argmax(fits_d))
    t = avail.copy()
    t[p0] -= drv
    with np.errstate(divide="ignore", invalid="ignore"):
        cap = np.where(exe > 0, t // np.where(exe > 0, exe, 1), 1 << 40)
    cap = np.where((t < 0).any(axis=1), 0, cap.min(axis=1))
    cap = np.minimum(cap, k)
    cs = np.cumsum(cap)
    if k > 0 and cs[-1] < k:
        continue
    take = np.zeros(n, dtype=np.int64)
    if k > 0:
        last = int(np.searchsorted(cs, k))
        take[:last] = cap[:last]
        take[last] = k - (cs[last - 1] if last else 0)
        first_exec.append(int(np.argmax(take > 0)) // 64)
    hosts = take[p0] > 0
    avail -= take[:, None] * exe[None, :]
    if not hosts:
        avail[p0] -= drv
    ch = np.unique(np.concatenate([np.nonzero(take)[0] // 64, [p0 // 64]]))
    touch[ch] += 1
tot = touch.sum()
nzc = np.nonzero(touch)[0]
print(f"config {cfg}: {n} slots = {len(touch)} chunks; {len(nzc)} chunks touched, {tot} (app, chunk) touches over {napps - 1} apps")
for pre in (50, 100, 150, 200, 400, 800):
    print(f"  prefix of {pre:4d} chunks holds {touch[:pre].sum() / tot:6.1%} of the touches")
top = np.argsort(-touch)[:40]
print("  hottest chunks (chunk: touches):", ", ".join(f"{c}:{touch[c]}" for c in top))
srt = np.sort(touch)[::-1]
for m in (50, 100, 150):
    print(f"  the {m} hottest chunks hold {srt[:m].sum() / tot:6.1%}")
