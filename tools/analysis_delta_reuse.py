"""could a chain checkpoint survive the snapshot delta of a SUCCESSFUL Filter?  (CPU only, the oracle's chain.)

A successful Filter creates a ResourceReservation: the next Filter's snapshot has the scheduled driver's REAL usage subtracted
(driver request on its node, one executor request per executor: UsageForNodes, LIB/resources/resources.go:31-43), the scheduled
driver leaves the pending queue, and the priority order is recomputed.  A checkpoint "table before application i" of the previous
chain stays valid for the new chain only if (a) the records in front of i are the same records, (b) no node the prefix visited
changed, (c) the priority order of the visited front is unchanged.

Measured here on the headline queue (10 000 nodes x 1 000 pending drivers, tightly-pack):
  1. the scheduled driver is the queue's HEAD (FIFO: the earliest pending driver is the one whose Filter succeeds): the new queue
     is the old one shifted by one, so what the old chain held "before application 1" is snapshot - quirk_usage(app 0), while the
     new chain starts from snapshot - real_usage(app 0).  Equal only when app 0's executors sit on K distinct nodes and none on the
     driver's node -> fraction of feasible applications for which quirk usage == real usage;
  2. a LATER driver j is scheduled: records 0 .. j-1 are unchanged, but its reservation lands on nodes of the front -> is any of
     its nodes inside the chunks (64 slots of the priority order) that the first 32 / 64 / ... applications visited?
  3. how often the real reservation changes the ORDER inside the visited front (free memory is the first sort key).
"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "k8s-spark-scheduler_amd")]
from gangfit import workloads as wl
from oracle import binding as ob

w = wl.headline(10000, 1000)
s = w.snapshot
apps = ob.make_apps(w.drv, w.exe, w.k, w.flags)
ref = ob.fit_fifo_chain(0, s.avail, apps, s.driver_order, s.exec_order, sched=s.sched)
pos_of = np.full(len(s.avail) + 1, -1, dtype=np.int64)
pos_of[s.exec_order] = np.arange(len(s.exec_order))
feas = np.nonzero(ref.results["has_capacity"])[0]
same_usage = 0
nodes_of = {}
for a in feas:
    d, _, ex = ref.placement(int(a))
    ex = np.asarray(ex)
    nodes_of[int(a)] = (int(d), ex)
    distinct = len(np.unique(ex)) == len(ex)
    if distinct and int(d) not in set(ex.tolist()):
        same_usage += 1
print(f"feasible applications: {len(feas)} of {len(apps)}")
print(f"1. quirk usage == real usage (K executors on K distinct nodes, none on the driver's): {same_usage} of {len(feas)} "
      f"= {same_usage / max(1, len(feas)):.3%}  -> a chain behind a scheduled HEAD can reuse nothing in the other cases")
# chunks visited by the prefix 0..i-1 (superset used: the chunks its placements and drivers touch)
touched = []
acc = set()
for a in range(len(apps)):
    if a in nodes_of:
        d, ex = nodes_of[a]
        for nidx in [d] + ex.tolist():
            acc.add(int(pos_of[nidx]) >> 6)
    touched.append(set(acc))
for i in (32, 64, 128, 256, 512):
    hit = tot = 0
    for j in feas:
        if j < i:
            continue
        d, ex = nodes_of[int(j)]
        ch = {int(pos_of[n]) >> 6 for n in [d] + ex.tolist()}
        tot += 1
        hit += bool(ch & touched[i - 1])
    print(f"2. checkpoint before application {i}: a later scheduled driver's reservation lands inside the chunks the prefix touched "
          f"for {hit} of {tot} candidates = {hit / max(1, tot):.1%}  (chunks touched by the prefix: {len(touched[i - 1])})")
# 3. order changes: the reservation lowers free memory of its nodes; does any of them change rank relative to a neighbour in the front?
order = s.exec_order
mem = s.avail[:, 1].astype(np.int64)
cpu = s.avail[:, 0].astype(np.int64)
moved = 0
for j in feas[:200]:
    d, ex = nodes_of[int(j)]
    newmem, newcpu = mem.copy(), cpu.copy()
    newmem[d] -= w.drv[j][1]; newcpu[d] -= w.drv[j][0]
    for n in ex.tolist():
        newmem[n] -= w.exe[j][1]; newcpu[n] -= w.exe[j][0]
    changed = np.unique(np.array([d] + ex.tolist()))
    # a node keeps its place iff it still sorts between its old neighbours
    keep = True
    for n in changed:
        p = int(pos_of[n])
        if p > 0:
            q = order[p - 1]
            if q not in changed and (newmem[n], newcpu[n]) < (newmem[q], newcpu[q]):
                keep = False
    moved += not keep
print(f"3. the real reservation of one scheduled driver moves at least one of its nodes forward in the priority order "
      f"(free memory is the first key) in {moved} of {min(200, len(feas))} cases = {moved / min(200, len(feas)):.1%}")
