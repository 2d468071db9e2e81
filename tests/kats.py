"""Known-answer cases for the gang-fit decision, shared by the oracle tests (CPU) and the HIP parity tests (GPU).

Two provenances, kept apart on purpose:

REFERENCE_PINNED — outcomes asserted by the reference's OWN tests, restated at the binpack boundary.  The reference
  tests run `single-az-tightly-pack` over 2 nodes that all land in zone "default" (SURVEY.md quirk 7), which is
  exactly plain TightlyPack; node shape from extendertest.NewNode (8 cpu, 8 GiB, 1 gpu,
  internal/extender/extendertest/extender_test_utils.go:239-271), app shape from StaticAllocationSparkPods
  (driver 1 cpu / "1" byte / 1 gpu, executors 1 cpu / "1" byte, :283-320).

HAND_DERIVED — traced by hand from the Go source.  NOT reference test vectors.

Quantities: [cpu milli, mem bytes, gpu].  Node names are indices; index >= n_nodes = name missing from metadata.
"""
GIB = 1 << 30
TIGHT, EVEN = 0, 1

_NODE = [8000, 8 * GIB, 1]  # extendertest.NewNode

REFERENCE_PINNED = [
    # TestScheduler (internal/extender/resource_test.go:27-71): driver Filter of "2-executor-app" succeeds; by code
    # trace the reservation is driver node1, executors [node1, node1].
    dict(name="T1 TestScheduler 2-executor-app fits", algo=TIGHT, avail=[_NODE, _NODE], D=[0, 1], X=[0, 1],
         drv=[1000, 1, 1], exe=[1000, 1, 0], k=2, feasible=True, driver=0, execs=[0, 0]),
    # TestUnschedulablePodMarker (unschedulablepods_test.go:24-53): 2 executors fit an empty 2-node cluster, 100 do not.
    dict(name="T3a UnschedulablePodMarker 2 executors fit", algo=TIGHT, avail=[_NODE, _NODE], D=[0, 1], X=[0, 1],
         drv=[1000, 1, 1], exe=[1000, 1, 0], k=2, feasible=True, driver=0, execs=[0, 0]),
    dict(name="T3b UnschedulablePodMarker 100 executors exceed", algo=TIGHT, avail=[_NODE, _NODE], D=[0, 1], X=[0, 1],
         drv=[1000, 1, 1], exe=[1000, 1, 0], k=100, feasible=False, driver=None, execs=[]),
    # TestSchedulerFailsToScheduleWhenNotEnoughNvidiaGPUs (unschedulablepods_test.go:55-80): driver + 2 executors need
    # 3 gpus, the cluster has 2 -> pins the third resource dimension.
    dict(name="T4 not enough nvidia gpus", algo=TIGHT, avail=[_NODE, _NODE], D=[0, 1], X=[0, 1],
         drv=[1000, 1, 1], exe=[1000, 1, 1], k=2, feasible=False, driver=None, execs=[]),
]

HAND_DERIVED = [
    dict(name="K1 tightly pack", algo=TIGHT, avail=[[4, 4, 0], [2, 2, 0], [8, 8, 0]], D=[0, 1, 2], X=[0, 1, 2],
         drv=[2, 2, 0], exe=[1, 1, 0], k=5, feasible=True, driver=0, execs=[0, 0, 1, 1, 2]),
    dict(name="K2 distribute evenly", algo=EVEN, avail=[[4, 4, 0], [2, 2, 0], [8, 8, 0]], D=[0, 1, 2], X=[0, 1, 2],
         drv=[2, 2, 0], exe=[1, 1, 0], k=5, feasible=True, driver=0, execs=[0, 1, 2, 0, 1]),
    dict(name="K3 driver skip (tight)", algo=TIGHT, avail=[[1, 1, 0], [10, 10, 0]], D=[0, 1], X=[0, 1],
         drv=[2, 2, 0], exe=[1, 1, 0], k=3, feasible=True, driver=1, execs=[0, 1, 1]),
    dict(name="K3 driver skip (even)", algo=EVEN, avail=[[1, 1, 0], [10, 10, 0]], D=[0, 1], X=[0, 1],
         drv=[2, 2, 0], exe=[1, 1, 0], k=3, feasible=True, driver=1, execs=[0, 1, 1]),
    dict(name="K4 infeasible (tight)", algo=TIGHT, avail=[[2, 2, 0], [3, 3, 0]], D=[0, 1], X=[0, 1],
         drv=[2, 2, 0], exe=[1, 1, 0], k=4, feasible=False, driver=None, execs=[]),
    dict(name="K4 infeasible (even)", algo=EVEN, avail=[[2, 2, 0], [3, 3, 0]], D=[0, 1], X=[0, 1],
         drv=[2, 2, 0], exe=[1, 1, 0], k=4, feasible=False, driver=None, execs=[]),
    dict(name="K5 zero executors (tight)", algo=TIGHT, avail=[[1, 1, 0], [5, 5, 0]], D=[0, 1], X=[0, 1],
         drv=[2, 2, 0], exe=[1, 1, 0], k=0, feasible=True, driver=1, execs=[]),
    dict(name="K5 zero executors (even)", algo=EVEN, avail=[[1, 1, 0], [5, 5, 0]], D=[0, 1], X=[0, 1],
         drv=[2, 2, 0], exe=[1, 1, 0], k=0, feasible=True, driver=1, execs=[]),
    # zero-size executor: adding zero never exceeds -> the first candidate absorbs everything
    dict(name="K6a zero-size executor", algo=TIGHT, avail=[[5, 5, 0], [1, 1, 0]], D=[0], X=[1, 0],
         drv=[1, 1, 0], exe=[0, 0, 0], k=3, feasible=True, driver=0, execs=[1, 1, 1]),
    # ... unless the node is already overcommitted in some dimension (0 > -1)
    dict(name="K6b zero-size executor, overcommitted node", algo=TIGHT, avail=[[5, 5, 0], [-1, 5, 0]], D=[0], X=[1, 0],
         drv=[1, 1, 0], exe=[0, 0, 0], k=3, feasible=True, driver=0, execs=[0, 0, 0]),
    # node name 7 is not in the metadata: contributes nothing, is skipped as a driver
    dict(name="K8 unknown node in orders (tight)", algo=TIGHT, avail=[[4, 4, 0], [4, 4, 0]], D=[7, 1, 0], X=[0, 7, 1],
         drv=[1, 1, 0], exe=[2, 2, 0], k=3, feasible=True, driver=1, execs=[0, 0, 1]),
    dict(name="K8 unknown node in orders (even)", algo=EVEN, avail=[[4, 4, 0], [4, 4, 0]], D=[7, 1, 0], X=[0, 7, 1],
         drv=[1, 1, 0], exe=[2, 2, 0], k=3, feasible=True, driver=1, execs=[0, 1, 0]),
    # the first fitting driver candidate steals the capacity the executors need; a later candidate works
    dict(name="K9 fallback driver choice", algo=TIGHT, avail=[[3, 3, 0], [1, 1, 0]], D=[0, 1], X=[0],
         drv=[1, 1, 0], exe=[1, 1, 0], k=3, feasible=True, driver=1, execs=[0, 0, 0]),
    # driver candidate that is not an executor candidate
    dict(name="K10 driver outside executor order", algo=EVEN, avail=[[2, 2, 0], [9, 9, 0]], D=[0], X=[1],
         drv=[2, 2, 0], exe=[2, 2, 0], k=4, feasible=True, driver=0, execs=[1, 1, 1, 1]),
    # multi-pass distribute evenly with uneven capacities: caps (3,1,2) -> passes [0,1,2],[0,2],[0]
    dict(name="K11 distribute evenly three passes", algo=EVEN, avail=[[3, 9, 0], [1, 9, 0], [2, 9, 0], [9, 9, 0]],
         D=[3], X=[0, 1, 2], drv=[1, 1, 0], exe=[1, 1, 0], k=6, feasible=True, driver=3, execs=[0, 1, 2, 0, 2, 0]),
    # gpu dimension binds
    dict(name="K12 gpu-bound capacity", algo=TIGHT, avail=[[64000, 256 * GIB, 2], [64000, 256 * GIB, 8]], D=[0, 1],
         X=[0, 1], drv=[1000, GIB, 0], exe=[1000, GIB, 1], k=5, feasible=True, driver=0, execs=[0, 0, 1, 1, 1]),
]

ALL = REFERENCE_PINNED + HAND_DERIVED

# FIFO replay quirk: earlier app drv (2,2) exe (3,3) K=3 tight on n1(10,10) n2(10,10) -> driver n1,
# execs [n1,n1,n2]; replay subtracts exe ONCE per distinct node and drops the driver entry that an executor overwrote:
# residual n1 (7,7), n2 (7,7).
FIFO_K7 = dict(avail=[[10, 10, 0], [10, 10, 0]], D=[0, 1], X=[0, 1],
               apps=[dict(drv=[2, 2, 0], exe=[3, 3, 0], k=3, skippable=False),
                     dict(drv=[1, 1, 0], exe=[7, 7, 0], k=2, skippable=False)],
               first=dict(driver=0, execs=[0, 0, 1]), residual=[[7, 7, 0], [7, 7, 0]],
               last=dict(feasible=False))
