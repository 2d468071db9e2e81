/*
 * gangfit_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the gang bin-packing hot path of palantir/k8s-spark-scheduler,
 * used ONLY as the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under k8s-spark-scheduler_amd/ may include, link or call this.
 *
 * Paths below are relative to /root/reference; LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg.
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - TightlyPack feasibility: pinned by the reference's own tests T1/T3/T4
 *     (internal/extender/resource_test.go:27-71, unschedulablepods_test.go:24-80) — tests/test_oracle_reference_kats.py.
 *   - DistributeEvenly placements and FIFO replay with >=1 earlier driver: PARITY UNPINNED — no reference test
 *     selects them and the Go reference cannot be built here (no Go toolchain). They follow the cited source
 *     line by line and are cross-checked against an independent pure-Python restatement (oracle/pyoracle.py).
 *
 * Conventions: a node "name" is a dense uint32 index; an index >= n_nodes is a name that is NOT a key of
 * nodesSchedulingMetadata (the `!ok` branches of the reference). Quantities are canonical int64:
 * cpu milli-cores, memory bytes, gpu count (resource.Quantity restricted to exactly-representable values).
 */
#ifndef GANGFIT_ORACLE_H
#define GANGFIT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GO_ALGO_TIGHTLY_PACK 0      /* LIB/binpack/pack_tightly.go:25-63 */
#define GO_ALGO_DISTRIBUTE_EVENLY 1 /* LIB/binpack/distribute_evenly.go:25-73 */
#define GO_ALGO_MINIMAL_FRAGMENTATION 2 /* LIB/binpack/minimal_fragmentation.go:33-137 */
#define GO_ALGO_AZ_AWARE_TIGHTLY_PACK 3 /* LIB/binpack/az_aware_pack_tightly.go:27-38 */
#define GO_ALGO_SINGLE_AZ_TIGHTLY_PACK 4 /* LIB/binpack/single_az_pack_tightly.go:21 + single_az.go:23-97 */
#define GO_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION 5 /* LIB/binpack/single_az_minimal_fragmentation.go:20 */

#define GO_NO_NODE 0xFFFFFFFFu
#define GO_APP_SKIPPABLE 1u /* shouldSkipDriverFifo() is true for this earlier driver, resource.go:264-270 */

typedef struct go_app {
    int64_t drv[3]; /* driver   cpu milli, mem bytes, gpu */
    int64_t exe[3]; /* executor cpu milli, mem bytes, gpu */
    int32_t k;      /* executorCount (MinExecutorCount for dynamic allocation, resource.go:242,325) */
    uint32_t flags;
} go_app;

typedef struct go_result {
    int32_t has_capacity;  /* PackingResult.HasCapacity */
    uint32_t driver_node;  /* PackingResult.DriverNode, GO_NO_NODE when empty */
    uint32_t exec_len;     /* len(PackingResult.ExecutorNodes): k when feasible, 0 otherwise */
    uint32_t evaluated;    /* 0 for apps after a FIFO chain abort */
} go_result;

/* One SparkBinPack evaluation, literal loops (binpack.go:60-87 + the chosen executor packer).
 * avail: n_nodes x 3 row-major (AvailableResources).  exec_out: room for app->k entries.
 * Returns has_capacity. */
int go_spark_binpack(int algo, const int64_t *avail, uint32_t n_nodes, const go_app *app,
                     const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order, uint32_t n_x,
                     uint32_t *driver_out, uint32_t *exec_out);

/* Same decision through the array/closed-form restatement (SURVEY.md section 8 "array restatement"):
 * capacities by floor division, O(N) driver choice.  Must agree with go_spark_binpack everywhere. */
int go_spark_binpack_closed_form(int algo, const int64_t *avail, uint32_t n_nodes, const go_app *app,
                                 const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                                 uint32_t n_x, uint32_t *driver_out, uint32_t *exec_out);

/* Batch of independent decisions against ONE snapshot (the unschedulable-pod scan shape,
 * internal/extender/unschedulablepods.go:93-166).  exec_out is the concatenation, app i at exec_off[i].
 * closed_form selects which restatement runs. */
void go_fit_independent(int algo, int closed_form, const int64_t *avail, uint32_t n_nodes, const go_app *apps,
                        uint32_t n_apps, const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                        uint32_t n_x, go_result *results, const uint64_t *exec_off, uint32_t *exec_out);

/* FIFO replay + final pack (internal/extender/resource.go:224-262, 304-328): apps[0..n_apps-2] are the earlier
 * drivers in creation order, apps[n_apps-1] is the driver being filtered.  avail is mutated exactly like
 * availableNodesSchedulingMetadata (sparkResourceUsage overwrite quirk, sparkpods.go:139-146).
 * Returns the index of the first non-skippable earlier driver that did not fit ("failure-earlier-driver"),
 * or -1. */
int32_t go_fit_fifo_chain(int algo, int closed_form, int64_t *avail, uint32_t n_nodes, const go_app *apps,
                          uint32_t n_apps, const uint32_t *driver_order, uint32_t n_d,
                          const uint32_t *exec_order, uint32_t n_x, go_result *results,
                          const uint64_t *exec_off, uint32_t *exec_out);

/* The same two batch shapes for EVERY registered packer (internal/binpacker/binpack.go:43-49), including the
 * zone-aware wrappers: sched = n_nodes x 3 SchedulableResources (efficiencies), zone = n_nodes zone ids
 * (NodeSchedulingMetadata.ZoneLabel; NULL = a single zone).  avg_out (nullable): n_apps x 4 doubles
 * {CPU, Memory, GPU, Max} = ComputeAvgPackingEfficiency over [driver] ++ executors of each returned result, in
 * slice order (what chooseBestResult compares, single_az.go:83-93); zeros for infeasible apps. */
void go_fit_independent_ex(int algo, int closed_form, const int64_t *avail, const int64_t *sched,
                           const uint32_t *zone, uint32_t n_nodes, const go_app *apps, uint32_t n_apps,
                           const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order, uint32_t n_x,
                           go_result *results, const uint64_t *exec_off, uint32_t *exec_out, double *avg_out);
int32_t go_fit_fifo_chain_ex(int algo, int closed_form, int64_t *avail, const int64_t *sched, const uint32_t *zone,
                             uint32_t n_nodes, const go_app *apps, uint32_t n_apps, const uint32_t *driver_order,
                             uint32_t n_d, const uint32_t *exec_order, uint32_t n_x, go_result *results,
                             const uint64_t *exec_off, uint32_t *exec_out);

/* go_fit_fifo_chain_ex (literal) plus what the reference's SparkBinPack also does on every successful pack: the per-node
 * PackingEfficiencies map over all nodes (binpack.go:77), discarded by fitEarlierDrivers.  Same results. */
int32_t go_fit_fifo_chain_with_efficiencies(int algo, int64_t *avail, const int64_t *sched, const uint32_t *zone,
                                            uint32_t n_nodes, const go_app *apps, uint32_t n_apps,
                                            const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                                            uint32_t n_x, go_result *results, const uint64_t *exec_off, uint32_t *exec_out);

/* ComputeAvgPackingEfficiency (efficiency.go:114-156) over nodeNames = [driver] ++ exec_nodes, duplicates counted,
 * summed in slice order.  reserved_includes_executors = 0 reproduces minimalFragmentation, which never writes its
 * placements into the `reserved` map (minimal_fragmentation.go:59-91): only the driver entry exists there. */
void go_avg_packing_efficiency_list(const int64_t *avail, const int64_t *sched, uint32_t n_nodes, const go_app *app,
                                    uint32_t driver_node, const uint32_t *exec_nodes, uint32_t n_exec,
                                    int reserved_includes_executors, double avg_out[4]);

/* capacity.GetNodeCapacity (LIB/capacity/capacity.go:36-75) on canonical int64; INT64_MAX stands for math.MaxInt. */
int64_t go_node_capacity(const int64_t avail[3], const int64_t reserved[3], const int64_t required[3]);

/* Packing efficiencies (LIB/binpack/efficiency.go:66-156).  sched: n_nodes x 3 SchedulableResources.
 * Per-node efficiencies after placing (driver_node, exec nodes) go to eff_out (n_nodes x 3 doubles: cpu, mem, gpu;
 * nullable).  avg_out[4] = {CPU, Memory, GPU, Max} of ComputeAvgPackingEfficiency over ALL nodes summed in
 * ascending node-index order (the reference sums in Go map order, which is unspecified —
 * internal/extender/resource.go:372-381). */
void go_packing_efficiency(const int64_t *avail, const int64_t *sched, uint32_t n_nodes, const go_app *app,
                           uint32_t driver_node, const uint32_t *exec_nodes, uint32_t n_exec, double *eff_out,
                           double avg_out[4]);

/* Single executor first-fit used by rescheduleExecutor (internal/extender/resource.go:658-662). */
uint32_t go_executor_first_fit(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3],
                               const uint32_t *exec_order, uint32_t n_x);

/* The same loop against the snapshot minus `reserved` (n_nodes x 3 or NULL): the overhead that the reference counts a
 * second time on this path (resource.go:640-643). */
uint32_t go_executor_first_fit_reserved(const int64_t *avail, uint32_t n_nodes, const int64_t *reserved,
                                        const int64_t exe[3], const uint32_t *exec_order, uint32_t n_x);

/* rescheduleExecutorWithMinimalFragmentation (internal/extender/resource.go:675-703). */
uint32_t go_executor_min_frag(const int64_t *avail, uint32_t n_nodes, const int64_t *reserved, const int64_t exe[3],
                              const uint32_t *exec_order, uint32_t n_x, const uint8_t *hosts);

/* The same FIFO chain (chain != 0) or independent batch (chain == 0) written with the reference's data structures — string
 * node names, hash maps for the metadata, `reserved`, the usage map and the per-node PackingEfficiencies of every
 * successful pack (oracle/gangfit_oracle_maps.cpp).  tightly-pack and distribute-evenly only.  Same results as
 * go_fit_fifo_chain / go_fit_independent; this is the CPU baseline whose COST resembles the Go code's. */
int32_t go_fit_maps(int algo, int chain, int64_t *avail, const int64_t *sched, uint32_t n_nodes, const go_app *apps,
                    uint32_t n_apps, const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order, uint32_t n_x,
                    go_result *results, const uint64_t *exec_off, uint32_t *exec_out);

/* findNodes (internal/extender/failover.go:412-436): tightly-pack with partial results, no driver, and no `Sub` of the
 * add that fails the comparison.  Returns the number of executors placed (<= executor_count); exec_out receives them;
 * adds_out (nullable, n_nodes, zeroed here) = number of `reserved[n].Add(executorResources)` calls per node, i.e. the
 * returned `reserved` map is adds_out[n] x exe (0 = the node has no entry). */
uint32_t go_find_nodes(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t executor_count,
                       const uint32_t *ordered_nodes, uint32_t n_o, uint32_t *exec_out, uint32_t *adds_out);
uint32_t go_find_nodes_closed_form(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t executor_count,
                                   const uint32_t *ordered_nodes, uint32_t n_o, uint32_t *exec_out, uint32_t *adds_out);
/* n_req findNodes calls in sequence, each followed by availableResources.Sub(reservedResources) (failover.go:159):
 * avail (n_nodes x 3) is mutated.  exe: n_req x 3; adds_out nullable, n_req x n_nodes. */
void go_find_nodes_chain(int closed_form, int64_t *avail, uint32_t n_nodes, const int64_t *exe, const int32_t *k,
                         uint32_t n_req, const uint32_t *ordered_nodes, uint32_t n_o, uint32_t *placed_out,
                         const uint64_t *exec_off, uint32_t *exec_out, uint32_t *adds_out);

#ifdef __cplusplus
}
#endif
#endif
