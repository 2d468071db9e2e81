/*
 * gangfit.h — C ABI of libgangfit, the MI355X-native gang-scheduling bin-packer.
 *
 * Drop-in boundary for ONE hot path of palantir/k8s-spark-scheduler: the gang-fit decision
 * (1 driver + K executors, 3-D cpu/mem/gpu, against a snapshot of per-node free capacity) behind the reference's
 * plug-in seam `binpack.SparkBinPackFunction`.  Everything here is what a cgo shim in the reference tree binds
 * (INTEGRATION.md shows that shim).  File:line citations are relative to the reference repository;
 * LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg.
 *
 * Conventions
 *   - Node identity crosses the boundary as a dense uint32 index into the snapshot arrays; the caller owns the
 *     name<->index table.  An index >= n_nodes inside an order vector is a node name that is not a key of
 *     nodesSchedulingMetadata (the `!ok` branches at LIB/binpack/binpack.go:68, pack_tightly.go:51,
 *     distribute_evenly.go:59): it never hosts anything.
 *   - Quantities are canonical int64: cpu in milli-cores, memory in bytes, gpu in devices
 *     (resource.Quantity restricted to exactly representable values; the shim must route anything else to the Go
 *     CPU path).  |available| < 2^62; driver/executor requests in [0, 2^62).
 *   - All host pointers are plain caller-owned arrays, copied before the call returns (cgo pointer rules); outputs
 *     are caller-allocated.  No callbacks, no exceptions, no stdout.
 *   - Return value: 0 = the call ran (feasible or not is in the results, like PackingResult.HasCapacity,
 *     LIB/binpack/binpack.go:25-40); < 0 = the accelerator could not serve the call — the shim must then fall back
 *     to the Go function so that Filter semantics never change.
 *   - Thread safety: every gf_ctx entry point is serialised by a per-context mutex (Predicate and the
 *     UnschedulablePodMarker goroutine may call concurrently: cmd/server.go:230, internal/extender/unschedulablepods.go:77-91).
 *     A SEQUENCE of calls that belongs together (gf_snapshot_set / gf_zones_set / gf_orders_set or gf_snapshot_build, then
 *     a fit on that snapshot) is NOT atomic by itself: bracket it with gf_ctx_lock / gf_ctx_unlock (or an equivalent
 *     mutex of the caller — the Go shim's Context.mu), otherwise another thread's snapshot can slip in between and the
 *     fit still returns GF_OK.  The *_dev entry points are asynchronous on the given stream: their host side (buffer
 *     growth, state flags) takes the context mutex, but the device work of two such calls on DIFFERENT streams is not
 *     ordered by the library.
 */
#ifndef GANGFIT_H
#define GANGFIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_VERSION 300 /* 0.3.0 */

/* ---- status codes ---- */
#define GF_OK 0
#define GF_ERR_NO_DEVICE (-1)  /* no usable gfx950 device / hipInit failed */
#define GF_ERR_HIP (-2)        /* a HIP runtime call failed; see gf_last_error */
#define GF_ERR_INVALID (-3)    /* bad argument (NULL, duplicate node in exec order, k < 0, value out of range...) */
#define GF_ERR_CAPACITY (-4)   /* caller's exec_nodes buffer is smaller than sum(k) */
#define GF_ERR_STATE (-5)      /* snapshot/orders not set yet */
#define GF_ERR_UNSUPPORTED (-6)/* algo/mode combination not served by the device path */

#define GF_NO_NODE 0xFFFFFFFFu
#define GF_MAX_K (1 << 20)     /* executorCount "is small (<1000)" — internal/extender/sparkpods.go:110 */
#define GF_MAX_ABS_QUANTITY (INT64_C(1) << 62)

/* Executor packers selectable by name in the reference (internal/binpacker/binpack.go:43-49). */
typedef enum gf_algo {
    GF_ALGO_TIGHTLY_PACK = 0,      /* "tightly-pack"      LIB/binpack/pack_tightly.go:25-63 */
    GF_ALGO_DISTRIBUTE_EVENLY = 1, /* "distribute-evenly" LIB/binpack/distribute_evenly.go:25-73 */
    GF_ALGO_MINIMAL_FRAGMENTATION = 2, /* binpack.MinimalFragmentation (not in the registry; the zone-less inner packer)
                                          LIB/binpack/minimal_fragmentation.go:27-137 */
    /* zone-aware packers: need gf_zones_set and the schedulable columns of gf_snapshot_set */
    GF_ALGO_AZ_AWARE_TIGHTLY_PACK = 3,  /* "az-aware-tightly-pack"  LIB/binpack/az_aware_pack_tightly.go:27-38 */
    GF_ALGO_SINGLE_AZ_TIGHTLY_PACK = 4, /* "single-az-tightly-pack" LIB/binpack/single_az_pack_tightly.go:21,
                                           single_az.go:23-97 — what every reference test selects */
    GF_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION = 5, /* "single-az-minimal-fragmentation"
                                           LIB/binpack/single_az_minimal_fragmentation.go:20 */
} gf_algo;

typedef enum gf_mode {
    /* n_apps independent SparkBinPack evaluations against the same snapshot — the batched form of
     * UnschedulablePodMarker.scanForUnschedulablePods (internal/extender/unschedulablepods.go:93-166). */
    GF_MODE_INDEPENDENT = 0,
    /* fitEarlierDrivers + final pack (internal/extender/resource.go:224-262, 309-328): apps[0..n-2] are the earlier
     * drivers in creation order, apps[n-1] is the driver being filtered; each feasible earlier app's usage is
     * subtracted from the working copy of the snapshot exactly like sparkResourceUsage + SubtractUsageIfExists
     * (internal/extender/sparkpods.go:139-146: ONE executor request per distinct executor node; the driver request only
     * if the driver node hosts no executor). */
    GF_MODE_FIFO_CHAIN = 1,
} gf_mode;

/* gf_app.flags */
#define GF_APP_SKIPPABLE 1u /* shouldSkipDriverFifo(driver) is true (resource.go:264-270): an unfit earlier driver is ignored */

/* One Spark application = types.SparkApplicationResources (internal/types/types.go:22-27) in canonical units. */
typedef struct gf_app {
    int64_t drv[3];    /* DriverResources   {cpu milli, memory bytes, gpu} */
    int64_t exe[3];    /* ExecutorResources {cpu milli, memory bytes, gpu} */
    int32_t k;         /* executorCount handed to the packer = MinExecutorCount (resource.go:242,325) */
    uint32_t flags;
    uint64_t exec_off; /* offset of this app's k placements in exec_nodes.  gf_fit_batch fills it (prefix sum of k)
                          in its private copy; callers of gf_fit_batch_dev must fill it themselves. */
} gf_app;              /* 64 bytes */

/* binpack.PackingResult (LIB/binpack/binpack.go:25-30) without the efficiency map (see gf_packing_efficiency). */
typedef struct gf_result {
    int32_t has_capacity; /* HasCapacity */
    uint32_t driver_node; /* DriverNode, GF_NO_NODE when empty */
    uint32_t exec_len;    /* len(ExecutorNodes): k when feasible, 0 otherwise; entries live at exec_nodes[app.exec_off ...) in
                             reservation order executor-1..executor-k (internal/extender/resourcereservations.go:501-502) */
    uint32_t evaluated;   /* 1 if the app was evaluated; 0 for apps behind a FIFO chain abort */
} gf_result;              /* 16 bytes */

typedef struct gf_ctx gf_ctx;

int gf_version(void);

/* Create a context.  Replaces nothing in the reference; the shim calls it right after SelectBinpacker (cmd/server.go:145).
 *   n_dev == 1 (or device_ids NULL, n_dev 0 = device 0): everything runs on that device.
 *   n_dev  > 1 (<= 16): ONE context over several devices of the box — the node table shards by range of the priority order
 *     (SURVEY.md section 8e).  gf_snapshot_set / gf_zones_set / gf_orders_set / gf_snapshot_build install the snapshot on
 *     every device; gf_fit_batch(GF_MODE_INDEPENDENT) with tightly-pack or distribute-evenly then evaluates every
 *     application on every device's range (four device steps per device) and stitches the result with three exchanges
 *     done by peer access over xGMI inside the call: two all-gathers of 16 B per application (written straight into the
 *     peers' tables) and one reduction of the placement buffer onto the first device.  Same gf_result / ExecutorNodes as on
 *     one device, bit for bit.  Everything else (FIFO chains — each commit must be visible to the next scan —, zone-aware
 *     and minimal-fragmentation packers, orders that do not merge into one, single executors, findNodes, efficiencies, the
 *     *_dev entry points) runs on the first device.  One submitting thread per device issues that device's launches
 *     (the calling thread is the first device's; the others park between batches), so a batch costs the host about what
 *     one device's half a dozen runtime calls cost.  A device id may repeat (several shards on one GPU): that is how the
 *     path is tested on a one-GPU box — the shards of one device then share a sub-context (one launch per step with a grid
 *     row per shard; the environment variable GANGFIT_TEST_GROUP_SPLIT=1, a test switch, gives every listed id a
 *     sub-context, a stream and a submitting thread of its own instead).  gf_shard_count returns the number of listed ids.  When two distinct devices cannot access each other's memory the context DEGRADES
 *     to the first device (GF_OK; gf_shard_count returns 1, gf_last_error says why): a host must not lose the accelerator
 *     because a topology lacks peer access.  The exchange buffers live in fine-grained memory.  Self-check: the first
 *     sharded batch on every newly installed snapshot is also answered by the first device alone; on a mismatch the
 *     context stops sharding (gf_shard_count 1, gf_last_error set) and serves the first device's answer — a wrong exchange
 *     never decides a Filter.  gf_set_option(ctx, "group_exchange", 1) moves the three exchanges onto RCCL collectives
 *     (librccl bound at run time, ncclCommInitAll over the context's devices: two all-gathers and one reduce per batch, each
 *     a grouped call over every device's stream); GF_ERR_UNSUPPORTED when the library is missing or refuses the device list
 *     (a repeated id), the peer-store exchange then stays in place.  Test switches of a multi-device context:
 *     "group_verify" (0 skips the self-check), "group_fault" (1 drops the placement reduction, 2 zeroes the other shards' capacity sums), "group_shard_off". */
int gf_init(const int *device_ids, int n_dev, gf_ctx **out);
/* Number of node-range shards independent batches are split into right now: the device count of gf_init, or 1 for a
 * single-device context, a degraded one, or one whose self-check failed. */
int gf_shard_count(gf_ctx *ctx);
void gf_destroy(gf_ctx *ctx);

/* A VIEW of a context: a context of its own (own stream, own working tables, own chain cache, own mutex) that fits on the
 * snapshot installed in `parent`, without a copy of it.  Predicate and the UnschedulablePodMarker run concurrently in the
 * reference (cmd/server.go:230), and so do Filters of different queues on one instance group's snapshot: N views run N FIFO
 * chains at the same time on N compute units of the one GPU, after ONE install.
 *   - a view serves the entry points that read the snapshot (gf_fit_batch[_dev], gf_spark_binpack, gf_residual_get,
 *     gf_snapshot_get, gf_executor_fit, gf_find_nodes, the efficiencies); the ones that install (gf_snapshot_set,
 *     gf_zones_set, gf_orders_set, gf_cluster_set, gf_snapshot_build*, gf_usage_*) return GF_ERR_STATE on it;
 *   - an install on the parent waits for the views' calls in flight and blocks new ones while it runs; the next call of a
 *     view then sees the new snapshot (and starts with an empty chain cache);
 *   - destroy the views before the parent.  GF_ERR_UNSUPPORTED for a multi-device parent. */
int gf_ctx_view(gf_ctx *parent, gf_ctx **out);

/* Sequence lock of a context: gf_ctx_lock blocks until no other caller holds it.  Unlike the internal per-call mutex it may
 * be released from a different OS thread than the one that took it (a goroutine can migrate between two cgo calls).  Not
 * re-entrant.  Every other entry point may be called with or without holding it. */
void gf_ctx_lock(gf_ctx *ctx);
void gf_ctx_unlock(gf_ctx *ctx);

/* Message for the last failing call on this context (C-owned, valid until the next call on ctx). */
const char *gf_last_error(gf_ctx *ctx);

/* Diagnostic / test switches of one context (a deployment needs none of them: every default is the fast path; tests use
 * them to drive the fallback kernels on inputs the fast paths would take).  Applies to what is installed or launched
 * afterwards.  Keys:
 *   "chain_cache"            0: every FIFO chain replays from the snapshot (also GANGFIT_CHAIN_CACHE=0 in the environment)
 *   "fifo_generic"           1: FIFO chains run on the wide / generic global-memory kernels only
 *   "lds_budget"             bytes of LDS one workgroup may use (<= the device's): smaller table fronts, global tails
 *   "minfrag_matrix", "minfrag_hist", "sparse_gpu", "zero_copy"   0 disables the respective structure
 *   "feasible_announce"      0: gf_fit_feasible waits for its stream instead of watching the answers arrive in pinned memory
 *                            (GANGFIT_WAIT=block has the same effect: no spinning on the answers)
 *   "zoned_fused"            0: the zone-aware packers' independent batches (gf_fit_batch, gf_fit_feasible) take the four-kernel
 *                            route (fit_zoned_kernel, avg_efficiency_kernel, zone_select_kernel, translate) instead of the one
 *                            launch of fit_zoned_fused_kernel (the default since round 5; views inherit the setting)
 *   "worker_sets", "worker_blocks_per_set", "worker_idle_us"   geometry and idle time of the resident worker (0 = the library's choice)
 *   "snapshot_finalize_host" 1: gf_snapshot_build* builds the slot tables through gf_orders_set on the host
 *   "force_general_layout"   1: gf_orders_set never merges the two orders into one slot order
 *   "sort_fault"             1: fault injection — the priority sort's grid barrier cannot complete; gf_snapshot_build* then
 *                            returns GF_ERR_HIP ("the priority sort's grid barrier gave up") instead of installing anything
 *   "rccl_selftest"          n: binds librccl at run time and checks a one-rank all-gather + reduce of n words on this device
 * GF_ERR_INVALID for an unknown key or a value out of range.  The only environment variables the library reads are
 * GANGFIT_WAIT=block (completion waits park the thread instead of polling) and GANGFIT_CHAIN_CACHE=0.  It never changes
 * the process environment: a deployment that runs chains concurrently (gf_ctx_view) sets GPU_MAX_HW_QUEUES=16 itself
 * before the process starts ("Deployment") — gf_ctx_view leaves a note in gf_last_error when it finds
 * fewer than 8. */
int gf_set_option(gf_ctx *ctx, const char *key, int64_t value);

/* Upload the per-node snapshot = the AvailableResources / SchedulableResources columns of
 * resources.NodeGroupSchedulingMetadata (LIB/resources/resources.go:61-100, 158-166).
 * avail_*: n_nodes values each (may be negative: overcommitted node).  sched_*: nullable (only needed by
 * gf_packing_efficiency). */
int gf_snapshot_set(gf_ctx *ctx, uint32_t n_nodes, const int64_t *avail_cpu_milli, const int64_t *avail_mem_bytes,
                    const int64_t *avail_gpu, const int64_t *sched_cpu_milli, const int64_t *sched_mem_bytes,
                    const int64_t *sched_gpu);

/* ---- the step BEFORE the decisions, on the device ----
 * Builds the snapshot from the flat cluster state and installs it (the equivalent of gf_snapshot_set + gf_zones_set +
 * gf_orders_set on the result):
 *   UsageForNodes — the ResourceReservation replay (LIB/resources/resources.go:31-43): res_* hold one entry per
 *     reservation of every ResourceReservation (and per soft reservation, resourcereservations.go:258-263) as
 *     (node index, cpu milli, memory bytes, gpus); entries whose node index is >= n_nodes are ignored;
 *   NodeSchedulingMetadataForNodes (resources.go:61-100): available = allocatable - (usage + overhead),
 *     schedulable = allocatable - overhead;
 *   NodeSorter.PotentialNodes (internal/sort/nodesorting.go:41-122): zones by free (memory, cpu) ascending — ties keep
 *     zone-id order, so assign ids in label order —, nodes by (zone rank, free memory, free cpu, name) with
 *     name_rank[n] = rank of node n's name in lexicographic order (a permutation of 0..n_nodes-1); driver candidates =
 *     nodes flagged GF_NODE_DRIVER_CANDIDATE (the Filter request's NodeNames), executor candidates = ready and not
 *     unschedulable; *_label_rank (nullable): rank of the node's value of the configured priority label, UINT32_MAX for
 *     "not ranked" — a stable re-sort of the respective list (:161-199).
 * driver_order_out / exec_order_out (nullable, room for n_nodes entries each) receive the two orders. */
#define GF_NODE_UNSCHEDULABLE 1u
#define GF_NODE_READY 2u
#define GF_NODE_DRIVER_CANDIDATE 4u
int gf_snapshot_build(gf_ctx *ctx, uint32_t n_nodes, const int64_t *alloc_cpu_milli, const int64_t *alloc_mem_bytes,
                      const int64_t *alloc_gpu, const int64_t *over_cpu_milli, const int64_t *over_mem_bytes,
                      const int64_t *over_gpu, uint32_t n_res, const uint32_t *res_node, const int64_t *res_cpu_milli,
                      const int64_t *res_mem_bytes, const int64_t *res_gpu, const uint32_t *node_flags,
                      const uint32_t *zone_of_node, uint32_t n_zones, const uint32_t *name_rank,
                      const uint32_t *driver_label_rank, const uint32_t *exec_label_rank, uint32_t *driver_order_out,
                      uint32_t *n_d_out, uint32_t *exec_order_out, uint32_t *n_x_out);
/* The same in two steps, for hosts that keep the cluster resident: gf_cluster_set uploads the columns that change only when
 * the node set does (allocatable, overhead, zone ids, name ranks, default node flags — an informer event in the Go host);
 * gf_snapshot_build_resident then builds and installs a snapshot from them and this request's reservation entries, so that
 * a Filter moves only the reservations (and, when node_flags is not NULL, its own candidate flags: the driver candidates
 * are the request's NodeNames) across PCIe.  gf_snapshot_build is exactly gf_cluster_set followed by
 * gf_snapshot_build_resident(node_flags = NULL). */
int gf_cluster_set(gf_ctx *ctx, uint32_t n_nodes, const int64_t *alloc_cpu_milli, const int64_t *alloc_mem_bytes,
                   const int64_t *alloc_gpu, const int64_t *over_cpu_milli, const int64_t *over_mem_bytes,
                   const int64_t *over_gpu, const uint32_t *node_flags, const uint32_t *zone_of_node, uint32_t n_zones,
                   const uint32_t *name_rank);
int gf_snapshot_build_resident(gf_ctx *ctx, uint32_t n_res, const uint32_t *res_node, const int64_t *res_cpu_milli,
                               const int64_t *res_mem_bytes, const int64_t *res_gpu, const uint32_t *node_flags,
                               const uint32_t *driver_label_rank, const uint32_t *exec_label_rank,
                               uint32_t *driver_order_out, uint32_t *n_d_out, uint32_t *exec_order_out, uint32_t *n_x_out);
/* The third step of keeping the snapshot resident ("keep snapshot resident on device across requests and
 * apply deltas"): the per-node usage sums of UsageForNodes (LIB/resources/resources.go:31-43) stay on the device next to the
 * cluster columns.  gf_cluster_set zeroes them (a new node set starts from nothing); gf_usage_apply adds (sign = +1: a
 * ResourceReservation or soft reservation appeared) or subtracts (sign = -1: it went away) the entries that changed since the
 * last call — an informer event in the Go host, K + 1 entries per application instead of every entry per Filter;
 * gf_snapshot_build_resident with n_res = GF_RESIDENT_USAGE then builds from the resident sums and moves no reservation at
 * all.  Sums of 64-bit integers: the result equals the replay of the full entry list bit for bit, whatever the order of the
 * updates.  Entries on nodes outside the cluster are ignored like the replay ignores them (resources.go:72). */
/* gf_usage_apply(sign = -1) returns GF_ERR_INVALID and leaves the sums as they were when an entry is removed from a node
 * that does not carry it (the node's sum would go negative: available above allocatable).  When an update fails half way
 * on a multi-device context the resident usage is unusable (GF_ERR_STATE) until gf_usage_reset. */
#define GF_RESIDENT_USAGE 0xFFFFFFFFu
int gf_usage_reset(gf_ctx *ctx);
int gf_usage_apply(gf_ctx *ctx, uint32_t n_entries, const uint32_t *res_node, const int64_t *res_cpu_milli,
                   const int64_t *res_mem_bytes, const int64_t *res_gpu, int sign /* +1 add, -1 remove */);
/* Generations of the state a context keeps between calls, for hosts that decide from them what a Filter must resend:
 *   out[0]  snapshot epoch: bumped by every call that installs a snapshot, zones or orders (the chain cache and recorded
 *           graphs are tied to it)
 *   out[1]  cluster generation: bumped by gf_cluster_set (and gf_snapshot_build, which calls it)
 *   out[2]  usage generation: bumped by gf_cluster_set, gf_usage_reset and every gf_usage_apply
 * A host that finds the generations it recorded after its own last call unchanged knows that no other user of the context
 * touched that state (internal/extender keeps one extender per context, but the UnschedulablePodMarker shares it). */
int gf_generation(gf_ctx *ctx, uint64_t out[3]);
/* The installed snapshot, n_nodes x 3 row-major each (either may be NULL). */
int gf_snapshot_get(gf_ctx *ctx, int64_t *avail_out, int64_t *sched_out);

/* Zone label of every node (NodeSchedulingMetadata.ZoneLabel, LIB/resources/resources.go:78-81, 158-166) as a dense id
 * the caller assigns per distinct label string.  Optional: without it every node is in one zone (the reference's
 * "default" label).  Call after gf_snapshot_set and before gf_orders_set (a new snapshot drops the zones). */
int gf_zones_set(gf_ctx *ctx, const uint32_t *zone_of_node /* n_nodes */);

/* Upload driverNodePriorityOrder / executorNodePriorityOrder (the two results of NodeSorter.PotentialNodes,
 * internal/sort/nodesorting.go:41-64) as node indices.  A known node may appear at most once in exec_order
 * (both vectors derive from map keys in the reference, nodesorting.go:153-159); violating this is GF_ERR_INVALID.
 * Must be called after gf_snapshot_set; a new snapshot invalidates the orders. */
int gf_orders_set(gf_ctx *ctx, const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order, uint32_t n_x);

/* The batched replacement of the BinpackFunc call sites (internal/extender/resource.go:238, :321,
 * internal/extender/unschedulablepods.go:156).
 *   results[n_apps]; exec_nodes[exec_nodes_cap] receives the concatenated ExecutorNodes (needs >= sum of k);
 *   chain_failed_at (nullable): FIFO mode — index of the first earlier driver that neither fit nor was skippable
 *   ("failure-earlier-driver", resource.go:249-251, 315-318), else -1.
 * Blocking.  Includes H2D of the app records and D2H of the results. */
int gf_fit_batch(gf_ctx *ctx, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app *apps, gf_result *results,
                 uint32_t *exec_nodes, uint64_t exec_nodes_cap, int32_t *chain_failed_at);

/* Feasibility of n_apps independent applications, nothing else: has_capacity[a] = PackingResult.HasCapacity of application a
 * against the installed snapshot (1 / 0).  What UnschedulablePodMarker reads — DoesPodExceedClusterCapacity returns
 * `!packingResult.HasCapacity` and drops the placements (internal/extender/unschedulablepods.go:132-166) —, batched over the
 * stale pending drivers of one scan (:93-129).  The same decision code as gf_fit_batch(GF_MODE_INDEPENDENT): the placements
 * are made and stay in device memory; one byte per application crosses the host link instead of a 16-byte result and 4 K
 * bytes of placements, and nothing is copied out but those bytes.  With mapped staging (the default) the bytes announce
 * themselves: each is preset to "not yet", one more workgroup of the kernel collects them in device memory and writes the whole array with system-scope
 * stores, and the call returns when the last byte has arrived — without the kernel-end write-back and the stream's completion
 * signal (a kernel that never answers is noticed after 2 ms through the stream).  No gf_scan_stats counters.  Blocking.  Every packer: the plain ones
 * through fit_independent_kernel's feasibility instantiation, the zone-aware ones through fit_zoned_fused_kernel's (one launch;
 * option "zoned_fused" = 0, more than 63 zones or missing schedulable columns: gf_fit_batch internally, of which only
 * HasCapacity is handed on — as for a multi-device context).  The zone-aware packers' answer is chooseBestResult's (single_az.go:75-97:
 * some zone fits AND its average Max efficiency is above 0); the averages are only computed where they could be 0 — a driver that asks
 * for neither cpu nor memory, or a snapshot in which some node's available quantity exceeds its schedulable one.  The kernel's
 * placements go to buffers no other entry point uses, so a call that follows on another stream never meets the tail of this one. */
int gf_fit_feasible(gf_ctx *ctx, gf_algo algo, uint32_t n_apps, const gf_app *apps, uint8_t *has_capacity);

/* Incremental FIFO chains.  The reference replays every earlier driver on every Filter (internal/extender/resource.go:309-328);
 * with an unchanged snapshot, driver j + 1's chain is driver j's chain plus one application.  gf_fit_batch(GF_MODE_FIFO_CHAIN)
 * therefore keeps the last queue, its results and a checkpoint of the working table every 32 applications (more for tables
 * whose 128 checkpoints would exceed 2 GiB), and a call whose queue starts with the same records resumes from the last
 * checkpoint inside the common prefix (the filtered driver of either queue excluded: nothing is committed behind it).
 * A chain on a table that fits LDS also keeps its TIP — the table before the application the chain ended at
 * (the driver being filtered, behind which nothing is committed, or the application it aborted at) —, and a queue that agrees
 * with the cached one up to there (the Filter of the next driver in creation order, the same Filter again) resumes from it:
 * one or two applications evaluated instead of everything since the last checkpoint, unless the chain crosses a checkpoint
 * boundary (it then starts from the checkpoint, so that the boundary's dump is made).
 * Every call that installs a snapshot, zones or orders (gf_snapshot_set, gf_zones_set, gf_orders_set, gf_snapshot_build*)
 * and gf_set_option drop the cache.  Results, placements, chain_failed_at and gf_residual_get are those of the full replay,
 * bit for bit: a checkpoint IS the table the replay holds at that application.  Served: every packer's LDS chain (plain,
 * zone-aware, minimal-fragmentation) on the merged layout with every request in scaled form; the wide / generic fallback
 * kernels replay.
 * out[0] = chains served with the cache armed, out[1] = of those resumed from a checkpoint, out[2] = applications
 * evaluated, out[3] = applications skipped (taken from the cache).  reset != 0 zeroes the counters afterwards. */
int gf_chain_cache_stats(gf_ctx *ctx, int reset, uint64_t out[4]);

/* Same decision kernels on DEVICE-resident buffers, asynchronous on `stream` (a hipStream_t; NULL = the context's own
 * stream).  d_apps must carry exec_off and must already be validated (k in [0, GF_MAX_K], requests in [0, 2^62)).
 * exec_nodes_len = number of uint32 entries in d_exec_nodes (>= sum of k).  d_chain_failed_at: device int32,
 * nullable (the result is then only kept inside the context).
 * Used by callers that keep the pending-app table resident (bench.py times this entry point). */
int gf_fit_batch_dev(gf_ctx *ctx, gf_mode mode, gf_algo algo, uint32_t n_apps, const gf_app *d_apps,
                     gf_result *d_results, uint32_t *d_exec_nodes, uint64_t exec_nodes_len,
                     int32_t *d_chain_failed_at, void *stream);

/* Replayable launch sequences.  A 1 000-application batch occupies the device for a few microseconds — about what the
 * host needs to submit one kernel — so a caller that evaluates the same device-resident tables over and over (bench.py; a
 * host that re-checks its pending queue on every event) is bound by its own launch rate.  gf_graph_begin starts recording
 * the device work of every *_dev call made on `stream` by the calling thread (nothing executes), gf_graph_end turns the
 * recording into a graph; gf_graph_launch replays it — same kernels, same arguments, one submission.  `stream` must not be
 * the NULL stream (NULL = the context's own stream, which is fine).  Snapshot, orders and every buffer the recorded calls
 * named must stay as they are until the graph is destroyed; buffers must have reached their final size before recording
 * (run the sequence once eagerly first).  The statistics counters of gf_scan_stats must be off while recording. */
/* gf_graph_launch returns GF_ERR_STATE when a snapshot, zones or orders were installed after gf_graph_end (the recording
 * names buffers an install may have replaced): record the sequence again. */
int gf_graph_begin(gf_ctx *ctx, void *stream);
int gf_graph_end(gf_ctx *ctx, void *stream, void **graph_out);
int gf_graph_launch(gf_ctx *ctx, void *graph, void *stream);
void gf_graph_destroy(gf_ctx *ctx, void *graph);

/* One decision through the batched path: the literal shape of binpack.SparkBinPackFunction
 * (LIB/binpack/binpack.go:43-48) for registry entries "gpu-tightly-pack" / "gpu-distribute-evenly". */
int gf_spark_binpack(gf_ctx *ctx, gf_algo algo, const gf_app *app, gf_result *result, uint32_t *exec_nodes,
                     uint64_t exec_nodes_cap);

/* binpack.AvgPackingEfficiency (LIB/binpack/efficiency.go:25-31). */
typedef struct gf_avg_efficiency {
    double cpu, memory, gpu, max;
} gf_avg_efficiency;

/* ComputeAvgPackingEfficiency (LIB/binpack/efficiency.go:114-156) over nodeNames = [DriverNode] ++ ExecutorNodes of
 * each of n_apps results of an INDEPENDENT batch, duplicates counted, summed in slice order — the value
 * chooseBestResult compares (LIB/binpack/single_az.go:83-93) — computed on the device against the current snapshot
 * (needs the schedulable columns).  `algo` selects how the packer left the `reserved` map: the minimal-fragmentation
 * packers never write executor placements into it (minimal_fragmentation.go:59-91).  Infeasible results give zeros
 * (WorstAvgPackingEfficiency).  float64 values are bit-identical to the reference's. */
int gf_avg_packing_efficiency(gf_ctx *ctx, gf_algo algo, uint32_t n_apps, const gf_app *apps, const gf_result *results,
                              const uint32_t *exec_nodes, uint64_t exec_nodes_len, gf_avg_efficiency *out);

/* ComputePackingEfficiencies (LIB/binpack/efficiency.go:66-103): PackingResult.PackingEfficiencies of ONE result as a
 * dense n_nodes x 3 array {CPU, Memory, GPU} in node-index order (the Go map has one entry per metadata key). */
int gf_packing_efficiencies(gf_ctx *ctx, gf_algo algo, const gf_app *app, const gf_result *result,
                            const uint32_t *exec_nodes, double *eff_out /* n_nodes * 3 */);

/* Placing single executors: the reschedule / extra-executor path of the executor Filter
 * (internal/extender/resource.go:594-703), n_req independent requests against the current snapshot and the executor
 * order of gf_orders_set.
 *   exe        n_req x 3 executor requests (canonical units, >= 0)
 *   reserved   nullable, n_nodes x 3 (row-major, >= 0): subtracted from the snapshot's available resources first — the
 *              overhead the reference counts twice for nodes that already carry reservations (`usage.Add(overhead)`
 *              after NodeSchedulingMetadataForNodes already did, resource.go:640-643) for the first-fit
 *              loop; the overhead map GetNodeCapacities receives as reservedResources (:682) for the other variant
 *   minimal_fragmentation == 0: the first node of the order the executor fits (:658-662)
 *   minimal_fragmentation != 0: rescheduleExecutorWithMinimalFragmentation (:675-703); hosts_app (nullable) holds, per
 *              request, ceil(n_nodes / 32) words with bit n set when node n already hosts executors of the application
 *   node_out   n_req node indices, GF_NO_NODE = "not enough capacity to reschedule the executor" (failure-fit) */
int gf_executor_fit(gf_ctx *ctx, int minimal_fragmentation, uint32_t n_req, const int64_t *exe, const int64_t *reserved,
                    const uint32_t *hosts_app, uint32_t *node_out);
/* The same with the zone step of the executor Filter on the device: filterNodesToZone (internal/extender/resource.go:462-478,
 * applied at :606-632 when a single-AZ packer runs with should-schedule-dynamically-allocated-executors-in-same-az and
 * getCommonZoneForExecutorsApplication, :493-519, found ONE zone).  Replaces: the narrowing of availableNodes / nodeNames
 * before NodeSchedulingMetadataForNodes and PotentialNodes in rescheduleExecutor.
 *   node_zone  n_nodes zone ids of the label THIS path reads, topology.kubernetes.io/zone (not
 *              necessarily the label gf_zones_set describes); no entry may be GF_ANY_ZONE
 *   req_zone   n_req zone ids: request q only considers nodes n with node_zone[n] == req_zone[q]; GF_ANY_ZONE = every node
 *              (the application's pods span several zones, :628-630)
 *   both NULL  = gf_executor_fit.
 * Exactness: the reference filters BEFORE it sorts.  Filtering the installed executor order gives the same sequence as
 * sorting the filtered set whenever the order INSIDE a zone does not depend on the other zones: true for the reference's
 * order (AZ priority, then free memory, free cpu, name: nodesorting.go:95-122) when node_zone partitions the nodes like the
 * zones that order ranked (both labels carry the same value — every cluster the reference's tests build), and for any order
 * when every node is in one zone.  A host whose two labels disagree filters on its side and calls gf_executor_fit
 * (host/extender.cpp does: always exact). */
#define GF_ANY_ZONE 0xFFFFFFFFu
int gf_executor_fit_zoned(gf_ctx *ctx, int minimal_fragmentation, uint32_t n_req, const int64_t *exe, const int64_t *reserved,
                          const uint32_t *hosts_app, const uint32_t *node_zone, const uint32_t *req_zone, uint32_t *node_out);

/* findNodes of the failover reconciler (internal/extender/failover.go:412-436; call site :368): reserve space for k
 * executors of each of n_req stale applications by walking the executor order of gf_orders_set (the reconciler's
 * orderedNodes) — tightly-pack with no driver, PARTIAL results (fewer than k nodes is not an error, :369-372), and the
 * reference's over-add kept: the `reserved[n].Add(executorResources)` that fails the comparison is not taken back before
 * the `break` (:424-427), so the returned `reserved` map holds (placed(n) + 1) x exe for every node that was filled up and
 * placed(n) x exe for the node on which k was reached.
 *   chained != 0  the requests run in order and each one's `reserved` map is subtracted from the working table before the
 *                 next (`r.availableResources[instanceGroup].Sub(reservedResources)`, :159); gf_residual_get returns the
 *                 table afterwards.  chained == 0: n_req independent calls against the snapshot.
 *   exe           n_req x 3 executor requests (canonical units, >= 0); k: n_req counts in [0, GF_MAX_K]
 *   results       per request: executors placed and the last node the loop reached (GF_NO_NODE if none).  The caller
 *                 rebuilds `reserved` from them: every node of the order up to last_node gets (multiplicity in the
 *                 placement list + 1) x exe, except last_node itself when placed == k (no over-add there).
 *   exec_nodes    concatenated placements; request q owns [sum of k before q, + results[q].placed)
 *   reserved_adds nullable, n_req x n_nodes: the `reserved` map in units of exe (0 = no entry) */
typedef struct gf_find_result {
    uint32_t placed;
    uint32_t last_node;
} gf_find_result; /* 8 bytes */
int gf_find_nodes(gf_ctx *ctx, int chained, uint32_t n_req, const int64_t *exe, const int32_t *k, gf_find_result *results,
                  uint32_t *exec_nodes, uint64_t exec_nodes_cap, uint32_t *reserved_adds);

/* ---- node-range sharding of an INDEPENDENT batch across the GPUs of one box ----
 * One gf_ctx per GPU, each given the same snapshot and orders; gf_shard_set tells it which contiguous range of the
 * priority order it owns (ranges are cut on 64-slot boundaries of the merged driver/executor order; GF_ERR_UNSUPPORTED
 * when the two orders cannot be merged into one).  A batch is then evaluated in four device steps, with one small
 * exchange over RCCL/xGMI between consecutive steps (the caller owns the communicator; k8s-spark-scheduler_amd/gangfit/
 * sharded.py drives it with torch.distributed):
 *     gf_shard_partials_dev -> all-gather -> gf_shard_drivers_dev -> all-gather -> gf_shard_emit_dev
 *     -> all-reduce(SUM, uint32) of d_exec2 -> gf_shard_finish_dev
 * after which EVERY rank holds the complete results: the same gf_result / ExecutorNodes as gf_fit_batch_dev on one GPU.
 * Tightly-pack and distribute-evenly only; the FIFO chain does not shard (each commit must be visible to the next
 * scan): it runs as replicas.  All pointers are device pointers; calls are asynchronous on `stream`. */
typedef struct gf_shard_partial {
    int64_t cap_sum;   /* sum over the range of min(capacity, K) with nothing reserved; may stop early once >= 2K */
    int64_t fit_count; /* nodes of the range with capacity >= 1 (distribute-evenly pass 1); same early stop */
} gf_shard_partial;    /* 16 bytes */
typedef struct gf_shard_driver {
    uint32_t pos;      /* first position of the range that passes the driver-fit check (LIB/binpack/binpack.go:69) and
                          leaves room for the gang, GF_NO_NODE if none */
    int32_t d_cap;     /* change of cap_sum / fit_count when the driver is reserved there */
    int32_t d_fit;
    uint32_t reserved;
} gf_shard_driver;     /* 16 bytes */

int gf_shard_set(gf_ctx *ctx, uint32_t shard, uint32_t n_shards); /* after gf_orders_set; a new gf_orders_set keeps it */
int gf_shard_partials_dev(gf_ctx *ctx, gf_algo algo, uint32_t n_apps, const gf_app *d_apps, gf_shard_partial *d_out,
                          void *stream);
int gf_shard_drivers_dev(gf_ctx *ctx, gf_algo algo, uint32_t n_apps, const gf_app *d_apps,
                         const gf_shard_partial *d_all_partials /* [n_shards][n_apps] */, gf_shard_driver *d_out,
                         void *stream);
/* d_exec2: 2 * half uint32 (half >= sum of k); zeroed here; [0, half) receives this shard's slice of the placements as
 * node index + 1, [half, 2 * half) the capacities distribute-evenly needs for passes >= 2. */
int gf_shard_emit_dev(gf_ctx *ctx, gf_algo algo, uint32_t n_apps, const gf_app *d_apps,
                      const gf_shard_partial *d_all_partials, const gf_shard_driver *d_all_drivers /* [n_shards][n_apps] */,
                      gf_result *d_results, uint32_t *d_exec2, uint64_t half, void *stream);
/* after the all-reduce: d_exec2[0, half) becomes the concatenated ExecutorNodes of gf_fit_batch_dev */
int gf_shard_finish_dev(gf_ctx *ctx, gf_algo algo, uint32_t n_apps, const gf_app *d_apps,
                        const gf_shard_partial *d_all_partials, const gf_shard_driver *d_all_drivers,
                        const gf_result *d_results, uint32_t *d_exec2, uint64_t half, void *stream);

/* Working copy of the available table after the last GF_MODE_FIFO_CHAIN call (n_nodes x 3, row-major) — lets tests
 * compare the replayed residuals with availableNodesSchedulingMetadata after fitEarlierDrivers. */
int gf_residual_get(gf_ctx *ctx, int64_t *avail_out /* n_nodes*3 */);

/* Kernel-only timing helper: HIP events recorded on `stream` (NULL = the context's own stream) around whatever is
 * launched on that stream between begin and end.  gf_timer_end blocks until the end event has completed. */
int gf_timer_begin(gf_ctx *ctx, void *stream);
int gf_timer_end(gf_ctx *ctx, float *elapsed_ms);

/* Visited-slot counters, for honest "visited bytes" reporting next to the algorithmic bytes (the scans are lazy, like
 * the reference's loops: they stop once K executors are placed).  enable != 0 makes subsequent launches count
 * (one atomic pair per app); out[0] = executor-order slots whose capacity was evaluated, out[1] = driver-order
 * positions whose fit was evaluated, accumulated since the last reset; out[2] / out[3] = shader-clock cycles and
 * 100 MHz real-time ticks spent inside the last FIFO-chain kernel (their ratio is the effective shader clock);
 * out[4..9] = that kernel's shader cycles by phase (app staging, driver scan, executor scan, slow path, commit, spare).
 * out may be NULL. */
int gf_scan_stats(gf_ctx *ctx, int enable, int reset, uint64_t out[10]);

/* Measurement helper (tools/probe_c5_heads.py): what the last FIFO chain of a plain packer did besides its common path, taken by
 * the instrumented kernel variant (gf_scan_stats enabled before the chain).  out[0..4] = applications that left the common path,
 * by ending: [1] a request without a shape id, [2] rejected by the capacity bound, [3] no driver candidate, [4] the gang did not
 * fit behind its first driver candidate (rollback + the generic decision); out[5..9] = shader cycles spent in those endings, same
 * index; out[10] / out[11] = the HW_ID / XCC_ID hardware registers of the chain's controlling wavefront (which compute unit of
 * which XCD the one workgroup ran on).  Nothing in the reference corresponds to it. */
int gf_chain_profile(gf_ctx *ctx, uint64_t out[12]);

/* On-device self-test of the wave primitives (DPP prefix scan, exact clamped 64-bit division) against plain
 * reference code on n_cases adversarial inputs per lane.  *mismatches == 0 means pass. */
int gf_selftest(gf_ctx *ctx, uint64_t seed, uint32_t n_cases, uint32_t *mismatches);

/* Bandwidth probe: streams `bytes` (a multiple of 16; use far more than the 256 MiB of last-level cache) `iters` times and
 * reports what this device delivers, in GB/s — the measured figures the rooflines in bench.py quote next to the spec peak:
 *   *read_gb_per_s   read-only stream (the access pattern of the scans), bytes / time
 *   *copy_gb_per_s   copy between two buffers, (read + write) bytes / time
 * Either pointer may be NULL (that leg is skipped). */
int gf_hbm_probe(gf_ctx *ctx, uint64_t bytes, uint32_t iters, double *read_gb_per_s, double *copy_gb_per_s);

/* Launch-floor probe: `iters` back-to-back launches of an empty one-wavefront kernel on `stream` (NULL = the context's own
 * stream) between two HIP events; *us_per_launch = what one dependent launch costs on this stream when there is nothing
 * to compute — the floor a latency-bound batch kernel is measured against. */
int gf_launch_floor(gf_ctx *ctx, void *stream, uint32_t iters, float *us_per_launch);

/* ---- The resident worker of the independent batch.
 * A batch of GF_MODE_INDEPENDENT is ~n_apps wavefronts that each walk a short chain of dependent misses: as a kernel launch it
 * pays the dispatch (~2.5 us) and cold caches, and two launches on one stream never overlap.  The worker is ONE launch that
 * stays on the device while batches keep coming: the host posts TICKETS into a ring in pinned memory and rings a doorbell word,
 * groups of wavefronts take the tickets in turn (several batches in flight), the wavefront that completes a ticket writes its
 * completion word to pinned memory.  No launch, no copy engine, no stream operation per batch.
 *   - it leaves the device by itself when no ticket has arrived for "worker_idle_us" (gf_set_option; default 200) and is
 *     launched again by the next submit; gf_worker_stop makes it leave at once (a host that wants to hipDeviceSynchronize);
 *   - it reads the installed snapshot: every install (gf_snapshot_* / gf_orders_set / gf_cluster_set ...) first serves what
 *     was posted and makes it leave;
 *   - plain packers only (tightly-pack, distribute-evenly, minimal-fragmentation); plain contexts only (no views, one
 *     device); results are bit-identical to gf_fit_batch(GF_MODE_INDEPENDENT) — same wave-level code.
 *   options: "worker_sets" (groups of wavefronts = batches in flight on the device, at most 16), "worker_blocks_per_set"
 *   (workgroups of sixteen wavefronts per group; each fills a CU), "worker_idle_us".  Both default to 0 = chosen at every
 *   launch of the worker from the first ticket it will serve: three applications per wavefront, one after the other, and as
 *   many sets as fit while sixteen CUs stay free for FIFO chains (1 000 applications: 11 sets of 21 workgroups);
 *   gf_worker_geometry reports what the last launch ran with.
 *
 * gf_worker_fit: one blocking batch with host arrays, like gf_fit_batch(GF_MODE_INDEPENDENT): the records are written into
 * a pinned slice the device reads in place, results and placements are written by the device into pinned memory (zero copy
 * both ways).
 *
 * gf_worker_submit_dev / gf_worker_wait: device-resident batches, asynchronously.  Tickets are numbered from 0 in posting
 * order; *first_ticket receives the number of batches[0].  A record array handed to the worker is read around the device's
 * L1 but through its L2: an array whose CONTENT is replaced while the worker is resident must be replaced by a stream
 * operation that completed before the submit (a finished copy or kernel), as for any kernel launch.  GF_WORKER_HOST_OUTPUTS:
 * d_results / d_exec_nodes are device addresses of PINNED HOST memory (no cache write-back before the completion word).
 * GF_WORKER_LEAVE_AFTER on the LAST batch of a submit: the caller has nothing more to post — a worker that this submit launches
 * (none was resident) serves what is posted and leaves the device by itself, without waiting to be told (gf_worker_stop /
 * gf_worker_wait behind it return as soon as the last answer is out; a bounded stream — K batches, then something else — saves the
 * two probes for a ticket that will not come).  Ignored when the worker is already resident. */
typedef struct gf_worker_batch {
    uint32_t n_apps;          /* > 0 */
    uint32_t flags;           /* GF_WORKER_* */
    const gf_app *d_apps;     /* exec_off filled by the caller, as for gf_fit_batch_dev */
    gf_result *d_results;
    uint32_t *d_exec_nodes;
    uint64_t exec_nodes_len;  /* sum of k over the batch */
} gf_worker_batch;
#define GF_WORKER_HOST_OUTPUTS 1u
#define GF_WORKER_LEAVE_AFTER 2u
int gf_worker_fit(gf_ctx *ctx, gf_algo algo, uint32_t n_apps, const gf_app *apps, gf_result *results, uint32_t *exec_nodes,
                  uint64_t exec_nodes_cap);
int gf_worker_submit_dev(gf_ctx *ctx, gf_algo algo, uint32_t n_batches, const gf_worker_batch *batches, uint64_t *first_ticket);
int gf_worker_wait(gf_ctx *ctx, uint64_t first_ticket, uint32_t n_tickets);
int gf_worker_stop(gf_ctx *ctx);
/* out[0] tickets posted, [1] tickets known complete (a prefix), [2] launches of the worker so far, [3] 1 = resident now */
int gf_worker_stats(gf_ctx *ctx, uint64_t out[4]);
/* out[0] sets, out[1] workgroups per set of the worker's last launch (0 0 before the first) */
int gf_worker_geometry(gf_ctx *ctx, uint32_t out[2]);
/* Measurement helper: where the last blocking gf_fit_batch(GF_MODE_INDEPENDENT) of a plain packer on the zero-copy path spent
 * its time on the host's clock, in microseconds: [0] validation + staging of the records into pinned memory, [1] the launch
 * call, [2] the wait for the stream (dispatch, the kernel — which reads the records from and writes the answers to pinned host
 * memory —, its completion signal), [3] copying results and placements to the
 * caller's arrays, [4] the whole call. */
int gf_call_phases(gf_ctx *ctx, double out_us[5]);
/* Measurement helper (bench.py's roofline): HIP events on the worker's own stream around its launch.  *ms = how long the
 * last FINISHED launch of the worker stayed on the device, *tickets = the tickets it relayed: ms / tickets is the device
 * time per 1 000-application batch in a stream of batches.  GF_ERR_STATE while no launch has finished (gf_worker_stop
 * makes the resident one leave).  Nothing in the reference corresponds to it. */
int gf_worker_kernel_time(gf_ctx *ctx, float *ms, uint64_t *tickets);

/* Device properties the host uses to size launches (also lets a caller verify it is talking to a gfx950). */
typedef struct gf_device_info {
    char name[128];
    char arch[64];
    int32_t compute_units;
    int32_t lds_bytes_per_cu;
    int32_t wavefront_size;
    int32_t clock_khz;
    int64_t hbm_bytes;
} gf_device_info;
int gf_device_info_get(gf_ctx *ctx, gf_device_info *out);

#ifdef __cplusplus
}
#endif
#endif /* GANGFIT_H */
