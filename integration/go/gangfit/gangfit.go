// Package gangfit is the cgo binding of libgangfit (include/gangfit.h) that a maintainer of
// palantir/k8s-spark-scheduler drops into the reference tree as internal/gangfit/gangfit.go.
//
// NOT COMPILED IN THE BUILD CONTAINER (no Go toolchain there): this file is the reference-side stub that
// INTEGRATION.md describes.  Build with CGO_ENABLED=1 (the reference's dist config sets it to 0,
// godel/config/dist-plugin.yml:7,26) and -lgangfit on the linker path.
//
// cgo pointer rules: every slice handed to C is a flat []int64 / []uint32 / []C.gf_app without Go pointers; the
// library copies inputs before returning and writes only into caller-allocated flat slices.
package gangfit

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -lgangfit
#include <stdlib.h>
#include "gangfit.h"
*/
import "C"

import (
	"context"
	"errors"
	"fmt"
	"sync"
	"unsafe"

	"github.com/palantir/k8s-spark-scheduler-lib/pkg/binpack"
	"github.com/palantir/k8s-spark-scheduler-lib/pkg/resources"
	"k8s.io/apimachinery/pkg/api/resource"
)

// ErrNotRepresentable makes the caller fall back to the Go packer: the value is not an exact canonical int64
// (sub-milli cpu, fractional bytes, |v| >= 2^62).  MilliValue()/Value() round UP (quantity.go:732-755), hence the
// explicit round-trip checks.
var ErrNotRepresentable = errors.New("gangfit: quantity is not exactly representable as canonical int64")

const maxAbs = int64(1) << 62

func milli(q resource.Quantity) (int64, error) {
	v := q.MilliValue()
	if v >= maxAbs || v <= -maxAbs || resource.NewMilliQuantity(v, q.Format).Cmp(q) != 0 {
		return 0, ErrNotRepresentable
	}
	return v, nil
}

func whole(q resource.Quantity) (int64, error) {
	v := q.Value()
	if v >= maxAbs || v <= -maxAbs || resource.NewQuantity(v, q.Format).Cmp(q) != 0 {
		return 0, ErrNotRepresentable
	}
	return v, nil
}

func canonical(r *resources.Resources) (out [3]int64, err error) {
	if out[0], err = milli(r.CPU); err != nil {
		return
	}
	if out[1], err = whole(r.Memory); err != nil {
		return
	}
	out[2], err = whole(r.NvidiaGPU)
	return
}

// Context owns one gf_ctx.  Created right after SelectBinpacker (cmd/server.go:145), destroyed in the server's cleanup
// func.  One device id = one MI355X; several ids = ONE context over several GPUs of the box: independent batches of the
// plain packers are then node-range sharded inside the library (SURVEY.md 8e), nothing else changes for the caller.
type Context struct {
	mu    sync.Mutex // snapshot+zones+orders+fit must not interleave between goroutines (Predicate vs UnschedulablePodMarker)
	ctx   *C.gf_ctx
	names []string // node order of the last ClusterSet (resident flow)
	// the snapshot the last FilterResident installed (gf_generation right after it) and the flags it was built with
	builtEpoch, builtCluster, builtUsage uint64
	builtFlags                          []uint32
}

func New(devices ...int) (*Context, error) {
	if len(devices) == 0 {
		devices = []int{0}
	}
	ids := make([]C.int, len(devices))
	for i, d := range devices {
		ids[i] = C.int(d)
	}
	var h *C.gf_ctx
	if rc := C.gf_init(&ids[0], C.int(len(ids)), &h); rc != C.GF_OK {
		return nil, fmt.Errorf("gf_init: %d", int(rc))
	}
	return &Context{ctx: h}, nil
}

// zone-aware packers compare packing efficiencies per zone (single_az.go:75-97): they need the zone ids and exact
// schedulable columns; minimal-fragmentation packers never write executor placements into `reserved`
// (minimal_fragmentation.go:59-91).
func zoneAware(algo int) bool { return algo == 3 || algo == 4 || algo == 5 }
func reservesExecutors(algo int) bool { return algo != 2 && algo != 5 }

func (c *Context) Close() {
	if c.ctx != nil {
		C.gf_destroy(c.ctx)
		c.ctx = nil
	}
}

func (c *Context) err(rc C.int) error {
	return fmt.Errorf("libgangfit %d: %s", int(rc), C.GoString(C.gf_last_error(c.ctx)))
}

// App is one pending application (types.SparkApplicationResources + the FIFO skip flag).
type App struct {
	Driver, Executor *resources.Resources
	ExecutorCount    int  // MinExecutorCount (resource.go:242,325)
	Skippable        bool // shouldSkipDriverFifo(driver, instanceGroup) (resource.go:264-270)
}

// Result mirrors binpack.PackingResult without the efficiency map.
type Result struct {
	HasCapacity   bool
	DriverNode    string
	ExecutorNodes []string
	Evaluated     bool
}

// table flattens nodesSchedulingMetadata + the two priority orders into the dense-index form of the C ABI.
type table struct {
	names        []string
	index        map[string]uint32
	avail, sched [3][]int64
	schedOK      bool // every SchedulableResources is an exact canonical value >= 0
	zone         []uint32
	dOrder       []uint32
	xOrder       []uint32
}

func flatten(meta resources.NodeGroupSchedulingMetadata, driverOrder, execOrder []string) (*table, error) {
	t := &table{index: make(map[string]uint32, len(meta)), schedOK: true}
	zoneID := map[string]uint32{} // dense id per distinct NodeSchedulingMetadata.ZoneLabel (resources.go:78-81)
	for name, m := range meta {
		a, err := canonical(m.AvailableResources)
		if err != nil {
			return nil, err
		}
		var s [3]int64
		if m.SchedulableResources == nil {
			t.schedOK = false
		} else if s2, err := canonical(m.SchedulableResources); err != nil || s2[0] < 0 || s2[1] < 0 || s2[2] < 0 {
			// e.g. CreateSchedulingMetadata's math.MaxInt64 (resources.go:260-266): not below 2^62.  The plain packers do
			// not need the column (it is then passed as NULL); the zone-aware ones fall back to Go (FitBatch).
			t.schedOK = false
		} else {
			s = s2
		}
		z, ok := zoneID[m.ZoneLabel]
		if !ok {
			z = uint32(len(zoneID))
			zoneID[m.ZoneLabel] = z
		}
		t.index[name] = uint32(len(t.names))
		t.names = append(t.names, name)
		t.zone = append(t.zone, z)
		for j := 0; j < 3; j++ {
			t.avail[j] = append(t.avail[j], a[j])
			t.sched[j] = append(t.sched[j], s[j])
		}
	}
	unknown := uint32(len(t.names)) // a name that is not a key of the metadata map: index >= n_nodes
	conv := func(order []string) []uint32 {
		out := make([]uint32, len(order))
		for i, n := range order {
			if ix, ok := t.index[n]; ok {
				out[i] = ix
			} else {
				out[i] = unknown
				unknown++
			}
		}
		return out
	}
	t.dOrder, t.xOrder = conv(driverOrder), conv(execOrder)
	return t, nil
}

func p64(s []int64) *C.int64_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.int64_t)(unsafe.Pointer(&s[0]))
}
func p32(s []uint32) *C.uint32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&s[0]))
}

// FitBatch is the batched replacement of the BinpackFunc call sites: mode GF_MODE_FIFO_CHAIN for
// fitEarlierDrivers + the final pack (resource.go:309-328), GF_MODE_INDEPENDENT for the unschedulable-pod scan
// (unschedulablepods.go:93-166).  failedAt is the index of the earlier driver that aborts the chain or -1.
func (c *Context) FitBatch(fifo bool, algo int, apps []App, driverOrder, execOrder []string,
	meta resources.NodeGroupSchedulingMetadata) (results []Result, failedAt int, err error) {
	t, err := flatten(meta, driverOrder, execOrder)
	if err != nil {
		return nil, -1, err
	}
	if zoneAware(algo) && !t.schedOK {
		return nil, -1, ErrNotRepresentable // chooseBestResult compares efficiencies: they must be exact, so Go decides
	}
	capps := make([]C.gf_app, len(apps))
	total := 0
	for i, a := range apps {
		d, err := canonical(a.Driver)
		if err != nil {
			return nil, -1, err
		}
		e, err := canonical(a.Executor)
		if err != nil {
			return nil, -1, err
		}
		if a.ExecutorCount < 0 || a.ExecutorCount > C.GF_MAX_K {
			return nil, -1, ErrNotRepresentable
		}
		for j := 0; j < 3; j++ {
			capps[i].drv[j], capps[i].exe[j] = C.int64_t(d[j]), C.int64_t(e[j])
		}
		capps[i].k = C.int32_t(a.ExecutorCount)
		if a.Skippable {
			capps[i].flags = C.GF_APP_SKIPPABLE
		}
		total += a.ExecutorCount
	}
	cres := make([]C.gf_result, len(apps))
	execNodes := make([]uint32, total+1)
	var failed C.int32_t = -1
	mode := C.gf_mode(C.GF_MODE_INDEPENDENT)
	if fifo {
		mode = C.GF_MODE_FIFO_CHAIN
	}

	c.mu.Lock()
	defer c.mu.Unlock()
	var s0, s1, s2 *C.int64_t // NULL schedulable columns: only the efficiencies need them
	if t.schedOK {
		s0, s1, s2 = p64(t.sched[0]), p64(t.sched[1]), p64(t.sched[2])
	}
	if rc := C.gf_snapshot_set(c.ctx, C.uint32_t(len(t.names)), p64(t.avail[0]), p64(t.avail[1]), p64(t.avail[2]),
		s0, s1, s2); rc != C.GF_OK {
		return nil, -1, c.err(rc)
	}
	// NodeSchedulingMetadata.ZoneLabel as dense ids: without them every node would sit in ONE zone and the single-AZ /
	// AZ-aware packers would silently answer for a one-zone cluster (after gf_snapshot_set, before gf_orders_set)
	if len(t.zone) > 0 {
		if rc := C.gf_zones_set(c.ctx, p32(t.zone)); rc != C.GF_OK {
			return nil, -1, c.err(rc)
		}
	}
	if rc := C.gf_orders_set(c.ctx, p32(t.dOrder), C.uint32_t(len(t.dOrder)), p32(t.xOrder), C.uint32_t(len(t.xOrder))); rc != C.GF_OK {
		return nil, -1, c.err(rc)
	}
	var pa *C.gf_app
	var pr *C.gf_result
	if len(apps) > 0 {
		pa, pr = &capps[0], &cres[0]
	}
	if rc := C.gf_fit_batch(c.ctx, mode, C.gf_algo(algo), C.uint32_t(len(apps)), pa, pr, p32(execNodes),
		C.uint64_t(total), &failed); rc != C.GF_OK {
		return nil, -1, c.err(rc)
	}
	results = make([]Result, len(apps))
	off := 0
	for i := range apps {
		r := &results[i]
		r.Evaluated = cres[i].evaluated != 0
		r.HasCapacity = cres[i].has_capacity != 0
		r.ExecutorNodes = make([]string, 0, int(cres[i].exec_len))
		if r.HasCapacity {
			r.DriverNode = t.names[cres[i].driver_node]
			for _, ix := range execNodes[off : off+int(cres[i].exec_len)] {
				r.ExecutorNodes = append(r.ExecutorNodes, t.names[ix])
			}
		}
		off += apps[i].ExecutorCount
	}
	return results, int(failed), nil
}

// SparkBinPackFunction returns a binpack.SparkBinPackFunction (LIB/binpack/binpack.go:43-48) served by the device,
// with `fallback` (binpack.TightlyPack / binpack.DistributeEvenly) taking over on ANY accelerator error so that Filter
// semantics never change.  PackingEfficiencies are recomputed on the host exactly like binpack.go:77.
func (c *Context) SparkBinPackFunction(algo int, fallback binpack.SparkBinPackFunction) binpack.SparkBinPackFunction {
	return func(ctx context.Context, driverResources, executorResources *resources.Resources, executorCount int,
		driverNodePriorityOrder, executorNodePriorityOrder []string,
		nodesSchedulingMetadata resources.NodeGroupSchedulingMetadata) *binpack.PackingResult {
		res, _, err := c.FitBatch(false, algo, []App{{driverResources, executorResources, executorCount, false}},
			driverNodePriorityOrder, executorNodePriorityOrder, nodesSchedulingMetadata)
		if err != nil {
			return fallback(ctx, driverResources, executorResources, executorCount, driverNodePriorityOrder,
				executorNodePriorityOrder, nodesSchedulingMetadata)
		}
		if !res[0].HasCapacity {
			return binpack.EmptyPackingResult()
		}
		reserved := make(resources.NodeGroupResources, len(res[0].ExecutorNodes)+1)
		reserved[res[0].DriverNode] = driverResources.Copy()
		if reservesExecutors(algo) { // minimalFragmentation leaves `reserved` with the driver only
			for _, n := range res[0].ExecutorNodes {
				if reserved[n] == nil {
					reserved[n] = resources.Zero()
				}
				reserved[n].Add(executorResources)
			}
		}
		return &binpack.PackingResult{
			DriverNode:          res[0].DriverNode,
			ExecutorNodes:       res[0].ExecutorNodes,
			HasCapacity:         true,
			PackingEfficiencies: binpack.ComputePackingEfficiencies(nodesSchedulingMetadata, reserved),
		}
	}
}

// FindNodes is findNodes of the failover reconciler (internal/extender/failover.go:412-436) for the stale applications of
// ONE instance group, chained like the reconciler's loop: request i sees availableResources after
// `availableResources.Sub(reservedResources)` of the requests before it (:159).  Returns, per request, the executor node
// names (possibly fewer than asked for) and the `reserved` map — with the reference's over-add: a node that was filled up
// carries one executor request more than it hosts (:424-427).  available is NOT modified; the caller applies Sub itself.
func (c *Context) FindNodes(counts []int, executor []*resources.Resources, available resources.NodeGroupResources,
	orderedNodes []string) (nodes [][]string, reserved []resources.NodeGroupResources, err error) {
	n := len(orderedNodes)
	var cols [3][]int64
	order := make([]uint32, n)
	for i, name := range orderedNodes {
		r, ok := available[name]
		if !ok {
			return nil, nil, fmt.Errorf("gangfit: node %s has no availableResources entry", name)
		}
		a, err := canonical(r)
		if err != nil {
			return nil, nil, err
		}
		for j := 0; j < 3; j++ {
			cols[j] = append(cols[j], a[j])
		}
		order[i] = uint32(i)
	}
	exe := make([]int64, 0, 3*len(counts))
	ks := make([]C.int32_t, len(counts))
	total := 0
	for q, k := range counts {
		e, err := canonical(executor[q])
		if err != nil {
			return nil, nil, err
		}
		if k <= 0 || k > C.GF_MAX_K {
			return nil, nil, ErrNotRepresentable
		}
		exe = append(exe, e[0], e[1], e[2])
		ks[q] = C.int32_t(k)
		total += k
	}
	res := make([]C.gf_find_result, len(counts))
	placed := make([]uint32, total+1)
	c.mu.Lock()
	defer c.mu.Unlock()
	if rc := C.gf_snapshot_set(c.ctx, C.uint32_t(n), p64(cols[0]), p64(cols[1]), p64(cols[2]), nil, nil, nil); rc != C.GF_OK {
		return nil, nil, c.err(rc)
	}
	if rc := C.gf_orders_set(c.ctx, nil, 0, p32(order), C.uint32_t(n)); rc != C.GF_OK {
		return nil, nil, c.err(rc)
	}
	if len(counts) == 0 {
		return nil, nil, nil
	}
	if rc := C.gf_find_nodes(c.ctx, 1, C.uint32_t(len(counts)), p64(exe), &ks[0], &res[0], p32(placed), C.uint64_t(total), nil); rc != C.GF_OK {
		return nil, nil, c.err(rc)
	}
	off := 0
	for q, k := range counts {
		got := placed[off : off+int(res[q].placed)]
		names := make([]string, len(got))
		mult := map[uint32]int64{}
		for i, ix := range got {
			names[i] = orderedNodes[ix]
			mult[ix]++
		}
		// the `reserved` map as include/gangfit.h documents it: every node up to last_node was reached and carries one
		// failing add on top of its placements, except last_node itself when the count was reached there
		rmap := resources.NodeGroupResources{}
		if res[q].last_node != C.GF_NO_NODE {
			for ix := uint32(0); ix <= uint32(res[q].last_node); ix++ {
				adds := mult[ix] + 1
				if ix == uint32(res[q].last_node) && int(res[q].placed) == k {
					adds--
				}
				r := resources.Zero()
				for i := int64(0); i < adds; i++ {
					r.Add(executor[q])
				}
				rmap[orderedNodes[ix]] = r
			}
		}
		nodes = append(nodes, names)
		reserved = append(reserved, rmap)
		off += k
	}
	return nodes, reserved, nil
}

// UsageApply keeps UsageForNodes (LIB/resources/resources.go:31-43) resident on the device next to the cluster columns of
// gf_cluster_set: call it from the ResourceReservation / soft-reservation informer handlers with the entries of the object
// that appeared (add = true) or went away (add = false) — node INDICES in the order given to gf_cluster_set — and build the
// request's snapshot with gf_snapshot_build_resident(n_res = GF_RESIDENT_USAGE): no reservation list crosses PCIe per Filter.
// Unverified here (no Go toolchain); the C entry points are exercised by tests/test_snapshot_build.py.
func (c *Context) UsageApply(nodes []uint32, requests []*resources.Resources, add bool) error {
	if len(nodes) != len(requests) {
		return fmt.Errorf("gangfit: %d nodes, %d requests", len(nodes), len(requests))
	}
	if len(nodes) == 0 {
		return nil
	}
	var cols [3][]int64
	for _, r := range requests {
		v, err := canonical(r)
		if err != nil {
			return err
		}
		for j := 0; j < 3; j++ {
			cols[j] = append(cols[j], v[j])
		}
	}
	sign := C.int(1)
	if !add {
		sign = -1
	}
	c.mu.Lock()
	defer c.mu.Unlock()
	if rc := C.gf_usage_apply(c.ctx, C.uint32_t(len(nodes)), p32(nodes), p64(cols[0]), p64(cols[1]), p64(cols[2]), sign); rc != C.GF_OK {
		return c.err(rc)
	}
	return nil
}

// UsageReset zeroes the resident usage (gf_cluster_set does it too: a new node set starts from nothing).
func (c *Context) UsageReset() error {
	c.mu.Lock()
	defer c.mu.Unlock()
	if rc := C.gf_usage_reset(c.ctx); rc != C.GF_OK {
		return c.err(rc)
	}
	return nil
}

// ---- the resident flow (INTEGRATION.md, L-resident): the node columns and the usage sums live on the device, a Filter sends
// its candidate flags and its queue.  Unverified here (no Go toolchain); host/extender.cpp::selectDriverNodeFlat is the C++
// version of the same calls and is tested.

// Cluster is the node-side state of one instance group in a FIXED node order (index i everywhere below = Names[i]).
type Cluster struct {
	Names    []string
	Alloc    [3][]int64 // allocatable: cpu milli, memory bytes, gpus
	Overhead [3][]int64 // nil columns: no overhead
	Flags    []uint32   // GF_NODE_READY | GF_NODE_UNSCHEDULABLE | GF_NODE_DRIVER_CANDIDATE defaults
	Zone     []uint32   // dense zone ids (nil: one zone)
	NZones   int
	NameRank []uint32 // rank of Names[i] in lexicographic order (nodesorting.go:92: the name breaks ties)
}

// ResidentUsage as n_res of SnapshotBuildResident: build from the sums UsageApply maintains.
const ResidentUsage = ^uint32(0)

// ClusterSet uploads the node columns (gf_cluster_set); it also zeroes the resident usage.
func (c *Context) ClusterSet(cl *Cluster) error {
	c.mu.Lock()
	defer c.mu.Unlock()
	nz := cl.NZones
	if nz < 1 {
		nz = 1
	}
	if rc := C.gf_cluster_set(c.ctx, C.uint32_t(len(cl.Names)), p64(cl.Alloc[0]), p64(cl.Alloc[1]), p64(cl.Alloc[2]),
		p64(cl.Overhead[0]), p64(cl.Overhead[1]), p64(cl.Overhead[2]), p32(cl.Flags), p32(cl.Zone), C.uint32_t(nz),
		p32(cl.NameRank)); rc != C.GF_OK {
		return c.err(rc)
	}
	c.names = append(c.names[:0], cl.Names...)
	return nil
}

// SnapshotBuildResident builds and installs this request's snapshot from the resident state
// (gf_snapshot_build_resident with n_res = GF_RESIDENT_USAGE); requestFlags may be nil (the cluster's defaults).
// NOT atomic with a following fit: a Filter must use FilterResident, which holds the lock across both.
func (c *Context) SnapshotBuildResident(requestFlags []uint32) error {
	c.mu.Lock()
	defer c.mu.Unlock()
	return c.buildResidentLocked(requestFlags)
}

func (c *Context) buildResidentLocked(requestFlags []uint32) error {
	if rc := C.gf_snapshot_build_resident(c.ctx, C.uint32_t(ResidentUsage), nil, nil, nil, nil, p32(requestFlags), nil, nil,
		nil, nil, nil, nil); rc != C.GF_OK {
		return c.err(rc)
	}
	return nil
}

// FilterResident is one driver Filter of the resident flow: build this request's snapshot from the resident cluster and
// usage, then run the FIFO chain (or an independent batch) on it — under ONE hold of the context lock, so that a FitBatch,
// FindNodes or ClusterSet of another goroutine (the UnschedulablePodMarker runs next to Predicate, cmd/server.go:230) can
// neither replace the snapshot between the two steps nor change the name table the answer is mapped through.
//
// When nothing changed since this context's previous FilterResident — same cluster and usage generations (gf_generation),
// same candidate flags, nobody installed another snapshot — the rebuild is skipped, and the chain of
// gf_fit_batch(GF_MODE_FIFO_CHAIN) resumes from the previous chain's checkpoints (include/gangfit.h, "Incremental FIFO
// chains"): the Filter of driver j + 1 after the Filter of driver j costs a few dozen applications instead of all of them.
func (c *Context) FilterResident(requestFlags []uint32, fifo bool, algo int, apps []App) (results []Result, failedAt int, err error) {
	capps, total, err := flattenApps(apps)
	if err != nil {
		return nil, -1, err
	}
	c.mu.Lock()
	defer c.mu.Unlock()
	var gen [3]C.uint64_t
	C.gf_generation(c.ctx, &gen[0])
	same := c.builtEpoch != 0 && uint64(gen[0]) == c.builtEpoch && uint64(gen[1]) == c.builtCluster &&
		uint64(gen[2]) == c.builtUsage && equalFlags(c.builtFlags, requestFlags)
	if !same {
		c.builtEpoch = 0
		if err := c.buildResidentLocked(requestFlags); err != nil {
			return nil, -1, err
		}
		C.gf_generation(c.ctx, &gen[0])
		c.builtEpoch, c.builtCluster, c.builtUsage = uint64(gen[0]), uint64(gen[1]), uint64(gen[2])
		c.builtFlags = append(c.builtFlags[:0], requestFlags...)
	}
	return c.fitLocked(fifo, algo, apps, capps, total, c.names)
}

func equalFlags(a, b []uint32) bool {
	if len(a) != len(b) {
		return false
	}
	for i := range a {
		if a[i] != b[i] {
			return false
		}
	}
	return true
}

func flattenApps(apps []App) ([]C.gf_app, int, error) {
	capps := make([]C.gf_app, len(apps))
	total := 0
	for i, a := range apps {
		d, err := canonical(a.Driver)
		if err != nil {
			return nil, 0, err
		}
		e, err := canonical(a.Executor)
		if err != nil {
			return nil, 0, err
		}
		if a.ExecutorCount < 0 || a.ExecutorCount > C.GF_MAX_K {
			return nil, 0, ErrNotRepresentable
		}
		for j := 0; j < 3; j++ {
			capps[i].drv[j], capps[i].exe[j] = C.int64_t(d[j]), C.int64_t(e[j])
		}
		capps[i].k = C.int32_t(a.ExecutorCount)
		if a.Skippable {
			capps[i].flags = C.GF_APP_SKIPPABLE
		}
		total += a.ExecutorCount
	}
	return capps, total, nil
}

// fitLocked runs gf_fit_batch on the installed snapshot and maps node indices through `names`; c.mu is held.
func (c *Context) fitLocked(fifo bool, algo int, apps []App, capps []C.gf_app, total int, names []string) ([]Result, int, error) {
	return c.fitLockedVia(false, fifo, algo, apps, capps, total, names)
}

// fitLockedVia: viaWorker sends an INDEPENDENT batch of a plain packer through the resident worker (gf_worker_fit: no kernel
// launch, records and answers in pinned memory) instead of gf_fit_batch — same answers.
func (c *Context) fitLockedVia(viaWorker, fifo bool, algo int, apps []App, capps []C.gf_app, total int, names []string) ([]Result, int, error) {
	cres := make([]C.gf_result, len(apps))
	execNodes := make([]uint32, total+1)
	var failed C.int32_t = -1
	mode := C.gf_mode(C.GF_MODE_INDEPENDENT)
	if fifo {
		mode = C.GF_MODE_FIFO_CHAIN
	}
	var pa *C.gf_app
	var pr *C.gf_result
	if len(apps) > 0 {
		pa, pr = &capps[0], &cres[0]
	}
	if viaWorker && !fifo {
		if rc := C.gf_worker_fit(c.ctx, C.gf_algo(algo), C.uint32_t(len(apps)), pa, pr, p32(execNodes), C.uint64_t(total)); rc != C.GF_OK {
			return nil, -1, c.err(rc)
		}
	} else if rc := C.gf_fit_batch(c.ctx, mode, C.gf_algo(algo), C.uint32_t(len(apps)), pa, pr, p32(execNodes),
		C.uint64_t(total), &failed); rc != C.GF_OK {
		return nil, -1, c.err(rc)
	}
	results := make([]Result, len(apps))
	off := 0
	for i := range apps {
		r := &results[i]
		r.Evaluated = cres[i].evaluated != 0
		r.HasCapacity = cres[i].has_capacity != 0
		r.ExecutorNodes = make([]string, 0, int(cres[i].exec_len))
		if r.HasCapacity {
			if int(cres[i].driver_node) >= len(names) {
				return nil, -1, fmt.Errorf("gangfit: node index %d outside the %d names of the installed cluster", int(cres[i].driver_node), len(names))
			}
			r.DriverNode = names[cres[i].driver_node]
			for _, ix := range execNodes[off : off+int(cres[i].exec_len)] {
				if int(ix) >= len(names) {
					return nil, -1, fmt.Errorf("gangfit: node index %d outside the %d names of the installed cluster", int(ix), len(names))
				}
				r.ExecutorNodes = append(r.ExecutorNodes, names[ix])
			}
		}
		off += apps[i].ExecutorCount
	}
	return results, int(failed), nil
}

// FitBatchOnInstalledSnapshot is FitBatch without the snapshot upload: node names come from the last ClusterSet.  For a
// caller that brackets the install and the fit with its own mutual exclusion; a Filter should call FilterResident.
func (c *Context) FitBatchOnInstalledSnapshot(fifo bool, algo int, apps []App) (results []Result, failedAt int, err error) {
	capps, total, err := flattenApps(apps)
	if err != nil {
		return nil, -1, err
	}
	c.mu.Lock()
	defer c.mu.Unlock()
	return c.fitLocked(fifo, algo, apps, capps, total, c.names)
}

// FitIndependentOnWorker is FitBatchOnInstalledSnapshot(fifo = false) through the RESIDENT WORKER (include/gangfit.h,
// gf_worker_*): for a caller that sends one independent batch after the other — the unschedulable-pod scan over many
// instance groups' pods (unschedulablepods.go:93-166), a capacity what-if sweep.  Calls on one context serialise on c.mu;
// the worker stays on the device for gf_set_option("worker_idle_us") after a batch and is launched again by the next one
// that finds it gone, so back-to-back batches pay no kernel launch.  (Its throughput shows with several tickets in flight:
// gf_worker_submit_dev, device-resident batches.)  Plain packers only (algo 0, 1, 2); any install (ClusterSet, SnapshotBuildResident, FitBatch) first serves what was
// posted and makes the worker leave.  Unverified here (no Go toolchain); tests/test_gpu_worker.py drives the C entry points.
func (c *Context) FitIndependentOnWorker(algo int, apps []App) ([]Result, error) {
	capps, total, err := flattenApps(apps)
	if err != nil {
		return nil, err
	}
	c.mu.Lock()
	defer c.mu.Unlock()
	results, _, err := c.fitLockedVia(true, false, algo, apps, capps, total, c.names)
	return results, err
}

// FitFeasibleOnInstalledSnapshot answers DoesPodExceedClusterCapacity for a batch of stale pending drivers
// (internal/extender/unschedulablepods.go:93-166): HasCapacity of every application against the installed snapshot and
// nothing else (include/gangfit.h, gf_fit_feasible) — the same decisions as FitBatchOnInstalledSnapshot(fifo = false), one byte
// per application back over the host link, and the call returns when the bytes have arrived instead of waiting for the
// kernel's completion signal (one pod: ~11 us against ~18 us; 1 000 pods: ~17 us against ~24 us on an MI355X).  Every packer.
// Unverified here (no Go toolchain); tests/test_gpu_feasible.py drives the C entry point.
func (c *Context) FitFeasibleOnInstalledSnapshot(algo int, apps []App) ([]bool, error) {
	capps, _, err := flattenApps(apps)
	if err != nil {
		return nil, err
	}
	if len(apps) == 0 {
		return nil, nil
	}
	fits := make([]C.uint8_t, len(apps))
	c.mu.Lock()
	defer c.mu.Unlock()
	if rc := C.gf_fit_feasible(c.ctx, C.gf_algo(algo), C.uint32_t(len(apps)), &capps[0], &fits[0]); rc != C.GF_OK {
		return nil, c.err(rc)
	}
	out := make([]bool, len(apps))
	for i := range fits {
		out[i] = fits[i] != 0
	}
	return out, nil
}

// ExecutorFitOnInstalledSnapshot is the node choice of rescheduleExecutor (internal/extender/resource.go:594-673) for a batch
// of executors against the installed snapshot and executor order: the first-fit loop (:658-662) or, minimalFragmentation = true,
// rescheduleExecutorWithMinimalFragmentation (:675-703).  zones == nil: no zone step.  Otherwise nodeZone[n] is the id of node
// n's topology.kubernetes.io/zone label (NOT the label the snapshot's zones come from, SURVEY.md quirk 7) and zones[q] the id
// of the zone getCommonZoneForExecutorsApplication (:493-519) found for request q, or AnyZone when the application's running
// pods span several zones (:628-630): filterNodesToZone (:462-478) then happens on the device (gf_executor_fit_zoned; exact
// when both zone labels agree — include/gangfit.h — otherwise filter nodeNames in Go first and pass zones = nil).
// Returns node names, "" = "not enough capacity to reschedule the executor".  Unverified here (no Go toolchain);
// tests/test_executor_fit.py drives the C entry point.
const AnyZone = ^uint32(0)

func (c *Context) ExecutorFitOnInstalledSnapshot(minimalFragmentation bool, executors []*resources.Resources, nodeZone, zones []uint32) ([]string, error) {
	if len(executors) == 0 {
		return nil, nil
	}
	exe := make([]int64, 0, 3*len(executors))
	for _, r := range executors {
		v, err := canonical(r)
		if err != nil {
			return nil, err
		}
		exe = append(exe, v[0], v[1], v[2])
	}
	mf := C.int(0)
	if minimalFragmentation {
		mf = 1
	}
	out := make([]uint32, len(executors))
	c.mu.Lock()
	defer c.mu.Unlock()
	var rc C.int
	if zones == nil {
		rc = C.gf_executor_fit(c.ctx, mf, C.uint32_t(len(executors)), p64(exe), nil, nil, p32(out))
	} else {
		rc = C.gf_executor_fit_zoned(c.ctx, mf, C.uint32_t(len(executors)), p64(exe), nil, nil, p32(nodeZone), p32(zones), p32(out))
	}
	if rc != C.GF_OK {
		return nil, c.err(rc)
	}
	names := make([]string, len(out))
	for i, n := range out {
		if n != ^uint32(0) && int(n) < len(c.names) {
			names[i] = c.names[n]
		}
	}
	return names, nil
}

// WorkerStop makes the resident worker leave the device now (it leaves by itself when idle).
func (c *Context) WorkerStop() error {
	c.mu.Lock()
	defer c.mu.Unlock()
	if rc := C.gf_worker_stop(c.ctx); rc != C.GF_OK {
		return c.err(rc)
	}
	return nil
}

