// Drop into /root/reference/internal/extender/ (package extender: it calls the unexported sparkResourceUsage and
// findNodes) next to copies of tests/golden/gangfit_golden_v1.json, gangfit_golden_v2.json and gangfit_golden_v3.json, then on a machine with
// Go 1.19:
//
//	go test -mod=vendor -run TestGangfitGolden ./internal/extender/              # diff the REAL packers against the fixtures
//	go test -mod=vendor -run TestGangfitGolden ./internal/extender/ -update      # rewrite the answers from the real packers
//
// What it closes: tests/golden/*.json are produced by oracle/gangfit_oracle.c — a restatement of this repository's
// code by the author of the HIP path.  Nothing in the build container can execute the reference (no Go toolchain there),
// so distributeExecutorsEvenly (LIB/binpack/distribute_evenly.go:34-73), the FIFO replay (internal/extender/resource.go:
// 224-262 with sparkResourceUsage, sparkpods.go:139-146) and findNodes (failover.go:412-436) are pinned by no reference
// test.  This test runs the reference's OWN functions on the fixtures' inputs and fails on the first difference; with
// -update it regenerates the fixtures, which tests/test_golden.py then holds the oracle and the HIP path to.
// UNVERIFIED in the build container.
//
// Fixture units: cpu in milli-cores, memory and gpu in whole units.  Node i is named n%05d, zone z is named z%d, an
// order entry >= n_nodes is a name that is not a key of the metadata map ("ghost-%d").
package extender

import (
	"context"
	"encoding/json"
	"flag"
	"fmt"
	"os"
	"reflect"
	"testing"

	"github.com/palantir/k8s-spark-scheduler-lib/pkg/binpack"
	"github.com/palantir/k8s-spark-scheduler-lib/pkg/resources"
	v1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"
)

var updateGolden = flag.Bool("update", false, "rewrite the golden answers from the reference's packers")

type goldenBatch struct {
	HasCapacity []int      `json:"has_capacity"`
	DriverNode  []uint32   `json:"driver_node"`
	ExecNodes   [][]uint32 `json:"exec_nodes"`
}

type goldenFifo struct {
	K           []int      `json:"k"`
	Exe         [][3]int64 `json:"exe,omitempty"` // v1 fixtures omit it: exe = max(exe, 1) per component
	FailedAt    int        `json:"failed_at"`
	HasCapacity []int      `json:"has_capacity"`
	Evaluated   []int      `json:"evaluated"`
	DriverNode  []uint32   `json:"driver_node"`
	ExecNodes   [][]uint32 `json:"exec_nodes"`
	AvailAfter  [][3]int64 `json:"avail_after"`
}

type goldenAnswer struct {
	Independent goldenBatch `json:"independent"`
	Fifo        goldenFifo  `json:"fifo"`
}

type goldenFindNodes struct {
	Order      []uint32   `json:"order"`
	K          []int      `json:"k"`
	Exe        [][3]int64 `json:"exe"`
	Placed     []int      `json:"placed"`
	ExecNodes  [][]uint32 `json:"exec_nodes"`
	Adds       [][]uint32 `json:"adds"` // reserved[n] in units of the executor request; 0 = no entry
	AvailAfter [][3]int64 `json:"avail_after"`
}

type goldenCase struct {
	Name      string                   `json:"name,omitempty"`
	Avail     [][3]int64               `json:"avail"`
	Sched     [][3]int64               `json:"sched"`
	Zone      []uint32                 `json:"zone"`
	D         []uint32                 `json:"D"`
	X         []uint32                 `json:"X"`
	Drv       [][3]int64               `json:"drv"`
	Exe       [][3]int64               `json:"exe"`
	K         []int                    `json:"k"`
	Flags     []uint32                 `json:"flags"`
	FifoK     []int                    `json:"fifo_k,omitempty"`
	FifoExe   [][3]int64               `json:"fifo_exe,omitempty"`
	Seed      int                      `json:"seed"`
	NNodes    int                      `json:"n_nodes"`
	Answers   map[string]*goldenAnswer `json:"answers"`
	FindNodes *goldenFindNodes         `json:"find_nodes,omitempty"`
}

type goldenFile struct {
	Generator string         `json:"generator"`
	Oracle    string         `json:"oracle"`
	Units     string         `json:"units,omitempty"`
	Algos     map[string]int `json:"algos"`
	Cases     []*goldenCase  `json:"cases"`
}

const noNode = uint32(0xFFFFFFFF)

func res3(v [3]int64) *resources.Resources {
	return &resources.Resources{
		CPU:       *resource.NewMilliQuantity(v[0], resource.DecimalSI),
		Memory:    *resource.NewQuantity(v[1], resource.BinarySI),
		NvidiaGPU: *resource.NewQuantity(v[2], resource.DecimalSI),
	}
}

func canon(r *resources.Resources) [3]int64 {
	return [3]int64{r.CPU.MilliValue(), r.Memory.Value(), r.NvidiaGPU.Value()}
}

func nodeName(i uint32, n int) string {
	if int(i) >= n {
		return fmt.Sprintf("ghost-%d", i)
	}
	return fmt.Sprintf("n%05d", i)
}

func names(ix []uint32, n int) []string {
	out := make([]string, len(ix))
	for i, v := range ix {
		out[i] = nodeName(v, n)
	}
	return out
}

func indices(t *testing.T, ns []string) []uint32 {
	out := make([]uint32, len(ns))
	for i, s := range ns {
		var v uint32
		if _, err := fmt.Sscanf(s, "n%05d", &v); err != nil {
			t.Fatalf("unexpected node name %q", s)
		}
		out[i] = v
	}
	return out
}

func metadata(c *goldenCase) resources.NodeGroupSchedulingMetadata {
	m := make(resources.NodeGroupSchedulingMetadata, len(c.Avail))
	for i := range c.Avail {
		m[nodeName(uint32(i), len(c.Avail))] = &resources.NodeSchedulingMetadata{
			AvailableResources:   res3(c.Avail[i]),
			SchedulableResources: res3(c.Sched[i]),
			ZoneLabel:            fmt.Sprintf("z%d", c.Zone[i]),
			Ready:                true,
		}
	}
	return m
}

// the packer behind each fixture name: the registry's entries (internal/binpacker/binpack.go:43-49) plus the zone-less
// minimal-fragmentation packer the library exports
var goldenPackers = map[string]binpack.SparkBinPackFunction{
	"tightly-pack":                    binpack.TightlyPack,
	"distribute-evenly":               binpack.DistributeEvenly,
	"minimal-fragmentation":           binpack.MinimalFragmentation,
	"az-aware-tightly-pack":           binpack.AzAwareTightlyPack,
	"single-az-tightly-pack":          binpack.SingleAZTightlyPack,
	"single-az-minimal-fragmentation": binpack.SingleAZMinimalFragmentation,
}

func runIndependent(t *testing.T, c *goldenCase, pack binpack.SparkBinPackFunction) goldenBatch {
	n := len(c.Avail)
	D, X := names(c.D, n), names(c.X, n)
	var out goldenBatch
	for a := range c.K {
		r := pack(context.Background(), res3(c.Drv[a]), res3(c.Exe[a]), c.K[a], D, X, metadata(c))
		if r.HasCapacity {
			out.HasCapacity = append(out.HasCapacity, 1)
			out.DriverNode = append(out.DriverNode, indices(t, []string{r.DriverNode})[0])
			out.ExecNodes = append(out.ExecNodes, indices(t, r.ExecutorNodes))
		} else {
			out.HasCapacity = append(out.HasCapacity, 0)
			out.DriverNode = append(out.DriverNode, noNode)
			out.ExecNodes = append(out.ExecNodes, []uint32{})
		}
	}
	return out
}

// fitEarlierDrivers + the final pack, literally (resource.go:224-262, 309-328): the earlier drivers in order, each
// feasible one's sparkResourceUsage subtracted with SubtractUsageIfExists; flags bit 0 = shouldSkipDriverFifo.
func runFifo(t *testing.T, c *goldenCase, want *goldenFifo, pack binpack.SparkBinPackFunction) goldenFifo {
	n := len(c.Avail)
	D, X := names(c.D, n), names(c.X, n)
	meta := metadata(c)
	out := goldenFifo{K: want.K, Exe: want.Exe, FailedAt: -1}
	exeOf := func(a int) [3]int64 {
		if len(want.Exe) > 0 {
			return want.Exe[a]
		}
		e := c.Exe[a]
		for j := range e {
			if e[j] < 1 {
				e[j] = 1
			}
		}
		return e
	}
	apps := len(c.K)
	aborted := false
	for a := 0; a < apps; a++ {
		if aborted {
			out.HasCapacity = append(out.HasCapacity, 0)
			out.Evaluated = append(out.Evaluated, 0)
			out.DriverNode = append(out.DriverNode, noNode)
			out.ExecNodes = append(out.ExecNodes, []uint32{})
			continue
		}
		drv, exe := res3(c.Drv[a]), res3(exeOf(a))
		r := pack(context.Background(), drv, exe, want.K[a], D, X, meta)
		out.Evaluated = append(out.Evaluated, 1)
		if !r.HasCapacity {
			out.HasCapacity = append(out.HasCapacity, 0)
			out.DriverNode = append(out.DriverNode, noNode)
			out.ExecNodes = append(out.ExecNodes, []uint32{})
			if a+1 < apps && c.Flags[a]&1 == 0 { // resource.go:249-251
				out.FailedAt = a
				aborted = true
			}
			continue
		}
		out.HasCapacity = append(out.HasCapacity, 1)
		out.DriverNode = append(out.DriverNode, indices(t, []string{r.DriverNode})[0])
		out.ExecNodes = append(out.ExecNodes, indices(t, r.ExecutorNodes))
		if a+1 < apps { // the driver being filtered is not subtracted (resource.go:321-328)
			meta.SubtractUsageIfExists(sparkResourceUsage(drv, exe, r.DriverNode, r.ExecutorNodes))
		}
	}
	for i := 0; i < n; i++ {
		out.AvailAfter = append(out.AvailAfter, canon(meta[nodeName(uint32(i), n)].AvailableResources))
	}
	return out
}

// the reconciler's loop over stale applications of one instance group (failover.go:132-160): findNodes, then
// availableResources.Sub(reservedResources)
func runFindNodes(t *testing.T, c *goldenCase, want *goldenFindNodes) goldenFindNodes {
	n := len(c.Avail)
	avail := make(resources.NodeGroupResources, n)
	for i := range c.Avail {
		avail[nodeName(uint32(i), n)] = res3(c.Avail[i])
	}
	ordered := make([]*v1.Node, len(want.Order))
	for i, ix := range want.Order {
		ordered[i] = &v1.Node{ObjectMeta: metav1.ObjectMeta{Name: nodeName(ix, n)}}
	}
	out := goldenFindNodes{Order: want.Order, K: want.K, Exe: want.Exe}
	for q := range want.K {
		exe := res3(want.Exe[q])
		got, reserved := findNodes(want.K[q], exe, avail, ordered)
		out.Placed = append(out.Placed, len(got))
		out.ExecNodes = append(out.ExecNodes, indices(t, got))
		adds := make([]uint32, n)
		for name, r := range reserved {
			ix := indices(t, []string{name})[0]
			e, v := want.Exe[q], canon(r)
			for j := 0; j < 3; j++ {
				if e[j] != 0 {
					adds[ix] = uint32(v[j] / e[j])
				}
			}
			if e == [3]int64{0, 0, 0} { // a zero request: the map cannot say how many adds there were; count placements
				for _, g := range got {
					if g == name {
						adds[ix]++
					}
				}
				if len(got) < want.K[q] || got[len(got)-1] != name {
					adds[ix]++
				}
			}
		}
		out.Adds = append(out.Adds, adds)
		avail.Sub(reserved) // failover.go:159
	}
	for i := 0; i < n; i++ {
		out.AvailAfter = append(out.AvailAfter, canon(avail[nodeName(uint32(i), n)]))
	}
	return out
}

func TestGangfitGolden(t *testing.T) {
	for _, file := range []string{"gangfit_golden_v1.json", "gangfit_golden_v2.json", "gangfit_golden_v3.json"} {
		raw, err := os.ReadFile(file)
		if err != nil {
			t.Fatalf("%s: %v (copy it from tests/golden/ of the gangfit repository)", file, err)
		}
		var g goldenFile
		if err := json.Unmarshal(raw, &g); err != nil {
			t.Fatalf("%s: %v", file, err)
		}
		for ci, c := range g.Cases {
			for name, want := range c.Answers {
				pack, ok := goldenPackers[name]
				if !ok {
					t.Fatalf("%s: unknown packer %q", file, name)
				}
				ind := runIndependent(t, c, pack)
				fifo := runFifo(t, c, &want.Fifo, pack)
				if *updateGolden {
					want.Independent, want.Fifo = ind, fifo
					continue
				}
				for a := range c.K { // placements of infeasible applications are unspecified on both sides
					if ind.HasCapacity[a] != want.Independent.HasCapacity[a] || ind.DriverNode[a] != want.Independent.DriverNode[a] ||
						(ind.HasCapacity[a] == 1 && !reflect.DeepEqual(ind.ExecNodes[a], want.Independent.ExecNodes[a])) {
						t.Errorf("%s case %d (%s) %s independent app %d: reference %v %v %v, fixture %v %v %v", file, ci, c.Name, name, a,
							ind.HasCapacity[a], ind.DriverNode[a], ind.ExecNodes[a], want.Independent.HasCapacity[a],
							want.Independent.DriverNode[a], want.Independent.ExecNodes[a])
					}
					if fifo.HasCapacity[a] != want.Fifo.HasCapacity[a] || fifo.Evaluated[a] != want.Fifo.Evaluated[a] ||
						fifo.DriverNode[a] != want.Fifo.DriverNode[a] ||
						(fifo.HasCapacity[a] == 1 && !reflect.DeepEqual(fifo.ExecNodes[a], want.Fifo.ExecNodes[a])) {
						t.Errorf("%s case %d (%s) %s fifo app %d differs from the fixture", file, ci, c.Name, name, a)
					}
				}
				if fifo.FailedAt != want.Fifo.FailedAt || !reflect.DeepEqual(fifo.AvailAfter, want.Fifo.AvailAfter) {
					t.Errorf("%s case %d (%s) %s fifo: failed_at %d vs %d, residuals equal: %v", file, ci, c.Name, name, fifo.FailedAt,
						want.Fifo.FailedAt, reflect.DeepEqual(fifo.AvailAfter, want.Fifo.AvailAfter))
				}
			}
			if c.FindNodes != nil {
				fn := runFindNodes(t, c, c.FindNodes)
				if *updateGolden {
					c.FindNodes = &fn
				} else if !reflect.DeepEqual(fn.Placed, c.FindNodes.Placed) || !reflect.DeepEqual(fn.ExecNodes, c.FindNodes.ExecNodes) ||
					!reflect.DeepEqual(fn.Adds, c.FindNodes.Adds) || !reflect.DeepEqual(fn.AvailAfter, c.FindNodes.AvailAfter) {
					t.Errorf("%s case %d (%s) findNodes differs from the fixture", file, ci, c.Name)
				}
			}
		}
		if *updateGolden {
			g.Oracle = "the reference's own packers (integration/go/golden_test.go -update)"
			out, err := json.Marshal(&g)
			if err != nil {
				t.Fatal(err)
			}
			if err := os.WriteFile(file, out, 0o644); err != nil {
				t.Fatal(err)
			}
		}
	}
}
