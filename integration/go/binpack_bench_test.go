// Drop into /root/reference/internal/binpacker/ and run on a machine with Go 1.19:
//   go test -mod=vendor -bench . -benchtime 20x ./internal/binpacker/
// Measures the TRUE Go number for BASELINE.json's metric (gang-fit decisions/sec at 10k nodes x 1k pending apps) on the
// same synthetic distributions as k8s-spark-scheduler_amd/gangfit/workloads.py::headline (splitmix64 streams; the
// generator below restates them).  UNVERIFIED in the build container: there is no Go toolchain there.
package binpacker

import (
	"context"
	"fmt"
	"math"
	"sort"
	"testing"

	"github.com/palantir/k8s-spark-scheduler-lib/pkg/resources"
)

const gib = int64(1) << 30

func splitmix(seed uint64, stream uint64, n int) []uint64 {
	out := make([]uint64, n)
	base := seed + stream*0xD1B54A32D192ED03
	for i := 0; i < n; i++ {
		z := base + 0x9E3779B97F4A7C15*uint64(i+1)
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
		z = (z ^ (z >> 27)) * 0x94D049BB133111EB
		out[i] = z ^ (z >> 31)
	}
	return out
}

func u01(v uint64) float64 { return float64(v>>11) / float64(uint64(1)<<53) }

type app struct {
	drv, exe *resources.Resources
	k        int
}

func workload(nNodes, nApps int, seed uint64, lo, hi float64) (resources.NodeGroupSchedulingMetadata, []string, []app) {
	cpuShapes := []int64{16000, 32000, 64000, 96000}
	memShapes := []int64{64 * gib, 128 * gib, 256 * gib, 384 * gib}
	s1, s2, s3, s4, s5, s6 := splitmix(seed, 1, nNodes), splitmix(seed, 2, nNodes), splitmix(seed, 3, nNodes),
		splitmix(seed, 4, nNodes), splitmix(seed, 5, nNodes), splitmix(seed, 6, nNodes)
	meta := make(resources.NodeGroupSchedulingMetadata, nNodes)
	type row struct {
		name     string
		cpu, mem int64
		idx      int
	}
	rows := make([]row, nNodes)
	for i := 0; i < nNodes; i++ {
		ac, am := cpuShapes[s1[i]%4], memShapes[s1[i]%4]
		ag := int64(0)
		if u01(s2[i]) < 0.10 {
			ag = 8
		}
		uc := int64(math.Floor((lo+(hi-lo)*u01(s3[i]))*float64(ac)/100.0)) * 100
		um := int64(math.Floor((lo+(hi-lo)*u01(s4[i]))*float64(am)/float64(256<<20))) * (256 << 20)
		ug := int64(math.Floor((lo + (hi-lo)*u01(s5[i])) * float64(ag)))
		if u01(s6[i]) < 0.01 {
			uc, um = ac+500, am+gib
		}
		name := fmt.Sprintf("node-%06d", i)
		m := resources.CreateSchedulingMetadataWithTotals((ac-uc+999)/1000, am-um, ag-ug, ac/1000, am, ag, "default")
		// cpu in milli-cores: overwrite with exact milli quantities
		m.AvailableResources.CPU.SetMilli(ac - uc)
		m.SchedulableResources.CPU.SetMilli(ac)
		meta[name] = m
		rows[i] = row{name, ac - uc, am - um, i}
	}
	sort.Slice(rows, func(a, b int) bool { // single zone: free memory asc, free cpu asc, name (nodesorting.go:74-122)
		if rows[a].mem != rows[b].mem {
			return rows[a].mem < rows[b].mem
		}
		if rows[a].cpu != rows[b].cpu {
			return rows[a].cpu < rows[b].cpu
		}
		return rows[a].idx < rows[b].idx
	})
	order := make([]string, nNodes)
	for i, r := range rows {
		order[i] = r.name
	}
	a11, a12, a13, a14, a15, a16 := splitmix(seed, 11, nApps), splitmix(seed, 12, nApps), splitmix(seed, 13, nApps),
		splitmix(seed, 14, nApps), splitmix(seed, 15, nApps), splitmix(seed, 16, nApps)
	apps := make([]app, nApps)
	for i := range apps {
		d := resources.CreateResources(0, []int64{2 * gib, 4 * gib, 8 * gib}[a12[i]%3], 0)
		d.CPU.SetMilli([]int64{1000, 2000, 4000}[a11[i]%3])
		g := int64(0)
		if u01(a15[i]) < 0.05 {
			g = 1
		}
		e := resources.CreateResources(0, []int64{4 * gib, 8 * gib, 16 * gib, 32 * gib}[a14[i]%4], g)
		e.CPU.SetMilli([]int64{1000, 2000, 4000, 8000}[a13[i]%4])
		k := 1 + int(math.Floor(math.Log1p(-u01(a16[i]))/math.Log1p(-1.0/12.0)))
		if k > 512 {
			k = 512
		}
		apps[i] = app{d, e, k}
	}
	return meta, order, apps
}

func benchIndependent(b *testing.B, name string, lo, hi float64) {
	meta, order, apps := workload(10000, 1000, 0x5EED0010, lo, hi)
	f := SelectBinpacker(name).BinpackFunc
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		for _, a := range apps {
			f(context.Background(), a.drv, a.exe, a.k, order, order, meta)
		}
	}
	b.ReportMetric(float64(b.N*len(apps))/b.Elapsed().Seconds(), "decisions/s")
}

func BenchmarkTightlyPackNominal(b *testing.B)        { benchIndependent(b, "tightly-pack", 0.0, 0.9) }
func BenchmarkDistributeEvenlyNominal(b *testing.B)   { benchIndependent(b, "distribute-evenly", 0.0, 0.9) }
func BenchmarkTightlyPackCongested(b *testing.B)      { benchIndependent(b, "tightly-pack", 0.95, 1.0) }
func BenchmarkDistributeEvenlyCongested(b *testing.B) { benchIndependent(b, "distribute-evenly", 0.95, 1.0) }
