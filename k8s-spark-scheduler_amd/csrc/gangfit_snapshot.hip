// gangfit_snapshot.hip — building the scheduling snapshot on the device: the step BEFORE the gang-fit kernels.
//
//   UsageForNodes ("ResourceReservation replay")      LIB/resources/resources.go:31-43
//   NodeSchedulingMetadataForNodes                    LIB/resources/resources.go:61-100
//   getNodeNamesInPriorityOrder                       internal/sort/nodesorting.go:95-122
// Inputs are the flat columns a host already holds (allocatable, overhead, one (node, request) pair per reservation,
// zone ids, the lexicographic rank of every node name).  The reservation replay is a scatter-add (64-bit atomics; at
// 20 k reservations x 13 entries the table is L2 resident), available/schedulable are one fused elementwise pass that
// also accumulates the per-zone free resources, and the priority order (zone rank, free memory, free cpu, name) is an
// LSD sequence of three stable radix sorts over a permutation that starts in name order.  The radix sort itself is
// rocPRIM's (through hipCUB), the way a GEMM would come from rocBLAS; everything else is written here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "gangfit_device.h"

namespace gangfit {

namespace {

__global__ void usage_scatter_kernel(uint32_t n_res, uint32_t n_nodes, const uint32_t* __restrict__ res_node,
                                     const int64_t* __restrict__ r0, const int64_t* __restrict__ r1,
                                     const int64_t* __restrict__ r2, unsigned long long* __restrict__ usage) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_res) return;
    const uint32_t n = res_node[i];
    if (n >= n_nodes) return;  // a reservation on a node outside the listed ones is never read (resources.go:72)
    atomicAdd(&usage[n], (unsigned long long)r0[i]);
    atomicAdd(&usage[(size_t)n_nodes + n], (unsigned long long)r1[i]);
    atomicAdd(&usage[2 * (size_t)n_nodes + n], (unsigned long long)r2[i]);
}

// available = allocatable - (usage + overhead), schedulable = allocatable - overhead (resources.go:76, 89-90);
// zone sums of the available memory / cpu feed the AZ order (nodesorting.go:124-134).
constexpr uint32_t kZoneLdsMax = 512;  // zones whose sums are first combined in LDS (one global atomic per block and zone)

__global__ __launch_bounds__(256) void metadata_kernel(uint32_t n_nodes, const int64_t* __restrict__ alloc,
                                                       const int64_t* __restrict__ overhead,
                                                       const int64_t* __restrict__ usage, const uint32_t* __restrict__ zone,
                                                       uint32_t n_zones, int64_t* __restrict__ avail,
                                                       int64_t* __restrict__ sched,
                                                       unsigned long long* __restrict__ zone_sum) {
    __shared__ unsigned long long zacc[3 * kZoneLdsMax];  // memory | cpu | population per zone
    const bool in_lds = n_zones <= kZoneLdsMax;
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i < 3 * n_zones; i += blockDim.x) zacc[i] = 0ull;
        __syncthreads();
    }
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_nodes) {
        int64_t a[3];
        for (int j = 0; j < 3; ++j) {
            const size_t k = (size_t)j * n_nodes + n;
            const int64_t o = overhead != nullptr ? overhead[k] : 0;
            a[j] = alloc[k] - (usage[k] + o);
            avail[k] = a[j];
            sched[k] = alloc[k] - o;
        }
        const uint32_t z = zone[n];
        if (z < n_zones) {
            if (in_lds) {  // a few hundred thousand same-address global atomics would serialise
                atomicAdd(&zacc[z], (unsigned long long)a[1]);
                atomicAdd(&zacc[n_zones + z], (unsigned long long)a[0]);
                atomicAdd(&zacc[2 * n_zones + z], 1ull);
            } else {
                atomicAdd(&zone_sum[2 * (size_t)z], (unsigned long long)a[1]);      // memory
                atomicAdd(&zone_sum[2 * (size_t)z + 1], (unsigned long long)a[0]);  // cpu
                atomicAdd(&zone_sum[2 * (size_t)n_zones + z], 1ull);                // population
            }
        }
    }
    if (in_lds) {
        __syncthreads();
        for (uint32_t z = threadIdx.x; z < n_zones; z += blockDim.x)
            if (zacc[2 * n_zones + z] != 0ull) {
                atomicAdd(&zone_sum[2 * (size_t)z], zacc[z]);
                atomicAdd(&zone_sum[2 * (size_t)z + 1], zacc[n_zones + z]);
                atomicAdd(&zone_sum[2 * (size_t)n_zones + z], zacc[2 * n_zones + z]);
            }
    }
}

// resourcesLessThan over the zones (nodesorting.go:73-80, 102-104): memory, then cpu, ascending; ties keep the
// caller's zone-id order.  A handful of zones: one thread, insertion sort.
__global__ void zone_rank_kernel(uint32_t n_zones, const long long* __restrict__ zone_sum, uint32_t* __restrict__ order,
                                 uint32_t* __restrict__ rank) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (uint32_t z = 0; z < n_zones; ++z) order[z] = z;
    for (uint32_t i = 1; i < n_zones; ++i) {
        const uint32_t z = order[i];
        const long long m = zone_sum[2 * (size_t)z], c = zone_sum[2 * (size_t)z + 1];
        uint32_t j = i;
        while (j > 0) {
            const uint32_t p = order[j - 1];
            const long long pm = zone_sum[2 * (size_t)p], pc = zone_sum[2 * (size_t)p + 1];
            const bool less = m != pm ? m < pm : c < pc;
            if (!less) break;
            order[j] = p;
            --j;
        }
        order[j] = z;
    }
    for (uint32_t i = 0; i < n_zones; ++i) rank[order[i]] = i;
}

__global__ void name_order_kernel(uint32_t n_nodes, const uint32_t* __restrict__ name_rank, uint32_t* __restrict__ perm) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_nodes) perm[name_rank[n]] = n;  // name_rank is a permutation (validated by the host layer)
}

__global__ void gather_i64_kernel(uint32_t n, const uint32_t* __restrict__ perm, const int64_t* __restrict__ col,
                                  int64_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = col[perm[i]];
}

__global__ void gather_zone_rank_kernel(uint32_t n, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ zone,
                                        const uint32_t* __restrict__ zrank, uint32_t n_zones, int64_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t z = zone[perm[i]];
    keys[i] = z < n_zones ? (int64_t)zrank[z] : (int64_t)n_zones;
}

}  // namespace

size_t snapshot_sort_temp_bytes(uint32_t n_nodes) {
    size_t bytes = 0;
    int64_t* k = nullptr;
    uint32_t* v = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, v, v, (int)n_nodes, 0, 64, (hipStream_t) nullptr);
    return bytes;
}

hipError_t launch_snapshot_build(const SnapshotBuild& b, hipStream_t stream) {
    const uint32_t n = b.n_nodes;
    if (n == 0) return hipSuccess;
    hipError_t e;
    if ((e = hipMemsetAsync(b.d_usage, 0, 3 * (size_t)n * sizeof(int64_t), stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.d_zone_sum, 0, 3 * (size_t)(b.n_zones ? b.n_zones : 1) * sizeof(int64_t), stream)) != hipSuccess)
        return e;
    const dim3 block(256);
    if (b.n_res > 0)
        hipLaunchKernelGGL(usage_scatter_kernel, dim3((b.n_res + 255) / 256), block, 0, stream, b.n_res, n, b.d_res_node,
                           b.d_res_req, b.d_res_req + b.n_res, b.d_res_req + 2 * (size_t)b.n_res,
                           reinterpret_cast<unsigned long long*>(b.d_usage));
    const dim3 grid((n + 255) / 256);
    hipLaunchKernelGGL(metadata_kernel, grid, block, 0, stream, n, b.d_alloc, b.d_overhead, (const int64_t*)b.d_usage,
                       b.d_zone, b.n_zones, b.d_avail, b.d_sched, reinterpret_cast<unsigned long long*>(b.d_zone_sum));
    hipLaunchKernelGGL(zone_rank_kernel, dim3(1), dim3(64), 0, stream, b.n_zones, (const long long*)b.d_zone_sum,
                       b.d_zone_order, b.d_zone_rank);
    hipLaunchKernelGGL(name_order_kernel, grid, block, 0, stream, n, b.d_name_rank, b.d_perm_a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // LSD: name order -> stable by cpu -> stable by memory -> stable by zone rank
    uint32_t* in = b.d_perm_a;
    uint32_t* out = b.d_perm_b;
    size_t temp = b.temp_bytes;
    for (int pass = 0; pass < 3; ++pass) {
        if (pass < 2)
            hipLaunchKernelGGL(gather_i64_kernel, grid, block, 0, stream, n, in,
                               b.d_avail + (size_t)(pass == 0 ? 0 : 1) * n, b.d_keys_a);
        else
            hipLaunchKernelGGL(gather_zone_rank_kernel, grid, block, 0, stream, n, in, b.d_zone, b.d_zone_rank, b.n_zones,
                               b.d_keys_a);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        e = hipcub::DeviceRadixSort::SortPairs(b.d_temp, temp, b.d_keys_a, b.d_keys_b, in, out, (int)n, 0,
                                              pass < 2 ? 64 : 32, stream);
        if (e != hipSuccess) return e;
        uint32_t* t = in;
        in = out;
        out = t;
    }
    // three passes: the result sits in d_perm_b
    return hipGetLastError();
}

}  // namespace gangfit
