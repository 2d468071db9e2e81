// gangfit_snapshot.hip — building the scheduling snapshot on the device: the step BEFORE the gang-fit kernels.
//
//   UsageForNodes ("ResourceReservation replay")      LIB/resources/resources.go:31-43
//   NodeSchedulingMetadataForNodes                    LIB/resources/resources.go:61-100
//   getNodeNamesInPriorityOrder                       internal/sort/nodesorting.go:95-122
// Inputs are the flat columns a host already holds (allocatable, overhead, one (node, request) pair per reservation,
// zone ids, the lexicographic rank of every node name).  The reservation replay is a scatter-add (64-bit atomics; at
// 20 k reservations x 13 entries the table is L2 resident), available/schedulable are one fused elementwise pass that
// also accumulates the per-zone free resources, and the priority order (zone rank, free memory, free cpu, name) is an
// stable LSD radix sort of ONE composite key per node over a permutation that starts in name order
// (priority_sort_kernel below: one workgroup, one launch; no sorting library).
#include <hip/hip_runtime.h>

#include "gangfit_device.h"

namespace gangfit {

namespace {

// sign = +1 / -1: two's-complement addition, so removing an entry is adding its negation
__global__ void usage_scatter_kernel(uint32_t n_res, uint32_t n_nodes, const uint32_t* __restrict__ res_node,
                                     const int64_t* __restrict__ r0, const int64_t* __restrict__ r1,
                                     const int64_t* __restrict__ r2, unsigned long long* __restrict__ usage, int sign = 1) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_res) return;
    const uint32_t n = res_node[i];
    if (n >= n_nodes) return;  // a reservation on a node outside the listed ones is never read (resources.go:72)
    atomicAdd(&usage[n], (unsigned long long)(r0[i] * sign));
    atomicAdd(&usage[(size_t)n_nodes + n], (unsigned long long)(r1[i] * sign));
    atomicAdd(&usage[2 * (size_t)n_nodes + n], (unsigned long long)(r2[i] * sign));
}

// available = allocatable - (usage + overhead), schedulable = allocatable - overhead (resources.go:76, 89-90);
// zone sums of the available memory / cpu feed the AZ order (nodesorting.go:124-134).
constexpr uint32_t kZoneLdsMax = 512;  // zones whose sums are first combined in LDS (one global atomic per block and zone)

// After a removal: a node of the update whose sum went negative had an entry removed that was never added there.
__global__ void usage_negative_kernel(uint32_t n_res, uint32_t n_nodes, const uint32_t* __restrict__ res_node,
                                      const int64_t* __restrict__ usage, uint32_t* __restrict__ negative) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_res) return;
    const uint32_t n = res_node[i];
    if (n >= n_nodes) return;
    if (usage[n] < 0 || usage[(size_t)n_nodes + n] < 0 || usage[2 * (size_t)n_nodes + n] < 0) atomicOr(negative, 1u);
}

// order-preserving map of a signed quantity to unsigned
__device__ __forceinline__ unsigned long long biased(int64_t v) { return (unsigned long long)v ^ 0x8000000000000000ull; }
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

// Also feeds the priority sort that follows (priority_sort_kernel): the permutation in name order (perm[name_rank[n]] = n) and
// the ranges of the two sort columns — sort_scal: max ~cpu | max cpu | max ~mem | max mem | OR (cpu - cpu[0]) | OR (mem - mem[0])
// over the biased free values (all zero at launch).
// Grid-stride over the nodes with at most kMetaBlocks workgroups, and ONE set of the six range atomics per workgroup: they are
// device-scope read-modify-writes of one cache line and queue behind each other — one set per wavefront (rounds 2-5a: 1 563
// wavefronts x 6 at 100 000 nodes) made this kernel 112 us of a 0.40 ms build, 15 us at 10 000 nodes.
constexpr uint32_t kMetaBlocks = 192;
__global__ __launch_bounds__(256) void metadata_kernel(uint32_t n_nodes, const int64_t* __restrict__ alloc,
                                                       const int64_t* __restrict__ overhead,
                                                       const int64_t* __restrict__ usage, const uint32_t* __restrict__ zone,
                                                       uint32_t n_zones, int64_t* __restrict__ avail,
                                                       int64_t* __restrict__ sched,
                                                       unsigned long long* __restrict__ zone_sum,
                                                       const uint32_t* __restrict__ name_rank, uint32_t* __restrict__ name_order,
                                                       unsigned long long* __restrict__ sort_scal) {
    __shared__ unsigned long long zacc[3 * kZoneLdsMax];  // memory | cpu | population per zone
    __shared__ unsigned long long red[6][4];              // the six range words per wavefront of the workgroup
    const bool in_lds = n_zones <= kZoneLdsMax;
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i < 3 * n_zones; i += blockDim.x) zacc[i] = 0ull;
        __syncthreads();
    }
    unsigned long long ncmn = 0ull, cmx = 0ull, nmmn = 0ull, mmx = 0ull, cor = 0ull, mor = 0ull;
    // node 0's free cpu / memory: the reference point of the common trailing zeros (every difference to the minimum
    // shares the trailing zeros of the differences to ANY one element)
    const int64_t o0 = overhead != nullptr ? overhead[0] : 0, o1 = overhead != nullptr ? overhead[n_nodes] : 0;
    const unsigned long long c0 = biased(alloc[0] - (usage[0] + o0)), m0 = biased(alloc[n_nodes] - (usage[n_nodes] + o1));
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < n_nodes; n += gridDim.x * blockDim.x) {
        int64_t a[3];
        for (int j = 0; j < 3; ++j) {
            const size_t k = (size_t)j * n_nodes + n;
            const int64_t o = overhead != nullptr ? overhead[k] : 0;
            a[j] = alloc[k] - (usage[k] + o);
            avail[k] = a[j];
            sched[k] = alloc[k] - o;
        }
        const unsigned long long c = biased(a[0]), m = biased(a[1]);
        ncmn = ~c > ncmn ? ~c : ncmn;
        cmx = c > cmx ? c : cmx;
        nmmn = ~m > nmmn ? ~m : nmmn;
        mmx = m > mmx ? m : mmx;
        cor |= c - c0;
        mor |= m - m0;
        name_order[name_rank[n]] = n;  // name_rank is a permutation (validated by the host layer)
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
    ncmn = wave_max_u64(ncmn);
    cmx = wave_max_u64(cmx);
    nmmn = wave_max_u64(nmmn);
    mmx = wave_max_u64(mmx);
    cor = wave_or_u64(cor);
    mor = wave_or_u64(mor);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) {
        red[0][wave] = ncmn;
        red[1][wave] = cmx;
        red[2][wave] = nmmn;
        red[3][wave] = mmx;
        red[4][wave] = cor;
        red[5][wave] = mor;
    }
    __syncthreads();
    // (a workgroup of the grid always holds nodes: the grid never has more workgroups than 256-node blocks; lanes without a
    //  node contribute zeros, the identity of max over biased values' complements and of OR)
    if (threadIdx.x < 6) {
        const uint32_t k = threadIdx.x, nw = blockDim.x >> 6;
        unsigned long long v = 0ull;
        for (uint32_t w = 0; w < nw; ++w) v = k < 4 ? (red[k][w] > v ? red[k][w] : v) : (v | red[k][w]);
        if (k < 4)
            atomicMax(&sort_scal[k], v);
        else if (v)
            atomicOr(&sort_scal[k], v);
    }
    if (in_lds) {
        for (uint32_t z = threadIdx.x; z < n_zones; z += blockDim.x)
            if (zacc[2 * n_zones + z] != 0ull) {
                atomicAdd(&zone_sum[2 * (size_t)z], zacc[z]);
                atomicAdd(&zone_sum[2 * (size_t)z + 1], zacc[n_zones + z]);
                atomicAdd(&zone_sum[2 * (size_t)n_zones + z], zacc[2 * n_zones + z]);
            }
    }
}

// Everything a build accumulates into starts at zero (the zone evaluation's first-slot table at all ones): ONE launch instead
// of five runtime fills of a few hundred bytes each — every one of them a kernel of its own on the stream.
struct SnapshotClear {
    uint32_t* zero[4];   // word ranges set to 0
    size_t zero_words[4];
    uint32_t* ones;      // word range set to 0xFFFFFFFF
    size_t ones_words;
};
__global__ __launch_bounds__(256) void snapshot_clear_kernel(SnapshotClear c) {
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x, stride = (size_t)gridDim.x * 256u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t* p = c.zero[r];
        const size_t n = c.zero_words[r];
        if (n == 0) continue;
        if (((uintptr_t)p & 15u) == 0) {  // the large ranges (count tables, the usage sums) are 16-byte aligned
            uint4* p4 = reinterpret_cast<uint4*>(p);
            for (size_t i = t; i < n / 4; i += stride) p4[i] = make_uint4(0, 0, 0, 0);
            for (size_t i = (n & ~(size_t)3) + t; i < n; i += stride) p[i] = 0u;
        } else {
            for (size_t i = t; i < n; i += stride) p[i] = 0u;
        }
    }
    for (size_t i = t; i < c.ones_words; i += stride) c.ones[i] = 0xFFFFFFFFu;
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

// ------------------------------------------------------------------------------------------------ the priority order
// getNodeNamesInPriorityOrder (internal/sort/nodesorting.go:95-122): nodes by (zone rank, free memory, free cpu, name), all
// ascending.  ONE launch of kSortWG single-wavefront workgroups that meet at a grid barrier:
//   1. the range of every column comes from metadata_kernel (min, max, common trailing zeros of the differences): a field
//      needs bits(max - min) bits, less the trailing zeros every difference shares (free memory is a multiple of hundreds
//      of MiB on real clusters: 39 bits shrink to about 11);
//   2. the fields are packed into ONE 64-bit key per node — zone rank | memory | cpu — when they fit (they do unless the
//      quantities are adversarial: then two or three keys are sorted one after the other, least significant first);
//   3. stable LSD radix sort of (key, node) pairs, 8 bits per pass, starting from the name order (the final tie-break).
//      Every wavefront owns a contiguous segment.  A pass: read the 64 x 256 digit counts, derive the segment's 256 output
//      offsets (digit-major exclusive scan, done redundantly by every wavefront: 64 KB of L2 reads instead of a barrier),
//      rank each element inside its 64-element chunk with a ballot-built peer mask (stable, no atomics on the offsets),
//      scatter — and count the NEXT pass's digit into the destination segment's row while the key is in a register
//      (order-free atomic adds).  A pass therefore costs ONE grid barrier, the key build (which counts the first digit) one,
//      and the last pass none: three rotating buffers are walked so that it lands in the output array.
// A wavefront alone on its SIMD issues at its full rate; 64 of them sort 100 000 nodes in a fraction of the millisecond one
// workgroup needed (860 us measured) and of the two dozen launches of a sorting library.  What is left is the barrier: the
// XCDs' L2s are not coherent with each other, so each one is a write-back, an invalidate and a round trip to memory.
constexpr uint32_t kSortWG = 64;          // workgroups = wavefronts
constexpr int kSortTile = 16;             // 64-element chunks a wavefront has in flight per step (independent loads)
constexpr int kSortKeyTile = 8;           // ... in the key build (a node id, then three gathers per element)
constexpr int kSortRowBatch = 32;         // rows of the count table requested together (behind a grid barrier every one of them
                                          // is a round trip to memory: eight at a time made a pass of 64 rows eight round trips)
constexpr uint32_t kSortHistWords = 3u * kSortWG * 256u;  // three rotating count tables [segment][digit]
constexpr uint32_t kSortStateWords = 4u;  // barrier count | barrier generation | error | spare
constexpr uint32_t kSortScalars = 16u;    // 64-bit words behind the state (the first six: metadata_kernel's ranges)

struct PrioritySort {
    uint32_t n, n_zones;
    const int64_t* cpu;        // free cpu by node
    const int64_t* mem;        // free memory by node
    const uint32_t* zone;      // zone id by node
    const uint32_t* zrank;     // zone id -> rank (zone_rank_kernel; only read when there are more than 64 zones)
    const long long* zone_sum; // free memory | free cpu per zone (metadata_kernel): up to 64 zones are ranked by the kernel itself
    unsigned long long* keys[3];
    uint32_t* perm[3];         // [0] holds the name order at launch, [1] receives the result: position -> node
    uint32_t* work;            // kSortHistWords + kSortStateWords uint32 + kSortScalars uint64; zero at launch but the ranges
    uint32_t spin_limit;       // probes of the barrier word before a wavefront gives up (2^24: about half a second)
};

__device__ __forceinline__ uint32_t bits_of(unsigned long long v) { return v ? 64u - (uint32_t)__clzll(v) : 0u; }

// Grid barrier of the kSortWG single-wavefront workgroups (sense by generation).  Everything a wavefront wrote before is
// visible to every wavefront behind it (agent-scope release / acquire: the XCDs' L2s are written back and invalidated).
// false = the others did not arrive in about half a second (a device saturated by someone else's endless kernel): the caller
// flags the error and leaves.
__device__ __forceinline__ bool sort_grid_sync(uint32_t* state, int lane, uint32_t spin_limit) {
    int ok = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) {
        const uint32_t gen = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t arrived = __hip_atomic_fetch_add(&state[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == kSortWG - 1u) {
            __hip_atomic_store(&state[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&state[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            uint32_t spins = 0;
            while (__hip_atomic_load(&state[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_limit) {
                    ok = 0;
                    break;
                }
            }
        }
    }
    ok = __shfl(ok, 0, 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok != 0;
}

// the peers of a lane: the valid lanes of the chunk that hold the same digit (eight ballots)
__device__ __forceinline__ unsigned long long digit_peers(uint32_t d, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

__global__ __launch_bounds__(64) void priority_sort_kernel(PrioritySort A) {
    __shared__ uint32_t off[256];  // this segment's next output position per digit
    __shared__ uint32_t zrank_l[64];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const uint32_t n = A.n;
    // resourcesLessThan over the zones (nodesorting.go:73-80, 102-104: memory, then cpu, ascending; ties keep the caller's zone-id
    // order): lane = zone, rank = the zones that sort before mine.  (A launch of its own — zone_rank_kernel — only beyond 64 zones.)
    const bool zr_local = A.n_zones <= 64u;
    if (zr_local) {
        long long zm = 0, zc = 0;
        if (lane < A.n_zones) {
            zm = A.zone_sum[2 * (size_t)lane];
            zc = A.zone_sum[2 * (size_t)lane + 1];
        }
        uint32_t r = 0;
        for (uint32_t y = 0; y < A.n_zones; ++y) {
            const long long ym = __shfl(zm, (int)y, 64), yc = __shfl(zc, (int)y, 64);
            r += (ym != zm ? ym < zm : (yc != zc ? yc < zc : y < lane)) ? 1u : 0u;
        }
        zrank_l[lane] = r;
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    uint32_t* const hist = A.work;
    uint32_t* const state = A.work + kSortHistWords;
    const unsigned long long* const scal = reinterpret_cast<const unsigned long long*>(A.work + kSortHistWords + kSortStateWords);
#define GF_SORT_SYNC()                                            \
    if (!sort_grid_sync(state, (int)lane, A.spin_limit)) {        \
        if (lane == 0) atomicOr(&state[2], 1u);                   \
        return;                                                   \
    }
    // ---- 1. field widths from the column ranges
    const unsigned long long cmin = ~scal[0], cmax = scal[1], mmin = ~scal[2], mmax = scal[3];
    const uint32_t tzc = scal[4] ? (uint32_t)__ffsll((long long)scal[4]) - 1u : 0u;
    const uint32_t tzm = scal[5] ? (uint32_t)__ffsll((long long)scal[5]) - 1u : 0u;
    const uint32_t wc = bits_of((cmax - cmin) >> tzc), wm = bits_of((mmax - mmin) >> tzm);
    const uint32_t wz = bits_of((unsigned long long)A.n_zones);  // unknown zone ids rank behind every zone (value n_zones)
    // ---- 2. key groups, least significant first: {cpu, mem, zone} | {cpu} {mem, zone} | {cpu} {mem} {zone}
    const uint32_t n_groups = (wc + wm + wz <= 64u) ? 1u : ((wm + wz <= 64u) ? 2u : 3u);
    uint32_t gw[3] = {0, 0, 0}, total_passes = 0;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const bool has_c = g == 0, has_m = n_groups == 1 || g == 1, has_z = g + 1 == n_groups;
        gw[g] = (has_c ? wc : 0u) + (has_m ? wm : 0u) + (has_z ? wz : 0u);
        total_passes += (gw[g] + 7u) / 8u;
    }
    // segments: a power of two of elements (whole chunks) so that the segment of an output position is a shift
    uint32_t seg_log = 6;
    while (((uint64_t)kSortWG << seg_log) < n) ++seg_log;
    const uint32_t seg = 1u << seg_log;
    const uint32_t n_seg = (n + seg - 1u) >> seg_log;  // segments that hold elements (<= kSortWG)
    const uint32_t lo = ((uint64_t)wave << seg_log) < n ? wave << seg_log : n, hi = lo + seg < n ? lo + seg : n;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t src = 0;  // buffer that holds the current order
    uint32_t P = 0;    // passes done so far (all groups): pass P reads count table P % 3, feeds (P + 1) % 3, clears (P + 2) % 3
    for (uint32_t g = 0; g < n_groups; ++g) {
        const bool has_c = g == 0, has_m = n_groups == 1 || g == 1, has_z = g + 1 == n_groups;
        const uint32_t width = gw[g];
        if (width == 0) continue;  // every key of the group equal: the order stands (uniform over the grid)
        // ---- keys of this group in the current order (this wavefront's segment) + the counts of their first digit
        {
            unsigned long long* kin = A.keys[src];
            const uint32_t* pin = A.perm[src];
            for (uint32_t d = lane; d < 256u; d += 64u) off[d] = 0u;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            // kSortKeyTile chunks at a time: their node ids, then all their gathers, then the keys — a chunk at a time was a chain
            // of three dependent round trips (id -> columns -> zone rank) per 64 elements
            for (uint32_t base = lo; base < hi; base += 64u * kSortKeyTile) {
                uint32_t node[kSortKeyTile], zid[kSortKeyTile];
                int64_t vc[kSortKeyTile], vm[kSortKeyTile];
#pragma unroll
                for (int t = 0; t < kSortKeyTile; ++t) {
                    const uint32_t i = base + (uint32_t)t * 64u + lane;
                    node[t] = i < hi ? pin[i] : 0u;
                }
#pragma unroll
                for (int t = 0; t < kSortKeyTile; ++t) {
                    const uint32_t i = base + (uint32_t)t * 64u + lane;
                    vc[t] = (has_c && i < hi) ? A.cpu[node[t]] : 0;
                    vm[t] = (has_m && i < hi) ? A.mem[node[t]] : 0;
                    zid[t] = (has_z && i < hi) ? A.zone[node[t]] : 0u;
                }
#pragma unroll
                for (int t = 0; t < kSortKeyTile; ++t) {
                    if (base + (uint32_t)t * 64u >= hi) break;  // wave-uniform
                    const uint32_t i = base + (uint32_t)t * 64u + lane;
                    const bool valid = i < hi;
                    unsigned long long key = 0ull;
                    if (valid) {
                        uint32_t sft = 0;
                        if (has_c) {
                            key = (biased(vc[t]) - cmin) >> tzc;
                            sft = wc;
                        }
                        if (has_m) {
                            if (sft < 64u) key |= ((biased(vm[t]) - mmin) >> tzm) << sft;
                            sft += wm;
                        }
                        if (has_z) {
                            const uint32_t z = zid[t];
                            const unsigned long long zr = z < A.n_zones ? (zr_local ? zrank_l[z] : A.zrank[z]) : A.n_zones;
                            if (sft < 64u) key |= zr << sft;
                        }
                        kin[i] = key;
                    }
                    const uint32_t d = (uint32_t)key & 255u;
                    const unsigned long long peers = digit_peers(d, valid);
                    if (valid && (peers & lt_mask) == 0ull) off[d] += (uint32_t)__popcll(peers);  // the peers' leader
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            uint32_t* row = hist + ((size_t)(P % 3u) * kSortWG + wave) * 256u;
            for (uint32_t d = lane; d < 256u; d += 64u) row[d] = off[d];
        }
        GF_SORT_SYNC()
        for (uint32_t shift = 0; shift < width; shift += 8u) {
            const uint32_t next = shift + 8u;
            const bool feeds = next < width;
            const uint32_t remaining = total_passes - P;  // passes left, this one included: the last one must land in buffer 1
            const uint32_t dst_buf = (remaining & 1u) ? 1u : (src == 2u ? 0u : 2u);
            const unsigned long long* kin = A.keys[src];
            const uint32_t* pin = A.perm[src];
            unsigned long long* kout = A.keys[dst_buf];
            uint32_t* pout = A.perm[dst_buf];
            const uint32_t* cur = hist + (size_t)(P % 3u) * kSortWG * 256u;
            uint32_t* nxt = hist + (size_t)((P + 1u) % 3u) * kSortWG * 256u;
            uint32_t* clr = hist + ((size_t)((P + 2u) % 3u) * kSortWG + wave) * 256u;
            // ---- this segment's output offsets: digit-major exclusive scan over (digit, segment); lane l owns digits 4l .. 4l+3
            {
                uint32_t tot[4] = {0, 0, 0, 0}, pre[4] = {0, 0, 0, 0};
                for (uint32_t r0 = 0; r0 < n_seg; r0 += (uint32_t)kSortRowBatch) {
                    uint4 rows[kSortRowBatch];
#pragma unroll
                    for (int t = 0; t < kSortRowBatch; ++t)
                        rows[t] = (r0 + (uint32_t)t < n_seg) ? reinterpret_cast<const uint4*>(cur + (size_t)(r0 + t) * 256u)[lane]
                                                             : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < kSortRowBatch; ++t) {
                        const bool before = r0 + (uint32_t)t < wave;
                        tot[0] += rows[t].x;
                        tot[1] += rows[t].y;
                        tot[2] += rows[t].z;
                        tot[3] += rows[t].w;
                        pre[0] += before ? rows[t].x : 0u;
                        pre[1] += before ? rows[t].y : 0u;
                        pre[2] += before ? rows[t].z : 0u;
                        pre[3] += before ? rows[t].w : 0u;
                    }
                }
                const uint32_t mine = tot[0] + tot[1] + tot[2] + tot[3];
                uint32_t incl = mine;
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_up(incl, o, 64);
                    if ((int)lane >= o) incl += t;
                }
                uint32_t base = incl - mine;
                off[4 * lane + 0] = base + pre[0];
                base += tot[0];
                off[4 * lane + 1] = base + pre[1];
                base += tot[1];
                off[4 * lane + 2] = base + pre[2];
                base += tot[2];
                off[4 * lane + 3] = base + pre[3];
                reinterpret_cast<uint4*>(clr)[lane] = make_uint4(0, 0, 0, 0);  // nobody touches this table during this pass
                __builtin_amdgcn_s_waitcnt(0xC07F);
            }
            // ---- stable scatter; the next pass's digit is counted into the destination segment's row
            for (uint32_t base = lo; base < hi; base += 64u * kSortTile) {
                unsigned long long k[kSortTile];
                uint32_t v[kSortTile];
#pragma unroll
                for (int t = 0; t < kSortTile; ++t) {
                    const uint32_t i = base + (uint32_t)t * 64u + lane;
                    k[t] = i < hi ? kin[i] : 0ull;
                    v[t] = i < hi ? pin[i] : 0u;
                }
#pragma unroll
                for (int t = 0; t < kSortTile; ++t) {
                    const uint32_t i = base + (uint32_t)t * 64u + lane;
                    if (base + (uint32_t)t * 64u >= hi) break;  // wave-uniform
                    const bool valid = i < hi;
                    const uint32_t d = (uint32_t)(k[t] >> shift) & 255u;
                    const unsigned long long peers = digit_peers(d, valid);
                    uint32_t o = 0;
                    if (valid) o = off[d];
                    const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
                    // (LDS operations of a wavefront execute in order: every lane has read the offset before its leader moves it on)
                    if (valid && rank == 0u) off[d] = o + (uint32_t)__popcll(peers);
                    if (valid) {
                        const uint32_t dst = o + rank;
                        kout[dst] = k[t];
                        pout[dst] = v[t];
                        if (feeds) atomicAdd(&nxt[(size_t)(dst >> seg_log) * 256u + ((uint32_t)(k[t] >> next) & 255u)], 1u);
                    }
                }
            }
            src = dst_buf;
            ++P;
            if (P < total_passes) GF_SORT_SYNC()  // nothing follows the last pass: it wrote the output array
        }
    }
#undef GF_SORT_SYNC
    // ---- no pass at all (every key equal): the name order is the result
    if (total_passes == 0)
        for (uint32_t i = lo + lane; i < hi; i += 64u) A.perm[1][i] = A.perm[0][i];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- slot tables
// What gf_orders_set builds on the host for the merged layout, built from the device-resident columns instead: slot s
// = position s of the priority order (every node gets a slot; nodes that are neither driver nor executor candidates
// simply have no candidate bit), one sentinel slot behind them.

__device__ __forceinline__ uint64_t gcd_u64(uint64_t a, uint64_t b) {
    while (b) {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// One thread per slot, 64-slot chunks = wavefronts.  Tables, candidate masks, chunk maxima, per-chunk gcds, zone facts.
constexpr uint32_t kFinalizeBlock = 1024;  // sixteen chunks per workgroup: its zone facts leave as one atomic per zone
__global__ __launch_bounds__(kFinalizeBlock) void finalize_slots_kernel(SnapshotFinalize f) {
    __shared__ uint32_t zf[kZoneLdsMax], zx[kZoneLdsMax];  // first driver slot / "has an executor candidate" per zone
    if (f.n_zones <= kZoneLdsMax)
        for (uint32_t zi = threadIdx.x; zi < f.n_zones; zi += blockDim.x) {
            zf[zi] = 0xFFFFFFFFu;
            zx[zi] = 0u;
        }
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = s >> 6;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n = f.n_nodes, ns = f.n_slots;
    const bool in_range = s < ns;
    const bool real = s < n;
    const uint32_t node = real ? f.d_perm[s] : GF_NO_NODE;
    int64_t a[3] = {-(INT64_C(1) << 62), -(INT64_C(1) << 62), -(INT64_C(1) << 62)};  // sentinel: never fits, never hosts
    int64_t sc[3] = {0, 0, 0};
    uint32_t flags = 0, z = 0;
    if (real) {
        for (int j = 0; j < 3; ++j) {
            a[j] = f.d_avail[(size_t)j * n + node];
            sc[j] = f.d_sched[(size_t)j * n + node];
            f.d_node_tab[(size_t)j * n + node] = a[j];
            f.d_node_tab[(size_t)(3 + j) * n + node] = sc[j];
        }
        flags = f.d_flags[node];
        z = f.d_zone[node];
        f.d_node_slot[node] = s;
        if (sc[0] < 0 || sc[1] < 0 || sc[2] < 0) {
            atomicOr(&f.d_scalars[2], 1u);
            if (f.h_out != nullptr) f.h_out[2] = 1u;  // (every writer stores the same value: no atomic over the host link)
        }
    }
    if (in_range) {
        for (int j = 0; j < 3; ++j) {
            f.d_snap[(size_t)j * ns + s] = a[j];
            f.d_sched_slot[(size_t)j * ns + s] = sc[j];
        }
        f.d_slot_node[s] = node;
        f.d_dslot[s] = s;
    }
    const bool xbit = real && !(flags & GF_NODE_UNSCHEDULABLE) && (flags & GF_NODE_READY);
    const bool dbit = real && (flags & GF_NODE_DRIVER_CANDIDATE);
    const uint64_t xm = __ballot(xbit), dm = __ballot(dbit);
    int64_t m[3];
    uint64_t g[3], mag[3];
    for (int j = 0; j < 3; ++j) {
        m[j] = in_range ? a[j] : INT64_MIN;
        uint64_t v = real ? (uint64_t)(a[j] < 0 ? -a[j] : a[j]) : 0ull;
        uint64_t any = v;
        mag[j] = v;  // largest magnitude of the chunk: how far a batch may refine the units (narrow_begin)
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t om = __shfl_xor(m[j], d, 64);
            m[j] = om > m[j] ? om : m[j];
            any |= (uint64_t)__shfl_xor((long long)any, d, 64);
            const uint64_t og = (uint64_t)__shfl_xor((long long)mag[j], d, 64);
            mag[j] = og > mag[j] ? og : mag[j];
        }
        // gcd of the chunk: the common power of two comes from the OR of the magnitudes; what is left of byte / milli
        // quantities almost always fits 32 bits, where Euclid's steps are cheap (gfx950 has no 64-bit divider)
        const int tz = any ? __builtin_ctzll(any) : 0;
        v >>= tz;
        if (__ballot(v >> 32) == 0) {
            // A candidate (some non-zero value of the chunk) and ROUNDS: every lane takes its value modulo the candidate — one
            // division per lane and round —; when every remainder is zero the candidate, itself a member, is the gcd, else it
            // is replaced by its gcd with one non-zero remainder (at most half of it: a real cluster's chunk of equal or
            // commensurable nodes ends after one or two rounds).  The butterfly this replaces ran Euclid's loop in every lane
            // at every one of its six levels.
            const uint32_t w = (uint32_t)v;
            const uint64_t nz = __ballot(w != 0u);
            uint32_t cand = 0;
            if (nz) {
                cand = (uint32_t)__shfl((int)w, __ffsll((unsigned long long)nz) - 1, 64);
                for (;;) {
                    const uint32_t r = w % cand;
                    const uint64_t left = __ballot(r != 0u);
                    if (left == 0) break;
                    uint32_t o = (uint32_t)__shfl((int)r, __ffsll((unsigned long long)left) - 1, 64);  // 0 < o < cand
                    while (o) {  // wave-uniform Euclid
                        const uint32_t t = cand % o;
                        cand = o;
                        o = t;
                    }
                }
            }
            g[j] = (uint64_t)cand << tz;
        } else {
            for (int d = 1; d < 64; d <<= 1) v = gcd_u64(v, (uint64_t)__shfl_xor((long long)v, d, 64));
            g[j] = v << tz;
        }
    }
    if (lane == 0 && c < f.n_chunks) {
        f.d_masks[c] = xm;
        f.d_masks[f.n_chunks + c] = dm;
        for (int j = 0; j < 3; ++j) {
            f.d_cmax[(size_t)j * f.n_chunks + c] = m[j];
            f.d_gcd_part[(size_t)j * f.n_chunks + c] = g[j];
            f.d_gcd_part[(size_t)(3 + j) * f.n_chunks + c] = mag[j];
        }
    }
    // zones by first appearance in the driver order, and whether they own an executor candidate (single_az.go:36-41)
    // (one atomic per wavefront and zone, combined per workgroup in LDS when the zones fit: the global ones are device-scope
    //  read-modify-writes of a handful of words and queue behind each other — 1 563 wavefronts x 3 zones x 2 at 100 000 nodes)
    const bool z_lds = f.n_zones <= kZoneLdsMax;
    if (z_lds) __syncthreads();  // zf / zx initialised (top of the kernel)
    uint64_t todo = __ballot((dbit || xbit) && z < f.n_zones);
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t zl = (uint32_t)__shfl((int)z, leader, 64);
        const uint64_t same = __ballot(z == zl && z < f.n_zones && real);
        const uint64_t dsame = __ballot(z == zl && dbit), xsame = __ballot(z == zl && xbit);
        if (lane == leader) {
            const uint32_t first = (s - (uint32_t)lane) + (uint32_t)(__ffsll((unsigned long long)dsame) - 1);
            if (z_lds) {
                if (dsame) atomicMin(&zf[zl], first);
                if (xsame) atomicOr(&zx[zl], 1u);
            } else {
                if (dsame) atomicMin(&f.d_zfirst[zl], first);
                if (xsame) atomicOr(&f.d_zhasx[zl], 1u);
            }
        }
        todo &= ~same;
    }
    if (z_lds) {
        __syncthreads();
        for (uint32_t zi = threadIdx.x; zi < f.n_zones; zi += blockDim.x) {
            if (zf[zi] != 0xFFFFFFFFu) atomicMin(&f.d_zfirst[zi], zf[zi]);
            if (zx[zi]) atomicOr(&f.d_zhasx[zi], 1u);
        }
    }
}

// gcd of two 64-bit values without the 64-bit software division where it can be avoided: equal operands (the rule — the chunks
// of a real cluster share their gcd) and zeros return at once; otherwise the common power of two is set aside and Euclid runs
// on the odd parts, in 32 bits when both fit (byte / milli quantities almost always do).
__device__ __forceinline__ uint64_t gcd_fast(uint64_t a, uint64_t b) {
    if (a == b || b == 0) return a;
    if (a == 0) return b;
    const int sh = __builtin_ctzll(a | b);
    a >>= __builtin_ctzll(a);
    b >>= __builtin_ctzll(b);
    if (((a | b) >> 32) == 0) {
        uint32_t x = (uint32_t)a, y = (uint32_t)b;
        while (y) {
            const uint32_t t = x % y;
            x = y;
            y = t;
        }
        return (uint64_t)x << sh;
    }
    return gcd_u64(a, b) << sh;
}

// One workgroup: the per-dimension units (gcd over the chunk gcds) and the zone evaluation list.  Four groups of 256 threads
// side by side — the three dimensions' gcd trees and the three magnitude maxima — instead of one after the other: the trees are
// chains of dependent gcds, and this kernel sits between two grid-wide ones on the build's critical path.
// Thread 0 also gathers what the host reads back into ONE range: d_scalars[3] = the priority sort's error word,
// d_scalars[4 .. 15] = the three units and the three largest scaled magnitudes as pairs of 32-bit words (one copy, not three).
constexpr uint32_t kReduceGroup = 256;
__global__ __launch_bounds__(4 * kReduceGroup) void finalize_reduce_kernel(SnapshotFinalize f, const uint32_t* __restrict__ sort_error) {
    __shared__ unsigned long long part[6][kReduceGroup];
    const uint32_t grp = threadIdx.x / kReduceGroup, t = threadIdx.x % kReduceGroup;
    if (grp < 3) {
        uint64_t g = 0;
        for (uint32_t c = t; c < f.n_chunks; c += kReduceGroup) g = gcd_fast(g, f.d_gcd_part[(size_t)grp * f.n_chunks + c]);
        part[grp][t] = g;
    } else {
        uint64_t mg[3] = {0, 0, 0};
        for (uint32_t c = t; c < f.n_chunks; c += kReduceGroup)
            for (int j = 0; j < 3; ++j) {
                const uint64_t v = f.d_gcd_part[(size_t)(3 + j) * f.n_chunks + c];
                mg[j] = v > mg[j] ? v : mg[j];
            }
        for (int j = 0; j < 3; ++j) part[3 + j][t] = mg[j];
    }
    __syncthreads();
    for (uint32_t w = kReduceGroup / 2; w > 0; w >>= 1) {
        if (t < w) {
            if (grp < 3) {
                part[grp][t] = gcd_fast(part[grp][t], part[grp][t + w]);
            } else {
                for (int j = 0; j < 3; ++j) {
                    const uint64_t o = part[3 + j][t + w];
                    if (o > part[3 + j][t]) part[3 + j][t] = o;
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int j = 0; j < 3; ++j) {
            const uint64_t g = part[j][0];
            const long long unit = g ? (long long)g : 1ll;
            const long long top = (long long)(part[3 + j][0] / (g ? g : 1ull));  // largest scaled magnitude
            f.d_units[j] = unit;
            f.d_units[3 + j] = top;
            f.d_scalars[4 + 2 * j] = (uint32_t)(unsigned long long)unit;
            f.d_scalars[5 + 2 * j] = (uint32_t)((unsigned long long)unit >> 32);
            f.d_scalars[10 + 2 * j] = (uint32_t)(unsigned long long)top;
            f.d_scalars[11 + 2 * j] = (uint32_t)((unsigned long long)top >> 32);
        }
        f.d_scalars[3] = sort_error != nullptr ? *sort_error : 0u;
        if (f.h_out != nullptr) {  // the host's copy, written in place (pinned, device-mapped; complete when the build's last kernel is)
            for (int k = 3; k < 16; ++k) f.h_out[k] = f.d_scalars[k];
        }
        // evaluation list: zones that have a driver candidate AND an executor candidate, ordered by their first driver slot
        uint32_t nz = 0;
        for (uint32_t z = 0; z < f.n_zones; ++z) f.d_zeval[z] = GF_NO_NODE;
        for (;;) {
            uint32_t best = GF_NO_NODE, best_first = GF_NO_NODE;
            for (uint32_t z = 0; z < f.n_zones; ++z)
                if (f.d_zeval[z] == GF_NO_NODE && f.d_zhasx[z] && f.d_zfirst[z] < best_first) {
                    best = z;
                    best_first = f.d_zfirst[z];
                }
            if (best == GF_NO_NODE) break;
            f.d_zeval[best] = nz++;
        }
        f.d_scalars[0] = nz;  // [1] = "a scaled value does not fit 2^30" (next kernel), [2] = "negative schedulable value"
        if (f.h_out != nullptr) f.h_out[0] = nz;
    }
}

// One thread per slot: the narrow (scaled int32) table + its chunk maxima, and the per-zone candidate masks.
__global__ __launch_bounds__(256) void finalize_narrow_zones_kernel(SnapshotFinalize f) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = s >> 6;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n = f.n_nodes, ns = f.n_slots;
    const bool real = s < n;
    const uint32_t node = real ? f.d_slot_node[s] : GF_NO_NODE;
    int32_t v[3] = {INT32_MIN / 2, INT32_MIN / 2, INT32_MIN / 2};
    bool bad = false;
    if (real)
        for (int j = 0; j < 3; ++j) {
            const int64_t q = f.d_snap[(size_t)j * ns + s] / f.d_units[j];
            bad = bad || q >= (INT64_C(1) << 30) || q <= -(INT64_C(1) << 30);
            v[j] = (int32_t)q;
        }
    if (__ballot(bad) && lane == 0) {
        atomicOr(&f.d_scalars[1], 1u);
        if (f.h_out != nullptr) f.h_out[1] = 1u;
    }
    for (int j = 0; j < 3; ++j) {
        if (s < ns) f.d_nsnap[(size_t)j * ns + s] = v[j];
        int32_t m = s < ns ? v[j] : INT32_MIN;
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t o = __shfl_xor(m, d, 64);
            m = o > m ? o : m;
        }
        if (lane == 0 && c < f.n_chunks) f.d_ncmax[(size_t)j * f.n_chunks + c] = m;
    }
    uint32_t ei = GF_NO_NODE;
    bool xbit = false, dbit = false;
    if (real) {
        const uint32_t flags = f.d_flags[node], z = f.d_zone[node];
        xbit = !(flags & GF_NODE_UNSCHEDULABLE) && (flags & GF_NODE_READY);
        dbit = flags & GF_NODE_DRIVER_CANDIDATE;
        if (z < f.n_zones) ei = f.d_zeval[z];
    }
    const uint32_t nz = f.d_scalars[0];
    for (uint32_t e = 0; e < nz; ++e) {
        const uint64_t zx = __ballot(ei == e && xbit), zd = __ballot(ei == e && dbit);
        if (lane == 0 && c < f.n_chunks) {
            f.d_zmasks[(size_t)e * f.n_chunks + c] = zx;
            f.d_zmasks[((size_t)f.n_zones + e) * f.n_chunks + c] = zd;
        }
    }
}

// (d_zfirst, d_zhasx and d_scalars[0 .. 16) were cleared by launch_snapshot_build's one clearing kernel: SnapshotBuild names them)
hipError_t launch_snapshot_finalize(const SnapshotFinalize& f, const uint32_t* d_sort_error, hipStream_t stream) {
    const dim3 block(256), grid((unsigned)(((size_t)f.n_chunks * 64 + 255) / 256));
    // sixteen chunks per workgroup on large tables (fewer zone atomics); on small ones four — sixteen wavefronts of gcd loops
    // share a CU's four SIMDs, and a 10 000-node table has no more wavefronts than the device has CUs (20 us against 13)
    const unsigned sblock = f.n_chunks > 512u ? kFinalizeBlock : 256u;
    const dim3 sgrid((unsigned)(((size_t)f.n_chunks * 64 + sblock - 1) / sblock));
    hipLaunchKernelGGL(finalize_slots_kernel, sgrid, dim3(sblock), 0, stream, f);
    hipLaunchKernelGGL(finalize_reduce_kernel, dim3(1), dim3(4 * kReduceGroup), 0, stream, f, d_sort_error);
    hipLaunchKernelGGL(finalize_narrow_zones_kernel, grid, block, 0, stream, f);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------- bandwidth / launch probes
// What THIS device delivers, measured next to the spec figures the rooflines quote:
//   stream_read_kernel  — read-only stream (the access pattern of the scans: loads, almost no stores), 16 bytes per lane,
//                         eight independent loads in flight per lane, far larger than L2 + MALL;
//   stream_copy_kernel  — read + write of the same size (a copy pays the write-allocate / turnaround traffic: lower);
//   empty_kernel        — the floor of one dependent kernel launch on a stream (nothing to do, one wavefront).
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4* __restrict__ src_, size_t n16, uint32_t* __restrict__ sink) {
    const probe_u32x4* __restrict__ src = reinterpret_cast<const probe_u32x4*>(src_);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        probe_u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) {
        const probe_u32x4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) *sink = acc;  // never true for the probe's fill pattern; keeps the loads alive
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(64) void empty_kernel(uint32_t* __restrict__ sink) {
    if (sink != nullptr && threadIdx.x == 1234567u) *sink = 0;
}

// Up to three word ranges copied by ONE kernel — the way a blocking entry point hands its answers to pinned host memory
// (posted PCIe writes, visible when the kernel has completed) without leaving the compute queue for a copy engine.
__global__ __launch_bounds__(256) void copy_out_kernel(CopyOut c) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 3; ++r)
        for (size_t i = t; i < c.words[r]; i += stride) c.dst[r][i] = c.src[r][i];
}

hipError_t launch_copy_out(const CopyOut& c, hipStream_t stream) {
    const size_t total = c.words[0] + c.words[1] + c.words[2];
    if (total == 0) return hipSuccess;
    size_t blocks = (total + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(copy_out_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, c);
    return hipGetLastError();
}

hipError_t launch_stream_copy(const void* src, void* dst, size_t bytes, hipStream_t stream) {
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 32), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, n16);
    return hipGetLastError();
}

hipError_t launch_stream_read(const void* src, size_t bytes, uint32_t* sink, hipStream_t stream) {
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 16), dim3(256), 0, stream, (const uint4*)src, n16, sink);
    return hipGetLastError();
}

hipError_t launch_empty(uint32_t* sink, hipStream_t stream) {
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, stream, sink);
    return hipGetLastError();
}

hipError_t launch_usage_apply(uint32_t n_entries, uint32_t n_nodes, const uint32_t* d_node, const int64_t* d_req, int sign,
                              int64_t* d_usage, uint32_t* d_negative, hipStream_t stream) {
    if (n_entries == 0 || n_nodes == 0) return hipSuccess;
    hipLaunchKernelGGL(usage_scatter_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, stream, n_entries, n_nodes, d_node,
                       d_req, d_req + n_entries, d_req + 2 * (size_t)n_entries, reinterpret_cast<unsigned long long*>(d_usage),
                       sign < 0 ? -1 : 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || sign > 0 || d_negative == nullptr) return e;
    hipLaunchKernelGGL(usage_negative_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, stream, n_entries, n_nodes, d_node,
                       (const int64_t*)d_usage, d_negative);
    return hipGetLastError();
}

hipError_t launch_snapshot_build(const SnapshotBuild& b, hipStream_t stream) {
    const uint32_t n = b.n_nodes;
    if (n == 0) return hipSuccess;
    hipError_t e;
    // the priority sort's work area (count tables, barrier and scalars start at zero; metadata_kernel leaves the column ranges),
    // the zone sums, the usage sums unless they are resident, and the finalize step's zone facts and scalars: one clearing launch
    const size_t work_bytes = (size_t)(kSortHistWords + kSortStateWords) * sizeof(uint32_t) + kSortScalars * sizeof(unsigned long long);
    {
        SnapshotClear c{};
        c.zero[0] = b.d_sort_work;
        c.zero_words[0] = work_bytes / sizeof(uint32_t);
        c.zero[1] = reinterpret_cast<uint32_t*>(b.d_zone_sum);
        c.zero_words[1] = 2 * 3 * (size_t)(b.n_zones ? b.n_zones : 1);
        if (!b.usage_resident) {
            c.zero[2] = reinterpret_cast<uint32_t*>(b.d_usage);
            c.zero_words[2] = 2 * 3 * (size_t)n;
        }
        if (b.d_zhasx != nullptr) {  // d_zhasx | d_zeval | d_scalars are one range (gf_snapshot_build's carve)
            c.zero[3] = b.d_zhasx;
            c.zero_words[3] = b.zhasx_to_scalars_words;
        }
        c.ones = b.d_zfirst;
        c.ones_words = b.d_zfirst != nullptr ? b.n_zones : 0;
        const size_t most = c.zero_words[2] > c.zero_words[0] ? c.zero_words[2] : c.zero_words[0];
        const unsigned blocks = (unsigned)((most / 4 + 255) / 256);
        hipLaunchKernelGGL(snapshot_clear_kernel, dim3(blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks)), dim3(256), 0, stream, c);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    const dim3 block(256);
    if (b.n_res > 0 && !b.usage_resident)
        hipLaunchKernelGGL(usage_scatter_kernel, dim3((b.n_res + 255) / 256), block, 0, stream, b.n_res, n, b.d_res_node,
                           b.d_res_req, b.d_res_req + b.n_res, b.d_res_req + 2 * (size_t)b.n_res,
                           reinterpret_cast<unsigned long long*>(b.d_usage), 1);
    unsigned long long* sort_scal = reinterpret_cast<unsigned long long*>(b.d_sort_work + kSortHistWords + kSortStateWords);
    const unsigned meta_blocks = (n + 255) / 256;
    const dim3 grid(meta_blocks > kMetaBlocks ? kMetaBlocks : meta_blocks);
    hipLaunchKernelGGL(metadata_kernel, grid, block, 0, stream, n, b.d_alloc, b.d_overhead, (const int64_t*)b.d_usage,
                       b.d_zone, b.n_zones, b.d_avail, b.d_sched, reinterpret_cast<unsigned long long*>(b.d_zone_sum),
                       b.d_name_rank, b.d_perm_a, sort_scal);
    if (b.n_zones > 64u)  // (up to 64 zones are ranked by the sort kernel itself: one launch less on the build's critical path)
        hipLaunchKernelGGL(zone_rank_kernel, dim3(1), dim3(64), 0, stream, b.n_zones, (const long long*)b.d_zone_sum,
                           b.d_zone_order, b.d_zone_rank);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // (zone rank, free memory, free cpu, name): one cooperative launch, the result sits in d_perm_b
    PrioritySort ps{n, b.n_zones, b.d_avail, b.d_avail + n, b.d_zone, b.d_zone_rank, (const long long*)b.d_zone_sum,
                    {reinterpret_cast<unsigned long long*>(b.d_keys_a), reinterpret_cast<unsigned long long*>(b.d_keys_b),
                     reinterpret_cast<unsigned long long*>(b.d_keys_c)},
                    {b.d_perm_a, b.d_perm_b, b.d_perm_c}, b.d_sort_work, b.sort_fault ? (1u << 12) : (1u << 24)};
    // An ordinary launch: sixty-four one-wavefront workgroups become resident as soon as sixty-four wave slots are free (every
    // other kernel of this library terminates on its own), and the grid barrier gives up with an error flag instead of
    // spinning forever.  (hipLaunchCooperativeKernel would also promise residency, but rocprofv3 crashes at process exit
    // behind a cooperative launch.)
    // (sort_fault: the last workgroup is not launched, so the barrier can never complete — what an oversubscribed device
    //  looks like to the other sixty-three; they flag the error and leave, and the host refuses the build)
    hipLaunchKernelGGL(priority_sort_kernel, dim3(b.sort_fault ? kSortWG - 1u : kSortWG), dim3(64), 0, stream, ps);
    return hipGetLastError();
}

size_t snapshot_sort_work_words() { return kSortHistWords + kSortStateWords + 2u * kSortScalars; }
uint32_t snapshot_sort_error_word() { return kSortHistWords + 2u; }

}  // namespace gangfit
