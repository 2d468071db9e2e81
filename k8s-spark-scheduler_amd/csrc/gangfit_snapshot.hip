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

// ------------------------------------------------------------------------------------------------ the priority order
// getNodeNamesInPriorityOrder (internal/sort/nodesorting.go:95-122): nodes by (zone rank, free memory, free cpu, name), all
// ascending.  One workgroup of sixteen wavefronts, one launch:
//   1. range of every column (min, max, OR of value - min): a field needs bits(max - min) bits, less its common trailing
//      zeros (free memory is a multiple of hundreds of MiB on real clusters: 39 bits shrink to about 11);
//   2. the fields are packed into ONE 64-bit key per node — zone rank | memory | cpu — when they fit (they do unless the
//      quantities are adversarial: then two or three keys are sorted one after the other, least significant first);
//   3. stable LSD radix sort of (key, node) pairs, 8 bits per pass, starting from the name order (the final tie-break):
//      every wavefront owns a contiguous segment, counts its digits with a ballot-built peer mask per 64 elements (no
//      atomics), the 16 x 256 counts are scanned digit-major, and the scatter ranks every element inside its 64-element
//      chunk with the same peer mask.  Passes whose digit is zero in every key are skipped.
// Keys and permutation ping-pong between two global buffers (L2 resident: 1.2 MB at 100 000 nodes); a workgroup shares one
// L1, so a barrier with vmcnt(0) orders one pass's stores before the next pass's loads.
constexpr int kSortWaves = 16;
constexpr int kSortThreads = kSortWaves * 64;
constexpr int kSortTile = 8;  // 64-element chunks a wavefront has in flight per step (independent loads)

struct PrioritySort {
    uint32_t n, n_zones;
    const int64_t* cpu;        // free cpu by node
    const int64_t* mem;        // free memory by node
    const uint32_t* zone;      // zone id by node
    const uint32_t* zrank;     // zone id -> rank
    const uint32_t* name_rank; // node -> rank of its name (a permutation)
    unsigned long long* keys_a;
    unsigned long long* keys_b;
    uint32_t* perm_a;
    uint32_t* perm_b;          // receives the result: position -> node
};

struct SortShared {
    uint32_t hist[kSortWaves][256];  // counts, then running offsets, of (wavefront, digit)
    unsigned long long red[3][kSortWaves];
    unsigned long long cmin, mmin, key_or;
    uint32_t tzc, tzm, wc, wm, wz;
};

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
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
__device__ __forceinline__ uint32_t bits_of(unsigned long long v) { return v ? 64u - (uint32_t)__clzll(v) : 0u; }

// order-preserving map of a signed quantity to unsigned
__device__ __forceinline__ unsigned long long biased(int64_t v) { return (unsigned long long)v ^ 0x8000000000000000ull; }

__global__ __launch_bounds__(kSortThreads) void priority_sort_kernel(PrioritySort A) {
    __shared__ SortShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n = A.n;
    // ---- 1. column ranges; the permutation starts in name order
    unsigned long long cmn = ~0ull, cmx = 0ull, mmn = ~0ull, mmx = 0ull;
    for (uint32_t i = tid; i < n; i += kSortThreads) {
        const unsigned long long c = biased(A.cpu[i]), m = biased(A.mem[i]);
        cmn = c < cmn ? c : cmn;
        cmx = c > cmx ? c : cmx;
        mmn = m < mmn ? m : mmn;
        mmx = m > mmx ? m : mmx;
        A.perm_a[A.name_rank[i]] = i;
    }
    cmn = wave_min_u64(cmn);
    cmx = wave_max_u64(cmx);
    mmn = wave_min_u64(mmn);
    mmx = wave_max_u64(mmx);
    if (lane == 0) {
        sh.red[0][wave] = cmn;
        sh.red[1][wave] = cmx;
        sh.red[2][wave] = mmn;
    }
    __syncthreads();
    if (wave == 0) {
        unsigned long long a = lane < kSortWaves ? sh.red[0][lane] : ~0ull, b = lane < kSortWaves ? sh.red[1][lane] : 0ull,
                           c = lane < kSortWaves ? sh.red[2][lane] : ~0ull;
        a = wave_min_u64(a);
        b = wave_max_u64(b);
        c = wave_min_u64(c);
        if (lane == 0) {
            sh.cmin = a;
            sh.mmin = c;
            sh.red[1][0] = b;
        }
    }
    __syncthreads();
    const unsigned long long cmin = sh.cmin, mmin = sh.mmin, cmax = sh.red[1][0];
    __syncthreads();
    if (lane == 0) sh.red[0][wave] = mmx;
    unsigned long long cor = 0ull, mor = 0ull;  // common trailing zeros of the offsets
    for (uint32_t i = tid; i < n; i += kSortThreads) {
        cor |= biased(A.cpu[i]) - cmin;
        mor |= biased(A.mem[i]) - mmin;
    }
    cor = wave_or_u64(cor);
    mor = wave_or_u64(mor);
    if (lane == 0) {
        sh.red[1][wave] = cor;
        sh.red[2][wave] = mor;
    }
    __syncthreads();
    if (wave == 0) {
        unsigned long long a = lane < kSortWaves ? sh.red[0][lane] : 0ull, b = lane < kSortWaves ? sh.red[1][lane] : 0ull,
                           c = lane < kSortWaves ? sh.red[2][lane] : 0ull;
        a = wave_max_u64(a);
        b = wave_or_u64(b);
        c = wave_or_u64(c);
        if (lane == 0) {
            const uint32_t tzc = b ? (uint32_t)__ffsll((long long)b) - 1u : 0u, tzm = c ? (uint32_t)__ffsll((long long)c) - 1u : 0u;
            sh.tzc = tzc;
            sh.tzm = tzm;
            sh.wc = bits_of((cmax - cmin) >> tzc);
            sh.wm = bits_of((a - mmin) >> tzm);
            sh.wz = bits_of((unsigned long long)A.n_zones);  // unknown zone ids rank behind every zone (value n_zones)
        }
    }
    __syncthreads();
    const uint32_t tzc = sh.tzc, tzm = sh.tzm, wc = sh.wc, wm = sh.wm, wz = sh.wz;
    // ---- 2. key groups, least significant first: {cpu, mem, zone} | {cpu} {mem, zone} | {cpu} {mem} {zone}
    const uint32_t n_groups = (wc + wm + wz <= 64u) ? 1u : ((wm + wz <= 64u) ? 2u : 3u);
    unsigned long long* kin = A.keys_a;
    unsigned long long* kout = A.keys_b;
    uint32_t* pin = A.perm_a;
    uint32_t* pout = A.perm_b;
    const uint32_t seg = ((n + kSortWaves - 1) / kSortWaves + 63u) & ~63u;  // elements per wavefront, whole chunks
    const uint32_t lo = wave * seg < n ? wave * seg : n, hi = lo + seg < n ? lo + seg : n;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const bool has_c = g == 0, has_m = n_groups == 1 || g == 1, has_z = g + 1 == n_groups;
        const uint32_t width = (has_c ? wc : 0u) + (has_m ? wm : 0u) + (has_z ? wz : 0u);
        // keys of this group for the current order
        __syncthreads();
        unsigned long long kor = 0ull;
        for (uint32_t i = tid; i < n; i += kSortThreads) {
            const uint32_t node = pin[i];
            unsigned long long key = 0ull;
            uint32_t sft = 0;
            if (has_c) {
                key = (biased(A.cpu[node]) - cmin) >> tzc;
                sft = wc;
            }
            if (has_m) {
                if (sft < 64u) key |= ((biased(A.mem[node]) - mmin) >> tzm) << sft;
                sft += wm;
            }
            if (has_z) {
                const uint32_t z = A.zone[node];
                const unsigned long long zr = z < A.n_zones ? A.zrank[z] : A.n_zones;
                if (sft < 64u) key |= zr << sft;
            }
            kin[i] = key;
            kor |= key;
        }
        kor = wave_or_u64(kor);
        if (lane == 0) sh.red[0][wave] = kor;
        __syncthreads();
        if (tid == 0) {
            unsigned long long o = 0ull;
            for (int w = 0; w < kSortWaves; ++w) o |= sh.red[0][w];
            sh.key_or = o;
        }
        __syncthreads();
        const unsigned long long key_or = sh.key_or;
        for (uint32_t shift = 0; shift < width; shift += 8u) {
            if (((key_or >> shift) & 255ull) == 0ull) continue;  // every key has digit 0 here: the pass is the identity
            // ---- counts per (wavefront, digit)
            for (uint32_t i = tid; i < kSortWaves * 256u; i += kSortThreads) (&sh.hist[0][0])[i] = 0u;
            __syncthreads();
            for (uint32_t base = lo; base < hi; base += 64u * kSortTile) {
                unsigned long long k[kSortTile];
#pragma unroll
                for (int t = 0; t < kSortTile; ++t) {
                    const uint32_t i = base + (uint32_t)t * 64u + lane;
                    k[t] = i < hi ? kin[i] : 0ull;
                }
#pragma unroll
                for (int t = 0; t < kSortTile; ++t) {
                    const uint32_t i = base + (uint32_t)t * 64u + lane;
                    if (base + (uint32_t)t * 64u >= hi) break;  // wave-uniform
                    const bool valid = i < hi;
                    const uint32_t d = (uint32_t)(k[t] >> shift) & 255u;
                    unsigned long long peers = __ballot(valid);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const bool bit = (d >> b) & 1u;
                        const unsigned long long bal = __ballot(bit);
                        peers &= bit ? bal : ~bal;
                    }
                    if (valid && (peers & lt_mask) == 0ull) sh.hist[wave][d] += (uint32_t)__popcll(peers);  // the peers' leader
                }
            }
            __syncthreads();
            // ---- digit-major exclusive scan: offset(w, d) = sum over d' < d of all counts + sum over w' < w of count(w', d)
            uint32_t run = 0, incl = 0;
            if (tid < 256u) {  // digit tid (wavefronts 0 .. 3)
                for (int w = 0; w < kSortWaves; ++w) {
                    const uint32_t c = sh.hist[w][tid];
                    sh.hist[w][tid] = run;
                    run += c;
                }
                incl = run;  // inclusive scan of the digit totals inside the wavefront
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_up(incl, o, 64);
                    if ((int)lane >= o) incl += t;
                }
                if (lane == 63u) sh.red[1][wave] = incl;
            }
            __syncthreads();
            if (tid < 256u) {
                uint32_t before = 0;
                for (uint32_t w = 0; w < wave; ++w) before += (uint32_t)sh.red[1][w];
                const uint32_t digit_base = before + incl - run;
                for (int w = 0; w < kSortWaves; ++w) sh.hist[w][tid] += digit_base;
            }
            __syncthreads();
            // ---- stable scatter
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
                    if (base + (uint32_t)t * 64u >= hi) break;
                    const bool valid = i < hi;
                    const uint32_t d = (uint32_t)(k[t] >> shift) & 255u;
                    unsigned long long peers = __ballot(valid);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const bool bit = (d >> b) & 1u;
                        const unsigned long long bal = __ballot(bit);
                        peers &= bit ? bal : ~bal;
                    }
                    uint32_t off = 0;
                    if (valid) off = sh.hist[wave][d];
                    const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
                    // (LDS operations of a wavefront execute in order: every lane has read the offset before the leader moves it on)
                    if (valid && rank == 0u) sh.hist[wave][d] = off + (uint32_t)__popcll(peers);
                    if (valid) {
                        kout[off + rank] = k[t];
                        pout[off + rank] = v[t];
                    }
                }
            }
            __syncthreads();  // vmcnt(0) + barrier: the scattered pairs are visible to the whole workgroup
            unsigned long long* tk = kin;
            kin = kout;
            kout = tk;
            uint32_t* tp = pin;
            pin = pout;
            pout = tp;
        }
    }
    // ---- the result belongs in perm_b
    __syncthreads();
    if (pin != A.perm_b)
        for (uint32_t i = tid; i < n; i += kSortThreads) A.perm_b[i] = pin[i];
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
__global__ __launch_bounds__(256) void finalize_slots_kernel(SnapshotFinalize f) {
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
        if (sc[0] < 0 || sc[1] < 0 || sc[2] < 0) atomicOr(&f.d_scalars[2], 1u);
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
            uint32_t w = (uint32_t)v;
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t o = (uint32_t)__shfl_xor((int)w, d, 64);
                while (o) {
                    const uint32_t t = w % o;
                    w = o;
                    o = t;
                }
            }
            g[j] = (uint64_t)w << tz;
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
    // (one atomic per wavefront and zone: a hundred thousand same-address atomics would serialise)
    uint64_t todo = __ballot((dbit || xbit) && z < f.n_zones);
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t zl = (uint32_t)__shfl((int)z, leader, 64);
        const uint64_t same = __ballot(z == zl && z < f.n_zones && real);
        const uint64_t dsame = __ballot(z == zl && dbit), xsame = __ballot(z == zl && xbit);
        if (lane == leader) {
            if (dsame) atomicMin(&f.d_zfirst[zl], (s - (uint32_t)lane) + (uint32_t)(__ffsll((unsigned long long)dsame) - 1));
            if (xsame) atomicOr(&f.d_zhasx[zl], 1u);
        }
        todo &= ~same;
    }
}

// One workgroup: the per-dimension units (gcd over the chunk gcds) and the zone evaluation list.
__global__ __launch_bounds__(256) void finalize_reduce_kernel(SnapshotFinalize f) {
    __shared__ unsigned long long part[6][256];
    for (int j = 0; j < 3; ++j) {
        uint64_t g = 0, mg = 0;
        for (uint32_t c = threadIdx.x; c < f.n_chunks; c += blockDim.x) {
            g = gcd_u64(g, f.d_gcd_part[(size_t)j * f.n_chunks + c]);
            const uint64_t v = f.d_gcd_part[(size_t)(3 + j) * f.n_chunks + c];
            mg = v > mg ? v : mg;
        }
        part[j][threadIdx.x] = g;
        part[3 + j][threadIdx.x] = mg;
    }
    __syncthreads();
    for (uint32_t w = blockDim.x / 2; w > 0; w >>= 1) {  // tree: a serial pass of one thread over 768 software gcds took 0.1 ms
        if (threadIdx.x < w)
            for (int j = 0; j < 3; ++j) {
                part[j][threadIdx.x] = gcd_u64(part[j][threadIdx.x], part[j][threadIdx.x + w]);
                const uint64_t o = part[3 + j][threadIdx.x + w];
                if (o > part[3 + j][threadIdx.x]) part[3 + j][threadIdx.x] = o;
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int j = 0; j < 3; ++j) {
            const uint64_t g = part[j][0];
            f.d_units[j] = g ? (long long)g : 1ll;
            f.d_units[3 + j] = (long long)(part[3 + j][0] / (g ? g : 1ull));  // largest scaled magnitude
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
    if (__ballot(bad) && lane == 0) atomicOr(&f.d_scalars[1], 1u);
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

hipError_t launch_snapshot_finalize(const SnapshotFinalize& f, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(f.d_zfirst, 0xFF, (size_t)f.n_zones * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(f.d_zhasx, 0, (size_t)f.n_zones * sizeof(uint32_t), stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(f.d_scalars, 0, 4 * sizeof(uint32_t), stream)) != hipSuccess) return e;
    const dim3 block(256), grid((unsigned)(((size_t)f.n_chunks * 64 + 255) / 256));
    hipLaunchKernelGGL(finalize_slots_kernel, grid, block, 0, stream, f);
    hipLaunchKernelGGL(finalize_reduce_kernel, dim3(1), block, 0, stream, f);
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
    if (!b.usage_resident && (e = hipMemsetAsync(b.d_usage, 0, 3 * (size_t)n * sizeof(int64_t), stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(b.d_zone_sum, 0, 3 * (size_t)(b.n_zones ? b.n_zones : 1) * sizeof(int64_t), stream)) != hipSuccess)
        return e;
    const dim3 block(256);
    if (b.n_res > 0 && !b.usage_resident)
        hipLaunchKernelGGL(usage_scatter_kernel, dim3((b.n_res + 255) / 256), block, 0, stream, b.n_res, n, b.d_res_node,
                           b.d_res_req, b.d_res_req + b.n_res, b.d_res_req + 2 * (size_t)b.n_res,
                           reinterpret_cast<unsigned long long*>(b.d_usage), 1);
    const dim3 grid((n + 255) / 256);
    hipLaunchKernelGGL(metadata_kernel, grid, block, 0, stream, n, b.d_alloc, b.d_overhead, (const int64_t*)b.d_usage,
                       b.d_zone, b.n_zones, b.d_avail, b.d_sched, reinterpret_cast<unsigned long long*>(b.d_zone_sum));
    hipLaunchKernelGGL(zone_rank_kernel, dim3(1), dim3(64), 0, stream, b.n_zones, (const long long*)b.d_zone_sum,
                       b.d_zone_order, b.d_zone_rank);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // (zone rank, free memory, free cpu, name): one launch, the result sits in d_perm_b
    PrioritySort ps{n, b.n_zones, b.d_avail, b.d_avail + n, b.d_zone, b.d_zone_rank, b.d_name_rank,
                    reinterpret_cast<unsigned long long*>(b.d_keys_a), reinterpret_cast<unsigned long long*>(b.d_keys_b),
                    b.d_perm_a, b.d_perm_b};
    hipLaunchKernelGGL(priority_sort_kernel, dim3(1), dim3(kSortThreads), 0, stream, ps);
    return hipGetLastError();
}

}  // namespace gangfit
