// gangfit_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for the gang-fit decision.
//
// What is computed (reference = palantir/k8s-spark-scheduler; LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg):
//   SparkBinPack            LIB/binpack/binpack.go:60-87        driver candidate loop + driver-fit check (:69)
//   tightlyPackExecutors    LIB/binpack/pack_tightly.go:34-63   fill nodes in priority order
//   distributeExecutorsEvenly LIB/binpack/distribute_evenly.go:34-73  round-robin one executor per node per pass
//   fitEarlierDrivers       internal/extender/resource.go:224-262   FIFO replay, usage subtraction quirk
//                           (internal/extender/sparkpods.go:139-146)
//
// Formulation (one wavefront = one application; 64 lanes = 64 consecutive nodes of the executor priority order):
//   cap(n) = number of consecutive successful "add one executor, then compare" steps on node n
//          = min over dims of floor((avail - base) / exe), 0 if any (avail - base) < 0, clamped to K.
//   The 64-bit floor division is done as one f64 multiply by a per-app reciprocal plus an exact int64 correction
//   (gfx950 has no 64-bit integer divide; f64 is half rate).  Placements are committed with a DPP wave prefix
//   scan (TightlyPack) or __ballot/popcount ranks (DistributeEvenly).  Pure integer results, bit-exact.
//   The scan is LAZY like the reference: it stops at the chunk where K executors are placed.
//
// Two kernels: fit_independent_kernel (one wave per app, table read through L1/L2 with coalesced 8-byte-per-lane
// loads) and fit_fifo_chain_kernel (one workgroup walks the chain, table front resident in LDS).
// No MFMA (nothing here is a contraction) — see DESIGN.md for the roofline discussion.

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "gangfit_device.h"

namespace gangfit {

namespace {

constexpr int kWave = 64;
#ifndef GF_MF_TEAM
#define GF_MF_TEAM 1  // minimal-fragmentation, independent batch: the wavefronts of a workgroup share ONE application's passes (team_minfrag_hist); 0 = one application per wavefront
#endif
constexpr int kMfHistBins = 256;  // minimal-fragmentation: capacities below this are counted in a histogram (Orders::mf_hist)
#ifndef GF_WAVES_PER_BLOCK
#define GF_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = GF_WAVES_PER_BLOCK;  // independent-batch kernel: apps (= waves) per workgroup
constexpr int kMfTeamMax = kWavesPerBlock;          // ... and the wavefronts of a minimal-fragmentation team

// ------------------------------------------------------------------------------------------------ wave primitives

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// DPP control words (gfx9 family): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
#define GF_DPP_ROW_SHR(n) (0x110 + (n))
#define GF_DPP_ROW_BCAST15 0x142
#define GF_DPP_ROW_BCAST31 0x143

// Inclusive prefix sum over the 64 lanes of a wave, 7 DPP adds, no LDS traffic.
__device__ __forceinline__ int32_t wave_inclusive_scan(int32_t v) {
#ifdef GF_SCAN_SHFL
    int32_t x = v;
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        int32_t y = __shfl_up(x, d, kWave);
        if (lane >= d) x += y;
    }
    return x;
#else
    int32_t x = v;
    x += __builtin_amdgcn_update_dpp(0, v, GF_DPP_ROW_SHR(1), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, v, GF_DPP_ROW_SHR(2), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, v, GF_DPP_ROW_SHR(3), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_SHR(4), 0xf, 0xe, false);
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_SHR(8), 0xf, 0xc, false);
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_BCAST15, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return x;
#endif
}

__device__ __forceinline__ int32_t read_lane(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int src) {
    return (uint32_t)__builtin_amdgcn_readlane((int32_t)v, src);
}
__device__ __forceinline__ int64_t read_lane(int64_t v, int src) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)v, src);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Maximum of a signed 64-bit value over the 64 lanes (wave-uniform result): row-wise running max with row_shr 1/2/4/8,
// then row_bcast15 / row_bcast31 carry the row maxima to lane 63.  ~30 VALU instructions, no LDS.
__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#define GF_MAX_STEP(ctrl, rmask)                                                                              \
    {                                                                                                         \
        const int32_t lo = (int32_t)(uint32_t)v, hi = (int32_t)(v >> 32);                                     \
        const uint32_t tlo = (uint32_t)__builtin_amdgcn_update_dpp(lo, lo, ctrl, rmask, 0xf, false);          \
        const int32_t thi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, rmask, 0xf, false);                     \
        const int64_t t = (int64_t)(((uint64_t)(uint32_t)thi << 32) | tlo);                                   \
        v = t > v ? t : v;                                                                                    \
    }
    GF_MAX_STEP(GF_DPP_ROW_SHR(1), 0xf)
    GF_MAX_STEP(GF_DPP_ROW_SHR(2), 0xf)
    GF_MAX_STEP(GF_DPP_ROW_SHR(4), 0xf)
    GF_MAX_STEP(GF_DPP_ROW_SHR(8), 0xf)
    GF_MAX_STEP(GF_DPP_ROW_BCAST15, 0xa)
    GF_MAX_STEP(GF_DPP_ROW_BCAST31, 0xc)
#undef GF_MAX_STEP
    return read_lane(v, kWave - 1);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0): with placement stores in
// flight every barrier would stall for a global-memory round trip (measured: ~2 us per app in the FIFO chain).
// Data exchanged through this barrier must live in LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ app registers

struct App {
    int64_t drv0, drv1, drv2;
    int64_t exe0, exe1, exe2;
    double rcp0, rcp1, rcp2;  // 1.0 / exe_j (only used when exe_j > 0)
    int32_t k;
    uint32_t flags;
    uint64_t exec_off;
};

// 1 / e for cap_dim, without the ~11 double-precision instructions (scale, rcp, four fma, fmas, fixup) of an IEEE division —
// per dimension, per wavefront, on the critical path of every decision.  cap_dim needs a relative error below 2^-21 only
// (K + 1 <= 2^20 + 1 quotient units must stay below one; its proof says 2^-50 because that is what the division gave): the
// hardware estimate (V_RCP_F64, good to 2^-23 or better) plus one Newton step — error squared — is far inside that.
// gf_selftest checks cap_dim with this reciprocal against a plain 64-bit divide on adversarial operands, on the device.
__device__ __forceinline__ double fast_rcp(double e) {
    const double r = __builtin_amdgcn_rcp(e);
    return __builtin_fma(__builtin_fma(-e, r, 1.0), r, r);
}

__device__ __forceinline__ App load_app(const gf_app* __restrict__ apps, uint32_t a) {
    const gf_app* p = apps + a;  // wave-uniform address: scalar loads
    App r;
    r.drv0 = p->drv[0];
    r.drv1 = p->drv[1];
    r.drv2 = p->drv[2];
    r.exe0 = p->exe[0];
    r.exe1 = p->exe[1];
    r.exe2 = p->exe[2];
    r.k = p->k;
    r.flags = p->flags;
    r.exec_off = p->exec_off;
    r.rcp0 = r.exe0 > 0 ? fast_rcp((double)r.exe0) : 0.0;
    r.rcp1 = r.exe1 > 0 ? fast_rcp((double)r.exe1) : 0.0;
    r.rcp2 = r.exe2 > 0 ? fast_rcp((double)r.exe2) : 0.0;
    return r;
}

// ------------------------------------------------------------------------------------------------ exact capacity

// min(floor(a / e), k) for one dimension; a = avail - base.  0 <= e < 2^62, |a| < 2^63, 0 <= k <= GF_MAX_K.
//   a < 0            -> 0   (base alone already exceeds avail; holds even when e == 0)
//   e == 0           -> k   (this dimension never limits)
//   otherwise        -> f64 estimate q^ = a * (1/e): relative error <= 2^-50, so when q^ < k+1 the truncated
//                       estimate is within +-1 of floor(a/e); one exact int64 multiply-subtract fixes it.
//                       q^ >= k+1 implies a/e > k (because (k+1) * 2^-50 < 1), i.e. the clamp.
__device__ __forceinline__ int32_t cap_dim(int64_t a, int64_t e, double rcp, int32_t k) {
    if (a < 0) return 0;
    if (e == 0) return k;  // wave-uniform branch (e is per app)
    const double qf = (double)a * rcp;
    if (qf >= (double)k + 1.0) return k;
    int32_t q = (int32_t)qf;  // qf >= 0: truncation == floor
    const int64_t rem = a - (int64_t)q * e;
    if (rem < 0)
        q -= 1;
    else if (rem >= e)
        q += 1;
    return q < k ? q : k;
}

#ifdef GF_PLAIN_DIVIDE
// Reference implementation used by the self-test and by -DGF_PLAIN_DIVIDE builds: compiler-emulated 64-bit divide.
__device__ __forceinline__ int32_t cap_dim_ref(int64_t a, int64_t e, int32_t k) {
    if (a < 0) return 0;
    if (e == 0) return k;
    const int64_t q = a / e;
    return q < (int64_t)k ? (int32_t)q : k;
}
#endif

__device__ __forceinline__ int32_t cap3(int64_t a0, int64_t a1, int64_t a2, const App& app) {
#ifdef GF_PLAIN_DIVIDE
    int32_t c = cap_dim_ref(a0, app.exe0, app.k);
    int32_t m = cap_dim_ref(a1, app.exe1, app.k);
    int32_t g = cap_dim_ref(a2, app.exe2, app.k);
#else
    int32_t c = cap_dim(a0, app.exe0, app.rcp0, app.k);
    int32_t m = cap_dim(a1, app.exe1, app.rcp1, app.k);
    int32_t g = cap_dim(a2, app.exe2, app.rcp2, app.k);
#endif
    c = c < m ? c : m;
    return c < g ? c : g;
}

// cap >= 1 without any division: base + exe <= avail in every dimension (the first add-then-compare step).
// Lane masks in scalar registers (the compares write them there) instead of per-lane booleans that a __ballot turns back into a
// mask by way of a select and a second compare: "every component at least r" for 64 lanes at once, ANDed with a wave-uniform
// candidate word; low_lanes(n) = the lanes below n.
__device__ __forceinline__ uint64_t ge3_mask(int64_t a0, int64_t a1, int64_t a2, int64_t r0, int64_t r1, int64_t r2) {
    constexpr int kSGE = 39;  // signed >=
    return __builtin_amdgcn_sicmpl(a0, r0, kSGE) & __builtin_amdgcn_sicmpl(a1, r1, kSGE) & __builtin_amdgcn_sicmpl(a2, r2, kSGE);
}
__device__ __forceinline__ uint64_t low_lanes(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }
__device__ __forceinline__ bool lane_in(uint64_t uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }

__device__ __forceinline__ bool cap_ge1(int64_t a0, int64_t a1, int64_t a2, const App& app) {
    return app.exe0 <= a0 && app.exe1 <= a1 && app.exe2 <= a2;
}

// !driverResources.GreaterThan(available)  (LIB/binpack/binpack.go:69, LIB/resources/resources.go:239-241)
__device__ __forceinline__ bool driver_fits(int64_t a0, int64_t a1, int64_t a2, const App& app) {
    return app.drv0 <= a0 && app.drv1 <= a1 && app.drv2 <= a2;
}

// ------------------------------------------------------------------------------------------------ table views

// Node table read straight from global memory (L1/L2 resident after first touch): independent-batch kernel.
struct GlobalView {
    int64_t* cpu;
    int64_t* mem;
    int64_t* gpu;
    const int64_t* cmax_cpu;  // per-64-slot-chunk maxima of each dimension (upper bounds; see NodeTable::cmax)
    const int64_t* cmax_mem;
    const int64_t* cmax_gpu;
    const uint64_t* xm;  // executor-candidate bits per chunk
    const uint64_t* dm;  // driver-candidate bits per chunk (merged layout)
    uint32_t n_chunks;
    static constexpr bool kCompactScan = true;  // tightly-pack scans gather sparse chunks first (wave_tight_scan_compact)
    // can ANY slot of chunk c offer r in every dimension?  false => capacity 0 / driver does not fit, for the whole chunk
    __device__ __forceinline__ void chunk_maxima(uint32_t c, int64_t& m0, int64_t& m1, int64_t& m2) const {
        m0 = cmax_cpu[c];
        m1 = cmax_mem[c];
        m2 = cmax_gpu[c];
    }
    __device__ __forceinline__ bool chunk_may_hold(uint32_t c, int64_t r0, int64_t r1, int64_t r2) const {
        // three loads in flight together: with && the second maximum is only requested after the first one has arrived
        int64_t m0, m1, m2;
        chunk_maxima(c, m0, m1, m2);
        return (bool)((m0 >= r0) & (m1 >= r1) & (m2 >= r2));
    }
    __device__ __forceinline__ uint64_t chunk_xmask(uint32_t c) const { return xm[c]; }
    __device__ __forceinline__ uint64_t chunk_dmask(uint32_t c) const { return dm[c]; }
    __device__ __forceinline__ bool xcand(uint32_t s) const { return (xm[s >> 6] >> (s & 63)) & 1ull; }
    __device__ __forceinline__ bool dcand(uint32_t s) const { return (dm[s >> 6] >> (s & 63)) & 1ull; }
    __device__ __forceinline__ void load(uint32_t s, int64_t& a0, int64_t& a1, int64_t& a2) const {
        a0 = cpu[s];
        a1 = mem[s];
        a2 = gpu[s];
    }
    __device__ __forceinline__ void sub(uint32_t s, int64_t r0, int64_t r1, int64_t r2) const {
        cpu[s] -= r0;
        mem[s] -= r1;
        gpu[s] -= r2;
    }
};

// FIFO chain: the first `lds_slots` slots of the (mutable) working table live in LDS — the front of the executor
// priority order is what every app of the chain scans — the tail stays in global memory.
// The pointers carry explicit address spaces: with generic pointers the compiler folds the two arms into ONE flat_load
// of a selected address, and every flat access waits on vmcnt(0) — i.e. on all placement stores still in flight.
typedef __attribute__((address_space(3))) int64_t lds_i64;
typedef __attribute__((address_space(1))) int64_t glb_i64;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
struct HybridView {
    lds_i64* lcpu;  // LDS, SoA: consecutive lanes read consecutive 8-byte words (ds_read_b64, conflict-free)
    lds_i64* lmem;
    lds_i64* lgpu;
    uint32_t lds_slots;
    glb_i64* cpu;  // global
    glb_i64* mem;
    glb_i64* gpu;
    lds_i64* cmax_cpu;  // chunk maxima, LDS copy (static upper bounds: the chain only ever subtracts)
    lds_i64* cmax_mem;
    lds_i64* cmax_gpu;
    lds_u64* xm;  // candidate bit masks, LDS copies
    lds_u64* dm;
    uint32_t n_chunks;
    __device__ __forceinline__ void chunk_maxima(uint32_t c, int64_t& m0, int64_t& m1, int64_t& m2) const {
        m0 = cmax_cpu[c];
        m1 = cmax_mem[c];
        m2 = cmax_gpu[c];
    }
    __device__ __forceinline__ bool chunk_may_hold(uint32_t c, int64_t r0, int64_t r1, int64_t r2) const {
        // three loads in flight together: with && the second maximum is only requested after the first one has arrived
        int64_t m0, m1, m2;
        chunk_maxima(c, m0, m1, m2);
        return (bool)((m0 >= r0) & (m1 >= r1) & (m2 >= r2));
    }
    __device__ __forceinline__ uint64_t chunk_xmask(uint32_t c) const { return xm[c]; }
    __device__ __forceinline__ uint64_t chunk_dmask(uint32_t c) const { return dm[c]; }
    __device__ __forceinline__ bool xcand(uint32_t s) const { return (xm[s >> 6] >> (s & 63)) & 1ull; }
    __device__ __forceinline__ bool dcand(uint32_t s) const { return (dm[s >> 6] >> (s & 63)) & 1ull; }
    __device__ __forceinline__ void load(uint32_t s, int64_t& a0, int64_t& a1, int64_t& a2) const {
        if (s < lds_slots) {
            a0 = lcpu[s];
            a1 = lmem[s];
            a2 = lgpu[s];
        } else {
            a0 = cpu[s];
            a1 = mem[s];
            a2 = gpu[s];
        }
    }
    // LDS-only accessors for wave-uniformly LDS-resident ranges: no vmcnt wait is generated, so the chain never
    // stalls on placement stores that are still in flight (stores count in vmcnt on gfx9-family ISAs).
    __device__ __forceinline__ void load_lds(uint32_t s, int64_t& a0, int64_t& a1, int64_t& a2) const {
        a0 = lcpu[s];
        a1 = lmem[s];
        a2 = lgpu[s];
    }
    __device__ __forceinline__ void sub_lds(uint32_t s, int64_t r0, int64_t r1, int64_t r2) const {
        lcpu[s] -= r0;
        lmem[s] -= r1;
        lgpu[s] -= r2;
    }
    __device__ __forceinline__ void sub(uint32_t s, int64_t r0, int64_t r1, int64_t r2) const {
        if (s < lds_slots) {
            lcpu[s] -= r0;
            lmem[s] -= r1;
            lgpu[s] -= r2;
        } else {
            cpu[s] -= r0;
            mem[s] -= r1;
            gpu[s] -= r2;
        }
    }
};

// Slot j's values together with its candidate bit: the mask word and the three values are requested in ONE round trip
// (testing the bit first costs a second, dependent miss per chunk on a cold L2).  The asm keeps the compiler from sinking
// the value loads below the bit test again.
#define GF_KEEP(x) asm volatile("" : "+v"(x))
template <bool DRV, class View>
__device__ __forceinline__ bool load_with_cand(const View& V, uint32_t j, uint32_t limit, int64_t& a0, int64_t& a1,
                                               int64_t& a2) {
    a0 = a1 = a2 = 0;
    bool cand = false;
    if (j < limit) {
        V.load(j, a0, a1, a2);
        cand = DRV ? V.dcand(j) : V.xcand(j);
        GF_KEEP(a0);
        GF_KEEP(a1);
        GF_KEEP(a2);
    }
    return cand;
}

// Index tables that never change during a launch.
typedef __attribute__((address_space(3))) uint32_t lds_u32h;
struct Orders {
    const uint32_t* slot_node;
    const uint32_t* dslot;
    uint32_t n_x;
    uint32_t n_d;
    bool d_identity;  // merged layout: driver position == slot, candidates flagged by the dmask bits
    bool dpos_mask = false;  // general layout: the view's dmask is indexed by driver POSITION and must be consulted
                             // (zone views; the plain general layout accepts every position)
    // minimal-fragmentation, histogram form (gangfit_minfrag.inc: wave_minfrag_hist): 3 * kMfHistBins words of LDS private to the
    // calling wavefront, 16-byte aligned, and the snapshot's scaled int32 columns (NodeTable::ncpu ..); nullptr = the
    // pass-per-question walk on the wide table
    lds_u32h* mf_hist = nullptr;
    // a TEAM of wavefronts on one application (gangfit_minfrag.inc: team_minfrag_hist): mf_team wavefronts, this one is mf_rank; the
    // rows of wavefront r start at mf_team_base + r * 3 * kMfHistBins (mf_hist = this wavefront's), mf_words: one word per wavefront
    uint32_t mf_team = 1, mf_rank = 0;
    lds_u32h* mf_team_base = nullptr;
    lds_u32h* mf_words = nullptr;
    bool mf_lent = false;  // (an LDS array may sit at LDS address 0, which compares equal to nullptr: the flag says whether mf_hist is lent)
    const int32_t* ncpu = nullptr;
    const int32_t* nmem = nullptr;
    const int32_t* ngpu = nullptr;
    int64_t nunit0 = 1, nunit1 = 1, nunit2 = 1;
#ifdef GF_MF_PROBE  // experiment build: where a minimal-fragmentation decision's cycles go (summed over the launch's applications)
    unsigned long long* mf_probe = nullptr;
#endif
    __device__ __forceinline__ void lend_minfrag(lds_u32h* lds, const NodeTable& T) {
        mf_hist = lds;
        mf_lent = true;
        ncpu = T.ncpu;
        nmem = T.nmem;
        ngpu = T.ngpu;
        nunit0 = T.nunit[0];
        nunit1 = T.nunit[1];
        nunit2 = T.nunit[2];
    }
    __device__ __forceinline__ uint32_t driver_slot(uint32_t i) const { return d_identity ? i : dslot[i]; }
};

// ------------------------------------------------------------------------------------------------ emission

// Lane `lane` owns a run of t copies of `id` starting at out[start].  Short runs: per-lane loop.  Long runs are
// written cooperatively by all 64 lanes (coalesced) so that one huge node cannot serialise the wave.
// wt (wave-uniform; constant false everywhere but in the resident worker): the placements leave as write-through
// (system-scope) stores — the destination is read by someone else while this kernel is still running.
__device__ __forceinline__ void put_out(uint32_t* __restrict__ p, uint32_t v, bool wt) {
#if defined(GF_WK_NOSTORE) && GF_WK_NOSTORE
    if (wt) return;  // (measurement build of the worker: see gangfit_worker.inc)
#endif
    if (wt)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        *p = v;
}
__device__ __forceinline__ void emit_runs(uint32_t* __restrict__ out, int64_t start, int32_t t, uint32_t id, int lane,
                                          const bool wt = false) {
    constexpr int32_t kLong = 16;
    uint64_t long_mask = __ballot(t > kLong);
    if (t > 0 && t <= kLong) {
        for (int32_t i = 0; i < t; ++i) put_out(out + start + i, id, wt);
    }
    while (long_mask) {
        const int src = __ffsll((unsigned long long)long_mask) - 1;
        long_mask &= long_mask - 1;
        const int64_t s = read_lane(start, src);
        const int32_t n = read_lane(t, src);
        const uint32_t v = read_lane(id, src);
        for (int32_t i = lane; i < n; i += kWave) put_out(out + s + i, v, wt);
    }
}

__device__ __forceinline__ uint32_t chunk_len(uint32_t n, uint32_t b, uint32_t width) {
    return (n - b) < width ? (n - b) : width;
}

// ------------------------------------------------------------------------------------------------ wave-level decision
// One wavefront evaluates one application.  SLOTS selects what is written to out[]: slot ids (FIFO kernel; translated
// by a follow-up kernel) or the caller's node indices (independent kernel).

// 64-bit mask over chunks [64g, 64g+64): bit set when the chunk may hold request r (every dimension's maximum >= r).
// A cleared bit is a proof that every slot of the chunk has capacity 0 for an executor of size r (cap_dim: a < e -> 0;
// reserving the driver only lowers a) resp. fails the driver-fit check for a driver of size r.
// DRV selects which candidate mask must be non-empty (driver scan vs executor scan).
// cand receives lane l's candidate word of chunk 64g + l (0 beyond the limit): read_lane(cand, b) is the mask of the chunk
// whose bit b is set, so the scans need not load it again.  The maxima and the mask are requested together (one round trip).
template <bool DRV, class View>
__device__ __forceinline__ uint64_t chunk_group_mask(const View& V, uint32_t g, uint32_t chunk_limit, int64_t r0,
                                                     int64_t r1, int64_t r2, int lane, uint64_t& cand) {
    const uint32_t c = g * kWave + lane;
    bool ok = false;
    cand = 0;
    if (c < chunk_limit) {
        cand = DRV ? V.chunk_dmask(c) : V.chunk_xmask(c);
        ok = V.chunk_may_hold(c, r0, r1, r2) & (cand != 0);
    }
    return __ballot(ok);
}
template <bool DRV, class View>
__device__ __forceinline__ uint64_t chunk_group_mask(const View& V, uint32_t g, uint32_t chunk_limit, int64_t r0,
                                                     int64_t r1, int64_t r2, int lane) {
    uint64_t cand;
    return chunk_group_mask<DRV>(V, g, chunk_limit, r0, r1, r2, lane, cand);
}

// First position p in [from, n_d) of driverNodePriorityOrder whose node passes the driver-fit check, else -1.
template <class View>
__device__ __forceinline__ int64_t wave_first_fitting_driver(const View& V, const Orders& O, const App& app,
                                                             uint32_t from, int lane, unsigned long long& visited) {
    if (O.d_identity) {  // position == slot: prune whole chunks with the maxima index
        const uint32_t dc = (O.n_d + kWave - 1) / kWave;
        for (uint32_t g = (from / kWave) / kWave; g * kWave < dc; ++g) {
            uint64_t cand;
            uint64_t m = chunk_group_mask<true>(V, g, dc, app.drv0, app.drv1, app.drv2, lane, cand);
            visited += kWave;
            while (m) {
                const int bit = __ffsll((unsigned long long)m) - 1;
                const uint32_t c = g * kWave + (uint32_t)bit;
                m &= m - 1;
                const uint32_t i = c * kWave + lane;
                const uint64_t cdm = (uint64_t)read_lane((int64_t)cand, bit);
                bool fit = false;
                if (i < O.n_d && i >= from && ((cdm >> lane) & 1ull)) {
                    int64_t a0, a1, a2;
                    V.load(i, a0, a1, a2);
                    fit = driver_fits(a0, a1, a2, app);
                }
                visited += chunk_len(O.n_d, c * kWave, kWave);
                const uint64_t fm = __ballot(fit);
                if (fm) return (int64_t)c * kWave + (__ffsll((unsigned long long)fm) - 1);
            }
        }
        return -1;
    }
    for (uint32_t b = from; b < O.n_d; b += kWave) {
        const uint32_t i = b + lane;
        bool fit = false;
        if (i < O.n_d && (!O.dpos_mask || V.dcand(i))) {
            int64_t a0, a1, a2;
            V.load(O.driver_slot(i), a0, a1, a2);
            fit = driver_fits(a0, a1, a2, app);
        }
        visited += chunk_len(O.n_d, b, kWave);
        const uint64_t m = __ballot(fit);
        if (m) return (int64_t)b + (__ffsll((unsigned long long)m) - 1);
    }
    return -1;
}

// O(N) driver choice on the fallback path: first position p >= from in driver order with
//   fit(p) && total(p) >= K,  total = S            if the node is not an executor candidate
//                                   = S - c0 + cd   otherwise           (SURVEY.md section 8 "O(N) driver choice")
template <class View>
__device__ __forceinline__ int64_t wave_next_feasible_driver(const View& V, const Orders& O, const App& app,
                                                             uint32_t from, int64_t S, int lane,
                                                             unsigned long long& visited) {
    for (uint32_t b = from; b < O.n_d; b += kWave) {
        const uint32_t i = b + lane;
        bool ok = false;
        if (i < O.n_d && ((!O.d_identity && !O.dpos_mask) || V.dcand(i))) {
            const uint32_t s = O.driver_slot(i);
            int64_t a0, a1, a2;
            V.load(s, a0, a1, a2);
            if (driver_fits(a0, a1, a2, app)) {
                int64_t total = S;
                if (s < O.n_x && V.xcand(s)) {
                    const int32_t c0 = cap3(a0, a1, a2, app);
                    const int32_t cd = cap3(a0 - app.drv0, a1 - app.drv1, a2 - app.drv2, app);
                    total = S - c0 + cd;
                }
                ok = total >= (int64_t)app.k;
            }
        }
        visited += chunk_len(O.n_d, b, kWave);
        const uint64_t m = __ballot(ok);
        if (m) return (int64_t)b + (__ffsll((unsigned long long)m) - 1);
    }
    return -1;
}

// Lane l's view of chunk l (group 0 of the chunk index) for both roles: the three maxima and the two candidate words.
// Nothing here depends on the application, so the independent kernel requests it BEFORE its app record: the two misses
// overlap instead of following each other.
struct Group0 {
    int64_t m0 = 0, m1 = 0, m2 = 0;
    uint64_t dcand = 0, xcand = 0;
    bool ind = false, inx = false;
};
template <class View>
__device__ __forceinline__ Group0 load_group0(const View& V, const Orders& O, int lane) {
    Group0 g;
    const uint32_t dc = (O.n_d + kWave - 1) / kWave, xc = (O.n_x + kWave - 1) / kWave;
    const uint32_t c = (uint32_t)lane;
    g.ind = c < dc && c < V.n_chunks;  // general layout: driver positions may outnumber the slots
    g.inx = c < xc && c < V.n_chunks;
    if (g.ind | g.inx) {
        V.chunk_maxima(c, g.m0, g.m1, g.m2);
        if (g.ind) g.dcand = V.chunk_dmask(c);
        if (g.inx) g.xcand = V.chunk_xmask(c);
    }
    return g;
}

template <class View, class = void>
struct HasCompactScan {
    static constexpr bool value = false;
};
template <class View>
struct HasCompactScan<View, decltype((void)View::kCompactScan)> {
    static constexpr bool value = true;
};

// What wave_decide_merged already holds when the executor scan starts: the chunk mask of group 0 and the slots of the
// first candidate chunk, requested together with the driver's (a decision is a chain of dependent global round trips —
// every launch starts with cold L2s — so requests that do not depend on each other must be in flight together).
struct ScanPre {
    bool on = false;
    uint64_t m = 0;     // group 0: chunks that may hold an executor
    uint64_t cand = 0;  // lane l: executor-candidate word of chunk l
    int c = -1;         // preloaded chunk, -1 = none
    int64_t a0 = 0, a1 = 0, a2 = 0;
    uint32_t node = 0;  // slot_node of this lane's slot in chunk c (SLOTS = false)
};

// tightlyPackExecutors with the driver reserved on slot ds.  Returns the sum of clamped capacities over the visited
// prefix (>= K  <=>  feasible; the scan stops at the first chunk where K is reached).  Writes placements.
// Only chunks the maxima index cannot rule out are loaded; a ruled-out chunk contributes exactly 0.
template <class View, bool SLOTS>
__device__ __forceinline__ int64_t wave_tight_scan(const View& V, const Orders& O, const App& app, uint32_t ds,
                                                   uint32_t* __restrict__ out, int lane,
                                                   unsigned long long& visited, const ScanPre& pre = ScanPre(),
                                                   const bool wt = false) {
    const int64_t K = app.k;
    const uint32_t xc = (O.n_x + kWave - 1) / kWave;
    int64_t taken = 0;
    for (uint32_t g = 0; g * kWave < xc; ++g) {
        uint64_t cand;
        uint64_t m;
        if (pre.on && g == 0) {
            m = pre.m;
            cand = pre.cand;
        } else {
            m = chunk_group_mask<false>(V, g, xc, app.exe0, app.exe1, app.exe2, lane, cand);
        }
        visited += kWave;
        while (m) {
            const int bit = __ffsll((unsigned long long)m) - 1;
            const uint32_t c = g * kWave + (uint32_t)bit;
            m &= m - 1;
            const uint32_t j = c * kWave + lane;
            const uint64_t cxm = (uint64_t)read_lane((int64_t)cand, bit);
            const bool have = pre.on && (int)c == pre.c;  // wave-uniform
            int32_t cp = 0;
            uint32_t node = 0;
            if (j < O.n_x && ((cxm >> lane) & 1ull)) {
                int64_t a0, a1, a2;
                if (have) {
                    a0 = pre.a0;
                    a1 = pre.a1;
                    a2 = pre.a2;
                    node = pre.node;
                } else {
                    V.load(j, a0, a1, a2);
                    if (!SLOTS) node = O.slot_node[j];  // requested with the slot, not after the capacities are known
                }
                if (j == ds) {
                    a0 -= app.drv0;
                    a1 -= app.drv1;
                    a2 -= app.drv2;
                }
                if (cap_ge1(a0, a1, a2, app)) cp = cap3(a0, a1, a2, app);  // no division for slots that hold nothing
            }
            visited += chunk_len(O.n_x, c * kWave, kWave);
            const int32_t incl = wave_inclusive_scan(cp);
            const int32_t tot = read_lane(incl, kWave - 1);
            if (tot > 0) {
                const int64_t start = taken + (int64_t)(incl - cp);
                const int64_t room = K - start;
                const int32_t t = room <= 0 ? 0 : (room < (int64_t)cp ? (int32_t)room : cp);
                const uint32_t id = SLOTS ? j : node;
                emit_runs(out, start, t, id, lane, wt);
            }
            taken += tot;
            if (taken >= K) return taken;
        }
    }
    return taken;
}

// tightlyPackExecutors for tables read from global memory, same result as wave_tight_scan.  A launch is as slow as its
// slowest wavefront, and that is the gang whose executors sit on a few slots of many chunks (gpu executors on the headline
// cluster: about six gpu nodes per 64-slot chunk with a device or two left, 7-20 chunks per gang).  A chunk visit costs
// ~450 instructions whatever the number of slots that can hold an executor (exact int64 capacities, the DPP scan, the run
// emission), so the slots that pass the cheap "holds at least one" test are first GATHERED, in slot order, into one
// 64-lane batch (ds_permute: no LDS memory involved) and the expensive part runs once per batch instead of once per
// chunk.  A chunk with many such slots is its own batch and is not moved.  The batch is processed as soon as it holds
// K - taken slots (each is good for at least one executor, so the scan ends there — as lazy as the reference's loop up
// to the chunk it stops in), when the next chunk does not fit in, and at the end of the candidates.
template <class View, bool SLOTS>
__device__ __forceinline__ int64_t wave_tight_scan_compact(const View& V, const Orders& O, const App& app, uint32_t ds,
                                                           uint32_t* __restrict__ out, int lane,
                                                           unsigned long long& visited, const ScanPre& pre = ScanPre(),
                                                           const bool wt = false) {
    constexpr uint32_t kDenseLanes = 20;
    const int64_t K = app.k;
    const uint32_t xc = (O.n_x + kWave - 1) / kWave;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    int64_t taken = 0;
    int64_t q0 = 0, q1 = 0, q2 = 0;  // the batch: table values (driver already reserved) and what a placement names
    uint32_t qid = 0;
    uint32_t fill = 0;  // lanes [0, fill) of the batch are occupied (wave-uniform)
    // capacities, prefix, run emission for the lanes marked live; true when K is reached
    auto process = [&](bool live, int64_t a0, int64_t a1, int64_t a2, uint32_t id) {
        const int32_t cp = live ? cap3(a0, a1, a2, app) : 0;
        const int32_t incl = wave_inclusive_scan(cp);
        const int32_t tot = read_lane(incl, kWave - 1);
        const int64_t start = taken + (int64_t)(incl - cp);
        const int64_t room = K - start;
        const int32_t t = room <= 0 ? 0 : (room < (int64_t)cp ? (int32_t)room : cp);
        emit_runs(out, start, t, id, lane, wt);
        taken += tot;
        return taken >= K;
    };
    for (uint32_t g = 0; g * kWave < xc; ++g) {
        uint64_t cand;
        uint64_t m;
        if (pre.on && g == 0) {
            m = pre.m;
            cand = pre.cand;
        } else {
            m = chunk_group_mask<false>(V, g, xc, app.exe0, app.exe1, app.exe2, lane, cand);
        }
        visited += kWave;
        while (m) {
            const int bit = __ffsll((unsigned long long)m) - 1;
            const uint32_t c = g * kWave + (uint32_t)bit;
            m &= m - 1;
            const uint32_t j = c * kWave + lane;
            const uint64_t cxm = (uint64_t)read_lane((int64_t)cand, bit);
            const bool have = pre.on && (int)c == pre.c;  // wave-uniform
            const uint64_t in_m = cxm & low_lanes(chunk_len(O.n_x, c * kWave, kWave));  // candidate slots of the order in this chunk
            const bool in = lane_in(in_m);
            int64_t a0 = 0, a1 = 0, a2 = 0;
            uint32_t node = 0;
            if (in) {
                if (have) {
                    a0 = pre.a0;
                    a1 = pre.a1;
                    a2 = pre.a2;
                    node = pre.node;
                } else {
                    V.load(j, a0, a1, a2);
                    if (!SLOTS) node = O.slot_node[j];  // requested with the slot, not after the capacities are known
                }
                if (j == ds) {
                    a0 -= app.drv0;
                    a1 -= app.drv1;
                    a2 -= app.drv2;
                }
            }
            visited += chunk_len(O.n_x, c * kWave, kWave);
            const uint64_t fm = ge3_mask(a0, a1, a2, app.exe0, app.exe1, app.exe2) & in_m;
            const bool fit = lane_in(fm);
            const uint32_t n = (uint32_t)__popcll((unsigned long long)fm);
            if (n == 0) continue;
            const uint32_t id = SLOTS ? j : node;
            if (fill + n > (uint32_t)kWave) {  // no room: the batch first (slot order is kept)
                if (process((uint32_t)lane < fill, q0, q1, q2, qid)) return taken;
                fill = 0;
            }
            if (fill == 0 && n > kDenseLanes) {  // a dense chunk is its own batch
                if (process(fit, a0, a1, a2, id)) return taken;
                continue;
            }
            // gather: the r-th fitting lane goes to lane fill + r; the other lanes take the remaining destinations so that
            // the permutation is a bijection (every lane must take part in ds_permute), and what they send is ignored
            const uint32_t rank = (uint32_t)__popcll((unsigned long long)(fm & lt_mask));
            const uint32_t dest = fit ? fill + rank : ((fill + n + ((uint32_t)lane - rank)) & 63u);
            const int addr = (int)(dest << 2);
            const uint32_t r0l = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)(uint32_t)a0);
            const uint32_t r0h = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)(uint32_t)((uint64_t)a0 >> 32));
            const uint32_t r1l = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)(uint32_t)a1);
            const uint32_t r1h = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)(uint32_t)((uint64_t)a1 >> 32));
            const uint32_t r2l = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)(uint32_t)a2);
            const uint32_t r2h = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)(uint32_t)((uint64_t)a2 >> 32));
            const uint32_t rid = (uint32_t)__builtin_amdgcn_ds_permute(addr, (int)id);
            const bool mine = (uint32_t)lane >= fill && (uint32_t)lane < fill + n;
            if (mine) {
                q0 = (int64_t)(((uint64_t)r0h << 32) | r0l);
                q1 = (int64_t)(((uint64_t)r1h << 32) | r1l);
                q2 = (int64_t)(((uint64_t)r2h << 32) | r2l);
                qid = rid;
            }
            fill += n;
            if (taken + (int64_t)fill >= K) {  // enough slots: each of them holds at least one executor
                process((uint32_t)lane < fill, q0, q1, q2, qid);
                return taken;
            }
        }
    }
    if (fill > 0) process((uint32_t)lane < fill, q0, q1, q2, qid);
    return taken;
}

// Pass r = 1 of distributeExecutorsEvenly, lazily: every node with cap >= 1, in order, until K are placed.
// Records the surviving slots (needed for passes >= 2) in surv[].  Returns the number of nodes with cap >= 1 seen
// (>= K means the app is placed entirely by pass 1).
template <class View, bool SLOTS>
__device__ __forceinline__ int64_t wave_even_pass1(const View& V, const Orders& O, const App& app, uint32_t ds,
                                                   uint32_t* __restrict__ out, uint32_t* __restrict__ surv, int lane,
                                                   unsigned long long& visited, const ScanPre& pre = ScanPre()) {
    const int64_t K = app.k;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t xc = (O.n_x + kWave - 1) / kWave;
    int64_t taken = 0;
    for (uint32_t g = 0; g * kWave < xc; ++g) {
        uint64_t cand;
        uint64_t cm;
        if (pre.on && g == 0) {
            cm = pre.m;
            cand = pre.cand;
        } else {
            cm = chunk_group_mask<false>(V, g, xc, app.exe0, app.exe1, app.exe2, lane, cand);
        }
        visited += kWave;
        while (cm) {
            const int bit = __ffsll((unsigned long long)cm) - 1;
            const uint32_t c = g * kWave + (uint32_t)bit;
            cm &= cm - 1;
            const uint32_t j = c * kWave + lane;
            const uint64_t cxm = (uint64_t)read_lane((int64_t)cand, bit);
            const bool have = pre.on && (int)c == pre.c;  // wave-uniform
            bool flag = false;
            uint32_t node = 0;
            if (j < O.n_x && ((cxm >> lane) & 1ull)) {
                int64_t a0, a1, a2;
                if (have) {
                    a0 = pre.a0;
                    a1 = pre.a1;
                    a2 = pre.a2;
                    node = pre.node;
                } else {
                    V.load(j, a0, a1, a2);
                    if (!SLOTS) node = O.slot_node[j];
                }
                if (j == ds) {
                    a0 -= app.drv0;
                    a1 -= app.drv1;
                    a2 -= app.drv2;
                }
                flag = cap_ge1(a0, a1, a2, app);
            }
            visited += chunk_len(O.n_x, c * kWave, kWave);
            const uint64_t m = __ballot(flag);
            const int64_t pos = taken + (int64_t)__popcll((unsigned long long)(m & lt_mask));
            if (flag && pos < K) {
                out[pos] = SLOTS ? j : node;
                surv[pos] = j;
            }
            taken += (int64_t)__popcll((unsigned long long)m);
            if (taken >= K) return taken;
        }
    }
    return taken;
}

// Passes r >= 2 when pass 1 found m1 < K nodes.  surv[0..m1) = surviving slots in order; caps[] receives their
// clamped capacities.  Returns S = sum of capacities (feasible <=> S >= K) and, when feasible, completes out[].
template <class View, bool SLOTS>
__device__ __forceinline__ int64_t wave_even_general(const View& V, const Orders& O, const App& app, uint32_t ds,
                                                     uint32_t* __restrict__ out, const uint32_t* __restrict__ surv,
                                                     uint32_t* __restrict__ caps, int64_t m1, int lane) {
    const int64_t K = app.k;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // surv[] was written rank-wise by other lanes in pass 1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int64_t S = 0;
    for (int64_t b = 0; b < m1; b += kWave) {
        const int64_t i = b + lane;
        int32_t c = 0;
        if (i < m1) {
            const uint32_t j = surv[i];
            int64_t a0, a1, a2;
            V.load(j, a0, a1, a2);
            if (j == ds) {
                a0 -= app.drv0;
                a1 -= app.drv1;
                a2 -= app.drv2;
            }
            c = cap3(a0, a1, a2, app);
            caps[i] = (uint32_t)c;
        }
        const int32_t incl = wave_inclusive_scan(c);
        S += read_lane(incl, kWave - 1);
    }
    if (S < K) return S;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // caps[] are re-read below by other lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // passes r = 2, 3, ...: nodes with cap >= r, in order, appended until K placements exist
    int64_t pos = m1;
    for (int32_t r = 2; pos < K; ++r) {
        for (int64_t b = 0; b < m1 && pos < K; b += kWave) {
            const int64_t i = b + lane;
            bool flag = false;
            uint32_t j = 0;
            if (i < m1) {
                flag = (int32_t)caps[i] >= r;
                j = surv[i];
            }
            const uint64_t m = __ballot(flag);
            const int64_t p = pos + (int64_t)__popcll((unsigned long long)(m & lt_mask));
            if (flag && p < K) out[p] = SLOTS ? j : O.slot_node[j];
            pos += (int64_t)__popcll((unsigned long long)m);
        }
    }
    return S;
}

// floor(a / e) for 0 <= a < 2^30 as a multiplication (Granlund & Montgomery, "Division by invariant integers using
// multiplication", N = 30): with l = ceil(log2 e) and m = ceil(2^(30 + l) / e) < 2^31, floor(a / e) == (m a) >> (30 + l), i.e.
// mulhi(2 m, 2 a) >> l — shift, multiply-high, shift: three instructions per dimension where the float estimate with its
// exact correction took fourteen.  The request's three multipliers are computed once per application (prepare_app).
__device__ __forceinline__ uint32_t narrow_shift(int32_t e) {  // l = ceil(log2 e); 0 for e <= 1
    return e <= 1 ? 0u : 32u - (uint32_t)__builtin_clz((uint32_t)e - 1u);
}
__device__ __forceinline__ void narrow_magic(int32_t e, uint32_t& mag, uint32_t& sh) {
    if (e <= 0) {
        mag = 0;
        sh = 0;
        return;
    }
    const uint32_t l = narrow_shift(e);
    const uint64_t m = ((1ull << (30u + l)) + (uint64_t)(uint32_t)e - 1ull) / (uint64_t)(uint32_t)e;
    mag = (uint32_t)(2ull * m);
    sh = l;
}

#include "gangfit_minfrag.inc"

struct Decision {
    bool feasible;
    uint32_t ds;    // driver slot
    int64_t pass1;  // DistributeEvenly: number of pass-1 placements (= distinct executor nodes); TightlyPack: unused
    uint32_t ds_node = GF_NO_NODE;  // slot_node[ds] when the decision already holds it (requested with the driver's chunk)
};

template <int ALGO, class View, bool SLOTS>
__device__ __forceinline__ int64_t wave_pack(const View& V, const Orders& O, const App& app, uint32_t ds,
                                             uint32_t* __restrict__ out, uint32_t* __restrict__ scratch_a,
                                             uint32_t* __restrict__ scratch_b, int lane, int64_t& pass1,
                                             unsigned long long& xvis, const ScanPre& pre = ScanPre(), const bool wt = false) {
    if (ALGO == GF_ALGO_TIGHTLY_PACK) {  // (wt: tightly-pack only — the other packers' callers never set it)
        if constexpr (HasCompactScan<View>::value)
            return wave_tight_scan_compact<View, SLOTS>(V, O, app, ds, out, lane, xvis, pre, wt);
        else
            return wave_tight_scan<View, SLOTS>(V, O, app, ds, out, lane, xvis, pre, wt);
    }
    if (ALGO == GF_ALGO_MINIMAL_FRAGMENTATION) return wave_minfrag<View, SLOTS>(V, O, app, ds, out, lane, xvis);
    pass1 = wave_even_pass1<View, SLOTS>(V, O, app, ds, out, scratch_a, lane, xvis, pre);
    if (pass1 >= (int64_t)app.k) return pass1;
    return wave_even_general<View, SLOTS>(V, O, app, ds, out, scratch_a, scratch_b, pass1, lane);
}

// Recovery after the first fitting candidate (position p0, slot ds0) could not host the gang: S_d is the exact
// capacity total with the driver reserved on ds0 (< K).  Finds the first later candidate that works and packs with it.
// Same answer as the reference's retry loop (binpack.go:67-85) in O(N) instead of O(|D| N).
template <int ALGO, class View, bool SLOTS>
__device__ __forceinline__ Decision wave_fallback(const View& V, const Orders& O, const App& app, int64_t p0,
                                                  uint32_t ds0, int64_t S_d, uint32_t* __restrict__ out,
                                                  uint32_t* __restrict__ scratch_a, uint32_t* __restrict__ scratch_b,
                                                  int lane, unsigned long long& xvis, unsigned long long& dvis,
                                                  const bool wt = false) {
    Decision dec;
    dec.feasible = false;
    dec.ds = ds0;
    dec.pass1 = 0;
    const int64_t K = app.k;
    int64_t S = S_d;
    if (ds0 < O.n_x && V.xcand(ds0)) {  // undo the driver reservation: S = S_d + cap(ds0, 0) - cap(ds0, drv)
        int64_t a0, a1, a2;
        V.load(ds0, a0, a1, a2);
        S = S_d + cap3(a0, a1, a2, app) - cap3(a0 - app.drv0, a1 - app.drv1, a2 - app.drv2, app);
    }
    if (S < K) return dec;
    const int64_t p1 = wave_next_feasible_driver(V, O, app, (uint32_t)p0 + 1, S, lane, dvis);
    if (p1 < 0) return dec;
    const uint32_t ds = O.driver_slot((uint32_t)p1);
    int64_t pass1 = 0;
    const int64_t S1 = wave_pack<ALGO, View, SLOTS>(V, O, app, ds, out, scratch_a, scratch_b, lane, pass1, xvis, ScanPre(), wt);
    dec.feasible = S1 >= K;  // always true here; kept as a guard so a logic error shows up as a parity failure
    dec.ds = ds;
    dec.pass1 = pass1;
    return dec;
}

// SparkBinPack for one app by one wave.  out = exec_nodes + exec_off.  scratch_a / scratch_b: K uint32 each.
template <int ALGO, class View, bool SLOTS>
__device__ __forceinline__ Decision wave_decide(const View& V, const Orders& O, const App& app,
                                                uint32_t* __restrict__ out, uint32_t* __restrict__ scratch_a,
                                                uint32_t* __restrict__ scratch_b, int lane, unsigned long long& xvis,
                                                unsigned long long& dvis, const Group0* g0p = nullptr,
                                                const SparseTable* gpu_view = nullptr, const bool wt = false) {
    Decision dec;
    dec.feasible = false;
    dec.ds = 0;
    dec.pass1 = 0;
    const int64_t K = app.k;
    int64_t p0 = -1;
    uint32_t p0_node = GF_NO_NODE;
    ScanPre pre;
    // executors that need a gpu are packed from the compact table of gpu nodes (SparseTable); wave-uniform
    const bool sparse = gpu_view != nullptr && gpu_view->n_x != 0 && app.exe2 > 0 && K > 0 && O.d_identity &&
                        (ALGO == GF_ALGO_TIGHTLY_PACK || ALGO == GF_ALGO_DISTRIBUTE_EVENLY);
    uint32_t p0_sub = GF_NO_NODE;
    bool p0_sub_known = false;  // the driver came from the first step below, which looked its sub-slot up on the way
    if (O.d_identity && ALGO != GF_ALGO_MINIMAL_FRAGMENTATION) {
        // Merged layout (driver position == slot): the chunk masks of group 0 for BOTH roles in one round trip (the
        // maxima are the same words), then the first candidate chunk of both roles — and the node ids — in one more.
        const uint32_t dc = (O.n_d + kWave - 1) / kWave;
        const Group0 g0 = g0p != nullptr ? *g0p : load_group0(V, O, lane);
        const uint64_t dcand = g0.dcand;
        pre.cand = g0.xcand;
        // (g0.ind / g0.inx = "lane < chunks of the order, and < chunks of the table": scalar masks)
        const uint32_t xc0 = (O.n_x + kWave - 1) / kWave;
        const uint64_t ind_m = low_lanes(dc < V.n_chunks ? dc : V.n_chunks), inx_m = low_lanes(xc0 < V.n_chunks ? xc0 : V.n_chunks);
        constexpr int kNE = 33;
        const uint64_t md = ge3_mask(g0.m0, g0.m1, g0.m2, app.drv0, app.drv1, app.drv2) & ind_m &
                            __builtin_amdgcn_uicmpl((unsigned long)dcand, 0ul, kNE);
        pre.m = sparse ? 0ull
                       : (ge3_mask(g0.m0, g0.m1, g0.m2, app.exe0, app.exe1, app.exe2) & inx_m &
                          __builtin_amdgcn_uicmpl((unsigned long)pre.cand, 0ul, kNE));
        pre.on = !sparse;
        dvis += kWave;
        const int bd = md ? __ffsll((unsigned long long)md) - 1 : -1;
        pre.c = (K > 0 && pre.m) ? __ffsll((unsigned long long)pre.m) - 1 : -1;
        const uint32_t limit = O.n_d > O.n_x ? O.n_d : O.n_x;
        int64_t d0 = 0, d1 = 0, d2 = 0;
        const uint32_t id = (uint32_t)(bd < 0 ? 0 : bd) * kWave + lane, ix = (uint32_t)(pre.c < 0 ? 0 : pre.c) * kWave + lane;
        uint32_t dnode = GF_NO_NODE, dsub = GF_NO_NODE;
        if (bd >= 0 && id < limit) {
            V.load(id, d0, d1, d2);
            if (!SLOTS) dnode = O.slot_node[id];  // the result needs the driver's node id: not one more round trip at the end
            if (sparse) dsub = gpu_view->sub_of_slot[id];  // where the driver's reservation lands in the compact table
        }
        if (pre.c >= 0 && ix < limit) {
            if (pre.c != bd) V.load(ix, pre.a0, pre.a1, pre.a2);
            if (!SLOTS) pre.node = O.slot_node[ix];
        }
        if (pre.c >= 0 && pre.c == bd) {
            pre.a0 = d0;
            pre.a1 = d1;
            pre.a2 = d2;
        }
        uint32_t from = dc > (uint32_t)kWave ? (uint32_t)kWave * kWave : O.n_d;  // where the generic search resumes
        if (bd >= 0) {
            const uint64_t cdm = (uint64_t)read_lane((int64_t)dcand, bd);
            dvis += chunk_len(O.n_d, (uint32_t)bd * kWave, kWave);
            // (lanes whose position is below n_d, candidate bits of the chunk, the driver-fit check of binpack.go:69: masks)
            const uint64_t fm = ge3_mask(d0, d1, d2, app.drv0, app.drv1, app.drv2) & cdm & low_lanes(chunk_len(O.n_d, (uint32_t)bd * kWave, kWave));
            if (fm) {
                const int fl = __ffsll((unsigned long long)fm) - 1;
                p0 = (int64_t)bd * kWave + fl;
                p0_node = read_lane(dnode, fl);
                p0_sub = read_lane(dsub, fl);
                p0_sub_known = true;
            }
            from = ((uint32_t)bd + 1) * kWave;
        }
        if (p0 < 0 && (bd >= 0 || dc > (uint32_t)kWave))  // a stale maximum, or candidates beyond group 0
            p0 = wave_first_fitting_driver(V, O, app, from, lane, dvis);
    } else {
        // (1) first driver candidate that passes the driver-fit check (binpack.go:67-71)
        p0 = wave_first_fitting_driver(V, O, app, 0, lane, dvis);
    }
    if (p0 < 0) return dec;
    const uint32_t ds = O.driver_slot((uint32_t)p0);
    dec.ds = ds;
    dec.ds_node = p0_node;
    if (K == 0) {  // pack_tightly.go:42-44 / distribute_evenly.go:46-48: nothing to place
        dec.feasible = true;
        return dec;
    }
    int64_t pass1 = 0;
    // (SLOTS callers — the zone packers' one-launch kernel — hand in a view whose `slot_node` names SLOTS of the full table and whose
    //  candidate words are the zone's)
    if constexpr (std::is_same<View, GlobalView>::value) {
        if (sparse) {
            // (2s) the same pack over the compact table of gpu nodes.  S_s = the capacity total with the driver reserved,
            //      exactly what the full order would give (every node left out has capacity 0 for this request).
            if (!p0_sub_known) p0_sub = gpu_view->sub_of_slot[ds];  // the driver came from the generic search
            const GlobalView VS{const_cast<int64_t*>(gpu_view->cpu), const_cast<int64_t*>(gpu_view->mem),
                                const_cast<int64_t*>(gpu_view->gpu), gpu_view->cmax, gpu_view->cmax + gpu_view->n_chunks,
                                gpu_view->cmax + 2 * (size_t)gpu_view->n_chunks, gpu_view->xmask, gpu_view->xmask,
                                gpu_view->n_chunks};
            const Orders OS{gpu_view->slot_node, nullptr, gpu_view->n_x, 0u, true};
            const int64_t S_s = wave_pack<ALGO, GlobalView, false>(VS, OS, app, p0_sub, out, scratch_a, scratch_b, lane, pass1,
                                                                   xvis, ScanPre(), wt);
            if (S_s >= K) {
                dec.feasible = true;
                dec.pass1 = pass1;
                return dec;
            }
            // short: without the driver's reservation the total is S = S_s + cap(ds, 0) - cap(ds, drv) (exact below K: every
            // term is).  Below K no other driver candidate can help (binpack.go:67-85 tries them all against the same
            // nodes); otherwise — rare — the decision is redone on the full order below.
            int64_t S = S_s;
            if (p0_sub != GF_NO_NODE) {
                int64_t a0, a1, a2;
                VS.load(p0_sub, a0, a1, a2);
                S += cap3(a0, a1, a2, app) - cap3(a0 - app.drv0, a1 - app.drv1, a2 - app.drv2, app);
            }
            if (S < K) return dec;
        }
    }
    // (2) executors with the driver reserved on that candidate — the common case ends here
    pass1 = 0;
    const int64_t S_d = wave_pack<ALGO, View, SLOTS>(V, O, app, ds, out, scratch_a, scratch_b, lane, pass1, xvis, pre, wt);
    if (S_d >= K) {
        dec.feasible = true;
        dec.pass1 = pass1;
        return dec;
    }
    // (3) rare: the first candidate's node is where the executors were needed
    return wave_fallback<ALGO, View, SLOTS>(V, O, app, p0, ds, S_d, out, scratch_a, scratch_b, lane, xvis, dvis, wt);
}

// sparkResourceUsage + SubtractUsageIfExists (internal/extender/sparkpods.go:139-146, LIB/resources/resources.go:129-135)
// applied from a placement list of SLOT ids: ONE executor request per distinct executor node; the driver request only
// if no executor sits on the driver node.  Wave-level; used on the slow path of the FIFO kernel.
template <int ALGO, class View>
__device__ __forceinline__ void wave_commit_from_list(const View& V, const App& app, const Decision& dec,
                                                      const uint32_t* __restrict__ out, int lane) {
    const int64_t K = app.k;
    bool driver_hosts_exec = false;
    // out[] was written by other lanes of this wave: make it visible before re-reading it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // tightly-pack and minimal-fragmentation placements are node-major runs (every node occupies ONE run)
    constexpr bool kRuns = ALGO != GF_ALGO_DISTRIBUTE_EVENLY;
    const int64_t first_region = kRuns ? K : (dec.pass1 < K ? dec.pass1 : K);
    for (int64_t b = 0; b < K; b += kWave) {
        const int64_t i = b + lane;
        bool first = false;
        uint32_t s = GF_NO_NODE;
        if (i < K) {
            s = out[i];
            if (kRuns)
                first = (i == 0) || (out[i - 1] != s);
            else
                first = i < first_region;  // pass 1 lists every executor node exactly once
        }
        if (first) V.sub(s, app.exe0, app.exe1, app.exe2);
        if (__ballot(i < K && s == dec.ds)) driver_hosts_exec = true;
    }
    if (!driver_hosts_exec && lane == 0) V.sub(dec.ds, app.drv0, app.drv1, app.drv2);
}

// ------------------------------------------------------------------------------------------------ independent batch

// One wave per app, 4 apps per workgroup.  Grid = ceil(n_apps / 4) >> 256 CUs at the target sizes.
#ifndef GF_IND_WAVES_PER_EU
#define GF_IND_WAVES_PER_EU 0  // experiment switch: > 0 asks the compiler for that many wavefronts per SIMD (VGPR and SGPR budget)
#endif
#if GF_IND_WAVES_PER_EU == 96
#define GF_IND_OCC __attribute__((amdgpu_num_sgpr(96), amdgpu_num_vgpr(64)))
#elif GF_IND_WAVES_PER_EU > 0
#define GF_IND_OCC __attribute__((amdgpu_waves_per_eu(GF_IND_WAVES_PER_EU, GF_IND_WAVES_PER_EU)))
#else
#define GF_IND_OCC
#endif
// One wavefront per application, four wavefronts per workgroup.  (Round 4 measured two applications per wavefront for batches
// that need more than one round of wavefronts — both records requested together, the second parked in two VGPR lanes while the
// first is decided, 5 000 wavefronts for config 3 instead of 10 000: 13.0-13.2 us against 11.5 us, profiles/r4a_variants.txt.
// The second decision starts behind the first one's stores and the kernel drops from seven to five wavefronts per SIMD; removed.)
// (Round 4 also measured a variant that sends the answers of a lone blocking batch to pinned host memory as write-through
// stores and announces its own completion there — arrival counters, a sequence word the caller polls — instead of the stream
// wait: 30.3 us per blocking 1 000-application call against 23.3 us; 13 000 dword write-through stores over the host link and
// their acknowledgement cost more than the kernel-end write-back and the completion signal they replace; removed.)
// gf_fit_feasible, device side (fit_independent_kernel<.., true>, fit_zoned_fused_kernel<.., true>): how HasCapacity of every
// application reaches the caller without the kernel having to end first.  `words`: ceil(n_apps / 4) words of DEVICE memory, zero
// between launches, one byte per application.
// feasible_announce: a deciding wavefront leaves 0x80 | HasCapacity in its byte with an atomic OR that returns nothing — it waits
// for nothing, not even for its own placement stores — and ends.
__device__ __forceinline__ void feasible_announce(uint32_t* words, uint32_t ai, bool feasible) {
    (void)__hip_atomic_fetch_or(words + (ai >> 2), (0x80u | (feasible ? 1u : 0u)) << (8u * (ai & 3u)), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT);
}
// feasible_collect: one wavefront of one more workgroup (the grid's last) watches the words until every byte carries its 0x80,
// writes the whole array to `dst` (device-mapped pinned memory, or a device buffer) with a handful of system-scope
// (written-through) stores and clears the words for the next launch.
__device__ __forceinline__ void feasible_collect(uint32_t* words, uint32_t* dst, uint32_t n_apps, int lane) {
    const uint32_t n_words = (n_apps + 3u) / 4u;
    for (uint32_t base = 0; base < n_words; base += kWave) {  // in order: the early words are usually complete first
        const uint32_t i = base + (uint32_t)lane;
        const uint32_t left = i < n_words ? n_apps - 4u * i : 0u;  // applications this word speaks for (>= 4: all four bytes)
        const uint32_t want = left >= 4u ? 0x80808080u : (left == 0u ? 0u : (0x80808080u >> (8u * (4u - left))));
        uint32_t v = 0;
        do {
            if (i < n_words) v = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (__ballot((v & want) != want) != 0ull);
        if (i < n_words) {
            __hip_atomic_store(dst + i, v & 0x01010101u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(words + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// FEAS (gf_fit_feasible): feasibility only.  `results` is then an array of n_apps BYTES (device-mapped pinned host memory, or a
// device buffer) that receives HasCapacity and nothing else, and `stats` carries the call's collection words instead of
// counters (feasible_announce / feasible_collect above): the caller watches the bytes arrive in pinned memory instead of
// waiting for the kernel-end write-back and the stream's completion signal.  Measured on the way
// (profiles/r5g_feasible_call_byte_stores.txt, r5h_feasible_call_last_wavefront.txt): one system-scope byte store per wavefront
// straight into pinned memory — 1 000 lone bytes over the host link cost 10 us more than they save —; an arrival counter whose
// last wavefront copies out — the counter's return value arrives behind the wavefront's own placement stores (memory
// operations return in order, a store is acknowledged after ~2 us): 37 us per call.  The placements are still made — same
// decision code — and stay in device memory.  A separate instantiation: the batch kernel proper carries none of this.
template <int ALGO, bool FEAS>
__global__ __launch_bounds__(kWave* kWavesPerBlock) GF_IND_OCC void fit_independent_kernel(
    NodeTable T, SparseTable G, uint32_t n_apps, const gf_app* __restrict__ apps, gf_result* __restrict__ results,
    uint32_t* __restrict__ exec_nodes, uint32_t* __restrict__ scratch, uint64_t scratch_half,
    ScanStats* __restrict__ stats) {
    const int lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool kMfTeam = ALGO == GF_ALGO_MINIMAL_FRAGMENTATION && GF_MF_TEAM != 0;
    const uint32_t a = kMfTeam ? blockIdx.x : blockIdx.x * kWavesPerBlock + wave;
    const bool reports = !kMfTeam || wave == 0u;  // (a team's wavefronts reach the same decision: one of them reports it)
    const uint32_t n_waves = n_apps;
    if (FEAS && blockIdx.x == gridDim.x - 1u) {  // the collecting workgroup (launch_fit_independent appends it)
        if (wave == 0) feasible_collect(reinterpret_cast<uint32_t*>(stats), reinterpret_cast<uint32_t*>(results), n_apps, lane);
        return;
    }
    if (a >= n_waves) return;
    GlobalView V{T.cpu, T.mem, T.gpu, T.cmax, T.cmax + T.n_chunks, T.cmax + 2 * (size_t)T.n_chunks, T.xmask, T.dmask,
                 T.n_chunks};
    Orders O{T.slot_node, T.dslot, T.n_x, T.n_d, T.d_identity != 0};
    if constexpr (ALGO == GF_ALGO_MINIMAL_FRAGMENTATION) {  // the capacity histogram rows of this wavefront
        __shared__ __attribute__((aligned(16))) uint32_t mf_hist[kWavesPerBlock * 3 * kMfHistBins];
        O.lend_minfrag((lds_u32h*)mf_hist + (size_t)wave * 3 * kMfHistBins, T);
        if constexpr (kMfTeam) {  // the workgroup is a team on ONE application (launch_fit_independent: a workgroup per application)
            __shared__ uint32_t mf_words[kMfTeamMax];
            O.mf_team = (uint32_t)kWavesPerBlock;
            O.mf_rank = wave;
            O.mf_team_base = (lds_u32h*)mf_hist;
            O.mf_words = (lds_u32h*)mf_words;
        }
#ifdef GF_MF_PROBE
        if (stats != nullptr) O.mf_probe = &stats->fifo_phase_cycles[0];
#endif
    }
#ifdef GF_MF_PROBE
    const unsigned long long t_kernel0 = __builtin_readcyclecounter();
#endif
    // every launch starts with cold L2s: the chunk index of group 0 and the app record are requested together
    //      (unconditionally — a branch here would make the compiler wait for the loads at the join; the general layout
    //      ignores the values, the buffers exist in both layouts)
    const Group0 g0 = load_group0(V, O, lane);
    const bool merged = T.d_identity != 0 && ALGO != GF_ALGO_MINIMAL_FRAGMENTATION;
    unsigned long long xvis = 0, dvis = 0;
    auto decide = [&](const App& app, uint32_t ai) {
        Decision dec = wave_decide<ALGO, GlobalView, false>(V, O, app, exec_nodes + app.exec_off, scratch + app.exec_off,
                                                            scratch + scratch_half + app.exec_off, lane, xvis, dvis,
                                                            merged ? &g0 : nullptr, &G);
        if (FEAS) {
            if (lane == 0 && reports) feasible_announce(reinterpret_cast<uint32_t*>(stats), ai, dec.feasible);
        } else if (lane == 0 && reports) {
            gf_result r;
            r.has_capacity = dec.feasible ? 1 : 0;
            if (dec.feasible && dec.ds_node == GF_NO_NODE) dec.ds_node = T.slot_node[dec.ds];
            r.driver_node = dec.feasible ? dec.ds_node : GF_NO_NODE;
            r.exec_len = dec.feasible ? (uint32_t)app.k : 0u;
            r.evaluated = 1;
            results[ai] = r;
        }
    };
    decide(load_app(apps, a), a);
#ifdef GF_MF_PROBE
    if constexpr (ALGO == GF_ALGO_MINIMAL_FRAGMENTATION)
        if (!FEAS && stats != nullptr && lane == 0 && reports) atomicAdd(&stats->fifo_phase_cycles[0], __builtin_readcyclecounter() - t_kernel0);
#endif
    if (!FEAS && stats != nullptr && lane == 0 && reports) {
        atomicAdd(&stats->exec_slots_visited, xvis);
        atomicAdd(&stats->driver_slots_visited, dvis);
    }
}

// ------------------------------------------------------------------------------------------------ FIFO chain

// Workgroup-level exchange of one 32-bit value per wave through LDS.  Two alternating buffers: one barrier per
// exchange is enough (a wave can only be one exchange ahead of the slowest wave).  16 entries regardless of NW (entries
// >= NW hold the identity) so that lanes 0..15 of every wave can combine them with 4 DPP steps instead of a serial loop.
struct Exchange {
    uint32_t first[2][16];
    int32_t tot[2][16];
};
constexpr uint32_t kNoPos = 0xFFFFFFFFu;

// Minimum over the waves of a wave-uniform position (kNoPos = none).
template <int NW>
__device__ __forceinline__ uint32_t block_min(Exchange* X, int& xb, uint32_t wave_value, int wave, int lane) {
    if (NW == 1) return wave_value;
    if (lane == 0) X->first[xb][wave] = wave_value;
    lds_barrier();
    int32_t x = (int32_t)X->first[xb][lane & 15];
#define GF_MIN_STEP(n)                                                                                  \
    {                                                                                                   \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(-1, x, GF_DPP_ROW_SHR(n), 0xf, 0xf, false); \
        x = (int32_t)(t < (uint32_t)x ? t : (uint32_t)x);                                               \
    }
    GF_MIN_STEP(1)
    GF_MIN_STEP(2)
    GF_MIN_STEP(4)
    GF_MIN_STEP(8)
#undef GF_MIN_STEP
    xb ^= 1;
    return (uint32_t)read_lane(x, 15);
}

// Exclusive prefix over the waves (in wave order) and total of a wave-uniform count.
template <int NW>
__device__ __forceinline__ void block_scan(Exchange* X, int& xb, int32_t wave_total, int wave, int lane,
                                           int32_t& prefix, int32_t& total) {
    if (NW == 1) {
        prefix = 0;
        total = wave_total;
        return;
    }
    if (lane == 0) X->tot[xb][wave] = wave_total;
    lds_barrier();
    int32_t x = X->tot[xb][lane & 15];
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_SHR(1), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_SHR(2), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_SHR(4), 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, GF_DPP_ROW_SHR(8), 0xf, 0xf, false);
    xb ^= 1;
    prefix = read_lane(x, wave) - wave_total;
    total = read_lane(x, 15);
}

constexpr int kAppStage = 32;  // app records staged into LDS per refill
constexpr int kRefresh = 8;    // apps between refreshes of the dirty entries of the chunk-maxima index

struct FifoShared {
    App apps[kAppStage];
    int32_t slow_feasible;
    uint32_t slow_ds;
    int64_t slow_pass1;
};

// State of one executor scan of the FIFO fast path (all fields workgroup-uniform except `hit`).
struct FifoScan {
    int32_t taken;  // placements so far (tightly-pack: sum of clamped capacities); < K + 2^30
    uint32_t end;   // one past the last slot looked at
    uint64_t hit;   // per thread: bit s set when this thread's slot in step s (slots [s*BLOCK, (s+1)*BLOCK)) hosts an executor
};

// One step of the executor scan: BLOCK consecutive slots starting at b = step * BLOCK, one 64-slot chunk per wave.
// `mine` (wave-uniform): the maxima index could not rule this wave's chunk out; ruled-out chunks contribute 0 unread.
// LDS_ONLY is a compile-time promise that every valid slot of the step lives in LDS: that instantiation contains no
// global LOAD, hence no vmcnt wait — placement stores still in flight (they count in vmcnt on gfx9-family ISAs)
// never stall the chain.
template <int ALGO, int NW, bool LDS_ONLY>
__device__ __forceinline__ void fifo_scan_step(const HybridView& V, const Orders& O, const App& app, uint32_t ds,
                                               uint32_t step, bool mine, uint32_t tid, int wave, int lane,
                                               Exchange* X, int& xb, uint32_t* __restrict__ out,
                                               uint32_t* __restrict__ surv, FifoScan& st) {
    constexpr uint32_t BLOCK = kWave * NW;
    const int32_t K = app.k;
    const uint32_t b = step * BLOCK;
    const uint32_t j = b + tid;
    int64_t a0 = -1, a1 = -1, a2 = -1;
    const bool cand = mine && j < O.n_x && V.xcand(j);
    if (cand) {
        if (LDS_ONLY)
            V.load_lds(j, a0, a1, a2);
        else
            V.load(j, a0, a1, a2);
        if (j == ds) {
            a0 -= app.drv0;
            a1 -= app.drv1;
            a2 -= app.drv2;
        }
    }
    st.end = b + BLOCK;
    const bool ge1 = cand && cap_ge1(a0, a1, a2, app);
    if (ALGO == GF_ALGO_TIGHTLY_PACK) {
        int32_t c = 0;
        if (ge1) c = cap3(a0, a1, a2, app);  // no division for slots (or whole waves) that hold nothing
        const int32_t incl = wave_inclusive_scan(c);
        int32_t prefix, total;
        block_scan<NW>(X, xb, read_lane(incl, kWave - 1), wave, lane, prefix, total);
        if (total > 0) {
            const int32_t start = st.taken + prefix + (incl - c);
            const int32_t room = K - start;
            const int32_t t = room <= 0 ? 0 : (room < c ? room : c);
            if (t > 0 && step < 64) st.hit |= 1ull << step;
            emit_runs(out, start, t, j, lane);
        }
        st.taken += total;
    } else {
        const uint64_t m = __ballot(ge1);
        int32_t prefix, total;
        block_scan<NW>(X, xb, (int32_t)__popcll((unsigned long long)m), wave, lane, prefix, total);
        const int32_t pos = st.taken + prefix + (int32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1ull)));
        if (ge1 && pos < K) {
            out[pos] = j;
            surv[pos] = j;  // survivor list for the (rare) multi-pass path
            if (step < 64) st.hit |= 1ull << step;
        }
        st.taken += total;
    }
}

// One step of the driver-candidate scan: returns the first fitting position of this step or kNoPos.
template <int NW, bool LDS_ONLY, bool DIDENT>
__device__ __forceinline__ uint32_t fifo_driver_step(const HybridView& V, const Orders& O, const App& app,
                                                    uint32_t step, bool mine, uint32_t tid, int wave, int lane,
                                                    Exchange* X, int& xb) {
    constexpr uint32_t BLOCK = kWave * NW;
    const uint32_t i = step * BLOCK + tid;
    bool fit = false;
    if (mine && i < O.n_d && (!DIDENT || V.dcand(i))) {
        int64_t a0, a1, a2;
        if (LDS_ONLY)
            V.load_lds(i, a0, a1, a2);  // identity mapping: position == slot
        else
            V.load(DIDENT ? i : O.driver_slot(i), a0, a1, a2);
        fit = driver_fits(a0, a1, a2, app);
    }
    const uint64_t m = __ballot(fit);
    const uint32_t wfirst = m ? (i - (uint32_t)lane) + (uint32_t)(__ffsll((unsigned long long)m) - 1) : kNoPos;
    return block_min<NW>(X, xb, wfirst, wave, lane);
}

// FIFO replay (internal/extender/resource.go:224-262 + :321): apps strictly in order, each against the residuals its
// predecessors left.  ONE workgroup of NW waves: nodes in parallel (NW*64 per step), apps sequential.  The working
// table's front and the chunk-maxima index live in LDS, so the per-app critical path is LDS latency + a few workgroup
// barriers instead of global-memory round trips.  Placements are written as SLOT ids (expand_translate_kernel maps them).
// DIDENT: the driver order is a prefix of the executor order (position == slot), the production shape; that
// instantiation has no driver-slot gather, i.e. no global load anywhere on the per-app fast path.
template <int ALGO, int NW, bool DIDENT>
__global__ __launch_bounds__(kWave* NW) void fit_fifo_chain_kernel(NodeTable T, uint32_t lds_slots, uint32_t n_apps,
                                                                  const gf_app* __restrict__ apps,
                                                                  gf_result* __restrict__ results,
                                                                  uint32_t* __restrict__ exec_nodes,
                                                                  uint32_t* __restrict__ scratch,
                                                                  uint64_t scratch_half,
                                                                  int32_t* __restrict__ chain_failed_at,
                                                                  ScanStats* __restrict__ stats,
                                                                  const int32_t* __restrict__ run_if_set) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t BLOCK = kWave * NW;
    constexpr uint32_t STEPS_PER_GROUP = kWave / NW;  // one 64-bit chunk mask covers this many steps
    if (run_if_set != nullptr && *run_if_set == 0) return;  // the narrow kernel served this launch
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long t0_cycles = __builtin_readcyclecounter();
    const unsigned long long t0_real = wall_clock64();

    // ---- LDS carve: table front | chunk maxima | exchange | shared scalars + staged apps
    lds_i64* lcpu = (lds_i64*)smem;
    lds_i64* lmem = lcpu + lds_slots;
    lds_i64* lgpu = lmem + lds_slots;
    lds_i64* lmax = lgpu + lds_slots;  // [3][n_chunks]
    lds_u64* lxm = (lds_u64*)(lmax + 3 * (size_t)T.n_chunks);  // [2][n_chunks] candidate masks
    Exchange* X = reinterpret_cast<Exchange*>(smem + 24 * ((size_t)lds_slots + T.n_chunks) + 16 * (size_t)T.n_chunks);
    FifoShared* sh = reinterpret_cast<FifoShared*>(X + 1);
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    lds_u8* dirty = (lds_u8*)(sh + 1);  // [n_chunks] chunk touched by a commit since the last maxima refresh
    for (uint32_t s = tid; s < lds_slots; s += BLOCK) {
        lcpu[s] = T.cpu[s];
        lmem[s] = T.mem[s];
        lgpu[s] = T.gpu[s];
    }
    for (uint32_t c = tid; c < 3 * T.n_chunks; c += BLOCK) lmax[c] = T.cmax[c];
    for (uint32_t c = tid; c < T.n_chunks; c += BLOCK) {
        dirty[c] = 0;
        lxm[c] = T.xmask[c];
        lxm[T.n_chunks + c] = T.dmask[c];
    }
    if (tid < 32) {  // exchange entries of waves that do not exist must hold the identity
        X->first[tid >> 4][tid & 15] = kNoPos;
        X->tot[tid >> 4][tid & 15] = 0;
    }
    HybridView V{lcpu, lmem, lgpu, lds_slots, (glb_i64*)T.cpu, (glb_i64*)T.mem, (glb_i64*)T.gpu,
                 lmax, lmax + T.n_chunks, lmax + 2 * (size_t)T.n_chunks, lxm, lxm + T.n_chunks, T.n_chunks};
    Orders O{T.slot_node, T.dslot, T.n_x, T.n_d, DIDENT};
    const uint32_t xc = (O.n_x + kWave - 1) / kWave;  // chunks of the executor order
    const uint32_t dc = (O.n_d + kWave - 1) / kWave;  // chunks of the driver order (identity mapping only)
    const uint32_t x_steps = (O.n_x + BLOCK - 1) / BLOCK;
    const uint32_t d_steps = (O.n_d + BLOCK - 1) / BLOCK;
    const bool mask_commit = x_steps <= 64u;  // a 64-bit per-thread hit mask covers a whole scan
    // a step is LDS-only when (step + 1) * BLOCK <= *_lds_limit (whole order resident -> no limit)
    const uint32_t x_lds_steps = O.n_x <= lds_slots ? 0xFFFFFFFFu : lds_slots / BLOCK;
    const uint32_t d_lds_steps = !DIDENT ? 0u : (O.n_d <= lds_slots ? 0xFFFFFFFFu : lds_slots / BLOCK);

    unsigned long long xvis = 0, dvis = 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};  // per-phase cycle accounting (only when stats are requested)
    const bool prof = stats != nullptr;
    unsigned long long tp = prof ? __builtin_readcyclecounter() : 0;
#define GF_PHASE(idx)                                             \
    if (prof) {                                                   \
        const unsigned long long tn = __builtin_readcyclecounter(); \
        ph[idx] += tn - tp;                                       \
        tp = tn;                                                  \
    }
    int xb = 0;
    int32_t failed_at = -1;
    uint32_t a = 0;
    for (; a < n_apps; ++a) {
        // ---- keep the maxima index tight: recompute the chunks that commits touched since the last refresh (the
        //      entries are upper bounds, so a stale one is only a missed skip, never a wrong answer)
        if ((a % kRefresh) == 0 && a != 0) {
            for (uint32_t c = (uint32_t)wave; c < T.n_chunks; c += NW) {
                if (!dirty[c]) continue;  // wave-uniform
                const uint32_t s = c * kWave + lane;
                int64_t a0 = INT64_MIN, a1 = INT64_MIN, a2 = INT64_MIN;
                if (s < T.n_slots) V.load(s, a0, a1, a2);
                a0 = wave_max_i64(a0);
                a1 = wave_max_i64(a1);
                a2 = wave_max_i64(a2);
                if (lane == 0) {
                    lmax[c] = a0;
                    lmax[T.n_chunks + c] = a1;
                    lmax[2 * (size_t)T.n_chunks + c] = a2;
                    dirty[c] = 0;
                }
            }
            __syncthreads();
        }
        // ---- stage the next kAppStage app records (incl. reciprocals) into LDS
        if ((a % kAppStage) == 0) {
            lds_barrier();  // previous stage fully consumed; also publishes the table / index fill
            const uint32_t n_stage = (n_apps - a) < (uint32_t)kAppStage ? (n_apps - a) : (uint32_t)kAppStage;
            if (tid < n_stage) sh->apps[tid] = load_app(apps, a + tid);
            lds_barrier();
        }
        const App app = sh->apps[a % kAppStage];
        const int32_t K = app.k;
        uint32_t* out = exec_nodes + app.exec_off;
        uint32_t* surv = scratch + app.exec_off;
        const bool last = (a + 1 == n_apps);
        GF_PHASE(0)

        // ---- (1) first fitting driver candidate, BLOCK candidates per step; steps whose chunks the maxima index
        //          rules out are skipped without a barrier (every wave evaluates the same 64-chunk mask)
        int64_t p0 = -1;
        {
            uint32_t f = kNoPos;
            for (uint32_t g = 0; g * STEPS_PER_GROUP < d_steps && f == kNoPos; ++g) {
                uint64_t gm = ~0ull;
                if (DIDENT) {
                    gm = chunk_group_mask<true>(V, g, dc, app.drv0, app.drv1, app.drv2, lane);
                    dvis += kWave;
                }
                for (uint32_t s = 0; s < STEPS_PER_GROUP; ++s) {
                    const uint32_t step = g * STEPS_PER_GROUP + s;
                    if (step >= d_steps) break;
                    const uint64_t bits = NW == 64 ? gm : ((gm >> (s * NW)) & ((1ull << NW) - 1ull));
                    if (bits == 0) continue;
                    const bool mine = (bits >> wave) & 1ull;
                    if (step < d_lds_steps)
                        f = fifo_driver_step<NW, true, DIDENT>(V, O, app, step, mine, tid, wave, lane, X, xb);
                    else
                        f = fifo_driver_step<NW, false, DIDENT>(V, O, app, step, mine, tid, wave, lane, X, xb);
                    dvis += chunk_len(O.n_d, step * BLOCK, BLOCK);
                    if (f != kNoPos) break;
                }
            }
            if (f != kNoPos) p0 = (int64_t)f;
        }
        GF_PHASE(1)

        Decision dec;
        dec.feasible = false;
        dec.ds = 0;
        dec.pass1 = 0;
        enum { kCommitDone, kCommitMask, kCommitList } commit = kCommitDone;
        FifoScan st;
        st.taken = 0;
        st.end = 0;
        st.hit = 0;
        if (p0 >= 0) {
            const uint32_t ds = DIDENT ? (uint32_t)p0 : O.driver_slot((uint32_t)p0);
            dec.ds = ds;
            if (K == 0) {
                dec.feasible = true;
                commit = kCommitMask;  // nothing placed: only the driver request is subtracted
            } else {
                // ---- (2) executors, BLOCK slots per step, lazy stop, ruled-out steps skipped
                for (uint32_t g = 0; g * STEPS_PER_GROUP < x_steps && st.taken < K; ++g) {
                    const uint64_t gm = chunk_group_mask<false>(V, g, xc, app.exe0, app.exe1, app.exe2, lane);
                    xvis += kWave;
                    for (uint32_t s = 0; s < STEPS_PER_GROUP; ++s) {
                        const uint32_t step = g * STEPS_PER_GROUP + s;
                        if (step >= x_steps) break;
                        const uint64_t bits = NW == 64 ? gm : ((gm >> (s * NW)) & ((1ull << NW) - 1ull));
                        if (bits == 0) continue;
                        const bool mine = (bits >> wave) & 1ull;
                        if (step < x_lds_steps)
                            fifo_scan_step<ALGO, NW, true>(V, O, app, ds, step, mine, tid, wave, lane, X, xb, out, surv,
                                                           st);
                        else
                            fifo_scan_step<ALGO, NW, false>(V, O, app, ds, step, mine, tid, wave, lane, X, xb, out,
                                                            surv, st);
                        xvis += (uint64_t)__popcll((unsigned long long)bits) * kWave;
                        if (st.taken >= K) break;
                    }
                }
                GF_PHASE(2)
                if (st.taken >= K) {
                    dec.feasible = true;
                    dec.pass1 = st.taken;
                    commit = mask_commit ? kCommitMask : kCommitList;
                } else {
                    // ---- (3) slow path, wave 0 alone: multi-pass distribute-evenly and/or driver fallback.
                    // st.taken is the exact capacity total S_d for tightly-pack, the pass-1 survivor count for evenly.
                    __syncthreads();  // out[] / surv[] writes of all waves are complete
                    if (wave == 0) {
                        uint32_t* sb = scratch + scratch_half + app.exec_off;
                        Decision d2;
                        d2.feasible = false;
                        d2.ds = ds;
                        d2.pass1 = st.taken;
                        int64_t S_d = st.taken;
                        if (ALGO != GF_ALGO_TIGHTLY_PACK)
                            S_d = wave_even_general<HybridView, true>(V, O, app, ds, out, surv, sb, st.taken, lane);
                        if (S_d >= K)
                            d2.feasible = true;
                        else
                            d2 = wave_fallback<ALGO, HybridView, true>(V, O, app, p0, ds, S_d, out, surv, sb, lane,
                                                                      xvis, dvis);
                        if (d2.feasible && !last) wave_commit_from_list<ALGO, HybridView>(V, app, d2, out, lane);
                        if (lane == 0) {
                            sh->slow_feasible = d2.feasible ? 1 : 0;
                            sh->slow_ds = d2.ds;
                            sh->slow_pass1 = d2.pass1;
                        }
                    }
                    __syncthreads();
                    dec.feasible = sh->slow_feasible != 0;
                    dec.ds = sh->slow_ds;
                    dec.pass1 = sh->slow_pass1;
                    commit = kCommitDone;
                    GF_PHASE(3)
                }
            }
        }

        if (tid == 0) {
            gf_result r;
            r.has_capacity = dec.feasible ? 1 : 0;
            r.driver_node = dec.feasible ? dec.ds : GF_NO_NODE;  // SLOT id; expand_translate_kernel maps it to the node index
            r.exec_len = dec.feasible ? (uint32_t)app.k : 0u;
            r.evaluated = 1;
            results[a] = r;
        }
        if (last) {
            ++a;
            break;  // the driver being filtered: nothing is subtracted after it (resource.go:321-328)
        }
        if (!dec.feasible) {
            if (app.flags & GF_APP_SKIPPABLE) continue;  // resource.go:244-248
            failed_at = (int32_t)a;                       // resource.go:249-251
            ++a;
            break;
        }
        // ---- (4) commit the usage of this earlier driver (sparkpods.go:139-146: one executor request per DISTINCT
        //          executor node; the driver request only if no executor landed on the driver's node)
        const bool lds_only = commit == kCommitMask && (st.end <= lds_slots || O.n_x <= lds_slots) &&
                              dec.ds < lds_slots;  // uniform: the commit touches LDS only
        if (commit == kCommitMask) {
            bool hosts = false;
            uint64_t h = st.hit;
            // thread (ds % BLOCK) is the one that looks at slot ds when its step runs; a slot whose step was skipped,
            // lies beyond the cut, or is driver-only (>= n_x) was never hit
            const uint32_t owner = dec.ds % BLOCK;
            if (lds_only) {
                while (h) {
                    const uint32_t step = (uint32_t)__ffsll((unsigned long long)h) - 1;
                    h &= h - 1;
                    const uint32_t s = step * BLOCK + tid;
                    V.sub_lds(s, app.exe0, app.exe1, app.exe2);
                    dirty[s / kWave] = 1;
                    if (s == dec.ds) hosts = true;
                }
                if (tid == owner && !hosts) {
                    V.sub_lds(dec.ds, app.drv0, app.drv1, app.drv2);
                    dirty[dec.ds / kWave] = 1;
                }
            } else {
                while (h) {
                    const uint32_t step = (uint32_t)__ffsll((unsigned long long)h) - 1;
                    h &= h - 1;
                    const uint32_t s = step * BLOCK + tid;
                    V.sub(s, app.exe0, app.exe1, app.exe2);
                    dirty[s / kWave] = 1;
                    if (s == dec.ds) hosts = true;
                }
                if (tid == owner && !hosts) {
                    V.sub(dec.ds, app.drv0, app.drv1, app.drv2);
                    dirty[dec.ds / kWave] = 1;
                }
            }
        } else if (commit == kCommitList) {
            __syncthreads();  // every wave's placements are written
            if (wave == 0) wave_commit_from_list<ALGO, HybridView>(V, app, dec, out, lane);
        }
        // residuals visible to every wave before the next app scans: LDS-only unless the global tail was touched
        if (lds_only)
            lds_barrier();
        else
            __syncthreads();
        GF_PHASE(4)
    }
#undef GF_PHASE
    // apps behind an abort are reported as not evaluated
    for (uint32_t r = a + tid; r < n_apps; r += BLOCK) {
        gf_result z;
        z.has_capacity = 0;
        z.driver_node = GF_NO_NODE;
        z.exec_len = 0;
        z.evaluated = 0;
        results[r] = z;
    }
    __syncthreads();
    // write the LDS-resident front of the working table back (gf_residual_get reads the global copy)
    for (uint32_t s = tid; s < lds_slots; s += BLOCK) {
        T.cpu[s] = lcpu[s];
        T.mem[s] = lmem[s];
        T.gpu[s] = lgpu[s];
    }
    if (tid == 0) {
        if (chain_failed_at != nullptr) *chain_failed_at = failed_at;
        if (stats != nullptr) {
            atomicAdd(&stats->exec_slots_visited, xvis);
            atomicAdd(&stats->driver_slots_visited, dvis);
            stats->fifo_shader_cycles = __builtin_readcyclecounter() - t0_cycles;
            stats->fifo_realtime_ticks = wall_clock64() - t0_real;
            for (int i = 0; i < 6; ++i) stats->fifo_phase_cycles[i] = ph[i];
        }
    }
}

#include "gangfit_worker.inc"
#include "gangfit_fifo_common.inc"
#include "gangfit_fifo_solo.inc"
#include "gangfit_zones.inc"
#include "gangfit_fifo_zoned.inc"
#include "gangfit_fifo_minfrag.inc"
#include "gangfit_shard.inc"
#include "gangfit_executor.inc"
#include "gangfit_findnodes.inc"

// ------------------------------------------------------------------------------------------------ self-test

__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Each lane checks (a) the DPP scan against a serial sum through LDS and (b) cap_dim against a plain 64-bit divide on
// adversarial operands (values adjacent to exact multiples, huge/small divisors, clamp boundaries).
__global__ __launch_bounds__(kWave) void selftest_kernel(uint64_t seed, uint32_t n_cases, uint32_t* mismatch) {
    __shared__ int32_t vals[kWave];
    const int lane = lane_id();
    uint64_t s = seed + 0x1234567ull * (blockIdx.x * (uint64_t)kWave + lane + 1);
    uint32_t bad = 0;
    for (uint32_t it = 0; it < n_cases; ++it) {
        // (a) scan
        const int32_t v = (int32_t)(splitmix64(s) & 0xFFFFF);
        vals[lane] = v;
        __syncthreads();
        int32_t ref = 0;
        for (int i = 0; i <= lane; ++i) ref += vals[i];
        __syncthreads();
        if (wave_inclusive_scan(v) != ref) ++bad;
        // (b) division
        const uint64_t r0 = splitmix64(s), r1 = splitmix64(s), r2 = splitmix64(s);
        const int ebits = 1 + (int)(r0 % 61);  // divisor magnitude 2^1 .. 2^61
        int64_t e = (int64_t)(r1 >> (64 - ebits));
        if (e == 0) e = 1;
        const int32_t k = (int32_t)(1 + (r2 % (uint64_t)GF_MAX_K));
        const int64_t lim = (int64_t)((1ull << 62) - 1);
        int64_t a;
        switch ((r0 >> 8) % 4) {
        case 0: a = (int64_t)(splitmix64(s) >> 2); break;  // uniform in [0, 2^62)
        case 1: {                                          // just around a multiple of e
            const int64_t mult = (int64_t)(splitmix64(s) % (uint64_t)(2 * (int64_t)k + 3));
            const int64_t maxq = lim / e;
            const int64_t qq = mult < maxq ? mult : maxq;
            a = qq * e + (int64_t)(splitmix64(s) % 3) - 1;
            break;
        }
        case 2: a = (int64_t)(splitmix64(s) % (uint64_t)(e)); break;  // below the divisor
        default: a = -(int64_t)(splitmix64(s) >> 3); break;           // negative availability
        }
        if (a > lim) a = lim;
        const double rcp = (it & 1u) ? fast_rcp((double)e) : 1.0 / (double)e;  // both reciprocals cap_dim is used with
        int32_t want;
        if (a < 0)
            want = 0;
        else {
            const int64_t q = a / e;
            want = q < (int64_t)k ? (int32_t)q : k;
        }
        if (cap_dim(a, e, rcp, k) != want) ++bad;
        if (cap_dim(a, 0, 0.0, k) != (a < 0 ? 0 : k)) ++bad;
        // (b2) the unclamped quotient of minimal-fragmentation (cap_dim_full): the same operands — quotients on both sides of
        //      its 2^40 switch to the plain division (a uniform in [0, 2^62) over a small divisor) —, both reciprocals
        if (cap_dim_full(a, e, rcp) != (a < 0 ? 0 : a / e)) ++bad;
        {
            const int64_t e2 = 1 + (int64_t)(r2 % 4093);  // small divisors: quotients around and above 2^40
            const int64_t a2 = (int64_t)((splitmix64(s) >> 2) >> (r1 % 24));
            const double rcp2 = (it & 1u) ? fast_rcp((double)e2) : 1.0 / (double)e2;
            if (cap_dim_full(a2, e2, rcp2) != a2 / e2) ++bad;
            const int64_t a3 = ((int64_t)1 << 40) * e2 + (int64_t)(splitmix64(s) % 5) - 2;  // right at the switch
            if (cap_dim_full(a3, e2, rcp2) != a3 / e2) ++bad;
        }
        if (cap_dim_full(a, 0, 0.0) != (a < 0 ? 0 : kCapInf)) ++bad;
        // (c) the narrow domain's division by multiplication (narrow_magic): |a| < 2^30, 0 < e < 2^30, the same operand families
        {
            const int nbits = 1 + (int)((r0 >> 16) % 30);
            int32_t ne = (int32_t)((r1 >> 3) & ((1u << nbits) - 1u));
            if (ne == 0) ne = 1;
            if ((r0 >> 24) % 7 == 0) ne = 1 << (nbits - 1);  // powers of two: m = 2^30 exactly
            const int32_t nlim = (1 << 30) - 1;
            int32_t na;
            switch ((r0 >> 32) % 4) {
            case 0: na = (int32_t)(splitmix64(s) & (uint64_t)nlim); break;
            case 1: {
                const int32_t maxq = nlim / ne;
                const int32_t mult = (int32_t)(splitmix64(s) % (uint64_t)(2 * (int64_t)k + 3));
                const int64_t t = (int64_t)(mult < maxq ? mult : maxq) * ne + (int64_t)(splitmix64(s) % 3) - 1;
                na = t > nlim ? nlim : (int32_t)t;
                break;
            }
            case 2: na = nlim - (int32_t)(splitmix64(s) % 3); break;  // the largest dividends
            default: na = -(int32_t)(splitmix64(s) & (uint64_t)nlim); break;
            }
            uint32_t mag, sh;
            narrow_magic(ne, mag, sh);
            const int32_t nwant = na < 0 ? 0 : ((na / ne) < k ? (na / ne) : k);
            if (ncap_dim(na, ne, mag, sh, k) != nwant) ++bad;
            if (ncap_dim(na, 0, 0u, 0u, k) != (na < 0 ? 0 : k)) ++bad;
            if (na >= 0 && ncap_full_dim(na, ne, mag, sh) != na / ne) ++bad;
            if (na >= ne) {  // a slot that fits: the variant the scans use
                NAppR q{};
                q.k = k;
                q.mag0 = mag;
                q.mag1 = 0;
                q.mag2 = mag;
                if (ncap3_fit(na, 5, na, q, sh, 0u, sh, 0u, 0xFFFFFFFFu, 0u) != nwant) ++bad;
            }
        }
    }
    if (bad) atomicAdd(mismatch, bad);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ launchers

// Workgroups of the worker kernel one CU holds at a time (what the runtime reports; the caller keeps a margin).
namespace {
// hipFuncAttributeMaxDynamicSharedMemorySize is per kernel and per device and only ever needs raising: the runtime call is
// made when a launch asks for more than any launch before it (one call less per chain otherwise).
hipError_t raise_dynamic_lds(const void* kernel, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> raised;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = raised[{kernel, dev}];
    if (lds <= have) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) have = lds;
    return e;
}

}  // namespace

hipError_t worker_blocks_per_cu(gf_algo algo, int* out) {
    const int threads = kWave * kWorkerWaves;
    if (algo == GF_ALGO_TIGHTLY_PACK)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fit_worker_kernel<GF_ALGO_TIGHTLY_PACK>, threads, 0);
    if (algo == GF_ALGO_DISTRIBUTE_EVENLY)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fit_worker_kernel<GF_ALGO_DISTRIBUTE_EVENLY>, threads, 0);
    if (algo == GF_ALGO_MINIMAL_FRAGMENTATION)
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, fit_worker_kernel<GF_ALGO_MINIMAL_FRAGMENTATION>, threads, 0);
    return hipErrorInvalidValue;
}

hipError_t launch_fit_worker(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, const WorkerArgs& args,
                             hipStream_t stream) {
    if (args.sets == 0 || args.blocks_per_set == 0 || args.host == nullptr || args.dev == nullptr) return hipErrorInvalidValue;
    const dim3 block(kWave * kWorkerWaves);
    const dim3 grid(1u + args.sets * args.blocks_per_set);
    if (algo == GF_ALGO_TIGHTLY_PACK)
        hipLaunchKernelGGL(fit_worker_kernel<GF_ALGO_TIGHTLY_PACK>, grid, block, 0, stream, table, gpu_view, args);
    else if (algo == GF_ALGO_DISTRIBUTE_EVENLY)
        hipLaunchKernelGGL(fit_worker_kernel<GF_ALGO_DISTRIBUTE_EVENLY>, grid, block, 0, stream, table, gpu_view, args);
    else if (algo == GF_ALGO_MINIMAL_FRAGMENTATION)
        hipLaunchKernelGGL(fit_worker_kernel<GF_ALGO_MINIMAL_FRAGMENTATION>, grid, block, 0, stream, table, gpu_view, args);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fit_independent(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, uint32_t n_apps,
                                  const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                                  uint64_t scratch_half, ScanStats* d_stats, hipStream_t stream, uint8_t* d_feasible,
                                  uint32_t* d_feasible_sync) {
    if (n_apps == 0) return hipSuccess;
    if (d_feasible != nullptr && d_feasible_sync == nullptr) return hipErrorInvalidValue;
    const dim3 block(kWave * kWavesPerBlock);
    // (minimal-fragmentation: a workgroup per application — its wavefronts share the passes over the executor order)
    const bool team = algo == GF_ALGO_MINIMAL_FRAGMENTATION && GF_MF_TEAM != 0;
    const dim3 grid(team ? n_apps : (n_apps + kWavesPerBlock - 1) / kWavesPerBlock);
    const dim3 grid_feas(grid.x + 1);  // + the collecting workgroup
#define GF_IND(ALGO)                                                                                                               \
    if (d_feasible != nullptr)                                                                                                     \
        hipLaunchKernelGGL((fit_independent_kernel<ALGO, true>), grid_feas, block, 0, stream, table, gpu_view, n_apps, d_apps,      \
                           reinterpret_cast<gf_result*>(d_feasible), d_exec_nodes, d_scratch, scratch_half,                        \
                           reinterpret_cast<ScanStats*>(d_feasible_sync));                                                         \
    else                                                                                                                           \
        hipLaunchKernelGGL((fit_independent_kernel<ALGO, false>), grid, block, 0, stream, table, gpu_view, n_apps, d_apps,          \
                           d_results, d_exec_nodes, d_scratch, scratch_half, d_stats)
    if (algo == GF_ALGO_TIGHTLY_PACK) {
        GF_IND(GF_ALGO_TIGHTLY_PACK);
    } else if (algo == GF_ALGO_MINIMAL_FRAGMENTATION) {
        GF_IND(GF_ALGO_MINIMAL_FRAGMENTATION);
    } else {
        GF_IND(GF_ALGO_DISTRIBUTE_EVENLY);
    }
#undef GF_IND
    return hipGetLastError();
}

size_t fifo_v2_lds_bytes(uint32_t lds_slots, uint32_t n_chunks) {
    return 24 * ((size_t)lds_slots + n_chunks) + 16 * (size_t)n_chunks + sizeof(Exchange) + sizeof(FifoShared) + 64 +
           ((n_chunks + 15) & ~15u);
}
size_t fifo_solo_lds_bytes(uint32_t lds_slots, uint32_t n_chunks) {
    const size_t nw = (n_chunks + 63u) / 64u, nwp = nw < 4 ? 4 : nw;
    size_t off = kFusedStage * sizeof(NApp) + sizeof(SoloShared) + 16 * (size_t)n_chunks + 8 * (size_t)kMaxShapes * nwp +
                 8 * (size_t)n_chunks + 16 * nw + 8 * (size_t)kMaxShapes;
    off = (off + 15) & ~(size_t)15;
    off += sizeof(ShapeEntry) * (kShapeHashSlots + kMaxShapes) + 4 * kShapeHashSlots + 16 + 4 * kMaxShapes +
           4 * (size_t)n_chunks + 12 * (size_t)n_chunks;
    off = (off + 15) & ~(size_t)15;
    return off + (size_t)(lds_slots / 64u) * (kBlkDwords * 4u);  // lds_slots is a multiple of 64
}

namespace {
template <class Kernel, class... Args>
hipError_t launch_one_workgroup(Kernel kernel, int n_waves, size_t lds, hipStream_t stream, Args... args) {
    const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(1), dim3(kWave * n_waves), lds, stream, args...);
    return hipGetLastError();
}

// The first kernel of a chain (ChainIo): records, scaled records, table copies, checkpoint overlay, run-head preset.
hipError_t launch_chain_prologue(const ChainIo& io, uint32_t n_apps, const gf_app* d_apps, NApp* d_napps, const int64_t* unit,
                                 int32_t* d_wide_needed, uint32_t* fill_dst, size_t fill_words, hipStream_t stream) {
    ChainPrologue p{};
    p.n_apps = (io.apps_src != nullptr || d_napps != nullptr) ? n_apps : 0u;
    p.apps_src = io.apps_src;
    p.apps_dst = const_cast<gf_app*>(d_apps);
    p.napps = d_napps;
    for (int j = 0; j < 3; ++j) p.unit[j] = unit ? unit[j] : 1;
    p.wide_needed = d_wide_needed;
    p.wide_clear = io.wide_clear;
    size_t most = p.n_apps;
    for (int r = 0; r < 2; ++r) {
        p.copy_src[r] = io.copy_src[r];
        p.copy_dst[r] = io.copy_dst[r];
        p.copy_words[r] = io.copy_words[r];
        most = std::max(most, io.copy_words[r] / 4);
    }
    p.overlay = io.overlay;
    p.overlay_stride = io.overlay_stride;
    p.overlay_count = io.overlay_count;
    p.overlay_dst = io.overlay_dst;
    p.overlay_slots = io.overlay_slots;
    p.overlay_chunks = io.overlay_chunks;
    if (io.overlay != nullptr) most = std::max(most, 3 * (size_t)io.overlay_slots / 4);
    p.fill_dst = fill_dst;
    p.fill_words = fill_words;
    most = std::max(most, fill_words);
    if (most == 0 && io.wide_clear == nullptr) return hipSuccess;
    size_t blocks = (most + 255) / 256;
    blocks = std::min<size_t>(std::max<size_t>(blocks, 1), 1024);
    hipLaunchKernelGGL(chain_prologue_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}
inline ChainOut chain_out_of(const ChainIo& io, const int32_t* d_failed) { return ChainOut{io.h_results, io.h_exec, d_failed, io.h_failed}; }

template <int ALGO>
hipError_t launch_v2(const FifoPlan& P, const NodeTable& T, uint32_t n_apps, const gf_app* d_apps, gf_result* d_results,
                     uint32_t* d_exec_nodes, uint32_t* d_scratch, uint64_t half, int32_t* d_failed, ScanStats* d_stats,
                     const int32_t* guard, hipStream_t stream) {
    const size_t lds = fifo_v2_lds_bytes(P.lds_slots_v2, T.n_chunks);
    constexpr int NW = 16;
    if (T.d_identity)
        return launch_one_workgroup(fit_fifo_chain_kernel<ALGO, NW, true>, NW, lds, stream, T, P.lds_slots_v2, n_apps, d_apps,
                                    d_results, d_exec_nodes, d_scratch, half, d_failed, d_stats, guard);
    return launch_one_workgroup(fit_fifo_chain_kernel<ALGO, NW, false>, NW, lds, stream, T, P.lds_slots_v2, n_apps, d_apps,
                                d_results, d_exec_nodes, d_scratch, half, d_failed, d_stats, guard);
}

template <int ALGO>
hipError_t launch_solo(const FifoPlan& P, const NodeTable& T, const NarrowTable& NT, uint32_t n_apps, NApp* d_napps,
                       const int32_t* d_wide_needed, gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                       uint64_t half, int32_t* d_failed, const ChainCkpt& ck, const SoloFused& F, ScanStats* d_stats,
                       hipStream_t stream) {
    const size_t lds = fifo_solo_lds_bytes(P.lds_slots_solo, T.n_chunks);
    constexpr int NW = GF_SOLO_WAVES;  // wavefront 0 walks the chain; all of them share the prologue, the checkpoints and the epilogue
    const bool resident = P.lds_slots_solo >= T.n_slots;
#define GF_SOLO(PR, RE)                                                                                                     \
    return launch_one_workgroup(fit_fifo_solo_kernel<ALGO, NW, PR, RE>, NW, lds, stream, T, NT, P.lds_slots_solo, n_apps,    \
                                d_napps, d_wide_needed, d_results, d_exec_nodes, d_scratch, half, d_failed, ck, F, d_stats)
    if (d_stats != nullptr) {
        if (resident) GF_SOLO(true, true);
        GF_SOLO(true, false);
    }
    if (resident) GF_SOLO(false, true);
    GF_SOLO(false, false);
#undef GF_SOLO
}

template <int ALGO>
hipError_t launch_fifo_algo(const FifoPlan& P, const NodeTable& T, const NarrowTable& NT, uint32_t n_apps,
                            const gf_app* d_apps, NApp* d_napps, int32_t* d_wide_needed, gf_result* d_results,
                            uint32_t* d_exec_nodes, uint32_t* d_scratch, uint64_t half, uint64_t heads_lo, int32_t* d_failed,
                            const ChainCkpt& ck, const ChainIo& io, ScanStats* d_stats, hipStream_t stream) {
    hipError_t e = hipSuccess;
    const int32_t* guard = nullptr;
    // run heads of the tightly-pack fast path: "no head here"
    const bool heads = P.narrow && ALGO == GF_ALGO_TIGHTLY_PACK && half > 1 + heads_lo;
    // The solo kernel is its own first kernel when the whole table lives in LDS, the host has proven every request's scaled
    // form (no wide twin) and nothing but a plain table copy was left to the prologue.
    SoloFused F{};
    F.enabled = (P.narrow && !P.wide && P.lds_slots_solo >= T.n_slots && io.overlay == nullptr && io.copy_words[1] == 0 &&
                 (io.copy_words[0] == 0 || io.copy_words[0] == 3 * (size_t)T.n_slots))
                    ? 1
                    : 0;
    if (F.enabled) {
        F.apps_src = io.apps_src;
        F.apps_dst = const_cast<gf_app*>(d_apps);
        F.table_src = io.copy_words[0] ? reinterpret_cast<const int32_t*>(io.copy_src[0]) : nullptr;
        for (int j = 0; j < 3; ++j) F.unit[j] = NT.unit[j];
        F.fill_dst = heads ? d_scratch + heads_lo : nullptr;
        F.fill_words = heads ? (size_t)(half - 1 - heads_lo) : 0;
        F.wide_clear = io.wide_clear;
    } else {
        e = launch_chain_prologue(io, n_apps, d_apps, P.narrow ? d_napps : nullptr, NT.unit, d_wide_needed,
                                  heads ? d_scratch + heads_lo : nullptr, heads ? (size_t)(half - 1 - heads_lo) : 0, stream);
        if (e != hipSuccess) return e;
    }
    if (P.narrow) {
        guard = d_wide_needed;
        e = launch_solo<ALGO>(P, T, NT, n_apps, d_napps, d_wide_needed, d_results, d_exec_nodes, d_scratch, half, d_failed, ck,
                              F, d_stats, stream);
        if (e != hipSuccess) return e;
    }
    if (P.wide) {
        e = launch_v2<ALGO>(P, T, n_apps, d_apps, d_results, d_exec_nodes, d_scratch, half, d_failed, d_stats, guard, stream);
        if (e != hipSuccess) return e;
    }
    const dim3 block(kWave * kWavesPerBlock);
    const dim3 grid((n_apps + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(expand_translate_kernel, grid, block, 0, stream, T.slot_node, n_apps, d_apps, d_results,
                       d_exec_nodes, d_scratch, chain_out_of(io, d_failed));
    return hipGetLastError();
}
}  // namespace

hipError_t launch_fit_fifo(gf_algo algo, const FifoPlan& plan, const NodeTable& table, const NarrowTable& ntable,
                           uint32_t n_apps, const gf_app* d_apps, NApp* d_napps, int32_t* d_wide_needed,
                           gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch, uint64_t scratch_half,
                           uint64_t heads_lo, int32_t* d_chain_failed_at, const ChainCkpt& ckpt, const ChainIo& io,
                           ScanStats* d_stats, hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    if (!plan.narrow && !plan.wide) return hipErrorInvalidValue;
    if (algo == GF_ALGO_TIGHTLY_PACK)
        return launch_fifo_algo<GF_ALGO_TIGHTLY_PACK>(plan, table, ntable, n_apps, d_apps, d_napps, d_wide_needed, d_results,
                                                      d_exec_nodes, d_scratch, scratch_half, heads_lo, d_chain_failed_at, ckpt,
                                                      io, d_stats, stream);
    return launch_fifo_algo<GF_ALGO_DISTRIBUTE_EVENLY>(plan, table, ntable, n_apps, d_apps, d_napps, d_wide_needed, d_results,
                                                       d_exec_nodes, d_scratch, scratch_half, heads_lo, d_chain_failed_at, ckpt,
                                                       io, d_stats, stream);
}

hipError_t launch_fit_zoned(int inner_algo, bool az_aware, bool reserve_execs, const NodeTable& table,
                            const ZoneTable& zones, const EffTables& eff, const ZoneBuffers& buf, uint32_t n_apps,
                            const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                            uint64_t scratch_half, hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    if (inner_algo != GF_ALGO_TIGHTLY_PACK && inner_algo != GF_ALGO_MINIMAL_FRAGMENTATION) return hipErrorInvalidValue;
    if (az_aware && inner_algo != GF_ALGO_TIGHTLY_PACK) return hipErrorInvalidValue;
    const dim3 block(kWave * kWavesPerBlock);
    const dim3 app_grid((n_apps + kWavesPerBlock - 1) / kWavesPerBlock);
    hipError_t e = hipSuccess;
    if (az_aware) {  // the plain TightlyPack answer first; the select kernel overwrites it where a zone wins
        e = launch_fit_independent(GF_ALGO_TIGHTLY_PACK, table, SparseTable{}, n_apps, d_apps, d_results, d_exec_nodes,
                                   d_scratch, scratch_half, nullptr, stream);
        if (e != hipSuccess) return e;
        if (buf.avg_out != nullptr) {
            e = launch_avg_efficiency(true, table, eff, buf.cnt, buf.n_cnt_waves, n_apps, d_apps, d_results,
                                      d_exec_nodes, buf.avg_out, stream);
            if (e != hipSuccess) return e;
        }
    }
    if (zones.n_zones > 0) {
        const uint64_t n_dec = (uint64_t)n_apps * zones.n_zones;
        const dim3 dec_grid((unsigned)((n_dec + kWavesPerBlock - 1) / kWavesPerBlock));
        if (inner_algo == GF_ALGO_MINIMAL_FRAGMENTATION)
            hipLaunchKernelGGL(fit_zoned_kernel<GF_ALGO_MINIMAL_FRAGMENTATION>, dec_grid, block, 0, stream, table, zones,
                               n_apps, d_apps, buf.zres, buf.zexec, buf.zexec_stride, d_scratch, scratch_half);
        else
            hipLaunchKernelGGL(fit_zoned_kernel<GF_ALGO_TIGHTLY_PACK>, dec_grid, block, 0, stream, table, zones, n_apps,
                               d_apps, buf.zres, buf.zexec, buf.zexec_stride, d_scratch, scratch_half);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        const dim3 eff_grid((buf.n_cnt_waves + kWavesPerBlock - 1) / kWavesPerBlock);
        if (reserve_execs)
            hipLaunchKernelGGL((avg_efficiency_kernel<true, true>), eff_grid, block, 0, stream, eff, table.node_slot,
                               table.n_slots, zones.n_zones, n_apps, d_apps, (const gf_result*)buf.zres,
                               (const uint32_t*)buf.zexec, buf.zexec_stride, buf.cnt, buf.n_cnt_waves, buf.zavg);
        else
            hipLaunchKernelGGL((avg_efficiency_kernel<false, true>), eff_grid, block, 0, stream, eff, table.node_slot,
                               table.n_slots, zones.n_zones, n_apps, d_apps, (const gf_result*)buf.zres,
                               (const uint32_t*)buf.zexec, buf.zexec_stride, buf.cnt, buf.n_cnt_waves, buf.zavg);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (az_aware)
        hipLaunchKernelGGL(zone_select_kernel<true>, app_grid, block, 0, stream, table.slot_node, zones.n_zones, n_apps,
                           d_apps, (const gf_result*)buf.zres, (const uint32_t*)buf.zexec, buf.zexec_stride,
                           (const double*)buf.zavg, d_results, d_exec_nodes, buf.avg_out);
    else
        hipLaunchKernelGGL(zone_select_kernel<false>, app_grid, block, 0, stream, table.slot_node, zones.n_zones, n_apps,
                           d_apps, (const gf_result*)buf.zres, (const uint32_t*)buf.zexec, buf.zexec_stride,
                           (const double*)buf.zavg, d_results, d_exec_nodes, buf.avg_out);
    return hipGetLastError();
}

hipError_t launch_fit_zoned_fused(int inner_algo, bool az_aware, const NodeTable& table, const SparseTable& gpu_view, const ZoneTable& zones,
                                  const int64_t* d_sched, uint32_t* d_zexec, uint64_t zexec_stride, uint32_t n_apps,
                                  const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                                  uint64_t scratch_half, hipStream_t stream, uint8_t* d_feasible, uint32_t* d_feasible_sync,
                                  bool eff_nonneg) {
    if (n_apps == 0) return hipSuccess;
    if (d_feasible != nullptr && d_feasible_sync == nullptr) return hipErrorInvalidValue;
    const uint32_t nonneg = eff_nonneg ? 1u : 0u;
    if (inner_algo != GF_ALGO_TIGHTLY_PACK && inner_algo != GF_ALGO_MINIMAL_FRAGMENTATION) return hipErrorInvalidValue;
    if (az_aware && inner_algo != GF_ALGO_TIGHTLY_PACK) return hipErrorInvalidValue;
    if (zones.n_zones + (az_aware ? 1u : 0u) > 64u) return hipErrorInvalidValue;
    const dim3 grid(n_apps), grid_feas(n_apps + 1) /* + the collecting workgroup */, block(kWave * kFusedWaves);
    // feasibility only (gf_fit_feasible): d_results is not written, d_exec_nodes receives nothing; d_feasible / d_feasible_sync
    // as for launch_fit_independent
#define GF_FUSED(ALGO, AZ)                                                                                                      \
    if (d_feasible != nullptr)                                                                                                  \
        hipLaunchKernelGGL((fit_zoned_fused_kernel<ALGO, AZ, true>), grid_feas, block, 0, stream, table, gpu_view, zones, d_sched, n_apps, \
                           d_apps, reinterpret_cast<gf_result*>(d_feasible), d_feasible_sync, d_zexec, zexec_stride, d_scratch, \
                           scratch_half, nonneg);                                                                               \
    else                                                                                                                        \
        hipLaunchKernelGGL((fit_zoned_fused_kernel<ALGO, AZ, false>), grid, block, 0, stream, table, gpu_view, zones, d_sched, n_apps,     \
                           d_apps, d_results, d_exec_nodes, d_zexec, zexec_stride, d_scratch, scratch_half, 0u)
    if (inner_algo == GF_ALGO_MINIMAL_FRAGMENTATION) {
        GF_FUSED(GF_ALGO_MINIMAL_FRAGMENTATION, false);
    } else if (az_aware) {
        GF_FUSED(GF_ALGO_TIGHTLY_PACK, true);
    } else {
        GF_FUSED(GF_ALGO_TIGHTLY_PACK, false);
    }
#undef GF_FUSED
    return hipGetLastError();
}

hipError_t launch_fit_fifo_generic(int inner_algo, bool zoned, bool az_aware, bool reserve_execs, const NodeTable& table,
                                   const ZoneTable& zones, const int64_t* d_sched, const ZoneBuffers& buf,
                                   uint32_t n_apps, const gf_app* d_apps, gf_result* d_results, uint32_t* d_exec_nodes,
                                   uint32_t* d_scratch, uint64_t scratch_half, int32_t* d_chain_failed_at,
                                   const int32_t* d_run_if, hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    const uint32_t n_cand = zoned ? zones.n_zones + (az_aware ? 1u : 0u) : 1u;
    if (n_cand > 64) return hipErrorInvalidValue;
    const uint32_t n_waves = n_cand < 1 ? 1 : (n_cand > 16 ? 16 : n_cand);
    const dim3 grid(1), block(kWave * n_waves);
#define GF_GEN(ALGO, ZO, AZ, RE)                                                                                        \
    hipLaunchKernelGGL((fit_fifo_generic_kernel<ALGO, ZO, AZ, RE>), grid, block, 0, stream, table, zones, d_sched, n_apps, \
                       d_apps, d_results, d_exec_nodes, buf.zexec, buf.zexec_stride, d_scratch, scratch_half, buf.cnt,  \
                       d_chain_failed_at, d_run_if)
    if (!zoned) {
        if (inner_algo == GF_ALGO_TIGHTLY_PACK)
            GF_GEN(GF_ALGO_TIGHTLY_PACK, false, false, true);
        else if (inner_algo == GF_ALGO_DISTRIBUTE_EVENLY)
            GF_GEN(GF_ALGO_DISTRIBUTE_EVENLY, false, false, true);
        else if (inner_algo == GF_ALGO_MINIMAL_FRAGMENTATION)
            GF_GEN(GF_ALGO_MINIMAL_FRAGMENTATION, false, false, false);
        else
            return hipErrorInvalidValue;
    } else {
        (void)reserve_execs;
        if (inner_algo == GF_ALGO_MINIMAL_FRAGMENTATION && !az_aware)
            GF_GEN(GF_ALGO_MINIMAL_FRAGMENTATION, true, false, false);  // minimalFragmentation reserves the driver only
        else if (inner_algo != GF_ALGO_TIGHTLY_PACK)
            return hipErrorInvalidValue;
        else if (az_aware)
            GF_GEN(GF_ALGO_TIGHTLY_PACK, true, true, true);
        else
            GF_GEN(GF_ALGO_TIGHTLY_PACK, true, false, true);
    }
#undef GF_GEN
    return hipGetLastError();
}

namespace {
inline dim3 app_grid_of(uint32_t n_apps) { return dim3((n_apps + kWavesPerBlock - 1) / kWavesPerBlock); }
}  // namespace

size_t fifo_zoned_lds_bytes(uint32_t lds_slots, uint32_t n_chunks, uint32_t n_zones, uint32_t n_cand, uint32_t n_shapes) {
    return fifo_zoned_fixed_lds(n_chunks, n_zones + 1, (uint32_t)fifo_zoned_waves(n_cand), n_shapes) + 12 * (size_t)lds_slots;
}

hipError_t launch_fit_fifo_zoned_lds(bool az_aware, const NodeTable& table, const NarrowTable& ntable, const ZoneTable& zones,
                                     const int64_t* d_sched, uint32_t lds_slots, uint32_t n_shapes, uint32_t n_apps,
                                     const gf_app* d_apps, NApp* d_napps, int32_t* d_wide_needed, gf_result* d_results,
                                     uint32_t* d_exec_nodes, uint32_t* d_spill, uint64_t spill_stride,
                                     int32_t* d_chain_failed_at, const ChainCkpt& ck, const ChainIo& io, ScanStats* d_stats,
                                     hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    if (zones.n_zones + (az_aware ? 1u : 0u) > 16 || !table.d_identity) return hipErrorInvalidValue;
    if (n_shapes == 0 || n_shapes > kZShapes) return hipErrorInvalidValue;
    hipError_t e = launch_chain_prologue(io, n_apps, d_apps, d_napps, ntable.unit, d_wide_needed, nullptr, 0, stream);
    if (e != hipSuccess) return e;
    // one wavefront per candidate view; the rest of the workgroup only helps with the prologue and shares every barrier and
    // the per-app control flow, i.e. takes issue slots from the views' wavefronts: no more wavefronts than views need
    const uint32_t n_cand = zones.n_zones + (az_aware ? 1u : 0u);
    const size_t lds = fifo_zoned_lds_bytes(lds_slots, table.n_chunks, zones.n_zones, n_cand, n_shapes);
    // ... plus one that expands the winner's placement next to the commit (none left with 16 views)
    const int wg_waves = fifo_zoned_waves(n_cand);
    const bool res = lds_slots >= table.n_slots;  // the whole table in LDS: the global-tail branch of every slot access compiles away
#define GF_ZL2(AZ, NWV, RS)                                                                                                     \
    e = launch_one_workgroup(fit_fifo_zoned_lds_kernel<AZ, NWV, RS>, NWV, lds, stream, table, ntable, zones, d_sched, lds_slots, \
                             n_apps, n_shapes, d_apps, (const NApp*)d_napps, (const int32_t*)d_wide_needed, d_results,          \
                             d_exec_nodes, d_spill, spill_stride, d_chain_failed_at, ck, d_stats)
#define GF_ZL(AZ, NWV)       \
    if (res)                 \
        GF_ZL2(AZ, NWV, true); \
    else                     \
        GF_ZL2(AZ, NWV, false)
    if (az_aware) {
        if (wg_waves == 4) {
            GF_ZL(true, 4);
        } else if (wg_waves == 8) {
            GF_ZL(true, 8);
        } else {
            GF_ZL(true, 16);
        }
    } else {
        if (wg_waves == 4) {
            GF_ZL(false, 4);
        } else if (wg_waves == 8) {
            GF_ZL(false, 8);
        } else {
            GF_ZL(false, 16);
        }
    }
#undef GF_ZL
#undef GF_ZL2
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(zoned_translate_kernel, app_grid_of(n_apps), dim3(kWave * kWavesPerBlock), 0, stream, table.slot_node,
                       n_apps, d_apps, d_results, d_exec_nodes, (const int32_t*)d_wide_needed, chain_out_of(io, d_chain_failed_at));
    return hipGetLastError();
}

size_t fifo_minfrag_hist_words(uint32_t n_zones, uint32_t n_shapes) {
    return 2 * (size_t)(n_zones ? n_zones : 1u) * n_shapes * (size_t)kMfBins;  // histograms, then first positions
}

size_t fifo_minfrag_lds_bytes(uint32_t lds_slots, uint32_t n_chunks, uint32_t n_zones, uint32_t n_shapes) {
    return fifo_minfrag_fixed_lds(n_chunks, n_zones + 1, n_shapes) + 12 * (size_t)lds_slots;
}

hipError_t launch_fit_fifo_minfrag_lds(bool zoned, const NodeTable& table, const NarrowTable& ntable, const ZoneTable& zones,
                                       const int64_t* d_sched, uint32_t lds_slots, uint32_t n_shapes, uint32_t n_idx,
                                       uint32_t n_apps,
                                       const gf_app* d_apps, NApp* d_napps, int32_t* d_wide_needed, gf_result* d_results,
                                       uint32_t* d_exec_nodes, uint32_t* d_spill, uint64_t spill_stride,
                                       int32_t* d_chain_failed_at, int32_t* d_capmat, int32_t* d_hist, const ChainCkpt& ck,
                                       const ChainIo& io, ScanStats* d_stats, hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    if ((zoned && zones.n_zones > 16) || !table.d_identity || n_shapes == 0 || n_shapes > kZShapes || n_idx > n_shapes)
        return hipErrorInvalidValue;
    if (d_capmat == nullptr) d_hist = nullptr;  // the histograms are patched together with the matrix
    // (no initialisation of d_hist: the row fill of a shape writes every bin of its histograms and first positions)
    hipError_t e = launch_chain_prologue(io, n_apps, d_apps, d_napps, ntable.unit, d_wide_needed, nullptr, 0, stream);
    if (e != hipSuccess) return e;
    const size_t lds = fifo_minfrag_lds_bytes(lds_slots, table.n_chunks, zoned ? zones.n_zones : 0u, n_idx);
    const bool res = lds_slots >= table.n_slots;
    // eight wavefronts (a 256-VGPR budget: no spills) when the candidate views, four patch helpers and the emitter fit, else sixteen
    const uint32_t n_cand_mf = zoned ? zones.n_zones : 1u;
    // ... and the whole table sits in LDS: on a 100 000-node table the block-cooperative parts (row fills, checkpoint dumps,
    // patches) want the sixteen (measured: 9.7 -> 10.2 ms plain, 15.0 -> 16.6 ms with three zones on eight)
    const bool eight = res && fifo_minfrag_waves(n_cand_mf) == 8u;
#define GF_MFL2(ZO, RS, NWV)                                                                                                   \
    e = launch_one_workgroup(fit_fifo_minfrag_lds_kernel<ZO, RS, NWV>, NWV, lds, stream, table, ntable, zones, d_sched,         \
                             lds_slots, n_apps, n_shapes, n_idx, d_apps, (const NApp*)d_napps, (const int32_t*)d_wide_needed,  \
                             d_results, d_exec_nodes, d_spill, spill_stride, d_chain_failed_at, d_capmat, d_hist, ck, d_stats)
#define GF_MFL(ZO, RS)          \
    if (eight)                  \
        GF_MFL2(ZO, RS, 8);     \
    else                        \
        GF_MFL2(ZO, RS, (int)kMfNWmax)
    if (zoned) {
        if (res) {
            GF_MFL(true, true);
        } else {
            GF_MFL(true, false);
        }
    } else {
        if (res) {
            GF_MFL(false, true);
        } else {
            GF_MFL(false, false);
        }
    }
#undef GF_MFL2
#undef GF_MFL
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(zoned_translate_kernel, app_grid_of(n_apps), dim3(kWave * kWavesPerBlock), 0, stream, table.slot_node,
                       n_apps, d_apps, d_results, d_exec_nodes, (const int32_t*)d_wide_needed, chain_out_of(io, d_chain_failed_at));
    return hipGetLastError();
}

hipError_t launch_avg_efficiency(bool reserve_execs, const NodeTable& table, const EffTables& eff, uint32_t* d_cnt,
                                 uint32_t n_cnt_waves, uint32_t n_apps, const gf_app* d_apps,
                                 const gf_result* d_results, const uint32_t* d_exec_nodes, double* d_avg_out,
                                 hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    const dim3 block(kWave * kWavesPerBlock);
    const dim3 grid((n_cnt_waves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (reserve_execs)
        hipLaunchKernelGGL((avg_efficiency_kernel<true, false>), grid, block, 0, stream, eff, table.node_slot,
                           table.n_slots, 1u, n_apps, d_apps, d_results, d_exec_nodes, (uint64_t)0, d_cnt, n_cnt_waves,
                           d_avg_out);
    else
        hipLaunchKernelGGL((avg_efficiency_kernel<false, false>), grid, block, 0, stream, eff, table.node_slot,
                           table.n_slots, 1u, n_apps, d_apps, d_results, d_exec_nodes, (uint64_t)0, d_cnt, n_cnt_waves,
                           d_avg_out);
    return hipGetLastError();
}

hipError_t launch_node_efficiencies(bool reserve_execs, const EffTables& eff_by_node, uint32_t n_nodes, int32_t k,
                                    const gf_app* d_app, const gf_result* d_result, const uint32_t* d_exec_nodes,
                                    int64_t* d_reserved, double* d_eff_out, hipStream_t stream) {
    if (n_nodes == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_reserved, 0, 3 * (size_t)n_nodes * sizeof(int64_t), stream);
    if (e != hipSuccess) return e;
    const unsigned k_blocks = (unsigned)(((k > 0 ? k : 1) + 255) / 256);
    if (reserve_execs)
        hipLaunchKernelGGL(reserved_scatter_kernel<true>, dim3(k_blocks), dim3(256), 0, stream, d_app, d_result,
                           d_exec_nodes, n_nodes, reinterpret_cast<unsigned long long*>(d_reserved));
    else
        hipLaunchKernelGGL(reserved_scatter_kernel<false>, dim3(1), dim3(64), 0, stream, d_app, d_result, d_exec_nodes,
                           n_nodes, reinterpret_cast<unsigned long long*>(d_reserved));
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(node_efficiency_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, stream, eff_by_node, n_nodes,
                       (const int64_t*)d_reserved, d_eff_out);
    return hipGetLastError();
}

// ---- node-range sharding (gangfit_shard.inc)
namespace {
inline dim3 app_grid(uint32_t n_apps) { return dim3((n_apps + kWavesPerBlock - 1) / kWavesPerBlock); }
}  // namespace

inline dim3 shard_grid(uint32_t n_apps, const ShardSet& set) { return dim3((n_apps + kWavesPerBlock - 1) / kWavesPerBlock, set.n); }

hipError_t launch_shard_partials(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, const ShardSet& set,
                                 uint32_t n_apps, const gf_app* d_apps, gf_shard_partial* d_out, const PeerPtrs& dsts,
                                 hipStream_t stream) {
    if (n_apps == 0 || set.n == 0) return hipSuccess;
    const dim3 block(kWave * kWavesPerBlock);
    if (algo == GF_ALGO_TIGHTLY_PACK)
        hipLaunchKernelGGL(shard_partials_kernel<GF_ALGO_TIGHTLY_PACK>, shard_grid(n_apps, set), block, 0, stream, table, gpu_view,
                           set, n_apps, d_apps, d_out, dsts);
    else
        hipLaunchKernelGGL(shard_partials_kernel<GF_ALGO_DISTRIBUTE_EVENLY>, shard_grid(n_apps, set), block, 0, stream, table,
                           gpu_view, set, n_apps, d_apps, d_out, dsts);
    return hipGetLastError();
}

hipError_t launch_shard_drivers(const NodeTable& table, const ShardSet& set, uint32_t n_apps, const gf_app* d_apps,
                                const gf_shard_partial* d_all_partials, gf_shard_driver* d_out, const PeerPtrs& dsts,
                                hipStream_t stream) {
    if (n_apps == 0 || set.n == 0) return hipSuccess;
    hipLaunchKernelGGL(shard_drivers_kernel, shard_grid(n_apps, set), dim3(kWave * kWavesPerBlock), 0, stream, table, set,
                       n_apps, d_apps, d_all_partials, d_out, dsts);
    return hipGetLastError();
}

hipError_t launch_shard_emit(gf_algo algo, const NodeTable& table, const SparseTable& gpu_view, const ShardSet& set,
                             uint32_t n_apps, const gf_app* d_apps, const gf_shard_partial* d_all_partials,
                             const gf_shard_driver* d_all_drivers, gf_result* d_results, uint32_t* d_exec2, uint64_t half,
                             hipStream_t stream) {
    if (n_apps == 0 || set.n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_exec2, 0, 2 * half * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const dim3 block(kWave * kWavesPerBlock);
    if (algo == GF_ALGO_TIGHTLY_PACK)
        hipLaunchKernelGGL(shard_emit_kernel<GF_ALGO_TIGHTLY_PACK>, shard_grid(n_apps, set), block, 0, stream, table, gpu_view,
                           set, n_apps, d_apps, d_all_partials, d_all_drivers, d_results, d_exec2, half);
    else
        hipLaunchKernelGGL(shard_emit_kernel<GF_ALGO_DISTRIBUTE_EVENLY>, shard_grid(n_apps, set), block, 0, stream, table,
                           gpu_view, set, n_apps, d_apps, d_all_partials, d_all_drivers, d_results, d_exec2, half);
    return hipGetLastError();
}

hipError_t launch_shard_finish(gf_algo algo, uint32_t n_shards, uint32_t n_apps, const gf_app* d_apps,
                               const gf_shard_partial* d_all_partials, const gf_shard_driver* d_all_drivers,
                               const gf_result* d_results, uint32_t* d_exec2, uint64_t half, hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    const dim3 block(kWave * kWavesPerBlock);
    if (algo == GF_ALGO_TIGHTLY_PACK)
        hipLaunchKernelGGL(shard_finish_kernel<GF_ALGO_TIGHTLY_PACK>, app_grid(n_apps), block, 0, stream, n_shards, n_apps,
                           d_apps, d_all_partials, d_all_drivers, d_results, d_exec2, half);
    else
        hipLaunchKernelGGL(shard_finish_kernel<GF_ALGO_DISTRIBUTE_EVENLY>, app_grid(n_apps), block, 0, stream, n_shards,
                           n_apps, d_apps, d_all_partials, d_all_drivers, d_results, d_exec2, half);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void narrow_rescale_kernel(const int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                             uint32_t n_slots, const int32_t* __restrict__ cm_src,
                                                             int32_t* __restrict__ cm_dst, uint32_t n_chunks, int32_t f0,
                                                             int32_t f1, int32_t f2) {
    const size_t n_tab = 3 * (size_t)n_slots, n_all = n_tab + 3 * (size_t)n_chunks;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += stride) {
        const bool tab = i < n_tab;
        const size_t k = tab ? i : i - n_tab;
        const uint32_t dim = (uint32_t)(k / (tab ? n_slots : n_chunks));
        const int32_t f = dim == 0 ? f0 : (dim == 1 ? f1 : f2);
        const int32_t v = tab ? src[k] : cm_src[k];
        const int32_t o = v <= INT32_MIN / 2 ? v : v * f;  // empty slots / all-empty chunks keep their sentinel
        if (tab)
            dst[k] = o;
        else
            cm_dst[k] = o;
    }
}

hipError_t launch_narrow_rescale(const int32_t* d_src, int32_t* d_dst, uint32_t n_slots, const int32_t* d_cmax_src,
                                 int32_t* d_cmax_dst, uint32_t n_chunks, const int32_t factor[3], hipStream_t stream) {
    const size_t n_all = 3 * ((size_t)n_slots + n_chunks);
    if (n_all == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n_all + 255) / 256 < 2048 ? (n_all + 255) / 256 : 2048);
    hipLaunchKernelGGL(narrow_rescale_kernel, dim3(blocks), dim3(256), 0, stream, d_src, d_dst, n_slots, d_cmax_src, d_cmax_dst,
                       n_chunks, factor[0], factor[1], factor[2]);
    return hipGetLastError();
}

hipError_t launch_shard_reduce_pull(const PeerPtrs& srcs, uint32_t* d_dst, size_t n, hipStream_t stream) {
    if (n == 0 || srcs.n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(shard_reduce_pull_kernel, dim3(blocks), dim3(256), 0, stream, srcs, d_dst, n);
    return hipGetLastError();
}

hipError_t launch_executor_fit(bool minimal_fragmentation, const NodeTable& table, const int64_t* d_reserved, uint32_t n_req,
                               const int64_t* d_exe, const uint32_t* d_hosts, uint32_t hosts_stride, const uint32_t* d_node_zone,
                               const uint32_t* d_req_zone, uint32_t* d_node_out, hipStream_t stream) {
    if (n_req == 0) return hipSuccess;
    const dim3 block(kWave * kWavesPerBlock);
    if (minimal_fragmentation)
        hipLaunchKernelGGL(executor_fit_kernel<true>, app_grid(n_req), block, 0, stream, table, d_reserved, n_req, d_exe,
                           d_hosts, hosts_stride, d_node_zone, d_req_zone, d_node_out);
    else
        hipLaunchKernelGGL(executor_fit_kernel<false>, app_grid(n_req), block, 0, stream, table, d_reserved, n_req, d_exe,
                           d_hosts, hosts_stride, d_node_zone, d_req_zone, d_node_out);
    return hipGetLastError();
}

hipError_t launch_find_nodes(bool chained, const NodeTable& table, uint32_t n_req, const int64_t* d_exe, const int32_t* d_k,
                             const uint64_t* d_exec_off, gf_find_result* d_results, uint32_t* d_exec_nodes,
                             uint32_t* d_adds_out, hipStream_t stream) {
    if (n_req == 0) return hipSuccess;
    if (chained)
        hipLaunchKernelGGL(find_nodes_kernel<true>, dim3(1), dim3(kWave * kWavesPerBlock), 0, stream, table, n_req, d_exe, d_k,
                           d_exec_off, d_results, d_exec_nodes, d_adds_out);
    else
        hipLaunchKernelGGL(find_nodes_kernel<false>, app_grid(n_req), dim3(kWave * kWavesPerBlock), 0, stream, table, n_req,
                           d_exe, d_k, d_exec_off, d_results, d_exec_nodes, d_adds_out);
    return hipGetLastError();
}

hipError_t launch_selftest(uint64_t seed, uint32_t n_cases, uint32_t* d_mismatch, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(64), dim3(kWave), 0, stream, seed, n_cases, d_mismatch);
    return hipGetLastError();
}

}  // namespace gangfit
