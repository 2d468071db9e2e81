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
// No MFMA (nothing here is a contraction), no LDS staging in this first version (node table is read through
// L1/L2 with coalesced 8-byte-per-lane loads) — see DESIGN.md for the roofline discussion.

#include "gangfit_device.h"

namespace gangfit {

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;  // independent-batch kernel: 4 apps per 256-thread workgroup

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

// ------------------------------------------------------------------------------------------------ app registers

struct App {
    int64_t drv0, drv1, drv2;
    int64_t exe0, exe1, exe2;
    double rcp0, rcp1, rcp2;  // 1.0 / exe_j (only used when exe_j > 0)
    int32_t k;
    uint32_t flags;
    uint64_t exec_off;
};

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
    r.rcp0 = r.exe0 > 0 ? 1.0 / (double)r.exe0 : 0.0;
    r.rcp1 = r.exe1 > 0 ? 1.0 / (double)r.exe1 : 0.0;
    r.rcp2 = r.exe2 > 0 ? 1.0 / (double)r.exe2 : 0.0;
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
__device__ __forceinline__ bool cap_ge1(int64_t a0, int64_t a1, int64_t a2, const App& app) {
    return app.exe0 <= a0 && app.exe1 <= a1 && app.exe2 <= a2;
}

// !driverResources.GreaterThan(available)  (LIB/binpack/binpack.go:69, LIB/resources/resources.go:239-241)
__device__ __forceinline__ bool driver_fits(int64_t a0, int64_t a1, int64_t a2, const App& app) {
    return app.drv0 <= a0 && app.drv1 <= a1 && app.drv2 <= a2;
}

// ------------------------------------------------------------------------------------------------ emission

// Lane `lane` owns a run of t copies of `node` starting at out[start].  Short runs: per-lane loop.  Long runs are
// written cooperatively by all 64 lanes (coalesced) so that one huge node cannot serialise the wave.
__device__ __forceinline__ void emit_runs(uint32_t* __restrict__ out, int64_t start, int32_t t, uint32_t node,
                                          int lane) {
    constexpr int32_t kLong = 16;
    uint64_t long_mask = __ballot(t > kLong);
    if (t > 0 && t <= kLong) {
        for (int32_t i = 0; i < t; ++i) out[start + i] = node;
    }
    while (long_mask) {
        const int src = __ffsll((unsigned long long)long_mask) - 1;
        long_mask &= long_mask - 1;
        const int64_t s = read_lane(start, src);
        const int32_t n = read_lane(t, src);
        const uint32_t nd = read_lane(node, src);
        for (int32_t i = lane; i < n; i += kWave) out[s + i] = nd;
    }
}

// ------------------------------------------------------------------------------------------------ driver scan

// First position p in [from, n_d) of driverNodePriorityOrder whose node passes the driver-fit check, else -1.
__device__ __forceinline__ int64_t first_fitting_driver(const NodeTable& T, const App& app, uint32_t from, int lane,
                                                        unsigned long long& visited) {
    for (uint32_t b = from; b < T.n_d; b += kWave) {
        const uint32_t i = b + lane;
        bool fit = false;
        if (i < T.n_d) {
            const uint32_t s = T.dslot[i];
            fit = driver_fits(T.cpu[s], T.mem[s], T.gpu[s], app);
        }
        visited += (T.n_d - b) < (uint32_t)kWave ? (T.n_d - b) : (uint32_t)kWave;
        const uint64_t m = __ballot(fit);
        if (m) return (int64_t)b + (__ffsll((unsigned long long)m) - 1);
    }
    return -1;
}

// Clamped capacities of one slot without / with the driver as base (wave-uniform slot -> every lane computes the
// same values; used only on the rare fallback path).
__device__ __forceinline__ void slot_caps(const NodeTable& T, const App& app, uint32_t s, int32_t& c0, int32_t& cd) {
    const int64_t a0 = T.cpu[s], a1 = T.mem[s], a2 = T.gpu[s];
    c0 = cap3(a0, a1, a2, app);
    cd = cap3(a0 - app.drv0, a1 - app.drv1, a2 - app.drv2, app);
}

// O(N) driver choice on the fallback path: first position p > after in driver order with
//   fit(p) && total(p) >= K,  total = S            if the node is not an executor candidate
//                                   = S - c0 + cd   otherwise           (SURVEY.md section 8 "O(N) driver choice")
__device__ __forceinline__ int64_t next_feasible_driver(const NodeTable& T, const App& app, uint32_t from, int64_t S,
                                                        int lane, unsigned long long& visited) {
    for (uint32_t b = from; b < T.n_d; b += kWave) {
        const uint32_t i = b + lane;
        bool ok = false;
        if (i < T.n_d) {
            const uint32_t s = T.dslot[i];
            const int64_t a0 = T.cpu[s], a1 = T.mem[s], a2 = T.gpu[s];
            if (driver_fits(a0, a1, a2, app)) {
                int64_t total = S;
                if (s < T.n_x) {
                    const int32_t c0 = cap3(a0, a1, a2, app);
                    const int32_t cd = cap3(a0 - app.drv0, a1 - app.drv1, a2 - app.drv2, app);
                    total = S - c0 + cd;
                }
                ok = total >= (int64_t)app.k;
            }
        }
        visited += (T.n_d - b) < (uint32_t)kWave ? (T.n_d - b) : (uint32_t)kWave;
        const uint64_t m = __ballot(ok);
        if (m) return (int64_t)b + (__ffsll((unsigned long long)m) - 1);
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------ TightlyPack scan

// tightlyPackExecutors with the driver reserved on slot ds.  Returns sum of clamped capacities over the visited
// prefix (>= K  <=>  feasible; the scan stops at the first chunk where K is reached).  Writes placements.
__device__ __forceinline__ int64_t tight_scan(const NodeTable& T, const App& app, uint32_t ds,
                                              uint32_t* __restrict__ out, int lane, unsigned long long& visited) {
    const int64_t K = app.k;
    int64_t taken = 0;
    for (uint32_t b = 0; b < T.n_x; b += kWave) {
        const uint32_t j = b + lane;
        int32_t c = 0;
        uint32_t node = GF_NO_NODE;
        if (j < T.n_x) {
            int64_t a0 = T.cpu[j], a1 = T.mem[j], a2 = T.gpu[j];
            node = T.slot_node[j];
            if (j == ds) {
                a0 -= app.drv0;
                a1 -= app.drv1;
                a2 -= app.drv2;
            }
            c = cap3(a0, a1, a2, app);
        }
        visited += (T.n_x - b) < (uint32_t)kWave ? (T.n_x - b) : (uint32_t)kWave;
        const int32_t incl = wave_inclusive_scan(c);
        const int32_t tot = read_lane(incl, kWave - 1);
        if (tot > 0) {
            const int64_t start = taken + (int64_t)(incl - c);
            const int64_t room = K - start;
            const int32_t t = room <= 0 ? 0 : (room < (int64_t)c ? (int32_t)room : c);
            emit_runs(out, start, t, node, lane);
        }
        taken += tot;
        if (taken >= K) break;
    }
    return taken;
}

// ------------------------------------------------------------------------------------------------ DistributeEvenly

// Pass r = 1 of distributeExecutorsEvenly, lazily: every node with cap >= 1, in order, until K are placed.
// Records the surviving slots (needed for passes >= 2) in surv[].  Returns the number of nodes with cap >= 1 seen
// (>= K means the app is placed entirely by pass 1).
__device__ __forceinline__ int64_t even_pass1(const NodeTable& T, const App& app, uint32_t ds,
                                              uint32_t* __restrict__ out, uint32_t* __restrict__ surv, int lane,
                                              unsigned long long& visited) {
    const int64_t K = app.k;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    int64_t taken = 0;
    for (uint32_t b = 0; b < T.n_x; b += kWave) {
        const uint32_t j = b + lane;
        bool flag = false;
        uint32_t node = GF_NO_NODE;
        if (j < T.n_x) {
            int64_t a0 = T.cpu[j], a1 = T.mem[j], a2 = T.gpu[j];
            node = T.slot_node[j];
            if (j == ds) {
                a0 -= app.drv0;
                a1 -= app.drv1;
                a2 -= app.drv2;
            }
            flag = cap_ge1(a0, a1, a2, app);
        }
        visited += (T.n_x - b) < (uint32_t)kWave ? (T.n_x - b) : (uint32_t)kWave;
        const uint64_t m = __ballot(flag);
        const int64_t pos = taken + (int64_t)__popcll((unsigned long long)(m & lt_mask));
        if (flag && pos < K) {
            out[pos] = node;
            surv[pos] = j;
        }
        taken += (int64_t)__popcll((unsigned long long)m);
        if (taken >= K) break;
    }
    return taken;
}

// Passes r >= 2 when pass 1 found m1 < K nodes.  surv[0..m1) = surviving slots in order; caps[] receives their
// clamped capacities.  Returns S = sum of capacities (feasible <=> S >= K) and, when feasible, completes out[].
__device__ __forceinline__ int64_t even_general(const NodeTable& T, const App& app, uint32_t ds,
                                                uint32_t* __restrict__ out, const uint32_t* __restrict__ surv,
                                                uint32_t* __restrict__ caps, int64_t m1, int lane) {
    const int64_t K = app.k;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // surv[] was written rank-wise by other lanes of this wave in pass 1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // capacities of the survivors
    int64_t S = 0;
    for (int64_t b = 0; b < m1; b += kWave) {
        const int64_t i = b + lane;
        int32_t c = 0;
        if (i < m1) {
            const uint32_t j = surv[i];
            int64_t a0 = T.cpu[j], a1 = T.mem[j], a2 = T.gpu[j];
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // caps[] written above are re-read below by other lanes
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
            if (flag && p < K) out[p] = T.slot_node[j];
            pos += (int64_t)__popcll((unsigned long long)m);
        }
    }
    return S;
}

// ------------------------------------------------------------------------------------------------ one decision

struct Decision {
    bool feasible;
    uint32_t dpos;     // position in driver order
    uint32_t ds;       // driver slot
    int64_t pass1;     // DistributeEvenly: number of pass-1 placements (first occurrences); TightlyPack: unused
};

// SparkBinPack for one app by one wave.  out = exec_nodes + exec_off.  scratch_a / scratch_b: K uint32 each.
template <int ALGO>
__device__ __forceinline__ Decision decide(const NodeTable& T, const App& app, uint32_t* __restrict__ out,
                                           uint32_t* __restrict__ scratch_a, uint32_t* __restrict__ scratch_b,
                                           int lane, unsigned long long& xvis, unsigned long long& dvis) {
    Decision dec;
    dec.feasible = false;
    dec.dpos = 0;
    dec.ds = 0;
    dec.pass1 = 0;
    const int64_t K = app.k;

    // (1) first driver candidate that passes the driver-fit check (binpack.go:67-71)
    int64_t p0 = first_fitting_driver(T, app, 0, lane, dvis);
    if (p0 < 0) return dec;
    uint32_t ds = T.dslot[p0];
    if (K == 0) {  // pack_tightly.go:42-44 / distribute_evenly.go:46-48: nothing to place
        dec.feasible = true;
        dec.dpos = (uint32_t)p0;
        dec.ds = ds;
        return dec;
    }

    // (2) executors with the driver reserved on that candidate — the common case ends here
    int64_t S_d;  // sum over executor order of min(cap(n, base_d), K) (exact whenever < K)
    int64_t pass1 = 0;
    if (ALGO == GF_ALGO_TIGHTLY_PACK) {
        S_d = tight_scan(T, app, ds, out, lane, xvis);
    } else {
        pass1 = even_pass1(T, app, ds, out, scratch_a, lane, xvis);
        S_d = pass1 >= K ? pass1 : even_general(T, app, ds, out, scratch_a, scratch_b, pass1, lane);
    }
    if (S_d >= K) {
        dec.feasible = true;
        dec.dpos = (uint32_t)p0;
        dec.ds = ds;
        dec.pass1 = pass1;
        return dec;
    }

    // (3) rare: the first candidate's node is where the executors were needed.  S_d is now the exact total with
    //     the driver on ds; recover S (no driver anywhere) and pick the first later candidate d with
    //     fit(d) && S - c0[d] + cd[d] >= K.  Same answer as the reference's retry loop (binpack.go:67-85).
    int64_t S = S_d;
    if (ds < T.n_x) {
        int32_t c0, cd;
        slot_caps(T, app, ds, c0, cd);
        S = S_d + c0 - cd;
    }
    if (S < K) return dec;
    const int64_t p1 = next_feasible_driver(T, app, (uint32_t)p0 + 1, S, lane, dvis);
    if (p1 < 0) return dec;
    ds = T.dslot[p1];
    if (ALGO == GF_ALGO_TIGHTLY_PACK) {
        S_d = tight_scan(T, app, ds, out, lane, xvis);
    } else {
        pass1 = even_pass1(T, app, ds, out, scratch_a, lane, xvis);
        S_d = pass1 >= K ? pass1 : even_general(T, app, ds, out, scratch_a, scratch_b, pass1, lane);
    }
    dec.feasible = S_d >= K;  // always true here; kept as a guard so a logic error shows up as a parity failure
    dec.dpos = (uint32_t)p1;
    dec.ds = ds;
    dec.pass1 = pass1;
    return dec;
}

__device__ __forceinline__ void write_result(gf_result* __restrict__ results, uint32_t a, const NodeTable& T,
                                             const App& app, const Decision& dec, int lane) {
    if (lane == 0) {
        gf_result r;
        r.has_capacity = dec.feasible ? 1 : 0;
        r.driver_node = dec.feasible ? T.slot_node[dec.ds] : GF_NO_NODE;
        r.exec_len = dec.feasible ? (uint32_t)app.k : 0u;
        r.evaluated = 1;
        results[a] = r;
    }
}

// ------------------------------------------------------------------------------------------------ kernels

// Independent batch: one wave per app, 4 apps per workgroup.  Grid = ceil(n_apps / 4) >> 256 CUs at the target sizes.
template <int ALGO>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void fit_independent_kernel(
    NodeTable T, uint32_t n_apps, const gf_app* __restrict__ apps, gf_result* __restrict__ results,
    uint32_t* __restrict__ exec_nodes, uint32_t* __restrict__ scratch, uint64_t scratch_half,
    ScanStats* __restrict__ stats) {
    const int lane = lane_id();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t a = blockIdx.x * kWavesPerBlock + wave;
    if (a >= n_apps) return;
    const App app = load_app(apps, a);
    unsigned long long xvis = 0, dvis = 0;
    const Decision dec = decide<ALGO>(T, app, exec_nodes + app.exec_off, scratch + app.exec_off,
                                      scratch + scratch_half + app.exec_off, lane, xvis, dvis);
    write_result(results, a, T, app, dec, lane);
    if (stats != nullptr && lane == 0) {
        atomicAdd(&stats->exec_slots_visited, xvis);
        atomicAdd(&stats->driver_slots_visited, dvis);
    }
}

// sparkResourceUsage + SubtractUsageIfExists (internal/extender/sparkpods.go:139-146, LIB/resources/resources.go:129-135)
// applied from the placement list the wave has just written: ONE executor request per distinct executor node;
// the driver request only if no executor sits on the driver node.
template <int ALGO>
__device__ __forceinline__ void commit_usage(const NodeTable& T, const App& app, const Decision& dec,
                                             const uint32_t* __restrict__ out, int lane) {
    const int64_t K = app.k;
    const uint32_t dnode = T.slot_node[dec.ds];
    bool driver_hosts_exec = false;
    // out[] was written by other lanes of this wave: make it visible before re-reading it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int64_t first_region = (ALGO == GF_ALGO_TIGHTLY_PACK) ? K : (dec.pass1 < K ? dec.pass1 : K);
    for (int64_t b = 0; b < K; b += kWave) {
        const int64_t i = b + lane;
        bool first = false;
        uint32_t node = GF_NO_NODE;
        if (i < K) {
            node = out[i];
            if (ALGO == GF_ALGO_TIGHTLY_PACK)
                first = (i == 0) || (out[i - 1] != node);  // placements are node-major runs
            else
                first = i < first_region;  // pass 1 lists every executor node exactly once
        }
        if (first) {
            const uint32_t s = T.node_slot[node];
            T.cpu[s] -= app.exe0;
            T.mem[s] -= app.exe1;
            T.gpu[s] -= app.exe2;
        }
        if (__ballot(i < K && node == dnode)) driver_hosts_exec = true;
    }
    if (!driver_hosts_exec && lane == 0) {
        T.cpu[dec.ds] -= app.drv0;
        T.mem[dec.ds] -= app.drv1;
        T.gpu[dec.ds] -= app.drv2;
    }
    // the next app's scan (other lanes) must observe the residuals
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// FIFO chain: sequential over apps (each sees the residuals left by its predecessors), nodes in parallel.
// Single wave in this version: no workgroup barriers on the critical path.
template <int ALGO>
__global__ __launch_bounds__(kWave) void fit_fifo_chain_kernel(NodeTable T, uint32_t n_apps,
                                                               const gf_app* __restrict__ apps,
                                                               gf_result* __restrict__ results,
                                                               uint32_t* __restrict__ exec_nodes,
                                                               uint32_t* __restrict__ scratch, uint64_t scratch_half,
                                                               int32_t* __restrict__ chain_failed_at,
                                                               ScanStats* __restrict__ stats) {
    const int lane = lane_id();
    unsigned long long xvis = 0, dvis = 0;
    int32_t failed_at = -1;
    uint32_t a = 0;
    for (; a < n_apps; ++a) {
        const App app = load_app(apps, a);
        uint32_t* out = exec_nodes + app.exec_off;
        const Decision dec = decide<ALGO>(T, app, out, scratch + app.exec_off, scratch + scratch_half + app.exec_off,
                                          lane, xvis, dvis);
        write_result(results, a, T, app, dec, lane);
        if (a + 1 == n_apps) {
            ++a;
            break;  // the driver being filtered: nothing is subtracted after it (resource.go:321-328)
        }
        if (!dec.feasible) {
            if (app.flags & GF_APP_SKIPPABLE) continue;  // resource.go:244-248
            failed_at = (int32_t)a;                       // resource.go:249-251
            ++a;
            break;
        }
        commit_usage<ALGO>(T, app, dec, out, lane);
    }
    // apps behind an abort are reported as not evaluated
    for (uint32_t r = a + lane; r < n_apps; r += kWave) {
        gf_result z;
        z.has_capacity = 0;
        z.driver_node = GF_NO_NODE;
        z.exec_len = 0;
        z.evaluated = 0;
        results[r] = z;
    }
    if (lane == 0) {
        if (chain_failed_at != nullptr) *chain_failed_at = failed_at;
        if (stats != nullptr) {
            atomicAdd(&stats->exec_slots_visited, xvis);
            atomicAdd(&stats->driver_slots_visited, dvis);
        }
    }
}

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
        const int ebits = 1 + (int)(r0 % 61);                    // divisor magnitude 2^1 .. 2^61
        int64_t e = (int64_t)(r1 >> (64 - ebits));
        if (e == 0) e = 1;
        const int32_t k = (int32_t)(1 + (r2 % (uint64_t)GF_MAX_K));
        const int64_t lim = (int64_t)((1ull << 62) - 1);
        int64_t a;
        switch ((r0 >> 8) % 4) {
        case 0: a = (int64_t)(splitmix64(s) >> 2); break;                       // uniform in [0, 2^62)
        case 1: {                                                                // just around a multiple of e
            const int64_t mult = (int64_t)(splitmix64(s) % (uint64_t)(2 * (int64_t)k + 3));
            const int64_t maxq = lim / e;
            const int64_t qq = mult < maxq ? mult : maxq;
            a = qq * e + (int64_t)(splitmix64(s) % 3) - 1;
            break;
        }
        case 2: a = (int64_t)(splitmix64(s) % (uint64_t)(e)) ; break;          // below the divisor
        default: a = -(int64_t)(splitmix64(s) >> 3); break;                     // negative availability
        }
        if (a > lim) a = lim;
        const double rcp = 1.0 / (double)e;
        int32_t want;
        if (a < 0) want = 0;
        else {
            const int64_t q = a / e;
            want = q < (int64_t)k ? (int32_t)q : k;
        }
        if (cap_dim(a, e, rcp, k) != want) ++bad;
        if (cap_dim(a, 0, 0.0, k) != (a < 0 ? 0 : k)) ++bad;
    }
    if (bad) atomicAdd(mismatch, bad);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ launchers

hipError_t launch_fit_independent(gf_algo algo, const NodeTable& table, uint32_t n_apps, const gf_app* d_apps,
                                  gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                                  uint64_t scratch_half, ScanStats* d_stats, hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    const dim3 block(kWave * kWavesPerBlock);
    const dim3 grid((n_apps + kWavesPerBlock - 1) / kWavesPerBlock);
    if (algo == GF_ALGO_TIGHTLY_PACK)
        hipLaunchKernelGGL(fit_independent_kernel<GF_ALGO_TIGHTLY_PACK>, grid, block, 0, stream, table, n_apps,
                           d_apps, d_results, d_exec_nodes, d_scratch, scratch_half, d_stats);
    else
        hipLaunchKernelGGL(fit_independent_kernel<GF_ALGO_DISTRIBUTE_EVENLY>, grid, block, 0, stream, table, n_apps,
                           d_apps, d_results, d_exec_nodes, d_scratch, scratch_half, d_stats);
    return hipGetLastError();
}

hipError_t launch_fit_fifo_chain(gf_algo algo, const NodeTable& table, uint32_t n_apps, const gf_app* d_apps,
                                 gf_result* d_results, uint32_t* d_exec_nodes, uint32_t* d_scratch,
                                 uint64_t scratch_half, int32_t* d_chain_failed_at, ScanStats* d_stats,
                                 hipStream_t stream) {
    if (n_apps == 0) return hipSuccess;
    if (algo == GF_ALGO_TIGHTLY_PACK)
        hipLaunchKernelGGL(fit_fifo_chain_kernel<GF_ALGO_TIGHTLY_PACK>, dim3(1), dim3(kWave), 0, stream, table,
                           n_apps, d_apps, d_results, d_exec_nodes, d_scratch, scratch_half, d_chain_failed_at,
                           d_stats);
    else
        hipLaunchKernelGGL(fit_fifo_chain_kernel<GF_ALGO_DISTRIBUTE_EVENLY>, dim3(1), dim3(kWave), 0, stream, table,
                           n_apps, d_apps, d_results, d_exec_nodes, d_scratch, scratch_half, d_chain_failed_at,
                           d_stats);
    return hipGetLastError();
}

hipError_t launch_selftest(uint64_t seed, uint32_t n_cases, uint32_t* d_mismatch, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(64), dim3(kWave), 0, stream, seed, n_cases, d_mismatch);
    return hipGetLastError();
}

}  // namespace gangfit
