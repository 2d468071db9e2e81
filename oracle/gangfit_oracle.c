/*
 * gangfit_oracle.c — CPU ORACLE (test infrastructure, NOT product code; see gangfit_oracle.h).
 *
 * Two independent restatements of the reference's gang bin-packing:
 *   (i)  literal   — the add-then-compare loops exactly as the Go code runs them, Go maps keyed by node name
 *                    replaced by arrays keyed by the dense node index;
 *   (ii) closed form — floor-division capacities + O(N) driver choice.
 * tests/ require (i) == (ii) on every input; the HIP path is compared with (i).
 *
 * File:line citations are relative to /root/reference;
 * LIB = vendor/github.com/palantir/k8s-spark-scheduler-lib/pkg.
 */
#include "gangfit_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- Resources (LIB/resources/resources.go) */

/* Resources.GreaterThan, resources.go:239-241: ANY component greater. */
static int res_greater_than(const int64_t a[3], const int64_t b[3]) {
    return a[0] > b[0] || a[1] > b[1] || a[2] > b[2];
}
static void res_add(int64_t a[3], const int64_t b[3]) { /* resources.go:201-205 */
    a[0] += b[0];
    a[1] += b[1];
    a[2] += b[2];
}
static void res_sub(int64_t a[3], const int64_t b[3]) { /* resources.go:207-211 */
    a[0] -= b[0];
    a[1] -= b[1];
    a[2] -= b[2];
}

/* ---------------------------------------------------------------- reservedResources map (NodeGroupResources) */

/* `reserved := make(resources.NodeGroupResources, ...)` (binpack.go:72): a fresh, empty map per driver candidate.
 * Array keyed by node index + a touched list so that "fresh" costs O(entries) instead of O(N). */
typedef struct {
    int64_t *r;        /* n_nodes x 3 */
    uint8_t *present;  /* key exists */
    uint32_t *touched;
    uint32_t n_touched;
    uint32_t n_nodes;
} reserved_map;

static int rm_init(reserved_map *m, uint32_t n_nodes) {
    m->n_nodes = n_nodes;
    m->n_touched = 0;
    m->r = (int64_t *)calloc((size_t)n_nodes * 3 + 3, sizeof(int64_t));
    m->present = (uint8_t *)calloc((size_t)n_nodes + 1, 1);
    m->touched = (uint32_t *)malloc(((size_t)n_nodes + 1) * sizeof(uint32_t));
    return m->r && m->present && m->touched;
}
static void rm_free(reserved_map *m) {
    free(m->r);
    free(m->present);
    free(m->touched);
}
static void rm_clear(reserved_map *m) {
    for (uint32_t i = 0; i < m->n_touched; ++i) {
        uint32_t n = m->touched[i];
        m->present[n] = 0;
        m->r[3 * n] = m->r[3 * n + 1] = m->r[3 * n + 2] = 0;
    }
    m->n_touched = 0;
}
/* `if reservedResources[n] == nil { reservedResources[n] = resources.Zero() }` */
static int64_t *rm_get_or_zero(reserved_map *m, uint32_t n) {
    if (!m->present[n]) {
        m->present[n] = 1;
        m->touched[m->n_touched++] = n;
    }
    return &m->r[3 * n];
}

/* ---------------------------------------------------------------- literal executor packers */

/* tightlyPackExecutors, LIB/binpack/pack_tightly.go:34-63 */
static int tightly_pack_executors(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t k,
                                  const uint32_t *order, uint32_t n_x, reserved_map *reserved,
                                  uint32_t *out) {
    int32_t count = 0;
    if (k == 0) return 1; /* :42-44 */
    for (uint32_t i = 0; i < n_x; ++i) {
        uint32_t n = order[i];
        if (n >= n_nodes) continue; /* :51 `!ok`: Add, Sub, break — no placement, entry is never read again */
        int64_t *rsv = rm_get_or_zero(reserved, n); /* :46-48 */
        for (;;) {
            res_add(rsv, exe);                               /* :50 */
            if (res_greater_than(rsv, &avail[3 * n])) {      /* :52 */
                res_sub(rsv, exe);                           /* :53 */
                break;
            }
            out[count++] = n;                                /* :56 */
            if (count == k) return 1;                        /* :57-59 */
        }
    }
    return 0; /* :62 */
}

/* distributeExecutorsEvenly, LIB/binpack/distribute_evenly.go:34-73 */
static int distribute_executors_evenly(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t k,
                                       const uint32_t *order, uint32_t n_x, reserved_map *reserved,
                                       uint32_t *out) {
    /* availableNodes := map[name]bool (:41-44).  Known names: flag by node index (so a duplicated name is ONE key,
     * as in the Go map).  Unknown names (>= n_nodes) are deleted on their first visit and can never place anything;
     * they are tracked per position. */
    uint8_t *avail_known = (uint8_t *)calloc((size_t)n_nodes + 1, 1);
    uint8_t *avail_unknown_pos = (uint8_t *)calloc((size_t)n_x + 1, 1);
    uint64_t n_available = 0;
    int32_t count = 0;
    int ok = 0;
    if (!avail_known || !avail_unknown_pos) goto done;
    for (uint32_t i = 0; i < n_x; ++i) {
        uint32_t n = order[i];
        if (n < n_nodes) {
            if (!avail_known[n]) {
                avail_known[n] = 1;
                ++n_available;
            }
        } else {
            avail_unknown_pos[i] = 1;
            ++n_available;
        }
    }
    if (k == 0) { /* :46-48 */
        ok = 1;
        goto done;
    }
    while (n_available > 0) {                                   /* :49 */
        for (uint32_t i = 0; i < n_x; ++i) {                    /* :50 */
            uint32_t n = order[i];
            if (n >= n_nodes) {
                if (!avail_unknown_pos[i]) continue;            /* :51-53 */
                avail_unknown_pos[i] = 0;                       /* :59-62 `!ok` -> delete */
                --n_available;
                continue;
            }
            if (!avail_known[n]) continue;                      /* :51-53 */
            int64_t *rsv = rm_get_or_zero(reserved, n);         /* :55-57 */
            res_add(rsv, exe);                                  /* :58 */
            if (res_greater_than(rsv, &avail[3 * n])) {         /* :60 */
                avail_known[n] = 0;                             /* :62 */
                --n_available;
                res_sub(rsv, exe);                              /* :63 */
            } else {
                out[count++] = n;                               /* :65 */
                if (count == k) {                               /* :66-68 */
                    ok = 1;
                    goto done;
                }
            }
        }
    }
done:
    free(avail_known);
    free(avail_unknown_pos);
    return ok;
}

/* ---------------------------------------------------------------- capacity (LIB/capacity/capacity.go) */

static int64_t floor_div(int64_t a, int64_t b) { /* b > 0 */
    int64_t q = a / b, r = a % b;
    return (r != 0 && r < 0) ? q - 1 : q;
}

/* getCapacityAgainstSingleDimension, capacity.go:36-56 */
static int64_t capacity_single_dim(int64_t available, int64_t reserved, int64_t required) {
    if (reserved > available) return 0;          /* :37-40 */
    if (required == 0) return INT64_MAX;         /* :42-45 math.MaxInt */
    return floor_div(available - reserved, required); /* :48-55 QuoRound(..., RoundFloor) */
}

/* GetNodeCapacity, capacity.go:59-75 */
int64_t go_node_capacity(const int64_t avail[3], const int64_t reserved[3], const int64_t required[3]) {
    int64_t c = capacity_single_dim(avail[0], reserved[0], required[0]);
    int64_t m = capacity_single_dim(avail[1], reserved[1], required[1]);
    int64_t g = capacity_single_dim(avail[2], reserved[2], required[2]);
    int64_t r = c < m ? c : m;
    return r < g ? r : g;
}

/* ---------------------------------------------------------------- minimalFragmentation (next-tier packer) */

typedef struct {
    uint32_t node;
    int64_t cap;
} node_cap;

/* sort.Search: smallest i in [0,n) with pred true (pred monotone), else n. */
static uint32_t search_cap_ge(const node_cap *v, uint32_t n, int64_t target) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (v[mid].cap >= target) hi = mid; else lo = mid + 1;
    }
    return lo;
}

/* internalMinimalFragmentation, minimal_fragmentation.go:93-137.  Returns 1 and writes k nodes on success. */
static int internal_min_frag(int64_t k, const node_cap *caps_in, uint32_t n, uint32_t *out) {
    node_cap *caps = (node_cap *)malloc(((size_t)n + 1) * sizeof(node_cap));
    int64_t count = 0;
    int ok = 0;
    if (!caps) return 0;
    memcpy(caps, caps_in, (size_t)n * sizeof(node_cap)); /* :96-97 */
    while (n > 0) {                                      /* :101 */
        uint32_t pos = search_cap_ge(caps, n, k);        /* :103-105 */
        if (pos != n) {                                  /* :107-110 */
            for (int64_t i = 0; i < k; ++i) out[count++] = caps[pos].node;
            ok = 1;
            break;
        }
        int64_t max_cap = caps[n - 1].cap;                               /* :113 */
        uint32_t first_max = search_cap_ge(caps, n, max_cap);            /* :114-116 */
        uint32_t cur = first_max;                                        /* :119 */
        for (; k >= max_cap && cur < n; ++cur) {                         /* :120 */
            for (int64_t i = 0; i < max_cap; ++i) out[count++] = caps[cur].node; /* :122 */
            k -= max_cap;                                                /* :123 */
        }
        if (k == 0) { /* :126-128 */
            ok = 1;
            break;
        }
        memmove(&caps[first_max], &caps[cur], (size_t)(n - cur) * sizeof(node_cap)); /* :130 */
        n -= (cur - first_max);
    }
    free(caps);
    return ok;
}

/* stable insertion-merge sort by capacity ascending (sort.SliceStable, minimal_fragmentation.go:64-66) */
static void stable_sort_caps(node_cap *v, uint32_t n) {
    if (n < 2) return;
    node_cap *tmp = (node_cap *)malloc((size_t)n * sizeof(node_cap));
    if (!tmp) return;
    for (uint32_t w = 1; w < n; w *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * w) {
            uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint32_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) tmp[o++] = (v[j].cap < v[i].cap) ? v[j++] : v[i++];
            while (i < mid) tmp[o++] = v[i++];
            while (j < hi) tmp[o++] = v[j++];
        }
        memcpy(v, tmp, (size_t)n * sizeof(node_cap));
    }
    free(tmp);
}

/* minimalFragmentation, minimal_fragmentation.go:33-91 */
static int minimal_fragmentation(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t k,
                                 const uint32_t *order, uint32_t n_x, reserved_map *reserved, uint32_t *out) {
    static const int64_t zero[3] = {0, 0, 0};
    if (k == 0) return 1; /* :40-42 */
    node_cap *caps = (node_cap *)malloc(((size_t)n_x + 1) * sizeof(node_cap));
    uint32_t n = 0;
    int ok = 0;
    if (!caps) return 0;
    /* capacity.GetNodeCapacities (capacity.go:78-102) then FilterOutNodesWithoutCapacity (:105-113) */
    for (uint32_t i = 0; i < n_x; ++i) {
        uint32_t node = order[i];
        if (node >= n_nodes) continue; /* capacity.go:87 `if ... ok` */
        const int64_t *rsv = reserved->present[node] ? &reserved->r[3 * node] : zero;
        int64_t c = go_node_capacity(&avail[3 * node], rsv, exe);
        if (c > 0) {
            caps[n].node = node;
            caps[n].cap = c;
            ++n;
        }
    }
    if (n == 0) goto done;             /* :46-48 */
    stable_sort_caps(caps, n);         /* :64-66 */
    {
        int64_t max_cap = caps[n - 1].cap; /* :67 */
        if ((int64_t)k < max_cap) {        /* :68 */
            /* (executorCount + maxCapacity) / 2 in Go int arithmetic; with maxCapacity == math.MaxInt the sum
             * wraps in Go too — mirror with unsigned wrap then signed divide. */
            int64_t target = (int64_t)((uint64_t)k + (uint64_t)max_cap) / 2; /* :69 */
            uint32_t first = search_cap_ge(caps, n, target);                   /* :70-72 */
            if (internal_min_frag(k, caps, first, out)) {                      /* :75-77 */
                ok = 1;
                goto done;
            }
        }
    }
    ok = internal_min_frag(k, caps, n, out); /* :81 */
done:
    free(caps);
    return ok;
}

/* ---------------------------------------------------------------- SparkBinPack, literal */

static int run_packer(int algo, const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t k,
                      const uint32_t *order, uint32_t n_x, reserved_map *reserved, uint32_t *out) {
    switch (algo) {
    case GO_ALGO_TIGHTLY_PACK:
        return tightly_pack_executors(avail, n_nodes, exe, k, order, n_x, reserved, out);
    case GO_ALGO_DISTRIBUTE_EVENLY:
        return distribute_executors_evenly(avail, n_nodes, exe, k, order, n_x, reserved, out);
    case GO_ALGO_MINIMAL_FRAGMENTATION:
        return minimal_fragmentation(avail, n_nodes, exe, k, order, n_x, reserved, out);
    default:
        return 0;
    }
}

/* SparkBinPack, LIB/binpack/binpack.go:60-87 */
static int spark_binpack_rm(int algo, const int64_t *avail, uint32_t n_nodes, const go_app *app,
                            const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                            uint32_t n_x, uint32_t *driver_out, uint32_t *exec_out, reserved_map *reserved) {
    for (uint32_t i = 0; i < n_d; ++i) {                                          /* :67 */
        uint32_t d = driver_order[i];
        if (d >= n_nodes || res_greater_than(app->drv, &avail[3 * d])) continue;  /* :68-71 driver-fit check */
        rm_clear(reserved);                                                       /* :72 fresh map */
        int64_t *rsv = rm_get_or_zero(reserved, d);
        rsv[0] = app->drv[0];                                                     /* :73 */
        rsv[1] = app->drv[1];
        rsv[2] = app->drv[2];
        if (run_packer(algo, avail, n_nodes, app->exe, app->k, exec_order, n_x, reserved, exec_out)) { /* :74-76 */
            *driver_out = d;
            return 1; /* :77-83 */
        }
    }
    *driver_out = GO_NO_NODE;
    return 0; /* :86 EmptyPackingResult */
}

int go_spark_binpack(int algo, const int64_t *avail, uint32_t n_nodes, const go_app *app,
                     const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order, uint32_t n_x,
                     uint32_t *driver_out, uint32_t *exec_out) {
    reserved_map rm;
    int ok = 0;
    if (rm_init(&rm, n_nodes))
        ok = spark_binpack_rm(algo, avail, n_nodes, app, driver_order, n_d, exec_order, n_x, driver_out,
                              exec_out, &rm);
    else
        *driver_out = GO_NO_NODE;
    rm_free(&rm);
    return ok;
}

/* ---------------------------------------------------------------- closed-form restatement */

/* cap(n, base) clamped to k: the number of consecutive successful add-then-compare steps on one node
 * (SURVEY.md section 8).  avail may be negative; base/exe are >= 0. */
static int64_t cap_clamped(const int64_t avail[3], const int64_t base[3], const int64_t exe[3], int64_t k) {
    int64_t c = k;
    for (int j = 0; j < 3; ++j) {
        int64_t a = avail[j] - base[j];
        if (a < 0) return 0;
        if (exe[j] == 0) continue;
        int64_t q = a / exe[j];
        if (q < c) c = q;
    }
    return c;
}

static int spark_binpack_closed(int algo, const int64_t *avail, uint32_t n_nodes, const go_app *app,
                                const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                                uint32_t n_x, uint32_t *driver_out, uint32_t *exec_out) {
    static const int64_t zero[3] = {0, 0, 0};
    if (algo == GO_ALGO_DISTRIBUTE_EVENLY) {
        /* A known name listed twice in the executor order is ONE key of availableNodes but is visited twice per pass by the
         * loop over the slice (distribute_evenly.go:50): the capacity form below (a duplicate has capacity 0) only matches
         * the literal code for tightly-pack.  The orders of the reference derive from map keys and never repeat a name
         * (nodesorting.go:153-159), and the device ABI rejects duplicates; if a caller builds one anyway, the literal loops
         * answer. */
        uint8_t *seen = (uint8_t *)calloc((size_t)n_nodes + 1, 1);
        int dup = 0;
        if (seen)
            for (uint32_t i = 0; i < n_x && !dup; ++i)
                if (exec_order[i] < n_nodes) {
                    dup = seen[exec_order[i]];
                    seen[exec_order[i]] = 1;
                }
        free(seen);
        if (dup)
            return go_spark_binpack(algo, avail, n_nodes, app, driver_order, n_d, exec_order, n_x, driver_out, exec_out);
    }
    const int64_t k = app->k;
    int64_t *c0 = (int64_t *)malloc(((size_t)n_x + 1) * sizeof(int64_t));
    /* pos_in_x[node] = position in exec order (first occurrence), or -1 */
    int64_t *pos_in_x = (int64_t *)malloc(((size_t)n_nodes + 1) * sizeof(int64_t));
    int ok = 0;
    *driver_out = GO_NO_NODE;
    if (!c0 || !pos_in_x) goto done;
    for (uint32_t n = 0; n < n_nodes; ++n) pos_in_x[n] = -1;
    int64_t s = 0;
    for (uint32_t i = 0; i < n_x; ++i) {
        uint32_t n = exec_order[i];
        c0[i] = 0;
        if (n >= n_nodes) continue;
        if (pos_in_x[n] >= 0) continue; /* duplicate name: its reserved entry is already saturated (tightly) */
        pos_in_x[n] = i;
        c0[i] = cap_clamped(&avail[3 * n], zero, app->exe, k);
        s += c0[i];
    }
    /* O(N) driver choice */
    uint32_t d = GO_NO_NODE;
    int64_t cd = 0;
    for (uint32_t i = 0; i < n_d; ++i) {
        uint32_t cand = driver_order[i];
        if (cand >= n_nodes || res_greater_than(app->drv, &avail[3 * cand])) continue;
        int64_t total = s, cdd = 0;
        if (pos_in_x[cand] >= 0) {
            cdd = cap_clamped(&avail[3 * cand], app->drv, app->exe, k);
            total = s - c0[pos_in_x[cand]] + cdd;
        }
        if (total >= k) {
            d = cand;
            cd = cdd;
            break;
        }
    }
    if (d == GO_NO_NODE) goto done;
    if (pos_in_x[d] >= 0) c0[pos_in_x[d]] = cd;
    *driver_out = d;
    ok = 1;
    if (k == 0) goto done;
    if (algo == GO_ALGO_TIGHTLY_PACK) {
        int64_t taken = 0;
        for (uint32_t i = 0; i < n_x && taken < k; ++i) {
            int64_t t = c0[i] < k - taken ? c0[i] : k - taken;
            for (int64_t j = 0; j < t; ++j) exec_out[taken + j] = exec_order[i];
            taken += t;
        }
    } else { /* distribute evenly: passes r = 1, 2, ...: every node with cap >= r, in order */
        int64_t taken = 0;
        for (int64_t r = 1; taken < k; ++r)
            for (uint32_t i = 0; i < n_x && taken < k; ++i)
                if (c0[i] >= r) exec_out[taken++] = exec_order[i];
    }
done:
    free(c0);
    free(pos_in_x);
    return ok;
}

int go_spark_binpack_closed_form(int algo, const int64_t *avail, uint32_t n_nodes, const go_app *app,
                                 const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                                 uint32_t n_x, uint32_t *driver_out, uint32_t *exec_out) {
    if (algo == GO_ALGO_MINIMAL_FRAGMENTATION) /* no closed form: same literal code */
        return go_spark_binpack(algo, avail, n_nodes, app, driver_order, n_d, exec_order, n_x, driver_out,
                                exec_out);
    return spark_binpack_closed(algo, avail, n_nodes, app, driver_order, n_d, exec_order, n_x, driver_out,
                                exec_out);
}

/* ---------------------------------------------------------------- efficiencies (LIB/binpack/efficiency.go) */

/* Quantity.Value(): rounded to the nearest integer AWAY from zero (K8S apimachinery quantity.go:731-734,
 * math.go:169-199).  unit = 1000 for cpu (canonical milli), 1 for memory bytes and gpu count. */
static int64_t value_round_away(int64_t v, int64_t unit) {
    if (unit == 1) return v;
    int64_t q = v / unit, r = v % unit;
    if (r > 0) return q + 1;
    if (r < 0) return q - 1;
    return q;
}
static int64_t normalize_resource(int64_t v) { return v == 0 ? 1 : v; } /* efficiency.go:105-110 */

static const int64_t k_value_unit[3] = {1000, 1, 1};

/* computePackingEfficiency, efficiency.go:79-103, for one node given its entry of the `reserved` map (zeros if absent).
 * Returns whether the node has gpus (SchedulableResources.NvidiaGPU.Value() != 0). */
static int node_efficiency(const int64_t avail[3], const int64_t sched[3], const int64_t rsv[3], double e[3]) {
    for (int j = 0; j < 3; ++j) {
        int64_t used = sched[j] - avail[j] + rsv[j];                       /* :84-88 */
        int64_t sv = value_round_away(sched[j], k_value_unit[j]);
        e[j] = (double)value_round_away(used, k_value_unit[j]) / (double)normalize_resource(sv);
    }
    int has_gpu = value_round_away(sched[2], 1) != 0;
    if (!has_gpu) e[2] = 0.0; /* :91-94 */
    return has_gpu;
}

/* Running state of ComputeAvgPackingEfficiency (efficiency.go:114-156): sums in the order entries are pushed. */
typedef struct {
    double cpu_sum, mem_sum, gpu_sum, max_sum;
    uint64_t with_gpu, len;
} avg_acc;
static void avg_push(avg_acc *a, const double e[3], int has_gpu) {
    a->cpu_sum += e[0];                      /* :128 */
    a->mem_sum += e[1];                      /* :129 */
    if (has_gpu) {                           /* :131-134 */
        a->gpu_sum += e[2];
        ++a->with_gpu;
    }
    double m = e[0] > e[1] ? e[0] : e[1];    /* :136 math.Max (no NaNs: every divisor is >= 1 in magnitude) */
    a->max_sum += e[2] > m ? e[2] : m;
    ++a->len;
}
static void avg_finish(const avg_acc *a, double out[4]) {
    if (a->len == 0) { /* WorstAvgPackingEfficiency, :119-121 */
        out[0] = out[1] = out[2] = out[3] = 0.0;
        return;
    }
    double len = (double)a->len;             /* :139 */
    out[0] = a->cpu_sum / len;
    out[1] = a->mem_sum / len;
    out[2] = a->with_gpu == 0 ? 1.0 : a->gpu_sum / (double)a->with_gpu; /* :141-145 */
    out[3] = a->max_sum / len;
}

void go_packing_efficiency(const int64_t *avail, const int64_t *sched, uint32_t n_nodes, const go_app *app,
                           uint32_t driver_node, const uint32_t *exec_nodes, uint32_t n_exec, double *eff_out,
                           double avg_out[4]) {
    int64_t *reserved = (int64_t *)calloc((size_t)n_nodes * 3 + 3, sizeof(int64_t));
    avg_acc acc;
    memset(&acc, 0, sizeof acc);
    if (!reserved) return;
    /* `reserved` as left by SparkBinPack: driver + one executorResources per placed executor */
    if (driver_node < n_nodes) res_add(&reserved[3 * driver_node], app->drv);
    for (uint32_t i = 0; i < n_exec; ++i)
        if (exec_nodes[i] < n_nodes) res_add(&reserved[3 * exec_nodes[i]], app->exe);
    for (uint32_t n = 0; n < n_nodes; ++n) {
        double e[3];
        int has_gpu = node_efficiency(&avail[3 * n], &sched[3 * n], &reserved[3 * n], e);
        if (eff_out) {
            eff_out[3 * n] = e[0];
            eff_out[3 * n + 1] = e[1];
            eff_out[3 * n + 2] = e[2];
        }
        avg_push(&acc, e, has_gpu);
    }
    avg_finish(&acc, avg_out);
    free(reserved);
}

/* chooseBestResult's average (single_az.go:83-89): ComputeAvgPackingEfficiency over nodeNames = [driver] ++ executors,
 * duplicates counted, summed in slice order, read from the per-node efficiencies of the given `reserved` map. */
static void avg_efficiency_of_list(const int64_t *avail, const int64_t *sched, uint32_t n_nodes,
                                   const reserved_map *reserved, uint32_t driver_node, const uint32_t *exec_nodes,
                                   uint32_t n_exec, double avg_out[4]) {
    static const int64_t zero[3] = {0, 0, 0};
    avg_acc acc;
    memset(&acc, 0, sizeof acc);
    for (uint32_t i = 0; i <= n_exec; ++i) {
        uint32_t n = i == 0 ? driver_node : exec_nodes[i - 1];
        double e[3];
        if (n >= n_nodes) continue; /* cannot happen for a feasible result: every placed node is a metadata key */
        const int64_t *rsv = reserved->present[n] ? &reserved->r[3 * n] : zero;
        int has_gpu = node_efficiency(&avail[3 * n], &sched[3 * n], rsv, e);
        avg_push(&acc, e, has_gpu);
    }
    avg_finish(&acc, avg_out);
}

void go_avg_packing_efficiency_list(const int64_t *avail, const int64_t *sched, uint32_t n_nodes, const go_app *app,
                                    uint32_t driver_node, const uint32_t *exec_nodes, uint32_t n_exec,
                                    int reserved_includes_executors, double avg_out[4]) {
    reserved_map rm;
    avg_out[0] = avg_out[1] = avg_out[2] = avg_out[3] = 0.0;
    if (!rm_init(&rm, n_nodes)) {
        rm_free(&rm);
        return;
    }
    if (driver_node < n_nodes) res_add(rm_get_or_zero(&rm, driver_node), app->drv);
    if (reserved_includes_executors)
        for (uint32_t i = 0; i < n_exec; ++i)
            if (exec_nodes[i] < n_nodes) res_add(rm_get_or_zero(&rm, exec_nodes[i]), app->exe);
    avg_efficiency_of_list(avail, sched, n_nodes, &rm, driver_node, exec_nodes, n_exec, avg_out);
    rm_free(&rm);
}

/* ---------------------------------------------------------------- single-AZ wrappers (LIB/binpack/single_az.go) */

static int is_single_az(int algo) {
    return algo == GO_ALGO_SINGLE_AZ_TIGHTLY_PACK || algo == GO_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION ||
           algo == GO_ALGO_AZ_AWARE_TIGHTLY_PACK;
}
static int inner_packer(int algo) {
    return algo == GO_ALGO_SINGLE_AZ_MINIMAL_FRAGMENTATION ? GO_ALGO_MINIMAL_FRAGMENTATION : GO_ALGO_TIGHTLY_PACK;
}

/* groupNodesByZone, single_az.go:57-72: zones in order of first appearance, names per zone in order; names that
 * are not metadata keys are dropped.  Returns the number of zones; zones_in_order[] receives them; the caller
 * filters per zone with zone_filter(). */
static uint32_t zones_in_order(const uint32_t *order, uint32_t n, const uint32_t *zone, uint32_t n_nodes,
                               uint32_t *zones_out /* room for n */) {
    uint32_t nz = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (order[i] >= n_nodes) continue;
        uint32_t z = zone ? zone[order[i]] : 0, seen = 0;
        for (uint32_t q = 0; q < nz && !seen; ++q) seen = zones_out[q] == z;
        if (!seen) zones_out[nz++] = z;
    }
    return nz;
}
static uint32_t zone_filter(const uint32_t *order, uint32_t n, const uint32_t *zone, uint32_t n_nodes, uint32_t z,
                            uint32_t *out) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (order[i] < n_nodes && (zone ? zone[order[i]] : 0) == z) out[m++] = order[i];
    return m;
}

/* getSingleAZSparkBinFunction (single_az.go:23-55) + chooseBestResult (:75-97).  The per-zone packs and their
 * efficiencies follow the literal path (reserved map as the packer left it) or, with closed_form, the array path
 * with `reserved` rebuilt from the placement (tightly-pack only). */
static int single_az_binpack(int algo, int closed_form, const int64_t *avail, const int64_t *sched,
                             const uint32_t *zone, uint32_t n_nodes, const go_app *app, const uint32_t *driver_order,
                             uint32_t n_d, const uint32_t *exec_order, uint32_t n_x, uint32_t *driver_out,
                             uint32_t *exec_out, reserved_map *rm, double avg_out[4]) {
    const int inner = inner_packer(algo);
    uint32_t *dz = (uint32_t *)malloc(((size_t)n_d + 1) * sizeof(uint32_t));
    uint32_t *xz = (uint32_t *)malloc(((size_t)n_x + 1) * sizeof(uint32_t));
    uint32_t *d_f = (uint32_t *)malloc(((size_t)n_d + 1) * sizeof(uint32_t));
    uint32_t *x_f = (uint32_t *)malloc(((size_t)n_x + 1) * sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)malloc(((size_t)(app->k > 0 ? app->k : 0) + 1) * sizeof(uint32_t));
    int best_ok = 0;
    double best_avg[4] = {0.0, 0.0, 0.0, 0.0}; /* WorstAvgPackingEfficiency, single_az.go:80 */
    *driver_out = GO_NO_NODE;
    if (avg_out) avg_out[0] = avg_out[1] = avg_out[2] = avg_out[3] = 0.0;
    if (!dz || !xz || !d_f || !x_f || !tmp) goto done;
    {
        const uint32_t n_dz = zones_in_order(driver_order, n_d, zone, n_nodes, dz);   /* :31 */
        const uint32_t n_xz = zones_in_order(exec_order, n_x, zone, n_nodes, xz);     /* :32 */
        for (uint32_t q = 0; q < n_dz; ++q) {                                         /* :36 */
            uint32_t z = dz[q], has_x = 0;
            for (uint32_t r = 0; r < n_xz && !has_x; ++r) has_x = xz[r] == z;
            if (!has_x) continue;                                                     /* :38-41 */
            uint32_t nd = zone_filter(driver_order, n_d, zone, n_nodes, z, d_f);
            uint32_t nx = zone_filter(exec_order, n_x, zone, n_nodes, z, x_f);
            uint32_t d = GO_NO_NODE;
            int ok;
            if (closed_form && inner != GO_ALGO_MINIMAL_FRAGMENTATION) {
                ok = spark_binpack_closed(inner, avail, n_nodes, app, d_f, nd, x_f, nx, &d, tmp);
                if (ok) { /* rebuild `reserved`: driver + one executor request per placement */
                    rm_clear(rm);
                    res_add(rm_get_or_zero(rm, d), app->drv);
                    for (int32_t i = 0; i < app->k; ++i) res_add(rm_get_or_zero(rm, tmp[i]), app->exe);
                }
            } else {
                ok = spark_binpack_rm(inner, avail, n_nodes, app, d_f, nd, x_f, nx, &d, tmp, rm); /* :42 */
            }
            if (!ok) continue;                                                        /* :44-46 */
            double avg[4];
            avg_efficiency_of_list(avail, sched, n_nodes, rm, d, tmp, (uint32_t)app->k, avg);
            if (best_avg[3] < avg[3]) { /* LessThan, efficiency.go:37-39: strictly higher Max replaces */
                best_ok = 1;
                *driver_out = d;
                memcpy(exec_out, tmp, (size_t)app->k * sizeof(uint32_t));
                memcpy(best_avg, avg, sizeof avg);
            }
        }
    }
    if (best_ok && avg_out) memcpy(avg_out, best_avg, sizeof best_avg);
done:
    free(dz);
    free(xz);
    free(d_f);
    free(x_f);
    free(tmp);
    return best_ok; /* no feasible zone, or no zone with Max > 0: EmptyPackingResult (:49-51, :78) */
}

/* ---------------------------------------------------------------- batch drivers */

typedef struct {
    const int64_t *sched;  /* n_nodes x 3 SchedulableResources; needed by the single-AZ packers (efficiencies) */
    const uint32_t *zone;  /* n_nodes zone ids (NodeSchedulingMetadata.ZoneLabel); NULL = one zone */
} cluster_aux;

/* One decision by packer name (internal/binpacker/binpack.go:43-49).  avg_out (nullable, 4 doubles) receives the
 * AvgPackingEfficiency over [driver] ++ executors of the returned result (zeros when infeasible). */
static int binpack_any(int algo, int closed_form, const int64_t *avail, const cluster_aux *aux, uint32_t n_nodes,
                       const go_app *app, const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                       uint32_t n_x, uint32_t *driver_out, uint32_t *exec_out, reserved_map *rm, double *avg_out) {
    int ok;
    if (is_single_az(algo)) {
        ok = aux && aux->sched &&
             single_az_binpack(algo, closed_form, avail, aux->sched, aux->zone, n_nodes, app, driver_order, n_d,
                               exec_order, n_x, driver_out, exec_out, rm, avg_out);
        if (ok || algo != GO_ALGO_AZ_AWARE_TIGHTLY_PACK) return ok;
        algo = GO_ALGO_TIGHTLY_PACK; /* az_aware_pack_tightly.go:33-37: fall back to plain TightlyPack */
    }
    if (closed_form && algo != GO_ALGO_MINIMAL_FRAGMENTATION) {
        ok = spark_binpack_closed(algo, avail, n_nodes, app, driver_order, n_d, exec_order, n_x, driver_out, exec_out);
        if (ok && avg_out && aux && aux->sched) {
            rm_clear(rm);
            res_add(rm_get_or_zero(rm, *driver_out), app->drv);
            for (int32_t i = 0; i < app->k; ++i) res_add(rm_get_or_zero(rm, exec_out[i]), app->exe);
        }
    } else {
        ok = spark_binpack_rm(algo, avail, n_nodes, app, driver_order, n_d, exec_order, n_x, driver_out, exec_out, rm);
    }
    if (avg_out) {
        avg_out[0] = avg_out[1] = avg_out[2] = avg_out[3] = 0.0;
        if (ok && aux && aux->sched)
            avg_efficiency_of_list(avail, aux->sched, n_nodes, rm, *driver_out, exec_out, (uint32_t)app->k, avg_out);
    }
    return ok;
}

void go_fit_independent_ex(int algo, int closed_form, const int64_t *avail, const int64_t *sched,
                           const uint32_t *zone, uint32_t n_nodes, const go_app *apps, uint32_t n_apps,
                           const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order, uint32_t n_x,
                           go_result *results, const uint64_t *exec_off, uint32_t *exec_out, double *avg_out) {
    reserved_map rm;
    cluster_aux aux = {sched, zone};
    int have_rm = rm_init(&rm, n_nodes);
    for (uint32_t a = 0; a < n_apps; ++a) {
        uint32_t d = GO_NO_NODE;
        int ok = have_rm && binpack_any(algo, closed_form, avail, &aux, n_nodes, &apps[a], driver_order, n_d,
                                        exec_order, n_x, &d, exec_out + exec_off[a], &rm,
                                        avg_out ? avg_out + 4 * (size_t)a : NULL);
        results[a].has_capacity = ok;
        results[a].driver_node = ok ? d : GO_NO_NODE;
        results[a].exec_len = ok ? (uint32_t)apps[a].k : 0;
        results[a].evaluated = 1;
    }
    rm_free(&rm);
}

void go_fit_independent(int algo, int closed_form, const int64_t *avail, uint32_t n_nodes, const go_app *apps,
                        uint32_t n_apps, const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                        uint32_t n_x, go_result *results, const uint64_t *exec_off, uint32_t *exec_out) {
    go_fit_independent_ex(algo, closed_form, avail, NULL, NULL, n_nodes, apps, n_apps, driver_order, n_d, exec_order,
                          n_x, results, exec_off, exec_out, NULL);
}

/* sparkResourceUsage + SubtractUsageIfExists (internal/extender/sparkpods.go:139-146, LIB/resources/resources.go:129-135):
 * res := map{}; res[driverNode] = drv; for n in execNodes { res[n] = exe }  — a MAP, so multiplicity is lost and a
 * driver entry is overwritten by an executor on the same node; then avail[n] -= res[n] for n in metadata. */
static void subtract_usage(int64_t *avail, uint32_t n_nodes, const go_app *app, uint32_t driver_node,
                           const uint32_t *exec_nodes, uint32_t n_exec, uint8_t *mark /* n_nodes zeros */) {
    int driver_overwritten = 0;
    for (uint32_t i = 0; i < n_exec; ++i) {
        uint32_t n = exec_nodes[i];
        if (n == driver_node) driver_overwritten = 1;
        if (n >= n_nodes || mark[n]) continue;
        mark[n] = 1;
        res_sub(&avail[3 * n], app->exe);
    }
    for (uint32_t i = 0; i < n_exec; ++i)
        if (exec_nodes[i] < n_nodes) mark[exec_nodes[i]] = 0;
    if (!driver_overwritten && driver_node < n_nodes) res_sub(&avail[3 * driver_node], app->drv);
}

static int32_t fifo_chain_impl(int algo, int closed_form, int64_t *avail, const int64_t *sched, const uint32_t *zone,
                               uint32_t n_nodes, const go_app *apps, uint32_t n_apps, const uint32_t *driver_order,
                               uint32_t n_d, const uint32_t *exec_order, uint32_t n_x, go_result *results,
                               const uint64_t *exec_off, uint32_t *exec_out, double *eff_scratch) {
    reserved_map rm;
    cluster_aux aux = {sched, zone};
    int have_rm = rm_init(&rm, n_nodes);
    uint8_t *mark = (uint8_t *)calloc((size_t)n_nodes + 1, 1);
    int32_t failed_at = -1;
    for (uint32_t a = 0; a < n_apps; ++a) {
        results[a].has_capacity = 0;
        results[a].driver_node = GO_NO_NODE;
        results[a].exec_len = 0;
        results[a].evaluated = 0;
    }
    if (!have_rm || !mark) goto done;
    for (uint32_t a = 0; a < n_apps; ++a) { /* resource.go:229 over earlier drivers, then :321 for the last */
        uint32_t d = GO_NO_NODE;
        int ok = binpack_any(algo, closed_form, avail, &aux, n_nodes, &apps[a], driver_order, n_d, exec_order, n_x, &d,
                             exec_out + exec_off[a], &rm, NULL);
        results[a].evaluated = 1;
        results[a].has_capacity = ok;
        results[a].driver_node = ok ? d : GO_NO_NODE;
        results[a].exec_len = ok ? (uint32_t)apps[a].k : 0;
        if (ok && eff_scratch && sched) { /* binpack.go:77: PackingEfficiencies of EVERY successful pack, replayed ones included */
            double avg[4];
            go_packing_efficiency(avail, sched, n_nodes, &apps[a], d, exec_out + exec_off[a], (uint32_t)apps[a].k, eff_scratch, avg);
        }
        if (a + 1 == n_apps) break; /* the driver being filtered: no subtraction afterwards */
        if (!ok) {
            if (apps[a].flags & GO_APP_SKIPPABLE) continue; /* resource.go:244-248 */
            failed_at = (int32_t)a;                          /* :249-251 -> "failure-earlier-driver" */
            break;
        }
        subtract_usage(avail, n_nodes, &apps[a], d, exec_out + exec_off[a], (uint32_t)apps[a].k, mark); /* :255-259 */
    }
done:
    free(mark);
    rm_free(&rm);
    return failed_at;
}

int32_t go_fit_fifo_chain_ex(int algo, int closed_form, int64_t *avail, const int64_t *sched, const uint32_t *zone,
                             uint32_t n_nodes, const go_app *apps, uint32_t n_apps, const uint32_t *driver_order,
                             uint32_t n_d, const uint32_t *exec_order, uint32_t n_x, go_result *results,
                             const uint64_t *exec_off, uint32_t *exec_out) {
    return fifo_chain_impl(algo, closed_form, avail, sched, zone, n_nodes, apps, n_apps, driver_order, n_d, exec_order, n_x,
                           results, exec_off, exec_out, NULL);
}

/* The chain as the reference's SparkBinPack really runs it: every successful pack also builds the per-node
 * PackingEfficiencies map over ALL nodes (binpack.go:77 -> efficiency.go:66-103), which fitEarlierDrivers then discards
 * (SURVEY.md section 8a row 9).  Same decisions; used by bench.py as the "reference-shaped" CPU baseline. */
int32_t go_fit_fifo_chain_with_efficiencies(int algo, int64_t *avail, const int64_t *sched, const uint32_t *zone,
                                            uint32_t n_nodes, const go_app *apps, uint32_t n_apps,
                                            const uint32_t *driver_order, uint32_t n_d, const uint32_t *exec_order,
                                            uint32_t n_x, go_result *results, const uint64_t *exec_off, uint32_t *exec_out) {
    double *eff = (double *)malloc(((size_t)n_nodes * 3 + 3) * sizeof(double));
    int32_t r = fifo_chain_impl(algo, 0, avail, sched, zone, n_nodes, apps, n_apps, driver_order, n_d, exec_order, n_x,
                                results, exec_off, exec_out, eff);
    free(eff);
    return r;
}

int32_t go_fit_fifo_chain(int algo, int closed_form, int64_t *avail, uint32_t n_nodes, const go_app *apps,
                          uint32_t n_apps, const uint32_t *driver_order, uint32_t n_d,
                          const uint32_t *exec_order, uint32_t n_x, go_result *results,
                          const uint64_t *exec_off, uint32_t *exec_out) {
    return go_fit_fifo_chain_ex(algo, closed_form, avail, NULL, NULL, n_nodes, apps, n_apps, driver_order, n_d,
                                exec_order, n_x, results, exec_off, exec_out);
}

/* ---------------------------------------------------------------- executor first-fit */

/* rescheduleExecutor first-fit, internal/extender/resource.go:658-662 */
uint32_t go_executor_first_fit(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3],
                               const uint32_t *exec_order, uint32_t n_x) {
    for (uint32_t i = 0; i < n_x; ++i) {
        uint32_t n = exec_order[i];
        if (n >= n_nodes) continue; /* order is derived from the metadata keys; defensive */
        if (!res_greater_than(exe, &avail[3 * n])) return n;
    }
    return GO_NO_NODE;
}

/* rescheduleExecutorWithMinimalFragmentation, internal/extender/resource.go:675-703: capacities through
 * capacity.GetNodeCapacities(executorNodeNames, metadata, overhead, executorResources) (capacity.go:78-102), then the
 * `best` switch (:686-699).  reserved: n_nodes x 3 or NULL; hosts: one byte per node (non-zero = the node already hosts
 * executors of this application) or NULL. */
uint32_t go_executor_min_frag(const int64_t *avail, uint32_t n_nodes, const int64_t *reserved, const int64_t exe[3],
                              const uint32_t *exec_order, uint32_t n_x, const uint8_t *hosts) {
    static const int64_t zero[3] = {0, 0, 0};
    uint32_t best = GO_NO_NODE;
    int64_t best_cap = 0;
    for (uint32_t i = 0; i < n_x; ++i) {
        uint32_t n = exec_order[i];
        if (n >= n_nodes) continue;                                                  /* capacity.go:87 */
        int64_t cap = go_node_capacity(&avail[3 * n], reserved ? &reserved[3 * n] : zero, exe);
        if (cap < 1) continue;                                                       /* :685 */
        int h = hosts && hosts[n], bh = best != GO_NO_NODE && hosts && hosts[best];
        if (best == GO_NO_NODE || (h && !bh) || (h == bh && cap < best_cap)) {        /* :687-698 */
            best = n;
            best_cap = cap;
        }
    }
    return best;
}

/* The first-fit loop of rescheduleExecutor (resource.go:658-662) against availableResources = the snapshot minus
 * `reserved` (the overhead counted a second time by `usage.Add(overhead)`, :640-643). */
uint32_t go_executor_first_fit_reserved(const int64_t *avail, uint32_t n_nodes, const int64_t *reserved,
                                        const int64_t exe[3], const uint32_t *exec_order, uint32_t n_x) {
    for (uint32_t i = 0; i < n_x; ++i) {
        uint32_t n = exec_order[i];
        if (n >= n_nodes) continue;
        int64_t a[3];
        for (int j = 0; j < 3; ++j) a[j] = avail[3 * n + j] - (reserved ? reserved[3 * n + j] : 0);
        if (!res_greater_than(exe, a)) return n;
    }
    return GO_NO_NODE;
}

/* ---------------------------------------------------------------- failover reconcile: findNodes */

/* findNodes, internal/extender/failover.go:412-436, literal: tightly-pack with PARTIAL results and no driver.  Every
 * visited node gets an entry in `reserved` (:419-421); the add that fails the comparison is NOT taken back before the
 * `break` (:424-428), so a node that was filled up ends with (placed + 1) x executorResources in the map, while the node on
 * which the count is reached returns at once (:430-432) and keeps exactly what was placed.  adds_out[n] = number of
 * `reserved[n].Add(executorResources)` calls (0 = no entry).  availableResources[n.Name] exists for every ordered node
 * (both derive from schedulableNodes, failover.go:286-322); an index >= n_nodes is skipped defensively.
 * executor_count <= 0 never happens (guarded at :367); it is treated as "nothing to do". */
uint32_t go_find_nodes(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t executor_count,
                       const uint32_t *ordered_nodes, uint32_t n_o, uint32_t *exec_out, uint32_t *adds_out) {
    uint32_t placed = 0;
    if (adds_out) memset(adds_out, 0, (size_t)n_nodes * sizeof(uint32_t));
    if (executor_count <= 0) return 0;
    for (uint32_t i = 0; i < n_o; ++i) {
        uint32_t n = ordered_nodes[i];
        if (n >= n_nodes) continue;
        int64_t reserved[3] = {0, 0, 0}; /* :419-421 (orderedNodes holds each node once) */
        uint32_t adds = 0;
        for (;;) {
            res_add(reserved, exe); /* :424 */
            ++adds;
            if (res_greater_than(reserved, &avail[3 * n])) break; /* :425-427 */
            exec_out[placed++] = n;                               /* :428 */
            if (placed == (uint32_t)executor_count) {             /* :429-431 */
                if (adds_out) adds_out[n] += adds;
                return placed;
            }
        }
        if (adds_out) adds_out[n] += adds;
    }
    return placed;
}

/* The same through capacities (cross-check): cap(n) = min over dims floor(avail / exe), 0 when any avail < 0. */
uint32_t go_find_nodes_closed_form(const int64_t *avail, uint32_t n_nodes, const int64_t exe[3], int32_t executor_count,
                                   const uint32_t *ordered_nodes, uint32_t n_o, uint32_t *exec_out, uint32_t *adds_out) {
    static const int64_t zero[3] = {0, 0, 0};
    uint32_t placed = 0;
    if (adds_out) memset(adds_out, 0, (size_t)n_nodes * sizeof(uint32_t));
    if (executor_count <= 0) return 0;
    for (uint32_t i = 0; i < n_o; ++i) {
        uint32_t n = ordered_nodes[i];
        if (n >= n_nodes) continue;
        int64_t left = (int64_t)executor_count - placed;
        int64_t cap = cap_clamped(&avail[3 * n], zero, exe, left);
        for (int64_t t = 0; t < cap; ++t) exec_out[placed++] = n;
        if (cap == left) {
            if (adds_out) adds_out[n] += (uint32_t)cap;
            return placed;
        }
        if (adds_out) adds_out[n] += (uint32_t)cap + 1;
    }
    return placed;
}

/* The reconcile loop over stale applications of ONE instance group (failover.go:132-160): each request runs findNodes
 * against the current availableResources, then `r.availableResources[instanceGroup].Sub(reservedResources)` (:159)
 * subtracts the map findNodes returned — over-adds included.  avail is mutated; exec_out is the concatenation (request q
 * at exec_off[q], room for k[q]); adds_out (nullable) n_req x n_nodes. */
void go_find_nodes_chain(int closed_form, int64_t *avail, uint32_t n_nodes, const int64_t *exe, const int32_t *k,
                         uint32_t n_req, const uint32_t *ordered_nodes, uint32_t n_o, uint32_t *placed_out,
                         const uint64_t *exec_off, uint32_t *exec_out, uint32_t *adds_out) {
    uint32_t *adds = (uint32_t *)malloc(((size_t)n_nodes + 1) * sizeof(uint32_t));
    if (!adds) return;
    for (uint32_t q = 0; q < n_req; ++q) {
        const int64_t *e = &exe[3 * (size_t)q];
        placed_out[q] = closed_form
                            ? go_find_nodes_closed_form(avail, n_nodes, e, k[q], ordered_nodes, n_o, exec_out + exec_off[q], adds)
                            : go_find_nodes(avail, n_nodes, e, k[q], ordered_nodes, n_o, exec_out + exec_off[q], adds);
        for (uint32_t n = 0; n < n_nodes; ++n) {
            if (!adds[n]) continue;
            for (int j = 0; j < 3; ++j) avail[3 * (size_t)n + j] -= (int64_t)adds[n] * e[j]; /* NodeGroupResources.Sub */
        }
        if (adds_out) memcpy(adds_out + (size_t)q * n_nodes, adds, (size_t)n_nodes * sizeof(uint32_t));
    }
    free(adds);
}
