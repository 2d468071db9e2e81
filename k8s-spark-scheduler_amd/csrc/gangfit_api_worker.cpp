// gangfit_api_worker.cpp — the resident worker of the independent batch (gf_worker_*; device side: gangfit_worker.inc).
#include "gangfit_ctx.h"
#include <cstdio>

using namespace gfapi;

namespace gfapi {

constexpr uint32_t kRing = gangfit::kWorkerRing;

inline uint64_t host_load(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void host_store(unsigned long long* p, uint64_t v) { __atomic_store_n(p, (unsigned long long)v, __ATOMIC_RELEASE); }

int worker_alloc(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (w.allocated) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    void* hp = nullptr;
    GF_HIP(ctx, hipHostMalloc(&hp, sizeof(gangfit::WorkerHostCtl), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(hp, 0, sizeof(gangfit::WorkerHostCtl));
    void* hd = nullptr;
    if (hipHostGetDevicePointer(&hd, hp, 0) != hipSuccess) {
        (void)hipHostFree(hp);
        return fail(ctx, GF_ERR_HIP, "the worker's control block cannot be mapped to the device");
    }
    void* dp = nullptr;
    // (ordinary device memory: the relaxed agent-scope loads of the pollers are served by their XCD's L2 — 32 workgroups
    //  probing one line cost one miss per update and XCD; in fine-grained memory every probe of every workgroup went to the
    //  one memory channel that holds the line)
    if (hipMalloc(&dp, sizeof(gangfit::WorkerDevCtl)) != hipSuccess ||
        hipMemset(dp, 0, sizeof(gangfit::WorkerDevCtl)) != hipSuccess) {
        if (dp) (void)hipFree(dp);
        (void)hipHostFree(hp);
        return fail(ctx, GF_ERR_HIP, "the worker's device control block cannot be allocated");
    }
    // its own non-blocking stream.  (A stream with a CU mask — to keep compute units free for FIFO chains — was measured first:
    // its first window cost 10 ms and, depending on the context, every ticket 3.8 instead of 2.4 us.  The free CUs come from the
    // worker's shape instead: workgroups of sixteen wavefronts that fill a CU's registers, fewer of them than the device has CUs.)
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipFree(dp);
        (void)hipHostFree(hp);
        return fail(ctx, GF_ERR_HIP, "the worker's stream cannot be created");
    }
    w.h = static_cast<gangfit::WorkerHostCtl*>(hp);
    w.h_dev = static_cast<gangfit::WorkerHostCtl*>(hd);
    w.d = static_cast<gangfit::WorkerDevCtl*>(dp);
    w.stream = st;
    (void)hipEventCreate(&w.ev0);
    (void)hipEventCreate(&w.ev1);
    w.allocated = true;
    return GF_OK;
}

void worker_advance(gf_ctx::Worker& w);
int worker_launch(gf_ctx* ctx, gf_algo algo, uint64_t first_ticket);

// The launch has left the device (its stream is idle): duration between the two events around it, tickets it relayed.
// (the duration itself is read from the events when somebody asks — gf_worker_kernel_time — or before they are recorded again:
//  hipEventElapsedTime is a microsecond or two of a caller's window that nobody else needs)
void worker_elapsed(gf_ctx::Worker& w) {
    if (!w.elapsed_pending) return;
    w.elapsed_pending = false;
    float ms = 0.0f;
    if (w.ev0 && w.ev1 && hipEventElapsedTime(&ms, w.ev0, w.ev1) == hipSuccess) w.last_ms = ms;
    (void)hipGetLastError();
}
void worker_finished(gf_ctx::Worker& w) {
    worker_elapsed(w);  // (a launch that finished earlier and was never asked about)
    const uint64_t consumed = host_load(&w.h->consumed);
    w.last_tickets = consumed > w.launch_first ? consumed - w.launch_first : 0;
    w.elapsed_pending = w.ev0 != nullptr && w.ev1 != nullptr;
}

// Makes the launch on the device (if any) leave once it has relayed and served every ticket posted so far, and waits for that.
// The leader may have idled out (or been stopped by worker_wait_ticket's 5 s limit) just as the last tickets were posted: it
// then left with consumed < posted.  Those tickets are re-driven here, on the still-installed snapshot, before the caller —
// usually an install — may go on; a context whose tickets cannot be served any more forgets them instead of refusing every
// later call.
int worker_join(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (!w.running) return GF_OK;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    for (int attempt = 0;; ++attempt) {
        host_store(&w.h->stop, w.posted + 2);  // "leave after ticket posted - 1" (gangfit_worker.inc)
        const hipError_t e = gf_wait_stream(w.stream);
        host_store(&w.h->stop, 0);
        w.running = false;
        if (e != hipSuccess) return fail(ctx, GF_ERR_HIP, "the worker did not leave the device: %s", hipGetErrorString(e));
        worker_finished(w);
        worker_advance(w);
        if (w.completed_upto == w.posted) return GF_OK;
        const uint64_t consumed = host_load(&w.h->consumed);
        if (attempt < 4 && w.algo >= 0 && consumed < w.posted && w.epoch == ctx->snap_epoch) {
            if (const int rc = worker_launch(ctx, (gf_algo)w.algo, consumed); rc != GF_OK) return rc;
            continue;
        }
        const uint64_t lost_lo = w.completed_upto, lost_hi = w.posted;
        w.completed_upto = w.posted;  // forget them: the ring is usable again (their callers were told, or never will wait)
        host_store(&w.h->consumed, w.posted);
        return fail(ctx, GF_ERR_HIP, "the worker left with tickets %llu .. %llu unserved (relayed %llu)", (unsigned long long)lost_lo,
                    (unsigned long long)lost_hi, (unsigned long long)consumed);
    }
}

// (Re)launches the worker for tickets >= first_ticket on the installed snapshot.
int worker_launch(gf_ctx* ctx, gf_algo algo, uint64_t first_ticket) {
    gf_ctx::Worker& w = ctx->worker;
    GF_HIP(ctx, hipSetDevice(ctx->device));
    host_store(&w.h->state, 0);
    host_store(&w.h->stop, 0);
    gangfit::WorkerArgs a{};
    a.host = w.h_dev;
    a.dev = w.d;
    a.generation = w.launches + 1;
    a.first_ticket = first_ticket;
    a.idle_ticks = (unsigned long long)w.idle_us * 100ull;  // wall_clock64 ticks at 100 MHz
    a.leave_after = (w.leave_after != 0 && w.leave_after == w.posted && first_ticket < w.posted) ? w.leave_after : 0ull;
    w.leave_after = 0;
    a.scratch = w.scratch.ptr;
    a.scratch_stride = w.scratch_stride;
    // every workgroup must be resident at once (a group that waits for a CU would leave its tickets unserved while the others
    // spin), and sixteen CUs stay free for FIFO chains (a chain needs a whole CU: sixteen wavefronts, the LDS): a workgroup of
    // the worker fills a CU's registers, so it has a CU to itself and the count of workgroups is the count of CUs taken
    uint32_t sets = w.sets, bps = w.blocks_per_set;
    {
        const uint32_t cus = (uint32_t)ctx->info.compute_units;
        // (the occupancy query is a runtime call of a microsecond or two, and a launch may sit inside a caller's timed window:
        //  asked once per packer)
        int& per_cu = w.per_cu[(unsigned)algo & 7u];
        if (per_cu == 0) GF_HIP(ctx, gangfit::worker_blocks_per_cu(algo, &per_cu));
        if (per_cu < 1) return fail(ctx, GF_ERR_HIP, "the worker kernel does not fit a CU");
        const uint32_t room = cus > 32 ? cus - 16u : cus;  // (per_cu is 1 for the tightly-pack instance; never count on more)
        // Unless the options say otherwise: THREE applications of a ticket per wavefront, one after the other, and as many sets as
        // then fit.  A ticket is a chain of dependent misses per application, not arithmetic — a set that gives every wavefront one
        // application (63 workgroups for 1 000) finishes a ticket in ~5.8 us but only three such sets fit the device; with three
        // applications per wavefront a ticket takes ~13 us and eleven are in flight on the same CUs: 1.40 against 2.06 us per
        // ticket in a long stream, 61 against 74 us for a window of twenty (profiles/r5j_worker_sets.txt; two per wavefront
        // 1.56 us, four 1.41 us but 68 us for the window of twenty).  Sized by the first ticket this launch will serve.
        // (gf_worker_fit — one blocking ticket at a time — asks for one application per wavefront: nothing else is in flight.)
        if (bps == 0) {
            uint32_t n_first = w.hint_apps;
            if (first_ticket < w.posted)
                n_first = (uint32_t)(host_load(&w.h->ring[first_ticket % kRing].word[5]) & 0xFFFFFFFFull);
            const uint32_t per_wave = w.hint_per_wave ? w.hint_per_wave : 1u;
            constexpr uint32_t kWavesPerGroup = 16;
            bps = (n_first + per_wave * kWavesPerGroup - 1) / (per_wave * kWavesPerGroup);
            if (bps < 1) bps = 1;
            if (1u + bps > room) bps = room > 1u ? room - 1u : 1u;  // (a one-CU device: refused below, never divided by zero)
        }
        if (sets == 0) {
            sets = (room - 1u) / bps;
            if (sets > 16u) sets = 16u;
            if (sets < 1u) sets = 1u;
        }
        while (sets > 1 && 1u + sets * bps > room) --sets;
        if (1u + sets * bps > room) return fail(ctx, GF_ERR_INVALID, "worker_blocks_per_set does not fit the device");
    }
    a.sets = sets;
    a.blocks_per_set = bps;
    w.cur_sets = sets;
    w.cur_blocks_per_set = bps;
    a.stats = ctx->stats_on ? ctx->d_stats.ptr : nullptr;
    // the tickets already posted for this launch (at most one per set) ride in its arguments
    a.n_inline = 0;
    for (uint64_t t = first_ticket; t < w.posted && a.n_inline < gangfit::kWorkerInline && a.n_inline < sets; ++t, ++a.n_inline)
        for (int k = 0; k < 6; ++k) a.inline_words[a.n_inline][k] = host_load(&w.h->ring[t % kRing].word[k]);
    w.launch_first = first_ticket;
    worker_elapsed(w);  // the previous launch's duration, before its events are recorded again
    if (w.ev0) (void)hipEventRecord(w.ev0, w.stream);
    GF_HIP(ctx, gangfit::launch_fit_worker(algo, make_table(ctx, ctx->d_snap.ptr), make_sparse(ctx), a, w.stream));
    if (w.ev1) (void)hipEventRecord(w.ev1, w.stream);
    w.running = true;
    w.algo = (int)algo;
    w.epoch = ctx->snap_epoch;
    ++w.launches;
    return GF_OK;
}

void worker_advance(gf_ctx::Worker& w) {
    while (w.completed_upto < w.posted && host_load(&w.h->done[w.completed_upto % kRing]) == w.completed_upto + 1) ++w.completed_upto;
}

// The leader leaves when no ticket has arrived for a while — possibly just as one was posted.  When it has left: the old
// launch is joined (its wavefronts work off everything it relayed first) and, if tickets were posted that it did not relay,
// the worker is launched again from the first of them.
int worker_revive(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (w.running) {
        if (host_load(&w.h->state) != 2) return GF_OK;
        GF_HIP(ctx, hipSetDevice(ctx->device));
        GF_HIP(ctx, gf_wait_stream(w.stream));
        w.running = false;
        worker_finished(w);
    }
    // not on the device: whatever was posted behind the last ticket the leader relayed needs a launch
    const uint64_t consumed = host_load(&w.h->consumed);
    if (w.algo >= 0 && consumed < w.posted) {
        if (w.epoch != ctx->snap_epoch) return fail(ctx, GF_ERR_STATE, "the snapshot changed under a posted ticket");
        return worker_launch(ctx, (gf_algo)w.algo, consumed);
    }
    return GF_OK;
}

// Waits for ticket t (t < posted).
int worker_wait_ticket(gf_ctx* ctx, uint64_t t) {
    gf_ctx::Worker& w = ctx->worker;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    for (;;) {
        if (t < w.completed_upto || host_load(&w.h->done[t % kRing]) == t + 1) return GF_OK;
        if (wait_blocking() && (spins & 0x7u) == 0x7u)  // GANGFIT_WAIT=block: the host cannot spare the core for the wait
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        if ((++spins & 0x3Fu) == 0) {
            if (const int rc = worker_revive(ctx); rc != GF_OK) return rc;
            if (!w.running && host_load(&w.h->done[t % kRing]) != t + 1)
                return fail(ctx, GF_ERR_STATE, "ticket %llu was never served (posted %llu, doorbell %llu, relayed %llu, complete below %llu, "
                            "completion word %llu, launches %llu)", (unsigned long long)t, (unsigned long long)w.posted,
                            (unsigned long long)host_load(&w.h->posted), (unsigned long long)host_load(&w.h->consumed),
                            (unsigned long long)w.completed_upto, (unsigned long long)host_load(&w.h->done[t % kRing]),
                            (unsigned long long)w.launches);
            if ((spins & 0xFFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                host_store(&w.h->stop, 1);
                return fail(ctx, GF_ERR_HIP, "the worker did not complete ticket %llu within 5 s", (unsigned long long)t);
            }
        }
    }
}

int worker_drain(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    worker_advance(w);
    for (uint64_t t = w.completed_upto; t < w.posted; ++t)
        if (const int rc = worker_wait_ticket(ctx, t); rc != GF_OK) return rc;
    worker_advance(w);
    return GF_OK;
}

// need_launch (nullable): instead of launching, report that a launch for tickets >= posted is needed — the caller posts its
// tickets first, so that the leader finds them at its first look (gf_worker_submit_dev).
int worker_prepare(gf_ctx* ctx, gf_algo algo, uint64_t max_total_k, bool* need_launch = nullptr) {
    gf_ctx::Worker& w = ctx->worker;
    if (!ctx->group.empty() || ctx->view_of != nullptr)
        return fail(ctx, GF_ERR_UNSUPPORTED, "the resident worker serves plain contexts (no views, one device)");
    if (algo != GF_ALGO_TIGHTLY_PACK && algo != GF_ALGO_DISTRIBUTE_EVENLY && algo != GF_ALGO_MINIMAL_FRAGMENTATION)
        return fail(ctx, GF_ERR_UNSUPPORTED, "the resident worker serves the plain packers");
    if (!ctx->have_orders) return fail(ctx, GF_ERR_STATE, "gf_snapshot_set + gf_orders_set must precede a fit");
    if (const int rc = worker_alloc(ctx); rc != GF_OK) return rc;
    if (const int rc = worker_revive(ctx); rc != GF_OK) return rc;
    const bool grow = max_total_k + 1 > w.scratch_stride;
    if (w.running && (w.algo != (int)algo || w.epoch != ctx->snap_epoch || grow)) {
        // another packer, another snapshot or a larger scratch: everything posted is served first, then the worker leaves
        if (const int rc = worker_join(ctx); rc != GF_OK) return rc;
    }
    if (grow) {
        if (const int rc = worker_drain(ctx); rc != GF_OK) return rc;
        uint64_t stride = w.scratch_stride ? w.scratch_stride : 1024;
        while (stride < max_total_k + 1) stride *= 2;
        GF_HIP(ctx, hipSetDevice(ctx->device));
        GF_HIP(ctx, w.scratch.reserve((size_t)kRing * 3 * stride));
        w.scratch_stride = stride;
    }
    // not running: every ticket posted so far was relayed and served (worker_revive re-drives the ones that were not)
    if (need_launch) *need_launch = !w.running;
    if (!w.running && !need_launch) return worker_launch(ctx, algo, w.posted);
    return GF_OK;
}

// Posts one ticket (the caller has made room in the ring).
void worker_post(gf_ctx::Worker& w, uint32_t n_apps, const gf_app* apps, gf_result* results, uint32_t* exec_nodes, uint64_t exec_len,
                 bool host_out) {
    gangfit::WorkerTicket& tk = w.h->ring[w.posted % kRing];
    const unsigned long long tag = gangfit::worker_tag(w.posted) << 48;
    tk.word[1] = (unsigned long long)reinterpret_cast<uintptr_t>(apps) | tag;
    tk.word[2] = (unsigned long long)reinterpret_cast<uintptr_t>(results) | tag;
    tk.word[3] = (unsigned long long)reinterpret_cast<uintptr_t>(exec_nodes) | tag;
    tk.word[4] = (unsigned long long)exec_len | tag;
    tk.word[5] = (unsigned long long)n_apps | ((unsigned long long)(host_out ? 1u : 0u) << 32) | tag;
    tk.word[0] = w.posted + 1;
    ++w.posted;
}
void worker_quiesce(gf_ctx* ctx) {
    gf_ctx::Worker& w = ctx->worker;
    if (!w.allocated) return;
    if (worker_revive(ctx) != GF_OK) return;  // (it may have left for lack of work just as tickets were posted)
    if (w.running)
        (void)worker_join(ctx);
    else
        (void)worker_drain(ctx);
}

}  // namespace gfapi

extern "C" {

int gf_worker_submit_dev(gf_ctx* ctx, gf_algo algo, uint32_t n_batches, const gf_worker_batch* batches, uint64_t* first_ticket) {
    if (!ctx || (n_batches > 0 && !batches)) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    uint64_t max_k = 0;
    for (uint32_t i = 0; i < n_batches; ++i) {
        if (batches[i].n_apps == 0 || !batches[i].d_apps || !batches[i].d_results)
            return fail(ctx, GF_ERR_INVALID, "batch %u: empty, or apps / results NULL", i);
        if (batches[i].exec_nodes_len > max_k) max_k = batches[i].exec_nodes_len;
    }
    bool need_launch = false;
    ctx->worker.hint_per_wave = 3;  // a stream of tickets: throughput (worker_launch)
    if (n_batches) ctx->worker.hint_apps = batches[0].n_apps;  // a launch sizes itself by the batch about to be posted, not by the last gf_worker_fit
    if (const int rc = worker_prepare(ctx, algo, max_k, &need_launch); rc != GF_OK) return rc;
    gf_ctx::Worker& w = ctx->worker;
    const uint64_t first = w.posted;
    if (first_ticket) *first_ticket = first;
    for (uint32_t i = 0; i < n_batches; ++i) {
        if (w.posted - w.completed_upto >= kRing) {  // the slot of ticket `posted` is free once ticket posted - ring is done
            host_store(&w.h->posted, w.posted);      // (ring the doorbell for what has been written so far)
            if (need_launch) {
                need_launch = false;
                if (const int rc = worker_launch(ctx, algo, first); rc != GF_OK) return rc;
            }
#ifdef GF_WORKER_HOST_DEBUG  // where the posting loop's time goes, and how many of the tickets in flight are done when the oldest is
            const auto td0 = std::chrono::steady_clock::now();
#endif
            if (const int rc = worker_wait_ticket(ctx, w.posted - kRing); rc != GF_OK) return rc;
#ifdef GF_WORKER_HOST_DEBUG
            {
                static double wait_ns = 0;
                static uint64_t n_wait = 0, done_behind = 0;
                wait_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - td0).count();
                for (uint64_t q = w.posted - kRing + 1; q < w.posted; ++q) done_behind += host_load(&w.h->done[q % kRing]) == q + 1 ? 1 : 0;
                if ((++n_wait % 1900) == 0)
                    std::fprintf(stderr, "[worker host] %llu waits: %.0f ns each, %.1f of the %u younger tickets already complete, relayed %llu of %llu posted\n",
                                 (unsigned long long)n_wait, wait_ns / n_wait, (double)done_behind / n_wait, kRing - 1,
                                 (unsigned long long)host_load(&w.h->consumed), (unsigned long long)w.posted);
            }
#endif
            worker_advance(w);
        }
        const gf_worker_batch& b = batches[i];
        worker_post(w, b.n_apps, b.d_apps, b.d_results, b.d_exec_nodes, b.exec_nodes_len, (b.flags & GF_WORKER_HOST_OUTPUTS) != 0);
    }
    host_store(&w.h->posted, w.posted);  // the doorbell: one word for the whole group
    if (need_launch) {
        // a bounded stream: the launch is told where it ends (only a launch that starts with everything already posted)
        if (n_batches != 0 && (batches[n_batches - 1].flags & GF_WORKER_LEAVE_AFTER) != 0) w.leave_after = w.posted;
        return worker_launch(ctx, algo, first);
    }
    return GF_OK;
}

int gf_worker_wait(gf_ctx* ctx, uint64_t first_ticket, uint32_t n_tickets) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    gf_ctx::Worker& w = ctx->worker;
    if (!w.allocated || first_ticket + n_tickets > w.posted) return fail(ctx, GF_ERR_INVALID, "tickets that were never posted");
    for (uint64_t t = first_ticket; t < first_ticket + n_tickets; ++t)
        if (const int rc = worker_wait_ticket(ctx, t); rc != GF_OK) return rc;
    worker_advance(w);
    return GF_OK;
}

int gf_worker_fit(gf_ctx* ctx, gf_algo algo, uint32_t n_apps, const gf_app* apps, gf_result* results, uint32_t* exec_nodes,
                  uint64_t exec_nodes_cap) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (n_apps > 0 && (!apps || !results)) return fail(ctx, GF_ERR_INVALID, "apps/results must not be NULL");
    if (n_apps == 0) return GF_OK;
    uint64_t total_k = 0;
    for (uint32_t a = 0; a < n_apps; ++a) {
        const gf_app& in = apps[a];
        if (in.k < 0 || in.k > GF_MAX_K) return fail(ctx, GF_ERR_INVALID, "apps[%u].k = %d outside [0, %d]", a, in.k, GF_MAX_K);
        for (int j = 0; j < 3; ++j)
            if (in.drv[j] < 0 || in.drv[j] >= GF_MAX_ABS_QUANTITY || in.exe[j] < 0 || in.exe[j] >= GF_MAX_ABS_QUANTITY)
                return fail(ctx, GF_ERR_INVALID, "apps[%u] request outside [0, 2^62)", a);
        total_k += (uint64_t)in.k;
    }
    if (total_k > exec_nodes_cap || (total_k > 0 && !exec_nodes))
        return fail(ctx, GF_ERR_CAPACITY, "exec_nodes holds %llu entries, %llu needed", (unsigned long long)exec_nodes_cap,
                    (unsigned long long)total_k);
    ctx->worker.hint_apps = n_apps;  // one blocking ticket: latency — an application per wavefront (worker_launch)
    ctx->worker.hint_per_wave = 1;
    if (const int rc = worker_prepare(ctx, algo, total_k); rc != GF_OK) return rc;
    gf_ctx::Worker& w = ctx->worker;
    // one pinned slice per ring slot: records in, results and placements out — the device reads and writes them in place
    if (n_apps > w.stage_apps || total_k + 1 > w.stage_k) {
        if (const int rc = worker_drain(ctx); rc != GF_OK) return rc;
        size_t na = w.stage_apps ? w.stage_apps : 1024, nk = w.stage_k ? w.stage_k : 16384;
        while (na < n_apps) na *= 2;
        while (nk < total_k + 1) nk *= 2;
        const size_t slice = na * (sizeof(gf_app) + sizeof(gf_result)) + nk * sizeof(uint32_t);
        void *hp = nullptr, *hd = nullptr;
        GF_HIP(ctx, hipSetDevice(ctx->device));
        GF_HIP(ctx, hipHostMalloc(&hp, slice * kRing, hipHostMallocMapped | hipHostMallocCoherent));
        if (hipHostGetDevicePointer(&hd, hp, 0) != hipSuccess) {
            (void)hipHostFree(hp);
            return fail(ctx, GF_ERR_HIP, "the worker's staging cannot be mapped to the device");
        }
        if (w.stage) (void)hipHostFree(w.stage);
        w.stage = hp;
        w.stage_dev = hd;
        w.stage_apps = na;
        w.stage_k = nk;
    }
    if (w.posted - w.completed_upto >= kRing) {
        if (const int rc = worker_wait_ticket(ctx, w.posted - kRing); rc != GF_OK) return rc;
        worker_advance(w);
    }
    const size_t slice = w.stage_apps * (sizeof(gf_app) + sizeof(gf_result)) + w.stage_k * sizeof(uint32_t);
    const size_t off = (size_t)(w.posted % kRing) * slice;
    char* hb = static_cast<char*>(w.stage) + off;
    char* db = static_cast<char*>(w.stage_dev) + off;
    gf_app* h_apps = reinterpret_cast<gf_app*>(hb);
    gf_result* h_res = reinterpret_cast<gf_result*>(hb + w.stage_apps * sizeof(gf_app));
    uint32_t* h_exec = reinterpret_cast<uint32_t*>(hb + w.stage_apps * (sizeof(gf_app) + sizeof(gf_result)));
    uint64_t k_off = 0;
    for (uint32_t a = 0; a < n_apps; ++a) {
        h_apps[a] = apps[a];
        h_apps[a].exec_off = k_off;
        k_off += (uint64_t)apps[a].k;
    }
    const uint64_t ticket = w.posted;
    worker_post(w, n_apps, reinterpret_cast<const gf_app*>(db), reinterpret_cast<gf_result*>(db + w.stage_apps * sizeof(gf_app)),
                reinterpret_cast<uint32_t*>(db + w.stage_apps * (sizeof(gf_app) + sizeof(gf_result))), total_k, true);
    host_store(&w.h->posted, w.posted);
    if (const int rc = worker_wait_ticket(ctx, ticket); rc != GF_OK) return rc;
    worker_advance(w);
    std::memcpy(results, h_res, (size_t)n_apps * sizeof(gf_result));
    if (total_k) std::memcpy(exec_nodes, h_exec, (size_t)total_k * sizeof(uint32_t));
    return GF_OK;
}

int gf_worker_stop(gf_ctx* ctx) {
    if (!ctx) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    worker_quiesce(ctx);
    return GF_OK;
}

int gf_worker_stats(gf_ctx* ctx, uint64_t out[4]) {
    if (!ctx || !out) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const gf_ctx::Worker& w = ctx->worker;
    out[0] = w.posted;
    out[1] = w.completed_upto;
    out[2] = w.launches;
    out[3] = (w.allocated && w.running && host_load(&w.h->state) != 2) ? 1 : 0;
    return GF_OK;
}

int gf_worker_geometry(gf_ctx* ctx, uint32_t out[2]) {
    if (!ctx || !out) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    out[0] = ctx->worker.cur_sets;
    out[1] = ctx->worker.cur_blocks_per_set;
    return GF_OK;
}

int gf_worker_kernel_time(gf_ctx* ctx, float* ms, uint64_t* tickets) {
    if (!ctx || !ms || !tickets) return GF_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const gf_ctx::Worker& w = ctx->worker;
    if (!w.allocated || w.launches == 0 || (w.running && w.launches == 1))
        return fail(ctx, GF_ERR_STATE, "no launch of the worker has finished yet (gf_worker_stop first)");
    if (!w.running) worker_elapsed(ctx->worker);
    *ms = w.last_ms;
    *tickets = w.last_tickets;
    return GF_OK;
}

}  // extern "C"
