// Exchange step of a multi-GPU product Q*v: ONE collective per product.
//
//   row-block panels (dense Q, streamed Gram):   in-place all-gather of the ranks' output slices
//   symmetric tile panels (kernel-built Gram):   in-place all-gather of the ranks' per-segment partial vectors, which every
//                                                rank then adds in segment order (bq_symv.hip) — bit-identical for any
//                                                rank count; BQ_SYM_EXCHANGE=allreduce: one ncclAllReduce(sum) instead
//
// RCCL (librccl.so, "nccl" on ROCm) is loaded lazily with dlopen so that single-GPU use never pays for it and a host
// without RCCL fails with a clear message only when a multi-rank context is requested; types and enums come from
// <rccl/rccl.h>.  One process per GPU.  Collectives are issued on the context's compute stream, so they are ordered after
// the producing kernel and before the consumers without host synchronisation.  Messages are small (0.8 MB per rank at
// n = 100 000): latency-bound over xGMI.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>

#include "bq_common.h"

namespace {
static_assert(sizeof(ncclUniqueId) == 128, "bq_comm_unique_id hands out 128 bytes");

struct rccl_api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
rccl_api g_rccl;

// ---- where the time of a communicator's creation goes ------------------------------------------------------
// Round 4 lost > 90 s inside Context.__init__ of a one-rank RCCL context on a cold box and the Python-level dump could not tell
// dlopen of the 573 MB librccl.so from ncclGetUniqueId from ncclCommInitRank.  Every stage is now clocked: the durations are kept
// (bq_comm_init_report), a stage that is still running after 5 s says so on stderr every 5 s WHILE it runs (so a hang names its
// stage), and a stage that took more than 5 s — or any stage under NCCL_DEBUG=INFO — is reported when it ends.
struct stage_log {
    std::mutex m;
    std::string text;   // "dlopen(librccl.so.1) 0.412 s; dlsym x8 0.000 s; ..."
};
stage_log g_stages;

struct stage_scope {
    const char *name;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::mutex m;
    std::condition_variable cv;
    bool over = false;
    std::thread ticker;
    explicit stage_scope(const char *nm) : name(nm) {
        ticker = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            while (!cv.wait_for(lk, std::chrono::seconds(5), [this] { return over; }))
                fprintf(stderr, "[bcqp] RCCL start-up: still inside %s after %.0f s\n", name, seconds());
        });
    }
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    ~stage_scope() {
        {
            std::lock_guard<std::mutex> lk(m);
            over = true;
        }
        cv.notify_all();
        ticker.join();
        const double dt = seconds();
        static const bool verbose = [] {
            const char *e = getenv("NCCL_DEBUG");   // RCCL's own switch: whoever asked RCCL to talk wants the stages too
            return e != nullptr && *e != 0 && strcmp(e, "VERSION") != 0 && strcmp(e, "WARN") != 0;
        }();
        if (dt > 5.0 || verbose) fprintf(stderr, "[bcqp] RCCL start-up: %s took %.3f s\n", name, dt);
        char buf[160];
        std::lock_guard<std::mutex> lk(g_stages.m);   // the separator looks at the text: under the lock too
        snprintf(buf, sizeof(buf), "%s%s %.3f s", g_stages.text.empty() ? "" : "; ", name, dt);
        g_stages.text += buf;
    }
};

int load_rccl() {
    if (g_rccl.handle) return BQ_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    {
        // RTLD_LAZY: the eight entry points used here are resolved by dlsym below; binding every other function of a 573 MB
        // library at load time (RTLD_NOW) buys nothing
        stage_scope st("dlopen(librccl.so)");
        for (const char *nm : names) {
            h = dlopen(nm, RTLD_LAZY | RTLD_GLOBAL);
            if (h) break;
        }
    }
    if (!h) {
        bq_set_error("cannot load RCCL (librccl.so): %s", dlerror());
        return BQ_ERR_RCCL;
    }
    {
        stage_scope st("dlsym x8");
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
        g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");
        g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
        g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    }
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.CommCount || !g_rccl.AllGather ||
        !g_rccl.AllReduce) {
        bq_set_error("librccl.so lacks a required symbol");
        dlclose(h);
        return BQ_ERR_RCCL;
    }
    g_rccl.handle = h;
    return BQ_OK;
}

int rccl_fail(const char *what, ncclResult_t r) {
    bq_set_error("%s failed: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", (int)r);
    return BQ_ERR_RCCL;
}

// The HIP-event pair around one collective: handed back to the context's pool on every path that does not reach end()
// (an error return must not leak the events — VERDICT r3 13c).
struct prof_scope {
    bq_ctx *c;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ended = false;
    explicit prof_scope(bq_ctx *ctx) : c(ctx) {}
    int begin() { return bq_prof_begin(c, BQ_PROF_EXCH, &e0, &e1); }
    int end() {
        ended = true;
        return bq_prof_end(c, BQ_PROF_EXCH, e0, e1);
    }
    ~prof_scope() {
        if (ended) return;
        if (e0) c->event_pool.push_back(e0);
        if (e1) c->event_pool.push_back(e1);
    }
};

// pinned host staging of the callback exchange
int pinned_reserve(bq_ctx *ctx, size_t bytes) {
    if (ctx->pinned_cap >= bytes) return BQ_OK;
    if (ctx->pinned) hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_cap = 0;
    BQ_HIP(hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_cap = bytes;
    return BQ_OK;
}

// callback transport, op 0: rows [r0, r1) of the n-vector `dev` are this rank's; on return every rank holds all rows
int callback_gather(bq_ctx *ctx, double *dev, int64_t n, int64_t r0, int64_t r1) {
    const size_t bytes = sizeof(double) * (size_t)n;
    BQ_TRY(pinned_reserve(ctx, bytes));
    if (r1 > r0)
        BQ_HIP(hipMemcpyAsync(ctx->pinned + r0, dev + r0, sizeof(double) * (size_t)(r1 - r0), hipMemcpyDeviceToHost,
                              ctx->stream));
    BQ_HIP(hipStreamSynchronize(ctx->stream));
    int rc = ctx->exch_fn(ctx->exch_user, ctx->pinned, n, r0, r1, 0);
    if (rc != 0) {
        bq_set_error("exchange callback returned %d", rc);
        return BQ_ERR_RCCL;
    }
    BQ_HIP(hipMemcpyAsync(dev, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return BQ_OK;
}
}  // namespace

extern "C" int bq_comm_unique_id(void *uid128) {
    BQ_ARG(uid128 != nullptr, "uid is NULL");
    BQ_TRY(load_rccl());
    ncclUniqueId id;
    ncclResult_t r;
    {
        stage_scope st("ncclGetUniqueId");
        r = g_rccl.GetUniqueId(&id);
    }
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(uid128, &id, sizeof(id));
    return BQ_OK;
}

// "stage seconds; stage seconds; ..." of every RCCL start-up stage this process has gone through so far (empty: RCCL never loaded)
extern "C" int bq_comm_init_report(char *buf, size_t cap) {
    BQ_ARG(buf != nullptr && cap > 0, "buf is NULL");
    std::lock_guard<std::mutex> lk(g_stages.m);
    snprintf(buf, cap, "%s", g_stages.text.c_str());
    return BQ_OK;
}

int bq_comm_init_rccl(bq_ctx *ctx, const void *uid128) {
    BQ_TRY(load_rccl());
    ncclUniqueId id;
    memcpy(&id, uid128, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r;
    {
        stage_scope st("ncclCommInitRank");
        r = g_rccl.CommInitRank(&comm, ctx->world, id, ctx->rank);
    }
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    int count = 0;
    {
        stage_scope st("ncclCommCount");
        r = g_rccl.CommCount(comm, &count);
    }
    if (r != ncclSuccess || count != ctx->world) {
        g_rccl.CommDestroy(comm);
        if (r != ncclSuccess) return rccl_fail("ncclCommCount", r);
        bq_set_error("RCCL communicator has %d ranks, expected %d", count, ctx->world);
        return BQ_ERR_RCCL;
    }
    ctx->nccl_comm = comm;
    ctx->comm_kind = BQ_COMM_RCCL;
    return BQ_OK;
}

int bq_comm_size(const bq_ctx *ctx) {   // ranks of the live RCCL communicator (0: none)
    if (ctx->comm_kind != BQ_COMM_RCCL || !ctx->nccl_comm) return 0;
    int count = 0;
    if (g_rccl.CommCount((ncclComm_t)ctx->nccl_comm, &count) != ncclSuccess) return 0;
    return count;
}

void bq_comm_destroy(bq_ctx *ctx) {
    if (ctx->nccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->nccl_comm);
    ctx->nccl_comm = nullptr;
}

// ---------------------------------------------------------------------------------------------------------------------
// Bounded collectives.  After ncclCommInitRank nothing in RCCL bounds a collective in time: if a peer
// stops taking part — an error on that rank only, a crashed process — this rank's collective kernel spins for ever and the
// host sits in the next wait on the stream.  A launcher that kills the job on the first non-zero exit hides that
// (bench.py spawn_ranks, torch.distributed.run); a library user has no such parent.  With a timeout set
// (bq_ctx_set_collective_timeout, BQ_COLLECTIVE_TIMEOUT_S) a watchdog thread watches the host's waits on the compute
// stream (bq_ctx_sync / bq_ctx_event_sync stamp them); a wait older than the limit gets the communicator aborted
// (ncclCommAbort ends its kernels), the wait returns and the call fails with BQ_ERR_RCCL, as does every later collective of
// the context.  The callback transport needs no watchdog: its exchange runs on the host inside the caller's communicator,
// whose own timeout (SocketComm / gloo / ThreadComm) ends the call.
// ---------------------------------------------------------------------------------------------------------------------
struct bq_watchdog {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    double timeout_s = 0.0;
    std::atomic<long long> wait_since_ns{0};   // 0: the host is not inside a wait that may sit behind a collective
    std::atomic<bool> fired{false};
    // a collective has been enqueued since the stream was last seen empty: only then can a wait be a wait for a PEER.  A wait with
    // nothing but this rank's own kernels ahead of it (a factorisation of a replicated solver, a preconditioner rebuild) is never
    // stamped, however long it lasts
    std::atomic<bool> collective_pending{false};
};

namespace {
long long now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void watchdog_loop(bq_ctx *ctx, bq_watchdog *wd) {
    std::unique_lock<std::mutex> lk(wd->m);
    while (!wd->stop) {
        wd->cv.wait_for(lk, std::chrono::milliseconds(20));
        if (wd->stop) break;
        const long long since = wd->wait_since_ns.load();
        if (since == 0 || wd->fired.load()) continue;
        if ((double)(now_ns() - since) * 1e-9 < wd->timeout_s) continue;
        // still under the mutex: the waiting thread cannot leave its wait scope (it takes the mutex to clear the stamp) while
        // the communicator is being torn down under it
        wd->fired.store(true);
        ctx->comm_aborted = true;
        if (ctx->comm_kind == BQ_COMM_RCCL && ctx->nccl_comm && g_rccl.CommAbort) {
            g_rccl.CommAbort((ncclComm_t)ctx->nccl_comm);
            ctx->nccl_comm = nullptr;   // aborted: never destroyed again
        }
    }
}

struct wait_scope {   // stamps a host wait that has a collective ahead of it; the stamp is cleared under the watchdog's mutex
    bq_watchdog *wd;
    explicit wait_scope(bq_ctx *ctx) : wd(ctx->watchdog) {
        if (wd && !wd->collective_pending.load()) wd = nullptr;
        if (wd) wd->wait_since_ns.store(now_ns());
    }
    // drained: the wait that ends was for the whole stream and succeeded, so no collective is outstanding any more
    void release(bool drained) {
        if (!wd) return;
        std::lock_guard<std::mutex> lk(wd->m);
        wd->wait_since_ns.store(0);
        if (drained && !wd->fired.load()) wd->collective_pending.store(false);
        wd = nullptr;
    }
    ~wait_scope() { release(false); }
};
inline void note_collective(bq_ctx *ctx) {
    if (ctx->watchdog) ctx->watchdog->collective_pending.store(true);
}

int aborted_error(bq_ctx *ctx) {
    bq_set_error("a collective did not complete within %.1f s (a peer rank stopped taking part?): the communicator of rank %d was "
                 "aborted; this context is unusable", ctx->watchdog ? ctx->watchdog->timeout_s : 0.0, ctx->rank);
    return BQ_ERR_RCCL;
}
}  // namespace

// (A spin-on-query before blocking was measured in round 4: dense ActiveSet at n = 20 000 to 'optimal' 18.03 s with, 18.01 s without —
// hipStreamSynchronize does not add a wake-up latency worth removing on this stack.)
int bq_ctx_sync(bq_ctx *ctx) {
    hipError_t e;
    {
        wait_scope scope(ctx);
        e = hipStreamSynchronize(ctx->stream);
        scope.release(e == hipSuccess);
    }
    if (ctx->comm_aborted) return aborted_error(ctx);
    if (e != hipSuccess) {
        bq_set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    return BQ_OK;
}

// The host waits for a word the DEVICE stores into mapped, coherent pinned memory (a kernel's last workgroup: system-scope fence, then
// the sequence number) instead of draining the stream: no copy command, no wait for the stream's completion signal — the data the
// host wants rides in the same pinned buffer and is complete when the word arrives (tools/look_probe.hip: 5.7 us per look against
// 13.5 us for hipMemcpyAsync + hipStreamSynchronize).  The spin looks at the stream now and then: a stream that has drained (or
// failed) without the word having arrived is an error, never an endless wait; the watchdog sees the wait like any other.
int bq_ctx_wait_flag(bq_ctx *ctx, const volatile int *flag, int want) {
    wait_scope scope(ctx);
    for (unsigned long long spins = 1;; ++spins) {
        if (__atomic_load_n(const_cast<const int *>(flag), __ATOMIC_ACQUIRE) == want) return BQ_OK;
        if ((spins & 0xffff) == 0) {
            if (ctx->comm_aborted) return aborted_error(ctx);
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {   // everything enqueued has run: the word is there now, or it never will be
                if (__atomic_load_n(const_cast<const int *>(flag), __ATOMIC_ACQUIRE) == want) return BQ_OK;
                bq_set_error("the stream drained without the device posting its record (expected %d, found %d)", want, (int)*flag);
                return BQ_ERR_HIP;
            }
            if (q != hipErrorNotReady) {
                bq_set_error("hipStreamQuery failed while waiting for a device record: %s", hipGetErrorString(q));
                return BQ_ERR_HIP;
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

int bq_ctx_event_sync(bq_ctx *ctx, hipEvent_t ev) {
    hipError_t e;
    {
        wait_scope scope(ctx);
        e = hipEventSynchronize(ev);
    }
    if (ctx->comm_aborted) return aborted_error(ctx);
    if (e != hipSuccess) {
        bq_set_error("hipEventSynchronize failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    return BQ_OK;
}

void bq_watchdog_stop(bq_ctx *ctx) {
    bq_watchdog *wd = ctx->watchdog;
    if (!wd) return;
    {
        std::lock_guard<std::mutex> lk(wd->m);
        wd->stop = true;
    }
    wd->cv.notify_all();
    if (wd->th.joinable()) wd->th.join();
    delete wd;
    ctx->watchdog = nullptr;
}

extern "C" int bq_ctx_set_collective_timeout(bq_ctx *ctx, double seconds) {
    BQ_ARG(ctx != nullptr, "ctx is NULL");
    BQ_ARG(seconds >= 0.0 && seconds <= 86400.0, "timeout in [0, 86400] seconds (0: none)");
    bq_watchdog_stop(ctx);
    // only an RCCL communicator can leave the host waiting for a peer for ever; on any other context the setting is accepted and
    // does nothing (a global BQ_COLLECTIVE_TIMEOUT_S must not cut short the long, legitimate waits of a single-GPU factorisation)
    if (seconds == 0.0 || ctx->comm_kind != BQ_COMM_RCCL) return BQ_OK;
    bq_watchdog *wd = new bq_watchdog();
    wd->timeout_s = seconds;
    ctx->watchdog = wd;
    wd->th = std::thread(watchdog_loop, ctx, wd);
    return BQ_OK;
}

// buf holds world*chunk doubles; this rank's chunk (at rank*chunk) is fresh on entry, all chunks on return
int bq_exchange_gather(bq_ctx *ctx, double *buf, int64_t chunk) {
    if (ctx->comm_kind == BQ_COMM_NONE || ctx->comm_kind == BQ_COMM_SHARE) return BQ_OK;
    if (ctx->comm_aborted) return aborted_error(ctx);
    prof_scope prof(ctx);
    BQ_TRY(prof.begin());
    if (ctx->comm_kind == BQ_COMM_RCCL) {
        note_collective(ctx);
        ncclResult_t r = g_rccl.AllGather(buf + (int64_t)ctx->rank * chunk, buf, (size_t)chunk, ncclDouble,
                                          (ncclComm_t)ctx->nccl_comm, ctx->stream);
        if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
    } else if (ctx->comm_kind == BQ_COMM_CALLBACK) {
        BQ_TRY(callback_gather(ctx, buf, chunk * ctx->world, chunk * ctx->rank, chunk * (ctx->rank + 1)));
    } else {
        bq_set_error("multi-rank context without an exchange");
        return BQ_ERR_BADARG;
    }
    return prof.end();
}

// s holds world*blk doubles; rows [r0,r1) (this rank's block, r0 == rank*blk, clipped to n) are fresh on entry
int bq_exchange_rows(bq_ctx *ctx, double *s, int64_t n, int64_t blk, int64_t r0, int64_t r1) {
    if (ctx->comm_kind == BQ_COMM_NONE || ctx->comm_kind == BQ_COMM_SHARE) return BQ_OK;
    if (ctx->comm_kind != BQ_COMM_CALLBACK) return bq_exchange_gather(ctx, s, blk);
    prof_scope prof(ctx);
    BQ_TRY(prof.begin());
    BQ_TRY(callback_gather(ctx, s, n, r0, r1));
    return prof.end();
}

// in-place all-reduce(sum) of v[0:count) — BQ_SYM_EXCHANGE=allreduce: every rank holds partial sums for every output block
int bq_exchange_sum(bq_ctx *ctx, double *v, int64_t count) {
    if (ctx->comm_kind == BQ_COMM_NONE || ctx->comm_kind == BQ_COMM_SHARE) return BQ_OK;
    if (ctx->comm_aborted) return aborted_error(ctx);
    prof_scope prof(ctx);
    BQ_TRY(prof.begin());
    if (ctx->comm_kind == BQ_COMM_RCCL) {
        note_collective(ctx);
        ncclResult_t r = g_rccl.AllReduce(v, v, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)ctx->nccl_comm, ctx->stream);
        if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    } else {
        const size_t bytes = sizeof(double) * (size_t)count;
        BQ_TRY(pinned_reserve(ctx, bytes));
        BQ_HIP(hipMemcpyAsync(ctx->pinned, v, bytes, hipMemcpyDeviceToHost, ctx->stream));
        BQ_HIP(hipStreamSynchronize(ctx->stream));
        int rc = ctx->exch_fn(ctx->exch_user, ctx->pinned, count, 0, count, 1);
        if (rc != 0) {
            bq_set_error("exchange callback returned %d", rc);
            return BQ_ERR_RCCL;
        }
        BQ_HIP(hipMemcpyAsync(v, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    return prof.end();
}

// The context's closing collective timed on its own (include/bcqp.h): `reps` calls, each between its own pair of HIP events on
// the context's stream.  kind 0: bq_exchange_gather of `count` doubles per rank; kind 1: bq_exchange_sum of `count` doubles.
extern "C" int bq_ctx_probe_exchange(bq_ctx *ctx, int kind, int64_t count, int reps, double *mean_us, double *min_us) {
    BQ_ARG(ctx && mean_us && min_us, "NULL argument");
    BQ_ARG(kind == 0 || kind == 1, "kind: 0 all-gather, 1 all-reduce");
    BQ_ARG(count >= 1 && reps >= 1 && reps <= 10000, "count >= 1, 1 <= reps <= 10000");
    BQ_HIP(hipSetDevice(ctx->device));
    const size_t len = (size_t)count * (kind == 0 ? (size_t)ctx->world : 1);
    double *buf = nullptr;
    BQ_HIP(hipMalloc(&buf, sizeof(double) * len));
    std::vector<hipEvent_t> ev((size_t)2 * reps, nullptr);
    const bool prof = ctx->profiling;
    ctx->profiling = false;
    int rc = BQ_OK;
    hipError_t e = hipMemsetAsync(buf, 0, sizeof(double) * len, ctx->stream);
    for (size_t i = 0; e == hipSuccess && i < ev.size(); ++i) e = hipEventCreate(&ev[i]);
    auto call = [&]() { return kind == 0 ? bq_exchange_gather(ctx, buf, count) : bq_exchange_sum(ctx, buf, count); };
    for (int i = 0; e == hipSuccess && rc == BQ_OK && i < 3; ++i) rc = call();
    for (int i = 0; e == hipSuccess && rc == BQ_OK && i < reps; ++i) {
        e = hipEventRecord(ev[2 * i], ctx->stream);
        if (e == hipSuccess) rc = call();
        if (e == hipSuccess && rc == BQ_OK) e = hipEventRecord(ev[2 * i + 1], ctx->stream);
    }
    if (e == hipSuccess && rc == BQ_OK) rc = bq_ctx_sync(ctx);   // behind `reps` collectives: the bounded wait
    double sum = 0.0, mn = 1e300;
    for (int i = 0; e == hipSuccess && rc == BQ_OK && i < reps; ++i) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        sum += ms * 1e3;
        mn = ms * 1e3 < mn ? ms * 1e3 : mn;
    }
    for (hipEvent_t x : ev)
        if (x) hipEventDestroy(x);
    hipFree(buf);
    ctx->profiling = prof;
    BQ_TRY(rc);
    if (e != hipSuccess) {
        bq_set_error("exchange probe failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    *mean_us = sum / reps;
    *min_us = mn;
    return BQ_OK;
}
