// C-ABI entry points (include/bcqp.h): contexts, the device-resident quadratic, solver drivers.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>

#include "bq_al.h"

static thread_local std::string g_last_error;

void bq_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char *bq_last_error(void) { return g_last_error.c_str(); }
extern "C" int bq_abi_version(void) { return BQ_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int bq_device_count(int *count) {
    BQ_ARG(count != nullptr, "count is NULL");
    BQ_HIP(hipGetDeviceCount(count));
    return BQ_OK;
}

static int ctx_new(int device, bq_ctx **out) {
    BQ_ARG(out != nullptr, "out is NULL");
    int ndev = 0;
    BQ_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        bq_set_error("device %d out of range (%d visible)", device, ndev);
        return BQ_ERR_BADARG;
    }
    BQ_HIP(hipSetDevice(device));
    bq_ctx *c = new bq_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    BQ_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    // hipDeviceProp_t::name comes back empty on some gfx950 boxes (BENCH_r03: " (gfx950:sramecc+:xnack-)"): say what is known
    snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name[0] ? prop.name : "AMD Instinct MI350-series [name not reported by the driver]",
             prop.gcnArchName);
    BQ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const char *mode = getenv("BQ_SYM_EXCHANGE");
    c->sym_allreduce = mode != nullptr && strcmp(mode, "allreduce") == 0;
    bq_ctx_register(c, true);
    *out = c;
    return BQ_OK;
}

extern "C" int bq_ctx_create(int device, bq_ctx **out) { return ctx_new(device, out); }

int bq_comm_init_rccl(bq_ctx *ctx, const void *uid128);  // bq_comm.cpp
void bq_comm_destroy(bq_ctx *ctx);
int bq_comm_size(const bq_ctx *ctx);

// the segment table (bq_seg_table) is a fixed-size struct passed to kernels by value: 8 * ceil(world / 8) segments must fit
#define BQ_ARG_WORLD(world) BQ_ARG(bq_sym_segments(world) <= BQ_SYM_SEG_MAX, "world is limited to 64 ranks (BQ_SYM_SEG_MAX canonical segments)")

extern "C" int bq_ctx_create_rccl(int device, int rank, int world, const void *uid128, bq_ctx **out) {
    BQ_ARG(world >= 1 && rank >= 0 && rank < world, "rank/world");
    BQ_ARG_WORLD(world);
    BQ_ARG(uid128 != nullptr, "uid is NULL");
    bq_ctx *c = nullptr;
    BQ_TRY(ctx_new(device, &c));
    c->rank = rank;
    c->world = world;
    int rc = bq_comm_init_rccl(c, uid128);
    if (rc != BQ_OK) {
        bq_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return BQ_OK;
}

extern "C" int bq_ctx_create_exchange(int device, int rank, int world, bq_exchange_fn fn, void *user, bq_ctx **out) {
    BQ_ARG(world >= 1 && rank >= 0 && rank < world, "rank/world");
    BQ_ARG(fn != nullptr || world == 1, "exchange callback is NULL");
    BQ_ARG_WORLD(world);
    bq_ctx *c = nullptr;
    BQ_TRY(ctx_new(device, &c));
    c->rank = rank;
    c->world = world;
    c->comm_kind = world > 1 ? BQ_COMM_CALLBACK : BQ_COMM_NONE;
    c->exch_fn = fn;
    c->exch_user = user;
    *out = c;
    return BQ_OK;
}

extern "C" int bq_ctx_create_share(int device, int rank, int world, bq_ctx **out) {
    BQ_ARG(world >= 1 && rank >= 0 && rank < world, "rank/world");
    BQ_ARG_WORLD(world);
    bq_ctx *c = nullptr;
    BQ_TRY(ctx_new(device, &c));
    c->rank = rank;
    c->world = world;
    c->comm_kind = BQ_COMM_SHARE;
    *out = c;
    return BQ_OK;
}

extern "C" int bq_ctx_destroy(bq_ctx *c) {
    if (c == nullptr) return BQ_OK;
    if (c->refs > 0) {   // problems still hold this context (and its stream): it goes with the last of them
        c->zombie = true;
        return BQ_OK;
    }
    hipSetDevice(c->device);
    bq_ctx_register(c, false);
    // the bounded wait first, the watchdog's end after it: a context closed behind a collective whose peer has gone must not sit
    // here for ever either; after an abort the stream holds nothing worth waiting for
    if (c->stream && !c->comm_aborted) (void)bq_ctx_sync(c);
    bq_watchdog_stop(c);
    bq_comm_destroy(c);
    for (auto &slot : c->prof)
        for (auto &pr : slot.pending) {
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
    if (c->prof_skip) hipFree(c->prof_skip);
    for (auto e : c->event_pool) hipEventDestroy(e);
    if (c->pinned) hipHostFree(c->pinned);
    bq_ctx_drop_cache(c);
    bq_ctx_release_held(c, nullptr);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return BQ_OK;
}

extern "C" int bq_ctx_info(const bq_ctx *c, int *device, int *rank, int *world, char *name, size_t cap) {
    BQ_ARG(c != nullptr, "ctx is NULL");
    if (device) *device = c->device;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (name && cap) snprintf(name, cap, "%s", c->name);
    return BQ_OK;
}

extern "C" int bq_ctx_comm_info(const bq_ctx *c, int *kind, int *comm_ranks, int *sym_allreduce) {
    BQ_ARG(c != nullptr, "ctx is NULL");
    if (kind) *kind = c->comm_kind;
    if (comm_ranks) *comm_ranks = bq_comm_size(c);
    if (sym_allreduce) *sym_allreduce = c->sym_allreduce ? 1 : 0;
    return BQ_OK;
}

extern "C" int bq_ctx_set_sym_allreduce(bq_ctx *c, int on) {
    BQ_ARG(c != nullptr, "ctx is NULL");
    c->sym_allreduce = on != 0;
    return BQ_OK;
}

extern "C" int bq_ctx_profile(bq_ctx *c, int enable) {
    BQ_ARG(c != nullptr, "ctx is NULL");
    c->profiling = enable != 0;
    return BQ_OK;
}

// bracket == true: the pair brackets a region of the stream (e0 is recorded here, e1 by bq_prof_end).  bracket == false: the pair is
// handed to ONE kernel launch (hipExtLaunchKernelGGL(..., e0, e1, ...)), whose own dispatch carries the two timestamps — the
// kernel's duration as the tracer sees it, and no event packets on the stream; bq_prof_end is then called with recorded == true.
int bq_prof_begin(bq_ctx *c, int which, hipEvent_t *e0, hipEvent_t *e1, bool bracket) {
    *e0 = *e1 = nullptr;
    if (!c->profiling) return BQ_OK;
    hipError_t err = hipSuccess;
    for (hipEvent_t *e : {e0, e1}) {
        if (!c->event_pool.empty()) {
            *e = c->event_pool.back();
            c->event_pool.pop_back();
        } else if ((err = hipEventCreate(e)) != hipSuccess) {
            *e = nullptr;
            break;
        }
    }
    if (err == hipSuccess && bracket) err = hipEventRecord(*e0, c->stream);
    if (err != hipSuccess) {   // nothing leaks on the error path: what was taken goes back to the pool
        for (hipEvent_t *e : {e0, e1}) {
            if (*e) c->event_pool.push_back(*e);
            *e = nullptr;
        }
        bq_set_error("profiling events: %s", hipGetErrorString(err));
        return BQ_ERR_HIP;
    }
    (void)which;
    return BQ_OK;
}

void bq_prof_drop(bq_ctx *c, hipEvent_t e0, hipEvent_t e1) {
    if (e0) c->event_pool.push_back(e0);
    if (e1) c->event_pool.push_back(e1);
    c->prof_cur_idx = -1;
}

void bq_prof_skip_arg(bq_ctx *c, hipEvent_t e0, int **slot, int *seq) {
    *slot = nullptr;
    *seq = 0;
    c->prof_cur_idx = -1;
    if (e0 == nullptr) return;   // not profiling
    if (c->prof_skip == nullptr) {
        if (hipMalloc(&c->prof_skip, sizeof(int) * BQ_PROF_SKIP_CAP) != hipSuccess ||
            hipMemsetAsync(c->prof_skip, 0, sizeof(int) * BQ_PROF_SKIP_CAP, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            c->prof_skip = nullptr;
            return;
        }
    }
    if ((int64_t)c->prof[BQ_PROF_MATVEC].pending.size() >= BQ_PROF_SKIP_CAP - 1) return;   // ring full until the next read
    c->prof_seq += 1;
    c->prof_cur_idx = (int)(c->prof_seq % BQ_PROF_SKIP_CAP);
    c->prof_cur_seq = (int)(c->prof_seq & 0x7fffffff) | 1;   // never 0 (the ring starts zeroed)
    *slot = c->prof_skip + c->prof_cur_idx;
    *seq = c->prof_cur_seq;
}

int bq_prof_end(bq_ctx *c, int which, hipEvent_t e0, hipEvent_t e1, bool recorded) {
    if (e0 == nullptr) return BQ_OK;
    if (!recorded) BQ_HIP(hipEventRecord(e1, c->stream));
    bq_prof_pending pe;
    pe.first = e0;
    pe.second = e1;
    if (which == BQ_PROF_MATVEC && c->prof_cur_idx >= 0) {
        pe.skip_idx = c->prof_cur_idx;
        pe.skip_seq = c->prof_cur_seq;
    }
    c->prof_cur_idx = -1;
    c->prof[which].pending.push_back(pe);
    return BQ_OK;
}

extern "C" int bq_ctx_profile_read(bq_ctx *c, int which, double *total_ms, int64_t *launches, int reset) {
    BQ_ARG(c != nullptr && which >= 0 && which < BQ_PROF_COUNT, "ctx/which");
    BQ_SYNC(c);
    bq_prof_slot &s = c->prof[which];
    std::vector<int> ring;
    bool any_tag = false;
    for (const bq_prof_pending &pe : s.pending) any_tag = any_tag || pe.skip_idx >= 0;
    if (any_tag && c->prof_skip != nullptr) {
        ring.resize(BQ_PROF_SKIP_CAP);
        BQ_HIP(hipMemcpy(ring.data(), c->prof_skip, sizeof(int) * BQ_PROF_SKIP_CAP, hipMemcpyDeviceToHost));
    }
    for (const bq_prof_pending &pe : s.pending) {
        float ms = 0.f;
        BQ_HIP(hipEventElapsedTime(&ms, pe.first, pe.second));
        c->event_pool.push_back(pe.first);
        c->event_pool.push_back(pe.second);
        // a launch that returned at once on its `done` flag said so itself (bq_prof_skip_arg): it is no product and stays out of
        // the mean — exactly those launches, whatever their duration
        if (pe.skip_idx >= 0 && !ring.empty() && ring[(size_t)pe.skip_idx] == pe.skip_seq) {
            s.skipped += 1;
            continue;
        }
        s.total_ms += ms;
        s.launches += 1;
    }
    s.pending.clear();
    if (total_ms) *total_ms = s.total_ms;
    if (launches) *launches = s.launches;
    if (reset) {
        s.total_ms = 0.0;
        s.launches = 0;
        s.skipped = 0;
    }
    return BQ_OK;
}

static int64_t row_block_size(int64_t n, int world) {
    const int64_t per = (n + world - 1) / world;
    return bq_round_up(per > 0 ? per : 1, 128);  // tile-aligned for the Gram kernel
}

extern "C" int bq_row_block(int64_t n, int rank, int world, int64_t *begin, int64_t *end) {
    BQ_ARG(n >= 0 && world >= 1 && rank >= 0 && rank < world, "n/rank/world");
    const int64_t blk = row_block_size(n, world);
    int64_t b = (int64_t)rank * blk;
    if (b > n) b = n;
    int64_t e = b + blk;
    if (e > n) e = n;
    if (begin) *begin = b;
    if (end) *end = e;
    return BQ_OK;
}

// Balanced triangular partition of nb tile rows into S canonical segments: segment s starts at round(nb * sqrt(s / S)), so
// every segment holds about nb^2 / (2 S) tiles of the lower triangle.  Rank k of G owns the segments
// [floor(k S / G), floor((k+1) S / G)) — for G dividing S that is I_k = round(nb * sqrt(k / G)), equal tile counts per rank.
int64_t bq_sym_seg_cut(int64_t nb, int s, int S) {
    if (s <= 0) return 0;
    if (s >= S) return nb;
    int64_t v = (int64_t)std::llround((double)nb * std::sqrt((double)s / (double)S));
    return v < 0 ? 0 : (v > nb ? nb : v);
}

static void sym_tile_rows(int64_t nb, int rank, int world, int64_t *I0, int64_t *I1) {
    const int S = bq_sym_segments(world);
    *I0 = bq_sym_seg_cut(nb, bq_sym_seg_first(rank, world, S), S);
    *I1 = bq_sym_seg_cut(nb, bq_sym_seg_first(rank + 1, world, S), S);
}

void bq_sym_seg_table(const bq_problem *p, bq_seg_table *tab) {
    const int world = p->ctx->world;   // <= 64: checked where a multi-rank context is created (BQ_ARG_WORLD)
    tab->count = p->seg_count;
    tab->lo = p->seg_lo;
    tab->hi = p->seg_hi;
    for (int s = 0; s <= p->seg_count; ++s) tab->cut[s] = bq_sym_seg_cut(p->nb, s, p->seg_count);
    for (int k = 0; k < world; ++k) {
        const int lo = bq_sym_seg_first(k, world, p->seg_count), hi = bq_sym_seg_first(k + 1, world, p->seg_count);
        for (int s = lo; s < hi; ++s) tab->slot[s] = k * p->seg_cmax + (s - lo);
    }
}

extern "C" int bq_sym_row_block(int64_t n, int rank, int world, int64_t *begin, int64_t *end) {
    BQ_ARG(n >= 0 && world >= 1 && rank >= 0 && rank < world, "n/rank/world");
    const int64_t nb = (n + BQ_SYM_TILE - 1) / BQ_SYM_TILE;
    int64_t I0, I1;
    sym_tile_rows(nb, rank, world, &I0, &I1);
    if (begin) *begin = std::min(n, I0 * BQ_SYM_TILE);
    if (end) *end = std::min(n, I1 * BQ_SYM_TILE);
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// the quadratic
// ---------------------------------------------------------------------------------------------
static int problem_alloc_common(bq_problem *p, const double *q_host) {
    bq_ctx *c = p->ctx;
    const int64_t nblk = p->ldN / BQ_VEC_TILE;
    BQ_HIP(hipMalloc(&p->q, sizeof(double) * p->ldN));
    BQ_HIP(hipMemsetAsync(p->q, 0, sizeof(double) * p->ldN, c->stream));
    BQ_HIP(hipMemcpyAsync(p->q, q_host, sizeof(double) * p->N, hipMemcpyHostToDevice, c->stream));
    BQ_HIP(hipMalloc(&p->w, sizeof(double) * p->ld));
    BQ_HIP(hipMemsetAsync(p->w, 0, sizeof(double) * p->ld, c->stream));
    const int64_t slen = bq_round_up(std::max(p->blk * c->world, p->nb * BQ_SYM_TILE), BQ_PAD);
    if (p->symmetric) {
        if (!p->streamed) BQ_HIP(hipMalloc(&p->slab, sizeof(double) * p->nb * p->nb * BQ_SYM_TILE));   // streamed: its own scratch
        if (c->comm_kind != BQ_COMM_NONE) {   // the gathered segment vectors of every rank (also a one-rank communicator)
            const size_t gl = sizeof(double) * (size_t)c->world * p->seg_cmax * p->nb * BQ_SYM_TILE;
            BQ_HIP(hipMalloc(&p->gath, gl));
            BQ_HIP(hipMemsetAsync(p->gath, 0, gl, c->stream));
        }
    }
    BQ_HIP(hipMalloc(&p->s, sizeof(double) * slen));
    BQ_HIP(hipMemsetAsync(p->s, 0, sizeof(double) * slen, c->stream));
    BQ_HIP(hipMalloc(&p->va, sizeof(double) * p->ldN));
    BQ_HIP(hipMalloc(&p->vb, sizeof(double) * p->ldN));
    BQ_HIP(hipMemsetAsync(p->va, 0, sizeof(double) * p->ldN, c->stream));
    BQ_HIP(hipMemsetAsync(p->vb, 0, sizeof(double) * p->ldN, c->stream));
    BQ_HIP(hipMalloc(&p->partials, sizeof(double) * BQ_MAX_PARTIAL_Q * nblk));
    BQ_HIP(hipMalloc(&p->scal, sizeof(double) * 16));
    return BQ_OK;
}

static int problem_layout(bq_problem *p, int64_t n, int64_t N) {
    bq_ctx *c = p->ctx;
    p->n = n;
    p->N = N;
    p->ld = bq_round_up(n, BQ_PAD);
    p->ldN = bq_round_up(N, BQ_PAD);
    p->blk = row_block_size(n, c->world);
    int64_t rows;
    if (p->symmetric) {
        p->nb = (n + BQ_SYM_TILE - 1) / BQ_SYM_TILE;
        sym_tile_rows(p->nb, c->rank, c->world, &p->I0, &p->I1);
        p->seg_count = bq_sym_segments(c->world);
        p->seg_lo = bq_sym_seg_first(c->rank, c->world, p->seg_count);
        p->seg_hi = bq_sym_seg_first(c->rank + 1, c->world, p->seg_count);
        p->seg_cmax = (p->seg_count + c->world - 1) / c->world;
        p->r0 = std::min(n, p->I0 * BQ_SYM_TILE);
        p->r1 = std::min(n, p->I1 * BQ_SYM_TILE);
        rows = (p->I1 - p->I0) * BQ_SYM_TILE;  // whole tiles, zero rows past n
    } else {
        BQ_TRY(bq_row_block(n, c->rank, c->world, &p->r0, &p->r1));
        rows = p->r1 - p->r0;
    }
    if (p->streamed) return BQ_OK;   // no resident panel
    const size_t esz = p->storage == BQ_F64 ? 8 : 4;
    // symmetric panels are stored packed: tile row I keeps (I+1)*256 columns
    const size_t elems = p->symmetric ? (size_t)(bq_sym_off(p->I1) - bq_sym_off(p->I0))
                                      : (size_t)(rows > 0 ? rows : 1) * (size_t)p->ld;
    size_t bytes = (elems > 0 ? elems : 1) * esz;
    const size_t data_bytes = bytes;
    hipError_t e = hipSuccess;
    size_t cached = 0;
    if (void *kept = bq_ctx_cache_take(c, bytes, &cached)) {
        p->panel = kept;   // the panel a destroyed problem left behind
        p->panel_bytes = cached;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        e = hipMalloc(&p->panel, bytes);   // bq_device_malloc: drops the cached panel and retries on failure
        p->alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        p->panel_bytes = bytes;
    }
    p->panel_alloc = p->panel;
    (void)data_bytes;
    if (e != hipSuccess) {
        p->panel = nullptr;
        p->panel_alloc = nullptr;
        bq_set_error("cannot allocate the %lld x %lld panel (%.1f GB): %s", (long long)rows, (long long)p->ld,
                     bytes / 1e9, hipGetErrorString(e));
        return BQ_ERR_NOMEM;
    }
    BQ_HIP(hipMemsetAsync(p->panel, 0, bytes, c->stream));
    return BQ_OK;
}

void bq_problem_unref(bq_problem *p) {
    if (p == nullptr) return;
    if (--p->refs <= 0 && p->zombie) {
        p->refs = 0;
        p->zombie = false;
        bq_problem_destroy(p);
    }
}

extern "C" int bq_problem_destroy(bq_problem *p) {
    if (p == nullptr) return BQ_OK;
    if (p->refs > 0) {   // solvers still run on this problem: it goes with the last of them
        p->zombie = true;
        return BQ_OK;
    }
    hipSetDevice(p->ctx->device);
    (void)bq_ctx_sync(p->ctx);   // may sit behind a collective: bounded when a collective timeout is set
    if (p->panel_alloc && p->panel_bytes >= ((size_t)1 << 30)) {   // keep one large panel for the next problem (see bq_ctx)
        bq_ctx_cache_put(p->ctx, p->panel_alloc, p->panel_bytes);
        p->panel_alloc = nullptr;
    }
    p->panel = nullptr;
    bq_ctx_release_held(p->ctx, p);
    for (void *ptr : {(void *)p->panel_alloc, (void *)p->q, (void *)p->sgn, (void *)p->X, (void *)p->w, (void *)p->s,
                      (void *)p->va, (void *)p->vb, (void *)p->partials, (void *)p->scal, (void *)p->slab, (void *)p->gath})
        if (ptr) hipFree(ptr);
    bq_stream_free(p->stream_img);
    bq_ctx *c = p->ctx;
    delete p;
    if (--c->refs <= 0 && c->zombie) {
        c->refs = 0;
        c->zombie = false;
        bq_ctx_destroy(c);
    }
    return BQ_OK;
}

static int place_panel(bq_problem *p, double first_alloc_ms);

static int dense_new(bq_ctx *c, int64_t n, const double *q, int storage, bool symmetric, bool place, bq_problem **out) {
    bq_problem *p = new bq_problem();
    p->ctx = c;
    c->refs += 1;
    p->structure = BQ_PLAIN;
    p->storage = storage;
    p->symmetric = symmetric;
    int rc = problem_layout(p, n, n);
    if (rc == BQ_OK) rc = problem_alloc_common(p, q);
    if (rc == BQ_OK && place && symmetric) rc = place_panel(p, p->alloc_ms);
    if (rc != BQ_OK) {
        bq_problem_destroy(p);
        return rc;
    }
    *out = p;
    return BQ_OK;
}

// Quadratic(Q, q) (optiml/opti/_base.py:228-256).  A Q that equals its transpose exactly goes into the packed lower tile rows of
// the kernel-built panels (half the HBM, half the bytes per product: symv_tiles_kernel); any other Q keeps whole row blocks and
// NumPy's `Q @ x` (gemv_rows_kernel).  bq_dense.hip has the upload and the check.
extern "C" int bq_problem_create_dense(bq_ctx *c, int64_t n, const double *Q, const double *q, int storage,
                                       bq_problem **out) {
    BQ_ARG(c && Q && q && out, "NULL argument");
    BQ_ARG(n >= 2, "Q is too small");  // optiml/opti/_base.py:249-250
    const bool force_rows = (storage & BQ_DENSE_ROWS) != 0, trust_lower = (storage & BQ_DENSE_LOWER) != 0;
    const bool place = (storage & BQ_PLACE_PANEL) != 0;
    storage &= ~(BQ_DENSE_ROWS | BQ_DENSE_LOWER | BQ_PLACE_PANEL);
    BQ_ARG(storage == BQ_F64 || storage == BQ_F32, "storage");
    BQ_ARG(!(force_rows && trust_lower), "BQ_DENSE_ROWS and BQ_DENSE_LOWER exclude each other");
    BQ_HIP(hipSetDevice(c->device));
    bq_problem *p = nullptr;
    // every rank is handed the same Q, so the look at the host matrix sends them all the same way
    bool try_sym = !force_rows && (trust_lower || bq_dense_host_spot_symmetric(Q, n));
    if (try_sym) {
        // rank-local steps (the packed panel's allocation, the upload, the comparison), then ONE agreement on a multi-rank context:
        // "some rank failed" ends the call with an error on EVERY rank, "some rank saw a difference" sends every rank to row blocks
        int rc = dense_new(c, n, q, storage, true, place, &p);
        int is_sym = 1;
        if (rc == BQ_OK) rc = bq_dense_upload_sym(p, Q, !trust_lower, &is_sym);
        if (rc == BQ_OK) rc = bq_ctx_sync(c);
        bool any_failed = rc != BQ_OK, any_asym = !is_sym;
        if (!trust_lower || c->world > 1) {
            int arc = bq_dense_agree(c, rc != BQ_OK, &any_failed);
            if (arc == BQ_OK && !any_failed && !trust_lower) arc = bq_dense_agree(c, !is_sym, &any_asym);
            if (rc == BQ_OK && arc != BQ_OK) rc = arc;
        }
        if (rc != BQ_OK || any_failed) {
            if (p) bq_problem_destroy(p);
            if (rc == BQ_OK) {
                bq_set_error("the dense Hessian could not be set up on another rank");
                rc = BQ_ERR_HIP;
            }
            return rc;
        }
        if (!any_asym) {
            *out = p;
            return BQ_OK;
        }
        bq_problem_destroy(p);   // not symmetric after all: row blocks
        bq_ctx_drop_cache(c);    // (the packed panel is half the size the row blocks need: nothing to keep it for)
        p = nullptr;
    }
    BQ_TRY(dense_new(c, n, q, storage, false, false, &p));
    int rc = bq_dense_upload_rows(p, Q);
    if (rc == BQ_OK) rc = bq_ctx_sync(c);
    if (rc != BQ_OK) {
        bq_problem_destroy(p);
        return rc;
    }
    *out = p;
    return BQ_OK;
}

// BQ_PLACE_PANEL (bcqp.h): time the product kernel on the freshly allocated, zeroed panel; while it streams below the "good" rate,
// the device can hold one more panel and the TIME BUDGET allows it, allocate another candidate and time it; keep the fastest,
// HOLD the rest until the first solver on the problem is destroyed (or the problem, or an allocation fails: released earlier, the allocator would hand the same memory
// back as the next candidate; released at the end — rounds 4-5 — the release itself slowed the solve that followed: bq_ctx::held).
//
// What a slow placement is (round 4, profiles/r04/placement_*.txt): a property of the physical region the driver handed out, stable
// for the life of the allocation (6.04 ... 6.50 ms in clusters) — NOT of the base address (the same allocation at 8 - 20 byte
// offsets up to 1 GiB: flat to 0.5 %), not of the page-table contiguity the API can ask for (hipDeviceMallocContiguous: same
// spread), not of the translation caches (UTCL1 misses equal); on a slow panel the fabric sees FEWER requests in flight and LOWER
// latency (TCC_EA0_RDREQ_LEVEL -10 %, DRAM credit stalls -30 %) while the vector L1 waits longer (TCP_PENDING_STALL +8 %).  Nothing
// on the allocation side changes it, so the remedy stays a choice among allocations — bounded in time so that it can be ON by
// default (SVC / SVR.fit and bench.py alike): BQ_PLACE_BUDGET_MS (200) covers the timing of the first panel (5 products) and
// as many further candidates as fit, each priced at what the FIRST allocation of this size cost (freshly released device memory
// is cleared by the driver before it is handed out: seconds, not milliseconds — then no candidate fits the budget).
// Per rank and before the first collective of a multi-rank job: the other ranks wait at most the budget.
static int place_panel(bq_problem *p, double first_alloc_ms) {
    bq_ctx *c = p->ctx;
    if (!p->symmetric || p->streamed || p->panel == nullptr || p->panel_bytes < ((size_t)1 << 30) || p->I1 <= p->I0) return BQ_OK;
    const int want = 3;   // allocations tried at most (a 4th never won in round 4's sweeps: profiles/r04/placement_*.txt)
    const double good_gbs = bq_hook_value("panel_good_gbs", 6500.0);
    const char *e = getenv("BQ_PLACE_BUDGET_MS");
    const double budget_min = c->place_min_ms >= 0.0 ? c->place_min_ms : (e ? atof(e) : 200.0);
    double budget_ms = budget_min;   // grows with the work the caller expects once the product's time is known (below)
    const auto t_start = std::chrono::steady_clock::now();
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    bq_seg_table tab;
    bq_sym_seg_table(p, &tab);
    struct scope {   // events and the profiling switch go back on every exit path
        bq_ctx *c;
        bool prof;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~scope() {
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
            c->profiling = prof;
        }
    } sc{c, c->profiling};
    c->profiling = false;
    BQ_HIP(hipEventCreate(&sc.e0));
    BQ_HIP(hipEventCreate(&sc.e1));
    auto time_on = [&](void *panel, double *ms_out) -> int {
        int rc = bq_launch_symv(c, panel, p->storage, p->add_one, p->nb, tab, p->w, p->slab, p->s, nullptr);   // warm
        hipEventRecord(sc.e0, c->stream);
        for (int i = 0; rc == BQ_OK && i < 4; ++i)
            rc = bq_launch_symv(c, panel, p->storage, p->add_one, p->nb, tab, p->w, p->slab, p->s, nullptr);
        hipEventRecord(sc.e1, c->stream);
        hipError_t he = hipEventSynchronize(sc.e1);
        float ms = 0.f;
        if (he == hipSuccess) he = hipEventElapsedTime(&ms, sc.e0, sc.e1);
        if (rc == BQ_OK && he != hipSuccess) {
            bq_set_error("timing a panel placement failed: %s", hipGetErrorString(he));
            rc = BQ_ERR_HIP;
        }
        *ms_out = (double)ms / 4.0;
        return rc;
    };
    std::vector<void *> losers;
    double best = 0.0;
    int rc = time_on(p->panel, &best);
    p->place_tried = 1;
    p->place_ms[0] = best;
    // a panel that came from the context's cache (a destroyed problem of the same size left it: fit loops) was chosen when IT was
    // allocated, and a fresh allocation right after that release is the slow kind: it is kept as it is
    if (first_alloc_ms <= 0.0) return rc;
    // bq_ctx_set_placement_budget: a slow placement costs ~5 % of every product, so up to 2 % of the products the caller expects to
    // run may be spent on finding a better one (within [min, max]): SVC.fit passes its max_iter, a steady-state measurement "many"
    if (c->place_products > 0.0 && c->place_max_ms > budget_min)
        budget_ms = std::min(c->place_max_ms, std::max(budget_min, 0.02 * c->place_products * best));
    // what one more candidate costs: an allocation like the first one + clearing it + five products on it
    const double cand_ms = first_alloc_ms + (double)p->panel_bytes / 4.0e9 + 6.0 * best;
    while (rc == BQ_OK && p->place_tried < want && (double)p->panel_bytes / (best * 1e-3) / 1e9 < good_gbs &&
           elapsed_ms() + cand_ms <= budget_ms) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < p->panel_bytes + p->panel_bytes / 16) break;
        // the losers are held while the problem lives (bq_ctx::held): a second copy only if a tenth of the device stays free beside
        // the two (config 5's 125 GB panel on 288 GB: yes, once; a failing allocation later gives the held one back: bq_alloc.cpp)
        if (2 * p->panel_bytes + total_b / 10 > total_b || free_b < p->panel_bytes + total_b / 10) break;
        void *cand = nullptr;
        if (hipMalloc(&cand, p->panel_bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        hipError_t he = hipMemsetAsync(cand, 0, p->panel_bytes, c->stream);
        double t = 0.0;
        if (he == hipSuccess) rc = time_on(cand, &t);
        if (he != hipSuccess || rc != BQ_OK) {
            hipFree(cand);
            if (he != hipSuccess) (void)hipGetLastError();
            break;
        }
        p->place_ms[p->place_tried++] = t;
        if (t < best) {
            losers.push_back(p->panel);
            p->panel = cand;
            p->panel_alloc = cand;
            best = t;
        } else {
            losers.push_back(cand);
        }
    }
    // not released here: a large hipFree is followed by a transient in which the whole process streams slower (bq_ctx::held)
    for (void *l : losers) bq_ctx_hold(c, l, p->panel_bytes, p);
    return rc;
}

extern "C" int bq_ctx_set_placement_budget(bq_ctx *c, double min_ms, double max_ms, double expected_products) {
    BQ_ARG(c != nullptr, "ctx is NULL");
    BQ_ARG(expected_products >= 0.0, "expected_products must be >= 0");
    c->place_min_ms = min_ms;      // < 0: BQ_PLACE_BUDGET_MS / 200 ms
    c->place_max_ms = max_ms;
    c->place_products = expected_products;
    return BQ_OK;
}

extern "C" int bq_problem_placement(const bq_problem *p, int *tried, double *ms, int cap) {
    BQ_ARG(p != nullptr && tried != nullptr, "NULL argument");
    *tried = p->place_tried;
    for (int i = 0; ms != nullptr && i < cap && i < p->place_tried; ++i) ms[i] = p->place_ms[i];
    return BQ_OK;
}

extern "C" int bq_problem_create_kernel(bq_ctx *c, int structure, int64_t n, int64_t d, const double *X,
                                        const double *y, int kernel, double gamma, double coef0, int degree,
                                        double diag_add, const double *q, int storage, bq_problem **out) {
    BQ_ARG(c && X && q && out, "NULL argument");
    const bool no_rank_one = (structure & BQ_NO_RANK_ONE) != 0, full_panel = (structure & BQ_FULL_PANEL) != 0;
    const bool place = (structure & BQ_PLACE_PANEL) != 0;
    structure &= ~(BQ_NO_RANK_ONE | BQ_FULL_PANEL | BQ_PLACE_PANEL);
    BQ_ARG(structure == BQ_PLAIN || structure == BQ_SVC || structure == BQ_SVR, "structure");
    BQ_ARG(structure != BQ_SVC || y != nullptr, "labels required for BQ_SVC");
    BQ_ARG(kernel >= BQ_KERNEL_LINEAR && kernel <= BQ_KERNEL_LAPLACIAN, "kernel");
    BQ_ARG(n >= 2 && d >= 1, "n/d");
    BQ_ARG(storage == BQ_F64 || storage == BQ_F32 || storage == BQ_STREAM, "storage");
    BQ_ARG(storage != BQ_STREAM || kernel != BQ_KERNEL_LAPLACIAN, "the streamed mode is built for the inner-product kernels");
    BQ_ARG(kernel != BQ_KERNEL_POLY || degree > 0, "degree must be > 0");
    BQ_HIP(hipSetDevice(c->device));
    bq_problem *p = new bq_problem();
    p->ctx = c;
    c->refs += 1;
    p->structure = structure;
    p->storage = storage;
    p->kernel = kernel;
    p->add_one = structure != BQ_PLAIN && !no_rank_one;
    p->gamma = gamma;
    p->coef0 = coef0;
    p->degree = degree;
    p->diag_add = diag_add;
    p->d = d;
    p->streamed = storage == BQ_STREAM;
    // Gram panels are symmetric: store and stream only the tiles on/below the diagonal, unless the caller asked for whole rows
    // (BQ_FULL_PANEL).  Streamed problems take the same segment partition of the tile rows and form each lower-triangle tile once.
    p->symmetric = p->streamed || !full_panel;
    int rc = problem_layout(p, n, structure == BQ_SVR ? 2 * n : n);
    if (rc == BQ_OK) rc = problem_alloc_common(p, q);
    if (rc == BQ_OK && place) rc = place_panel(p, p->alloc_ms);
    if (rc != BQ_OK) {
        bq_problem_destroy(p);
        return rc;
    }
    auto fail = [&](hipError_t e) {
        bq_set_error("kernel problem setup failed: %s", hipGetErrorString(e));
        bq_problem_destroy(p);
        return BQ_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc(&p->X, sizeof(double) * n * d)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(p->X, X, sizeof(double) * n * d, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(e);
    if (structure == BQ_SVC) {
        if ((e = hipMalloc(&p->sgn, sizeof(double) * p->ld)) != hipSuccess) return fail(e);
        if ((e = hipMemsetAsync(p->sgn, 0, sizeof(double) * p->ld, c->stream)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(p->sgn, y, sizeof(double) * n, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(e);
    }
    if (p->streamed)
        rc = bq_stream_prepare(c, p->X, n, d, p->I0 * BQ_SYM_TILE, p->I1 * BQ_SYM_TILE, &p->stream_img);
    else
        rc = bq_launch_gram(c, p->X, n, d, p->r0, p->r1, kernel, gamma, coef0, degree, p->panel, storage, p->ld, p->symmetric);
    if (rc != BQ_OK) {
        bq_problem_destroy(p);
        return rc;
    }
    if (const int src = bq_ctx_sync(c)) {   // (bounded on a multi-rank context)
        bq_problem_destroy(p);
        return src;
    }
    *out = p;
    return BQ_OK;
}

extern "C" int bq_problem_dims(const bq_problem *p, int64_t *n_dual, int64_t *n_rows, int64_t *rb, int64_t *re) {
    BQ_ARG(p != nullptr, "problem is NULL");
    if (n_dual) *n_dual = p->N;
    if (n_rows) *n_rows = p->n;
    if (rb) *rb = p->r0;
    if (re) *re = p->r1;
    return BQ_OK;
}

extern "C" int bq_problem_layout(const bq_problem *p, int *packed, int *streamed, int64_t *panel_bytes) {
    BQ_ARG(p != nullptr, "problem is NULL");
    if (packed) *packed = p->symmetric ? 1 : 0;
    if (streamed) *streamed = p->streamed ? 1 : 0;
    if (panel_bytes) *panel_bytes = (int64_t)p->panel_bytes;
    return BQ_OK;
}

static int upload_padded(bq_problem *p, const double *host, double *dev, int64_t len, int64_t padded) {
    BQ_HIP(hipMemsetAsync(dev, 0, sizeof(double) * padded, p->ctx->stream));
    BQ_HIP(hipMemcpyAsync(dev, host, sizeof(double) * len, hipMemcpyHostToDevice, p->ctx->stream));
    return BQ_OK;
}

extern "C" int bq_problem_matvec(bq_problem *p, const double *v, double *out) {
    BQ_ARG(p && v && out, "NULL argument");
    BQ_HIP(hipSetDevice(p->ctx->device));
    BQ_TRY(upload_padded(p, v, p->va, p->N, p->ldN));
    BQ_TRY(bq_problem_apply(p, p->va, p->vb, nullptr));
    BQ_HIP(hipMemcpyAsync(out, p->vb, sizeof(double) * p->N, hipMemcpyDeviceToHost, p->ctx->stream));
    BQ_SYNC(p->ctx);
    return BQ_OK;
}

extern "C" int bq_problem_eval(bq_problem *p, const double *x, double *f, double *g) {
    BQ_ARG(p && x && f, "NULL argument");
    BQ_HIP(hipSetDevice(p->ctx->device));
    BQ_TRY(upload_padded(p, x, p->va, p->N, p->ldN));
    BQ_TRY(bq_problem_apply(p, p->va, p->vb, nullptr));
    BQ_TRY(bq_vec_eval_f(p, p->va, p->vb, g ? p->vb : nullptr, p->scal));
    BQ_HIP(hipMemcpyAsync(f, p->scal, sizeof(double), hipMemcpyDeviceToHost, p->ctx->stream));
    if (g) BQ_HIP(hipMemcpyAsync(g, p->vb, sizeof(double) * p->N, hipMemcpyDeviceToHost, p->ctx->stream));
    BQ_SYNC(p->ctx);
    return BQ_OK;
}

extern "C" int bq_problem_gram_matvec(bq_problem *p, const double *w, double *out) {
    BQ_ARG(p && w && out, "NULL argument");
    BQ_ARG(p->kernel >= 0, "not a kernel-structured problem");
    bq_ctx *c = p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    BQ_HIP(hipMemsetAsync(p->w, 0, sizeof(double) * p->ld, c->stream));
    BQ_HIP(hipMemcpyAsync(p->w, w, sizeof(double) * p->n, hipMemcpyHostToDevice, c->stream));
    BQ_TRY(bq_panel_product(p, false, p->w, nullptr));
    BQ_HIP(hipMemcpyAsync(out, p->s, sizeof(double) * p->n, hipMemcpyDeviceToHost, c->stream));
    BQ_SYNC(c);
    return BQ_OK;
}

extern "C" int bq_problem_panel_rows(bq_problem *p, int64_t row0, int64_t nrows, double *out) {
    BQ_ARG(p && out, "NULL argument");
    BQ_ARG(row0 >= p->r0 && row0 + nrows <= p->r1 && nrows >= 0, "rows outside this rank's block");
    BQ_ARG(!p->streamed, "a streamed problem keeps no panel");
    bq_ctx *c = p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    if (nrows == 0) return BQ_OK;
    if (p->symmetric) {
        // packed layout: row i holds its (I+1)*256 leading columns; the rest of the output row (strictly-upper tiles) is 0
        const size_t esz = p->storage == BQ_F64 ? 8 : 4;
        std::vector<unsigned char> tmp((size_t)p->n * esz);
        for (int64_t r = 0; r < nrows; ++r) {
            const int64_t i = row0 + r;
            const int64_t len = std::min(p->n, bq_sym_pitch(i / BQ_SYM_TILE));
            BQ_HIP(hipMemcpyAsync(tmp.data(), (const unsigned char *)p->panel + (size_t)bq_sym_addr(i, 0, p->I0) * esz,
                                  (size_t)len * esz, hipMemcpyDeviceToHost, c->stream));
            BQ_SYNC(c);
            double *o = out + r * p->n;
            for (int64_t j = 0; j < len; ++j)
                o[j] = p->storage == BQ_F64 ? ((const double *)tmp.data())[j] : (double)((const float *)tmp.data())[j];
            for (int64_t j = len; j < p->n; ++j) o[j] = 0.0;
        }
        return BQ_OK;
    }
    const int64_t lr = row0 - p->r0;
    if (p->storage == BQ_F64) {
        BQ_HIP(hipMemcpy2DAsync(out, p->n * 8, (const double *)p->panel + lr * p->ld, p->ld * 8, p->n * 8, nrows,
                                hipMemcpyDeviceToHost, c->stream));
        BQ_SYNC(c);
    } else {
        std::vector<float> tmp((size_t)nrows * p->n);
        BQ_HIP(hipMemcpy2DAsync(tmp.data(), p->n * 4, (const float *)p->panel + lr * p->ld, p->ld * 4, p->n * 4, nrows,
                                hipMemcpyDeviceToHost, c->stream));
        BQ_SYNC(c);
        for (size_t i = 0; i < tmp.size(); ++i) out[i] = (double)tmp[i];
    }
    return BQ_OK;
}

extern "C" int bq_problem_time_matvec(bq_problem *p, int reps, double *mean_ms) {
    BQ_ARG(p && mean_ms && reps > 0, "argument");
    bq_ctx *c = p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    hipEvent_t e0, e1;
    BQ_HIP(hipEventCreate(&e0));
    BQ_HIP(hipEventCreate(&e1));
    const bool prof = c->profiling;
    c->profiling = false;
    bq_seg_table tab;
    if (p->symmetric) bq_sym_seg_table(p, &tab);
    auto local = [&]() {
        if (p->streamed)
            return bq_stream_sym_product(c, p->stream_img, p->n, p->nb, tab, p->kernel, p->gamma, p->coef0, p->degree, p->add_one, p->w,
                                         p->s, 0, nullptr);
        return p->symmetric ? bq_launch_symv(c, p->panel, p->storage, p->add_one, p->nb, tab, p->w, p->slab, p->s, nullptr)
                            : bq_launch_gemv(c, p->panel, p->storage, p->add_one, p->r1 - p->r0, p->ld, p->w,
                                             p->s + p->r0, nullptr);
    };
    int rc = local();  // warm
    hipEventRecord(e0, c->stream);
    for (int i = 0; rc == BQ_OK && i < reps; ++i) rc = local();
    hipEventRecord(e1, c->stream);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    c->profiling = prof;
    BQ_TRY(rc);
    BQ_HIP(e);
    *mean_ms = (double)ms / reps;
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// HBM streaming probe: what a plain read-only sweep and a device-to-device copy reach on this GPU, so that a roofline
// fraction can be quoted against the measured ceiling as well as the nominal 8 TB/s
// ---------------------------------------------------------------------------------------------
typedef double probe_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void probe_read_kernel(const probe_d2 *__restrict__ src, int64_t n2,
                                                         double *__restrict__ sink) {
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const probe_d2 v = __builtin_nontemporal_load(src + i);
        acc += v.x + v.y;
    }
    if (acc == 12345.678) *sink = acc;   // keeps the loads alive, practically never true
}

extern "C" int bq_ctx_probe_bandwidth(bq_ctx *c, int64_t bytes, int reps, double *read_gbs, double *copy_gbs) {
    BQ_ARG(c && read_gbs && copy_gbs, "NULL argument");
    BQ_ARG(bytes >= (1 << 20) && reps >= 1, "bytes >= 1 MiB, reps >= 1");
    BQ_HIP(hipSetDevice(c->device));
    bytes &= ~(int64_t)4095;
    double *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a, (size_t)bytes);
    if (e == hipSuccess) e = hipMalloc(&b, (size_t)bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, (size_t)bytes, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, (size_t)bytes, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms_read = 0.f, ms_copy = 0.f;
    if (e == hipSuccess) {
        const int64_t n2 = bytes / 16;
        const unsigned grid = (unsigned)(c->num_cu * 16);
        probe_read_kernel<<<grid, 256, 0, c->stream>>>((const probe_d2 *)a, n2, b);   // warm
        hipEventRecord(e0, c->stream);
        for (int r = 0; r < reps; ++r) probe_read_kernel<<<grid, 256, 0, c->stream>>>((const probe_d2 *)a, n2, b);
        hipEventRecord(e1, c->stream);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_read, e0, e1);
    }
    if (e == hipSuccess) {
        hipMemcpyAsync(b, a, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream);   // warm
        hipEventRecord(e0, c->stream);
        for (int r = 0; r < reps; ++r) hipMemcpyAsync(b, a, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream);
        hipEventRecord(e1, c->stream);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms_copy, e0, e1);
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (a) hipFree(a);
    if (b) hipFree(b);
    if (e != hipSuccess) {
        bq_set_error("bandwidth probe failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    *read_gbs = (double)bytes * reps / (ms_read * 1e-3) / 1e9;
    *copy_gbs = 2.0 * (double)bytes * reps / (ms_copy * 1e-3) / 1e9;   // bytes read + bytes written
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// fp64 MFMA probe: what back-to-back v_mfma_f64_16x16x4_f64 with register operands sustain on this GPU at the clock it
// holds under that load — the yardstick beside the nominal 78.6 TFLOP/s for the Cholesky's roofline fraction
// ---------------------------------------------------------------------------------------------
typedef double probe_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void probe_mfma_kernel(double *sink, int iters, double seed) {
    probe_d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (probe_d4){0.0, 0.0, 0.0, 0.0};
    double a = seed + 1e-3 * (double)(threadIdx.x & 15), b = 1.0 - 1e-3 * (double)(threadIdx.x >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a = -a;   // keeps the sums bounded without touching the MFMA stream's shape
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678) *sink = s;   // keeps the MFMAs alive, practically never true
}

extern "C" int bq_ctx_probe_mfma_f64(bq_ctx *c, double seconds, double *tflops) {
    BQ_ARG(c && tflops, "NULL argument");
    BQ_ARG(seconds > 0.0 && seconds <= 10.0, "seconds in (0, 10]");
    BQ_HIP(hipSetDevice(c->device));
    double *sink = nullptr;
    BQ_HIP(hipMalloc(&sink, sizeof(double)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const int iters = 4096;
    const unsigned grid = (unsigned)(c->num_cu * 2);           // 2 workgroups of 4 waves per CU: two waves per SIMD
    const double flop_per_launch = (double)grid * 4.0 * iters * 16.0 * 2048.0;   // waves x iterations x MFMAs x 2*16*16*4
    float ms = 0.f;
    int launches = 0;
    if (e == hipSuccess) {
        // warm up for half the requested time so that the clock has settled under the load, then time the other half
        probe_mfma_kernel<<<grid, 256, 0, c->stream>>>(sink, iters, 0.5);
        hipEventRecord(e0, c->stream);
        probe_mfma_kernel<<<grid, 256, 0, c->stream>>>(sink, iters, 0.5);
        hipEventRecord(e1, c->stream);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        const int per_half = ms > 0.f ? std::max(1, (int)(seconds * 500.0 / ms)) : 1;
        for (int i = 0; e == hipSuccess && i < per_half; ++i) probe_mfma_kernel<<<grid, 256, 0, c->stream>>>(sink, iters, 0.5);
        if (e == hipSuccess) {
            hipEventRecord(e0, c->stream);
            for (int i = 0; i < per_half; ++i) probe_mfma_kernel<<<grid, 256, 0, c->stream>>>(sink, iters, 0.5);
            hipEventRecord(e1, c->stream);
            e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            launches = per_half;
        }
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(sink);
    if (e != hipSuccess || ms <= 0.f) {
        bq_set_error("MFMA probe failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    *tflops = flop_per_launch * launches / (ms * 1e-3) / 1e12;
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// Stream occupancy probe (tests of the collective watchdog): ONE lane spins on the constant-rate wall clock for the given
// time and then returns — it always terminates by itself, whatever the host does
// ---------------------------------------------------------------------------------------------
__global__ void probe_stall_kernel(long long ticks, long long *sink) {
    const long long t0 = wall_clock64();
    long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (ticks < 0) *sink = t;   // never: keeps the loop
}

extern "C" int bq_ctx_probe_stall(bq_ctx *c, double milliseconds, int behind_collective) {
    BQ_ARG(c != nullptr, "ctx is NULL");
    BQ_ARG(milliseconds > 0.0 && milliseconds <= 5000.0, "milliseconds in (0, 5000]");
    BQ_HIP(hipSetDevice(c->device));
    int rate_khz = 0;
    BQ_HIP(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, c->device));
    if (rate_khz <= 0) rate_khz = 100000;   // 100 MHz on every part this library is built for
    double *one = nullptr;
    if (behind_collective) {
        BQ_HIP(hipMalloc(&one, sizeof(double)));
        BQ_HIP(hipMemsetAsync(one, 0, sizeof(double), c->stream));
    }
    probe_stall_kernel<<<1, 1, 0, c->stream>>>((long long)(milliseconds * (double)rate_khz), nullptr);
    int rc = hipGetLastError() == hipSuccess ? BQ_OK : BQ_ERR_HIP;
    if (rc == BQ_OK && behind_collective) rc = bq_exchange_sum(c, one, 1);
    if (rc == BQ_OK) rc = bq_ctx_sync(c);
    if (one) hipFree(one);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// solvers
// ---------------------------------------------------------------------------------------------
static int alloc_vec(bq_solver *s, double **v) {
    BQ_HIP(hipMalloc(v, sizeof(double) * s->ldN));
    BQ_HIP(hipMemsetAsync(*v, 0, sizeof(double) * s->ldN, s->p->ctx->stream));
    return BQ_OK;
}

extern "C" int bq_solver_destroy(bq_solver *s) {
    if (s == nullptr) return BQ_OK;
    hipSetDevice(s->p->ctx->device);
    (void)bq_ctx_sync(s->p->ctx);   // may sit behind a collective: bounded when a collective timeout is set
    for (void *ptr : {(void *)s->x, (void *)s->g, (void *)s->d, (void *)s->Qd, (void *)s->lb, (void *)s->ub,
                      (void *)s->lp, (void *)s->lm, (void *)s->rhs, (void *)s->hd, (void *)s->dlp, (void *)s->dlm,
                      (void *)s->mL, (void *)s->mU, (void *)s->partials, (void *)s->sc, (void *)s->stats})
        if (ptr) hipFree(ptr);
    if (s->chol) bq_chol_ws_destroy(s->chol);
    if (s->flag_host) {
        hipHostFree(s->flag_host);
        hipEventDestroy(s->flag_event);
    }
    bq_as_free(s);
    delete s->resume;
    if (s->al) {
        bq_al_vecs &V = s->al->V;   // x, g, step (= s->d) and Qx (= s->Qd) are owned by the common slots above
        for (void *ptr : {(void *)V.xe, (void *)V.chk, (void *)V.s1, (void *)V.s2, (void *)V.s3, (void *)V.a, (void *)V.llb, (void *)V.lub,
                          (void *)V.lr_sched, (void *)V.mom_sched})
            if (ptr) hipFree(ptr);
        delete s->al;
    }
    bq_problem *p = s->p;
    delete s;
    // the solve the placement choice was made for is over: what it held back to keep this solve undisturbed can go now (bq_ctx::held)
    bq_ctx_release_held(p->ctx, p);
    bq_problem_unref(p);
    return BQ_OK;
}

extern "C" int bq_solver_create(bq_problem *p, int kind, const double *lb, const double *ub, const double *x0,
                                double eps, int64_t max_iter, double fw_t, bq_solver **out) {
    BQ_ARG(p && ub && out, "NULL argument");
    BQ_ARG(kind == BQ_PG || kind == BQ_FW || kind == BQ_AS || kind == BQ_IP || kind == BQ_AS_CG, "solver kind");
    const bool as_cg = kind == BQ_AS_CG;   // ActiveSet on products only: none of the dense-factor restrictions below
    if (as_cg) kind = BQ_AS;
    BQ_ARG(max_iter > 0, "max_iter must be > 0");             // optiml/opti/_base.py:73-74
    BQ_ARG(fw_t >= 0.0 && fw_t < 1.0, "t has to lie in [0, 1)");  // frank_wolfe.py:84-85
    if ((kind == BQ_IP || kind == BQ_AS) && !as_cg && p->ctx->world > 1) {
        bq_set_error("InteriorPoint/ActiveSet factorise the whole Hessian: use a single-rank context (replicas only)");
        return BQ_ERR_BADARG;
    }
    if ((kind == BQ_IP || kind == BQ_AS) && !as_cg && p->streamed) {
        bq_set_error("InteriorPoint/ActiveSet assemble their systems from the resident panel: not available in the streamed mode");
        return BQ_ERR_BADARG;
    }
    if ((kind == BQ_IP || kind == BQ_AS) && p->structure != BQ_PLAIN && !p->add_one) {
        bq_set_error("InteriorPoint/ActiveSet are not built for the BQ_NO_RANK_ONE duals (svm/_base.py:621-624)");
        return BQ_ERR_BADARG;
    }
    bq_ctx *c = p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    bq_solver *s = new bq_solver();
    s->p = p;
    p->refs += 1;
    s->kind = kind;
    s->as_cg = as_cg;
    s->N = p->N;
    s->ldN = p->ldN;
    s->nblk = p->ldN / BQ_VEC_TILE;
    int rc = BQ_OK;
    for (double **v : {&s->x, &s->g, &s->d, &s->Qd, &s->lb, &s->ub})
        if (rc == BQ_OK) rc = alloc_vec(s, v);
    if (rc == BQ_OK && kind == BQ_IP)
        for (double **v : {&s->lp, &s->lm, &s->rhs, &s->hd, &s->dlp, &s->dlm})
            if (rc == BQ_OK) rc = alloc_vec(s, v);
    if (rc != BQ_OK) {
        bq_solver_destroy(s);
        return rc;
    }
    std::vector<double> h((size_t)s->N);
    auto up = [&](double *dev, const double *src) {
        return hipMemcpyAsync(dev, src, sizeof(double) * s->N, hipMemcpyHostToDevice, c->stream);
    };
    hipError_t e = up(s->ub, ub);
    if (e == hipSuccess && lb) e = up(s->lb, lb);
    if (e == hipSuccess) {
        if (x0) {
            e = up(s->x, x0);
        } else {
            for (int64_t i = 0; i < s->N; ++i) h[i] = ((lb ? lb[i] : 0.0) + ub[i]) / 2;  // constrained/_base.py:65
            e = up(s->x, h.data());
        }
    }
    if (e == hipSuccess) e = hipMalloc(&s->partials, sizeof(double) * BQ_MAX_PARTIAL_Q * s->nblk);
    if (e == hipSuccess) e = hipMalloc(&s->sc, sizeof(bq_scal));
    if (e == hipSuccess) {
        memset(&s->host, 0, sizeof(bq_scal));
        s->host.max_iter = max_iter;
        s->host.eps = eps;
        s->host.fw_t = fw_t;
        s->host.status = BQ_STATUS_UNKNOWN;
        s->host.best_lb = -INFINITY;
        s->host.f = NAN;
        e = hipMemcpyAsync(s->sc, &s->host, sizeof(bq_scal), hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) {   // may sit behind a collective of an earlier solve: the bounded wait
        const int src = bq_ctx_sync(c);
        if (src != BQ_OK) {
            bq_solver_destroy(s);
            return src;
        }
    }
    if (e != hipSuccess) {
        bq_set_error("solver setup failed: %s", hipGetErrorString(e));
        bq_solver_destroy(s);
        return BQ_ERR_HIP;
    }
    if ((kind == BQ_IP || kind == BQ_AS) && !as_cg) {
        // InteriorPoint on the SVR structure factorises the reduced n x n system (bq_ip.hip)
        rc = bq_chol_ws_create(c, (kind == BQ_IP && p->structure == BQ_SVR && bq_ip_svr_reduced()) ? p->n : s->N,
                               &s->chol);
        if (rc != BQ_OK) {
            bq_solver_destroy(s);
            return rc;
        }
    }
    *out = s;
    return BQ_OK;
}

extern "C" int bq_solver_set_inner(bq_solver *s, double rtol, int64_t max_iter) {
    BQ_ARG(s, "NULL solver");
    BQ_ARG(s->kind == BQ_AS && s->as_cg, "only the conjugate-gradient ActiveSet (BQ_AS_CG) has an inner iteration");
    BQ_ARG(rtol > 0.0 && rtol < 1.0, "inner tolerance must lie in (0, 1)");
    BQ_ARG(max_iter >= 0, "inner iteration cap must be >= 0 (0: 2 |A| + 50)");
    s->inner_rtol = rtol;
    s->inner_max = max_iter;
    return BQ_OK;
}

extern "C" int bq_solver_inner_iters(bq_solver *s, int64_t *total) {
    BQ_ARG(s && total, "NULL argument");
    *total = s->kind == BQ_AS ? bq_as_inner_iters(s) : 0;
    return BQ_OK;
}

extern "C" int bq_solver_counter(bq_solver *s, int which, int64_t *value) {
    BQ_ARG(s && value, "NULL argument");
    BQ_ARG(which >= BQ_COUNT_INNER && which <= BQ_COUNT_NO_PRODUCT, "which: BQ_COUNT_*");
    *value = s->kind == BQ_AS ? bq_as_counter(s, which) : 0;
    return BQ_OK;
}

extern "C" int bq_al_solver_create(bq_problem *p, const bq_al_params *prm, const double *a_eq, const double *lb,
                                   const double *ub, const double *x0, const double *dual0, bq_solver **out) {
    BQ_ARG(p && prm && x0 && out, "NULL argument");
    BQ_ARG(prm->rule >= BQ_RULE_SGD && prm->rule <= BQ_RULE_RMSPROP, "update rule");
    BQ_ARG(prm->momentum_type >= BQ_MOM_NONE && prm->momentum_type <= BQ_MOM_NESTEROV, "unknown momentum type");
    BQ_ARG(prm->momentum_type == BQ_MOM_NONE || (prm->rule != BQ_RULE_ADAGRAD && prm->rule != BQ_RULE_ADADELTA),
           "AdaGrad / AdaDelta take no momentum");
    BQ_ARG(prm->step_size > 0.0, "step_size must be > 0");                       // stochastic/_base.py:86-87
    BQ_ARG(prm->momentum >= 0.0 && prm->momentum < 1.0, "momentum must be between 0 and 1");   // :240-241
    BQ_ARG(prm->epochs > 0, "max_iter must be > 0");                             // optiml/opti/_base.py:73-74
    BQ_ARG(prm->rho > 0.0, "rho must be must > 0");                              // constrained/_base.py:276-277
    BQ_ARG(prm->offset > 0.0 || prm->rule == BQ_RULE_SGD, "offset must be > 0");
    BQ_ARG(prm->beta1 >= 0.0 && prm->beta1 < 1.0, "beta1 has to lie in [0, 1)");
    BQ_ARG(prm->beta2 >= 0.0 && prm->beta2 < 1.0, "beta2 has to lie in [0, 1)");
    BQ_ARG(prm->decay >= 0.0 && prm->decay < 1.0, "decay has to lie in [0, 1)");
    bq_ctx *c = p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    bq_solver *s = new bq_solver();
    s->p = p;
    p->refs += 1;
    s->kind = BQ_AL;
    s->N = p->N;
    s->ldN = p->ldN;
    s->nblk = p->ldN / BQ_VEC_TILE;
    s->al = new bq_al_state();
    s->al->prm = *prm;
    bq_al_vecs &V = s->al->V;
    memset(&V, 0, sizeof(V));
    int rc = BQ_OK;
    for (double **v : {&s->x, &s->g, &s->d, &s->Qd, &V.xe, &V.s1})
        if (rc == BQ_OK) rc = alloc_vec(s, v);
    if (rc == BQ_OK && hipMalloc(&V.chk, sizeof(double) * 3 * s->ldN) != hipSuccess) {
        bq_set_error("cannot allocate the solver's vectors");
        rc = BQ_ERR_NOMEM;
    }
    if (rc == BQ_OK) BQ_HIP(hipMemsetAsync(V.chk, 0, sizeof(double) * 3 * s->ldN, c->stream));
    const bool two = prm->rule == BQ_RULE_ADAM || prm->rule == BQ_RULE_AMSGRAD || prm->rule == BQ_RULE_ADAMAX ||
                     prm->rule == BQ_RULE_ADADELTA;
    if (rc == BQ_OK && two) rc = alloc_vec(s, &V.s2);
    if (rc == BQ_OK && prm->rule == BQ_RULE_AMSGRAD) rc = alloc_vec(s, &V.s3);
    if (rc == BQ_OK && a_eq) rc = alloc_vec(s, &V.a);
    if (rc == BQ_OK && lb) rc = alloc_vec(s, &s->lb);
    if (rc == BQ_OK && lb) rc = alloc_vec(s, &V.llb);
    if (rc == BQ_OK && ub) rc = alloc_vec(s, &s->ub);
    if (rc == BQ_OK && ub) rc = alloc_vec(s, &V.lub);
    if (rc != BQ_OK) {
        bq_solver_destroy(s);
        return rc;
    }
    V.x = s->x;
    V.g = s->g;
    V.step = s->d;
    V.Qx = s->Qd;
    V.q = p->q;
    V.lb = lb ? s->lb : nullptr;
    V.ub = ub ? s->ub : nullptr;
    auto up = [&](double *dev, const double *src) {
        return hipMemcpyAsync(dev, src, sizeof(double) * s->N, hipMemcpyHostToDevice, c->stream);
    };
    hipError_t e = up(s->x, x0);
    if (e == hipSuccess) e = up(V.xe, x0);
    if (e == hipSuccess && a_eq) e = up(V.a, a_eq);
    if (e == hipSuccess && lb) e = up(s->lb, lb);
    if (e == hipSuccess && ub) e = up(s->ub, ub);
    std::vector<double> ones;
    if (e == hipSuccess && prm->rule == BQ_RULE_RMSPROP) {   // rmsprop.py: moving_mean_squared starts at ones
        ones.assign((size_t)s->N, 1.0);
        e = up(V.s1, ones.data());
    }
    memset(&s->host, 0, sizeof(bq_scal));
    if (dual0) {
        const double *d0 = dual0;
        if (a_eq) s->host.al_mu = *d0++;
        if (e == hipSuccess && lb) {
            e = up(V.llb, d0);
            d0 += s->N;
        }
        if (e == hipSuccess && ub) e = up(V.lub, d0);
    }
    if (e == hipSuccess) e = hipMalloc(&s->partials, sizeof(double) * BQ_MAX_PARTIAL_Q * s->nblk);
    if (e == hipSuccess) e = hipMalloc(&s->sc, sizeof(bq_scal));
    if (e == hipSuccess) {
        s->host.max_iter = prm->epochs;
        s->host.eps = prm->tol;
        s->host.status = BQ_STATUS_UNKNOWN;
        s->host.f = NAN;
        s->host.al_pf = NAN;
        e = hipMemcpyAsync(s->sc, &s->host, sizeof(bq_scal), hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) {   // may sit behind a collective of an earlier solve: the bounded wait
        const int src = bq_ctx_sync(c);
        if (src != BQ_OK) {
            bq_solver_destroy(s);
            return src;
        }
    }
    if (e != hipSuccess) {
        bq_set_error("solver setup failed: %s", hipGetErrorString(e));
        bq_solver_destroy(s);
        return BQ_ERR_HIP;
    }
    *out = s;
    return BQ_OK;
}

extern "C" int bq_al_solver_set_schedules(bq_solver *s, const double *step_sizes, const double *momenta, int64_t count) {
    BQ_ARG(s != nullptr && s->al != nullptr, "not an augmented-Lagrangian solver");
    BQ_ARG(count >= 1 && (step_sizes || momenta), "count >= 1 and at least one schedule");
    BQ_ARG(!s->initialised && s->host.iter == 0, "schedules are set before the first run");
    bq_ctx *c = s->p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    for (int64_t i = 0; i < count; ++i) {
        BQ_ARG(!step_sizes || step_sizes[i] > 0.0, "step sizes must be > 0");
        BQ_ARG(!momenta || (momenta[i] >= 0.0 && momenta[i] < 1.0), "momentum must be between 0 and 1");
    }
    bq_al_vecs &V = s->al->V;
    for (int k = 0; k < 2; ++k) {
        const double *src = k == 0 ? step_sizes : momenta;
        if (!src) continue;
        double *dev = nullptr;
        BQ_HIP(hipMalloc(&dev, sizeof(double) * count));
        BQ_HIP(hipMemcpy(dev, src, sizeof(double) * count, hipMemcpyHostToDevice));
        if (k == 0) {
            if (V.lr_sched) hipFree((void *)V.lr_sched);
            V.lr_sched = dev;
        } else {
            if (V.mom_sched) hipFree((void *)V.mom_sched);
            V.mom_sched = dev;
        }
    }
    V.sched_len = count;
    return BQ_OK;
}

extern "C" int bq_al_solver_dual_size(const bq_solver *s, int64_t *n_dual) {
    BQ_ARG(s && n_dual, "NULL argument");
    BQ_ARG(s->al != nullptr, "not an augmented-Lagrangian solver");
    const bq_al_vecs &V = s->al->V;
    *n_dual = (V.a ? 1 : 0) + (V.llb ? s->N : 0) + (V.lub ? s->N : 0);
    return BQ_OK;
}

static int solver_first(bq_solver *s) {
    if (s->kind == BQ_AL) return BQ_OK;   // nothing to prepare: every iteration evaluates Q x afresh
    bq_solver::resume_t *r = s->resume;
    const int have = r ? r->have : 0;
    hipStream_t st = s->p->ctx->stream;
    auto up = [&](double *dev, const std::vector<double> &src) {
        return hipMemcpyAsync(dev, src.data(), sizeof(double) * s->N, hipMemcpyHostToDevice, st);
    };
    switch (s->kind) {
        case BQ_PG:
        case BQ_FW:
            // g given: the start product is not needed — (x, g) is the whole state of these two (bq_solver_set_state)
            if (have & BQ_STATE_G) BQ_HIP(up(s->g, r->g));
            else BQ_TRY(bq_pgfw_start(s));
            break;
        case BQ_IP:
            // interior_point.py:180-186 where something is missing; what was given then replaces what the start formed
            if ((have & (BQ_STATE_G | BQ_STATE_MULT)) != (BQ_STATE_G | BQ_STATE_MULT)) BQ_TRY(bq_ip_start(s));
            if (have & BQ_STATE_G) BQ_HIP(up(s->g, r->g));
            if (have & BQ_STATE_MULT) {
                BQ_HIP(up(s->lp, r->lp));
                BQ_HIP(up(s->lm, r->lm));
            }
            break;
        default:
            BQ_TRY(bq_as_start(s));   // takes the masks from s->resume itself (they are allocated there)
            if (have & BQ_STATE_G) BQ_HIP(up(s->g, r->g));
            break;
    }
    if (r) {
        BQ_SYNC(s->p->ctx);   // the uploads read the vectors about to be released
        delete r;
        s->resume = nullptr;
    }
    return BQ_OK;
}

static int solver_iterate(bq_solver *s) {
    if (s->kind == BQ_AL) return bq_al_iterate(s);
    switch (s->kind) {
        case BQ_PG:
        case BQ_FW:
            return bq_pgfw_iterate(s);
        case BQ_IP:
            return bq_ip_iterate(s);
        default:
            return bq_as_iterate(s);
    }
}

extern "C" int bq_solver_run(bq_solver *s, int64_t max_steps, bq_iter_stat *stats, int64_t stats_cap, int64_t *n_stats,
                             int *status) {
    BQ_ARG(s && n_stats && status, "NULL argument");
    BQ_ARG(max_steps > 0, "max_steps must be > 0");
    BQ_ARG(stats == nullptr || stats_cap >= max_steps, "stats capacity must be >= max_steps");
    bq_ctx *c = s->p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    *n_stats = 0;
    *status = s->host.status;
    if (s->host.done) return BQ_OK;
    if (s->stats_cap < max_steps) {
        if (s->stats) BQ_HIP(hipFree(s->stats));
        s->stats = nullptr;
        BQ_HIP(hipMalloc(&s->stats, sizeof(bq_iter_stat) * max_steps));
        s->stats_cap = max_steps;
    }
    if (s->flag_host == nullptr) {   // pinned flag + event of the lagged done-flag polling (best effort)
        if (hipHostMalloc((void **)&s->flag_host, sizeof(int), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s->flag_event, hipEventDisableTiming) != hipSuccess) {
            if (s->flag_host) hipHostFree(s->flag_host);
            s->flag_host = nullptr;
            (void)hipGetLastError();
        } else {
            *s->flag_host = 0;
        }
    }
    if (s->al) s->al->w_ready = false;   // p->w belongs to the problem: anything may have used it since the last run
    const long long base = s->host.iter;
    long long hdr[2] = {base, (long long)max_steps};
    BQ_HIP(hipMemcpyAsync(&s->sc->stat_base, hdr, sizeof(hdr), hipMemcpyHostToDevice, c->stream));
    if (!s->initialised) {
        BQ_TRY(solver_first(s));
        s->initialised = true;
    }
    // How often the host looks at the device `done` flag: kernels early-exit on the flag, so a late look only costs a few
    // no-op launches (~20 us per skipped iteration).  The factorising solvers are host-enqueued O(n^3) work per iteration
    // and look every time; PG/FW look about every 20 ms of estimated panel streaming time.  A look is a 4-byte copy to the
    // host plus an event on the stream — not free between two short kernels: at 2 ms (round 3) BASELINE config 2 ran 0.3035 ms
    // per iteration, at 20 ms 0.286 ms (profiles/r04/share_gaps_events.txt).
    int64_t poll = 1;
    if (s->kind == BQ_PG || s->kind == BQ_FW || s->kind == BQ_AL) {
        // The interval must be the SAME on every rank (each iteration contains a collective: ranks that stopped enqueueing at
        // different iterations would leave the others inside it), so it is computed from the mean share n / world, not from
        // this rank's own rows.
        const double esz = s->p->storage == BQ_F64 ? 8.0 : 4.0;
        const double rows = (double)s->p->n / (double)c->world;
        const double iter_s = rows * (double)s->p->n * esz * (s->p->symmetric ? 0.5 : 1.0) / 5.0e12 + 30e-6;
        poll = (int64_t)(20.0e-3 / iter_s);
        poll = poll < 1 ? 1 : (poll > 64 ? 64 : poll);
    }
    // The product-bound solvers look at the flag with a LAG of one chunk: the copy of the flag is followed by an event,
    // the next chunk is enqueued at once, and only then does the host wait for that event — the device never runs dry
    // while the host decides (the extra chunk early-exits on the flag).  The factorising solvers enqueue O(n^3) work per
    // iteration from the host and check before every iteration.
    const bool lagged = poll > 0 && (s->kind == BQ_PG || s->kind == BQ_FW || s->kind == BQ_AL) && s->flag_host != nullptr;
    bool pending = false;
    for (int64_t k = 0; k < max_steps; ++k) {
        BQ_TRY(solver_iterate(s));
        if (s->kind == BQ_AS) {
            // ActiveSet looks at the device at the top of every iteration anyway (bq_as_iterate: the order of the restricted
            // system comes from there) and leaves s->host current: a second look here was one more stream drain per iteration
            if (s->host.done) break;
            continue;
        }
        if ((k + 1) % poll == 0 && k + 1 < max_steps) {
            if (lagged) {
                if (pending) {
                    BQ_TRY(bq_ctx_event_sync(c, s->flag_event));
                    if (*s->flag_host) break;
                }
                BQ_HIP(hipMemcpyAsync(s->flag_host, &s->sc->done, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                BQ_HIP(hipEventRecord(s->flag_event, c->stream));
                pending = true;
            } else {
                BQ_HIP(hipMemcpyAsync(&s->host, s->sc, sizeof(bq_scal), hipMemcpyDeviceToHost, c->stream));
                BQ_SYNC(c);
                if (s->host.done) break;
            }
        }
    }
    if (s->al) BQ_TRY(bq_al_flush(s));
    BQ_HIP(hipMemcpyAsync(&s->host, s->sc, sizeof(bq_scal), hipMemcpyDeviceToHost, c->stream));
    BQ_SYNC(c);
    if (s->host.status < 0) {  // a kernel flagged a numerical failure (codes mirror BQ_ERR_*)
        bq_set_error(s->host.status == BQ_ERR_NOT_PD ? "Cholesky met a non-positive pivot"
                                                     : "non-finite values in the solver state");
        return s->host.status;
    }
    int64_t rows = s->host.iter - base + (s->host.done ? 1 : 0);
    if (rows > max_steps) rows = max_steps;
    if (rows < 0) rows = 0;
    if (stats && rows > 0) {
        BQ_HIP(hipMemcpyAsync(stats, s->stats, sizeof(bq_iter_stat) * rows, hipMemcpyDeviceToHost, c->stream));
        BQ_SYNC(c);
    }
    *n_stats = rows;
    *status = s->host.status;
    return BQ_OK;
}

extern "C" int bq_solver_state(const bq_solver *s, int64_t *iter, int *status, double *f_x) {
    BQ_ARG(s != nullptr, "solver is NULL");
    if (iter) *iter = s->host.iter;
    if (status) *status = s->host.status;
    if (f_x) *f_x = s->host.f;
    return BQ_OK;
}

extern "C" int bq_solver_get(bq_solver *s, int what, double *out) {
    BQ_ARG(s && out, "NULL argument");
    bq_ctx *c = s->p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    const double *src = nullptr;
    if (what == BQ_GET_DUAL) {
        BQ_ARG(s->al != nullptr, "multipliers exist for the augmented-Lagrangian solver only");
        const bq_al_vecs &V = s->al->V;
        double *o = out;
        if (V.a) *o++ = s->host.al_mu;
        for (const double *v : {(const double *)V.llb, (const double *)V.lub})
            if (v) {
                BQ_HIP(hipMemcpyAsync(o, v, sizeof(double) * s->N, hipMemcpyDeviceToHost, c->stream));
                o += s->N;
            }
        BQ_SYNC(c);
        return BQ_OK;
    }
    switch (what) {
        case BQ_GET_X: src = s->kind == BQ_AS ? bq_as_view(s, BQ_GET_X) : (s->al ? s->al->V.xe : s->x); break;
        case BQ_GET_G: src = s->kind == BQ_AS ? bq_as_view(s, BQ_GET_G) : s->g; break;
        case BQ_GET_X_NOW: src = s->x; break;
        case BQ_GET_G_NOW: src = s->g; break;
        case BQ_GET_D: src = s->d; break;
        case BQ_GET_LP: src = s->lp; break;
        case BQ_GET_LM: src = s->lm; break;
        default: break;
    }
    if (what == BQ_GET_MASK_L || what == BQ_GET_MASK_U) {
        const unsigned char *m = what == BQ_GET_MASK_L ? s->mL : s->mU;
        BQ_ARG(m != nullptr, "masks exist for ActiveSet only");
        std::vector<unsigned char> tmp((size_t)s->N);
        BQ_HIP(hipMemcpyAsync(tmp.data(), m, (size_t)s->N, hipMemcpyDeviceToHost, c->stream));
        BQ_SYNC(c);
        for (int64_t i = 0; i < s->N; ++i) out[i] = tmp[i] ? 1.0 : 0.0;
        return BQ_OK;
    }
    BQ_ARG(src != nullptr, "vector not available for this solver");
    BQ_HIP(hipMemcpyAsync(out, src, sizeof(double) * s->N, hipMemcpyDeviceToHost, c->stream));
    BQ_SYNC(c);
    return BQ_OK;
}

// ---- checkpoint / resume (bcqp.h) -----------------------------------------------------------------------------
// v + t dv with the device's rounding: one rounded product, one sum (pgfw_update_kernel / ip_update_kernel use __dmul_rn).
#pragma clang fp contract(off)
static void state_apply_step(std::vector<double> &v, const std::vector<double> &dv, double t) {
    for (size_t i = 0; i < v.size(); ++i) {
        const double prod = t * dv[i];
        v[i] = v[i] + prod;
    }
}

extern "C" int bq_solver_get_state(bq_solver *s, bq_solver_snapshot *out) {
    BQ_ARG(s && out, "NULL argument");
    BQ_ARG(s->kind != BQ_AL, "the augmented-Lagrangian solver keeps its state in x and BQ_GET_DUAL");
    bq_ctx *c = s->p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    const size_t N = (size_t)s->N;
    out->iter = s->host.iter;
    out->kind = s->as_cg ? BQ_AS_CG : s->kind;
    out->f = s->host.f;
    out->best_lb = s->host.best_lb;
    out->have = 0;
    // a decided step waits on the device until the next iteration's first kernel applies it — unless the run has ended
    // (the deciding kernels return before they form one) or has not begun
    const bool pending = s->started && !s->host.done && s->kind != BQ_AS;
    const double t = s->kind == BQ_IP ? s->host.step : s->host.t;
    std::vector<double> v(N), dv(N);
    auto down = [&](std::vector<double> &dst, const double *src) {
        return hipMemcpyAsync(dst.data(), src, sizeof(double) * N, hipMemcpyDeviceToHost, c->stream);
    };
    auto fetch = [&](double *dst, const double *vec, const double *dvec) -> int {
        BQ_HIP(down(v, vec));
        if (pending) BQ_HIP(down(dv, dvec));
        BQ_SYNC(c);
        if (pending) state_apply_step(v, dv, t);
        memcpy(dst, v.data(), sizeof(double) * N);
        return BQ_OK;
    };
    const bool pgfw = s->kind == BQ_PG || s->kind == BQ_FW;
    if (out->x) {
        BQ_TRY(fetch(out->x, s->x, s->d));   // d: PG / FW direction, IP dx (ip_vecs)
        out->have |= BQ_STATE_X;
    }
    // the gradient exists once the start-up has run (PG / FW keep it current: g + t Qd; IP and ActiveSet keep the reference's
    // stale self.g_x: interior_point.py:180, active_set.py:157)
    if (out->g && s->initialised) {
        if (pgfw) {
            BQ_TRY(fetch(out->g, s->g, s->Qd));
        } else {
            BQ_HIP(down(v, s->g));
            BQ_SYNC(c);
            memcpy(out->g, v.data(), sizeof(double) * N);
        }
        out->have |= BQ_STATE_G;
    }
    if (s->kind == BQ_IP && out->lp && out->lm && s->initialised) {
        BQ_TRY(fetch(out->lp, s->lp, s->dlp));
        BQ_TRY(fetch(out->lm, s->lm, s->dlm));
        out->have |= BQ_STATE_MULT;
    }
    if (s->kind == BQ_AS && out->mask_l && out->mask_u && s->mL && s->mU) {
        BQ_TRY(bq_solver_get(s, BQ_GET_MASK_L, out->mask_l));
        BQ_TRY(bq_solver_get(s, BQ_GET_MASK_U, out->mask_u));
        out->have |= BQ_STATE_MASKS;
    }
    return BQ_OK;
}

extern "C" int bq_solver_set_state(bq_solver *s, const bq_solver_snapshot *in) {
    BQ_ARG(s && in, "NULL argument");
    BQ_ARG(s->kind != BQ_AL, "the augmented-Lagrangian solver is resumed through x0 / dual0 of bq_al_solver_create");
    BQ_ARG(!s->initialised, "the state can only be set before the solver's first run");
    BQ_ARG(in->kind == -1 || in->kind == (s->as_cg ? BQ_AS_CG : s->kind), "the state was taken from another kind of solver");
    BQ_ARG((in->have & BQ_STATE_X) && in->x, "a state holds x at least");
    BQ_ARG(in->iter >= 0, "iter must be >= 0");
    BQ_ARG(!(in->have & BQ_STATE_G) || in->g, "BQ_STATE_G without g");
    BQ_ARG(!(in->have & BQ_STATE_MULT) || (s->kind == BQ_IP && in->lp && in->lm), "BQ_STATE_MULT: lp and lm, InteriorPoint only");
    BQ_ARG(!(in->have & BQ_STATE_MASKS) || (s->kind == BQ_AS && in->mask_l && in->mask_u), "BQ_STATE_MASKS: mask_l and mask_u, ActiveSet only");
    bq_ctx *c = s->p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    const size_t N = (size_t)s->N;
    delete s->resume;
    s->resume = new bq_solver::resume_t();
    bq_solver::resume_t *r = s->resume;
    r->have = in->have & (BQ_STATE_G | BQ_STATE_MULT | BQ_STATE_MASKS);
    if (r->have & BQ_STATE_G) r->g.assign(in->g, in->g + N);
    if (r->have & BQ_STATE_MULT) {
        r->lp.assign(in->lp, in->lp + N);
        r->lm.assign(in->lm, in->lm + N);
    }
    if (r->have & BQ_STATE_MASKS) {
        r->mL.resize(N);
        r->mU.resize(N);
        for (size_t i = 0; i < N; ++i) {
            r->mL[i] = in->mask_l[i] != 0.0;
            r->mU[i] = in->mask_u[i] != 0.0;
            if (r->mL[i] && r->mU[i]) {
                bq_set_error("state: index %zu is in both masks (L and U are disjoint: active_set.py:85)", i);
                return BQ_ERR_BADARG;
            }
        }
    }
    s->host.iter = in->iter;
    if (s->kind == BQ_FW && in->best_lb == in->best_lb) s->host.best_lb = in->best_lb;
    BQ_HIP(hipMemcpyAsync(s->x, in->x, sizeof(double) * N, hipMemcpyHostToDevice, c->stream));
    BQ_HIP(hipMemcpyAsync(s->sc, &s->host, sizeof(bq_scal), hipMemcpyHostToDevice, c->stream));
    BQ_SYNC(c);
    return BQ_OK;
}
#pragma clang fp contract(on)

extern "C" int bq_decision_function(bq_ctx *c, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                                    const double *SV, const double *coef, double intercept, int64_t t,
                                    const double *Xt, double *out) {
    BQ_ARG(c && SV && coef && Xt && out, "NULL argument");
    BQ_ARG(m >= 1 && d >= 1 && t >= 1, "m/d/t");
    BQ_HIP(hipSetDevice(c->device));
    return bq_launch_decision(c, kernel, gamma, coef0, degree, m, d, SV, coef, intercept, t, Xt, out);
}

extern "C" int bq_gram_matrix(bq_ctx *c, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                              const double *A, int64_t t, const double *B, double *out) {
    BQ_ARG(c && A && out, "NULL argument");
    BQ_ARG(m >= 1 && d >= 1 && (B == nullptr || t >= 1), "m/d/t");
    BQ_HIP(hipSetDevice(c->device));
    return bq_launch_gram_matrix(c, kernel, gamma, coef0, degree, m, d, A, t, B, out);
}

int bq_chol_solve_dense_impl(bq_ctx *ctx, int64_t n, const double *A, const double *b, double *x, double *factor_ms);

extern "C" int bq_cholesky_solve(bq_ctx *c, int64_t n, const double *A, const double *b, double *x, double *factor_ms) {
    BQ_ARG(c && A && b && x, "NULL argument");
    BQ_ARG(n >= 1, "n");
    BQ_HIP(hipSetDevice(c->device));
    return bq_chol_solve_dense_impl(c, n, A, b, x, factor_ms);
}

