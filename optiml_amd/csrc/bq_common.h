// Internal declarations shared by the HIP translation units of libbcqp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bcqp.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void bq_set_error(const char *fmt, ...);

#define BQ_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            bq_set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, hipGetErrorString(_e)); \
            return BQ_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define BQ_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc != BQ_OK) return _rc; \
    } while (0)

#define BQ_ARG(cond, msg)                 \
    do {                                  \
        if (!(cond)) {                    \
            bq_set_error("bad argument: %s", msg); \
            return BQ_ERR_BADARG;         \
        }                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------
// test hooks: ONE environment variable, BQ_TEST_HOOKS="name=value,name=value", read at every use (the tests switch hooks between
// solves of one process).  They force paths that a heuristic would not take at test sizes, or switch a shortcut off so that a test
// can hold it against the long way — nothing a user of the library needs (round 6: these were eighteen BQ_* variables).
//   rows_per_step=4|8  stream_unit=U  minres_big_min=N  as_schur=0  as_schur_min=N  as_schur_limit=N  as_mailbox=0  as_f_chain=0
//   as_cg_warm=0  as_cg_incq=0  as_cg_colq=0  as_cg_pc_incr=0  as_cg_pc_class=0..3  ip_svr_reduced=0  smo_helpers=N
//   panel_good_gbs=G  alloc_fail_above=BYTES  decision_chunk_rows=R  sweep_block=1024|2048|4096
// ---------------------------------------------------------------------------------------------
bool bq_hook(const char *name, double *value);                       // true (and *value) when the hook is set
static inline double bq_hook_value(const char *name, double dflt) {
    double v = dflt;
    return bq_hook(name, &v) ? v : dflt;
}
static inline bool bq_hook_on(const char *name) { return bq_hook_value(name, 1.0) != 0.0; }   // default on; "name=0" switches off

// ---------------------------------------------------------------------------------------------
// device allocation: EVERY hipMalloc of the library goes through bq_device_malloc, which on failure gives the panels the
// live contexts keep cached (bq_ctx.panel_cache) back to the driver and tries once more
// ---------------------------------------------------------------------------------------------
hipError_t bq_device_malloc(void **ptr, size_t bytes);
template <typename T>
static inline hipError_t bq_device_malloc(T **ptr, size_t bytes) { return bq_device_malloc((void **)ptr, bytes); }
#ifndef BQ_NO_MALLOC_REDIRECT
#define hipMalloc(ptr, bytes) bq_device_malloc((ptr), (bytes))
#endif

// ---------------------------------------------------------------------------------------------
// layout constants
// ---------------------------------------------------------------------------------------------
// Every device vector and every panel row is padded to a multiple of BQ_PAD elements and the pad is
// kept at zero, so the streaming kernels never need a column tail.
constexpr int64_t BQ_PAD = 1024;
constexpr int BQ_VEC_BLOCK = 256;                 // threads per block in the O(n) kernels
constexpr int BQ_VEC_ITEMS = 4;                   // elements per thread
constexpr int BQ_VEC_TILE = BQ_VEC_BLOCK * BQ_VEC_ITEMS;  // 1024 elements per block == BQ_PAD
constexpr int BQ_MAX_PARTIAL_Q = 32;              // reduced quantities per block of 1024 elements (PG / FW: five per block of 256 rows; AL: eight)

static inline int64_t bq_round_up(int64_t a, int64_t m) { return (a + m - 1) / m * m; }

enum { BQ_PROF_MATVEC = 0, BQ_PROF_GRAM = 1, BQ_PROF_CHOL = 2, BQ_PROF_EXCH = 3, BQ_PROF_PCSHARD = 4, BQ_PROF_COUNT = 5 };
// BQ_COMM_SHARE: one rank's share of a `world`-way partition with NO transport behind it — every collective is a no-op, so the
// products hold this rank's contributions only.  It exists to time and inspect a share on a single GPU (bq_ctx_create_share).
enum { BQ_COMM_NONE = 0, BQ_COMM_RCCL = 1, BQ_COMM_CALLBACK = 2, BQ_COMM_SHARE = 3 };

struct bq_prof_pending {
    hipEvent_t first = nullptr, second = nullptr;
    int skip_idx = -1, skip_seq = 0;   // slot of bq_ctx::prof_skip the kernel writes skip_seq into when it returned on a `done` flag
};
struct bq_prof_slot {
    double total_ms = 0.0;
    int64_t launches = 0, skipped = 0;
    std::vector<bq_prof_pending> pending;
};
constexpr int BQ_PROF_SKIP_CAP = 16384;

struct bq_ctx {
    int device = 0, rank = 0, world = 1;
    int num_cu = 256;
    char name[128] = {0};
    hipStream_t stream = nullptr;
    int comm_kind = BQ_COMM_NONE;
    bool sym_allreduce = false;   // BQ_SYM_EXCHANGE=allreduce: symmetric products end in ncclAllReduce instead of the
                                  // deterministic segment all-gather
    void *nccl_comm = nullptr;
    bq_exchange_fn exch_fn = nullptr;
    void *exch_user = nullptr;
    double *pinned = nullptr;  // host staging for the callback exchange
    size_t pinned_cap = 0;
    bool profiling = false;
    bq_prof_slot prof[BQ_PROF_COUNT];
    // a product enqueued behind a solver's `done` flag returns at once when the flag is up: such a launch writes its sequence
    // number into its slot of this ring, and bq_ctx_profile_read leaves exactly those launches out of the mean (it
    // used to guess them from their duration)
    int *prof_skip = nullptr;
    long long prof_seq = 0;
    int prof_cur_idx = -1, prof_cur_seq = 0;   // the slot handed to the launch between bq_prof_begin and bq_prof_end
    std::vector<hipEvent_t> event_pool;
    // One released panel is kept for the next problem of about the same size: on this platform a 40 GB hipMalloc issued
    // right after a 40 GB hipFree takes 1-2 s instead of 0.04 s (measured), and fits in a loop — multi-class, parameter
    // sweeps, cross-validation — create panels of the same size over and over.  Dropped when ANY device allocation of the
    // library fails (bq_device_malloc, which every hipMalloc here is routed through) and with the context.
    void *panel_cache = nullptr;
    size_t panel_cache_bytes = 0;
    // Allocations the placement choice did not keep (place_panel), HELD until a solver of their problem (or the problem) goes: hipFree of a large allocation is
    // followed by a transient — ~0.5 s after 3 GB, longer after 40 GB — during which every kernel of the process streams 1.5 - 4.5 %
    // slower (round 5, profiles/r05/placement_release_transient.txt: the driver clears what was released), i.e. exactly while the
    // solve that the choice was made for runs.  Given back earlier when ANY device allocation of the library fails (bq_alloc.cpp).
    // budget of the placement choice (bq_ctx_set_placement_budget): min / max ms and the products the caller expects to run
    double place_min_ms = -1.0, place_max_ms = -1.0, place_products = 0.0;
    struct held_t {
        void *ptr;
        size_t bytes;
        const void *owner;
    };
    std::vector<held_t> held;
    // problems alive on this context; a context destroyed while some are is only marked and goes with the last of them
    int refs = 0;
    bool zombie = false;
    // bounded collectives (bq_ctx_set_collective_timeout): a watchdog thread looks at how long the host has been inside a wait on
    // the compute stream; past the limit it aborts the RCCL communicator — the kernel of a collective whose peer never arrives
    // ends, the wait returns, and every later call on this context fails with BQ_ERR_RCCL
    struct bq_watchdog *watchdog = nullptr;
    std::atomic<bool> comm_aborted{false};   // written by the watchdog thread
};
// every host wait on the compute stream of a (possibly multi-rank) context: stamps the wait for the watchdog and turns an abort
// into BQ_ERR_RCCL
int bq_ctx_sync(bq_ctx *ctx);
int bq_ctx_event_sync(bq_ctx *ctx, hipEvent_t ev);
int bq_ctx_wait_flag(bq_ctx *ctx, const volatile int *flag, int want);   // a word the device posts into mapped pinned memory
void bq_watchdog_stop(bq_ctx *ctx);   // joins the thread (context destruction)
#define BQ_SYNC(ctx) BQ_TRY(bq_ctx_sync(ctx))

struct bq_problem {
    bq_ctx *ctx = nullptr;
    int refs = 0;          // solvers alive on this problem; destroyed while some are: marked, goes with the last of them
    bool zombie = false;
    int structure = BQ_PLAIN, storage = BQ_F64, kernel = -1;
    bool add_one = false;  // panel holds K and the Hessian entry is +-(K+1)
    int64_t n = 0;         // panel columns (= samples for kernel problems)
    int64_t N = 0;         // dual dimension (n, or 2n for BQ_SVR)
    int64_t ld = 0;        // padded leading dimension of the panel (elements)
    int64_t ldN = 0;       // padded length of dual-dim vectors
    int64_t r0 = 0, r1 = 0, blk = 0;  // my rows [r0,r1) and the per-rank block size (row-block mode)
    // symmetric mode (kernel-built panels): only tiles on/below the diagonal are stored and streamed; this rank owns
    // the 256-row tile rows [I0, I1) of nb, panel row 0 is global row I0*256
    size_t panel_bytes = 0;    // allocated size of `panel_alloc`
    void *panel_alloc = nullptr;   // what the allocator handed out; `panel` = panel_alloc (round 4's offset experiment — no effect — is gone)
    int place_tried = 0;       // BQ_PLACE_PANEL: placements timed, and the product's launch time on each
    double place_ms[32] = {0.0};
    double alloc_ms = 0.0;     // what the driver took to hand out the panel (0: it came from the context's cache)
    bool symmetric = false;
    bool streamed = false;     // BQ_STREAM: no panel, Gram tiles recomputed inside every product (stream_img)
    void *stream_img = nullptr;
    int64_t I0 = 0, I1 = 0, nb = 0;
    double *slab = nullptr;   // nb x nb x 256 partial products
    // canonical segments of the tile rows (bq_sym_segments): this rank owns segments [seg_lo, seg_hi) of seg_count; the
    // per-segment partial products are gathered into gath[world][seg_cmax][nb*256] and summed in segment order
    int seg_count = 0, seg_lo = 0, seg_hi = 0, seg_cmax = 0;
    double *gath = nullptr;
    void *panel = nullptr;
    double *q = nullptr;    // ldN
    double *sgn = nullptr;  // ld (labels, BQ_SVC) or null
    double *X = nullptr;    // n x d (kernel problems)
    int64_t d = 0;
    double gamma = 0, coef0 = 0, diag_add = 0;
    int degree = 0;
    double *w = nullptr;   // ld: panel-product input
    double *s = nullptr;   // world*blk (>= n): panel-product output, gathered
    double *va = nullptr, *vb = nullptr;  // ldN scratch for the host-vector entry points
    double *partials = nullptr;           // BQ_MAX_PARTIAL_Q * nblocks(ldN)
    double *scal = nullptr;               // a few device scalars for the host-vector entry points
};

// device-resident scalar state of a solver; one instance per solver, read back in one copy
struct bq_scal {
    long long iter, max_iter, stat_base, stat_cap;
    int status, done, fw_clip, al_pending;     // al_pending: an update has run whose stop test / multiplier step is still due (bq_epilogue.h, kind 2)
    double eps, fw_t;
    double f, ng, gd, max_t, den, t;          // PG / FW
    double low, best_lb, gap;                  // FW
    double p, mu, xr, step;                    // IP
    double al_mu, al_ax, al_pf;                // augmented Lagrangian: multiplier of the equality row, a'x, primal value
    long long al_epoch;
    unsigned int ticket[2], pad1[2];           // last-block tickets of the fused reduce-and-decide kernels (PG / FW)
    long long al_last;                         // the epoch test fired: the gradient of the last record is still due
    double aux[8];
};

struct bq_chol_ws;

// augmented-Lagrangian driver (bq_al.hip): device vectors (null pointer = that constraint family is absent)
struct bq_al_vecs {
    double *x, *xe, *g, *Qx, *q, *step, *s1, *s2, *s3;
    double *chk;   // 3 x ldN: an update's per-element terms of the stop test (|c(x_new)|^2, |d dual|^2, |d x|^2), summed by the next closing kernel
    double *a, *lb, *ub, *llb, *lub;   // equality row; bounds; their multipliers
    const double *lr_sched, *mom_sched;   // optional per-iteration step sizes / momenta (index: iteration), else null
    long long sched_len;
};
struct bq_al_state {
    bq_al_params prm;
    bq_al_vecs V;
    bool w_ready = false;   // p->w holds the structure map of the current x (the last update kernel wrote it): cleared at the top of
                            // every bq_solver_run, other calls on the problem use p->w too
};

struct bq_solver {
    bq_problem *p = nullptr;
    int kind = BQ_PG;
    int64_t N = 0, ldN = 0, nblk = 0;
    double *x = nullptr, *g = nullptr, *d = nullptr, *Qd = nullptr, *lb = nullptr, *ub = nullptr;
    double *lp = nullptr, *lm = nullptr, *rhs = nullptr, *hd = nullptr, *dlp = nullptr, *dlm = nullptr;  // IP
    unsigned char *mL = nullptr, *mU = nullptr;                                                         // AS
    double *partials = nullptr;
    bq_scal *sc = nullptr;
    bq_iter_stat *stats = nullptr;
    int64_t stats_cap = 0;
    bq_scal host;  // mirror after the last run
    bool initialised = false;  // the one-time start-up product has run
    bool started = false;      // an evaluation has run, so a pending step may exist
    bq_chol_ws *chol = nullptr;
    void *as_ws = nullptr;
    bool as_cg = false;                // BQ_AS_CG: restricted systems by conjugate gradients instead of a dense factor
    double inner_rtol = 1e-13;
    long long inner_max = 0;           // 0: 2 |A| + 50
    bq_al_state *al = nullptr;
    int *flag_host = nullptr;          // pinned copy of sc->done + its event (lagged polling in bq_solver_run)
    hipEvent_t flag_event = nullptr;
    // bq_solver_set_state: what the start-up must take instead of forming it (x itself is uploaded at once)
    struct resume_t {
        int have = 0;                                 // BQ_STATE_* bits
        std::vector<double> g, lp, lm;
        std::vector<unsigned char> mL, mU;
    };
    resume_t *resume = nullptr;
};

// ---------------------------------------------------------------------------------------------
// cross-file launchers (each defined next to its kernels)
// ---------------------------------------------------------------------------------------------
// bq_ctx.cpp / bq_api.hip
int bq_prof_begin(bq_ctx *ctx, int which, hipEvent_t *e0, hipEvent_t *e1, bool bracket = true);
int bq_prof_end(bq_ctx *ctx, int which, hipEvent_t e0, hipEvent_t e1, bool recorded = false);
void bq_prof_drop(bq_ctx *ctx, hipEvent_t e0, hipEvent_t e1);   // a pair that was taken and never used goes back to the pool
// between bq_prof_begin and the launch of a kernel that may return on a `done` flag: where that kernel reports "skipped"
// (*slot null when not profiling); the kernel does `if (*done) { if (slot && first thread) *slot = seq; return; }`
void bq_prof_skip_arg(bq_ctx *ctx, hipEvent_t e0, int **slot, int *seq);
int bq_exchange_rows(bq_ctx *ctx, double *s, int64_t n, int64_t blk, int64_t r0, int64_t r1);
int bq_exchange_sum(bq_ctx *ctx, double *v, int64_t count);  // all-reduce(sum) of a replicated-length vector
// in-place all-gather of equal chunks: buf holds world*chunk doubles, this rank's chunk (at rank*chunk) is fresh on entry
int bq_exchange_gather(bq_ctx *ctx, double *buf, int64_t chunk);
void bq_problem_unref(bq_problem *p);   // a solver of p has gone: destroys p if that was pending
void bq_ctx_register(bq_ctx *c, bool alive);   // bq_alloc.cpp: the contexts whose cached panels a failing allocation may drop
// bq_alloc.cpp: the cached panel of a context, every access under one mutex (a failing allocation on another thread may drop it)
void bq_ctx_drop_cache(bq_ctx *ctx);   // give the cached panel back to the driver
void *bq_ctx_cache_take(bq_ctx *ctx, size_t bytes, size_t *cap);   // the cached panel if it fits `bytes` (<= 25 % spare), else null
void bq_ctx_cache_put(bq_ctx *ctx, void *panel, size_t bytes);
void bq_ctx_hold(bq_ctx *ctx, void *ptr, size_t bytes, const void *owner);   // keep an unused allocation until bq_ctx_release_held
void bq_ctx_release_held(bq_ctx *ctx, const void *owner);                    // owner's blocks (null: all) go back to the driver

// bq_symv.hip: symmetric tile product over tile rows [I0, I1) -> out (nb*256 partial sums)
constexpr int64_t BQ_SYM_TILE = 256;
// Packed layout of a symmetric (kernel-built) panel: tile row I (256 rows) keeps only its columns [0, (I+1)*256), stored
// row-major with pitch (I+1)*256; tile rows are concatenated.  Element (i, j), j's tile <= i's tile, of a panel whose
// first stored tile row is I0:
__host__ __device__ inline int64_t bq_sym_off(int64_t I) { return BQ_SYM_TILE * BQ_SYM_TILE * (I * (I + 1) / 2); }
__host__ __device__ inline int64_t bq_sym_pitch(int64_t I) { return (I + 1) * BQ_SYM_TILE; }
__host__ __device__ inline int64_t bq_sym_addr(int64_t i, int64_t j, int64_t I0) {
    const int64_t I = i / BQ_SYM_TILE;
    return bq_sym_off(I) - bq_sym_off(I0) + (i - I * BQ_SYM_TILE) * bq_sym_pitch(I) + j;
}
// Canonical segments of a symmetric panel's tile rows.  The partial products of the lower-triangle tiles are summed per
// SEGMENT (a contiguous range of tile rows holding 1/S of the triangle's tiles, boundaries a function of nb and S only) and the
// S segment vectors are then added in segment order on every rank, so the product is bit-identical for any number of
// ranks that own whole segments: 1, 2, 4 and 8 GPUs see the same iterates.  S = 8 up to 8 ranks.
constexpr int BQ_SYM_SEG = 8;
constexpr int BQ_SYM_SEG_MAX = 64;
static inline int bq_sym_segments(int world) { return BQ_SYM_SEG * ((world + BQ_SYM_SEG - 1) / BQ_SYM_SEG); }
int64_t bq_sym_seg_cut(int64_t nb, int s, int S);                              // first tile row of segment s (s = S: nb)
static inline int bq_sym_seg_first(int rank, int world, int S) { return (int)((int64_t)rank * S / world); }
struct bq_seg_table {
    int count;                              // S
    int lo, hi;                             // the segments summed by this launch (this rank's)
    int slot[BQ_SYM_SEG_MAX];               // segment -> slot of the gathered buffer (owner rank * cmax + local index)
    long long cut[BQ_SYM_SEG_MAX + 1];      // tile-row boundaries
};
// tiles of the segments [tab.lo, tab.hi) + their sum in segment order -> out (nb*256)
struct bq_epilogue;   // bq_epilogue.h: the PG / FW step fused into the kernel that finishes the product (null: none)
int bq_launch_symv(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nb, const bq_seg_table &tab,
                   const double *w, double *slab, double *out, const int *done, const bq_epilogue *epi = nullptr);
// the same with the segment partials written to `gath` (slots of this rank) instead of one summed vector; and the closing
// sum of all S gathered segment vectors -> out
int bq_launch_symv_segments(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nb, const bq_seg_table &tab,
                            const double *w, double *slab, double *gath, const int *done);
int bq_launch_symv_segsum(bq_ctx *ctx, int64_t nb, const bq_seg_table &tab, const double *gath, double *out, const int *done,
                          const bq_epilogue *epi = nullptr);
void bq_sym_seg_table(const bq_problem *p, bq_seg_table *tab);
// -> p->s (complete on all ranks).  epi: fuse the PG / FW epilogue into the closing kernel where the path has one (*fused says so)
int bq_panel_product(bq_problem *p, bool add_one, const double *w, const int *done, const bq_epilogue *epi = nullptr,
                     bool *fused = nullptr);

// bq_dense.hip: a dense host Hessian into the resident panel — packed lower tile rows when Q == Q' exactly (checked on the device
// while uploading, agreed across ranks), else row blocks
bool bq_dense_host_spot_symmetric(const double *Q, int64_t n);
int bq_dense_upload_sym(bq_problem *p, const double *Q, bool check, int *symmetric);
int bq_dense_agree(bq_ctx *c, bool mine, bool *any);   // a flag summed over the ranks (one all-reduce; no-op on one rank)
int bq_dense_upload_rows(bq_problem *p, const double *Q);

// bq_gemv.hip: s[r0 + i] = sum_j elem(panel[i][j]) * w[j], i in [0, nrows)
int bq_launch_gemv(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nrows, int64_t ld,
                   const double *w, double *s_rows, const int *done_flag);

// bq_gram.hip: panel rows [r0,r1) of kernel(X, X) (n x n), written in `storage` dtype with row pitch ld
int bq_launch_gram(bq_ctx *ctx, const double *X, int64_t n, int64_t d, int64_t r0, int64_t r1, int kernel,
                   double gamma, double coef0, int degree, void *panel, int storage, int64_t ld, bool sym_packed);
// rectangular cross-Gram fused with a coefficient contraction (decision function)
int bq_launch_decision(bq_ctx *ctx, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                       const double *SV, const double *coef, double intercept, int64_t t, const double *Xt,
                       double *out);

// streamed mode (bq_gram.hip): persistent k-major image of X + the fused Gram-tile x vector product
// rows [r0, r1): this rank's 256-row tile rows (whole canonical segments; r1 may exceed n)
int bq_stream_prepare(bq_ctx *ctx, const double *Xdev, int64_t n, int64_t d, int64_t r0, int64_t r1, void **out);
// the streamed product: segments / slots as for bq_launch_symv (mode 0: summed over this rank's segments into out; mode 1: per
// segment into the gathered buffer)
int bq_stream_sym_product(bq_ctx *ctx, void *h, int64_t n, int64_t nb, const bq_seg_table &tab, int kernel, double gamma, double coef0,
                          int degree, bool add_one, const double *w, double *out, int mode, const int *done);
void bq_stream_free(void *h);

int bq_launch_gram_matrix(bq_ctx *ctx, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                          const double *A, int64_t t, const double *B, double *out);

// bq_vec.hip
int bq_problem_apply(bq_problem *p, const double *v, double *out, const int *done_flag);  // out = Q v, all ranks
int bq_launch_prep(bq_problem *p, const double *v, const int *done_flag);   // p->w = the structure map of v (BQ_SVC: y o v; BQ_SVR: v+ - v-)
int bq_vec_eval_f(bq_problem *p, const double *x, const double *Qx, double *g_out, double *f_dev);

int bq_pgfw_start(bq_solver *s);
int bq_pgfw_iterate(bq_solver *s);
int bq_ip_start(bq_solver *s);
int bq_ip_iterate(bq_solver *s);
bool bq_ip_svr_reduced();   // n x n Schur reduction of the SVR Newton system (default; hook ip_svr_reduced=0 disables)
int bq_as_start(bq_solver *s);
int bq_as_iterate(bq_solver *s);
void bq_as_free(bq_solver *s);
int bq_al_iterate(bq_solver *s);   // bq_al.hip
int bq_al_flush(bq_solver *s);     // closes the last iteration of a run (its stop test is otherwise taken by the next closing kernel)
const double *bq_as_view(bq_solver *s, int what);
long long bq_as_inner_iters(bq_solver *s);
long long bq_as_counter(bq_solver *s, int which);

// bq_chol.hip
int bq_chol_ws_create(bq_ctx *ctx, int64_t n, bq_chol_ws **out);
void bq_chol_ws_destroy(bq_chol_ws *ws);
