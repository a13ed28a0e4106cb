/*
 * bcqp.h — C ABI of the MI355X-native box-constrained dual-QP hot path (libbcqp_hip.so).
 *
 * The reference (dmeoli/optiml) is pure Python/NumPy and has NO FFI; its "operator API" for this
 * path is the Python class surface.  Each entry point below replaces the NumPy/SciPy arithmetic of
 * one reference site (paths relative to the reference checkout); the Python classes in
 * optiml_amd/ keep the reference's names and ctor arguments and bind these symbols with ctypes
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 *   min 1/2 x'Qx + q'x   s.t.  lb <= x <= ub
 *
 * Conventions
 *   - every function returns an int: 0 = ok, <0 = error class (BQ_ERR_*); the message of the last
 *     error on the calling thread is returned by bq_last_error().
 *   - host buffers are BORROWED for the duration of the call, never retained.
 *   - device memory is owned by the opaque handles and released by the *_destroy functions.
 *   - a bq_ctx is not thread-safe; every call returns after its HIP stream has been synchronised.
 *   - all host-visible vectors are fp64.  `storage` selects the dtype of the resident Hessian /
 *     Gram panel only (fp32 storage still accumulates in fp64).
 *   - multi-GPU: one process per GPU; all n-vectors are replicated and each product Q*v is completed
 *     by ONE collective (RCCL over xGMI, or a caller-supplied exchange callback): kernel-built
 *     symmetric panels are split into tile rows with equal shares of the lower triangle
 *     (bq_sym_row_block: runs of 8 canonical segments) and end in an all-gather of the per-segment
 *     partial vectors, which every rank adds in segment order — bit-identical iterates for 1/2/4/8
 *     ranks (BQ_SYM_EXCHANGE=allreduce: one all-reduce(sum) instead); a dense Q that equals its transpose
 *     exactly is stored and split the same way; any other dense Q is split into equal row blocks
 *     (bq_row_block) and ends in an all-gather.  Solvers that factorise the Hessian (InteriorPoint,
 *     ActiveSet) and SMO need a single-rank context.
 */
#ifndef BCQP_H
#define BCQP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BQ_OK 0
#define BQ_ERR_HIP (-1)       /* a HIP runtime call failed */
#define BQ_ERR_RCCL (-2)      /* RCCL missing or a collective failed */
#define BQ_ERR_NOT_PD (-3)    /* Cholesky met a non-positive pivot (scipy: LinAlgError) */
#define BQ_ERR_NONFINITE (-4) /* inf/nan where the reference would raise (e.g. IP with ub=inf) */
#define BQ_ERR_BADARG (-5)
#define BQ_ERR_NOMEM (-6)

/* 2 (round 6): bq_ctx_probe_stall takes behind_collective, bq_problem_create_dense takes layout flags, bq_problem_layout,
 * bq_ctx_release_held and the state snapshot were added: a consumer built against version 1 must be rebuilt */
#define BQ_ABI_VERSION 2

typedef struct bq_ctx bq_ctx;
typedef struct bq_problem bq_problem;
typedef struct bq_solver bq_solver;

/* panel storage: fp64, fp32 (fp64 accumulation), or none at all — BQ_STREAM recomputes the Gram tiles on the MFMA inside
 * every product (kernel problems with an inner-product kernel; PG / FW / augmented-Lagrangian solvers only): the
 * fallback for n^2 s > HBM */
enum { BQ_F64 = 0, BQ_F32 = 1, BQ_STREAM = 2 };
/* linear, poly, rbf: kernels.py:40-129 (the named configs); sigmoid, laplacian: kernels.py:132-201 */
enum { BQ_KERNEL_LINEAR = 0, BQ_KERNEL_POLY = 1, BQ_KERNEL_RBF = 2, BQ_KERNEL_SIGMOID = 3, BQ_KERNEL_LAPLACIAN = 4 };
enum { BQ_PLAIN = 0, BQ_SVC = 1, BQ_SVR = 2 };                       /* Hessian structure */
/* OR-ed into the structure: leave out the rank-one term yy' / ee' of the regularised intercept, i.e. the
 * reg_intercept=False duals Q = K*yy' and Q = [[K,-K],[-K,K]] (svm/_base.py:552-559, 1096-1099).  Products only:
 * InteriorPoint / ActiveSet refuse such a problem (the reference has no box solver for it either, :621-624). */
#define BQ_NO_RANK_ONE 16
/* OR-ed into the structure: keep the WHOLE n x n Gram panel (row blocks, pitch round_up(n, 1024)) instead of the packed
 * lower-triangular tile rows: twice the memory and twice the bytes per product, but every row is contiguous and K keeps
 * the reference's own (not exactly symmetric) rounding of the RBF distances.  An option, not a default: measured on SMO
 * (single-row gathers) it changes nothing, n=100k fit 0.93 s either way — the sweeps are bound by one CU's gather rate. */
#define BQ_FULL_PANEL 32
/* OR-ed into the structure: choose WHERE the resident panel lives.  The rate at which the panel product streams a panel is a
 * stable property of the physical region the allocation landed in (five panels of one n = 100 000 problem held at once ran 6.04 -
 * 6.50 ms per product, each reproducible to 0.005 ms; NOT of its base address, nor of the contiguity the API can ask for:
 * profiles/r04/placement_*.txt): with this flag the product kernel is timed on the freshly allocated (still empty) panel and, if
 * it streams below ~6.5 TB/s, further allocations are tried while they fit a time budget (BQ_PLACE_BUDGET_MS, default 200 ms; at
 * most 3; a candidate is priced at what the first allocation cost; bq_ctx_set_placement_budget lets the budget grow with the work
 * the caller expects) and the fastest is kept (the others are held until the solve is over: bq_ctx_set_placement_budget; a candidate is only tried while a
 * tenth of the device stays free beside it).  For the product-bound solvers (PG, FW, ActiveSetCG, the
 * augmented-Lagrangian rules), whose every iteration streams the panel — SVC / SVR.fit set it for those; pointless for
 * InteriorPoint / ActiveSet / SMO.  Per rank, before the first collective. */
#define BQ_PLACE_PANEL 64
enum { BQ_PG = 0, BQ_FW = 1, BQ_AS = 2, BQ_IP = 3,                   /* solver kind */
       /* ActiveSet (active_set.py:82-237, same outer logic) whose restricted systems Q[A,A] xs = rhs are solved by
        * conjugate gradients on the masked panel product instead of a dense Cholesky factor: no n_A x n_A copy, so it
        * also runs on sharded (multi-rank), fp32-stored and streamed panels — SURVEY 7's plan for config 5.  The inner
        * iteration stops at |r| <= rtol (|(Q x)_A| + |q_A|); a direction of non-positive curvature is BQ_ERR_NOT_PD. */
       BQ_AS_CG = 5 };
enum { BQ_STATUS_UNKNOWN = 0, BQ_STATUS_OPTIMAL = 1, BQ_STATUS_STOPPED = 2 };
/* BQ_GET_X / BQ_GET_G: the point (and gradient) the LAST ITERATION RECORD was evaluated at — what the
 * reference's callback sees at the top of that iteration.  BQ_GET_X_NOW / BQ_GET_G_NOW: the current iterate
 * (differs only for ActiveSet, whose loop body moves x after the record: active_set.py:156-160, 208). */
enum { BQ_GET_X = 0, BQ_GET_G = 1, BQ_GET_LP = 2, BQ_GET_LM = 3, BQ_GET_D = 4,
       BQ_GET_MASK_L = 5, BQ_GET_MASK_U = 6, BQ_GET_X_NOW = 7, BQ_GET_G_NOW = 8,
       BQ_GET_DUAL = 9 /* augmented Lagrangian: [mu (if a_eq); lambda_lb (if lb); lambda_ub (if ub)] */ };

/* One row per evaluation at the top of a solver iteration (what the reference's callback/verbose
 * line sees).  r1/r2/r3 by solver:  PG: |proj grad|_2, step t, max_t;  FW: best lower bound, gap,
 * step a;  IP: dual value p, gap, step;  AS: |L|+|U|, event (0 step / 1 release-L / 2 release-U /
 * 3 optimal) , index or count. */
typedef struct bq_iter_stat {
    int64_t iter;
    double f;
    double r1, r2, r3;
} bq_iter_stat;

/* Exchange callback for multi-process runs without RCCL (tests; hosts without xGMI).  Return 0 on success.
 *   op 0 (gather): on entry buf holds this rank's rows [row_begin,row_end) of an n-vector; on return the whole
 *                  vector must be filled with every rank's rows   (row-block panels; the segment partials of the
 *                  symmetric tile panels: n = world * chunk, equal chunks)
 *   op 1 (sum):    on return buf[0:n) must hold the element-wise sum over ranks   (BQ_SYM_EXCHANGE=allreduce) */
typedef int (*bq_exchange_fn)(void *user, double *buf, int64_t n, int64_t row_begin, int64_t row_end, int op);

int bq_abi_version(void);
const char *bq_last_error(void);

/* ---- context -------------------------------------------------------------------------------- */
int bq_device_count(int *count);
int bq_ctx_create(int device, bq_ctx **out);
/* uid: 128-byte RCCL unique id, created on rank 0 by bq_comm_unique_id and broadcast by the caller */
int bq_comm_unique_id(void *uid128);
int bq_ctx_create_rccl(int device, int rank, int world, const void *uid128, bq_ctx **out);
/* where the time of RCCL's start-up went in this process so far: "dlopen(librccl.so) 0.41 s; dlsym x8 0.00 s; ncclGetUniqueId ...;
 * ncclCommInitRank ...; ncclCommCount ..." (empty when RCCL was never loaded).  A stage still running after 5 s also says so on
 * stderr every 5 s while it runs; NCCL_DEBUG=INFO prints every stage as it ends. */
int bq_comm_init_report(char *buf, size_t cap);
int bq_ctx_create_exchange(int device, int rank, int world, bq_exchange_fn fn, void *user, bq_ctx **out);
/* ONE rank's share of a `world`-way partition with no transport behind it: panels, segments and every kernel of the
 * per-rank iteration are exactly those of rank `rank` of `world`, every collective is a no-op — so products hold this rank's
 * contributions only.  For timing and inspecting a share on a single GPU (bench.py --emulate-shares); never a solver. */
int bq_ctx_create_share(int device, int rank, int world, bq_ctx **out);
int bq_ctx_destroy(bq_ctx *ctx);
int bq_ctx_info(const bq_ctx *ctx, int *device, int *rank, int *world, char *name, size_t name_cap);
/* the exchange behind a multi-rank context: kind 0 none / 1 RCCL / 2 callback / 3 share (bq_ctx_create_share); comm_ranks = ncclCommCount of the live
 * communicator (0 without RCCL); sym_allreduce = 1 when symmetric products end in an all-reduce (BQ_SYM_EXCHANGE=allreduce)
 * instead of the default all-gather of segment partials summed in a fixed order (bit-identical for any rank count) */
int bq_ctx_comm_info(const bq_ctx *ctx, int *kind, int *comm_ranks, int *sym_allreduce);
/* choose the closing collective of symmetric products (before the first problem is created): 0 = all-gather of segment
 * partials + ordered sum (default), 1 = all-reduce(sum) */
int bq_ctx_set_sym_allreduce(bq_ctx *ctx, int on);
/* HIP-event timing of the dominant kernels (on the stream they run on).  which: 0 = Q*v panel
 * product, 1 = Gram build, 2 = Cholesky factorisation, 3 = row-block exchange, 4 = the sample-sharded part of ActiveSetCG's
 * preconditioner (the implicit order-2 remainder: what divides by the rank count; bench.py's prediction for config 5 over 8). */
int bq_ctx_profile(bq_ctx *ctx, int enable);
int bq_ctx_profile_read(bq_ctx *ctx, int which, double *total_ms, int64_t *launches, int reset);
/* measured HBM ceilings of this GPU on a scratch buffer of `bytes`: a read-only streaming sweep and a
 * device-to-device copy (read + written bytes), in GB/s — the yardstick beside the nominal 8 TB/s */
int bq_ctx_probe_bandwidth(bq_ctx *ctx, int64_t bytes, int reps, double *read_gbs, double *copy_gbs);
/* measured fp64 matrix-core ceiling of this GPU: back-to-back v_mfma_f64_16x16x4_f64 on register operands for about `seconds`
 * (first half warm-up: the clock settles under the load), in TFLOP/s — the yardstick beside the nominal 78.6 TFLOP/s for
 * the roofline fraction of the Cholesky (SURVEY 8d asks for nominal-peak and measured fractions) */
int bq_ctx_probe_mfma_f64(bq_ctx *ctx, double seconds, double *tflops);
/* measured cost of this context's closing collective of a product (ONE collective per Q*v), on the context's stream
 * with HIP events around every call: kind 0 = the in-place all-gather of `count` doubles per rank (buffer world * count), kind 1
 * = the in-place all-reduce(sum) of `count` doubles.  `reps` calls after 3 warm-up calls; mean and minimum in microseconds.
 * On a ONE-rank RCCL communicator this is the launch + local-copy floor of the collective (a lower bound of what N > 1 ranks pay
 * over xGMI, nothing more); on N > 1 ranks every rank must call it with the same arguments. */
int bq_ctx_probe_exchange(bq_ctx *ctx, int kind, int64_t count, int reps, double *mean_us, double *min_us);
/* Bound the collectives of an RCCL context in time (0: no bound, the default; also BQ_COLLECTIVE_TIMEOUT_S through the Python
 * Context).  RCCL itself never gives up on a collective whose peer does not arrive (a rank-local error, a dead process): with a
 * timeout a watchdog thread aborts the communicator (ncclCommAbort) once the host has waited for longer on a stream that holds a
 * collective enqueued since it was last seen empty; the call in progress returns BQ_ERR_RCCL and the context is unusable afterwards.
 * A wait with only this rank's own kernels ahead of it is never cut short.  The callback transport is bounded by the caller's own
 * communicator (and on every context without an RCCL communicator the setting is accepted and does nothing).  Set it from the thread
 * that owns the context, while no call is in progress. */
int bq_ctx_set_collective_timeout(bq_ctx *ctx, double seconds);
/* occupy the compute stream for `milliseconds` (one lane spinning on the wall clock; it ends by itself) and wait for it through
 * the library's bounded wait.  behind_collective != 0: the context's collective (an all-reduce of one double) is enqueued BEHIND the
 * occupation first — what a peer that arrives `milliseconds` late looks like from this rank: the end-to-end test of the watchdog on
 * one GPU (BQ_ERR_RCCL when it fired).  behind_collective == 0: nothing but this rank's own work is ahead of the wait, which the
 * watchdog must leave alone however long it lasts (a factorisation, a preconditioner rebuild). */
int bq_ctx_probe_stall(bq_ctx *ctx, double milliseconds, int behind_collective);
/* row block [begin,end) of an n-row panel owned by `rank` out of `world` (pure arithmetic): equal 128-aligned
 * blocks for dense panels; bq_sym_row_block: the balanced triangular partition (256-aligned) of the symmetric
 * kernel panels, whose ranks stream only the tiles on/below the diagonal */
int bq_row_block(int64_t n, int rank, int world, int64_t *begin, int64_t *end);
int bq_sym_row_block(int64_t n, int rank, int world, int64_t *begin, int64_t *end);

/* ---- the quadratic ("Quadratic", optiml/opti/_base.py:228-300) ------------------------------- */
/* dense Q (n x n row-major fp64) and q: replaces the host copy at optiml/opti/_base.py:243.
 * Layout of the resident copy (bq_problem_layout tells which one was taken):
 *   - Q == Q' exactly (every pair of elements compared as stored, on the device, while the rows are uploaded; on a multi-rank
 *     context the ranks agree with one all-reduce, so the call is collective there): the packed lower-triangular 256-tile rows of
 *     the kernel-built panels — half the HBM and half the bytes per product (40 GB and ~6 ms instead of 80 GB and ~12 ms at
 *     n = 100 000), sharded by the canonical segments, bit-identical products for any rank count;
 *   - any other Q (the reference never checks symmetry, opti/_base.py:249-256, and computes `Q @ x`): whole row blocks.
 * OR-ed into `storage`:  BQ_DENSE_ROWS   row blocks whatever Q is (the layout of rounds 1-5; what a comparison needs);
 *                        BQ_DENSE_LOWER  the caller vouches for symmetry: only the lower triangle of the host matrix is read
 *                                        (LAPACK's uplo = 'L'), nothing is compared, half the PCIe traffic;
 *                        BQ_PLACE_PANEL  as for kernel problems (packed layout only). */
#define BQ_DENSE_ROWS 256
#define BQ_DENSE_LOWER 512
int bq_problem_create_dense(bq_ctx *ctx, int64_t n, const double *Q, const double *q, int storage,
                            bq_problem **out);
/* kernel-structured Hessian built on the device from X (n x d row-major fp64):
 *   K = kernel(X, X)                        optiml/ml/svm/kernels.py:49-51 / 91-95 / 125-129
 *   BQ_SVC: Q = K*yy' + yy' (+ diag_add*I), dual dim n     optiml/ml/svm/_base.py:552-555, 628, 729-730
 *   BQ_SVR: Q = [[K,-K],[-K,K]] + ee', dual dim 2n         optiml/ml/svm/_base.py:1096-1099, 1178
 *   BQ_PLAIN: Q = K (+ diag_add*I)
 * gamma must already be resolved ('scale'/'auto' are host-side scalars of X).  q has the dual dim. */
int bq_problem_create_kernel(bq_ctx *ctx, int structure, int64_t n, int64_t d, const double *X,
                             const double *y, int kernel, double gamma, double coef0, int degree,
                             double diag_add, const double *q, int storage, bq_problem **out);
/* destroy: device memory goes back to the driver, except that a context keeps the ONE most recently released panel of
 * >= 1 GB for the next problem of about that size (a large hipMalloc right after a large hipFree costs seconds on this
 * platform); it is released when an allocation fails and with the context. */
int bq_problem_destroy(bq_problem *p);
int bq_problem_dims(const bq_problem *p, int64_t *n_dual, int64_t *n_rows, int64_t *row_begin,
                    int64_t *row_end);
/* how the Hessian is resident on this rank: *packed = 1 for the packed lower tile rows (kernel-built panels; a dense Q == Q'),
 * 0 for row blocks; *streamed = 1 for BQ_STREAM (no panel); *panel_bytes = the size of this rank's panel allocation */
int bq_problem_layout(const bq_problem *p, int *packed, int *streamed, int64_t *panel_bytes);
/* out = Q v (dual dim; every rank gets the full vector)       optiml/opti/_base.py:291 (minus q) */
int bq_problem_matvec(bq_problem *p, const double *v, double *out);
/* f = 1/2 x'Qx + q'x and (optionally, g != NULL) g = Qx + q   optiml/opti/_base.py:282, 291 */
int bq_problem_eval(bq_problem *p, const double *x, double *f, double *g);
/* x_out = argmin 1/2 x'Qx + q'x without the box: Quadratic.x_star(), optiml/opti/_base.py:259-269 —
 * cho_solve(cho_factor(Q), -q) on the device (H assembled from the resident panel, blocked MFMA Cholesky); when a pivot
 * is <= 0 (scipy: LinAlgError) the reference's fallback scipy.sparse.linalg.minres(Q, -q)[0] with its defaults (rtol 1e-5,
 * 5 N iterations), one panel product per iteration.  *method: 0 Cholesky, 1 MINRES; *minres_iters: its iteration count.
 * Single-rank contexts, resident panels. */
int bq_problem_x_star(bq_problem *p, double *x_out, int *method, int64_t *minres_iters);
/* out = K w with the raw Gram panel (kernel problems only; n-vectors)  svm/_base.py:877-880 */
int bq_problem_gram_matvec(bq_problem *p, const double *w, double *out);
/* copy rows [row0,row0+nrows) of this rank's resident panel (n columns each) to the host as fp64 */
int bq_problem_panel_rows(bq_problem *p, int64_t row0, int64_t nrows, double *out);
/* time `reps` launches of the panel product with HIP events; returns the mean in ms */
int bq_problem_time_matvec(bq_problem *p, int reps, double *mean_ms);
/* BQ_PLACE_PANEL: how many placements of the panel were timed (0: the flag was not given or did not apply) and the product's
 * launch time on each, in the order tried (ms[0 .. min(*tried, cap))); the panel kept is the fastest of them */
int bq_problem_placement(const bq_problem *p, int *tried, double *ms, int cap);
/* Budget of the BQ_PLACE_PANEL choice for the problems created on this context from now on.  A slow placement costs ~5 % of every
 * product of the solve that follows, so the choice may spend up to 2 % of the products the caller EXPECTS to run
 * (expected_products x the product's measured time), never less than min_ms (< 0: BQ_PLACE_BUDGET_MS, default 200) and never
 * more than max_ms.  SVC / SVR.fit pass their max_iter (svm/_base.py:187-240: the optimizer's iteration cap), a steady-state
 * measurement passes a large number; expected_products = 0 (the default) keeps the fixed min_ms budget.  Candidates that were
 * not kept stay allocated until the first solver created on the problem is destroyed (or the problem, or a device allocation of
 * the library fails): releasing them before the solve slowed it down (profiles/r05/placement_release_transient.txt). */
int bq_ctx_set_placement_budget(bq_ctx *ctx, double min_ms, double max_ms, double expected_products);
/* give the allocations the placement choice is holding back (all of them on this context: up to two panel-sized blocks per problem
 * whose choice tried candidates) to the driver NOW — for a caller that needs the memory for something this library does not see
 * (another allocator in the process, a communicator about to be created).  *bytes (optional): what was released.  The ~0.5 s
 * transient that follows a large release (everything streams 1.5 - 4.5 % slower) is then the caller's. */
int bq_ctx_release_held_memory(bq_ctx *ctx, int64_t *bytes);

/* ---- solvers (optiml/opti/constrained/, the four .py files) --------------------------------------------------- */
/* lb/ub/x0: dual-dim fp64 host vectors (lb, x0 may be NULL: 0 and mid-box, constrained/_base.py:61-65).
 * eps: stopping accuracy; max_iter: as the reference; fw_t: FrankWolfe trust radius in [0,1).
 * BQ_AS (active_set.py:82-237): same masks, candidate / ratio step, Bland release and tolerances as the reference; the
 * restricted system of line :141 is solved from a Cholesky factor that is KEPT across iterations (the free set moves by
 * one index at a time: the changes are carried through a Schur complement on a base factor, which is rebuilt every 96 to 512
 * changes; hook as_schur=0 re-factorises in every iteration like the reference).  A non-positive pivot takes
 * the reference's minres branch (:142-151). */
int bq_solver_create(bq_problem *p, int kind, const double *lb, const double *ub, const double *x0,
                     double eps, int64_t max_iter, double fw_t, bq_solver **out);
int bq_solver_destroy(bq_solver *s);
/* advance by at most max_steps iterations, device-resident; stats receives one row per evaluated
 * iteration (capacity stats_cap rows, at least max_steps+1).  *status is BQ_STATUS_UNKNOWN while the
 * solver can continue. */
int bq_solver_run(bq_solver *s, int64_t max_steps, bq_iter_stat *stats, int64_t stats_cap,
                  int64_t *n_stats, int *status);
int bq_solver_state(const bq_solver *s, int64_t *iter, int *status, double *f_x);
/* BQ_AS_CG only: relative residual level (default 1e-13) and iteration cap (0 = 2 |A| + 50) of the inner conjugate
 * gradients; total inner iterations so far (0 for the other solvers). */
int bq_solver_set_inner(bq_solver *s, double rtol, int64_t max_iter);
int bq_solver_inner_iters(bq_solver *s, int64_t *total);
/* ActiveSet bookkeeping, totals since the solver was created (0 for the other solvers):
 *   BQ_COUNT_MINRES    iterations whose restricted system Q[A,A] was NOT factorised (non-positive pivot) and took the reference's
 *                      minres branch (active_set.py:142-151) — what a test of the pivot threshold (BQ_AS_PIVOT_REL) looks at
 *   BQ_COUNT_REFACTOR  base-set factorisations of the kept-factor path;  BQ_COUNT_REUSED  iterations solved through a kept factor
 *   BQ_COUNT_INNER     = bq_solver_inner_iters
 *   BQ_COUNT_NO_PRODUCT  ratio-step iterations (active_set.py:152-176) whose f(x) came from the line-search identity
 *                      f(x + t d) = f(x) + (t - t^2/2) g_A'd_A instead of a product with Q (hook as_f_chain=0: none) */
#define BQ_COUNT_INNER 0
#define BQ_COUNT_MINRES 1
#define BQ_COUNT_REFACTOR 2
#define BQ_COUNT_REUSED 3
#define BQ_COUNT_NO_PRODUCT 4
int bq_solver_counter(bq_solver *s, int which, int64_t *value);
int bq_solver_get(bq_solver *s, int what, double *out);

/* ---- checkpoint / resume (SURVEY 5 "checkpoint / resume") -------------------------------------------------------
 * What the reference's loop holds at the TOP of an iteration, so that a run which was stopped (max_iter, a callback's
 * StopIteration, a process that has to go) can be continued in a NEW solver: the reference offers `x=` only
 * (constrained/_base.py:61-65) and keeps the rest as locals — InteriorPoint's multipliers lp / lm (interior_point.py:181-186),
 * ActiveSet's masks L / U (active_set.py:91-92), FrankWolfe's best lower bound (frank_wolfe.py:90,105-106), self.g_x.
 *
 * bq_solver_get_state: the state at the top of the NEXT iteration (iteration `iter`): a step that has been decided but not yet
 * applied to the device vectors (PG / FW: x + t d, g + t Qd; IP: x + t dx, lp + t dlp, lm + t dlm) is applied to the copies
 * handed out, with the device's own rounding (one rounded product, one sum).  Vector pointers that are NULL are skipped; `have`
 * says which of the others were filled (BQ_STATE_*): lp / lm exist for BQ_IP, the masks (1.0 / 0.0 per index) for BQ_AS / BQ_AS_CG.
 *
 * bq_solver_set_state: into a solver that has not run yet (same problem, kind, bounds).  x is required; g, lp & lm, the masks are
 * taken when given (`have`), otherwise formed as the reference forms them at a start point (g = Qx + q by one product, lp / lm
 * from g, empty masks).  With everything given, InteriorPoint, ProjectedGradient and FrankWolfe continue BIT-IDENTICALLY to the
 * run that was stopped (every later quantity is a function of this state); ActiveSet continues from the same point and masks with a
 * freshly built factor (the stopped run's was updated incrementally: same iterates to rounding).
 * max_iter keeps counting from `iter`. */
#define BQ_STATE_X 1
#define BQ_STATE_G 2
#define BQ_STATE_MULT 4   /* lp and lm */
#define BQ_STATE_MASKS 8  /* mask_l and mask_u */
typedef struct bq_solver_snapshot {
    int64_t iter;      /* iterations completed = index of the next iteration record */
    int kind;          /* BQ_PG ... (get: filled; set: must match the solver, or -1 = not checked) */
    int have;          /* BQ_STATE_* bits: which vectors are filled (get) / given (set) */
    double f;          /* objective of the last record (NaN before the first); informational on set */
    double best_lb;    /* FrankWolfe: best lower bound so far (-inf before the first record) */
    double *x, *g, *lp, *lm, *mask_l, *mask_u;   /* dual-dim fp64 host vectors owned by the caller, NULL = absent */
} bq_solver_snapshot;
int bq_solver_get_state(bq_solver *s, bq_solver_snapshot *state);
int bq_solver_set_state(bq_solver *s, const bq_solver_snapshot *state);

/* ---- augmented-Lagrangian dual + first-order update rules ------------------------------------
 * min 1/2 x'Qx + q'x  s.t.  a_eq'x = 0 (optional), lb <= x <= ub (each optional), relaxed into
 *   L(x) = f(x) + dual'c(x) + rho/2 |[c_eq; max(c_in, 0)]|^2,  c = [a_eq'x; lb - x; x - ub]
 * (AugmentedLagrangianQuadratic, optiml/opti/constrained/_base.py:224-410) and minimised by one of the reference's
 * full-batch "stochastic" rules (optiml/opti/unconstrained/stochastic/, the seven rule files) with the multiplier update and stop tests
 * of optiml/opti/_base.py:129-146.  This is the SVC/SVR dual branch for unconstrained optimizers
 * (optiml/ml/svm/_base.py:638-723, :1188-1270), including reg_intercept=False (equality row y / [1;-1]).
 * Records: f = L(x), r1 = primal value f(x), r2 = |c(x_new)|_2, r3 = |d dual|_2 + |d x|_2 (the two stop-test values).
 * With no constraint at all (a_eq = lb = ub = NULL) it is the plain rule on the quadratic: runs `epochs` iterations. */
enum { BQ_RULE_SGD = 0, BQ_RULE_ADAM = 1, BQ_RULE_AMSGRAD = 2, BQ_RULE_ADAMAX = 3, BQ_RULE_ADAGRAD = 4,
       BQ_RULE_ADADELTA = 5, BQ_RULE_RMSPROP = 6 };
enum { BQ_MOM_NONE = 0, BQ_MOM_POLYAK = 1, BQ_MOM_NESTEROV = 2 };
typedef struct bq_al_params {
    int32_t rule, momentum_type;   /* AdaGrad and AdaDelta have no momentum (the reference classes take none) */
    double step_size, momentum;    /* constants; per-iteration schedules: bq_al_solver_set_schedules */
    double beta1, beta2;           /* Adam, AMSGrad, AdaMax */
    double decay;                  /* AdaDelta, RMSProp */
    double offset;
    double rho, tol;               /* penalty (> 0); tolerance of the two stop tests */
    int64_t epochs;                /* full batch: one epoch per iteration; 'stopped' when reached */
} bq_al_params;
/* x0 is required (the reference draws it uniform(0,1): optiml/opti/_base.py:36-57 — host side); dual0 (layout of
 * BQ_GET_DUAL) may be NULL = zeros.  The returned solver is driven by bq_solver_run / _state / _get / _destroy. */
int bq_al_solver_create(bq_problem *p, const bq_al_params *prm, const double *a_eq, const double *lb,
                        const double *ub, const double *x0, const double *dual0, bq_solver **out);
/* optional schedules (stochastic/schedules.py; the reference draws one value per iteration from an iterable step_size /
 * momentum): entry k is used by iteration k, the last entry continues; either pointer may be NULL.  Before the first run. */
int bq_al_solver_set_schedules(bq_solver *s, const double *step_sizes, const double *momenta, int64_t count);
/* number of multipliers (length of BQ_GET_DUAL) */
int bq_al_solver_dual_size(const bq_solver *s, int64_t *n_dual);

/* ---- SMO on the resident Gram panel (optiml/ml/svm/smo.py) -----------------------------------------
 * SMOClassifier (:99-357) / SMORegression (:386-797): the reg_intercept=False duals through SVC/SVR(optimizer='smo')
 * (optiml/ml/svm/_base.py:560-573, :1106-1120).  `p` is a kernel-built problem of n samples (only its Gram panel K is
 * used; single-rank contexts); y: labels in {+1,-1} (BQ_SVC) or targets (BQ_SVR).  One call of bq_smo_run advances
 * by at most max_outer outer iterations (one sweep over the samples each, smo.py:327-351) and reports the total
 * number done and whether the loop has terminated.  Ties between free samples with bit-identical cached errors are
 * resolved towards the smaller index (the reference: CPython set order) — see oracle/smo_oracle.py. */
typedef struct bq_smo bq_smo;
enum { BQ_SMO_ALPHAS = 0,   /* n (BQ_SVC) or [alpha+; alpha-] 2n (BQ_SVR) */
       BQ_SMO_ERRORS = 1,   /* n: the error cache */
       BQ_SMO_SCALARS = 2,  /* 6: b_up, b_low, b_up_idx, b_low_idx, successful pair steps, intercept b */
       BQ_SMO_STATS = 3     /* 4: helper workgroups per full sweep, error sums they delivered, sums rejected because the list
                             * entries read did not hash to the published value, results rejected by their own checksum —
                             * the two rejection counts are self-checks of the inter-workgroup hand-off and must be 0 */ };
int bq_smo_create(bq_problem *p, int task, const double *y, double C, double epsilon, double tol, bq_smo **out);
int bq_smo_run(bq_smo *s, int64_t max_outer, int64_t *outer_iters, int *finished);
int bq_smo_get(bq_smo *s, int what, double *out);
int bq_smo_destroy(bq_smo *s);

/* ---- prediction (optiml/ml/svm/_base.py:284-287) ------------------------------- */
/* out[t] = sum_m coef[m] * kernel(SV[m], Xt[t]) + intercept   (SV: m x d, Xt: t x d, row-major fp64) */
int bq_decision_function(bq_ctx *ctx, int kernel, double gamma, double coef0, int degree, int64_t m,
                         int64_t d, const double *SV, const double *coef, double intercept, int64_t t,
                         const double *Xt, double *out);

/* dense Gram matrix out (m x t, row-major) = kernel(A (m x d), B (t x d)); B == NULL means B is A (t ignored)
 * — the kernel functors' __call__, optiml/ml/svm/kernels.py:49-51 / 91-95 / 125-129 */
int bq_gram_matrix(bq_ctx *ctx, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                   const double *A, int64_t t, const double *B, double *out);

/* x = A^-1 b for a dense symmetric positive definite A (n x n row-major fp64, lower triangle read):
 * scipy.linalg.cho_solve(cho_factor(A), b) at optiml/opti/constrained/interior_point.py:235 and
 * active_set.py:141.  BQ_ERR_NOT_PD mirrors LinAlgError.  factor_ms (optional): HIP-event time of the
 * factorisation alone. */
int bq_cholesky_solve(bq_ctx *ctx, int64_t n, const double *A, const double *b, double *x, double *factor_ms);

#ifdef __cplusplus
}
#endif
#endif /* BCQP_H */
