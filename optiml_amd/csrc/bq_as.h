// Internals shared by the translation units of the ActiveSet driver (round 5 split what was one 2 700-line file):
//   bq_as.hip        the outer iteration (active_set.py:84-230): masks, compaction, ratio step / release, the classic per-iteration
//                    factorisation and the minres branch, start / iterate / free
//   bq_as_schur.hip  factor re-use: the kept base factor and its Schur-complement updates (dense ActiveSet)
//   bq_as_cg.hip     BQ_AS_CG: conjugate gradients on the masked panel operator, warm start, product-free bookkeeping
//   bq_as_pc.hip     the diagonal + low-rank preconditioner of those conjugate gradients (features, Woodbury, rank-one updates)
#pragma once
#include <cmath>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

#include "bq_chol.h"
#include "bq_qelem.h"
#define BQ_EXP_ATTR __device__ __forceinline__
#define BQ_EXP_LOINT(t) __double2loint(t)
#include "bq_exp.h"

#define ACT_TOL 1e-12
constexpr int AS_SCHUR_MAX = 1536;   // capacity of the update slots

#define VEC_LOOP(i)                                                             \
    const int64_t _base = (int64_t)blockIdx.x * BQ_VEC_TILE + threadIdx.x;      \
    _Pragma("unroll") for (int _j = 0; _j < BQ_VEC_ITEMS; ++_j)                 \
        for (int64_t i = _base + (int64_t)_j * BQ_VEC_BLOCK, _once = 1; _once; _once = 0)

static inline dim3 vgrid(int64_t ldN) { return dim3((unsigned)(ldN / BQ_VEC_TILE)); }

// device-resident scalar recurrences of the conjugate-gradient inner solver
struct as_cg_scal {
    double rr, alpha, beta, tol2;
    long long iters, max_iters;
    int done, info;
    unsigned int ticket[2];
    double rz;   // preconditioned runs: r'z (alpha = rz / p'Qp, beta = rz_new / rz)
    int pc, pad;
};

// the sample segments of the sharded parts of the preconditioner (bq_as_pc.hip, bq_as_pc2.hip): segment k = sample blocks [blk[k], blk[k + 1]) of 1024 samples,
// this rank owns [lo, hi); gathered buffers hold world * cmax slots of maxlen blocks (a kernel argument: plain data)
struct as_pc_part {
    int S, lo, hi, cmax;
    int slot[BQ_SYM_SEG_MAX];
    long long blk[BQ_SYM_SEG_MAX + 1];
    long long maxlen;
};
// Preconditioner of the inner conjugate gradients: P = D + Phi Phi' restricted to the free set, Phi (N x m, m << N) an
// explicit low-rank factor (stored in fp32) of the smooth part of the Hessian and D the diagonal left over.  Applied through Woodbury:
//   P_AA^-1 r = D^-1 r - D^-1 Phi_A G^-1 Phi_A' D^-1 r,   G = I + Phi_A' D_A^-1 Phi_A  (m x m, factorised once per outer iteration).
// RBF panels: the first-order Taylor features of exp(-g|x-x'|^2) = e^{-g|x|^2} e^{-g|x'|^2} (1 + 2g x.x' + ...), i.e.
// Phi_i = y_i e^{-g|x_i|^2} [1, sqrt(2g) x_i] (+ the y_i column of the rank-one term): the d + 1 directions whose eigenvalues
// grow like n.  Linear panels: Phi = y o [X, 1] is exact.  The other kernels run unpreconditioned.
struct as_pc {
    int m = 0;               // features
    int64_t mp = 0;          // m padded to the factorisation block
    float *Phi = nullptr;    // m8 x ldN fp32, feature-major (a column of the N x m matrix is contiguous; rows m .. m8 are zero).
                             // fp32: P = D + Phi Phi' only has to be positive definite and the same on every rank, and a relative
                             // error of 6e-8 on directions whose eigenvalues are ~1e5 stays far below the bulk (~1); the two passes
                             // over Phi per inner iteration move half the bytes (round 4)
    int64_t m8 = 0;          // m rounded up to the feature group of the t = Phi' D^-1 r kernel
    double *tpart = nullptr; // nblk x mp: per-sample-block partial sums of t
    double *dinv = nullptr;  // ldN: 1 / D_i
    double *z = nullptr;     // ldN: preconditioned residual
    double *Gpart = nullptr; // slices x mp x mp partial Gram sums
    double *Ginv = nullptr;  // mp x mp, full symmetric storage: G^-1, G = I + Phi_A' D_A^-1 Phi_A of the free set in `prev`.  Rebuilt
                             // from a Cholesky factor of G summed afresh (first iteration, every 128th, > 64 changes at once, after a
                             // failed update) and carried between rebuilds by Sherman-Morrison updates, one per sample that entered
                             // or left the free set, in index order (as_pc_sm_kernel)
    double *u = nullptr;     // mp: G^-1 t
    int *sm_fail = nullptr;  // device flag: an update met a denominator <= 1e-8 (the caller rebuilds)
    unsigned char *prev = nullptr;   // ldN: the free set G^-1 was last brought up to
    int *chg = nullptr;      // [0] changed indices since then, [1] need a full rebuild, [2 ..] index and sign (+1 freed / -1 bound)
    int host_chg[2] = {0, 1};        // chg[0 .. 1] as read at the top of the outer iteration
    int age = 0;             // outer iterations since the last full rebuild (rounding of the rank-one updates); 0: rebuild now
    long long rebuilds = 0;
    double *cls = nullptr;   // class statistics (as_pc_class_kernel), BQ_SVC + RBF only
    bq_chol_ws *ws = nullptr;
    // the implicit order-2 remainder behind a degree-1 Chebyshev polynomial (bq_as_pc2.hip); null: the explicit model alone
    struct as_pc2 *r2 = nullptr;
    int top0 = 0, ntop = 0;       // the Phi_top columns (the projected order-2 directions) of the explicit model
    double *y1 = nullptr, *v2 = nullptr, *z2 = nullptr, *ones = nullptr;   // ldN each: y = P1^-1 r, R y, P1^-1 R y; weights 1
    double *ttop = nullptr;       // mp: Phi_top' y
    // the explicit model's two passes over Phi are sums over / values of SAMPLES: sharded by the canonical sample segments like the
    // remainder (round 6) — a rank walks its own sample blocks, the per-segment sums of t (S x mp doubles) and the rank's z are
    // gathered, everything is added / unpacked in segment order on every rank (and in the same order on one rank)
    as_pc_part part;
    double *tg = nullptr;         // (world * cmax) x mp: per-segment sums of t = Phi' D^-1 r, gathered
    double *zg = nullptr;         // (world * cmax) x maxlen x 1024: z in the gathered layout
    long long lambda_nA = 0;      // size of the free set the spectrum bound of the remainder was estimated on (a larger set needs a new one)
};

struct as_ws {
    int *idx = nullptr;        // compacted free set
    int *ints = nullptr;       // [0] nA, [1] nB, [2] feasible, [3] h_lower, [4] h_upper, [5] nL_new, [6] nU_new
    double *cand = nullptr;    // candidate point (ldN)
    double *z = nullptr;       // bound contribution vector (ldN)
    double *Qz = nullptr;      // (ldN)
    double *x_eval = nullptr;  // x / g at the top of the current iteration (what a callback must see)
    double *g_eval = nullptr;
    // PINNED host copies of the per-iteration records: a device-to-host copy into pageable memory goes through the runtime's
    // staging path and cost ~30 us of stream idle per look at the device (three looks per ActiveSet iteration: 0.18 ms of a 1.4 ms
    // iteration at n = 20 000, profiles/r04/as_n20k_stream_idle_before.txt); into pinned memory it is a plain DMA
    int *host_ints = nullptr;        // 32 ints: the copy of `ints`
    bq_scal *host_scal = nullptr;    // the copy of the solver's device scalars (moved into s->host after the wait)
    int *host_info = nullptr;        // 8 ints: [0] factorisation info, [1] preconditioner info, [2] update failure, [4 .. 5] chg[0 .. 1]
    as_cg_scal *host_cg = nullptr;   // the inner solver's scalars at the end of a solve
    // The dense iteration's three looks at the device without a copy command or a stream drain (second half of round 4): host_ints /
    // host_scal (and the kept-factor path's small_pin) are MAPPED, coherent pinned memory; the kernel that completes a record stores
    // it there itself and then posts a sequence number into `mail` — [0] top of the iteration (as_top_kernel), [1] the dot
    // products of a new slot (as_schur_dots_kernel), [2] the candidate's feasibility (as_cand_scatter_kernel) — on which the host
    // spins (bq_ctx_wait_flag).  hook as_mailbox=0: copies + hipStreamSynchronize as before.
    int *mail = nullptr;
    int *mail_d = nullptr, *host_ints_d = nullptr, *host_scal_d = nullptr;   // the device's addresses of mail / host_ints / host_scal
    int mail_seq[3] = {0, 0, 0};
    unsigned int *mail_ticket = nullptr;   // device: last-workgroup tickets of the two multi-block posters
    bool mailbox = true;
    // product-free f of a ratio step (as_step_min_kernel): allowed at all / a run is open / its length since f was last formed by a
    // product / how many steps went without a product (bq_solver_counter)
    bool f_chain = true, chain_ok = false;
    double *g0 = nullptr;             // the gradient at the starting point (device, ldN)
    const double *gref = nullptr;     // the gradient the run scales: g0 until the first release, s->g after it
    int chain_len = 0;
    long long chain_steps = 0;
    // BQ_AS_TIMING=1: where the host's time goes per kept-factor iteration (printed by bq_as_free): [0] wait for the top record,
    // [1] host work up to the launch of the dot products, [2] wait for them, [3] the small system on the host, [4] launches up
    // to the candidate, [5] wait for its record, [6] launches of the branch, [7] iterations counted
    double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool timing = false;
    double tm_last = 0.0, tm_classic = 0.0, classic_order = 0.0;   // iterations that factorise Q[A,A] afresh: host time from their
    long long n_classic = 0;                                       // launch to the end of bq_as_iterate, and the mean order
    long long minres_calls = 0;
    struct as_schur *sch = nullptr;   // factor re-use (Schur-complement updates of a base factorisation)
    int last_branch = -1;             // what the previous iteration did: 1 release, 0 ratio step + absorb, -1 nothing yet
    // conjugate-gradient inner solver (BQ_AS_CG)
    double *dlt = nullptr, *r = nullptr, *pv = nullptr, *Qp = nullptr, *sol = nullptr;
    as_cg_scal *cg = nullptr;
    int *cg_flag_host = nullptr;   // pinned: {done, info}
    hipEvent_t cg_event = nullptr; // recorded behind the copy of the flag (lagged polling of the inner iteration)
    long long cg_iters = 0;
    double *Qdl = nullptr, *Qcand = nullptr;   // Q delta accumulated over the inner iterations; Q cand = Q z + Q delta
    bool incq = true;              // hook as_cg_incq=0: a fresh product Q x after every outer iteration (round 2)
    bool colq = false;             // the start product of a warm-started solve is Q cand + a few columns of Q formed from X
    double *sq = nullptr;          // ldN: squared row norms of X (the columns' RBF distances)
    int *zchg = nullptr;           // [0] count, [1] 1 = columns suffice (the start product is skipped), [2 ..] indices
    double *zdl = nullptr;         // bound - cand of those indices
    int since_refresh = 0;         // outer iterations since Q x was last formed by a product
    bool anchor = false;           // the next solve forms its start product Q z by a real product (re-anchors Q z -> Q cand -> Q z ...)
    long long pc_rebuilds = 0, pc_dropped = 0;   // Woodbury system not positive definite: G summed afresh / preconditioner given up
    as_pc *pc = nullptr;           // null: plain conjugate gradients
    bool have_cand = false;        // w->cand holds the candidate of the previous outer iteration (the warm start)
    bool warm = true;              // hook as_cg_warm=0: start every inner solve from the current point
};


__device__ __forceinline__ double as_wsum_any(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double as_wmin(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}

// a record for the host: system-scope release store of its sequence number, after the data (as_ws::mail)
__device__ __forceinline__ void as_post(int *flag, int seq) {
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one wave; lane 0 opens the iteration's record, then (mailbox) the wave hands `ints` and the solver's scalars to the host
// ---------------------------------------------------------------------------------------------------------------
// The per-iteration O(N) steps, multi-block (round 2).  Their single-block predecessors walked N elements with 256
// threads: 35-110 us each at n = 20 000, four of them per iteration = 16 % of an ActiveSet iteration once the triangular
// sweeps were fixed.  Same arithmetic, same results (minima, counts and index lists do not depend on the block order);
// the last-finishing block of a launch closes the step (fixed-order final reduction over the per-block partials).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool as_last_block_mb(unsigned int *ticket) {
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();   // this block's partials are visible device-wide before the ticket is taken
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last) {
        __threadfence();
        if (threadIdx.x == 0) *ticket = 0u;
    }
    return last != 0;
}

__device__ __forceinline__ double as_wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double as_block_sum(double v, double *sh) {
    v = as_wsum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double as_final_sum(const double *part, int64_t nblk, double *sh) {
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < nblk; i += BQ_VEC_BLOCK) a += part[i];
    return as_block_sum(a, sh);
}
__device__ __forceinline__ bool as_last_block(unsigned int *ticket) {
    __shared__ int last;
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last) __threadfence();
    return last != 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Short, index-ordered lists of the samples where a per-sample predicate holds (which variables moved since the previous outer
// iteration), multi-block: pass 1 counts per block of 1024 samples and the last block to finish turns the counts into offsets
// (the total and the "too many" verdict with them); pass 2 writes entry `offset + rank inside the block` when it is below the
// list's capacity.  Index order = (block, item j, wave, lane) as everywhere in these kernels, so the list — and what is done in
// its order — is the same on every rank and for every launch geometry.  (Round 3: one workgroup walked all N samples with three
// barriers per 1024: 0.25 ms per list at n = 250 000.)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void as_list_count(int changed_bits, int *__restrict__ cnt, unsigned int *ticket, int *total_out) {
    // changed_bits: bit j = item j of this thread is in the list
    __shared__ int wt[4];
    int c = __popc(changed_bits);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&cnt[blockIdx.x], wt[0] + wt[1] + wt[2] + wt[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (as_last_block_mb(ticket) && threadIdx.x == 0) {
        int run = 0;
        for (unsigned int b = 0; b < gridDim.x; ++b) {
            const int v = __hip_atomic_load(&cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cnt[b] = run;
            run += v;
        }
        *total_out = run;
    }
}
// position of item j of this thread in the list (valid where bit j of changed_bits is set)
__device__ __forceinline__ void as_list_positions(int changed_bits, const int *__restrict__ cnt, int pos[BQ_VEC_ITEMS]) {
    __shared__ int wt[BQ_VEC_ITEMS][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int within[BQ_VEC_ITEMS];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const unsigned long long bal = __ballot((changed_bits >> j) & 1);
        within[j] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wt[j][wv] = __popcll(bal);
    }
    __syncthreads();
    int off = cnt[blockIdx.x];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        int o = off;
        for (int k = 0; k < wv; ++k) o += wt[j][k];
        pos[j] = o + within[j];
        off += wt[j][0] + wt[j][1] + wt[j][2] + wt[j][3];
    }
}

static __global__ void as_copy_kernel(int64_t N, const double *__restrict__ src, double *__restrict__ dst) {
    VEC_LOOP(i) {
        if (i < N) dst[i] = src[i];
    }
}

// the kept-factor candidate in two multi-block steps: cand = bound values / 0 everywhere (+ the feasibility flag raised), then
// the base variables that are still free and the freed ones scatter their values and lower the flag where a value leaves the box
static __global__ void as_cand_fill_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                    const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ cand,
                                    int *__restrict__ ints) {
    VEC_LOOP(i) {
        if (i < N) cand[i] = mU[i] ? ub[i] : (mL[i] ? lb[i] : 0.0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ints[2] = 1;
}
// ... and the plain restricted solve: the solution on the compacted free set (ints[0] entries of idx) scatters the same way
static __global__ void as_cand_scatter_idx_kernel(const int *__restrict__ idx, int *__restrict__ ints, const double *__restrict__ sol,
                                           const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ cand) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= ints[0]) return;
    const int i = idx[a];
    const double v = sol[a];
    cand[i] = v;
    if (!(v <= ub[i] + ACT_TOL && v >= lb[i] - ACT_TOL)) ints[2] = 0;   // benign race: every writer stores 0
}
// ---- host-side helpers ----------------------------------------------------------------------------------------
static inline as_ws *get_ws(bq_solver *s) { return reinterpret_cast<as_ws *>(s->as_ws); }
static inline bool as_env_on(const char *name) {   // a user-level switch: BQ_AS_CG_PC=0
    const char *e = getenv(name);
    return !(e && atoi(e) == 0);
}
// mapped, coherent pinned memory: the device stores into / loads from it in place (as_ws::mail)
constexpr unsigned int AS_MAPPED = hipHostMallocMapped | hipHostMallocCoherent;
template <typename P>
static inline P *as_dev(P *host) {
    void *d = nullptr;
    return (host != nullptr && hipHostGetDevicePointer(&d, host, 0) == hipSuccess) ? static_cast<P *>(d) : host;
}
// one look of the host at the device: the record `which` (as_ws::mail) has been posted / the stream has drained behind the copies
static inline int as_look(bq_ctx *ctx, as_ws *w, int which) {
    if (!w->mailbox) return bq_ctx_sync(ctx);
    return bq_ctx_wait_flag(ctx, w->mail + which, w->mail_seq[which]);
}

static inline void as_tick(as_ws *w, int slot) {   // BQ_AS_TIMING: the time since the previous tick goes to `slot`
    if (!w->timing) return;
    const double now = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (slot >= 0) w->tm[slot] += now - w->tm_last;
    w->tm_last = now;
}

// ---- factor re-use (bq_as_schur.hip) ---------------------------------------------------------------------------
struct as_schur {
    bool valid = false;
    int64_t n0 = 0, np0 = 0, cap = 0;
    int *idx0 = nullptr;              // device: the base set, ascending
    int *pos0 = nullptr;              // device: variable -> position in the base, -1 outside
    std::vector<int> hpos0;           // host mirror of pos0
    std::vector<int> kind, var;       // slots: kind 0 = base variable now at a bound, 1 = variable freed since
    std::vector<double> C;            // AS_SCHUR_MAX x AS_SCHUR_MAX, symmetric, host
    std::vector<double> Lc, Dc;       // C = Lc diag(Dc) Lc' of its leading ldl_n rows (unit lower Lc, no pivoting: C is
    int ldl_n = 0;                    // symmetric quasi-definite), grown by one row per new slot
    double *U = nullptr, *W = nullptr;   // device: AS_SCHUR_MAX x cap columns u_k and Q00^-1 u_k
    double *y0 = nullptr, *y = nullptr;  // device: cap
    double *small = nullptr;          // device: AS_SCHUR_MAX results / coefficients
    int *meta = nullptr;              // device: kind[], var[] of the slots (2 x AS_SCHUR_MAX)
    // pinned staging of the per-iteration transfers (slot table up, dot products down, coefficients up): asynchronous copies
    // from / to pageable stack arrays needed a stream synchronisation each just to keep the array alive
    // (mailbox, as_ws::mail: the three are mapped, coherent pinned memory and the kernels use them in place — the slot table and
    // the coefficients are READ by the kernels straight from the host's buffer, the dot products are WRITTEN there: no copy commands)
    int *meta_pin = nullptr;
    double *small_pin = nullptr, *coef_pin = nullptr;
    int *meta_pin_d = nullptr;                                // ... and the device's addresses of the three
    double *small_pin_d = nullptr, *coef_pin_d = nullptr;
    long long refreshes = 0, reused = 0;
    long long rows_extended = 0, rows_solved = 0, drops = 0;   // BQ_AS_TIMING: rows the small factorisation (re)built / orders solved / slots dropped
    bool timing = false;                  // BQ_AS_TIMING: host microseconds spent in ...
    double t_ldl[4] = {0, 0, 0, 0};       // ... the new row | the forward solve | the backward solve | the residual check
    double t_c = 0.0;                     // ... storing the new row / column of C
    bool y0_valid = false;   // y0 = Q00^-1 b0 is current: b0 only moves when a variable OUTSIDE the base changes sides
};
int as_schur_min();
bool as_schur_enabled();
void as_schur_free(as_ws *w);
int as_schur_step(bq_solver *s, as_ws *w, int64_t nA, bool *solved);
double ldl_now();

// ---- the outer iteration's launch helpers (bq_as.hip) -----------------------------------------------------------
void as_launch_compact(bq_solver *s, as_ws *w, hipStream_t st);
void as_launch_step(bq_solver *s, as_ws *w, hipStream_t st, int chain = 0);
void as_launch_absorb(bq_solver *s, as_ws *w, hipStream_t st);
void as_launch_release(bq_solver *s, as_ws *w, hipStream_t st);
int as_eval_f(bq_solver *s, double *g_out);   // Qd = Q x ; f = 1/2 x'Qx + q'x -> sc->f ; optionally g = Qx + q
int as_finish_iteration(bq_solver *s, as_ws *w, hipStream_t st, bool exact);

// ---- conjugate gradients (bq_as_cg.hip) -----------------------------------------------------------------------
int as_cg_create(bq_solver *s, as_ws *w);      // buffers, switches and the preconditioner of a BQ_AS_CG solver (bq_as_start)
int as_cg_iterate(bq_solver *s, as_ws *w);     // the body of one outer iteration once the top record has been read

// ---- preconditioner (bq_as_pc.hip, bq_as_pc2.hip) ------------------------------------------------------------
constexpr int PC_MAX_M = 1024;   // features the apply kernels keep in LDS
typedef float as_f4 __attribute__((ext_vector_type(4)));
void as_pc_make_part(const bq_ctx *ctx, int64_t nblk, as_pc_part *pt);   // the canonical sample segments of nblk blocks of 1024 samples
int as_pc2_create(bq_solver *s, double *bdiag_out, int64_t tail, as_pc2 **out);
double *as_pc2_tail(const as_pc2 *r, int64_t *gstride);
void as_pc2_free(as_pc2 *r);
int as_pc2_bpart(bq_solver *s, as_pc2 *r, const double *y, const as_cg_scal *cg);
const double *as_pc2_ypart(const as_pc2 *r, int *tiles);
const double *as_pc2_c(const as_pc2 *r);
const as_pc_part *as_pc2_part(const as_pc2 *r);
double *as_pc2_vg(const as_pc2 *r);
double as_pc2_lambda(const as_pc2 *r);
void as_pc2_set_lambda(as_pc2 *r, double lambda);
void as_pc2_coefs(const as_pc2 *r, double *alpha, double *beta);
int as_pc_create(bq_solver *s, as_pc **out);
void as_pc_free(as_pc *pc);
void as_pc_track(bq_solver *s, as_ws *w, hipStream_t st);   // which samples entered / left the free set since G^-1 was brought up to date
int as_pc_update(bq_solver *s, as_ws *w);                   // G^-1 for this outer iteration's free set (rebuild or rank-one updates)
int as_pc_apply(bq_solver *s, as_ws *w, int first);         // z = P_AA^-1 r (+ r'z and beta on the device)
