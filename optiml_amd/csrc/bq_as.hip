// Active-set driver for the box QP: mask bookkeeping, gathers, KKT sign scan and the restricted Cholesky solve on
// the device.  Restates optiml/opti/constrained/active_set.py:84-230.
//
// Per iteration (reference line numbers in parentheses):
//   top      record (iter, f, |L|+|U|), stop at max_iter (:103-114)
//   restrict idx = compact(A);  z = ub on U, lb on L, 0 on A;  rhs_A = -(q_A + (Q z)_A)  — one panel product instead
//            of the reference's two gathered products Q[A,U] ub_U + Q[A,L] lb_L (:132-136)
//   solve    H = Q[A,A] gathered straight from the resident panel, blocked Cholesky, two triangular solves (:141)
//   branch   candidate inside the box (+-1e-12)?  (:153)
//     yes:   x = candidate; f, g = evaluate; release the FIRST index of L with g < -1e-12, else the first of U with
//            g > 1e-12 (Bland), else 'optimal' (:156-189)
//     no:    ratio step towards the candidate, f = evaluate, move variables that hit a bound into L / U (:195-220)
// The restricted order |A| changes every iteration, so the host reads small records per iteration (|A|, the dot products of a
// new slot, {pivot info, feasible}) — posted by the kernels into mapped pinned memory, three sequence numbers the host spins on
// (as_ws::mail: no copy commands, no stream drains); everything else stays on the stream.  f(x) after a ratio step comes from the
// step's own identity (as_step_min_kernel); a panel product is spent on f only at release iterations.  When Q[A,A] is not positive definite the reference
// silently switches to scipy's minres on the normal equations (:142-151): here a persistent single-workgroup MINRES
// kernel (bq_minres.hip) takes over for |A| <= 8192.
//
// Factor re-use (default for every non-empty free set; BQ_AS_SCHUR=0 re-factorises every iteration as the reference does): between
// two consecutive iterations the free set changes by one or a few indices, so the Cholesky factor of a BASE set A0 is
// kept and the current restricted system is solved through its Schur complement — variables of A0 that have reached a
// bound since are pinned by a multiplier row (x_k = bound), variables released since are bordered on:
//     [ Q00  U ] [y]   [b0]        U = [ Q[A0, added] | e_removed ],  V = [ Q[added, added] 0 ; 0 0 ]
//     [ U'   V ] [w] = [b1]        C = V - U' Q00^-1 U  (m x m, m <= 96 ... 512),  w = C^-1 (b1 - U' Q00^-1 b0),  y = Q00^-1 (b0 - U w)
// One new column Q00^-1 u (two triangular sweeps) per changed index and one Q00^-1 b0 per iteration replace the
// |A|^3/3 factorisation; C is kept on the host and solved there.  The base is re-factorised after 160 (|A| < 8 192) to 1536 (|A| >= 80 000) changes, when C
// is numerically singular, or when the classic path is needed (non-positive pivot -> the reference's minres branch).
//
// BQ_AS_CG (SURVEY 7 "hard parts": ActiveSet beyond the sizes a dense factor fits): the same outer logic, but the
// restricted system is solved by conjugate gradients on the masked panel operator v -> m.(Q (m.v)), m = indicator of
// A — no H, no factorisation, one panel product per inner iteration, so it runs on sharded, fp32-stored and streamed
// panels and on any number of ranks (the vectors and scalar recurrences are replicated, reductions fixed-order).  The
// iteration starts from the candidate of the previous outer iteration (the free set has moved by one index since) and, on
// RBF / linear panels, is preconditioned by a diagonal + low-rank model of Q[A,A] built from explicit features (struct
// as_pc): at BASELINE config 5 that takes the inner iterations per outer iteration from ~80 to ~30 (round 3).
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
    unsigned int *tticket = nullptr;
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
    // spins (bq_ctx_wait_flag).  BQ_AS_MAILBOX=0: copies + hipStreamSynchronize as before.
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
    bool incq = true;              // BQ_AS_CG_INCQ=0: a fresh product Q x after every outer iteration (round 2)
    bool colq = false;             // the start product of a warm-started solve is Q cand + a few columns of Q formed from X
    double *sq = nullptr;          // ldN: squared row norms of X (the columns' RBF distances)
    int *zchg = nullptr;           // [0] count, [1] 1 = columns suffice (the start product is skipped), [2 ..] indices
    double *zdl = nullptr;         // bound - cand of those indices
    int since_refresh = 0;         // outer iterations since Q x was last formed by a product
    bool anchor = false;           // the next solve forms its start product Q z by a real product (re-anchors Q z -> Q cand -> Q z ...)
    long long pc_rebuilds = 0, pc_dropped = 0;   // Woodbury system not positive definite: G summed afresh / preconditioner given up
    as_pc *pc = nullptr;           // null: plain conjugate gradients
    bool have_cand = false;        // w->cand holds the candidate of the previous outer iteration (the warm start)
    bool warm = true;              // BQ_AS_CG_WARM=0: start every inner solve from the current point
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
__global__ __launch_bounds__(64) void as_top_kernel(bq_scal *sc, int *__restrict__ ints, bq_iter_stat *stats, int n_all,
                                                    int *__restrict__ mail_ints, int *__restrict__ mail_scal, int *mail, int seq) {
    if (threadIdx.x == 0) {
        ints[27] = n_all;   // as_release_mb_kernel: running minima of the candidate indices
        ints[28] = n_all;
        if (!sc->done) {
            const long long row = sc->iter - sc->stat_base;
            if (row >= 0 && row < sc->stat_cap) {
                bq_iter_stat st;
                st.iter = sc->iter;
                st.f = sc->f;
                st.r1 = (double)ints[1];
                st.r2 = -1.0;
                st.r3 = 0.0;
                stats[row] = st;
            }
            if (sc->iter >= sc->max_iter) {
                sc->status = BQ_STATUS_STOPPED;
                sc->done = 1;
            }
        }
        __threadfence();
    }
    if (mail == nullptr) return;
    __syncthreads();
    const int l = threadIdx.x;
    if (l < 32) mail_ints[l] = __hip_atomic_load(ints + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int j = l; j < (int)(sizeof(bq_scal) / sizeof(int)); j += 64)
        mail_scal[j] = __hip_atomic_load(reinterpret_cast<int *>(sc) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    if (l == 0) as_post(mail, seq);
}

__global__ void as_make_z_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                 const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ z) {
    VEC_LOOP(i) {
        if (i < N) z[i] = mU[i] ? ub[i] : (mL[i] ? lb[i] : 0.0);
    }
}

// rhs[a] = -(q[idx[a]] + Qz[idx[a]]) for a < nA, 0 on the pad
__global__ void as_gather_rhs_kernel(const int *__restrict__ ints, const int *__restrict__ idx,
                                     const double *__restrict__ q, const double *__restrict__ Qz,
                                     double *__restrict__ rhs, int64_t np) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= np) return;
    const int nA = ints[0];
    rhs[a] = (a < nA) ? -(q[idx[a]] + Qz[idx[a]]) : 0.0;
}

__global__ void as_copy_kernel(int64_t N, const double *__restrict__ src, double *__restrict__ dst) {
    VEC_LOOP(i) {
        if (i < N) dst[i] = src[i];
    }
}
// the iteration's snapshot of x and g in one launch
__global__ void as_copy2_kernel(int64_t N, const double *__restrict__ a, double *__restrict__ da, const double *__restrict__ b,
                                double *__restrict__ db) {
    VEC_LOOP(i) {
        if (i < N) {
            da[i] = a[i];
            db[i] = b[i];
        }
    }
}

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

// free-set compaction, pass 1: part[b] = number of free indices in block b's 1024 elements; the last block turns the counts
// into exclusive offsets (in place) and writes ints[0] = |A|, ints[1] = |L| + |U|
__global__ __launch_bounds__(256) void as_count_free_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                            const unsigned char *__restrict__ mU, int *__restrict__ cnt,
                                                            int *__restrict__ ints, unsigned int *ticket) {
    __shared__ int wt[4];
    int c = 0;
    VEC_LOOP(i) c += (i < N && !(mL[i] | mU[i])) ? 1 : 0;
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
        ints[0] = run;
        ints[1] = (int)N - run;
    }
}
// pass 2: idx[offset of the block + rank inside the block] = i, ascending (element j of thread t: index b*1024 + j*256 + t)
__global__ __launch_bounds__(256) void as_write_free_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                            const unsigned char *__restrict__ mU, const int *__restrict__ cnt,
                                                            int *__restrict__ idx) {
    __shared__ int wt[BQ_VEC_ITEMS][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int flag[BQ_VEC_ITEMS], within[BQ_VEC_ITEMS];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        flag[j] = (i < N && !(mL[i] | mU[i])) ? 1 : 0;
        const unsigned long long bal = __ballot(flag[j]);
        within[j] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wt[j][wv] = __popcll(bal);
    }
    __syncthreads();
    int off = cnt[blockIdx.x];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        int o = off;
        for (int w = 0; w < wv; ++w) o += wt[j][w];
        if (flag[j]) idx[o + within[j]] = (int)((int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x);
        off += wt[j][0] + wt[j][1] + wt[j][2] + wt[j][3];
    }
}

// ratio test of the step towards the candidate: sc->step = min over the free set (active_set.py:166-173), partial minima per
// block, the last block takes the minimum over them.
//
// chain != 0 — f(x + t d) WITHOUT a panel product (the second half of round 4).  The reference evaluates f afresh after the ratio
// step (active_set.py:172-176: three products with Q).  But d = cand - x lives on the free set A and cand solves the restricted
// system, so Q_AA d_A = -g_A(x) (a Newton step on A), hence  d'Qd = -g_A'd_A  and
//       f(x + t d) = f(x) + t g_A'd_A + t^2/2 d'Qd = f(x) + (t - t^2/2) g_A'd_A,      g_A(x + t d) = (1 - t) g_A(x):
// along a run of ratio steps the gradient on the (shrinking) free set is the gradient g0 of the last RELEASE iteration — the one
// s->g still holds, fresh from that iteration's product — times gamma = prod (1 - t_j).  So the kernel also sums g0_A'd_A over the
// free set (fixed order: per block, then over the blocks) and its last block moves f and gamma (sc->aux[0]; as_release_mb_kernel
// sets it back to 1).  The host asks for this only while every candidate since that release came from a factorisation (not from
// the minimum-residual branch) and forms f by a product again every 64th step of a run (chain == 2: only gamma moves; as_finish_iteration);
// BQ_AS_F_CHAIN=0: never.
__global__ __launch_bounds__(256) void as_step_min_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                          const unsigned char *__restrict__ mU, const double *__restrict__ cand,
                                                          const double *__restrict__ lb, const double *__restrict__ ub,
                                                          const double *__restrict__ x, const double *__restrict__ g0, double *part,
                                                          double *part_s, bq_scal *sc, int chain) {
    __shared__ double sh[4], shs[4];
    double rmin = INFINITY, sgd = 0.0;
    VEC_LOOP(i) {
        if (i < N && !(mL[i] | mU[i])) {
            const double d = cand[i] - x[i];
            if (d > 0.0) rmin = fmin(rmin, (ub[i] - x[i]) / d);
            if (d < 0.0) rmin = fmin(rmin, (lb[i] - x[i]) / d);
            if (chain == 1) sgd = fma(g0[i], d, sgd);
        }
    }
    rmin = as_wmin(rmin);
    if (chain) sgd = as_wsum_any(sgd);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = rmin;
        shs[threadIdx.x >> 6] = sgd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[blockIdx.x], fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (chain) __hip_atomic_store(&part_s[blockIdx.x], ((shs[0] + shs[1]) + shs[2]) + shs[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (as_last_block_mb(&sc->ticket[0])) {
        double m = INFINITY, a = 0.0;
        for (unsigned int b = threadIdx.x; b < gridDim.x; b += 256) {
            m = fmin(m, __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (chain) a += __hip_atomic_load(&part_s[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        m = as_wmin(m);
        if (chain) a = as_wsum_any(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            sh[threadIdx.x >> 6] = m;
            shs[threadIdx.x >> 6] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
            sc->step = t;
            if (chain == 1) {
                const double gd = sc->aux[0] * (((shs[0] + shs[1]) + shs[2]) + shs[3]);
                sc->f = sc->f + (t - 0.5 * t * t) * gd;
            }
            if (chain) sc->aux[0] = sc->aux[0] * (1.0 - t);   // chain == 2: f comes from a product this time, the run goes on
        }
    }
}
__global__ void as_step_apply_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                     const double *__restrict__ cand, double *__restrict__ x, const bq_scal *sc) {
    const double t = sc->step;
    VEC_LOOP(i) {
        if (i < N && !(mL[i] | mU[i])) x[i] = x[i] + __dmul_rn(t, cand[i] - x[i]);
    }
}

// free variables that reached a bound move into L / U (L first, as the reference); ints[5], ints[6] = how many, ints[8] = the
// total, ints[9 ..] = up to 16 of their indices (the host sorts them); the last block closes the iteration.
// ints[26] is the slot counter of this launch (zeroed by the last block for the next one; ints[7] belongs to MINRES).
__global__ __launch_bounds__(256) void as_absorb_mb_kernel(int64_t N, unsigned char *__restrict__ mL, unsigned char *__restrict__ mU,
                                                           const double *__restrict__ x, const double *__restrict__ lb,
                                                           const double *__restrict__ ub, bq_scal *sc, int *__restrict__ ints,
                                                           int *__restrict__ cnt, bq_iter_stat *stats) {
    __shared__ int wl[4], wu[4];
    int nl = 0, nu = 0;
    VEC_LOOP(i) {
        if (i < N && !(mL[i] | mU[i])) {
            bool hit = false;
            if (x[i] <= lb[i] + ACT_TOL) {
                mL[i] = 1;
                ++nl;
                hit = true;
            } else if (x[i] >= ub[i] - ACT_TOL) {
                mU[i] = 1;
                ++nu;
                hit = true;
            }
            if (hit) {
                const int slot = atomicAdd(&ints[26], 1);
                if (slot < 16) ints[9 + slot] = (int)i;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nl += __shfl_down(nl, off, 64);
        nu += __shfl_down(nu, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        wl[threadIdx.x >> 6] = nl;
        wu[threadIdx.x >> 6] = nu;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&cnt[2 * blockIdx.x], wl[0] + wl[1] + wl[2] + wl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cnt[2 * blockIdx.x + 1], wu[0] + wu[1] + wu[2] + wu[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (as_last_block_mb(&sc->ticket[1]) && threadIdx.x == 0) {
        int tl = 0, tu = 0;
        for (unsigned int b = 0; b < gridDim.x; ++b) {
            tl += __hip_atomic_load(&cnt[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tu += __hip_atomic_load(&cnt[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ints[5] = tl;
        ints[6] = tu;
        ints[8] = __hip_atomic_load(&ints[26], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ints[26], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            stats[row].r2 = 0.0;
            stats[row].r3 = (double)(((long long)tl << 32) | (long long)tu);
        }
        sc->iter += 1;
    }
}

// first index of L with g < -tol and of U with g > tol (Bland's rule, active_set.py:178-199): block minima meet in two atomic
// minima (ints[27], ints[28], reset by as_top_kernel); the last block releases the variable, writes the record and closes the
// iteration — or declares the point optimal
__global__ __launch_bounds__(256) void as_release_mb_kernel(int64_t N, const double *__restrict__ g, unsigned char *__restrict__ mL,
                                                            unsigned char *__restrict__ mU, bq_scal *sc, int *__restrict__ ints,
                                                            bq_iter_stat *stats) {
    __shared__ int sl[4], su[4];
    int hl = (int)N, hu = (int)N;
    VEC_LOOP(i) {
        if (i < N) {
            if (mL[i] && g[i] < -ACT_TOL && (int)i < hl) hl = (int)i;
            if (mU[i] && g[i] > ACT_TOL && (int)i < hu) hu = (int)i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        hl = min(hl, __shfl_down(hl, off, 64));
        hu = min(hu, __shfl_down(hu, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        sl[threadIdx.x >> 6] = hl;
        su[threadIdx.x >> 6] = hu;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hl = min(min(sl[0], sl[1]), min(sl[2], sl[3]));
        hu = min(min(su[0], su[1]), min(su[2], su[3]));
        if (hl < (int)N) atomicMin(&ints[27], hl);
        if (hu < (int)N) atomicMin(&ints[28], hu);
    }
    if (as_last_block_mb(&sc->pad1[1]) && threadIdx.x == 0) {
        const long long ghl = __hip_atomic_load(&ints[27], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long ghu = __hip_atomic_load(&ints[28], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ints[3] = (int)ghl;
        ints[4] = (int)ghu;
        sc->aux[0] = 1.0;   // g is fresh: a run of product-free ratio steps starts here (as_step_min_kernel)
        const long long row = sc->iter - sc->stat_base;
        const bool rec = row >= 0 && row < sc->stat_cap;
        if (ghl < N) {
            mL[ghl] = 0;
            if (rec) {
                stats[row].r2 = 1.0;
                stats[row].r3 = (double)ghl;
            }
            sc->iter += 1;
        } else if (ghu < N) {
            mU[ghu] = 0;
            if (rec) {
                stats[row].r2 = 2.0;
                stats[row].r3 = (double)ghu;
            }
            sc->iter += 1;
        } else {
            if (rec) stats[row].r2 = 3.0;
            sc->status = BQ_STATUS_OPTIMAL;
            sc->done = 1;
        }
    }
}

// the kept-factor candidate in two multi-block steps: cand = bound values / 0 everywhere (+ the feasibility flag raised), then
// the base variables that are still free and the freed ones scatter their values and lower the flag where a value leaves the box
__global__ void as_cand_fill_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                    const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ cand,
                                    int *__restrict__ ints) {
    VEC_LOOP(i) {
        if (i < N) cand[i] = mU[i] ? ub[i] : (mL[i] ? lb[i] : 0.0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ints[2] = 1;
}
// ... and the plain restricted solve: the solution on the compacted free set (ints[0] entries of idx) scatters the same way
__global__ void as_cand_scatter_idx_kernel(const int *__restrict__ idx, int *__restrict__ ints, const double *__restrict__ sol,
                                           const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ cand) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= ints[0]) return;
    const int i = idx[a];
    const double v = sol[a];
    cand[i] = v;
    if (!(v <= ub[i] + ACT_TOL && v >= lb[i] - ACT_TOL)) ints[2] = 0;   // benign race: every writer stores 0
}
__global__ void as_cand_scatter_kernel(int64_t n0, int m, const int *__restrict__ idx0, const int *__restrict__ meta,
                                       const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                       const double *__restrict__ lb, const double *__restrict__ ub,
                                       const double *__restrict__ y, const double *__restrict__ coef,
                                       double *__restrict__ cand, int *__restrict__ ints, unsigned int *ticket,
                                       int *__restrict__ mail_ints, int *mail, int seq) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (a < n0) {
        const int i = idx0[a];
        if (!(mL[i] | mU[i])) {
            const double v = y[a];
            cand[i] = v;
            bad = !(v <= ub[i] + ACT_TOL && v >= lb[i] - ACT_TOL);
        }
    }
    if (a < m && meta[a] == 1) {
        const int i = meta[AS_SCHUR_MAX + a];
        const double v = coef[a];
        cand[i] = v;
        bad = bad || !(v <= ub[i] + ACT_TOL && v >= lb[i] - ACT_TOL);
    }
    if (bad) ints[2] = 0;   // benign race: every writer stores 0
    if (mail == nullptr) return;
    // the last workgroup to get here hands the record (feasibility flag and all) to the host
    __shared__ int last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 32) mail_ints[threadIdx.x] = __hip_atomic_load(ints + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    if (threadIdx.x == 0) {
        *ticket = 0;
        as_post(mail, seq);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conjugate gradients on Q[A,A] (BQ_AS_CG).  Vectors are full length and zero outside A.
// ---------------------------------------------------------------------------------------------------------------
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

// xt = the current point with the bound values on L and U (what the reference substitutes, active_set.py:132-136)
__global__ void as_make_xt_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                  const double *__restrict__ lb, const double *__restrict__ ub,
                                  const double *__restrict__ x, double *__restrict__ xt) {
    VEC_LOOP(i) {
        if (i < N) xt[i] = mU[i] ? ub[i] : (mL[i] ? lb[i] : x[i]);
    }
}

// r = p = -(Q xt + q) on A, 0 elsewhere; delta = 0; the stop level is rtol * (|(Q xt)_A| + |q_A|)
// (Qlevel: the product the stop level is taken from — Q x of the CURRENT point when the iteration starts somewhere else, so
// that a warm start does not change what "solved to rtol" means)
__global__ void as_cg_init_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                  const double *__restrict__ Qxt, const double *__restrict__ Qlevel,
                                  const double *__restrict__ q, double *__restrict__ dlt,
                                  double *__restrict__ r, double *__restrict__ pv, double *part, int64_t nblk,
                                  as_cg_scal *cg, double rtol, long long max_iters, int pc) {
    __shared__ double sh[4];
    double srr = 0.0, sqx = 0.0, sq = 0.0;
    VEC_LOOP(i) {
        if (i < N) {
            const bool fr = !(mL[i] | mU[i]);
            const double a = Qxt[i], b = q[i];
            const double ri = fr ? -(a + b) : 0.0;
            r[i] = ri;
            pv[i] = ri;
            dlt[i] = 0.0;
            srr += __dmul_rn(ri, ri);
            if (fr) {
                const double al = Qlevel[i];
                sqx += __dmul_rn(al, al);
                sq += __dmul_rn(b, b);
            }
        }
    }
    srr = as_block_sum(srr, sh);
    sqx = as_block_sum(sqx, sh);
    sq = as_block_sum(sq, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = srr;
        part[nblk + blockIdx.x] = sqx;
        part[2 * nblk + blockIdx.x] = sq;
    }
    if (as_last_block(&cg->ticket[0])) {
        const double rr = as_final_sum(part, nblk, sh);
        const double nqx = as_final_sum(part + nblk, nblk, sh), nq = as_final_sum(part + 2 * nblk, nblk, sh);
        if (threadIdx.x == 0) {
            const double level = rtol * (sqrt(nqx) + sqrt(nq));
            cg->ticket[0] = 0;
            cg->rr = rr;
            cg->tol2 = level * level;
            cg->alpha = 0.0;
            cg->beta = 0.0;
            cg->iters = 0;
            cg->max_iters = max_iters;
            cg->info = 0;
            cg->pc = pc;
            cg->rz = rr;
            cg->done = (rr <= cg->tol2) ? 1 : 0;
        }
    }
}

// alpha = r'r / p'Qp (p vanishes outside A, so the sum needs no mask); a curvature <= 0 means Q[A,A] is not positive definite
__global__ void as_cg_pap_kernel(int64_t N, const double *__restrict__ pv, const double *__restrict__ Qp, double *part,
                                 int64_t nblk, as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double sh[4];
    double s = 0.0;
    VEC_LOOP(i) {
        if (i < N) s += __dmul_rn(pv[i], Qp[i]);
    }
    s = as_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (as_last_block(&cg->ticket[0])) {
        const double pAp = as_final_sum(part, nblk, sh);
        if (threadIdx.x == 0) {
            cg->ticket[0] = 0;
            if (!(pAp > 0.0) || !isfinite(pAp)) {
                cg->info = 1;
                cg->alpha = 0.0;
            } else {
                cg->alpha = (cg->pc ? cg->rz : cg->rr) / pAp;
            }
        }
    }
}

// delta += alpha p;  r -= alpha m.(Qp);  beta = r'r(new) / r'r(old); stop tests.  Qdl += alpha Qp on ALL rows: Q delta, which
// with the start point's product gives Q cand without another product (as_qcand_kernel)
__global__ void as_cg_update_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                    double *__restrict__ dlt, double *__restrict__ r, const double *__restrict__ pv,
                                    const double *__restrict__ Qp, double *__restrict__ Qdl, double *part, int64_t nblk,
                                    as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double sh[4];
    const double alpha = cg->alpha;
    double s = 0.0;
    VEC_LOOP(i) {
        if (i < N) Qdl[i] = Qdl[i] + __dmul_rn(alpha, Qp[i]);
        if (i < N && !(mL[i] | mU[i])) {
            dlt[i] = dlt[i] + __dmul_rn(alpha, pv[i]);
            const double ri = r[i] - __dmul_rn(alpha, Qp[i]);
            r[i] = ri;
            s += __dmul_rn(ri, ri);
        }
    }
    s = as_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (as_last_block(&cg->ticket[1])) {
        const double rr = as_final_sum(part, nblk, sh);
        if (threadIdx.x == 0) {
            cg->ticket[1] = 0;
            if (!cg->pc) cg->beta = cg->rr > 0.0 ? rr / cg->rr : 0.0;   // preconditioned: beta = rz_new / rz (as_pc_apply_kernel)
            cg->rr = rr;
            cg->iters += 1;
            if (cg->info || rr <= cg->tol2 || cg->iters >= cg->max_iters || !isfinite(rr)) cg->done = 1;
        }
    }
}

// p = r + beta p
__global__ void as_cg_dir_kernel(int64_t N, const double *__restrict__ r, double *__restrict__ pv, const as_cg_scal *cg) {
    if (cg->done) return;
    const double beta = cg->beta;
    VEC_LOOP(i) {
        if (i < N) pv[i] = r[i] + __dmul_rn(beta, pv[i]);
    }
}

// Q cand = Q z + Q delta (z: the start point with the bound values, delta: what the iteration added on the free set)
__global__ void as_qcand_kernel(int64_t N, const double *__restrict__ Qz, const double *__restrict__ Qdl, double *__restrict__ Qc) {
    VEC_LOOP(i) {
        Qc[i] = i < N ? Qz[i] + Qdl[i] : 0.0;
    }
}

// after the ratio step x += t (cand - x):  Q x += t (Q cand - Q x)
__global__ void as_qx_lerp_kernel(int64_t N, const bq_scal *sc, const double *__restrict__ Qc, double *__restrict__ Qx) {
    const double t = sc->step;
    VEC_LOOP(i) {
        if (i < N) Qx[i] = Qx[i] + __dmul_rn(t, Qc[i] - Qx[i]);
    }
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

constexpr int AS_MAX_COLS = 16;
// Where the start vector z (the previous candidate with the CURRENT bound values) differs from that candidate: the variables
// that reached a bound in the step since — a handful.
// zchg[1] = 1: at most AS_MAX_COLS of them, so Q z = Q cand + sum_j (bound_j - cand_j) Q[:, j] and the product is skipped.
__device__ __forceinline__ double as_zdiff_of(int64_t i, int64_t N, const unsigned char *mL, const unsigned char *mU,
                                              const double *lb, const double *ub, const double *cand) {
    if (i < N && (mL[i] | mU[i])) return (mU[i] ? ub[i] : lb[i]) - cand[i];
    return 0.0;
}
__global__ __launch_bounds__(256) void as_zdiff_count_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                             const unsigned char *__restrict__ mU, const double *__restrict__ lb,
                                                             const double *__restrict__ ub, const double *__restrict__ cand,
                                                             int *__restrict__ cnt, unsigned int *ticket, int *__restrict__ zchg) {
    int bits = 0;
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        if (as_zdiff_of(i, N, mL, mU, lb, ub, cand) != 0.0) bits |= 1 << j;
    }
    __shared__ int total;
    if (threadIdx.x == 0) total = -1;
    as_list_count(bits, cnt, ticket, &total);
    if (threadIdx.x == 0 && total >= 0) {
        zchg[0] = total;
        zchg[1] = total <= AS_MAX_COLS ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void as_zdiff_write_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                             const unsigned char *__restrict__ mU, const double *__restrict__ lb,
                                                             const double *__restrict__ ub, const double *__restrict__ cand,
                                                             const int *__restrict__ cnt, int *__restrict__ zchg,
                                                             double *__restrict__ zdl) {
    int bits = 0;
    double dl[BQ_VEC_ITEMS];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        dl[j] = as_zdiff_of(i, N, mL, mU, lb, ub, cand);
        if (dl[j] != 0.0) bits |= 1 << j;
    }
    int pos[BQ_VEC_ITEMS];
    as_list_positions(bits, cnt, pos);
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j)
        if (((bits >> j) & 1) && pos[j] < AS_MAX_COLS) {
            zchg[2 + pos[j]] = (int)((int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x);
            zdl[pos[j]] = dl[j];
        }
}

// Qz = Q cand + sum_c dl_c Q[:, j_c] with the columns formed from X (replicated on every rank, so no exchange): the entry the
// panel holds up to the rounding of its own dot products — K as the kernel maps of bq_gram.hip define it, rounded to the
// panel's storage type, then the structure of the dual (bq_qelem.h).  Runs only when as_zdiff_kernel said the columns suffice.
__global__ __launch_bounds__(256) void as_qz_cols_kernel(int64_t n, int64_t d, const double *__restrict__ X, const double *__restrict__ sq,
                                                         const double *__restrict__ sgn, int kernel, double gamma, double coef0,
                                                         int degree, int add_one, double diag_add, int f32,
                                                         const int *__restrict__ zchg, const double *__restrict__ zdl,
                                                         const double *__restrict__ Qc, double *__restrict__ Qz) {
    if (!zchg[1]) return;
    // a WAVE per row, lanes along the features: a row of X is read in 512-byte runs (round 3 gave every lane a row of its own: 64
    // cache lines per load instruction, 0.53 ms for the 0.5 GB of X at BASELINE config 5); the dot product is the lane-strided sum
    // + a halving butterfly, the same on every rank
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int cnt = zchg[0];
    for (int64_t i = wave; i < n; i += nwaves) {
        const double *xi = X + i * d;
        double acc = Qc[i];
        for (int c = 0; c < cnt; ++c) {
            const int64_t j = zchg[2 + c];
            const double *xj = X + j * d;
            double dot = 0.0;
            for (int64_t k = lane; k < d; k += 64) dot = fma(xi[k], xj[k], dot);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
            double kv;
            if (kernel == BQ_KERNEL_RBF) {
                double dist = -2.0 * dot;
                dist += sq[i];
                dist += sq[j];
                dist = fmax(dist, 0.0);
                if (i == j) dist = 0.0;
                kv = bq_exp(-gamma * dist);
            } else if (kernel == BQ_KERNEL_POLY) {
                const double b = gamma * dot + coef0;
                kv = degree == 2 ? b * b : (degree == 3 ? b * b * b : pow(b, (double)degree));
            } else if (kernel == BQ_KERNEL_SIGMOID) {
                kv = tanh(gamma * dot + coef0);
            } else {
                kv = dot;
            }
            if (f32) kv = (double)(float)kv;
            double q = kv + (add_one ? 1.0 : 0.0);
            if (sgn) q *= sgn[i] * sgn[j];
            if (i == j) q += diag_add;
            acc = fma(zdl[c], q, acc);
        }
        if (lane == 0) Qz[i] = acc;
    }
}

__global__ void as_row_norms_kernel(const double *__restrict__ X, int64_t n, int64_t d, double *__restrict__ out, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    double v = 0.0;
    if (i < n) {
        const double *row = X + i * d;
        for (int64_t k = 0; k < d; ++k) v = fma(row[k], row[k], v);
    }
    out[i] = v;
}

// sol[a] = x[idx[a]] + delta[idx[a]] (x: the point the iteration started from): the restricted solution in the compact order
// as_candidate_kernel reads
__global__ void as_cg_gather_kernel(const int *__restrict__ ints, const int *__restrict__ idx,
                                    const double *__restrict__ x, const double *__restrict__ dlt,
                                    double *__restrict__ sol, int64_t N) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < ints[0] && a < N) sol[a] = x[idx[a]] + dlt[idx[a]];
}

// ---------------------------------------------------------------------------------------------------------------
// the diagonal + low-rank preconditioner of the inner iteration (struct as_pc)
// ---------------------------------------------------------------------------------------------------------------
constexpr int PC_MAX_M = 1024;   // features the apply kernel keeps in LDS
constexpr int PC_T = 32, PC_C = 64, PC_SLICES = 8;

// class statistics of the samples (BQ_SVC panels): cls[k] = a_k = (mean_+ - mean_-)_k / 2, cls[d + k] = m0_k = (mean_+ + mean_-)_k / 2,
// cls[2d] = |a|.  One workgroup per feature column for the sums, the last one to finish closes.
__global__ __launch_bounds__(256) void as_pc_class_kernel(int64_t n, int64_t d, const double *__restrict__ X,
                                                          const double *__restrict__ sgn, double *__restrict__ cls,
                                                          unsigned int *ticket) {
    __shared__ double sh[4];
    const int64_t k = blockIdx.x;
    double sp = 0.0, sm = 0.0, cp = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double v = X[i * d + k];
        if (sgn[i] > 0.0) {
            sp += v;
            cp += 1.0;
        } else {
            sm += v;
        }
    }
    sp = as_block_sum(sp, sh);
    sm = as_block_sum(sm, sh);
    cp = as_block_sum(cp, sh);
    if (threadIdx.x == 0) {
        const double cm = (double)n - cp;
        const double mp = cp > 0.0 ? sp / cp : 0.0, mm = cm > 0.0 ? sm / cm : 0.0;
        cls[k] = (cp > 0.0 && cm > 0.0) ? 0.5 * (mp - mm) : 0.0;
        cls[d + k] = (cp > 0.0 && cm > 0.0) ? 0.5 * (mp + mm) : (cp > 0.0 ? mp : mm);
    }
    if (as_last_block(ticket)) {
        double s2 = 0.0;
        for (int64_t j = threadIdx.x; j < d; j += 256) s2 += cls[j] * cls[j];
        s2 = as_block_sum(s2, sh);
        if (threadIdx.x == 0) {
            cls[2 * d] = sqrt(s2);
            *ticket = 0;
        }
    }
}

// Phi (feature-major).  One thread per sample.  RBF:
//   family 1 (d + 1 columns): y e^{-g|x|^2} [1, sqrt(2g) x]                    the order-0/1 terms of e^{2g x.x'}
//   family 2, BQ_SVC panels — the directions of the ORDER-2 term (2g x.x')^2 / 2 = 2g^2 <x x', x' x''> whose eigenvalues grow like
//   n |class mean|^2 (the rest of that term is a flat bulk of d (d + 1) / 2 directions no low-rank model captures:
//   tools/pc_nystrom_study.py, tools/pc_order2_cpu_study.py):
//     fam2 == 2 (2d columns, round 5): the EXACT projection of the order-2 feature sqrt(2) g y e vec(x x') onto the span of
//       U_{c,k} = m_c e_k' + e_k m_c' (c = the two class means, k < d) — raw coordinates f_{c,k} = <x x', U_{c,k}> = 2 (x.m_c) x_k
//       (as_pc_raw_kernel), orthonormalised by the inverse Cholesky factor of the 2d x 2d Gram matrix of the U's
//       (as_pc_project_kernel).  A projection of a positive semi-definite term: P = D + Phi Phi' never over-counts Q.
//     fam2 == 1 (d columns, rounds 3-4; kept for d too large for 3d + 2 features): 2g |a| e (x - m0 - y a), the cross term
//       2 (y y' |a|^2)(e.e') under the assumption m0 ~ 0 — not a projection (it over-counts when m0 is not small): measured against
//       fam2 == 2 at d = 64, n = 20 000: 39 against 23 conjugate-gradient iterations to 1e-8 (profiles/r05/pc_projected_study.txt)
// share[i] <- Q_ii and, for fam2 != 2, the sum of squares of the stored features in s1[i]; as_pc_diag_kernel turns them into 1 / D.
__global__ void as_pc_features_kernel(int kernel, int64_t n, int64_t d, int64_t ld, const double *__restrict__ X,
                                      const double *__restrict__ sgn, const double *__restrict__ cls, int fam2, double gamma,
                                      int add_one, double diag_add, int m, float *__restrict__ Phi, double *__restrict__ qdiag,
                                      double *__restrict__ raw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ld) return;
    if (i >= n) {
        for (int j = 0; j < m; ++j) Phi[(int64_t)j * ld + i] = 0.f;
        qdiag[i] = 0.0;
        if (raw != nullptr) raw[i] = raw[ld + i] = raw[2 * ld + i] = 0.0;
        return;
    }
    const double y = sgn ? sgn[i] : 1.0;
    const double *x = X + i * d;
    double sq = 0.0;
    for (int64_t k = 0; k < d; ++k) sq = fma(x[k], x[k], sq);
    double qii;
    int col = 0;
    auto put = [&](int64_t j, double v) { Phi[j * ld + i] = (float)v; };   // D is what the STORED (fp32) values leave of the diagonal
    if (kernel == BQ_KERNEL_RBF) {
        const double e = exp(-gamma * sq);
        const double c0 = y * e, c1 = c0 * sqrt(2.0 * gamma);
        put(0, c0);
        for (int64_t k = 0; k < d; ++k) put(1 + k, c1 * x[k]);
        col = (int)d + 1;
        if (fam2 == 1) {
            const double c2 = 2.0 * gamma * cls[2 * d] * e;
            for (int64_t k = 0; k < d; ++k) put(col + k, c2 * (x[k] - cls[d + k] - y * cls[k]));
            col += (int)d;
        } else if (fam2 == 2) {   // the columns are written by as_pc_project_kernel from these three per-sample numbers
            double sp = 0.0, sm = 0.0;
            for (int64_t k = 0; k < d; ++k) {
                sp = fma(x[k], cls[d + k] + cls[k], sp);   // x . m_+,  m_+ = m0 + a
                sm = fma(x[k], cls[d + k] - cls[k], sm);   // x . m_-,  m_- = m0 - a
            }
            raw[i] = 2.0 * sp;
            raw[ld + i] = 2.0 * sm;
            raw[2 * ld + i] = sqrt(2.0) * gamma * c0;      // sqrt(2 g^2) y e
            col += 2 * (int)d;
        }
        qii = 1.0;
    } else {   // linear: exact features (up to their fp32 rounding)
        for (int64_t k = 0; k < d; ++k) put(k, y * x[k]);
        col = (int)d;
        qii = sq;
    }
    if (add_one) {
        put(col, y);
        qii += 1.0;
    }
    qdiag[i] = qii + diag_add;
}

// family 2, fam2 == 2: Phi[col0 + j][i] = scale_i * sum_{l <= j} F[i][l] Rinv[l][j],  F[i][c d + k] = (2 x_i.m_c) x_ik — a
// (samples x 2d) x (2d x 2d upper triangular) product, 64 x 64 output tiles, 4 x 4 per thread, fp64 accumulation, once per solver.
__global__ __launch_bounds__(256) void as_pc_project_kernel(int64_t n, int64_t d, int64_t ld, const double *__restrict__ X,
                                                            const double *__restrict__ raw, const double *__restrict__ Rinv,
                                                            int col0, float *__restrict__ Phi) {
    __shared__ double As[16][65], Bs[16][65];
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int j0 = (int)blockIdx.y * 64, n2 = 2 * (int)d;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double acc[4][4] = {};
    const int kend = j0 + 64 < n2 ? j0 + 64 : n2;   // Rinv is upper triangular: rows beyond the tile's last column are zero
    for (int k0 = 0; k0 < kend; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t + q * 256, kk = e & 15, ii = e >> 4;   // consecutive threads: consecutive k of one sample (row-major X)
            const int64_t i = i0 + ii;
            const int l = k0 + kk;
            double v = 0.0;
            if (i < n && l < n2) v = raw[(l >= d ? ld : 0) + i] * X[i * d + (l >= d ? l - d : l)];
            As[kk][ii] = v;
            const int jj = e & 63, kb = e >> 6;
            Bs[kb][jj] = (k0 + kb < n2 && j0 + jj < n2) ? Rinv[(int64_t)(k0 + kb) * n2 + j0 + jj] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = As[kk][ty * 4 + r];
                b[r] = Bs[kk][tx * 4 + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + ty * 4 + r;
        if (i >= n) continue;
        const double sc = raw[2 * ld + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + tx * 4 + c;
            if (j < n2) Phi[(int64_t)(col0 + j) * ld + i] = (float)(sc * acc[r][c]);
        }
    }
}

// 1 / D_i and the share of the diagonal the features leave, D_i = Q_ii - |Phi_i|^2 over the STORED features (floored at 1e-8 Q_ii: P
// only has to be positive definite).  qdiag_share: Q_ii in, share out.
__global__ void as_pc_diag_kernel(int64_t n, int64_t ld, int m, const float *__restrict__ Phi, double *__restrict__ dinv,
                                  double *__restrict__ qdiag_share) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ld) return;
    if (i >= n) {
        dinv[i] = 0.0;
        qdiag_share[i] = 1.0;
        return;
    }
    double s = 0.0;
    for (int j = 0; j < m; ++j) {
        const double v = (double)Phi[(int64_t)j * ld + i];
        s = fma(v, v, s);
    }
    const double qii = qdiag_share[i];
    dinv[i] = 1.0 / fmax(qii - s, 1e-8 * qii);
    qdiag_share[i] = (qii - s) / qii;   // what the features leave of the diagonal: the host refuses a model that leaves too little
}

// Gpart[slice][a][b] = sum over the slice's FREE samples of Phi[a][i] Phi[b][i] / D_i, lower tiles (b-tile <= a-tile)
__global__ __launch_bounds__(256) void as_pc_gram_kernel(int m, int64_t mp, int64_t N, int64_t ld, const float *__restrict__ Phi,
                                                         const double *__restrict__ dinv, const unsigned char *__restrict__ mL,
                                                         const unsigned char *__restrict__ mU, double *__restrict__ Gpart) {
    __shared__ double As[PC_T][PC_C + 1], Bs[PC_T][PC_C + 1];
    // (ta, tb) from the linear lower-triangle tile index
    int ta = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((ta + 1) * (ta + 2) / 2 <= (int)blockIdx.x) ++ta;
    while (ta * (ta + 1) / 2 > (int)blockIdx.x) --ta;
    const int tb = (int)blockIdx.x - ta * (ta + 1) / 2;
    const int a0 = ta * PC_T, b0 = tb * PC_T;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int64_t per = ((N + PC_SLICES - 1) / PC_SLICES + PC_C - 1) / PC_C * PC_C;
    const int64_t i0 = (int64_t)blockIdx.y * per, i1 = i0 + per < N ? i0 + per : N;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c = i0; c < i1; c += PC_C) {
#pragma unroll
        for (int k = 0; k < (PC_T * PC_C) / 256; ++k) {
            const int e = t + k * 256, row = e / PC_C, cc = e % PC_C;
            const int64_t i = c + cc;
            double wgt = 0.0;
            if (i < i1 && !(mL[i] | mU[i])) wgt = dinv[i];
            As[row][cc] = (a0 + row < m && wgt != 0.0) ? (double)Phi[(int64_t)(a0 + row) * ld + i] * wgt : 0.0;
            Bs[row][cc] = (b0 + row < m && i < i1) ? (double)Phi[(int64_t)(b0 + row) * ld + i] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < PC_C; ++k) {
            const double bv = Bs[tx][k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fma(As[ty * 4 + j][k], bv, acc[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        Gpart[((int64_t)blockIdx.y * mp + a0 + ty * 4 + j) * mp + b0 + tx] = acc[j];
}

// full rebuild: H = G = I + the slices of Gpart added in slice order (lower triangle; identity on the pad rows)
__global__ void as_pc_gram_reduce_kernel(int m, int64_t mp, const double *__restrict__ Gpart, double *__restrict__ H, int64_t ldh) {
    const int64_t a = blockIdx.y, b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b > a || b >= mp) return;
    double v = (a == b) ? 1.0 : 0.0;
    if (a < m)
        for (int sidx = 0; sidx < PC_SLICES; ++sidx) v += Gpart[((int64_t)sidx * mp + a) * mp + b];
    H[a * ldh + b] = v;
}

// G^-1 = L^-T L^-1 from the explicit inverse factor bq_chol_prepare_sweeps leaves (MT = L^-T, upper triangular, pitch 1024):
// Ginv[a][b] = sum_k MT[a][k] MT[b][k] — 16 x 16 tiles through LDS, k ascending: one fixed order
__global__ __launch_bounds__(256) void as_pc_ginv_kernel(int64_t mp, const double *__restrict__ MT, int64_t ldm, double *__restrict__ Ginv) {
    __shared__ double As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t a0 = (int64_t)blockIdx.y * 16, b0 = (int64_t)blockIdx.x * 16;
    double acc = 0.0;
    const int64_t kstart = (a0 > b0 ? a0 : b0);   // MT[a][k] = 0 for k < a
    for (int64_t k0 = kstart; k0 < mp; k0 += 16) {
        As[ty][tx] = MT[(a0 + ty) * ldm + k0 + tx];
        Bs[ty][tx] = MT[(b0 + ty) * ldm + k0 + tx];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(As[ty][k], Bs[tx][k], acc);
        __syncthreads();
    }
    Ginv[(a0 + ty) * mp + b0 + tx] = acc;
}

// u = Ginv t: a wave per row, lanes along the columns, xor butterfly (every lane ends with the sum)
__global__ __launch_bounds__(256) void as_pc_gemv_kernel(int64_t mp, const double *__restrict__ Ginv, const double *__restrict__ t,
                                                         double *__restrict__ u, const as_cg_scal *cg) {
    if (cg->done) return;
    const int lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= mp) return;
    const double *row = Ginv + a * mp;
    double acc = 0.0;
    for (int64_t b = lane; b < mp; b += 64) acc = fma(row[b], t[b], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) u[a] = acc;
}

constexpr int PC_MAX_CHG = 64;
// which samples entered / left the free set since G was last brought up to date, in index order (the order of the rank-one
// updates must not depend on the launch geometry: as_list_count / as_list_positions); more than PC_MAX_CHG of them, or `force`:
// full rebuild
__global__ __launch_bounds__(256) void as_pc_diff_count_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                               const unsigned char *__restrict__ mU,
                                                               const unsigned char *__restrict__ prev, int *__restrict__ cnt,
                                                               unsigned int *ticket, int *__restrict__ chg, int force) {
    int bits = 0;
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        if (i < N && (int)!(mL[i] | mU[i]) != (int)prev[i]) bits |= 1 << j;
    }
    __shared__ int total;
    if (threadIdx.x == 0) total = -1;
    as_list_count(bits, cnt, ticket, &total);
    if (threadIdx.x == 0 && total >= 0) {
        chg[0] = total;
        chg[1] = (force || total > PC_MAX_CHG) ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void as_pc_diff_write_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                               const unsigned char *__restrict__ mU, unsigned char *__restrict__ prev,
                                                               const int *__restrict__ cnt, int *__restrict__ chg) {
    int bits = 0, fr[BQ_VEC_ITEMS];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        fr[j] = 0;
        if (i < N) {
            fr[j] = !(mL[i] | mU[i]);
            if (fr[j] != (int)prev[i]) bits |= 1 << j;
            prev[i] = (unsigned char)fr[j];
        }
    }
    int pos[BQ_VEC_ITEMS];
    as_list_positions(bits, cnt, pos);
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j)
        if (((bits >> j) & 1) && pos[j] < PC_MAX_CHG) {
            chg[2 + 2 * pos[j]] = (int)((int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x);
            chg[3 + 2 * pos[j]] = fr[j] ? 1 : -1;
        }
}

// G -> G + s phi phi' / D for every sample that entered (s = +1) or left (s = -1) the free set, in list order, carried on the
// INVERSE (Sherman-Morrison):  v = Ginv phi,  Ginv -= (s / D) / (1 + (s / D) phi'v) v v'.  One workgroup: the updates are a
// chain, each is two passes over the m x m inverse (3 MB at m = 514: L2), and there are one or two of them per outer iteration
// — against the m^3 / 3 factorisation + explicit inverse of round 3's every outer iteration (0.9 ms at BASELINE config 5).
// G - phi phi'/D stays >= I, so every denominator is positive; one at or below 1e-8 (or not finite) raises *fail and the caller
// sums G afresh.  Fixed order throughout: the same bits on every rank.
__global__ __launch_bounds__(1024) void as_pc_sm_kernel(int m, int64_t mp, int64_t ld, const float *__restrict__ Phi,
                                                        const double *__restrict__ dinv, const int *__restrict__ chg,
                                                        double *__restrict__ Ginv, int *__restrict__ fail) {
    __shared__ double phi[PC_MAX_M], v[PC_MAX_M];
    __shared__ double red[16];
    __shared__ double coef_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cnt = chg[0];
    for (int c = 0; c < cnt; ++c) {
        const int64_t i = chg[2 + 2 * c];
        const double wgt = (double)chg[3 + 2 * c] * dinv[i];
        for (int j = tid; j < m; j += 1024) phi[j] = (double)Phi[(int64_t)j * ld + i];
        __syncthreads();
        for (int a0 = wv * 2; a0 < m; a0 += 32) {   // two rows per wave and turn: two independent chains
            const int a1 = a0 + 1 < m ? a0 + 1 : a0;
            const double *r0 = Ginv + (int64_t)a0 * mp, *r1 = Ginv + (int64_t)a1 * mp;
            double s0 = 0.0, s1 = 0.0;
            for (int b = lane; b < m; b += 64) {
                s0 = fma(r0[b], phi[b], s0);
                s1 = fma(r1[b], phi[b], s1);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                s0 += __shfl_xor(s0, off, 64);
                s1 += __shfl_xor(s1, off, 64);
            }
            if (lane == 0) {
                v[a0] = s0;
                v[a1] = s1;
            }
        }
        __syncthreads();
        double part = 0.0;
        for (int j = tid; j < m; j += 1024) part = fma(phi[j], v[j], part);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) red[wv] = part;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int k = 0; k < 16; ++k) tot += red[k];
            const double denom = 1.0 + wgt * tot;
            if (!(denom > 1e-8) || !isfinite(denom)) {
                *fail = 1;
                coef_s = 0.0;
            } else {
                coef_s = wgt / denom;
            }
        }
        __syncthreads();
        const double coef = coef_s;
        for (int a = wv; a < m; a += 16) {
            double *row = Ginv + (int64_t)a * mp;
            const double ca = coef * v[a];
            for (int b = lane; b < m; b += 64) row[b] = fma(-ca, v[b], row[b]);
        }
        __syncthreads();
    }
}

typedef float as_f4 __attribute__((ext_vector_type(4)));
constexpr int PC_FG = 8;   // features per butterfly group of the t kernel

// t[j] = sum_i Phi[j][i] r_i / D_i   (r vanishes outside the free set).  One workgroup per block of 1024 samples: a lane keeps
// w = r / D of its four consecutive samples in registers and walks all features (one 16-byte load per feature: a 4 KiB run per
// workgroup and feature row), eight features at a time through a halving butterfly over the 64 lanes (3 + 3 shuffle-adds for
// eight sums), the four wave sums meet in LDS -> tpart[block][j]; the last workgroup to finish adds the blocks in block order.
// Round 3 had one workgroup per FEATURE re-reading r and 1 / D for each of them: 2.5 GB of L2 traffic beside the 1 GB of
// features, 0.40 ms per call at BASELINE config 5.  Fixed order throughout: the same bits on every rank.
constexpr int PC_TSLICES_MAX = 4;   // feature slices of the t kernel (gridDim.y)
__global__ __launch_bounds__(256) void as_pc_tphi_kernel(int m, int64_t m8, int64_t mp, int64_t N, int64_t ld,
                                                         const float *__restrict__ Phi, const double *__restrict__ dinv,
                                                         const double *__restrict__ r, double *__restrict__ tpart,
                                                         double *__restrict__ tvec, unsigned int *ticket, const as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double wsum[4][PC_MAX_M];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * BQ_VEC_TILE + 4 * tid;   // ld is a multiple of the tile: always in range
    // this workgroup's feature groups: a contiguous quarter of the m8 / 8 groups
    const int64_t ngroups = m8 / PC_FG;
    const int64_t g_lo = ngroups * blockIdx.y / gridDim.y, g_hi = ngroups * (blockIdx.y + 1) / gridDim.y;
    const int64_t j_lo = g_lo * PC_FG, j_hi = g_hi * PC_FG;
    double w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = (base + k < N) ? r[base + k] * dinv[base + k] : 0.0;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    const int rho = (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0);
    for (int64_t j0 = j_lo; j0 < j_hi; j0 += PC_FG) {
        double a[PC_FG];
#pragma unroll
        for (int f = 0; f < PC_FG; ++f) {
            const as_f4 v = *reinterpret_cast<const as_f4 *>(Phi + (j0 + f) * ld + base);
            a[f] = fma((double)v.w, w4[3], fma((double)v.z, w4[2], fma((double)v.y, w4[1], (double)v.x * w4[0])));
        }
        double u[4], t2[2], s1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double send = b5 ? a[i] : a[i + 4];
            const double keep = b5 ? a[i + 4] : a[i];
            u[i] = keep + __shfl_xor(send, 32, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double send = b4 ? u[i] : u[i + 2];
            const double keep = b4 ? u[i + 2] : u[i];
            t2[i] = keep + __shfl_xor(send, 16, 64);
        }
        {
            const double send = b3 ? t2[0] : t2[1];
            const double keep = b3 ? t2[1] : t2[0];
            s1 = keep + __shfl_xor(send, 8, 64);
        }
        s1 += __shfl_xor(s1, 4, 64);
        s1 += __shfl_xor(s1, 2, 64);
        s1 += __shfl_xor(s1, 1, 64);
        if ((lane & 7) == 0) wsum[wv][j0 - j_lo + rho] = s1;
    }
    __syncthreads();
    double *mine = tpart + (int64_t)blockIdx.x * mp;
    for (int64_t j = j_lo + tid; j < j_hi; j += 256) {
        const int64_t c = j - j_lo;
        mine[j] = j < m ? ((wsum[0][c] + wsum[1][c]) + wsum[2][c]) + wsum[3][c] : 0.0;
    }
    (void)tvec;
    (void)ticket;
}

// t[j] = the sample blocks' partial sums added in block order: 16 features x 16 interleaved runs of blocks per workgroup (a run's
// loads are independent of each other: ~15 in flight per lane), the 16 runs then combined in run order.  (As the tail of the
// kernel above, one workgroup walking 245 dependent, 5 KB-strided loads per feature, it cost more than the pass over Phi.)
__global__ __launch_bounds__(256) void as_pc_treduce_kernel(int64_t m8, int64_t mp, int64_t nb, const double *__restrict__ tpart,
                                                            double *__restrict__ tvec, const as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + tx;
    double acc = 0.0;
    if (j < m8)
        for (int64_t b = ty; b < nb; b += 16) acc += tpart[b * mp + j];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && j < mp) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][tx];
        tvec[j] = v;
    }
}

// z = P^-1 r on the free set: z_i = (r_i - Phi_i . u) / D_i, u = G^-1 t;  rz = r'z;  beta = rz / rz_old (first: 0).
// A lane owns four consecutive samples (one 16-byte load per feature row), eight feature rows in flight.
__global__ __launch_bounds__(256) void as_pc_apply_kernel(int m, int64_t m8, int64_t N, int64_t ld, const float *__restrict__ Phi,
                                                          const double *__restrict__ dinv, const unsigned char *__restrict__ mL,
                                                          const unsigned char *__restrict__ mU, const double *__restrict__ r,
                                                          const double *__restrict__ u, double *__restrict__ z, double *part,
                                                          int64_t nblk, as_cg_scal *cg, int first) {
    if (cg->done) return;
    __shared__ double us[PC_MAX_M];
    __shared__ double sh[4];
    for (int j = threadIdx.x; j < m8; j += BQ_VEC_BLOCK) us[j] = j < m ? u[j] : 0.0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * BQ_VEC_TILE + 4 * threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t j0 = 0; j0 < m8; j0 += PC_FG) {
        as_f4 v[PC_FG];
#pragma unroll
        for (int f = 0; f < PC_FG; ++f) v[f] = *reinterpret_cast<const as_f4 *>(Phi + (j0 + f) * ld + base);
#pragma unroll
        for (int f = 0; f < PC_FG; ++f) {
            const double uj = us[j0 + f];
            acc[0] = fma((double)v[f].x, uj, acc[0]);
            acc[1] = fma((double)v[f].y, uj, acc[1]);
            acc[2] = fma((double)v[f].z, uj, acc[2]);
            acc[3] = fma((double)v[f].w, uj, acc[3]);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k;
        double zi = 0.0;
        if (i < N && !(mL[i] | mU[i])) {
            zi = dinv[i] * (r[i] - acc[k]);
            s += __dmul_rn(r[i], zi);
        }
        z[i] = zi;
    }
    s = as_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (as_last_block(&cg->ticket[0])) {
        const double rz = as_final_sum(part, nblk, sh);
        if (threadIdx.x == 0) {
            cg->ticket[0] = 0;
            cg->beta = (first || !(cg->rz > 0.0)) ? 0.0 : rz / cg->rz;
            cg->rz = rz;
            if (!(rz > 0.0) || !isfinite(rz)) cg->info = 2;   // P is positive definite: r'z <= 0 means r = 0 or a broken factor
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// factor re-use: Schur-complement updates of a base factorisation (see the header comment)
// ---------------------------------------------------------------------------------------------------------------
// changed indices carried before the base is re-factorised: the larger the factor, the longer it is worth keeping
// (n^3/3 to rebuild against one more small-system row per carried index)
static int as_schur_limit(int64_t np0) {
    if (const char *e = getenv("BQ_AS_SCHUR_LIMIT")) return std::max(1, std::min(atoi(e), AS_SCHUR_MAX));   // tests
    // (re-tuned in round 2 for the two-launches-per-1024-rows sweeps: a solve is ~4x cheaper, so the n^3/3 of a rebuild is
    // amortised over more iterations: n = 50 000: 0.77 s per rebuild = 3 ms per iteration at 256 carried changes)
    // (below |A| = 8 192: 96 through round 4's first half; swept again once every free set went through the kept factor and the
    // looks became cheap — 48 / 96 / 160 / 256: 11.15 / 10.42 / 10.13 / 10.22 s to 'optimal' at n = 20 000, profiles/r04/as_f_chain.txt)
    return np0 < 8192 ? 160 : (np0 < 40000 ? 384 : (np0 < 80000 ? 768 : 1536));
}

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

__global__ void as_schur_pos_kernel(int64_t N, int64_t n0, const int *__restrict__ idx0, int *__restrict__ pos0, int pass) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pass == 0) {
        if (t < N) pos0[t] = -1;
    } else if (t < n0) {
        pos0[idx0[t]] = (int)t;
    }
}

// z = the bound value on every bound variable OUTSIDE the base (those inside are pinned by a multiplier row), else 0
__global__ void as_schur_z_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                  const double *__restrict__ lb, const double *__restrict__ ub,
                                  const int *__restrict__ pos0, double *__restrict__ z) {
    VEC_LOOP(i) {
        if (i < N) z[i] = (pos0[i] < 0 && (mL[i] | mU[i])) ? (mU[i] ? ub[i] : lb[i]) : 0.0;
    }
}

__global__ void as_schur_rhs0_kernel(int64_t n0, int64_t np0, const int *__restrict__ idx0, const double *__restrict__ q,
                                     const double *__restrict__ Qz, double *__restrict__ rhs) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < np0) rhs[a] = a < n0 ? -(q[idx0[a]] + Qz[idx0[a]]) : 0.0;
}

// the column of a new slot: e_pos for a base variable that reached a bound, Q[A0, var] for a freed variable
template <typename T>
__global__ void as_schur_col_kernel(int kind, int var, int pos, int64_t n0, int64_t np0, const int *__restrict__ idx0,
                                    int structure, const T *__restrict__ panel, int64_t ldp, int packed, int64_t n,
                                    const double *__restrict__ sgn, double diag_add, double *__restrict__ out,
                                    double *__restrict__ out2) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= np0) return;
    double v = 0.0;
    if (kind == 0)
        v = a == pos ? 1.0 : 0.0;
    else if (a < n0)
        v = bq_q_elem(structure, panel, ldp, packed, n, sgn, diag_add, (int64_t)idx0[a], (int64_t)var);
    out[a] = v;
    out2[a] = v;   // the right-hand side of the solve that follows (was a device-to-device copy)
}

// block i: out[i] = extra_i - U[i]'v.  mode 0 (v = W[k]): extra = V_ik = Q[var_i, var_k] when both were freed, else 0.
// mode 1 (v = y0): extra = the right-hand side of row i: the bound of a pinned variable, -(q + Qz) of a freed one.
// mode 2: BOTH in one pass over U[i] (the usual iteration: one new slot k = m - 1, then the right-hand side): out[i] as mode 0
// with v = v0, out1[i] as mode 1 with v = v1 — each dot product summed exactly as in its own launch (same bits).
template <typename T>
__global__ __launch_bounds__(256) void as_schur_dots_kernel(int mode, int k, const double *__restrict__ U,
                                                            const double *__restrict__ v, const double *__restrict__ v1,
                                                            int64_t cap, int64_t np0,
                                                            const int *__restrict__ meta, int structure,
                                                            const T *__restrict__ panel, int64_t ldp, int packed, int64_t n,
                                                            const double *__restrict__ sgn, double diag_add,
                                                            const unsigned char *__restrict__ mU,
                                                            const double *__restrict__ lb, const double *__restrict__ ub,
                                                            const double *__restrict__ q, const double *__restrict__ Qz,
                                                            double *__restrict__ out, double *__restrict__ out1,
                                                            unsigned int *ticket, int *mail, int seq) {
    __shared__ double sh[4], sh1[4];
    const int i = blockIdx.x;
    const double *u = U + (int64_t)i * cap;
    // the slot table may live in the HOST's (mapped) buffer: thread 0 asks for its four entries before the dot products, not after
    int ki = 0, vi = 0, kk = 0, vk = 0;
    if (threadIdx.x == 0) {
        ki = meta[i];
        vi = meta[AS_SCHUR_MAX + i];
        kk = meta[k];
        vk = meta[AS_SCHUR_MAX + k];
    }
    double s = 0.0, t = 0.0;
    if (mode == 2) {
        for (int64_t a = threadIdx.x; a < np0; a += 256) {
            const double ua = u[a];
            s += __dmul_rn(ua, v[a]);
            t += __dmul_rn(ua, v1[a]);
        }
    } else {
        for (int64_t a = threadIdx.x; a < np0; a += 256) s += __dmul_rn(u[a], v[a]);
    }
    s = as_wsum_any(s);
    if (mode == 2) t = as_wsum_any(t);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = s;
        sh1[threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double dot = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        if (mode != 1) {
            double extra = 0.0;
            if (ki == 1 && kk == 1) extra = bq_q_elem(structure, panel, ldp, packed, n, sgn, diag_add, (int64_t)vi, (int64_t)vk);
            out[i] = extra - dot;
        }
        if (mode != 0) {
            const double d1 = mode == 1 ? dot : ((sh1[0] + sh1[1]) + sh1[2]) + sh1[3];
            const double extra = ki == 0 ? (mU[vi] ? ub[vi] : lb[vi]) : -(q[vi] + Qz[vi]);
            (mode == 1 ? out : out1)[i] = extra - d1;
        }
        if (mail != nullptr) {   // out / out1 are the host's (mapped) buffers: the last workgroup posts the record
            __threadfence_system();
            if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
                *ticket = 0;
                __threadfence_system();
                as_post(mail, seq);
            }
        }
    }
}

// y = y0 - sum_k coef[k] W[k]: 256 rows x 4 interleaved runs of slots per workgroup (1024 threads: four times the loads in
// flight of the one-thread-per-row form, which ran this m x np0 product at 0.65 TB/s), the four runs added in run order
__global__ __launch_bounds__(1024) void as_schur_combine_kernel(int64_t np0, int m, const double *__restrict__ y0,
                                                                const double *__restrict__ W, int64_t cap,
                                                                const double *__restrict__ coef, double *__restrict__ y) {
    __shared__ double part[4][256];
    __shared__ double cf[AS_SCHUR_MAX];   // the coefficients may live in the HOST's (mapped) buffer: one coalesced read per workgroup,
    for (int k = threadIdx.x; k < m; k += 1024) cf[k] = coef[k];   // not one uncached trip over PCIe per term of the sum
    __syncthreads();
    const int r = threadIdx.x & 255, g = threadIdx.x >> 8;
    const int64_t a = (int64_t)blockIdx.x * 256 + r;
    double v = 0.0;
    if (a < np0)
        for (int k = g; k < m; k += 4) v += __dmul_rn(cf[k], W[(int64_t)k * cap + a]);
    part[g][r] = v;
    __syncthreads();
    if (g == 0 && a < np0) y[a] = y0[a] - (((part[0][r] + part[1][r]) + part[2][r]) + part[3][r]);
}

// C = L D L' of the m x m Schur complement, kept on the host and grown by one row per new slot (O(m^2)); C is symmetric
// quasi-definite — minus a positive definite block for the pinned variables, a positive definite one for the freed —
// so it factorises without pivoting in any order.  as_ldl_solve checks the residual; on failure the caller falls back
// to the pivoted elimination below (and from there to a fresh base factor).
// The host's share of a kept-factor iteration is this O(m^2) work between two waits on the stream (the device idles meanwhile:
// profiles/r04/as_n20k_stream_idle_*.txt), so it is compiled a second time for AVX2 + FMA hosts and picked at load time
// (function multi-versioning; the device pass of hipcc does not know the attribute).  The sums may be re-associated by the
// vectoriser: the small system's solution moves in its last bits with the host's vector width, as it would with another BLAS.
#if defined(__HIP_DEVICE_COMPILE__)
#define BQ_HOST_SIMD
#else
#define BQ_HOST_SIMD __attribute__((target_clones("arch=x86-64-v3", "default")))
#endif

// sum_j a[j] * b[j]
BQ_HOST_SIMD static double as_dot4(const double *__restrict__ a, const double *__restrict__ b, int n) {
#pragma clang fp reassociate(on)
    double s = 0.0;
#pragma clang loop vectorize(enable) interleave_count(4)
    for (int j = 0; j < n; ++j) s += a[j] * b[j];
    return s;
}
// sum_j |a[j] * b[j]|
BQ_HOST_SIMD static double as_absdot(const double *__restrict__ a, const double *__restrict__ b, int n) {
#pragma clang fp reassociate(on)
    double s = 0.0;
#pragma clang loop vectorize(enable) interleave_count(4)
    for (int j = 0; j < n; ++j) s += std::fabs(a[j] * b[j]);
    return s;
}
// y[0:n) -= l[0:n) * w
BQ_HOST_SIMD static void as_axpy_neg(double *__restrict__ y, const double *__restrict__ l, double w, int n) {
#pragma clang loop vectorize(enable) interleave_count(4)
    for (int j = 0; j < n; ++j) y[j] -= l[j] * w;
}

// C = L D L' of the m x m Schur complement, kept on the host and grown by one row per new slot (O(m^2)); C is symmetric
// quasi-definite — minus a positive definite block for the pinned variables, a positive definite one for the freed —
// so it factorises without pivoting in any order.  as_ldl_solve checks the residual; on failure the caller falls back
// to the pivoted elimination below (and from there to a fresh base factor).
static bool as_ldl_extend(as_schur *c, int m) {
    const size_t ld = AS_SCHUR_MAX;
    std::vector<double> z;
    c->rows_extended += m - c->ldl_n;
    c->rows_solved += m;
    for (int k = c->ldl_n; k < m; ++k) {
        double *lk = &c->Lc[(size_t)k * ld];
        // L z = C[0:k, k]  (forward), l = z / D, d = C[k][k] - sum l z
        z.assign((size_t)k + 1, 0.0);
        const double *ck = &c->C[(size_t)k * ld];   // row k = column k (symmetric): contiguous
        for (int i = 0; i < k; ++i) {
            const double v = ck[i] - as_dot4(&c->Lc[(size_t)i * ld], z.data(), i);
            z[i] = v;
            lk[i] = v / c->Dc[i];
        }
        const double d = c->C[(size_t)k * ld + k] - as_dot4(lk, z.data(), k);
        if (!std::isfinite(d) || d == 0.0) return false;
        c->Dc[k] = d;
        lk[k] = 1.0;
        c->ldl_n = k + 1;
    }
    return true;
}

static inline double ldl_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool as_ldl_solve(as_schur *c, int m, const double *t, double *w) {
    const double t0 = c->timing ? ldl_now() : 0.0;
    if (!as_ldl_extend(c, m)) return false;
    const double t1 = c->timing ? ldl_now() : 0.0;
    const size_t ld = AS_SCHUR_MAX;
    std::vector<double> y(t, t + m);
    for (int i = 0; i < m; ++i) y[i] -= as_dot4(&c->Lc[(size_t)i * ld], y.data(), i);   // L y = t, rows of L contiguous
    for (int i = 0; i < m; ++i) y[i] /= c->Dc[i];
    const double t2 = c->timing ? ldl_now() : 0.0;
    // L' w = y by columns of L' = rows of L: once w[i] is final it is eliminated from the unknowns above it (contiguous row i;
    // the dot-product form walked a COLUMN of the 1536-pitch factor per unknown: one cache line per element)
    for (int i = m - 1; i >= 0; --i) {
        const double wi = y[i];
        w[i] = wi;
        as_axpy_neg(y.data(), &c->Lc[(size_t)i * ld], wi, i);
    }
    const double t3 = c->timing ? ldl_now() : 0.0;
    c->t_ldl[0] += t1 - t0;
    c->t_ldl[1] += t2 - t1;
    c->t_ldl[2] += t3 - t2;
    struct tail {
        as_schur *c;
        double t;
        ~tail() {
            if (c->timing) c->t_ldl[3] += ldl_now() - t;
        }
    } tl{c, t3};
    // residual against the stored C
    double worst = 0.0, scale = 0.0;
    for (int i = 0; i < m; ++i) {
        const double *ci = &c->C[(size_t)i * ld];
        const double r = t[i] - as_dot4(ci, w, m);
        if (!std::isfinite(r)) return false;
        worst = std::max(worst, std::fabs(r));
        scale = std::max(scale, std::fabs(t[i]) + as_absdot(ci, w, m));
    }
    return worst <= 1e-11 * scale;
}

// dense m x m solve on the host (partial pivoting); false when a pivot is negligible or the result is not finite
static bool as_small_solve(int m, const std::vector<double> &C, const double *t, double *w) {
    std::vector<double> A((size_t)m * m);
    std::vector<double> b(t, t + m);
    double scale = 0.0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            A[(size_t)i * m + j] = C[(size_t)i * AS_SCHUR_MAX + j];
            scale = std::max(scale, std::fabs(A[(size_t)i * m + j]));
        }
    for (int c = 0; c < m; ++c) {
        int piv = c;
        for (int r = c + 1; r < m; ++r)
            if (std::fabs(A[(size_t)r * m + c]) > std::fabs(A[(size_t)piv * m + c])) piv = r;
        if (!(std::fabs(A[(size_t)piv * m + c]) > 1e-13 * scale)) return false;
        if (piv != c) {
            for (int j = 0; j < m; ++j) std::swap(A[(size_t)piv * m + j], A[(size_t)c * m + j]);
            std::swap(b[piv], b[c]);
        }
        for (int r = c + 1; r < m; ++r) {
            const double f = A[(size_t)r * m + c] / A[(size_t)c * m + c];
            if (f == 0.0) continue;
            for (int j = c; j < m; ++j) A[(size_t)r * m + j] -= f * A[(size_t)c * m + j];
            b[r] -= f * b[c];
        }
    }
    for (int r = m - 1; r >= 0; --r) {
        double v = b[r];
        for (int j = r + 1; j < m; ++j) v -= A[(size_t)r * m + j] * w[j];
        w[r] = v / A[(size_t)r * m + r];
        if (!std::isfinite(w[r])) return false;
    }
    return true;
}

static as_ws *get_ws(bq_solver *s) { return reinterpret_cast<as_ws *>(s->as_ws); }

// launches of the multi-block per-iteration steps; their per-block partials live in three disjoint slices of s->partials
static_assert(BQ_MAX_PARTIAL_Q >= 3, "as_launch_*: three slices of the partials buffer");
static void as_launch_compact(bq_solver *s, as_ws *w, hipStream_t st) {
    int *cnt = reinterpret_cast<int *>(s->partials + s->nblk);
    as_count_free_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, cnt, w->ints, &s->sc->pad1[0]);
    as_write_free_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, cnt, w->idx);
}
static_assert(BQ_MAX_PARTIAL_Q >= 4, "as_launch_step: a fourth slice of the partials buffer");
static void as_launch_step(bq_solver *s, as_ws *w, hipStream_t st, int chain = 0) {
    as_step_min_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, w->cand, s->lb, s->ub, s->x, w->gref ? w->gref : s->g,
                                                              s->partials, s->partials + 3 * s->nblk, s->sc, chain);
    as_step_apply_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, w->cand, s->x, s->sc);
}
static void as_launch_absorb(bq_solver *s, as_ws *w, hipStream_t st) {
    int *cnt = reinterpret_cast<int *>(s->partials + 2 * s->nblk);
    as_absorb_mb_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, s->x, s->lb, s->ub, s->sc, w->ints, cnt, s->stats);
}

static int eval_f(bq_solver *s, double *g_out) {
    // Qd = Q x ; f = 1/2 x'Qx + q'x -> sc->f ; optionally g = Qx + q
    BQ_TRY(bq_problem_apply(s->p, s->x, s->Qd, nullptr));
    return bq_vec_eval_f(s->p, s->x, s->Qd, g_out, &s->sc->f);
}

// the dense iteration's two branches once the candidate is known (w->host_ints[2]: it is feasible).  `exact`: the candidate came
// from a factorisation (kept or fresh), not from the minimum-residual branch — the condition of the product-free f of a ratio step
static int as_finish_iteration(bq_solver *s, as_ws *w, hipStream_t st, bool exact) {
    const int64_t N = s->N;
    if (w->host_ints[2]) {
        as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, w->cand, s->x);
        BQ_TRY(eval_f(s, s->g));
        as_release_mb_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->g, s->mL, s->mU, s->sc, w->ints, s->stats);
        w->chain_ok = exact && w->f_chain;   // x solves its restricted system and g is fresh: a run may start
        w->gref = s->g;
        w->chain_len = 0;
    } else {
        const bool run = w->chain_ok && exact;          // the run's identity holds for this step
        const bool chain = run && w->chain_len < 64;    // ... and f is taken from it (every 64th step of a run: from a product)
        as_launch_step(s, w, st, chain ? 1 : (run ? 2 : 0));
        if (chain) {
            w->chain_len += 1;
            w->chain_steps += 1;
        } else {
            BQ_TRY(eval_f(s, nullptr));
            w->chain_len = 0;          // f is anchored again; the run itself goes on as long as the candidates stay exact
            if (!exact) w->chain_ok = false;
        }
        as_launch_absorb(s, w, st);
    }
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

// ---- factor re-use: host side --------------------------------------------------------------------------------
// Smallest free set that goes through the kept factor.  Through round 4's first half this was 1024: below it every iteration
// factorised Q[A,A] afresh — a small factorisation, but with it the bound product Q z (a whole panel product) and a blocking look
// per iteration: 3 921 of the 22 897 iterations of BASELINE config 2's shape, 0.67 ms each.  Measured to 'optimal' at n = 20 000
// (profiles/r04/as_schur_min_sweep.txt): 1024 17.5 s, 256 16.5 s, 64 16.3 s, 16 16.1 s, 0 16.1 s — any non-empty free set now.
static int as_schur_min() {   // read per iteration: tests switch it between solves
    const char *e = getenv("BQ_AS_SCHUR_MIN");
    return e ? atoi(e) : 1;
}
static bool as_schur_enabled() {
    const char *e = getenv("BQ_AS_SCHUR");
    return e ? atoi(e) != 0 : true;
}

#define AS_PANEL_ARGS(T) p->structure, (const T *)p->panel, p->ld, p->symmetric ? 1 : 0, p->n, p->sgn, p->diag_add

static void as_schur_free(as_ws *w) {
    as_schur *c = w->sch;
    if (!c) return;
    for (void *ptr : {(void *)c->idx0, (void *)c->pos0, (void *)c->U, (void *)c->W, (void *)c->y0, (void *)c->y,
                      (void *)c->small, (void *)c->meta})
        if (ptr) hipFree(ptr);
    for (void *ptr : {(void *)c->meta_pin, (void *)c->small_pin, (void *)c->coef_pin})
        if (ptr) hipHostFree(ptr);
    delete c;
    w->sch = nullptr;
}

// mapped, coherent pinned memory: the device stores into / loads from it in place (as_ws::mail)
constexpr unsigned int AS_MAPPED = hipHostMallocMapped | hipHostMallocCoherent;
template <typename P>
static P *as_dev(P *host) {
    void *d = nullptr;
    return (host != nullptr && hipHostGetDevicePointer(&d, host, 0) == hipSuccess) ? static_cast<P *>(d) : host;
}
// one look of the host at the device: the record `which` (as_ws::mail) has been posted / the stream has drained behind the copies
static int as_look(bq_ctx *ctx, as_ws *w, int which) {
    if (!w->mailbox) return bq_ctx_sync(ctx);
    return bq_ctx_wait_flag(ctx, w->mail + which, w->mail_seq[which]);
}

static inline void as_tick(as_ws *w, int slot) {   // BQ_AS_TIMING: the time since the previous tick goes to `slot`
    if (!w->timing) return;
    const double now = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (slot >= 0) w->tm[slot] += now - w->tm_last;
    w->tm_last = now;
}

static int as_schur_setup(bq_solver *s, as_ws *w) {
    if (w->sch) return BQ_OK;
    as_schur *c = new as_schur();
    w->sch = c;
    c->timing = w->timing;
    c->cap = s->chol->cap;
    BQ_HIP(hipMalloc(&c->idx0, sizeof(int) * (s->N + 1)));
    BQ_HIP(hipMalloc(&c->pos0, sizeof(int) * (s->N + 1)));
    BQ_HIP(hipMalloc(&c->U, sizeof(double) * AS_SCHUR_MAX * c->cap));
    BQ_HIP(hipMalloc(&c->W, sizeof(double) * AS_SCHUR_MAX * c->cap));
    BQ_HIP(hipMalloc(&c->y0, sizeof(double) * c->cap));
    BQ_HIP(hipMalloc(&c->y, sizeof(double) * c->cap));
    BQ_HIP(hipMalloc(&c->small, sizeof(double) * 3 * AS_SCHUR_MAX));   // dots of a new column | coefficients | dots with y0
    BQ_HIP(hipMalloc(&c->meta, sizeof(int) * 2 * AS_SCHUR_MAX));
    BQ_HIP(hipHostMalloc(&c->meta_pin, sizeof(int) * 2 * AS_SCHUR_MAX, AS_MAPPED));
    BQ_HIP(hipHostMalloc(&c->small_pin, sizeof(double) * 2 * AS_SCHUR_MAX, AS_MAPPED));
    BQ_HIP(hipHostMalloc(&c->coef_pin, sizeof(double) * AS_SCHUR_MAX, AS_MAPPED));
    memset(c->meta_pin, 0, sizeof(int) * 2 * AS_SCHUR_MAX);
    c->meta_pin_d = as_dev(c->meta_pin);
    c->small_pin_d = as_dev(c->small_pin);
    c->coef_pin_d = as_dev(c->coef_pin);
    c->hpos0.assign((size_t)s->N, -1);
    c->C.assign((size_t)AS_SCHUR_MAX * AS_SCHUR_MAX, 0.0);
    c->Lc.assign((size_t)AS_SCHUR_MAX * AS_SCHUR_MAX, 0.0);
    c->Dc.assign((size_t)AS_SCHUR_MAX, 0.0);
    return BQ_OK;
}

// base := the current free set (w->idx holds it, compacted at the top of this iteration); *ok = false when its
// factorisation meets a non-positive pivot (the classic path then takes the reference's minres branch)
static int as_schur_refresh(bq_solver *s, as_ws *w, int64_t nA, bool *ok) {
    as_schur *c = w->sch;
    bq_chol_ws *ws = s->chol;
    hipStream_t st = s->p->ctx->stream;
    const int64_t N = s->N;
    c->valid = false;
    std::vector<int> hidx((size_t)nA);
    BQ_HIP(hipMemcpyAsync(c->idx0, w->idx, sizeof(int) * nA, hipMemcpyDeviceToDevice, st));
    BQ_HIP(hipMemcpyAsync(hidx.data(), w->idx, sizeof(int) * nA, hipMemcpyDeviceToHost, st));
    as_schur_pos_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(N, nA, c->idx0, c->pos0, 0);
    as_schur_pos_kernel<<<(unsigned)((nA + 255) / 256), 256, 0, st>>>(N, nA, c->idx0, c->pos0, 1);
    int64_t np0 = 0;
    BQ_TRY(bq_chol_build_h(ws, s->p, c->idx0, nA, nullptr, &np0));
    BQ_TRY(bq_chol_factor(ws, np0));
    BQ_HIP(hipMemcpyAsync(w->host_info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    const int info = w->host_info[0];
    if (info != 0) {
        *ok = false;
        return BQ_OK;
    }
    // this factor is kept for up to hundreds of iterations: make its sweeps short chains of full-chip products
    static const bool fast_sweeps = [] {
        const char *e = getenv("BQ_AS_FAST_SWEEPS");
        return e == nullptr || atoi(e) != 0;
    }();
    if (fast_sweeps) BQ_TRY(bq_chol_prepare_sweeps(ws, np0));
    std::fill(c->hpos0.begin(), c->hpos0.end(), -1);
    for (int64_t a = 0; a < nA; ++a) c->hpos0[(size_t)hidx[(size_t)a]] = (int)a;
    c->n0 = nA;
    c->np0 = np0;
    c->kind.clear();
    c->var.clear();
    c->ldl_n = 0;
    c->valid = true;
    c->y0_valid = false;
    c->refreshes += 1;
    *ok = true;
    return BQ_OK;
}

// drop slot j (swap with the last one): metadata, its row / column of C, its two device columns
static int as_schur_drop(bq_solver *s, as_schur *c, int j) {
    const int last = (int)c->kind.size() - 1;
    if (j != last) {
        hipStream_t st = s->p->ctx->stream;
        c->kind[j] = c->kind[last];
        c->var[j] = c->var[last];
        for (int i = 0; i <= last; ++i) c->C[(size_t)j * AS_SCHUR_MAX + i] = c->C[(size_t)last * AS_SCHUR_MAX + i];
        for (int i = 0; i <= last; ++i) c->C[(size_t)i * AS_SCHUR_MAX + j] = c->C[(size_t)i * AS_SCHUR_MAX + last];
        c->C[(size_t)j * AS_SCHUR_MAX + j] = c->C[(size_t)last * AS_SCHUR_MAX + last];
        BQ_HIP(hipMemcpyAsync(c->U + (int64_t)j * c->cap, c->U + (int64_t)last * c->cap, sizeof(double) * c->np0,
                              hipMemcpyDeviceToDevice, st));
        BQ_HIP(hipMemcpyAsync(c->W + (int64_t)j * c->cap, c->W + (int64_t)last * c->cap, sizeof(double) * c->np0,
                              hipMemcpyDeviceToDevice, st));
    }
    c->kind.pop_back();
    c->var.pop_back();
    c->drops += 1;
    c->ldl_n = std::min(c->ldl_n, j);   // rows < j of the small factorisation only know C[0:j, 0:j], which the swap left alone
    return BQ_OK;
}

// what the previous iteration did to the free set -> slots.  *computed = slots whose columns exist (the new ones are
// appended behind them); *ok = false when the change cannot be carried (too many indices at once)
static int as_schur_event(bq_solver *s, as_ws *w, int *computed, bool *ok) {
    as_schur *c = w->sch;
    const int64_t N = s->N;
    std::vector<int> freed, bound;
    if (w->last_branch == 1) {
        const int hl = w->host_ints[3], hu = w->host_ints[4];
        if (hl < N)
            freed.push_back(hl);
        else if (hu < N)
            freed.push_back(hu);
    } else if (w->last_branch == 0) {
        const int cnt = w->host_ints[8];
        if (cnt > 16) {
            *ok = false;
            return BQ_OK;
        }
        for (int k = 0; k < cnt; ++k) bound.push_back(w->host_ints[9 + k]);
    }
    auto find = [&](int kind, int v) {
        for (size_t j = 0; j < c->kind.size(); ++j)
            if (c->kind[j] == kind && c->var[j] == v) return (int)j;
        return -1;
    };
    // removals of slots first (everything still in the list has its columns), then the new slots at the end
    std::vector<std::pair<int, int>> add;
    for (int v : freed) {
        const int j = find(0, v);
        if (j >= 0) {
            BQ_TRY(as_schur_drop(s, c, j));
        } else {
            add.push_back({1, v});
            c->y0_valid = false;   // a variable outside the base left its bound: z, and with it b0, moves
        }
    }
    for (int v : bound) {
        const int j = find(1, v);
        if (j >= 0) {
            BQ_TRY(as_schur_drop(s, c, j));
            c->y0_valid = false;
        } else {
            add.push_back({0, v});
        }
    }
    *computed = (int)c->kind.size();
    for (auto &kv : add) {
        if (kv.first == 0 && c->hpos0[(size_t)kv.second] < 0) {   // cannot happen: a variable that reached a bound was free
            *ok = false;
            return BQ_OK;
        }
        c->kind.push_back(kv.first);
        c->var.push_back(kv.second);
    }
    *ok = (int)c->kind.size() <= as_schur_limit(c->np0);
    return BQ_OK;
}

template <typename T>
static int as_schur_solve_t(bq_solver *s, as_ws *w, int computed, bool *good) {
    as_schur *c = w->sch;
    bq_chol_ws *ws = s->chol;
    bq_problem *p = s->p;
    hipStream_t st = p->ctx->stream;
    const int64_t N = s->N, np0 = c->np0, n0 = c->n0;
    const int m = (int)c->kind.size();
    const unsigned gb = (unsigned)((np0 + 255) / 256);
    *good = false;
    int *hmeta = c->meta_pin;   // the previous iteration's copy is long complete (every pass through here ends in a synchronisation)
    for (int k = 0; k < m; ++k) {
        hmeta[k] = c->kind[k];
        hmeta[AS_SCHUR_MAX + k] = c->var[k];
    }
    const bool mbx = w->mailbox;
    const int *meta = mbx ? c->meta_pin_d : c->meta;
    if (!mbx) BQ_HIP(hipMemcpyAsync(c->meta, hmeta, sizeof(int) * 2 * AS_SCHUR_MAX, hipMemcpyHostToDevice, st));
    // where the dot products go and how the host learns that they are there
    double *dots_out = mbx ? c->small_pin_d : c->small, *t_out = mbx ? c->small_pin_d + AS_SCHUR_MAX : c->small + 2 * AS_SCHUR_MAX;
    unsigned int *tk = mbx ? w->mail_ticket : nullptr;
    int *post = mbx ? w->mail_d + 1 : nullptr;
    if (!c->y0_valid) {   // while only base variables reach bounds, b0 and Q00^-1 b0 stay what they were
        as_schur_z_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, c->pos0, w->z);
        BQ_TRY(bq_problem_apply(p, w->z, w->Qz, nullptr));
        as_schur_rhs0_kernel<<<gb, 256, 0, st>>>(n0, np0, c->idx0, p->q, w->Qz, ws->rhs);
        BQ_TRY(bq_chol_solve(ws, np0));
        BQ_HIP(hipMemcpyAsync(c->y0, ws->rhs, sizeof(double) * np0, hipMemcpyDeviceToDevice, st));
        c->y0_valid = true;
    }
    double *host_small = c->small_pin, *host_t = c->small_pin + AS_SCHUR_MAX;
    bool have_t = false;
    for (int k = computed; k < m; ++k) {   // the columns of the new slots and their rows of C
        double *uk = c->U + (int64_t)k * c->cap, *wk = c->W + (int64_t)k * c->cap;
        // the column goes to its slot AND to the right-hand side of the solve; the solve leaves its result in the slot of W too
        as_schur_col_kernel<T><<<gb, 256, 0, st>>>(c->kind[k], c->var[k], c->kind[k] == 0 ? c->hpos0[(size_t)c->var[k]] : -1, n0,
                                                  np0, c->idx0, AS_PANEL_ARGS(T), uk, ws->rhs);
        // a pinned base variable's column is a unit vector: the forward sweep starts at its row
        BQ_TRY(bq_chol_solve(ws, np0, c->kind[k] == 0 ? (int64_t)c->hpos0[(size_t)c->var[k]] : 0, wk));
        if (k == m - 1) {   // the right-hand side of the small system needs nothing from the host: same pass over U, same round trip
            as_schur_dots_kernel<T><<<m, 256, 0, st>>>(2, k, c->U, wk, c->y0, c->cap, np0, meta, AS_PANEL_ARGS(T), s->mU, s->lb,
                                                      s->ub, p->q, w->Qz, dots_out, t_out, tk, post, ++w->mail_seq[1]);
            if (!mbx) {
                BQ_HIP(hipMemcpyAsync(host_small, c->small, sizeof(double) * (k + 1), hipMemcpyDeviceToHost, st));
                BQ_HIP(hipMemcpyAsync(host_t, c->small + 2 * AS_SCHUR_MAX, sizeof(double) * m, hipMemcpyDeviceToHost, st));
            }
            have_t = true;
        } else {
            as_schur_dots_kernel<T><<<k + 1, 256, 0, st>>>(0, k, c->U, wk, nullptr, c->cap, np0, meta, AS_PANEL_ARGS(T), s->mU,
                                                          s->lb, s->ub, p->q, w->Qz, dots_out, nullptr, tk, post, ++w->mail_seq[1]);
            if (!mbx) BQ_HIP(hipMemcpyAsync(host_small, c->small, sizeof(double) * (k + 1), hipMemcpyDeviceToHost, st));
        }
        as_tick(w, 1);
        BQ_TRY(as_look(s->p->ctx, w, 1));
        as_tick(w, 2);
        const double tc0 = w->timing ? ldl_now() : 0.0;
        for (int i = 0; i <= k; ++i) {
            if (!std::isfinite(host_small[i])) return BQ_OK;
            c->C[(size_t)i * AS_SCHUR_MAX + k] = host_small[i];
            c->C[(size_t)k * AS_SCHUR_MAX + i] = host_small[i];
        }
        if (w->timing) c->t_c += ldl_now() - tc0;
    }
    double *coef = c->coef_pin;
    if (m > 0) {
        if (!have_t) {
            as_schur_dots_kernel<T><<<m, 256, 0, st>>>(1, 0, c->U, c->y0, nullptr, c->cap, np0, meta, AS_PANEL_ARGS(T), s->mU, s->lb,
                                                      s->ub, p->q, w->Qz, t_out, nullptr, tk, post, ++w->mail_seq[1]);
            if (!mbx) BQ_HIP(hipMemcpyAsync(host_t, c->small + 2 * AS_SCHUR_MAX, sizeof(double) * m, hipMemcpyDeviceToHost, st));
            BQ_TRY(as_look(s->p->ctx, w, 1));
        }
        if (!as_ldl_solve(c, m, host_t, coef)) {   // incremental factorisation first, pivoted elimination as the fallback
            c->ldl_n = 0;
            if (!as_small_solve(m, c->C, host_t, coef)) return BQ_OK;
        }
        if (!mbx) BQ_HIP(hipMemcpyAsync(c->small + AS_SCHUR_MAX, coef, sizeof(double) * m, hipMemcpyHostToDevice, st));
    }
    as_tick(w, 3);
    const double *coef_dev = mbx ? c->coef_pin_d : c->small + AS_SCHUR_MAX;
    as_schur_combine_kernel<<<gb, 1024, 0, st>>>(np0, m, c->y0, c->W, c->cap, coef_dev, c->y);
    as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
    {
        const int64_t span = n0 > m ? (n0 > 0 ? n0 : 1) : (int64_t)m;
        as_cand_scatter_kernel<<<dim3((unsigned)((span + 255) / 256)), 256, 0, st>>>(n0, m, c->idx0, meta, s->mL, s->mU, s->lb, s->ub,
                                                                                  c->y, coef_dev, w->cand, w->ints,
                                                                                  mbx ? w->mail_ticket + 1 : nullptr, w->host_ints_d,
                                                                                  mbx ? w->mail_d + 2 : nullptr, ++w->mail_seq[2]);
    }
    BQ_HIP(hipGetLastError());
    if (!mbx) BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    as_tick(w, 4);
    BQ_TRY(as_look(s->p->ctx, w, 2));
    as_tick(w, 5);
    *good = true;
    return BQ_OK;
}

// one restricted solve through the kept factor; *solved = false leaves the iteration to the classic path
static int as_schur_step(bq_solver *s, as_ws *w, int64_t nA, bool *solved) {
    *solved = false;
    BQ_TRY(as_schur_setup(s, w));
    as_schur *c = w->sch;
    int computed = 0;
    bool ok = c->valid;
    if (ok) BQ_TRY(as_schur_event(s, w, &computed, &ok));
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!ok) {
            BQ_TRY(as_schur_refresh(s, w, nA, &ok));
            if (!ok) return BQ_OK;   // not positive definite: classic path (minres branch)
            computed = 0;
        } else if (attempt == 0) {
            c->reused += 1;
        }
        bool good = false;
        if (s->p->storage == BQ_F64)
            BQ_TRY(as_schur_solve_t<double>(s, w, computed, &good));
        else
            BQ_TRY(as_schur_solve_t<float>(s, w, computed, &good));
        if (good) {
            *solved = true;
            return BQ_OK;
        }
        ok = false;   // numerically singular update: start again from a fresh factor of the current set
    }
    c->valid = false;
    return BQ_OK;
}

static bool as_env_on(const char *name) {
    const char *e = getenv(name);
    return !(e && atoi(e) == 0);
}

static void as_pc_free(as_pc *pc);

// The model P = D + Phi Phi' is only used when it leaves every sample a diagonal share D_i / Q_ii of at least this much: a
// feature set that explains (or over-explains: D_i <= 0) the whole diagonal of some sample is a Taylor expansion outside its
// range (2 g |x|^2 not small) and would make P far worse conditioned than Q itself.
constexpr double PC_MIN_DIAG_SHARE = 0.1;

// build the preconditioner's features once per solver (null: the panel's kernel has none, the features do not fit this data,
// or BQ_AS_CG_PC=0)
static int as_pc_create(bq_solver *s, as_pc **out) {
    *out = nullptr;
    bq_problem *p = s->p;
    if (!as_env_on("BQ_AS_CG_PC")) return BQ_OK;
    if (p->X == nullptr || (p->structure != BQ_PLAIN && p->structure != BQ_SVC)) return BQ_OK;
    if (p->kernel != BQ_KERNEL_RBF && !(p->kernel == BQ_KERNEL_LINEAR && p->diag_add > 0.0)) return BQ_OK;
    bq_ctx *ctx = p->ctx;
    // family 2 of the RBF features on BQ_SVC panels: 2 = projected order-2 directions (2d columns), 1 = the class-mean cross term of
    // rounds 3-4 (d columns), 0 = none.  BQ_AS_CG_PC_CLASS=0|1|2 caps it (tests compare them); a family that does not fit PC_MAX_M
    // features, or whose model leaves a sample too little of its diagonal, steps down.
    int fam2 = 0;
    if (p->kernel == BQ_KERNEL_RBF && p->structure == BQ_SVC) {
        const char *e = getenv("BQ_AS_CG_PC_CLASS");
        fam2 = e ? std::max(0, std::min(atoi(e), 2)) : 2;
    }
    std::vector<double> share((size_t)p->n);
    for (; fam2 >= 0; --fam2) {
        const bool classes = fam2 > 0;
        int m = p->kernel == BQ_KERNEL_RBF ? (int)p->d + 1 + fam2 * (int)p->d : (int)p->d;
        if (p->add_one) m += 1;
        if (m > PC_MAX_M) {   // the apply kernel keeps the coefficients of all features in LDS
            if (fam2 == 0) return BQ_OK;
            continue;
        }
        as_pc *pc = new as_pc();
        pc->m = m;
        pc->mp = bq_round_up(m, 128);
        int rc = bq_chol_ws_create(ctx, pc->mp, &pc->ws);
        hipError_t e = hipSuccess;
        pc->m8 = bq_round_up(m, PC_FG);
        if (rc == BQ_OK) e = hipMalloc(&pc->Phi, sizeof(float) * (size_t)pc->m8 * s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->Phi, 0, sizeof(float) * (size_t)pc->m8 * s->ldN, ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->tpart, sizeof(double) * (size_t)s->nblk * pc->mp);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->tticket, sizeof(unsigned int));
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->tticket, 0, sizeof(unsigned int), ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->dinv, sizeof(double) * s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->z, sizeof(double) * s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->Gpart, sizeof(double) * PC_SLICES * pc->mp * pc->mp);
        if (rc == BQ_OK && e == hipSuccess)
            e = hipMemsetAsync(pc->Gpart, 0, sizeof(double) * PC_SLICES * pc->mp * pc->mp, ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->Ginv, sizeof(double) * pc->mp * pc->mp);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->u, sizeof(double) * pc->mp);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->sm_fail, sizeof(int));
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->sm_fail, 0, sizeof(int), ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->prev, (size_t)s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->prev, 0, (size_t)s->ldN, ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->chg, sizeof(int) * (2 + 2 * PC_MAX_CHG));
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->chg, 0, sizeof(int) * (2 + 2 * PC_MAX_CHG), ctx->stream);
        if (rc == BQ_OK && e == hipSuccess && classes) {
            e = hipMalloc(&pc->cls, sizeof(double) * (2 * p->d + 2));
            if (e == hipSuccess) e = hipMemsetAsync(pc->cls, 0, sizeof(double) * (2 * p->d + 2), ctx->stream);
        }
        // Do ALL ranks hold their features?  Every rank must run the SAME inner iteration (each product is a collective): a rank
        // that fell back to plain conjugate gradients alone — or returned an error alone — would leave the others waiting in the
        // next collective for ever.  So the outcome is agreed on (one all-reduce of a flag) and, if any rank has no room, every rank
        // runs unpreconditioned.
        double failed = (rc != BQ_OK || e != hipSuccess) ? 1.0 : 0.0;
        if (failed != 0.0) (void)hipGetLastError();
        if (ctx->world > 1 && ctx->comm_kind != BQ_COMM_SHARE) {
            hipError_t ae = hipMemcpyAsync(s->partials, &failed, sizeof(double), hipMemcpyHostToDevice, ctx->stream);
            int arc = ae == hipSuccess ? bq_exchange_sum(ctx, s->partials, 1) : BQ_ERR_HIP;
            if (arc == BQ_OK) ae = hipMemcpyAsync(&failed, s->partials, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
            if (arc == BQ_OK && ae == hipSuccess) arc = bq_ctx_sync(ctx);   // behind a collective: the bounded wait
            if (arc != BQ_OK || ae != hipSuccess) {
                as_pc_free(pc);
                if (arc == BQ_OK) bq_set_error("agreeing on the preconditioner across ranks failed: %s", hipGetErrorString(ae));
                return arc != BQ_OK ? arc : BQ_ERR_HIP;
            }
        }
        if (failed != 0.0) {   // no room for the features on some rank: all ranks run plain conjugate gradients
            as_pc_free(pc);
            return BQ_OK;
        }
        double *raw = nullptr, *rinv_d = nullptr;   // fam2 == 2: per-sample numbers and the orthonormalising factor (setup only)
        hipError_t fe = hipSuccess;
        int frc = BQ_OK;
        if (classes) {
            unsigned int *ticket = (unsigned int *)(pc->cls + 2 * p->d + 1);   // the spare slot, zeroed above
            as_pc_class_kernel<<<(unsigned)p->d, 256, 0, ctx->stream>>>(p->n, p->d, p->X, p->sgn, pc->cls, ticket);
            fe = hipGetLastError();
        }
        if (fam2 == 2 && fe == hipSuccess) {
            // the 2d x 2d Gram matrix of U_{c,k} = m_c e_k' + e_k m_c' (Frobenius): <U_{c,k}, U_{c',l}> = 2 (m_c.m_c') [k == l] + 2 m_c[l] m_c'[k],
            // its Cholesky factor R (columns whose pivot falls below 1e-8 of their diagonal are dropped: m_- = -m_+ leaves d directions)
            // and Rinv — on the host, from the class means (deterministic: the same on every rank)
            const int d = (int)p->d, n2 = 2 * d;
            std::vector<double> cls((size_t)n2 + 2), mc((size_t)n2);
            fe = hipMemcpyAsync(cls.data(), pc->cls, sizeof(double) * (n2 + 1), hipMemcpyDeviceToHost, ctx->stream);
            if (fe == hipSuccess) frc = bq_ctx_sync(ctx);
            if (fe == hipSuccess && frc == BQ_OK) {
                for (int k = 0; k < d; ++k) {
                    mc[k] = cls[d + k] + cls[k];
                    mc[d + k] = cls[d + k] - cls[k];
                }
                double dots[2][2] = {{0, 0}, {0, 0}};
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        for (int k = 0; k < d; ++k) dots[a][b] += mc[a * d + k] * mc[b * d + k];
                auto gram = [&](int i, int j) {
                    const int a = i / d, k = i % d, b = j / d, l = j % d;
                    return 2.0 * ((k == l ? dots[a][b] : 0.0) + mc[a * d + l] * mc[b * d + k]);
                };
                std::vector<double> R((size_t)n2 * n2, 0.0), Rinv((size_t)n2 * n2, 0.0);
                std::vector<int> kept;
                std::vector<double> c((size_t)n2);
                for (int j = 0; j < n2; ++j) {   // up-looking Cholesky over the kept columns
                    double piv = gram(j, j);
                    const double gjj = piv;
                    for (size_t a = 0; a < kept.size(); ++a) {
                        const int ia = kept[a];
                        double v = gram(ia, j);
                        for (size_t b = 0; b < a; ++b) v -= R[(size_t)kept[b] * n2 + ia] * c[b];
                        c[a] = v / R[(size_t)ia * n2 + ia];
                        piv -= c[a] * c[a];
                    }
                    if (!(piv > 1e-8 * gjj) || !(gjj > 0.0)) continue;   // (numerically) inside the span of the kept ones
                    for (size_t a = 0; a < kept.size(); ++a) R[(size_t)kept[a] * n2 + j] = c[a];
                    R[(size_t)j * n2 + j] = sqrt(piv);
                    kept.push_back(j);
                }
                for (size_t b = 0; b < kept.size(); ++b) {   // Rinv over the kept set by back substitution, column by column
                    const int jb = kept[b];
                    Rinv[(size_t)jb * n2 + jb] = 1.0 / R[(size_t)jb * n2 + jb];
                    for (size_t a = b; a-- > 0;) {
                        const int ia = kept[a];
                        double v = 0.0;
                        for (size_t q = a + 1; q <= b; ++q) v += R[(size_t)ia * n2 + kept[q]] * Rinv[(size_t)kept[q] * n2 + jb];
                        Rinv[(size_t)ia * n2 + jb] = -v / R[(size_t)ia * n2 + ia];
                    }
                }
                fe = hipMalloc(&raw, sizeof(double) * 3 * s->ldN);
                if (fe == hipSuccess) fe = hipMalloc(&rinv_d, sizeof(double) * (size_t)n2 * n2);
                if (fe == hipSuccess) fe = hipMemcpyAsync(rinv_d, Rinv.data(), sizeof(double) * (size_t)n2 * n2, hipMemcpyHostToDevice, ctx->stream);
                if (fe == hipSuccess) frc = bq_ctx_sync(ctx);   // Rinv leaves this scope
            }
        }
        if (fe == hipSuccess && frc == BQ_OK) {
            as_pc_features_kernel<<<(unsigned)(s->ldN / 256), 256, 0, ctx->stream>>>(p->kernel, p->n, p->d, s->ldN, p->X, p->sgn, pc->cls, fam2,
                                                                                     p->gamma, p->add_one ? 1 : 0, p->diag_add, m,
                                                                                     pc->Phi, pc->z, raw);
            if (fam2 == 2)
                as_pc_project_kernel<<<dim3((unsigned)(s->ldN / 64), (unsigned)((2 * p->d + 63) / 64)), 256, 0, ctx->stream>>>(
                    p->n, p->d, s->ldN, p->X, raw, rinv_d, (int)p->d + 1, pc->Phi);
            as_pc_diag_kernel<<<(unsigned)(s->ldN / 256), 256, 0, ctx->stream>>>(p->n, s->ldN, m, pc->Phi, pc->dinv, pc->z);
            fe = hipGetLastError();
        }
        if (fe == hipSuccess && frc == BQ_OK) fe = hipMemcpyAsync(share.data(), pc->z, sizeof(double) * p->n, hipMemcpyDeviceToHost, ctx->stream);
        if (fe == hipSuccess && frc == BQ_OK) frc = bq_ctx_sync(ctx);
        if (raw) hipFree(raw);
        if (rinv_d) hipFree(rinv_d);
        if (fe != hipSuccess || frc != BQ_OK) {
            as_pc_free(pc);
            if (fe != hipSuccess) bq_set_error("building the preconditioner features failed: %s", hipGetErrorString(fe));
            return fe != hipSuccess ? BQ_ERR_HIP : frc;
        }
        double lo = 1.0;
        for (double v : share) lo = std::min(lo, v);
        if (std::isfinite(lo) && (p->kernel == BQ_KERNEL_LINEAR || lo >= PC_MIN_DIAG_SHARE)) {   // linear: the model is exact
            *out = pc;
            return BQ_OK;
        }
        as_pc_free(pc);   // the model leaves some sample too little of its diagonal: the next smaller family
    }
    return BQ_OK;
}

static void as_pc_free(as_pc *pc) {
    if (!pc) return;
    if (pc->ws) bq_chol_ws_destroy(pc->ws);
    for (void *ptr : {(void *)pc->Phi, (void *)pc->dinv, (void *)pc->z, (void *)pc->Gpart, (void *)pc->cls, (void *)pc->Ginv,
                      (void *)pc->u, (void *)pc->sm_fail, (void *)pc->prev, (void *)pc->chg, (void *)pc->tpart, (void *)pc->tticket})
        if (ptr) hipFree(ptr);
    delete pc;
}

// z = P_AA^-1 r (+ r'z and beta on the device)
static int as_pc_apply(bq_solver *s, as_ws *w, int first) {
    as_pc *pc = w->pc;
    hipStream_t st = s->p->ctx->stream;
    static const int tslices = [] {
        const char *e = getenv("BQ_AS_PC_TSLICES");
        return e ? std::max(1, std::min(atoi(e), PC_TSLICES_MAX)) : 1;
    }();
    as_pc_tphi_kernel<<<dim3(vgrid(s->ldN).x, (unsigned)tslices), BQ_VEC_BLOCK, 0, st>>>(pc->m, pc->m8, pc->mp, s->N, s->ldN, pc->Phi, pc->dinv, w->r,
                                                                                        pc->tpart, pc->ws->rhs, pc->tticket, w->cg);
    as_pc_treduce_kernel<<<(unsigned)(pc->mp / 16), 256, 0, st>>>(pc->m8, pc->mp, (int64_t)vgrid(s->ldN).x, pc->tpart, pc->ws->rhs, w->cg);
    as_pc_gemv_kernel<<<(unsigned)((pc->mp + 3) / 4), 256, 0, st>>>(pc->mp, pc->Ginv, pc->ws->rhs, pc->u, w->cg);   // u = G^-1 t
    as_pc_apply_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(pc->m, pc->m8, s->N, s->ldN, pc->Phi, pc->dinv, s->mL, s->mU, w->r,
                                                               pc->u, pc->z, s->partials, s->nblk, w->cg, first);
    return BQ_OK;
}

// the restricted solve of one outer iteration by conjugate gradients; leaves the candidate in w->cand and the
// feasibility flag in w->host_ints[2].
//   start: the candidate of the previous outer iteration (the free set has moved by one index since, so it solves the new
//   system up to one column of Q), else the current point;  BQ_AS_CG_WARM=0: always the current point
//   preconditioner: struct as_pc (RBF and linear panels);   BQ_AS_CG_PC=0: none
static int as_cg_solve_once(bq_solver *s, as_ws *w, int *pc_failed) {
    *pc_failed = 0;
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    const int64_t N = s->N, nblk = s->nblk;
    const int64_t nA = w->host_ints[0];
    const dim3 grid = vgrid(s->ldN);
    const long long cap = s->inner_max > 0 ? s->inner_max : 2 * (long long)nA + 50;
    as_pc *pc = w->pc;
    const double *start = (w->warm && w->have_cand) ? w->cand : s->x;
    as_make_xt_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, start, w->z);
    const bool anchor = w->anchor;
    w->anchor = false;
    if (w->colq && start == w->cand && !anchor) {
        // z is the previous candidate except at the variables that reached a bound since: Q z = Q cand + those columns of Q,
        // formed from X — the product below returns at once (its `done` flag) unless too many variables moved
        bq_problem *p = s->p;
        int *lcnt = reinterpret_cast<int *>(s->partials + s->nblk);   // the slice as_launch_compact uses between its two passes
        as_zdiff_count_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, lcnt, &s->sc->pad1[0], w->zchg);
        as_zdiff_write_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, lcnt, w->zchg, w->zdl);
        BQ_TRY(bq_problem_apply(p, w->z, w->Qz, w->zchg + 1));
        as_qz_cols_kernel<<<(unsigned)std::min<int64_t>((p->n + 3) / 4, 8192), 256, 0, st>>>(p->n, p->d, p->X, w->sq, p->sgn, p->kernel, p->gamma, p->coef0,
                                                                         p->degree, p->add_one ? 1 : 0, p->diag_add,
                                                                         p->storage == BQ_F32 ? 1 : 0, w->zchg, w->zdl, w->Qcand, w->Qz);
    } else {
        BQ_TRY(bq_problem_apply(s->p, w->z, w->Qz, nullptr));
    }
    BQ_HIP(hipMemsetAsync(w->Qdl, 0, sizeof(double) * s->ldN, st));
    // s->Qd = Q x of the current point (eval_f at the end of the previous outer iteration, or of bq_as_start)
    as_cg_init_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, w->Qz, start == s->x ? w->Qz : s->Qd, s->p->q, w->dlt, w->r,
                                                     w->pv, s->partials, nblk, w->cg, s->inner_rtol, cap, pc ? 1 : 0);
    const double *zr = w->r;   // what the next direction is built from: the residual, or the preconditioned residual
    if (pc) {
        // G^-1, G = I + Phi_A' D_A^-1 Phi_A, for the free set of this outer iteration.  The set moves by an index or two per outer
        // iteration (the list was made at the top of the iteration, as_pc_diff_*): G^-1 follows by Sherman-Morrison updates in index
        // order (the same bits on every rank), and is rebuilt from G summed afresh over all samples at the start, every 128 outer
        // iterations, when more than 64 samples moved at once, and after an update that failed.
        const bool rebuild = pc->host_chg[1] != 0 || pc->age == 0;
        pc->age = rebuild ? 1 : pc->age + 1;
        if (rebuild) {
            const int tiles = (int)(pc->mp / PC_T);
            const dim3 gtri((unsigned)((pc->mp + 255) / 256), (unsigned)pc->mp);
            as_pc_gram_kernel<<<dim3((unsigned)(tiles * (tiles + 1) / 2), PC_SLICES), 256, 0, st>>>(pc->m, pc->mp, N, s->ldN, pc->Phi,
                                                                                                   pc->dinv, s->mL, s->mU, pc->Gpart);
            as_pc_gram_reduce_kernel<<<gtri, 256, 0, st>>>(pc->m, pc->mp, pc->Gpart, pc->ws->H, pc->ws->ldh);
            BQ_TRY(bq_chol_factor(pc->ws, pc->mp));
            BQ_TRY(bq_chol_prepare_sweeps(pc->ws, pc->mp));   // the explicit inverse factor (mp <= 1024: one block)
            as_pc_ginv_kernel<<<dim3((unsigned)(pc->mp / 16), (unsigned)(pc->mp / 16)), 256, 0, st>>>(pc->mp, pc->ws->bigMT, 1024, pc->Ginv);
            BQ_HIP(hipMemsetAsync(pc->sm_fail, 0, sizeof(int), st));
            pc->rebuilds += 1;
        } else {
            as_pc_sm_kernel<<<1, 1024, 0, st>>>(pc->m, pc->mp, s->ldN, pc->Phi, pc->dinv, pc->chg, pc->Ginv, pc->sm_fail);
        }
        BQ_TRY(as_pc_apply(s, w, 1));
        as_cg_dir_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, pc->z, w->pv, w->cg);   // beta = 0: p = z
        zr = pc->z;
    }
    // The host looks at the `done` flag after EVERY inner iteration, one iteration late: iteration k + 1 is enqueued, then the
    // host waits for the copy of the flag recorded behind iteration k.  The device never runs dry (an iteration is a panel
    // product: 2.5 - 20 ms at BASELINE config 5, far longer than the host's turn) and exactly ONE enqueued iteration is wasted per
    // solve — it returns at once on the flag — where round 3's batches of 8 -> 32 wasted 12 launches per solve, 0.17 ms each for the
    // empty 60 000-workgroup product grid alone (profiles/r04/c5_per_outer_iteration_kernel_ms_before.csv).  Every rank sees the
    // same flag values at the same iteration (replicated, bit-identical scalars), so all ranks enqueue the same collectives.
    long long queued = 0;
    BQ_HIP(hipMemcpyAsync(w->cg_flag_host, &w->cg->done, 2 * sizeof(int), hipMemcpyDeviceToHost, st));   // the flag the start leaves
    BQ_HIP(hipEventRecord(w->cg_event, st));
    while (queued < cap) {
        BQ_TRY(bq_problem_apply(s->p, w->pv, w->Qp, &w->cg->done));
        as_cg_pap_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, w->pv, w->Qp, s->partials, nblk, w->cg);
        as_cg_update_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, w->dlt, w->r, w->pv, w->Qp, w->Qdl, s->partials,
                                                           nblk, w->cg);
        if (pc) BQ_TRY(as_pc_apply(s, w, 0));
        as_cg_dir_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, zr, w->pv, w->cg);
        ++queued;
        BQ_TRY(bq_ctx_event_sync(ctx, w->cg_event));   // the flag as it stood BEFORE the iteration just enqueued
        if (w->cg_flag_host[0]) break;
        BQ_HIP(hipMemcpyAsync(w->cg_flag_host, &w->cg->done, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        BQ_HIP(hipEventRecord(w->cg_event, st));
    }
    BQ_HIP(hipMemcpyAsync(w->host_cg, w->cg, sizeof(as_cg_scal), hipMemcpyDeviceToHost, st));
    w->host_info[1] = w->host_info[2] = 0;
    if (pc) BQ_HIP(hipMemcpyAsync(w->host_info + 1, pc->ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    if (pc) BQ_HIP(hipMemcpyAsync(w->host_info + 2, pc->sm_fail, sizeof(int), hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    const int pc_info = w->host_info[1], sm_fail = w->host_info[2];
    const as_cg_scal h = *w->host_cg;
    w->cg_iters += h.iters;
    if (pc_info != 0 || sm_fail != 0 || h.info == 2) {   // the caller sums G afresh and tries again, then gives the preconditioner up
        bq_set_error("the preconditioner of the inner conjugate gradients is not positive definite (pivot %d, |A| = %lld): "
                     "BQ_AS_CG_PC=0 runs without it", pc_info, (long long)nA);
        *pc_failed = 1;
        return BQ_OK;
    }
    if (h.info != 0 || !std::isfinite(h.rr)) {
        bq_set_error("conjugate gradients on the restricted Hessian Q[A,A] (|A| = %lld) met a direction of non-positive "
                     "curvature after %lld iterations: the system is not positive definite",
                     (long long)nA, (long long)h.iters);
        return BQ_ERR_NOT_PD;
    }
    as_qcand_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, w->Qz, w->Qdl, w->Qcand);
    as_cg_gather_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(w->ints, w->idx, w->z, w->dlt, w->sol, N);
    as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
    as_cand_scatter_idx_kernel<<<vgrid(s->ldN).x * (BQ_VEC_TILE / 256), 256, 0, st>>>(w->idx, w->ints, w->sol, s->lb, s->ub, w->cand);
    w->have_cand = true;
    BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    return BQ_OK;
}

// A Woodbury system G that is not positive definite (or r'z <= 0) says something about the MODEL P, not about Q: its Gram matrix
// has drifted under the rank-one updates, or the features do not fit this free set.  First G is summed afresh and the solve
// repeated; if that fails too the preconditioner is dropped for the rest of the run (plain conjugate gradients).  Every rank
// reads the same replicated scalars, so all ranks take the same turn here and stay in the same collectives.
static int as_cg_solve(bq_solver *s, as_ws *w) {
    for (int attempt = 0; attempt < 3; ++attempt) {
        int pc_failed = 0;
        BQ_TRY(as_cg_solve_once(s, w, &pc_failed));
        if (!pc_failed) return BQ_OK;
        if (w->pc == nullptr) break;
        if (attempt == 0) {
            w->pc->age = 0;
            w->pc_rebuilds += 1;
        } else {
            as_pc_free(w->pc);
            w->pc = nullptr;
            w->pc_dropped += 1;
        }
    }
    return BQ_ERR_NOT_PD;
}

int bq_as_start(bq_solver *s) {
    if (s->chol) {   // a numerically singular Q_AA must take the minres branch, not the sign of its rounding noise (bq_chol.h)
        const char *e = getenv("BQ_AS_PIVOT_REL");
        s->chol->pivot_rel = e ? atof(e) : 1e-13;
    }
    bq_ctx *ctx = s->p->ctx;
    as_ws *w = new as_ws();
    s->as_ws = w;
    BQ_HIP(hipMalloc(&w->idx, sizeof(int) * (s->N + 1)));
    BQ_HIP(hipMemsetAsync(w->idx, 0, sizeof(int) * (s->N + 1), ctx->stream));
    BQ_HIP(hipMalloc(&w->ints, sizeof(int) * 32));
    BQ_HIP(hipMemsetAsync(w->ints, 0, sizeof(int) * 32, ctx->stream));
    BQ_HIP(hipHostMalloc(&w->host_ints, sizeof(int) * 32, AS_MAPPED));
    BQ_HIP(hipHostMalloc(&w->host_scal, sizeof(bq_scal), AS_MAPPED));
    BQ_HIP(hipHostMalloc(&w->mail, sizeof(int) * 16, AS_MAPPED));
    memset(w->mail, 0, sizeof(int) * 16);
    w->mail_d = as_dev(w->mail);
    w->host_ints_d = as_dev(w->host_ints);
    w->host_scal_d = reinterpret_cast<int *>(as_dev(w->host_scal));
    BQ_HIP(hipMalloc(&w->mail_ticket, sizeof(unsigned int) * 2));
    BQ_HIP(hipMemsetAsync(w->mail_ticket, 0, sizeof(unsigned int) * 2, s->p->ctx->stream));
    w->mailbox = !s->as_cg && as_env_on("BQ_AS_MAILBOX");
    w->f_chain = !s->as_cg && as_env_on("BQ_AS_F_CHAIN");
    if (const char *e = getenv("BQ_AS_TIMING")) w->timing = atoi(e) != 0;
    BQ_HIP(hipHostMalloc(&w->host_info, sizeof(int) * 8));
    BQ_HIP(hipHostMalloc(&w->host_cg, sizeof(as_cg_scal)));
    memset(w->host_ints, 0, sizeof(int) * 32);
    memset(w->host_info, 0, sizeof(int) * 8);
    for (double **v : {&w->cand, &w->z, &w->Qz, &w->x_eval, &w->g_eval}) {
        BQ_HIP(hipMalloc(v, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(*v, 0, sizeof(double) * s->ldN, ctx->stream));
    }
    if (s->as_cg) {
        for (double **v : {&w->dlt, &w->r, &w->pv, &w->Qp, &w->sol, &w->Qdl, &w->Qcand}) {
            BQ_HIP(hipMalloc(v, sizeof(double) * s->ldN));
            BQ_HIP(hipMemsetAsync(*v, 0, sizeof(double) * s->ldN, ctx->stream));
        }
        BQ_HIP(hipMalloc(&w->cg, sizeof(as_cg_scal)));
        BQ_HIP(hipMemsetAsync(w->cg, 0, sizeof(as_cg_scal), ctx->stream));
        BQ_HIP(hipHostMalloc(&w->cg_flag_host, 2 * sizeof(int)));
        BQ_HIP(hipEventCreateWithFlags(&w->cg_event, hipEventDisableTiming));
        w->warm = as_env_on("BQ_AS_CG_WARM");
        w->incq = as_env_on("BQ_AS_CG_INCQ");
        {
            bq_problem *p = s->p;
            w->colq = w->warm && as_env_on("BQ_AS_CG_COLQ") && p->X != nullptr && !p->streamed &&
                      (p->structure == BQ_PLAIN || p->structure == BQ_SVC) && p->kernel >= BQ_KERNEL_LINEAR &&
                      p->kernel <= BQ_KERNEL_SIGMOID;
            if (w->colq) {
                BQ_HIP(hipMalloc(&w->sq, sizeof(double) * s->ldN));
                BQ_HIP(hipMalloc(&w->zchg, sizeof(int) * (2 + AS_MAX_COLS)));
                BQ_HIP(hipMemsetAsync(w->zchg, 0, sizeof(int) * (2 + AS_MAX_COLS), ctx->stream));
                BQ_HIP(hipMalloc(&w->zdl, sizeof(double) * AS_MAX_COLS));
                as_row_norms_kernel<<<(unsigned)(s->ldN / 256), 256, 0, ctx->stream>>>(p->X, p->n, p->d, w->sq, s->ldN);
            }
        }
        BQ_TRY(as_pc_create(s, &w->pc));
    }
    BQ_HIP(hipMalloc(&s->mL, (size_t)s->ldN));
    BQ_HIP(hipMalloc(&s->mU, (size_t)s->ldN));
    BQ_HIP(hipMemsetAsync(s->mL, 0, (size_t)s->ldN, ctx->stream));
    BQ_HIP(hipMemsetAsync(s->mU, 0, (size_t)s->ldN, ctx->stream));
    // bq_solver_set_state: continue from given masks L / U (the reference starts with both empty, active_set.py:91-92).  The host
    // vectors live until solver_first has synchronised.
    const bool resumed = s->resume && (s->resume->have & BQ_STATE_MASKS);
    if (resumed) {
        BQ_HIP(hipMemcpyAsync(s->mL, s->resume->mL.data(), (size_t)s->N, hipMemcpyHostToDevice, ctx->stream));
        BQ_HIP(hipMemcpyAsync(s->mU, s->resume->mU.data(), (size_t)s->N, hipMemcpyHostToDevice, ctx->stream));
    }
    if (w->f_chain && resumed) {
        // the product-free f of ratio steps needs a point that solves its restricted system and the gradient there: neither is
        // known for a restored point, so the first run starts at the next release (as_finish_iteration), as after a minres step
        BQ_HIP(hipMalloc(&w->g0, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(w->g0, 0, sizeof(double) * s->ldN, ctx->stream));
        static const double one = 1.0;
        BQ_HIP(hipMemcpyAsync(&s->sc->aux[0], &one, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        w->gref = nullptr;
        w->chain_ok = false;
        return eval_f(s, nullptr);
    }
    if (w->f_chain) {
        // the gradient at x0 rides on the product of f(x0): with every index free, x0 + t d obeys the ratio step's identity from the
        // first iteration on (as_step_min_kernel).  Kept apart from s->g, which the reference does not touch before a release.
        BQ_HIP(hipMalloc(&w->g0, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(w->g0, 0, sizeof(double) * s->ldN, ctx->stream));
        static const double one = 1.0;   // gamma of the first run (as_release_mb_kernel sets it for the later ones)
        BQ_HIP(hipMemcpyAsync(&s->sc->aux[0], &one, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        w->gref = w->g0;
        w->chain_ok = true;
        return eval_f(s, w->g0);
    }
    return eval_f(s, nullptr);  // f(x0), active_set.py:84
}

void bq_as_free(bq_solver *s) {
    as_ws *w = get_ws(s);
    if (!w) return;
    if (w->timing && w->tm[7] > 0) {
        const double it = w->tm[7];
        fprintf(stderr, "BQ_AS_TIMING  %.0f iterations, us per iteration: wait top %.1f | host to dots %.1f | wait dots %.1f | small system %.1f | "
                        "launch to candidate %.1f | wait candidate %.1f | launch branch + top %.1f\n", it, w->tm[0] / it, w->tm[1] / it, w->tm[2] / it,
                w->tm[3] / it, w->tm[4] / it, w->tm[5] / it, w->tm[6] / it);
        if (w->sch)
            fprintf(stderr, "BQ_AS_TIMING  host LDL, us per iteration: new row %.1f | forward %.1f | back %.1f | residual check %.1f | C row/column %.1f\n",
                    w->sch->t_ldl[0] / it, w->sch->t_ldl[1] / it, w->sch->t_ldl[2] / it, w->sch->t_ldl[3] / it, w->sch->t_c / it);
        fprintf(stderr, "BQ_AS_TIMING  iterations that factorised Q[A,A] afresh: %lld, mean order %.0f, %.1f us each (launch to the candidate's record)\n", w->n_classic,
                w->classic_order / std::max(1.0, (double)w->n_classic), w->tm_classic / std::max(1.0, (double)w->n_classic));
        if (w->sch)
            fprintf(stderr, "BQ_AS_TIMING  small system: %lld solves, mean order %.1f, rows (re)factorised per solve %.2f, slots dropped %lld, base factorisations %lld\n",
                    w->sch->reused + w->sch->refreshes, (double)w->sch->rows_solved / std::max(1.0, (double)(w->sch->reused + w->sch->refreshes)),
                    (double)w->sch->rows_extended / std::max(1.0, (double)(w->sch->reused + w->sch->refreshes)), w->sch->drops, w->sch->refreshes);
    }
    for (void *p : {(void *)w->idx, (void *)w->ints, (void *)w->cand, (void *)w->z, (void *)w->Qz, (void *)w->x_eval,
                    (void *)w->g_eval, (void *)w->dlt, (void *)w->r, (void *)w->pv, (void *)w->Qp, (void *)w->sol,
                    (void *)w->cg, (void *)w->Qdl, (void *)w->Qcand, (void *)w->sq, (void *)w->zchg, (void *)w->zdl,
                    (void *)w->mail_ticket, (void *)w->g0})
        if (p) hipFree(p);
    for (void *hp : {(void *)w->host_ints, (void *)w->host_scal, (void *)w->host_info, (void *)w->host_cg, (void *)w->mail})
        if (hp) hipHostFree(hp);
    if (w->cg_flag_host) hipHostFree(w->cg_flag_host);
    if (w->cg_event) hipEventDestroy(w->cg_event);
    as_pc_free(w->pc);
    as_schur_free(w);
    delete w;
    s->as_ws = nullptr;
}

// x / g as the callback of the last recorded iteration must see them (the body has already moved on)
long long bq_as_inner_iters(bq_solver *s) {
    as_ws *w = get_ws(s);
    return w ? w->cg_iters : 0;
}

long long bq_as_counter(bq_solver *s, int which) {
    as_ws *w = get_ws(s);
    if (!w) return 0;
    switch (which) {
        case BQ_COUNT_INNER: return w->cg_iters;
        case BQ_COUNT_MINRES: return w->minres_calls;
        case BQ_COUNT_REFACTOR: return w->sch ? w->sch->refreshes : 0;
        case BQ_COUNT_REUSED: return w->sch ? w->sch->reused : 0;
        case BQ_COUNT_NO_PRODUCT: return w->chain_steps;
        default: return 0;
    }
}

const double *bq_as_view(bq_solver *s, int what) {
    as_ws *w = get_ws(s);
    if (!w || !s->started) return what == BQ_GET_X ? s->x : s->g;
    return what == BQ_GET_X ? w->x_eval : w->g_eval;
}

int bq_as_iterate(bq_solver *s) {
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    as_ws *w = get_ws(s);
    bq_chol_ws *ws = s->chol;
    const int64_t N = s->N;

    as_launch_compact(s, w, st);
    const bool mbx = w->mailbox;
    as_top_kernel<<<1, 64, 0, st>>>(s->sc, w->ints, s->stats, (int)N, w->host_ints_d, w->host_scal_d,
                                   mbx ? w->mail_d : nullptr, mbx ? ++w->mail_seq[0] : 0);
    if (!s->host.done) {  // snapshot of the point this record describes
        as_copy2_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->x, w->x_eval, s->g, w->g_eval);
        s->started = true;
    }
    if (s->as_cg && w->pc) {
        // which samples entered / left the free set since the preconditioner's G^-1 was brought up to date: the list (and whether
        // it is short enough for rank-one updates) rides on this iteration's one look at the device, so that the host knows
        // whether to enqueue the update kernel or a rebuild without a synchronisation of its own
        as_pc *pc = w->pc;
        const int force = (pc->age == 0 || pc->age >= 128 || !as_env_on("BQ_AS_CG_PC_INCR")) ? 1 : 0;
        int *lcnt = reinterpret_cast<int *>(s->partials + s->nblk);
        as_pc_diff_count_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, pc->prev, lcnt, &s->sc->pad1[0], pc->chg, force);
        as_pc_diff_write_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, pc->prev, lcnt, pc->chg);
        BQ_HIP(hipMemcpyAsync(w->host_info + 4, pc->chg, 2 * sizeof(int), hipMemcpyDeviceToHost, st));   // pinned, like the rest
    }
    if (!mbx) {
        BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
        BQ_HIP(hipMemcpyAsync(w->host_scal, s->sc, sizeof(bq_scal), hipMemcpyDeviceToHost, st));
    }
    BQ_HIP(hipGetLastError());
    as_tick(w, w->tm[7] > 0 ? 6 : -1);
    BQ_TRY(as_look(ctx, w, 0));
    as_tick(w, 0);
    w->tm[7] += 1;
    s->host = *w->host_scal;
    if (s->as_cg && w->pc) {
        w->pc->host_chg[0] = w->host_info[4];
        w->pc->host_chg[1] = w->host_info[5];
    }
    if (s->host.done) return BQ_OK;
    const int64_t nA = w->host_ints[0];

    if (s->as_cg) {
        BQ_TRY(as_cg_solve(s, w));
        // Q x of the new point without a product: Q cand is known from the inner iteration (Q z + Q delta), the ratio step is
        // a convex combination.  Every 64th outer iteration forms it afresh so that rounding cannot accumulate.
        const bool inc = w->incq && ++w->since_refresh < 64;
        if (!inc) {
            w->since_refresh = 0;
            // ... and the start product of the NEXT solve is a real product too: Q z = Q cand + columns, Q cand = Q z + Q delta is a
            // chain that the refresh of Q x alone does not re-anchor (the columns are rounded like the panel's entries,
            // not bit-equal to them, and BASELINE config 5 runs ~n outer iterations)
            w->anchor = true;
        }
        if (w->host_ints[2]) {
            as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, w->cand, s->x);
            if (inc) {
                as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, w->Qcand, s->Qd);
                BQ_TRY(bq_vec_eval_f(s->p, s->x, s->Qd, s->g, &s->sc->f));
            } else {
                BQ_TRY(eval_f(s, s->g));
            }
            as_release_mb_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->g, s->mL, s->mU, s->sc, w->ints, s->stats);
        } else {
            as_launch_step(s, w, st);
            if (inc) {
                as_qx_lerp_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->sc, w->Qcand, s->Qd);
                BQ_TRY(bq_vec_eval_f(s->p, s->x, s->Qd, nullptr, &s->sc->f));
            } else {
                BQ_TRY(eval_f(s, nullptr));
            }
            as_launch_absorb(s, w, st);
        }
        BQ_HIP(hipGetLastError());
        return BQ_OK;
    }
    bool solved = false;
    // (an empty free set has nothing to keep: with BQ_AS_SCHUR_MIN=0 it reached the kept-factor path and launched empty grids)
    if (as_schur_enabled() && nA > 0 && nA >= as_schur_min()) BQ_TRY(as_schur_step(s, w, nA, &solved));
    if (!solved && w->sch) w->sch->valid = false;   // the classic path below overwrites the kept factor
    if (solved) {
        w->last_branch = w->host_ints[2] ? 1 : 0;
        return as_finish_iteration(s, w, st, true);
    }
    w->n_classic += 1;
    w->classic_order += (double)nA;
    const double tcl0 = w->timing ? ldl_now() : 0.0;
    struct classic_tail {
        as_ws *w;
        double t0;
        ~classic_tail() {
            if (w->timing) w->tm_classic += ldl_now() - t0;
        }
    } ctail{w, tcl0};
    as_make_z_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->z);
    BQ_TRY(bq_problem_apply(s->p, w->z, w->Qz, nullptr));
    int64_t np = 0;
    BQ_TRY(bq_chol_build_h(ws, s->p, w->idx, nA, nullptr, &np));
    as_gather_rhs_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(w->ints, w->idx, s->p->q, w->Qz, ws->rhs, np);
    BQ_TRY(bq_chol_factor(ws, np));
    BQ_TRY(bq_chol_solve(ws, np));
    as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
    as_cand_scatter_idx_kernel<<<vgrid(s->ldN).x * (BQ_VEC_TILE / 256), 256, 0, st>>>(w->idx, w->ints, ws->rhs, s->lb, s->ub, w->cand);
    BQ_HIP(hipMemcpyAsync(w->host_info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    const int info = w->host_info[0];
    if (info != 0) {
        // Q[A,A] is not positive definite: the reference's bare `except` switches to scipy's minres on the normal
        // equations (active_set.py:142-151).  Rebuild the (destroyed) restricted Hessian with both triangles, solve,
        // and redo the feasibility test on the minimum-residual candidate.
        if (!ws->mr_vec) BQ_HIP(hipMalloc(&ws->mr_vec, sizeof(double) * 10 * ws->cap));
        BQ_TRY(bq_chol_build_h(ws, s->p, w->idx, nA, nullptr, &np, true));
        as_gather_rhs_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(w->ints, w->idx, s->p->q, w->Qz, ws->rhs, np);
        BQ_TRY(bq_minres_normal(ws, w->ints, nA, np, ws->mr_vec, w->ints + 7));
        as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
        as_cand_scatter_idx_kernel<<<vgrid(s->ldN).x * (BQ_VEC_TILE / 256), 256, 0, st>>>(w->idx, w->ints, ws->rhs, s->lb, s->ub, w->cand);
        BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
        BQ_SYNC(s->p->ctx);
        w->minres_calls += 1;
    }
    w->last_branch = w->host_ints[2] ? 1 : 0;
    return as_finish_iteration(s, w, st, info == 0);
}
