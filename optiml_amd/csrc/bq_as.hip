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
// The restricted order |A| changes every iteration, so the host reads two small records per iteration (|A| and
// {pivot info, feasible}); everything else stays on the stream.  When Q[A,A] is not positive definite the reference
// silently switches to scipy's minres on the normal equations (:142-151): here a persistent single-workgroup MINRES
// kernel (bq_minres.hip) takes over for |A| <= 8192.
#include <cmath>

#include "bq_chol.h"

#define ACT_TOL 1e-12

#define VEC_LOOP(i)                                                             \
    const int64_t _base = (int64_t)blockIdx.x * BQ_VEC_TILE + threadIdx.x;      \
    _Pragma("unroll") for (int _j = 0; _j < BQ_VEC_ITEMS; ++_j)                 \
        for (int64_t i = _base + (int64_t)_j * BQ_VEC_BLOCK, _once = 1; _once; _once = 0)

static inline dim3 vgrid(int64_t ldN) { return dim3((unsigned)(ldN / BQ_VEC_TILE)); }

struct as_ws {
    int *idx = nullptr;        // compacted free set
    int *ints = nullptr;       // [0] nA, [1] nB, [2] feasible, [3] h_lower, [4] h_upper, [5] nL_new, [6] nU_new
    double *cand = nullptr;    // candidate point (ldN)
    double *z = nullptr;       // bound contribution vector (ldN)
    double *Qz = nullptr;      // (ldN)
    double *x_eval = nullptr;  // x / g at the top of the current iteration (what a callback must see)
    double *g_eval = nullptr;
    int host_ints[8];
    long long minres_calls = 0;
};

__device__ __forceinline__ double as_wmin(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}

// idx = ascending indices of the free set A = !(L | U); ints[0] = |A|, ints[1] = |L| + |U|
__global__ __launch_bounds__(256) void as_compact_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                         const unsigned char *__restrict__ mU, int *__restrict__ idx,
                                                         int *__restrict__ ints) {
    __shared__ int wtot[4];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < N; c0 += 256) {
        const int64_t i = c0 + tid;
        const int flag = (i < N) && !(mL[i] | mU[i]);
        const unsigned long long bal = __ballot(flag);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += wtot[w];
        if (flag) idx[off + within] = (int)i;
        __syncthreads();
        if (tid == 0) base += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    if (tid == 0) {
        ints[0] = base;
        ints[1] = (int)N - base;
    }
}

__global__ void as_top_kernel(bq_scal *sc, const int *__restrict__ ints, bq_iter_stat *stats) {
    if (sc->done) return;
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

// cand = z on the bounds, solution on A; ints[2] = every free coordinate inside [lb - tol, ub + tol]
__global__ __launch_bounds__(256) void as_candidate_kernel(const int *__restrict__ idx, int *__restrict__ ints,
                                                           const double *__restrict__ sol, const double *__restrict__ z,
                                                           int64_t N, const double *__restrict__ lb,
                                                           const double *__restrict__ ub, double *__restrict__ cand) {
    // single block: N is swept twice, the feasibility flag needs no atomics
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < N; i += 256) cand[i] = z[i];
    __syncthreads();
    const int nA = ints[0];
    int mybad = 0;
    for (int a = threadIdx.x; a < nA; a += 256) {
        const int i = idx[a];
        const double v = sol[a];
        cand[i] = v;
        if (!(v <= ub[i] + ACT_TOL && v >= lb[i] - ACT_TOL)) mybad = 1;
    }
    if (mybad) bad = 1;  // benign race: every writer stores 1
    __syncthreads();
    if (threadIdx.x == 0) ints[2] = bad ? 0 : 1;
}

__global__ void as_copy_kernel(int64_t N, const double *__restrict__ src, double *__restrict__ dst) {
    VEC_LOOP(i) {
        if (i < N) dst[i] = src[i];
    }
}

// first index of L with g < -tol (ints[3]) and of U with g > tol (ints[4]); N if none.  Single block.
__global__ __launch_bounds__(256) void as_release_kernel(int64_t N, const double *__restrict__ g,
                                                         unsigned char *__restrict__ mL, unsigned char *__restrict__ mU,
                                                         bq_scal *sc, int *__restrict__ ints, bq_iter_stat *stats) {
    __shared__ long long sl[256], su[256];
    long long hl = N, hu = N;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        if (mL[i] && g[i] < -ACT_TOL && i < hl) hl = i;
        if (mU[i] && g[i] > ACT_TOL && i < hu) hu = i;
    }
    sl[threadIdx.x] = hl;
    su[threadIdx.x] = hu;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sl[threadIdx.x] = sl[threadIdx.x] < sl[threadIdx.x + s] ? sl[threadIdx.x] : sl[threadIdx.x + s];
            su[threadIdx.x] = su[threadIdx.x] < su[threadIdx.x + s] ? su[threadIdx.x] : su[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hl = sl[0];
        hu = su[0];
        ints[3] = (int)hl;
        ints[4] = (int)hu;
        const long long row = sc->iter - sc->stat_base;
        const bool rec = row >= 0 && row < sc->stat_cap;
        if (hl < N) {
            mL[hl] = 0;
            if (rec) {
                stats[row].r2 = 1.0;
                stats[row].r3 = (double)hl;
            }
            sc->iter += 1;
        } else if (hu < N) {
            mU[hu] = 0;
            if (rec) {
                stats[row].r2 = 2.0;
                stats[row].r3 = (double)hu;
            }
            sc->iter += 1;
        } else {
            if (rec) stats[row].r2 = 3.0;
            sc->status = BQ_STATUS_OPTIMAL;
            sc->done = 1;
        }
    }
}

// ratio step towards the candidate on the free set:  x += max_t (cand - x)     (single block)
__global__ __launch_bounds__(256) void as_step_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                      const unsigned char *__restrict__ mU, const double *__restrict__ cand,
                                                      const double *__restrict__ lb, const double *__restrict__ ub,
                                                      double *__restrict__ x, bq_scal *sc) {
    __shared__ double sh[4];
    double rmin = INFINITY;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        if (mL[i] | mU[i]) continue;
        const double d = cand[i] - x[i];
        if (d > 0.0) rmin = fmin(rmin, (ub[i] - x[i]) / d);
        if (d < 0.0) rmin = fmin(rmin, (lb[i] - x[i]) / d);
    }
    rmin = as_wmin(rmin);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = rmin;
    __syncthreads();
    const double t = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        if (mL[i] | mU[i]) continue;
        const double d = cand[i] - x[i];
        x[i] = x[i] + __dmul_rn(t, d);
    }
    if (threadIdx.x == 0) sc->step = t;
}

// move free variables that reached a bound into L / U (L first, as the reference), count them, advance iter
__global__ __launch_bounds__(256) void as_absorb_kernel(int64_t N, unsigned char *__restrict__ mL,
                                                        unsigned char *__restrict__ mU, const double *__restrict__ x,
                                                        const double *__restrict__ lb, const double *__restrict__ ub,
                                                        bq_scal *sc, int *__restrict__ ints, bq_iter_stat *stats) {
    __shared__ int cl[256], cu[256];
    int nl = 0, nu = 0;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        if (mL[i] | mU[i]) continue;
        if (x[i] <= lb[i] + ACT_TOL) {
            mL[i] = 1;
            ++nl;
        } else if (x[i] >= ub[i] - ACT_TOL) {
            mU[i] = 1;
            ++nu;
        }
    }
    cl[threadIdx.x] = nl;
    cu[threadIdx.x] = nu;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            cl[threadIdx.x] += cl[threadIdx.x + s];
            cu[threadIdx.x] += cu[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ints[5] = cl[0];
        ints[6] = cu[0];
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            stats[row].r2 = 0.0;
            stats[row].r3 = (double)(((long long)cl[0] << 32) | (long long)cu[0]);
        }
        sc->iter += 1;
    }
}

static as_ws *get_ws(bq_solver *s) { return reinterpret_cast<as_ws *>(s->as_ws); }

static int eval_f(bq_solver *s, double *g_out) {
    // Qd = Q x ; f = 1/2 x'Qx + q'x -> sc->f ; optionally g = Qx + q
    BQ_TRY(bq_problem_apply(s->p, s->x, s->Qd, nullptr));
    return bq_vec_eval_f(s->p, s->x, s->Qd, g_out, &s->sc->f);
}

int bq_as_start(bq_solver *s) {
    bq_ctx *ctx = s->p->ctx;
    as_ws *w = new as_ws();
    s->as_ws = w;
    BQ_HIP(hipMalloc(&w->idx, sizeof(int) * (s->N + 1)));
    BQ_HIP(hipMemsetAsync(w->idx, 0, sizeof(int) * (s->N + 1), ctx->stream));
    BQ_HIP(hipMalloc(&w->ints, sizeof(int) * 8));
    BQ_HIP(hipMemsetAsync(w->ints, 0, sizeof(int) * 8, ctx->stream));
    for (double **v : {&w->cand, &w->z, &w->Qz, &w->x_eval, &w->g_eval}) {
        BQ_HIP(hipMalloc(v, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(*v, 0, sizeof(double) * s->ldN, ctx->stream));
    }
    BQ_HIP(hipMalloc(&s->mL, (size_t)s->ldN));
    BQ_HIP(hipMalloc(&s->mU, (size_t)s->ldN));
    BQ_HIP(hipMemsetAsync(s->mL, 0, (size_t)s->ldN, ctx->stream));
    BQ_HIP(hipMemsetAsync(s->mU, 0, (size_t)s->ldN, ctx->stream));
    return eval_f(s, nullptr);  // f(x0), active_set.py:84
}

void bq_as_free(bq_solver *s) {
    as_ws *w = get_ws(s);
    if (!w) return;
    for (void *p : {(void *)w->idx, (void *)w->ints, (void *)w->cand, (void *)w->z, (void *)w->Qz, (void *)w->x_eval,
                    (void *)w->g_eval})
        if (p) hipFree(p);
    delete w;
    s->as_ws = nullptr;
}

// x / g as the callback of the last recorded iteration must see them (the body has already moved on)
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

    as_compact_kernel<<<1, 256, 0, st>>>(N, s->mL, s->mU, w->idx, w->ints);
    as_top_kernel<<<1, 1, 0, st>>>(s->sc, w->ints, s->stats);
    if (!s->host.done) {  // snapshot of the point this record describes
        as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->x, w->x_eval);
        as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->g, w->g_eval);
        s->started = true;
    }
    BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
    BQ_HIP(hipMemcpyAsync(&s->host, s->sc, sizeof(bq_scal), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    if (s->host.done) return BQ_OK;
    const int64_t nA = w->host_ints[0];

    as_make_z_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->z);
    BQ_TRY(bq_problem_apply(s->p, w->z, w->Qz, nullptr));
    int64_t np = 0;
    BQ_TRY(bq_chol_build_h(ws, s->p, w->idx, nA, nullptr, &np));
    as_gather_rhs_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(w->ints, w->idx, s->p->q, w->Qz, ws->rhs, np);
    BQ_TRY(bq_chol_factor(ws, np));
    BQ_TRY(bq_chol_solve(ws, np));
    as_candidate_kernel<<<1, 256, 0, st>>>(w->idx, w->ints, ws->rhs, w->z, N, s->lb, s->ub, w->cand);
    int info = 0;
    BQ_HIP(hipMemcpyAsync(&info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    if (info != 0) {
        // Q[A,A] is not positive definite: the reference's bare `except` switches to scipy's minres on the normal
        // equations (active_set.py:142-151).  Rebuild the (destroyed) restricted Hessian with both triangles, solve,
        // and redo the feasibility test on the minimum-residual candidate.
        if (nA > 8192) {
            bq_set_error("restricted Hessian Q[A,A] (|A| = %lld) is not positive definite at pivot %d and too large for "
                         "the single-workgroup MINRES fallback (limit 8192)", (long long)nA, info);
            return BQ_ERR_NOT_PD;
        }
        if (!ws->mr_vec) BQ_HIP(hipMalloc(&ws->mr_vec, sizeof(double) * 10 * ws->cap));
        BQ_TRY(bq_chol_build_h(ws, s->p, w->idx, nA, nullptr, &np, true));
        as_gather_rhs_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(w->ints, w->idx, s->p->q, w->Qz, ws->rhs, np);
        BQ_TRY(bq_minres_normal(ws, w->ints, ws->cap, ws->mr_vec, w->ints + 7));
        as_candidate_kernel<<<1, 256, 0, st>>>(w->idx, w->ints, ws->rhs, w->z, N, s->lb, s->ub, w->cand);
        BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
        BQ_HIP(hipStreamSynchronize(st));
        w->minres_calls += 1;
    }
    if (w->host_ints[2]) {
        as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, w->cand, s->x);
        BQ_TRY(eval_f(s, s->g));
        as_release_kernel<<<1, 256, 0, st>>>(N, s->g, s->mL, s->mU, s->sc, w->ints, s->stats);
    } else {
        as_step_kernel<<<1, 256, 0, st>>>(N, s->mL, s->mU, w->cand, s->lb, s->ub, s->x, s->sc);
        BQ_TRY(eval_f(s, nullptr));
        as_absorb_kernel<<<1, 256, 0, st>>>(N, s->mL, s->mU, s->x, s->lb, s->ub, s->sc, w->ints, s->stats);
    }
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
