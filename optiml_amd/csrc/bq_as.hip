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
//
// BQ_AS_CG (SURVEY 7 "hard parts": ActiveSet beyond the sizes a dense factor fits): the same outer logic, but the
// restricted system is solved by conjugate gradients on the masked panel operator v -> m.(Q (m.v)), m = indicator of
// A — no H, no factorisation, one panel product per inner iteration, so it runs on sharded, fp32-stored and streamed
// panels and on any number of ranks (the vectors and scalar recurrences are replicated, reductions fixed-order).  The
// iteration starts from the current point (delta = 0), which after a one-index change of A is already close.
#include <cmath>

#include "bq_chol.h"

#define ACT_TOL 1e-12

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
};

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
    // conjugate-gradient inner solver (BQ_AS_CG)
    double *dlt = nullptr, *r = nullptr, *pv = nullptr, *Qp = nullptr, *sol = nullptr;
    as_cg_scal *cg = nullptr;
    int *cg_flag_host = nullptr;   // pinned: {done, info}
    long long cg_iters = 0;
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
__global__ void as_cg_init_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                  const double *__restrict__ Qxt, const double *__restrict__ q, double *__restrict__ dlt,
                                  double *__restrict__ r, double *__restrict__ pv, double *part, int64_t nblk,
                                  as_cg_scal *cg, double rtol, long long max_iters) {
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
                sqx += __dmul_rn(a, a);
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
                cg->alpha = cg->rr / pAp;
            }
        }
    }
}

// delta += alpha p;  r -= alpha m.(Qp);  beta = r'r(new) / r'r(old); stop tests
__global__ void as_cg_update_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                    double *__restrict__ dlt, double *__restrict__ r, const double *__restrict__ pv,
                                    const double *__restrict__ Qp, double *part, int64_t nblk, as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double sh[4];
    const double alpha = cg->alpha;
    double s = 0.0;
    VEC_LOOP(i) {
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
            cg->beta = cg->rr > 0.0 ? rr / cg->rr : 0.0;
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

// sol[a] = x[idx[a]] + delta[idx[a]]: the restricted solution in the compact order as_candidate_kernel reads
__global__ void as_cg_gather_kernel(const int *__restrict__ ints, const int *__restrict__ idx,
                                    const double *__restrict__ x, const double *__restrict__ dlt,
                                    double *__restrict__ sol, int64_t N) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < ints[0] && a < N) sol[a] = x[idx[a]] + dlt[idx[a]];
}

static as_ws *get_ws(bq_solver *s) { return reinterpret_cast<as_ws *>(s->as_ws); }

static int eval_f(bq_solver *s, double *g_out) {
    // Qd = Q x ; f = 1/2 x'Qx + q'x -> sc->f ; optionally g = Qx + q
    BQ_TRY(bq_problem_apply(s->p, s->x, s->Qd, nullptr));
    return bq_vec_eval_f(s->p, s->x, s->Qd, g_out, &s->sc->f);
}

// the restricted solve of one outer iteration by conjugate gradients; leaves the candidate in w->cand and the
// feasibility flag in w->host_ints[2]
static int as_cg_solve(bq_solver *s, as_ws *w) {
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    const int64_t N = s->N, nblk = s->nblk;
    const int64_t nA = w->host_ints[0];
    const dim3 grid = vgrid(s->ldN);
    const long long cap = s->inner_max > 0 ? s->inner_max : 2 * (long long)nA + 50;
    as_make_xt_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, s->x, w->z);
    BQ_TRY(bq_problem_apply(s->p, w->z, w->Qz, nullptr));
    as_cg_init_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, w->Qz, s->p->q, w->dlt, w->r, w->pv, s->partials,
                                                     nblk, w->cg, s->inner_rtol, cap);
    // batches of inner iterations between looks at the done flag; after it is set the vector kernels return at once
    // and only the products of the rest of the batch are wasted, so batches start small
    int batch = 4;
    long long queued = 0;
    while (true) {
        BQ_HIP(hipMemcpyAsync(w->cg_flag_host, &w->cg->done, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        BQ_HIP(hipStreamSynchronize(st));
        if (w->cg_flag_host[0] || queued >= cap) break;
        for (int b = 0; b < batch; ++b) {
            BQ_TRY(bq_problem_apply(s->p, w->pv, w->Qp, nullptr));
            as_cg_pap_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, w->pv, w->Qp, s->partials, nblk, w->cg);
            as_cg_update_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, w->dlt, w->r, w->pv, w->Qp, s->partials,
                                                               nblk, w->cg);
            as_cg_dir_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, w->r, w->pv, w->cg);
        }
        queued += batch;
        if (batch < 32) batch *= 2;
    }
    as_cg_scal h;
    BQ_HIP(hipMemcpyAsync(&h, w->cg, sizeof(as_cg_scal), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    w->cg_iters += h.iters;
    if (h.info != 0 || !std::isfinite(h.rr)) {
        bq_set_error("conjugate gradients on the restricted Hessian Q[A,A] (|A| = %lld) met a direction of non-positive "
                     "curvature after %lld iterations: the system is not positive definite",
                     (long long)nA, (long long)h.iters);
        return BQ_ERR_NOT_PD;
    }
    as_cg_gather_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(w->ints, w->idx, s->x, w->dlt, w->sol, N);
    as_candidate_kernel<<<1, 256, 0, st>>>(w->idx, w->ints, w->sol, w->z, N, s->lb, s->ub, w->cand);
    BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 8, hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    return BQ_OK;
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
    if (s->as_cg) {
        for (double **v : {&w->dlt, &w->r, &w->pv, &w->Qp, &w->sol}) {
            BQ_HIP(hipMalloc(v, sizeof(double) * s->ldN));
            BQ_HIP(hipMemsetAsync(*v, 0, sizeof(double) * s->ldN, ctx->stream));
        }
        BQ_HIP(hipMalloc(&w->cg, sizeof(as_cg_scal)));
        BQ_HIP(hipMemsetAsync(w->cg, 0, sizeof(as_cg_scal), ctx->stream));
        BQ_HIP(hipHostMalloc(&w->cg_flag_host, 2 * sizeof(int)));
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
                    (void *)w->g_eval, (void *)w->dlt, (void *)w->r, (void *)w->pv, (void *)w->Qp, (void *)w->sol,
                    (void *)w->cg})
        if (p) hipFree(p);
    if (w->cg_flag_host) hipHostFree(w->cg_flag_host);
    delete w;
    s->as_ws = nullptr;
}

// x / g as the callback of the last recorded iteration must see them (the body has already moved on)
long long bq_as_inner_iters(bq_solver *s) {
    as_ws *w = get_ws(s);
    return w ? w->cg_iters : 0;
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

    if (s->as_cg) {
        BQ_TRY(as_cg_solve(s, w));
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
