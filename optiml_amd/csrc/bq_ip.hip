// Interior-point driver: primal-dual feasible IP for the box QP, one Cholesky solve per iteration.
//
// Restates optiml/opti/constrained/interior_point.py:180-267 on the device: start-up multipliers from the
// gradient (:180-186), bounds p / gap / mu (:192-195, :225), Newton system H dx = w with H = Q + diag (:227-235),
// multiplier increments (:237-239), the four step-to-boundary ratios scaled by 0.9995 (:242-263) and the update
// (:265-267).  One panel product Q x per iteration gives both f and x'Qx (the reference forms them with two
// products).  As in the drivers of PG/FW the step computed at the end of an iteration is applied at the start
// of the next evaluation, so x on the device is always the point the last iteration record was evaluated at.
#include <cmath>
#include <cstdlib>

#include "bq_chol.h"

#define VEC_LOOP(i)                                                             \
    const int64_t _base = (int64_t)blockIdx.x * BQ_VEC_TILE + threadIdx.x;      \
    _Pragma("unroll") for (int _j = 0; _j < BQ_VEC_ITEMS; ++_j)                 \
        for (int64_t i = _base + (int64_t)_j * BQ_VEC_BLOCK, _once = 1; _once; _once = 0)

static inline dim3 vgrid(int64_t ldN) { return dim3((unsigned)(ldN / BQ_VEC_TILE)); }

__device__ __forceinline__ double ip_wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double ip_wmin(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ double ip_bsum(double v, double *sh) {
    v = ip_wsum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double ip_bmin(double v, double *sh) {
    v = ip_wmin(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
    __syncthreads();
    return r;
}
__device__ __forceinline__ double ip_fsum(const double *part, int64_t nblk, double *sh) {
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < nblk; i += BQ_VEC_BLOCK) a += part[i];
    return ip_bsum(a, sh);
}
__device__ __forceinline__ double ip_fmin(const double *part, int64_t nblk, double *sh) {
    double a = INFINITY;
    for (int64_t i = threadIdx.x; i < nblk; i += BQ_VEC_BLOCK) a = fmin(a, part[i]);
    return ip_bmin(a, sh);
}

struct ipv {
    double *x, *g, *Qx, *q, *lb, *ub, *lp, *lm, *rhs, *hd, *dlp, *dlm, *dx;
};

// g = Qx + q at the start point; lp/lm = 1e-6 + [-g]_+ / [g]_+     (interior_point.py:180-186)
__global__ void ip_init_kernel(int64_t N, ipv V) {
    VEC_LOOP(i) {
        if (i < N) {
            const double g = V.Qx[i] + V.q[i];
            V.g[i] = g;
            double lp = 1e-6, lm = 1e-6;
            if (g >= 0.0)
                lm = lm + g;
            else
                lp = lp - g;
            V.lp[i] = lp;
            V.lm[i] = lm;
        }
    }
}

// apply the pending step:  x += t dx, lp += t dlp, lm += t dlm      (interior_point.py:265-267)
__global__ void ip_update_kernel(int64_t N, ipv V, const bq_scal *sc) {
    if (sc->done) return;
    const double t = sc->step;
    VEC_LOOP(i) {
        if (i < N) {
            V.x[i] = V.x[i] + __dmul_rn(t, V.dx[i]);
            V.lp[i] = V.lp[i] + __dmul_rn(t, V.dlp[i]);
            V.lm[i] = V.lm[i] + __dmul_rn(t, V.dlm[i]);
        }
    }
}

__global__ void ip_eval_kernel(int64_t N, ipv V, const bq_scal *sc, double *__restrict__ part, int64_t nblk) {
    if (sc->done) return;
    __shared__ double sh[4];
    double xr = 0.0, qx = 0.0, lu = 0.0, ll = 0.0;
    VEC_LOOP(i) {
        if (i < N) {
            const double x = V.x[i];
            xr += x * V.Qx[i];
            qx += V.q[i] * x;
            lu += V.lp[i] * V.ub[i];
            ll += V.lm[i] * V.lb[i];
        }
    }
    xr = ip_bsum(xr, sh);
    qx = ip_bsum(qx, sh);
    lu = ip_bsum(lu, sh);
    ll = ip_bsum(ll, sh);
    if (threadIdx.x == 0) {
        part[0 * nblk + blockIdx.x] = xr;
        part[1 * nblk + blockIdx.x] = qx;
        part[2 * nblk + blockIdx.x] = lu;
        part[3 * nblk + blockIdx.x] = ll;
    }
}

__global__ void ip_decide_kernel(bq_scal *sc, int64_t N, const double *__restrict__ part, int64_t nblk,
                                 bq_iter_stat *stats) {
    if (sc->done) return;
    __shared__ double sh[4];
    const double xr = ip_fsum(part + 0 * nblk, nblk, sh);
    const double qx = ip_fsum(part + 1 * nblk, nblk, sh);
    const double lu = ip_fsum(part + 2 * nblk, nblk, sh);
    const double ll = ip_fsum(part + 3 * nblk, nblk, sh);
    if (threadIdx.x == 0) {
        const double f = 0.5 * xr + qx;
        const double p = -lu + ll - 0.5 * xr;
        const double gap = (f - p) / fmax(fabs(f), 1.0);
        sc->f = f;
        sc->p = p;
        sc->gap = gap;
        sc->mu = (f - p) / (4.0 * (double)N * (double)N);
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            bq_iter_stat st;
            st.iter = sc->iter;
            st.f = f;
            st.r1 = p;
            st.r2 = gap;
            st.r3 = NAN;
            stats[row] = st;
        }
        if (!(f == f) || !(p == p)) {  // NaN: the reference's cho_factor would raise on non-finite input
            sc->status = BQ_ERR_NONFINITE;
            sc->done = 1;
        } else if (gap <= sc->eps) {
            sc->status = BQ_STATUS_OPTIMAL;
            sc->done = 1;
        } else if (sc->iter >= sc->max_iter) {
            sc->status = BQ_STATUS_STOPPED;
            sc->done = 1;
        }
    }
}

// hd = lp/(ub-x) + lm/(x-lb);  w = mu (ub+lb-2x)/((ub-x)(x-lb)) + lp - lm     (interior_point.py:227-231)
__global__ void ip_prep_kernel(int64_t N, ipv V, const bq_scal *sc, double *__restrict__ w_out) {
    if (sc->done) return;
    const double mu = sc->mu;
    VEC_LOOP(i) {
        if (i < N) {
            const double x = V.x[i], ub = V.ub[i], lb = V.lb[i], lp = V.lp[i], lm = V.lm[i];
            const double umx = ub - x, xml = x - lb;
            V.hd[i] = lp / umx + lm / xml;
            w_out[i] = mu * (ub + lb - 2.0 * x) / (umx * xml) + lp - lm;
        }
    }
}

// dlp, dlm and the partial minima of the four step-to-boundary ratios     (interior_point.py:237-259)
__global__ void ip_ratio_kernel(int64_t N, ipv V, const bq_scal *sc, const double *__restrict__ dxs,
                                double *__restrict__ part) {
    if (sc->done) return;
    __shared__ double sh[4];
    const double mu = sc->mu;
    double rmin = INFINITY;
    VEC_LOOP(i) {
        if (i < N) {
            const double x = V.x[i], ub = V.ub[i], lb = V.lb[i], lp = V.lp[i], lm = V.lm[i], dx = dxs[i];
            const double umx = ub - x, xml = x - lb;
            const double dlp = (mu + lp * dx) / umx - lp;
            const double dlm = (mu - lm * dx) / xml - lm;
            V.dx[i] = dx;
            V.dlp[i] = dlp;
            V.dlm[i] = dlm;
            if (dx < 0.0) rmin = fmin(rmin, (lb - x) / dx);
            if (dx > 0.0) rmin = fmin(rmin, umx / dx);
            if (dlp < 0.0) rmin = fmin(rmin, -lp / dlp);
            if (dlm < 0.0) rmin = fmin(rmin, -lm / dlm);
        }
    }
    rmin = ip_bmin(rmin, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = rmin;
}

__global__ void ip_step_kernel(bq_scal *sc, const double *__restrict__ part, int64_t nblk, const int *__restrict__ info,
                               bq_iter_stat *stats) {
    if (sc->done) return;
    __shared__ double sh[4];
    const double rmin = ip_fmin(part, nblk, sh);
    if (threadIdx.x == 0) {
        if (*info != 0) {
            sc->status = BQ_ERR_NOT_PD;
            sc->done = 1;
            sc->aux[0] = (double)*info;
            return;
        }
        const double t = rmin * 0.9995;
        sc->step = t;
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) stats[row].r3 = t;
        sc->iter += 1;
    }
}

__global__ void ip_copy_kernel(int64_t n, const double *__restrict__ src, double *__restrict__ dst, int64_t npad,
                               const bq_scal *sc) {
    if (sc->done) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) dst[i] = (i < n) ? src[i] : 0.0;
}

// SVR: the 2n x 2n Newton system [[P + D1, -P], [-P, P + D2]] [dx1; dx2] = [w1; w2] (P = K + 1, D = diag part hd)
// reduces to an n x n SPD system in u = dx1 - dx2.  Adding the two block rows gives D1 dx1 + D2 dx2 = w1 + w2, hence
//   (P + C) u = (D2 w1 - D1 w2) / (D1 + D2),   C = D1 D2 / (D1 + D2),
//   dx1 = (w1 + w2 + D2 u) / (D1 + D2),   dx2 = (w1 + w2 - D1 u) / (D1 + D2).
// Every division is by D1 + D2 (never by one of the two alone), so the elimination is symmetric in the two halves and
// does not lose digits when one of them is tiny.  An 8x cheaper factorisation than the reference's 2n x 2n one
// ("SVR 2n structure + IP"), same solution.
__global__ void ip_svr_reduce_kernel(int64_t n, const double *__restrict__ hd, const double *__restrict__ w,
                                     double *__restrict__ cdiag, double *__restrict__ t, const bq_scal *sc) {
    if (sc->done) return;
    VEC_LOOP(i) {
        double c = 0.0, tv = 0.0;
        if (i < n) {
            const double d1 = hd[i], d2 = hd[n + i], inv = 1.0 / (d1 + d2);
            c = d1 * d2 * inv;
            tv = (d2 * w[i] - d1 * w[n + i]) * inv;
        }
        cdiag[i] = c;   // both arrays are padded: the pad stays 0
        t[i] = tv;
    }
}

__global__ void ip_svr_rhs_kernel(int64_t n, int64_t np, const double *__restrict__ t, double *__restrict__ rhs,
                                  const bq_scal *sc) {
    if (sc->done) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) rhs[i] = (i < n) ? t[i] : 0.0;
}

// in place: w (2n) -> dx (2n)
__global__ void ip_svr_expand_kernel(int64_t n, const double *__restrict__ hd, const double *__restrict__ u,
                                     double *__restrict__ w, const bq_scal *sc) {
    if (sc->done) return;
    VEC_LOOP(i) {
        if (i < n) {
            const double d1 = hd[i], d2 = hd[n + i], inv = 1.0 / (d1 + d2);
            const double sum = w[i] + w[n + i], ui = u[i];
            w[i] = (sum + d2 * ui) * inv;
            w[n + i] = (sum - d1 * ui) * inv;
        }
    }
}

// Default ON: on the reference's SVR fixtures the reduced system reproduces the 2n x 2n trajectory (same iteration
// count, objective to 1e-14, alpha+ - alpha- to 1e-11).  hook ip_svr_reduced=0 selects the reference's own 2n x 2n
// factorisation instead.
bool bq_ip_svr_reduced() {
    return bq_hook_on("ip_svr_reduced");
}

static ipv ip_vecs(bq_solver *s) {
    ipv V;
    V.x = s->x;
    V.g = s->g;
    V.Qx = s->Qd;
    V.q = s->p->q;
    V.lb = s->lb;
    V.ub = s->ub;
    V.lp = s->lp;
    V.lm = s->lm;
    V.rhs = s->rhs;
    V.hd = s->hd;
    V.dlp = s->dlp;
    V.dlm = s->dlm;
    V.dx = s->d;
    return V;
}

int bq_ip_start(bq_solver *s) {
    hipStream_t st = s->p->ctx->stream;
    BQ_TRY(bq_problem_apply(s->p, s->x, s->Qd, nullptr));
    ip_init_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, ip_vecs(s));
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

int bq_ip_iterate(bq_solver *s) {
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    bq_chol_ws *ws = s->chol;
    const int *done = &s->sc->done;
    ipv V = ip_vecs(s);
    if (s->started) ip_update_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, V, s->sc);
    s->started = true;
    BQ_TRY(bq_problem_apply(s->p, s->x, s->Qd, done));
    ip_eval_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, V, s->sc, s->partials, s->nblk);
    ip_decide_kernel<<<1, BQ_VEC_BLOCK, 0, st>>>(s->sc, s->N, s->partials, s->nblk, s->stats);
    ip_prep_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, V, s->sc, s->rhs);
    // The factorisation below is host-enqueued and cannot early-exit on the device flag; after `done` it works on
    // stale data and its result is ignored by the (early-exiting) consumers.
    int64_t np = 0;
    if (s->p->structure == BQ_SVR && bq_ip_svr_reduced()) {
        bq_problem *p = s->p;
        const int64_t n = p->n;
        double *cdiag = s->dlp;   // scratch until ip_ratio_kernel rewrites dlp
        ip_svr_reduce_kernel<<<vgrid(p->ld), BQ_VEC_BLOCK, 0, st>>>(n, s->hd, s->rhs, cdiag, p->w, s->sc);
        BQ_TRY(bq_chol_build_h(ws, p, nullptr, n, cdiag, &np, false, BQ_H_KPLUS1));
        ip_svr_rhs_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(n, np, p->w, ws->rhs, s->sc);
        BQ_TRY(bq_chol_factor(ws, np));
        BQ_TRY(bq_chol_solve(ws, np));
        ip_svr_expand_kernel<<<vgrid(p->ld), BQ_VEC_BLOCK, 0, st>>>(n, s->hd, ws->rhs, s->rhs, s->sc);
        ip_ratio_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, V, s->sc, s->rhs, s->partials);
    } else {
        BQ_TRY(bq_chol_build_h(ws, s->p, nullptr, s->N, s->hd, &np));
        ip_copy_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(s->N, s->rhs, ws->rhs, np, s->sc);
        BQ_TRY(bq_chol_factor(ws, np));
        BQ_TRY(bq_chol_solve(ws, np));
        ip_ratio_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, V, s->sc, ws->rhs, s->partials);
    }
    ip_step_kernel<<<1, BQ_VEC_BLOCK, 0, st>>>(s->sc, s->partials, s->nblk, ws->info, s->stats);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

