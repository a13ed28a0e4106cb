// Unconstrained minimiser of the quadratic: Quadratic.x_star() / f_star() (optiml/opti/_base.py:259-273).
//
//   x* = cho_solve(cho_factor(Q), -q)                 when Q is positive definite          (:263-264)
//   x* = scipy.sparse.linalg.minres(Q, -q)[0]         when the factorisation raises        (:265-269)
//
// The Hessian is assembled on the device from the resident panel (dense, SVC- or SVR-structured: bq_chol_build_h) and
// factorised by the blocked MFMA Cholesky the interior-point solver uses; LAPACK's "pivot <= 0" test decides the branch.
// The MINRES branch restates scipy's Paige-Saunders iteration with its defaults (rtol 1e-5, maxiter 5 N, x0 = 0, no shift,
// no preconditioner; scipy 1.15.3 here, unpinned in the reference): the O(N^2) product per iteration is the panel product
// of the problem, the three-term recurrences run in two single-workgroup kernels with fixed-order reductions, and the
// scalar rotations + stopping tests (same tests, same order) on the host between them.  Not a hot path: two stream
// synchronisations per iteration.
#include <cfloat>
#include <cmath>

#include "bq_chol.h"

namespace {
constexpr int XT = 1024;

__device__ __forceinline__ double xs_bsum(double v, double *sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < XT / 64; ++w) r += sh[w];
    return r;
}

// rhs = -q ; r1 = r2 = rhs ; x = w = w2 = 0 ; out[0] = rhs'rhs
__global__ __launch_bounds__(XT) void xs_init_kernel(int64_t N, const double *__restrict__ q, double *r1, double *r2, double *x,
                                                     double *w, double *w2, double *out) {
    __shared__ double sh[XT / 64];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += XT) {
        const double b = -q[i];
        r1[i] = b;
        r2[i] = b;
        x[i] = 0.0;
        w[i] = 0.0;
        w2[i] = 0.0;
        s = fma(b, b, s);
    }
    s = xs_bsum(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

// v = (1 / beta) y
__global__ __launch_bounds__(256) void xs_scale_kernel(int64_t N, double s, const double *__restrict__ y, double *__restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) v[i] = __dmul_rn(s, y[i]);
}

// y -= c1 r1 (itn >= 2);  alfa = v'y;  y -= (alfa / beta) r2;  out = {alfa, y'y}
__global__ __launch_bounds__(XT) void xs_lanczos_kernel(int64_t N, int use_r1, double c1, double beta, const double *__restrict__ v,
                                                        const double *__restrict__ r1, const double *__restrict__ r2,
                                                        double *__restrict__ y, double *out) {
    __shared__ double sh[XT / 64];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += XT) {
        double yi = y[i];
        if (use_r1) yi = yi - __dmul_rn(c1, r1[i]);
        y[i] = yi;
        s = fma(v[i], yi, s);
    }
    const double alfa = xs_bsum(s, sh);
    const double c2 = alfa / beta;
    s = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += XT) {
        const double yi = y[i] - __dmul_rn(c2, r2[i]);
        y[i] = yi;
        s = fma(yi, yi, s);
    }
    s = xs_bsum(s, sh);
    if (threadIdx.x == 0) {
        out[0] = alfa;
        out[1] = s;
    }
}

// wn = (v - oldeps w_{k-2} - delta w_{k-1}) denom ; x += phi wn ; out = x'x
__global__ __launch_bounds__(XT) void xs_update_kernel(int64_t N, double oldeps, double delta, double denom, double phi,
                                                       const double *__restrict__ v, const double *__restrict__ wkm2,
                                                       const double *__restrict__ wkm1, double *__restrict__ wn_out,
                                                       double *__restrict__ x, double *out) {
    __shared__ double sh[XT / 64];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += XT) {
        const double wn = __dmul_rn((v[i] - __dmul_rn(oldeps, wkm2[i])) - __dmul_rn(delta, wkm1[i]), denom);
        wn_out[i] = wn;
        const double xi = x[i] + __dmul_rn(phi, wn);
        x[i] = xi;
        s = fma(xi, xi, s);
    }
    s = xs_bsum(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ void xs_neg_kernel(int64_t N, int64_t np, const double *__restrict__ q, double *__restrict__ rhs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < np) rhs[i] = i < N ? -q[i] : 0.0;
}

struct dev_vecs {
    double *buf = nullptr;
    ~dev_vecs() {
        if (buf) hipFree(buf);
    }
};

int minres_solve(bq_problem *p, double *x_host, int64_t *iters) {
    bq_ctx *c = p->ctx;
    hipStream_t st = c->stream;
    const int64_t N = p->N, ld = p->ldN;
    dev_vecs mem;
    BQ_HIP(hipMalloc(&mem.buf, sizeof(double) * (8 * ld + 8)));
    BQ_HIP(hipMemsetAsync(mem.buf, 0, sizeof(double) * (8 * ld + 8), st));
    double *x = mem.buf, *r1 = x + ld, *r2 = r1 + ld, *y = r2 + ld, *v = y + ld, *w = v + ld, *w1 = w + ld, *w2 = w1 + ld,
           *sc = w2 + ld;
    double h[2] = {0.0, 0.0};
    auto fetch = [&](int k) {
        hipError_t e = hipMemcpyAsync(h, sc, sizeof(double) * k, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        return e;
    };
    xs_init_kernel<<<1, XT, 0, st>>>(N, p->q, r1, r2, x, w, w2, sc);
    BQ_HIP(fetch(1));
    const double eps = DBL_EPSILON, rtol = 1e-5;
    const int64_t maxiter = 5 * N;
    int64_t itn = 0;
    double beta1 = h[0];
    if (beta1 > 0.0) {   // beta1 == 0 (q == 0): x = 0 is returned as is
        beta1 = std::sqrt(beta1);
        double oldb = 0.0, beta = beta1, dbar = 0.0, epsln = 0.0, phibar = beta1, rhs1 = beta1, rhs2 = 0.0, tnorm2 = 0.0,
               gmax = 0.0, gmin = DBL_MAX, cs = -1.0, sn = 0.0;
        int istop = 0;
        const unsigned vg = (unsigned)((N + 255) / 256);
        while (itn < maxiter) {
            itn += 1;
            xs_scale_kernel<<<vg, 256, 0, st>>>(N, 1.0 / beta, r2, v);      // y (= r2, no preconditioner) scaled
            BQ_TRY(bq_problem_apply(p, v, y, nullptr));                      // y = Q v
            xs_lanczos_kernel<<<1, XT, 0, st>>>(N, itn >= 2 ? 1 : 0, itn >= 2 ? beta / oldb : 0.0, beta, v, r1, r2, y, sc);
            BQ_HIP(fetch(2));
            const double alfa = h[0];
            {   // r1 = r2; r2 = y; the buffer of the old r1 receives the next product
                double *t = r1;
                r1 = r2;
                r2 = y;
                y = t;
            }
            oldb = beta;
            if (h[1] < 0.0 || !std::isfinite(h[1])) {
                bq_set_error("minres: non-finite Lanczos vector");
                return BQ_ERR_NONFINITE;
            }
            beta = std::sqrt(h[1]);
            tnorm2 += alfa * alfa + oldb * oldb + beta * beta;
            if (itn == 1 && beta / beta1 <= 10 * eps) istop = -1;
            const double oldeps = epsln;
            const double delta = cs * dbar + sn * alfa;
            const double gbar = sn * dbar - cs * alfa;
            epsln = sn * beta;
            dbar = -cs * beta;
            const double root = std::hypot(gbar, dbar);
            double gamma = std::hypot(gbar, beta);
            gamma = std::max(gamma, eps);
            cs = gbar / gamma;
            sn = beta / gamma;
            const double phi = cs * phibar;
            phibar = sn * phibar;
            const double denom = 1.0 / gamma;
            xs_update_kernel<<<1, XT, 0, st>>>(N, oldeps, delta, denom, phi, v, w2, w, w1, x, sc);   // w1: the spare buffer
            {   // scipy: w1 = w2; w2 = w; w = the new vector — the old w2 becomes the spare
                double *t = w2;
                w2 = w;
                w = w1;
                w1 = t;
            }
            BQ_HIP(fetch(1));
            gmax = std::max(gmax, gamma);
            gmin = std::min(gmin, gamma);
            const double z = rhs1 / gamma;
            rhs1 = rhs2 - delta * z;
            rhs2 = -epsln * z;
            const double Anorm = std::sqrt(tnorm2), ynorm = std::sqrt(h[0]);
            const double epsx = Anorm * ynorm * eps;
            const double rnorm = phibar;
            const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? INFINITY : rnorm / (Anorm * ynorm);
            const double test2 = (Anorm == 0.0) ? INFINITY : root / Anorm;
            const double Acond = gmax / gmin;
            if (istop == 0) {
                if (1.0 + test2 <= 1.0) istop = 2;
                if (1.0 + test1 <= 1.0) istop = 1;
                if (itn >= maxiter) istop = 6;
                if (Acond >= 0.1 / eps) istop = 4;
                if (epsx >= beta1) istop = 3;
                if (test2 <= rtol) istop = 2;
                if (test1 <= rtol) istop = 1;
            }
            if (istop != 0) break;
        }
    }
    BQ_HIP(hipMemcpyAsync(x_host, x, sizeof(double) * N, hipMemcpyDeviceToHost, st));
    BQ_HIP(hipStreamSynchronize(st));
    if (iters) *iters = itn;
    return BQ_OK;
}
}  // namespace

extern "C" int bq_problem_x_star(bq_problem *p, double *x_out, int *method, int64_t *minres_iters) {
    BQ_ARG(p && x_out, "NULL argument");
    bq_ctx *c = p->ctx;
    BQ_ARG(c->world == 1, "x_star factorises the whole Hessian: single-rank contexts only");
    BQ_ARG(!p->streamed, "x_star assembles the Hessian from the resident panel: not available in the streamed mode");
    BQ_HIP(hipSetDevice(c->device));
    if (method) *method = 0;
    if (minres_iters) *minres_iters = 0;
    bq_chol_ws *ws = nullptr;
    BQ_TRY(bq_chol_ws_create(c, p->N, &ws));
    int64_t np = 0;
    int rc = bq_chol_build_h(ws, p, nullptr, p->N, nullptr, &np);
    int info = 0;
    hipError_t e = hipSuccess;
    if (rc == BQ_OK) {
        xs_neg_kernel<<<(unsigned)((np + 255) / 256), 256, 0, c->stream>>>(p->N, np, p->q, ws->rhs);
        rc = bq_chol_factor(ws, np);
    }
    if (rc == BQ_OK) e = hipMemcpyAsync(&info, ws->info, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (rc == BQ_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (rc == BQ_OK && e == hipSuccess && info == 0) {
        rc = bq_chol_solve(ws, np);
        if (rc == BQ_OK) e = hipMemcpyAsync(x_out, ws->rhs, sizeof(double) * p->N, hipMemcpyDeviceToHost, c->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    bq_chol_ws_destroy(ws);
    BQ_TRY(rc);
    BQ_HIP(e);
    if (info == 0) return BQ_OK;
    // not positive definite (scipy: LinAlgError): the minimum-residual solution, optiml/opti/_base.py:265-269
    if (method) *method = 1;
    return minres_solve(p, x_out, minres_iters);
}
