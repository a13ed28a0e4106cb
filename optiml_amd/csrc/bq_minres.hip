// MINRES on the normal equations of the restricted system — the active-set fallback for a singular Q[A,A].
//
// The reference catches the failed Cholesky and calls scipy.sparse.linalg.minres(Q_AA Q_AA^T, -(Q_AA^T q_A)) with its
// defaults (optiml/opti/constrained/active_set.py:142-151; scipy 1.15.3 here, unpinned in the reference): rtol 1e-5,
// maxiter 5 n, x0 = 0, no shift, no preconditioner.  This is a restatement of that Paige-Saunders iteration (Lanczos
// three-term recurrence + Givens rotations, the same stopping tests in the same order) as ONE persistent workgroup:
// the iteration is a strict chain of small matrix-vector products and scalar recurrences, so a single 1024-thread
// workgroup with the restricted Hessian resident in L2 avoids ~10 launches and 3 host round-trips per iteration.
// The operator is applied as H (H v) with the symmetric H = Q[A,A] instead of forming H H^T.
// Parity can only be loose: the singular case is decided by rounding in the reference as well.
#include "bq_chol.h"

constexpr int MT = 1024;   // threads of the persistent workgroup

__device__ __forceinline__ double mr_wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// every thread receives the block-wide sum (fixed order)
__device__ __forceinline__ double mr_bsum(double v, double *sh) {
    v = mr_wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < MT / 64; ++w) r += sh[w];
    return r;
}

__device__ __forceinline__ double mr_dot(const double *a, const double *b, int n, double *sh) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += MT) s = fma(a[i], b[i], s);
    return mr_bsum(s, sh);
}

// y = H x for the symmetric n x n matrix H (row pitch ld): one wave per row, lanes along the columns
__device__ __forceinline__ void mr_matvec(const double *__restrict__ H, int64_t ld, int n, const double *x, double *y) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = wv; r < n; r += MT / 64) {
        const double *row = H + (int64_t)r * ld;
        double s = 0.0;
        for (int c = lane; c < n; c += 64) s = fma(row[c], x[c], s);
        s = mr_wsum(s);
        if (lane == 0) y[r] = s;
    }
    __syncthreads();
}

// vec: 10 scratch vectors of length np.  rhs holds q' = -(q_A + ...) on entry and the solution on exit.
__global__ __launch_bounds__(MT) void minres_normal_kernel(const double *__restrict__ H, int64_t ld, const int *nA_ptr,
                                                           double *__restrict__ rhs, double *__restrict__ vec,
                                                           int64_t np, double rtol, int *__restrict__ iters) {
    __shared__ double sh[MT / 64];
    const int n = *nA_ptr;
    const int tid = threadIdx.x;
    // Ra/Rb/Rc rotate through the roles (r1, r2 = y, next Lanczos vector) of the three-term recurrence
    double *b = vec, *x = vec + np, *Ra = vec + 2 * np, *Rb = vec + 3 * np, *Rc = vec + 4 * np, *v = vec + 5 * np,
           *w = vec + 6 * np, *w1 = vec + 7 * np, *w2 = vec + 8 * np, *tmp = vec + 9 * np;
    const double eps = 2.220446049250313e-16;
    const int maxiter = 5 * n;

    // b = H q'   (= -(Q_AA^T q_A));  x = 0;  r1 = b;  y = r1
    mr_matvec(H, ld, n, rhs, b);
    for (int i = tid; i < n; i += MT) {
        x[i] = 0.0;
        Ra[i] = b[i];
        w[i] = 0.0;
        w2[i] = 0.0;
    }
    __syncthreads();
    double beta1 = mr_dot(Ra, Ra, n, sh);
    int itn = 0;
    if (beta1 > 0.0) {   // beta1 == 0: the exact solution is x0 = 0 (also covers b == 0)
        beta1 = sqrt(beta1);
        double oldb = 0.0, beta = beta1, dbar = 0.0, epsln = 0.0, phibar = beta1, rhs1 = beta1, rhs2 = 0.0;
        double tnorm2 = 0.0, gmax = 0.0, gmin = 1.7976931348623157e308, cs = -1.0, sn = 0.0;
        double *r1p = Ra, *r2p = Ra, *y = Ra;   // r1 = b, r2 = r1, y = psolve(r1) = r1
        int istop = 0;
        while (itn < maxiter) {
            itn += 1;
            const double s = 1.0 / beta;
            for (int i = tid; i < n; i += MT) v[i] = s * y[i];
            __syncthreads();
            mr_matvec(H, ld, n, v, tmp);     // operator = H H^T = H H
            double *ynew = (Ra != r1p && Ra != r2p) ? Ra : ((Rb != r1p && Rb != r2p) ? Rb : Rc);
            mr_matvec(H, ld, n, tmp, ynew);
            if (itn >= 2) {
                const double c = beta / oldb;
                for (int i = tid; i < n; i += MT) ynew[i] = ynew[i] - c * r1p[i];
                __syncthreads();
            }
            const double alfa = mr_dot(v, ynew, n, sh);
            {
                const double c = alfa / beta;
                for (int i = tid; i < n; i += MT) ynew[i] = ynew[i] - c * r2p[i];
                __syncthreads();
            }
            // r1 = r2; r2 = y; y = psolve(r2) = r2
            r1p = r2p;
            r2p = ynew;
            y = ynew;
            oldb = beta;
            beta = mr_dot(r2p, y, n, sh);
            if (beta < 0.0) {
                istop = 7;
                break;
            }
            beta = sqrt(beta);
            tnorm2 += alfa * alfa + oldb * oldb + beta * beta;
            if (itn == 1 && beta / beta1 <= 10.0 * eps) istop = -1;
            const double oldeps = epsln;
            const double delta = cs * dbar + sn * alfa;
            const double gbar = sn * dbar - cs * alfa;
            epsln = sn * beta;
            dbar = -cs * beta;
            const double root = hypot(gbar, dbar);
            double gamma = hypot(gbar, beta);
            gamma = fmax(gamma, eps);
            cs = gbar / gamma;
            sn = beta / gamma;
            const double phi = cs * phibar;
            phibar = sn * phibar;
            const double denom = 1.0 / gamma;
            // w1 = w2; w2 = w; w = (v - oldeps*w1 - delta*w2) * denom; x += phi*w
            double *wold = w1;
            w1 = w2;
            w2 = w;
            w = wold;
            double xn = 0.0;
            for (int i = tid; i < n; i += MT) {
                const double wn = (v[i] - oldeps * w1[i] - delta * w2[i]) * denom;
                w[i] = wn;
                const double xi = x[i] + phi * wn;
                x[i] = xi;
                xn = fma(xi, xi, xn);
            }
            const double ynorm = sqrt(mr_bsum(xn, sh));
            gmax = fmax(gmax, gamma);
            gmin = fmin(gmin, gamma);
            const double z = rhs1 / gamma;
            rhs1 = rhs2 - delta * z;
            rhs2 = -epsln * z;
            const double Anorm = sqrt(tnorm2);
            const double epsx = Anorm * ynorm * eps;
            const double rnorm = phibar;
            const double test1 = (ynorm == 0.0 || Anorm == 0.0) ? INFINITY : rnorm / (Anorm * ynorm);
            const double test2 = (Anorm == 0.0) ? INFINITY : root / Anorm;
            const double Acond = gmax / gmin;
            if (istop == 0) {
                const double t1 = 1.0 + test1, t2 = 1.0 + test2;
                if (t2 <= 1.0) istop = 2;
                if (t1 <= 1.0) istop = 1;
                if (itn >= maxiter) istop = 6;
                if (Acond >= 0.1 / eps) istop = 4;
                if (epsx >= beta1) istop = 3;
                if (test2 <= rtol) istop = 2;
                if (test1 <= rtol) istop = 1;
            }
            if (istop != 0) break;
        }
    }
    __syncthreads();
    for (int i = tid; i < (int)np; i += MT) rhs[i] = (i < n) ? x[i] : 0.0;
    if (tid == 0 && iters) *iters = itn;
}

int bq_minres_normal(bq_chol_ws *ws, const int *nA_dev, int64_t np, double *vec, int *iters_dev) {
    minres_normal_kernel<<<1, MT, 0, ws->ctx->stream>>>(ws->H, ws->ldh, nA_dev, ws->rhs, vec, np, 1e-5, iters_dev);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
