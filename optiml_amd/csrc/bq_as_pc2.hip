// The implicit ORDER-2 remainder of ActiveSetCG's preconditioner (round 5).
//
// The RBF kernel e^{-g|x-x'|^2} = e e' exp(2g x.x') has the order-2 Taylor term  B_ij = c_i c_j (x_i.x_j)^2,  c = sqrt(2) g y e.
// Its d (d + 1) / 2 directions form a FLAT bulk of eigenvalues ~ 0.27 n / (d (d + 1) / 2) (2.05 at BASELINE config 5) on top of the
// ~0.8 the diagonal leaves: no explicit low-rank model captures it (tools/pc_nystrom_study.py), 33 154 explicit features do
// (tools/pc_order2_study.py: 53 -> 13 iterations at n = 100 000) but cost 33 GB and a 33 154^2 inverse.  B needs no features to be
// APPLIED, though:   (B v)_i = c_i x_i' M x_i,   M = sum_j (c_j v_j) x_j x_j'   (d x d)
// — two n x d x d products on the fp64 matrix cores (~0.5 ms each at n = 250 000, d = 256, against a 20 ms panel product).
//
// The explicit model P1 = D + Phi Phi' (bq_as_pc.hip) already holds the part of B whose eigenvalues grow like n |class mean|^2 — the
// exact projection Phi_top of B's features onto the 2d class-mean directions — so what is left,  R = B - Phi_top Phi_top'  (positive
// semi-definite: a projection was removed), has its spectrum in the bulk: P1^-1 (P1 + R) lies in [1, 1 + lambda], lambda ~ 4, and ONE
// application of R inside a degree-1 Chebyshev polynomial inverts P = P1 + R to 10 %:
//      P^-1 r  ~  alpha y - beta P1^-1 (R y),   y = P1^-1 r            (a fixed symmetric positive definite operator: plain PCG)
// tools/pc_projected_cpu_study.py (d = 64, n = 20 000): 23 -> 18 conjugate-gradient iterations, the exact order-2 model needs 17.
//
// Everything is a fixed-order sum: the same bits on every rank (the operator is replicated, like the rest of the preconditioner).
#include "bq_as.h"
#include "bq_mfma_tile.h"

struct as_pc2 {
    int64_t n = 0, d = 0, dp = 0, ld = 0;   // samples, features, features padded to 128, vector pitch (s->ldN)
    int64_t S = 0, kc = 0, rows = 0;        // split-K of M: S slices of kc samples, rows = S * kc >= ld
    double *Xp = nullptr;     // rows x dp, row-major, zero padded: the k-major image of X with k = sample
    double *Xt = nullptr;     // dp x ld: the k-major image with k = feature
    double *W = nullptr;      // rows x dp: diag(w) Xp, w = c o y
    double *Mpart = nullptr;  // S x dp x dp
    double *M = nullptr;      // dp x dp
    double *ypart = nullptr;  // (dp / 128) x ld: x_i' M x_i, one partial per column tile
    double *c = nullptr;      // ld: sqrt(2) g y e
    double alpha = 1.0, beta = 0.0, lambda = -1.0;   // the polynomial; lambda < 0: not estimated yet
};

// Xp (rows x dp) and Xt (dp x ld) from X (n x d)
__global__ void pc2_pad_kernel(const double *__restrict__ X, int64_t n, int64_t d, int64_t dp, int64_t rows, int64_t ld,
                               double *__restrict__ Xp, double *__restrict__ Xt) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = r0 + j, k = k0 + tx;
        const double v = (r < n && k < d) ? X[r * d + k] : 0.0;
        tile[j][tx] = v;
        if (r < rows && k < dp) Xp[r * dp + k] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t k = k0 + j, r = r0 + tx;
        if (k < dp && r < ld) Xt[k * ld + r] = tile[tx][j];
    }
}

// W[i][:] = (c_i y_i) Xp[i][:]  (y vanishes outside the free set: so does W)
__global__ __launch_bounds__(256) void pc2_scale_kernel(int64_t ld, int64_t dp, const double *__restrict__ Xp, const double *__restrict__ c,
                                                        const double *__restrict__ y, double *__restrict__ W, const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // a wave per sample row
    if (i >= ld) return;
    const double w = c[i] * y[i];
    const double *src = Xp + i * dp;
    double *dst = W + i * dp;
    for (int64_t k = 2 * (threadIdx.x & 63); k < dp; k += 128) {
        bq_d2 v = *reinterpret_cast<const bq_d2 *>(src + k);
        v.x *= w;
        v.y *= w;
        *reinterpret_cast<bq_d2 *>(dst + k) = v;
    }
}

// Mpart[s] (tile ta, tb) = Xp[slice s]' W[slice s], tb <= ta only (M is symmetric: the reduction mirrors): blockIdx.x = the tile's
// index in the lower triangle, blockIdx.y = slice
__global__ __launch_bounds__(256, 2) void pc2_moment_kernel(int64_t dp, int64_t kc, const double *__restrict__ Xp, const double *__restrict__ W,
                                                            double *__restrict__ Mpart, const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    int64_t ta = 0;
    while ((ta + 1) * (ta + 2) / 2 <= (int64_t)blockIdx.x) ++ta;
    const int64_t tb = (int64_t)blockIdx.x - ta * (ta + 1) / 2;
    const int64_t k0 = (int64_t)blockIdx.y * kc;
    bq_d4 acc[4][4];
    bq_tile_zero(acc);
    bq_mfma_tile_128(Xp + k0 * dp, dp, ta * BQ_GT, W + k0 * dp, dp, tb * BQ_GT, kc, sm, acc);
    bq_tile_store(acc, Mpart + ((int64_t)blockIdx.y * dp + ta * BQ_GT) * dp + tb * BQ_GT, dp);
}

// M = the slices added in slice order
__global__ __launch_bounds__(256) void pc2_moment_reduce_kernel(int64_t dp, int64_t S, const double *__restrict__ Mpart, double *__restrict__ M,
                                                                const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= dp * dp) return;
    const int64_t ra = o / dp, rb = o % dp;
    const int64_t e = (ra / BQ_GT >= rb / BQ_GT) ? o : rb * dp + ra;   // tiles above the diagonal: the mirrored element
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four interleaved chains (loads in flight), combined in a fixed order
    int64_t s = 0;
    for (; s + 4 <= S; s += 4) {
        a0 += Mpart[(s + 0) * dp * dp + e];
        a1 += Mpart[(s + 1) * dp * dp + e];
        a2 += Mpart[(s + 2) * dp * dp + e];
        a3 += Mpart[(s + 3) * dp * dp + e];
    }
    for (; s < S; ++s) a0 += Mpart[s * dp * dp + e];
    M[o] = (a0 + a1) + (a2 + a3);
}

// ypart[ct][i] = sum over the 128 columns b of column tile ct of (X M)[i][b] X[i][b]: blockIdx.x = sample tile, blockIdx.y = ct
__global__ __launch_bounds__(256, 2) void pc2_bilinear_kernel(int64_t dp, int64_t ld, const double *__restrict__ Xt, const double *__restrict__ Xp,
                                                              const double *__restrict__ M, double *__restrict__ ypart, const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    __shared__ double half[2][BQ_GT];
    const int64_t i0 = (int64_t)blockIdx.x * BQ_GT, b0 = (int64_t)blockIdx.y * BQ_GT;
    bq_d4 acc[4][4];
    bq_tile_zero(acc);
    // acc[row = sample][col = b] = sum_k Xt[k][i0 + row] M[k][b0 + col]   (M is symmetric: its rows are its k-major image)
    bq_mfma_tile_128(Xt, ld, i0, M, dp, b0, dp, sm, acc);
    const int lane = threadIdx.x & 63, wc = (threadIdx.x >> 6) & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = bq_acc_row(i, v);
            const double *xr = Xp + (i0 + row) * dp + b0;
            double p = 0.0;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const bq_d2 x2 = *reinterpret_cast<const bq_d2 *>(xr + bq_acc_col(2 * jp));
                p = fma(acc[i][2 * jp][v], x2.x, p);
                p = fma(acc[i][2 * jp + 1][v], x2.y, p);
            }
            // the sixteen lanes that share this row (same lane >> 4) hold its other columns of this wave's half
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            p += __shfl_xor(p, 4, 64);
            p += __shfl_xor(p, 8, 64);
            if ((lane & 15) == 0) half[wc][row] = p;
        }
    __syncthreads();
    if (threadIdx.x < BQ_GT) ypart[(int64_t)blockIdx.y * ld + i0 + threadIdx.x] = half[0][threadIdx.x] + half[1][threadIdx.x];
}

// c_i = sqrt(2) g y_i e^{-g |x_i|^2} and the diagonal of B, c_i^2 |x_i|^4
__global__ void pc2_coef_kernel(int64_t n, int64_t d, int64_t ld, const double *__restrict__ X, const double *__restrict__ sgn, double gamma,
                                double *__restrict__ c, double *__restrict__ bdiag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ld) return;
    double ci = 0.0, bd = 0.0;
    if (i < n) {
        const double *x = X + i * d;
        double sq = 0.0;
        for (int64_t k = 0; k < d; ++k) sq = fma(x[k], x[k], sq);
        ci = sqrt(2.0) * gamma * (sgn ? sgn[i] : 1.0) * exp(-gamma * sq);
        bd = ci * ci * sq * sq;
    }
    c[i] = ci;
    bdiag[i] = bd;
}

void as_pc2_free(as_pc2 *r) {
    if (!r) return;
    for (void *p : {(void *)r->Xp, (void *)r->Xt, (void *)r->W, (void *)r->Mpart, (void *)r->M, (void *)r->ypart, (void *)r->c})
        if (p) hipFree(p);
    delete r;
}

// buffers and images; bdiag_out (device, ldN): the diagonal of B, which the caller takes out of D
int as_pc2_create(bq_solver *s, double *bdiag_out, as_pc2 **out) {
    *out = nullptr;
    bq_problem *p = s->p;
    hipStream_t st = p->ctx->stream;
    as_pc2 *r = new as_pc2();
    r->n = p->n;
    r->d = p->d;
    r->dp = bq_round_up(p->d, BQ_GT);
    r->ld = s->ldN;
    const int64_t T = r->dp / BQ_GT;
    // slices of the moment matrix: about two workgroups per CU in all, at least 256 samples each
    int64_t S = std::max<int64_t>(1, 512 / (T * (T + 1) / 2));
    S = std::min<int64_t>(S, std::max<int64_t>(1, r->ld / 256));
    r->kc = bq_round_up((r->ld + S - 1) / S, BQ_GK);
    r->S = (r->ld + r->kc - 1) / r->kc;
    r->rows = r->S * r->kc;
    hipError_t e = hipMalloc(&r->Xp, sizeof(double) * r->rows * r->dp);
    if (e == hipSuccess) e = hipMalloc(&r->Xt, sizeof(double) * r->dp * r->ld);
    if (e == hipSuccess) e = hipMalloc(&r->W, sizeof(double) * r->rows * r->dp);
    if (e == hipSuccess) e = hipMalloc(&r->Mpart, sizeof(double) * r->S * r->dp * r->dp);
    if (e == hipSuccess) e = hipMalloc(&r->M, sizeof(double) * r->dp * r->dp);
    if (e == hipSuccess) e = hipMalloc(&r->ypart, sizeof(double) * T * r->ld);
    if (e == hipSuccess) e = hipMalloc(&r->c, sizeof(double) * r->ld);
    if (e == hipSuccess) e = hipMemsetAsync(r->Xp, 0, sizeof(double) * r->rows * r->dp, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->W, 0, sizeof(double) * r->rows * r->dp, st);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        as_pc2_free(r);
        return BQ_OK;   // no room: the explicit model alone (the caller sees *out == nullptr)
    }
    pc2_pad_kernel<<<dim3((unsigned)((r->rows + 31) / 32), (unsigned)((r->dp + 31) / 32)), 256, 0, st>>>(p->X, p->n, p->d, r->dp, r->rows, r->ld,
                                                                                                      r->Xp, r->Xt);
    pc2_coef_kernel<<<(unsigned)(r->ld / 256), 256, 0, st>>>(p->n, p->d, r->ld, p->X, p->sgn, p->gamma, r->c, bdiag_out);
    if (hipGetLastError() != hipSuccess) {   // a launch that cannot be made: the explicit model alone
        as_pc2_free(r);
        return BQ_OK;
    }
    *out = r;
    return BQ_OK;
}

// the B part of v = R y:  ypart[ct][i] = the column tile's share of x_i' M x_i, M = sum_j (c_j y_j) x_j x_j'  (y: zero outside the free set);
// bq_as_pc.hip (as_pc_r_apply) adds the tiles, scales by c_i and takes Phi_top Phi_top' y off
int as_pc2_bpart(bq_solver *s, as_pc2 *r, const double *y, const as_cg_scal *cg) {
    hipStream_t st = s->p->ctx->stream;
    const int64_t T = r->dp / BQ_GT;
    pc2_scale_kernel<<<(unsigned)((r->ld + 3) / 4), 256, 0, st>>>(r->ld, r->dp, r->Xp, r->c, y, r->W, cg);
    pc2_moment_kernel<<<dim3((unsigned)(T * (T + 1) / 2), (unsigned)r->S), 256, 0, st>>>(r->dp, r->kc, r->Xp, r->W, r->Mpart, cg);
    pc2_moment_reduce_kernel<<<(unsigned)((r->dp * r->dp + 255) / 256), 256, 0, st>>>(r->dp, r->S, r->Mpart, r->M, cg);
    pc2_bilinear_kernel<<<dim3((unsigned)(r->ld / BQ_GT), (unsigned)T), 256, 0, st>>>(r->dp, r->ld, r->Xt, r->Xp, r->M, r->ypart, cg);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
const double *as_pc2_ypart(const as_pc2 *r, int *tiles) {
    *tiles = (int)(r->dp / BQ_GT);
    return r->ypart;
}
const double *as_pc2_c(const as_pc2 *r) { return r->c; }

double as_pc2_lambda(const as_pc2 *r) { return r->lambda; }

// the degree-1 Chebyshev polynomial for the spectrum [1, 1 + lambda] of P1^-1 (P1 + R):  P^-1 r ~ alpha y - beta P1^-1 (R y)
void as_pc2_set_lambda(as_pc2 *r, double lambda) {
    r->lambda = lambda;
    const double a = 1.0, b = 1.0 + lambda;
    const double theta = 0.5 * (a + b), delta = 0.5 * (b - a);
    if (!(delta > 1e-12)) {   // nothing left of R: the explicit model alone
        r->alpha = 1.0;
        r->beta = 0.0;
        return;
    }
    const double sigma = theta / delta, rho0 = 1.0 / sigma, rho1 = 1.0 / (2.0 * sigma - rho0);
    // two Chebyshev steps from zero: z = d0 + d1, d0 = y / theta, res1 = y - (d0 + P1^-1 R d0), d1 = rho1 rho0 d0 + (2 rho1 / delta) res1
    r->alpha = (1.0 + rho1 * rho0) / theta + (2.0 * rho1 / delta) * (1.0 - 1.0 / theta);
    r->beta = (2.0 * rho1 / delta) / theta;
}
void as_pc2_coefs(const as_pc2 *r, double *alpha, double *beta) {
    *alpha = r->alpha;
    *beta = r->beta;
}
