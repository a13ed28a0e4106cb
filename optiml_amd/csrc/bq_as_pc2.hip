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
// Everything is a fixed-order sum.  The two products are sums over SAMPLES and are sharded like the panel product (round 5): the
// sample blocks are cut into the canonical segments of bq_sym_segments(world) (8 up to 8 ranks), a rank owns a run of them and forms
// their slices of M, the per-segment sums are gathered (8 x d^2 doubles: 4 MB at d = 256) and added in segment order on every rank;
// x_i' M x_i and v = R y are formed for the rank's own samples and v is gathered (n doubles).  Two collectives per application, the
// same bits for any rank count (and the same association on one rank).  The explicit model P1 stays replicated.
#include "bq_as.h"
#include "bq_mfma_tile.h"

struct as_pc2 {
    int64_t n = 0, d = 0, dp = 0, ld = 0;   // samples, features, features padded to 128, vector pitch (s->ldN)
    int64_t rows = 0;                       // rows of Xp / W (= ld)
    // the sample segments (in blocks of 1024 samples) and their slices (<= 2 blocks each: the split-K of M)
    as_pc_part part;
    int nsl = 0, own_sl_lo = 0, own_sl_hi = 0;   // slices in all; this rank's (its segments are a contiguous run)
    long long *sl_row = nullptr, *sl_k = nullptr; // device: first row and length (samples) of slice s
    int *sl_first = nullptr;                      // device: first slice of segment k (k = S: nsl)
    int64_t gstride = 0;      // doubles per slot of Mg: dp x dp + the caller's tail (as_pc_r_apply: the segment sums of Phi_top' y ride on
                              // the same all-gather)
    double *Mg = nullptr;     // (world * cmax) x gstride: the per-segment sums of M (+ tail), gathered
    double *vg = nullptr;     // (world * cmax) x maxlen x 1024: v in the gathered layout
    double *Xp = nullptr;     // rows x dp, row-major, zero padded: the k-major image of X with k = sample
    double *Xt = nullptr;     // dp x ld: the k-major image with k = feature
    double *W = nullptr;      // rows x dp: diag(w) Xp, w = c o y
    double *Mpart = nullptr;  // nsl x dp x dp (this rank fills its own slices)
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
__global__ __launch_bounds__(256) void pc2_scale_kernel(int64_t row0, int64_t row1, int64_t dp, const double *__restrict__ Xp,
                                                        const double *__restrict__ c, const double *__restrict__ y, double *__restrict__ W,
                                                        const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    const int64_t i = row0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // a wave per sample row, this rank's rows
    if (i >= row1) return;
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
__global__ __launch_bounds__(256, 2) void pc2_moment_kernel(int64_t dp, int sl0, const long long *__restrict__ sl_row,
                                                            const long long *__restrict__ sl_k, const double *__restrict__ Xp,
                                                            const double *__restrict__ W, double *__restrict__ Mpart, const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    int64_t ta = 0;
    while ((ta + 1) * (ta + 2) / 2 <= (int64_t)blockIdx.x) ++ta;
    const int64_t tb = (int64_t)blockIdx.x - ta * (ta + 1) / 2;
    const int sl = sl0 + (int)blockIdx.y;
    const int64_t k0 = sl_row[sl], kc = sl_k[sl];
    bq_d4 acc[4][4];
    bq_tile_zero(acc);
    bq_mfma_tile_128(Xp + k0 * dp, dp, ta * BQ_GT, W + k0 * dp, dp, tb * BQ_GT, kc, sm, acc);
    bq_tile_store(acc, Mpart + ((int64_t)sl * dp + ta * BQ_GT) * dp + tb * BQ_GT, dp);
}

// Mg[slot(k)] = the slices of segment k added in slice order (blockIdx.y = k - first own segment); tiles above the diagonal mirrored
__global__ __launch_bounds__(256) void pc2_moment_seg_kernel(int64_t dp, int64_t gstride, as_pc_part part, const int *__restrict__ sl_first,
                                                             const double *__restrict__ Mpart, double *__restrict__ Mg, const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= dp * dp) return;
    const int k = part.lo + (int)blockIdx.y;
    const int64_t ra = o / dp, rb = o % dp;
    const int64_t e = (ra / BQ_GT >= rb / BQ_GT) ? o : rb * dp + ra;
    double a0 = 0.0, a1 = 0.0;   // two interleaved chains, combined in a fixed order (a segment has at most a few dozen slices)
    int sl = sl_first[k];
    const int end = sl_first[k + 1];
    for (; sl + 2 <= end; sl += 2) {
        a0 += Mpart[(int64_t)sl * dp * dp + e];
        a1 += Mpart[(int64_t)(sl + 1) * dp * dp + e];
    }
    if (sl < end) a0 += Mpart[(int64_t)sl * dp * dp + e];
    Mg[(int64_t)part.slot[k] * gstride + o] = a0 + a1;
}
// M = the gathered per-segment sums added in segment order (every rank: the same bits)
__global__ __launch_bounds__(256) void pc2_moment_final_kernel(int64_t dp, int64_t gstride, as_pc_part part, const double *__restrict__ Mg,
                                                               double *__restrict__ M, const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= dp * dp) return;
    double a = 0.0;
    for (int k = 0; k < part.S; ++k) a += Mg[(int64_t)part.slot[k] * gstride + o];
    M[o] = a;
}

// ypart[ct][i] = sum over the 128 columns b of column tile ct of (X M)[i][b] X[i][b]: blockIdx.x = sample tile, blockIdx.y = ct
__global__ __launch_bounds__(256, 2) void pc2_bilinear_kernel(int64_t dp, int64_t ld, int64_t tile0, const double *__restrict__ Xt,
                                                              const double *__restrict__ Xp, const double *__restrict__ M,
                                                              double *__restrict__ ypart, const as_cg_scal *cg) {
    if (cg != nullptr && cg->done) return;
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    __shared__ double half[2][BQ_GT];
    const int64_t i0 = (tile0 + (int64_t)blockIdx.x) * BQ_GT, b0 = (int64_t)blockIdx.y * BQ_GT;
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
    for (void *p : {(void *)r->Xp, (void *)r->Xt, (void *)r->W, (void *)r->Mpart, (void *)r->M, (void *)r->ypart, (void *)r->c, (void *)r->sl_row,
                    (void *)r->sl_k, (void *)r->sl_first, (void *)r->Mg, (void *)r->vg})
        if (p) hipFree(p);
    delete r;
}

// buffers and images; bdiag_out (device, ldN): the diagonal of B, which the caller takes out of D
int as_pc2_create(bq_solver *s, double *bdiag_out, int64_t tail, as_pc2 **out) {
    *out = nullptr;
    bq_problem *p = s->p;
    hipStream_t st = p->ctx->stream;
    as_pc2 *r = new as_pc2();
    r->n = p->n;
    r->d = p->d;
    r->dp = bq_round_up(p->d, BQ_GT);
    r->ld = s->ldN;
    const int64_t T = r->dp / BQ_GT;
    r->rows = r->ld;
    r->gstride = r->dp * r->dp + tail;
    // the canonical sample segments (as many as the panel's: 8 up to 8 ranks), in blocks of 1024 samples, this rank's run of them
    // and the slot of every segment in the gathered buffers; slices of at most two blocks (2 048 samples: ~3 x 123 workgroups at
    // config 5) inside every segment
    bq_ctx *ctx = p->ctx;
    as_pc_part &pt = r->part;
    const int64_t nblk = r->ld / BQ_VEC_TILE;
    as_pc_make_part(ctx, nblk, &pt);
    std::vector<long long> sl_row, sl_k;
    std::vector<int> sl_first((size_t)pt.S + 1, 0);
    for (int k = 0; k < pt.S; ++k) {
        sl_first[(size_t)k] = (int)sl_row.size();
        if (k == pt.lo) r->own_sl_lo = (int)sl_row.size();
        for (long long b = pt.blk[k]; b < pt.blk[k + 1]; b += 2) {
            sl_row.push_back(b * BQ_VEC_TILE);
            sl_k.push_back(std::min<long long>(2, pt.blk[k + 1] - b) * BQ_VEC_TILE);
        }
        if (k + 1 == pt.hi) r->own_sl_hi = (int)sl_row.size();
    }
    sl_first[(size_t)pt.S] = (int)sl_row.size();
    if (pt.hi <= pt.lo) r->own_sl_lo = r->own_sl_hi = 0;
    r->nsl = (int)sl_row.size();
    const size_t nsl1 = (size_t)std::max(r->nsl, 1);
    hipError_t e = hipMalloc(&r->Xp, sizeof(double) * r->rows * r->dp);
    if (e == hipSuccess) e = hipMalloc(&r->sl_row, sizeof(long long) * nsl1);
    if (e == hipSuccess) e = hipMalloc(&r->sl_k, sizeof(long long) * nsl1);
    if (e == hipSuccess) e = hipMalloc(&r->sl_first, sizeof(int) * ((size_t)pt.S + 1));
    if (e == hipSuccess) e = hipMalloc(&r->Mg, sizeof(double) * (size_t)ctx->world * pt.cmax * r->gstride);
    if (e == hipSuccess) e = hipMemsetAsync(r->Mg, 0, sizeof(double) * (size_t)ctx->world * pt.cmax * r->gstride, st);
    if (e == hipSuccess) e = hipMalloc(&r->vg, sizeof(double) * (size_t)ctx->world * pt.cmax * pt.maxlen * BQ_VEC_TILE);
    if (e == hipSuccess) e = hipMemsetAsync(r->vg, 0, sizeof(double) * (size_t)ctx->world * pt.cmax * pt.maxlen * BQ_VEC_TILE, st);
    if (e == hipSuccess && r->nsl > 0) e = hipMemcpyAsync(r->sl_row, sl_row.data(), sizeof(long long) * r->nsl, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && r->nsl > 0) e = hipMemcpyAsync(r->sl_k, sl_k.data(), sizeof(long long) * r->nsl, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(r->sl_first, sl_first.data(), sizeof(int) * ((size_t)pt.S + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && bq_ctx_sync(ctx) != BQ_OK) e = hipErrorUnknown;   // the three tables leave this scope (the bounded wait: the stream may hold collectives)
    if (e == hipSuccess) e = hipMalloc(&r->Xt, sizeof(double) * r->dp * r->ld);
    if (e == hipSuccess) e = hipMalloc(&r->W, sizeof(double) * r->rows * r->dp);
    if (e == hipSuccess) e = hipMalloc(&r->Mpart, sizeof(double) * nsl1 * r->dp * r->dp);
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
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    const int64_t T = r->dp / BQ_GT;
    const as_pc_part &pt = r->part;
    const int64_t row0 = pt.blk[pt.lo] * BQ_VEC_TILE, row1 = pt.blk[pt.hi] * BQ_VEC_TILE;   // this rank's samples
    const int own_sl = r->own_sl_hi - r->own_sl_lo, own_seg = pt.hi - pt.lo;
    const unsigned ge = (unsigned)((r->dp * r->dp + 255) / 256);
    if (row1 > row0) pc2_scale_kernel<<<(unsigned)((row1 - row0 + 3) / 4), 256, 0, st>>>(row0, row1, r->dp, r->Xp, r->c, y, r->W, cg);
    if (own_sl > 0)
        pc2_moment_kernel<<<dim3((unsigned)(T * (T + 1) / 2), (unsigned)own_sl), 256, 0, st>>>(r->dp, r->own_sl_lo, r->sl_row, r->sl_k, r->Xp, r->W,
                                                                                            r->Mpart, cg);
    if (own_seg > 0) pc2_moment_seg_kernel<<<dim3(ge, (unsigned)own_seg), 256, 0, st>>>(r->dp, r->gstride, pt, r->sl_first, r->Mpart, r->Mg, cg);
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_exchange_gather(ctx, r->Mg, (int64_t)pt.cmax * r->gstride));
    pc2_moment_final_kernel<<<ge, 256, 0, st>>>(r->dp, r->gstride, pt, r->Mg, r->M, cg);
    if (row1 > row0)
        pc2_bilinear_kernel<<<dim3((unsigned)((row1 - row0) / BQ_GT), (unsigned)T), 256, 0, st>>>(r->dp, r->ld, row0 / BQ_GT, r->Xt, r->Xp, r->M,
                                                                                               r->ypart, cg);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
const as_pc_part *as_pc2_part(const as_pc2 *r) { return &r->part; }
double *as_pc2_tail(const as_pc2 *r, int64_t *gstride) {   // the caller's tail of slot 0 of the gathered per-segment buffer
    *gstride = r->gstride;
    return r->Mg + r->dp * r->dp;
}
double *as_pc2_vg(const as_pc2 *r) { return r->vg; }
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
