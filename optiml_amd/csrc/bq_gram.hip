// Gram-panel build on fp64 matrix cores: K[r0:r1, :] = kernel(X[r0:r1], X)   (one-time, MFMA-bound).
//
// Replaces optiml/ml/svm/kernels.py:49-51 (linear), :91-95 (poly), :125-129 (gaussian; the squared
// distances follow sklearn's euclidean_distances formula -2xy' + |x|^2 + |y|^2, clamped at 0, zero diagonal
// when both arguments are the same matrix).
//
// Data flow: X (n x d, row-major) is transposed once into a k-major, zero-padded image Xt[dp][np] so that a
// 128-row tile of one k-slice is 1 KiB contiguous: global loads are coalesced 16-byte loads and the LDS image
// [k][row] needs no transposition.  One 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile; each
// wave accumulates a 64x64 block as 4x4 v_mfma_f64_16x16x4_f64 tiles (128 accumulator VGPRs), stepping K in
// chunks of 16 through double-buffered LDS (row pitch 144 doubles: the four k-slices a wave reads in one
// ds_read_b64 land on disjoint bank halves).  The kernel epilogue applies the RBF / polynomial map and
// stores in the panel's storage dtype.
#include "bq_common.h"

#include "bq_mfma_tile.h"

#include <type_traits>

constexpr int GT = BQ_GT;
constexpr int GK = BQ_GK;

__global__ void transpose_pad_kernel(const double *__restrict__ X, int64_t n, int64_t d, double *__restrict__ Xt,
                                     int64_t np, int64_t dp) {
    __shared__ double tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int64_t r = r0 + j, k = k0 + tx;
        tile[j][tx] = (r < n && k < d) ? X[r * d + k] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int64_t k = k0 + j, r = r0 + tx;
        if (k < dp && r < np) Xt[k * np + r] = tile[tx][j];
    }
}

__global__ void row_norms_kernel(const double *__restrict__ X, int64_t n, int64_t d, double *__restrict__ out,
                                 int64_t np) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    double s = 0.0;
    if (i < n) {
        const double *row = X + i * d;
        for (int64_t k = 0; k < d; ++k) s = fma(row[k], row[k], s);
    }
    out[i] = s;
}

template <typename T> __device__ __forceinline__ void store_elem(T *p, double v);
template <> __device__ __forceinline__ void store_elem<double>(double *p, double v) { __builtin_nontemporal_store(v, p); }
template <> __device__ __forceinline__ void store_elem<float>(float *p, double v) { __builtin_nontemporal_store((float)v, p); }
// two adjacent elements (p 16-byte aligned for double, 8-byte for float): one store instruction
__device__ __forceinline__ void store_pair(double *p, double a, double b) {
    bq_d2 v;
    v.x = a;
    v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<bq_d2 *>(p));
}
__device__ __forceinline__ void store_pair(float *p, double a, double b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v;
    v.x = (float)a;
    v.y = (float)b;
    __builtin_nontemporal_store(v, reinterpret_cast<f2 *>(p));
}

// (gamma <x, y> + coef0)^degree: degrees 2 and 3 (the reference's default) by multiplication — pow() is ~150 vector
// instructions per element, x * x * x two; <= 1 ulp from the reference's pow(x, 3.0).  DEG = 0: any degree, pow().
template <int DEG>
__device__ __forceinline__ double bq_poly_map(double x, int degree) {
    if (DEG == 2) return x * x;
    if (DEG == 3) return x * x * x;
    return pow(x, (double)degree);
}
#define BQ_EXP_ATTR __device__ __forceinline__
#define BQ_EXP_LOINT(t) __double2loint(t)
#include "bq_exp.h"

struct gram_params {
    const double *At, *Bt;   // k-major padded images: At[dp][mp], Bt[dp][np]
    const double *a2, *b2;   // squared row norms (padded)
    int64_t m, n;            // rows of A covered by this launch / rows of B (= output columns)
    int64_t mp, np, dp;
    int64_t arow0, arow1;    // global A rows [arow0, arow1) of this launch (arow0 tile-aligned)
    int same;                // A and B are the same matrix -> exact zero distance on the diagonal
    int lower_only;          // symmetric panel: skip tiles strictly above the 256-tile diagonal, packed output layout
    int kernel, degree;
    double gamma, coef0;
    int64_t ld;              // output pitch (elements)
    int64_t ntiles;          // tiles of this launch (XCD-aware remap bound)
};

// One workgroup owns a 128-row tile row and a strip of GRAM_STRIP consecutive 128-column tiles (blockIdx.y): looping over
// the strip keeps the row slice of the image hot and amortises the workgroup start over several tiles (one tile per
// workgroup ran at 36 % MFMA-pipe utilisation: `profiles/r01/pmc_mfma_ip_n50000.txt`).
constexpr int GRAM_STRIP = 8;
template <typename T, int KIND>
__global__ __launch_bounds__(256, 2) void gram_mfma_kernel(gram_params P, T *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    const int64_t arow = P.arow0 + (int64_t)blockIdx.x * GT;
    int64_t tiles_n = (P.n + GT - 1) / GT;
    if (P.lower_only) {   // symmetric panel: columns up to the end of this row's 256-tile on the diagonal
        const int64_t lim = ((arow / BQ_SYM_TILE) + 1) * (BQ_SYM_TILE / GT);
        tiles_n = lim < tiles_n ? lim : tiles_n;
    }
    const int64_t j0 = (int64_t)blockIdx.y * GRAM_STRIP;
    const int64_t j1 = j0 + GRAM_STRIP < tiles_n ? j0 + GRAM_STRIP : tiles_n;
    if (j0 >= j1) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1, ccol = lane & 15, crow = lane >> 4;
    const int64_t I0 = P.arow0 / BQ_SYM_TILE;
    __shared__ double rowsq[4][64];   // squared norms of each wave's 64 rows: through LDS on purpose (see below)
    rowsq[wv][lane] = P.a2[arow + wr * 64 + lane];
    for (int64_t J = j0; J < j1; ++J) {
        const int64_t bcol = J * GT;
        bq_d4 acc[4][4];
        bq_tile_zero(acc);
        bq_mfma_tile_128(P.At, P.mp, arow, P.Bt, P.np, bcol, P.dp, sm, acc);
        // The row bases and norms of the epilogue do not depend on J; hoisted out of this loop they would stay live across
        // the MFMA loop (32+ VGPRs) and spill.  An opaque zero makes them per-iteration values.
        int64_t opaque = 0;
        asm volatile("" : "+s"(opaque));
        // epilogue, row by row: the kernel map per element, two adjacent columns per store.  Instances: only the tile on the
        // diagonal of a symmetric build pays for the "exact zero distance at i == j" test, only tiles that stick out of the
        // matrix for the bounds tests (their branches also cut the rows into separate blocks: two exp chains in flight
        // instead of four).  A row's address is this lane's base (once per tile) + a uniform multiple of the pitch.
        const int64_t pitch = P.lower_only ? bq_sym_pitch(arow / BQ_SYM_TILE) : P.ld;
        const int64_t tile0 = P.lower_only ? bq_sym_addr(arow, 0, I0) : (arow - P.arow0) * P.ld;   // uniform
        T *const lane_base = out + tile0 + (int64_t)(wr * 64 + 2 * crow) * pitch + bcol + wc * 64 + 2 * ccol;
        auto epilogue = [&](auto on_diag, auto on_edge, auto deg) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t gi = opaque + arow + bq_acc_row(i, v);
                    if (decltype(on_edge)::value && gi >= P.arow1) continue;
                    T *rowp = lane_base + (opaque + (i >> 1) * 32 + 8 * v + (i & 1)) * pitch;
                    const double ai = rowsq[wv][bq_acc_row64(i, v)];
#pragma unroll
                    for (int jp = 0; jp < 2; ++jp) {   // columns j = 2 jp and 2 jp + 1 of this lane are adjacent (even, odd)
                        const int64_t gj = bcol + bq_acc_col(2 * jp);
                        if (decltype(on_edge)::value && gj >= P.n) continue;
                        double kv[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const double dot = acc[i][2 * jp + h][v];
                            if (KIND == BQ_KERNEL_RBF) {
                                double dist = -2.0 * dot;
                                dist += ai;
                                dist += P.b2[gj + h];   // padded to the image pitch
                                dist = fmax(dist, 0.0);
                                if (decltype(on_diag)::value && P.same && gi == gj + h) dist = 0.0;
                                kv[h] = bq_exp(-P.gamma * dist);
                            } else if (KIND == BQ_KERNEL_POLY) {
                                kv[h] = bq_poly_map<decltype(deg)::value>(P.gamma * dot + P.coef0, P.degree);
                            } else if (KIND == BQ_KERNEL_SIGMOID) {
                                kv[h] = tanh(P.gamma * dot + P.coef0);
                            } else {
                                kv[h] = dot;
                            }
                        }
                        if (!decltype(on_edge)::value || gj + 1 < P.n)
                            store_pair(rowp + jp * 32, kv[0], kv[1]);
                        else
                            store_elem<T>(rowp + jp * 32, kv[0]);
                    }
                    if (KIND == BQ_KERNEL_RBF || KIND == BQ_KERNEL_POLY) __builtin_amdgcn_sched_barrier(0);   // four chains in flight
                }
            }
        };
        typedef std::integral_constant<int, 0> deg_any;
        typedef std::integral_constant<int, 2> deg_2;
        typedef std::integral_constant<int, 3> deg_3;
        if (arow + GT > P.arow1 || bcol + GT > P.n)
            epilogue(std::true_type{}, std::true_type{}, deg_any{});   // rare: the test for the diagonal rides along
        else if (KIND == BQ_KERNEL_RBF && P.same && arow == bcol)
            epilogue(std::true_type{}, std::false_type{}, deg_any{});
        else if (KIND == BQ_KERNEL_POLY && P.degree == 3)
            epilogue(std::false_type{}, std::false_type{}, deg_3{});
        else if (KIND == BQ_KERNEL_POLY && P.degree == 2)
            epilogue(std::false_type{}, std::false_type{}, deg_2{});
        else
            epilogue(std::false_type{}, std::false_type{}, deg_any{});
        __syncthreads();   // the next tile's prologue refills the LDS buffers
    }
}

// Laplacian kernel exp(-gamma * |x - y|_1) (kernels.py:159-163; sklearn's manhattan_distances sums |x_k - y_k| over
// the features in order).  No GEMM form: a VALU kernel on the same k-major images, 64 x 64 outputs per workgroup,
// 4 x 4 per thread, 32-deep k-chunks through LDS.
constexpr int LT = 64, LK = 32;
template <typename T>
__global__ __launch_bounds__(256) void gram_l1_kernel(gram_params P, T *__restrict__ out) {
    __shared__ double As[LK][LT + 2], Bs[LK][LT + 2];
    const int64_t tiles_n = (P.n + LT - 1) / LT;
    const int64_t tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int64_t arow = P.arow0 + tm * LT, bcol = tn * LT;
    if (P.lower_only && (bcol / BQ_SYM_TILE) > (arow / BQ_SYM_TILE)) return;
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;   // 16 x 16 threads, 4 x 4 outputs each
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
    for (int64_t kc = 0; kc < P.dp; kc += LK) {
        __syncthreads();
        for (int e = tid; e < LK * LT; e += 256) {
            const int k = e / LT, r = e % LT;
            const bool kin = kc + k < P.dp;
            As[k][r] = (kin && arow + r < P.mp) ? P.At[(kc + k) * P.mp + arow + r] : 0.0;
            Bs[k][r] = (kin && bcol + r < P.np) ? P.Bt[(kc + k) * P.np + bcol + r] : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < LK; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = As[k][4 * tr + u];
                b[u] = Bs[k][4 * tc + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += fabs(a[u] - b[v]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t gi = arow + 4 * tr + u, gj = bcol + 4 * tc + v;
            if (gi < P.arow1 && gj < P.n)
                store_elem<T>(out + (P.lower_only ? bq_sym_addr(gi, gj, P.arow0 / BQ_SYM_TILE) : (gi - P.arow0) * P.ld + gj),
                              bq_exp(-P.gamma * acc[u][v]));
        }
}

struct gram_images {
    double *At = nullptr, *a2 = nullptr;
    int64_t mp = 0, dp = 0;
};

static int make_image(bq_ctx *ctx, const double *Xdev, int64_t n, int64_t d, gram_images *img) {
    img->mp = bq_round_up(n, GT);
    img->dp = bq_round_up(d, GK);
    BQ_HIP(hipMalloc(&img->At, sizeof(double) * img->mp * img->dp));
    BQ_HIP(hipMalloc(&img->a2, sizeof(double) * img->mp));
    dim3 grid((unsigned)((img->mp + 31) / 32), (unsigned)((img->dp + 31) / 32));
    transpose_pad_kernel<<<grid, 256, 0, ctx->stream>>>(Xdev, n, d, img->At, img->mp, img->dp);
    row_norms_kernel<<<(unsigned)((img->mp + 255) / 256), 256, 0, ctx->stream>>>(Xdev, n, d, img->a2, img->mp);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

static void free_image(gram_images *img) {
    if (img->At) hipFree(img->At);
    if (img->a2) hipFree(img->a2);
    img->At = img->a2 = nullptr;
}

static int run_gram(bq_ctx *ctx, const gram_images &A, const gram_images &B, int64_t m_rows0, int64_t m_rows1,
                    int64_t n, bool same, int kernel, double gamma, double coef0, int degree, void *out,
                    int storage, int64_t ld, bool lower_only = false) {
    gram_params P;
    P.lower_only = lower_only ? 1 : 0;
    P.At = A.At;
    P.Bt = B.At;
    P.a2 = A.a2;
    P.b2 = B.a2;
    P.m = m_rows1 - m_rows0;
    P.n = n;
    P.mp = A.mp;
    P.np = B.mp;
    P.dp = A.dp;
    P.arow0 = m_rows0;
    P.arow1 = m_rows1;
    P.same = same ? 1 : 0;
    P.kernel = kernel;
    P.degree = degree;
    P.gamma = gamma;
    P.coef0 = coef0;
    P.ld = ld;
    const int64_t tiles_m = (P.m + GT - 1) / GT, tiles_n = (n + GT - 1) / GT;
    if (tiles_m * tiles_n <= 0) return BQ_OK;
    BQ_ARG(tiles_m * tiles_n < (int64_t)2147483647, "Gram grid too large");
    // the A tile of the last tile-row may run past arow1 but never past mp because arow0 is tile-aligned
    BQ_ARG(m_rows0 % GT == 0 || m_rows0 + tiles_m * GT <= A.mp, "row block must keep tiles inside the padded image");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_GRAM, &e0, &e1));
    if (kernel == BQ_KERNEL_LAPLACIAN) {
        const int64_t lm = (P.m + LT - 1) / LT, ln = (n + LT - 1) / LT;
        BQ_ARG(lm * ln < (int64_t)2147483647, "Gram grid too large");
        dim3 lgrid((unsigned)(lm * ln));
        if (storage == BQ_F64)
            gram_l1_kernel<double><<<lgrid, 256, 0, ctx->stream>>>(P, reinterpret_cast<double *>(out));
        else
            gram_l1_kernel<float><<<lgrid, 256, 0, ctx->stream>>>(P, reinterpret_cast<float *>(out));
    } else {
        P.ntiles = tiles_m * tiles_n;
        dim3 grid((unsigned)tiles_m, (unsigned)((tiles_n + GRAM_STRIP - 1) / GRAM_STRIP));
        // one instantiation per kernel map (all of exp / pow / tanh inlined in the 64-element epilogue costs registers)
#define BQ_GRAM_LAUNCH(KIND)                                                                                           \
    do {                                                                                                               \
        if (storage == BQ_F64)                                                                                         \
            gram_mfma_kernel<double, KIND><<<grid, 256, 0, ctx->stream>>>(P, reinterpret_cast<double *>(out));        \
        else                                                                                                           \
            gram_mfma_kernel<float, KIND><<<grid, 256, 0, ctx->stream>>>(P, reinterpret_cast<float *>(out));          \
    } while (0)
        switch (kernel) {
            case BQ_KERNEL_RBF: BQ_GRAM_LAUNCH(BQ_KERNEL_RBF); break;
            case BQ_KERNEL_POLY: BQ_GRAM_LAUNCH(BQ_KERNEL_POLY); break;
            case BQ_KERNEL_SIGMOID: BQ_GRAM_LAUNCH(BQ_KERNEL_SIGMOID); break;
            default: BQ_GRAM_LAUNCH(BQ_KERNEL_LINEAR); break;
        }
#undef BQ_GRAM_LAUNCH
    }
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_prof_end(ctx, BQ_PROF_GRAM, e0, e1));
    return BQ_OK;
}

int bq_launch_gram(bq_ctx *ctx, const double *X, int64_t n, int64_t d, int64_t r0, int64_t r1, int kernel,
                   double gamma, double coef0, int degree, void *panel, int storage, int64_t ld, bool sym_packed) {
    if (r1 <= r0) return BQ_OK;   // a rank that owns no rows (more ranks than tile rows) has nothing to build
    gram_images img;
    // pad the image so that a tile starting at any r0 stays inside it
    int rc = make_image(ctx, X, n, d, &img);
    if (rc != BQ_OK) {
        free_image(&img);
        return rc;
    }
    if (r0 % GT != 0) {
        // row blocks are handed out tile-aligned by bq_row_block; anything else needs a larger pad
        free_image(&img);
        bq_set_error("row block start %lld is not a multiple of %d", (long long)r0, GT);
        return BQ_ERR_BADARG;
    }
    rc = run_gram(ctx, img, img, r0, r1, n, true, kernel, gamma, coef0, degree, panel, storage, ld, sym_packed);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    free_image(&img);
    if (rc != BQ_OK) return rc;
    BQ_HIP(e);
    return BQ_OK;
}

// out[t] = sum_m coef[m] * kernel(SV[m], Xt[t]) + intercept : the m x t cross-Gram goes through the same MFMA
// kernel (A = Xt rows, B = SV rows -> panel t x m), then the panel product with coef.
int bq_launch_decision(bq_ctx *ctx, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                       const double *SV, const double *coef, double intercept, int64_t t, const double *Xt,
                       double *out) {
    gram_images A, B;
    double *dSV = nullptr, *dXt = nullptr, *panel = nullptr, *w = nullptr, *s = nullptr;
    const int64_t ld = bq_round_up(m, BQ_PAD);
    int rc = BQ_OK;
    auto cleanup = [&]() {
        free_image(&A);
        free_image(&B);
        if (dSV) hipFree(dSV);
        if (dXt) hipFree(dXt);
        if (panel) hipFree(panel);
        if (w) hipFree(w);
        if (s) hipFree(s);
    };
#define DEC_HIP(e)                                                          \
    do {                                                                    \
        hipError_t _e = (e);                                                \
        if (_e != hipSuccess) {                                             \
            bq_set_error("%s failed: %s", #e, hipGetErrorString(_e));       \
            cleanup();                                                      \
            return BQ_ERR_HIP;                                              \
        }                                                                   \
    } while (0)
    DEC_HIP(hipMalloc(&dSV, sizeof(double) * m * d));
    DEC_HIP(hipMalloc(&dXt, sizeof(double) * t * d));
    DEC_HIP(hipMalloc(&panel, sizeof(double) * t * ld));
    DEC_HIP(hipMalloc(&w, sizeof(double) * ld));
    DEC_HIP(hipMalloc(&s, sizeof(double) * t));
    DEC_HIP(hipMemcpyAsync(dSV, SV, sizeof(double) * m * d, hipMemcpyHostToDevice, ctx->stream));
    DEC_HIP(hipMemcpyAsync(dXt, Xt, sizeof(double) * t * d, hipMemcpyHostToDevice, ctx->stream));
    DEC_HIP(hipMemsetAsync(panel, 0, sizeof(double) * t * ld, ctx->stream));
    DEC_HIP(hipMemsetAsync(w, 0, sizeof(double) * ld, ctx->stream));
    DEC_HIP(hipMemcpyAsync(w, coef, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = make_image(ctx, dXt, t, d, &A)) != BQ_OK || (rc = make_image(ctx, dSV, m, d, &B)) != BQ_OK) {
        cleanup();
        return rc;
    }
    // kernel(SV, Xt)[m][t] == kernel(Xt, SV)[t][m] for all three kernels (gamma is passed in resolved)
    rc = run_gram(ctx, A, B, 0, t, m, false, kernel, gamma, coef0, degree, panel, BQ_F64, ld);
    if (rc == BQ_OK) rc = bq_launch_gemv(ctx, panel, BQ_F64, false, t, ld, w, s, nullptr);
    if (rc != BQ_OK) {
        cleanup();
        return rc;
    }
    DEC_HIP(hipMemcpyAsync(out, s, sizeof(double) * t, hipMemcpyDeviceToHost, ctx->stream));
    DEC_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < t; ++i) out[i] += intercept;
    cleanup();
#undef DEC_HIP
    return BQ_OK;
}

// Dense Gram matrix kernel(A, B) (m x t) for host callers (the Kernel functors' __call__): B == nullptr means
// B is A (exact zero distance on the diagonal, as sklearn does when Y is X).
int bq_launch_gram_matrix(bq_ctx *ctx, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                          const double *A, int64_t t, const double *B, double *out) {
    gram_images ia, ib;
    double *dA = nullptr, *dB = nullptr, *panel = nullptr;
    const bool same = (B == nullptr);
    if (same) t = m;
    const int64_t ld = bq_round_up(t, BQ_PAD);
    int rc = BQ_OK;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() {
        free_image(&ia);
        if (!same) free_image(&ib);
        if (dA) hipFree(dA);
        if (dB) hipFree(dB);
        if (panel) hipFree(panel);
    };
    do {
        if ((e = hipMalloc(&dA, sizeof(double) * m * d)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(dA, A, sizeof(double) * m * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if (!same) {
            if ((e = hipMalloc(&dB, sizeof(double) * t * d)) != hipSuccess) break;
            if ((e = hipMemcpyAsync(dB, B, sizeof(double) * t * d, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        }
        if ((e = hipMalloc(&panel, sizeof(double) * m * ld)) != hipSuccess) break;
        if ((rc = make_image(ctx, dA, m, d, &ia)) != BQ_OK) break;
        if (!same && (rc = make_image(ctx, dB, t, d, &ib)) != BQ_OK) break;
        if ((rc = run_gram(ctx, ia, same ? ia : ib, 0, m, t, same, kernel, gamma, coef0, degree, panel, BQ_F64, ld)) != BQ_OK) break;
        if ((e = hipMemcpy2DAsync(out, t * 8, panel, ld * 8, t * 8, m, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    cleanup();
    if (rc != BQ_OK) return rc;
    if (e != hipSuccess) {
        bq_set_error("gram matrix failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    return BQ_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Streamed mode (BQ_STREAM): no resident panel — every product recomputes the Gram tiles on the MFMA and contracts them
// with the input vector inside the same kernel (the fallback when n^2 s does not fit in HBM).
// One workgroup owns a 128-row tile of this rank's row block and a chunk of column tiles: per column tile the shared
// tile kernel yields the 128 x 128 dot products, the kernel map turns them into K (or K + 1) and each lane folds its 64
// values into 16 per-row sums.  At the end the 16 lanes that share a row are combined with shuffles, the two waves that
// share a row block through LDS, and the chunk's 128 row sums go to S[chunk][row]; stream_reduce_kernel adds the chunks
// in order.  No symmetry credit (each tile is computed where it is used): 2 n^2 d flop per product.
// ---------------------------------------------------------------------------------------------------------------
// exp / pow chains kept in flight per lane in the streamed product's epilogue (a chain is ~40 dependent fp64 instructions: one
// at a time left the VALU idle; measured 62.5 / 60.4 / 59.8 ms per n=100k product for 1 / 2 / 4)
#ifndef STREAM_EXP_ILP
#define STREAM_EXP_ILP 4
#endif
struct bq_stream_images {
    gram_images img;
    double *S = nullptr;       // nchunk x rows_pad partial products
    int64_t rows_pad = 0;
    int nchunk = 1;
};

#ifdef BQ_DIAG_STAMPS   // diagnostic build only: per-wave phase stamps of four workgroups of the third product -> stderr
__device__ long long bq_stream_stamps[4][4][1 + 3 * 8];
#define STREAM_STAMP(slot)                                                                                         \
    do {                                                                                                           \
        if (sblk >= 0 && (J - j0) < 8 && lane == 0) bq_stream_stamps[sblk][wv][1 + 3 * (J - j0) + (slot)] = wall_clock64(); \
    } while (0)
#else
#define STREAM_STAMP(slot) do { } while (0)
#endif

template <int KIND>
__global__ __launch_bounds__(256, 2) void gram_stream_kernel(gram_params P, const double *__restrict__ w, int add_one,
                                                             int64_t tiles_per_chunk, double *__restrict__ S,
                                                             int64_t rows_pad, const int *done) {
    if (done != nullptr && *done) return;
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    __shared__ double rowsum[4][64];   // per wave: running sums of its 64 rows over its 64 columns of every tile so far
    __shared__ double rowsq[4][64];    // per wave: squared norms of its 64 rows (kept out of the registers on purpose: as
                                       // loop invariants they would stay live across the MFMA loop and force spills)
    const int64_t tiles_n = (P.n + GT - 1) / GT;
    const int64_t tm = blockIdx.x, chunk = blockIdx.y;
    const int64_t arow = P.arow0 + tm * GT;
    const int64_t j0 = chunk * tiles_per_chunk, j1 = (j0 + tiles_per_chunk < tiles_n) ? j0 + tiles_per_chunk : tiles_n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1, ccol = lane & 15;
    rowsum[wv][lane] = 0.0;
    rowsq[wv][lane] = P.a2[arow + (wv >> 1) * 64 + lane];
#ifdef BQ_DIAG_STAMPS
    const int sblk = (blockIdx.y == 1 && (blockIdx.x == 0 || blockIdx.x == 1 || blockIdx.x == 256 || blockIdx.x == 257))
                         ? (int)(blockIdx.x & 1) + 2 * (int)(blockIdx.x >> 8) : -1;
    if (sblk >= 0 && lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        bq_stream_stamps[sblk][wv][0] = ((long long)xcc << 32) | hw;
    }
#endif
    for (int64_t J = j0; J < j1; ++J) {
        const int64_t bcol = J * GT;
        bq_d4 acc[4][4];
        bq_tile_zero(acc);
        STREAM_STAMP(0);
        bq_mfma_tile_128(P.At, P.mp, arow, P.Bt, P.np, bcol, P.dp, sm, acc);
        STREAM_STAMP(1);
        __builtin_amdgcn_sched_barrier(0);   // keep the epilogue's loads (w, norms) below the MFMA loop: hoisted above it
                                             // they stay live across it and spill
        int64_t opaque = 0;                  // ... and keep the J-invariant row indices per-iteration values (same reason)
        asm volatile("" : "+s"(opaque));
        // epilogue: kernel map, contraction with this tile's slice of w, fold over the 16 lanes that share a row; the
        // running row sums live in LDS so that no accumulator stays in registers across the MFMA loop.  Two instances: only
        // the tile on the diagonal pays for the "exact zero distance at i == j" test
        const double one = add_one ? 1.0 : 0.0;   // K + 1 without a select per element (x + 0.0 == x for every x the maps produce)
        auto epilogue = [&](auto on_diag, auto deg) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t gi = opaque + arow + bq_acc_row(i, v);
                    const double ai = KIND == BQ_KERNEL_RBF ? rowsq[wv][bq_acc_row64(i, v)] : 0.0;
                    double part = 0.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t gj = bcol + bq_acc_col(j);
                        const double dot = acc[i][j][v];
                        double kv;
                        if (KIND == BQ_KERNEL_RBF) {
                            double dist = -2.0 * dot;
                            dist += ai;
                            dist += P.b2[gj];
                            dist = fmax(dist, 0.0);
                            if (decltype(on_diag)::value && gi == gj) dist = 0.0;
                            kv = bq_exp(-P.gamma * dist);
                        } else if (KIND == BQ_KERNEL_POLY) {
                            kv = bq_poly_map<decltype(deg)::value>(P.gamma * dot + P.coef0, P.degree);
                        } else if (KIND == BQ_KERNEL_SIGMOID) {
                            kv = tanh(P.gamma * dot + P.coef0);
                        } else {
                            kv = dot;
                        }
                        kv += one;
                        part = fma(kv, w[gj], part);   // w is zero beyond n (padded to the panel pitch)
                        if ((KIND == BQ_KERNEL_RBF || KIND == BQ_KERNEL_POLY) && ((j + 1) % STREAM_EXP_ILP == 0))
                            __builtin_amdgcn_sched_barrier(0);   // STREAM_EXP_ILP exp / pow chains in flight (register pressure)
                    }
                    part += __shfl_xor(part, 1, 64);
                    part += __shfl_xor(part, 2, 64);
                    part += __shfl_xor(part, 4, 64);
                    part += __shfl_xor(part, 8, 64);
                    if (ccol == 0) rowsum[wv][bq_acc_row64(i, v)] += part;
                }
            }
        };
        if (KIND == BQ_KERNEL_RBF && arow == bcol)
            epilogue(std::true_type{}, std::integral_constant<int, 0>{});
        else if (KIND == BQ_KERNEL_POLY && P.degree == 3)
            epilogue(std::false_type{}, std::integral_constant<int, 3>{});
        else if (KIND == BQ_KERNEL_POLY && P.degree == 2)
            epilogue(std::false_type{}, std::integral_constant<int, 2>{});
        else
            epilogue(std::false_type{}, std::integral_constant<int, 0>{});
        STREAM_STAMP(2);
        __syncthreads();   // the next tile's prologue refills the LDS buffers
    }
    __syncthreads();
    if (wc == 0) {   // the two waves of a row block (column halves) are combined in a fixed order
        const int64_t gi = arow + wr * 64 + lane;
        S[chunk * rows_pad + (gi - P.arow0)] = gi < P.arow1 ? rowsum[wv][lane] + rowsum[wv + 1][lane] : 0.0;
    }
}

__global__ void stream_reduce_kernel(const double *__restrict__ S, int64_t rows, int64_t rows_pad, int nchunk,
                                     double *__restrict__ out, const int *done) {
    if (done != nullptr && *done) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    double a = 0.0;
    for (int c = 0; c < nchunk; ++c) a += S[(int64_t)c * rows_pad + i];
    out[i] = a;
}

int bq_stream_prepare(bq_ctx *ctx, const double *Xdev, int64_t n, int64_t d, int64_t r0, int64_t r1, void **out) {
    bq_stream_images *st = new bq_stream_images();
    int rc = make_image(ctx, Xdev, n, d, &st->img);
    if (rc != BQ_OK) {
        delete st;
        return rc;
    }
    const int64_t tiles_m = (r1 - r0 + GT - 1) / GT, tiles_n = (n + GT - 1) / GT;
    st->rows_pad = (tiles_m > 0 ? tiles_m : 1) * GT;
    // ~24 rounds of workgroups over the chip's 2 x num_cu slots (all workgroups last the same: with 4.6 rounds — three chunks
    // at n = 100 000 — the fifth, half-empty round cost 9 %), but never more chunks than column tiles
    int nchunk = (int)((24 * (int64_t)ctx->num_cu * 2 + tiles_m - 1) / (tiles_m > 0 ? tiles_m : 1));
    nchunk = nchunk < 1 ? 1 : (nchunk > 32 ? 32 : nchunk);
    if (nchunk > tiles_n) nchunk = (int)tiles_n;
    st->nchunk = nchunk;
    if (hipMalloc(&st->S, sizeof(double) * st->rows_pad * nchunk) != hipSuccess) {
        bq_set_error("cannot allocate the streamed-product scratch");
        free_image(&st->img);
        delete st;
        return BQ_ERR_NOMEM;
    }
    *out = st;
    return BQ_OK;
}

void bq_stream_free(void *h) {
    if (!h) return;
    bq_stream_images *st = (bq_stream_images *)h;
    free_image(&st->img);
    if (st->S) hipFree(st->S);
    delete st;
}

// out_rows[0 : r1 - r0) = (K or K + 1)[r0:r1, :] w
int bq_stream_product(bq_ctx *ctx, void *h, int64_t n, int64_t r0, int64_t r1, int kernel, double gamma, double coef0,
                      int degree, bool add_one, const double *w, double *out_rows, const int *done) {
    if (r1 <= r0) return BQ_OK;
    bq_stream_images *st = (bq_stream_images *)h;
    BQ_ARG(kernel != BQ_KERNEL_LAPLACIAN, "the streamed mode is built for the inner-product kernels (linear, poly, rbf, sigmoid)");
    gram_params P;
    P.lower_only = 0;
    P.At = P.Bt = st->img.At;
    P.a2 = P.b2 = st->img.a2;
    P.m = r1 - r0;
    P.n = n;
    P.mp = P.np = st->img.mp;
    P.dp = st->img.dp;
    P.arow0 = r0;
    P.arow1 = r1;
    P.same = 1;
    P.kernel = kernel;
    P.degree = degree;
    P.gamma = gamma;
    P.coef0 = coef0;
    P.ld = 0;
    P.ntiles = 0;
    const int64_t tiles_m = (P.m + GT - 1) / GT, tiles_n = (n + GT - 1) / GT;
    BQ_ARG(r0 % GT == 0, "row block must be tile aligned");
    const int64_t per = (tiles_n + st->nchunk - 1) / st->nchunk;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_MATVEC, &e0, &e1));
    dim3 grid((unsigned)tiles_m, (unsigned)st->nchunk);
    // one instantiation per kernel map: with all of exp / pow / tanh inlined in the 64-element epilogue the kernel spilled
    switch (kernel) {
        case BQ_KERNEL_RBF:
            gram_stream_kernel<BQ_KERNEL_RBF><<<grid, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, per, st->S, st->rows_pad, done);
            break;
        case BQ_KERNEL_POLY:
            gram_stream_kernel<BQ_KERNEL_POLY><<<grid, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, per, st->S, st->rows_pad, done);
            break;
        case BQ_KERNEL_SIGMOID:
            gram_stream_kernel<BQ_KERNEL_SIGMOID><<<grid, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, per, st->S, st->rows_pad, done);
            break;
        default:
            gram_stream_kernel<BQ_KERNEL_LINEAR><<<grid, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, per, st->S, st->rows_pad, done);
            break;
    }
    stream_reduce_kernel<<<(unsigned)((P.m + 255) / 256), 256, 0, ctx->stream>>>(st->S, P.m, st->rows_pad, st->nchunk,
                                                                                 out_rows, done);
    BQ_HIP(hipGetLastError());
#ifdef BQ_DIAG_STAMPS
    {
        static int calls = 0;
        if (++calls == 3) {
            static long long h[4][4][25];
            hipStreamSynchronize(ctx->stream);
            hipMemcpyFromSymbol(h, HIP_SYMBOL(bq_stream_stamps), sizeof(h));
            for (int b = 0; b < 4; ++b)
                for (int w4 = 0; w4 < 4; ++w4) {
                    fprintf(stderr, "stamps blk %d wave %d xcc %lld hw_id 0x%llx t0 %lld:", b, w4, h[b][w4][0] >> 32,
                            h[b][w4][0] & 0xffffffffll, h[b][w4][1]);
                    for (int j = 0; j < 8; ++j)
                        fprintf(stderr, "  [mfma %.2f epi %.2f gap %.2f]", (h[b][w4][2 + 3 * j] - h[b][w4][1 + 3 * j]) / 100.0,
                                (h[b][w4][3 + 3 * j] - h[b][w4][2 + 3 * j]) / 100.0,
                                j < 7 ? (h[b][w4][4 + 3 * j] - h[b][w4][3 + 3 * j]) / 100.0 : 0.0);
                    fprintf(stderr, "\n");
                }
        }
    }
#endif
    BQ_TRY(bq_prof_end(ctx, BQ_PROF_MATVEC, e0, e1));
    return BQ_OK;
}
