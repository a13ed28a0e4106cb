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
#include <vector>

constexpr int GT = BQ_GT;
constexpr int GK = BQ_GK;
#ifndef BQ_STREAM_FOLD
#define BQ_STREAM_FOLD 0
#endif

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
    int rect_rb, rect_sb;    // > 0: 1-D grid walked in XCD-local rectangles of GRAM_RECT_R tile rows x GRAM_RECT_S strips
    int tiles_m, strips;
};

// One workgroup owns a 128-row tile row and a strip of GRAM_STRIP consecutive 128-column tiles (blockIdx.y): looping over
// the strip keeps the row slice of the image hot and amortises the workgroup start over several tiles (one tile per
// workgroup ran at 36 % MFMA-pipe utilisation: `profiles/r01/pmc_mfma_ip_n50000.txt`).
constexpr int GRAM_STRIP = 8;
// Workgroup order.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, MI355X_MICROARCH.md), each XCD
// has its own 4 MiB L2 and keeps 64 workgroups in flight.  The 64 consecutive work items of ONE XCD form a rectangle of
// GRAM_RECT_R tile rows x GRAM_RECT_S column strips: its 16 row slices of the image (2 MiB) stay in that L2 while every
// column tile streams through once for 16 consumers, instead of every workgroup re-fetching its row slice for each of its
// 8 column tiles (round 2: 44 GB of L2 fills per n = 100 000 build for a 0.1 GB operand).  Speed only: any placement is correct.
constexpr int GRAM_RECT_R = 16, GRAM_RECT_S = 4;
// (Round 4 tried to break the lockstep of the co-resident workgroups — 512 MFMAs per wave, then 64 map evaluations and 32 stores per
// lane with the matrix pipe idle, chip-wide in phase — by delaying the first round of workgroups by a phase: five modes x three
// units measured 25.7 - 26.4 ms against 25.7 - 26.2 ms without, profiles/r04/gram_stagger.txt.  No effect; removed in round 5.)
template <typename T, int KIND>
__global__ __launch_bounds__(256, 2) void gram_mfma_kernel(gram_params P, T *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    int64_t bx = blockIdx.x, by = blockIdx.y;
    if (P.rect_rb > 0) {
        const int64_t b = blockIdx.x;
        const int64_t xcd = b & 7, slot = b >> 3;
        const int64_t rect = (slot / (GRAM_RECT_R * GRAM_RECT_S)) * 8 + xcd, idx = slot % (GRAM_RECT_R * GRAM_RECT_S);
        if (rect >= (int64_t)P.rect_rb * P.rect_sb) return;
        bx = (rect % P.rect_rb) * GRAM_RECT_R + idx % GRAM_RECT_R;
        by = (rect / P.rect_rb) * GRAM_RECT_S + idx / GRAM_RECT_R;
        if (bx >= P.tiles_m || by >= P.strips) return;
    }
    const int64_t arow = P.arow0 + bx * GT;
    int64_t tiles_n = (P.n + GT - 1) / GT;
    if (P.lower_only) {   // symmetric panel: columns up to the end of this row's 256-tile on the diagonal
        const int64_t lim = ((arow / BQ_SYM_TILE) + 1) * (BQ_SYM_TILE / GT);
        tiles_n = lim < tiles_n ? lim : tiles_n;
    }
    const int64_t j0 = by * GRAM_STRIP;
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
        double bj[4];   // the squared norms of this lane's four columns, once per tile (b2 is padded to the image pitch)
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = KIND == BQ_KERNEL_RBF ? P.b2[bcol + bq_acc_col(2 * (j >> 1)) + (j & 1)] : 0.0;
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
                                dist += bj[2 * jp + h];
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
    P.rect_rb = P.rect_sb = P.tiles_m = P.strips = 0;
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
        const int64_t strips = (tiles_n + GRAM_STRIP - 1) / GRAM_STRIP;
        dim3 grid((unsigned)tiles_m, (unsigned)strips);
        P.rect_rb = P.rect_sb = 0;
        P.tiles_m = (int)tiles_m;
        P.strips = (int)strips;
        if (tiles_m * strips >= 8 * GRAM_RECT_R * GRAM_RECT_S) {
            P.rect_rb = (int)((tiles_m + GRAM_RECT_R - 1) / GRAM_RECT_R);
            P.rect_sb = (int)((strips + GRAM_RECT_S - 1) / GRAM_RECT_S);
            const int64_t nrect = (int64_t)P.rect_rb * P.rect_sb;
            grid = dim3((unsigned)(((nrect + 7) / 8) * 8 * GRAM_RECT_R * GRAM_RECT_S), 1);
        }
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

// out[t] = sum_m coef[m] * kernel(SV[m], Xt[t]) + intercept : the cross-Gram goes through the same MFMA kernel (A = Xt rows,
// B = SV rows -> panel rows x m), then the row-block product with coef — in CHUNKS of test points, so that the t x m panel is never
// held whole (round 6: 10^6 test points against 50 000 support vectors would have asked for 400 GB; a chunk's panel is <= 2 GiB).
// Every output row is formed by the same kernels on the same operands whatever the chunking: the values do not depend on it
// (hook decision_chunk_rows forces small chunks in the tests).
int bq_launch_decision(bq_ctx *ctx, int kernel, double gamma, double coef0, int degree, int64_t m, int64_t d,
                       const double *SV, const double *coef, double intercept, int64_t t, const double *Xt,
                       double *out) {
    gram_images A, B;
    double *dSV = nullptr, *dXt = nullptr, *panel = nullptr, *w = nullptr, *s = nullptr;
    const int64_t ld = bq_round_up(m, BQ_PAD);
    int64_t chunk = std::max<int64_t>(GT, ((int64_t)1 << 28) / ld / GT * GT);   // rows of test points per pass: <= 2 GiB of panel
    {
        double hv = 0.0;
        if (bq_hook("decision_chunk_rows", &hv) && hv >= 1.0) chunk = bq_round_up((int64_t)hv, GT);
    }
    chunk = std::min(chunk, bq_round_up(t, GT));
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
    DEC_HIP(hipMalloc(&panel, sizeof(double) * chunk * ld));
    DEC_HIP(hipMalloc(&w, sizeof(double) * ld));
    DEC_HIP(hipMalloc(&s, sizeof(double) * bq_round_up(t, GT)));
    DEC_HIP(hipMemcpyAsync(dSV, SV, sizeof(double) * m * d, hipMemcpyHostToDevice, ctx->stream));
    DEC_HIP(hipMemcpyAsync(dXt, Xt, sizeof(double) * t * d, hipMemcpyHostToDevice, ctx->stream));
    DEC_HIP(hipMemsetAsync(panel, 0, sizeof(double) * chunk * ld, ctx->stream));   // the pad columns m .. ld stay zero: the row product reads them
    DEC_HIP(hipMemsetAsync(w, 0, sizeof(double) * ld, ctx->stream));
    DEC_HIP(hipMemcpyAsync(w, coef, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = make_image(ctx, dXt, t, d, &A)) != BQ_OK || (rc = make_image(ctx, dSV, m, d, &B)) != BQ_OK) {
        cleanup();
        return rc;
    }
    // kernel(SV, Xt)[m][t] == kernel(Xt, SV)[t][m] for all the kernels (gamma is passed in resolved)
    for (int64_t r0 = 0; rc == BQ_OK && r0 < t; r0 += chunk) {
        const int64_t r1 = std::min(t, r0 + chunk);
        rc = run_gram(ctx, A, B, r0, r1, m, false, kernel, gamma, coef0, degree, panel, BQ_F64, ld);
        if (rc == BQ_OK) rc = bq_launch_gemv(ctx, panel, BQ_F64, false, r1 - r0, ld, w, s + r0, nullptr);
    }
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
// ---------------------------------------------------------------------------------------------------------------
// exp / pow chains kept in flight per lane in the epilogue (a chain is ~20-40 dependent fp64 instructions: one at a time left
// the vector ALU idle)
#ifndef STREAM_SYM_EXP_ILP
#define STREAM_SYM_EXP_ILP 4   // measured 24.7 / 24.3 / 24.0 ms per n = 100 000 product for 1 / 2 / 4
#endif
struct bq_stream_images {
    gram_images img;
    // this rank owns the 128-row tiles [t0, t1) (whole canonical segments) and forms every tile (I, J), J <= I, of those rows
    // once; work units = (row tile, range of U column tiles)
    int64_t T = 0, t0 = 0, t1 = 0, nunits = 0;
    int64_t U = 1, kmax = 1;    // column tiles per unit (a function of n only: sums associate the same for any rank count), ranges per row
    int *unit = nullptr;        // nunits x 4: row tile, first column tile, end column tile, slot in SU ((row - t0) * kmax + range)
    double *SU = nullptr;       // (t1 - t0) x kmax x 128: row sums of the units
    double *slab = nullptr;     // column sums of the off-diagonal tile (I, J), I in [t0, t1), at (I (I - 1) - t0 (t0 - 1)) / 2 + J
};

// ---------------------------------------------------------------------------------------------------------------
// K is symmetric, so a tile (I, J), J < I, is formed ONCE and used twice: its row sums K_IJ w_J go to y_I, its column sums
// K_IJ^T w_I to y_J — half the MFMAs and half the kernel-map evaluations of a row-block form (rounds 1-2a: every rank formed all
// tiles of its rows; n = 100 000 RBF 44.6 ms per product against 24.1 ms now).
// Ranks own whole canonical segments of tile rows, exactly as for the resident symmetric panels (bq_sym_segments): a rank
// forms the tiles (I, J <= I) of its rows.  A workgroup owns a unit = (row tile I, a range of U column tiles): row sums
// accumulate in LDS over the unit and land in SU[unit]; the 128 column sums of every off-diagonal tile land in the slab (each
// lane folds its 16 rows per column, the four lane groups of a wave meet through shuffles, the two waves of a column half
// through LDS).  The reduce kernels then form, per segment s and output tile t, y_s[t] = (the units of row t in range order, if
// t lies in s) + (the column sums of the tiles (I, t), I in s, I > t, ascending), and add the segment vectors in segment order
// — on one rank in the same kernel, across ranks after the all-gather (bq_launch_symv_segsum).  U depends on n only, so every
// sum associates the same way for any rank count: streamed products are bit-identical for 1 / 2 / 4 / 8 ranks like the
// resident ones.  No atomics.
// ---------------------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256, 2) void gram_stream_sym_kernel(gram_params P, const double *__restrict__ w, int add_one,
                                                                 const int *__restrict__ unit, double *__restrict__ SU,
                                                                 double *__restrict__ slab, int64_t t0, const int *done,
                                                                 int *__restrict__ skip, int skip_seq) {
    if (done != nullptr && *done) {
        if (skip != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *skip = skip_seq;   // bq_prof_skip_arg
        return;
    }
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    __shared__ double rowsum[4][64];
    __shared__ double rowsq[4][64];
    __shared__ double wrow[4][64];      // w over the unit's rows, per wave like rowsq
    __shared__ double colsum[2][128];   // per row half: column sums of the current tile
    const int I = unit[4 * blockIdx.x], j0 = unit[4 * blockIdx.x + 1], j1 = unit[4 * blockIdx.x + 2], slot = unit[4 * blockIdx.x + 3];
    const int64_t arow = (int64_t)I * GT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1, ccol = lane & 15, crow = lane >> 4;
    rowsum[wv][lane] = 0.0;
    rowsq[wv][lane] = P.a2[arow + wr * 64 + lane];
    wrow[wv][lane] = w[arow + wr * 64 + lane];
    for (int J = j0; J < j1; ++J) {
        const int64_t bcol = (int64_t)J * GT;
        bq_d4 acc[4][4];
        bq_tile_zero(acc);
        bq_mfma_tile_128(P.At, P.mp, arow, P.Bt, P.np, bcol, P.dp, sm, acc);
        __builtin_amdgcn_sched_barrier(0);
        int64_t opaque = 0;
        asm volatile("" : "+s"(opaque));
        const double one = add_one ? 1.0 : 0.0;
        // the squared norms of this lane's four columns, once per tile (round 6: read per element they were 64 dependent loads per
        // lane and tile between the accumulators and the exponentials: profiles/r06/gram_epilogue_variants.txt)
        double bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = KIND == BQ_KERNEL_RBF ? P.b2[bcol + bq_acc_col(j)] : 0.0;
        double wj[4];   // ... and the product's input over them (zero beyond n: padded to the panel pitch)
#pragma unroll
        for (int j = 0; j < 4; ++j) wj[j] = w[bcol + bq_acc_col(j)];
#if BQ_STREAM_FOLD
        // MEASURED VARIANT, not the product (build with -DBQ_STREAM_FOLD=1): the argument of the exponential as
        // fma(2 gamma, x.y, c_i + c_j), c = -gamma |x|^2 per row / per column, clamped from above at 0 — three vector instructions per
        // element instead of five, another rounding of the distances than sklearn's (-2 x.y + |x|^2 + |y|^2, clamped at 0).
        // profiles/r06/gram_epilogue_variants.txt has what it buys.
        const double g2 = 2.0 * P.gamma;
        double cj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cj[j] = -P.gamma * bj[j];
#endif
        auto epilogue = [&](auto on_diag, auto deg) {
            double col[4] = {0.0, 0.0, 0.0, 0.0};
            double part[16];   // one per accumulator row of this lane; folded over the 16 lanes of a row after the maps
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t gi = opaque + arow + bq_acc_row(i, v);
                    const double ai = KIND == BQ_KERNEL_RBF ? rowsq[wv][bq_acc_row64(i, v)] : 0.0;
                    const double wi = wrow[wv][bq_acc_row64(i, v)];   // zero beyond n
                    double pr = 0.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t gj = bcol + bq_acc_col(j);
                        const double dot = acc[i][j][v];
                        double kv;
                        if (KIND == BQ_KERNEL_RBF) {
#if BQ_STREAM_FOLD
                            double arg = fma(g2, dot, -P.gamma * ai + cj[j]);   // (-gamma a_i: once per row, hoisted by the compiler)
                            arg = fmin(arg, 0.0);
                            if (decltype(on_diag)::value && gi == gj) arg = 0.0;
                            kv = bq_exp(arg);
#else
                            double dist = -2.0 * dot;
                            dist += ai;
                            dist += bj[j];
                            dist = fmax(dist, 0.0);
                            if (decltype(on_diag)::value && gi == gj) dist = 0.0;
                            kv = bq_exp(-P.gamma * dist);
#endif
                        } else if (KIND == BQ_KERNEL_POLY) {
                            kv = bq_poly_map<decltype(deg)::value>(P.gamma * dot + P.coef0, P.degree);
                        } else if (KIND == BQ_KERNEL_SIGMOID) {
                            kv = tanh(P.gamma * dot + P.coef0);
                        } else {
                            kv = dot;
                        }
                        kv += one;
                        pr = fma(kv, wj[j], pr);
                        if (!decltype(on_diag)::value) col[j] = fma(kv, wi, col[j]);   // the diagonal tile is used once
                        if ((KIND == BQ_KERNEL_RBF || (KIND == BQ_KERNEL_POLY && decltype(deg)::value == 0)) && ((j + 1) % STREAM_SYM_EXP_ILP == 0))
                            __builtin_amdgcn_sched_barrier(0);
                    }
                    part[4 * i + v] = pr;
                }
            }
            // one straight block up to here (a branch per row cut it into 16, with values carried — and spilled — between them)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double q = part[r];
                q += __shfl_xor(q, 1, 64);
                q += __shfl_xor(q, 2, 64);
                q += __shfl_xor(q, 4, 64);
                q += __shfl_xor(q, 8, 64);
                part[r] = q;
            }
            if (!decltype(on_diag)::value) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double c = col[j];
                    c += __shfl_xor(c, 16, 64);
                    c += __shfl_xor(c, 32, 64);
                    col[j] = c;
                }
            }
            if (ccol == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rowsum[wv][bq_acc_row64(r >> 2, r & 3)] += part[r];
            }
            if (!decltype(on_diag)::value && crow == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) colsum[wr][wc * 64 + bq_acc_col64(j)] = col[j];
            }
        };
        if (I == J)
            epilogue(std::true_type{}, std::integral_constant<int, 0>{});
        else if (KIND == BQ_KERNEL_POLY && P.degree == 3)
            epilogue(std::false_type{}, std::integral_constant<int, 3>{});
        else if (KIND == BQ_KERNEL_POLY && P.degree == 2)
            epilogue(std::false_type{}, std::integral_constant<int, 2>{});
        else
            epilogue(std::false_type{}, std::integral_constant<int, 0>{});
        __syncthreads();   // colsum complete; the next tile's prologue may refill the operand buffers
        if (I != J && threadIdx.x < 128)
            slab[(((int64_t)I * (I - 1) - t0 * (t0 - 1)) / 2 + J) * 128 + threadIdx.x] = colsum[0][threadIdx.x] + colsum[1][threadIdx.x];
    }
    __syncthreads();
    if (wc == 0) SU[(int64_t)slot * 128 + wr * 64 + lane] = rowsum[wv][lane] + rowsum[wv + 1][lane];
}

// y_s over output tile t, element e: the units of row t in range order (if t lies in the segment's 128-row tiles [lo, hi)),
// then the column sums of the tiles (I, t), I in [lo, hi), I > t, ascending
struct stream_sym_geom {
    int64_t T, t0, U, kmax, n, len;
};
__device__ __forceinline__ double stream_seg_partial(const double *__restrict__ SU, const double *__restrict__ slab,
                                                     const stream_sym_geom &g, int64_t t, int e, int64_t lo, int64_t hi) {
    double a = 0.0;
    if (t >= lo && t < hi)
        for (int64_t k = 0; k * g.U <= t; ++k) a += SU[((t - g.t0) * g.kmax + k) * 128 + e];
    for (int64_t I = (t + 1 > lo ? t + 1 : lo); I < hi; ++I) a += slab[((I * (I - 1) - g.t0 * (g.t0 - 1)) / 2 + t) * 128 + e];
    return a;
}
// one rank: the segment vectors are added in segment order right here
__global__ __launch_bounds__(128) void stream_sym_reduce_kernel(const double *__restrict__ SU, const double *__restrict__ slab,
                                                                stream_sym_geom g, bq_seg_table tab, double *__restrict__ out,
                                                                const int *done) {
    if (done != nullptr && *done) return;
    const int64_t t = blockIdx.x;
    const int e = threadIdx.x;
    double acc = 0.0;
    for (int s = tab.lo; s < tab.hi; ++s) {
        const int64_t lo = 2 * tab.cut[s], hi = 2 * tab.cut[s + 1] < g.T ? 2 * tab.cut[s + 1] : g.T;
        acc += stream_seg_partial(SU, slab, g, t, e, lo, hi);
    }
    out[t * GT + e] = t * GT + e < g.n ? acc : 0.0;
}
// several ranks: this rank's segment vectors, each to its slot of the gathered buffer
__global__ __launch_bounds__(128) void stream_sym_reduce_seg_kernel(const double *__restrict__ SU, const double *__restrict__ slab,
                                                                    stream_sym_geom g, bq_seg_table tab, double *__restrict__ gath,
                                                                    const int *done) {
    if (done != nullptr && *done) return;
    const int64_t t = blockIdx.x;
    const int s = tab.lo + (int)blockIdx.y;
    const int e = threadIdx.x;
    const int64_t lo = 2 * tab.cut[s], hi = 2 * tab.cut[s + 1] < g.T ? 2 * tab.cut[s + 1] : g.T;
    const double v = stream_seg_partial(SU, slab, g, t, e, lo, hi);
    gath[(int64_t)tab.slot[s] * g.len + t * GT + e] = t * GT + e < g.n ? v : 0.0;
}

int bq_stream_prepare(bq_ctx *ctx, const double *Xdev, int64_t n, int64_t d, int64_t r0, int64_t r1, void **out) {
    bq_stream_images *st = new bq_stream_images();
    int rc = make_image(ctx, Xdev, n, d, &st->img);
    if (rc != BQ_OK) {
        delete st;
        return rc;
    }
    const int64_t tiles_n = (n + GT - 1) / GT;
    {   // rows [r0, r1) are whole 256-row tile rows of canonical segments (r1 may exceed n)
        const int64_t T = tiles_n, total = T * (T + 1) / 2;
        const int64_t t0 = r0 / GT, t1 = (r1 + GT - 1) / GT < T ? (r1 + GT - 1) / GT : T;
        // units sized for ~24 rounds of workgroups on each of 8 ranks, from n alone (the association of the row sums must not
        // depend on the rank count)
        const int64_t target = 24 * 512 * 8;
        int64_t U = total / target > 0 ? (total + target - 1) / target : 1;
        {   // hook stream_unit: several tiles per unit at small n (tests)
            double hv = 0.0;
            if (bq_hook("stream_unit", &hv) && hv >= 1.0) U = (int64_t)hv;
        }
        const int64_t kmax = (T + U - 1) / U;
        // launch order: column range by column range, rows descending inside a range — the workgroups in flight walk the SAME
        // column tiles at about the same time, like the row-block form (measured equal to row-major unit order at n = 100 000:
        // the operand images live in the last-level cache either way)
        std::vector<int> unit;
        for (int64_t k = 0; k < kmax; ++k)
            for (int64_t t = t1 - 1; t >= t0 && t >= k * U; --t) {
                unit.push_back((int)t);
                unit.push_back((int)(k * U));
                unit.push_back((int)(k * U + U < t + 1 ? k * U + U : t + 1));
                unit.push_back((int)((t - t0) * kmax + k));
            }
        st->T = T;
        st->t0 = t0;
        st->t1 = t1;
        st->U = U;
        st->kmax = kmax;
        st->nunits = (int64_t)unit.size() / 4;
        const int64_t rows = t1 > t0 ? t1 - t0 : 0;
        const int64_t nslab = rows > 0 ? (t1 * (t1 - 1) - t0 * (t0 - 1)) / 2 : 0;
        hipError_t e = hipMalloc(&st->unit, sizeof(int) * (unit.size() + 4));
        if (e == hipSuccess) e = hipMalloc(&st->SU, sizeof(double) * (rows * kmax + 1) * 128);
        if (e == hipSuccess) e = hipMalloc(&st->slab, sizeof(double) * (nslab + 1) * 128);
        if (e == hipSuccess && !unit.empty()) e = hipMemcpy(st->unit, unit.data(), sizeof(int) * unit.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            bq_set_error("cannot allocate the symmetric streamed-product scratch: %s", hipGetErrorString(e));
            bq_stream_free(st);
            return BQ_ERR_NOMEM;
        }
    }
    *out = st;
    return BQ_OK;
}

void bq_stream_free(void *h) {
    if (!h) return;
    bq_stream_images *st = (bq_stream_images *)h;
    free_image(&st->img);
    if (st->unit) hipFree(st->unit);
    if (st->SU) hipFree(st->SU);
    if (st->slab) hipFree(st->slab);
    delete st;
}

// mode 0: out (nb * 256) = the sum of this rank's segment vectors in segment order; mode 1: each of this
// rank's segment vectors to its slot of the gathered buffer `out` (slots of nb * 256)
int bq_stream_sym_product(bq_ctx *ctx, void *h, int64_t n, int64_t nb, const bq_seg_table &tab, int kernel, double gamma, double coef0,
                          int degree, bool add_one, const double *w, double *out, int mode, const int *done) {
    bq_stream_images *st = (bq_stream_images *)h;
    BQ_ARG(kernel != BQ_KERNEL_LAPLACIAN, "the streamed mode is built for the inner-product kernels (linear, poly, rbf, sigmoid)");
    gram_params P;
    P.lower_only = 0;
    P.At = P.Bt = st->img.At;
    P.a2 = P.b2 = st->img.a2;
    P.m = n;
    P.n = n;
    P.mp = P.np = st->img.mp;
    P.dp = st->img.dp;
    P.arow0 = 0;
    P.arow1 = n;
    P.same = 1;
    P.kernel = kernel;
    P.degree = degree;
    P.gamma = gamma;
    P.coef0 = coef0;
    P.ld = 0;
    P.ntiles = 0;
    P.rect_rb = P.rect_sb = P.tiles_m = P.strips = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_MATVEC, &e0, &e1));
    int *skip = nullptr, skip_seq = 0;
    if (done != nullptr) bq_prof_skip_arg(ctx, e0, &skip, &skip_seq);
    const unsigned nu = (unsigned)st->nunits;
    if (nu > 0) {
        switch (kernel) {
            case BQ_KERNEL_RBF:
                gram_stream_sym_kernel<BQ_KERNEL_RBF><<<nu, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, st->unit, st->SU, st->slab, st->t0, done, skip, skip_seq);
                break;
            case BQ_KERNEL_POLY:
                gram_stream_sym_kernel<BQ_KERNEL_POLY><<<nu, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, st->unit, st->SU, st->slab, st->t0, done, skip, skip_seq);
                break;
            case BQ_KERNEL_SIGMOID:
                gram_stream_sym_kernel<BQ_KERNEL_SIGMOID><<<nu, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, st->unit, st->SU, st->slab, st->t0, done, skip, skip_seq);
                break;
            default:
                gram_stream_sym_kernel<BQ_KERNEL_LINEAR><<<nu, 256, 0, ctx->stream>>>(P, w, add_one ? 1 : 0, st->unit, st->SU, st->slab, st->t0, done, skip, skip_seq);
                break;
        }
    }
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_prof_end(ctx, BQ_PROF_MATVEC, e0, e1));
    stream_sym_geom g{st->T, st->t0, st->U, st->kmax, n, nb * BQ_SYM_TILE};
    if (mode == 0)
        stream_sym_reduce_kernel<<<(unsigned)st->T, 128, 0, ctx->stream>>>(st->SU, st->slab, g, tab, out, done);
    else if (tab.hi > tab.lo)
        stream_sym_reduce_seg_kernel<<<dim3((unsigned)st->T, (unsigned)(tab.hi - tab.lo)), 128, 0, ctx->stream>>>(st->SU, st->slab, g, tab, out, done);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
