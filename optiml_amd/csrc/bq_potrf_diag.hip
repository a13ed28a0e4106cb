// Diagonal-block kernel of the blocked Cholesky: factor a 128 x 128 SPD block and invert its triangular factor,
// all inside one workgroup's LDS (no trip to HBM between its internal phases).
//
// Layout: one 128 x 129 fp64 square M in LDS.  Lower triangle + diagonal = A, then L.  The strict upper triangle
// receives X = L^-1 transposed (X[p][q], p > q, lives at M[q][p]) — which is exactly the k-major image
// LinvT[k][j] = Linv[j][k] that the TRSM-as-GEMM tile kernel and the triangular solves consume; diag(X) = 1/diag(L)
// sits in a side vector.  A 3 x 32 x 33 scratch area holds panel results / partial products.
//
// Algorithm (32-wide sub-blocks, 8 waves).  The code is kept COMPACT (runtime loops, small unroll factors): this
// kernel runs every phase once per launch, so a fully unrolled body is bound by instruction fetch, not by math
// (measured: 240 us unrolled).
//   for each sub-block column: (1) wave 0 factors the 32 x 32 diagonal block and inverts its factor with rows held in
//   registers and cross-row operands fetched by v_readlane (no LDS round trip on the sequential chain); (2) the rows below get
//   L_ik = A_ik X_kk^T; (3) the trailing lower triangle gets A_ij -= L_ik L_jk^T in 4 x 4 register micro-tiles.
//   Then the off-diagonal blocks of X by block forward substitution along the block sub-diagonals:
//   X_ic = -X_ii (sum_j L_ij X_jc).
// A non-positive (or NaN) pivot is reported through *info (1-based global index), like LAPACK's potrf.
#include "bq_common.h"
#include <type_traits>

constexpr int PB = 128;        // block order
constexpr int PS = 32;         // sub-block
constexpr int PL = PB + 1;     // LDS pitch
constexpr int PW = PS + 1;     // scratch pitch
constexpr int PT = 512;        // threads per workgroup
constexpr size_t POTRF_LDS = sizeof(double) * (PB * PL + PB + 3 * PS * PW) + 16;

// order the LDS traffic of ONE wave: all lanes' earlier LDS writes are visible to later reads of the same wave
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v_readlane of a double (lane index uniform)
__device__ __forceinline__ double rdlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(PT) void potrf_diag128_kernel(double *__restrict__ H, int64_t ldh, int64_t k0,
                                                           double *__restrict__ LinvT, int *__restrict__ info, long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *M = smem;
    double *dinv = M + PB * PL;
    double *W = dinv + PB;
    int *flag = reinterpret_cast<int *>(W + 3 * PS * PW);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef BQ_DIAG_STAMPS   // diagnostic build only: phase boundaries of launch #3 -> stderr (100 MHz wall clock)
    int sidx = 0;
#define STAMP()                                                          \
    do {                                                                 \
        __syncthreads();                                                 \
        if (stamps && tid == 0) stamps[sidx] = wall_clock64();           \
        ++sidx;                                                          \
    } while (0)
#else
#define STAMP() do { } while (0)
#endif
    STAMP();
    for (int e = tid; e < PB * PB; e += PT) {
        const int i = e >> 7, j = e & 127;
        M[i * PL + j] = (j <= i) ? H[(k0 + i) * ldh + k0 + j] : 0.0;
    }
    if (tid == 0) *flag = 0;
    __syncthreads();
    STAMP();

    for (int kb = 0; kb < PB / PS; ++kb) {
        const int o = kb * PS;
        double *D = M + o * PL + o;   // the 32 x 32 diagonal block, D[i * PL + j]
        // ---- (1) diagonal block: factor, then invert the factor; wave 0 only, wave-synchronous ---------------------
        if (wv == 0) {
            // Lane i (both half-waves compute the same thing; only lanes < 32 store) owns ROW i of the block.
            // Factor: left-looking over panels of 8 columns held in registers; every cross-row operand is another
            // lane's register, fetched with v_readlane (uniform lane index), so the sequential chain has no LDS round
            // trips.  Rows above the diagonal carry harmless finite garbage and are never stored.
            const int i = lane & 31;
            int bad = 0;
            double myri = 1.0;   // 1 / L[i][i]
#pragma unroll 1
            for (int pb = 0; pb < PS / 8; ++pb) {
                const int c0 = 8 * pb;
                double a[8];
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) a[cc] = D[i * PL + c0 + cc];
                for (int k = 0; k < c0; ++k) {
                    const double lik = D[i * PL + k];   // L[c0 + cc][k] is lane (c0 + cc)'s lik
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) a[cc] = fma(-lik, rdlane(lik, c0 + cc), a[cc]);
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = c0 + jj;
                    double d = rdlane(a[jj], j);
                    if (!(d > 0.0)) {   // uniform
                        if (bad == 0) bad = o + j + 1;
                        d = 1.0;
                    }
                    double ri = rsqrt(d);
                    ri = ri * fma(-0.5 * d * ri, ri, 1.5);   // Newton step: 1/sqrt(d) to full fp64 accuracy
                    const double l = (i == j) ? d * ri : a[jj] * ri;
                    a[jj] = l;
                    if (i == j) myri = ri;
#pragma unroll
                    for (int cc = jj + 1; cc < 8; ++cc) a[cc] = fma(-l, rdlane(l, c0 + cc), a[cc]);
                }
                if (lane < 32) {
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc)
                        if (i >= c0 + cc) D[i * PL + c0 + cc] = a[cc];
                }
                wave_lds_fence();
            }
            if (lane < 32) dinv[o + i] = myri;
            // Inverse of the factor, X = L^-1: lane i owns row i of Y with X[i][:] = myri * Y[i][:] and
            // Y[i][:] = e_i - sum_{k < i} L[i][k] myri_k Y[k][:]; step j subtracts row j (final by then) from the rows
            // below it.  Y[j][c] = 0 for c > j bounds the column range of each group of 8 steps.
            double y[PS];
#pragma unroll
            for (int c = 0; c < PS; ++c) y[c] = (c == i) ? 1.0 : 0.0;
            auto steps = [&](auto tag) {
                constexpr int J0 = decltype(tag)::value;
#pragma unroll 1
                for (int j = J0; j < J0 + 8; ++j) {
                    const double m = (i > j) ? D[i * PL + j] * rdlane(myri, j) : 0.0;
#pragma unroll
                    for (int c = 0; c < J0 + 8; ++c) y[c] = fma(-m, rdlane(y[c], j), y[c]);
                }
            };
            steps(std::integral_constant<int, 0>{});
            steps(std::integral_constant<int, 8>{});
            steps(std::integral_constant<int, 16>{});
            steps(std::integral_constant<int, 24>{});
            wave_lds_fence();
            if (lane < 32) {
#pragma unroll
                for (int c = 0; c < PS - 1; ++c)
                    if (c < i) D[c * PL + i] = y[c] * myri;   // X[i][c] at its transposed home (row c, upper)
            }
            if (bad != 0 && lane == 0) *flag = bad;
        }
        __syncthreads();
        STAMP();
        if (*flag != 0) break;   // uniform
        const int R0 = o + PS;   // first row below the diagonal block
        const int m = PB - R0;   // rows below
        if (m == 0) break;
        // ---- (2) panel: L[r][o + j] = sum_{t <= j} A[r][o + t] * X[o + j][o + t], X(j, t) at D[t][j] / dinv -------
        {
            const int r = tid >> 2, h = tid & 3;    // four threads per row, 8 outputs each, staged through W
            if (r < m) {
                const double *arow = M + (R0 + r) * PL + o;
                double acc[PS / 4];
#pragma unroll
                for (int jj = 0; jj < PS / 4; ++jj) acc[jj] = 0.0;
                const int jbase = h * (PS / 4);
#pragma unroll 2
                for (int t = 0; t < jbase + PS / 4; ++t) {
                    const double at = arow[t];
                    const double dt = dinv[o + t];
#pragma unroll
                    for (int jj = 0; jj < PS / 4; ++jj) {
                        const int j = jbase + jj;
                        const double xv = (t == j) ? dt : D[t * PL + j];
                        if (t <= j) acc[jj] = fma(at, xv, acc[jj]);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < PS / 4; ++jj) W[r * PW + jbase + jj] = acc[jj];
            }
            __syncthreads();
            for (int e = tid; e < m * PS; e += PT) M[(R0 + e / PS) * PL + o + (e % PS)] = W[(e / PS) * PW + (e % PS)];
        }
        __syncthreads();
        STAMP();
        // ---- (3) trailing update in 4 x 4 micro-tiles of the lower triangle ---------------------------------------
        {
            const int mt = m / 4;
            const int ntiles = mt * (mt + 1) / 2;
            for (int idx = tid; idx < ntiles; idx += PT) {
                int tr = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
                while ((tr + 1) * (tr + 2) / 2 <= idx) ++tr;
                while (tr * (tr + 1) / 2 > idx) --tr;
                const int tc = idx - tr * (tr + 1) / 2;
                const int rr = R0 + 4 * tr, cc = R0 + 4 * tc;
                double acc[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
#pragma unroll 4
                for (int t = 0; t < PS; ++t) {
                    double ar[4], bc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ar[u] = M[(rr + u) * PL + o + t];
                        bc[u] = M[(cc + u) * PL + o + t];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[u][v] = fma(ar[u], bc[v], acc[u][v]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (cc + v <= rr + u) M[(rr + u) * PL + cc + v] -= acc[u][v];
            }
        }
        __syncthreads();
        STAMP();
    }

    const int bad = *flag;
    if (bad != 0) {
        if (tid == 0 && *info == 0) *info = (int)(k0 + bad);
        return;
    }

    // ---- off-diagonal blocks of X = L^-1 along the block sub-diagonals ------------------------------------------------
    // X(p, q) for p > q is M[q][p]; X(p, p) = dinv[p]; X(p, q) = 0 for p < q.
    for (int dd = 1; dd < PB / PS; ++dd) {
        const int nblk = PB / PS - dd;   // targets (i = c + dd, c), c = 0 .. nblk-1
        // stage A: W_b = sum_{j = c}^{i-1} L_ij X_jc
        for (int idx = tid; idx < nblk * PS * PS; idx += PT) {
            const int b = idx / (PS * PS), u = (idx / PS) % PS, v = idx % PS;
            const int c = b, i = b + dd;
            double s0 = 0.0, s1 = 0.0;
            for (int j = c; j < i; ++j) {
                const double *Lrow = M + (PS * i + u) * PL + PS * j;
                const double *Xcol = M + (PS * c + v) * PL + PS * j;   // X[32 j + t][32 c + v], t = 0..31
                if (j == c) {
                    // diagonal block of X: lower triangular, rows t >= v
                    s0 = fma(Lrow[v], dinv[PS * c + v], s0);
#pragma unroll 2
                    for (int t = v + 1; t < PS; ++t) s1 = fma(Lrow[t], Xcol[t], s1);
                } else {
#pragma unroll 4
                    for (int t = 0; t < PS; t += 2) {
                        s0 = fma(Lrow[t], Xcol[t], s0);
                        s1 = fma(Lrow[t + 1], Xcol[t + 1], s1);
                    }
                }
            }
            W[b * PS * PW + u * PW + v] = s0 + s1;
        }
        __syncthreads();
        // stage B: X_ic = -X_ii W_b   (X_ii lower triangular: t <= u)
        for (int idx = tid; idx < nblk * PS * PS; idx += PT) {
            const int b = idx / (PS * PS), u = (idx / PS) % PS, v = idx % PS;
            const int c = b, i = b + dd;
            const double *Wb = W + b * PS * PW + v;
            const double *Xii = M + PS * i * PL + PS * i + u;   // X_ii[u][t] (t < u) at M[32 i + t][32 i + u]
            double s0 = dinv[PS * i + u] * Wb[u * PW], s1 = 0.0;
#pragma unroll 2
            for (int t = 0; t < u; ++t) s1 = fma(Xii[t * PL], Wb[t * PW], s1);
            M[(PS * c + v) * PL + PS * i + u] = -(s0 + s1);   // X[32 i + u][32 c + v] at its transposed home
        }
        __syncthreads();
        STAMP();
    }

    for (int e = tid; e < PB * PB; e += PT) {
        const int i = e >> 7, j = e & 127;
        if (j <= i) H[(k0 + i) * ldh + k0 + j] = M[i * PL + j];
        LinvT[e] = (i < j) ? M[i * PL + j] : ((i == j) ? dinv[i] : 0.0);   // LinvT[k = i][j] = Linv[j][k]
    }
    STAMP();
}

int bq_potrf_diag_setup() {
    BQ_HIP(hipFuncSetAttribute((const void *)potrf_diag128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)POTRF_LDS));
    return BQ_OK;
}

void bq_launch_potrf_diag(hipStream_t st, double *H, int64_t ldh, int64_t k0, double *LinvT, int *info) {
#ifdef BQ_DIAG_STAMPS
    static long long *dstamps = nullptr;
    static int calls = 0;
    if (!dstamps) hipMalloc(&dstamps, 64 * sizeof(long long));
    potrf_diag128_kernel<<<1, PT, POTRF_LDS, st>>>(H, ldh, k0, LinvT, info, calls == 3 ? dstamps : nullptr);
    if (calls == 3) {
        long long h[64];
        hipStreamSynchronize(st);
        hipMemcpy(h, dstamps, sizeof(h), hipMemcpyDeviceToHost);
        for (int i = 1; i < 17; ++i) fprintf(stderr, "stamp %d: +%.2f us\n", i, (h[i] - h[i - 1]) / 100.0);
    }
    ++calls;
#else
    potrf_diag128_kernel<<<1, PT, POTRF_LDS, st>>>(H, ldh, k0, LinvT, info, nullptr);
#endif
}
