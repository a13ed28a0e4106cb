// Dense fp64 Cholesky (lower, in place, row-major) + triangular solves for the interior-point Newton system
// and the active-set restricted system.  Replaces scipy's cho_factor / cho_solve at
// optiml/opti/constrained/interior_point.py:235 and active_set.py:141.
//
// Blocked right-looking factorisation with 128-wide block columns:
//   1. potrf_diag   one workgroup factors the 128x128 diagonal block in LDS (packed lower triangle, 66 KB) and
//                   inverts the triangular factor next to it (another 66 KB); the inverse is kept per block (it
//                   turns both TRSM and the two triangular solves into matrix products);
//   2. trsm as GEMM the block column below the diagonal, X = A_ik L_kk^-T, on the shared fp64-MFMA 128x128 tile
//                   kernel (operands as k-major images so global loads are coalesced);
//   3. syrk         the trailing lower triangle, A_ij -= X_i X_j^T, one workgroup per 128x128 tile on the same
//                   MFMA kernel — n^3/3 flops, the MFMA-bound bulk of every interior-point iteration.
// The solves walk the block columns with one row-panel product + one 128x128 product per block (forward
// left-looking, backward right-looking), all with fixed reduction orders.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "bq_chol.h"
#include "bq_qelem.h"
#include "bq_mfma_tile.h"

constexpr int NB = 128;
// 1. diagonal block (factor + invert): bq_potrf_diag.hip
int bq_potrf_diag_setup();
void bq_launch_potrf_diag(hipStream_t st, double *H, int64_t ldh, int64_t k0, double *LinvT, int *info, const double *thr);

// ---------------------------------------------------------------------------------------------
// 2./3. block column: TRSM as GEMM (which also leaves the k-major image of its result), SYRK on the trailing triangle
// ---------------------------------------------------------------------------------------------
// X = A_ik * Linv_kk^T for the row tiles below the diagonal block.  A is read straight from H (row-major operand of the
// tile kernel: no transposing pre-pass); X is written back in place AND as the k-major image Wt[c][row] that the trailing
// updates consume (no transposing post-pass either).
__global__ __launch_bounds__(256, 2) void trsm_gemm_kernel(double *__restrict__ H, int64_t ldh, int64_t k0, int64_t i0,
                                                           double *__restrict__ Wt, const double *__restrict__ LinvT) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    const int64_t arow = i0 + (int64_t)blockIdx.x * NB;
    bq_d4 acc[4][4];
    bq_tile_zero(acc);
    bq_mfma_tile_128<false, true>(H + k0, ldh, arow, LinvT, NB, 0, NB, sm, acc);
    bq_tile_foreach(acc, [&](int r, int c, double v) {
        H[(arow + r) * ldh + k0 + c] = v;
        Wt[(int64_t)c * ldh + arow + r] = v;
    });
}

// A_ij -= X_i X_j^T over the lower-triangular tiles (ti >= tj) of the trailing matrix starting at i0; X is the k-major
// image Wt with kdim rows (128, or 256 when two factored block columns are applied in one pass: twice the flops per
// byte of C-tile traffic)
// Tile order: the 128 x 128 tiles are grouped into 8 x 8 super-tiles and every super-tile is dealt to ONE XCD (workgroups
// go round-robin over the 8 XCDs by block id).  An XCD runs 64 workgroups at a time (32 CUs x 2), i.e. one super-tile:
// its 8 row slices + 8 column slices of the image (16 x 256 KiB at K = 256) fit the XCD's 4 MiB L2 and are each reused
// 8 times, instead of every tile streaming its own column slice from HBM/MALL (row-major order: 64 different column
// slices in flight per XCD).  Blocks of a diagonal super-tile that fall above the diagonal exit at once.
__global__ __launch_bounds__(256, 2) void syrk_kernel(double *__restrict__ H, int64_t ldh, int64_t i0,
                                                      const double *__restrict__ Wt, int kdim, int64_t T, int64_t nsuper) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    const int64_t bid = blockIdx.x, slot = bid / 8;
    int64_t ti, tj;
    if (nsuper == 0) {   // small grids (a few rounds at most): plain row-major triangle, every XCD equally loaded
        if (bid >= T * (T + 1) / 2) return;
        ti = (int64_t)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
        while (ti * (ti + 1) / 2 > bid) --ti;
        tj = bid - ti * (ti + 1) / 2;
    } else {
        const int64_t sidx = (slot / 64) * 8 + bid % 8;
        if (sidx >= nsuper) return;
        int64_t si = (int64_t)((sqrt(8.0 * (double)sidx + 1.0) - 1.0) * 0.5);
        while ((si + 1) * (si + 2) / 2 <= sidx) ++si;
        while (si * (si + 1) / 2 > sidx) --si;
        const int64_t sj = sidx - si * (si + 1) / 2;
        const int local = (int)(slot % 64);
        ti = 8 * si + local / 8;
        tj = 8 * sj + local % 8;
        if (ti >= T || tj > ti) return;
    }
    const int64_t arow = i0 + ti * NB, bcol = i0 + tj * NB;
    bq_d4 acc[4][4];
    double *Ct = H + arow * ldh + bcol;
    bq_tile_zero(acc);   // acc = X_i X_j^T, then C -= acc: the C tile is read after the loop (bq_tile_sub_store)
    bq_mfma_tile_128(Wt, ldh, arow, Wt, ldh, bcol, kdim, sm, acc);
    bq_tile_sub_store(acc, Ct, ldh);
}

// the same update restricted to the first block column of the trailing matrix (tiles (ti, 0)): makes the next block
// column current so that it can be factored before the wide update runs
__global__ __launch_bounds__(256, 2) void syrk_col_kernel(double *__restrict__ H, int64_t ldh, int64_t i0,
                                                          const double *__restrict__ Wt) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    const int64_t arow = i0 + (int64_t)blockIdx.x * NB;
    bq_d4 acc[4][4];
    double *Ct = H + arow * ldh + i0;
    bq_tile_zero(acc);   // acc = X_i X_j^T, then C -= acc
    bq_mfma_tile_128(Wt, ldh, arow, Wt, ldh, i0, NB, sm, acc);
    bq_tile_sub_store(acc, Ct, ldh);
}

// the update restricted to the first NC block columns of the trailing matrix starting at r0 (tiles (ti >= c, c), c < NC):
// the part the next narrow work waits for.  NC = 2 with the K = 256 images of one pass, NC = 4 with the K = 512 images of a
// super-pass.
__global__ __launch_bounds__(256, 2) void syrk_head_kernel(double *__restrict__ H, int64_t ldh, int64_t r0,
                                                           const double *__restrict__ Wt, int kdim, int64_t T, int nc) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    int64_t b = blockIdx.x, tj = 0;
    while (tj < nc - 1 && b >= T - tj) {   // column tj holds T - tj tiles
        b -= T - tj;
        ++tj;
    }
    const int64_t ti = b + tj;
    if (ti >= T) return;
    const int64_t arow = r0 + ti * NB, bcol = r0 + tj * NB;
    bq_d4 acc[4][4];
    double *Ct = H + arow * ldh + bcol;
    bq_tile_zero(acc);   // acc = X_i X_j^T, then C -= acc: the C tile is read after the loop (bq_tile_sub_store)
    bq_mfma_tile_128(Wt, ldh, arow, Wt, ldh, bcol, kdim, sm, acc);
    bq_tile_sub_store(acc, Ct, ldh);
}

// ---------------------------------------------------------------------------------------------
// triangular solves with the blocked factor (rhs overwritten by the solution)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wsum_c(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void diag_mv_body(const double *__restrict__ LinvT, int transpose, const double *in,
                                             double *out);
__device__ __forceinline__ bool chol_last_block(unsigned int *ticket);

// part[s][r] = sum_{c in slice s of [c0, k0)} L[k0 + r][c] * rhs[c]      (workgroup (r, s): row r, column slice s)
// ... and the last workgroup to finish forms tmp = rhs[k0 : k0+128) - sum_s part[s] (fixed order) and applies the inverse
// of the diagonal block: rhs[k0 : k0+128) = Linv_kk tmp.  Long rows are cut into up to four slices so that no single
// CU has to pull a whole 400 KB row (n = 50 000) by itself.
__global__ __launch_bounds__(256) void fwd_panel_kernel(const double *__restrict__ H, int64_t ldh, int64_t k0, double *rhs,
                                                        double *part, const double *__restrict__ LinvT,
                                                        unsigned int *ticket, int64_t c0, int64_t slice) {
    __shared__ double red[4];
    __shared__ double comb[NB];
    const int r = blockIdx.x, tid = threadIdx.x;
    const double *row = H + (k0 + r) * ldh;
    const int64_t lo = c0 + (int64_t)blockIdx.y * slice, hi = lo + slice < k0 ? lo + slice : k0;
    double a = 0.0;
    // c0: the right-hand side is known to vanish before column c0 (a multiple of 128)
    for (int64_t c = lo + 2 * tid; c < hi; c += 512) {  // slices are multiples of 512, k0 of 128: pairs never straddle
        const bq_d2 l = *reinterpret_cast<const bq_d2 *>(row + c);
        const bq_d2 y = *reinterpret_cast<const bq_d2 *>(rhs + c);
        a = fma(l.y, y.y, fma(l.x, y.x, a));
    }
    a = wsum_c(a);
    if ((tid & 63) == 0) red[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) part[(int64_t)blockIdx.y * NB + r] = ((red[0] + red[1]) + red[2]) + red[3];
    if (chol_last_block(ticket)) {
        if (tid < NB) {
            double sum = part[tid];
            for (unsigned int sl = 1; sl < gridDim.y; ++sl) sum += part[(int64_t)sl * NB + tid];
            comb[tid] = rhs[k0 + tid] - sum;
        }
        __syncthreads();
        diag_mv_body(LinvT, 0, comb, rhs + k0);
    }
}

// out[i] = sum_j M[i][j] in[j] with M = Linv (transpose == 0) or Linv^T (transpose == 1); LinvT[k][j] = Linv[j][k].
// Two threads per output element split the 128-long sum; reads of LinvT are coalesced in the non-transposed case
// (consecutive i) and row-contiguous per thread in the transposed one (L2-resident 128 KB).
__device__ __forceinline__ void diag_mv_body(const double *__restrict__ LinvT, int transpose, const double *in,
                                             double *out) {
    __shared__ double v[NB];
    __shared__ double half[2][NB];
    const int i = threadIdx.x & 127, h = threadIdx.x >> 7;
    if (threadIdx.x < NB) v[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    // LinvT is a full 128 x 128 square (zeros outside the triangle), so the loops have fixed trip counts and are unrolled
    // with independent accumulators: the 64 loads of a thread are in flight together (the triangular bounds made this a
    // chain of dependent-latency loads: 24 us per call).
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int j0 = h * 64;
    if (transpose) {   // Linv^T[i][j] = LinvT[i][j]: row i, contiguous
        const bq_d2 *row = reinterpret_cast<const bq_d2 *>(LinvT + i * NB + j0);
#pragma unroll 8
        for (int jj = 0; jj < 32; jj += 2) {
            const bq_d2 l0 = row[jj], l1 = row[jj + 1];
            acc[0] = fma(l0.x, v[j0 + 2 * jj], acc[0]);
            acc[1] = fma(l0.y, v[j0 + 2 * jj + 1], acc[1]);
            acc[2] = fma(l1.x, v[j0 + 2 * jj + 2], acc[2]);
            acc[3] = fma(l1.y, v[j0 + 2 * jj + 3], acc[3]);
        }
    } else {           // Linv[i][j] = LinvT[j][i]: column i, coalesced across the threads
#pragma unroll 16
        for (int jj = 0; jj < 64; jj += 4) {
            acc[0] = fma(LinvT[(j0 + jj) * NB + i], v[j0 + jj], acc[0]);
            acc[1] = fma(LinvT[(j0 + jj + 1) * NB + i], v[j0 + jj + 1], acc[1]);
            acc[2] = fma(LinvT[(j0 + jj + 2) * NB + i], v[j0 + jj + 2], acc[2]);
            acc[3] = fma(LinvT[(j0 + jj + 3) * NB + i], v[j0 + jj + 3], acc[3]);
        }
    }
    const double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    half[h][i] = s;
    __syncthreads();
    if (threadIdx.x < NB) out[i] = half[0][i] + half[1][i];
}
__global__ __launch_bounds__(256) void diag_mv_kernel(const double *__restrict__ LinvT, int transpose, const double *in,
                                                      double *out) {
    diag_mv_body(LinvT, transpose, in, out);
}

// the block that takes the last ticket of a launch runs the 128 x 128 product that depends on all of them (one launch
// and one dependent-kernel gap less per block step of the solves)
__device__ __forceinline__ bool chol_last_block(unsigned int *ticket) {
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last) {
        __threadfence();
        if (threadIdx.x == 0) *ticket = 0;
    }
    return last != 0;
}

// rhs[c] -= sum_{r < 128} L[k0 + r][c] * x[r]  for c < k0; the last block to finish then solves the next (previous)
// diagonal block in place: rhs[k0-128 : k0) = Linv^T_{k-1} rhs[k0-128 : k0)
// One workgroup per 128 columns: 64 column pairs x 4 groups of 32 rows, a thread's 32 row loads in flight together, the
// four partial sums combined in a fixed order through LDS.  (Its first version gave a workgroup 512 columns and every
// thread all 128 rows: k0/512 workgroups, each pulling 512 KB through one CU — 19 us per launch in the ActiveSet trace.)
__global__ __launch_bounds__(256) void bwd_update_kernel(const double *__restrict__ H, int64_t ldh, int64_t k0,
                                                         const double *xk, double *rhs,
                                                         const double *__restrict__ LinvT_prev, unsigned int *ticket) {
    __shared__ double x[NB];
    __shared__ double part[3][NB];
    if (threadIdx.x < NB) x[threadIdx.x] = xk[threadIdx.x];
    __syncthreads();
    const int cp = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * NB + 2 * cp;   // k0 is a multiple of 128: the slab never straddles k0
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const int r = rg * 32 + rr;
        const bq_d2 l = *reinterpret_cast<const bq_d2 *>(H + (k0 + r) * ldh + c);
        a0 = fma(l.x, x[r], a0);
        a1 = fma(l.y, x[r], a1);
    }
    if (rg > 0) {
        part[rg - 1][2 * cp] = a0;
        part[rg - 1][2 * cp + 1] = a1;
    }
    __syncthreads();
    if (rg == 0) {
        a0 = ((a0 + part[0][2 * cp]) + part[1][2 * cp]) + part[2][2 * cp];
        a1 = ((a1 + part[0][2 * cp + 1]) + part[1][2 * cp + 1]) + part[2][2 * cp + 1];
        rhs[c] -= a0;
        rhs[c + 1] -= a1;
    }
    if (chol_last_block(ticket)) diag_mv_body(LinvT_prev, 1, rhs + k0 - NB, rhs + k0 - NB);
}

// ---------------------------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------------------------
int bq_chol_ws_create(bq_ctx *ctx, int64_t n, bq_chol_ws **out) {
    bq_chol_ws *ws = new bq_chol_ws();
    ws->ctx = ctx;
    ws->cap = bq_round_up(n, NB);
    ws->ldh = ws->cap;
    const int64_t nblk = ws->cap / NB;
    // bq_device_malloc: a panel kept for re-use that is in the way is dropped and the allocation retried
    hipError_t e = hipMalloc(&ws->H, sizeof(double) * ws->ldh * ws->cap);
    if (e != hipSuccess) {
        bq_set_error("cannot allocate the %lld x %lld factorisation workspace (%.1f GB): %s", (long long)ws->cap,
                     (long long)ws->cap, 8e-9 * ws->ldh * ws->cap, hipGetErrorString(e));
        delete ws;
        return BQ_ERR_NOMEM;
    }
    ws->super_max = ws->cap >= 24576 ? 4 : 2;   // passes per super-pass the image buffers are sized for
    if (e == hipSuccess) e = hipMalloc(&ws->Wt, sizeof(double) * 2 * ws->super_max * 2 * NB * ws->ldh);   // 2 super-passes of images
    if (e == hipSuccess) e = hipMalloc(&ws->LinvT, sizeof(double) * nblk * NB * NB);
    if (e == hipSuccess) e = hipMalloc(&ws->rhs, sizeof(double) * (ws->cap + NB));
    if (e == hipSuccess) e = hipMalloc(&ws->tmp, sizeof(double) * 4 * NB);   // up to four column slices of a block row
    if (e == hipSuccess) e = hipMalloc(&ws->info, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&ws->ticket, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemset(ws->ticket, 0, sizeof(unsigned int));
    if (e == hipSuccess && bq_potrf_diag_setup() != BQ_OK) e = hipErrorUnknown;
    if (e != hipSuccess) {
        bq_set_error("factorisation workspace setup failed: %s", hipGetErrorString(e));
        bq_chol_ws_destroy(ws);
        return BQ_ERR_HIP;
    }
    // Look-ahead: the narrow work of pass p+1 is issued on a high-priority stream beside the wide update of pass p
    // (priority streams; issuing everything in order on one stream was 5 % slower at n = 50 000, CU-masked streams did not finish a
    // run at n = 16 384 within its limit: both removed in round 5).  The
    // diagonal-block kernel needs a whole free CU (136 KB of LDS, 8 waves) and the wide update keeps every CU's
    // register file full, so the priority stream mostly gets its turn in the tail of the wide kernel.
    const int la_mode = 2;   // priority streams
    if (la_mode != 0 && ctx->num_cu >= 64) {
        bool ok;
        if (la_mode == 2) {
            int lo = 0, hi = 0;
            ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
                 hipStreamCreateWithPriority(&ws->s_main, hipStreamNonBlocking, lo) == hipSuccess &&
                 hipStreamCreateWithPriority(&ws->s_side, hipStreamNonBlocking, hi) == hipSuccess;
        } else {
            const int words = (ctx->num_cu + 31) / 32;
            std::vector<uint32_t> side(words, 0u), mainm(words, 0u);
            const int reserved = 16;
            for (int cu = 0; cu < ctx->num_cu; ++cu) {
                if (cu < reserved)
                    side[cu / 32] |= 1u << (cu % 32);
                else
                    mainm[cu / 32] |= 1u << (cu % 32);
            }
            ok = hipExtStreamCreateWithCUMask(&ws->s_main, (uint32_t)words, mainm.data()) == hipSuccess &&
                 hipExtStreamCreateWithCUMask(&ws->s_side, (uint32_t)words, side.data()) == hipSuccess;
        }
        for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&ws->ev[i], hipEventDisableTiming) == hipSuccess;
        ws->lookahead = ok;
        if (!ok) (void)hipGetLastError();
    }
    *out = ws;
    return BQ_OK;
}

void bq_chol_ws_destroy(bq_chol_ws *ws) {
    if (!ws) return;
    for (hipEvent_t e : ws->ev)
        if (e) hipEventDestroy(e);
    if (ws->s_main) hipStreamDestroy(ws->s_main);
    if (ws->s_side) hipStreamDestroy(ws->s_side);
    for (void *p : {(void *)ws->H, (void *)ws->Wt, (void *)ws->LinvT, (void *)ws->rhs, (void *)ws->tmp, (void *)ws->info, (void *)ws->ticket,
                    (void *)ws->mr_vec, (void *)ws->mr_state, (void *)ws->mr_part, (void *)ws->bigM, (void *)ws->bigMT,
                    (void *)ws->big_scratch, (void *)ws->sw_t, (void *)ws->pivot_thr})
        if (p) hipFree(p);
    if (ws->mr_flag) hipHostFree(ws->mr_flag);
    delete ws;
}

__global__ void pivot_thr_kernel(const double *__restrict__ H, int64_t ldh, int64_t np, double rel, double *__restrict__ thr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) thr[i] = rel * fabs(H[i * ldh + i]);
}

// factor the leading np x np block of ws->H (np multiple of 128, lower triangle valid) in place
int bq_chol_factor(bq_chol_ws *ws, int64_t np) {
    bq_ctx *ctx = ws->ctx;
    hipStream_t st = ctx->stream;
    BQ_ARG(np % NB == 0 && np <= ws->cap, "factor size");
    ws->sweep_np = 0;   // a new factor: the fast sweeps have to be prepared again (bq_chol_prepare_sweeps)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_CHOL, &e0, &e1));
    BQ_HIP(hipMemsetAsync(ws->info, 0, sizeof(int), st));
    const int64_t ldh = ws->ldh;
    const double *thr = nullptr;
    if (ws->pivot_rel > 0.0) {   // thresholds from the diagonal as given, before any update touches it
        if (!ws->pivot_thr) BQ_HIP(hipMalloc(&ws->pivot_thr, sizeof(double) * ws->cap));
        pivot_thr_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(ws->H, ldh, np, ws->pivot_rel, ws->pivot_thr);
        thr = ws->pivot_thr;
    }
    // Two block columns per pass p (A_p = 2p*NB, B_p = A_p + NB), two passes per SUPER-PASS q (passes 2q, 2q+1):
    //   narrow(p): diag(A) ; TRSM(A) ; narrow update of column B ; diag(B) ; TRSM(B)        -> images of pass p
    //   head_in(p): the two block columns pass p+1 factors next  -= the images of the super-pass so far
    //   wide(q)  : everything behind the super-pass -= X X^T with ALL its images in ONE K = P * 256 pass — a C tile is read
    //              and written once per P * 256 columns factored (tools/syrk_probe.hip, final tile kernel: 69.5 / 73.6 /
    //              74.3 TFLOP/s at K = 512 / 768 / 1024; with the round-1 kernel it was 0.71 of the MFMA peak at K = 256);
    //              split into head (the 2P block columns the next chain works on) and rest.
    // The images of a super-pass are contiguous in k, two buffers.
    // With look-ahead the narrow chain of super-pass q+1 (it touches only the 2P block columns head(q) has finished) runs on
    // the high-priority side stream while rest(q) keeps the chip busy.
    // P passes per super-pass (by size): the wide update then has K = P * 256.  A larger P cuts the C
    // traffic and the per-tile prologue per flop further but lengthens the narrow chain between two wide updates.
    const int P = [&] {
        int v = np >= 65536 ? 4 : (np >= 24576 ? 3 : 2);   // measured (final kernel): n=50k 622 / 608 / 621 ms, n=32k 192 / 190 / 194 ms for P = 2 / 3 / 4
        return v < 1 ? 1 : (v > ws->super_max ? ws->super_max : v);
    }();
    auto wimg = [&](int64_t p) {   // images of a super-pass are contiguous in k; two buffers
        return ws->Wt + ((p / P) & 1) * (int64_t)(P * 2 * NB) * ldh + (p % P) * (int64_t)(2 * NB) * ldh;
    };
    auto narrow = [&](int64_t p, hipStream_t s) {
        const int64_t a0 = 2 * p * NB;
        if (a0 >= np) return;
        double *WtA = wimg(p), *WtB = WtA + (int64_t)NB * ldh;
        auto panel = [&](int64_t k0, double *Wimg) {   // diag(k0) done: TRSM the rows below, leave their image in Wimg
            const int64_t i0 = k0 + NB;
            const int64_t T = (np - i0) / NB;
            const double *LinvT = ws->LinvT + (k0 / NB) * NB * NB;
            trsm_gemm_kernel<<<(unsigned)T, 256, 0, s>>>(ws->H, ldh, k0, i0, Wimg, LinvT);
        };
        bq_launch_potrf_diag(s, ws->H, ldh, a0, ws->LinvT + (a0 / NB) * NB * NB, ws->info, thr);
        const int64_t b0 = a0 + NB;
        if (b0 >= np) return;
        panel(a0, WtA);
        syrk_col_kernel<<<(unsigned)((np - b0) / NB), 256, 0, s>>>(ws->H, ldh, b0, WtA);
        bq_launch_potrf_diag(s, ws->H, ldh, b0, ws->LinvT + (b0 / NB) * NB * NB, ws->info, thr);
        if (b0 + NB >= np) return;
        panel(b0, WtB);
    };
    auto head_grid = [](int64_t T, int nc) {
        int64_t g = 0;
        for (int c = 0; c < nc && c < T; ++c) g += T - c;
        return (unsigned)g;
    };
    // the two block columns pass p+1 factors  -=  the images of ALL passes of p's super-pass so far (K = 256 .. (P-1) 256)
    auto head_in = [&](int64_t p, hipStream_t s) {
        const int64_t r0 = 2 * p * NB + 2 * NB;
        if (r0 >= np) return;
        const int64_t T = (np - r0) / NB;
        const int64_t first = p / P * P;
        syrk_head_kernel<<<head_grid(T, 2), 256, 0, s>>>(ws->H, ldh, r0, wimg(first), (int)((p - first + 1) * 2 * NB), T, 2);
    };
    // the narrow chain of super-pass q: narrow, head_in, narrow, ..., narrow
    auto chain = [&](int64_t q, hipStream_t s) {
        for (int j = 0; j < P; ++j) {
            narrow(q * P + j, s);
            if (j + 1 < P) head_in(q * P + j, s);
        }
    };
    // everything behind super-pass q  -=  X X^T with its 2P images in one K = P * 256 pass: first the 2P block columns the next
    // chain works on, then the rest
    auto wide_head = [&](int64_t q, hipStream_t s) {
        const int64_t r0 = (q + 1) * P * 2 * NB;
        if (r0 >= np) return;
        const int64_t T = (np - r0) / NB;
        syrk_head_kernel<<<head_grid(T, 2 * P), 256, 0, s>>>(ws->H, ldh, r0, wimg(q * P), P * 2 * NB, T, 2 * P);
    };
    auto wide_rest = [&](int64_t q, hipStream_t s) {
        const int64_t r0 = (q + 1) * P * 2 * NB + 2 * P * NB;
        if (r0 >= np) return;
        const int64_t T = (np - r0) / NB;
        const int64_t S = (T + 7) / 8, ntiles = T * (T + 1) / 2;
        // super-tiles pay off once an XCD has many of them; below ~8 rounds of the chip the even split wins
        const int64_t nsuper = ntiles >= 8 * 512 ? S * (S + 1) / 2 : 0;
        const unsigned grid = nsuper ? (unsigned)(((nsuper + 7) / 8) * 8 * 64) : (unsigned)ntiles;
        syrk_kernel<<<grid, 256, 0, s>>>(ws->H, ldh, r0, wimg(q * P), P * 2 * NB, T, nsuper);
    };
    const int64_t npass = (np + 2 * NB - 1) / (2 * NB), nsup = (npass + P - 1) / P;
    if (!ws->lookahead || np < 16 * NB) {
        for (int64_t q = 0; q < nsup; ++q) {
            chain(q, st);
            wide_head(q, st);
            wide_rest(q, st);
        }
    } else {
        hipStream_t sm = ws->s_main, ss = ws->s_side;
        BQ_HIP(hipEventRecord(ws->ev[0], st));       // everything enqueued so far (H assembly) precedes the factorisation
        BQ_HIP(hipStreamWaitEvent(sm, ws->ev[0], 0));
        chain(0, sm);
        for (int64_t q = 0; q < nsup; ++q) {
            hipEvent_t e_head = ws->ev[1 + (q % 3)], e_narrow = ws->ev[4 + (q % 3)];
            wide_head(q, sm);
            BQ_HIP(hipEventRecord(e_head, sm));
            BQ_HIP(hipStreamWaitEvent(ss, e_head, 0));
            chain(q + 1, ss);
            BQ_HIP(hipEventRecord(e_narrow, ss));
            wide_rest(q, sm);
            BQ_HIP(hipStreamWaitEvent(sm, e_narrow, 0));
        }
        BQ_HIP(hipEventRecord(ws->ev[7], sm));
        BQ_HIP(hipStreamWaitEvent(st, ws->ev[7], 0));   // the solves (on the context stream) follow the factorisation
    }
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_prof_end(ctx, BQ_PROF_CHOL, e0, e1));
    return BQ_OK;
}

// solve (L L^T) x = ws->rhs in place (ws->rhs padded to np, pad entries zero)

// ---------------------------------------------------------------------------------------------
// Fast sweeps for a factor that is solved with MANY times (the ActiveSet keeps the factor of a base set for hundreds of
// iterations, each with one or two solves).  The block-by-block solves above take one dependent launch per 128 rows and
// direction (782 launches for a 10 GB factor at n = 50 000: ~10 % of the HBM rate).  bq_chol_prepare_sweeps() spends one
// pass over the factor to make every sweep a short chain of full-chip row-panel products instead:
//   * the bb x bb (1024 ... 4096) diagonal blocks of L are inverted once (M_K = L_KK^-1, built from the 128 x 128 inverses the
//     factorisation left behind: X_ij = -Linv_ii sum_{j <= k < i} L_ik X_kj on the MFMA tile kernel) and kept together with
//     their transposes;
//   * L^T is mirrored into the (unused) upper triangle of H, so that the backward sweep reads contiguous rows too.
// A sweep then takes two launches per bb rows: t = b_K - L[K, :K] y (all CUs streaming the row panel once) and
// y_K = M_K t.  n = 20 000: 40 launches instead of 154 per direction; the factor is streamed once per direction at the
// product's rate.  Every sum has a fixed order (no atomics): results do not depend on the launch geometry.
// ---------------------------------------------------------------------------------------------
// rows of a big block: bq_chol_ws::bb, chosen when the sweeps are prepared (1024 for small workspaces, 2048 from order 4096, 4096
// from order 8192 on: a quarter of the launches of a sweep for four times the inverse-block traffic — profiles/r06/as_sweep_block.txt)

// X_ij blocks of one big block's inverse, one workgroup per (column j of sub-blocks, big block)
__global__ __launch_bounds__(256, 2) void big_inverse_kernel(const double *__restrict__ H, int64_t ldh, int64_t np, int64_t BB,
                                                             const double *__restrict__ LinvT, double *__restrict__ M,
                                                             double *__restrict__ scratch) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    const int64_t K = blockIdx.y, j = blockIdx.x;
    const int64_t r0 = K * BB;
    const int64_t nsub = (np - r0 < BB ? np - r0 : BB) / NB;
    if (j >= nsub) return;
    double *Mk = M + K * BB * BB;
    double *S = scratch + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NB * NB;
    const double *Lj = LinvT + (r0 / NB + j) * NB * NB;
    // X_jj = Linv_jj  (LinvT[k][r] = Linv[r][k])
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        Mk[(j * NB + r) * BB + j * NB + c] = Lj[c * NB + r];
    }
    __syncthreads();
    for (int64_t i = j + 1; i < nsub; ++i) {
        bq_d4 acc[4][4];
        bq_tile_zero(acc);
        // S = L[i, j..i-1] X[j..i-1, j]: A row-major from H, B the k-major image X[k][c] = M rows
        bq_mfma_tile_128<false, true>(H + r0 + j * NB, ldh, r0 + i * NB, Mk + (j * NB) * BB + j * NB, BB, 0, (i - j) * NB, sm, acc);
        bq_tile_store(acc, S, NB);
        __syncthreads();
        // X_ij = -Linv_ii S: A image [k][r] = LinvT_i, B image [k][c] = S rows
        bq_tile_zero(acc);
        bq_mfma_tile_128<true, false>(LinvT + (r0 / NB + i) * NB * NB, NB, 0, S, NB, 0, NB, sm, acc);
        bq_tile_store(acc, Mk + (i * NB) * BB + j * NB, BB);
        __syncthreads();
    }
}

// MT_K = M_K^T (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void big_transpose_kernel(const double *__restrict__ M, double *__restrict__ MT, int64_t BB) {
    __shared__ double tile[32][33];
    const double *src = M + (int64_t)blockIdx.z * BB * BB;
    double *dst = MT + (int64_t)blockIdx.z * BB * BB;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8) tile[k][tx] = src[(r0 + k) * BB + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8) dst[(c0 + k) * BB + r0 + tx] = tile[tx][k];
}

// H[c][r] = H[r][c] for r > c (tiles of 32 x 32; the diagonal tiles mirror inside themselves)
__global__ __launch_bounds__(256) void mirror_lower_kernel(double *__restrict__ H, int64_t ldh, int64_t np) {
    __shared__ double tile[32][33];
    const int64_t tr = blockIdx.y, tc = blockIdx.x;
    if (tc > tr) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = tr * 32, c0 = tc * 32;
    for (int k = ty; k < 32; k += 8) tile[k][tx] = H[(r0 + k) * ldh + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = c0 + k, c = r0 + tx;   // element (r, c) of the upper part = lower (c, r) = tile[tx][k]
        if (c > r) H[r * ldh + c] = tile[tx][k];
    }
    (void)np;
}

// out[r] = base[r] + sign * sum_{c in [lo, hi)} A[(row0 + r) * lda + c] * v[c],  r < nrows; 4 rows per workgroup.
// mode 0: [lo, hi) = [c_lo, c_hi) for every row; mode 1 / 2: the non-zero 128-blocks of a lower / upper triangular
// 1024-block (rows r, columns up to / from r's own 128-block).  c_lo even, rows 16-byte aligned.
__global__ __launch_bounds__(256) void sweep_gemv_kernel(const double *__restrict__ A, int64_t lda, int64_t row0, int64_t nrows,
                                                         int64_t c_lo, int64_t c_hi, int mode, const double *__restrict__ v,
                                                         const double *base, double sign, double *out, double *out2) {
    constexpr int R = 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    int64_t lo = c_lo, hi = c_hi;
    if (mode == 1) hi = (r0 / NB + 1) * NB < c_hi ? (r0 / NB + 1) * NB : c_hi;
    if (mode == 2) lo = (r0 / NB) * NB;
    const double *rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rp[r] = A + (row0 + (r0 + r < nrows ? r0 + r : nrows - 1)) * lda;
    double acc[R] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int64_t c = lo + 2 * tid; c < hi; c += 512) {
        const bq_d2 x = *reinterpret_cast<const bq_d2 *>(v + c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bq_d2 a = __builtin_nontemporal_load(reinterpret_cast<const bq_d2 *>(rp[r] + c));
            acc[r] = fma(a.y, x.y, fma(a.x, x.x, acc[r]));
        }
    }
    __shared__ double red[4][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double s = wsum_c(acc[r]);
        if (lane == 0) red[wv][r] = s;
    }
    __syncthreads();
    if (tid < R && r0 + tid < nrows) {
        const double s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        const double val = (base ? base[r0 + tid] : 0.0) + sign * s;
        out[r0 + tid] = val;
        if (out2 != nullptr) out2[r0 + tid] = val;   // the caller's copy of the solution (saves a device-to-device copy per solve)
    }
}

// the block a sweep starts from, copied aside (the block's product with its inverted diagonal block overwrites it in place):
// a launch of this instead of a device-to-device copy COMMAND, twice per solve (a copy command costs the host ~3 x a launch and
// the stream a blit kernel between two barriers)
__global__ void sweep_copy_kernel(const double *__restrict__ src, double *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

int bq_chol_prepare_sweeps(bq_chol_ws *ws, int64_t np) {
    hipStream_t st = ws->ctx->stream;
    // the big block: a sweep is two launches per block and direction, so fewer, larger blocks shorten the launch chain of a sweep at
    // the price of more inverse-block traffic (np * bb * 8 B per direction); hook sweep_block forces it (tests, the sweep)
    // measured (profiles/r06/as_sweep_block.txt, dense ActiveSet to 'optimal'): n = 20 000 9.91 / 9.16 / 9.07 / 10.18 s and n = 50 000
    // 79.2 / 72.2 / 70.6 / 73.5 s with 1024 / 2048 / 4096 / 8192
    int64_t BB = ws->cap >= 8192 ? 4096 : (ws->cap >= 4096 ? 2048 : 1024);
    {
        double hv = 0.0;
        if (bq_hook("sweep_block", &hv) && (hv == 1024.0 || hv == 2048.0 || hv == 4096.0 || hv == 8192.0)) BB = (int64_t)hv;
    }
    const int64_t nbb = (np + BB - 1) / BB;
    if (ws->big_cap < nbb || ws->bb != BB) {
        for (double **p : {&ws->bigM, &ws->bigMT, &ws->big_scratch, &ws->sw_t})
            if (*p) {
                hipFree(*p);
                *p = nullptr;
            }
        ws->big_cap = 0;
        const int64_t want = (ws->cap + BB - 1) / BB;
        BQ_HIP(hipMalloc(&ws->bigM, sizeof(double) * want * BB * BB));
        BQ_HIP(hipMalloc(&ws->bigMT, sizeof(double) * want * BB * BB));
        BQ_HIP(hipMalloc(&ws->big_scratch, sizeof(double) * want * (BB / NB) * NB * NB));
        BQ_HIP(hipMalloc(&ws->sw_t, sizeof(double) * BB));
        ws->big_cap = want;
        ws->bb = BB;
    }
    BQ_HIP(hipMemsetAsync(ws->bigM, 0, sizeof(double) * nbb * BB * BB, st));
    big_inverse_kernel<<<dim3((unsigned)(BB / NB), (unsigned)nbb), 256, 0, st>>>(ws->H, ws->ldh, np, BB, ws->LinvT, ws->bigM, ws->big_scratch);
    big_transpose_kernel<<<dim3((unsigned)(BB / 32), (unsigned)(BB / 32), (unsigned)nbb), 256, 0, st>>>(ws->bigM, ws->bigMT, BB);
    mirror_lower_kernel<<<dim3((unsigned)(np / 32), (unsigned)(np / 32)), 256, 0, st>>>(ws->H, ws->ldh, np);
    BQ_HIP(hipGetLastError());
    ws->sweep_np = np;
    return BQ_OK;
}

static int chol_solve_fast(bq_chol_ws *ws, int64_t np, int64_t first_nonzero, double *also) {
    hipStream_t st = ws->ctx->stream;
    const int64_t BB = ws->bb;
    const int64_t ldh = ws->ldh, nbb = (np + BB - 1) / BB;
    auto rows_of = [&](int64_t K) { return np - K * BB < BB ? np - K * BB : BB; };
    double *rhs = ws->rhs, *t = ws->sw_t;
    // forward: t = b_K - L[K, kb:K] y;  y_K = M_K t
    const int64_t Kb = first_nonzero / BB;
    for (int64_t K = Kb; K < nbb; ++K) {
        const int64_t r0 = K * BB, nr = rows_of(K);
        const unsigned g = (unsigned)((nr + 3) / 4);
        const double *in = rhs + r0;
        if (K > Kb) {
            sweep_gemv_kernel<<<g, 256, 0, st>>>(ws->H, ldh, r0, nr, Kb * BB, r0, 0, rhs, rhs + r0, -1.0, t, nullptr);
            in = t;
        } else {
            sweep_copy_kernel<<<(unsigned)((nr + 255) / 256), 256, 0, st>>>(rhs + r0, t, nr);
            in = t;
        }
        sweep_gemv_kernel<<<g, 256, 0, st>>>(ws->bigM + K * BB * BB, BB, 0, nr, 0, nr, 1, in, nullptr, 1.0, rhs + r0, nullptr);
    }
    // backward: t = y_K - L^T[K, K+1:] x (L^T lives in the upper triangle of H);  x_K = M_K^T t
    for (int64_t K = nbb - 1; K >= 0; --K) {
        const int64_t r0 = K * BB, nr = rows_of(K);
        const unsigned g = (unsigned)((nr + 3) / 4);
        if (K < nbb - 1)
            sweep_gemv_kernel<<<g, 256, 0, st>>>(ws->H, ldh, r0, nr, r0 + BB, np, 0, rhs, rhs + r0, -1.0, t, nullptr);
        else
            sweep_copy_kernel<<<(unsigned)((nr + 255) / 256), 256, 0, st>>>(rhs + r0, t, nr);
        sweep_gemv_kernel<<<g, 256, 0, st>>>(ws->bigMT + K * BB * BB, BB, 0, nr, 0, nr, 2, t, nullptr, 1.0, rhs + r0,
                                             also ? also + r0 : nullptr);
    }
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

int bq_chol_solve(bq_chol_ws *ws, int64_t np, int64_t first_nonzero, double *also) {
    if (ws->sweep_np == np) return chol_solve_fast(ws, np, first_nonzero, also);
    hipStream_t st = ws->ctx->stream;
    const int64_t ldh = ws->ldh;
    auto Linv = [&](int64_t k0) { return ws->LinvT + (k0 / NB) * NB * NB; };
    // forward: y_k = Linv_kk (b_k - L_k,0:k y_0:k); one launch per block row (the panel product's last block applies Linv).
    // A right-hand side that vanishes before row first_nonzero (a unit vector: the ActiveSet update columns) leaves
    // y = 0 there, so the sweep starts at that row's block.
    const int64_t kb = (first_nonzero / NB) * NB;
    diag_mv_kernel<<<1, 256, 0, st>>>(Linv(kb), 0, ws->rhs + kb, ws->rhs + kb);
    for (int64_t k0 = kb + NB; k0 < np; k0 += NB) {
        const int64_t span = k0 - kb;
        const int slices = (int)std::min<int64_t>(4, (span + 16383) / 16384);
        const int64_t slice = ((span + slices - 1) / slices + 511) / 512 * 512;
        fwd_panel_kernel<<<dim3(NB, (unsigned)slices), 256, 0, st>>>(ws->H, ldh, k0, ws->rhs, ws->tmp, Linv(k0), ws->ticket, kb,
                                                                    slice);
    }
    // backward: x_k = Linv_kk^T y_k, then y_0:k -= L_k,0:k^T x_k (whose last block solves block k-1)
    diag_mv_kernel<<<1, 256, 0, st>>>(Linv(np - NB), 1, ws->rhs + np - NB, ws->rhs + np - NB);
    for (int64_t k0 = np - NB; k0 > 0; k0 -= NB)
        bwd_update_kernel<<<(unsigned)(k0 / NB), 256, 0, st>>>(ws->H, ldh, k0, ws->rhs + k0, ws->rhs, Linv(k0 - NB),
                                                               ws->ticket);
    if (also != nullptr) BQ_HIP(hipMemcpyAsync(also, ws->rhs, sizeof(double) * np, hipMemcpyDeviceToDevice, st));
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// H assembly: H[a][b] = Q[idx[a]][idx[b]] (+ hd[a] on the diagonal) for b <= a < m; identity on the pad
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void build_h_kernel(int structure, const T *__restrict__ panel, int64_t ldp, int packed, int64_t n,
                               const double *__restrict__ sgn, double diag_add, const int *__restrict__ idx, int64_t m,
                               int64_t np, const double *__restrict__ hd, double *__restrict__ H, int64_t ldh, int full) {
    const int64_t a0 = (int64_t)blockIdx.y * 32, b0 = (int64_t)blockIdx.x * 32;
    if (!full && b0 > a0 + 31) return;  // strictly-upper tile
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t b = b0 + tx;
    for (int j = ty; j < 32; j += 8) {
        const int64_t a = a0 + j;
        if (a >= np || (!full && b > a) || b >= np) continue;
        double v;
        if (a >= m || b >= m) {   // pad rows AND pad columns (a full build visits b > a): never index idx[] there
            v = (a == b) ? 1.0 : 0.0;
        } else {
            const int64_t i = idx ? idx[a] : a, jj = idx ? idx[b] : b;
            // (a full symmetric build visits the upper half too: bq_q_elem mirrors, panels hold the lower tiles)
            v = bq_q_elem(structure, panel, ldp, packed, n, sgn, diag_add, i, jj);
            if (i == jj && hd) v += hd[a];
        }
        H[a * ldh + b] = v;
    }
}

int bq_chol_build_h(bq_chol_ws *ws, bq_problem *p, const int *idx, int64_t m, const double *hd, int64_t *np_out,
                    bool full, int structure_override) {
    const int structure = structure_override >= 0 ? structure_override : p->structure;
    const int64_t np = bq_round_up(m > 0 ? m : 1, NB);
    BQ_ARG(np <= ws->cap, "H larger than the workspace");
    BQ_ARG(p->r0 == 0 && p->r1 == p->n, "the factorisation needs the whole panel on this rank");
    dim3 grid((unsigned)((np + 31) / 32), (unsigned)((np + 31) / 32));
    if (p->storage == BQ_F64)
        build_h_kernel<double><<<grid, 256, 0, ws->ctx->stream>>>(structure, (const double *)p->panel, p->ld, p->symmetric ? 1 : 0, p->n, p->sgn,
                                                                 p->diag_add, idx, m, np, hd, ws->H, ws->ldh, full ? 1 : 0);
    else
        build_h_kernel<float><<<grid, 256, 0, ws->ctx->stream>>>(structure, (const float *)p->panel, p->ld, p->symmetric ? 1 : 0, p->n, p->sgn,
                                                                p->diag_add, idx, m, np, hd, ws->H, ws->ldh, full ? 1 : 0);
    BQ_HIP(hipGetLastError());
    *np_out = np;
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// stand-alone entry: x = A^-1 b for a dense SPD host matrix (cho_factor + cho_solve), used by tests and benchmarks
// ---------------------------------------------------------------------------------------------
__global__ void pad_identity_kernel(double *H, int64_t ldh, int64_t n, int64_t np) {
    const int64_t i = n + (int64_t)blockIdx.x;
    for (int64_t j = threadIdx.x; j <= i; j += blockDim.x) H[i * ldh + j] = (i == j) ? 1.0 : 0.0;
    (void)np;
}

int bq_chol_solve_dense_impl(bq_ctx *ctx, int64_t n, const double *A, const double *b, double *x, double *factor_ms) {
    bq_chol_ws *ws = nullptr;
    BQ_TRY(bq_chol_ws_create(ctx, n, &ws));
    const int64_t np = bq_round_up(n, NB);
    hipStream_t st = ctx->stream;
    int rc = BQ_OK;
    hipError_t e = hipMemcpy2DAsync(ws->H, ws->ldh * 8, A, n * 8, n * 8, n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && np > n) pad_identity_kernel<<<(unsigned)(np - n), 128, 0, st>>>(ws->H, ws->ldh, n, np);
    if (e == hipSuccess) e = hipMemsetAsync(ws->rhs, 0, sizeof(double) * np, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ws->rhs, b, sizeof(double) * n, hipMemcpyHostToDevice, st);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    if (e == hipSuccess) rc = bq_chol_factor(ws, np);
    if (e == hipSuccess && rc == BQ_OK) e = hipEventRecord(e1, st);
    if (e == hipSuccess && rc == BQ_OK) rc = bq_chol_solve(ws, np);
    int info = 0;
    if (e == hipSuccess && rc == BQ_OK) e = hipMemcpyAsync(&info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && rc == BQ_OK) e = hipMemcpyAsync(x, ws->rhs, sizeof(double) * n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    float ms = 0.f;
    if (e == hipSuccess && rc == BQ_OK && e0 && e1) e = hipEventElapsedTime(&ms, e0, e1);
    if (factor_ms) *factor_ms = ms;
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    bq_chol_ws_destroy(ws);
    if (rc != BQ_OK) return rc;
    if (e != hipSuccess) {
        bq_set_error("dense Cholesky solve failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    if (info != 0) {
        bq_set_error("%d-th leading minor of the array is not positive definite", info);
        return BQ_ERR_NOT_PD;
    }
    return BQ_OK;
}
