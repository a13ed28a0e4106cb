// Symmetric panel product: out = K w reading only the tiles on or below the diagonal — half the HBM bytes of the
// row-block product for the (symmetric) Gram panels of the SVM dual.
//
// The panel is cut into 256 x 256 tiles (I, J), J <= I, and stored PACKED: tile row I keeps only its (I+1)*256 leading
// columns (row-major), so the lower triangle costs half the HBM of the square panel.  A work item is a STRIP: up to JG consecutive tiles of one
// tile row, i.e. 256 rows x (JG * 2 KiB) contiguous bytes per row.  The workgroup streams the strip once (16-byte
// non-temporal loads, lanes along the columns) and produces BOTH contributions the strip is responsible for:
//   row part   S[I][J0][r]  = sum over the strip's columns of elem(K[I*T+r][c]) * w[c]        (to output block I)
//   col parts  S[J][I][c]   = sum_r elem(K[I*T+r][J*T+c]) * w[I*T+r],  each J != I of the strip  (to output block J)
// into a slab S[nb][nb][T]; a second kernel sums the slab entries of each output block in a fixed order ->
// deterministic, no atomics.  Column sums are lane-local (a lane owns 4 columns of every tile); the row partials of a
// step are accumulated over the whole strip and then reduced across the 64 lanes with a halving butterfly
// (4 row sums in 7 shuffle-adds).  Slab traffic is about n^2*8/T*(1/2 + 1/(2 JG)) bytes each way (~1 % of the tile bytes).
//
// Multi-GPU: the tile rows are cut into S canonical SEGMENTS with equal shares of the triangle's tiles (bq_common.h); a rank
// owns a contiguous run of segments and produces, per segment, a partial vector for every output block.  One in-place
// all-gather per product hands every rank all S segment vectors, which are then added in segment order — the same
// association for any rank count, so 1/2/4/8-GPU iterates are bit-identical.  (BQ_SYM_EXCHANGE=allreduce: each rank adds
// its own segments and the rank partials meet in one ncclAllReduce — the sum's association then depends on the ring.)
#include <cstdlib>

#include <hip/hip_ext.h>

#include "bq_common.h"
#include "bq_epilogue.h"

typedef double d2_t __attribute__((ext_vector_type(2)));

constexpr int ST = 256;   // tile edge (== BQ_SYM_TILE)
constexpr int JG_DEFAULT = 8;   // tiles per strip (8 x 2 KiB contiguous per row; best of the measured variants)

typedef float f4_t __attribute__((ext_vector_type(4)));

// One 16-byte non-temporal load per lane and row.  fp64: a lane owns columns {2l, 2l+1} and {128+2l, 128+2l+1} of a tile
// (two loads); fp32: columns {4l .. 4l+3} (one float4 load).  c0 / c1 are the first columns of the two pairs.
template <typename T> struct tile_ld;
template <> struct tile_ld<double> {
    static __device__ __forceinline__ int c0(int lane) { return 2 * lane; }
    static __device__ __forceinline__ int c1(int lane) { return 128 + 2 * lane; }
    static __device__ __forceinline__ void get(const double *row, int lane, d2_t &a, d2_t &b) {
        a = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(row + 2 * lane));
        b = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(row + 128 + 2 * lane));
    }
};
template <> struct tile_ld<float> {
    static __device__ __forceinline__ int c0(int lane) { return 4 * lane; }
    static __device__ __forceinline__ int c1(int lane) { return 4 * lane + 2; }
    static __device__ __forceinline__ void get(const float *row, int lane, d2_t &a, d2_t &b) {
        const f4_t v = __builtin_nontemporal_load(reinterpret_cast<const f4_t *>(row + 4 * lane));
        a = (d2_t){(double)v.x, (double)v.y};
        b = (d2_t){(double)v.z, (double)v.w};
    }
};

// strips of tile row I: g = 0 .. I / JG ; linear index over tile rows [I0, I1)
template <int JG>
__device__ __host__ __forceinline__ int64_t strips_before(int64_t I) {  // sum_{i < I} (i / JG + 1)
    const int64_t qq = I / JG, rr = I % JG;
    return JG * qq * (qq + 1) / 2 + rr * (qq + 1);
}

// One workgroup of four waves streams all 256 rows of a strip: wave `wv` owns rows [64 wv, 64 wv + 64).
// (Round 4 also carried a row-cut variant — two 128-thread workgroups per strip with a three-entry column slab — for short grids: built,
// bit-identical, measured no faster (1/8 shares 0.796 - 0.885 ms cut against 0.801 - 0.832 ms uncut, profiles/r04/symv_row_cut_strips.txt)
// and removed in round 5 together with the <4,4> <4,8> <2,8> <2,4> tuning variants: HISTORY.md.)
template <typename T, bool ADD_ONE, int JG, int SR>
__global__ __launch_bounds__(256, 2) void symv_tiles_kernel(const T *__restrict__ panel, int64_t I0, int64_t nb,
                                                          const double *__restrict__ w, double *__restrict__ slab,
                                                          const int *__restrict__ done, int *__restrict__ skip, int skip_seq) {
    if (done != nullptr && *done) {
        if (skip != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *skip = skip_seq;   // bq_prof_skip_arg: "this launch was no product"
        return;
    }
    __shared__ double colred[4][ST];
    // decode (I, g) from the linear strip index
    const int64_t t = (int64_t)blockIdx.x + strips_before<JG>(I0);
    int64_t I = (int64_t)sqrt(2.0 * (double)JG * (double)t);
    if (I >= nb) I = nb - 1;
    while (I > 0 && strips_before<JG>(I) > t) --I;
    while (strips_before<JG>(I + 1) <= t) ++I;
    const int64_t g = t - strips_before<JG>(I);
    const int64_t J0 = g * JG;
    const int nj = (int)((J0 + JG <= I + 1) ? JG : (I + 1 - J0));  // tiles in this strip (J <= I)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // packed symmetric layout: tile row I has pitch (I+1)*256 and starts at bq_sym_off(I) - bq_sym_off(I0)
    const int64_t pitch = bq_sym_pitch(I);
    const T *rows = panel + (bq_sym_off(I) - bq_sym_off(I0)) + (int64_t)(wv * 64) * pitch + J0 * ST;
    const double *wI = w + I * ST + wv * 64;
    const int c0 = tile_ld<T>::c0(lane), c1 = tile_ld<T>::c1(lane);
    d2_t wj0[JG], wj1[JG];
    double ca[JG][4];
#pragma unroll
    for (int j = 0; j < JG; ++j) {
        const double *wJ = w + (J0 + (j < nj ? j : 0)) * ST;
        wj0[j] = *reinterpret_cast<const d2_t *>(wJ + c0);
        wj1[j] = *reinterpret_cast<const d2_t *>(wJ + c1);
        ca[j][0] = ca[j][1] = ca[j][2] = ca[j][3] = 0.0;
    }
    double *rowout = slab + (I * nb + J0) * ST + wv * 64;
    const bool b5 = lane & 32, b4 = lane & 16;

#pragma unroll 1
    for (int step = 0; step < 64 / SR; ++step) {
        double rp[SR];
        double wi[SR];
#pragma unroll
        for (int k = 0; k < SR; ++k) {
            rp[k] = 0.0;
            wi[k] = wI[step * SR + k];
        }
#pragma unroll
        for (int j = 0; j < JG; ++j) {
            if (j < nj) {
                d2_t a[SR], b[SR];
#pragma unroll
                for (int k = 0; k < SR; ++k) {
                    const T *row = rows + (int64_t)(step * SR + k) * pitch + j * ST;
                    tile_ld<T>::get(row, lane, a[k], b[k]);
                }
#pragma unroll
                for (int k = 0; k < SR; ++k) {
                    if (ADD_ONE) {
                        a[k].x += 1.0;
                        a[k].y += 1.0;
                        b[k].x += 1.0;
                        b[k].y += 1.0;
                    }
                    rp[k] = fma(b[k].y, wj1[j].y, fma(b[k].x, wj1[j].x, fma(a[k].y, wj0[j].y, fma(a[k].x, wj0[j].x, rp[k]))));
                    ca[j][0] = fma(a[k].x, wi[k], ca[j][0]);
                    ca[j][1] = fma(a[k].y, wi[k], ca[j][1]);
                    ca[j][2] = fma(b[k].x, wi[k], ca[j][2]);
                    ca[j][3] = fma(b[k].y, wi[k], ca[j][3]);
                }
            }
        }
        // halving butterfly: SR row partials x 64 lanes -> one full row sum per (64 / SR)-lane group
        double s1;
        int rho;
        if constexpr (SR == 8) {
            const bool b3 = lane & 8;
            double u[4], t2[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double send = b5 ? rp[i] : rp[i + 4];
                const double keep = b5 ? rp[i + 4] : rp[i];
                u[i] = keep + __shfl_xor(send, 32, 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double send = b4 ? u[i] : u[i + 2];
                const double keep = b4 ? u[i + 2] : u[i];
                t2[i] = keep + __shfl_xor(send, 16, 64);
            }
            {
                const double send = b3 ? t2[0] : t2[1];
                const double keep = b3 ? t2[1] : t2[0];
                s1 = keep + __shfl_xor(send, 8, 64);
            }
            rho = (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0);
        } else {
            static_assert(SR == 4 || SR == 8, "SR");
            double u[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double send = b5 ? rp[i] : rp[i + 2];
                const double keep = b5 ? rp[i + 2] : rp[i];
                u[i] = keep + __shfl_xor(send, 32, 64);
            }
            {
                const double send = b4 ? u[0] : u[1];
                const double keep = b4 ? u[1] : u[0];
                s1 = keep + __shfl_xor(send, 16, 64);
            }
            s1 += __shfl_xor(s1, 8, 64);
            rho = (b5 ? 2 : 0) + (b4 ? 1 : 0);
        }
        s1 += __shfl_xor(s1, 4, 64);
        s1 += __shfl_xor(s1, 2, 64);
        s1 += __shfl_xor(s1, 1, 64);
        if ((lane & (64 / SR - 1)) == 0) rowout[step * SR + rho] = s1;
    }
    // column parts of every off-diagonal tile of the strip
#pragma unroll
    for (int j = 0; j < JG; ++j) {
        if (j < nj && J0 + j != I) {   // uniform across the workgroup
            __syncthreads();
            colred[wv][c0] = ca[j][0];
            colred[wv][c0 + 1] = ca[j][1];
            colred[wv][c1] = ca[j][2];
            colred[wv][c1 + 1] = ca[j][3];
            __syncthreads();
            const int64_t entry = (J0 + j) * nb + I;
            slab[entry * ST + tid] = ((colred[0][tid] + colred[1][tid]) + colred[2][tid]) + colred[3][tid];
        }
    }
}

// Partial sum of output block a over the slab entries S[a][b] that the tile rows [c0, c1) produced, b ascending:
//   row parts live at b = first tile of a strip (b % JG == 0, b <= a) when tile row a lies in [c0, c1),
//   col parts at every b > a inside [c0, c1).
// 1024 threads: thread (r, q) sums every 4th entry of that fixed entry list (two independent chains each for load-level
// parallelism); the four partial sums are combined in the fixed order q = 0..3.  Every thread returns the combined value.
template <int JG>
__device__ __forceinline__ double seg_thread_sum(const double *__restrict__ p, int64_t a, int64_t c0, int64_t c1, int q) {
    double s0 = 0.0, s1 = 0.0;
    int64_t e = 0;   // running index over the entry list: row parts (b = 0, JG, 2JG, ... <= a) then col parts (b > a)
    // Two chains (s0: entries k, k + 8, ...; s1: k + 4, k + 12, ...), each added in its own order — but the LOADS of four turns are
    // issued together (round 5): written as one load per turn the loop was a chain of L2 round trips (~11 of them at nb = 79, ~50 at
    // nb = 391: most of this kernel's 7 us on short grids); the association, and with it every bit, is what it was.
    auto walk = [&](const double *base, int64_t first, int64_t count, int64_t stride) {
        int64_t k = first;
        for (; k + 28 < count; k += 32) {   // four turns of both chains: all eight entries exist
            const double a0 = base[(k) * stride], b0 = base[(k + 4) * stride], a1 = base[(k + 8) * stride], b1 = base[(k + 12) * stride];
            const double a2 = base[(k + 16) * stride], b2 = base[(k + 20) * stride], a3 = base[(k + 24) * stride], b3 = base[(k + 28) * stride];
            s0 += a0;
            s1 += b0;
            s0 += a1;
            s1 += b1;
            s0 += a2;
            s1 += b2;
            s0 += a3;
            s1 += b3;
        }
        for (; k < count; k += 8) {
            s0 += base[k * stride];
            if (k + 4 < count) s1 += base[(k + 4) * stride];
        }
    };
    if (a >= c0 && a < c1) {
        const int64_t nrow = a / JG + 1;
        walk(p, q, nrow, (int64_t)JG * ST);
        e = nrow;
    }
    const int64_t bs = (a + 1 > c0) ? a + 1 : c0;
    const int64_t ncol = c1 > bs ? c1 - bs : 0;
    // keep the q-assignment a function of the position in the whole list (row parts first)
    const int64_t shift = (4 - (e & 3)) & 3;
    walk(p + bs * ST, (q + shift) & 3, ncol, ST);
    return s0 + s1;   // this thread's share (every 4th entry, q = its phase) of the segment's entry list
}

// the four phases of a segment's sum combined in the fixed order q = 0..3; every thread returns the combined value
template <int JG>
__device__ __forceinline__ double seg_partial(const double *__restrict__ p, int64_t a, int64_t c0, int64_t c1, int q, int r,
                                              double (*part)[ST]) {
    part[q][r] = seg_thread_sum<JG>(p, a, c0, c1, q);
    __syncthreads();
    const double v = ((part[0][r] + part[1][r]) + part[2][r]) + part[3][r];
    __syncthreads();
    return v;
}

// out[a*T + r] = sum over the segments [tab.lo, tab.hi) of their partial sums, in segment order — the canonical order of
// the product: the same association whether the segments were summed here (one rank) or gathered from their owners
// EPI: the PG / FW epilogue of bq_epilogue.h goes on from the summed product (out is still written: other consumers read p->s)
template <int JG, int EPI>
__global__ __launch_bounds__(1024) void symv_reduce_kernel(const double *__restrict__ slab, int64_t nb, bq_seg_table tab,
                                                           double *__restrict__ out, const int *__restrict__ done, bq_epilogue epi) {
    if (done != nullptr && *done) return;
    // the phase sums of up to eight segments (all of them on one rank) meet in LDS in ONE round of barriers — one round per segment
    // made sixteen barriers of 1 024 threads, a third of this kernel's length; the association per segment and the segment order
    // of the final sum are what they were
    __shared__ double part[BQ_SYM_SEG][4][ST];
    const int64_t a = blockIdx.x;
    const int r = threadIdx.x & (ST - 1), q = threadIdx.x >> 8;
    const double *p = slab + a * nb * ST + r;
    bq_epi_pre pre;
    bq_al_pre apre;
    if constexpr (EPI == BQ_EPI_PGFW) pre = bq_epi_preload(epi, a * ST + r, q == 0);   // in flight while the slab is walked
    if constexpr (EPI == BQ_EPI_AL) apre = bq_al_epi_preload(epi, a * ST + r, q == 0, true);
    double acc = 0.0;
    for (int s0 = tab.lo; s0 < tab.hi; s0 += BQ_SYM_SEG) {
        const int ns = tab.hi - s0 < BQ_SYM_SEG ? tab.hi - s0 : BQ_SYM_SEG;
        for (int k = 0; k < ns; ++k) part[k][q][r] = seg_thread_sum<JG>(p, a, tab.cut[s0 + k], tab.cut[s0 + k + 1], q);
        __syncthreads();
        if (q == 0)
            for (int k = 0; k < ns; ++k) acc += ((part[k][0][r] + part[k][1][r]) + part[k][2][r]) + part[k][3][r];
        __syncthreads();
    }
    if (q == 0) out[a * ST + r] = acc;
    if constexpr (EPI == BQ_EPI_PGFW) bq_epi_finish(epi, a, nb, bq_epi_element(epi, pre, a * ST + r, acc), gridDim.x);
    if constexpr (EPI == BQ_EPI_AL) bq_al_epi_finish<true>(epi, a, nb, bq_al_epi_element(epi, apre, a * ST + r, acc, true), gridDim.x);
}

// the per-segment partial vectors of this rank's segments, each to its slot of the gathered buffer
template <int JG>
__global__ __launch_bounds__(1024) void symv_reduce_seg_kernel(const double *__restrict__ slab, int64_t nb, bq_seg_table tab,
                                                               double *__restrict__ gath, const int *__restrict__ done) {
    if (done != nullptr && *done) return;
    __shared__ double part[4][ST];
    const int64_t a = blockIdx.x;
    const int s = tab.lo + (int)blockIdx.y;
    const int r = threadIdx.x & (ST - 1), q = threadIdx.x >> 8;
    const double v = seg_partial<JG>(slab + a * nb * ST + r, a, tab.cut[s], tab.cut[s + 1], q, r, part);
    if (q == 0) gath[((int64_t)tab.slot[s] * nb + a) * ST + r] = v;
}

// out = sum of all S gathered segment vectors in segment order (every rank: identical bits); EPI: as symv_reduce_kernel
template <int EPI>
__global__ __launch_bounds__(256) void symv_segsum_kernel(const double *__restrict__ gath, int64_t len, bq_seg_table tab,
                                                          double *__restrict__ out, const int *__restrict__ done, bq_epilogue epi) {
    if (done != nullptr && *done) return;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // len = nb * 256: always in range
    bq_epi_pre pre;
    bq_al_pre apre;
    if constexpr (EPI == BQ_EPI_PGFW) pre = bq_epi_preload(epi, i, true);
    if constexpr (EPI == BQ_EPI_AL) apre = bq_al_epi_preload(epi, i, true, true);
    double acc = 0.0;
    int s = 0;
    for (; s + 8 <= tab.count; s += 8) {   // the canonical eight segments: eight loads in flight, added in segment order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gath[(int64_t)tab.slot[s + u] * len + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < tab.count; ++s) acc += gath[(int64_t)tab.slot[s] * len + i];
    out[i] = acc;
    if constexpr (EPI == BQ_EPI_PGFW) bq_epi_finish(epi, blockIdx.x, gridDim.x, bq_epi_element(epi, pre, i, acc), gridDim.x);
    if constexpr (EPI == BQ_EPI_AL) bq_al_epi_finish<true>(epi, blockIdx.x, gridDim.x, bq_al_epi_element(epi, apre, i, acc, true), gridDim.x);
}

// The timed launch: when the context is profiling, the kernel's own dispatch carries the two timestamps (hipExtLaunchKernelGGL with a
// start and a stop event) instead of two events recorded around it on the stream: the duration is the kernel's own, as rocprofv3's
// kernel trace reports it, and the stream carries nothing a solve without profiling would not (measured: 1-2 us per product less than
// the bracketing, profiles/r04/share_gaps_events.txt).
template <typename K, typename... A>
static hipError_t launch_timed(bq_ctx *ctx, bool ext, hipEvent_t e0, hipEvent_t e1, K kernel, dim3 grid, dim3 block, A... args) {
    if (ext)
        hipExtLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, e0, e1, 0, args...);
    else
        kernel<<<grid, block, 0, ctx->stream>>>(args...);
    return hipGetLastError();
}

template <int JG, int SR>
static int launch_tiles(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t I0, int64_t I1, int64_t nb,
                        const double *w, double *slab, const int *done) {
    constexpr bool bracket = false;   // (round 4's BQ_PROF_BRACKET=1 — events recorded around the launch — cost 1-2 us per product: removed)
    const int64_t nstrips = strips_before<JG>(I1) - strips_before<JG>(I0);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_MATVEC, &e0, &e1, bracket));
    if (nstrips <= 0) {
        if (!bracket) {
            bq_prof_drop(ctx, e0, e1);
            return BQ_OK;
        }
        return bq_prof_end(ctx, BQ_PROF_MATVEC, e0, e1);
    }
    int *skip = nullptr, skip_seq = 0;
    if (done != nullptr) bq_prof_skip_arg(ctx, e0, &skip, &skip_seq);
    const bool ext = !bracket && e0 != nullptr;
    const dim3 grid((unsigned)nstrips), block(256);
    hipError_t err;
    if (storage == BQ_F64) {
        if (add_one)
            err = launch_timed(ctx, ext, e0, e1, symv_tiles_kernel<double, true, JG, SR>, grid, block, (const double *)panel, I0, nb, w, slab, done, skip, skip_seq);
        else
            err = launch_timed(ctx, ext, e0, e1, symv_tiles_kernel<double, false, JG, SR>, grid, block, (const double *)panel, I0, nb, w, slab, done, skip, skip_seq);
    } else {
        if (add_one)
            err = launch_timed(ctx, ext, e0, e1, symv_tiles_kernel<float, true, JG, SR>, grid, block, (const float *)panel, I0, nb, w, slab, done, skip, skip_seq);
        else
            err = launch_timed(ctx, ext, e0, e1, symv_tiles_kernel<float, false, JG, SR>, grid, block, (const float *)panel, I0, nb, w, slab, done, skip, skip_seq);
    }
    if (err != hipSuccess) {
        bq_prof_drop(ctx, e0, e1);
        bq_set_error("symv_tiles launch: %s", hipGetErrorString(err));
        return BQ_ERR_HIP;
    }
    return bq_prof_end(ctx, BQ_PROF_MATVEC, e0, e1, ext);
}

// mode 0: tiles + the sum over this launch's segments -> out (nb*256);  mode 1: tiles + one vector per segment -> gath slots
template <int SR>
static int launch_variant(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nb, const bq_seg_table &tab,
                          const double *w, double *slab, double *out, int mode, const int *done, const bq_epilogue *epi) {
    constexpr int JG = JG_DEFAULT;
    BQ_TRY((launch_tiles<JG, SR>(ctx, panel, storage, add_one, tab.cut[tab.lo], tab.cut[tab.hi], nb, w, slab, done)));
    if (mode == 0 && bq_epi_mode(epi) == BQ_EPI_PGFW)
        symv_reduce_kernel<JG, BQ_EPI_PGFW><<<(unsigned)nb, 1024, 0, ctx->stream>>>(slab, nb, tab, out, done, *epi);
    else if (mode == 0 && bq_epi_mode(epi) == BQ_EPI_AL)
        symv_reduce_kernel<JG, BQ_EPI_AL><<<(unsigned)nb, 1024, 0, ctx->stream>>>(slab, nb, tab, out, done, *epi);
    else if (mode == 0)
        symv_reduce_kernel<JG, BQ_EPI_NONE><<<(unsigned)nb, 1024, 0, ctx->stream>>>(slab, nb, tab, out, done, bq_epilogue{});
    else if (tab.hi > tab.lo)
        symv_reduce_seg_kernel<JG><<<dim3((unsigned)nb, (unsigned)(tab.hi - tab.lo)), 1024, 0, ctx->stream>>>(slab, nb, tab, out, done);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

static int launch_any(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nb, const bq_seg_table &tab,
                      const double *w, double *slab, double *out, int mode, const int *done, const bq_epilogue *epi = nullptr) {
    // Rows per step (loads in flight per wave).  fp32 tiles are half as wide in bytes: 8 rows per step keep the same bytes in
    // flight per lane.  fp64: 4 rows per step, except on short grids whose LAST round of workgroups is sparsely filled (two
    // workgroups per CU = 512 slots on this chip; a 1/8 share of n = 100 000 is ~1 200 strips = 2.3 rounds): there 8 rows per
    // step keep HBM busy while that round drains.  Measured (profiles/r03/symv_rows_per_step.txt): 1 200 strips 0.85 -> 0.81 ms,
    // 1 650 strips (n = 40 000) 1.07 -> 1.02 ms; grids whose last round is well filled are 2-6 % faster with 4, long grids do not
    // care.  The rows of a step are independent sums and the lane butterfly pairs the same lanes in the same order: the bits do
    // not depend on the choice (tests/test_distributed.py::test_rows_per_step_variants_are_bit_identical).
    const int force = (int)bq_hook_value("rows_per_step", 0.0);   // 4 or 8: no heuristic (the bit-identity test switches it)
    const int64_t strips = strips_before<JG_DEFAULT>(tab.cut[tab.hi]) - strips_before<JG_DEFAULT>(tab.cut[tab.lo]);
    const int64_t slots = 2 * (int64_t)(ctx->num_cu > 0 ? ctx->num_cu : 256);
    const bool sparse_tail = strips >= slots && strips < 8 * slots && 2 * (strips % slots) < slots;
    const bool eight = force == 8 || (force != 4 && (storage == BQ_F32 || sparse_tail));
    if (eight) return launch_variant<8>(ctx, panel, storage, add_one, nb, tab, w, slab, out, mode, done, epi);
    return launch_variant<4>(ctx, panel, storage, add_one, nb, tab, w, slab, out, mode, done, epi);
}

int bq_launch_symv(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nb, const bq_seg_table &tab,
                   const double *w, double *slab, double *out, const int *done, const bq_epilogue *epi) {
    return launch_any(ctx, panel, storage, add_one, nb, tab, w, slab, out, 0, done, epi);
}

int bq_launch_symv_segments(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nb, const bq_seg_table &tab,
                            const double *w, double *slab, double *gath, const int *done) {
    return launch_any(ctx, panel, storage, add_one, nb, tab, w, slab, gath, 1, done);
}

int bq_launch_symv_segsum(bq_ctx *ctx, int64_t nb, const bq_seg_table &tab, const double *gath, double *out, const int *done,
                          const bq_epilogue *epi) {
    const int64_t len = nb * ST;
    if (bq_epi_mode(epi) == BQ_EPI_PGFW)
        symv_segsum_kernel<BQ_EPI_PGFW><<<(unsigned)nb, 256, 0, ctx->stream>>>(gath, len, tab, out, done, *epi);
    else if (bq_epi_mode(epi) == BQ_EPI_AL)
        symv_segsum_kernel<BQ_EPI_AL><<<(unsigned)nb, 256, 0, ctx->stream>>>(gath, len, tab, out, done, *epi);
    else
        symv_segsum_kernel<BQ_EPI_NONE><<<(unsigned)nb, 256, 0, ctx->stream>>>(gath, len, tab, out, done, bq_epilogue{});
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
