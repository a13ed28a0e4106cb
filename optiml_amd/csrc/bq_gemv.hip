// Row-block panel product: the HBM-streaming kernel behind every Q*v of the dual-QP solvers.
//
// Replaces the dense `Q @ x` of the reference (optiml/opti/_base.py:282,291 and the d'Qd products at
// optiml/opti/constrained/projected_gradient.py:121, frank_wolfe.py:143, interior_point.py:193).
//
// Layout: the panel is row-major with a pitch `ld` that is a multiple of 1024 elements and a zero pad,
// so a 256-thread workgroup walks whole rows with 16-byte loads and no column tail.  One workgroup owns
// R consecutive rows for the full width: each lane keeps R fp64 accumulators, re-uses one 16-byte load
// of w for all R rows (w stays L2-resident; the panel is streamed once with non-temporal loads), and the
// row sums are combined by a fixed shuffle tree + a fixed 4-wave order -> bit-reproducible on any GPU
// count.  The kernel is bound by HBM: algorithmic bytes per launch = nrows*n*sizeof(T) + O(n).
#include "bq_common.h"

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <bool ADD_ONE>
__device__ __forceinline__ double dot16(d2_t p, d2_t w0, d2_t /*w1*/) {
    if (ADD_ONE) {
        p.x += 1.0;
        p.y += 1.0;
    }
    return fma(p.y, w0.y, p.x * w0.x);
}

template <bool ADD_ONE>
__device__ __forceinline__ double dot16(f4_t p, d2_t w0, d2_t w1) {
    double a = (double)p.x, b = (double)p.y, c = (double)p.z, e = (double)p.w;
    if (ADD_ONE) {
        a += 1.0;
        b += 1.0;
        c += 1.0;
        e += 1.0;
    }
    return fma(e, w1.y, fma(c, w1.x, fma(b, w0.y, a * w0.x)));
}

template <typename VT> struct vt_traits;
template <> struct vt_traits<d2_t> { static constexpr int elems = 2; };
template <> struct vt_traits<f4_t> { static constexpr int elems = 4; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename VT, int R, bool ADD_ONE, int UNROLL>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const VT *__restrict__ panel, int64_t ldv, int64_t nrows,
                                                        const d2_t *__restrict__ w, double *__restrict__ out,
                                                        const int *__restrict__ done, int *__restrict__ skip, int skip_seq) {
    if (done != nullptr && *done) {
        if (skip != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *skip = skip_seq;   // bq_prof_skip_arg
        return;
    }
    constexpr int WPL = vt_traits<VT>::elems / 2;  // d2 loads of w per lane per step
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const VT *rp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t row = row0 + r;
        if (row > nrows - 1) row = nrows - 1;
        rp[r] = panel + row * ldv;
    }
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;

    const int64_t steps = ldv / 256;  // ldv counts 16-byte vectors per row
#pragma unroll UNROLL
    for (int64_t c = 0; c < steps; ++c) {
        const int64_t i = c * 256 + tid;
        d2_t w0 = w[i * WPL];
        d2_t w1 = (WPL == 2) ? w[i * WPL + 1] : w0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            VT pv = __builtin_nontemporal_load(rp[r] + i);
            acc[r] += dot16<ADD_ONE>(pv, w0, w1);
        }
    }

    __shared__ double red[4][R];
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double s = wave_sum(acc[r]);
        if (lane == 0) red[wv][r] = s;
    }
    __syncthreads();
    if (tid < R) {
        int64_t row = row0 + tid;
        if (row < nrows) out[row] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    }
}

template <typename VT, bool ADD_ONE>
static void launch_r(hipStream_t st, const void *panel, int64_t ldv, int64_t nrows, const double *w, double *out,
                     const int *done, int *skip, int skip_seq) {
    const VT *P = reinterpret_cast<const VT *>(panel);
    const d2_t *W = reinterpret_cast<const d2_t *>(w);
    if (nrows >= 8192) {
        gemv_rows_kernel<VT, 8, ADD_ONE, 2><<<dim3((unsigned)((nrows + 7) / 8)), dim3(256), 0, st>>>(P, ldv, nrows, W, out, done, skip, skip_seq);
    } else if (nrows >= 2048) {
        gemv_rows_kernel<VT, 4, ADD_ONE, 2><<<dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st>>>(P, ldv, nrows, W, out, done, skip, skip_seq);
    } else if (nrows >= 512) {
        gemv_rows_kernel<VT, 2, ADD_ONE, 4><<<dim3((unsigned)((nrows + 1) / 2)), dim3(256), 0, st>>>(P, ldv, nrows, W, out, done, skip, skip_seq);
    } else {
        gemv_rows_kernel<VT, 1, ADD_ONE, 4><<<dim3((unsigned)nrows), dim3(256), 0, st>>>(P, ldv, nrows, W, out, done, skip, skip_seq);
    }
}

int bq_launch_gemv(bq_ctx *ctx, const void *panel, int storage, bool add_one, int64_t nrows, int64_t ld,
                   const double *w, double *s_rows, const int *done_flag) {
    if (nrows <= 0) return BQ_OK;
    BQ_ARG(ld % BQ_PAD == 0, "panel pitch must be a multiple of 1024 elements");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_MATVEC, &e0, &e1));
    int *skip = nullptr, skip_seq = 0;
    if (done_flag != nullptr) bq_prof_skip_arg(ctx, e0, &skip, &skip_seq);
    if (storage == BQ_F64) {
        const int64_t ldv = ld / 2;
        if (add_one)
            launch_r<d2_t, true>(ctx->stream, panel, ldv, nrows, w, s_rows, done_flag, skip, skip_seq);
        else
            launch_r<d2_t, false>(ctx->stream, panel, ldv, nrows, w, s_rows, done_flag, skip, skip_seq);
    } else {
        const int64_t ldv = ld / 4;
        if (add_one)
            launch_r<f4_t, true>(ctx->stream, panel, ldv, nrows, w, s_rows, done_flag, skip, skip_seq);
        else
            launch_r<f4_t, false>(ctx->stream, panel, ldv, nrows, w, s_rows, done_flag, skip, skip_seq);
    }
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_prof_end(ctx, BQ_PROF_MATVEC, e0, e1));
    return BQ_OK;
}
