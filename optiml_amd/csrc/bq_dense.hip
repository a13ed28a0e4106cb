// Upload of a dense host Hessian (Quadratic(Q, q), optiml/opti/_base.py:243) into the resident panel.
//
// Two layouts.  ROW BLOCKS: all of Q, this rank's 128-aligned rows, pitch round_up(n, 1024) — what gemv_rows_kernel streams
// (n^2 s bytes per product); the only layout a Q that is NOT symmetric can have (the reference never checks symmetry,
// opti/_base.py:249-256, and `Q @ x` of such a Q is what it computes).  PACKED LOWER TILE ROWS: the layout of the kernel-built
// Gram panels (bq_common.h: bq_sym_addr) — tile row I keeps its (I+1)*256 leading columns — which symv_tiles_kernel streams once
// for both the row and the column contributions: half the HBM and half the bytes per product.  Every Hessian the reference's
// formulas are valid for is symmetric (`Qx + q` is the gradient of 1/2 x'Qx only then, opti/_base.py:291), so the packed layout is
// the default WHEN Q == Q' HOLDS EXACTLY, element for element as stored (fp64 bits; fp32 storage: the rounded values) — checked on
// the device while the rows go up, so a Q that differs from its transpose in one last bit keeps the row blocks and NumPy's product.
//
// The check costs no extra pass over the host matrix: a rank uploads its tile rows from the LAST to the first, whole rows; the
// columns up to the diagonal tile go into the packed panel, the columns beyond it are compared with the transposed tiles that are
// already there (they belong to later tile rows).  A rank of a multi-rank context also uploads the column strip ABOVE its rows
// (rows of earlier ranks, its own columns) and compares it with its lower-left rectangle; the ranks then agree on the outcome with
// one all-reduce of a flag (all of them were handed the same Q).
#include <algorithm>

#include "bq_common.h"

namespace {

template <typename T> struct bits_of;
template <> struct bits_of<double> {
    static __device__ __forceinline__ long long get(double v) { return __double_as_longlong(v); }
};
template <> struct bits_of<float> {
    static __device__ __forceinline__ long long get(float v) { return (long long)__float_as_int(v); }
};

// rows x width block of fp64 values (pitch ldt) -> storage type, row-major with the given pitch
template <typename T>
__global__ __launch_bounds__(256) void dense_pack_kernel(const double *__restrict__ tmp, int64_t ldt, int64_t width,
                                                         T *__restrict__ dst, int64_t pitch) {
    const int64_t r = blockIdx.y;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < width; c += (int64_t)gridDim.x * 256)
        dst[r * pitch + c] = (T)tmp[r * ldt + c];
}

// BQ_DENSE_LOWER: the diagonal tile is stored whole (the tile kernel reads all of it for its row part), so its upper half is
// filled from its lower half — the host's upper triangle is never looked at.  diag: first element of the tile in the panel.
template <typename T>
__global__ __launch_bounds__(256) void dense_mirror_diag_kernel(T *__restrict__ diag, int64_t pitch, int rows) {
    const int r = blockIdx.x, c = threadIdx.x;
    if (r < rows && c < rows && c > r) diag[(int64_t)r * pitch + c] = diag[(int64_t)c * pitch + r];
}

// The uploaded rows [I*256, I*256 + rows) x global columns [c_lo, c_hi) (tmp column 0 is global column c_org) against the
// transpose kept in the packed panel: element (i, c) must equal panel(c, i) as stored.  32 x 32 sub-tiles through LDS so that both
// sides are read along their rows.  Any difference raises *flag (every writer writes 1).
template <typename T>
__global__ __launch_bounds__(256) void dense_symcheck_kernel(const double *__restrict__ tmp, int64_t ldt, int64_t c_org, int64_t I,
                                                             int rows, int64_t c_lo, int64_t c_hi, const T *__restrict__ panel,
                                                             int64_t I0, int *__restrict__ flag) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t cb = c_lo + (int64_t)blockIdx.x * 32;
    const int rb = (int)blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int64_t c = cb + k;
        const int r = rb + tx;
        T v = (T)0;
        if (c < c_hi && r < rows) v = panel[bq_sym_addr(c, I * BQ_SYM_TILE + r, I0)];
        tile[k][tx] = v;
    }
    __syncthreads();
    bool bad = false;
    for (int k = ty; k < 32; k += 8) {
        const int r = rb + k;
        const int64_t c = cb + tx;
        if (r < rows && c < c_hi) {
            const T mine = (T)tmp[(int64_t)r * ldt + (c - c_org)];
            bad = bad || bits_of<T>::get(mine) != bits_of<T>::get(tile[tx][k]);
        }
    }
    if (bad) *flag = 1;
}

template <typename T>
int upload_sym_t(bq_problem *p, const double *Q, bool check, int *symmetric) {
    bq_ctx *c = p->ctx;
    const int64_t n = p->n, T256 = BQ_SYM_TILE;
    *symmetric = 1;
    if (p->I1 <= p->I0) return BQ_OK;   // this rank owns no tile row
    struct scratch {
        double *tmp = nullptr;
        int *flag = nullptr;
        ~scratch() {
            if (tmp) hipFree(tmp);
            if (flag) hipFree(flag);
        }
    } s;
    const int64_t ldt = p->ld;
    BQ_HIP(hipMalloc(&s.tmp, sizeof(double) * (size_t)T256 * (size_t)ldt));
    BQ_HIP(hipMalloc(&s.flag, sizeof(int)));
    BQ_HIP(hipMemsetAsync(s.flag, 0, sizeof(int), c->stream));
    T *panel = reinterpret_cast<T *>(p->panel);
    const int64_t col_end = std::min(n, p->I1 * T256);
    int seen = 0, blocks = 0;
    auto look = [&]() -> int {   // has a difference been seen so far?  (one 4-byte copy; the stream is drained by it)
        BQ_HIP(hipMemcpyAsync(&seen, s.flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        BQ_SYNC(c);
        return BQ_OK;
    };
    for (int64_t I = p->I1 - 1; I >= p->I0; --I) {
        const int rows = (int)std::min<int64_t>(T256, n - I * T256);
        if (rows <= 0) continue;
        const int64_t lower = std::min(n, (I + 1) * T256);
        const int64_t width = check ? col_end : lower;
        // the host call returns when the pageable source has been staged: tmp is free again for the next block only after the
        // kernels of this one, which the stream orders
        BQ_HIP(hipMemcpy2DAsync(s.tmp, (size_t)ldt * 8, Q + I * T256 * n, (size_t)n * 8, (size_t)width * 8, (size_t)rows,
                                hipMemcpyHostToDevice, c->stream));
        dense_pack_kernel<T><<<dim3((unsigned)std::min<int64_t>((lower + 255) / 256, 1024), (unsigned)rows), 256, 0, c->stream>>>(
            s.tmp, ldt, lower, panel + (bq_sym_off(I) - bq_sym_off(p->I0)), bq_sym_pitch(I));
        if (!check)
            dense_mirror_diag_kernel<T><<<(unsigned)rows, 256, 0, c->stream>>>(panel + (bq_sym_off(I) - bq_sym_off(p->I0)) + I * T256,
                                                                                bq_sym_pitch(I), rows);
        if (check) {
            const int64_t c_lo = I * T256;
            dense_symcheck_kernel<T><<<dim3((unsigned)((col_end - c_lo + 31) / 32), (unsigned)((rows + 31) / 32)), 256, 0, c->stream>>>(
                s.tmp, ldt, 0, I, rows, c_lo, col_end, panel, p->I0, s.flag);
            // a Q that is symmetric only up to rounding (a Gram matrix assembled as sklearn does) shows on the first block
            if (blocks == 0 || blocks % 32 == 31) {
                BQ_TRY(look());
                if (seen) break;
            }
        }
        ++blocks;
        BQ_HIP(hipGetLastError());
    }
    // the strip above this rank's rows (rows of the earlier ranks, this rank's columns) against the lower-left rectangle
    if (check && !seen && p->I0 > 0) {
        const int64_t c_lo = p->I0 * T256;
        for (int64_t Ib = 0; Ib < p->I0; ++Ib) {
            BQ_HIP(hipMemcpy2DAsync(s.tmp, (size_t)ldt * 8, Q + Ib * T256 * n + c_lo, (size_t)n * 8, (size_t)(col_end - c_lo) * 8,
                                    (size_t)T256, hipMemcpyHostToDevice, c->stream));
            dense_symcheck_kernel<T><<<dim3((unsigned)((col_end - c_lo + 31) / 32), (unsigned)(T256 / 32)), 256, 0, c->stream>>>(
                s.tmp, ldt, c_lo, Ib, (int)T256, c_lo, col_end, panel, p->I0, s.flag);
            if (Ib % 32 == 31) {
                BQ_TRY(look());
                if (seen) break;
            }
        }
        BQ_HIP(hipGetLastError());
    }
    if (check) {
        BQ_TRY(look());
        *symmetric = seen ? 0 : 1;
    }
    return BQ_OK;
}

}   // namespace

// a cheap look at the host matrix before anything is allocated: 512 pairs (i, j) drawn by a fixed generator — a Q whose
// asymmetry is spread over the matrix (rounding of a Gram assembly) is told apart here, without a panel being allocated for it
bool bq_dense_host_spot_symmetric(const double *Q, int64_t n) {
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        return (int64_t)((st >> 20) % (unsigned long long)n);
    };
    for (int k = 0; k < 512; ++k) {
        const int64_t i = next(), j = next();
        long long a, b;
        memcpy(&a, &Q[i * n + j], 8);
        memcpy(&b, &Q[j * n + i], 8);
        if (a != b) return false;
    }
    return true;
}

// One flag summed over the ranks of a multi-rank context (no-op elsewhere): *any = some rank raised it.  Every rank that entered
// bq_problem_create_dense's packed attempt reaches this, whatever happened to it before — a rank that failed alone must not leave the
// others waiting in the collective.
int bq_dense_agree(bq_ctx *c, bool mine, bool *any) {
    *any = mine;
    if (c->comm_kind == BQ_COMM_NONE || c->comm_kind == BQ_COMM_SHARE) return BQ_OK;
    double *fd = nullptr;
    const double bad = mine ? 1.0 : 0.0;
    if (hipMalloc(&fd, sizeof(double)) != hipSuccess) {
        bq_set_error("cannot allocate the agreement flag");
        return BQ_ERR_HIP;
    }
    int rc = BQ_OK;
    if (hipMemcpyAsync(fd, &bad, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = BQ_ERR_HIP;
    if (rc == BQ_OK) rc = bq_exchange_sum(c, fd, 1);
    double total = 0.0;
    if (rc == BQ_OK && hipMemcpyAsync(&total, fd, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = BQ_ERR_HIP;
    if (rc == BQ_OK) rc = bq_ctx_sync(c);
    hipFree(fd);
    if (rc == BQ_OK) *any = total > 0.0;
    return rc;
}

// p: a dense problem laid out symmetric (problem_layout); fills the packed panel from Q.  check: compare with the transpose while
// uploading — *symmetric == 0 means this rank saw a difference (the caller agrees across ranks and falls back to row blocks)
int bq_dense_upload_sym(bq_problem *p, const double *Q, bool check, int *symmetric) {
    return p->storage == BQ_F64 ? upload_sym_t<double>(p, Q, check, symmetric) : upload_sym_t<float>(p, Q, check, symmetric);
}

__global__ void f64_to_f32_rows_kernel(const double *__restrict__ src, int64_t n, float *__restrict__ dst, int64_t ld) {
    const int64_t r = blockIdx.y;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        dst[r * ld + j] = (float)src[r * n + j];
}

// row blocks: this rank's rows [r0, r1) of Q, all n columns, pitch p->ld
int bq_dense_upload_rows(bq_problem *p, const double *Q) {
    bq_ctx *c = p->ctx;
    const int64_t n = p->n, rows = p->r1 - p->r0;
    if (rows <= 0) return BQ_OK;
    if (p->storage == BQ_F64) {
        BQ_HIP(hipMemcpy2DAsync(p->panel, p->ld * 8, Q + p->r0 * n, n * 8, n * 8, rows, hipMemcpyHostToDevice, c->stream));
        return BQ_OK;
    }
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(256ll << 20) / (n * 8));
    double *tmp = nullptr;
    hipError_t e = hipMalloc(&tmp, sizeof(double) * chunk * n);
    for (int64_t r = 0; e == hipSuccess && r < rows; r += chunk) {
        const int64_t cr = std::min(chunk, rows - r);
        e = hipMemcpyAsync(tmp, Q + (p->r0 + r) * n, sizeof(double) * cr * n, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) break;
        dim3 grid((unsigned)std::min<int64_t>((n + 255) / 256, 1024), (unsigned)cr);
        f64_to_f32_rows_kernel<<<grid, 256, 0, c->stream>>>(tmp, n, (float *)p->panel + r * p->ld, p->ld);
        e = hipStreamSynchronize(c->stream);
    }
    if (tmp) hipFree(tmp);
    if (e != hipSuccess) {
        bq_set_error("panel upload failed: %s", hipGetErrorString(e));
        return BQ_ERR_HIP;
    }
    return BQ_OK;
}
