// O(n) kernels of the dual-QP solvers: fused vector updates, masks and deterministic reductions.
//
// Every vector is replicated on every rank and padded with zeros to a multiple of 1024; block b always owns
// elements [1024 b, 1024 (b+1)), lane t the elements b*1024 + j*256 + t.  Reductions are two-stage with a fixed
// order (4 items per lane -> shuffle tree -> 4 waves -> partials[b] -> one block sums the partials strided by
// 256 + the same tree), so the iterates are bit-identical for any GPU count.  The scalar state of a solver lives
// in one device struct (bq_scal) and the decisions are taken on the device: once `done` is set every later kernel
// of the run returns immediately.  PG / FW: the update before the product is one elementwise pass
// (pgfw_update_kernel); every sum and decision of the iteration belongs to the kernel that closes the product
// (bq_epilogue.h: per 256-row block, the same tree in every closing kernel).
//
// Reference sites: projected_gradient.py:82-129, frank_wolfe.py:96-151, interior_point.py:180-267
// (all under optiml/opti/constrained/), objective/gradient optiml/opti/_base.py:282,291.
#include "bq_common.h"
#include "bq_epilogue.h"

#include <cmath>
#include <cstdlib>


__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// all threads of a 256-thread block get the result
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wsum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double final_sum(const double *part, int64_t nblk, double *sh) {
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < nblk; i += BQ_VEC_BLOCK) a += part[i];
    return block_sum(a, sh);
}
#define VEC_LOOP(i)                                                             \
    const int64_t _base = (int64_t)blockIdx.x * BQ_VEC_TILE + threadIdx.x;      \
    _Pragma("unroll") for (int _j = 0; _j < BQ_VEC_ITEMS; ++_j)                 \
        for (int64_t i = _base + (int64_t)_j * BQ_VEC_BLOCK, _once = 1; _once; _once = 0)

static inline dim3 vec_grid(int64_t ldN) { return dim3((unsigned)(ldN / BQ_VEC_TILE)); }

// ---------------------------------------------------------------------------------------------
// Hessian application: prep (v -> panel input w), finish (gathered panel output s -> Q v)
// ---------------------------------------------------------------------------------------------
__global__ void prep_kernel(int structure, int64_t n, const double *__restrict__ v, const double *__restrict__ sgn,
                            double *__restrict__ w, const int *done) {
    if (done != nullptr && *done) return;
    VEC_LOOP(i) {
        double r = 0.0;
        if (i < n) r = (structure == BQ_SVC) ? sgn[i] * v[i] : v[i] - v[n + i];
        w[i] = r;
    }
}

__global__ void finish_kernel(int structure, int64_t n, int64_t N, double diag_add, const double *__restrict__ s,
                              const double *__restrict__ v, const double *__restrict__ sgn, double *__restrict__ out,
                              const int *done) {
    if (done != nullptr && *done) return;
    VEC_LOOP(i) {
        double r = 0.0;
        if (i < N) {
            if (structure == BQ_PLAIN)
                r = s[i];
            else if (structure == BQ_SVC)
                r = sgn[i] * s[i];
            else
                r = (i < n) ? s[i] : -s[i - n];
            if (diag_add != 0.0) r += diag_add * v[i];
        }
        out[i] = r;
    }
}

// p->s = P w on every rank, P the resident panel (elements optionally +1).  Row-block panels: local rows + all-gather;
// symmetric tile panels: local lower-triangle tiles (both contributions), one partial vector per canonical segment,
// all-gather of the segment vectors + their sum in segment order (or, BQ_SYM_EXCHANGE=allreduce, one all-reduce(sum)).
int bq_panel_product(bq_problem *p, bool add_one, const double *w, const int *done, const bq_epilogue *epi, bool *fused) {
    bq_ctx *ctx = p->ctx;
    if (fused) *fused = false;
    if (p->streamed) {   // no panel: the lower-triangle Gram tiles of this rank's segments recomputed inside the product, each
                         // used for its rows and its columns; exchange exactly as for the resident symmetric panels
        bq_seg_table tab;
        bq_sym_seg_table(p, &tab);
        auto prod = [&](double *out, int mode) {
            return bq_stream_sym_product(ctx, p->stream_img, p->n, p->nb, tab, p->kernel, p->gamma, p->coef0, p->degree, add_one, w, out,
                                         mode, done);
        };
        if (ctx->comm_kind == BQ_COMM_NONE) {
            BQ_TRY(prod(p->s, 0));
        } else if (ctx->sym_allreduce) {
            BQ_TRY(prod(p->s, 0));
            BQ_TRY(bq_exchange_sum(ctx, p->s, p->nb * BQ_SYM_TILE));
        } else {
            BQ_TRY(prod(p->gath, 1));
            BQ_TRY(bq_exchange_gather(ctx, p->gath, (int64_t)p->seg_cmax * p->nb * BQ_SYM_TILE));
            BQ_TRY(bq_launch_symv_segsum(ctx, p->nb, tab, p->gath, p->s, done, epi));
            if (fused) *fused = epi != nullptr;
        }
        return BQ_OK;
    }
    if (p->symmetric) {
        bq_seg_table tab;
        bq_sym_seg_table(p, &tab);
        if (ctx->comm_kind == BQ_COMM_NONE) {
            BQ_TRY(bq_launch_symv(ctx, p->panel, p->storage, add_one, p->nb, tab, w, p->slab, p->s, done, epi));
            if (fused) *fused = epi != nullptr;
        } else if (ctx->sym_allreduce) {   // rank partials meet in one all-reduce(sum): association depends on the transport
            BQ_TRY(bq_launch_symv(ctx, p->panel, p->storage, add_one, p->nb, tab, w, p->slab, p->s, done));
            BQ_TRY(bq_exchange_sum(ctx, p->s, p->nb * BQ_SYM_TILE));
        } else {   // default: all-gather of the segment vectors, summed in segment order on every rank (bit-identical for any world)
            BQ_TRY(bq_launch_symv_segments(ctx, p->panel, p->storage, add_one, p->nb, tab, w, p->slab, p->gath, done));
            BQ_TRY(bq_exchange_gather(ctx, p->gath, (int64_t)p->seg_cmax * p->nb * BQ_SYM_TILE));
            BQ_TRY(bq_launch_symv_segsum(ctx, p->nb, tab, p->gath, p->s, done, epi));
            if (fused) *fused = epi != nullptr;
        }
    } else {
        BQ_TRY(bq_launch_gemv(ctx, p->panel, p->storage, add_one, p->r1 - p->r0, p->ld, w, p->s + p->r0, done));
        if (ctx->comm_kind != BQ_COMM_NONE) BQ_TRY(bq_exchange_rows(ctx, p->s, p->n, p->blk, p->r0, p->r1));
    }
    return BQ_OK;
}

int bq_launch_prep(bq_problem *p, const double *v, const int *done) {
    prep_kernel<<<vec_grid(p->ld), BQ_VEC_BLOCK, 0, p->ctx->stream>>>(p->structure, p->n, v, p->sgn, p->w, done);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

int bq_problem_apply(bq_problem *p, const double *v, double *out, const int *done) {
    bq_ctx *ctx = p->ctx;
    const double *w = v;
    if (p->structure != BQ_PLAIN) {
        prep_kernel<<<vec_grid(p->ld), BQ_VEC_BLOCK, 0, ctx->stream>>>(p->structure, p->n, v, p->sgn, p->w, done);
        w = p->w;
    }
    BQ_TRY(bq_panel_product(p, p->add_one, w, done));
    finish_kernel<<<vec_grid(p->ldN), BQ_VEC_BLOCK, 0, ctx->stream>>>(p->structure, p->n, p->N, p->diag_add, p->s, v,
                                                                      p->sgn, out, done);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

// f = 1/2 x'(Qx) + q'x, g = Qx + q       (host-vector entry point bq_problem_eval)
__global__ void evalf_partial_kernel(int64_t N, const double *__restrict__ x, const double *__restrict__ Qx,
                                     const double *__restrict__ q, double *__restrict__ g, double *__restrict__ part,
                                     int64_t nblk) {
    __shared__ double sh[4];
    double a = 0.0, b = 0.0;
    VEC_LOOP(i) {
        if (i < N) {
            a += x[i] * Qx[i];
            b += q[i] * x[i];
            if (g != nullptr) g[i] = Qx[i] + q[i];
        }
    }
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = a;
        part[nblk + blockIdx.x] = b;
    }
}
__global__ void evalf_final_kernel(const double *part, int64_t nblk, double *f) {
    __shared__ double sh[4];
    double a = final_sum(part, nblk, sh);
    double b = final_sum(part + nblk, nblk, sh);
    if (threadIdx.x == 0) *f = 0.5 * a + b;
}

int bq_vec_eval_f(bq_problem *p, const double *x, const double *Qx, double *g_out, double *f_dev) {
    const int64_t nblk = p->ldN / BQ_VEC_TILE;
    evalf_partial_kernel<<<vec_grid(p->ldN), BQ_VEC_BLOCK, 0, p->ctx->stream>>>(p->N, x, Qx, p->q, g_out, p->partials, nblk);
    evalf_final_kernel<<<1, BQ_VEC_BLOCK, 0, p->ctx->stream>>>(p->partials, nblk, f_dev);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// Projected gradient and Frank-Wolfe
// ---------------------------------------------------------------------------------------------
struct vecs {
    double *x, *g, *d, *Qd, *q, *lb, *ub;
};

// g = Qx + q from the product already in Qd (first evaluation only)
__global__ void grad_init_kernel(int64_t N, vecs V) {
    VEC_LOOP(i) {
        if (i < N) V.g[i] = V.Qd[i] + V.q[i];
    }
}

// An iteration's update (projected_gradient.py:81-98, frank_wolfe.py:96-110): x += t d, g += t Qd, the new direction, and — BQ_SVC —
// the panel-product input w = y o d.  One elementwise pass, no sums: they are taken, with the decisions, by the kernel that closes
// the product (bq_epilogue.h).
__global__ __launch_bounds__(256) void pgfw_update_kernel(bq_epilogue e, double *__restrict__ w_out) {
    if (e.sc->done) return;
    double t, tr;
    bq_epi_scalars(e, t, tr);
    const bool upd = e.do_update != 0;
    // one panel column per thread — its dual element, or its two (BQ_SVR: i and n + i) — so the kernel is one round trip long and
    // writes the product's input w itself for every structure (prep_kernel's map: y o d, d+ - d-, or d)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= e.n) return;
    const bool two = e.structure == BQ_SVR;
    const bq_pgfw_raw r0 = bq_pgfw_load(e, i, upd), r1 = bq_pgfw_load(e, two ? e.n + i : i, upd);
    const bq_pgfw_elem a = bq_pgfw_compute(e.kind, r0, upd, t, tr);
    if (upd) {
        e.x[i] = a.x;
        e.g[i] = a.g;
    }
    e.d[i] = a.d;
    double w = a.d;
    if (two) {
        const bq_pgfw_elem b = bq_pgfw_compute(e.kind, r1, upd, t, tr);
        if (upd) {
            e.x[e.n + i] = b.x;
            e.g[e.n + i] = b.g;
        }
        e.d[e.n + i] = b.d;
        w = a.d - b.d;
    } else if (e.structure == BQ_SVC) {
        w = e.sgn[i] * a.d;
    }
    if (w_out != nullptr) w_out[i] = w;
}

// The stand-alone closing kernel of an iteration (bq_epilogue.h) for the paths whose product has no closing kernel of its own to
// carry it (dense row-block panels, the one-rank streamed product, BQ_SYM_EXCHANGE=allreduce).  One workgroup per 256 rows, exactly
// as inside symv_reduce_kernel / symv_segsum_kernel: the sums — and with them the step length — have the same bits whichever
// kernel closed the product.
__global__ __launch_bounds__(256) void finish_den_kernel(const double *__restrict__ sv, bq_epilogue epi) {
    if (epi.sc->done) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bq_epi_pre pre = bq_epi_preload(epi, i, true);
    bq_epi_finish(epi, blockIdx.x, gridDim.x, bq_epi_element(epi, pre, i, i < epi.n ? sv[i] : 0.0), gridDim.x);
}

static vecs solver_vecs(bq_solver *s) {
    vecs V;
    V.x = s->x;
    V.g = s->g;
    V.d = s->d;
    V.Qd = s->Qd;
    V.q = s->p->q;
    V.lb = s->lb;
    V.ub = s->ub;
    return V;
}

int bq_pgfw_start(bq_solver *s) {
    // g = Q x0 + q : the only full-gradient product of the run (later iterations update g incrementally)
    hipStream_t st = s->p->ctx->stream;
    BQ_TRY(bq_problem_apply(s->p, s->x, s->Qd, nullptr));
    grad_init_kernel<<<vec_grid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, solver_vecs(s));
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

int bq_pgfw_iterate(bq_solver *s) {
    hipStream_t st = s->p->ctx->stream;
    const int *done = &s->sc->done;
    bq_problem *p = s->p;
    bq_epilogue epi;
    epi.structure = p->structure;
    epi.kind = s->kind == BQ_PG ? 0 : 1;
    epi.do_update = s->started ? 1 : 0;
    epi.n = p->n;
    epi.N = p->N;
    epi.diag_add = p->diag_add;
    epi.x = s->x;
    epi.g = s->g;
    epi.d = s->d;
    epi.q = p->q;
    epi.lb = s->lb;
    epi.ub = s->ub;
    epi.sgn = p->sgn;
    epi.Qd = s->Qd;
    epi.sc = s->sc;
    epi.part = s->partials;
    epi.stats = s->stats;
    static_assert(BQ_MAX_PARTIAL_Q >= 20, "five sums per block of 256 rows = 20 per block of 1024");
    s->started = true;
    // BQ_PLAIN: the product's input is the direction itself (same padded length as the panel width); else the kernel writes it
    const double *w = p->structure == BQ_PLAIN ? s->d : p->w;
    pgfw_update_kernel<<<(unsigned)((p->n + 255) / 256), 256, 0, st>>>(epi, p->structure == BQ_PLAIN ? nullptr : p->w);
    bool fused = false;
    BQ_TRY(bq_panel_product(p, p->add_one, w, done, &epi, &fused));
    if (!fused) finish_den_kernel<<<(unsigned)((p->n + 255) / 256), 256, 0, st>>>(p->s, epi);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
