// The kernel that closes the panel product of a PG / FW iteration (projected_gradient.py:81-129, frank_wolfe.py:96-151): the slab
// reduction of one rank (symv_reduce_kernel), the ordered segment sum after the all-gather (symv_segsum_kernel) or the stand-alone
// finish_den_kernel goes on, for its own 256 rows, to
//   Qd = structure(s) (+ diag_add d),  the block's partial sums of d'Qd AND of the iteration's other reductions (|d|^2 or FW's
//   g'(y - x), g'd, x'(g + q), PG's minimum ratio), re-derived from the element's x, g, bounds (bq_pgfw_compute: the function the
//   update kernel stored d with),
// and its LAST block (a ticket) takes the iteration's decisions — objective, |d| or the gap, the record, the stop tests — and the
// step length t = min(-g'd / d'Qd, cap).  Round 3 had a separate launch for d'Qd and t; round 4 fused that here; round 5 moved the
// other sums and the decisions here too: the kernel before the product (pgfw_update_kernel, bq_vec.hip) is one elementwise pass
// without sums, ticket or second reduction (config 2: 28.3 -> 24.8 us per iteration beside the product).  The price is one product
// enqueued past the iteration that stops.  The partial sums are per 256-row block by the block's first four waves in EVERY closing
// kernel (same lanes, same tree) and are added by the last block strided by 256 in a fixed order: the iterates have the same bits on
// one rank and on any number of ranks.
// (Also tried in round 5 and NOT kept: the tile kernel forming its input from the state in a prologue, which makes the update kernel
// unnecessary — two launches per iteration — but costs a single-round grid more than it saves: profiles/r05/pgfw_fused_tile_prologue_probe.txt.)
#pragma once
#include "bq_common.h"

#define BQ_CURV_TOL 1e-16
#define BQ_ACT_TOL 1e-12

struct bq_epilogue {
    int structure;            // BQ_PLAIN / BQ_SVC / BQ_SVR
    int kind;                 // 0: PG, 1: FW
    int do_update;            // a step is pending (every iteration but a solver's first): pgfw_update_kernel applies it
    long long n, N;
    double diag_add;
    double *x, *g, *d;        // the state (N)
    const double *q, *lb, *ub, *sgn;   // linear term, box (N); labels (n, BQ_SVC)
    double *Qd;               // update kernel: Q d of the pending step (in); closing kernel: Q d of the new direction (out; N, padded to ldN)
    bq_scal *sc;
    double *part;             // 5 x nblocks partial sums (nblocks = the closing kernel's blocks of 256 rows)
    bq_iter_stat *stats;
};

__device__ __forceinline__ double bq_epi_wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double bq_epi_wmin(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}

// element u of the new iterate from element u of the old state (x, g, d, Qd and the pending step t); y0 = FW's unclipped vertex
// (its lower bound needs it).  `update` clear: (x, g) is the new iterate already and only d is re-derived.  Loading and arithmetic
// are separate so that a caller with two elements (BQ_SVR) issues all loads before the first use.
struct bq_pgfw_raw {
    double x, g, d, Qd, lb, ub;
};
struct bq_pgfw_elem {
    double x, g, d, y0;
};
__device__ __forceinline__ bq_pgfw_raw bq_pgfw_load(const bq_epilogue &e, long long u, bool update) {
    bq_pgfw_raw r;
    r.x = e.x[u];
    r.g = e.g[u];
    r.lb = e.lb[u];
    r.ub = e.ub[u];
    r.d = update ? e.d[u] : 0.0;
    r.Qd = update ? e.Qd[u] : 0.0;
    return r;
}
__device__ __forceinline__ bq_pgfw_elem bq_pgfw_compute(int kind, const bq_pgfw_raw &in, bool update, double t, double tr) {
    bq_pgfw_elem r;
    double xi = in.x, gi = in.g;
    if (update) {
        xi = xi + __dmul_rn(t, in.d);
        gi = gi + __dmul_rn(t, in.Qd);
    }
    double di;
    if (kind == 0) {
        di = -gi;
        if (in.ub - xi <= BQ_ACT_TOL && di > 0.0) di = 0.0;
        if (xi - in.lb <= BQ_ACT_TOL && di < 0.0) di = 0.0;
        r.y0 = 0.0;
    } else {
        double yi = (gi < 0.0) ? in.ub : in.lb;
        r.y0 = yi;
        if (tr > 0.0) {
            const double rad = tr * (in.ub - in.lb);
            yi = fmin(fmax(yi, xi - rad), xi + rad);
        }
        di = yi - xi;
    }
    r.x = xi;
    r.g = gi;
    r.d = di;
    return r;
}
__device__ __forceinline__ bq_pgfw_elem bq_pgfw_element(const bq_epilogue &e, long long u, bool update, double t, double tr) {
    return bq_pgfw_compute(e.kind, bq_pgfw_load(e, u, update), update, t, tr);
}

// the pending step and FW's trust radius
__device__ __forceinline__ void bq_epi_scalars(const bq_epilogue &e, double &t, double &tr) {
    t = e.do_update ? e.sc->t : 0.0;
    tr = e.sc->fw_t;
}

// a thread's contributions to the iteration's sums
struct bq_epi_sums {
    double den, a, gd, xg, rmin;   // d'Qd;  PG |d|^2 / FW g'(y0 - x);  g'd;  x'(g + q);  PG min ratio
};
__device__ __forceinline__ bq_epi_sums bq_epi_zero() { return bq_epi_sums{0.0, 0.0, 0.0, 0.0, INFINITY}; }

// what an output row's epilogue reads that does not depend on the product: loaded by the closing kernels BEFORE they sum the product
// (the loads then overlap the slab walk / the gathered segment loads instead of following them: one round trip less per iteration)
struct bq_epi_pre {
    bq_pgfw_raw r0, r1;   // the element(s) of row i: u0 = i, and u1 = n + i for BQ_SVR (else u1 = u0: the same lines)
    double q0, q1, sg, tr;
    bool live;
};
__device__ __forceinline__ bq_epi_pre bq_epi_preload(const bq_epilogue &e, long long i, bool active) {
    bq_epi_pre p = {};
    p.live = active && i < e.n;
    if (!p.live) return p;   // (symv_reduce_kernel: three of a block's four 256-thread groups carry no row)
    const long long u0 = i, u1 = e.structure == BQ_SVR ? e.n + u0 : u0;
    p.r0 = bq_pgfw_load(e, u0, false);
    p.r1 = bq_pgfw_load(e, u1, false);
    p.q0 = e.q[u0];
    p.q1 = e.q[u1];
    p.sg = e.structure == BQ_SVC ? e.sgn[u0] : 1.0;
    p.tr = e.sc->fw_t;
    return p;
}

// one output row: i = a * 256 + r of the n-vector s (both halves for BQ_SVR): Qd and the element's contributions to the sums
__device__ __forceinline__ bq_epi_sums bq_epi_element(const bq_epilogue &e, const bq_epi_pre &p, long long i, double sv) {
    bq_epi_sums c = bq_epi_zero();
    if (!p.live) return c;
    const bool two = e.structure == BQ_SVR;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 0 || two) {
            const long long u = h == 0 ? i : e.n + i;
            const bq_pgfw_raw &in = h == 0 ? p.r0 : p.r1;
            const bq_pgfw_elem el = bq_pgfw_compute(e.kind, in, false, 0.0, p.tr);   // d as the update kernel stored it, and FW's y0
            double r;
            if (two)
                r = h == 0 ? sv : -sv;
            else
                r = e.structure == BQ_SVC ? p.sg * sv : sv;
            if (e.diag_add != 0.0) r += e.diag_add * el.d;
            e.Qd[u] = r;
            c.den += el.d * r;
            c.gd += el.g * el.d;
            c.xg += el.x * (el.g + (h == 0 ? p.q0 : p.q1));
            if (e.kind == 0) {
                c.a += el.d * el.d;
                if (el.d > 0.0) c.rmin = fmin(c.rmin, (in.ub - el.x) / el.d);
                if (el.d < 0.0) c.rmin = fmin(c.rmin, (in.lb - el.x) / el.d);
            } else {
                c.a += el.g * (el.y0 - el.x);
            }
        }
    }
    return c;
}

// called by ALL threads of the workgroup (>= 256 threads; threads 0 .. 255 carry the contributions `c` of block `a`'s 256 rows):
// block partials -> part[q][a]; the last block of the grid takes the iteration's decisions and the step length
__device__ __forceinline__ void bq_epi_finish(const bq_epilogue &e, long long a, long long nblocks, bq_epi_sums c, unsigned int nwg) {
    __shared__ double sh[5][4];
    __shared__ int last;
    const int tid = threadIdx.x;
    if (tid < 256) {
        c.den = bq_epi_wsum(c.den);
        c.a = bq_epi_wsum(c.a);
        c.gd = bq_epi_wsum(c.gd);
        c.xg = bq_epi_wsum(c.xg);
        c.rmin = bq_epi_wmin(c.rmin);
        if ((tid & 63) == 0) {
            const int w = tid >> 6;
            sh[0][w] = c.den;
            sh[1][w] = c.a;
            sh[2][w] = c.gd;
            sh[3][w] = c.xg;
            sh[4][w] = c.rmin;
        }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e.part[q * nblocks + a] = ((sh[q][0] + sh[q][1]) + sh[q][2]) + sh[q][3];
        e.part[4 * nblocks + a] = fmin(fmin(sh[4][0], sh[4][1]), fmin(sh[4][2], sh[4][3]));
        __threadfence();   // this block's partial sums are visible device-wide before the ticket is taken
        last = atomicAdd(&e.sc->ticket[1], 1u) == nwg - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // The last block of the grid closes the iteration: the sums over the blocks' partial sums (strided by 256, wave tree, four waves
    // in order), then projected_gradient.py:99-129 / frank_wolfe.py:111-151 — objective, |d| or the gap, the record, the stop tests,
    // the step length.  A stop leaves the state at the iterate it was decided on, with no step pending.
    double acc[4] = {0.0, 0.0, 0.0, 0.0}, am = INFINITY;
    if (tid < 256) {
        for (long long k = tid; k < nblocks; k += 256) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += e.part[q * nblocks + k];
            am = fmin(am, e.part[4 * nblocks + k]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = bq_epi_wsum(acc[q]);
        am = bq_epi_wmin(am);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sh[q][tid >> 6] = acc[q];
            sh[4][tid >> 6] = am;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double den = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
        const double ra = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
        const double sgd = ((sh[2][0] + sh[2][1]) + sh[2][2]) + sh[2][3];
        const double sxg = ((sh[3][0] + sh[3][1]) + sh[3][2]) + sh[3][3];
        const double rmin = fmin(fmin(sh[4][0], sh[4][1]), fmin(sh[4][2], sh[4][3]));
        bq_scal *sc = e.sc;
        const double f = 0.5 * sxg;
        bq_iter_stat st;
        st.iter = sc->iter;
        st.f = f;
        bool stop;
        sc->f = f;
        sc->gd = sgd;
        if (e.kind == 0) {
            const double ng = sqrt(ra);
            sc->ng = ng;
            sc->max_t = rmin;
            st.r1 = ng;
            st.r2 = NAN;
            st.r3 = rmin;
            stop = ng <= sc->eps;
        } else {
            const double low = f + ra;
            if (low > sc->best_lb) sc->best_lb = low;
            const double gap = (f - sc->best_lb) / fmax(fabs(f), 1.0);
            sc->low = low;
            sc->gap = gap;
            st.r1 = sc->best_lb;
            st.r2 = gap;
            st.r3 = NAN;
            stop = gap <= sc->eps;
        }
        if (stop) {
            sc->status = BQ_STATUS_OPTIMAL;
            sc->done = 1;
        } else if (sc->iter >= sc->max_iter) {
            sc->status = BQ_STATUS_STOPPED;
            sc->done = 1;
            stop = true;
        }
        if (!stop) {
            const double cap = (e.kind == 0) ? rmin : 1.0;
            const double t = (den <= BQ_CURV_TOL) ? cap : fmin(-sgd / den, cap);
            sc->den = den;
            sc->t = t;
            if (e.kind == 0)
                st.r2 = t;
            else
                st.r3 = t;
        }
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) e.stats[row] = st;
        if (!stop) sc->iter += 1;
        sc->ticket[1] = 0;
    }
}
