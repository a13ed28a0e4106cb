// Epilogue of a panel product inside a PG / FW iteration, fused into the kernel that finishes the product (round 4): the slab
// reduction of a single rank (symv_reduce_kernel) or the ordered segment sum after the all-gather (symv_segsum_kernel) goes on to
//   Qd = structure(s) (+ diag_add d),   partial sums of d'Qd per 256-row block,   and — in the last block to finish — the step
//   length t = min(-g'd / d'Qd, cap) that closes the iteration (projected_gradient.py:112-129, frank_wolfe.py:130-151),
// which round 3 did in a separate launch (finish_den_kernel: 9 us + a dependent-launch gap per iteration).  The partial sums are
// taken per 256-row block by the block's first four waves in BOTH kernels (same lanes, same tree), and added by the last block
// strided by 256 in a fixed order: the step length has the same bits on one rank and on any number of ranks.
#pragma once
#include "bq_common.h"

#define BQ_CURV_TOL 1e-16

struct bq_epilogue {
    int structure;            // BQ_PLAIN / BQ_SVC / BQ_SVR
    int kind;                 // 0: PG, 1: FW
    long long n, N;
    double diag_add;
    const double *d, *sgn;    // the direction (N); labels (BQ_SVC)
    double *Qd;               // out (N, padded to ldN)
    bq_scal *sc;
    double *part;             // >= nb partial sums
    bq_iter_stat *stats;
    const double *dec;        // the update / evaluation kernel's per-block partial sums (4 x nblk: PG |d|^2, g'd, x'(g+q), min ratio; FW g'(y-x), g'd,
    long long nblk;           //   x'(g+q)), reduced and DECIDED on here since round 5 (was: by a ticket + last block inside that kernel)
};

__device__ __forceinline__ double bq_epi_wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// one output element: i = a * 256 + r of the n-vector s -> Qd (both halves for BQ_SVR); returns its contribution to d'Qd
__device__ __forceinline__ double bq_epi_element(const bq_epilogue &e, long long i, double sv) {
    double c = 0.0;
    if (i < e.n) {
        if (e.structure == BQ_SVR) {
            const double d0 = e.d[i], d1 = e.d[e.n + i];
            double r0 = sv, r1 = -sv;
            if (e.diag_add != 0.0) {
                r0 += e.diag_add * d0;
                r1 += e.diag_add * d1;
            }
            e.Qd[i] = r0;
            e.Qd[e.n + i] = r1;
            c = d0 * r0 + d1 * r1;
        } else {
            double r = e.structure == BQ_SVC ? e.sgn[i] * sv : sv;
            const double di = e.d[i];
            if (e.diag_add != 0.0) r += e.diag_add * di;
            e.Qd[i] = r;
            c = di * r;
        }
    }
    return c;
}

// called by ALL threads of the workgroup (>= 256 threads; threads 0 .. 255 carry the contributions `c` of block `a`'s 256 rows):
// block partial -> part[a]; the last block of the grid turns the partials into the step length and closes the iteration
__device__ __forceinline__ void bq_epi_finish(const bq_epilogue &e, long long a, long long nblocks, double c, unsigned int nwg) {
    __shared__ double sh[4];
    __shared__ int last;
    const int tid = threadIdx.x;
    if (tid < 256) {
        c = bq_epi_wsum(c);
        if ((tid & 63) == 0) sh[tid >> 6] = c;
    }
    __syncthreads();
    if (tid == 0) {
        e.part[a] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        __threadfence();   // this block's partial sum is visible device-wide before the ticket is taken
        last = atomicAdd(&e.sc->ticket[1], 1u) == nwg - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // The last block of the grid closes the iteration.  Round 5: it first takes the decisions that the update / evaluation kernel's own
    // last block took before (projected_gradient.py:99-110, frank_wolfe.py:111-128: objective, |d| or the gap, the record, the stop
    // tests) — that kernel is now one pass without a ticket, a fence and a dependent second reduction (~5 us of its 12 per iteration);
    // the price is one product enqueued past the iteration that stops.  Same sums in the same association as pg_decide_body /
    // fw_decide_body had (strided by 256 over the blocks' partial sums, wave tree, four waves in order).
    __shared__ double shd[5][4];
    double acc = 0.0, da = 0.0, db = 0.0, dc = 0.0, dm = INFINITY;
    if (tid < 256) {
        for (long long k = tid; k < nblocks; k += 256) acc += e.part[k];
        for (long long k = tid; k < e.nblk; k += 256) {
            da += e.dec[0 * e.nblk + k];
            db += e.dec[1 * e.nblk + k];
            dc += e.dec[2 * e.nblk + k];
            if (e.kind == 0) dm = fmin(dm, e.dec[3 * e.nblk + k]);
        }
        acc = bq_epi_wsum(acc);
        da = bq_epi_wsum(da);
        db = bq_epi_wsum(db);
        dc = bq_epi_wsum(dc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dm = fmin(dm, __shfl_down(dm, off, 64));
        if ((tid & 63) == 0) {
            shd[0][tid >> 6] = acc;
            shd[1][tid >> 6] = da;
            shd[2][tid >> 6] = db;
            shd[3][tid >> 6] = dc;
            shd[4][tid >> 6] = dm;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const double den = ((shd[0][0] + shd[0][1]) + shd[0][2]) + shd[0][3];
        const double ra = ((shd[1][0] + shd[1][1]) + shd[1][2]) + shd[1][3];
        const double sgd = ((shd[2][0] + shd[2][1]) + shd[2][2]) + shd[2][3];
        const double sxg = ((shd[3][0] + shd[3][1]) + shd[3][2]) + shd[3][3];
        const double rmin = fmin(fmin(shd[4][0], shd[4][1]), fmin(shd[4][2], shd[4][3]));
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
