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
    double acc = 0.0;
    if (tid < 256)
        for (long long k = tid; k < nblocks; k += 256) acc += e.part[k];
    if (tid < 256) {
        acc = bq_epi_wsum(acc);
        if ((tid & 63) == 0) sh[tid >> 6] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        const double den = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        bq_scal *sc = e.sc;
        const double cap = (e.kind == 0) ? sc->max_t : 1.0;
        const double t = (den <= BQ_CURV_TOL) ? cap : fmin(-sc->gd / den, cap);
        sc->den = den;
        sc->t = t;
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            if (e.kind == 0)
                e.stats[row].r2 = t;
            else
                e.stats[row].r3 = t;
        }
        sc->iter += 1;
        sc->ticket[1] = 0;
    }
}
