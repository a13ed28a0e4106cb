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
    int kind;                 // 0: PG, 1: FW, 2: the evaluation of an augmented-Lagrangian iteration (bq_al_epi_*, below)
    int do_update;            // a step is pending (every iteration but a solver's first): pgfw_update_kernel applies it
    long long n, N;
    double diag_add;
    double *x, *g, *d;        // the state (N)
    const double *q, *lb, *ub, *sgn;   // linear term, box (N); labels (n, BQ_SVC)
    double *Qd;               // update kernel: Q d of the pending step (in); closing kernel: Q d of the new direction (out; N, padded to ldN)
    bq_scal *sc;
    double *part;             // 5 x nblocks partial sums (nblocks = the closing kernel's blocks of 256 rows)
    bq_iter_stat *stats;
    // kind 2 only: x is the point (the product's input is its structure map), Qd receives Q x; lb / ub may be null there
    const double *a, *llb, *lub;   // equality row; multipliers of the bound rows (null: that family is absent)
    const double *chk;             // 3 x ldN: the last update's per-element terms of the stop test
    long long ldN;
    double rho, tol;
    long long epochs;
    int has_rows;                  // some constraint family is present (else nothing ever stops but the epoch count)
};
#define BQ_EPI_NONE 0
#define BQ_EPI_PGFW 1
#define BQ_EPI_AL 2
static inline int bq_epi_mode(const bq_epilogue *e) { return e == nullptr ? BQ_EPI_NONE : (e->kind == 2 ? BQ_EPI_AL : BQ_EPI_PGFW); }

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

// ---------------------------------------------------------------------------------------------------------------------------
// kind 2: an augmented-Lagrangian iteration (optiml/opti/constrained/_base.py:326-340, the callback of optiml/opti/_base.py:96-117,
// the epoch test of the stochastic rules, e.g. adagrad.py:85-101, and the multiplier step / stop tests of optiml/opti/_base.py:129-146)
// in the kernel that closes the product Q x.  For the block's 256 rows: Qx = structure(s) (+ diag_add x), the partial sums of the
// EVALUATION at x (x'Qx, q'x, a'x, dual'c, |max(c_in, 0)|^2) and of the STOP TEST the previous update left pending (|c(x)|^2 over
// the bound rows, |d dual|^2, |d x|^2: per-element terms the update kernel stored; a'x is shared).  The LAST block first closes the
// previous iteration — multiplier of the equality row, the two stop tests, iter += 1 — and then, unless that stopped the solve,
// takes this iteration's value, primal value, record and epoch test.  Rounds 2-5 ran finish_kernel + al_eval_kernel behind the
// closing kernel and a second reduce-and-decide chain inside the update kernel; now the update kernel is one elementwise pass and an
// iteration has ONE chain of sums (the structure PG / FW got in round 5).  The price, as there: one product enqueued past the
// iteration that stops.  Same partial-sum layout and tree in every closing kernel and in al_flush_kernel (which closes the last
// iteration of a bq_solver_run): identical bits however the iterations are cut into runs, on any rank count.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int BQ_AL_NQ = 8;
struct bq_al_pre {
    double x[2], q[2], a[2], lb[2], ub[2], ll[2], lu[2], ck[2][3], sg;
    bool live, pending;
};
// eval: the evaluation's operands too (the closing kernels); else only what the pending stop test needs (al_flush_kernel)
__device__ __forceinline__ bq_al_pre bq_al_epi_preload(const bq_epilogue &e, long long i, bool active, bool eval) {
    bq_al_pre p = {};
    p.pending = e.sc->al_pending != 0;
    p.live = active && i < e.n;
    if (!p.live) return p;
    const int halves = e.structure == BQ_SVR ? 2 : 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h < halves) {
            const long long u = h == 0 ? i : e.n + i;
            p.x[h] = e.x[u];
            if (e.a) p.a[h] = e.a[u];
            if (p.pending) {
#pragma unroll
                for (int k = 0; k < 3; ++k) p.ck[h][k] = e.chk[k * e.ldN + u];
            }
            if (eval) {
                p.q[h] = e.q[u];
                if (e.lb) {
                    p.lb[h] = e.lb[u];
                    p.ll[h] = e.llb[u];
                }
                if (e.ub) {
                    p.ub[h] = e.ub[u];
                    p.lu[h] = e.lub[u];
                }
            }
        }
    }
    p.sg = e.structure == BQ_SVC ? e.sgn[i] : 1.0;
    return p;
}

struct bq_al_sums {
    double v[BQ_AL_NQ];   // x'Qx, q'x, a'x, dual'c of the bound rows, |max(c_in, 0)|^2;  pending: |c|^2 of the bound rows, |d dual|^2, |d x|^2
};
__device__ __forceinline__ bq_al_sums bq_al_epi_element(const bq_epilogue &e, const bq_al_pre &p, long long i, double sv, bool eval) {
    bq_al_sums c;
#pragma unroll
    for (int q = 0; q < BQ_AL_NQ; ++q) c.v[q] = 0.0;
    if (!p.live) return c;
    const bool two = e.structure == BQ_SVR;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 0 || two) {
            const long long u = h == 0 ? i : e.n + i;
            const double x = p.x[h];
            if (e.a) c.v[2] += p.a[h] * x;
            if (p.pending) {
#pragma unroll
                for (int k = 0; k < 3; ++k) c.v[5 + k] += p.ck[h][k];
            }
            if (eval) {
                double r;
                if (two)
                    r = h == 0 ? sv : -sv;
                else
                    r = e.structure == BQ_SVC ? p.sg * sv : sv;
                if (e.diag_add != 0.0) r += e.diag_add * x;
                e.Qd[u] = r;
                c.v[0] += x * r;
                c.v[1] += p.q[h] * x;
                if (e.lb) {
                    const double cc = p.lb[h] - x;
                    c.v[3] += p.ll[h] * cc;
                    if (cc > 0.0) c.v[4] += cc * cc;
                }
                if (e.ub) {
                    const double cc = x - p.ub[h];
                    c.v[3] += p.lu[h] * cc;
                    if (cc > 0.0) c.v[4] += cc * cc;
                }
            }
        }
    }
    return c;
}

// called by ALL threads of the workgroup (threads 0 .. 255 carry block `a`'s contributions), as bq_epi_finish.  EVAL: the closing
// kernels (pending stop test, then the evaluation); else al_flush_kernel (the pending stop test alone)
template <bool EVAL>
__device__ __forceinline__ void bq_al_epi_finish(const bq_epilogue &e, long long a, long long nblocks, bq_al_sums c, unsigned int nwg) {
    __shared__ double sh[BQ_AL_NQ][4];
    __shared__ int last;
    const int tid = threadIdx.x;
    if (tid < 256) {
#pragma unroll
        for (int q = 0; q < BQ_AL_NQ; ++q) c.v[q] = bq_epi_wsum(c.v[q]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < BQ_AL_NQ; ++q) sh[q][tid >> 6] = c.v[q];
        }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < BQ_AL_NQ; ++q) e.part[q * nblocks + a] = ((sh[q][0] + sh[q][1]) + sh[q][2]) + sh[q][3];
        __threadfence();
        last = atomicAdd(&e.sc->ticket[0], 1u) == nwg - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double acc[BQ_AL_NQ];
#pragma unroll
    for (int q = 0; q < BQ_AL_NQ; ++q) acc[q] = 0.0;
    if (tid < 256) {
        for (long long k = tid; k < nblocks; k += 256) {
#pragma unroll
            for (int q = 0; q < BQ_AL_NQ; ++q) acc[q] += e.part[q * nblocks + k];
        }
#pragma unroll
        for (int q = 0; q < BQ_AL_NQ; ++q) acc[q] = bq_epi_wsum(acc[q]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < BQ_AL_NQ; ++q) sh[q][tid >> 6] = acc[q];
        }
    }
    __syncthreads();
    if (tid != 0) return;
    double t[BQ_AL_NQ];
#pragma unroll
    for (int q = 0; q < BQ_AL_NQ; ++q) t[q] = ((sh[q][0] + sh[q][1]) + sh[q][2]) + sh[q][3];
    bq_scal *sc = e.sc;
    sc->ticket[0] = 0;
    const double ax = t[2];
    if (sc->al_pending) {
        // the previous iteration's close (optiml/opti/_base.py:129-146): multiplier of the equality row from a'x at the new point
        // (this very sum), the two stop tests, iter += 1
        double cn = t[5], dl = t[6];
        const double dx = t[7];
        if (e.a) {
            const double dmu = e.rho * ax;
            sc->al_mu = sc->al_mu + dmu;
            cn += ax * ax;
            dl += dmu * dmu;
        }
        const double cnorm = sqrt(cn), moved = sqrt(dl) + sqrt(dx);
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            e.stats[row].r2 = cnorm;
            e.stats[row].r3 = moved;
        }
        sc->al_pending = 0;
        if (e.has_rows && (moved <= e.tol || cnorm <= e.tol)) {
            sc->status = BQ_STATUS_OPTIMAL;
            sc->done = 1;
            return;   // the evaluation below belongs to an iteration that does not take place
        }
        sc->iter += 1;
    }
    if (!EVAL) return;
    const double xqx = t[0], qx = t[1];
    const double pf = 0.5 * xqx + qx;
    double dual_c = t[3], sq = t[4];
    if (e.a) {
        dual_c += sc->al_mu * ax;
        sq += ax * ax;
    }
    const double f = pf + dual_c + 0.5 * e.rho * sq;
    sc->f = f;
    sc->al_pf = pf;
    sc->al_ax = ax;
    const long long row = sc->iter - sc->stat_base;
    if (row >= 0 && row < sc->stat_cap) {
        bq_iter_stat st;
        st.iter = sc->iter;
        st.f = f;
        st.r1 = pf;
        st.r2 = 0.0;
        st.r3 = 0.0;
        e.stats[row] = st;
    }
    sc->al_epoch += 1;
    if (sc->al_epoch >= e.epochs) {
        sc->status = BQ_STATUS_STOPPED;
        sc->done = 1;
        sc->al_last = 1;   // the update kernel still owes g_x at this point (no step follows)
    }
}
