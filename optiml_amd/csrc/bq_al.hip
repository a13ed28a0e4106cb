// Augmented-Lagrangian dual driven by the first-order ("stochastic", full batch) update rules — SURVEY 8(f).3.
//
// Restates on the device, for a QP  min 1/2 x'Qx + q'x  s.t.  a'x = 0 (optional row), lb <= x <= ub (each optional):
//   constraints  c = [a'x; lb - x; x - ub]                         optiml/opti/constrained/_base.py:239-296, 317-325
//   value        L = f(x) + dual'c + rho/2 |[c_eq; max(c_in, 0)]|^2           :326-340
//   gradient     Qx + q + dual'AG + rho AG_act'(AG_act x - bh_act)            :374-404
//   callback     primal value, past_x                                         optiml/opti/_base.py:96-117
//   multipliers  dual += rho c(x_new), inequality part clipped at 0; 'optimal' when |d dual| + |d x| <= tol or
//                |c| <= tol                                                   optiml/opti/_base.py:129-146
//   loop + rules optiml/opti/unconstrained/stochastic/{gradient_descent,adam,amsgrad,adamax,adagrad,adadelta,
//                rmsprop}.py (minimize): [nesterov jump] -> evaluate -> callback -> epoch test -> rule step ->
//                momentum -> multiplier update / stop test -> iter += 1.
// [A; -I; I] is never formed: every row is a coordinate or the single dense row a.  One panel product Q x per
// iteration gives the value, the primal value and the gradient (the reference spends three).  All O(n) work and all
// reductions are redundant per rank in a fixed order; only the panel product is sharded.
#include <cmath>

#include "bq_al.h"

#define VEC_LOOP(i)                                                             \
    const int64_t _base = (int64_t)blockIdx.x * BQ_VEC_TILE + threadIdx.x;      \
    _Pragma("unroll") for (int _j = 0; _j < BQ_VEC_ITEMS; ++_j)                 \
        for (int64_t i = _base + (int64_t)_j * BQ_VEC_BLOCK, _once = 1; _once; _once = 0)

static inline dim3 vgrid(int64_t ldN) { return dim3((unsigned)(ldN / BQ_VEC_TILE)); }

__device__ __forceinline__ double al_wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// NQ block sums in ONE round of barriers, and the NQ final sums over a launch's per-block partials (quantity q at part + q * nblk)
// likewise: each quantity through the tree the one-quantity helpers of rounds 2-4 used (wave tree, four waves in order: same bits); NQ separate calls were 2 NQ barriers in kernels
// whose length is their chain of dependent steps (round 5; config-2-sized AdaGrad: 42 -> 38 us beside the product)
template <int NQ>
__device__ __forceinline__ void al_bsum_n(double (&v)[NQ], double (*sh)[4]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = al_wsum(v[q]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) sh[q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = ((sh[q][0] + sh[q][1]) + sh[q][2]) + sh[q][3];
    __syncthreads();
}
template <int NQ>
__device__ __forceinline__ void al_fsum_n(const double *part, int64_t nblk, double (&v)[NQ], double (*sh)[4]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = 0.0;
    for (int64_t i = threadIdx.x; i < nblk; i += BQ_VEC_BLOCK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] += part[q * nblk + i];
    }
    al_bsum_n<NQ>(v, sh);
}

// the block that takes the last ticket of a launch finishes the reduction and takes the scalar decisions in the same
// kernel (same fixed-order final sums as a separate one-block kernel: results do not depend on which block is last)
__device__ __forceinline__ bool al_last_block(unsigned int *ticket) {
    __shared__ int last;
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last) __threadfence();
    return last != 0;
}

__device__ __forceinline__ void al_record_body(bq_scal *sc, const bq_al_params &prm, int has_eq, const double *part,
                                               int64_t nblk, bq_iter_stat *stats);
__device__ __forceinline__ void al_check_body(bq_scal *sc, const bq_al_params &prm, int has_eq, int has_rows,
                                              const double *part, int64_t nblk, bq_iter_stat *stats);

// nesterov: x += momentum * previous step, before the gradient is taken (gradient_descent.py:78-81 and the like)
// value of a schedule at the current iteration (the last entry continues), or the constant
__device__ __forceinline__ double al_sched(const double *sched, long long len, long long it, double constant) {
    return sched != nullptr ? sched[it < len ? it : len - 1] : constant;
}

__global__ void al_jump_kernel(int64_t N, bq_al_vecs V, double mom_const, const bq_scal *sc) {
    if (sc->done) return;
    const double mom = al_sched(V.mom_sched, V.sched_len, sc->iter, mom_const);
    VEC_LOOP(i) {
        if (i < N) V.x[i] = V.x[i] + __dmul_rn(mom, V.step[i]);
    }
}

// partial sums of everything the value needs at x (Qx already computed)
__global__ void al_eval_kernel(int64_t N, bq_al_vecs V, bq_scal *sc, double *part, int64_t nblk, bq_al_params prm,
                               int has_eq, bq_iter_stat *stats) {
    if (sc->done) return;
    double xqx = 0.0, qx = 0.0, ax = 0.0, dc = 0.0, cl = 0.0;
    VEC_LOOP(i) {
        if (i < N) {
            const double x = V.x[i];
            xqx += x * V.Qx[i];
            qx += V.q[i] * x;
            if (V.a) ax += V.a[i] * x;
            if (V.lb) {
                const double c = V.lb[i] - x;
                dc += V.llb[i] * c;
                if (c > 0.0) cl += c * c;
            }
            if (V.ub) {
                const double c = x - V.ub[i];
                dc += V.lub[i] * c;
                if (c > 0.0) cl += c * c;
            }
        }
    }
    __shared__ double sh5[5][4];
    double v5[5] = {xqx, qx, ax, dc, cl};
    al_bsum_n<5>(v5, sh5);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) part[q * nblk + blockIdx.x] = v5[q];
    }
    if (al_last_block(&sc->ticket[0])) {
        al_record_body(sc, prm, has_eq, part, nblk, stats);
        if (threadIdx.x == 0) sc->ticket[0] = 0;
    }
}

// last block of al_eval_kernel: value, primal value, iteration record, epoch test     (adagrad.py:85-101 and the like)
__device__ __forceinline__ void al_record_body(bq_scal *sc, const bq_al_params &prm, int has_eq, const double *part,
                                               int64_t nblk, bq_iter_stat *stats) {
    __shared__ double sh5[5][4];
    double v5[5];
    al_fsum_n<5>(part, nblk, v5, sh5);
    const double xqx = v5[0], qx = v5[1], ax = v5[2], dc = v5[3], cl = v5[4];
    if (threadIdx.x == 0) {
        const double pf = 0.5 * xqx + qx;
        double dual_c = dc, sq = cl;
        if (has_eq) {
            dual_c += sc->al_mu * ax;
            sq += ax * ax;
        }
        const double f = pf + dual_c + 0.5 * prm.rho * sq;
        sc->f = f;
        sc->al_pf = pf;
        sc->al_ax = ax;
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            stats[row].iter = sc->iter;
            stats[row].f = f;
            stats[row].r1 = pf;
            stats[row].r2 = 0.0;
            stats[row].r3 = 0.0;
        }
        sc->al_epoch += 1;
        if (sc->al_epoch >= prm.epochs) {
            sc->status = BQ_STATUS_STOPPED;
            sc->done = 1;
            sc->al_last = 1;   // the update kernel still owes g_x at this point (no step follows)
        }
    }
}

// gradient, rule step, momentum, x update, multiplier update of the coordinate rows, and the partial sums the stop
// test needs at the new point; the last block then updates the equality multiplier and runs the stop tests
__global__ void al_update_kernel(int64_t N, bq_al_vecs V, bq_al_params prm, bq_scal *sc, double *part, int64_t nblk,
                                 int has_eq, int has_rows, bq_iter_stat *stats) {
    const bool last = sc->al_last != 0;   // 'stopped' at this evaluation: write its gradient, take no step
    if (sc->done && !last) return;
    const double ax = sc->al_ax, mu = sc->al_mu, rho = prm.rho;
    const double lr = al_sched(V.lr_sched, V.sched_len, sc->iter, prm.step_size);
    const double mom = al_sched(V.mom_sched, V.sched_len, sc->iter, prm.momentum);
    const bool eq_act = V.a != nullptr && ax != 0.0;
    const double t = (double)(sc->iter + 1);
    double c1 = 1.0, c2 = 1.0;   // bias corrections 1 - beta^t
    if (prm.rule == BQ_RULE_ADAM || prm.rule == BQ_RULE_ADAMAX) c1 = 1.0 - pow(prm.beta1, t);
    if (prm.rule == BQ_RULE_ADAM) c2 = 1.0 - pow(prm.beta2, t);
    double axn = 0.0, cn = 0.0, dl = 0.0, dx = 0.0;
    VEC_LOOP(i) {
        if (i < N) {
            const double x = V.x[i];
            // ---- gradient at x -------------------------------------------------------------------------------
            double g = V.Qx[i] + V.q[i];
            double dual_ag = 0.0, t3 = 0.0, t4 = 0.0;
            if (V.a) {
                dual_ag = __dmul_rn(mu, V.a[i]);
                if (eq_act) t3 = __dmul_rn(V.a[i], ax);
            }
            if (V.lb) {
                dual_ag -= V.llb[i];
                if (V.lb[i] - x > 0.0) {
                    t3 += x;
                    t4 += V.lb[i];
                }
            }
            if (V.ub) {
                dual_ag += V.lub[i];
                if (x - V.ub[i] > 0.0) {
                    t3 += x;
                    t4 += V.ub[i];
                }
            }
            g = ((g + dual_ag) + __dmul_rn(rho, t3)) - __dmul_rn(rho, t4);
            V.g[i] = g;
            V.xe[i] = x;
            if (last) continue;
            // ---- rule step -----------------------------------------------------------------------------------
            const double d = -g, g2 = __dmul_rn(g, g);
            double s;
            switch (prm.rule) {
                case BQ_RULE_ADAM: {
                    const double m = __dmul_rn(prm.beta1, V.s1[i]) + __dmul_rn(1.0 - prm.beta1, d);
                    const double v = __dmul_rn(prm.beta2, V.s2[i]) + __dmul_rn(1.0 - prm.beta2, g2);
                    V.s1[i] = m;
                    V.s2[i] = v;
                    s = __dmul_rn(lr, m / c1) / (sqrt(v / c2) + prm.offset);
                    break;
                }
                case BQ_RULE_AMSGRAD: {
                    const double m = __dmul_rn(prm.beta1, V.s1[i]) + __dmul_rn(1.0 - prm.beta1, d);
                    const double v = __dmul_rn(prm.beta2, V.s2[i]) + __dmul_rn(1.0 - prm.beta2, g2);
                    const double vm = fmax(v, V.s3[i]);
                    V.s1[i] = m;
                    V.s2[i] = v;
                    V.s3[i] = vm;
                    s = __dmul_rn(lr, m) / (sqrt(vm) + prm.offset);
                    break;
                }
                case BQ_RULE_ADAMAX: {
                    const double m = __dmul_rn(prm.beta1, V.s1[i]) + __dmul_rn(1.0 - prm.beta1, d);
                    const double u = fmax(__dmul_rn(prm.beta2, V.s2[i]), fabs(g));
                    V.s1[i] = m;
                    V.s2[i] = u;
                    s = __dmul_rn(lr, m / c1) / (u + prm.offset);
                    break;
                }
                case BQ_RULE_ADAGRAD: {
                    const double acc = V.s1[i] + g2;
                    V.s1[i] = acc;
                    s = __dmul_rn(lr, d) / sqrt(acc + prm.offset);
                    break;
                }
                case BQ_RULE_ADADELTA: {
                    const double acc = __dmul_rn(prm.decay, V.s1[i]) + __dmul_rn(1.0 - prm.decay, g2);
                    V.s1[i] = acc;
                    s = __dmul_rn(__dmul_rn(lr, d), sqrt(V.s2[i] + prm.offset) / sqrt(acc + prm.offset));
                    break;
                }
                case BQ_RULE_RMSPROP: {
                    const double acc = __dmul_rn(prm.decay, V.s1[i]) + __dmul_rn(1.0 - prm.decay, g2);
                    V.s1[i] = acc;
                    s = __dmul_rn(lr, d) / sqrt(acc + prm.offset);
                    break;
                }
                default: s = __dmul_rn(lr, d); break;   // BQ_RULE_SGD
            }
            // ---- momentum: step = momentum * previous step + s in both variants; nesterov already moved x by the
            // first term before the gradient was taken ------------------------------------------------------------
            double step = s, xn;
            if (prm.momentum_type == BQ_MOM_POLYAK) {
                step = __dmul_rn(mom, V.step[i]) + s;
                xn = x + step;
            } else if (prm.momentum_type == BQ_MOM_NESTEROV) {
                step = __dmul_rn(mom, V.step[i]) + s;
                xn = x + s;
            } else {
                xn = x + step;
            }
            V.step[i] = step;
            V.x[i] = xn;
            if (prm.rule == BQ_RULE_ADADELTA)   // adadelta.py:122 (only reached when the stop test below fails)
                V.s2[i] = __dmul_rn(prm.decay, V.s2[i]) + __dmul_rn(1.0 - prm.decay, __dmul_rn(step, step));
            // ---- constraints at the new point, multiplier update of the coordinate rows --------------------------
            if (V.a) axn += V.a[i] * xn;
            if (V.lb) {
                const double c = V.lb[i] - xn, old = V.llb[i];
                const double nw = fmax(old + __dmul_rn(rho, c), 0.0);
                V.llb[i] = nw;
                cn += c * c;
                dl += (nw - old) * (nw - old);
            }
            if (V.ub) {
                const double c = xn - V.ub[i], old = V.lub[i];
                const double nw = fmax(old + __dmul_rn(rho, c), 0.0);
                V.lub[i] = nw;
                cn += c * c;
                dl += (nw - old) * (nw - old);
            }
            dx += (xn - x) * (xn - x);
        }
    }
    if (last) {   // uniform: no step after the last evaluation; the flag is consumed by the block that finishes last
        if (al_last_block(&sc->ticket[1]) && threadIdx.x == 0) {
            sc->al_last = 0;
            sc->ticket[1] = 0;
        }
        return;
    }
    __shared__ double sh4[4][4];
    double v4[4] = {axn, cn, dl, dx};
    al_bsum_n<4>(v4, sh4);
    axn = v4[0];
    cn = v4[1];
    dl = v4[2];
    dx = v4[3];
    if (threadIdx.x == 0) {
        part[0 * nblk + blockIdx.x] = axn;
        part[1 * nblk + blockIdx.x] = cn;
        part[2 * nblk + blockIdx.x] = dl;
        part[3 * nblk + blockIdx.x] = dx;
    }
    if (al_last_block(&sc->ticket[1])) {
        al_check_body(sc, prm, has_eq, has_rows, part, nblk, stats);
        if (threadIdx.x == 0) sc->ticket[1] = 0;
    }
}

// last block of al_update_kernel: multiplier of the equality row, the two stop tests, iter += 1
// (optiml/opti/_base.py:129-146)
__device__ __forceinline__ void al_check_body(bq_scal *sc, const bq_al_params &prm, int has_eq, int has_rows,
                                              const double *part, int64_t nblk, bq_iter_stat *stats) {
    __shared__ double sh4[4][4];
    double v4[4];
    al_fsum_n<4>(part, nblk, v4, sh4);
    const double axn = v4[0], dx = v4[3];
    double cn = v4[1], dl = v4[2];
    if (threadIdx.x == 0) {
        if (has_eq) {
            const double dmu = prm.rho * axn;
            sc->al_mu = sc->al_mu + dmu;
            cn += axn * axn;
            dl += dmu * dmu;
        }
        const double cnorm = sqrt(cn), moved = sqrt(dl) + sqrt(dx);
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            stats[row].r2 = cnorm;
            stats[row].r3 = moved;
        }
        if (has_rows && (moved <= prm.tol || cnorm <= prm.tol)) {
            sc->status = BQ_STATUS_OPTIMAL;
            sc->done = 1;
        } else {
            sc->iter += 1;
        }
    }
}

int bq_al_iterate(bq_solver *s) {
    bq_al_state *al = s->al;
    bq_problem *p = s->p;
    hipStream_t st = p->ctx->stream;
    const int *done = &s->sc->done;
    const bq_al_params &prm = al->prm;
    const int has_eq = al->V.a != nullptr, has_rows = has_eq || al->V.lb != nullptr || al->V.ub != nullptr;
    if (prm.momentum_type == BQ_MOM_NESTEROV)
        al_jump_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, al->V, prm.momentum, s->sc);
    BQ_TRY(bq_problem_apply(p, al->V.x, al->V.Qx, done));
    al_eval_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, al->V, s->sc, s->partials, s->nblk, prm, has_eq, s->stats);
    al_update_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, al->V, prm, s->sc, s->partials, s->nblk, has_eq,
                                                             has_rows, s->stats);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
