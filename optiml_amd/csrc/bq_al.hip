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
//
// An iteration is THREE launches with ONE chain of sums (round 6; six launches and two chains in rounds 2-5: prep, tiles, slab
// reduction, finish, eval + decisions, update + decisions):
//   the tile kernel;
//   the kernel that closes the product (bq_epilogue.h, kind 2): Qx, the sums of this evaluation AND of the previous iteration's stop
//   test per 256-row block; its last block closes the previous iteration (multiplier of the equality row, stop tests, iter += 1)
//   and takes this one's value, record and epoch test;
//   al_update_kernel: one elementwise pass — gradient, rule step, momentum, x, the multipliers of the bound rows, the per-element
//   terms of the stop test (summed by the NEXT closing kernel) and the next product's input.
// al_flush_kernel closes the last iteration of a bq_solver_run (same sums, same tree: the bits do not depend on how the iterations
// are cut into runs).  BQ_SVR structure keeps a launch of its own for the product's input; Nesterov momentum moves x before the
// gradient is taken, so its iterations are closed at once (flush after every update: five launches).
#include <cmath>

#include "bq_al.h"
#include "bq_epilogue.h"

#define VEC_LOOP(i)                                                             \
    const int64_t _base = (int64_t)blockIdx.x * BQ_VEC_TILE + threadIdx.x;      \
    _Pragma("unroll") for (int _j = 0; _j < BQ_VEC_ITEMS; ++_j)                 \
        for (int64_t i = _base + (int64_t)_j * BQ_VEC_BLOCK, _once = 1; _once; _once = 0)

static inline dim3 vgrid(int64_t ldN) { return dim3((unsigned)(ldN / BQ_VEC_TILE)); }

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

// The stand-alone closing kernel of an iteration's evaluation (bq_epilogue.h, kind 2) for the paths whose product has no closing
// kernel of its own to carry it (dense row-block panels, the one-rank streamed product, BQ_SYM_EXCHANGE=allreduce): one workgroup
// per 256 rows, the sums have the same bits whichever kernel closed the product.
__global__ __launch_bounds__(256) void finish_al_kernel(const double *__restrict__ sv, bq_epilogue epi) {
    if (epi.sc->done) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bq_al_pre pre = bq_al_epi_preload(epi, i, true, true);
    bq_al_epi_finish<true>(epi, blockIdx.x, gridDim.x, bq_al_epi_element(epi, pre, i, i < epi.n ? sv[i] : 0.0, true), gridDim.x);
}

// closes the iteration whose update has run but whose stop test is still pending (the end of a bq_solver_run; every iteration
// with Nesterov momentum): the sums of bq_al_epi_finish without an evaluation
__global__ __launch_bounds__(256) void al_flush_kernel(bq_epilogue epi) {
    if (epi.sc->done || !epi.sc->al_pending) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bq_al_pre pre = bq_al_epi_preload(epi, i, true, false);
    bq_al_epi_finish<false>(epi, blockIdx.x, gridDim.x, bq_al_epi_element(epi, pre, i, 0.0, false), gridDim.x);
}

// gradient, rule step, momentum, x update, multiplier update of the coordinate rows, and the per-element terms of the stop test at
// the new point (the next closing kernel, or al_flush_kernel, sums them, updates the equality multiplier and decides): one
// elementwise pass, an element per thread
// w_out (BQ_SVC without Nesterov momentum): the next product's input y o x_new, so that no prep launch precedes the tile kernel
__global__ __launch_bounds__(256) void al_update_kernel(int64_t N, int64_t ldN, bq_al_vecs V, bq_al_params prm, bq_scal *sc,
                                                        const double *__restrict__ sgn, double *__restrict__ w_out) {
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
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
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
        if (!last) {
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
            if (w_out != nullptr) w_out[i] = sgn[i] * xn;
            if (prm.rule == BQ_RULE_ADADELTA)   // adadelta.py:122 (only reached when the stop test fails: a stop ends the solve anyway)
                V.s2[i] = __dmul_rn(prm.decay, V.s2[i]) + __dmul_rn(1.0 - prm.decay, __dmul_rn(step, step));
            // ---- constraints at the new point, multiplier update of the coordinate rows --------------------------
            double cn = 0.0, dl = 0.0;
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
            V.chk[i] = cn;
            V.chk[ldN + i] = dl;
            V.chk[2 * ldN + i] = (xn - x) * (xn - x);
        }
    }
    if (last) {   // uniform: no step after the last evaluation; the flag is consumed by the block that finishes last
        if (al_last_block(&sc->ticket[1]) && threadIdx.x == 0) {
            sc->al_last = 0;
            sc->ticket[1] = 0;
        }
        return;
    }
    if (i == 0) sc->al_pending = 1;   // read by the NEXT kernel (the closing kernel of the next product, or al_flush_kernel)
}

static bq_epilogue al_epilogue(bq_solver *s) {
    bq_al_state *al = s->al;
    bq_problem *p = s->p;
    bq_epilogue epi = {};
    epi.structure = p->structure;
    epi.kind = 2;
    epi.n = p->n;
    epi.N = p->N;
    epi.diag_add = p->diag_add;
    epi.x = al->V.x;
    epi.q = al->V.q;
    epi.lb = al->V.lb;
    epi.ub = al->V.ub;
    epi.sgn = p->sgn;
    epi.Qd = al->V.Qx;
    epi.sc = s->sc;
    epi.part = s->partials;
    epi.stats = s->stats;
    epi.a = al->V.a;
    epi.llb = al->V.llb;
    epi.lub = al->V.lub;
    epi.chk = al->V.chk;
    epi.ldN = s->ldN;
    epi.rho = al->prm.rho;
    epi.tol = al->prm.tol;
    epi.epochs = al->prm.epochs;
    epi.has_rows = (al->V.a != nullptr || al->V.lb != nullptr || al->V.ub != nullptr) ? 1 : 0;
    return epi;
}

// the end of a bq_solver_run: the last update's stop test must not stay pending across the call (its record leaves with this run)
int bq_al_flush(bq_solver *s) {
    al_flush_kernel<<<(unsigned)((s->p->n + 255) / 256), 256, 0, s->p->ctx->stream>>>(al_epilogue(s));
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

int bq_al_iterate(bq_solver *s) {
    bq_al_state *al = s->al;
    bq_problem *p = s->p;
    hipStream_t st = p->ctx->stream;
    const int *done = &s->sc->done;
    const bq_al_params &prm = al->prm;
    const bool nesterov = prm.momentum_type == BQ_MOM_NESTEROV;
    if (nesterov) al_jump_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, al->V, prm.momentum, s->sc);
    // the product's input: x itself (BQ_PLAIN), what the previous update kernel left in p->w (BQ_SVC, nothing has moved x since),
    // else the structure map of x by a launch of its own
    const double *w = al->V.x;
    if (p->structure != BQ_PLAIN) {
        if (!al->w_ready) BQ_TRY(bq_launch_prep(p, al->V.x, done));
        w = p->w;
    }
    const bq_epilogue epi = al_epilogue(s);
    bool fused = false;
    BQ_TRY(bq_panel_product(p, p->add_one, w, done, &epi, &fused));
    if (!fused) finish_al_kernel<<<(unsigned)((p->n + 255) / 256), 256, 0, st>>>(p->s, epi);
    double *w_out = (p->structure == BQ_SVC && !nesterov) ? p->w : nullptr;
    al_update_kernel<<<(unsigned)((s->N + 255) / 256), 256, 0, st>>>(s->N, s->ldN, al->V, prm, s->sc, p->sgn, w_out);
    al->w_ready = w_out != nullptr;
    // Nesterov's jump of the NEXT iteration moves x before its gradient is taken: whether that iteration takes place must be known
    // before the jump, so the iteration is closed here instead of inside the next closing kernel
    if (nesterov) al_flush_kernel<<<(unsigned)((p->n + 255) / 256), 256, 0, st>>>(epi);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}
