// BQ_AS_CG: the restricted systems of ActiveSet solved by conjugate gradients on the masked panel operator (header comment of
// bq_as.hip).  Split out of bq_as.hip in round 5.
#include "bq_as.h"

// xt = the current point with the bound values on L and U (what the reference substitutes, active_set.py:132-136)
__global__ void as_make_xt_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                  const double *__restrict__ lb, const double *__restrict__ ub,
                                  const double *__restrict__ x, double *__restrict__ xt) {
    VEC_LOOP(i) {
        if (i < N) xt[i] = mU[i] ? ub[i] : (mL[i] ? lb[i] : x[i]);
    }
}

// r = p = -(Q xt + q) on A, 0 elsewhere; delta = 0; the stop level is rtol * (|(Q xt)_A| + |q_A|)
// (Qlevel: the product the stop level is taken from — Q x of the CURRENT point when the iteration starts somewhere else, so
// that a warm start does not change what "solved to rtol" means)
__global__ void as_cg_init_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                  const double *__restrict__ Qxt, const double *__restrict__ Qlevel,
                                  const double *__restrict__ q, double *__restrict__ dlt,
                                  double *__restrict__ r, double *__restrict__ pv, double *part, int64_t nblk,
                                  as_cg_scal *cg, double rtol, long long max_iters, int pc) {
    __shared__ double sh[4];
    double srr = 0.0, sqx = 0.0, sq = 0.0;
    VEC_LOOP(i) {
        if (i < N) {
            const bool fr = !(mL[i] | mU[i]);
            const double a = Qxt[i], b = q[i];
            const double ri = fr ? -(a + b) : 0.0;
            r[i] = ri;
            pv[i] = ri;
            dlt[i] = 0.0;
            srr += __dmul_rn(ri, ri);
            if (fr) {
                const double al = Qlevel[i];
                sqx += __dmul_rn(al, al);
                sq += __dmul_rn(b, b);
            }
        }
    }
    srr = as_block_sum(srr, sh);
    sqx = as_block_sum(sqx, sh);
    sq = as_block_sum(sq, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = srr;
        part[nblk + blockIdx.x] = sqx;
        part[2 * nblk + blockIdx.x] = sq;
    }
    if (as_last_block(&cg->ticket[0])) {
        const double rr = as_final_sum(part, nblk, sh);
        const double nqx = as_final_sum(part + nblk, nblk, sh), nq = as_final_sum(part + 2 * nblk, nblk, sh);
        if (threadIdx.x == 0) {
            const double level = rtol * (sqrt(nqx) + sqrt(nq));
            cg->ticket[0] = 0;
            cg->rr = rr;
            cg->tol2 = level * level;
            cg->alpha = 0.0;
            cg->beta = 0.0;
            cg->iters = 0;
            cg->max_iters = max_iters;
            cg->info = 0;
            cg->pc = pc;
            cg->rz = rr;
            cg->done = (rr <= cg->tol2) ? 1 : 0;
        }
    }
}

// alpha = r'r / p'Qp (p vanishes outside A, so the sum needs no mask); a curvature <= 0 means Q[A,A] is not positive definite
__global__ void as_cg_pap_kernel(int64_t N, const double *__restrict__ pv, const double *__restrict__ Qp, double *part,
                                 int64_t nblk, as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double sh[4];
    double s = 0.0;
    VEC_LOOP(i) {
        if (i < N) s += __dmul_rn(pv[i], Qp[i]);
    }
    s = as_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (as_last_block(&cg->ticket[0])) {
        const double pAp = as_final_sum(part, nblk, sh);
        if (threadIdx.x == 0) {
            cg->ticket[0] = 0;
            if (!(pAp > 0.0) || !isfinite(pAp)) {
                cg->info = 1;
                cg->alpha = 0.0;
            } else {
                cg->alpha = (cg->pc ? cg->rz : cg->rr) / pAp;
            }
        }
    }
}

// delta += alpha p;  r -= alpha m.(Qp);  beta = r'r(new) / r'r(old); stop tests.  Qdl += alpha Qp on ALL rows: Q delta, which
// with the start point's product gives Q cand without another product (as_qcand_kernel)
__global__ void as_cg_update_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                    double *__restrict__ dlt, double *__restrict__ r, const double *__restrict__ pv,
                                    const double *__restrict__ Qp, double *__restrict__ Qdl, double *part, int64_t nblk,
                                    as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double sh[4];
    const double alpha = cg->alpha;
    double s = 0.0;
    VEC_LOOP(i) {
        if (i < N) Qdl[i] = Qdl[i] + __dmul_rn(alpha, Qp[i]);
        if (i < N && !(mL[i] | mU[i])) {
            dlt[i] = dlt[i] + __dmul_rn(alpha, pv[i]);
            const double ri = r[i] - __dmul_rn(alpha, Qp[i]);
            r[i] = ri;
            s += __dmul_rn(ri, ri);
        }
    }
    s = as_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (as_last_block(&cg->ticket[1])) {
        const double rr = as_final_sum(part, nblk, sh);
        if (threadIdx.x == 0) {
            cg->ticket[1] = 0;
            if (!cg->pc) cg->beta = cg->rr > 0.0 ? rr / cg->rr : 0.0;   // preconditioned: beta = rz_new / rz (as_pc_apply_kernel)
            cg->rr = rr;
            cg->iters += 1;
            if (cg->info || rr <= cg->tol2 || cg->iters >= cg->max_iters || !isfinite(rr)) cg->done = 1;
        }
    }
}

// p = r + beta p
__global__ void as_cg_dir_kernel(int64_t N, const double *__restrict__ r, double *__restrict__ pv, const as_cg_scal *cg) {
    if (cg->done) return;
    const double beta = cg->beta;
    VEC_LOOP(i) {
        if (i < N) pv[i] = r[i] + __dmul_rn(beta, pv[i]);
    }
}

// Q cand = Q z + Q delta (z: the start point with the bound values, delta: what the iteration added on the free set)
__global__ void as_qcand_kernel(int64_t N, const double *__restrict__ Qz, const double *__restrict__ Qdl, double *__restrict__ Qc) {
    VEC_LOOP(i) {
        Qc[i] = i < N ? Qz[i] + Qdl[i] : 0.0;
    }
}

// after the ratio step x += t (cand - x):  Q x += t (Q cand - Q x)
__global__ void as_qx_lerp_kernel(int64_t N, const bq_scal *sc, const double *__restrict__ Qc, double *__restrict__ Qx) {
    const double t = sc->step;
    VEC_LOOP(i) {
        if (i < N) Qx[i] = Qx[i] + __dmul_rn(t, Qc[i] - Qx[i]);
    }
}

constexpr int AS_MAX_COLS = 16;
// Where the start vector z (the previous candidate with the CURRENT bound values) differs from that candidate: the variables
// that reached a bound in the step since — a handful.
// zchg[1] = 1: at most AS_MAX_COLS of them, so Q z = Q cand + sum_j (bound_j - cand_j) Q[:, j] and the product is skipped.
__device__ __forceinline__ double as_zdiff_of(int64_t i, int64_t N, const unsigned char *mL, const unsigned char *mU,
                                              const double *lb, const double *ub, const double *cand) {
    if (i < N && (mL[i] | mU[i])) return (mU[i] ? ub[i] : lb[i]) - cand[i];
    return 0.0;
}
__global__ __launch_bounds__(256) void as_zdiff_count_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                             const unsigned char *__restrict__ mU, const double *__restrict__ lb,
                                                             const double *__restrict__ ub, const double *__restrict__ cand,
                                                             int *__restrict__ cnt, unsigned int *ticket, int *__restrict__ zchg) {
    int bits = 0;
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        if (as_zdiff_of(i, N, mL, mU, lb, ub, cand) != 0.0) bits |= 1 << j;
    }
    __shared__ int total;
    if (threadIdx.x == 0) total = -1;
    as_list_count(bits, cnt, ticket, &total);
    if (threadIdx.x == 0 && total >= 0) {
        zchg[0] = total;
        zchg[1] = total <= AS_MAX_COLS ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void as_zdiff_write_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                             const unsigned char *__restrict__ mU, const double *__restrict__ lb,
                                                             const double *__restrict__ ub, const double *__restrict__ cand,
                                                             const int *__restrict__ cnt, int *__restrict__ zchg,
                                                             double *__restrict__ zdl) {
    int bits = 0;
    double dl[BQ_VEC_ITEMS];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        dl[j] = as_zdiff_of(i, N, mL, mU, lb, ub, cand);
        if (dl[j] != 0.0) bits |= 1 << j;
    }
    int pos[BQ_VEC_ITEMS];
    as_list_positions(bits, cnt, pos);
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j)
        if (((bits >> j) & 1) && pos[j] < AS_MAX_COLS) {
            zchg[2 + pos[j]] = (int)((int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x);
            zdl[pos[j]] = dl[j];
        }
}

// Qz = Q cand + sum_c dl_c Q[:, j_c] with the columns formed from X (replicated on every rank, so no exchange): the entry the
// panel holds up to the rounding of its own dot products — K as the kernel maps of bq_gram.hip define it, rounded to the
// panel's storage type, then the structure of the dual (bq_qelem.h).  Runs only when as_zdiff_kernel said the columns suffice.
__global__ __launch_bounds__(256) void as_qz_cols_kernel(int64_t n, int64_t d, const double *__restrict__ X, const double *__restrict__ sq,
                                                         const double *__restrict__ sgn, int kernel, double gamma, double coef0,
                                                         int degree, int add_one, double diag_add, int f32,
                                                         const int *__restrict__ zchg, const double *__restrict__ zdl,
                                                         const double *__restrict__ Qc, double *__restrict__ Qz) {
    if (!zchg[1]) return;
    // a WAVE per row, lanes along the features: a row of X is read in 512-byte runs (round 3 gave every lane a row of its own: 64
    // cache lines per load instruction, 0.53 ms for the 0.5 GB of X at BASELINE config 5); the dot product is the lane-strided sum
    // + a halving butterfly, the same on every rank
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int cnt = zchg[0];
    for (int64_t i = wave; i < n; i += nwaves) {
        const double *xi = X + i * d;
        double acc = Qc[i];
        for (int c = 0; c < cnt; ++c) {
            const int64_t j = zchg[2 + c];
            const double *xj = X + j * d;
            double dot = 0.0;
            for (int64_t k = lane; k < d; k += 64) dot = fma(xi[k], xj[k], dot);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
            double kv;
            if (kernel == BQ_KERNEL_RBF) {
                double dist = -2.0 * dot;
                dist += sq[i];
                dist += sq[j];
                dist = fmax(dist, 0.0);
                if (i == j) dist = 0.0;
                kv = bq_exp(-gamma * dist);
            } else if (kernel == BQ_KERNEL_POLY) {
                const double b = gamma * dot + coef0;
                kv = degree == 2 ? b * b : (degree == 3 ? b * b * b : pow(b, (double)degree));
            } else if (kernel == BQ_KERNEL_SIGMOID) {
                kv = tanh(gamma * dot + coef0);
            } else {
                kv = dot;
            }
            if (f32) kv = (double)(float)kv;
            double q = kv + (add_one ? 1.0 : 0.0);
            if (sgn) q *= sgn[i] * sgn[j];
            if (i == j) q += diag_add;
            acc = fma(zdl[c], q, acc);
        }
        if (lane == 0) Qz[i] = acc;
    }
}

__global__ void as_row_norms_kernel(const double *__restrict__ X, int64_t n, int64_t d, double *__restrict__ out, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    double v = 0.0;
    if (i < n) {
        const double *row = X + i * d;
        for (int64_t k = 0; k < d; ++k) v = fma(row[k], row[k], v);
    }
    out[i] = v;
}

// sol[a] = x[idx[a]] + delta[idx[a]] (x: the point the iteration started from): the restricted solution in the compact order
// as_candidate_kernel reads
__global__ void as_cg_gather_kernel(const int *__restrict__ ints, const int *__restrict__ idx,
                                    const double *__restrict__ x, const double *__restrict__ dlt,
                                    double *__restrict__ sol, int64_t N) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < ints[0] && a < N) sol[a] = x[idx[a]] + dlt[idx[a]];
}

// the restricted solve of one outer iteration by conjugate gradients; leaves the candidate in w->cand and the
// feasibility flag in w->host_ints[2].
//   start: the candidate of the previous outer iteration (the free set has moved by one index since, so it solves the new
//   system up to one column of Q), else the current point;  hook as_cg_warm=0: always the current point
//   preconditioner: struct as_pc (RBF and linear panels);   BQ_AS_CG_PC=0: none
static int as_cg_solve_once(bq_solver *s, as_ws *w, int *pc_failed) {
    *pc_failed = 0;
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    const int64_t N = s->N, nblk = s->nblk;
    const int64_t nA = w->host_ints[0];
    const dim3 grid = vgrid(s->ldN);
    const long long cap = s->inner_max > 0 ? s->inner_max : 2 * (long long)nA + 50;
    as_pc *pc = w->pc;
    const double *start = (w->warm && w->have_cand) ? w->cand : s->x;
    as_make_xt_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, start, w->z);
    const bool anchor = w->anchor;
    w->anchor = false;
    if (w->colq && start == w->cand && !anchor) {
        // z is the previous candidate except at the variables that reached a bound since: Q z = Q cand + those columns of Q,
        // formed from X — the product below returns at once (its `done` flag) unless too many variables moved
        bq_problem *p = s->p;
        int *lcnt = reinterpret_cast<int *>(s->partials + s->nblk);   // the slice as_launch_compact uses between its two passes
        as_zdiff_count_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, lcnt, &s->sc->pad1[0], w->zchg);
        as_zdiff_write_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, lcnt, w->zchg, w->zdl);
        BQ_TRY(bq_problem_apply(p, w->z, w->Qz, w->zchg + 1));
        as_qz_cols_kernel<<<(unsigned)std::min<int64_t>((p->n + 3) / 4, 8192), 256, 0, st>>>(p->n, p->d, p->X, w->sq, p->sgn, p->kernel, p->gamma, p->coef0,
                                                                         p->degree, p->add_one ? 1 : 0, p->diag_add,
                                                                         p->storage == BQ_F32 ? 1 : 0, w->zchg, w->zdl, w->Qcand, w->Qz);
    } else {
        BQ_TRY(bq_problem_apply(s->p, w->z, w->Qz, nullptr));
    }
    BQ_HIP(hipMemsetAsync(w->Qdl, 0, sizeof(double) * s->ldN, st));
    // s->Qd = Q x of the current point (eval_f at the end of the previous outer iteration, or of bq_as_start)
    as_cg_init_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, w->Qz, start == s->x ? w->Qz : s->Qd, s->p->q, w->dlt, w->r,
                                                     w->pv, s->partials, nblk, w->cg, s->inner_rtol, cap, pc ? 1 : 0);
    const double *zr = w->r;   // what the next direction is built from: the residual, or the preconditioned residual
    if (pc) {
        BQ_TRY(as_pc_update(s, w));
        BQ_TRY(as_pc_apply(s, w, 1));
        as_cg_dir_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, pc->z, w->pv, w->cg);   // beta = 0: p = z
        zr = pc->z;
    }
    // The host looks at the `done` flag after EVERY inner iteration, one iteration late: iteration k + 1 is enqueued, then the
    // host waits for the copy of the flag recorded behind iteration k.  The device never runs dry (an iteration is a panel
    // product: 2.5 - 20 ms at BASELINE config 5, far longer than the host's turn) and exactly ONE enqueued iteration is wasted per
    // solve — it returns at once on the flag — where round 3's batches of 8 -> 32 wasted 12 launches per solve, 0.17 ms each for the
    // empty 60 000-workgroup product grid alone (profiles/r04/c5_per_outer_iteration_kernel_ms_before.csv).  Every rank sees the
    // same flag values at the same iteration (replicated, bit-identical scalars), so all ranks enqueue the same collectives.
    long long queued = 0;
    BQ_HIP(hipMemcpyAsync(w->cg_flag_host, &w->cg->done, 2 * sizeof(int), hipMemcpyDeviceToHost, st));   // the flag the start leaves
    BQ_HIP(hipEventRecord(w->cg_event, st));
    while (queued < cap) {
        BQ_TRY(bq_problem_apply(s->p, w->pv, w->Qp, &w->cg->done));
        as_cg_pap_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, w->pv, w->Qp, s->partials, nblk, w->cg);
        as_cg_update_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, w->dlt, w->r, w->pv, w->Qp, w->Qdl, s->partials,
                                                           nblk, w->cg);
        if (pc) BQ_TRY(as_pc_apply(s, w, 0));
        as_cg_dir_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, zr, w->pv, w->cg);
        ++queued;
        BQ_TRY(bq_ctx_event_sync(ctx, w->cg_event));   // the flag as it stood BEFORE the iteration just enqueued
        if (w->cg_flag_host[0]) break;
        BQ_HIP(hipMemcpyAsync(w->cg_flag_host, &w->cg->done, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        BQ_HIP(hipEventRecord(w->cg_event, st));
    }
    BQ_HIP(hipMemcpyAsync(w->host_cg, w->cg, sizeof(as_cg_scal), hipMemcpyDeviceToHost, st));
    w->host_info[1] = w->host_info[2] = 0;
    if (pc) BQ_HIP(hipMemcpyAsync(w->host_info + 1, pc->ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    if (pc) BQ_HIP(hipMemcpyAsync(w->host_info + 2, pc->sm_fail, sizeof(int), hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    const int pc_info = w->host_info[1], sm_fail = w->host_info[2];
    const as_cg_scal h = *w->host_cg;
    w->cg_iters += h.iters;
    if (pc_info != 0 || sm_fail != 0 || h.info == 2) {   // the caller sums G afresh and tries again, then gives the preconditioner up
        bq_set_error("the preconditioner of the inner conjugate gradients is not positive definite (pivot %d, |A| = %lld): "
                     "BQ_AS_CG_PC=0 runs without it", pc_info, (long long)nA);
        *pc_failed = 1;
        return BQ_OK;
    }
    if (h.info != 0 || !std::isfinite(h.rr)) {
        bq_set_error("conjugate gradients on the restricted Hessian Q[A,A] (|A| = %lld) met a direction of non-positive "
                     "curvature after %lld iterations: the system is not positive definite",
                     (long long)nA, (long long)h.iters);
        return BQ_ERR_NOT_PD;
    }
    as_qcand_kernel<<<grid, BQ_VEC_BLOCK, 0, st>>>(N, w->Qz, w->Qdl, w->Qcand);
    as_cg_gather_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(w->ints, w->idx, w->z, w->dlt, w->sol, N);
    as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
    as_cand_scatter_idx_kernel<<<vgrid(s->ldN).x * (BQ_VEC_TILE / 256), 256, 0, st>>>(w->idx, w->ints, w->sol, s->lb, s->ub, w->cand);
    w->have_cand = true;
    BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    return BQ_OK;
}

// A Woodbury system G that is not positive definite (or r'z <= 0) says something about the MODEL P, not about Q: its Gram matrix
// has drifted under the rank-one updates, or the features do not fit this free set.  First G is summed afresh and the solve
// repeated (with the spectrum bound of the implicit remainder estimated afresh on THIS free set: an underestimate is the other way
// the operator turns indefinite); then the remainder is given up and the explicit model kept; if that fails too the preconditioner
// is dropped for the rest of the run (plain conjugate gradients).  Every rank reads the same replicated scalars, so all ranks take
// the same turn here and stay in the same collectives.
static int as_cg_solve(bq_solver *s, as_ws *w) {
    for (int attempt = 0; attempt < 4; ++attempt) {
        int pc_failed = 0;
        BQ_TRY(as_cg_solve_once(s, w, &pc_failed));
        if (!pc_failed) return BQ_OK;
        if (w->pc == nullptr) break;
        if (attempt == 0) {
            w->pc->age = 0;
            if (w->pc->r2) as_pc2_set_lambda(w->pc->r2, -1.0);
            w->pc_rebuilds += 1;
        } else if (w->pc->r2 != nullptr) {
            as_pc2_free(w->pc->r2);
            w->pc->r2 = nullptr;
            w->pc->age = 0;
            w->pc_rebuilds += 1;
        } else {
            as_pc_free(w->pc);
            w->pc = nullptr;
            w->pc_dropped += 1;
        }
    }
    return BQ_ERR_NOT_PD;
}

// the body of one outer iteration of BQ_AS_CG once the top record has been read (bq_as_iterate): restricted solve, then the
// reference's two branches (active_set.py:153-220) with Q x carried along without a product where it can be
int as_cg_iterate(bq_solver *s, as_ws *w) {
    hipStream_t st = s->p->ctx->stream;
    const int64_t N = s->N;
    BQ_TRY(as_cg_solve(s, w));
    // Q x of the new point without a product: Q cand is known from the inner iteration (Q z + Q delta), the ratio step is
    // a convex combination.  Every 64th outer iteration forms it afresh so that rounding cannot accumulate.
    const bool inc = w->incq && ++w->since_refresh < 64;
    if (!inc) {
        w->since_refresh = 0;
        // ... and the start product of the NEXT solve is a real product too: Q z = Q cand + columns, Q cand = Q z + Q delta is a
        // chain that the refresh of Q x alone does not re-anchor (the columns are rounded like the panel's entries,
        // not bit-equal to them, and BASELINE config 5 runs ~n outer iterations)
        w->anchor = true;
    }
    if (w->host_ints[2]) {
        as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, w->cand, s->x);
        if (inc) {
            as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, w->Qcand, s->Qd);
            BQ_TRY(bq_vec_eval_f(s->p, s->x, s->Qd, s->g, &s->sc->f));
        } else {
            BQ_TRY(as_eval_f(s, s->g));
        }
        as_launch_release(s, w, st);
    } else {
        as_launch_step(s, w, st);
        if (inc) {
            as_qx_lerp_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->sc, w->Qcand, s->Qd);
            BQ_TRY(bq_vec_eval_f(s->p, s->x, s->Qd, nullptr, &s->sc->f));
        } else {
            BQ_TRY(as_eval_f(s, nullptr));
        }
        as_launch_absorb(s, w, st);
    }
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

// buffers, switches and the preconditioner of a BQ_AS_CG solver (bq_as_start)
int as_cg_create(bq_solver *s, as_ws *w) {
    bq_ctx *ctx = s->p->ctx;
    for (double **v : {&w->dlt, &w->r, &w->pv, &w->Qp, &w->sol, &w->Qdl, &w->Qcand}) {
        BQ_HIP(hipMalloc(v, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(*v, 0, sizeof(double) * s->ldN, ctx->stream));
    }
    BQ_HIP(hipMalloc(&w->cg, sizeof(as_cg_scal)));
    BQ_HIP(hipMemsetAsync(w->cg, 0, sizeof(as_cg_scal), ctx->stream));
    BQ_HIP(hipHostMalloc(&w->cg_flag_host, 2 * sizeof(int)));
    BQ_HIP(hipEventCreateWithFlags(&w->cg_event, hipEventDisableTiming));
    w->warm = bq_hook_on("as_cg_warm");
    w->incq = bq_hook_on("as_cg_incq");
    {
        bq_problem *p = s->p;
        w->colq = w->warm && bq_hook_on("as_cg_colq") && p->X != nullptr && !p->streamed &&
                  (p->structure == BQ_PLAIN || p->structure == BQ_SVC) && p->kernel >= BQ_KERNEL_LINEAR &&
                  p->kernel <= BQ_KERNEL_SIGMOID;
        if (w->colq) {
            BQ_HIP(hipMalloc(&w->sq, sizeof(double) * s->ldN));
            BQ_HIP(hipMalloc(&w->zchg, sizeof(int) * (2 + AS_MAX_COLS)));
            BQ_HIP(hipMemsetAsync(w->zchg, 0, sizeof(int) * (2 + AS_MAX_COLS), ctx->stream));
            BQ_HIP(hipMalloc(&w->zdl, sizeof(double) * AS_MAX_COLS));
            as_row_norms_kernel<<<(unsigned)(s->ldN / 256), 256, 0, ctx->stream>>>(p->X, p->n, p->d, w->sq, s->ldN);
        }
    }
    BQ_TRY(as_pc_create(s, &w->pc));
    return BQ_OK;
}
