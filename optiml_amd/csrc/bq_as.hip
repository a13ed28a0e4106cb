// Active-set driver for the box QP: mask bookkeeping, gathers, KKT sign scan and the restricted Cholesky solve on
// the device.  Restates optiml/opti/constrained/active_set.py:84-230.
//
// Per iteration (reference line numbers in parentheses):
//   top      record (iter, f, |L|+|U|), stop at max_iter (:103-114)
//   restrict idx = compact(A);  z = ub on U, lb on L, 0 on A;  rhs_A = -(q_A + (Q z)_A)  — one panel product instead
//            of the reference's two gathered products Q[A,U] ub_U + Q[A,L] lb_L (:132-136)
//   solve    H = Q[A,A] gathered straight from the resident panel, blocked Cholesky, two triangular solves (:141)
//   branch   candidate inside the box (+-1e-12)?  (:153)
//     yes:   x = candidate; f, g = evaluate; release the FIRST index of L with g < -1e-12, else the first of U with
//            g > 1e-12 (Bland), else 'optimal' (:156-189)
//     no:    ratio step towards the candidate, f = evaluate, move variables that hit a bound into L / U (:195-220)
// The restricted order |A| changes every iteration, so the host reads small records per iteration (|A|, the dot products of a
// new slot, {pivot info, feasible}) — posted by the kernels into mapped pinned memory, three sequence numbers the host spins on
// (as_ws::mail: no copy commands, no stream drains); everything else stays on the stream.  f(x) after a ratio step comes from the
// step's own identity (as_step_min_kernel); a panel product is spent on f only at release iterations.  When Q[A,A] is not positive definite the reference
// silently switches to scipy's minres on the normal equations (:142-151): here a persistent single-workgroup MINRES
// kernel (bq_minres.hip) takes over for |A| <= 8192.
//
// Factor re-use (default for every non-empty free set; hook as_schur=0 re-factorises every iteration as the reference does): between
// two consecutive iterations the free set changes by one or a few indices, so the Cholesky factor of a BASE set A0 is
// kept and the current restricted system is solved through its Schur complement — variables of A0 that have reached a
// bound since are pinned by a multiplier row (x_k = bound), variables released since are bordered on:
//     [ Q00  U ] [y]   [b0]        U = [ Q[A0, added] | e_removed ],  V = [ Q[added, added] 0 ; 0 0 ]
//     [ U'   V ] [w] = [b1]        C = V - U' Q00^-1 U  (m x m, m <= 96 ... 512),  w = C^-1 (b1 - U' Q00^-1 b0),  y = Q00^-1 (b0 - U w)
// One new column Q00^-1 u (two triangular sweeps) per changed index and one Q00^-1 b0 per iteration replace the
// |A|^3/3 factorisation; C is kept on the host and solved there.  The base is re-factorised after 160 (|A| < 8 192) to 1536 (|A| >= 80 000) changes, when C
// is numerically singular, or when the classic path is needed (non-positive pivot -> the reference's minres branch).
//
// BQ_AS_CG (SURVEY 7 "hard parts": ActiveSet beyond the sizes a dense factor fits): the same outer logic, but the
// restricted system is solved by conjugate gradients on the masked panel operator v -> m.(Q (m.v)), m = indicator of
// A — no H, no factorisation, one panel product per inner iteration, so it runs on sharded, fp32-stored and streamed
// panels and on any number of ranks (the vectors and scalar recurrences are replicated, reductions fixed-order).  The
// iteration starts from the candidate of the previous outer iteration (the free set has moved by one index since) and, on
// RBF / linear panels, is preconditioned by a diagonal + low-rank model of Q[A,A] built from explicit features (struct
// as_pc): at BASELINE config 5 that takes the inner iterations per outer iteration from ~80 to ~30 (round 3).
#include "bq_as.h"

__global__ __launch_bounds__(64) void as_top_kernel(bq_scal *sc, int *__restrict__ ints, bq_iter_stat *stats, int n_all,
                                                    int *__restrict__ mail_ints, int *__restrict__ mail_scal, int *mail, int seq) {
    if (threadIdx.x == 0) {
        ints[27] = n_all;   // as_release_mb_kernel: running minima of the candidate indices
        ints[28] = n_all;
        if (!sc->done) {
            const long long row = sc->iter - sc->stat_base;
            if (row >= 0 && row < sc->stat_cap) {
                bq_iter_stat st;
                st.iter = sc->iter;
                st.f = sc->f;
                st.r1 = (double)ints[1];
                st.r2 = -1.0;
                st.r3 = 0.0;
                stats[row] = st;
            }
            if (sc->iter >= sc->max_iter) {
                sc->status = BQ_STATUS_STOPPED;
                sc->done = 1;
            }
        }
        __threadfence();
    }
    if (mail == nullptr) return;
    __syncthreads();
    const int l = threadIdx.x;
    if (l < 32) mail_ints[l] = __hip_atomic_load(ints + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int j = l; j < (int)(sizeof(bq_scal) / sizeof(int)); j += 64)
        mail_scal[j] = __hip_atomic_load(reinterpret_cast<int *>(sc) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    if (l == 0) as_post(mail, seq);
}

__global__ void as_make_z_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                 const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ z) {
    VEC_LOOP(i) {
        if (i < N) z[i] = mU[i] ? ub[i] : (mL[i] ? lb[i] : 0.0);
    }
}

// rhs[a] = -(q[idx[a]] + Qz[idx[a]]) for a < nA, 0 on the pad
__global__ void as_gather_rhs_kernel(const int *__restrict__ ints, const int *__restrict__ idx,
                                     const double *__restrict__ q, const double *__restrict__ Qz,
                                     double *__restrict__ rhs, int64_t np) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= np) return;
    const int nA = ints[0];
    rhs[a] = (a < nA) ? -(q[idx[a]] + Qz[idx[a]]) : 0.0;
}

// the iteration's snapshot of x and g in one launch
__global__ void as_copy2_kernel(int64_t N, const double *__restrict__ a, double *__restrict__ da, const double *__restrict__ b,
                                double *__restrict__ db) {
    VEC_LOOP(i) {
        if (i < N) {
            da[i] = a[i];
            db[i] = b[i];
        }
    }
}

// free-set compaction, pass 1: part[b] = number of free indices in block b's 1024 elements; the last block turns the counts
// into exclusive offsets (in place) and writes ints[0] = |A|, ints[1] = |L| + |U|
__global__ __launch_bounds__(256) void as_count_free_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                            const unsigned char *__restrict__ mU, int *__restrict__ cnt,
                                                            int *__restrict__ ints, unsigned int *ticket) {
    __shared__ int wt[4];
    int c = 0;
    VEC_LOOP(i) c += (i < N && !(mL[i] | mU[i])) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) wt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&cnt[blockIdx.x], wt[0] + wt[1] + wt[2] + wt[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (as_last_block_mb(ticket) && threadIdx.x == 0) {
        int run = 0;
        for (unsigned int b = 0; b < gridDim.x; ++b) {
            const int v = __hip_atomic_load(&cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cnt[b] = run;
            run += v;
        }
        ints[0] = run;
        ints[1] = (int)N - run;
    }
}
// pass 2: idx[offset of the block + rank inside the block] = i, ascending (element j of thread t: index b*1024 + j*256 + t)
__global__ __launch_bounds__(256) void as_write_free_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                            const unsigned char *__restrict__ mU, const int *__restrict__ cnt,
                                                            int *__restrict__ idx) {
    __shared__ int wt[BQ_VEC_ITEMS][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int flag[BQ_VEC_ITEMS], within[BQ_VEC_ITEMS];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        flag[j] = (i < N && !(mL[i] | mU[i])) ? 1 : 0;
        const unsigned long long bal = __ballot(flag[j]);
        within[j] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wt[j][wv] = __popcll(bal);
    }
    __syncthreads();
    int off = cnt[blockIdx.x];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        int o = off;
        for (int w = 0; w < wv; ++w) o += wt[j][w];
        if (flag[j]) idx[o + within[j]] = (int)((int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x);
        off += wt[j][0] + wt[j][1] + wt[j][2] + wt[j][3];
    }
}

// ratio test of the step towards the candidate: sc->step = min over the free set (active_set.py:166-173), partial minima per
// block, the last block takes the minimum over them.
//
// chain != 0 — f(x + t d) WITHOUT a panel product (the second half of round 4).  The reference evaluates f afresh after the ratio
// step (active_set.py:172-176: three products with Q).  But d = cand - x lives on the free set A and cand solves the restricted
// system, so Q_AA d_A = -g_A(x) (a Newton step on A), hence  d'Qd = -g_A'd_A  and
//       f(x + t d) = f(x) + t g_A'd_A + t^2/2 d'Qd = f(x) + (t - t^2/2) g_A'd_A,      g_A(x + t d) = (1 - t) g_A(x):
// along a run of ratio steps the gradient on the (shrinking) free set is the gradient g0 of the last RELEASE iteration — the one
// s->g still holds, fresh from that iteration's product — times gamma = prod (1 - t_j).  So the kernel also sums g0_A'd_A over the
// free set (fixed order: per block, then over the blocks) and its last block moves f and gamma (sc->aux[0]; as_release_mb_kernel
// sets it back to 1).  The host asks for this only while every candidate since that release came from a factorisation (not from
// the minimum-residual branch) and forms f by a product again every 64th step of a run (chain == 2: only gamma moves; as_finish_iteration);
// hook as_f_chain=0: never.
__global__ __launch_bounds__(256) void as_step_min_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                          const unsigned char *__restrict__ mU, const double *__restrict__ cand,
                                                          const double *__restrict__ lb, const double *__restrict__ ub,
                                                          const double *__restrict__ x, const double *__restrict__ g0, double *part,
                                                          double *part_s, bq_scal *sc, int chain) {
    __shared__ double sh[4], shs[4];
    double rmin = INFINITY, sgd = 0.0;
    VEC_LOOP(i) {
        if (i < N && !(mL[i] | mU[i])) {
            const double d = cand[i] - x[i];
            if (d > 0.0) rmin = fmin(rmin, (ub[i] - x[i]) / d);
            if (d < 0.0) rmin = fmin(rmin, (lb[i] - x[i]) / d);
            if (chain == 1) sgd = fma(g0[i], d, sgd);
        }
    }
    rmin = as_wmin(rmin);
    if (chain) sgd = as_wsum_any(sgd);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = rmin;
        shs[threadIdx.x >> 6] = sgd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&part[blockIdx.x], fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (chain) __hip_atomic_store(&part_s[blockIdx.x], ((shs[0] + shs[1]) + shs[2]) + shs[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (as_last_block_mb(&sc->ticket[0])) {
        double m = INFINITY, a = 0.0;
        for (unsigned int b = threadIdx.x; b < gridDim.x; b += 256) {
            m = fmin(m, __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (chain) a += __hip_atomic_load(&part_s[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        m = as_wmin(m);
        if (chain) a = as_wsum_any(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            sh[threadIdx.x >> 6] = m;
            shs[threadIdx.x >> 6] = a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
            sc->step = t;
            if (chain == 1) {
                const double gd = sc->aux[0] * (((shs[0] + shs[1]) + shs[2]) + shs[3]);
                sc->f = sc->f + (t - 0.5 * t * t) * gd;
            }
            if (chain) sc->aux[0] = sc->aux[0] * (1.0 - t);   // chain == 2: f comes from a product this time, the run goes on
        }
    }
}
__global__ void as_step_apply_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                     const double *__restrict__ cand, double *__restrict__ x, const bq_scal *sc) {
    const double t = sc->step;
    VEC_LOOP(i) {
        if (i < N && !(mL[i] | mU[i])) x[i] = x[i] + __dmul_rn(t, cand[i] - x[i]);
    }
}

// free variables that reached a bound move into L / U (L first, as the reference); ints[5], ints[6] = how many, ints[8] = the
// total, ints[9 ..] = up to 16 of their indices (the host sorts them); the last block closes the iteration.
// ints[26] is the slot counter of this launch (zeroed by the last block for the next one; ints[7] belongs to MINRES).
__global__ __launch_bounds__(256) void as_absorb_mb_kernel(int64_t N, unsigned char *__restrict__ mL, unsigned char *__restrict__ mU,
                                                           const double *__restrict__ x, const double *__restrict__ lb,
                                                           const double *__restrict__ ub, bq_scal *sc, int *__restrict__ ints,
                                                           int *__restrict__ cnt, bq_iter_stat *stats) {
    __shared__ int wl[4], wu[4];
    int nl = 0, nu = 0;
    VEC_LOOP(i) {
        if (i < N && !(mL[i] | mU[i])) {
            bool hit = false;
            // absorbed within the 1e-12 tolerance but further from the bound than the rounding of the step that brought it there (a
            // blocking variable lands on its bound to an ulp or two: 1e-14 relative is that, with room) — a NEAR-TIE of the ratio test.
            // The reference does not snap x to the bound either (active_set.py:208-217), so neither does this kernel; see ints[29].
            bool off = false;
            if (x[i] <= lb[i] + ACT_TOL) {
                mL[i] = 1;
                ++nl;
                hit = true;
                off = fabs(x[i] - lb[i]) > 1e-14 * fmax(1.0, fabs(lb[i]));
            } else if (x[i] >= ub[i] - ACT_TOL) {
                mU[i] = 1;
                ++nu;
                hit = true;
                off = fabs(x[i] - ub[i]) > 1e-14 * fmax(1.0, fabs(ub[i]));
            }
            if (off) __hip_atomic_store(&ints[30], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every writer stores 1
            if (hit) {
                const int slot = atomicAdd(&ints[26], 1);
                if (slot < 16) ints[9 + slot] = (int)i;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nl += __shfl_down(nl, off, 64);
        nu += __shfl_down(nu, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        wl[threadIdx.x >> 6] = nl;
        wu[threadIdx.x >> 6] = nu;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&cnt[2 * blockIdx.x], wl[0] + wl[1] + wl[2] + wl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cnt[2 * blockIdx.x + 1], wu[0] + wu[1] + wu[2] + wu[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (as_last_block_mb(&sc->ticket[1]) && threadIdx.x == 0) {
        int tl = 0, tu = 0;
        for (unsigned int b = 0; b < gridDim.x; ++b) {
            tl += __hip_atomic_load(&cnt[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tu += __hip_atomic_load(&cnt[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ints[5] = tl;
        ints[6] = tu;
        ints[8] = __hip_atomic_load(&ints[26], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ints[26], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ints[29]: THIS step bound a variable that sits off its bound by 1e-14 .. 1e-12 — the product-free f of the following ratio
        // steps assumes bound variables ON their bounds (as_step_min_kernel), so the host ends the run (as_finish_iteration)
        ints[29] = __hip_atomic_load(&ints[30], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ints[30], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long row = sc->iter - sc->stat_base;
        if (row >= 0 && row < sc->stat_cap) {
            stats[row].r2 = 0.0;
            stats[row].r3 = (double)(((long long)tl << 32) | (long long)tu);
        }
        sc->iter += 1;
    }
}

// first index of L with g < -tol and of U with g > tol (Bland's rule, active_set.py:178-199): block minima meet in two atomic
// minima (ints[27], ints[28], reset by as_top_kernel); the last block releases the variable, writes the record and closes the
// iteration — or declares the point optimal
__global__ __launch_bounds__(256) void as_release_mb_kernel(int64_t N, const double *__restrict__ g, unsigned char *__restrict__ mL,
                                                            unsigned char *__restrict__ mU, bq_scal *sc, int *__restrict__ ints,
                                                            bq_iter_stat *stats) {
    __shared__ int sl[4], su[4];
    int hl = (int)N, hu = (int)N;
    VEC_LOOP(i) {
        if (i < N) {
            if (mL[i] && g[i] < -ACT_TOL && (int)i < hl) hl = (int)i;
            if (mU[i] && g[i] > ACT_TOL && (int)i < hu) hu = (int)i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        hl = min(hl, __shfl_down(hl, off, 64));
        hu = min(hu, __shfl_down(hu, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        sl[threadIdx.x >> 6] = hl;
        su[threadIdx.x >> 6] = hu;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hl = min(min(sl[0], sl[1]), min(sl[2], sl[3]));
        hu = min(min(su[0], su[1]), min(su[2], su[3]));
        if (hl < (int)N) atomicMin(&ints[27], hl);
        if (hu < (int)N) atomicMin(&ints[28], hu);
    }
    if (as_last_block_mb(&sc->pad1[1]) && threadIdx.x == 0) {
        const long long ghl = __hip_atomic_load(&ints[27], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long ghu = __hip_atomic_load(&ints[28], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ints[3] = (int)ghl;
        ints[4] = (int)ghu;
        sc->aux[0] = 1.0;   // g is fresh: a run of product-free ratio steps starts here (as_step_min_kernel)
        const long long row = sc->iter - sc->stat_base;
        const bool rec = row >= 0 && row < sc->stat_cap;
        if (ghl < N) {
            mL[ghl] = 0;
            if (rec) {
                stats[row].r2 = 1.0;
                stats[row].r3 = (double)ghl;
            }
            sc->iter += 1;
        } else if (ghu < N) {
            mU[ghu] = 0;
            if (rec) {
                stats[row].r2 = 2.0;
                stats[row].r3 = (double)ghu;
            }
            sc->iter += 1;
        } else {
            if (rec) stats[row].r2 = 3.0;
            sc->status = BQ_STATUS_OPTIMAL;
            sc->done = 1;
        }
    }
}



// launches of the multi-block per-iteration steps; their per-block partials live in three disjoint slices of s->partials
static_assert(BQ_MAX_PARTIAL_Q >= 3, "as_launch_*: three slices of the partials buffer");
void as_launch_compact(bq_solver *s, as_ws *w, hipStream_t st) {
    int *cnt = reinterpret_cast<int *>(s->partials + s->nblk);
    as_count_free_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, cnt, w->ints, &s->sc->pad1[0]);
    as_write_free_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, cnt, w->idx);
}
static_assert(BQ_MAX_PARTIAL_Q >= 4, "as_launch_step: a fourth slice of the partials buffer");
void as_launch_step(bq_solver *s, as_ws *w, hipStream_t st, int chain) {
    as_step_min_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, w->cand, s->lb, s->ub, s->x, w->gref ? w->gref : s->g,
                                                              s->partials, s->partials + 3 * s->nblk, s->sc, chain);
    as_step_apply_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, w->cand, s->x, s->sc);
}
void as_launch_absorb(bq_solver *s, as_ws *w, hipStream_t st) {
    int *cnt = reinterpret_cast<int *>(s->partials + 2 * s->nblk);
    as_absorb_mb_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->mL, s->mU, s->x, s->lb, s->ub, s->sc, w->ints, cnt, s->stats);
}

int as_eval_f(bq_solver *s, double *g_out) {
    // Qd = Q x ; f = 1/2 x'Qx + q'x -> sc->f ; optionally g = Qx + q
    BQ_TRY(bq_problem_apply(s->p, s->x, s->Qd, nullptr));
    return bq_vec_eval_f(s->p, s->x, s->Qd, g_out, &s->sc->f);
}

// the dense iteration's two branches once the candidate is known (w->host_ints[2]: it is feasible).  `exact`: the candidate came
// from a factorisation (kept or fresh), not from the minimum-residual branch — the condition of the product-free f of a ratio step
void as_launch_release(bq_solver *s, as_ws *w, hipStream_t st) {
    as_release_mb_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, s->g, s->mL, s->mU, s->sc, w->ints, s->stats);
}

int as_finish_iteration(bq_solver *s, as_ws *w, hipStream_t st, bool exact) {
    const int64_t N = s->N;
    if (w->host_ints[2]) {
        as_copy_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, w->cand, s->x);
        BQ_TRY(as_eval_f(s, s->g));
        as_release_mb_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->g, s->mL, s->mU, s->sc, w->ints, s->stats);
        w->chain_ok = exact && w->f_chain;   // x solves its restricted system and g is fresh: a run may start
        w->gref = s->g;
        w->chain_len = 0;
    } else {
        // a variable absorbed OFF its bound in the previous step (1e-14 < |x - bound| <= 1e-12: a near-tie of the ratio test):
        // Q_AA d_A = -g_A(x) then carries an O(1e-12 |Q_AB|) error per such index until x is replaced by a feasible candidate — the run ends here, f is formed by products
        // until the next release re-anchors it (host_ints: this iteration's top record, i.e. the state the previous body left)
        if (w->host_ints[29]) w->chain_ok = false;
        const bool run = w->chain_ok && exact;          // the run's identity holds for this step
        const bool chain = run && w->chain_len < 64;    // ... and f is taken from it (every 64th step of a run: from a product)
        as_launch_step(s, w, st, chain ? 1 : (run ? 2 : 0));
        if (chain) {
            w->chain_len += 1;
            w->chain_steps += 1;
        } else {
            BQ_TRY(as_eval_f(s, nullptr));
            w->chain_len = 0;          // f is anchored again; the run itself goes on as long as the candidates stay exact
            if (!exact) w->chain_ok = false;
        }
        as_launch_absorb(s, w, st);
    }
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

int bq_as_start(bq_solver *s) {
    if (s->chol) {   // a numerically singular Q_AA must take the minres branch, not the sign of its rounding noise (bq_chol.h)
        const char *e = getenv("BQ_AS_PIVOT_REL");
        s->chol->pivot_rel = e ? atof(e) : 1e-13;
    }
    bq_ctx *ctx = s->p->ctx;
    as_ws *w = new as_ws();
    s->as_ws = w;
    BQ_HIP(hipMalloc(&w->idx, sizeof(int) * (s->N + 1)));
    BQ_HIP(hipMemsetAsync(w->idx, 0, sizeof(int) * (s->N + 1), ctx->stream));
    BQ_HIP(hipMalloc(&w->ints, sizeof(int) * 32));
    BQ_HIP(hipMemsetAsync(w->ints, 0, sizeof(int) * 32, ctx->stream));
    BQ_HIP(hipHostMalloc(&w->host_ints, sizeof(int) * 32, AS_MAPPED));
    BQ_HIP(hipHostMalloc(&w->host_scal, sizeof(bq_scal), AS_MAPPED));
    BQ_HIP(hipHostMalloc(&w->mail, sizeof(int) * 16, AS_MAPPED));
    memset(w->mail, 0, sizeof(int) * 16);
    w->mail_d = as_dev(w->mail);
    w->host_ints_d = as_dev(w->host_ints);
    w->host_scal_d = reinterpret_cast<int *>(as_dev(w->host_scal));
    BQ_HIP(hipMalloc(&w->mail_ticket, sizeof(unsigned int) * 2));
    BQ_HIP(hipMemsetAsync(w->mail_ticket, 0, sizeof(unsigned int) * 2, s->p->ctx->stream));
    w->mailbox = !s->as_cg && bq_hook_on("as_mailbox");
    w->f_chain = !s->as_cg && bq_hook_on("as_f_chain");
#ifdef BQ_AS_TIMING   // diagnostic build (BQ_EXTRA_CXXFLAGS=-DBQ_AS_TIMING): where the host's time goes per iteration, printed by bq_as_free
    w->timing = true;
#endif
    BQ_HIP(hipHostMalloc(&w->host_info, sizeof(int) * 8));
    BQ_HIP(hipHostMalloc(&w->host_cg, sizeof(as_cg_scal)));
    memset(w->host_ints, 0, sizeof(int) * 32);
    memset(w->host_info, 0, sizeof(int) * 8);
    for (double **v : {&w->cand, &w->z, &w->Qz, &w->x_eval, &w->g_eval}) {
        BQ_HIP(hipMalloc(v, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(*v, 0, sizeof(double) * s->ldN, ctx->stream));
    }
    if (s->as_cg) BQ_TRY(as_cg_create(s, w));
    BQ_HIP(hipMalloc(&s->mL, (size_t)s->ldN));
    BQ_HIP(hipMalloc(&s->mU, (size_t)s->ldN));
    BQ_HIP(hipMemsetAsync(s->mL, 0, (size_t)s->ldN, ctx->stream));
    BQ_HIP(hipMemsetAsync(s->mU, 0, (size_t)s->ldN, ctx->stream));
    // bq_solver_set_state: continue from given masks L / U (the reference starts with both empty, active_set.py:91-92).  The host
    // vectors live until solver_first has synchronised.
    const bool resumed = s->resume && (s->resume->have & BQ_STATE_MASKS);
    if (resumed) {
        BQ_HIP(hipMemcpyAsync(s->mL, s->resume->mL.data(), (size_t)s->N, hipMemcpyHostToDevice, ctx->stream));
        BQ_HIP(hipMemcpyAsync(s->mU, s->resume->mU.data(), (size_t)s->N, hipMemcpyHostToDevice, ctx->stream));
    }
    if (w->f_chain && resumed) {
        // the product-free f of ratio steps needs a point that solves its restricted system and the gradient there: neither is
        // known for a restored point, so the first run starts at the next release (as_finish_iteration), as after a minres step
        BQ_HIP(hipMalloc(&w->g0, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(w->g0, 0, sizeof(double) * s->ldN, ctx->stream));
        static const double one = 1.0;
        BQ_HIP(hipMemcpyAsync(&s->sc->aux[0], &one, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        w->gref = nullptr;
        w->chain_ok = false;
        return as_eval_f(s, nullptr);
    }
    if (w->f_chain) {
        // the gradient at x0 rides on the product of f(x0): with every index free, x0 + t d obeys the ratio step's identity from the
        // first iteration on (as_step_min_kernel).  Kept apart from s->g, which the reference does not touch before a release.
        BQ_HIP(hipMalloc(&w->g0, sizeof(double) * s->ldN));
        BQ_HIP(hipMemsetAsync(w->g0, 0, sizeof(double) * s->ldN, ctx->stream));
        static const double one = 1.0;   // gamma of the first run (as_release_mb_kernel sets it for the later ones)
        BQ_HIP(hipMemcpyAsync(&s->sc->aux[0], &one, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        w->gref = w->g0;
        w->chain_ok = true;
        return as_eval_f(s, w->g0);
    }
    return as_eval_f(s, nullptr);  // f(x0), active_set.py:84
}

void bq_as_free(bq_solver *s) {
    as_ws *w = get_ws(s);
    if (!w) return;
    if (w->timing && w->tm[7] > 0) {
        const double it = w->tm[7];
        fprintf(stderr, "BQ_AS_TIMING  %.0f iterations, us per iteration: wait top %.1f | host to dots %.1f | wait dots %.1f | small system %.1f | "
                        "launch to candidate %.1f | wait candidate %.1f | launch branch + top %.1f\n", it, w->tm[0] / it, w->tm[1] / it, w->tm[2] / it,
                w->tm[3] / it, w->tm[4] / it, w->tm[5] / it, w->tm[6] / it);
        if (w->sch)
            fprintf(stderr, "BQ_AS_TIMING  host LDL, us per iteration: new row %.1f | forward %.1f | back %.1f | residual check %.1f | C row/column %.1f\n",
                    w->sch->t_ldl[0] / it, w->sch->t_ldl[1] / it, w->sch->t_ldl[2] / it, w->sch->t_ldl[3] / it, w->sch->t_c / it);
        fprintf(stderr, "BQ_AS_TIMING  iterations that factorised Q[A,A] afresh: %lld, mean order %.0f, %.1f us each (launch to the candidate's record)\n", w->n_classic,
                w->classic_order / std::max(1.0, (double)w->n_classic), w->tm_classic / std::max(1.0, (double)w->n_classic));
        if (w->sch)
            fprintf(stderr, "BQ_AS_TIMING  small system: %lld solves, mean order %.1f, rows (re)factorised per solve %.2f, slots dropped %lld, base factorisations %lld\n",
                    w->sch->reused + w->sch->refreshes, (double)w->sch->rows_solved / std::max(1.0, (double)(w->sch->reused + w->sch->refreshes)),
                    (double)w->sch->rows_extended / std::max(1.0, (double)(w->sch->reused + w->sch->refreshes)), w->sch->drops, w->sch->refreshes);
    }
    for (void *p : {(void *)w->idx, (void *)w->ints, (void *)w->cand, (void *)w->z, (void *)w->Qz, (void *)w->x_eval,
                    (void *)w->g_eval, (void *)w->dlt, (void *)w->r, (void *)w->pv, (void *)w->Qp, (void *)w->sol,
                    (void *)w->cg, (void *)w->Qdl, (void *)w->Qcand, (void *)w->sq, (void *)w->zchg, (void *)w->zdl,
                    (void *)w->mail_ticket, (void *)w->g0})
        if (p) hipFree(p);
    for (void *hp : {(void *)w->host_ints, (void *)w->host_scal, (void *)w->host_info, (void *)w->host_cg, (void *)w->mail})
        if (hp) hipHostFree(hp);
    if (w->cg_flag_host) hipHostFree(w->cg_flag_host);
    if (w->cg_event) hipEventDestroy(w->cg_event);
    as_pc_free(w->pc);
    as_schur_free(w);
    delete w;
    s->as_ws = nullptr;
}

// x / g as the callback of the last recorded iteration must see them (the body has already moved on)
long long bq_as_inner_iters(bq_solver *s) {
    as_ws *w = get_ws(s);
    return w ? w->cg_iters : 0;
}

long long bq_as_counter(bq_solver *s, int which) {
    as_ws *w = get_ws(s);
    if (!w) return 0;
    switch (which) {
        case BQ_COUNT_INNER: return w->cg_iters;
        case BQ_COUNT_MINRES: return w->minres_calls;
        case BQ_COUNT_REFACTOR: return w->sch ? w->sch->refreshes : 0;
        case BQ_COUNT_REUSED: return w->sch ? w->sch->reused : 0;
        case BQ_COUNT_NO_PRODUCT: return w->chain_steps;
        default: return 0;
    }
}

const double *bq_as_view(bq_solver *s, int what) {
    as_ws *w = get_ws(s);
    if (!w || !s->started) return what == BQ_GET_X ? s->x : s->g;
    return what == BQ_GET_X ? w->x_eval : w->g_eval;
}

int bq_as_iterate(bq_solver *s) {
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    as_ws *w = get_ws(s);
    bq_chol_ws *ws = s->chol;
    const int64_t N = s->N;

    as_launch_compact(s, w, st);
    const bool mbx = w->mailbox;
    as_top_kernel<<<1, 64, 0, st>>>(s->sc, w->ints, s->stats, (int)N, w->host_ints_d, w->host_scal_d,
                                   mbx ? w->mail_d : nullptr, mbx ? ++w->mail_seq[0] : 0);
    if (!s->host.done) {  // snapshot of the point this record describes
        as_copy2_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->x, w->x_eval, s->g, w->g_eval);
        s->started = true;
    }
    if (s->as_cg && w->pc) as_pc_track(s, w, st);
    if (!mbx) {
        BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
        BQ_HIP(hipMemcpyAsync(w->host_scal, s->sc, sizeof(bq_scal), hipMemcpyDeviceToHost, st));
    }
    BQ_HIP(hipGetLastError());
    as_tick(w, w->tm[7] > 0 ? 6 : -1);
    BQ_TRY(as_look(ctx, w, 0));
    as_tick(w, 0);
    w->tm[7] += 1;
    s->host = *w->host_scal;
    if (s->as_cg && w->pc) {
        w->pc->host_chg[0] = w->host_info[4];
        w->pc->host_chg[1] = w->host_info[5];
    }
    if (s->host.done) return BQ_OK;
    const int64_t nA = w->host_ints[0];

    if (s->as_cg) return as_cg_iterate(s, w);
    bool solved = false;
    // (an empty free set has nothing to keep: with hook as_schur_min=0 it reached the kept-factor path and launched empty grids)
    if (as_schur_enabled() && nA > 0 && nA >= as_schur_min()) BQ_TRY(as_schur_step(s, w, nA, &solved));
    if (!solved && w->sch) w->sch->valid = false;   // the classic path below overwrites the kept factor
    if (solved) {
        w->last_branch = w->host_ints[2] ? 1 : 0;
        return as_finish_iteration(s, w, st, true);
    }
    w->n_classic += 1;
    w->classic_order += (double)nA;
    const double tcl0 = w->timing ? ldl_now() : 0.0;
    struct classic_tail {
        as_ws *w;
        double t0;
        ~classic_tail() {
            if (w->timing) w->tm_classic += ldl_now() - t0;
        }
    } ctail{w, tcl0};
    as_make_z_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->z);
    BQ_TRY(bq_problem_apply(s->p, w->z, w->Qz, nullptr));
    int64_t np = 0;
    BQ_TRY(bq_chol_build_h(ws, s->p, w->idx, nA, nullptr, &np));
    as_gather_rhs_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(w->ints, w->idx, s->p->q, w->Qz, ws->rhs, np);
    BQ_TRY(bq_chol_factor(ws, np));
    BQ_TRY(bq_chol_solve(ws, np));
    as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
    as_cand_scatter_idx_kernel<<<vgrid(s->ldN).x * (BQ_VEC_TILE / 256), 256, 0, st>>>(w->idx, w->ints, ws->rhs, s->lb, s->ub, w->cand);
    BQ_HIP(hipMemcpyAsync(w->host_info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    const int info = w->host_info[0];
    if (info != 0) {
        // Q[A,A] is not positive definite: the reference's bare `except` switches to scipy's minres on the normal
        // equations (active_set.py:142-151).  Rebuild the (destroyed) restricted Hessian with both triangles, solve,
        // and redo the feasibility test on the minimum-residual candidate.
        if (!ws->mr_vec) BQ_HIP(hipMalloc(&ws->mr_vec, sizeof(double) * 10 * ws->cap));
        BQ_TRY(bq_chol_build_h(ws, s->p, w->idx, nA, nullptr, &np, true));
        as_gather_rhs_kernel<<<(unsigned)((np + 255) / 256), 256, 0, st>>>(w->ints, w->idx, s->p->q, w->Qz, ws->rhs, np);
        BQ_TRY(bq_minres_normal(ws, w->ints, nA, np, ws->mr_vec, w->ints + 7));
        as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
        as_cand_scatter_idx_kernel<<<vgrid(s->ldN).x * (BQ_VEC_TILE / 256), 256, 0, st>>>(w->idx, w->ints, ws->rhs, s->lb, s->ub, w->cand);
        BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
        BQ_SYNC(s->p->ctx);
        w->minres_calls += 1;
    }
    w->last_branch = w->host_ints[2] ? 1 : 0;
    return as_finish_iteration(s, w, st, info == 0);
}
