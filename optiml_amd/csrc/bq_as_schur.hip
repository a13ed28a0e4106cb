// Factor re-use of the dense ActiveSet: the Cholesky factor of a base free set is kept and the current restricted system is solved
// through its Schur complement (header comment of bq_as.hip).  Split out of bq_as.hip in round 5.
#include "bq_as.h"

__global__ void as_cand_scatter_kernel(int64_t n0, int m, const int *__restrict__ idx0, const int *__restrict__ meta,
                                       const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                       const double *__restrict__ lb, const double *__restrict__ ub,
                                       const double *__restrict__ y, const double *__restrict__ coef,
                                       double *__restrict__ cand, int *__restrict__ ints, unsigned int *ticket,
                                       int *__restrict__ mail_ints, int *mail, int seq) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (a < n0) {
        const int i = idx0[a];
        if (!(mL[i] | mU[i])) {
            const double v = y[a];
            cand[i] = v;
            bad = !(v <= ub[i] + ACT_TOL && v >= lb[i] - ACT_TOL);
        }
    }
    if (a < m && meta[a] == 1) {
        const int i = meta[AS_SCHUR_MAX + a];
        const double v = coef[a];
        cand[i] = v;
        bad = bad || !(v <= ub[i] + ACT_TOL && v >= lb[i] - ACT_TOL);
    }
    if (bad) ints[2] = 0;   // benign race: every writer stores 0
    if (mail == nullptr) return;
    // the last workgroup to get here hands the record (feasibility flag and all) to the host
    __shared__ int last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 32) mail_ints[threadIdx.x] = __hip_atomic_load(ints + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    if (threadIdx.x == 0) {
        *ticket = 0;
        as_post(mail, seq);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// factor re-use: Schur-complement updates of a base factorisation (see the header comment)
// ---------------------------------------------------------------------------------------------------------------
// changed indices carried before the base is re-factorised: the larger the factor, the longer it is worth keeping
// (n^3/3 to rebuild against one more small-system row per carried index)
static int as_schur_limit(int64_t np0) {
    double hv = 0.0;
    if (bq_hook("as_schur_limit", &hv)) return std::max(1, std::min((int)hv, AS_SCHUR_MAX));   // tests
    // (re-tuned in round 2 for the two-launches-per-1024-rows sweeps: a solve is ~4x cheaper, so the n^3/3 of a rebuild is
    // amortised over more iterations: n = 50 000: 0.77 s per rebuild = 3 ms per iteration at 256 carried changes)
    // (below |A| = 8 192: 96 through round 4's first half; swept again once every free set went through the kept factor and the
    // looks became cheap — 48 / 96 / 160 / 256: 11.15 / 10.42 / 10.13 / 10.22 s to 'optimal' at n = 20 000, profiles/r04/as_f_chain.txt)
    return np0 < 8192 ? 160 : (np0 < 40000 ? 384 : (np0 < 80000 ? 768 : 1536));
}



__global__ void as_schur_pos_kernel(int64_t N, int64_t n0, const int *__restrict__ idx0, int *__restrict__ pos0, int pass) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pass == 0) {
        if (t < N) pos0[t] = -1;
    } else if (t < n0) {
        pos0[idx0[t]] = (int)t;
    }
}

// z = the bound value on every bound variable OUTSIDE the base (those inside are pinned by a multiplier row), else 0
__global__ void as_schur_z_kernel(int64_t N, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                  const double *__restrict__ lb, const double *__restrict__ ub,
                                  const int *__restrict__ pos0, double *__restrict__ z) {
    VEC_LOOP(i) {
        if (i < N) z[i] = (pos0[i] < 0 && (mL[i] | mU[i])) ? (mU[i] ? ub[i] : lb[i]) : 0.0;
    }
}

__global__ void as_schur_rhs0_kernel(int64_t n0, int64_t np0, const int *__restrict__ idx0, const double *__restrict__ q,
                                     const double *__restrict__ Qz, double *__restrict__ rhs) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < np0) rhs[a] = a < n0 ? -(q[idx0[a]] + Qz[idx0[a]]) : 0.0;
}

// the column of a new slot: e_pos for a base variable that reached a bound, Q[A0, var] for a freed variable
template <typename T>
__global__ void as_schur_col_kernel(int kind, int var, int pos, int64_t n0, int64_t np0, const int *__restrict__ idx0,
                                    int structure, const T *__restrict__ panel, int64_t ldp, int packed, int64_t n,
                                    const double *__restrict__ sgn, double diag_add, double *__restrict__ out,
                                    double *__restrict__ out2) {
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= np0) return;
    double v = 0.0;
    if (kind == 0)
        v = a == pos ? 1.0 : 0.0;
    else if (a < n0)
        v = bq_q_elem(structure, panel, ldp, packed, n, sgn, diag_add, (int64_t)idx0[a], (int64_t)var);
    out[a] = v;
    out2[a] = v;   // the right-hand side of the solve that follows (was a device-to-device copy)
}

// block i: out[i] = extra_i - U[i]'v.  mode 0 (v = W[k]): extra = V_ik = Q[var_i, var_k] when both were freed, else 0.
// mode 1 (v = y0): extra = the right-hand side of row i: the bound of a pinned variable, -(q + Qz) of a freed one.
// mode 2: BOTH in one pass over U[i] (the usual iteration: one new slot k = m - 1, then the right-hand side): out[i] as mode 0
// with v = v0, out1[i] as mode 1 with v = v1 — each dot product summed exactly as in its own launch (same bits).
template <typename T>
__global__ __launch_bounds__(256) void as_schur_dots_kernel(int mode, int k, const double *__restrict__ U,
                                                            const double *__restrict__ v, const double *__restrict__ v1,
                                                            int64_t cap, int64_t np0,
                                                            const int *__restrict__ meta, int structure,
                                                            const T *__restrict__ panel, int64_t ldp, int packed, int64_t n,
                                                            const double *__restrict__ sgn, double diag_add,
                                                            const unsigned char *__restrict__ mU,
                                                            const double *__restrict__ lb, const double *__restrict__ ub,
                                                            const double *__restrict__ q, const double *__restrict__ Qz,
                                                            double *__restrict__ out, double *__restrict__ out1,
                                                            unsigned int *ticket, int *mail, int seq) {
    __shared__ double sh[4], sh1[4];
    const int i = blockIdx.x;
    const double *u = U + (int64_t)i * cap;
    // the slot table may live in the HOST's (mapped) buffer: thread 0 asks for its four entries before the dot products, not after
    int ki = 0, vi = 0, kk = 0, vk = 0;
    if (threadIdx.x == 0) {
        ki = meta[i];
        vi = meta[AS_SCHUR_MAX + i];
        kk = meta[k];
        vk = meta[AS_SCHUR_MAX + k];
    }
    double s = 0.0, t = 0.0;
    if (mode == 2) {
        for (int64_t a = threadIdx.x; a < np0; a += 256) {
            const double ua = u[a];
            s += __dmul_rn(ua, v[a]);
            t += __dmul_rn(ua, v1[a]);
        }
    } else {
        for (int64_t a = threadIdx.x; a < np0; a += 256) s += __dmul_rn(u[a], v[a]);
    }
    s = as_wsum_any(s);
    if (mode == 2) t = as_wsum_any(t);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = s;
        sh1[threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double dot = ((sh[0] + sh[1]) + sh[2]) + sh[3];
        if (mode != 1) {
            double extra = 0.0;
            if (ki == 1 && kk == 1) extra = bq_q_elem(structure, panel, ldp, packed, n, sgn, diag_add, (int64_t)vi, (int64_t)vk);
            out[i] = extra - dot;
        }
        if (mode != 0) {
            const double d1 = mode == 1 ? dot : ((sh1[0] + sh1[1]) + sh1[2]) + sh1[3];
            const double extra = ki == 0 ? (mU[vi] ? ub[vi] : lb[vi]) : -(q[vi] + Qz[vi]);
            (mode == 1 ? out : out1)[i] = extra - d1;
        }
        if (mail != nullptr) {   // out / out1 are the host's (mapped) buffers: the last workgroup posts the record
            __threadfence_system();
            if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
                *ticket = 0;
                __threadfence_system();
                as_post(mail, seq);
            }
        }
    }
}

// y = y0 - sum_k coef[k] W[k]: 256 rows x 4 interleaved runs of slots per workgroup (1024 threads: four times the loads in
// flight of the one-thread-per-row form, which ran this m x np0 product at 0.65 TB/s), the four runs added in run order
__global__ __launch_bounds__(1024) void as_schur_combine_kernel(int64_t np0, int m, const double *__restrict__ y0,
                                                                const double *__restrict__ W, int64_t cap,
                                                                const double *__restrict__ coef, double *__restrict__ y) {
    __shared__ double part[4][256];
    __shared__ double cf[AS_SCHUR_MAX];   // the coefficients may live in the HOST's (mapped) buffer: one coalesced read per workgroup,
    for (int k = threadIdx.x; k < m; k += 1024) cf[k] = coef[k];   // not one uncached trip over PCIe per term of the sum
    __syncthreads();
    const int r = threadIdx.x & 255, g = threadIdx.x >> 8;
    const int64_t a = (int64_t)blockIdx.x * 256 + r;
    double v = 0.0;
    if (a < np0)
        for (int k = g; k < m; k += 4) v += __dmul_rn(cf[k], W[(int64_t)k * cap + a]);
    part[g][r] = v;
    __syncthreads();
    if (g == 0 && a < np0) y[a] = y0[a] - (((part[0][r] + part[1][r]) + part[2][r]) + part[3][r]);
}

// C = L D L' of the m x m Schur complement, kept on the host and grown by one row per new slot (O(m^2)); C is symmetric
// quasi-definite — minus a positive definite block for the pinned variables, a positive definite one for the freed —
// so it factorises without pivoting in any order.  as_ldl_solve checks the residual; on failure the caller falls back
// to the pivoted elimination below (and from there to a fresh base factor).
// The host's share of a kept-factor iteration is this O(m^2) work between two waits on the stream (the device idles meanwhile:
// profiles/r04/as_n20k_stream_idle_*.txt), so it is compiled a second time for AVX2 + FMA hosts and picked at load time
// (function multi-versioning; the device pass of hipcc does not know the attribute).
#if defined(__HIP_DEVICE_COMPILE__)
#define BQ_HOST_SIMD
#else
#define BQ_HOST_SIMD __attribute__((target_clones("arch=x86-64-v3", "default")))
#endif

// The three loops below are compiled twice (AVX2 and baseline x86-64) and must give the SAME bits in both: the small system's
// solution feeds the 1e-12 feasibility decision of the iteration, and a trajectory must not depend on the host's vector width
// (ADVICE r4).  So the association is written out — eight independent partial sums over j mod 8, combined in one fixed tree — and
// contraction is off (the AVX2 clone has fused multiply-add, the baseline has not: a fused product rounds once, a separate one
// twice).  The vectoriser needs no licence to re-associate for this shape.
#pragma clang fp contract(off)
// sum_j a[j] * b[j]
BQ_HOST_SIMD static double as_dot4(const double *__restrict__ a, const double *__restrict__ b, int n) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int j = 0;
    for (; j + 8 <= n; j += 8)
        for (int l = 0; l < 8; ++l) acc[l] += a[j + l] * b[j + l];
    for (int l = 0; j < n; ++j, ++l) acc[l] += a[j] * b[j];
    return ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
}
// sum_j |a[j] * b[j]|
BQ_HOST_SIMD static double as_absdot(const double *__restrict__ a, const double *__restrict__ b, int n) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int j = 0;
    for (; j + 8 <= n; j += 8)
        for (int l = 0; l < 8; ++l) acc[l] += std::fabs(a[j + l] * b[j + l]);
    for (int l = 0; j < n; ++j, ++l) acc[l] += std::fabs(a[j] * b[j]);
    return ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
}
// y[0:n) -= l[0:n) * w
BQ_HOST_SIMD static void as_axpy_neg(double *__restrict__ y, const double *__restrict__ l, double w, int n) {
#pragma clang loop vectorize(enable) interleave_count(4)
    for (int j = 0; j < n; ++j) {
        const double prod = l[j] * w;
        y[j] -= prod;
    }
}
#pragma clang fp contract(on)

// C = L D L' of the m x m Schur complement, kept on the host and grown by one row per new slot (O(m^2)); C is symmetric
// quasi-definite — minus a positive definite block for the pinned variables, a positive definite one for the freed —
// so it factorises without pivoting in any order.  as_ldl_solve checks the residual; on failure the caller falls back
// to the pivoted elimination below (and from there to a fresh base factor).
static bool as_ldl_extend(as_schur *c, int m) {
    const size_t ld = AS_SCHUR_MAX;
    std::vector<double> z;
    c->rows_extended += m - c->ldl_n;
    c->rows_solved += m;
    for (int k = c->ldl_n; k < m; ++k) {
        double *lk = &c->Lc[(size_t)k * ld];
        // L z = C[0:k, k]  (forward), l = z / D, d = C[k][k] - sum l z
        z.assign((size_t)k + 1, 0.0);
        const double *ck = &c->C[(size_t)k * ld];   // row k = column k (symmetric): contiguous
        for (int i = 0; i < k; ++i) {
            const double v = ck[i] - as_dot4(&c->Lc[(size_t)i * ld], z.data(), i);
            z[i] = v;
            lk[i] = v / c->Dc[i];
        }
        const double d = c->C[(size_t)k * ld + k] - as_dot4(lk, z.data(), k);
        if (!std::isfinite(d) || d == 0.0) return false;
        c->Dc[k] = d;
        lk[k] = 1.0;
        c->ldl_n = k + 1;
    }
    return true;
}

double ldl_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool as_ldl_solve(as_schur *c, int m, const double *t, double *w) {
    const double t0 = c->timing ? ldl_now() : 0.0;
    if (!as_ldl_extend(c, m)) return false;
    const double t1 = c->timing ? ldl_now() : 0.0;
    const size_t ld = AS_SCHUR_MAX;
    std::vector<double> y(t, t + m);
    for (int i = 0; i < m; ++i) y[i] -= as_dot4(&c->Lc[(size_t)i * ld], y.data(), i);   // L y = t, rows of L contiguous
    for (int i = 0; i < m; ++i) y[i] /= c->Dc[i];
    const double t2 = c->timing ? ldl_now() : 0.0;
    // L' w = y by columns of L' = rows of L: once w[i] is final it is eliminated from the unknowns above it (contiguous row i;
    // the dot-product form walked a COLUMN of the 1536-pitch factor per unknown: one cache line per element)
    for (int i = m - 1; i >= 0; --i) {
        const double wi = y[i];
        w[i] = wi;
        as_axpy_neg(y.data(), &c->Lc[(size_t)i * ld], wi, i);
    }
    const double t3 = c->timing ? ldl_now() : 0.0;
    c->t_ldl[0] += t1 - t0;
    c->t_ldl[1] += t2 - t1;
    c->t_ldl[2] += t3 - t2;
    struct tail {
        as_schur *c;
        double t;
        ~tail() {
            if (c->timing) c->t_ldl[3] += ldl_now() - t;
        }
    } tl{c, t3};
    // residual against the stored C
    double worst = 0.0, scale = 0.0;
    for (int i = 0; i < m; ++i) {
        const double *ci = &c->C[(size_t)i * ld];
        const double r = t[i] - as_dot4(ci, w, m);
        if (!std::isfinite(r)) return false;
        worst = std::max(worst, std::fabs(r));
        scale = std::max(scale, std::fabs(t[i]) + as_absdot(ci, w, m));
    }
    return worst <= 1e-11 * scale;
}

// dense m x m solve on the host (partial pivoting); false when a pivot is negligible or the result is not finite
static bool as_small_solve(int m, const std::vector<double> &C, const double *t, double *w) {
    std::vector<double> A((size_t)m * m);
    std::vector<double> b(t, t + m);
    double scale = 0.0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            A[(size_t)i * m + j] = C[(size_t)i * AS_SCHUR_MAX + j];
            scale = std::max(scale, std::fabs(A[(size_t)i * m + j]));
        }
    for (int c = 0; c < m; ++c) {
        int piv = c;
        for (int r = c + 1; r < m; ++r)
            if (std::fabs(A[(size_t)r * m + c]) > std::fabs(A[(size_t)piv * m + c])) piv = r;
        if (!(std::fabs(A[(size_t)piv * m + c]) > 1e-13 * scale)) return false;
        if (piv != c) {
            for (int j = 0; j < m; ++j) std::swap(A[(size_t)piv * m + j], A[(size_t)c * m + j]);
            std::swap(b[piv], b[c]);
        }
        for (int r = c + 1; r < m; ++r) {
            const double f = A[(size_t)r * m + c] / A[(size_t)c * m + c];
            if (f == 0.0) continue;
            for (int j = c; j < m; ++j) A[(size_t)r * m + j] -= f * A[(size_t)c * m + j];
            b[r] -= f * b[c];
        }
    }
    for (int r = m - 1; r >= 0; --r) {
        double v = b[r];
        for (int j = r + 1; j < m; ++j) v -= A[(size_t)r * m + j] * w[j];
        w[r] = v / A[(size_t)r * m + r];
        if (!std::isfinite(w[r])) return false;
    }
    return true;
}

// ---- factor re-use: host side --------------------------------------------------------------------------------
// Smallest free set that goes through the kept factor.  Through round 4's first half this was 1024: below it every iteration
// factorised Q[A,A] afresh — a small factorisation, but with it the bound product Q z (a whole panel product) and a blocking look
// per iteration: 3 921 of the 22 897 iterations of BASELINE config 2's shape, 0.67 ms each.  Measured to 'optimal' at n = 20 000
// (profiles/r04/as_schur_min_sweep.txt): 1024 17.5 s, 256 16.5 s, 64 16.3 s, 16 16.1 s, 0 16.1 s — any non-empty free set now.
int as_schur_min() {   // read per iteration: tests switch it between solves
    return (int)bq_hook_value("as_schur_min", 1.0);
}
bool as_schur_enabled() {
    return bq_hook_on("as_schur");
}

#define AS_PANEL_ARGS(T) p->structure, (const T *)p->panel, p->ld, p->symmetric ? 1 : 0, p->n, p->sgn, p->diag_add

void as_schur_free(as_ws *w) {
    as_schur *c = w->sch;
    if (!c) return;
    for (void *ptr : {(void *)c->idx0, (void *)c->pos0, (void *)c->U, (void *)c->W, (void *)c->y0, (void *)c->y,
                      (void *)c->small, (void *)c->meta})
        if (ptr) hipFree(ptr);
    for (void *ptr : {(void *)c->meta_pin, (void *)c->small_pin, (void *)c->coef_pin})
        if (ptr) hipHostFree(ptr);
    delete c;
    w->sch = nullptr;
}

static int as_schur_setup(bq_solver *s, as_ws *w) {
    if (w->sch) return BQ_OK;
    as_schur *c = new as_schur();
    w->sch = c;
    c->timing = w->timing;
    c->cap = s->chol->cap;
    BQ_HIP(hipMalloc(&c->idx0, sizeof(int) * (s->N + 1)));
    BQ_HIP(hipMalloc(&c->pos0, sizeof(int) * (s->N + 1)));
    BQ_HIP(hipMalloc(&c->U, sizeof(double) * AS_SCHUR_MAX * c->cap));
    BQ_HIP(hipMalloc(&c->W, sizeof(double) * AS_SCHUR_MAX * c->cap));
    BQ_HIP(hipMalloc(&c->y0, sizeof(double) * c->cap));
    BQ_HIP(hipMalloc(&c->y, sizeof(double) * c->cap));
    BQ_HIP(hipMalloc(&c->small, sizeof(double) * 3 * AS_SCHUR_MAX));   // dots of a new column | coefficients | dots with y0
    BQ_HIP(hipMalloc(&c->meta, sizeof(int) * 2 * AS_SCHUR_MAX));
    BQ_HIP(hipHostMalloc(&c->meta_pin, sizeof(int) * 2 * AS_SCHUR_MAX, AS_MAPPED));
    BQ_HIP(hipHostMalloc(&c->small_pin, sizeof(double) * 2 * AS_SCHUR_MAX, AS_MAPPED));
    BQ_HIP(hipHostMalloc(&c->coef_pin, sizeof(double) * AS_SCHUR_MAX, AS_MAPPED));
    memset(c->meta_pin, 0, sizeof(int) * 2 * AS_SCHUR_MAX);
    c->meta_pin_d = as_dev(c->meta_pin);
    c->small_pin_d = as_dev(c->small_pin);
    c->coef_pin_d = as_dev(c->coef_pin);
    c->hpos0.assign((size_t)s->N, -1);
    c->C.assign((size_t)AS_SCHUR_MAX * AS_SCHUR_MAX, 0.0);
    c->Lc.assign((size_t)AS_SCHUR_MAX * AS_SCHUR_MAX, 0.0);
    c->Dc.assign((size_t)AS_SCHUR_MAX, 0.0);
    return BQ_OK;
}

// base := the current free set (w->idx holds it, compacted at the top of this iteration); *ok = false when its
// factorisation meets a non-positive pivot (the classic path then takes the reference's minres branch)
static int as_schur_refresh(bq_solver *s, as_ws *w, int64_t nA, bool *ok) {
    as_schur *c = w->sch;
    bq_chol_ws *ws = s->chol;
    hipStream_t st = s->p->ctx->stream;
    const int64_t N = s->N;
    c->valid = false;
    std::vector<int> hidx((size_t)nA);
    BQ_HIP(hipMemcpyAsync(c->idx0, w->idx, sizeof(int) * nA, hipMemcpyDeviceToDevice, st));
    BQ_HIP(hipMemcpyAsync(hidx.data(), w->idx, sizeof(int) * nA, hipMemcpyDeviceToHost, st));
    as_schur_pos_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(N, nA, c->idx0, c->pos0, 0);
    as_schur_pos_kernel<<<(unsigned)((nA + 255) / 256), 256, 0, st>>>(N, nA, c->idx0, c->pos0, 1);
    int64_t np0 = 0;
    BQ_TRY(bq_chol_build_h(ws, s->p, c->idx0, nA, nullptr, &np0));
    BQ_TRY(bq_chol_factor(ws, np0));
    BQ_HIP(hipMemcpyAsync(w->host_info, ws->info, sizeof(int), hipMemcpyDeviceToHost, st));
    BQ_SYNC(s->p->ctx);
    const int info = w->host_info[0];
    if (info != 0) {
        *ok = false;
        return BQ_OK;
    }
    // this factor is kept for up to hundreds of iterations: make its sweeps short chains of full-chip products
    BQ_TRY(bq_chol_prepare_sweeps(ws, np0));
    std::fill(c->hpos0.begin(), c->hpos0.end(), -1);
    for (int64_t a = 0; a < nA; ++a) c->hpos0[(size_t)hidx[(size_t)a]] = (int)a;
    c->n0 = nA;
    c->np0 = np0;
    c->kind.clear();
    c->var.clear();
    c->ldl_n = 0;
    c->valid = true;
    c->y0_valid = false;
    c->refreshes += 1;
    *ok = true;
    return BQ_OK;
}

// drop slot j (swap with the last one): metadata, its row / column of C, its two device columns
static int as_schur_drop(bq_solver *s, as_schur *c, int j) {
    const int last = (int)c->kind.size() - 1;
    if (j != last) {
        hipStream_t st = s->p->ctx->stream;
        c->kind[j] = c->kind[last];
        c->var[j] = c->var[last];
        for (int i = 0; i <= last; ++i) c->C[(size_t)j * AS_SCHUR_MAX + i] = c->C[(size_t)last * AS_SCHUR_MAX + i];
        for (int i = 0; i <= last; ++i) c->C[(size_t)i * AS_SCHUR_MAX + j] = c->C[(size_t)i * AS_SCHUR_MAX + last];
        c->C[(size_t)j * AS_SCHUR_MAX + j] = c->C[(size_t)last * AS_SCHUR_MAX + last];
        BQ_HIP(hipMemcpyAsync(c->U + (int64_t)j * c->cap, c->U + (int64_t)last * c->cap, sizeof(double) * c->np0,
                              hipMemcpyDeviceToDevice, st));
        BQ_HIP(hipMemcpyAsync(c->W + (int64_t)j * c->cap, c->W + (int64_t)last * c->cap, sizeof(double) * c->np0,
                              hipMemcpyDeviceToDevice, st));
    }
    c->kind.pop_back();
    c->var.pop_back();
    c->drops += 1;
    c->ldl_n = std::min(c->ldl_n, j);   // rows < j of the small factorisation only know C[0:j, 0:j], which the swap left alone
    return BQ_OK;
}

// what the previous iteration did to the free set -> slots.  *computed = slots whose columns exist (the new ones are
// appended behind them); *ok = false when the change cannot be carried (too many indices at once)
static int as_schur_event(bq_solver *s, as_ws *w, int *computed, bool *ok) {
    as_schur *c = w->sch;
    const int64_t N = s->N;
    std::vector<int> freed, bound;
    if (w->last_branch == 1) {
        const int hl = w->host_ints[3], hu = w->host_ints[4];
        if (hl < N)
            freed.push_back(hl);
        else if (hu < N)
            freed.push_back(hu);
    } else if (w->last_branch == 0) {
        const int cnt = w->host_ints[8];
        if (cnt > 16) {
            *ok = false;
            return BQ_OK;
        }
        for (int k = 0; k < cnt; ++k) bound.push_back(w->host_ints[9 + k]);
    }
    auto find = [&](int kind, int v) {
        for (size_t j = 0; j < c->kind.size(); ++j)
            if (c->kind[j] == kind && c->var[j] == v) return (int)j;
        return -1;
    };
    // removals of slots first (everything still in the list has its columns), then the new slots at the end
    std::vector<std::pair<int, int>> add;
    for (int v : freed) {
        const int j = find(0, v);
        if (j >= 0) {
            BQ_TRY(as_schur_drop(s, c, j));
        } else {
            add.push_back({1, v});
            c->y0_valid = false;   // a variable outside the base left its bound: z, and with it b0, moves
        }
    }
    for (int v : bound) {
        const int j = find(1, v);
        if (j >= 0) {
            BQ_TRY(as_schur_drop(s, c, j));
            c->y0_valid = false;
        } else {
            add.push_back({0, v});
        }
    }
    *computed = (int)c->kind.size();
    for (auto &kv : add) {
        if (kv.first == 0 && c->hpos0[(size_t)kv.second] < 0) {   // cannot happen: a variable that reached a bound was free
            *ok = false;
            return BQ_OK;
        }
        c->kind.push_back(kv.first);
        c->var.push_back(kv.second);
    }
    *ok = (int)c->kind.size() <= as_schur_limit(c->np0);
    return BQ_OK;
}

template <typename T>
static int as_schur_solve_t(bq_solver *s, as_ws *w, int computed, bool *good) {
    as_schur *c = w->sch;
    bq_chol_ws *ws = s->chol;
    bq_problem *p = s->p;
    hipStream_t st = p->ctx->stream;
    const int64_t N = s->N, np0 = c->np0, n0 = c->n0;
    const int m = (int)c->kind.size();
    const unsigned gb = (unsigned)((np0 + 255) / 256);
    *good = false;
    int *hmeta = c->meta_pin;   // the previous iteration's copy is long complete (every pass through here ends in a synchronisation)
    for (int k = 0; k < m; ++k) {
        hmeta[k] = c->kind[k];
        hmeta[AS_SCHUR_MAX + k] = c->var[k];
    }
    const bool mbx = w->mailbox;
    const int *meta = mbx ? c->meta_pin_d : c->meta;
    if (!mbx) BQ_HIP(hipMemcpyAsync(c->meta, hmeta, sizeof(int) * 2 * AS_SCHUR_MAX, hipMemcpyHostToDevice, st));
    // where the dot products go and how the host learns that they are there
    double *dots_out = mbx ? c->small_pin_d : c->small, *t_out = mbx ? c->small_pin_d + AS_SCHUR_MAX : c->small + 2 * AS_SCHUR_MAX;
    unsigned int *tk = mbx ? w->mail_ticket : nullptr;
    int *post = mbx ? w->mail_d + 1 : nullptr;
    if (!c->y0_valid) {   // while only base variables reach bounds, b0 and Q00^-1 b0 stay what they were
        as_schur_z_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, c->pos0, w->z);
        BQ_TRY(bq_problem_apply(p, w->z, w->Qz, nullptr));
        as_schur_rhs0_kernel<<<gb, 256, 0, st>>>(n0, np0, c->idx0, p->q, w->Qz, ws->rhs);
        BQ_TRY(bq_chol_solve(ws, np0));
        BQ_HIP(hipMemcpyAsync(c->y0, ws->rhs, sizeof(double) * np0, hipMemcpyDeviceToDevice, st));
        c->y0_valid = true;
    }
    double *host_small = c->small_pin, *host_t = c->small_pin + AS_SCHUR_MAX;
    bool have_t = false;
    for (int k = computed; k < m; ++k) {   // the columns of the new slots and their rows of C
        double *uk = c->U + (int64_t)k * c->cap, *wk = c->W + (int64_t)k * c->cap;
        // the column goes to its slot AND to the right-hand side of the solve; the solve leaves its result in the slot of W too
        as_schur_col_kernel<T><<<gb, 256, 0, st>>>(c->kind[k], c->var[k], c->kind[k] == 0 ? c->hpos0[(size_t)c->var[k]] : -1, n0,
                                                  np0, c->idx0, AS_PANEL_ARGS(T), uk, ws->rhs);
        // a pinned base variable's column is a unit vector: the forward sweep starts at its row
        BQ_TRY(bq_chol_solve(ws, np0, c->kind[k] == 0 ? (int64_t)c->hpos0[(size_t)c->var[k]] : 0, wk));
        if (k == m - 1) {   // the right-hand side of the small system needs nothing from the host: same pass over U, same round trip
            as_schur_dots_kernel<T><<<m, 256, 0, st>>>(2, k, c->U, wk, c->y0, c->cap, np0, meta, AS_PANEL_ARGS(T), s->mU, s->lb,
                                                      s->ub, p->q, w->Qz, dots_out, t_out, tk, post, ++w->mail_seq[1]);
            if (!mbx) {
                BQ_HIP(hipMemcpyAsync(host_small, c->small, sizeof(double) * (k + 1), hipMemcpyDeviceToHost, st));
                BQ_HIP(hipMemcpyAsync(host_t, c->small + 2 * AS_SCHUR_MAX, sizeof(double) * m, hipMemcpyDeviceToHost, st));
            }
            have_t = true;
        } else {
            as_schur_dots_kernel<T><<<k + 1, 256, 0, st>>>(0, k, c->U, wk, nullptr, c->cap, np0, meta, AS_PANEL_ARGS(T), s->mU,
                                                          s->lb, s->ub, p->q, w->Qz, dots_out, nullptr, tk, post, ++w->mail_seq[1]);
            if (!mbx) BQ_HIP(hipMemcpyAsync(host_small, c->small, sizeof(double) * (k + 1), hipMemcpyDeviceToHost, st));
        }
        as_tick(w, 1);
        BQ_TRY(as_look(s->p->ctx, w, 1));
        as_tick(w, 2);
        const double tc0 = w->timing ? ldl_now() : 0.0;
        for (int i = 0; i <= k; ++i) {
            if (!std::isfinite(host_small[i])) return BQ_OK;
            c->C[(size_t)i * AS_SCHUR_MAX + k] = host_small[i];
            c->C[(size_t)k * AS_SCHUR_MAX + i] = host_small[i];
        }
        if (w->timing) c->t_c += ldl_now() - tc0;
    }
    double *coef = c->coef_pin;
    if (m > 0) {
        if (!have_t) {
            as_schur_dots_kernel<T><<<m, 256, 0, st>>>(1, 0, c->U, c->y0, nullptr, c->cap, np0, meta, AS_PANEL_ARGS(T), s->mU, s->lb,
                                                      s->ub, p->q, w->Qz, t_out, nullptr, tk, post, ++w->mail_seq[1]);
            if (!mbx) BQ_HIP(hipMemcpyAsync(host_t, c->small + 2 * AS_SCHUR_MAX, sizeof(double) * m, hipMemcpyDeviceToHost, st));
            BQ_TRY(as_look(s->p->ctx, w, 1));
        }
        if (!as_ldl_solve(c, m, host_t, coef)) {   // incremental factorisation first, pivoted elimination as the fallback
            c->ldl_n = 0;
            if (!as_small_solve(m, c->C, host_t, coef)) return BQ_OK;
        }
        if (!mbx) BQ_HIP(hipMemcpyAsync(c->small + AS_SCHUR_MAX, coef, sizeof(double) * m, hipMemcpyHostToDevice, st));
    }
    as_tick(w, 3);
    const double *coef_dev = mbx ? c->coef_pin_d : c->small + AS_SCHUR_MAX;
    as_schur_combine_kernel<<<gb, 1024, 0, st>>>(np0, m, c->y0, c->W, c->cap, coef_dev, c->y);
    as_cand_fill_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, s->lb, s->ub, w->cand, w->ints);
    {
        const int64_t span = n0 > m ? (n0 > 0 ? n0 : 1) : (int64_t)m;
        as_cand_scatter_kernel<<<dim3((unsigned)((span + 255) / 256)), 256, 0, st>>>(n0, m, c->idx0, meta, s->mL, s->mU, s->lb, s->ub,
                                                                                  c->y, coef_dev, w->cand, w->ints,
                                                                                  mbx ? w->mail_ticket + 1 : nullptr, w->host_ints_d,
                                                                                  mbx ? w->mail_d + 2 : nullptr, ++w->mail_seq[2]);
    }
    BQ_HIP(hipGetLastError());
    if (!mbx) BQ_HIP(hipMemcpyAsync(w->host_ints, w->ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    as_tick(w, 4);
    BQ_TRY(as_look(s->p->ctx, w, 2));
    as_tick(w, 5);
    *good = true;
    return BQ_OK;
}

// one restricted solve through the kept factor; *solved = false leaves the iteration to the classic path
int as_schur_step(bq_solver *s, as_ws *w, int64_t nA, bool *solved) {
    *solved = false;
    BQ_TRY(as_schur_setup(s, w));
    as_schur *c = w->sch;
    int computed = 0;
    bool ok = c->valid;
    if (ok) BQ_TRY(as_schur_event(s, w, &computed, &ok));
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!ok) {
            BQ_TRY(as_schur_refresh(s, w, nA, &ok));
            if (!ok) return BQ_OK;   // not positive definite: classic path (minres branch)
            computed = 0;
        } else if (attempt == 0) {
            c->reused += 1;
        }
        bool good = false;
        if (s->p->storage == BQ_F64)
            BQ_TRY(as_schur_solve_t<double>(s, w, computed, &good));
        else
            BQ_TRY(as_schur_solve_t<float>(s, w, computed, &good));
        if (good) {
            *solved = true;
            return BQ_OK;
        }
        ok = false;   // numerically singular update: start again from a fresh factor of the current set
    }
    c->valid = false;
    return BQ_OK;
}
