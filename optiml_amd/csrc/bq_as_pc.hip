// The preconditioner of ActiveSetCG's inner conjugate gradients: P = D + Phi Phi' on the free set through Woodbury (struct as_pc in
// bq_as.h).  Split out of bq_as.hip in round 5.
#include "bq_as.h"

// ---------------------------------------------------------------------------------------------------------------
// the diagonal + low-rank preconditioner of the inner iteration (struct as_pc)
// ---------------------------------------------------------------------------------------------------------------
constexpr int PC_T = 32, PC_C = 64, PC_SLICES = 8;

// class statistics of the samples (BQ_SVC panels): cls[k] = a_k = (mean_+ - mean_-)_k / 2, cls[d + k] = m0_k = (mean_+ + mean_-)_k / 2,
// cls[2d] = |a|.  One workgroup per feature column for the sums, the last one to finish closes.
__global__ __launch_bounds__(256) void as_pc_class_kernel(int64_t n, int64_t d, const double *__restrict__ X,
                                                          const double *__restrict__ sgn, double *__restrict__ cls,
                                                          unsigned int *ticket) {
    __shared__ double sh[4];
    const int64_t k = blockIdx.x;
    double sp = 0.0, sm = 0.0, cp = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double v = X[i * d + k];
        if (sgn[i] > 0.0) {
            sp += v;
            cp += 1.0;
        } else {
            sm += v;
        }
    }
    sp = as_block_sum(sp, sh);
    sm = as_block_sum(sm, sh);
    cp = as_block_sum(cp, sh);
    if (threadIdx.x == 0) {
        const double cm = (double)n - cp;
        const double mp = cp > 0.0 ? sp / cp : 0.0, mm = cm > 0.0 ? sm / cm : 0.0;
        cls[k] = (cp > 0.0 && cm > 0.0) ? 0.5 * (mp - mm) : 0.0;
        cls[d + k] = (cp > 0.0 && cm > 0.0) ? 0.5 * (mp + mm) : (cp > 0.0 ? mp : mm);
    }
    if (as_last_block(ticket)) {
        double s2 = 0.0;
        for (int64_t j = threadIdx.x; j < d; j += 256) s2 += cls[j] * cls[j];
        s2 = as_block_sum(s2, sh);
        if (threadIdx.x == 0) {
            cls[2 * d] = sqrt(s2);
            *ticket = 0;
        }
    }
}

// Phi (feature-major).  One thread per sample.  RBF:
//   family 1 (d + 1 columns): y e^{-g|x|^2} [1, sqrt(2g) x]                    the order-0/1 terms of e^{2g x.x'}
//   family 2, BQ_SVC panels — the directions of the ORDER-2 term (2g x.x')^2 / 2 = 2g^2 <x x', x' x''> whose eigenvalues grow like
//   n |class mean|^2 (the rest of that term is a flat bulk of d (d + 1) / 2 directions no low-rank model captures:
//   tools/pc_nystrom_study.py, tools/pc_order2_cpu_study.py):
//     fam2 == 2 (2d columns, round 5): the EXACT projection of the order-2 feature sqrt(2) g y e vec(x x') onto the span of
//       U_{c,k} = m_c e_k' + e_k m_c' (c = the two class means, k < d) — raw coordinates f_{c,k} = <x x', U_{c,k}> = 2 (x.m_c) x_k
//       (as_pc_raw_kernel), orthonormalised by the inverse Cholesky factor of the 2d x 2d Gram matrix of the U's
//       (as_pc_project_kernel).  A projection of a positive semi-definite term: P = D + Phi Phi' never over-counts Q.
//     fam2 == 1 (d columns, rounds 3-4; kept for d too large for 3d + 2 features): 2g |a| e (x - m0 - y a), the cross term
//       2 (y y' |a|^2)(e.e') under the assumption m0 ~ 0 — not a projection (it over-counts when m0 is not small): measured against
//       fam2 == 2 at d = 64, n = 20 000: 39 against 23 conjugate-gradient iterations to 1e-8 (profiles/r05/pc_projected_study.txt)
// share[i] <- Q_ii and, for fam2 != 2, the sum of squares of the stored features in s1[i]; as_pc_diag_kernel turns them into 1 / D.
__global__ void as_pc_features_kernel(int kernel, int64_t n, int64_t d, int64_t ld, const double *__restrict__ X,
                                      const double *__restrict__ sgn, const double *__restrict__ cls, int fam2, double gamma,
                                      int add_one, double diag_add, int m, float *__restrict__ Phi, double *__restrict__ qdiag,
                                      double *__restrict__ raw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ld) return;
    if (i >= n) {
        for (int j = 0; j < m; ++j) Phi[(int64_t)j * ld + i] = 0.f;
        qdiag[i] = 0.0;
        if (raw != nullptr) raw[i] = raw[ld + i] = raw[2 * ld + i] = 0.0;
        return;
    }
    const double y = sgn ? sgn[i] : 1.0;
    const double *x = X + i * d;
    double sq = 0.0;
    for (int64_t k = 0; k < d; ++k) sq = fma(x[k], x[k], sq);
    double qii;
    int col = 0;
    auto put = [&](int64_t j, double v) { Phi[j * ld + i] = (float)v; };   // D is what the STORED (fp32) values leave of the diagonal
    if (kernel == BQ_KERNEL_RBF) {
        const double e = exp(-gamma * sq);
        const double c0 = y * e, c1 = c0 * sqrt(2.0 * gamma);
        put(0, c0);
        for (int64_t k = 0; k < d; ++k) put(1 + k, c1 * x[k]);
        col = (int)d + 1;
        if (fam2 == 1) {
            const double c2 = 2.0 * gamma * cls[2 * d] * e;
            for (int64_t k = 0; k < d; ++k) put(col + k, c2 * (x[k] - cls[d + k] - y * cls[k]));
            col += (int)d;
        } else if (fam2 == 2) {   // the columns are written by as_pc_project_kernel from these three per-sample numbers
            double sp = 0.0, sm = 0.0;
            for (int64_t k = 0; k < d; ++k) {
                sp = fma(x[k], cls[d + k] + cls[k], sp);   // x . m_+,  m_+ = m0 + a
                sm = fma(x[k], cls[d + k] - cls[k], sm);   // x . m_-,  m_- = m0 - a
            }
            raw[i] = 2.0 * sp;
            raw[ld + i] = 2.0 * sm;
            raw[2 * ld + i] = sqrt(2.0) * gamma * c0;      // sqrt(2 g^2) y e
            col += 2 * (int)d;
        }
        qii = 1.0;
    } else {   // linear: exact features (up to their fp32 rounding)
        for (int64_t k = 0; k < d; ++k) put(k, y * x[k]);
        col = (int)d;
        qii = sq;
    }
    if (add_one) {
        put(col, y);
        qii += 1.0;
    }
    qdiag[i] = qii + diag_add;
}

// family 2, fam2 == 2: Phi[col0 + j][i] = scale_i * sum_{l <= j} F[i][l] Rinv[l][j],  F[i][c d + k] = (2 x_i.m_c) x_ik — a
// (samples x 2d) x (2d x 2d upper triangular) product, 64 x 64 output tiles, 4 x 4 per thread, fp64 accumulation, once per solver.
__global__ __launch_bounds__(256) void as_pc_project_kernel(int64_t n, int64_t d, int64_t ld, const double *__restrict__ X,
                                                            const double *__restrict__ raw, const double *__restrict__ Rinv,
                                                            int col0, float *__restrict__ Phi) {
    __shared__ double As[16][65], Bs[16][65];
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int j0 = (int)blockIdx.y * 64, n2 = 2 * (int)d;
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    double acc[4][4] = {};
    const int kend = j0 + 64 < n2 ? j0 + 64 : n2;   // Rinv is upper triangular: rows beyond the tile's last column are zero
    for (int k0 = 0; k0 < kend; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = t + q * 256, kk = e & 15, ii = e >> 4;   // consecutive threads: consecutive k of one sample (row-major X)
            const int64_t i = i0 + ii;
            const int l = k0 + kk;
            double v = 0.0;
            if (i < n && l < n2) v = raw[(l >= d ? ld : 0) + i] * X[i * d + (l >= d ? l - d : l)];
            As[kk][ii] = v;
            const int jj = e & 63, kb = e >> 6;
            Bs[kb][jj] = (k0 + kb < n2 && j0 + jj < n2) ? Rinv[(int64_t)(k0 + kb) * n2 + j0 + jj] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = As[kk][ty * 4 + r];
                b[r] = Bs[kk][tx * 4 + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fma(a[r], b[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + ty * 4 + r;
        if (i >= n) continue;
        const double sc = raw[2 * ld + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = j0 + tx * 4 + c;
            if (j < n2) Phi[(int64_t)(col0 + j) * ld + i] = (float)(sc * acc[r][c]);
        }
    }
}

// 1 / D_i and the share of the diagonal the features leave, D_i = Q_ii - |Phi_i|^2 over the STORED features (floored at 1e-8 Q_ii: P
// only has to be positive definite).  qdiag_share: Q_ii in, share out.
// With the implicit order-2 remainder R = B - Phi_top Phi_top' (bq_as_pc2.hip) the model is P = D + Phi Phi' + R: D also gives up
// R's diagonal, bdiag_i - |Phi_top,i|^2 (bdiag: the diagonal of B; top0 / ntop: the Phi_top columns; null / 0: no remainder).
__global__ void as_pc_diag_kernel(int64_t n, int64_t ld, int m, const float *__restrict__ Phi, double *__restrict__ dinv,
                                  double *__restrict__ qdiag_share, const double *__restrict__ bdiag, int top0, int ntop) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ld) return;
    if (i >= n) {
        dinv[i] = 0.0;
        qdiag_share[i] = 1.0;
        return;
    }
    double s = 0.0;
    for (int j = 0; j < m; ++j) {
        if (bdiag != nullptr && j >= top0 && j < top0 + ntop) continue;   // Phi_top Phi_top' + R = B on these directions
        const double v = (double)Phi[(int64_t)j * ld + i];
        s = fma(v, v, s);
    }
    if (bdiag != nullptr) s += bdiag[i];
    const double qii = qdiag_share[i];
    dinv[i] = 1.0 / fmax(qii - s, 1e-8 * qii);
    qdiag_share[i] = (qii - s) / qii;   // what the features leave of the diagonal: the host refuses a model that leaves too little
}

// Gpart[slice][a][b] = sum over the slice's FREE samples of Phi[a][i] Phi[b][i] / D_i, lower tiles (b-tile <= a-tile)
__global__ __launch_bounds__(256) void as_pc_gram_kernel(int m, int64_t mp, int64_t N, int64_t ld, const float *__restrict__ Phi,
                                                         const double *__restrict__ dinv, const unsigned char *__restrict__ mL,
                                                         const unsigned char *__restrict__ mU, double *__restrict__ Gpart) {
    __shared__ double As[PC_T][PC_C + 1], Bs[PC_T][PC_C + 1];
    // (ta, tb) from the linear lower-triangle tile index
    int ta = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((ta + 1) * (ta + 2) / 2 <= (int)blockIdx.x) ++ta;
    while (ta * (ta + 1) / 2 > (int)blockIdx.x) --ta;
    const int tb = (int)blockIdx.x - ta * (ta + 1) / 2;
    const int a0 = ta * PC_T, b0 = tb * PC_T;
    const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
    const int64_t per = ((N + PC_SLICES - 1) / PC_SLICES + PC_C - 1) / PC_C * PC_C;
    const int64_t i0 = (int64_t)blockIdx.y * per, i1 = i0 + per < N ? i0 + per : N;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t c = i0; c < i1; c += PC_C) {
#pragma unroll
        for (int k = 0; k < (PC_T * PC_C) / 256; ++k) {
            const int e = t + k * 256, row = e / PC_C, cc = e % PC_C;
            const int64_t i = c + cc;
            double wgt = 0.0;
            if (i < i1 && !(mL[i] | mU[i])) wgt = dinv[i];
            As[row][cc] = (a0 + row < m && wgt != 0.0) ? (double)Phi[(int64_t)(a0 + row) * ld + i] * wgt : 0.0;
            Bs[row][cc] = (b0 + row < m && i < i1) ? (double)Phi[(int64_t)(b0 + row) * ld + i] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < PC_C; ++k) {
            const double bv = Bs[tx][k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fma(As[ty * 4 + j][k], bv, acc[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        Gpart[((int64_t)blockIdx.y * mp + a0 + ty * 4 + j) * mp + b0 + tx] = acc[j];
}

// full rebuild: H = G = I + the slices of Gpart added in slice order (lower triangle; identity on the pad rows)
__global__ void as_pc_gram_reduce_kernel(int m, int64_t mp, const double *__restrict__ Gpart, double *__restrict__ H, int64_t ldh) {
    const int64_t a = blockIdx.y, b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b > a || b >= mp) return;
    double v = (a == b) ? 1.0 : 0.0;
    if (a < m)
        for (int sidx = 0; sidx < PC_SLICES; ++sidx) v += Gpart[((int64_t)sidx * mp + a) * mp + b];
    H[a * ldh + b] = v;
}

// G^-1 = L^-T L^-1 from the explicit inverse factor bq_chol_prepare_sweeps leaves (MT = L^-T, upper triangular, pitch 1024):
// Ginv[a][b] = sum_k MT[a][k] MT[b][k] — 16 x 16 tiles through LDS, k ascending: one fixed order
__global__ __launch_bounds__(256) void as_pc_ginv_kernel(int64_t mp, const double *__restrict__ MT, int64_t ldm, double *__restrict__ Ginv) {
    __shared__ double As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t a0 = (int64_t)blockIdx.y * 16, b0 = (int64_t)blockIdx.x * 16;
    double acc = 0.0;
    const int64_t kstart = (a0 > b0 ? a0 : b0);   // MT[a][k] = 0 for k < a
    for (int64_t k0 = kstart; k0 < mp; k0 += 16) {
        As[ty][tx] = MT[(a0 + ty) * ldm + k0 + tx];
        Bs[ty][tx] = MT[(b0 + ty) * ldm + k0 + tx];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(As[ty][k], Bs[tx][k], acc);
        __syncthreads();
    }
    Ginv[(a0 + ty) * mp + b0 + tx] = acc;
}

// u = Ginv t: a wave per row, lanes along the columns, xor butterfly (every lane ends with the sum)
__global__ __launch_bounds__(256) void as_pc_gemv_kernel(int64_t mp, const double *__restrict__ Ginv, const double *__restrict__ t,
                                                         double *__restrict__ u, const as_cg_scal *cg) {
    if (cg->done) return;
    const int lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= mp) return;
    const double *row = Ginv + a * mp;
    double acc = 0.0;
    for (int64_t b = lane; b < mp; b += 64) acc = fma(row[b], t[b], acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) u[a] = acc;
}

constexpr int PC_MAX_CHG = 64;
// which samples entered / left the free set since G was last brought up to date, in index order (the order of the rank-one
// updates must not depend on the launch geometry: as_list_count / as_list_positions); more than PC_MAX_CHG of them, or `force`:
// full rebuild
__global__ __launch_bounds__(256) void as_pc_diff_count_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                               const unsigned char *__restrict__ mU,
                                                               const unsigned char *__restrict__ prev, int *__restrict__ cnt,
                                                               unsigned int *ticket, int *__restrict__ chg, int force) {
    int bits = 0;
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        if (i < N && (int)!(mL[i] | mU[i]) != (int)prev[i]) bits |= 1 << j;
    }
    __shared__ int total;
    if (threadIdx.x == 0) total = -1;
    as_list_count(bits, cnt, ticket, &total);
    if (threadIdx.x == 0 && total >= 0) {
        chg[0] = total;
        chg[1] = (force || total > PC_MAX_CHG) ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void as_pc_diff_write_kernel(int64_t N, const unsigned char *__restrict__ mL,
                                                               const unsigned char *__restrict__ mU, unsigned char *__restrict__ prev,
                                                               const int *__restrict__ cnt, int *__restrict__ chg) {
    int bits = 0, fr[BQ_VEC_ITEMS];
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j) {
        const int64_t i = (int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x;
        fr[j] = 0;
        if (i < N) {
            fr[j] = !(mL[i] | mU[i]);
            if (fr[j] != (int)prev[i]) bits |= 1 << j;
            prev[i] = (unsigned char)fr[j];
        }
    }
    int pos[BQ_VEC_ITEMS];
    as_list_positions(bits, cnt, pos);
#pragma unroll
    for (int j = 0; j < BQ_VEC_ITEMS; ++j)
        if (((bits >> j) & 1) && pos[j] < PC_MAX_CHG) {
            chg[2 + 2 * pos[j]] = (int)((int64_t)blockIdx.x * BQ_VEC_TILE + (int64_t)j * BQ_VEC_BLOCK + threadIdx.x);
            chg[3 + 2 * pos[j]] = fr[j] ? 1 : -1;
        }
}

// G -> G + s phi phi' / D for every sample that entered (s = +1) or left (s = -1) the free set, in list order, carried on the
// INVERSE (Sherman-Morrison):  v = Ginv phi,  Ginv -= (s / D) / (1 + (s / D) phi'v) v v'.  One workgroup: the updates are a
// chain, each is two passes over the m x m inverse (3 MB at m = 514: L2), and there are one or two of them per outer iteration
// — against the m^3 / 3 factorisation + explicit inverse of round 3's every outer iteration (0.9 ms at BASELINE config 5).
// G - phi phi'/D stays >= I, so every denominator is positive; one at or below 1e-8 (or not finite) raises *fail and the caller
// sums G afresh.  Fixed order throughout: the same bits on every rank.
__global__ __launch_bounds__(1024) void as_pc_sm_kernel(int m, int64_t mp, int64_t ld, const float *__restrict__ Phi,
                                                        const double *__restrict__ dinv, const int *__restrict__ chg,
                                                        double *__restrict__ Ginv, int *__restrict__ fail) {
    __shared__ double phi[PC_MAX_M], v[PC_MAX_M];
    __shared__ double red[16];
    __shared__ double coef_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int cnt = chg[0];
    for (int c = 0; c < cnt; ++c) {
        const int64_t i = chg[2 + 2 * c];
        const double wgt = (double)chg[3 + 2 * c] * dinv[i];
        for (int j = tid; j < m; j += 1024) phi[j] = (double)Phi[(int64_t)j * ld + i];
        __syncthreads();
        for (int a0 = wv * 2; a0 < m; a0 += 32) {   // two rows per wave and turn: two independent chains
            const int a1 = a0 + 1 < m ? a0 + 1 : a0;
            const double *r0 = Ginv + (int64_t)a0 * mp, *r1 = Ginv + (int64_t)a1 * mp;
            double s0 = 0.0, s1 = 0.0;
            for (int b = lane; b < m; b += 64) {
                s0 = fma(r0[b], phi[b], s0);
                s1 = fma(r1[b], phi[b], s1);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                s0 += __shfl_xor(s0, off, 64);
                s1 += __shfl_xor(s1, off, 64);
            }
            if (lane == 0) {
                v[a0] = s0;
                v[a1] = s1;
            }
        }
        __syncthreads();
        double part = 0.0;
        for (int j = tid; j < m; j += 1024) part = fma(phi[j], v[j], part);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (lane == 0) red[wv] = part;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int k = 0; k < 16; ++k) tot += red[k];
            const double denom = 1.0 + wgt * tot;
            if (!(denom > 1e-8) || !isfinite(denom)) {
                *fail = 1;
                coef_s = 0.0;
            } else {
                coef_s = wgt / denom;
            }
        }
        __syncthreads();
        const double coef = coef_s;
        for (int a = wv; a < m; a += 16) {
            double *row = Ginv + (int64_t)a * mp;
            const double ca = coef * v[a];
            for (int b = lane; b < m; b += 64) row[b] = fma(-ca, v[b], row[b]);
        }
        __syncthreads();
    }
}

constexpr int PC_FG = 8;   // features per butterfly group of the t kernel

// t[j] = sum_i Phi[j][i] r_i / D_i   (r vanishes outside the free set).  One workgroup per block of 1024 samples: a lane keeps
// w = r / D of its four consecutive samples in registers and walks all features (one 16-byte load per feature: a 4 KiB run per
// workgroup and feature row), eight features at a time through a halving butterfly over the 64 lanes (3 + 3 shuffle-adds for
// eight sums), the four wave sums meet in LDS -> tpart[block][j]; the last workgroup to finish adds the blocks in block order.
// Round 3 had one workgroup per FEATURE re-reading r and 1 / D for each of them: 2.5 GB of L2 traffic beside the 1 GB of
// features, 0.40 ms per call at BASELINE config 5.  Fixed order throughout: the same bits on every rank.
// blockIdx.x counts this rank's sample blocks from block0 on (round 6: the pass is sharded by samples, as_pc_part).
__global__ __launch_bounds__(256) void as_pc_tphi_kernel(int m, int64_t m8, int64_t mp, int64_t N, int64_t ld, int64_t block0,
                                                         const float *__restrict__ Phi, const double *__restrict__ dinv,
                                                         const double *__restrict__ r, double *__restrict__ tpart, const as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double wsum[4][PC_MAX_M];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t blk = block0 + blockIdx.x;
    const int64_t base = blk * BQ_VEC_TILE + 4 * tid;   // ld is a multiple of the tile: always in range
    // this workgroup's feature groups: a contiguous quarter of the m8 / 8 groups
    const int64_t ngroups = m8 / PC_FG;
    const int64_t g_lo = ngroups * blockIdx.y / gridDim.y, g_hi = ngroups * (blockIdx.y + 1) / gridDim.y;
    const int64_t j_lo = g_lo * PC_FG, j_hi = g_hi * PC_FG;
    double w4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w4[k] = (base + k < N) ? r[base + k] * dinv[base + k] : 0.0;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    const int rho = (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0);
    for (int64_t j0 = j_lo; j0 < j_hi; j0 += PC_FG) {
        double a[PC_FG];
#pragma unroll
        for (int f = 0; f < PC_FG; ++f) {
            const as_f4 v = *reinterpret_cast<const as_f4 *>(Phi + (j0 + f) * ld + base);
            a[f] = fma((double)v.w, w4[3], fma((double)v.z, w4[2], fma((double)v.y, w4[1], (double)v.x * w4[0])));
        }
        double u[4], t2[2], s1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double send = b5 ? a[i] : a[i + 4];
            const double keep = b5 ? a[i + 4] : a[i];
            u[i] = keep + __shfl_xor(send, 32, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double send = b4 ? u[i] : u[i + 2];
            const double keep = b4 ? u[i + 2] : u[i];
            t2[i] = keep + __shfl_xor(send, 16, 64);
        }
        {
            const double send = b3 ? t2[0] : t2[1];
            const double keep = b3 ? t2[1] : t2[0];
            s1 = keep + __shfl_xor(send, 8, 64);
        }
        s1 += __shfl_xor(s1, 4, 64);
        s1 += __shfl_xor(s1, 2, 64);
        s1 += __shfl_xor(s1, 1, 64);
        if ((lane & 7) == 0) wsum[wv][j0 - j_lo + rho] = s1;
    }
    __syncthreads();
    double *mine = tpart + blk * mp;
    for (int64_t j = j_lo + tid; j < j_hi; j += 256) {
        const int64_t c = j - j_lo;
        mine[j] = j < m ? ((wsum[0][c] + wsum[1][c]) + wsum[2][c]) + wsum[3][c] : 0.0;
    }
}

// The sample blocks' partial sums of t, per canonical sample SEGMENT (blockIdx.y counts this rank's segments): 16 features x 16
// interleaved runs of the segment's blocks per workgroup (a run's loads are independent of each other), the 16 runs then combined
// in run order -> the segment's slot of the gathered buffer.  (Round 5 added ALL blocks in one such pass on every rank; the
// association is per segment now, then over the segments in order: as_pc_tsum_kernel — the same on one rank and on eight.)
__global__ __launch_bounds__(256) void as_pc_tseg_kernel(int64_t m8, int64_t mp, as_pc_part part, const double *__restrict__ tpart,
                                                         double *__restrict__ tg, int64_t gstride, const as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + tx;
    const int k = part.lo + (int)blockIdx.y;
    double acc = 0.0;
    if (j < m8)
        for (int64_t b = part.blk[k] + ty; b < part.blk[k + 1]; b += 16) acc += tpart[b * mp + j];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && j < mp) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += red[q][tx];
        tg[(int64_t)part.slot[k] * gstride + j] = v;
    }
}
// t = the gathered per-segment sums added in segment order (every rank: the same bits)
__global__ __launch_bounds__(256) void as_pc_tsum_kernel(int64_t mp, as_pc_part part, const double *__restrict__ tg, int64_t gstride,
                                                         double *__restrict__ tvec, const as_cg_scal *cg) {
    if (cg->done) return;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= mp) return;
    double a = 0.0;
    for (int k = 0; k < part.S; ++k) a += tg[(int64_t)part.slot[k] * gstride + j];
    tvec[j] = a;
}

// z = P^-1 r on the free set for THIS RANK's sample blocks (block0 on): z_i = (r_i - Phi_i . u) / D_i, u = G^-1 t, written in the
// gathered layout (the slot of the block's segment); as_pc_zunpack_kernel brings every rank's part home and takes r'z.
// A lane owns four consecutive samples (one 16-byte load per feature row), eight feature rows in flight.
__global__ __launch_bounds__(256) void as_pc_apply_kernel(int m, int64_t m8, int64_t N, int64_t ld, int64_t block0, as_pc_part part,
                                                          const float *__restrict__ Phi, const double *__restrict__ dinv,
                                                          const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                                          const double *__restrict__ r, const double *__restrict__ u,
                                                          double *__restrict__ zg, const as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double us[PC_MAX_M];
    for (int j = threadIdx.x; j < m8; j += BQ_VEC_BLOCK) us[j] = j < m ? u[j] : 0.0;
    __syncthreads();
    const int64_t blk = block0 + blockIdx.x;
    int k = part.lo;
    while (k + 1 < part.hi && blk >= part.blk[k + 1]) ++k;
    double *dst = zg + ((int64_t)part.slot[k] * part.maxlen + (blk - part.blk[k])) * BQ_VEC_TILE + 4 * threadIdx.x;
    const int64_t base = blk * BQ_VEC_TILE + 4 * threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t j0 = 0; j0 < m8; j0 += PC_FG) {
        as_f4 v[PC_FG];
#pragma unroll
        for (int f = 0; f < PC_FG; ++f) v[f] = *reinterpret_cast<const as_f4 *>(Phi + (j0 + f) * ld + base);
#pragma unroll
        for (int f = 0; f < PC_FG; ++f) {
            const double uj = us[j0 + f];
            acc[0] = fma((double)v[f].x, uj, acc[0]);
            acc[1] = fma((double)v[f].y, uj, acc[1]);
            acc[2] = fma((double)v[f].z, uj, acc[2]);
            acc[3] = fma((double)v[f].w, uj, acc[3]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + q;
        double zi = 0.0;
        if (i < N && !(mL[i] | mU[i])) zi = dinv[i] * (r[i] - acc[q]);
        dst[q] = zi;
    }
}
// z (contiguous, every sample) from the gathered layout;  fin: rz = r'z and beta = rz / rz_old (first: 0) of the conjugate gradients
// (fin == 0: z only — an inner application of P1^-1 inside the polynomial of the order-2 remainder: as_pc_apply)
__global__ __launch_bounds__(256) void as_pc_zunpack_kernel(int64_t N, as_pc_part part, const double *__restrict__ zg,
                                                            const double *__restrict__ r, double *__restrict__ z, double *pt_sums,
                                                            int64_t nblk, as_cg_scal *cg, int first, int fin) {
    if (cg->done) return;
    __shared__ double sh[4];
    const int64_t blk = blockIdx.x;
    int k = 0;
    while (k + 1 < part.S && blk >= part.blk[k + 1]) ++k;
    const double *src = zg + ((int64_t)part.slot[k] * part.maxlen + (blk - part.blk[k])) * BQ_VEC_TILE + 4 * threadIdx.x;
    const int64_t base = blk * BQ_VEC_TILE + 4 * threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double zi = src[q];
        z[base + q] = zi;
        if (base + q < N) s += __dmul_rn(r[base + q], zi);   // z vanishes outside the free set
    }
    if (!fin) return;
    s = as_block_sum(s, sh);
    if (threadIdx.x == 0) pt_sums[blockIdx.x] = s;
    if (as_last_block(&cg->ticket[0])) {
        const double rz = as_final_sum(pt_sums, nblk, sh);
        if (threadIdx.x == 0) {
            cg->ticket[0] = 0;
            cg->beta = (first || !(cg->rz > 0.0)) ? 0.0 : rz / cg->rz;
            cg->rz = rz;
            if (!(rz > 0.0) || !isfinite(rz)) cg->info = 2;   // P is positive definite: r'z <= 0 means r = 0 or a broken factor
        }
    }
}


// v_i = c_i (the column tiles' partials of x_i' M x_i, added in tile order) - Phi_top[:, i] . t   on the free set, 0 elsewhere:
// v = R y, R = B - Phi_top Phi_top' (bq_as_pc2.hip formed the B part for THIS RANK's samples, as_pc_tphi_kernel t = Phi_top' y).
// blockIdx.x = a block of this rank's samples (block0 on); written in the gathered layout (slot of the block's segment).
__global__ __launch_bounds__(256) void as_pc_r_finish_kernel(int64_t N, int64_t ld, int64_t block0, as_pc_part part, int tiles, int ncols,
                                                             const float *__restrict__ Phitop, const double *__restrict__ c,
                                                             const double *__restrict__ ypart, const double *__restrict__ t,
                                                             const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                                             double *__restrict__ vg, const as_cg_scal *cg) {
    if (cg->done) return;
    __shared__ double ts[PC_MAX_M];
    for (int j = threadIdx.x; j < ncols; j += 256) ts[j] = t[j];
    __syncthreads();
    const int64_t blk = block0 + blockIdx.x;
    int k = part.lo;
    while (k + 1 < part.hi && blk >= part.blk[k + 1]) ++k;
    double *dst = vg + ((int64_t)part.slot[k] * part.maxlen + (blk - part.blk[k])) * BQ_VEC_TILE + 4 * threadIdx.x;
    const int64_t base = blk * BQ_VEC_TILE + 4 * threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = 0; j < ncols; ++j) {
        const as_f4 f = *reinterpret_cast<const as_f4 *>(Phitop + (int64_t)j * ld + base);
        const double tj = ts[j];
        acc[0] = fma((double)f.x, tj, acc[0]);
        acc[1] = fma((double)f.y, tj, acc[1]);
        acc[2] = fma((double)f.z, tj, acc[2]);
        acc[3] = fma((double)f.w, tj, acc[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + q;
        double r = 0.0;
        if (i < N && !(mL[i] | mU[i])) {
            double b = 0.0;
            for (int ct = 0; ct < tiles; ++ct) b += ypart[(int64_t)ct * ld + i];
            r = c[i] * b - acc[q];
        }
        dst[q] = r;
    }
}
// v (contiguous, every sample) from the gathered layout
__global__ __launch_bounds__(256) void as_pc_unpack_kernel(int64_t ld, as_pc_part part, const double *__restrict__ vg, double *__restrict__ v,
                                                           const as_cg_scal *cg) {
    if (cg->done) return;
    const int64_t blk = blockIdx.x;
    int k = 0;
    while (k + 1 < part.S && blk >= part.blk[k + 1]) ++k;
    const double *src = vg + ((int64_t)part.slot[k] * part.maxlen + (blk - part.blk[k])) * BQ_VEC_TILE;
    double *dst = v + blk * BQ_VEC_TILE;
#pragma unroll
    for (int q = 0; q < BQ_VEC_ITEMS; ++q) dst[threadIdx.x + q * BQ_VEC_BLOCK] = src[threadIdx.x + q * BQ_VEC_BLOCK];
    (void)ld;
}

// z = alpha y - beta z2 (both vanish outside the free set);  rz = r'z and beta of the conjugate gradients, as as_pc_apply_kernel does
__global__ __launch_bounds__(256) void as_pc_combine_kernel(int64_t N, double alpha, double beta, const double *__restrict__ r,
                                                            const double *__restrict__ y, const double *__restrict__ z2,
                                                            double *__restrict__ z, double *part, int64_t nblk, as_cg_scal *cg, int first) {
    if (cg->done) return;
    __shared__ double sh[4];
    double s = 0.0;
    VEC_LOOP(i) {
        double zi = 0.0;
        if (i < N) {
            zi = alpha * y[i] - beta * z2[i];
            s += __dmul_rn(r[i], zi);
        }
        z[i] = zi;
    }
    s = as_block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
    if (as_last_block(&cg->ticket[0])) {
        const double rz = as_final_sum(part, nblk, sh);
        if (threadIdx.x == 0) {
            cg->ticket[0] = 0;
            cg->beta = (first || !(cg->rz > 0.0)) ? 0.0 : rz / cg->rz;
            cg->rz = rz;
            if (!(rz > 0.0) || !isfinite(rz)) cg->info = 2;
        }
    }
}

__global__ void as_pc_fill_kernel(int64_t ld, double value, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < ld) out[i] = value;
}
// v = 1 on the free set, 0 elsewhere (start vector of the power iteration)
__global__ void as_pc_mask_ones_kernel(int64_t N, int64_t ld, const unsigned char *__restrict__ mL, const unsigned char *__restrict__ mU,
                                       double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < ld) out[i] = (i < N && !(mL[i] | mU[i])) ? 1.0 : 0.0;
}

// The model P = D + Phi Phi' is only used when it leaves every sample a diagonal share D_i / Q_ii of at least this much: a
// feature set that explains (or over-explains: D_i <= 0) the whole diagonal of some sample is a Taylor expansion outside its
// range (2 g |x|^2 not small) and would make P far worse conditioned than Q itself.
constexpr double PC_MIN_DIAG_SHARE = 0.1;

// the canonical sample segments (as many as the panel's: 8 up to 8 ranks) in blocks of 1024 samples, this rank's run of them and the
// slot of every segment in the gathered buffers — a function of the block count and the rank count only
void as_pc_make_part(const bq_ctx *ctx, int64_t nblk, as_pc_part *out) {
    as_pc_part &pt = *out;
    pt.S = bq_sym_segments(ctx->world);
    pt.lo = bq_sym_seg_first(ctx->rank, ctx->world, pt.S);
    pt.hi = bq_sym_seg_first(ctx->rank + 1, ctx->world, pt.S);
    pt.cmax = (pt.S + ctx->world - 1) / ctx->world;
    pt.maxlen = 0;
    for (int k = 0; k <= pt.S; ++k) pt.blk[k] = (long long)((int64_t)k * nblk / pt.S);
    for (int k = 0; k < pt.S; ++k) pt.maxlen = std::max(pt.maxlen, pt.blk[k + 1] - pt.blk[k]);
    for (int q = 0; q < ctx->world; ++q) {
        const int lo = bq_sym_seg_first(q, ctx->world, pt.S), hi = bq_sym_seg_first(q + 1, ctx->world, pt.S);
        for (int k = lo; k < hi; ++k) pt.slot[k] = q * pt.cmax + (k - lo);
    }
}

// build the preconditioner's features once per solver (null: the panel's kernel has none, the features do not fit this data,
// or BQ_AS_CG_PC=0)
int as_pc_create(bq_solver *s, as_pc **out) {
    *out = nullptr;
    bq_problem *p = s->p;
    if (!as_env_on("BQ_AS_CG_PC")) return BQ_OK;
    if (p->X == nullptr || (p->structure != BQ_PLAIN && p->structure != BQ_SVC)) return BQ_OK;
    if (p->kernel != BQ_KERNEL_RBF && !(p->kernel == BQ_KERNEL_LINEAR && p->diag_add > 0.0)) return BQ_OK;
    bq_ctx *ctx = p->ctx;
    // family 2 of the RBF features on BQ_SVC panels: 2 = projected order-2 directions (2d columns), 1 = the class-mean cross term of
    // rounds 3-4 (d columns), 0 = none.  hook as_cg_pc_class=0|1|2 caps it (tests compare them); a family that does not fit PC_MAX_M
    // features, or whose model leaves a sample too little of its diagonal, steps down.  With family 2 the REST of the order-2 term is
    // applied without features behind a degree-1 Chebyshev polynomial (bq_as_pc2.hip) where its bulk stands out of the diagonal —
    // n >= 65 536 (its eigenvalues grow like n / (d (d + 1) / 2); below, it costs more launches than it saves products) or
    // hook as_cg_pc_class=3 (tests).
    int fam2 = 0;
    bool want_r2 = false;
    if (p->kernel == BQ_KERNEL_RBF && p->structure == BQ_SVC) {
        double hv = 2.0;
        const bool forced = bq_hook("as_cg_pc_class", &hv);
        const int v = std::max(0, std::min((int)hv, 3));
        fam2 = std::min(v, 2);
        want_r2 = v == 3 || (!forced && p->n >= 65536);
    }
    std::vector<double> share((size_t)p->n);
    // the sum over ranks of a flag, so that every rank takes the same turn (each product of the solve is a collective: a rank that fell
    // back, or returned an error, alone would leave the others waiting in the next one for ever)
    auto agree = [&](double *flag) -> int {
        if (ctx->world <= 1 || ctx->comm_kind == BQ_COMM_SHARE) return BQ_OK;
        hipError_t ae = hipMemcpyAsync(s->partials, flag, sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        int arc = ae == hipSuccess ? bq_exchange_sum(ctx, s->partials, 1) : BQ_ERR_HIP;
        if (arc == BQ_OK) ae = hipMemcpyAsync(flag, s->partials, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (arc == BQ_OK && ae == hipSuccess) arc = bq_ctx_sync(ctx);   // behind a collective: the bounded wait
        if (arc == BQ_OK && ae != hipSuccess) {
            bq_set_error("agreeing on the preconditioner across ranks failed: %s", hipGetErrorString(ae));
            arc = BQ_ERR_HIP;
        }
        return arc;
    };
    for (; fam2 >= 0; --fam2) {
        const bool classes = fam2 > 0;
        int m = p->kernel == BQ_KERNEL_RBF ? (int)p->d + 1 + fam2 * (int)p->d : (int)p->d;
        if (p->add_one) m += 1;
        if (m > PC_MAX_M) {   // the apply kernel keeps the coefficients of all features in LDS
            if (fam2 == 0) return BQ_OK;
            continue;
        }
        as_pc *pc = new as_pc();
        pc->m = m;
        pc->mp = bq_round_up(m, 128);
        int rc = bq_chol_ws_create(ctx, pc->mp, &pc->ws);
        hipError_t e = hipSuccess;
        pc->m8 = bq_round_up(m, PC_FG);
        // (PC_FG spare zero rows: the t kernel walks feature groups of PC_FG from any first column — the Phi_top block of the remainder)
        if (rc == BQ_OK) e = hipMalloc(&pc->Phi, sizeof(float) * (size_t)(pc->m8 + PC_FG) * s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->Phi, 0, sizeof(float) * (size_t)(pc->m8 + PC_FG) * s->ldN, ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->tpart, sizeof(double) * (size_t)s->nblk * pc->mp);
        as_pc_make_part(ctx, s->nblk, &pc->part);
        {
            const size_t tgl = sizeof(double) * (size_t)ctx->world * pc->part.cmax * pc->mp;
            const size_t zgl = sizeof(double) * (size_t)ctx->world * pc->part.cmax * pc->part.maxlen * BQ_VEC_TILE;
            if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->tg, tgl);
            if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->tg, 0, tgl, ctx->stream);
            if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->zg, zgl);
            if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->zg, 0, zgl, ctx->stream);
        }
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->dinv, sizeof(double) * s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->z, sizeof(double) * s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->Gpart, sizeof(double) * PC_SLICES * pc->mp * pc->mp);
        if (rc == BQ_OK && e == hipSuccess)
            e = hipMemsetAsync(pc->Gpart, 0, sizeof(double) * PC_SLICES * pc->mp * pc->mp, ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->Ginv, sizeof(double) * pc->mp * pc->mp);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->u, sizeof(double) * pc->mp);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->sm_fail, sizeof(int));
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->sm_fail, 0, sizeof(int), ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->prev, (size_t)s->ldN);
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->prev, 0, (size_t)s->ldN, ctx->stream);
        if (rc == BQ_OK && e == hipSuccess) e = hipMalloc(&pc->chg, sizeof(int) * (2 + 2 * PC_MAX_CHG));
        if (rc == BQ_OK && e == hipSuccess) e = hipMemsetAsync(pc->chg, 0, sizeof(int) * (2 + 2 * PC_MAX_CHG), ctx->stream);
        if (rc == BQ_OK && e == hipSuccess && classes) {
            e = hipMalloc(&pc->cls, sizeof(double) * (2 * p->d + 2));
            if (e == hipSuccess) e = hipMemsetAsync(pc->cls, 0, sizeof(double) * (2 * p->d + 2), ctx->stream);
        }
        // fam2 == 2: per-sample numbers and the orthonormalising factor (set-up only) — allocated HERE, with everything else a rank may
        // have no room for, so that their failure lands in the agreed flag (they used to come after the agreement)
        double *raw = nullptr, *rinv_d = nullptr;
        if (rc == BQ_OK && e == hipSuccess && fam2 == 2) {
            e = hipMalloc(&raw, sizeof(double) * 3 * s->ldN);
            if (e == hipSuccess) e = hipMalloc(&rinv_d, sizeof(double) * (size_t)(2 * p->d) * (size_t)(2 * p->d));
        }
        // Do ALL ranks hold their features?  Every rank must run the SAME inner iteration (each product is a collective): a rank
        // that fell back to plain conjugate gradients alone — or returned an error alone — would leave the others waiting in the
        // next collective for ever.  So the outcome is agreed on (one all-reduce of a flag) and, if any rank has no room, every rank
        // runs unpreconditioned.
        // the implicit order-2 remainder (family 2 only): its images and buffers — a rank without room for them drops IT on every rank
        // (the 1e-3 digit of the agreed flag), not the explicit model
        double *bdiag = nullptr;
        if (rc == BQ_OK && e == hipSuccess && fam2 == 2 && want_r2) {
            hipError_t re = hipMalloc(&bdiag, sizeof(double) * s->ldN);
            for (double **v : {&pc->y1, &pc->v2, &pc->z2, &pc->ones})
                if (re == hipSuccess) re = hipMalloc(v, sizeof(double) * s->ldN);
            if (re == hipSuccess) re = hipMalloc(&pc->ttop, sizeof(double) * pc->mp);
            if (re == hipSuccess && as_pc2_create(s, bdiag, pc->mp, &pc->r2) != BQ_OK) re = hipErrorOutOfMemory;
            if (re == hipSuccess && pc->r2 != nullptr) {
                as_pc_fill_kernel<<<(unsigned)(s->ldN / 256), 256, 0, ctx->stream>>>(s->ldN, 1.0, pc->ones);
                pc->top0 = (int)p->d + 1;
                pc->ntop = 2 * (int)p->d;
            } else {
                (void)hipGetLastError();
                as_pc2_free(pc->r2);
                pc->r2 = nullptr;
            }
        }
        const bool r2_missing = fam2 == 2 && want_r2 && pc->r2 == nullptr;
        double failed = ((rc != BQ_OK || e != hipSuccess) ? 1.0 : 0.0) + (r2_missing ? 1e-3 : 0.0);
        if (failed >= 0.5) (void)hipGetLastError();
        {
            const int arc = agree(&failed);
            if (arc != BQ_OK) {
                if (bdiag) hipFree(bdiag);
                if (raw) hipFree(raw);
                if (rinv_d) hipFree(rinv_d);
                as_pc_free(pc);
                return arc;
            }
        }
        if (failed >= 0.5) {   // no room for the features on some rank: all ranks run plain conjugate gradients
            if (bdiag) hipFree(bdiag);
            if (raw) hipFree(raw);
            if (rinv_d) hipFree(rinv_d);
            as_pc_free(pc);
            return BQ_OK;
        }
        if (failed > 1e-6 && pc->r2 != nullptr) {   // some rank has no room for the remainder: nobody uses it
            as_pc2_free(pc->r2);
            pc->r2 = nullptr;
        }
        hipError_t fe = hipSuccess;
        int frc = BQ_OK;
        if (classes) {
            unsigned int *ticket = (unsigned int *)(pc->cls + 2 * p->d + 1);   // the spare slot, zeroed above
            as_pc_class_kernel<<<(unsigned)p->d, 256, 0, ctx->stream>>>(p->n, p->d, p->X, p->sgn, pc->cls, ticket);
            fe = hipGetLastError();
        }
        if (fam2 == 2 && fe == hipSuccess) {
            // the 2d x 2d Gram matrix of U_{c,k} = m_c e_k' + e_k m_c' (Frobenius): <U_{c,k}, U_{c',l}> = 2 (m_c.m_c') [k == l] + 2 m_c[l] m_c'[k],
            // its Cholesky factor R (columns whose pivot falls below 1e-8 of their diagonal are dropped: m_- = -m_+ leaves d directions)
            // and Rinv — on the host, from the class means (deterministic: the same on every rank)
            const int d = (int)p->d, n2 = 2 * d;
            std::vector<double> cls((size_t)n2 + 2), mc((size_t)n2);
            fe = hipMemcpyAsync(cls.data(), pc->cls, sizeof(double) * (n2 + 1), hipMemcpyDeviceToHost, ctx->stream);
            if (fe == hipSuccess) frc = bq_ctx_sync(ctx);
            if (fe == hipSuccess && frc == BQ_OK) {
                for (int k = 0; k < d; ++k) {
                    mc[k] = cls[d + k] + cls[k];
                    mc[d + k] = cls[d + k] - cls[k];
                }
                double dots[2][2] = {{0, 0}, {0, 0}};
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b)
                        for (int k = 0; k < d; ++k) dots[a][b] += mc[a * d + k] * mc[b * d + k];
                auto gram = [&](int i, int j) {
                    const int a = i / d, k = i % d, b = j / d, l = j % d;
                    return 2.0 * ((k == l ? dots[a][b] : 0.0) + mc[a * d + l] * mc[b * d + k]);
                };
                std::vector<double> R((size_t)n2 * n2, 0.0), Rinv((size_t)n2 * n2, 0.0);
                std::vector<int> kept;
                std::vector<double> c((size_t)n2);
                for (int j = 0; j < n2; ++j) {   // up-looking Cholesky over the kept columns
                    double piv = gram(j, j);
                    const double gjj = piv;
                    for (size_t a = 0; a < kept.size(); ++a) {
                        const int ia = kept[a];
                        double v = gram(ia, j);
                        for (size_t b = 0; b < a; ++b) v -= R[(size_t)kept[b] * n2 + ia] * c[b];
                        c[a] = v / R[(size_t)ia * n2 + ia];
                        piv -= c[a] * c[a];
                    }
                    if (!(piv > 1e-8 * gjj) || !(gjj > 0.0)) continue;   // (numerically) inside the span of the kept ones
                    for (size_t a = 0; a < kept.size(); ++a) R[(size_t)kept[a] * n2 + j] = c[a];
                    R[(size_t)j * n2 + j] = sqrt(piv);
                    kept.push_back(j);
                }
                for (size_t b = 0; b < kept.size(); ++b) {   // Rinv over the kept set by back substitution, column by column
                    const int jb = kept[b];
                    Rinv[(size_t)jb * n2 + jb] = 1.0 / R[(size_t)jb * n2 + jb];
                    for (size_t a = b; a-- > 0;) {
                        const int ia = kept[a];
                        double v = 0.0;
                        for (size_t q = a + 1; q <= b; ++q) v += R[(size_t)ia * n2 + kept[q]] * Rinv[(size_t)kept[q] * n2 + jb];
                        Rinv[(size_t)ia * n2 + jb] = -v / R[(size_t)ia * n2 + ia];
                    }
                }
                fe = hipMemcpyAsync(rinv_d, Rinv.data(), sizeof(double) * (size_t)n2 * n2, hipMemcpyHostToDevice, ctx->stream);
                if (fe == hipSuccess) frc = bq_ctx_sync(ctx);   // Rinv leaves this scope
            }
        }
        if (fe == hipSuccess && frc == BQ_OK) {
            as_pc_features_kernel<<<(unsigned)(s->ldN / 256), 256, 0, ctx->stream>>>(p->kernel, p->n, p->d, s->ldN, p->X, p->sgn, pc->cls, fam2,
                                                                                     p->gamma, p->add_one ? 1 : 0, p->diag_add, m,
                                                                                     pc->Phi, pc->z, raw);
            if (fam2 == 2)
                as_pc_project_kernel<<<dim3((unsigned)(s->ldN / 64), (unsigned)((2 * p->d + 63) / 64)), 256, 0, ctx->stream>>>(
                    p->n, p->d, s->ldN, p->X, raw, rinv_d, (int)p->d + 1, pc->Phi);
            as_pc_diag_kernel<<<(unsigned)(s->ldN / 256), 256, 0, ctx->stream>>>(p->n, s->ldN, m, pc->Phi, pc->dinv, pc->z,
                                                                                 pc->r2 ? bdiag : nullptr, pc->top0, pc->ntop);
            fe = hipGetLastError();
        }
        if (fe == hipSuccess && frc == BQ_OK) fe = hipMemcpyAsync(share.data(), pc->z, sizeof(double) * p->n, hipMemcpyDeviceToHost, ctx->stream);
        if (fe == hipSuccess && frc == BQ_OK) frc = bq_ctx_sync(ctx);
        if (raw) hipFree(raw);
        if (rinv_d) hipFree(rinv_d);
        if (bdiag) hipFree(bdiag);
        // the feature build is rank-local work that can fail on one rank alone (a launch, a copy, a bounded wait): a second agreement,
        // so that EVERY rank leaves solver creation with an error when any of them has one — none enters the solve's
        // collectives without the others
        double build_failed = (fe != hipSuccess || frc != BQ_OK) ? 1.0 : 0.0;
        if (build_failed > 0.0) (void)hipGetLastError();
        const int brc = agree(&build_failed);
        if (brc != BQ_OK || build_failed > 0.0) {
            as_pc_free(pc);
            if (brc != BQ_OK) return brc;
            if (fe != hipSuccess)
                bq_set_error("building the preconditioner features failed: %s", hipGetErrorString(fe));
            else if (frc == BQ_OK)
                bq_set_error("building the preconditioner features failed on another rank");
            return frc != BQ_OK ? frc : BQ_ERR_HIP;
        }
        double lo = 1.0;
        for (double v : share) lo = std::min(lo, v);
        if (std::isfinite(lo) && (p->kernel == BQ_KERNEL_LINEAR || lo >= PC_MIN_DIAG_SHARE)) {   // linear: the model is exact
            *out = pc;
            return BQ_OK;
        }
        as_pc_free(pc);   // the model leaves some sample too little of its diagonal: the next smaller family
    }
    return BQ_OK;
}

void as_pc_free(as_pc *pc) {
    if (!pc) return;
    if (pc->ws) bq_chol_ws_destroy(pc->ws);
    as_pc2_free(pc->r2);
    for (void *ptr : {(void *)pc->y1, (void *)pc->v2, (void *)pc->z2, (void *)pc->ones, (void *)pc->ttop})
        if (ptr) hipFree(ptr);
    for (void *ptr : {(void *)pc->Phi, (void *)pc->dinv, (void *)pc->z, (void *)pc->Gpart, (void *)pc->cls, (void *)pc->Ginv,
                      (void *)pc->u, (void *)pc->sm_fail, (void *)pc->prev, (void *)pc->chg, (void *)pc->tpart, (void *)pc->tg, (void *)pc->zg})
        if (ptr) hipFree(ptr);
    delete pc;
}

// t = Phi' (dinv o in) over `m` feature rows of `Phi`, sharded by samples: this rank's blocks -> per-segment sums into its slots of
// the gathered buffer `tg` (slot pitch gstride); the caller gathers and then adds the segments in order (as_pc_tsum_kernel)
static void as_pc_tpart(bq_solver *s, as_ws *w, int m, int64_t m8, const float *Phi, const double *dinv, const double *in, double *tg,
                        int64_t gstride) {
    as_pc *pc = w->pc;
    hipStream_t st = s->p->ctx->stream;
    const as_pc_part &pt = pc->part;
    const int64_t b0 = pt.blk[pt.lo], b1 = pt.blk[pt.hi];
    if (b1 <= b0) return;
    as_pc_tphi_kernel<<<dim3((unsigned)(b1 - b0), 1), BQ_VEC_BLOCK, 0, st>>>(m, m8, pc->mp, s->N, s->ldN, b0, Phi, dinv, in, pc->tpart, w->cg);
    as_pc_tseg_kernel<<<dim3((unsigned)(pc->mp / 16), (unsigned)(pt.hi - pt.lo)), 256, 0, st>>>(m8, pc->mp, pt, pc->tpart, tg, gstride, w->cg);
}

// out = P1_AA^-1 in (in vanishes outside the free set); fin: also r'z and beta of the conjugate gradients (in is their residual).
// Two passes over Phi, both over THIS RANK's samples only; two collectives: the per-segment sums of t (S x mp doubles) and z.
static int as_pc_solve1(bq_solver *s, as_ws *w, const double *in, double *out, int first, int fin) {
    as_pc *pc = w->pc;
    bq_ctx *ctx = s->p->ctx;
    hipStream_t st = ctx->stream;
    const as_pc_part &pt = pc->part;
    const int64_t b0 = pt.blk[pt.lo], b1 = pt.blk[pt.hi];
    as_pc_tpart(s, w, pc->m, pc->m8, pc->Phi, pc->dinv, in, pc->tg, pc->mp);
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_exchange_gather(ctx, pc->tg, (int64_t)pt.cmax * pc->mp));
    as_pc_tsum_kernel<<<(unsigned)((pc->mp + 255) / 256), 256, 0, st>>>(pc->mp, pt, pc->tg, pc->mp, pc->ws->rhs, w->cg);
    as_pc_gemv_kernel<<<(unsigned)((pc->mp + 3) / 4), 256, 0, st>>>(pc->mp, pc->Ginv, pc->ws->rhs, pc->u, w->cg);   // u = G^-1 t
    if (b1 > b0)
        as_pc_apply_kernel<<<(unsigned)(b1 - b0), BQ_VEC_BLOCK, 0, st>>>(pc->m, pc->m8, s->N, s->ldN, b0, pt, pc->Phi, pc->dinv, s->mL, s->mU, in,
                                                                       pc->u, pc->zg, w->cg);
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_exchange_gather(ctx, pc->zg, (int64_t)pt.cmax * pt.maxlen * BQ_VEC_TILE));
    as_pc_zunpack_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, pt, pc->zg, in, out, s->partials, s->nblk, w->cg, first, fin);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

// v2 = R y on the free set, R = B - Phi_top Phi_top' (y vanishes outside the free set)
static int as_pc_r_apply(bq_solver *s, as_ws *w, const double *y) {
    as_pc *pc = w->pc;
    hipStream_t st = s->p->ctx->stream;
    // Phi_top' y: this rank's samples, per-segment sums into the TAIL of the remainder's gathered slots — they ride on its all-gather
    const float *Phitop = pc->Phi + (int64_t)pc->top0 * s->ldN;
    const int64_t m8t = bq_round_up(pc->ntop, PC_FG);   // (Phi carries PC_FG spare zero rows behind its last feature)
    int64_t gstride = 0;
    double *tail = as_pc2_tail(pc->r2, &gstride);
    as_pc_tpart(s, w, pc->ntop, m8t, Phitop, pc->ones, y, tail, gstride);
    BQ_TRY(as_pc2_bpart(s, pc->r2, y, w->cg));   // (gathers M's per-segment sums and the tail)
    as_pc_tsum_kernel<<<(unsigned)((pc->mp + 255) / 256), 256, 0, st>>>(pc->mp, pc->part, tail, gstride, pc->ttop, w->cg);
    int tiles = 0;
    const double *ypart = as_pc2_ypart(pc->r2, &tiles);
    const as_pc_part pt = *as_pc2_part(pc->r2);
    double *vg = as_pc2_vg(pc->r2);
    const int64_t b0 = pt.blk[pt.lo], b1 = pt.blk[pt.hi];   // this rank's sample blocks
    if (b1 > b0)
        as_pc_r_finish_kernel<<<(unsigned)(b1 - b0), BQ_VEC_BLOCK, 0, st>>>(s->N, s->ldN, b0, pt, tiles, pc->ntop, Phitop, as_pc2_c(pc->r2), ypart,
                                                                          pc->ttop, s->mL, s->mU, vg, w->cg);
    BQ_HIP(hipGetLastError());
    BQ_TRY(bq_exchange_gather(s->p->ctx, vg, (int64_t)pt.cmax * pt.maxlen * BQ_VEC_TILE));
    as_pc_unpack_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->ldN, pt, vg, pc->v2, w->cg);
    BQ_HIP(hipGetLastError());
    return BQ_OK;
}

// z = P_AA^-1 r (+ r'z and beta on the device).  Explicit model: one Woodbury application.  With the order-2 remainder:
// z = alpha y - beta P1^-1 (R y), y = P1^-1 r — the degree-1 Chebyshev polynomial of bq_as_pc2.hip (alpha = 1, beta = 0 until the
// spectrum bound has been estimated).  More Chebyshev steps were measured at config 5 (two and three applications of R per call): the
// outer iteration still takes 5.0 products and gets 7 / 18 ms longer (profiles/r05/c5_chebyshev_steps.txt) — the order-2 MODEL is the
// limit, not how exactly it is inverted; the CPU study said the same (18 / 18 / 17 iterations).
int as_pc_apply(bq_solver *s, as_ws *w, int first) {
    as_pc *pc = w->pc;
    hipStream_t st = s->p->ctx->stream;
    double alpha = 1.0, beta = 0.0;
    if (pc->r2) as_pc2_coefs(pc->r2, &alpha, &beta);
    // profiling (BQ_PROF_PCSHARD): one application of the preconditioner — all of its passes over samples are sharded since round 6
    // (what stays replicated inside it: the m x m product, the segment sums and the unpacking of two n-vectors: microseconds)
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    BQ_TRY(bq_prof_begin(s->p->ctx, BQ_PROF_PCSHARD, &pe0, &pe1));
    if (pc->r2 == nullptr || beta == 0.0) {
        BQ_TRY(as_pc_solve1(s, w, w->r, pc->z, first, 1));   // two collectives
    } else {
        BQ_TRY(as_pc_solve1(s, w, w->r, pc->y1, first, 0));
        BQ_TRY(as_pc_r_apply(s, w, pc->y1));                 // six collectives in all
        BQ_TRY(as_pc_solve1(s, w, pc->v2, pc->z2, first, 0));
        as_pc_combine_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(s->N, alpha, beta, w->r, pc->y1, pc->z2, pc->z, s->partials, s->nblk, w->cg,
                                                                     first);
        BQ_HIP(hipGetLastError());
    }
    return bq_prof_end(s->p->ctx, BQ_PROF_PCSHARD, pe0, pe1);
}

// which samples entered / left the free set since the preconditioner's G^-1 was brought up to date: the list (and whether it is short
// enough for rank-one updates) rides on the outer iteration's one look at the device, so that the host knows whether to enqueue the
// update kernel or a rebuild without a synchronisation of its own
void as_pc_track(bq_solver *s, as_ws *w, hipStream_t st) {
    as_pc *pc = w->pc;
    const int64_t N = s->N;
    const int force = (pc->age == 0 || pc->age >= 128 || !bq_hook_on("as_cg_pc_incr")) ? 1 : 0;
    int *lcnt = reinterpret_cast<int *>(s->partials + s->nblk);
    as_pc_diff_count_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, pc->prev, lcnt, &s->sc->pad1[0], pc->chg, force);
    as_pc_diff_write_kernel<<<vgrid(s->ldN), BQ_VEC_BLOCK, 0, st>>>(N, s->mL, s->mU, pc->prev, lcnt, pc->chg);
    (void)hipMemcpyAsync(w->host_info + 4, pc->chg, 2 * sizeof(int), hipMemcpyDeviceToHost, st);   // pinned, like the rest
}

int as_pc_update(bq_solver *s, as_ws *w) {
    as_pc *pc = w->pc;
    hipStream_t st = s->p->ctx->stream;
    const int64_t N = s->N;
    // G^-1, G = I + Phi_A' D_A^-1 Phi_A, for the free set of this outer iteration.  The set moves by an index or two per outer
    // iteration (the list was made at the top of the iteration, as_pc_diff_*): G^-1 follows by Sherman-Morrison updates in index
    // order (the same bits on every rank), and is rebuilt from G summed afresh over all samples at the start, every 128 outer
    // iterations, when more than 64 samples moved at once, and after an update that failed.
    const bool rebuild = pc->host_chg[1] != 0 || pc->age == 0;
    pc->age = rebuild ? 1 : pc->age + 1;
    if (rebuild) {
        const int tiles = (int)(pc->mp / PC_T);
        const dim3 gtri((unsigned)((pc->mp + 255) / 256), (unsigned)pc->mp);
        as_pc_gram_kernel<<<dim3((unsigned)(tiles * (tiles + 1) / 2), PC_SLICES), 256, 0, st>>>(pc->m, pc->mp, N, s->ldN, pc->Phi,
                                                                                               pc->dinv, s->mL, s->mU, pc->Gpart);
        as_pc_gram_reduce_kernel<<<gtri, 256, 0, st>>>(pc->m, pc->mp, pc->Gpart, pc->ws->H, pc->ws->ldh);
        BQ_TRY(bq_chol_factor(pc->ws, pc->mp));
        BQ_TRY(bq_chol_prepare_sweeps(pc->ws, pc->mp));   // the explicit inverse factor (mp <= 1024: one block)
        as_pc_ginv_kernel<<<dim3((unsigned)(pc->mp / 16), (unsigned)(pc->mp / 16)), 256, 0, st>>>(pc->mp, pc->ws->bigMT, pc->ws->bb, pc->Ginv);
        BQ_HIP(hipMemsetAsync(pc->sm_fail, 0, sizeof(int), st));
        pc->rebuilds += 1;
    } else {
        as_pc_sm_kernel<<<1, 1024, 0, st>>>(pc->m, pc->mp, s->ldN, pc->Phi, pc->dinv, pc->chg, pc->Ginv, pc->sm_fail);
    }
    // the bound was estimated on a free set of lambda_nA samples: restricting to a smaller set only lowers lambda_max, a LARGER one (a
    // resumed solve that starts from given masks, releases after an early estimate) may exceed it — and an underestimate is what
    // turns the polynomial indefinite (profiles/r05/c5_lambda_scale.txt): estimate again.  Replicated numbers: every
    // rank takes the same turn.
    if (pc->r2 && rebuild && as_pc2_lambda(pc->r2) >= 0.0 && (long long)w->host_ints[0] > pc->lambda_nA + pc->lambda_nA / 8)
        as_pc2_set_lambda(pc->r2, -1.0);
    if (pc->r2 && rebuild && as_pc2_lambda(pc->r2) < 0.0) {
        // lambda_max of P1^-1 R on this free set by a power iteration from the all-ones vector (twelve applications, two norms on the
        // host; repeated only when the free set has grown by more than an eighth since).  Every rank computes the same bits.
        // The polynomial stays positive up to 1 + the assumed bound, so an estimate from BELOW must be widened, never trusted: at
        // config 5 six applications x 1.15 still gave 5.0 products per outer iteration, x 0.85 gave 8.0 and x 0.6 36 (the operator
        // turns indefinite); x 1.6 is as good as x 1.15 (profiles/r05/c5_lambda_scale.txt) — wide is cheap, narrow is not.
        // A spectrum bound that cannot be formed (a solve that was over before it began: the kernels return on `done`) is retried
        // at the next rebuild; until then the explicit model alone is used.
        std::vector<double> a((size_t)N), b((size_t)N);
        as_pc_mask_ones_kernel<<<(unsigned)(s->ldN / 256), 256, 0, st>>>(N, s->ldN, s->mL, s->mU, pc->y1);
        double lam = -1.0;
        for (int it = 0; it < 12; ++it) {
            if (it == 11) BQ_HIP(hipMemcpyAsync(a.data(), pc->y1, sizeof(double) * N, hipMemcpyDeviceToHost, st));
            BQ_TRY(as_pc_r_apply(s, w, pc->y1));
            BQ_TRY(as_pc_solve1(s, w, pc->v2, pc->y1, 1, 0));
        }
        BQ_HIP(hipMemcpyAsync(b.data(), pc->y1, sizeof(double) * N, hipMemcpyDeviceToHost, st));
        BQ_TRY(bq_ctx_sync(s->p->ctx));
        double na = 0.0, nb = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            na += a[(size_t)i] * a[(size_t)i];
            nb += b[(size_t)i] * b[(size_t)i];
        }
        if (na > 0.0 && std::isfinite(na) && std::isfinite(nb)) lam = sqrt(nb / na);
#ifndef BQ_PC2_LAMBDA_SCALE
#define BQ_PC2_LAMBDA_SCALE 1.5   // a power iteration comes from below (swept at config 5: profiles/r05/c5_lambda_scale.txt)
#endif
        if (lam >= 0.0 && std::isfinite(lam)) {
            as_pc2_set_lambda(pc->r2, BQ_PC2_LAMBDA_SCALE * lam);
            pc->lambda_nA = (long long)w->host_ints[0];
        }
    }
    return BQ_OK;
}
