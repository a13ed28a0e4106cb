// Factorisation workspace shared by the interior-point and active-set drivers.
#pragma once
#include "bq_common.h"

struct bq_chol_ws {
    bq_ctx *ctx = nullptr;
    int64_t cap = 0;   // largest padded order the workspace can hold (multiple of 128)
    int64_t ldh = 0;   // row pitch of H
    double *H = nullptr;      // cap x ldh, row-major; lower triangle = matrix, then its Cholesky factor
    double *Wt = nullptr;     // 2 x (super_max x 256 x ldh) k-major images of the block columns of a super-pass (double-buffered)
    int super_max = 2;        // passes per super-pass the image buffers hold
    double *LinvT = nullptr;  // per diagonal block: 128 x 128 image of the inverse triangular factor
    double *rhs = nullptr;    // right-hand side / solution (padded)
    double *tmp = nullptr;    // 128 scratch
    int *info = nullptr;      // 0 = ok, else 1 + index of the first non-positive pivot
    unsigned int *ticket = nullptr;   // last-block ticket of the fused solve kernels
    double *mr_vec = nullptr; // 10 x cap scratch vectors of the MINRES fallback (allocated on first use)
    void *mr_state = nullptr; // device scalars, partial sums and the pinned done flag of its multi-workgroup form
    double *mr_part = nullptr;
    int64_t mr_part_cap = 0;
    int *mr_flag = nullptr;
    // fast sweeps (bq_chol_prepare_sweeps): inverses of the 1024 x 1024 diagonal blocks of the factor and their transposes,
    // scratch of their construction, the 1024-vector between the two launches of a block step; sweep_np = the factor order
    // they were prepared for (0: not prepared; bq_chol_solve then walks the 128-row blocks)
    double *bigM = nullptr, *bigMT = nullptr, *big_scratch = nullptr, *sw_t = nullptr;
    int64_t big_cap = 0, sweep_np = 0;
    int64_t bb = 1024;        // rows of a big block of the prepared sweeps (bq_chol_prepare_sweeps chooses it)
    // look-ahead: the narrow work of pass p+1 (diagonal blocks, TRSM, column update) runs on a side stream that owns
    // a few reserved CUs while the wide trailing update of pass p runs on the rest of the chip
    hipStream_t s_main = nullptr, s_side = nullptr;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool lookahead = false;
    // pivot_rel > 0: a pivot <= pivot_rel * (the diagonal entry of the matrix as given) counts as non-positive.  LAPACK's test is
    // "<= 0", which on a numerically singular matrix is decided by the sign of rounding noise; ActiveSet sets 1e-13 so that a
    // singular Q_AA reliably takes the reference's minres branch (active_set.py:142-151).  0: LAPACK's test (InteriorPoint, linalg).
    double pivot_rel = 0.0;
    double *pivot_thr = nullptr;   // cap thresholds, filled at the start of a factorisation when pivot_rel > 0
};

int bq_chol_factor(bq_chol_ws *ws, int64_t np);
// rhs[0:first_nonzero) is known to be zero; also (may be null): the solution is written there as well as into ws->rhs
int bq_chol_solve(bq_chol_ws *ws, int64_t np, int64_t first_nonzero = 0, double *also = nullptr);
// after a successful bq_chol_factor of a factor that will be solved with many times: two launches per 1024 rows and
// direction instead of one per 128 (writes L^T into the upper triangle of H)
int bq_chol_prepare_sweeps(bq_chol_ws *ws, int64_t np);
constexpr int BQ_H_KPLUS1 = 3;   // internal H-assembly mode: entries K_ij + 1 of the n x n panel
int bq_chol_build_h(bq_chol_ws *ws, bq_problem *p, const int *idx, int64_t m, const double *hd, int64_t *np_out,
                    bool full = false, int structure_override = -1);
// bq_minres.hip: x = argmin |H x - q'| via MINRES on H H^T x = H q' (H = ws->H full symmetric, q' = ws->rhs)
int bq_minres_normal(bq_chol_ws *ws, const int *nA_dev, int64_t nA, int64_t np, double *vec, int *iters_dev);
