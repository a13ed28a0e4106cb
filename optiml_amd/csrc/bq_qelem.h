// One entry Q[i][j] of a problem's Hessian read from its resident panel (the H assembly of the factorisations and the
// ActiveSet update columns share it): dense panels as stored, kernel panels through the structure flags
// (BQ_SVC: y_i y_j (K_ij + 1), BQ_SVR: +-(K + 1) on the n x n blocks, BQ_H_KPLUS1: K + 1), plus diag_add on the diagonal.
#pragma once
#include "bq_common.h"

constexpr int BQ_H_KPLUS1_MODE = 3;   // == BQ_H_KPLUS1 (bq_chol.h)

template <typename T>
__device__ __forceinline__ double bq_q_elem(int structure, const T *__restrict__ panel, int64_t ldp, int packed, int64_t n,
                                            const double *__restrict__ sgn, double diag_add, int64_t i, int64_t jj) {
    if (jj > i) {   // kernel-built panels keep only the tiles on/below the diagonal: always read (max, min)
        const int64_t t = i;
        i = jj;
        jj = t;
    }
    double v;
    if (structure == BQ_PLAIN) {
        v = (double)panel[packed ? bq_sym_addr(i, jj, 0) : i * ldp + jj];
    } else if (structure == BQ_SVC) {
        v = sgn[i] * sgn[jj] * ((double)panel[bq_sym_addr(i, jj, 0)] + 1.0);
    } else if (structure == BQ_H_KPLUS1_MODE) {
        v = (double)panel[bq_sym_addr(i, jj, 0)] + 1.0;
    } else {
        const int64_t ii = i >= n ? i - n : i, jn = jj >= n ? jj - n : jj;
        const int64_t hi = ii > jn ? ii : jn, lo = ii > jn ? jn : ii;
        const double pv = (double)panel[bq_sym_addr(hi, lo, 0)] + 1.0;
        v = ((i >= n) == (jj >= n)) ? pv : -pv;
    }
    if (i == jj && diag_add != 0.0) v += diag_add;
    return v;
}
