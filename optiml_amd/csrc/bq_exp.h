// exp for the kernel maps, ONE definition for the device (bq_gram.hip) and for the host check (tools/exp_check.c, run by
// tests/test_host_math.py): plain C99 + fma / fmax / ldexp.
//
// 18 vector instructions against ~25 of the device library's exp: round-to-nearest of x log2(e) by the 1.5 * 2^52 trick (the
// integer is the low word of the sum: no rndne / cvt), two-term Cody-Waite reduction, degree-11 polynomial with the
// Chebyshev-node coefficients of (exp(r) - 1 - r) / r^2 on |r| <= ln(2) / 2 (tools/exp_fit.py: 1.7e-17 relative), one ldexp that
// also produces the subnormal results; the clamp at -746 replaces the library's range selects.  <= 1 ulp from glibc's exp on
// [-746, 0], exp(0) = 1 exactly; correct up to the overflow threshold for positive arguments as well.  A NaN argument gives
// 0 (max drops it), like the fmax(dist, 0) the kernel maps put in front of it.  Matters because vector instructions are paid
// for in matrix-pipe time on gfx950 (bq_mfma_tile.h): every instruction of a tile epilogue is taken from the MFMAs.
#pragma once
#ifndef BQ_EXP_ATTR
#define BQ_EXP_ATTR static inline
#endif
#ifndef BQ_EXP_LOINT   // low 32 bits of the representation of a double, as int
#include <stdint.h>
#include <string.h>
static inline int bq_exp_loint_host(double t) {
    int64_t b;
    memcpy(&b, &t, 8);
    return (int)(int32_t)(uint32_t)b;
}
#define BQ_EXP_LOINT(t) bq_exp_loint_host(t)
#endif
BQ_EXP_ATTR double bq_exp(double x) {
#ifdef BQ_EXP_DIAG_CHEAP   /* diagnostic builds only: what the kernel maps would cost with a 2-instruction "exp" (wrong values) */
    return fma(x, 0.5, 1.0);
#endif
    x = fmax(x, -746.0);
    const double magic = 6755399441055744.0;   // 1.5 * 2^52
    const double t = fma(x, 1.4426950408889634074, magic);
    const double n = t - magic;
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 0x1.af3a57ea0843fp-26;
    p = fma(p, r, 0x1.2891a1928aa16p-22);
    p = fma(p, r, 0x1.71de0c9540aa2p-19);
    p = fma(p, r, 0x1.a019b8f77d16ep-16);
    p = fma(p, r, 0x1.a01a01a8454fcp-13);
    p = fma(p, r, 0x1.6c16c1789064ap-10);
    p = fma(p, r, 0x1.1111111110834p-7);
    p = fma(p, r, 0x1.5555555553d5ep-5);
    p = fma(p, r, 0x1.5555555555556p-3);
    p = fma(p, r, 0x1.0000000000001p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, BQ_EXP_LOINT(t));
}
