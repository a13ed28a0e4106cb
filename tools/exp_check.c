#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static double exp_nonpos(double x) {
    x = fmax(x, -746.0);
    const double MAGIC = 6755399441055744.0;
    const double t = fma(x, 1.4426950408889634074, MAGIC);
    const double n = t - MAGIC;
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
    int64_t ti; memcpy(&ti, &t, 8);
    return ldexp(p, (int)(int32_t)(uint32_t)ti);
}
static double ulp_err(double a, double b) {
    if (a == b) return 0;
    int e; frexp(b, &e);
    double u = ldexp(1.0, e - 53);
    if (b != 0 && fabs(b) < 2.3e-308) u = 4.9406564584124654e-324;
    return fabs(a - b) / u;
}
int main() {
    double worst = 0, wx = 0; long bad = 0;
    srand48(1);
    for (long i = 0; i < 40000000; ++i) {
        double x;
        int m = i % 4;
        if (m == 0) x = -drand48() * 746.0;
        else if (m == 1) x = -drand48() * 40.0;
        else if (m == 2) x = -drand48();
        else x = -exp(-drand48() * 40.0);
        double a = exp_nonpos(x), b = exp(x);
        double e = ulp_err(a, b);
        if (e > worst) { worst = e; wx = x; }
        if (e > 1.0) ++bad;
    }
    printf("worst %.3f ulp at x=%.17g  (>1ulp: %ld)\n", worst, wx, bad);
    printf("exp(0)=%.17g exp(-0.0)=%.17g exp(-746)=%g exp(-1e308)=%g exp(-inf)=%g exp(-745)=%g vs %g\n", exp_nonpos(0.0), exp_nonpos(-0.0), exp_nonpos(-746.0), exp_nonpos(-1e308), exp_nonpos(-INFINITY), exp_nonpos(-745.0), exp(-745.0));
    printf("exp(-708.5)=%.17g vs %.17g ; exp(-720)=%.17g vs %.17g\n", exp_nonpos(-708.5), exp(-708.5), exp_nonpos(-720.0), exp(-720.0));
    return 0;
}
