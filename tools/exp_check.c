/* Host check of the kernel maps' exp (optiml_amd/csrc/bq_exp.h, the very definition the device compiles) against libm:
 *   gcc -O2 -I optiml_amd/csrc tools/exp_check.c -lm -o exp_check && ./exp_check [points]
 * exit status 0 iff the worst error is <= 1 ulp and the edge values are right. */
#define _XOPEN_SOURCE 600
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "bq_exp.h"
static double ulp_err(double a, double b) {
    if (a == b) return 0;
    int e;
    frexp(b, &e);
    double u = ldexp(1.0, e - 53);
    if (b != 0 && fabs(b) < 2.3e-308) u = 4.9406564584124654e-324;
    return fabs(a - b) / u;
}
int main(int argc, char **argv) {
    const long points = argc > 1 ? atol(argv[1]) : 40000000;
    double worst = 0, wx = 0;
    srand48(1);
    for (long i = 0; i < points; ++i) {
        double x;
        const int m = i % 4;
        if (m == 0) x = -drand48() * 746.0;
        else if (m == 1) x = -drand48() * 40.0;
        else if (m == 2) x = -drand48();
        else x = -exp(-drand48() * 40.0);
        const double e = ulp_err(bq_exp(x), exp(x));
        if (e > worst) { worst = e; wx = x; }
    }
    printf("worst %.3f ulp at x=%.17g over %ld points\n", worst, wx, points);
    int bad = worst > 1.0;
    bad |= bq_exp(0.0) != 1.0 || bq_exp(-0.0) != 1.0 || bq_exp(-746.0) != 0.0 || bq_exp(-1e308) != 0.0 || bq_exp(-INFINITY) != 0.0;
    bad |= bq_exp(-745.0) != exp(-745.0) || bq_exp(-720.0) != exp(-720.0) || bq_exp(-708.5) != exp(-708.5);
    bad |= ulp_err(bq_exp(1.0), exp(1.0)) > 1.0 || ulp_err(bq_exp(700.0), exp(700.0)) > 1.0 || !isinf(bq_exp(710.0));
    printf("edges %s\n", bad ? "WRONG" : "ok");
    return bad;
}
