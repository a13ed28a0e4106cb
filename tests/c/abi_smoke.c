/* Plain-C consumer of include/bcqp.h: proves the boundary is a C ABI (no C++/torch types in the signatures).
 * Built and run by tests/test_abi.py::test_c_consumer.  With a GPU (argv[1] == "gpu") it also solves a tiny box QP. */
#include <stdio.h>
#include <string.h>

#include "bcqp.h"

int main(int argc, char **argv) {
    int64_t b = -1, e = -1;
    if (bq_abi_version() != BQ_ABI_VERSION) return 10;
    if (bq_row_block(1000, 1, 4, &b, &e) != BQ_OK || b != 256 || e != 512) return 11;
    if (bq_sym_row_block(100000, 7, 8, &b, &e) != BQ_OK || e != 100000 || b % 256 != 0) return 12;
    if (bq_row_block(10, 5, 2, NULL, NULL) != BQ_ERR_BADARG || strlen(bq_last_error()) == 0) return 13;
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
        /* min 1/2 x'Qx + q'x, 0 <= x <= 1 with Q = [[2,0],[0,2]], q = [-1,-4]  ->  x* = (0.5, 1) */
        const double Q[4] = {2, 0, 0, 2}, q[2] = {-1, -4}, ub[2] = {1, 1};
        double x[2] = {0, 0};
        bq_ctx *ctx = NULL;
        bq_problem *p = NULL;
        bq_solver *s = NULL;
        bq_iter_stat rows[64];
        int64_t n = 0;
        int status = 0;
        if (bq_ctx_create(0, &ctx) != BQ_OK) return 20;
        if (bq_problem_create_dense(ctx, 2, Q, q, BQ_F64, &p) != BQ_OK) return 21;
        if (bq_solver_create(p, BQ_PG, NULL, ub, NULL, 1e-6, 1000, 0.0, &s) != BQ_OK) return 22;
        if (bq_solver_run(s, 64, rows, 64, &n, &status) != BQ_OK) return 23;
        if (status != BQ_STATUS_OPTIMAL) return 24;
        if (bq_solver_get(s, BQ_GET_X_NOW, x) != BQ_OK) return 25;
        if (x[0] < 0.4999 || x[0] > 0.5001 || x[1] < 0.9999 || x[1] > 1.0001) return 26;
        bq_solver_destroy(s);
        {
            /* checkpoint / resume through the plain-C struct: InteriorPoint stopped after 3 iterations continues in a NEW solver and
             * ends bit for bit where an uninterrupted one ends (bq_solver_get_state / bq_solver_set_state) */
            double xa[2], xb[2], sx[2], sg[2], slp[2], slm[2];
            bq_solver_snapshot st;
            bq_solver *a = NULL, *b2 = NULL, *c2 = NULL;
            if (bq_solver_create(p, BQ_IP, NULL, ub, NULL, 1e-10, 1000, 0.0, &a) != BQ_OK) return 30;
            if (bq_solver_run(a, 64, rows, 64, &n, &status) != BQ_OK || status != BQ_STATUS_OPTIMAL) return 31;
            if (bq_solver_get(a, BQ_GET_X_NOW, xa) != BQ_OK) return 32;
            if (bq_solver_create(p, BQ_IP, NULL, ub, NULL, 1e-10, 3, 0.0, &b2) != BQ_OK) return 33;
            if (bq_solver_run(b2, 8, rows, 64, &n, &status) != BQ_OK || status != BQ_STATUS_STOPPED) return 34;
            memset(&st, 0, sizeof(st));
            st.x = sx;
            st.g = sg;
            st.lp = slp;
            st.lm = slm;
            if (bq_solver_get_state(b2, &st) != BQ_OK || st.iter != 3 || st.kind != BQ_IP) return 35;
            if ((st.have & (BQ_STATE_X | BQ_STATE_G | BQ_STATE_MULT)) != (BQ_STATE_X | BQ_STATE_G | BQ_STATE_MULT)) return 36;
            if (bq_solver_create(p, BQ_IP, NULL, ub, NULL, 1e-10, 1000, 0.0, &c2) != BQ_OK) return 37;
            if (bq_solver_set_state(c2, &st) != BQ_OK) return 38;
            if (bq_solver_run(c2, 64, rows, 64, &n, &status) != BQ_OK || status != BQ_STATUS_OPTIMAL) return 39;
            if (bq_solver_get(c2, BQ_GET_X_NOW, xb) != BQ_OK) return 40;
            if (memcmp(xa, xb, sizeof(xa)) != 0) return 41;
            if (bq_solver_set_state(c2, &st) != BQ_ERR_BADARG) return 42;   /* only before the first run */
            bq_solver_destroy(a);
            bq_solver_destroy(b2);
            bq_solver_destroy(c2);
        }
        bq_problem_destroy(p);
        bq_ctx_destroy(ctx);
        printf("c abi gpu ok: x = (%.6f, %.6f) after %lld records; InteriorPoint resumed bit for bit\n", x[0], x[1], (long long)n);
    }
    printf("c abi ok\n");
    return 0;
}
