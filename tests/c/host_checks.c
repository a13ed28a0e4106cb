/* Host-only walk over the C ABI for the sanitizer build (tests/test_sanitize.py): pure-arithmetic entry points over a grid of
 * arguments, every bad-argument path that returns before touching a device, the error-string plumbing, and — where no GPU is
 * visible — the failing context / device paths.  Exit code 0 = every expectation held (ASan / UBSan abort on their own). */
#include <stdio.h>
#include <string.h>

#include "bcqp.h"

#define EXPECT(cond, code) do { if (!(cond)) { fprintf(stderr, "host_checks: expectation %d failed: %s\n", code, #cond); return code; } } while (0)

int main(void) {
    int64_t b, e, prev;
    EXPECT(bq_abi_version() == BQ_ABI_VERSION, 1);
    for (int64_t n = 2; n < 300000; n = n * 3 + 1)
        for (int world = 1; world <= 8; ++world) {
            prev = 0;
            for (int r = 0; r < world; ++r) {
                EXPECT(bq_row_block(n, r, world, &b, &e) == BQ_OK && b == prev && e >= b && e <= n, 2);
                prev = e;
            }
            EXPECT(prev == n, 3);
            prev = 0;
            for (int r = 0; r < world; ++r) {
                EXPECT(bq_sym_row_block(n, r, world, &b, &e) == BQ_OK && b == prev && e >= b && e <= n, 4);
                EXPECT(b % 256 == 0 || b == n, 5);
                prev = e;
            }
            EXPECT(prev == n, 6);
        }
    EXPECT(bq_row_block(10, 3, 2, NULL, NULL) == BQ_ERR_BADARG && strstr(bq_last_error(), "rank") != NULL, 7);
    EXPECT(bq_sym_row_block(-1, 0, 1, &b, &e) == BQ_ERR_BADARG, 8);
    EXPECT(bq_row_block(10, 0, 1, NULL, NULL) == BQ_OK, 9);   /* NULL outputs are allowed */
    /* NULL handles and arguments: reported, never dereferenced */
    EXPECT(bq_problem_matvec(NULL, NULL, NULL) == BQ_ERR_BADARG, 10);
    EXPECT(bq_problem_eval(NULL, NULL, NULL, NULL) == BQ_ERR_BADARG, 11);
    EXPECT(bq_problem_dims(NULL, NULL, NULL, NULL, NULL) == BQ_ERR_BADARG, 12);
    EXPECT(bq_solver_run(NULL, 1, NULL, 0, NULL, NULL) == BQ_ERR_BADARG, 13);
    EXPECT(bq_solver_get(NULL, 0, NULL) == BQ_ERR_BADARG, 14);
    EXPECT(bq_solver_state(NULL, NULL, NULL, NULL) == BQ_ERR_BADARG, 15);
    EXPECT(bq_smo_run(NULL, 1, NULL, NULL) == BQ_ERR_BADARG, 16);
    EXPECT(bq_ctx_info(NULL, NULL, NULL, NULL, NULL, 0) == BQ_ERR_BADARG, 17);
    EXPECT(bq_ctx_comm_info(NULL, NULL, NULL, NULL) == BQ_ERR_BADARG, 18);
    EXPECT(bq_comm_unique_id(NULL) == BQ_ERR_BADARG, 19);
    EXPECT(bq_problem_destroy(NULL) == BQ_OK && bq_solver_destroy(NULL) == BQ_OK && bq_ctx_destroy(NULL) == BQ_OK &&
           bq_smo_destroy(NULL) == BQ_OK, 20);
    {   /* without a GPU the device paths fail with a message; with one they work — either way no sanitizer finding */
        int ndev = -1;
        bq_ctx *ctx = NULL;
        const int rc = bq_device_count(&ndev);
        if (rc != BQ_OK || ndev == 0) {
            EXPECT(bq_ctx_create(0, &ctx) != BQ_OK && ctx == NULL && strlen(bq_last_error()) > 0, 21);
            EXPECT(bq_ctx_create_exchange(0, 0, 1, NULL, NULL, &ctx) != BQ_OK, 22);
        } else {
            char name[64];
            EXPECT(bq_ctx_create(0, &ctx) == BQ_OK && ctx != NULL, 23);
            EXPECT(bq_ctx_info(ctx, NULL, NULL, NULL, name, sizeof(name)) == BQ_OK && strlen(name) > 0, 24);
            EXPECT(bq_ctx_destroy(ctx) == BQ_OK, 25);
        }
        EXPECT(bq_ctx_create(-1, &ctx) != BQ_OK, 26);
    }
    printf("host_checks ok\n");
    return 0;
}
