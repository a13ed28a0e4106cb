// placeholder until the Cholesky path lands
#include "bq_common.h"
int bq_ip_start(bq_solver *) { bq_set_error("InteriorPoint not built yet"); return BQ_ERR_BADARG; }
int bq_ip_iterate(bq_solver *) { bq_set_error("InteriorPoint not built yet"); return BQ_ERR_BADARG; }
int bq_as_start(bq_solver *) { bq_set_error("ActiveSet not built yet"); return BQ_ERR_BADARG; }
int bq_as_iterate(bq_solver *) { bq_set_error("ActiveSet not built yet"); return BQ_ERR_BADARG; }
int bq_chol_ws_create(bq_ctx *, int64_t, bq_chol_ws **out) { *out = nullptr; return BQ_OK; }
void bq_chol_ws_destroy(bq_chol_ws *) {}
