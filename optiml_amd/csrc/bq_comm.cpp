// Row-block exchange for multi-GPU runs: one all-gather of the panel-product slices per Q*v.
//
// RCCL (librccl.so, "nccl" on ROCm) is loaded lazily with dlopen so that single-GPU use never pays for it and a
// host without RCCL fails with a clear message only when a multi-rank context is requested.  One process per GPU:
// rank r contributes rows [r*blk, (r+1)*blk) of the n-vector; the gather is in place
// (sendbuff == recvbuff + rank*blk), issued on the context's compute stream so it is ordered after the GEMV and
// before the consumers without host synchronisation.  Messages are tiny (<= 1 MB): latency-bound over xGMI.
#include <dlfcn.h>

#include "bq_common.h"

namespace {
typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
typedef int nccl_result_t;
constexpr int NCCL_FLOAT64 = 8;  // ncclDouble / ncclFloat64 in rccl.h's ncclDataType_t
constexpr int NCCL_SUM = 0;      // ncclSum in rccl.h's ncclRedOp_t

struct rccl_api {
    void *handle = nullptr;
    nccl_result_t (*GetUniqueId)(nccl_uid_t *) = nullptr;
    nccl_result_t (*CommInitRank)(nccl_comm_t *, int, nccl_uid_t, int) = nullptr;
    nccl_result_t (*CommDestroy)(nccl_comm_t) = nullptr;
    nccl_result_t (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    nccl_result_t (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(nccl_result_t) = nullptr;
};
rccl_api g_rccl;

int load_rccl() {
    if (g_rccl.handle) return BQ_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        bq_set_error("cannot load RCCL (librccl.so): %s", dlerror());
        return BQ_ERR_RCCL;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.AllReduce) {
        bq_set_error("librccl.so lacks a required symbol");
        dlclose(h);
        return BQ_ERR_RCCL;
    }
    g_rccl.handle = h;
    return BQ_OK;
}

int rccl_fail(const char *what, nccl_result_t r) {
    bq_set_error("%s failed: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?", r);
    return BQ_ERR_RCCL;
}
}  // namespace

extern "C" int bq_comm_unique_id(void *uid128) {
    BQ_ARG(uid128 != nullptr, "uid is NULL");
    BQ_TRY(load_rccl());
    nccl_uid_t id;
    nccl_result_t r = g_rccl.GetUniqueId(&id);
    if (r != 0) return rccl_fail("ncclGetUniqueId", r);
    memcpy(uid128, &id, sizeof(id));
    return BQ_OK;
}

int bq_comm_init_rccl(bq_ctx *ctx, const void *uid128) {
    BQ_TRY(load_rccl());
    nccl_uid_t id;
    memcpy(&id, uid128, sizeof(id));
    nccl_comm_t comm = nullptr;
    nccl_result_t r = g_rccl.CommInitRank(&comm, ctx->world, id, ctx->rank);
    if (r != 0) return rccl_fail("ncclCommInitRank", r);
    ctx->nccl_comm = comm;
    ctx->comm_kind = BQ_COMM_RCCL;
    return BQ_OK;
}

void bq_comm_destroy(bq_ctx *ctx) {
    if (ctx->nccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy((nccl_comm_t)ctx->nccl_comm);
    ctx->nccl_comm = nullptr;
}

// s holds world*blk doubles; rows [r0,r1) (this rank's block, r0 == rank*blk) are fresh on entry
int bq_exchange_rows(bq_ctx *ctx, double *s, int64_t n, int64_t blk, int64_t r0, int64_t r1) {
    if (ctx->comm_kind == BQ_COMM_NONE) return BQ_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_EXCH, &e0, &e1));
    if (ctx->comm_kind == BQ_COMM_RCCL) {
        nccl_result_t r = g_rccl.AllGather(s + (int64_t)ctx->rank * blk, s, (size_t)blk, NCCL_FLOAT64,
                                           (nccl_comm_t)ctx->nccl_comm, ctx->stream);
        if (r != 0) return rccl_fail("ncclAllGather", r);
    } else if (ctx->comm_kind == BQ_COMM_CALLBACK) {
        const size_t bytes = sizeof(double) * (size_t)n;
        if (ctx->pinned_cap < bytes) {
            if (ctx->pinned) hipHostFree(ctx->pinned);
            ctx->pinned = nullptr;
            BQ_HIP(hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
            ctx->pinned_cap = bytes;
        }
        if (r1 > r0)
            BQ_HIP(hipMemcpyAsync(ctx->pinned + r0, s + r0, sizeof(double) * (size_t)(r1 - r0), hipMemcpyDeviceToHost,
                                  ctx->stream));
        BQ_HIP(hipStreamSynchronize(ctx->stream));
        int rc = ctx->exch_fn(ctx->exch_user, ctx->pinned, n, r0, r1, 0);
        if (rc != 0) {
            bq_set_error("exchange callback returned %d", rc);
            return BQ_ERR_RCCL;
        }
        BQ_HIP(hipMemcpyAsync(s, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        bq_set_error("multi-rank context without an exchange");
        return BQ_ERR_BADARG;
    }
    BQ_TRY(bq_prof_end(ctx, BQ_PROF_EXCH, e0, e1));
    return BQ_OK;
}

// in-place all-reduce(sum) of v[0:count) — the exchange of the symmetric tile product (every rank holds partial
// sums for every output block)
int bq_exchange_sum(bq_ctx *ctx, double *v, int64_t count) {
    if (ctx->comm_kind == BQ_COMM_NONE) return BQ_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    BQ_TRY(bq_prof_begin(ctx, BQ_PROF_EXCH, &e0, &e1));
    if (ctx->comm_kind == BQ_COMM_RCCL) {
        nccl_result_t r = g_rccl.AllReduce(v, v, (size_t)count, NCCL_FLOAT64, NCCL_SUM, (nccl_comm_t)ctx->nccl_comm,
                                           ctx->stream);
        if (r != 0) return rccl_fail("ncclAllReduce", r);
    } else {
        const size_t bytes = sizeof(double) * (size_t)count;
        if (ctx->pinned_cap < bytes) {
            if (ctx->pinned) hipHostFree(ctx->pinned);
            ctx->pinned = nullptr;
            BQ_HIP(hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
            ctx->pinned_cap = bytes;
        }
        BQ_HIP(hipMemcpyAsync(ctx->pinned, v, bytes, hipMemcpyDeviceToHost, ctx->stream));
        BQ_HIP(hipStreamSynchronize(ctx->stream));
        int rc = ctx->exch_fn(ctx->exch_user, ctx->pinned, count, 0, count, 1);
        if (rc != 0) {
            bq_set_error("exchange callback returned %d", rc);
            return BQ_ERR_RCCL;
        }
        BQ_HIP(hipMemcpyAsync(v, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    BQ_TRY(bq_prof_end(ctx, BQ_PROF_EXCH, e0, e1));
    return BQ_OK;
}
