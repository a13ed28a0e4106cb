// Device allocation with the retry the panel cache needs (see bq_common.h): one place, used by every hipMalloc of the library.
#define BQ_NO_MALLOC_REDIRECT
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "bq_common.h"

namespace {
std::mutex g_mu;
std::vector<bq_ctx *> g_live;   // contexts that may hold a cached panel
}  // namespace

void bq_ctx_register(bq_ctx *c, bool alive) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = std::find(g_live.begin(), g_live.end(), c);
    if (alive && it == g_live.end()) g_live.push_back(c);
    if (!alive && it != g_live.end()) g_live.erase(it);
}

hipError_t bq_device_malloc(void **ptr, size_t bytes) {
    // BQ_TEST_ALLOC_FAIL_ABOVE=<bytes>: test hook — requests above the threshold fail ONCE per request while a cached
    // panel exists, so that the drop-and-retry path is exercised without exhausting a 288 GB device
    static const long long fail_above = [] {
        const char *e = getenv("BQ_TEST_ALLOC_FAIL_ABOVE");
        return e ? atoll(e) : -1ll;
    }();
    hipError_t e = hipSuccess;
    bool simulated = false;
    if (fail_above >= 0 && (long long)bytes > fail_above) {
        std::lock_guard<std::mutex> lk(g_mu);
        for (bq_ctx *c : g_live) simulated = simulated || c->panel_cache != nullptr;
    }
    if (simulated)
        e = hipErrorOutOfMemory;
    else
        e = hipMalloc(ptr, bytes);
    if (e == hipSuccess) return e;
    bool dropped = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (bq_ctx *c : g_live)
            if (c->panel_cache) {
                int dev = 0;
                hipGetDevice(&dev);
                hipSetDevice(c->device);
                bq_ctx_drop_cache(c);
                hipSetDevice(dev);
                dropped = true;
            }
    }
    // a failed hipMalloc leaves its error pending: the next hipGetLastError() of an unrelated call would report "out of
    // memory" (seen: after a panel that does not fit, the next — small — problem failed once).  The caller gets the code.
    (void)hipGetLastError();
    if (!dropped) return e;
    e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}
