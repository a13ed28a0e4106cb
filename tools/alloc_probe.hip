// How long does the first large device allocation of a process take, and does it depend on the API?  (calibration tool)
//   alloc_probe [GiB]   -> one line per variant; every variant also touches the memory with a fill kernel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(double *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main(int argc, char **argv) {
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 40;
    const size_t bytes = gib << 30;
    double t = now();
    hipFree(nullptr);
    printf("runtime init                         %.3f s\n", now() - t);
    void *p = nullptr;
    t = now();
    hipError_t e = hipMalloc(&p, bytes);
    printf("first hipMalloc of %zu GiB            %.3f s (%s)\n", gib, now() - t, hipGetErrorString(e));
    t = now();
    fill<<<2048, 256>>>((double *)p, bytes / 8);
    hipDeviceSynchronize();
    printf("first touch (fill kernel)            %.3f s\n", now() - t);
    t = now();
    fill<<<2048, 256>>>((double *)p, bytes / 8);
    hipDeviceSynchronize();
    printf("second fill                          %.3f s\n", now() - t);
    t = now();
    hipFree(p);
    printf("hipFree                              %.3f s\n", now() - t);
    t = now();
    e = hipMalloc(&p, bytes);
    printf("second hipMalloc (after the free)    %.3f s (%s)\n", now() - t, hipGetErrorString(e));
    hipFree(p);
    hipStream_t s;
    hipStreamCreate(&s);
    t = now();
    e = hipMallocAsync(&p, bytes, s);
    hipStreamSynchronize(s);
    printf("hipMallocAsync + sync                %.3f s (%s)\n", now() - t, hipGetErrorString(e));
    if (e == hipSuccess) {
        t = now();
        fill<<<2048, 256, 0, s>>>((double *)p, bytes / 8);
        hipStreamSynchronize(s);
        printf("first touch after hipMallocAsync     %.3f s\n", now() - t);
        hipFreeAsync(p, s);
        hipStreamSynchronize(s);
    }
    t = now();
    e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocDefault);
    printf("hipExtMallocWithFlags(default)       %.3f s (%s)\n", now() - t, hipGetErrorString(e));
    if (e == hipSuccess) hipFree(p);
    // many smaller pieces
    t = now();
    void *q[64];
    const size_t piece = bytes / 64;
    for (int i = 0; i < 64; ++i) hipMalloc(&q[i], piece);
    printf("64 hipMalloc of 1/64 each            %.3f s\n", now() - t);
    for (int i = 0; i < 64; ++i) hipFree(q[i]);
    return 0;
}
