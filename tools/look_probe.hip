// What does one look of the host at the device cost?  (dense ActiveSet looks three times per iteration, bq_as.hip.)
//   hipcc --offload-arch=gfx950 -O2 -o build/look_probe tools/look_probe.hip && build/look_probe [cycles]
// A cycle = a short chain of small kernels (the work between two looks) whose last kernel leaves 32 ints for the host, then the look:
//   A  hipMemcpyAsync(pinned <- device, 128 B) + hipStreamSynchronize                      (what bq_as.hip did through round 4)
//   B  the last kernel stores the 32 ints and then a sequence number straight into mapped pinned memory (system-scope fence between);
//      the host spins on the sequence number                                                (no copy, no stream drain, no interrupt)
//   C  as B, but the host waits with hipStreamSynchronize                                   (the copy taken out, the drain kept)
// Prints wall microseconds per cycle for each and the same chain without any look (enqueue-only floor).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            return 1;                                                            \
        }                                                                        \
    } while (0)

__global__ void work_kernel(double *v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] * 1.0000001 + 1e-9;
}

__global__ void last_kernel(double *v, int n, int *ints, int *mapped, int seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] * 0.9999999;
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        ints[threadIdx.x] = seq + threadIdx.x;
        if (mapped != nullptr) {
            mapped[threadIdx.x] = seq + threadIdx.x;
            __threadfence_system();
            if (threadIdx.x == 0) __hip_atomic_store(mapped + 32, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int main(int argc, char **argv) {
    const int cycles = argc > 1 ? atoi(argv[1]) : 3000;
    const int n = 20000;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double *v;
    int *ints, *pinned, *mapped;
    CK(hipMalloc(&v, sizeof(double) * n));
    CK(hipMemset(v, 0, sizeof(double) * n));
    CK(hipMalloc(&ints, sizeof(int) * 64));
    CK(hipHostMalloc(&pinned, sizeof(int) * 64));
    CK(hipHostMalloc(&mapped, sizeof(int) * 64, hipHostMallocMapped | hipHostMallocCoherent));
    for (int i = 0; i < 64; ++i) mapped[i] = pinned[i] = 0;
    const dim3 grid((n + 255) / 256), block(256);
    volatile int *flag = mapped + 32;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {   // rep 0 warms up
            CK(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            long long spins = 0;
            for (int c = 1; c <= cycles; ++c) {
                const int seq = (mode * 2 + rep) * 1000000 + c * 40;
                for (int k = 0; k < 3; ++k) work_kernel<<<grid, block, 0, st>>>(v, n);
                last_kernel<<<grid, block, 0, st>>>(v, n, ints, mode == 0 || mode == 3 ? nullptr : mapped, seq);
                if (mode == 0) {
                    CK(hipMemcpyAsync(pinned, ints, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
                    CK(hipStreamSynchronize(st));
                    if (pinned[5] != seq + 5) {
                        fprintf(stderr, "A: stale\n");
                        return 1;
                    }
                } else if (mode == 1) {
                    while (*flag != seq) {
                        if ((++spins & 0xfffff) == 0 && hipStreamQuery(st) == hipSuccess && *flag != seq) {
                            fprintf(stderr, "B: the stream drained and the flag never came\n");
                            return 1;
                        }
                    }
                    if (((volatile int *)mapped)[5] != seq + 5) {
                        fprintf(stderr, "B: stale\n");
                        return 1;
                    }
                } else if (mode == 2) {
                    CK(hipStreamSynchronize(st));
                    if (((volatile int *)mapped)[5] != seq + 5) {
                        fprintf(stderr, "C: stale\n");
                        return 1;
                    }
                }
            }
            CK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / cycles;
            if (rep == 1)
                printf("%s  %8.2f us per cycle (4 small kernels + the look)\n",
                       mode == 0   ? "A  memcpyAsync D2H + hipStreamSynchronize      "
                       : mode == 1 ? "B  stores to mapped pinned memory + host spin  "
                       : mode == 2 ? "C  stores to mapped pinned memory + stream sync"
                                   : "-  no look (enqueue only)                      ",
                       us);
        }
    }
    return 0;
}
