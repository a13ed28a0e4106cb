// Do fp64 VALU instructions of one wave slow the fp64 MFMAs of another wave on the same SIMD?  (not part of the library)
// One workgroup of 512 threads per CU = two waves per SIMD: waves 0-3 issue back-to-back v_mfma_f64_16x16x4_f64, waves 4-7 run
// a dependent VALU chain of the chosen kind until the matrix waves are done.  Prints the matrix waves' rate per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE, int CHAINS, int PRIO>
__global__ __launch_bounds__(512, 1) void k(double *sink, int iters, double a0, double b0, unsigned long long *clk, long long *vops) {
    __shared__ volatile int finished;
    if (threadIdx.x == 0) finished = 0;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    if (wv < 4) {
        d4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
        double a = a0 * (1.0 + 1e-3 * (double)(threadIdx.x & 15)), b = b0 * (1.0 - 1e-3 * (double)(threadIdx.x >> 4));
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (s == 12345.678) *sink = s;
        if ((threadIdx.x & 63) == 0) atomicAdd((int *)&finished, 1);
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    } else if (MODE > 0) {
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        double x[CHAINS];
        float f[CHAINS];
        unsigned u[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) { x[c] = a0 + c; f[c] = (float)b0 + c; u[c] = threadIdx.x + c; }
        long long n = 0;
        while (finished < 4) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) {
                    if (MODE == 1) x[c] = __builtin_fma(x[c], 0.999, 1e-3);
                    if (MODE == 2) f[c] = __builtin_fmaf(f[c], 0.999f, 1e-3f);
                    if (MODE == 3) u[c] = u[c] * 1664525u + 1013904223u;
                    if (MODE == 4) x[c] = x[c] + 1e-3;
                }
            }
            n += 32 * CHAINS;
        }
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) s += x[c] + f[c] + u[c];
        if (s == 12345.678) *sink = s;
        if (threadIdx.x == 256 && blockIdx.x == 0) *vops = n;
    }
}
template <int MODE, int CHAINS, int PRIO = 0>
void run(const char *what) {
    double *sink; unsigned long long *clk, h; long long *vops, hv = 0;
    hipMalloc(&sink, 8); hipMalloc(&clk, 8); hipMalloc(&vops, 8); hipMemset(vops, 0, 8);
    const int iters = 4096;
    for (int w = 0; w < 5; ++w) k<MODE, CHAINS, PRIO><<<256, 512>>>(sink, iters, 0.5, 1.0, clk, vops);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) k<MODE, CHAINS, PRIO><<<256, 512>>>(sink, iters, 0.5, 1.0, clk, vops);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); hipMemcpy(&hv, vops, 8, hipMemcpyDeviceToHost);
    const double tf = 256.0 * 4 * iters * 16 * 2048.0 * reps / (ms * 1e-3) / 1e12;
    const double us = (double)h / 100.0;
    printf("%-44s chains %d : matrix waves %6.2f TFLOP/s (%.0f us in-kernel), VALU ops per lane of the other wave %lld (%.1f ns each)\n",
           what, CHAINS, tf, us, hv, hv ? us * 1e3 / (double)hv : 0.0);
    hipFree(sink); hipFree(clk); hipFree(vops);
}
int main() {
    run<0, 1>("matrix waves alone");
    run<1, 1>("+ fp64 FMA chain");
    run<1, 4>("+ fp64 FMA chains");
    run<1, 8>("+ fp64 FMA chains");
    run<4, 4>("+ fp64 ADD chains");
    run<2, 4>("+ fp32 FMA chains");
    run<3, 4>("+ int32 mul-add chains");
    run<1, 1, 3>("+ fp64 FMA chain, s_setprio 3");
    run<1, 4, 3>("+ fp64 FMA chains, s_setprio 3");
    run<1, 8, 3>("+ fp64 FMA chains, s_setprio 3");
    run<2, 4, 3>("+ fp32 FMA chains, s_setprio 3");
    run<3, 4, 3>("+ int32 mul-add chains, s_setprio 3");
    run<1, 4, 1>("+ fp64 FMA chains, s_setprio 1");
    return 0;
}
