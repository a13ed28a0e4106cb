// Can vector instructions run beside a dense stream of matrix instructions?  (calibration tool, not part of the library)
// 1. Other wave, same SIMD: one workgroup of 512 threads per CU = two waves per SIMD; waves 0-3 issue back-to-back
//    v_mfma_f64_16x16x4_f64 (or v_mfma_f32_32x32x2_f32), waves 4-7 run dependent vector chains of the chosen kind until the
//    matrix waves are done.  Prints the matrix waves' rate and how many vector instructions the other wave got through.
// 2. Same wave: k independent vector instructions pinned after every MFMA; prints the MFMA period.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
// the same experiment with fp32 matrix instructions (v_mfma_f32_32x32x2_f32, 16 passes like the fp64 one) as the contrast
template <int MODE, int CHAINS>
__global__ __launch_bounds__(512, 1) void k32(double *sink, int iters, float a0, float b0, unsigned long long *clk, long long *vops) {
    __shared__ volatile int finished;
    if (threadIdx.x == 0) finished = 0;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    if (wv < 4) {
        f16v acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        float a = a0 * (1.f + 1e-3f * (float)(threadIdx.x & 15)), b = b0 * (1.f - 1e-3f * (float)(threadIdx.x >> 4));
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][e];
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        if (s == 12345.678f) *sink = s;
        if ((threadIdx.x & 63) == 0) atomicAdd((int *)&finished, 1);
        if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
    } else if (MODE > 0) {
        double x[CHAINS];
        float f[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) { x[c] = a0 + c; f[c] = b0 + c; }
        long long n = 0;
        while (finished < 4) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) {
                    if (MODE == 1) x[c] = __builtin_fma(x[c], 0.999, 1e-3);
                    if (MODE == 2) f[c] = __builtin_fmaf(f[c], 0.999f, 1e-3f);
                }
            }
            n += 32 * CHAINS;
        }
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) s += x[c] + f[c];
        if (s == 12345.678) *sink = s;
        if (threadIdx.x == 256 && blockIdx.x == 0) *vops = n;
    }
}
template <int MODE, int CHAINS>
void run32(const char *what) {
    double *sink; unsigned long long *clk, h; long long *vops, hv = 0;
    hipMalloc(&sink, 8); hipMalloc(&clk, 8); hipMalloc(&vops, 8); hipMemset(vops, 0, 8);
    const int iters = 8192;
    for (int w = 0; w < 5; ++w) k32<MODE, CHAINS><<<256, 512>>>(sink, iters, 0.5f, 1.0f, clk, vops);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) k32<MODE, CHAINS><<<256, 512>>>(sink, iters, 0.5f, 1.0f, clk, vops);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); hipMemcpy(&hv, vops, 8, hipMemcpyDeviceToHost);
    const double tf = 256.0 * 4 * iters * 8 * 4096.0 * reps / (ms * 1e-3) / 1e12;   // 32 x 32 x 2 x 2 flop per instruction
    const double us = (double)h / 100.0;
    printf("fp32 MFMA %-34s chains %d : matrix waves %6.2f TFLOP/s (%.0f us in-kernel), VALU ops per lane of the other wave %lld (%.1f ns each)\n",
           what, CHAINS, tf, us, hv, hv ? us * 1e3 / (double)hv : 0.0);
    hipFree(sink); hipFree(clk); hipFree(vops);
}
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

// Same wave: VPM independent vector instructions after every matrix instruction (pinned with sched_group_barrier).  If they
// issue in the shadow of the MFMA's passes the matrix rate does not move; if the MFMA occupies the vector ALU it drops by
// VPM * 4 (or 8) cycles per 64.
template <int KIND, int VK, int VPM>
__global__ __launch_bounds__(256, 2) void ksame(double *sink, int iters, double a0, double b0) {
    typedef typename std::conditional<KIND == 0, d4, f16v>::type acc_t;
    constexpr int NACC = KIND == 0 ? 16 : 8, NE = KIND == 0 ? 4 : 16;
    acc_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[i][e] = 0;
    double a = a0 * (1.0 + 1e-3 * (double)(threadIdx.x & 15)), b = b0 * (1.0 - 1e-3 * (double)(threadIdx.x >> 4));
    float fa = (float)a, fb = (float)b;
    double x[8];
    float f[8];
    unsigned u[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { x[c] = a0 + c; f[c] = (float)b0 + c; u[c] = threadIdx.x + c; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            else acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i & 7], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VPM; ++v) {
                const int c = (i * VPM + v) & 7;
                if (VK == 0) x[c] = __builtin_fma(x[c], 0.999, 1e-3);
                if (VK == 1) f[c] = __builtin_fmaf(f[c], 0.999f, 1e-3f);
                if (VK == 2) u[c] = u[c] * 1664525u + 1013904223u;
            }
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (VPM) __builtin_amdgcn_sched_group_barrier(0x2, VPM, 0);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < NE; ++e) s += acc[i][e];
#pragma unroll
    for (int c = 0; c < 8; ++c) s += x[c] + f[c] + u[c];
    if (s == 12345.678) *sink = s;
}
template <int KIND, int VK, int VPM>
void runsame(const char *what) {
    double *sink; hipMalloc(&sink, 8);
    const int iters = 4096;
    for (int w = 0; w < 3; ++w) ksame<KIND, VK, VPM><<<256, 256>>>(sink, iters, 0.5, 1.0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) ksame<KIND, VK, VPM><<<256, 256>>>(sink, iters, 0.5, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = KIND == 0 ? 2048.0 : 4096.0;
    const double tf = 256.0 * 4 * iters * 16 * per * reps / (ms * 1e-3) / 1e12;
    const double cyc = ms * 1e-3 / reps * 2.4e9 / ((double)iters * 16);
    printf("same wave, %s MFMA + %d %s per MFMA: %7.2f TFLOP/s  (%.1f cycles per MFMA at 2.4 GHz)\n", KIND == 0 ? "fp64" : "fp32", VPM, what, tf, cyc);
    hipFree(sink);
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
    runsame<0, 0, 0>("(none)");
    runsame<0, 0, 1>("fp64 FMA");
    runsame<0, 0, 2>("fp64 FMA");
    runsame<0, 0, 4>("fp64 FMA");
    runsame<0, 1, 2>("fp32 FMA");
    runsame<0, 1, 4>("fp32 FMA");
    runsame<0, 2, 4>("int32 mul-add");
    runsame<1, 0, 0>("(none)");
    runsame<1, 0, 2>("fp64 FMA");
    runsame<1, 0, 4>("fp64 FMA");
    runsame<1, 1, 4>("fp32 FMA");
    runsame<1, 1, 8>("fp32 FMA");
    runsame<1, 2, 4>("int32 mul-add");
    run32<0, 1>("alone");
    run32<1, 4>("+ fp64 FMA chains");
    run32<2, 4>("+ fp32 FMA chains");
    return 0;
}
