// Calibration of the fp64 matrix-core ceiling (not part of the library): back-to-back v_mfma_f64_16x16x4_f64 under
// different loads.  usage: mfma_probe  -> one line per variant: workgroups, waves/SIMD, data, TFLOP/s, effective MHz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void k(double *sink, int iters, double a0, double b0, unsigned long long *clk) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = a0 * (1.0 + 1e-3 * (double)(threadIdx.x & 15)), b = b0 * (1.0 - 1e-3 * (double)(threadIdx.x >> 4));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678) *sink = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NACC>
void run(int grid, int threads, double a, double b, const char *what) {
    double *sink; unsigned long long *clk, h[2];
    hipMalloc(&sink, 8); hipMalloc(&clk, 16);
    const int iters = 8192;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 40; ++w) k<NACC><<<grid, threads>>>(sink, iters, a, b, clk);
    hipEventRecord(e0);
    const int reps = 40;
    for (int w = 0; w < reps; ++w) k<NACC><<<grid, threads>>>(sink, iters, a, b, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double waves = (double)grid * threads / 64.0;
    const double tf = waves * iters * NACC * 2048.0 * reps / (ms * 1e-3) / 1e12;
    printf("%-34s grid %4d x %4d acc %2d : %7.2f TFLOP/s   in-kernel clock %6.0f MHz   cycles/MFMA/wave %.1f\n", what, grid, threads, NACC, tf,
           (double)h[0] / (double)h[1] * 100.0, (double)h[0] / ((double)iters * NACC));
    hipFree(sink); hipFree(clk);
}
int main() {
    run<16>(256, 256, 0.5, 1.0, "1 wave/SIMD, all CUs, random-ish");
    run<16>(512, 256, 0.5, 1.0, "2 waves/SIMD, all CUs, random-ish");
    run<16>(512, 256, 0.0, 0.0, "2 waves/SIMD, all CUs, zeros");
    run<16>(64, 256, 0.5, 1.0, "1 wave/SIMD, 64 CUs");
    run<16>(8, 256, 0.5, 1.0, "1 wave/SIMD, 8 CUs");
    run<4>(256, 256, 0.5, 1.0, "1 wave/SIMD, 4 accumulators");
    run<16>(1024, 256, 0.5, 1.0, "4 waves/SIMD, all CUs");
    return 0;
}
