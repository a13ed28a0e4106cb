// Where does the trailing update lose its MFMA cycles?  The library's tile kernel with parts switched off (calibration tool,
// not part of the library).  Output: one line per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../optiml_amd/csrc/bq_mfma_tile.h"
constexpr int NB = 128;
template <int VAR>
__global__ __launch_bounds__(256, 2) void syrk_var(double *__restrict__ H, int64_t ldh, const double *__restrict__ Wt, int kdim, int64_t T) {
    __shared__ __attribute__((aligned(16))) bq_tile_smem sm;
    __shared__ double pad[(VAR == 3 || VAR == 4) ? 10240 : 1];   // VAR 3 / 4: 80 KB more LDS -> one workgroup per CU
    if ((VAR == 3 || VAR == 4) && threadIdx.x == 9999) pad[0] = 1.0;
    const int64_t bid = blockIdx.x;
    int64_t ti = (int64_t)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    while (ti * (ti + 1) / 2 > bid) --ti;
    const int64_t tj = bid - ti * (ti + 1) / 2;
    if (ti >= T) return;
    const int64_t arow = ti * NB, bcol = tj * NB;
    bq_d4 acc[4][4];
    double *Ct = H + arow * ldh + bcol;
    if (VAR == 0 || VAR == 3) bq_tile_load(acc, Ct, ldh); else bq_tile_zero(acc);   // VAR 5: C is read after the loop (the shipped form)
    if (VAR == 2) {   // no global traffic in the loop: stage one chunk, then run the same number of MFMA chunks on it
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1, fr = lane & 15, fk = lane >> 4;
        for (int e = tid; e < 16 * BQ_GP; e += 256) { (&sm.A[0][0][0])[e] = 1e-3 * e; (&sm.B[0][0][0])[e] = 1e-3; }
        __syncthreads();
        for (int c = 0; c < kdim / 16; ++c) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double a[4], b[4];
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    const bq_d2 va = *reinterpret_cast<const bq_d2 *>(&sm.A[0][kk * 4 + fk][wr * 64 + tp * 32 + 2 * fr]);
                    const bq_d2 vb = *reinterpret_cast<const bq_d2 *>(&sm.B[0][kk * 4 + fk][wc * 64 + tp * 32 + 2 * fr]);
                    a[2 * tp] = va.x; a[2 * tp + 1] = va.y; b[2 * tp] = vb.x; b[2 * tp + 1] = vb.y;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    } else if (VAR == 5) {
        bq_mfma_tile_128<false>(Wt, ldh, arow, Wt, ldh, bcol, kdim, sm, acc);
    } else {
        bq_mfma_tile_128<true>(Wt, ldh, arow, Wt, ldh, bcol, kdim, sm, acc);
    }
    if (VAR == 5) bq_tile_sub_store(acc, Ct, ldh);
    else if (VAR == 0 || VAR == 3) bq_tile_store(acc, Ct, ldh);
    else {
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
        if (sum == 12345.678) Ct[0] = sum;
    }
}
template <int VAR> void run(const char *what, double *H, int64_t ldh, double *Wt, int kdim, int64_t T) {
    const int64_t tiles = T * (T + 1) / 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    syrk_var<VAR><<<(unsigned)tiles, 256>>>(H, ldh, Wt, kdim, T);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) syrk_var<VAR><<<(unsigned)tiles, 256>>>(H, ldh, Wt, kdim, T);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)tiles * 2.0 * 128 * 128 * kdim * reps;
    printf("%-58s T=%lld K=%d: %8.3f ms/launch  %6.2f TFLOP/s  (%.1f us per tile-slot)\n", what, (long long)T, kdim, ms / reps, fl / (ms * 1e-3) / 1e12,
           ms / reps * 1e3 / ((double)tiles / 512.0));
}
int main(int argc, char **argv) {
    const int64_t T = argc > 1 ? atoll(argv[1]) : 256;
    const int64_t n = T * NB, ldh = n;
    double *H, *Wt;
    hipMalloc(&H, sizeof(double) * n * ldh);
    hipMalloc(&Wt, sizeof(double) * 1024 * ldh);
    hipMemset(H, 0, sizeof(double) * n * ldh);
    std::vector<double> w((size_t)1024 * ldh);
    for (size_t i = 0; i < w.size(); ++i) w[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    hipMemcpy(Wt, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice);
    for (int kdim : {512, 768, 1024}) {
        run<0>("acc = C, K loop with negated A, C store (rounds 1-2a)", H, ldh, Wt, kdim, T);
        run<1>("no C traffic (acc = 0, no store)", H, ldh, Wt, kdim, T);
        run<2>("no global traffic at all (LDS reads + MFMA + barriers)", H, ldh, Wt, kdim, T);
        run<3>("full, ONE workgroup per CU", H, ldh, Wt, kdim, T);
        run<4>("no C traffic, ONE workgroup per CU", H, ldh, Wt, kdim, T);
        run<5>("C read after the loop (C -= acc, no negation in the loop)", H, ldh, Wt, kdim, T);
    }
    return 0;
}
