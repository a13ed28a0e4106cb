// 128x128 fp64 MFMA tile product shared by the Gram build, the Cholesky TRSM-as-GEMM and the SYRK update.
//
//   acc[i][j] (+)= sum_k A[arow + .][k] * B[bcol + .][k]      (both operands "row x k", i.e. C = A * B^T)
//
// Operands are k-major, zero-padded images: At[k][row] with pitch lda, Bt[k][row] with pitch ldb, so one k-slice
// of a 128-row tile is 1 KiB contiguous (coalesced 16-byte loads) and the LDS image [k][row] needs no transpose.
// 256 threads = 4 waves in a 2x2 arrangement; each wave owns a 64x64 block as 4x4 v_mfma_f64_16x16x4_f64 tiles
// (64 fp64 accumulators = 128 VGPRs per lane).  K advances in chunks of 16 through double-buffered LDS; the row
// pitch of 144 doubles puts the k-slices that one ds_read_b64 touches (lanes 0-15 / 16-31 of a 32-lane group) on
// disjoint bank halves.  Lane maps of v_mfma_f64_16x16x4_f64: A/B operand: row = lane & 15, k = lane >> 4;
// C/D: col = lane & 15, row = (lane >> 4) + 4 * reg.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef double bq_d2 __attribute__((ext_vector_type(2)));
typedef double bq_d4 __attribute__((ext_vector_type(4)));

constexpr int BQ_GT = 128;          // tile edge
constexpr int BQ_GK = 16;           // k-chunk
constexpr int BQ_GP = BQ_GT + 16;   // LDS row pitch (doubles)

struct bq_tile_smem {
    double A[2][BQ_GK][BQ_GP];
    double B[2][BQ_GK][BQ_GP];
};

// kdim must be a multiple of 16; arow/bcol multiples of 2 with arow+127 < lda, bcol+127 < ldb.
// NEG_A: accumulate -A*B^T (the A slice is negated while it is staged), so that a kernel can start from acc = C and
// finish with plain stores instead of a serialised read-modify-write epilogue.
// A_ROWS: the A operand is NOT an image but the row-major matrix itself, element (arow + r, k) at At[(arow + r) * lda + k]
// (16 contiguous doubles per row and chunk); it is transposed on its way into LDS.  Saves the transposing pre-pass for
// operands that are consumed once (the TRSM input).
template <bool NEG_A = false, bool A_ROWS = false>
__device__ __forceinline__ void bq_mfma_tile_128(const double *__restrict__ At, int64_t lda, int64_t arow,
                                                 const double *__restrict__ Bt, int64_t ldb, int64_t bcol,
                                                 int64_t kdim, bq_tile_smem &sm, bq_d4 (&acc)[4][4]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    bq_d2 ra[4], rb[4];
    auto gload = [&](int64_t kc) {   // chunk kc / 16 of both operands -> the staging registers
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = tid + 256 * u;
            const int k = j >> 6, c2 = j & 63;
            if (A_ROWS)
                ra[u] = *reinterpret_cast<const bq_d2 *>(At + (arow + (j >> 3)) * lda + kc + 2 * (j & 7));
            else
                ra[u] = *reinterpret_cast<const bq_d2 *>(At + (kc + k) * lda + arow + 2 * c2);
            rb[u] = *reinterpret_cast<const bq_d2 *>(Bt + (kc + k) * ldb + bcol + 2 * c2);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = tid + 256 * u;
            const int k = j >> 6, c2 = j & 63;
            const bq_d2 va = NEG_A ? -ra[u] : ra[u];
            if (A_ROWS) {
                sm.A[buf][2 * (j & 7)][j >> 3] = va.x;
                sm.A[buf][2 * (j & 7) + 1][j >> 3] = va.y;
            } else {
                *reinterpret_cast<bq_d2 *>(&sm.A[buf][k][2 * c2]) = va;
            }
            *reinterpret_cast<bq_d2 *>(&sm.B[buf][k][2 * c2]) = rb[u];
        }
    };
    const int64_t nchunks = kdim / BQ_GK;
    const int fr = lane & 15, fk = lane >> 4;
    // fragments of one k-slice of 4 (this lane: 4 A rows, 4 B rows), two register sets so that the reads of slice s+1 are in
    // flight while the 16 MFMAs of slice s issue
    double fa[2][4], fb[2][4];
    auto rd = [&](int set, int buf, int kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[set][t] = sm.A[buf][kk * 4 + fk][wr * 64 + t * 16 + fr];
            fb[set][t] = sm.B[buf][kk * 4 + fk][wc * 64 + t * 16 + fr];
        }
    };
    auto mm = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
    // The chunk loop is software-pipelined ACROSS the chunk boundary so that the matrix pipe always has MFMAs to issue while
    // the staging happens (tools/syrk_probe.hip: with "64 MFMAs, then stage, barrier, first reads" per chunk the pipe idled
    // ~10 % — LDS-write completion + barrier + first-read latency at every boundary — and the co-resident workgroup, in
    // lockstep, did not fill it).  Per chunk c (LDS buffer B = c & 1, O the other):
    //   slice 0: MFMAs (fragments pre-read at the end of chunk c-1)                    | reads of slice 1
    //   slice 1: MFMAs | registers (chunk c+1, loaded during chunk c-1) -> LDS O, then the global loads of chunk c+2
    //   slice 2: MFMAs | reads of slice 3 ; lgkmcnt(0) ; BARRIER — every LDS read of B and every write of O is done
    //   slice 3: MFMAs (operands in registers)                                          | reads of slice 0 of chunk c+1 from O
    // One barrier per chunk.  O may be overwritten during chunk c because every wave finished reading it before the barrier
    // of chunk c-1 (its slice-3 fragments were in registers by then); B may be read in chunk c because its writes preceded
    // that same barrier.  The global prefetch stays in flight across the barrier (plain loads: the fence of
    // __syncthreads() waits for LDS only) and has a whole chunk to land.  No load in the steady-state body is conditional
    // (the last two chunks are peeled): with a conditional prefetch hipcc's s_waitcnt pass joined an "issued" and a "not
    // issued" path and waited vmcnt(3..0) in front of the chunk's first MFMAs (round 1).
    gload(0);
    lstore(0);
    if (nchunks > 1) gload(BQ_GK);
    __syncthreads();
    rd(0, 0, 0);
    int64_t c = 0;
    // sched_group_barrier pins the interleave inside each region (masks: MFMA 0x8, VALU 0x2, VMEM read 0x20, DS read 0x100,
    // DS write 0x200): the next slice's four ds_read2 go out first, the staging is threaded through the MFMAs one
    // instruction per MFMA — left alone, hipcc put all reads, writes and loads of a region in front of its MFMAs.
    for (; c + 2 < nchunks; ++c) {
        const int B = (int)(c & 1), O = B ^ 1;
        rd(1, B, 1);
        mm(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd(0, B, 2);
        mm(1);
        lstore(O);
        gload((c + 2) * BQ_GK);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        rd(1, B, 3);
        mm(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
        __builtin_amdgcn_sched_barrier(0);   // the barrier stays BEHIND slice 2's MFMAs: its wait is covered by them
        __syncthreads();
        rd(0, O, 0);
        mm(1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < nchunks) {   // second-to-last chunk: its successor waits in the registers, nothing left to prefetch
        const int B = (int)(c & 1), O = B ^ 1;
        rd(1, B, 1);
        mm(0);
        rd(0, B, 2);
        mm(1);
        lstore(O);
        rd(1, B, 3);
        mm(0);
        __syncthreads();
        rd(0, O, 0);
        mm(1);
        ++c;
    }
    {   // last chunk
        const int B = (int)(c & 1);
        rd(1, B, 1);
        mm(0);
        rd(0, B, 2);
        mm(1);
        rd(1, B, 3);
        mm(0);
        mm(1);
    }
}

// Load / store a whole accumulator tile from / to a row-major matrix (pitch ld): 64 independent 8-byte accesses per lane,
// issued back to back (no read-after-write chain between them).
__device__ __forceinline__ void bq_tile_load(bq_d4 (&acc)[4][4], const double *__restrict__ C, int64_t ld) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1, ccol = lane & 15, crow = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                acc[i][j][v] = C[(int64_t)(wr * 64 + i * 16 + crow + 4 * v) * ld + wc * 64 + j * 16 + ccol];
}
__device__ __forceinline__ void bq_tile_store(const bq_d4 (&acc)[4][4], double *__restrict__ C, int64_t ld) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1, ccol = lane & 15, crow = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                C[(int64_t)(wr * 64 + i * 16 + crow + 4 * v) * ld + wc * 64 + j * 16 + ccol] = acc[i][j][v];
}

// Visit every accumulator element of this lane: f(row_in_tile, col_in_tile, value)
template <typename F>
__device__ __forceinline__ void bq_tile_foreach(bq_d4 (&acc)[4][4], F &&f) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int ccol = lane & 15, crow = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) f(wr * 64 + i * 16 + crow + 4 * v, wc * 64 + j * 16 + ccol, acc[i][j][v]);
}

__device__ __forceinline__ void bq_tile_zero(bq_d4 (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (bq_d4){0.0, 0.0, 0.0, 0.0};
}
