// 128x128 fp64 MFMA tile product shared by the Gram build, the Cholesky TRSM-as-GEMM and the SYRK update.
//
//   acc[i][j] (+)= sum_k A[arow + .][k] * B[bcol + .][k]      (both operands "row x k", i.e. C = A * B^T)
//
// Operands are k-major, zero-padded images: At[k][row] with pitch lda, Bt[k][row] with pitch ldb, so one k-slice
// of a 128-row tile is 1 KiB contiguous (coalesced 16-byte loads) and the LDS image [k][row] needs no transpose.
// 256 threads = 4 waves in a 2x2 arrangement; each wave owns a 64x64 block as 4x4 v_mfma_f64_16x16x4_f64 tiles
// (64 fp64 accumulators = 128 VGPRs per lane).  K advances in chunks of 16 through double-buffered LDS.
// Lane maps of v_mfma_f64_16x16x4_f64: A/B operand: row = lane & 15, k = lane >> 4; C/D: col = lane & 15,
// row = (lane >> 4) + 4 * reg.
//
// Which 16 rows of a wave's 64 form MFMA row block t is free to choose; here block t = 2 tp + b holds the rows
//     w * 64 + tp * 32 + 2 m + b          (m = 0..15 the MFMA row index = lane & 15 of the operand)
// so that the four values a lane feeds to its four row blocks are two aligned 16-byte pairs of the plain [k][row] image:
// a k-slice of 4 is 2 + 2 ds_read_b128 per lane, every one of them at ONE per-lane base (per operand) + a 16-bit
// immediate — no address arithmetic in the loop (see below why that matters) and no family of base registers (with 16
// consecutive rows per block the reads were ds_read2_b64, whose offsets reach 2 KB: ~30 base registers once the addresses
// were loop-invariant, and they spilled).  Sixteen consecutive lanes read 256 contiguous bytes.  The accumulator element
// acc[i][j][v] of a lane is therefore tile element (bq_acc_row(i, v), bq_acc_col(j)) — j = 0, 1 (and 2, 3) are adjacent
// columns, so C tiles move as 16-byte accesses.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef double bq_d2 __attribute__((ext_vector_type(2)));
typedef double bq_d4 __attribute__((ext_vector_type(4)));

constexpr int BQ_GT = 128;          // tile edge
constexpr int BQ_GK = 16;           // k-chunk
#ifndef BQ_TILE_PITCH
#define BQ_TILE_PITCH 128
#endif
constexpr int BQ_GP = BQ_TILE_PITCH;   // LDS row pitch (doubles)

struct bq_tile_smem {
    double A[2][BQ_GK][BQ_GP];
    double B[2][BQ_GK][BQ_GP];
};
// row / column inside the wave's 64 x 64 block of accumulator element acc[i][j][v] of this lane
__device__ __forceinline__ int bq_acc_row64(int i, int v) { return (i >> 1) * 32 + 2 * ((int)((threadIdx.x & 63) >> 4) + 4 * v) + (i & 1); }
__device__ __forceinline__ int bq_acc_col64(int j) { return (j >> 1) * 32 + 2 * (int)(threadIdx.x & 15) + (j & 1); }
// ... and inside the 128 x 128 tile
__device__ __forceinline__ int bq_acc_row(int i, int v) { return (int)(threadIdx.x >> 7) * 64 + bq_acc_row64(i, v); }
__device__ __forceinline__ int bq_acc_col(int j) { return (int)((threadIdx.x >> 6) & 1) * 64 + bq_acc_col64(j); }

// kdim must be a multiple of 16; arow/bcol multiples of 2 with arow+127 < lda, bcol+127 < ldb.
// NEG_A: accumulate -A*B^T (the A slice is negated while it is staged), so that a kernel can start from acc = C and
// finish with plain stores instead of a serialised read-modify-write epilogue.
// A_ROWS: the A operand is NOT an image but the row-major matrix itself, element (arow + r, k) at At[(arow + r) * lda + k]
// (16 contiguous doubles per row and chunk); it is transposed on its way into LDS.  Saves the transposing pre-pass for
// operands that are consumed once (the TRSM input).
// Address arithmetic stays off the vector ALU inside the chunk loop.  Vector instructions are paid for in matrix-pipe time
// on this part (tools/coissue_probe.hip): a wave with independent MFMAs queued back to back leaves a co-resident wave of
// its SIMD ~0.1 % of its stand-alone vector issue rate (fp64 / fp32 / int32 chains alike, with or without s_setprio), and
// k vector instructions placed after each MFMA of the same wave lengthen its 65-cycle period by ~13 + 5 (k - 1) cycles.
// The round-2a loop spent 40 vector instructions per 64 MFMAs on the global and LDS addresses of the staging (5-7 % of the
// tile rate, tools/syrk_probe.hip).  Here the operands are fetched with raw buffer loads (per-lane offset fixed for the
// whole tile, the chunk advance is a scalar add on the resource's base) and the loop is unrolled over the two LDS buffers so
// that every LDS address is base + immediate: 2 vector instructions per 128 MFMAs in the steady state.
__device__ __forceinline__ const double *bq_uniform(const double *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const double *>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ bq_d2 bq_buffer_load_d2(const double *base, int voffset, int soffset) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, -1, 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
    return __builtin_bit_cast(bq_d2, v);
}

template <bool NEG_A = false, bool A_ROWS = false>
__device__ __forceinline__ void bq_mfma_tile_128(const double *__restrict__ At, int64_t lda, int64_t arow,
                                                 const double *__restrict__ Bt, int64_t ldb, int64_t bcol,
                                                 int64_t kdim, bq_tile_smem &sm, bq_d4 (&acc)[4][4]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    bq_d2 ra[4], rb[4];
    // this lane's byte offset inside a chunk (fixed), the four pieces u of a chunk as scalar offsets, the tile's origin
    const int voffA = A_ROWS ? (int)(((int64_t)(tid >> 3) * lda + 2 * (tid & 7)) * 8) : (int)(((int64_t)(tid >> 6) * lda + 2 * (tid & 63)) * 8);
    const int voffB = (int)(((int64_t)(tid >> 6) * ldb + 2 * (tid & 63)) * 8);
    const int stepA = __builtin_amdgcn_readfirstlane((int)((A_ROWS ? 32 : 4) * lda * 8));
    const int stepB = __builtin_amdgcn_readfirstlane((int)(4 * ldb * 8));
    const double *pA = bq_uniform(A_ROWS ? At + arow * lda : At + arow);
    const double *pB = bq_uniform(Bt + bcol);
    const int64_t advA = A_ROWS ? 1 : lda;
    auto gload = [&](int64_t kc) {   // chunk kc / 16 of both operands -> the staging registers
        const double *ca = pA + kc * advA, *cb = pB + kc * ldb;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = bq_buffer_load_d2(ca, voffA, u * stepA);
            rb[u] = bq_buffer_load_d2(cb, voffB, u * stepB);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bq_d2 va = NEG_A ? -ra[u] : ra[u];
            if (A_ROWS) {
                sm.A[buf][2 * (tid & 7)][(tid >> 3) + 32 * u] = va.x;
                sm.A[buf][2 * (tid & 7) + 1][(tid >> 3) + 32 * u] = va.y;
            } else {
                *reinterpret_cast<bq_d2 *>(&sm.A[buf][wv + 4 * u][2 * lane]) = va;
            }
            *reinterpret_cast<bq_d2 *>(&sm.B[buf][wv + 4 * u][2 * lane]) = rb[u];
        }
    };
    const int64_t nchunks = kdim / BQ_GK;
    const int fr = lane & 15, fk = lane >> 4;
    // fragments of one k-slice of 4 (this lane: 4 A rows, 4 B rows), two register sets so that the reads of slice s+1 are in
    // flight while the 16 MFMAs of slice s issue
    double fa[2][4], fb[2][4];
    const double *const rA = &sm.A[0][fk][wr * 64 + fr * 2];
    const double *const rB = &sm.B[0][fk][wc * 64 + fr * 2];
    auto rd = [&](int set, int buf, int kk) {
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const bq_d2 a = *reinterpret_cast<const bq_d2 *>(rA + (buf * BQ_GK + kk * 4) * BQ_GP + tp * 32);
            const bq_d2 b = *reinterpret_cast<const bq_d2 *>(rB + (buf * BQ_GK + kk * 4) * BQ_GP + tp * 32);
            fa[set][2 * tp] = a.x;
            fa[set][2 * tp + 1] = a.y;
            fb[set][2 * tp] = b.x;
            fb[set][2 * tp + 1] = b.y;
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
    // sched_group_barrier pins the interleave inside each region (masks: MFMA 0x8, VALU 0x2, VMEM read 0x20, DS read 0x100,
    // DS write 0x200): the next slice's four ds_read2 go out first, the staging is threaded through the MFMAs one
    // instruction per MFMA — left alone, hipcc put all reads, writes and loads of a region in front of its MFMAs.
    // (Requesting chunk 0 of the NEXT tile before the epilogue of the current one — a kernel walking a strip of d = 128 tiles has
    // only 8 chunks per tile — was measured too: panel build unchanged, streamed RBF product 44.5 -> 48.2 ms.  Not kept.)
    auto steady = [&](int B, int O, int64_t c) {   // B, O are literals at every call site: LDS addresses fold to immediates
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
            __builtin_amdgcn_sched_group_barrier(0x2, NEG_A ? 2 : 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
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
    };
    auto slice = [&](int set_next, int B, int kk_next, int set) {
        rd(set_next, B, kk_next);
        mm(set);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    gload(0);
    lstore(0);
    if (nchunks > 1) gload(BQ_GK);
    __syncthreads();
    rd(0, 0, 0);
    int64_t c = 0;
    for (; c + 3 < nchunks; c += 2) {   // c is even here
        steady(0, 1, c);
        steady(1, 0, c + 1);
    }
    // at most three chunks are left.  ONE straight-line tail with the buffer index as a value (a few address additions per
    // chunk, on two chunks per tile at most): alternatives per parity made the accumulators live in different registers per
    // path, and the joins cost ~1000 spilled registers
    for (; c + 1 < nchunks; ++c) {
        const int B = (int)(c & 1), O = B ^ 1;
        slice(1, B, 1, 0);
        rd(0, B, 2);
        mm(1);
        lstore(O);   // chunk c + 1 waits in the registers
        if (c + 2 < nchunks) gload((c + 2) * BQ_GK);
        __builtin_amdgcn_sched_barrier(0);
        slice(1, B, 3, 0);
        __syncthreads();
        slice(0, O, 0, 1);
    }
    {
        const int B = (int)(c & 1);
        slice(1, B, 1, 0);
        slice(0, B, 2, 1);
        slice(1, B, 3, 0);
        mm(1);
    }
}

// Load / store a whole accumulator tile from / to a row-major matrix (pitch ld, even, C 16-byte aligned): 32 independent
// 16-byte accesses per lane (columns j = 0, 1 and j = 2, 3 of an accumulator row are adjacent), issued back to back; sixteen
// lanes cover 256 contiguous bytes of a row.
__device__ __forceinline__ void bq_tile_load(bq_d4 (&acc)[4][4], const double *__restrict__ C, int64_t ld) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const bq_d2 c = *reinterpret_cast<const bq_d2 *>(C + (int64_t)bq_acc_row(i, v) * ld + bq_acc_col(2 * jp));
                acc[i][2 * jp][v] = c.x;
                acc[i][2 * jp + 1][v] = c.y;
            }
}
__device__ __forceinline__ void bq_tile_store(const bq_d4 (&acc)[4][4], double *__restrict__ C, int64_t ld) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                bq_d2 c;
                c.x = acc[i][2 * jp][v];
                c.y = acc[i][2 * jp + 1][v];
                *reinterpret_cast<bq_d2 *>(C + (int64_t)bq_acc_row(i, v) * ld + bq_acc_col(2 * jp)) = c;
            }
}

// C -= acc (the accumulators hold the product X_i X_j^T: no negation while staging): the tile is read AFTER the K loop,
// 8 rows at a time.  tools/syrk_probe.hip: "acc = C, loop, store" ran the tile update at 65.0 / 69.0 / 70.1 TFLOP/s for
// K = 512 / 768 / 1024, this form at 69.5 / 73.6 / 74.3 = the rate without any C traffic — the 32 strided C loads (one
// page per row) no longer sit in front of the first operand loads of the tile.  Warming the tile's lines with scratch loads
// under the MFMAs on top of it was 1-3 % slower and is not kept.
__device__ __forceinline__ void bq_tile_sub_store(const bq_d4 (&acc)[4][4], double *__restrict__ C, int64_t ld) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bq_d2 c[4][2];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
                c[v][jp] = *reinterpret_cast<const bq_d2 *>(C + (int64_t)bq_acc_row(i, v) * ld + bq_acc_col(2 * jp));
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                c[v][jp].x -= acc[i][2 * jp][v];
                c[v][jp].y -= acc[i][2 * jp + 1][v];
                *reinterpret_cast<bq_d2 *>(C + (int64_t)bq_acc_row(i, v) * ld + bq_acc_col(2 * jp)) = c[v][jp];
            }
    }
}

// Visit every accumulator element of this lane: f(row_in_tile, col_in_tile, value)
template <typename F>
__device__ __forceinline__ void bq_tile_foreach(bq_d4 (&acc)[4][4], F &&f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) f(bq_acc_row(i, v), bq_acc_col(j), acc[i][j][v]);
}

__device__ __forceinline__ void bq_tile_zero(bq_d4 (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (bq_d4){0.0, 0.0, 0.0, 0.0};
}
