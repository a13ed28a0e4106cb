// SMO on the resident Gram panel — SURVEY 8(f).4.
//
// Restates on the device the reference's sequential minimal optimisation for the reg_intercept=False duals
// (optiml/ml/svm/smo.py): classification (:99-357, Platt's SMO with the two-threshold rule of Keerthi et al.) and
// regression (:386-797, Shevade et al.).  The algorithm is a sequential sweep over the samples with data-dependent
// control flow, so it is latency-bound, not bandwidth-bound: ONE persistent 1024-thread workgroup runs a whole outer
// iteration per launch.  Thread 0 carries the scalar pair logic; the other threads share the three vector jobs:
//   (1) E_i = sum_j coef_j K[i][j] for a sample whose error is not cached — over a compact, ascending list of the
//       samples with a non-zero coefficient (mirrored in LDS, kept current by in-place insert / erase / update after
//       every pair step), so an examine costs O(n_sv / 1024) panel reads per thread instead of the reference's dense
//       O(n) dot;
//   (2) the error-cache update of the free set after a successful pair step (two panel rows);
//   (3) the re-computation of the two thresholds over the free set.
// The index sets I0..I4 of the reference are functions of (alpha, y) and are not stored.
// TIE RULE: where the reference's `for i in self.I0` loop (CPython set order) decides between two free samples with
// bit-identical cached errors, this kernel takes the smaller index (oracle/smo_oracle.py, tie='index').
// K[i][j] comes from the packed lower-triangular tile-row panel (bq_sym_addr: the row part is contiguous, the part right
// of the diagonal tile is read down the column of the later tile rows — one page per support vector) or, preferably, from
// a full square panel (BQ_FULL_PANEL: every row contiguous, one page per examined sample).
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <algorithm>

#include "bq_common.h"

// -DBQ_SMO_STAMPS: thread 0 of the SVC walker accumulates the time of each phase of a full-sweep round (100 MHz
// wall clock) and prints the totals at the end of the launch — diagnostic builds only
#ifdef BQ_SMO_STAMPS
#include <cstdio>
#define SMO_STAMP(k)                                  \
    if (threadIdx.x == 0) {                           \
        const unsigned long long _t = wall_clock64(); \
        stamp_acc[k] += _t - stamp_t;                 \
        stamp_cnt[k] += 1;                            \
        stamp_t = _t;                                 \
    }
#else
#define SMO_STAMP(k)
#endif

constexpr int SMO_T = 1024;

struct bq_smo_scal {
    double b_up, b_low;
    long long i_up, i_low;
    long long outer, steps, changed;
    int sweep_all, finished, err_flag, pad;
};

struct bq_smo {
    bq_problem *p = nullptr;
    int task = BQ_SVC;
    int64_t n = 0;
    double C = 1.0, eps = 0.0, tol = 1e-3;
    double *y = nullptr, *a = nullptr, *am = nullptr, *err = nullptr;   // a: alpha (SVC) / alpha+ (SVR); am: alpha-
    int *nz = nullptr;                                                   // n: support list (ascending indices)
    double *cf = nullptr;                                                // n: and its coefficients
    bq_smo_scal *sc = nullptr;
    bq_smo_scal host;
    // helper workgroups of the full sweeps (see "Helpers" below)
    unsigned int *ctl = nullptr;       // SPEC_* words
    void *spec_res = nullptr;          // n x 16 bytes: {bit pattern of sum_q c_q K[s][idx_q] formed by a helper, the list
                                       // version it was formed under, checksum of the bits} — ONE 16-byte store / load
    unsigned int epoch = 0;            // launch counter
    int helpers = 0;                   // helper workgroups per full-sweep launch (0: none)
};

// ---------------------------------------------------------------------------------------------------------------
// Helpers.  What a full sweep spends its time on is gathering panel entries: one error is n_sv scattered 8-byte reads,
// and ONE workgroup lives on one CU, whose miss handling sustains about 0.44 G gathers/s — 127 ms for a quiet sweep over
// n = 100 000 samples with 558 support vectors, with 255 CUs idle.  So a full-sweep launch carries `helpers` more
// workgroups whose waves form the SAME sums (same list order, same lane assignment, same tree: bit-identical) for the
// samples the walker (workgroup 0) is about to examine, out of the support list the walker keeps in global memory:
//   * walker -> helpers: a control block {version, list length, position, window}.  The version is a sequence lock: odd
//     while the walker edits the list (a pair step), bumped to the next even value afterwards.
//   * helpers -> walker: res[s] = {sum bits, version it was formed under, checksum}, one 16-byte write-through store
//     (the data is its own flag: no fence on either side).  A helper wave re-reads the version after its gathers and
//     drops the sum if the list moved meanwhile.
//   * sample s of the window [position, position + window) belongs to helper wave ((s - base) mod 16 H), base = the
//     walker's position when the version was published: consecutive samples go to different CUs, and right after a
//     pair step (window H) it is wave 0 of each helper that forms one sum; every quiet batch doubles the window up to
//     16 H samples.  Only wave 0 of a helper reads the control block in memory (every ~0.9 us when idle) and copies it
//     into the workgroup's LDS for its other fifteen waves: 2048 waves polling one line slowed the walker's own stores
//     2.6 times, and waves that back off instead wake too late to be ahead of the walk after a pair step.
// Nobody waits for anybody: the walker polls the tag of its sample between the chunks of its own gather loop and takes
// whichever result is there first, and the helpers leave when the walker publishes the end of the launch — a helper
// that never gets scheduled costs nothing.  Trajectories do not depend on the helpers: a tagged sum is the same
// double the walker would have formed.
// ---------------------------------------------------------------------------------------------------------------
enum { SPEC_VER = 0, SPEC_NNZ = 1, SPEC_WIN = 2, SPEC_DONE = 3, SPEC_POS = 4 /* 64-bit, words 4-5 */,
       SPEC_BASE = 6 /* 64-bit: the walker's position when the version was published */,
       SPEC_HASH = 8 /* 64-bit: multiset hash of the list entries of the published version */,
       // diagnostics, accumulated over the launches of a fit (bq_smo_get BQ_SMO_STATS): sums a helper formed under a stable
       // version whose entries did not hash to the published value (a torn list: must stay 0), results whose checksum did
       // not match their bits on the walker's side (a torn 16-byte granule: must stay 0), sums delivered
       SPEC_REJ_HASH = 10, SPEC_REJ_CHK = 11, SPEC_DELIVERED = 12, SPEC_WORDS = 16 };
typedef unsigned int smo_u4 __attribute__((ext_vector_type(4)));
struct SmoSpec {
    unsigned int *ctl;
    smo_u4 *res;
    unsigned int res_bytes;
    unsigned int epoch;
    int helpers;
};
// res[s]: one naturally aligned 16-byte granule, written by ONE write-through (sc1) store of one lane and read by ONE sc1
// load (MI355X_MICROARCH.md, inter-workgroup visibility: R2 — the data is the flag)
__device__ __forceinline__ smo_u4 res_load(const SmoSpec &P, int64_t s) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)P.res, 0, (int)P.res_bytes, 0x00020000);
    // aux: bit 4 = sc1, bit 31 = volatile (every call is a fresh load: the walker polls these words inside its gather loop)
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(s * 16), 0, (int)(0x80000000u | 16u));
}
__device__ __forceinline__ smo_u4 res_load_uni(const SmoSpec &P, int64_t s) {   // wave-uniform address: lane 0's copy
    const smo_u4 v = res_load(P, s);
    smo_u4 o;
    o.x = (unsigned int)__builtin_amdgcn_readfirstlane((int)v.x);
    o.y = (unsigned int)__builtin_amdgcn_readfirstlane((int)v.y);
    o.z = (unsigned int)__builtin_amdgcn_readfirstlane((int)v.z);
    o.w = (unsigned int)__builtin_amdgcn_readfirstlane((int)v.w);
    return o;
}
__device__ __forceinline__ void res_store(const SmoSpec &P, int64_t s, long long bits, unsigned int ver, unsigned int chk) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)P.res, 0, (int)P.res_bytes, 0x00020000);
    const smo_u4 v = {(unsigned int)bits, (unsigned int)((unsigned long long)bits >> 32), ver, chk};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(s * 16), 0, 16);
}
__device__ __forceinline__ long long res_bits(const smo_u4 &v) { return (long long)(((unsigned long long)v.y << 32) | v.x); }
// Loads of the shared words.  Every lane of a wave reads the same address, but nothing makes the 64 lanes of one load
// instruction observe the same store, and the code that follows branches on the value: lane 0's copy is broadcast so
// that the whole wave takes one decision.
// Every shared word is accessed through a GLOBAL (address space 1) pointer: global_load / global_store ... sc1, never flat_
// (flat operations complete out of order with respect to the vmcnt the hand-off waits on).
typedef __attribute__((address_space(1))) unsigned int smo_gu32;
typedef __attribute__((address_space(1))) long long smo_gi64;
#define SMO_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ unsigned int ld_rlx(const unsigned int *p) {
    return (unsigned int)__builtin_amdgcn_readfirstlane(
        (int)__hip_atomic_load((const smo_gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ long long ld_rlx64(const long long *p) {
    const long long v = __hip_atomic_load((const smo_gi64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long ld_uni64(long long v) {   // lane 0's copy of a 64-bit value
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void st_rlx(unsigned int *p, unsigned int v) {
    __hip_atomic_store((smo_gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_rlx64(long long *p, long long v) {
    __hip_atomic_store((smo_gi64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a flag behind everything this WORKGROUP stored before its last barrier: agent-scope release (L2 write-back), an explicit
// wait (the compiler may drop the fence's own), then the write-through store
__device__ __forceinline__ void st_rel(unsigned int *p, unsigned int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    SMO_DRAIN();
    __hip_atomic_store((smo_gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fence_acq() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// Ordering (MI355X_MICROARCH.md, inter-workgroup visibility).  Walker -> helpers: the list is edited with WRITE-THROUGH
// (sc1) stores by all sixteen waves; EVERY wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup barrier, then
// thread 0 stores the control words write-through, waits again, and only then the new even version — the "sc1 payload,
// drained, one lane's sc1 flag" form, which needs no L2 write-back per pair step.  (Round 1 had a workgroup-scope fence in front of the barrier: it compiles to nothing another CU
// can observe, so fifteen waves' list stores could still be in flight when the version appeared — a helper then formed a
// sum from old entries under the new version: the 1-in-3-runs path change with 240+ helpers.)  Helpers: wave 0 polls
// the version relaxed and runs ONE agent-scope acquire when it moves; every consuming wave runs its own acquire when it
// picks a new version up, reads the list with agent-scope loads, runs an acquire between those loads and the re-check
// of the version, and delivers only under an unchanged even version.
// On top of that the results validate themselves, and the validations are COUNTED (they must never fire):
//  * the walker publishes, with every list version, the sum over the list entries of a 64-bit mix of (index,
//    coefficient bits) — order-free, so thread 0 keeps it current with one subtraction / addition per edit; a helper
//    wave adds up the same mix over the entries it actually read and delivers only if the two agree (SPEC_REJ_HASH);
//  * a delivered result carries a checksum of its own bits (SPEC_REJ_CHK on the walker's side).
__device__ __forceinline__ unsigned long long entry_mix(unsigned int idx, double c) {
    unsigned long long x = (unsigned long long)__double_as_longlong(c) ^ ((unsigned long long)idx * 0x9E3779B97F4A7C15ull);
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return x;
}
__device__ __forceinline__ unsigned int dot_check(long long bits) {
    const unsigned long long u = (unsigned long long)bits * 0x9E3779B97F4A7C15ull;
    return (unsigned int)(u >> 32) ^ 0x5bd1e995u;
}

template <typename T>
struct KView {
    const T *panel;
    int64_t ld;   // 0: packed lower-triangular tile rows; else the pitch of a full square panel (BQ_FULL_PANEL)
    __device__ __forceinline__ double at(int64_t i, int64_t j) const {
        if (ld != 0) return (double)panel[i * ld + j];
        const int64_t ti = i / BQ_SYM_TILE, tj = j / BQ_SYM_TILE;
        return (double)(tj <= ti ? panel[bq_sym_addr(i, j, 0)] : panel[bq_sym_addr(j, i, 0)]);
    }
};

struct ValIdx {
    double v;
    long long i;
};
// maximum with the smaller index on ties (sign = +1) / minimum with the smaller index on ties (sign = -1)
__device__ __forceinline__ ValIdx better(ValIdx a, ValIdx b, int sign) {
    const bool b_wins = sign > 0 ? (b.v > a.v || (b.v == a.v && b.i < a.i && b.i >= 0))
                                 : (b.v < a.v || (b.v == a.v && b.i < a.i && b.i >= 0));
    if (a.i < 0) return b.i >= 0 ? b : a;
    if (b.i < 0) return a;
    return b_wins ? b : a;
}
__device__ __forceinline__ ValIdx smo_bbest(ValIdx x, int sign, double *shv, long long *shi) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ValIdx o;
        o.v = __shfl_down(x.v, off, 64);
        o.i = __shfl_down(x.i, off, 64);
        x = better(x, o, sign);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        shv[threadIdx.x >> 6] = x.v;
        shi[threadIdx.x >> 6] = x.i;
    }
    __syncthreads();
    ValIdx r{shv[0], shi[0]};
#pragma unroll
    for (int w = 1; w < SMO_T / 64; ++w) r = better(r, ValIdx{shv[w], shi[w]}, sign);
    return r;
}

struct SmoShared {
    double red[SMO_T / 64];
    double bv[SMO_T / 64];
    long long bi[SMO_T / 64];
    // scalar state, owned by thread 0 between barriers
    double b_up, b_low;
    long long i_up, i_low;
    // broadcast slots of the current examine
    double E2, c1, c2;
    long long i1, next;
    int go, fail;
    int pos, found;               // sup_apply
    long long i2;                 // the examined sample of a pair step
    int used, stop, fast;         // batch bookkeeping of the sweep drivers
    double ba[SMO_T / 64], by[SMO_T / 64], bE[SMO_T / 64], bm[SMO_T / 64];   // batch: multiplier(s), label/target, error
    int mem1, mem2;               // new list membership of the two touched samples
    double cf1, cf2;              // and their new coefficients
    int nnz;                      // length of the support list
    unsigned int ver;             // helpers: current (even) list version
    unsigned long long hash;      // helpers: sum of entry_mix over the list (thread 0)
    double oldcf;                 // sup_apply: coefficient of the entry found
    unsigned int win;             // helpers: look-ahead window, in samples
    int scan[SMO_T / 64];
};

// exclusive prefix sum of one int per thread (wave scan, then the 16 wave totals); *total gets the block sum
__device__ __forceinline__ int smo_bscan(int v, int *sh, int *total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SMO_T / 64; ++w) {
        if (w < wv) base += sh[w];
        tot += sh[w];
    }
    *total = tot;
    return base + inc - v;
}

// ascending list of the samples whose coefficient is non-zero (thread t compacts the contiguous chunk t)
template <typename Pred>
__device__ __forceinline__ void smo_rebuild(int64_t n, int *__restrict__ nz, SmoShared &S, Pred nonzero) {
    const int64_t chunk = (n + SMO_T - 1) / SMO_T;
    const int64_t lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    int cnt = 0;
#pragma unroll 4
    for (int64_t j = lo; j < hi; ++j) cnt += nonzero(j) ? 1 : 0;
    int total;
    int pos = smo_bscan(cnt, S.scan, &total);
    for (int64_t j = lo; j < hi; ++j)
        if (nonzero(j))   // write-through like every store to the global list (see sup_put)
            __hip_atomic_store((__attribute__((address_space(1))) int *)&nz[pos++], (int)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) S.nnz = total;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Support list: the samples with a non-zero multiplier, ascending, with their coefficients c_j (alpha_j y_j, or
// alpha+_j - alpha-_j).  Global arrays hold the whole list; its first SMO_CAP entries are mirrored in LDS so that an
// error evaluation costs one level of global latency (the panel entry) instead of three (index, multiplier, entry).
// A pair step changes at most two entries: they are inserted / erased / updated in place (binary search + a parallel
// shift), not by rebuilding the list.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SMO_CAP = 4096;
struct SupList {
    int nz[SMO_CAP];
    double cf[SMO_CAP];
};
struct SupGlobal {
    int *nz;
    double *cf;
};
__device__ __forceinline__ int sup_idx(const SupList &L, const SupGlobal &G, int q) { return q < SMO_CAP ? L.nz[q] : G.nz[q]; }
__device__ __forceinline__ double sup_cf(const SupList &L, const SupGlobal &G, int q) { return q < SMO_CAP ? L.cf[q] : G.cf[q]; }
// the global copy is what the helper workgroups read: WRITE-THROUGH (sc1) stores, so that publishing an edit needs no L2
// write-back (MI355X_MICROARCH.md, visibility: payload stored sc1 + every storing wave drained + flag = a valid hand-off for
// sc1 loads; the release fence it replaces cost a buffer_wbl2 per pair step)
__device__ __forceinline__ void sup_put(SupList &L, const SupGlobal &G, int q, int idx, double c) {
    __hip_atomic_store((__attribute__((address_space(1))) int *)&G.nz[q], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((__attribute__((address_space(1))) long long *)&G.cf[q], __double_as_longlong(c), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    if (q < SMO_CAP) {
        L.nz[q] = idx;
        L.cf[q] = c;
    }
}

// after smo_rebuild filled G.nz: coefficients + LDS mirror
template <typename Coef>
__device__ __forceinline__ void sup_fill(SupList &L, const SupGlobal &G, SmoShared &S, Coef coef) {
    unsigned long long h = 0ull;
    for (int q = threadIdx.x; q < S.nnz; q += SMO_T) {
        const int j = G.nz[q];
        const double c = coef(j);
        sup_put(L, G, q, j, c);
        h += entry_mix((unsigned int)j, c);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += (unsigned long long)__shfl_down((long long)h, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) S.bi[threadIdx.x >> 6] = (long long)h;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0ull;
        for (int w = 0; w < SMO_T / 64; ++w) t += (unsigned long long)S.bi[w];
        S.hash = t;
    }
    __syncthreads();
}

// make the list reflect sample idx: member = its multiplier(s) are non-zero now, c = its coefficient.  All threads call
// this with the same arguments.
__device__ __forceinline__ void sup_apply(SupList &L, const SupGlobal &G, SmoShared &S, int idx, bool member, double c) {
    if (threadIdx.x == 0) {   // lower bound of idx
        int lo = 0, hi = S.nnz;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sup_idx(L, G, mid) < idx)
                lo = mid + 1;
            else
                hi = mid;
        }
        S.pos = lo;
        S.found = (lo < S.nnz && sup_idx(L, G, lo) == idx) ? 1 : 0;
        if (S.found) S.hash -= entry_mix((unsigned int)idx, sup_cf(L, G, lo));
        if (member) S.hash += entry_mix((unsigned int)idx, c);
    }
    __syncthreads();
    const int pos = S.pos, nnz = S.nnz;
    const bool found = S.found != 0;
    if (member && found) {
        if (threadIdx.x == 0) sup_put(L, G, pos, idx, c);
    } else if (member) {   // insert: shift [pos, nnz) one to the right, highest chunk first
        for (int top = nnz - 1; top >= pos; top -= SMO_T) {
            const int q = top - (int)threadIdx.x;
            int vi = 0;
            double vc = 0.0;
            if (q >= pos) {
                vi = sup_idx(L, G, q);
                vc = sup_cf(L, G, q);
            }
            __syncthreads();
            if (q >= pos) sup_put(L, G, q + 1, vi, vc);
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            sup_put(L, G, pos, idx, c);
            S.nnz = nnz + 1;
        }
    } else if (found) {    // erase: shift (pos, nnz) one to the left, lowest chunk first
        for (int bot = pos + 1; bot < nnz; bot += SMO_T) {
            const int q = bot + (int)threadIdx.x;
            int vi = 0;
            double vc = 0.0;
            if (q < nnz) {
                vi = sup_idx(L, G, q);
                vc = sup_cf(L, G, q);
            }
            __syncthreads();
            if (q < nnz) sup_put(L, G, q - 1, vi, vc);
            __syncthreads();
        }
        if (threadIdx.x == 0) S.nnz = nnz - 1;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Sweep drivers shared by both tasks.
// Full sweep (every sample in index order): a batch of 16 samples is prepared at once, one per wave — the wave reads
// the sample's multiplier(s) and, unless its error is cached, forms E = sum_q c_q K[s][idx_q] over the support list
// (lane l adds list positions l, l + 64, ... in ascending order with separately rounded products, the 64 lanes fold
// by halves: the order oracle/smo_oracle.py reproduces with dot='tree').  Thread 0 then examines the 16 samples IN
// ORDER with exactly the reference's scalar logic; a successful pair step invalidates the rest of the batch (their
// errors were formed with the old multipliers), which is simply prepared again after the step.  Most examined samples
// are no violators, so a batch usually costs four barriers for 16 samples.
// Free-set sweep (smo.py:336-342): only samples with a multiplier strictly inside the box, whose errors are cached —
// thread 0 walks the support list (the free set is inside it) one sample at a time.
// ---------------------------------------------------------------------------------------------------------------
// walker: bracket an edit of the global support list (all threads; no-ops without helpers).  `at` = the position the
// walk continues from.
__device__ __forceinline__ void spec_begin(const SmoSpec &P, SmoShared &S) {
    if (P.helpers == 0) return;
    if (threadIdx.x == 0) {
        st_rlx(&P.ctl[SPEC_VER], S.ver + 1u);   // odd: the list is being edited (write-through store)
        SMO_DRAIN();                            // ... and has left this CU before any edit store is issued
    }
    __syncthreads();
}
__device__ __forceinline__ void spec_end(const SmoSpec &P, SmoShared &S, long long at) {
    if (P.helpers == 0) return;
    SMO_DRAIN();   // EVERY wave: its write-through list stores have left the CU and are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        S.ver += 2u;
        S.win = (unsigned int)P.helpers;   // one sample for wave 0 of every helper workgroup
        st_rlx(&P.ctl[SPEC_NNZ], (unsigned int)S.nnz);
        st_rlx(&P.ctl[SPEC_WIN], S.win);
        st_rlx64((long long *)&P.ctl[SPEC_BASE], at);
        st_rlx64((long long *)&P.ctl[SPEC_POS], at);
        st_rlx64((long long *)&P.ctl[SPEC_HASH], (long long)S.hash);
        SMO_DRAIN();                                         // the words of this version are out before the version itself
        st_rlx(&P.ctl[SPEC_VER], S.ver);
    }
    __syncthreads();
}

// what a helper workgroup knows about the walk: wave 0 copies the control block here every time it looks, the other
// fifteen waves read this copy (LDS, on their own CU) instead of the shared line in memory
struct HelpBoard {
    unsigned int seq;   // sequence lock of this copy: odd while wave 0 rewrites it
    unsigned int v, win, nnz, quit;
    unsigned int pos_lo, pos_hi, base_lo, base_hi;
};
__device__ __forceinline__ unsigned int lds_uni(const volatile unsigned int *p) {
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)*p);
}

// the sums of this wave's samples in the window [pos, pos + win) under list version v; true if one was delivered
template <typename T>
__device__ __forceinline__ bool helper_sums(const KView<T> &K, int64_t n, const SupGlobal &G, const SmoSpec &P, unsigned int v,
                                            unsigned int win, int nnz, long long pos, long long base, long long mine,
                                            long long stride, unsigned int &delivered) {
    const int lane = threadIdx.x & 63;
    bool did = false;
    long long s = base + mine;
    if (s < pos) s += (pos - s + stride - 1) / stride * stride;
    for (; s < n && s < pos + (long long)win; s += stride) {
        if (res_load_uni(P, s).z == v) continue;   // already delivered under this version
        double part = 0.0;   // the sum of wave_dot, from the global copy of the list
        unsigned long long seen = 0ull;   // entry_mix over the entries this lane read
        for (int q0 = lane; q0 < nnz; q0 += 64 * 8) {
            double kv[8], cf[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = q0 + 64 * u;
                const bool in = q < nnz;
                // The list is read with agent-scope (sc1) loads, which do not go through this CU's L1
                cf[u] = in ? __longlong_as_double(__hip_atomic_load((const smo_gi64 *)&G.cf[q], __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT))
                           : 0.0;
                // while the walker edits the list an entry can be anything; the sum is dropped below, but the
                // panel read must stay inside the panel
                const unsigned int j = in ? (unsigned int)__hip_atomic_load((const __attribute__((address_space(1))) int *)&G.nz[q],
                                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                kv[u] = in ? K.at(s, j < (unsigned long long)n ? (int64_t)j : 0) : 0.0;
                if (in) seen += entry_mix(j, cf[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (q0 + 64 * u < nnz) part = part + __dmul_rn(cf[u], kv[u]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) seen += (unsigned long long)__shfl_down((long long)seen, off, 64);
        seen = (unsigned long long)ld_uni64((long long)seen);
        fence_acq();   // the list reads above are complete before the control block is looked at again
        const unsigned long long want = (unsigned long long)ld_rlx64((const long long *)&P.ctl[SPEC_HASH]);
        if (ld_rlx(&P.ctl[SPEC_VER]) != v) break;   // the list moved under the gathers: drop the sum
        if (seen != want) {                         // entries of another version under a stable version: must not happen
            if (lane == 0) atomicAdd(&P.ctl[SPEC_REJ_HASH], 1u);
            continue;
        }
        if (lane == 0) {
            const long long bits = __double_as_longlong(part);
            res_store(P, s, bits, v, dot_check(bits));
        }
        did = true;
        ++delivered;
    }
    return did;
}

// helper workgroups (see "Helpers" above): one barrier at the start, none afterwards — every wave runs on its own
template <typename T>
__device__ void smo_helper(const KView<T> &K, int64_t n, const SupGlobal &G, const SmoSpec &P) {
    __shared__ HelpBoard HB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long stride = (long long)P.helpers * (SMO_T / 64);
    const long long mine = (long long)(blockIdx.x - 1) + (long long)P.helpers * wv;   // residue of (s - base) modulo stride
    volatile HelpBoard *hb = &HB;
    if (threadIdx.x == 0) {
        hb->seq = 0u;
        hb->v = 1u;   // odd: nothing to do yet
        hb->win = 0u;
        hb->quit = 0u;
    }
    __syncthreads();
    unsigned int delivered = 0u;
    if (wv == 0) {
        unsigned int seq = 0u, last_v = 1u;
        while (true) {
            const unsigned int quit = ld_rlx(&P.ctl[SPEC_DONE]) == P.epoch ? 1u : 0u;
            const unsigned int v = ld_rlx(&P.ctl[SPEC_VER]);   // ONE relaxed poll ...
            if (v != last_v) {
                fence_acq();                                    // ... ONE agent-scope acquire when the version has moved
                last_v = v;
            }
            const unsigned int win = ld_rlx(&P.ctl[SPEC_WIN]);
            const int nnz = (int)ld_rlx(&P.ctl[SPEC_NNZ]);
            const long long pos = ld_rlx64((const long long *)&P.ctl[SPEC_POS]);
            const long long base = ld_rlx64((const long long *)&P.ctl[SPEC_BASE]);
            if (lane == 0) {
                hb->seq = seq + 1u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                hb->v = v;
                hb->win = win;
                hb->nnz = (unsigned int)nnz;
                hb->quit = quit;
                hb->pos_lo = (unsigned int)pos;
                hb->pos_hi = (unsigned int)((unsigned long long)pos >> 32);
                hb->base_lo = (unsigned int)base;
                hb->base_hi = (unsigned int)((unsigned long long)base >> 32);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                hb->seq = seq + 2u;
            }
            seq += 2u;
            if (quit) break;
            bool did = false;
            if (!(v & 1u) && win != 0u) did = helper_sums(K, n, G, P, v, win, nnz, pos, base, mine, stride, delivered);
            if (!did) __builtin_amdgcn_s_sleep(32);   // ~0.9 us between looks at the shared line
        }
    } else {
        unsigned int seen = 0u, last_v = 1u;
        while (true) {
            const unsigned int s1 = lds_uni(&hb->seq);
            if ((s1 & 1u) || s1 == seen) {   // being rewritten, or nothing new since the last pass
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const unsigned int v = lds_uni(&hb->v), win = lds_uni(&hb->win), nnz = lds_uni(&hb->nnz), quit = lds_uni(&hb->quit);
            const unsigned int plo = lds_uni(&hb->pos_lo), phi = lds_uni(&hb->pos_hi);
            const unsigned int blo = lds_uni(&hb->base_lo), bhi = lds_uni(&hb->base_hi);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lds_uni(&hb->seq) != s1) continue;   // wave 0 rewrote the copy meanwhile
            seen = s1;
            if (quit) break;
            if ((v & 1u) || win == 0u) continue;
            if (v != last_v) {   // this wave's own agent-scope acquire for the version it is about to read the list under
                fence_acq();
                last_v = v;
            }
            helper_sums(K, n, G, P, v, win, (int)nnz, (long long)(((unsigned long long)phi << 32) | plo),
                        (long long)(((unsigned long long)bhi << 32) | blo), mine, stride, delivered);
        }
    }
    delivered = (unsigned int)__builtin_amdgcn_readfirstlane((int)delivered);
    if (lane == 0 && delivered) atomicAdd(&P.ctl[SPEC_DELIVERED], delivered);
}

constexpr int SMO_B = SMO_T / 64;

template <typename T>
__device__ __forceinline__ double wave_dot(const KView<T> &K, const SupList &L, const SupGlobal &G, int nnz, int64_t s,
                                           const SmoSpec &P, unsigned int ver, bool ahead) {
    double part = 0.0;
    // eight panel entries of this lane are fetched together (independent loads in flight), then added in list order —
    // the summation order is that of the plain loop; only the latency of the reads overlaps
    for (int q0 = threadIdx.x & 63; q0 < nnz; q0 += 64 * 8) {
        // Right after a pair step the tag of this sample travels with the chunk's gathers (one more load in flight, no
        // extra round trip: the helpers have had no time yet).  Once the look-ahead has grown the helpers are ahead of
        // the walk and the tag is looked at first — gathers the walker does not issue are what makes it fast.
        smo_u4 res = {0u, 0u, 0u, 0u};
        if (P.helpers != 0) {
            res = res_load_uni(P, s);
            if (ahead && res.z == ver) {
                if (dot_check(res_bits(res)) == res.w) return __longlong_as_double(res_bits(res));
                if ((threadIdx.x & 63) == 0) atomicAdd(&P.ctl[SPEC_REJ_CHK], 1u);
            }
        }
        double kv[8], cf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = q0 + 64 * u;
            const bool in = q < nnz;
            cf[u] = in ? sup_cf(L, G, q) : 0.0;
            kv[u] = in ? K.at(s, sup_idx(L, G, q)) : 0.0;
        }
        if (P.helpers != 0 && !ahead && res.z == ver) {   // wave-uniform: a helper has formed this very sum
            if (dot_check(res_bits(res)) == res.w) return __longlong_as_double(res_bits(res));
            if ((threadIdx.x & 63) == 0) atomicAdd(&P.ctl[SPEC_REJ_CHK], 1u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q0 + 64 * u < nnz) part = part + __dmul_rn(cf[u], kv[u]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    return part;   // complete in lane 0
}

// smallest list position whose sample index is >= idx (thread 0)
__device__ __forceinline__ int sup_lower_bound(const SupList &L, const SupGlobal &G, int nnz, long long idx) {
    int lo = 0, hi = nnz;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sup_idx(L, G, mid) < idx)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------------------------
// classification (smo.py:130-319)
// ---------------------------------------------------------------------------------------------------------------
// thread 0: examine sample i2 (smo.py:278-319) and, if it violates, solve the pair (:130-224).  E2 is the sample's
// error (cached, or just formed).  Returns true when a pair step was taken; its data is left in S for svc_after_step.
template <typename T>
__device__ __forceinline__ bool svc_examine(SmoShared &S, const KView<T> &K, const double *y, double *a, double *err,
                                            double C, double tol, long long i2, double a2, double y2, double E2) {
    const bool free2 = a2 > 0.0 && a2 < C;
    const bool up2 = (y2 == 1.0 && a2 == 0.0) || (y2 == -1.0 && a2 == C);     // I1 or I2
    const bool low2 = (y2 == 1.0 && a2 == C) || (y2 == -1.0 && a2 == 0.0);    // I3 or I4
    if (!free2) {
        err[i2] = E2;
        if (up2 && E2 < S.b_up) {
            S.b_up = E2;
            S.i_up = i2;
        } else if (low2 && E2 > S.b_low) {
            S.b_low = E2;
            S.i_low = i2;
        }
    }
    long long i1 = -1;
    if ((free2 || up2) && S.b_low - E2 > 2 * tol) i1 = S.i_low;
    if ((free2 || low2) && E2 - S.b_up > 2 * tol) i1 = S.i_up;
    if (i1 >= 0 && free2) i1 = (S.b_low - E2 > E2 - S.b_up) ? S.i_low : S.i_up;
    if (i1 < 0 || i1 == i2) return false;
    const double a1 = a[i1], y1 = y[i1], E1 = err[i1];
    double L, H;
    if (y1 != y2) {
        L = fmax(0.0, a2 - a1);
        H = fmin(C, C + a2 - a1);
    } else {
        L = fmax(0.0, a2 + a1 - C);
        H = fmin(C, a2 + a1);
    }
    if (L == H) return false;
    const double k11 = K.at(i1, i1), k22 = K.at(i2, i2), k12 = K.at(i1, i2);
    const double eta = k11 + k22 - 2 * k12;
    double n2;
    if (eta > 0.0) {
        n2 = fmax(L, fmin(a2 + __dmul_rn(y2, E1 - E2) / eta, H));
    } else {
        const double lo = __dmul_rn(__dmul_rn(y2, E1 - E2), L), hi = __dmul_rn(__dmul_rn(y2, E1 - E2), H);
        n2 = lo > hi + 1e-12 ? L : (lo < hi - 1e-12 ? H : a2);
    }
    if (fabs(n2 - a2) < __dmul_rn(1e-12, n2 + a2 + 1e-12)) return false;
    double n1 = a1 + __dmul_rn(__dmul_rn(y1, y2), a2 - n2);
    const double c1 = __dmul_rn(y1, n1 - a1), c2 = __dmul_rn(y2, n2 - a2);
    // own entries of the error cache, with the old multipliers (smo.py:203-204)
    err[i1] = E1 + (__dmul_rn(c1, k11) + __dmul_rn(c2, k12));
    err[i2] = E2 + (__dmul_rn(c1, k12) + __dmul_rn(c2, k22));
    n2 = n2 > C - __dmul_rn(1e-8, C) ? C : (n2 <= __dmul_rn(1e-8, C) ? 0.0 : n2);
    n1 = n1 > C - __dmul_rn(1e-8, C) ? C : (n1 <= __dmul_rn(1e-8, C) ? 0.0 : n1);
    a[i1] = n1;
    a[i2] = n2;
    S.mem1 = n1 != 0.0;
    S.mem2 = n2 != 0.0;
    S.cf1 = __dmul_rn(n1, y1);
    S.cf2 = __dmul_rn(n2, y2);
    S.c1 = c1;
    S.c2 = c2;
    S.i1 = i1;
    S.i2 = i2;
    return true;
}

// all threads, after a pair step: support list, error cache of the free set, thresholds (smo.py:199-201, :241-271)
template <typename T>
__device__ __forceinline__ void svc_after_step(SmoShared &S, SupList &L, const SupGlobal &G, const KView<T> &K,
                                               const double *y, double *a, double *err, double C, const SmoSpec &P) {
    const int tid = threadIdx.x;
    const long long i1 = S.i1, i2 = S.i2;
    spec_begin(P, S);
    sup_apply(L, G, S, (int)i1, S.mem1 != 0, S.cf1);   // the list now reflects the new multipliers
    sup_apply(L, G, S, (int)i2, S.mem2 != 0, S.cf2);
    spec_end(P, S, i2 + 1);   // a full sweep continues behind the examined sample
    const double c1 = S.c1, c2 = S.c2;
    ValIdx hi{-DBL_MAX, -1}, lo{DBL_MAX, -1};
    for (int q = tid; q < S.nnz; q += SMO_T) {   // the free set is a subset of the support list
        const int64_t j = sup_idx(L, G, q);
        const double aj = a[j];
        if (aj > 0.0 && aj < C) {
            double e = err[j];
            if (j != i1 && j != i2) {
                e += __dmul_rn(c1, K.at(i1, j)) + __dmul_rn(c2, K.at(i2, j));
                err[j] = e;
            }
            if (e > hi.v) hi = ValIdx{e, (long long)j};     // ascending j: first maximum / minimum kept
            if (e < lo.v) lo = ValIdx{e, (long long)j};
        }
    }
    hi = smo_bbest(hi, +1, S.bv, S.bi);
    lo = smo_bbest(lo, -1, S.bv, S.bi);
    if (tid == 0) {
        S.b_up = DBL_MAX;
        S.b_low = -DBL_MAX;
        S.i_up = -1;
        S.i_low = -1;
        if (hi.i >= 0 && hi.v > S.b_low) {
            S.b_low = hi.v;
            S.i_low = hi.i;
        }
        if (lo.i >= 0 && lo.v < S.b_up) {
            S.b_up = lo.v;
            S.i_up = lo.i;
        }
        const long long pair[2] = {i1, i2};
        for (int k = 0; k < 2; ++k) {   // the two touched samples when they left the free set (smo.py:253-268)
            const long long i = pair[k];
            const double ai = a[i], yi = y[i], ei = err[i];
            if (ai > 0.0 && ai < C) continue;
            const bool low_i = (yi == 1.0 && ai == C) || (yi == -1.0 && ai == 0.0);
            if (low_i) {
                if (ei > S.b_low) {
                    S.b_low = ei;
                    S.i_low = i;
                }
            } else if (ei < S.b_up) {
                S.b_up = ei;
                S.i_up = i;
            }
        }
        if (S.i_low < 0 || S.i_up < 0) S.fail = 1;   // 'unexpected status'
    }
    __syncthreads();
}

// One wave, one lane per sample of a batch of B <= 64 samples starting at i (lane w: multiplier a2, label y2, error E2;
// `in` = w < B), no side effects until it is known to apply: within a batch without a pair step b_up only falls and
// b_low only rises (smo.py:285-291), so a sample that does not violate the thresholds the batch ENDS with violates none
// of the intermediate ones — then the sequential examine of the batch reduces to storing the new errors and to a
// first-minimum / first-maximum over the batch, which the lanes do at once.  Returns false (nothing written) when any
// sample is a possible violator: the batch then goes down the sequential walk.
__device__ __forceinline__ bool svc_quiet_batch(SmoShared &S, double *err, double C, double tol, int64_t i, int B, bool in,
                                                double a2, double y2, double E2) {
    const int w = threadIdx.x & 63;
    const bool free2 = in && a2 > 0.0 && a2 < C;
    const bool up2 = in && ((y2 == 1.0 && a2 == 0.0) || (y2 == -1.0 && a2 == C));
    const bool low2 = in && ((y2 == 1.0 && a2 == C) || (y2 == -1.0 && a2 == 0.0));
    ValIdx lo{(up2 && !free2) ? E2 : DBL_MAX, (up2 && !free2) ? (long long)w : -1};
    ValIdx hi{(low2 && !free2) ? E2 : -DBL_MAX, (low2 && !free2) ? (long long)w : -1};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ValIdx o;
        o.v = __shfl_xor(lo.v, off, 64);
        o.i = __shfl_xor(lo.i, off, 64);
        lo = better(lo, o, -1);
        o.v = __shfl_xor(hi.v, off, 64);
        o.i = __shfl_xor(hi.i, off, 64);
        hi = better(hi, o, +1);
    }
    const bool up_moves = lo.i >= 0 && lo.v < S.b_up, low_moves = hi.i >= 0 && hi.v > S.b_low;
    const double bu = up_moves ? lo.v : S.b_up, bl = low_moves ? hi.v : S.b_low;
    const bool viol = in && (((free2 || up2) && bl - E2 > 2 * tol) || ((free2 || low2) && E2 - bu > 2 * tol));
    const bool fast = __ballot(viol) == 0ull;
    if (fast) {
        if (in && !free2) err[i + w] = E2;
        if (w == 0) {
            if (up_moves) {
                S.b_up = lo.v;
                S.i_up = i + lo.i;
            }
            if (low_moves) {
                S.b_low = hi.v;
                S.i_low = i + hi.i;
            }
            S.go = 0;
            S.used = B;
        }
    }
    return fast;
}

template <typename T>
__global__ __launch_bounds__(SMO_T) void smo_svc_kernel(KView<T> K, int64_t n, const double *__restrict__ y,
                                                        double *a, double *err, SupGlobal G, double C,
                                                        double tol, bq_smo_scal *sc, SmoSpec P) {
    if (blockIdx.x != 0) {
        smo_helper(K, n, G, P);
        return;
    }
    __shared__ SmoShared S;
    __shared__ SupList L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) {
        S.b_up = sc->b_up;
        S.b_low = sc->b_low;
        S.i_up = sc->i_up;
        S.i_low = sc->i_low;
        S.fail = 0;
        S.go = 0;
        S.ver = P.helpers ? ld_rlx(&P.ctl[SPEC_VER]) : 0u;
        S.win = 0;
    }
    __syncthreads();
    if (sc->finished) {
        if (tid == 0 && P.helpers) st_rel(&P.ctl[SPEC_DONE], P.epoch);
        return;
    }
    const bool sweep_all = sc->sweep_all != 0;
    long long changed = 0, steps = 0;
    spec_begin(P, S);
    smo_rebuild(n, G.nz, S, [&](int64_t j) { return a[j] != 0.0; });
    sup_fill(L, G, S, [&](int j) { return __dmul_rn(a[j], y[j]); });
    spec_end(P, S, 0);
    if (sweep_all) {
        int64_t i = 0;
#ifdef BQ_SMO_STAMPS
        unsigned long long stamp_acc[8] = {0}, stamp_cnt[8] = {0}, stamp_t = wall_clock64();
#endif
        while (i < n) {
            SMO_STAMP(7)   // loop tail of the previous round
            if (tid == 0 && P.helpers) {   // where the walk stands; a batch without a pair step doubles the look-ahead
                const unsigned int cap = (unsigned int)(P.helpers * SMO_B);
                if (!S.go) S.win = S.win * 2u < cap ? S.win * 2u : cap;
                st_rlx64((long long *)&P.ctl[SPEC_POS], (long long)i);
                st_rlx(&P.ctl[SPEC_WIN], S.win);
            }
            if (P.helpers) {
                // Once the look-ahead has grown, the helpers are ahead of the walk: wave 0 tries 64 samples at once, one per
                // lane — multiplier, label, the helper's tagged sum (or the cached error) — and the same quiet-batch
                // test as below.  A missing tag or a possible violator leaves everything to the 16-sample batch.
                if (wv == 0) {
                    bool wide = false;
                    if (S.win > (unsigned int)SMO_B) {   // same wave as thread 0, which has just written it
                        const int B = n - i < 64 ? (int)(n - i) : 64;
                        const bool in = lane < B;
                        const int64_t sw = i + lane;
                        const double a2 = in ? a[sw] : 0.0, y2 = in ? y[sw] : 0.0;
                        const bool free2 = in && a2 > 0.0 && a2 < C;
                        const bool need = in && !free2;
                        smo_u4 res = {0u, 0u, 0u, 0u};
                        if (need) res = res_load(P, sw);
                        const long long bits = res_bits(res);
                        if (__ballot(need && res.z != S.ver) == 0ull) {
                            const unsigned long long bad = __ballot(need && dot_check(bits) != res.w);
                            if (bad != 0ull && lane == 0) atomicAdd(&P.ctl[SPEC_REJ_CHK], (unsigned int)__popcll(bad));
                            if (bad == 0ull) {
                                double E2 = 0.0;
                                if (need)
                                    E2 = __longlong_as_double(bits) - y2;
                                else if (in)
                                    E2 = err[sw];
                                wide = svc_quiet_batch(S, err, C, tol, i, B, in, a2, y2, E2);
                            }
                        }
                    }
                    if (lane == 0) S.fast = wide ? 2 : 0;
                }
                __syncthreads();
                if (S.fast == 2) {
                    SMO_STAMP(0)   // a wide quiet round
                    i += S.used;
                    __syncthreads();   // S.fast / S.used are rewritten in the next round
                    continue;
                }
                SMO_STAMP(1)       // a wide attempt that fell through
            }
            const int64_t s = i + wv;
            if (s < n) {   // wave-uniform
                const double as = a[s], ys = y[s];
                double E = 0.0;
                if (!(as > 0.0 && as < C)) E = wave_dot(K, L, G, S.nnz, s, P, S.ver, S.win > (unsigned int)SMO_B) - ys;
                if (lane == 0) {
                    S.ba[wv] = as;
                    S.by[wv] = ys;
                    S.bE[wv] = (as > 0.0 && as < C) ? err[s] : E;
                }
            }
            __syncthreads();
            SMO_STAMP(2)   // the sixteen errors of the batch
            const int B = n - i < SMO_B ? (int)(n - i) : SMO_B;
            // fast path: wave 0 examines the whole batch at once when it holds no possible violator
            if (wv == 0) {
                const bool in = lane < B;
                const bool fast = svc_quiet_batch(S, err, C, tol, i, B, in, in ? S.ba[lane] : 0.0, in ? S.by[lane] : 0.0,
                                                  in ? S.bE[lane] : 0.0);
                if (lane == 0) S.fast = fast ? 1 : 0;
            }
            __syncthreads();
            SMO_STAMP(3)   // quiet-batch test
            if (tid == 0 && !S.fast) {
                int used = B;
                S.go = 0;
                for (int w = 0; w < B; ++w) {
                    if (svc_examine(S, K, y, a, err, C, tol, i + w, S.ba[w], S.by[w], S.bE[w])) {
                        S.go = 1;
                        used = w + 1;
                        break;
                    }
                }
                S.used = used;
            }
            __syncthreads();
            SMO_STAMP(4)   // sequential walk of the batch (incl. the pair solve)
            if (S.go) {
                svc_after_step(S, L, G, K, y, a, err, C, P);
                ++changed;
                ++steps;
                SMO_STAMP(5)   // list edit, error cache, thresholds
            }
            if (S.fail) break;
            i += S.used;
            __syncthreads();   // S.used / S.go are rewritten by thread 0 in the next round
        }
#ifdef BQ_SMO_STAMPS
        if (tid == 0) {
            // phases: 0 wide quiet round, 1 wide attempt that fell through, 2 the sixteen errors of a batch, 3 quiet-batch
            // test, 4 sequential walk + pair solve, 5 after a step (list edit, error cache, thresholds), 7 loop tail
            for (int k = 0; k < 8; ++k)
                if (stamp_cnt[k])
                    printf("[smo stamps] phase %d: %llu x %.2f us = %.2f ms\n", k, stamp_cnt[k],
                           0.01 * (double)stamp_acc[k] / (double)stamp_cnt[k], 1e-5 * (double)stamp_acc[k]);
        }
#endif
    } else {
        long long last = -1;
        while (true) {
            if (tid == 0) {
                S.go = 0;
                S.stop = 1;
                for (int q = sup_lower_bound(L, G, S.nnz, last + 1); q < S.nnz; ++q) {
                    const long long j = sup_idx(L, G, q);
                    const double aj = a[j];
                    if (aj > 0.0 && aj < C) {
                        S.stop = 0;
                        S.i2 = j;
                        S.go = svc_examine(S, K, y, a, err, C, tol, j, aj, y[j], err[j]) ? 1 : 0;
                        break;
                    }
                }
            }
            __syncthreads();
            if (S.stop) break;
            if (S.go) {
                svc_after_step(S, L, G, K, y, a, err, C, P);
                ++changed;
                ++steps;
            }
            if (S.fail) break;
            if (S.b_up > S.b_low - 2 * tol) {   // optimality on the free set (smo.py:339-342)
                changed = 0;
                break;
            }
            last = S.i2;
            __syncthreads();
        }
    }
    if (tid == 0) {
        sc->b_up = S.b_up;
        sc->b_low = S.b_low;
        sc->i_up = S.i_up;
        sc->i_low = S.i_low;
        sc->steps += steps;
        sc->changed = changed;
        sc->err_flag = S.fail;
        int next_all = sweep_all ? 0 : (changed == 0 ? 1 : 0);
        sc->sweep_all = next_all;
        sc->outer += 1;
        sc->finished = (S.fail || !(changed > 0 || next_all)) ? 1 : 0;
        if (P.helpers) {   // every exit path of the walker ends here (or in the early return above): release the helpers
            st_rlx(&P.ctl[SPEC_WIN], 0u);
            st_rel(&P.ctl[SPEC_DONE], P.epoch);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// regression (smo.py:447-758)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int svr_kind(double p, double m, double C) {
    if ((p > 0.0 && p < C) || (m > 0.0 && m < C)) return 0;
    if (p == 0.0 && m == 0.0) return 1;
    if (p == 0.0 && m == C) return 2;
    if (p == C && m == 0.0) return 3;
    return -1;
}
__device__ __forceinline__ double svr_clip(double v, double C) {
    return v > C - __dmul_rn(1e-10, C) ? C : (v <= __dmul_rn(1e-10, C) ? 0.0 : v);
}
// minimiser of the pair sub-problem on [L, H]: Newton step when eta > 0, else the better end point
__device__ __forceinline__ double svr_solve(double L, double H, double base, double num, double lin, double eta) {
    if (eta > 0.0) return fmax(L, fmin(base + num / eta, H));
    return __dmul_rn(L, lin) > __dmul_rn(H, lin) ? L : H;
}
__device__ __forceinline__ long long svr_pick(const SmoShared &S, double vlow, double vup, double tol) {
    if (S.b_low - vlow > 2 * tol) return (vlow - S.b_up > S.b_low - vlow) ? S.i_up : S.i_low;
    if (vup - S.b_up > 2 * tol) return (S.b_low - vup > vup - S.b_up) ? S.i_low : S.i_up;
    return -1;
}

// thread 0: examine sample i2 (smo.py:677-758) and, if it violates, solve the pair (:447-600).  E2 is its error (cached,
// or just formed).  Returns true when a pair step was taken; its data is left in S for svr_after_step.
template <typename T>
__device__ __forceinline__ bool svr_examine(SmoShared &S, const KView<T> &K, const double *y, double *ap, double *an,
                                            double *err, double C, double eps, double tol, long long i2, double p2,
                                            double m2, double E2) {
    const int k2 = svr_kind(p2, m2, C);
    if (k2 != 0) {
        err[i2] = E2;
        if (k2 == 1) {
            if (E2 + eps < S.b_up) {
                S.b_up = E2 + eps;
                S.i_up = i2;
            } else if (E2 - eps > S.b_low) {
                S.b_low = E2 - eps;
                S.i_low = i2;
            }
        } else if (k2 == 2 && E2 + eps > S.b_low) {
            S.b_low = E2 + eps;
            S.i_low = i2;
        } else if (k2 == 3 && E2 - eps < S.b_up) {
            S.b_up = E2 - eps;
            S.i_up = i2;
        }
    }
    long long i1 = -1;
    if (k2 == 0) {
        if (p2 > 0.0 && p2 < C)
            i1 = svr_pick(S, E2 - eps, E2 - eps, tol);
        else if (m2 > 0.0 && m2 < C)
            i1 = svr_pick(S, E2 + eps, E2 + eps, tol);
    } else if (k2 == 1) {
        i1 = svr_pick(S, E2 + eps, E2 - eps, tol);
    } else if (k2 == 2) {
        if ((E2 + eps) - S.b_up > 2 * tol) i1 = S.i_up;
    } else if (k2 == 3) {
        if (S.b_low - (E2 - eps) > 2 * tol) i1 = S.i_low;
    } else {
        S.fail = 1;   // 'the index could not be found'
        return false;
    }
    if (i1 < 0 || i1 == i2) return false;
    const double p1o = ap[i1], m1o = an[i1];
    double p1 = p1o, m1 = m1o, q2 = p2, r2 = m2;   // q2 / r2: working copies of alpha2+ / alpha2-
    const double k11 = K.at(i1, i1), k22 = K.at(i2, i2), k12 = K.at(i1, i2);
    const double eta = fmax(k11 + k22 - 2 * k12, 0.0);
    const double gamma = p1 - m1 + q2 - r2;
    double dE = err[i1] - E2;
    bool tried[4] = {false, false, false, false};
    bool moved = false, done = false;
    while (!done) {   // at most three rounds (smo.py:471)
        if (!tried[0] && (p1 > 0 || (m1 == 0 && dE > 0)) && (q2 > 0 || (r2 == 0 && dE < 0))) {
            const double L = fmax(0.0, gamma - C), H = fmin(C, gamma);
            if (L < H) {
                const double v2 = svr_solve(L, H, q2, -dE, -dE, eta), v1 = p1 - (v2 - q2);
                if (fabs(v1 - p1) > 1e-12 || fabs(v2 - q2) > 1e-12) {
                    p1 = v1;
                    q2 = v2;
                    moved = true;
                }
            } else {
                done = true;
            }
            tried[0] = true;
        } else if (!tried[1] && (p1 > 0 || (m1 == 0 && dE > 2 * eps)) &&
                   (r2 > 0 || (q2 == 0 && dE > 2 * eps))) {
            const double L = fmax(0.0, -gamma), H = fmin(C, -gamma + C);
            if (L < H) {
                const double v2 = svr_solve(L, H, r2, dE - 2 * eps, -2 * eps + dE, eta), v1 = p1 + (v2 - r2);
                if (fabs(v1 - p1) > 1e-12 || fabs(v2 - r2) > 1e-12) {
                    p1 = v1;
                    r2 = v2;
                    moved = true;
                }
            } else {
                done = true;
            }
            tried[1] = true;
        } else if (!tried[2] && (m1 > 0 || (p1 == 0 && dE < -2 * eps)) &&
                   (q2 > 0 || (r2 == 0 && dE < -2 * eps))) {
            const double L = fmax(0.0, gamma), H = fmin(C, C + gamma);
            if (L < H) {
                const double v2 = svr_solve(L, H, q2, -(dE + 2 * eps), -(2 * eps + dE), eta);
                const double v1 = m1 + (v2 - q2);
                if (fabs(v1 - m1) > 1e-12 || fabs(v2 - q2) > 1e-12) {
                    m1 = v1;
                    q2 = v2;
                    moved = true;
                }
            } else {
                done = true;
            }
            tried[2] = true;
        } else if (!tried[3] && (m1 > 0 || (p1 == 0 && dE < 0)) && (r2 > 0 || (q2 == 0 && dE > 0))) {
            const double L = fmax(0.0, -gamma - C), H = fmin(C, -gamma);
            if (L < H) {
                const double v2 = svr_solve(L, H, r2, dE, dE, eta), v1 = m1 - (v2 - r2);
                if (fabs(v1 - m1) > 1e-12 || fabs(v2 - r2) > 1e-12) {
                    m1 = v1;
                    r2 = v2;
                    moved = true;
                }
            } else {
                done = true;
            }
            tried[3] = true;
        } else {
            done = true;
        }
        dE += __dmul_rn(eta, (q2 - r2) - (p2 - m2));
    }
    if (moved) {
        const double c1 = (p1o - m1o) - (p1 - m1), c2 = (p2 - m2) - (q2 - r2);
        err[i1] = err[i1] + (__dmul_rn(c1, k11) + __dmul_rn(c2, k12));
        err[i2] = E2 + (__dmul_rn(c1, k12) + __dmul_rn(c2, k22));
        const double np1 = svr_clip(p1, C), nm1 = svr_clip(m1, C), np2 = svr_clip(q2, C), nm2 = svr_clip(r2, C);
        ap[i1] = np1;
        an[i1] = nm1;
        ap[i2] = np2;
        an[i2] = nm2;
        S.mem1 = np1 != 0.0 || nm1 != 0.0;
        S.mem2 = np2 != 0.0 || nm2 != 0.0;
        S.cf1 = np1 - nm1;
        S.cf2 = np2 - nm2;
        S.c1 = c1;
        S.c2 = c2;
        S.i1 = i1;
        S.i2 = i2;
        return true;
    }
    return false;
}

// all threads, after a pair step: support list, error cache of the free set, thresholds (smo.py:601-675)
template <typename T>
__device__ __forceinline__ void svr_after_step(SmoShared &S, SupList &L, const SupGlobal &G, const KView<T> &K,
                                               double *ap, double *an, double *err, double C, double eps,
                                               const SmoSpec &P) {
    const int tid = threadIdx.x;
    const long long i1 = S.i1, i2 = S.i2;
    spec_begin(P, S);
    sup_apply(L, G, S, (int)i1, S.mem1 != 0, S.cf1);
    sup_apply(L, G, S, (int)i2, S.mem2 != 0, S.cf2);
    spec_end(P, S, i2 + 1);   // a full sweep continues behind the examined sample
    const double c1 = S.c1, c2 = S.c2;
    ValIdx hi{-DBL_MAX, -1}, lo{DBL_MAX, -1};
    for (int q = tid; q < S.nnz; q += SMO_T) {
        const int64_t j = sup_idx(L, G, q);
        const double pj = ap[j], mj = an[j];
        const bool pin = pj > 0.0 && pj < C, nin = mj > 0.0 && mj < C;
        if (pin || nin) {
            double e = err[j];
            if (j != i1 && j != i2) {
                e += __dmul_rn(c1, K.at(i1, j)) + __dmul_rn(c2, K.at(i2, j));
                err[j] = e;
            }
            // smo.py:641-653: alpha+ inside is tried first, alpha- inside only when that test fails
            if (pin && e - eps > hi.v)
                hi = ValIdx{e - eps, (long long)j};
            else if (nin && e + eps > hi.v)
                hi = ValIdx{e + eps, (long long)j};
            if (pin && e - eps < lo.v)
                lo = ValIdx{e - eps, (long long)j};
            else if (nin && e + eps < lo.v)
                lo = ValIdx{e + eps, (long long)j};
        }
    }
    hi = smo_bbest(hi, +1, S.bv, S.bi);
    lo = smo_bbest(lo, -1, S.bv, S.bi);
    if (tid == 0) {
        S.b_up = DBL_MAX;
        S.b_low = -DBL_MAX;
        S.i_up = -1;
        S.i_low = -1;
        if (hi.i >= 0 && hi.v > S.b_low) {
            S.b_low = hi.v;
            S.i_low = hi.i;
        }
        if (lo.i >= 0 && lo.v < S.b_up) {
            S.b_up = lo.v;
            S.i_up = lo.i;
        }
        const long long pair[2] = {i1, (long long)i2};
        for (int k = 0; k < 2; ++k) {   // smo.py:654-668
            const long long i = pair[k];
            const int ki = svr_kind(ap[i], an[i], C);
            const double ei = err[i];
            if (ki == 0) continue;
            if (ki == 2 && ei + eps > S.b_low) {
                S.b_low = ei + eps;
                S.i_low = i;
            } else if (ki == 1 && ei - eps > S.b_low) {
                S.b_low = ei - eps;
                S.i_low = i;
            }
            if (ki == 3 && ei - eps < S.b_up) {
                S.b_up = ei - eps;
                S.i_up = i;
            } else if (ki == 1 && ei + eps < S.b_up) {
                S.b_up = ei + eps;
                S.i_up = i;
            }
        }
        if (S.i_low < 0 || S.i_up < 0) S.fail = 1;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(SMO_T) void smo_svr_kernel(KView<T> K, int64_t n, const double *__restrict__ y,
                                                        double *ap, double *an, double *err, SupGlobal G,
                                                        double C, double eps, double tol, bq_smo_scal *sc, SmoSpec P) {
    if (blockIdx.x != 0) {
        smo_helper(K, n, G, P);
        return;
    }
    __shared__ SmoShared S;
    __shared__ SupList L;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) {
        S.b_up = sc->b_up;
        S.b_low = sc->b_low;
        S.i_up = sc->i_up;
        S.i_low = sc->i_low;
        S.fail = 0;
        S.go = 0;
        S.ver = P.helpers ? ld_rlx(&P.ctl[SPEC_VER]) : 0u;
        S.win = 0;
    }
    __syncthreads();
    if (sc->finished) {
        if (tid == 0 && P.helpers) st_rel(&P.ctl[SPEC_DONE], P.epoch);
        return;
    }
    const bool sweep_all = sc->sweep_all != 0;
    long long changed = 0, steps = 0;
    spec_begin(P, S);
    smo_rebuild(n, G.nz, S, [&](int64_t j) { return ap[j] != 0.0 || an[j] != 0.0; });   // a superset of the free set
    sup_fill(L, G, S, [&](int j) { return ap[j] - an[j]; });
    spec_end(P, S, 0);
    if (sweep_all) {
        int64_t i = 0;
        while (i < n) {
            if (tid == 0 && P.helpers) {
                const unsigned int cap = (unsigned int)(P.helpers * SMO_B);
                if (!S.go) S.win = S.win * 2u < cap ? S.win * 2u : cap;
                st_rlx64((long long *)&P.ctl[SPEC_POS], (long long)i);
                st_rlx(&P.ctl[SPEC_WIN], S.win);
            }
            const int64_t s = i + wv;
            if (s < n) {   // wave-uniform
                const double ps = ap[s], ms = an[s];
                const bool cached = svr_kind(ps, ms, C) == 0;
                double E = 0.0;
                if (!cached) E = y[s] - wave_dot(K, L, G, S.nnz, s, P, S.ver, S.win > (unsigned int)SMO_B);
                if (lane == 0) {
                    S.ba[wv] = ps;
                    S.bm[wv] = ms;
                    S.bE[wv] = cached ? err[s] : E;
                }
            }
            __syncthreads();
            if (tid == 0) {
                const int B = n - i < SMO_B ? (int)(n - i) : SMO_B;
                int used = B;
                S.go = 0;
                for (int w = 0; w < B; ++w) {
                    const bool took = svr_examine(S, K, y, ap, an, err, C, eps, tol, i + w, S.ba[w], S.bm[w], S.bE[w]);
                    if (took || S.fail) {
                        S.go = took ? 1 : 0;
                        used = w + 1;
                        break;
                    }
                }
                S.used = used;
            }
            __syncthreads();
            if (S.go) {
                svr_after_step(S, L, G, K, ap, an, err, C, eps, P);
                ++changed;
                ++steps;
            }
            if (S.fail) break;
            i += S.used;
            __syncthreads();   // S.used / S.go are rewritten by thread 0 in the next round
        }
    } else {
        long long last = -1;
        while (true) {
            if (tid == 0) {
                S.go = 0;
                S.stop = 1;
                for (int q = sup_lower_bound(L, G, S.nnz, last + 1); q < S.nnz; ++q) {
                    const long long j = sup_idx(L, G, q);
                    const double pj = ap[j], mj = an[j];
                    if (svr_kind(pj, mj, C) == 0) {
                        S.stop = 0;
                        S.i2 = j;
                        S.go = svr_examine(S, K, y, ap, an, err, C, eps, tol, j, pj, mj, err[j]) ? 1 : 0;
                        break;
                    }
                }
            }
            __syncthreads();
            if (S.stop) break;
            if (S.go) {
                svr_after_step(S, L, G, K, ap, an, err, C, eps, P);
                ++changed;
                ++steps;
            }
            if (S.fail) break;
            if (S.b_up > S.b_low - 2 * tol) {
                changed = 0;
                break;
            }
            last = S.i2;
            __syncthreads();
        }
    }
    if (tid == 0) {
        sc->b_up = S.b_up;
        sc->b_low = S.b_low;
        sc->i_up = S.i_up;
        sc->i_low = S.i_low;
        sc->steps += steps;
        sc->changed = changed;
        sc->err_flag = S.fail;
        int next_all = sweep_all ? 0 : (changed == 0 ? 1 : 0);
        sc->sweep_all = next_all;
        sc->outer += 1;
        sc->finished = (S.fail || !(changed > 0 || next_all)) ? 1 : 0;
        if (P.helpers) {   // every exit path of the walker ends here (or in the early return above): release the helpers
            st_rlx(&P.ctl[SPEC_WIN], 0u);
            st_rel(&P.ctl[SPEC_DONE], P.epoch);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int bq_smo_destroy(bq_smo *s) {
    if (s == nullptr) return BQ_OK;
    hipSetDevice(s->p->ctx->device);
    hipStreamSynchronize(s->p->ctx->stream);
    for (void *ptr : {(void *)s->y, (void *)s->a, (void *)s->am, (void *)s->err, (void *)s->nz, (void *)s->cf, (void *)s->sc,
                      (void *)s->ctl, (void *)s->spec_res})
        if (ptr) hipFree(ptr);
    bq_problem *p = s->p;
    delete s;
    bq_problem_unref(p);
    return BQ_OK;
}

extern "C" int bq_smo_create(bq_problem *p, int task, const double *y, double C, double epsilon, double tol,
                             bq_smo **out) {
    BQ_ARG(p && y && out, "NULL argument");
    BQ_ARG(task == BQ_SVC || task == BQ_SVR, "task must be BQ_SVC or BQ_SVR");
    BQ_ARG(p->kernel >= 0 && !p->streamed, "SMO needs a kernel-built problem with a resident Gram panel");
    BQ_ARG(C > 0.0, "C must be > 0");
    BQ_ARG(epsilon >= 0.0, "epsilon must be >= 0");
    BQ_ARG(tol > 0.0, "tol must be > 0");
    if (p->ctx->world > 1) {
        bq_set_error("SMO walks the samples sequentially over the whole panel: use a single-rank context (replicas only)");
        return BQ_ERR_BADARG;
    }
    bq_ctx *c = p->ctx;
    const int64_t n = p->n;
    int64_t first_pos = -1, first_neg = -1;
    if (task == BQ_SVC) {
        for (int64_t i = 0; i < n; ++i) {
            BQ_ARG(y[i] == 1.0 || y[i] == -1.0, "labels must be +1 / -1");
            if (y[i] == 1.0 && first_pos < 0) first_pos = i;
            if (y[i] == -1.0 && first_neg < 0) first_neg = i;
        }
        BQ_ARG(first_pos >= 0 && first_neg >= 0, "both classes are needed");
    }
    BQ_HIP(hipSetDevice(c->device));
    bq_smo *s = new bq_smo();
    s->p = p;
    p->refs += 1;
    s->task = task;
    s->n = n;
    s->C = C;
    s->eps = epsilon;
    s->tol = tol;
    hipError_t e = hipSuccess;
    for (double **v : {&s->y, &s->a, &s->am, &s->err}) {
        if (e == hipSuccess) e = hipMalloc(v, sizeof(double) * n);
        if (e == hipSuccess) e = hipMemsetAsync(*v, 0, sizeof(double) * n, c->stream);
    }
    if (e == hipSuccess) e = hipMalloc(&s->nz, sizeof(int) * n);
    if (e == hipSuccess) e = hipMalloc(&s->cf, sizeof(double) * n);
    if (e == hipSuccess) e = hipMalloc(&s->sc, sizeof(bq_smo_scal));
    // helper workgroups of the full sweeps: hook smo_helpers (0 = the walker alone).  Default 48: measured at n = 100 000
    // the sweeps take 455 / 357 / 344 / 355 / 365 / 380 / 433 / 479 ms with 16 / 32 / 48 / 64 / 96 / 128 / 192 / 255
    // helpers — past ~48 the extra pollers of the control line cost the walker more than the extra gather rate brings
    int cus = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
    s->helpers = std::min(48, cus / 4);
    s->helpers = (int)bq_hook_value("smo_helpers", (double)s->helpers);
    s->helpers = std::max(0, std::min(s->helpers, cus - 1));
    if (s->helpers < SMO_T / 64) s->helpers = 0;   // the first batch after a pair step wants one CU per sample
    if (e == hipSuccess && s->helpers) {
        const unsigned int ctl0[SPEC_WORDS] = {2u};   // version 2: a zeroed result (version 0) never matches
        e = hipMalloc(&s->ctl, sizeof(ctl0));
        if (e == hipSuccess) e = hipMemcpyAsync(s->ctl, ctl0, sizeof(ctl0), hipMemcpyHostToDevice, c->stream);
        if (n * 16 >= ((int64_t)1 << 31)) e = hipErrorInvalidValue;   // 32-bit byte offsets of the result buffer
        if (e == hipSuccess) e = hipMalloc(&s->spec_res, (size_t)16 * n);
        if (e == hipSuccess) e = hipMemsetAsync(s->spec_res, 0, (size_t)16 * n, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // ctl0 lives on this stack frame
    }
    if (e == hipSuccess) e = hipMemcpyAsync(s->y, y, sizeof(double) * n, hipMemcpyHostToDevice, c->stream);
    memset(&s->host, 0, sizeof(s->host));
    s->host.sweep_all = 1;
    if (task == BQ_SVC) {   // smo.py:119-128
        s->host.b_up = -1.0;
        s->host.b_low = 1.0;
        s->host.i_up = first_pos;
        s->host.i_low = first_neg;
        const double m1 = -1.0, p1 = 1.0;
        if (e == hipSuccess) e = hipMemcpyAsync(s->err + first_pos, &m1, sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s->err + first_neg, &p1, sizeof(double), hipMemcpyHostToDevice, c->stream);
    } else {                // smo.py:442-445
        s->host.b_up = y[0] + epsilon;
        s->host.b_low = y[0] - epsilon;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(s->sc, &s->host, sizeof(bq_smo_scal), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        bq_set_error("SMO setup failed: %s", hipGetErrorString(e));
        bq_smo_destroy(s);
        return BQ_ERR_HIP;
    }
    *out = s;
    return BQ_OK;
}

extern "C" int bq_smo_run(bq_smo *s, int64_t max_outer, int64_t *outer_iters, int *finished) {
    BQ_ARG(s && outer_iters && finished, "NULL argument");
    BQ_ARG(max_outer > 0, "max_outer must be > 0");
    bq_problem *p = s->p;
    bq_ctx *c = p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    for (int64_t k = 0; k < max_outer && !s->host.finished; ++k) {
        // full sweeps take the helper workgroups along; free-set sweeps walk the (short) support list alone
        const int helpers = s->host.sweep_all ? s->helpers : 0;
        const SmoSpec P{s->ctl, (smo_u4 *)s->spec_res, (unsigned int)(16 * s->n), helpers ? ++s->epoch : 0u, helpers};
        const dim3 grid(1 + helpers);
        const SupGlobal G{s->nz, s->cf};
        const int64_t ld = p->symmetric ? 0 : p->ld;
        if (s->task == BQ_SVC) {
            if (p->storage == BQ_F64)
                smo_svc_kernel<double><<<grid, SMO_T, 0, c->stream>>>(KView<double>{(const double *)p->panel, ld}, s->n, s->y, s->a,
                                                                       s->err, G, s->C, s->tol, s->sc, P);
            else
                smo_svc_kernel<float><<<grid, SMO_T, 0, c->stream>>>(KView<float>{(const float *)p->panel, ld}, s->n, s->y, s->a,
                                                                     s->err, G, s->C, s->tol, s->sc, P);
        } else {
            if (p->storage == BQ_F64)
                smo_svr_kernel<double><<<grid, SMO_T, 0, c->stream>>>(KView<double>{(const double *)p->panel, ld}, s->n, s->y, s->a,
                                                                       s->am, s->err, G, s->C, s->eps, s->tol, s->sc, P);
            else
                smo_svr_kernel<float><<<grid, SMO_T, 0, c->stream>>>(KView<float>{(const float *)p->panel, ld}, s->n, s->y, s->a,
                                                                     s->am, s->err, G, s->C, s->eps, s->tol, s->sc, P);
        }
        BQ_HIP(hipGetLastError());
        BQ_HIP(hipMemcpyAsync(&s->host, s->sc, sizeof(bq_smo_scal), hipMemcpyDeviceToHost, c->stream));
        BQ_HIP(hipStreamSynchronize(c->stream));
        if (s->host.err_flag) {
            bq_set_error("SMO reached an unexpected status (no threshold index)");   // smo.py:270-271, :756
            return BQ_ERR_NONFINITE;
        }
    }
    *outer_iters = s->host.outer;
    *finished = s->host.finished;
    return BQ_OK;
}

extern "C" int bq_smo_get(bq_smo *s, int what, double *out) {
    BQ_ARG(s && out, "NULL argument");
    bq_ctx *c = s->p->ctx;
    BQ_HIP(hipSetDevice(c->device));
    switch (what) {
        case BQ_SMO_ALPHAS:
            BQ_HIP(hipMemcpyAsync(out, s->a, sizeof(double) * s->n, hipMemcpyDeviceToHost, c->stream));
            if (s->task == BQ_SVR)
                BQ_HIP(hipMemcpyAsync(out + s->n, s->am, sizeof(double) * s->n, hipMemcpyDeviceToHost, c->stream));
            break;
        case BQ_SMO_ERRORS:
            BQ_HIP(hipMemcpyAsync(out, s->err, sizeof(double) * s->n, hipMemcpyDeviceToHost, c->stream));
            break;
        case BQ_SMO_SCALARS:
            out[0] = s->host.b_up;
            out[1] = s->host.b_low;
            out[2] = (double)s->host.i_up;
            out[3] = (double)s->host.i_low;
            out[4] = (double)s->host.steps;
            out[5] = s->task == BQ_SVC ? -(s->host.b_low + s->host.b_up) / 2 : (s->host.b_low + s->host.b_up) / 2;
            break;
        case BQ_SMO_STATS: {
            unsigned int w[SPEC_WORDS] = {0};
            if (s->ctl) BQ_HIP(hipMemcpy(w, s->ctl, sizeof(w), hipMemcpyDeviceToHost));
            out[0] = (double)s->helpers;
            out[1] = (double)w[SPEC_DELIVERED];
            out[2] = (double)w[SPEC_REJ_HASH];
            out[3] = (double)w[SPEC_REJ_CHK];
            break;
        }
        default:
            bq_set_error("bad argument: what");
            return BQ_ERR_BADARG;
    }
    BQ_HIP(hipStreamSynchronize(c->stream));
    return BQ_OK;
}
