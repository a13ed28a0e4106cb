// tools/barrier_probe.hip — what does a grid-wide barrier cost on this chip, against a kernel boundary?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_probe tools/barrier_probe.hip && /tmp/barrier_probe
//
// The question behind it: a short iteration (BASELINE config 2: 0.257 ms of panel product, 29 us of three
// dependent small kernels) could run as ONE cooperative kernel whose phases are separated by grid barriers instead of kernel
// boundaries — worth it only if a barrier over every resident workgroup is clearly cheaper than the ~5 us a dependent launch costs.
// Measured here, for grids of 256 / 512 / 1024 workgroups of 256 threads:
//   flat      one arrival counter (agent-scope atomic add), everybody spins on the generation word
//   tree      per-group arrival counters (8 groups by blockIdx % 8 — the XCD a workgroup lands on), the last arrival of a group
//             adds to a root counter, the last group flips the generation word
//   launches  the same number of EMPTY dependent kernels on one stream (the kernel-boundary cost the barrier would replace)
// Every barrier has a bounded spin (it gives up after ~2 s of wall clock and flags it), so a mistake cannot hang the box.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

struct bar_state {
    unsigned int count;        // flat: arrivals of the current generation
    unsigned int gen;          // generation word everybody spins on
    unsigned int root;         // tree: groups that have arrived
    unsigned int pad[13];
    unsigned int group[8][16]; // tree: arrivals per group (64-byte apart)
    unsigned int timeout;      // set when a spin gave up
};

__device__ __forceinline__ bool spin_until(unsigned int *word, unsigned int want, long long t_end) {
    unsigned int it = 0;
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 0x3ff) == 0 && wall_clock64() > t_end) return false;
    }
    return true;
}

// flat barrier; `g` = the generation this call completes (1, 2, ...)
__device__ __forceinline__ void barrier_flat(bar_state *b, unsigned int g, unsigned int nwg, long long t_end) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int prev = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == g * nwg - 1) {
            __hip_atomic_store(&b->gen, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!spin_until(&b->gen, g, t_end)) {
            b->timeout = 1;
        }
        __threadfence();
    }
    __syncthreads();
}

__device__ __forceinline__ void barrier_tree(bar_state *b, unsigned int g, unsigned int nwg, long long t_end) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int grp = blockIdx.x & 7;
        const unsigned int members = (nwg + 7 - grp) / 8;   // workgroups with blockIdx % 8 == grp
        const unsigned int prev = __hip_atomic_fetch_add(&b->group[grp][0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        bool released = false;
        if (prev == g * members - 1) {
            const unsigned int r = __hip_atomic_fetch_add(&b->root, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (r == g * 8 - 1) {
                __hip_atomic_store(&b->gen, g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                released = true;
            }
        }
        if (!released && !spin_until(&b->gen, g, t_end)) b->timeout = 1;
        __threadfence();
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void barrier_kernel(bar_state *b, int reps, double *sink, long long ticks_2s) {
    const long long t_end = wall_clock64() + ticks_2s;
    double acc = 0.0;
    for (int k = 1; k <= reps; ++k) {
        // a token amount of work between barriers: one store + one load of a word another workgroup wrote
        sink[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + k;
        if (MODE == 0)
            barrier_flat(b, (unsigned)k, gridDim.x, t_end);
        else if (MODE == 1)
            barrier_tree(b, (unsigned)k, gridDim.x, t_end);
        else
            cooperative_groups::this_grid().sync();
        acc += sink[(size_t)((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
        if (b->timeout) break;
    }
    if (acc == -1.0) sink[0] = acc;
}

// the library's "last block closes" pattern without any waiting: per block one store, a fence, one agent-scope atomic on a shared ticket;
// the block that takes the last ticket reads every block's value.  How long is a launch of G such blocks?
__global__ __launch_bounds__(256) void ticket_kernel(double *part, unsigned int *ticket, double *out, int k) {
    __shared__ int last;
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (double)(blockIdx.x + k);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double a = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256) a += part[i];
    if (threadIdx.x == 0) {
        out[0] = a;
        *ticket = 0;
    }
}
// the same blocks without the ticket (what the launch costs by itself at this grid size)
__global__ __launch_bounds__(256) void noticket_kernel(double *part, int k) {
    if (threadIdx.x == 0) part[blockIdx.x] = (double)(blockIdx.x + k);
}

__global__ void empty_kernel(double *sink, int k) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && k < 0) sink[0] = 1.0;
}

// a dependent kernel that also leaves dirty lines behind (what the real boundary pays: release at the end, acquire at the start)
__global__ __launch_bounds__(256) void dirty_kernel(double *buf, size_t n, int k) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = buf[i] + k;
}

int main() {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    if (rate_khz <= 0) rate_khz = 100000;
    printf("device: %s, %d CUs, cooperative launch %d, wall clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.cooperativeLaunch, rate_khz);
    bar_state *b;
    double *sink;
    CK(hipMalloc(&b, sizeof(bar_state)));
    CK(hipMalloc(&sink, sizeof(double) * 2048 * 256));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 2000;
    const long long ticks_2s = 2LL * rate_khz * 1000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int nwg : {256, 512, 1024}) {
            int per_cu = 0;
            void *fn = mode == 0 ? (void *)barrier_kernel<0> : mode == 1 ? (void *)barrier_kernel<1> : (void *)barrier_kernel<2>;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0));
            if (per_cu * prop.multiProcessorCount < nwg) {
                printf("mode %d grid %d: not co-resident (%d per CU), skipped\n", mode, nwg, per_cu);
                continue;
            }
            for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
                CK(hipMemsetAsync(b, 0, sizeof(bar_state), st));
                int r = reps;
                long long t2 = ticks_2s;
                void *args[] = {&b, &r, &sink, &t2};
                CK(hipEventRecord(e0, st));
                CK(hipLaunchCooperativeKernel(fn, dim3(nwg), dim3(256), args, 0, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                bar_state h;
                CK(hipMemcpy(&h, b, sizeof(h), hipMemcpyDeviceToHost));
                if (pass == 1)
                    printf("%-8s grid %4d x 256: %7.3f us per barrier (%d barriers in %.3f ms)%s\n",
                           mode == 0 ? "flat" : mode == 1 ? "tree" : "cg.sync", nwg, 1e3 * ms / reps, reps, ms, h.timeout ? "  TIMED OUT" : "");
                if (h.timeout) break;
            }
        }
    }
    // a launch of G blocks that meet in a ticket against the same launch without one
    {
        unsigned int *ticket;
        CK(hipMalloc(&ticket, sizeof(unsigned int)));
        CK(hipMemsetAsync(ticket, 0, sizeof(unsigned int), st));
        for (int g : {32, 128, 512, 1024, 2048, 4096}) {
            for (int kind = 0; kind < 2; ++kind)
                for (int pass = 0; pass < 2; ++pass) {
                    CK(hipEventRecord(e0, st));
                    for (int k = 0; k < reps; ++k) {
                        if (kind == 0)
                            ticket_kernel<<<g, 256, 0, st>>>(sink, ticket, sink + 8192, k);
                        else
                            noticket_kernel<<<g, 256, 0, st>>>(sink, k);
                    }
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    float ms = 0.f;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (pass == 1)
                        printf("%-8s grid %4d x 256: %7.3f us per dependent launch\n", kind == 0 ? "ticket" : "noticket", g, 1e3 * ms / reps);
                }
        }
        CK(hipFree(ticket));
    }
    // the kernel boundary it would replace
    for (int kind = 0; kind < 2; ++kind) {
        const size_t n = (size_t)4 << 20;   // 32 MB of doubles: every launch leaves dirty lines behind
        double *buf = nullptr;
        if (kind == 1) {
            CK(hipMalloc(&buf, sizeof(double) * n));
            CK(hipMemsetAsync(buf, 0, sizeof(double) * n, st));
        }
        for (int pass = 0; pass < 2; ++pass) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < reps; ++k) {
                if (kind == 0)
                    empty_kernel<<<512, 256, 0, st>>>(sink, k);
                else
                    dirty_kernel<<<512, 256, 0, st>>>(buf, (size_t)512 * 256 * 4, k);   // 4 MB touched per launch
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass == 1)
                printf("%-8s grid  512 x 256: %7.3f us per dependent launch (%d launches in %.3f ms)\n", kind == 0 ? "empty" : "4MB r/w", 1e3 * ms / reps, reps, ms);
        }
        if (buf) CK(hipFree(buf));
    }
    return 0;
}
