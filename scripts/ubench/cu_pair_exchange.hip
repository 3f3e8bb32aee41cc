// Micro-benchmark for VERDICT r3 item 2(b): what a 2-CU / 4-CU COLUMN SPLIT of the persistent LSTM kernels would trade.
//
// Today one 512-thread workgroup per CU owns 32 sequences and streams the WHOLE fp16 weight matrix [x | h] -> 4H
// (720 KB) from L2 in every one of the T = 4 steps.  Split over G CUs of one XCD, a workgroup would stream 1/G of the
// columns (360 / 180 KB) for G x the rows and, after each step's pointwise epilogue, publish its slice of h_t
// (rows x H/G fp16 = 16 KB) and gather the other G - 1 slices before the next step's h-part can start.
//
// This program times exactly that trade, nothing else: per step
//   stream  W bytes of a read-only buffer through registers (16-B loads, 8 in flight per lane: the weight stream)
//   publish 16 KB with `sc1` write-through stores, s_waitcnt vmcnt(0), workgroup barrier, `sc1` flag store   (split only)
//   gather  (G - 1) x 16 KB from the partners with `sc1` loads into LDS after polling their flags               (split only)
// for 256 workgroups (one per CU, forced by 129 KB of LDS), 4 steps, partners = blockIdx ^ 8, ^ 16 (same XCD under the
// observed round-robin placement; correctness does not depend on it).  Polls are bounded: a partner that never arrives
// sets the `failed` flag instead of hanging the chip.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_pair_exchange scripts/ubench/cu_pair_exchange.hip && /tmp/cu_pair_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int THREADS = 512, STEPS = 4, SLICE_BYTES = 16 * 1024;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // (a register-class operand of inline asm)
__device__ __forceinline__ void load2_sc1(const uint4* pa, const uint4* pb, uint4& a, uint4& b) {      // both in flight, one wait
    u32x4 ra, rb;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n global_load_dwordx4 %1, %3, off sc1\n s_waitcnt vmcnt(0)"
                 : "=&v"(ra), "=&v"(rb) : "v"(pa), "v"(pb) : "memory");
    a = make_uint4(ra.x, ra.y, ra.z, ra.w);
    b = make_uint4(rb.x, rb.y, rb.z, rb.w);
}
__device__ __forceinline__ void store_sc1(uint4* p, uint4 v) {
    const u32x4 r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ unsigned load_flag(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_flag(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// G = 1: no exchange.  launch_tag: flags count up over launches so that nothing has to be cleared between them.
template <int G>
__global__ __launch_bounds__(THREADS) void k(const uint4* __restrict__ weights, long long stream_bytes, uint4* slices,
                                             unsigned* flags, unsigned launch_tag, unsigned long long* t_out,
                                             unsigned* failed, float* sink) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    unsigned long long t0 = wall_clock64();
    uint4 acc = make_uint4(0, 0, 0, 0);
    const long long n16 = stream_bytes / 16;
    // each workgroup streams its own window of the (L2-resident) weight buffer: column slice b % G of the matrix
    const uint4* wbase = weights + (long long)(b % G) * n16;
    for (int step = 0; step < STEPS; ++step) {
        // ---- weight stream: 8 independent 16-B loads in flight per lane
        for (long long i = tid; i + 7 * THREADS < n16; i += 8 * THREADS) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = wbase[i + u * THREADS];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
        }
        if (G > 1) {
            // ---- publish this workgroup's 16 KB slice of h_t (sc1 write-through), drain, barrier, flag
            uint4* mine = slices + ((long long)(step & 1) * gridDim.x + b) * (SLICE_BYTES / 16);
            for (int i = tid; i < SLICE_BYTES / 16; i += THREADS) store_sc1(mine + i, make_uint4(acc.x + i, b, step, launch_tag));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const unsigned want = launch_tag * STEPS + step + 1;
            if (tid == 0) store_flag(flags + b * 32, want);
            // ---- gather the partners' slices into LDS
            for (int g = 1; g < G; ++g) {
                const int partner = b ^ (8 * g);
                if (tid == 0) {
                    int spins = 0;
                    while (load_flag(flags + partner * 32) < want) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1 << 22)) { *failed = 1; break; }
                    }
                }
                __syncthreads();
                const uint4* theirs = slices + ((long long)(step & 1) * gridDim.x + partner) * (SLICE_BYTES / 16);
                static_assert(SLICE_BYTES / 16 == 2 * THREADS, "two 16-B pieces per thread and slice");
                uint4 va, vb;
                load2_sc1(theirs + tid, theirs + THREADS + tid, va, vb);
                lds[(g - 1) * (SLICE_BYTES / 16) + tid] = va;
                lds[(g - 1) * (SLICE_BYTES / 16) + THREADS + tid] = vb;
            }
            __syncthreads();
            acc.x ^= lds[tid].x;
        }
    }
    unsigned long long t1 = wall_clock64();
    if (tid == 0) t_out[b] = t1 - t0;
    if (acc.x == 0x12345u && acc.y == 7u) sink[0] = 1.0f;      // keep the stream alive
}

int main() {
    const int blocks = 256;
    const long long full = 720 * 1024;
    uint4 *w, *slices;
    unsigned *flags, *failed;
    unsigned long long* t;
    float* sink;
    CHECK(hipMalloc(&w, 4 * full));
    CHECK(hipMemset(w, 1, 4 * full));
    CHECK(hipMalloc(&slices, 2ll * blocks * SLICE_BYTES));
    CHECK(hipMalloc(&flags, blocks * 32 * sizeof(unsigned)));
    CHECK(hipMemset(flags, 0, blocks * 32 * sizeof(unsigned)));
    CHECK(hipMalloc(&failed, 4));
    CHECK(hipMemset(failed, 0, 4));
    CHECK(hipMalloc(&t, blocks * sizeof(unsigned long long)));
    CHECK(hipMalloc(&sink, 4));
    const size_t lds = 129 * 1024;      // one workgroup per CU, as the LSTM kernels
    CHECK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    unsigned tag = 0;
    printf("# 256 workgroups x 512 threads, 4 steps; per step: weight stream of W KB (+ publish 16 KB + gather (G-1) x 16 KB)\n");
    printf("# %-34s %10s %12s %12s\n", "variant", "kernel us", "us / step", "wall_clock p50 us/step");
    for (int variant = 0; variant < 3; ++variant) {
        const int G = variant == 0 ? 1 : (variant == 1 ? 2 : 4);
        const long long bytes = full / G;
        float best = 1e9f;
        std::vector<unsigned long long> ht(blocks);
        for (int rep = 0; rep < 30; ++rep) {
            ++tag;
            CHECK(hipEventRecord(e0));
            if (G == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(THREADS), lds, 0, w, bytes, slices, flags, tag, t, failed, sink);
            else if (G == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(THREADS), lds, 0, w, bytes, slices, flags, tag, t, failed, sink);
            else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(THREADS), lds, 0, w, bytes, slices, flags, tag, t, failed, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 5) best = std::min(best, ms);
        }
        CHECK(hipMemcpy(ht.data(), t, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(ht.begin(), ht.end());
        unsigned f = 0;
        CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
        char name[96];
        snprintf(name, sizeof name, "G = %d: stream %lld KB / step%s", G, bytes / 1024, G > 1 ? " + exchange" : "");
        printf("  %-34s %10.2f %12.2f %12.2f%s\n", name, best * 1e3, best * 1e3 / STEPS, ht[blocks / 2] / 100.0 / STEPS,
               f ? "   (a poll gave up: partner not resident?)" : "");
    }
    return 0;
}
