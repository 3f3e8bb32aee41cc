// Micro-benchmark: VALU issue cost of ONE wave per CU on gfx950 (the regime vine_step_kernel runs in).
// Prints cycles per instruction for dependent / independent fma chains, SGPR operands, rsq, and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float s0, float s1) {
    float a = threadIdx.x * 1e-3f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f, e = a + 4.0f, f = a + 5.0f, g = a + 6.0f, h = a + 7.0f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {  // dependent chain, VGPR operands
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
        } else if (MODE == 1) {  // 4 independent chains interleaved
            REP8(REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                              : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));))
        } else if (MODE == 2) {  // dependent chain, one SGPR operand
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "s"(s0), "v"(c));)
        } else if (MODE == 3) {  // 2 independent chains
            REP8(REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c));))
        } else if (MODE == 4) {  // dependent v_mul (VOP2)
            REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if (MODE == 5) {  // rsq chain
            REP64(asm volatile("v_rsq_f32 %0, %0" : "+v"(a));)
        } else if (MODE == 6) {  // rsq independent from fma stream: 1 rsq + 7 fma
            REP8(asm volatile("v_rsq_f32 %1, %1\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3"
                              : "+v"(a), "+v"(d) : "v"(b), "v"(c));)
        } else if (MODE == 7) {  // 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b), "+v"(c) : "v"(s0), "v"(s1));)
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int threads, int instr_per_iter) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-44s blocks %4d x %4d thr: %.2f cycles/instr (per wave)\n", name, blocks, threads, s / blocks / (64.0 * instr_per_iter));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("dependent v_fma (VGPR)", 256, 64, 64);
    run<3>("2 independent v_fma chains", 256, 64, 128);
    run<1>("4 independent v_fma chains", 256, 64, 256);
    run<7>("8 independent v_fma chains", 256, 64, 64);
    run<2>("dependent v_fma, SGPR operand", 256, 64, 64);
    run<4>("dependent v_mul (VOP2)", 256, 64, 64);
    run<5>("dependent v_rsq", 256, 64, 64);
    run<6>("1 rsq + 7 dependent fma", 256, 64, 64);
    run<0>("dependent v_fma, 2 waves/SIMD (512 thr)", 256, 512, 64);
    run<1>("4 indep chains, 2 waves/SIMD (512 thr)", 256, 512, 256);
    run<0>("dependent v_fma, 1 wave/SIMD (256 thr)", 256, 256, 64);
    run<1>("4 indep chains, 1 wave/SIMD (256 thr)", 256, 256, 256);
    return 0;
}
