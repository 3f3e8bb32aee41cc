// ppo_kernels.hip — fused pointwise kernels of the PPO update for gfx950 (C ABI: include/vine_ppo.h).
// HBM-bound elementwise work: one thread per float4 of hidden units, 16-B loads/stores, grid capped and strided.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../include/vine.h"
#include "../../include/vine_ppo.h"

namespace {

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU instructions): the LSTM step kernel spent 800 of
// its 1800 VALU instructions per wave on the 80 divisions of its gate non-linearities
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh via exp of -2|x|: accurate to ~2e-7 relative, no overflow
    const float ax = fabsf(x);
    const float e = __expf(-2.0f * ax);
    const float t = (1.0f - e) * rcpf_(1.0f + e);
    return copysignf(t, x);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// 16-bit storage of the mixed-precision update (GEMM operands and backward-only saved activations; arithmetic and
// accumulation stay fp32).  The format is a build-time choice of this translation unit:
//   default        IEEE half (fp16: 11-bit significand) -- what the reference's `mixed_precision: True` autocasts to
//                  (Vine5LinkMovingBasePPO.yaml:53), with its GradScaler restated on the device (loss scale applied by the
//                  loss kernel, overflow flagged where a gradient is rounded to 16 bits, skip / back-off / growth in the
//                  Adam kernel): v_mfma_f32_16x16x32_f16, same rate as the bf16 form;
//   -DVINE_LP_BF16 bfloat16 (8-bit significand, fp32 range, no loss scaling needed): the round-2 format, kept as an A/B build.
// vine_lp16_format() reports which one the library was built with; the host allocates torch.float16 / torch.bfloat16 to match.
typedef unsigned short lp16_t;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2_hw;
#ifdef VINE_LP_BF16
typedef __bf16 lp16_hw;
#define VINE_LP16_NAME "bf16"
#define MFMA_LP16_BUILTIN __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4bf16
typedef __bf16 lp16_tr_hw;
#define LP16_MAX 3.3895314e38f
#else
typedef _Float16 lp16_hw;
#define VINE_LP16_NAME "fp16"
#define MFMA_LP16_BUILTIN __builtin_amdgcn_mfma_f32_16x16x32_f16
#define DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4f16
typedef __fp16 lp16_tr_hw;
#define LP16_MAX 65504.0f
#endif
typedef __attribute__((__vector_size__(2 * sizeof(lp16_hw)))) lp16_hw lp16x2_hw;
// round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_f16_f32)
__device__ __forceinline__ lp16_t f2lp(float f) {
    const lp16_hw h = (lp16_hw)f;
    return *reinterpret_cast<const lp16_t*>(&h);
}
__device__ __forceinline__ unsigned f2lp2(float lo, float hi) {      // two values -> one packed dword
    const f32x2_hw v = {lo, hi};
    const lp16x2_hw h = __builtin_convertvector(v, lp16x2_hw);
    return *reinterpret_cast<const unsigned*>(&h);
}
#ifdef VINE_LP_BF16
__device__ __forceinline__ float lp2f(lp16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void unpack2(unsigned r, float& lo, float& hi) {
    lo = __uint_as_float(r << 16);
    hi = __uint_as_float(r & 0xFFFF0000u);
}
#else
__device__ __forceinline__ float lp2f(lp16_t h) { return (float)*reinterpret_cast<const lp16_hw*>(&h); }
__device__ __forceinline__ void unpack2(unsigned r, float& lo, float& hi) {
    const lp16x2_hw h = *reinterpret_cast<const lp16x2_hw*>(&r);
    lo = (float)h[0];
    hi = (float)h[1];
}
#endif
// 4 consecutive elements, fp32 or 16-bit storage
__device__ __forceinline__ float4 ld4(const lp16_t* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    float4 v;
    unpack2(r.x, v.x, v.y);
    unpack2(r.y, v.z, v.w);
    return v;
}
__device__ __forceinline__ void st4(lp16_t* p, float4 v) {
    uint2 r;
    r.x = f2lp2(v.x, v.y);
    r.y = f2lp2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
}

template <typename HP>
__global__ void lstm_fwd_kernel(long long B, int H, const float* __restrict__ igates, long long ig_stride,
                                const float* __restrict__ hgates, const float* __restrict__ bias,
                                const float* __restrict__ c_prev, const unsigned char* __restrict__ done,
                                long long done_stride, float* __restrict__ h_out, long long h_stride,
                                float* __restrict__ c_out, HP* __restrict__ gates_act,
                                HP* __restrict__ hp_next, const unsigned char* __restrict__ done_next,
                                long long done_next_stride, long long hp_stride) {
    const int H4 = H >> 2;
    const long long total = B * H4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / H4;
        const int j = (int)(idx - b * H4) << 2;
        const float keep = done ? 1.0f - (float)done[b * done_stride] : 1.0f;
        const float* ig = igates + b * ig_stride;
        float4 g[4];
        if (hgates) {
            const float* hg = hgates + b * 4LL * H;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 a = ld4(ig + k * H + j), h = ld4(hg + k * H + j), bb = ld4(bias + k * H + j);
                g[k] = make_float4(a.x + keep * h.x + bb.x, a.y + keep * h.y + bb.y, a.z + keep * h.z + bb.z,
                                   a.w + keep * h.w + bb.w);
            }
        } else {      // igates already holds x W_ih^T + h W_hh^T (one GEMM over the concatenated [x | h] operand)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 a = ld4(ig + k * H + j), bb = ld4(bias + k * H + j);
                g[k] = make_float4(a.x + bb.x, a.y + bb.y, a.z + bb.z, a.w + bb.w);
            }
        }
        const float4 cp = ld4(c_prev + b * H + j);
        float gi[4] = {g[0].x, g[0].y, g[0].z, g[0].w}, gf[4] = {g[1].x, g[1].y, g[1].z, g[1].w};
        float gg[4] = {g[2].x, g[2].y, g[2].z, g[2].w}, go[4] = {g[3].x, g[3].y, g[3].z, g[3].w};
        const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
        float cn[4], hn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gi[u] = sigmoidf_(gi[u]);
            gf[u] = sigmoidf_(gf[u]);
            gg[u] = tanhf_(gg[u]);
            go[u] = sigmoidf_(go[u]);
            cn[u] = gf[u] * (keep * cpv[u]) + gi[u] * gg[u];
            hn[u] = go[u] * tanhf_(cn[u]);
        }
        st4(c_out + b * H + j, make_float4(cn[0], cn[1], cn[2], cn[3]));
        st4(h_out + b * h_stride + j, make_float4(hn[0], hn[1], hn[2], hn[3]));
        if (hp_next) {   // the masked hidden state step t+1 consumes (operand of the recurrent weight gradient)
            const float kn = done_next ? 1.0f - (float)done_next[b * done_next_stride] : 1.0f;
            st4(hp_next + b * hp_stride + j, make_float4(kn * hn[0], kn * hn[1], kn * hn[2], kn * hn[3]));
        }
        if (gates_act) {
            HP* ga = gates_act + b * 4LL * H;
            st4(ga + 0 * H + j, make_float4(gi[0], gi[1], gi[2], gi[3]));
            st4(ga + 1 * H + j, make_float4(gf[0], gf[1], gf[2], gf[3]));
            st4(ga + 2 * H + j, make_float4(gg[0], gg[1], gg[2], gg[3]));
            st4(ga + 3 * H + j, make_float4(go[0], go[1], go[2], go[3]));
        }
    }
}

// ---- LSTM step with the recurrent GEMM fused in: gates = A W^T (+ igates) + bias on the matrix cores, pointwise
// epilogue on the accumulators; the [B, 4H] pre-activations never touch HBM.
//   A [B, K] bf16 (masked h, or the rollout's [x | h] operand), W [4H, K] bf16 (w_hh, or [w_ih | w_hh]).
// Tiling: workgroup = 4 waves = 64 batch rows x 16 hidden units (x 4 gates); each wave 16 rows.  Computed transposed
// (D^T = W A^T) with v_mfma_f32_16x16x32_bf16 so that a lane ends up with 4 CONSECUTIVE hidden units of ONE batch
// row for each gate (C/D map: col = lane & 15 -> batch row, row = 4 (lane >> 4) + reg -> hidden unit): float4 I/O.
// W's 64 x K slab is staged once in LDS (row pitch K*2 + 16 B: the 16 lanes of a fragment read hit distinct banks;
// 33 KB at K = 256: 4 workgroups per CU); the A fragments (16 B per lane and k-step) come straight from global memory.
// Measured at B = 8192, K = 256: 33 us against 16 us (hipBLASLt GEMM) + 24 us (pointwise kernel) unfused.  Two
// re-tilings were tried and rejected: 32 units per workgroup with the whole slab in LDS (68 KB, 2 workgroups per CU:
// 46 us), 32 units with W streamed through LDS in double-buffered 64-wide k chunks (36 KB: 52 us), and keeping the
// slab while one workgroup walks over several 64-row blocks (fewer, longer workgroups: 37 us at K = 352).
// What did pay for the K = 352 [x | h] shapes: lstm_step_mfma64_kernel below (64 units per workgroup, the A fragments
// are re-read 4 instead of 16 times, W streamed one 32-wide k-step at a time): 28.9 us against 32.3 us.
typedef __attribute__((__vector_size__(8 * sizeof(lp16_hw)))) lp16_hw lp16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
__device__ __forceinline__ f32x4_t MFMA_LP16(lp16x8_t a, lp16x8_t b, f32x4_t c) { return MFMA_LP16_BUILTIN(a, b, c, 0, 0, 0); }
#define LSTM_MFMA_MAX_K 512

template <int KSTEPS, int KS1>      // K = 32 * KSTEPS; the first KS1 k-steps of A come from A, the rest from A2
__global__ __launch_bounds__(256) void lstm_step_mfma_kernel(
    long long B, int H, const lp16_t* __restrict__ A, long long lda, const lp16_t* __restrict__ A2, long long lda2,
    const lp16_t* __restrict__ W, long long ldw,
    const float* __restrict__ igates, long long ig_stride, const float* __restrict__ bias,
    const float* __restrict__ c_prev, const unsigned char* __restrict__ done, long long done_stride,
    float* __restrict__ h_out, long long h_stride, float* __restrict__ c_out, lp16_t* __restrict__ gates_act,
    lp16_t* __restrict__ hp_next, const unsigned char* __restrict__ done_next, long long done_next_stride,
    long long hp_stride) {
    constexpr int K = 32 * KSTEPS;
    constexpr int PITCH = K + 8;                               // bf16 elements: K*2 + 16 bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    lp16_t* wl = reinterpret_cast<lp16_t*>(lds_raw);           // [4 gates x 16 units][PITCH]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.y * 16;                            // first hidden unit of this workgroup
    const long long b0 = (long long)blockIdx.x * 64 + wave * 16;
    // Everything this wave will need from HBM is requested up front, so that its latency runs under the weight staging
    // and the MFMA loop: A fragments (lane holds A[b0 + (lane & 15)][32 kk + 8 (lane >> 4) + 0..7]) and the epilogue's
    // operands (this lane owns batch row b, hidden units j .. j+3 of every gate).
    const long long b = b0 + (lane & 15);
    const int j = u0 + 4 * (lane >> 4);
    lp16x8_t af[KSTEPS];
    const lp16_t* arow = A + b * lda + 8 * (lane >> 4);
    const lp16_t* arow2 = KS1 < KSTEPS && KS1 > 0 ? A2 + b * lda2 + 8 * (lane >> 4) : arow;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk)
        af[kk] = (KS1 == 0 || kk < KS1) ? *reinterpret_cast<const lp16x8_t*>(arow + 32 * kk)
                                        : *reinterpret_cast<const lp16x8_t*>(arow2 + 32 * (kk - KS1));
    float4 igv[4], bbv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bbv[g] = ld4(bias + g * H + j);
        igv[g] = igates ? ld4(igates + b * ig_stride + g * H + j) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const float4 cp = ld4(c_prev + b * H + j);
    const float keep = done ? 1.0f - (float)done[b * done_stride] : 1.0f;
    const float kn = (hp_next && done_next) ? 1.0f - (float)done_next[b * done_next_stride] : 1.0f;
    // stage W: LDS row g*16 + i  <-  W[g*H + u0 + i][0:K]   (16-B chunks, coalesced along k)
    constexpr int CHUNKS = K / 8;
    // (64 * CHUNKS / 256 = KSTEPS pieces per thread; unrolled with all loads ahead of the LDS stores: left rolled, each
    //  iteration waited out a full L2 round trip before the next load was issued -- 12 us of a 37 us launch at K = 352)
    {
        uint4 stage[KSTEPS];
#pragma unroll
        for (int it = 0; it < KSTEPS; ++it) {
            const int c = threadIdx.x + 256 * it;
            const int row = c / CHUNKS, ck = c - row * CHUNKS;
            const int g = row >> 4, i = row & 15;
            stage[it] = *reinterpret_cast<const uint4*>(W + (long long)(g * H + u0 + i) * ldw + ck * 8);
        }
#pragma unroll
        for (int it = 0; it < KSTEPS; ++it) {
            const int c = threadIdx.x + 256 * it;
            const int row = c / CHUNKS, ck = c - row * CHUNKS;
            *reinterpret_cast<uint4*>(&wl[row * PITCH + ck * 8]) = stage[it];
        }
    }
    __syncthreads();
    f32x4_t acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const lp16x8_t wf =
                *reinterpret_cast<const lp16x8_t*>(&wl[(g * 16 + (lane & 15)) * PITCH + 32 * kk + 8 * (lane >> 4)]);
            acc[g] = MFMA_LP16(wf, af[kk], acc[g]);
        }
    }
    float pre[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        pre[g][0] = acc[g][0] + bbv[g].x + igv[g].x; pre[g][1] = acc[g][1] + bbv[g].y + igv[g].y;
        pre[g][2] = acc[g][2] + bbv[g].z + igv[g].z; pre[g][3] = acc[g][3] + bbv[g].w + igv[g].w;
    }
    const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
    float gi[4], gf[4], gg[4], go[4], cn[4], hn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        gi[u] = sigmoidf_(pre[0][u]);
        gf[u] = sigmoidf_(pre[1][u]);
        gg[u] = tanhf_(pre[2][u]);
        go[u] = sigmoidf_(pre[3][u]);
        cn[u] = gf[u] * (keep * cpv[u]) + gi[u] * gg[u];
        hn[u] = go[u] * tanhf_(cn[u]);
    }
    st4(c_out + b * H + j, make_float4(cn[0], cn[1], cn[2], cn[3]));
    st4(h_out + b * h_stride + j, make_float4(hn[0], hn[1], hn[2], hn[3]));
    if (hp_next) st4(hp_next + b * hp_stride + j, make_float4(kn * hn[0], kn * hn[1], kn * hn[2], kn * hn[3]));
    if (gates_act) {
        lp16_t* ga = gates_act + b * 4LL * H;
        st4(ga + 0 * H + j, make_float4(gi[0], gi[1], gi[2], gi[3]));
        st4(ga + 1 * H + j, make_float4(gf[0], gf[1], gf[2], gf[3]));
        st4(ga + 2 * H + j, make_float4(gg[0], gg[1], gg[2], gg[3]));
        st4(ga + 3 * H + j, make_float4(go[0], go[1], go[2], go[3]));
    }
}

// ---- the same LSTM step with 64 hidden units per workgroup and the weights STREAMED through LDS: chunk = one MFMA
// k-step (32 k) of the 4 x 64 weight rows (20 KB), double-buffered, one barrier per chunk, everything unrolled.
// Against the resident-slab kernel above this re-reads every A row 4 times instead of 16, and its epilogue goes
// through LDS so that 16 lanes cover one row's 64 units: 256 contiguous bytes per row and array instead of 64.
template <int KSTEPS, int KS1>
__global__ __launch_bounds__(256) void lstm_step_mfma64_kernel(
    long long B, int H, const lp16_t* __restrict__ A, long long lda, const lp16_t* __restrict__ A2, long long lda2,
    const lp16_t* __restrict__ W, long long ldw, const float* __restrict__ igates, long long ig_stride,
    const float* __restrict__ bias, const float* __restrict__ c_prev, const unsigned char* __restrict__ done,
    long long done_stride, float* __restrict__ h_out, long long h_stride, float* __restrict__ c_out,
    lp16_t* __restrict__ gates_act, lp16_t* __restrict__ hp_next, const unsigned char* __restrict__ done_next,
    long long done_next_stride, long long hp_stride) {
    constexpr int PITCH = 32 + 8;                              // 80 B rows: 16 fragment rows x 16 B tile the banks
    constexpr int GP = 4 * 64 + 4;                             // floats per row of the pre-activation hand-over tile
    // one LDS block: the two weight buffers (2 x 20 KB) during the product, the [64][GP] fp32 tile (65 KB) after it
    __shared__ __attribute__((aligned(16))) float smem[64 * GP];
    lp16_t (*wl)[256 * PITCH] = reinterpret_cast<lp16_t (*)[256 * PITCH]>(smem);
    float* gt = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.y * 64;
    const long long b = (long long)blockIdx.x * 64 + wave * 16 + (lane & 15);
    // A fragments, all k-steps up front
    lp16x8_t af[KSTEPS];
    const lp16_t* arow = A + b * lda + 8 * (lane >> 4);
    const lp16_t* arow2 = KS1 < KSTEPS && KS1 > 0 ? A2 + b * lda2 + 8 * (lane >> 4) : arow;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk)
        af[kk] = (KS1 == 0 || kk < KS1) ? *reinterpret_cast<const lp16x8_t*>(arow + 32 * kk)
                                        : *reinterpret_cast<const lp16x8_t*>(arow2 + 32 * (kk - KS1));
    // weight chunk kk: LDS row g*64 + i <- W[g*H + u0 + i][32 kk : 32 kk + 32]; 1024 16-B pieces, 4 per thread
    // (piece p = tid + 256 q: row p >> 2, 16-B column p & 3)
    const int pcol = threadIdx.x & 3;
    int prow[4];
    const lp16_t* wsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        prow[q] = ((int)threadIdx.x + 256 * q) >> 2;
        const int g = prow[q] >> 6, i = prow[q] & 63;
        wsrc[q] = W + (long long)(g * H + u0 + i) * ldw + pcol * 8;
    }
    // three register sets for the weight stream (named scalars: an indexed array lands in scratch): chunk kk + 2 is
    // requested while chunk kk is multiplied, so a request has two whole k-steps of matrix work to hide its L2 round
    // trip behind instead of one
    uint4 wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3, wc0, wc1, wc2, wc3;
#define LOAD_W(S, kk)                                                       \
    S##0 = *reinterpret_cast<const uint4*>(wsrc[0] + 32 * (kk));            \
    S##1 = *reinterpret_cast<const uint4*>(wsrc[1] + 32 * (kk));            \
    S##2 = *reinterpret_cast<const uint4*>(wsrc[2] + 32 * (kk));            \
    S##3 = *reinterpret_cast<const uint4*>(wsrc[3] + 32 * (kk))
#define STORE_W(S, kk)                                                                        \
    *reinterpret_cast<uint4*>(&wl[(kk) & 1][prow[0] * PITCH + pcol * 8]) = S##0;              \
    *reinterpret_cast<uint4*>(&wl[(kk) & 1][prow[1] * PITCH + pcol * 8]) = S##1;              \
    *reinterpret_cast<uint4*>(&wl[(kk) & 1][prow[2] * PITCH + pcol * 8]) = S##2;              \
    *reinterpret_cast<uint4*>(&wl[(kk) & 1][prow[3] * PITCH + pcol * 8]) = S##3
    f32x4_t acc[4][4];                                         // [gate][unit tile]
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[g][t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    // step kk: request chunk kk + 2 into set SL (stored to LDS one step ago), multiply chunk kk, move chunk kk + 1 from
    // set SS into the other LDS buffer
#define LSTM_STEP(kk, SL, SS)                                                                                  \
    if (KSTEPS > (kk)) {                                                                                       \
        if ((kk) > 0 && (kk) + 2 < KSTEPS) { LOAD_W(SL, (kk) + 2); }                                           \
        const lp16_t* wb_ = wl[(kk) & 1];                                                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
                const lp16x8_t wf = *reinterpret_cast<const lp16x8_t*>(                                        \
                    wb_ + (g * 64 + t * 16 + (lane & 15)) * PITCH + 8 * (lane >> 4));                          \
                acc[g][t] = MFMA_LP16(wf, af[(kk) < KSTEPS ? (kk) : 0], acc[g][t]); \
            }                                                                                                  \
        if ((kk) + 1 < KSTEPS) { STORE_W(SS, (kk) + 1); }                                                      \
        __syncthreads();                                                                                       \
    }
    LOAD_W(wa, 0);
    if (KSTEPS > 1) { LOAD_W(wb, 1); }
    if (KSTEPS > 2) { LOAD_W(wc, 2); }
    // operands of the pointwise part (thread -> unit quad pq of rows prg + 16 p), requested now so that their HBM
    // round trip runs under the matrix work instead of after it
    const int pq = threadIdx.x & 15, prg = threadIdx.x >> 4;
    float4 cpre[4];
    float keep_pre[4], kn_pre[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long long bp = (long long)blockIdx.x * 64 + prg + 16 * p;
        cpre[p] = ld4(c_prev + bp * H + u0 + 4 * pq);
        keep_pre[p] = done ? 1.0f - (float)done[bp * done_stride] : 1.0f;
        kn_pre[p] = (hp_next && done_next) ? 1.0f - (float)done_next[bp * done_next_stride] : 1.0f;
    }
    STORE_W(wa, 0);
    __syncthreads();
    LSTM_STEP(0, wc, wb) LSTM_STEP(1, wa, wc) LSTM_STEP(2, wb, wa) LSTM_STEP(3, wc, wb)
    LSTM_STEP(4, wa, wc) LSTM_STEP(5, wb, wa) LSTM_STEP(6, wc, wb) LSTM_STEP(7, wa, wc)
    LSTM_STEP(8, wb, wa) LSTM_STEP(9, wc, wb) LSTM_STEP(10, wa, wc) LSTM_STEP(11, wb, wa)
    LSTM_STEP(12, wc, wb) LSTM_STEP(13, wa, wc) LSTM_STEP(14, wb, wa) LSTM_STEP(15, wc, wb)
    static_assert(KSTEPS <= 16, "lstm_step_mfma64_kernel: K <= 512");
#undef LSTM_STEP
#undef LOAD_W
#undef STORE_W
    // hand-over through LDS: accumulators -> [row][gate][unit] tile, then the pointwise part runs with 16 lanes per
    // row (whole 256-B fp32 / 128-B bf16 row segments per array) instead of 64-B pieces of 16 different rows
    {
        const int r = wave * 16 + (lane & 15);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<float4*>(&gt[r * GP + g * 64 + 16 * t + 4 * (lane >> 4)]) =
                    make_float4(acc[g][t][0], acc[g][t][1], acc[g][t][2], acc[g][t][3]);
    }
    __syncthreads();
    const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j = u0 + 4 * q;
    float bv[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 bb = ld4(bias + g * H + j);
        bv[g][0] = bb.x; bv[g][1] = bb.y; bv[g][2] = bb.z; bv[g][3] = bb.w;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = rg + 16 * p;
        const long long b = (long long)blockIdx.x * 64 + r;
        const float keep = keep_pre[p], kn = kn_pre[p];
        float pre[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(&gt[r * GP + g * 64 + 4 * q]);
            pre[g][0] = a.x + bv[g][0]; pre[g][1] = a.y + bv[g][1]; pre[g][2] = a.z + bv[g][2]; pre[g][3] = a.w + bv[g][3];
            if (igates) {
                const float4 ig = ld4(igates + b * ig_stride + g * H + j);
                pre[g][0] += ig.x; pre[g][1] += ig.y; pre[g][2] += ig.z; pre[g][3] += ig.w;
            }
        }
        const float4 cp = cpre[p];
        const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
        float gi[4], gf[4], gg[4], go[4], cn[4], hn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gi[u] = sigmoidf_(pre[0][u]);
            gf[u] = sigmoidf_(pre[1][u]);
            gg[u] = tanhf_(pre[2][u]);
            go[u] = sigmoidf_(pre[3][u]);
            cn[u] = gf[u] * (keep * cpv[u]) + gi[u] * gg[u];
            hn[u] = go[u] * tanhf_(cn[u]);
        }
        st4(c_out + b * H + j, make_float4(cn[0], cn[1], cn[2], cn[3]));
        st4(h_out + b * h_stride + j, make_float4(hn[0], hn[1], hn[2], hn[3]));
        if (hp_next) st4(hp_next + b * hp_stride + j, make_float4(kn * hn[0], kn * hn[1], kn * hn[2], kn * hn[3]));
        if (gates_act) {
            lp16_t* ga = gates_act + b * 4LL * H;
            st4(ga + 0 * H + j, make_float4(gi[0], gi[1], gi[2], gi[3]));
            st4(ga + 1 * H + j, make_float4(gf[0], gf[1], gf[2], gf[3]));
            st4(ga + 2 * H + j, make_float4(gg[0], gg[1], gg[2], gg[3]));
            st4(ga + 3 * H + j, make_float4(go[0], go[1], go[2], go[3]));
        }
    }
}

// ================================================================================================================
// Persistent LSTM sequence kernels (mixed precision, H = 256): ONE launch runs all T time steps of a block of 32
// sequences.  Against one launch per step (lstm_step_mfma64_kernel / lstm_bwd_mfma_kernel, where all 512 workgroups
// march load -> matrix work -> hand-over -> pointwise -> store in lock-step and the phases add up):
//   * h_t (forward) / dG_t (backward) stay in LDS as the next step's matrix operand, c_t / dc_t stay in registers:
//     their HBM round trips and T - 1 kernel boundaries disappear;
//   * the weights are streamed from L2 straight into MFMA operand registers, from a copy that is PRE-TILED in fragment
//     order (every load instruction of a wave is one contiguous 1 KiB), through a register ring that keeps ~2.75
//     k-steps in flight and runs ahead ACROSS the step boundary (the weights do not depend on h): no LDS staging of
//     weights, no barrier inside the k-loop -- the only workgroup barrier is the one per time step that publishes h_t;
//   * 8 waves per workgroup (2 per SIMD): while one wave of a SIMD is in its pointwise epilogue / issuing stores the
//     other one can be in its matrix loop.
// Workgroup = 512 threads = 8 waves; wave w owns hidden units [32 w, 32 w + 32) for all 32 rows: 2 unit tiles x
// 4 gates x 2 row tiles of v_mfma_f32_16x16x32_bf16, computed transposed (D^T = W X^T) as in the step kernels so that
// a lane ends up with consecutive hidden units of ONE batch row.  The weight rows of a tile are permuted -- tile
// `ut`, row 4 q + j  <->  unit 32 w + 8 q + 4 ut + j -- so that the lane group q holds the 8 CONSECUTIVE units
// 32 w + 8 q .. + 7 over its two tiles: 32-B fp32 / 16-B bf16 pieces per lane, 128 / 64 contiguous bytes per row.
// Tiled weight layout (built by lstm_tile_weights_kernel, one 16-B chunk per lane):
//   chunk(w, kk, j = 2 g + ut, lane)  =  Wsrc[row(g, unit)][32 kk + 8 (lane >> 4) .. + 8],
//   unit = 32 w + 8 ((lane & 15) >> 2) + 4 ut + (lane & 3),  at chunk index ((w * KSTEPS + kk) * NJ + j) * 64 + lane.
__device__ __forceinline__ float dpp_i2f(int x);
__device__ __forceinline__ int dpp_f2i(float x);
constexpr int SEQ_ROWS = 32;           // sequences per workgroup
// Row skew of the 16-bit LDS operand images that are read as MFMA fragments (lane = row + 16 x chunk, one ds_read_b128): on
// gfx950 such a read is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- over 64 banks,
// and the 16 pieces of a group tile the banks only when the row pitch is an ODD multiple of 32 bytes (round 5: the pitches
// were K + 8 elements = an odd multiple of 16 bytes, right for groups of 8 consecutive lanes: SQ_LDS_BANK_CONFLICT was 36 % of
// SQ_LDS_IDX_ACTIVE in the four-phase launch, 45 % in the one-launch MLP)
constexpr int LDS_SKEW = 16;           // elements (32 bytes); K is a multiple of 32 elements everywhere it is used
constexpr int SEQ_FWD_NC = 11;         // weight fragments per wave kept in LDS by the forward kernel (88 KB)
constexpr int SEQ_H = 256;             // hidden units (8 waves x 32)
#ifndef SEQ_BWD_RING
#define SEQ_BWD_RING 8      // weight ring of the backward kernel: 16 does not fit the register file without spills (76.7 us against 65.3 us per 4-step sequence)
#endif

__device__ __forceinline__ void seq_barrier() {
    // LDS hand-over between the waves of the workgroup: only the LDS counter has to drain -- the weight ring and
    // the epilogue's stores stay in flight across the barrier (a __syncthreads() would wait for vmcnt(0) too)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// dst chunk -> source elements.  transposed = 0: Wsrc [rows, ld] row-major, row(g, unit) = g * H + unit, k along the
// row (forward: [w_ih | 0 | w_hh], K = 32 KSTEPS).  transposed = 1: element (unit, k) = Wsrc[k * ld + unit] (backward:
// Wsrc = w_hh [4H, H], the product dG W_hh sums over k = gate-major 4H index; NJ = 2, row(g, unit) = unit).
__global__ __launch_bounds__(256) void lstm_tile_weights_kernel(const lp16_t* __restrict__ src, long long ld, int H, int ksteps,
                                                                int nj, int transposed, lp16_t* __restrict__ dst) {
    const long long chunks = (long long)(H / 32) * ksteps * nj * 64;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63);
        long long r = c >> 6;
        const int j = (int)(r % nj); r /= nj;
        const int kk = (int)(r % ksteps);
        const int w = (int)(r / ksteps);
        const int ut = nj == 8 ? (j & 1) : j, g = nj == 8 ? (j >> 1) : 0;
        const int unit = 32 * w + 8 * ((lane & 15) >> 2) + 4 * ut + (lane & 3);
        const int k0 = 32 * kk + 8 * (lane >> 4);
        uint4 v;
        if (!transposed) {
            v = *reinterpret_cast<const uint4*>(src + (long long)(g * H + unit) * ld + k0);
        } else {
            lp16_t e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = src[(long long)(k0 + i) * ld + unit];
            v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
            v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
        }
        *reinterpret_cast<uint4*>(dst + c * 8) = v;
    }
}

typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4_t;
__device__ __forceinline__ uint4 seq_load_frag(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 pack_lp16x8(const float (&v)[8]) {
    uint4 r;
    r.x = f2lp2(v[0], v[1]); r.y = f2lp2(v[2], v[3]); r.z = f2lp2(v[4], v[5]); r.w = f2lp2(v[6], v[7]);
    return r;
}

// CT: storage type of the saved cell states c_1 .. c_{T-1} (read again by the backward kernel only): float, or lp16_t
// -- the recurrence itself always runs on the fp32 registers, and c_T then goes to `c_last` in fp32.
// HT: type of the hidden states written out.  float: h_out [B*T, H] fp32 for the LayerNorm, and the MASKED 16-bit h_{t-1}
// a second time into `hp` (slot t) for the recurrent weight gradient.  lp16_t ("h once"): h_out [B, T + 1, H] in the
// 16-bit format, unmasked, is the only copy -- slot 0 = h0, slot t + 1 = h_t; the LayerNorm reads slots 1 .. T as an
// autocast LSTM's output, the weight-gradient kernel reads slots 0 .. T - 1 and applies done[k] itself (`hp` unused):
// 30 MB fewer stores per launch at the update's shape.
#ifdef SPLIT_TIMING
__device__ unsigned long long mlp_split_t[4096 * 16];
#endif
#if defined(SPLIT_TIMING) && defined(BWD_STAMPS)
// (debug build -DSPLIT_TIMING -DBWD_STAMPS: stamps 5, 6, 7 of scripts/ubench/trunk_phases_clock.py move from the loss phase
// into the LSTM backward phase: first step done / second step's matrix loop done / second step done)
#define BSTAMP(i)                                                                                                       \
    if ((threadIdx.x & 63) == 0) {                                                                                      \
        mlp_split_t[((blockIdx.x * 8 + (threadIdx.x >> 6)) & 4095) * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime();      \
        mlp_split_t[((blockIdx.x * 8 + (threadIdx.x >> 6)) & 4095) * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define BSTAMP(i)
#endif
#ifdef SEQ_TIMING
__device__ unsigned long long seq_t[256 * 8 * 16];      // (debug build: 4 stamps per step and wave, scripts/ubench/lstm_seq_phases.py)
#endif
// NC (round 4): the first NC fragments of every wave's per-step weight stream are kept in LDS after step 0 (a wave-private
// 16 NC KB region behind the operand rows: no barrier) and re-read from there in steps 1 .. T - 1 -- the kernel is bound
// by the L2 -> CU weight stream (720 KB per step at 51 of the CU's 64 B/clk), and with T <= 4 the operand rows leave 96 KB
// of the CU's 160 KB of LDS unused: 12 of a wave's 88 fragments, 13.6 % of the stream of three of the four steps.
template <int KS1, int RING, typename CT, typename HT, int NC>      // K = 32 (KS1 + 8): the x block (KS1 k-steps, zero-padded) then the 256 hidden units
__device__ __forceinline__ void lstm_seq_fwd_body(
    int T, long long B, const lp16_t* __restrict__ x, long long ldx, lp16_t* hp, long long hp_stride,
    const uint4* __restrict__ Wt, const float* __restrict__ bias, const float* __restrict__ c0,
    const unsigned char* __restrict__ done, HT* __restrict__ h_out, CT* __restrict__ c_all,
    lp16_t* __restrict__ gates, int ablate, float* __restrict__ c_last, const float* __restrict__ h0) {
    constexpr int H = SEQ_H, KSTEPS = KS1 + 8, KX = 32 * KS1, NF = KSTEPS * 8;
    // LDS operand rows (16-B row skew; both pitches put the 16 rows of a fragment read on distinct bank groups): the x
    // blocks of ALL T steps, staged once in the prologue, and the masked h_{t-1} in two buffers.  The x k-steps of a step
    // need nothing from the step before, so the one workgroup barrier per step sits BEHIND them: a wave that finishes its
    // pointwise epilogue early runs the next step's x part while the slowest wave is still storing (the waves reach the
    // barrier up to 2 us apart: per-step stamps, profiles/r03/update_kernel_experiments.txt).
    constexpr int XP = KX + LDS_SKEW, HP = H + LDS_SKEW;
    static_assert(NF % RING == 0, "the ring must close on a step boundary");
    // (dynamic: T x 6.5 KB + 33 KB + 4 KB goes past the 64 KB a kernel gets without asking)
    extern __shared__ __attribute__((aligned(16))) unsigned char seqf_lds[];
    float* bias_l = reinterpret_cast<float*>(seqf_lds);                                             // [4H]
    lp16_t (*hl)[SEQ_ROWS * HP] = reinterpret_cast<lp16_t (*)[SEQ_ROWS * HP]>(seqf_lds + 4 * H * sizeof(float));   // [2]: masked h_{t-1}, ping-pong
    lp16_t (*xs)[SEQ_ROWS * XP] = reinterpret_cast<lp16_t (*)[SEQ_ROWS * XP]>(seqf_lds + 4 * H * sizeof(float) +
                                                                            2 * SEQ_ROWS * HP * sizeof(lp16_t));   // [T]: x_t of every step
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the stream base below lives in SGPRs
    const int col = lane & 15, lq = lane >> 4;
    const long long b0 = (long long)blockIdx.x * SEQ_ROWS;
    const int U0 = 32 * w + 8 * lq;                              // this lane's 8 consecutive hidden units
    // ---- weight ring: the wave's own contiguous stream of NF fragments per step, RING of them in flight
    // (scalar base + one 32-bit lane offset + immediates: 88 per-load 64-bit addresses would not fit the register file)
    // addressed through a buffer descriptor: scalar base (this wave's stream) + ONE 32-bit lane offset + a scalar
    // fragment offset per load -- plain pointers made the compiler keep 88 loop-invariant 64-bit addresses in VGPRs
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(Wt) + (long long)w * NF * 1024), 0, NF * 1024, 0x00020000);
    const int wlane = lane * 16;
#define SEQ_WFRAG(f) seq_load_frag(wrsrc, wlane, (f) * 1024)
    uint4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = SEQ_WFRAG(i);
    static_assert(NC <= NF - RING && NC <= RING, "cached fragments are re-requested from the tail of the previous step only");
    uint4* wcache = reinterpret_cast<uint4*>(seqf_lds + 4 * H * sizeof(float) + 2 * SEQ_ROWS * HP * sizeof(lp16_t) +
                                             (size_t)T * SEQ_ROWS * XP * sizeof(lp16_t)) + (w * NC) * 64 + lane;
    // ---- x rows of all steps, masked h of step 0, bias, cell state, done flags
    {
        for (int p = tid; p < T * SEQ_ROWS * (KX / 8); p += 512) {
            const int tt = p / (SEQ_ROWS * (KX / 8)), q = p - tt * (SEQ_ROWS * (KX / 8));
            const int r = q / (KX / 8), c8 = q - r * (KX / 8);
            *reinterpret_cast<uint4*>(&xs[tt][r * XP + 8 * c8]) =
                *reinterpret_cast<const uint4*>(x + ((b0 + r) * T + tt) * ldx + 8 * c8);
        }
        for (int p = tid; p < SEQ_ROWS * (H / 8); p += 512) {
            const int r = p >> 5, c8 = p & 31;
            uint4 hv;
            if (h0) {
                // the masked initial state is formed here from the fp32 state and the first done flag (formerly a job
                // of the batched operand-copy launch: three quarters of its threads), and written to slot 0 of hp for
                // the weight-gradient kernel
                const float keep0 = (done && done[(b0 + r) * T]) ? 0.0f : 1.0f;
                const float4 a = ld4(h0 + (b0 + r) * H + 8 * c8), b = ld4(h0 + (b0 + r) * H + 8 * c8 + 4);
                const float v[8] = {keep0 * a.x, keep0 * a.y, keep0 * a.z, keep0 * a.w, keep0 * b.x, keep0 * b.y, keep0 * b.z, keep0 * b.w};
                hv = pack_lp16x8(v);
                if (sizeof(HT) == 2) {      // "h once": slot 0 of h_out receives the UNMASKED 16-bit h0
                    const float u8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    *reinterpret_cast<uint4*>(reinterpret_cast<lp16_t*>(h_out) + (b0 + r) * (T + 1) * H + 8 * c8) = pack_lp16x8(u8);
                } else {
                    *reinterpret_cast<uint4*>(hp + (b0 + r) * hp_stride + 8 * c8) = hv;
                }
            } else {
                hv = *reinterpret_cast<const uint4*>(hp + (b0 + r) * hp_stride + 8 * c8);      // slot 0: masked h_{-1}
            }
            *reinterpret_cast<uint4*>(&hl[0][r * HP + 8 * c8]) = hv;
        }
        for (int p = tid; p < H; p += 512) st4(&bias_l[4 * p], ld4(bias + 4 * p));
    }
    float c[2][8];
    unsigned dmask[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const long long b = b0 + 16 * rt + col;
        const float4 a = ld4(c0 + b * H + U0), bb = ld4(c0 + b * H + U0 + 4);
        c[rt][0] = a.x; c[rt][1] = a.y; c[rt][2] = a.z; c[rt][3] = a.w;
        c[rt][4] = bb.x; c[rt][5] = bb.y; c[rt][6] = bb.z; c[rt][7] = bb.w;
        unsigned m = 0;
        if (done)
            for (int t = 0; t < T; ++t) m |= (done[b * T + t] ? 1u : 0u) << t;
        dmask[rt] = m;
    }
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
#ifdef SEQ_TIMING
        if (lane == 0 && t < 4) seq_t[(blockIdx.x * 8 + w) * 16 + 4 * t + 0] = wall_clock64();
#endif
        if (t == 0) seq_barrier();                               // xs (all steps), hl[0] and bias_l complete
        const lp16_t* xb = xs[t];
        const lp16_t* hb = hl[t & 1];
        lp16_t* hnext = hl[(t + 1) & 1];
        const bool last = t == T - 1;
        f32x4_t acc[4][2][2];                                    // [gate][unit tile][row tile]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[g][ut][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            // (the barrier of steps 1 .. T-1: every wave's masked h_{t-1} is in hl[t & 1]; placed after the x k-steps)
#ifdef SEQ_TIMING
            if (kk == KS1 && lane == 0 && t < 4) seq_t[(blockIdx.x * 8 + w) * 16 + 4 * t + 1] = wall_clock64();
#endif
            if (kk == KS1 && t > 0) seq_barrier();
            const lp16_t* ob = kk < KS1 ? xb + 32 * kk : hb + 32 * (kk - KS1);
            const int op = kk < KS1 ? XP : HP;
            const lp16x8_t a0 = *reinterpret_cast<const lp16x8_t*>(ob + col * op + 8 * lq);
            const lp16x8_t a1 = *reinterpret_cast<const lp16x8_t*>(ob + (16 + col) * op + 8 * lq);
#ifdef SEQ_FWD_HALF_STREAM
            uint4 wr_prev = make_uint4(0, 0, 0, 0);
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = kk * 8 + j, slot = f % RING;
#ifdef SEQ_FWD_HALF_STREAM
                // (timing experiment, debug build only: the odd fragments are never loaded -- each even one is used twice --
                // = the weight stream of a 2-CU column split, with none of its other costs; results are garbage)
                const uint4 wr = (f & 1) ? wr_prev : ring[slot];      // (the even fragment's registers: its slot is reloading)
                wr_prev = wr;
#else
                const uint4 wr = ring[slot];
#endif
                const lp16x8_t wf = __builtin_bit_cast(lp16x8_t, wr);
                acc[j >> 1][j & 1][0] = MFMA_LP16(wf, a0, acc[j >> 1][j & 1][0]);
                acc[j >> 1][j & 1][1] = MFMA_LP16(wf, a1, acc[j >> 1][j & 1][1]);
                if (f < NC && t == 0) wcache[f * 64] = wr;       // (wave-private LDS copy for the later steps)
#ifdef SEQ_FWD_HALF_STREAM
                if (!(ablate & 2) && !(f & 1)) {
#else
                if (!(ablate & 2)) {
#endif
                    const int fn = (f + RING) % NF;              // the fragment RING places further down the stream (folded: unrolled)
                    if (fn < NC && f + RING >= NF) ring[slot] = wcache[fn * 64];      // next step's: cached in step 0
                    else ring[slot] = SEQ_WFRAG(fn);
                }
                // pin the order {2 MFMAs, reload}: left alone, the scheduler sinks every reload down to its use one
                // step later (to save registers) and the ring degenerates into load -> wait -> use
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef SEQ_TIMING
        if (lane == 0 && t < 4) seq_t[(blockIdx.x * 8 + w) * 16 + 4 * t + 2] = wall_clock64();
#endif
        // ---- pointwise epilogue on the accumulators: lane = batch row 16 rt + col, units U0 .. U0 + 7 (4 per unit tile;
        // the bf16 outputs of tile 0 wait, packed, for tile 1 so that every bf16 store is one 16-B piece)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long long b = b0 + 16 * rt + col;
            const float keep = ((dmask[rt] >> t) & 1u) ? 0.0f : 1.0f;
            const float kn = ((dmask[rt] >> (t + 1)) & 1u) ? 0.0f : 1.0f;
            CT* cp = c_all + ((long long)(t + 1) * B + b) * H + U0;
            float* cl = c_last + b * H + U0;                       // (only dereferenced in the low-precision mode)
            HT* hpo = h_out + (sizeof(HT) == 2 ? b * (T + 1) + t + 1 : b * T + t) * H + U0;
            uint2 lo[6];                                         // i, f, g, o, masked h, h of tile 0
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) {
                const float4 bi = *reinterpret_cast<const float4*>(&bias_l[0 * H + U0 + 4 * ut]);
                const float4 bf_ = *reinterpret_cast<const float4*>(&bias_l[1 * H + U0 + 4 * ut]);
                const float4 bg = *reinterpret_cast<const float4*>(&bias_l[2 * H + U0 + 4 * ut]);
                const float4 bo = *reinterpret_cast<const float4*>(&bias_l[3 * H + U0 + 4 * ut]);
                const float bv[4][4] = {{bi.x, bi.y, bi.z, bi.w}, {bf_.x, bf_.y, bf_.z, bf_.w}, {bg.x, bg.y, bg.z, bg.w},
                                        {bo.x, bo.y, bo.z, bo.w}};
                float gi[4], gf[4], gg[4], go[4], hn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    gi[u] = sigmoidf_(acc[0][ut][rt][u] + bv[0][u]);
                    gf[u] = sigmoidf_(acc[1][ut][rt][u] + bv[1][u]);
                    gg[u] = tanhf_(acc[2][ut][rt][u] + bv[2][u]);
                    go[u] = sigmoidf_(acc[3][ut][rt][u] + bv[3][u]);
                    const float cn = gf[u] * (keep * c[rt][4 * ut + u]) + gi[u] * gg[u];
                    c[rt][4 * ut + u] = cn;
                    hn[u] = go[u] * tanhf_(cn);
                }
                if (!(ablate & 1)) {
                    const float4 cv = make_float4(c[rt][4 * ut], c[rt][4 * ut + 1], c[rt][4 * ut + 2], c[rt][4 * ut + 3]);
                    if (sizeof(CT) == 2 && last) st4(cl + 4 * ut, cv);
                    else st4(cp + 4 * ut, cv);
                    if (sizeof(HT) == 4) st4(hpo + 4 * ut, make_float4(hn[0], hn[1], hn[2], hn[3]));
                }
                uint2 pk[6];
                pk[5] = make_uint2(f2lp2(hn[0], hn[1]), f2lp2(hn[2], hn[3]));
                pk[0] = make_uint2(f2lp2(gi[0], gi[1]), f2lp2(gi[2], gi[3]));
                pk[1] = make_uint2(f2lp2(gf[0], gf[1]), f2lp2(gf[2], gf[3]));
                pk[2] = make_uint2(f2lp2(gg[0], gg[1]), f2lp2(gg[2], gg[3]));
                pk[3] = make_uint2(f2lp2(go[0], go[1]), f2lp2(go[2], go[3]));
                pk[4] = make_uint2(f2lp2(kn * hn[0], kn * hn[1]), f2lp2(kn * hn[2], kn * hn[3]));
                if (ut == 0) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) lo[a] = pk[a];
                } else {
                    if (sizeof(HT) == 2 && !(ablate & 1))
                        *reinterpret_cast<uint4*>(hpo) = make_uint4(lo[5].x, lo[5].y, pk[5].x, pk[5].y);
                    if (gates && !(ablate & 1)) {
                        lp16_t* ga = gates + ((long long)t * B + b) * 4 * H + U0;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<uint4*>(ga + g * H) = make_uint4(lo[g].x, lo[g].y, pk[g].x, pk[g].y);
                    }
                    if (!last) {   // the masked state step t + 1 consumes: next operand (LDS) + weight-gradient operand (HBM)
                        const uint4 hm = make_uint4(lo[4].x, lo[4].y, pk[4].x, pk[4].y);
                        *reinterpret_cast<uint4*>(&hnext[(16 * rt + col) * HP + U0]) = hm;
                        if (sizeof(HT) == 4 && !(ablate & 1))
                            *reinterpret_cast<uint4*>(hp + b * hp_stride + (long long)(t + 1) * H + U0) = hm;
                    }
                }
            }
        }
#ifdef SEQ_TIMING
        if (lane == 0 && t < 4) seq_t[(blockIdx.x * 8 + w) * 16 + 4 * t + 3] = wall_clock64();
#endif
    }
}
template <int KS1, int RING, typename CT, typename HT, int NC>
__global__ __launch_bounds__(512) void lstm_seq_fwd_kernel(
    int T, long long B, const lp16_t* __restrict__ x, long long ldx, lp16_t* hp, long long hp_stride,
    const uint4* __restrict__ Wt, const float* __restrict__ bias, const float* __restrict__ c0,
    const unsigned char* __restrict__ done, HT* __restrict__ h_out, CT* __restrict__ c_all,
    lp16_t* __restrict__ gates, int ablate, float* __restrict__ c_last, const float* __restrict__ h0) {
    lstm_seq_fwd_body<KS1, RING, CT, HT, NC>(T, B, x, ldx, hp, hp_stride, Wt, bias, c0, done, h_out, c_all, gates, ablate, c_last,
                                             h0);
}

// ---- backward twin: all T steps of lstm_bwd_mfma_kernel for 32 sequences in one launch.  dG_{t+1} [32, 4H] stays in
// LDS (bf16, the operand of the recurrent product dG_{t+1} W_hh), dc_t and c_t stay in registers, the bias-gradient
// partial sums are carried in registers over the T steps and folded over the 32 rows once at the end (DPP row sums:
// the 16 lanes of a DPP row are exactly the 16 batch rows of a unit group).  Wave w owns hidden units [32 w, 32 w + 32):
// 2 unit tiles x 2 row tiles, K = 4H = 1024 streamed from the fragment-ordered copy of w_hh^T (64 fragments per step).
__device__ __forceinline__ float dpp_row_sum16(float v) {
    // sum over the 16 lanes of a DPP row, valid in lane 15 of the row
    float s = v + dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(v), 0x111, 0xf, 0xf, true));      // row_shr:1
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(v), 0x112, 0xf, 0xf, true));                // row_shr:2
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(v), 0x113, 0xf, 0xf, true));                // row_shr:3
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(s), 0x114, 0xf, 0xe, true));                // row_shr:4
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(s), 0x118, 0xf, 0xc, true));                // row_shr:8
    return s;
}
__device__ __forceinline__ void unpack_lp16x8(uint4 r, float (&v)[8]) {
    unpack2(r.x, v[0], v[1]); unpack2(r.y, v[2], v[3]); unpack2(r.z, v[4], v[5]); unpack2(r.w, v[6], v[7]);
}

// CT: storage type of the saved cell states (see the forward kernel; with lp16_t, c_T comes from `c_last` in fp32);
// GT: type of the incoming gradient w.r.t. the hidden states (float, or lp16_t as written by ln_heads_loss_kernel).
template <int RING, typename CT, typename GT>
__device__ __forceinline__ void lstm_seq_bwd_body(
    int T, long long B, const GT* __restrict__ g_out, const uint4* __restrict__ Wt, const lp16_t* __restrict__ gates,
    const CT* __restrict__ c_all, const float* __restrict__ c0, const unsigned char* __restrict__ done,
    lp16_t* __restrict__ dG, float* __restrict__ bias_partial, int ablate, const float* __restrict__ c_last) {
    constexpr int H = SEQ_H, K = 4 * H, KSTEPS = K / 32, NF = KSTEPS * 2;
    constexpr int PITCH = K + LDS_SKEW;
    static_assert(NF % RING == 0, "the ring must close on a step boundary");
    extern __shared__ __attribute__((aligned(16))) unsigned char seq_lds[];
    lp16_t (*dgl)[SEQ_ROWS * PITCH] = reinterpret_cast<lp16_t (*)[SEQ_ROWS * PITCH]>(seq_lds);     // [2][32 rows][4H + 8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, lq = lane >> 4;
    const long long b0 = (long long)blockIdx.x * SEQ_ROWS;
    const int U0 = 32 * w + 8 * lq;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(Wt) + (long long)w * NF * 1024), 0, NF * 1024, 0x00020000);
    const int wlane = lane * 16;
    uint4 ring[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = SEQ_WFRAG(i);
    unsigned dmask[2];
    float dcarry[2][8], cnew[2][8], bsum[4][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const long long b = b0 + 16 * rt + col;
        unsigned m = 0;
        if (done)
            for (int t = 0; t < T; ++t) m |= (done[b * T + t] ? 1u : 0u) << t;
        dmask[rt] = m;
        float4 a, bb;
        if (sizeof(CT) == 2) { a = ld4(c_last + b * H + U0); bb = ld4(c_last + b * H + U0 + 4); }
        else { a = ld4(c_all + ((long long)T * B + b) * H + U0); bb = ld4(c_all + ((long long)T * B + b) * H + U0 + 4); }
        cnew[rt][0] = a.x; cnew[rt][1] = a.y; cnew[rt][2] = a.z; cnew[rt][3] = a.w;
        cnew[rt][4] = bb.x; cnew[rt][5] = bb.y; cnew[rt][6] = bb.z; cnew[rt][7] = bb.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) dcarry[rt][e] = 0.0f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[g][e] = 0.0f;
    // (round 5: the first step -- no matrix loop -- is its own instance of the step: `first` is a compile-time constant in
    // both instances, 2-8 VGPRs fewer in every backward kernel and one spill fewer in the four-phase launch)
    auto step = [&](const int t, auto first_c) __attribute__((always_inline)) {
        constexpr bool first = decltype(first_c)::value;
        // operands of the pointwise part: those of row tile 0 are requested ahead of the matrix loop (their HBM round
        // trip hides under it), those of row tile 1 when tile 0's arithmetic starts (registers: 32 fewer live in the loop)
        float4 go4[2][2], cp4[2][2];
        uint4 gpk[2][4];
#define SEQ_BWD_LOAD(rt)                                                                                   \
        {                                                                                                  \
            const long long b_ = b0 + 16 * (rt) + col;                                                     \
            const GT* gp_ = g_out + (b_ * T + t) * H + U0;                                                 \
            go4[rt][0] = ld4(gp_); go4[rt][1] = ld4(gp_ + 4);                                              \
            if (t == 0) {                                                                                  \
                cp4[rt][0] = ld4(c0 + b_ * H + U0); cp4[rt][1] = ld4(c0 + b_ * H + U0 + 4);                \
            } else {                                                                                       \
                const CT* cpp_ = c_all + ((long long)t * B + b_) * H + U0;                                 \
                cp4[rt][0] = ld4(cpp_); cp4[rt][1] = ld4(cpp_ + 4);                                        \
            }                                                                                              \
            const lp16_t* ga_ = gates + ((long long)t * B + b_) * 4 * H + U0;                              \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) gpk[rt][g] = *reinterpret_cast<const uint4*>(ga_ + g * H); \
        }
        SEQ_BWD_LOAD(0)
        f32x4_t acc[2][2];                                       // [unit tile][row tile]
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[ut][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        if (!first) {
            seq_barrier();                                       // dG_{t+1} complete in dgl[(t + 1) & 1]
            const lp16_t* xb = dgl[(t + 1) & 1];
            if (t == T - 2) { BSTAMP(5) }
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const lp16x8_t a0 = *reinterpret_cast<const lp16x8_t*>(xb + col * PITCH + 32 * kk + 8 * lq);
                const lp16x8_t a1 = *reinterpret_cast<const lp16x8_t*>(xb + (16 + col) * PITCH + 32 * kk + 8 * lq);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int f = kk * 2 + j, slot = f % RING;
                    const lp16x8_t wf = __builtin_bit_cast(lp16x8_t, ring[slot]);
                    acc[j][0] = MFMA_LP16(wf, a0, acc[j][0]);
                    acc[j][1] = MFMA_LP16(wf, a1, acc[j][1]);
                    if (!(ablate & 2)) ring[slot] = SEQ_WFRAG((f + RING) % NF);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (t == T - 2) { BSTAMP(6) }
        }
        lp16_t* dgn = dgl[t & 1];
        SEQ_BWD_LOAD(1)
#undef SEQ_BWD_LOAD
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long long b = b0 + 16 * rt + col;
            const float keep = ((dmask[rt] >> t) & 1u) ? 0.0f : 1.0f;
            const float keep_n = ((dmask[rt] >> (t + 1)) & 1u) ? 0.0f : 1.0f;
            lp16_t* dgp = dG + (b * T + t) * 4 * H + U0;
            uint2 lo[4];                                         // packed dG of unit tile 0 (waits for tile 1: 16-B pieces)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) {
                // the 4 units of this tile: gate activations from the packed loads, the rest from the fp32 quads
                float gi[4], gf[4], gg[4], go[4];
                const unsigned ia = ut ? gpk[rt][0].z : gpk[rt][0].x, ib = ut ? gpk[rt][0].w : gpk[rt][0].y;
                const unsigned fa = ut ? gpk[rt][1].z : gpk[rt][1].x, fb = ut ? gpk[rt][1].w : gpk[rt][1].y;
                const unsigned ga_ = ut ? gpk[rt][2].z : gpk[rt][2].x, gb_ = ut ? gpk[rt][2].w : gpk[rt][2].y;
                const unsigned oa = ut ? gpk[rt][3].z : gpk[rt][3].x, ob = ut ? gpk[rt][3].w : gpk[rt][3].y;
                unpack2(ia, gi[0], gi[1]); unpack2(ib, gi[2], gi[3]);
                unpack2(fa, gf[0], gf[1]); unpack2(fb, gf[2], gf[3]);
                unpack2(ga_, gg[0], gg[1]); unpack2(gb_, gg[2], gg[3]);
                unpack2(oa, go[0], go[1]); unpack2(ob, go[2], go[3]);
                const float gout[4] = {go4[rt][ut].x, go4[rt][ut].y, go4[rt][ut].z, go4[rt][ut].w};
                const float cp[4] = {cp4[rt][ut].x, cp4[rt][ut].y, cp4[rt][ut].z, cp4[rt][ut].w};
                float di[4], df[4], dg[4], dout[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = 4 * ut + u;
                    float dh = gout[u];
                    if (!first) dh += keep_n * acc[ut][rt][u];
                    float dc = first ? 0.0f : keep_n * dcarry[rt][e];
                    // (with `first` a compile-time constant the product above would be contracted into the sum below as one
                    // fma -- the step kernels, where it sits behind a select, round it separately: keep the two bit-identical)
                    if (!first) asm volatile("" : "+v"(dc));
                    const float tc = tanhf_(cnew[rt][e]);
                    const float d_o = dh * tc;
                    const float d_c = dc + dh * go[u] * (1.0f - tc * tc);
                    di[u] = d_c * gg[u] * gi[u] * (1.0f - gi[u]);
                    df[u] = d_c * (keep * cp[u]) * gf[u] * (1.0f - gf[u]);
                    dg[u] = d_c * gi[u] * (1.0f - gg[u] * gg[u]);
                    dout[u] = d_o * go[u] * (1.0f - go[u]);
                    dcarry[rt][e] = d_c * gf[u];
                    cnew[rt][e] = cp[u];                          // c_{t-1} is c_new of the next (earlier) step
                    bsum[0][e] += di[u]; bsum[1][e] += df[u]; bsum[2][e] += dg[u]; bsum[3][e] += dout[u];
                }
                uint2 pk[4];
                pk[0] = make_uint2(f2lp2(di[0], di[1]), f2lp2(di[2], di[3]));
                pk[1] = make_uint2(f2lp2(df[0], df[1]), f2lp2(df[2], df[3]));
                pk[2] = make_uint2(f2lp2(dg[0], dg[1]), f2lp2(dg[2], dg[3]));
                pk[3] = make_uint2(f2lp2(dout[0], dout[1]), f2lp2(dout[2], dout[3]));
                if (ut == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) lo[g] = pk[g];
                } else {
                    lp16_t* row = dgn + (16 * rt + col) * PITCH + U0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const uint4 v = make_uint4(lo[g].x, lo[g].y, pk[g].x, pk[g].y);
                        if (!(ablate & 1)) *reinterpret_cast<uint4*>(dgp + g * H) = v;
                        if (t > 0) *reinterpret_cast<uint4*>(row + g * H) = v;
                    }
                }
            }
        }
        if (t == T - 2) { BSTAMP(7) }
    };
    step(T - 1, std::true_type{});
#pragma unroll 1
    for (int t = T - 2; t >= 0; --t) step(t, std::false_type{});
    if (bias_partial) {
        float* outp = bias_partial + (long long)blockIdx.x * 4 * H + U0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = dpp_row_sum16(bsum[g][e]);
            if (col == 15) {
                st4(outp + g * H, make_float4(r[0], r[1], r[2], r[3]));
                st4(outp + g * H + 4, make_float4(r[4], r[5], r[6], r[7]));
            }
        }
    }
}
template <int RING, typename CT, typename GT>
__global__ __launch_bounds__(512) void lstm_seq_bwd_kernel(
    int T, long long B, const GT* __restrict__ g_out, const uint4* __restrict__ Wt, const lp16_t* __restrict__ gates,
    const CT* __restrict__ c_all, const float* __restrict__ c0, const unsigned char* __restrict__ done,
    lp16_t* __restrict__ dG, float* __restrict__ bias_partial, int ablate, const float* __restrict__ c_last) {
    lstm_seq_bwd_body<RING, CT, GT>(T, B, g_out, Wt, gates, c_all, c0, done, dG, bias_partial, ablate, c_last);
}
#undef SEQ_WFRAG

// ---- Linear + bias + ELU on the matrix cores (the MLP layers of the mixed-precision path): out = elu(A W^T + bias),
// A [n, K] bf16, W [N, K] bf16, out [n, N] bf16.  Same scheme as lstm_step_mfma_kernel: transposed product so that a
// lane holds 4 consecutive output units of one row per 16-unit tile, W's 64 x K slab in LDS, A fragments from global
// memory; workgroup = 64 rows x 64 units, so every output row segment is one full 128-B line of bf16.
template <int KSTEPS>
__global__ __launch_bounds__(256) void linear_elu_mfma_kernel(long long n, int N, const lp16_t* __restrict__ A,
                                                              long long lda, const lp16_t* __restrict__ W, long long ldw,
                                                              const float* __restrict__ bias, float alpha,
                                                              lp16_t* __restrict__ out, long long out_stride) {
    constexpr int K = 32 * KSTEPS;
    constexpr int PITCH = K + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    lp16_t* wl = reinterpret_cast<lp16_t*>(lds_raw);           // [64 units][PITCH]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.y * 64;
    const long long b = (long long)blockIdx.x * 64 + wave * 16 + (lane & 15);
    lp16x8_t af[KSTEPS];
    const lp16_t* arow = A + b * lda + 8 * (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) af[kk] = *reinterpret_cast<const lp16x8_t*>(arow + 32 * kk);
    float4 bbv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bbv[t] = ld4(bias + u0 + 16 * t + 4 * (lane >> 4));
    constexpr int CHUNKS = K / 8;
    {
        uint4 stage[KSTEPS];
#pragma unroll
        for (int it = 0; it < KSTEPS; ++it) {
            const int c = threadIdx.x + 256 * it;
            const int row = c / CHUNKS, ck = c - row * CHUNKS;
            stage[it] = *reinterpret_cast<const uint4*>(W + (long long)(u0 + row) * ldw + ck * 8);
        }
#pragma unroll
        for (int it = 0; it < KSTEPS; ++it) {
            const int c = threadIdx.x + 256 * it;
            const int row = c / CHUNKS, ck = c - row * CHUNKS;
            *reinterpret_cast<uint4*>(&wl[row * PITCH + ck * 8]) = stage[it];
        }
    }
    __syncthreads();
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const lp16x8_t wf =
                *reinterpret_cast<const lp16x8_t*>(&wl[(t * 16 + (lane & 15)) * PITCH + 32 * kk + 8 * (lane >> 4)]);
            acc[t] = MFMA_LP16(wf, af[kk], acc[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float x[4] = {acc[t][0] + bbv[t].x, acc[t][1] + bbv[t].y, acc[t][2] + bbv[t].z, acc[t][3] + bbv[t].w};
        float y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = x[u] > 0.0f ? x[u] : alpha * (__expf(x[u]) - 1.0f);
        st4(out + b * out_stride + u0 + 16 * t + 4 * (lane >> 4), make_float4(y[0], y[1], y[2], y[3]));
    }
}

// ---- Several small 2-D element moves in ONE launch: the operand preparation of an optimiser step (concatenated /
// padded / transposed bf16 weight operands, the merged head weights, fp32 -> bf16 casts of the observations) is a
// dozen strided copies of a few hundred KB each; as separate launches they cost ~5 us apiece.
// dst[r, c] (rows dst_stride apart) = op(src) for r < rows, c < cols:
//   op 0 copy        src[r, c]            (elem = 2 or 4 bytes, same type both sides)
//   op 1 zero
//   op 2 transpose   src[c, r]            (elem = 2 or 4 bytes)
//   op 3 f32 -> bf16 src[r, c]
//   op 4 add (f32)   src[r, c] + src2[r, c]
//   op 5 masked      src[r, c] * (1 - mask[r * aux])   src f32, mask = src2 (uint8), dst f32 or bf16 by elem
//   op 6 / 7         fragment-ordered LSTM weights for the persistent sequence kernels (see the kernel body)
struct CopyJob {
    const void* src;
    const void* src2;
    void* dst;
    long long rows, cols, src_stride, dst_stride, aux;
    int op, elem, first_block, vec;
};
#define VINE_COPY_MAX_JOBS 24
struct CopyBatchArgs {
    CopyJob job[VINE_COPY_MAX_JOBS];
    int njobs;
};
__device__ void copy_scatter_forms(const CopyBatchArgs& batch, int vblock, int vtid);
// (the body of one 256-thread block of the launch: also run as a SIDE JOB by the workgroups of mlp3_elu_mfma_kernel, which
// opens the optimiser step's forward pass -- the operands it builds are derived from the parameters alone and are first
// read by the kernels behind it, so they ride in that launch instead of one of their own: vine_mlp3_elu_mfma_prep)
__device__ __forceinline__ void copy_batched_body(const CopyBatchArgs& batch, int vblock, int vtid) {
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < batch.njobs; ++k)
        if (vblock >= batch.job[k].first_block) j = k;
    const CopyJob& J = batch.job[j];
    const long long total = J.rows * J.cols;
    const long long base = ((long long)(vblock - J.first_block) * 256 + vtid) * 4;
    if (base >= total) return;
    if (J.op >= 8) { copy_scatter_forms(batch, vblock, vtid); return; }      // (coalesced-load forms: copy_item_load)
    if (J.op >= 6) {
        // fragment-ordered LSTM weight for the persistent kernels (layout: lstm_tile_weights_kernel).  dst is flat;
        // element index -> (wave w, k-step kk, fragment j, lane, i) -> (unit, k).
        //   op 6 (forward operand [w_ih | 0 | w_hh]): src = w_ih [4H, src_stride] with aux & 0xffff valid columns padded
        //        with zeros to K1 = aux >> 16, src2 = w_hh [4H, dst_stride]; 8 fragments per k-step (j = 2 gate + tile)
        //   op 7 (backward operand w_hh^T): src = w_hh [4H, src_stride], element (unit, k) = src[k][unit]; 2 fragments
        const int H = SEQ_H;
        const int cols1 = (int)(J.aux & 0xffff), K1 = (int)(J.aux >> 16);
        const int nj = J.op == 6 ? 8 : 2, ksteps = J.op == 6 ? (K1 + H) / 32 : (4 * H) / 32;
        const lp16_t* s1 = reinterpret_cast<const lp16_t*>(J.src);
        const lp16_t* s2 = reinterpret_cast<const lp16_t*>(J.src2);
        lp16_t* d = reinterpret_cast<lp16_t*>(J.dst);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const long long idx = base + q4;
            if (idx >= total) break;
            const int i = (int)(idx & 7);
            long long c = idx >> 3;
            const int lane = (int)(c & 63); c >>= 6;
            const int jj = (int)(c % nj); c /= nj;
            const int kk = (int)(c % ksteps);
            const int wv = (int)(c / ksteps);
            const int ut = J.op == 6 ? (jj & 1) : jj, g = J.op == 6 ? (jj >> 1) : 0;
            const int unit = 32 * wv + 8 * ((lane & 15) >> 2) + 4 * ut + (lane & 3);
            const int k = 32 * kk + 8 * (lane >> 4) + i;
            lp16_t v;
            if (J.op == 6) {
                const long long row = (long long)g * H + unit;
                v = k < K1 ? (k < cols1 ? s1[row * J.src_stride + k] : (lp16_t)0) : s2[row * J.dst_stride + (k - K1)];
            } else {
                v = s1[(long long)k * J.src_stride + unit];
            }
            d[idx] = v;
        }
        return;
    }
    // (row, column) of the first of this thread's 4 consecutive elements, then carried along: one division per thread
    long long r = base / J.cols;
    int c = (int)(base - r * J.cols);
    const int cols = (int)J.cols;
    if (J.vec) {
        // aligned job (cols, strides multiples of 4, pointers aligned): the 4 elements are one 8- or 16-B access
        const long long d = r * J.dst_stride + c, sidx = r * J.src_stride + c;
        switch (J.op) {
            case 0:
                if (J.elem == 2)
                    *reinterpret_cast<uint2*>(reinterpret_cast<lp16_t*>(J.dst) + d) =
                        *reinterpret_cast<const uint2*>(reinterpret_cast<const lp16_t*>(J.src) + sidx);
                else
                    st4(reinterpret_cast<float*>(J.dst) + d, ld4(reinterpret_cast<const float*>(J.src) + sidx));
                break;
            case 1:
                if (J.elem == 2) *reinterpret_cast<uint2*>(reinterpret_cast<lp16_t*>(J.dst) + d) = make_uint2(0u, 0u);
                else st4(reinterpret_cast<float*>(J.dst) + d, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                break;
            case 3:
                st4(reinterpret_cast<lp16_t*>(J.dst) + d, ld4(reinterpret_cast<const float*>(J.src) + sidx));
                break;
            case 4: {
                const float4 a = ld4(reinterpret_cast<const float*>(J.src) + sidx);
                const float4 b = ld4(reinterpret_cast<const float*>(J.src2) + sidx);
                st4(reinterpret_cast<float*>(J.dst) + d, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
                break;
            }
            default: {
                const float keep = J.src2 ? 1.0f - (float)reinterpret_cast<const unsigned char*>(J.src2)[r * J.aux] : 1.0f;
                const float4 a = ld4(reinterpret_cast<const float*>(J.src) + sidx);
                const float4 v = make_float4(a.x * keep, a.y * keep, a.z * keep, a.w * keep);
                if (J.elem == 2) st4(reinterpret_cast<lp16_t*>(J.dst) + d, v);
                else st4(reinterpret_cast<float*>(J.dst) + d, v);
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (base + q >= total) break;
        const long long d = r * J.dst_stride + c;
        const long long sidx = (J.op == 2) ? (long long)c * J.src_stride + r : r * J.src_stride + c;
        switch (J.op) {
            case 0:
            case 2:
                if (J.elem == 2) reinterpret_cast<lp16_t*>(J.dst)[d] = reinterpret_cast<const lp16_t*>(J.src)[sidx];
                else reinterpret_cast<float*>(J.dst)[d] = reinterpret_cast<const float*>(J.src)[sidx];
                break;
            case 1:
                if (J.elem == 2) reinterpret_cast<lp16_t*>(J.dst)[d] = 0;
                else reinterpret_cast<float*>(J.dst)[d] = 0.0f;
                break;
            case 3:
                reinterpret_cast<lp16_t*>(J.dst)[d] = f2lp(reinterpret_cast<const float*>(J.src)[sidx]);
                break;
            case 4:
                reinterpret_cast<float*>(J.dst)[d] =
                    reinterpret_cast<const float*>(J.src)[sidx] + reinterpret_cast<const float*>(J.src2)[sidx];
                break;
            default: {
                const float keep = J.src2 ? 1.0f - (float)reinterpret_cast<const unsigned char*>(J.src2)[r * J.aux] : 1.0f;
                const float v = reinterpret_cast<const float*>(J.src)[sidx] * keep;
                if (J.elem == 2) reinterpret_cast<lp16_t*>(J.dst)[d] = f2lp(v);
                else reinterpret_cast<float*>(J.dst)[d] = v;
            }
        }
        if (++c == cols) { c = 0; ++r; }
    }
}
__global__ __launch_bounds__(256) void copy_batched_kernel(CopyBatchArgs batch) {
    copy_batched_body(batch, (int)blockIdx.x, (int)threadIdx.x);
}

// ---- The same moves in two phases, for a kernel that carries them as a side job (mlp3_elu_mfma_kernel): `copy_item_load`
// issues the loads of one thread's 4 elements of one virtual block and keeps them in registers, `copy_item_store` writes
// them -- the host kernel issues the loads of ALL its virtual blocks and its own first loads before anything is waited for,
// so the side job costs it no memory round trip of its own (run one block after the other inside the launch it cost three
// round trips: measured, no gain).  Covers the moves an optimiser step's parameter-derived operands need: the LSTM weight
// tiles (op 6 / 7), 16-bit transposes (op 2), and the vector forms of copy / zero / add (`side_job_supported`).
struct CopyItem {
    int mode, n;                 // 0 nothing; 1: n 16-bit scalars at d[q]; 2: 8 B at d[0] (16-bit elements); 3: 16 B at d[0] (floats);
                                 // 4: n 32-bit scalars at d[q]
    unsigned int v[4];
    long long d[4];
    void* dst;
};
__host__ __device__ inline bool side_job_supported(const CopyJob& J) {
    if (J.rows * J.cols >= (1ll << 31) || J.cols >= (1ll << 31)) return false;
    // (ops 8-10 exist in the two-phase form only)
    return J.op >= 6 || (J.op == 2 && J.elem == 2) || (J.vec && (J.op == 0 || J.op == 1 || J.op == 4)) ||
           (J.op == 0 && J.elem == 4);
}
__device__ __forceinline__ void copy_item_load(const CopyBatchArgs& batch, int vblock, int vtid, CopyItem& it) {
    it.mode = 0; it.n = 0;
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < batch.njobs; ++k)
        if (vblock >= batch.job[k].first_block) j = k;
    const CopyJob& J = batch.job[j];
    const long long total = J.rows * J.cols;
    const long long base = ((long long)(vblock - J.first_block) * 256 + vtid) * 4;
    if (base >= total) return;
    it.dst = J.dst;
    if (J.op >= 8) {
        // ---- scatter forms (round 4): the thread's 4 elements are consecutive in the SOURCE (one coalesced 8-B load; the
        // gather forms read 2 bytes per 64-B line: 47 MB of L2 -> CU traffic per optimiser step for 1.5 MB of operands)
        // and go to their places in the destination: 8 contiguous bytes (op 8) or four 16-bit stores (ops 9, 10), which
        // nothing waits for.
        const unsigned ub = (unsigned)base;
        const int H = SEQ_H;
        uint2 w = make_uint2(0u, 0u);
        if (J.op == 8) {         // virtual source [w_ih | 0 | w_hh]: rows 4H, K = K1 + H columns -> forward tile order
            const int cols1 = (int)(J.aux & 0xffff), K1 = (int)(J.aux >> 16);
            const unsigned K = (unsigned)(K1 + H), ksteps = K / 32u;
            const unsigned row = ub / K;
            const int k = (int)(ub - row * K);
            if (k < cols1) w = *reinterpret_cast<const uint2*>(reinterpret_cast<const lp16_t*>(J.src) + (long long)row * J.src_stride + k);
            else if (k >= K1) w = *reinterpret_cast<const uint2*>(reinterpret_cast<const lp16_t*>(J.src2) + (long long)row * J.dst_stride + (k - K1));
            const int g = (int)(row >> 8), unit = (int)(row & 255u);
            const int wv = unit >> 5, u5 = unit & 31, lane15 = 4 * (u5 >> 3) + (u5 & 3), ut = (u5 >> 2) & 1;
            const int kk = k >> 5, k5 = k & 31, lane = (k5 >> 3) * 16 + lane15, i = k5 & 7, jj = 2 * g + ut;
            it.d[0] = ((((long long)wv * ksteps + kk) * 8 + jj) * 64 + lane) * 8 + i;
            it.v[0] = w.x; it.v[1] = w.y;
            it.n = 4;
            it.mode = 2;
            return;
        }
        unsigned sr, sc;          // source row / first source column of the 4 elements
        long long dbase, dstep;  // destination index of element 0 and the step to the next one
        if (J.op == 9) {         // source w_hh [4H (k), H (unit)] -> backward tile order (w_hh^T fragments)
            sr = ub >> 8; sc = ub & 255u;
            const int k = (int)sr, unit = (int)sc;
            const int wv = unit >> 5, u5 = unit & 31, lane15 = 4 * (u5 >> 3) + (u5 & 3), ut = (u5 >> 2) & 1;
            const int kk = k >> 5, k5 = k & 31, lane = (k5 >> 3) * 16 + lane15, i = k5 & 7;
            dbase = ((((long long)wv * (4 * H / 32) + kk) * 2 + ut) * 64 + lane) * 8 + i;
            dstep = 8;           // unit + 1 = the next lane of the fragment
        } else {                 // op 10: dst [rows, cols] = src^T, src [cols, rows]: 4 consecutive source columns
            const unsigned R = (unsigned)J.rows;
            sr = ub / R; sc = ub - sr * R;
            dbase = (long long)sc * J.dst_stride + sr;
            dstep = J.dst_stride;
        }
        w = *reinterpret_cast<const uint2*>(reinterpret_cast<const lp16_t*>(J.src) + (long long)sr * J.src_stride + sc);
        it.v[0] = w.x & 0xffffu; it.v[1] = w.x >> 16; it.v[2] = w.y & 0xffffu; it.v[3] = w.y >> 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) it.d[q] = dbase + q * dstep;
        it.n = 4;
        it.mode = 1;
        return;
    }
    if (J.op >= 6) {
        // (index arithmetic of copy_batched_body, decoded ONCE per thread and in 32 bits: the thread's 4 elements are
        // i .. i + 3 of one (wave, k-step, fragment, lane) -- base is a multiple of 4 and a fragment row holds 8 -- and the
        // four 64-bit divisions per ELEMENT of the generic body were most of a side job's time)
        const int H = SEQ_H;
        const int cols1 = (int)(J.aux & 0xffff), K1 = (int)(J.aux >> 16);
        const unsigned ksteps = J.op == 6 ? (unsigned)(K1 + H) / 32u : (unsigned)(4 * H) / 32u;
        const lp16_t* s1 = reinterpret_cast<const lp16_t*>(J.src);
        const lp16_t* s2 = reinterpret_cast<const lp16_t*>(J.src2);
        const unsigned ub = (unsigned)base;
        const int i0 = (int)(ub & 7u);
        unsigned c = ub >> 3;
        const int lane = (int)(c & 63u); c >>= 6;
        const int jj = J.op == 6 ? (int)(c & 7u) : (int)(c & 1u);
        c >>= (J.op == 6 ? 3 : 1);
        const unsigned wv = c / ksteps, kk = c - wv * ksteps;
        const int ut = J.op == 6 ? (jj & 1) : jj, g = J.op == 6 ? (jj >> 1) : 0;
        const int unit = 32 * (int)wv + 8 * ((lane & 15) >> 2) + 4 * ut + (lane & 3);
        const int k0 = 32 * (int)kk + 8 * (lane >> 4) + i0;
        const int n = (int)(total - base < 4 ? total - base : 4);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            it.v[q4] = 0;
            it.d[q4] = base + q4;
            if (q4 < n) {
                const int k = k0 + q4;
                lp16_t v;
                if (J.op == 6) {
                    const long long row = (long long)g * H + unit;
                    v = k < K1 ? (k < cols1 ? s1[row * J.src_stride + k] : (lp16_t)0) : s2[row * J.dst_stride + (k - K1)];
                } else {
                    v = s1[(long long)k * J.src_stride + unit];
                }
                it.v[q4] = (unsigned int)v;
            }
        }
        it.n = n;
        it.mode = n == 4 ? 2 : 1;
        if (n == 4) { it.v[0] |= it.v[1] << 16; it.v[1] = it.v[2] | (it.v[3] << 16); }
        return;
    }
    const int cols = (int)J.cols;
    long long r = (long long)((unsigned)base / (unsigned)cols);      // (32-bit: side jobs are below 2^31 elements)
    int c = (int)((unsigned)base - (unsigned)r * (unsigned)cols);
    if (J.op == 2) {             // 16-bit transpose: 4 scalars
        const lp16_t* sp = reinterpret_cast<const lp16_t*>(J.src);
        int n = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            it.v[q] = 0; it.d[q] = 0;
            if (base + q < total) {
                it.d[q] = r * J.dst_stride + c;
                it.v[q] = (unsigned int)sp[(long long)c * J.src_stride + r];
                ++n;
                if (++c == cols) { c = 0; ++r; }
            }
        }
        it.n = n;
        it.mode = 1;
        return;
    }
    if (!J.vec) {                // op 0, fp32, unaligned or short rows (the head biases): up to 4 scalars
        const float* sp = reinterpret_cast<const float*>(J.src);
        int n = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            it.v[q] = 0; it.d[q] = 0;
            if (base + q < total) {
                it.d[q] = r * J.dst_stride + c;
                it.v[q] = __float_as_uint(sp[r * J.src_stride + c]);
                ++n;
                if (++c == cols) { c = 0; ++r; }
            }
        }
        it.n = n;
        it.mode = 4;
        return;
    }
    // vector forms: the 4 elements are one aligned 8- or 16-B access on both sides
    const long long d = r * J.dst_stride + c, sidx = r * J.src_stride + c;
    it.d[0] = d;
    it.n = 4;
    if (J.op == 1) {
        it.v[0] = it.v[1] = it.v[2] = it.v[3] = 0u;
        it.mode = J.elem == 2 ? 2 : 3;
    } else if (J.op == 0 && J.elem == 2) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const lp16_t*>(J.src) + sidx);
        it.v[0] = w.x; it.v[1] = w.y;
        it.mode = 2;
    } else {
        float4 a = ld4(reinterpret_cast<const float*>(J.src) + sidx);
        if (J.op == 4) {
            const float4 b = ld4(reinterpret_cast<const float*>(J.src2) + sidx);
            a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        it.v[0] = __float_as_uint(a.x); it.v[1] = __float_as_uint(a.y); it.v[2] = __float_as_uint(a.z); it.v[3] = __float_as_uint(a.w);
        it.mode = 3;
    }
}
__device__ __forceinline__ void copy_item_store(const CopyItem& it) {
    if (it.mode == 1) {
        lp16_t* d = reinterpret_cast<lp16_t*>(it.dst);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < it.n) d[it.d[q]] = (lp16_t)it.v[q];
    } else if (it.mode == 2) {
        *reinterpret_cast<uint2*>(reinterpret_cast<lp16_t*>(it.dst) + it.d[0]) = make_uint2(it.v[0], it.v[1]);
    } else if (it.mode == 3) {
        *reinterpret_cast<uint4*>(reinterpret_cast<float*>(it.dst) + it.d[0]) = make_uint4(it.v[0], it.v[1], it.v[2], it.v[3]);
    } else if (it.mode == 4) {
        unsigned int* d = reinterpret_cast<unsigned int*>(it.dst);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < it.n) d[it.d[q]] = it.v[q];
    }
}
__device__ void copy_scatter_forms(const CopyBatchArgs& batch, int vblock, int vtid) {
    CopyItem it;
    copy_item_load(batch, vblock, vtid, it);
    copy_item_store(it);
}
#define MLP3_SIDE_ITEMS 4        // virtual blocks per 256 threads of the host kernel (more: the host falls back to its own launch)

// ---- The whole MLP of the default network in ONE kernel (mixed precision):
//   x [n, 32] bf16 (the normalised observation block of the LSTM operand buffer + its zero pad; or, with `raw`, built
//   here from the fp32 observations and the RunningMeanStd statistics and written there) -> Linear(32, C1) + ELU ->
//   Linear(C1, C2) + ELU -> Linear(C2, C3) + ELU -> out [n, C3] bf16 (the MLP block of the LSTM operand buffer), with
//   the two intermediate activations stored for the backward pass when asked for (act1 / act2; the rollout passes NULL).
// Three launches of linear_elu_mfma_kernel (+ one of normalize_obs_kernel) cost ~4.5 us of dispatch and drain each in the
// replayed graphs; here a wave carries its 16 rows through all three layers IN REGISTERS.  That needs no cross-lane
// exchange because the tile -> unit mapping of a producing layer is chosen to match the operand layout of the consuming
// one (as in the LSTM sequence kernels): output tiles (2 kk, 2 kk + 1) of layer L hold, in lane group q, the units
// 32 kk + 8 q + {0..3} and + {4..7} -- exactly the 8 consecutive k's that lane needs as its B fragment of k-step kk of
// layer L + 1 (v_mfma_f32_16x16x32_bf16, transposed product D^T = W X^T as in linear_elu_mfma_kernel).  The weights
// are staged once per workgroup into LDS with their rows permuted into tile order, so fragment reads stay on
// consecutive LDS rows (16-B skew per row: conflict-free).  Workgroup = 16 rows per wave, 4 or 8 waves; LDS 103 KB.
__device__ __forceinline__ int mlp3_tile_row(int u) {      // unit -> LDS row: 32 kk + 8 q + 4 ut + j  ->  32 kk + 16 ut + 4 q + j
    return (u & ~31) | ((u & 4) << 2) | ((u >> 1) & 12) | (u & 3);
}
__device__ __forceinline__ float elu1(float x, float alpha) { return x > 0.0f ? x : alpha * (__expf(x) - 1.0f); }

#ifdef SPLIT_TIMING
// (debug build, scripts/ubench/mlp_split_clock.py / mlp_mfma_clock.py) per wave: s_memtime / s_memrealtime at up to 8 points
// (mlp_split_t itself is declared in front of the persistent LSTM kernels, which stamp into it with -DBWD_STAMPS)
#define MLPM_STAMP(i)                                                                                                   \
    if (lane == 0) {                                                                                                    \
        mlp_split_t[((blockIdx.x * NW + wave) & 4095) * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime();                    \
        mlp_split_t[((blockIdx.x * NW + wave) & 4095) * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();            \
    }
#else
#define MLPM_STAMP(i)
#endif
template <int C1, int C2, int C3, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 2))) void mlp3_elu_mfma_kernel(long long n, lp16_t* x, long long ldx,
                                                            const float* __restrict__ raw, int F_in,
                                                            const double* __restrict__ mean, const double* __restrict__ var,
                                                            float eps, float clip, const lp16_t* __restrict__ w1,
                                                            const float* __restrict__ b1, const lp16_t* __restrict__ w2,
                                                            long long ldw2, const float* __restrict__ b2,
                                                            const lp16_t* __restrict__ w3, long long ldw3,
                                                            const float* __restrict__ b3, float alpha,
                                                            lp16_t* __restrict__ act1, lp16_t* __restrict__ act2,
                                                            lp16_t* out, long long out_stride, int ldw1,
                                                            const CopyBatchArgs side, int side_blocks) {
    constexpr int P1 = 32 + LDS_SKEW, P2 = C1 + LDS_SKEW, P3 = C2 + LDS_SKEW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    lp16_t* w1l = reinterpret_cast<lp16_t*>(lds_raw);            // [C1][P1], rows in tile order
    lp16_t* w2l = w1l + C1 * P1;                                 // [C2][P2], rows in tile order
    lp16_t* w3l = w2l + C2 * P2;                                 // [C3][P3], natural order (its output goes to memory)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const long long b = ((long long)blockIdx.x * NW + wave) * 16 + i;
    MLPM_STAMP(0)
    // ---- weights -> LDS.  Every request leaves before anything is waited for (compile-time trip counts: the staging is
    // one L2 round trip, not one per chunk); layer 1 starts as soon as ITS weights are in LDS, the two larger weights
    // arrive under its arithmetic.  Order of the prologue (round 4): W1, the observations and their statistics, W2 / W3, the
    // side job's requests (its job table is a chain of dependent loads), THEN the observation arithmetic and the side job's
    // stores -- the weight requests used to stand behind the observation arithmetic and the side job, three round trips in
    // a row.
    constexpr int TH = 64 * NW, N1 = C1 * 4 / TH, N2 = C2 * (C1 / 8) / TH, N3 = C3 * (C2 / 8) / TH;
    static_assert(C1 * 4 % TH == 0 && C2 * (C1 / 8) % TH == 0 && C3 * (C2 / 8) % TH == 0, "whole chunks per thread");
    u32x4_t s1[N1], s2[N2], s3[N3];      // native vectors: arrays of the HIP uint4 struct stay in scratch memory here
    if (ldw1 == 32) {            // w1 zero-padded to 32 columns (w1p)
#pragma unroll
        for (int it = 0; it < N1; ++it) {
            const int c = tid + TH * it;
            s1[it] = *reinterpret_cast<const u32x4_t*>(w1 + (c >> 2) * 32 + 8 * (c & 3));
        }
    } else {                     // w1 as stored, [C1, ldw1] with ldw1 = F_in (even) valid columns: padded here
#pragma unroll
        for (int it = 0; it < N1; ++it) {
            const int c = tid + TH * it, row = c >> 2, c0 = 8 * (c & 3);
            unsigned int d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                d[e] = c0 + 2 * e < ldw1 ? *reinterpret_cast<const unsigned int*>(w1 + row * ldw1 + c0 + 2 * e) : 0u;
            s1[it] = u32x4_t{d[0], d[1], d[2], d[3]};
        }
    }
    // ---- this lane's B fragment of layer 1: columns 8 q .. 8 q + 7 of its row.  The float64 statistics are read and turned
    // into (mean, sqrt(var + eps)) by 32 threads, once per workgroup, and handed round through LDS: 16 float64 loads per
    // thread compile to one memory round trip each (the conversion of one is waited for before the next is requested),
    // and per-column `if`s around the observation loads to one round trip per column -- clamped addresses instead
    __shared__ float stat_m[32], stat_sd[32];
    float rv[8];
    double stat_mean = 0.0, stat_var = 1.0;
    if (raw) {
        if (tid < 32) {
            stat_mean = mean[min(tid, F_in - 1)];
            stat_var = var[min(tid, F_in - 1)];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = raw[b * F_in + min(8 * q + e, F_in - 1)];
    }
    // (the two larger weights behind the observations: their 120 KB per CU would stand in front of them otherwise)
#pragma unroll
    for (int it = 0; it < N2; ++it) {
        const int c = tid + TH * it, row = c / (C1 / 8), ck = c - row * (C1 / 8);
        s2[it] = *reinterpret_cast<const u32x4_t*>(w2 + (long long)row * ldw2 + 8 * ck);
    }
#pragma unroll
    for (int it = 0; it < N3; ++it) {
        const int c = tid + TH * it, row = c / (C2 / 8), ck = c - row * (C2 / 8);
        s3[it] = *reinterpret_cast<const u32x4_t*>(w3 + (long long)row * ldw3 + 8 * ck);
    }
    // ---- side job (vine_mlp3_elu_mfma_prep): this workgroup's share of the optimiser step's operand preparation, the
    // former copy_batched launch in front of this one (nothing this kernel reads)
    CopyItem items[MLP3_SIDE_ITEMS];
#pragma unroll
    for (int k = 0; k < MLP3_SIDE_ITEMS; ++k) {
        const int vb = ((int)blockIdx.x + k * (int)gridDim.x) * (TH / 256) + (tid >> 8);
        items[k].mode = 0;
        if (vb < side_blocks) copy_item_load(side, vb, tid & 255, items[k]);
    }
    lp16x8_t af1;
    if (raw) {
        if (tid < 32) {
            // same arithmetic as normalize_obs_kernel: statistics cast to float first
            stat_m[tid] = (float)stat_mean;
            stat_sd[tid] = sqrtf((float)stat_var + eps);
        }
        __syncthreads();
        float v[8];
        const float4 m0 = ld4(stat_m + 8 * q), m1 = ld4(stat_m + 8 * q + 4), d0 = ld4(stat_sd + 8 * q), d1 = ld4(stat_sd + 8 * q + 4);
        const float ms[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const float sds[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = fminf(fmaxf((rv[e] - ms[e]) / sds[e], -clip), clip);
            v[e] = 8 * q + e < F_in ? y : 0.0f;
        }
        const uint4 pk = pack_lp16x8(v);
        af1 = __builtin_bit_cast(lp16x8_t, pk);
        *reinterpret_cast<uint4*>(x + b * ldx + 8 * q) = pk;     // the LSTM operand's observation block (+ zero pad)
    } else {
        af1 = *reinterpret_cast<const lp16x8_t*>(x + b * ldx + 8 * q);
    }
    if (side_blocks > 0) {
#pragma unroll
        for (int k = 0; k < MLP3_SIDE_ITEMS; ++k) copy_item_store(items[k]);
    }
#pragma unroll
    for (int it = 0; it < N1; ++it) {
        const int c = tid + TH * it;
        *reinterpret_cast<u32x4_t*>(&w1l[mlp3_tile_row(c >> 2) * P1 + 8 * (c & 3)]) = s1[it];
    }
    __syncthreads();
    MLPM_STAMP(1)
    // ---- layer 1: K = 32, C1 / 32 pairs of output tiles; pair kk becomes the B fragment of k-step kk of layer 2
    lp16x8_t af2[C1 / 32];
#pragma unroll
    for (int kk = 0; kk < C1 / 32; ++kk) {
        float y[8];
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const lp16x8_t wf = *reinterpret_cast<const lp16x8_t*>(&w1l[(32 * kk + 16 * ut + i) * P1 + 8 * q]);
            const f32x4_t a = MFMA_LP16(wf, af1, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f});
            const float4 bb = ld4(b1 + 32 * kk + 8 * q + 4 * ut);
            y[4 * ut + 0] = elu1(a[0] + bb.x, alpha); y[4 * ut + 1] = elu1(a[1] + bb.y, alpha);
            y[4 * ut + 2] = elu1(a[2] + bb.z, alpha); y[4 * ut + 3] = elu1(a[3] + bb.w, alpha);
        }
        const uint4 pk = pack_lp16x8(y);
        af2[kk] = __builtin_bit_cast(lp16x8_t, pk);
        if (act1) *reinterpret_cast<uint4*>(act1 + b * C1 + 32 * kk + 8 * q) = pk;
    }
    MLPM_STAMP(2)
#pragma unroll
    for (int it = 0; it < N2; ++it) {
        const int c = tid + TH * it, row = c / (C1 / 8), ck = c - row * (C1 / 8);
        *reinterpret_cast<u32x4_t*>(&w2l[mlp3_tile_row(row) * P2 + 8 * ck]) = s2[it];
    }
#pragma unroll
    for (int it = 0; it < N3; ++it) {
        const int c = tid + TH * it, row = c / (C2 / 8), ck = c - row * (C2 / 8);
        *reinterpret_cast<u32x4_t*>(&w3l[row * P3 + 8 * ck]) = s3[it];
    }
    __syncthreads();
    MLPM_STAMP(3)
    // ---- layer 2: K = C1
    lp16x8_t af3[C2 / 32];
#pragma unroll
    for (int kp = 0; kp < C2 / 32; ++kp) {
        f32x4_t a[2] = {f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int kk = 0; kk < C1 / 32; ++kk)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) {
                const lp16x8_t wf = *reinterpret_cast<const lp16x8_t*>(&w2l[(32 * kp + 16 * ut + i) * P2 + 32 * kk + 8 * q]);
                a[ut] = MFMA_LP16(wf, af2[kk], a[ut]);
            }
        float y[8];
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const float4 bb = ld4(b2 + 32 * kp + 8 * q + 4 * ut);
            y[4 * ut + 0] = elu1(a[ut][0] + bb.x, alpha); y[4 * ut + 1] = elu1(a[ut][1] + bb.y, alpha);
            y[4 * ut + 2] = elu1(a[ut][2] + bb.z, alpha); y[4 * ut + 3] = elu1(a[ut][3] + bb.w, alpha);
        }
        const uint4 pk = pack_lp16x8(y);
        af3[kp] = __builtin_bit_cast(lp16x8_t, pk);
        if (act2) *reinterpret_cast<uint4*>(act2 + b * C2 + 32 * kp + 8 * q) = pk;
    }
    MLPM_STAMP(4)
    // ---- layer 3: K = C2, natural tile order: lane holds units 16 t + 4 q + {0..3} of its row
#pragma unroll
    for (int t = 0; t < C3 / 16; ++t) {
        f32x4_t a = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = 0; kk < C2 / 32; ++kk) {
            const lp16x8_t wf = *reinterpret_cast<const lp16x8_t*>(&w3l[(16 * t + i) * P3 + 32 * kk + 8 * q]);
            a = MFMA_LP16(wf, af3[kk], a);
        }
        const float4 bb = ld4(b3 + 16 * t + 4 * q);
        st4(out + b * out_stride + 16 * t + 4 * q,
            make_float4(elu1(a[0] + bb.x, alpha), elu1(a[1] + bb.y, alpha), elu1(a[2] + bb.z, alpha), elu1(a[3] + bb.w, alpha)));
    }
    MLPM_STAMP(5)
}

// ---- backward of a Linear through the previous layer's ELU on the matrix cores:
//   gz[b, u] = (G W)[b, u] * elu'(a[b, u])      G [n, K] bf16 (gradient w.r.t. this layer's pre-activation),
//                                               Wt [N, K] bf16 = W^T (N = input width of the layer),
//                                               a [n, N] bf16 = the previous layer's ELU output
// plus one row of column sums of gz per workgroup (= partial bias gradient of the previous layer).  Same transposed
// product scheme as linear_elu_mfma_kernel; the fp32 input gradient is never stored.
template <int KSTEPS>
__global__ __launch_bounds__(256) void linear_bwd_elu_mfma_kernel(long long n, int N, const lp16_t* __restrict__ G,
                                                                  long long ldg, const lp16_t* __restrict__ Wt,
                                                                  long long ldw, const lp16_t* __restrict__ a,
                                                                  long long a_stride, float alpha,
                                                                  lp16_t* __restrict__ gz, long long gz_stride,
                                                                  float* __restrict__ partial) {
    constexpr int K = 32 * KSTEPS;
    constexpr int PITCH = K + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    lp16_t* wl = reinterpret_cast<lp16_t*>(lds_raw);           // [64 units][PITCH], later reused for the column sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.y * 64;
    const long long b = (long long)blockIdx.x * 64 + wave * 16 + (lane & 15);
    lp16x8_t gf[KSTEPS];
    const lp16_t* grow = G + b * ldg + 8 * (lane >> 4);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) gf[kk] = *reinterpret_cast<const lp16x8_t*>(grow + 32 * kk);
    float4 av[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) av[t] = ld4(a + b * a_stride + u0 + 16 * t + 4 * (lane >> 4));
    constexpr int CHUNKS = K / 8;
    {
        uint4 stage[KSTEPS];
#pragma unroll
        for (int it = 0; it < KSTEPS; ++it) {
            const int c = threadIdx.x + 256 * it;
            const int row = c / CHUNKS, ck = c - row * CHUNKS;
            stage[it] = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + row) * ldw + ck * 8);
        }
#pragma unroll
        for (int it = 0; it < KSTEPS; ++it) {
            const int c = threadIdx.x + 256 * it;
            const int row = c / CHUNKS, ck = c - row * CHUNKS;
            *reinterpret_cast<uint4*>(&wl[row * PITCH + ck * 8]) = stage[it];
        }
    }
    __syncthreads();
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const lp16x8_t wf =
                *reinterpret_cast<const lp16x8_t*>(&wl[(t * 16 + (lane & 15)) * PITCH + 32 * kk + 8 * (lane >> 4)]);
            acc[t] = MFMA_LP16(wf, gf[kk], acc[t]);
        }
    }
    float d[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float aa[4] = {av[t].x, av[t].y, av[t].z, av[t].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) d[t][u] = acc[t][u] * (aa[u] > 0.0f ? 1.0f : aa[u] + alpha);
        st4(gz + b * gz_stride + u0 + 16 * t + 4 * (lane >> 4), make_float4(d[t][0], d[t][1], d[t][2], d[t][3]));
    }
    if (partial) {
        // column sums over the workgroup's 64 rows: [row 0..63][unit 0..63] through LDS, then 64 threads x 64 adds
        __syncthreads();                                       // everyone is done with the weight slab
        float* red = reinterpret_cast<float*>(lds_raw);        // 64 x 65 floats (16.6 KB <= slab size for K >= 128)
        const int r = wave * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) red[r * 65 + 16 * t + 4 * (lane >> 4) + u] = d[t][u];
        __syncthreads();
        if (threadIdx.x < 64) {
            float sum = 0.0f;
            for (int rr = 0; rr < 64; ++rr) sum += red[rr * 65 + threadIdx.x];
            partial[(long long)blockIdx.x * N + u0 + threadIdx.x] = sum;
        }
    }
}

// ---- the same backward product for a LONG reduction (K = 128 * NCH, e.g. the 1024 gate columns of the LSTM): both
// operands stream in 128-wide k chunks, the weight chunk double-buffered through LDS (2 x 17 KB), the G fragments
// double-buffered in registers; one barrier per chunk.  Everything is written out unrolled (no index arrays in
// scratch).  N = 64 output units per workgroup; with N == 64 every G row is read exactly once from HBM.
template <int NCH>
__global__ __launch_bounds__(256) void linear_bwd_elu_mfma_chunked_kernel(
    long long n, int N, const lp16_t* __restrict__ G, long long ldg, const lp16_t* __restrict__ Wt, long long ldw,
    const lp16_t* __restrict__ a, long long a_stride, float alpha, lp16_t* __restrict__ gz, long long gz_stride,
    float* __restrict__ partial) {
    constexpr int CK = 128;                                    // k per chunk = 4 MFMA k-steps
    constexpr int PITCH = CK + 8;
    __shared__ __attribute__((aligned(16))) lp16_t wl[2][64 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.y * 64;
    const long long b = (long long)blockIdx.x * 64 + wave * 16 + (lane & 15);
    const lp16_t* grow = G + b * ldg + 8 * (lane >> 4);
    // this thread's 4 weight pieces of a chunk: piece p = tid + 256 q -> row p >> 4, 16-B column p & 15
    const int prow[4] = {(int)threadIdx.x >> 4, ((int)threadIdx.x + 256) >> 4, ((int)threadIdx.x + 512) >> 4,
                         ((int)threadIdx.x + 768) >> 4};
    const int pcol = threadIdx.x & 15;
    uint4 w0, w1, w2, w3;
    lp16x8_t g0, g1, g2, g3, h0, h1, h2, h3;
#define LOAD_W(c)                                                                                              \
    w0 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[0]) * ldw + (c) * CK + pcol * 8);          \
    w1 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[1]) * ldw + (c) * CK + pcol * 8);          \
    w2 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[2]) * ldw + (c) * CK + pcol * 8);          \
    w3 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[3]) * ldw + (c) * CK + pcol * 8)
#define STORE_W(buf)                                                                                           \
    *reinterpret_cast<uint4*>(&wl[buf][prow[0] * PITCH + pcol * 8]) = w0;                                      \
    *reinterpret_cast<uint4*>(&wl[buf][prow[1] * PITCH + pcol * 8]) = w1;                                      \
    *reinterpret_cast<uint4*>(&wl[buf][prow[2] * PITCH + pcol * 8]) = w2;                                      \
    *reinterpret_cast<uint4*>(&wl[buf][prow[3] * PITCH + pcol * 8]) = w3
#define LOAD_G(c, x0, x1, x2, x3)                                                                              \
    x0 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK);                                                  \
    x1 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK + 32);                                             \
    x2 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK + 64);                                             \
    x3 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK + 96)
    LOAD_W(0);
    LOAD_G(0, g0, g1, g2, g3);
    float4 av[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) av[t] = ld4(a + b * a_stride + u0 + 16 * t + 4 * (lane >> 4));
    STORE_W(0);
    __syncthreads();
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
            LOAD_W(c + 1);
            LOAD_G(c + 1, h0, h1, h2, h3);
        }
        const lp16_t* wb = wl[c & 1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const lp16_t* wr = wb + (t * 16 + (lane & 15)) * PITCH + 8 * (lane >> 4);
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr), g0, acc[t]);
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr + 32), g1, acc[t]);
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr + 64), g2, acc[t]);
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr + 96), g3, acc[t]);
        }
        if (c + 1 < NCH) {
            STORE_W((c + 1) & 1);                              // buffer (c+1)&1 was last read before the previous barrier
            g0 = h0; g1 = h1; g2 = h2; g3 = h3;
        }
        __syncthreads();
    }
#undef LOAD_W
#undef STORE_W
#undef LOAD_G
    float d[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float aa[4] = {av[t].x, av[t].y, av[t].z, av[t].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) d[t][u] = acc[t][u] * (aa[u] > 0.0f ? 1.0f : aa[u] + alpha);
        st4(gz + b * gz_stride + u0 + 16 * t + 4 * (lane >> 4), make_float4(d[t][0], d[t][1], d[t][2], d[t][3]));
    }
    if (partial) {
        float* red = reinterpret_cast<float*>(&wl[0][0]);      // 64 x 65 floats = 16.6 KB <= one weight buffer (17.4 KB)
        const int r = wave * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) red[r * 65 + 16 * t + 4 * (lane >> 4) + u] = d[t][u];
        __syncthreads();
        if (threadIdx.x < 64) {
            float sum = 0.0f;
            for (int rr = 0; rr < 64; ++rr) sum += red[rr * 65 + threadIdx.x];
            partial[(long long)blockIdx.x * N + u0 + threadIdx.x] = sum;
        }
    }
}

// ---- The backward pass of the whole MLP in ONE kernel (mixed precision), the twin of mlp3_elu_mfma_kernel:
//   gz3 [n, C3] = (dG [n, K0] Wt0^T) * elu'(a3)     Wt0 [C3, K0] = the MLP block of w_ih, transposed (K0 = 4H, streamed)
//   gz2 [n, C2] = (gz3 Wt1^T) * elu'(a2)            Wt1 [C2, C3] = W3^T
//   gz1 [n, C1] = (gz2 Wt2^T) * elu'(a1)            Wt2 [C1, C2] = W2^T
// plus one row of column sums of each gz per workgroup (the partial bias gradients).  Formerly three launches
// (linear_bwd_elu_mfma_chunked_kernel<8>, linear_bwd_elu_mfma_kernel<2>, <4>: 15 + 9 + 18 us in the update).  A wave
// carries its 16 rows through the three products in registers: the output tiles of a product are ordered so that a lane
// ends up with the 8 consecutive units it needs as its B fragment of the next product (mlp3_tile_row), which also makes
// every gz store and every activation load one 16-B piece.  Stage 1 streams K0 in 128-wide chunks exactly like the
// chunked kernel (weight chunk double-buffered in LDS, dG fragments double-buffered in registers); the two small
// weights are requested at the start and parked in LDS after stage 1.  Column sums: DPP row sums over the 16 rows of
// a wave (dpp_row_sum16), then across the waves through LDS -- fixed order, no atomics.
template <int C3, int C2, int C1, int NCH, int NW>
__device__ __forceinline__ void mlp3_bwd_elu_mfma_body(
    long long n, const lp16_t* __restrict__ G, long long ldg, const lp16_t* __restrict__ Wt0, long long ldw0,
    const lp16_t* __restrict__ Wt1, long long ldw1, const lp16_t* __restrict__ Wt2, long long ldw2,
    const lp16_t* __restrict__ a3, long long a3_stride, const lp16_t* __restrict__ a2, const lp16_t* __restrict__ a1,
    float alpha, lp16_t* __restrict__ gz3, lp16_t* __restrict__ gz2, lp16_t* __restrict__ gz1,
    float* __restrict__ part3, float* __restrict__ part2, float* __restrict__ part1) {
    constexpr int CK = 128, PC = CK + LDS_SKEW, P1 = C3 + LDS_SKEW, P2 = C2 + LDS_SKEW, TH = 64 * NW;
    constexpr int NPC = C3 * (CK / 8) / TH;                     // weight pieces of a chunk per thread
    constexpr int N1 = C2 * (C3 / 8) / TH, N2 = C1 * (C2 / 8) / TH;
    static_assert(C3 == 64 && C3 * (CK / 8) % TH == 0 && C2 * (C3 / 8) % TH == 0 && C1 * (C2 / 8) % TH == 0, "shapes");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    lp16_t* wc = reinterpret_cast<lp16_t*>(lds_raw);            // [2][C3][PC] chunk buffers of Wt0, rows in tile order
    lp16_t* w1l = wc + 2 * C3 * PC;                             // [C2][P1], rows in tile order
    lp16_t* w2l = w1l + C2 * P1;                                // [C1][P2], rows in tile order
    float* red = reinterpret_cast<float*>(w2l + C1 * P2);       // [NW][C3 + C2 + C1] column sums of the waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const long long b = ((long long)blockIdx.x * NW + wave) * 16 + i;
    const lp16_t* grow = G + b * ldg + 8 * q;
    // ---- requests: the two small weights, chunk 0 of Wt0, this lane's dG fragments of chunk 0
    u32x4_t s1[N1], s2[N2], wp[NPC];
#pragma unroll
    for (int it = 0; it < N1; ++it) {
        const int c = tid + TH * it, row = c / (C3 / 8), ck = c - row * (C3 / 8);
        s1[it] = *reinterpret_cast<const u32x4_t*>(Wt1 + (long long)row * ldw1 + 8 * ck);
    }
#pragma unroll
    for (int it = 0; it < N2; ++it) {
        const int c = tid + TH * it, row = c / (C2 / 8), ck = c - row * (C2 / 8);
        s2[it] = *reinterpret_cast<const u32x4_t*>(Wt2 + (long long)row * ldw2 + 8 * ck);
    }
#define MB_LOAD_W(ch)                                                                                          \
    _Pragma("unroll") for (int it = 0; it < NPC; ++it) {                                                       \
        const int c_ = tid + TH * it;                                                                          \
        wp[it] = *reinterpret_cast<const u32x4_t*>(Wt0 + (long long)(c_ >> 4) * ldw0 + (ch) * CK + 8 * (c_ & 15)); \
    }
#define MB_STORE_W(buf)                                                                                        \
    _Pragma("unroll") for (int it = 0; it < NPC; ++it) {                                                       \
        const int c_ = tid + TH * it;                                                                          \
        *reinterpret_cast<u32x4_t*>(&wc[((buf) * C3 + mlp3_tile_row(c_ >> 4)) * PC + 8 * (c_ & 15)]) = wp[it]; \
    }
    lp16x8_t gq[3][4];                                          // dG fragments of chunks c, c + 1, c + 2 (two in flight)
#define MB_LOAD_G(ch)                                                                                          \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_)                                                           \
        gq[(ch) % 3][k_] = *reinterpret_cast<const lp16x8_t*>(grow + (ch) * CK + 32 * k_);
    MB_LOAD_W(0)
    MB_LOAD_G(0)
    if (NCH > 1) { MB_LOAD_G(1) }
    // the stored activations of this lane's units (ELU' of the three layers): requested here, 56 registers, so that their
    // 29 MB travel under stage 1's stream instead of as 14 exposed round trips in the finishing steps (31.3 -> us)
    uint4 pa3[C3 / 32], pa2[C2 / 32], pa1[C1 / 32];
#pragma unroll
    for (int kk = 0; kk < C3 / 32; ++kk) pa3[kk] = *reinterpret_cast<const uint4*>(a3 + b * a3_stride + 32 * kk + 8 * q);
#pragma unroll
    for (int kp = 0; kp < C2 / 32; ++kp) pa2[kp] = *reinterpret_cast<const uint4*>(a2 + b * C2 + 32 * kp + 8 * q);
#pragma unroll
    for (int kp = 0; kp < C1 / 32; ++kp) pa1[kp] = *reinterpret_cast<const uint4*>(a1 + b * C1 + 32 * kp + 8 * q);
    MB_STORE_W(0)
    __syncthreads();
    // ---- stage 1: C3 = 64 units = tiles (pair kk, ut), K0 streamed
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) { MB_LOAD_W(c + 1) }
        if (c + 2 < NCH) { MB_LOAD_G(c + 2) }
        const lp16_t* wb = wc + (c & 1) * C3 * PC;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const lp16_t* wr = wb + (t * 16 + i) * PC + 8 * q;
#pragma unroll
            for (int k_ = 0; k_ < 4; ++k_)
                acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr + 32 * k_), gq[c % 3][k_], acc[t]);
        }
        if (c + 1 < NCH) { MB_STORE_W((c + 1) & 1) }            // buffer (c+1)&1 was last read before the previous barrier
        __syncthreads();
    }
#undef MB_LOAD_W
#undef MB_STORE_W
#undef MB_LOAD_G
    // the two small weights -> LDS (their loads have long landed)
#pragma unroll
    for (int it = 0; it < N1; ++it) {
        const int c = tid + TH * it, row = c / (C3 / 8), ck = c - row * (C3 / 8);
        *reinterpret_cast<u32x4_t*>(&w1l[mlp3_tile_row(row) * P1 + 8 * ck]) = s1[it];
    }
#pragma unroll
    for (int it = 0; it < N2; ++it) {
        const int c = tid + TH * it, row = c / (C2 / 8), ck = c - row * (C2 / 8);
        *reinterpret_cast<u32x4_t*>(&w2l[mlp3_tile_row(row) * P2 + 8 * ck]) = s2[it];
    }
    float* myred = red + wave * (C3 + C2 + C1);
    // gz = acc * elu'(a) for the pair of tiles holding units u0 .. u0 + 7 of this lane; returns the packed bf16 piece,
    // stores it, and leaves the wave's column sums of the 8 units in LDS (lane 15 of every DPP row)
#define MB_FINISH(a_lo, a_hi, areg, gzptr, redoff, pk)                                                         \
    {                                                                                                          \
        float av_[8], d_[8];                                                                                   \
        unpack_lp16x8(areg, av_);                                                                              \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
            d_[e] = (a_lo)[e] * (av_[e] > 0.0f ? 1.0f : av_[e] + alpha);                                       \
            d_[4 + e] = (a_hi)[e] * (av_[4 + e] > 0.0f ? 1.0f : av_[4 + e] + alpha);                           \
        }                                                                                                      \
        pk = pack_lp16x8(d_);                                                                                  \
        *reinterpret_cast<uint4*>(gzptr) = pk;                                                                 \
        float r_[8];                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) r_[e] = dpp_row_sum16(d_[e]);                            \
        if (i == 15) {                                                                                         \
            st4(myred + (redoff), make_float4(r_[0], r_[1], r_[2], r_[3]));                                    \
            st4(myred + (redoff) + 4, make_float4(r_[4], r_[5], r_[6], r_[7]));                                \
        }                                                                                                      \
    }
    lp16x8_t gf2[C3 / 32];
#pragma unroll
    for (int kk = 0; kk < C3 / 32; ++kk) {
        uint4 pk;
        MB_FINISH(acc[2 * kk], acc[2 * kk + 1], pa3[kk], gz3 + b * C3 + 32 * kk + 8 * q, 32 * kk + 8 * q, pk)
        gf2[kk] = __builtin_bit_cast(lp16x8_t, pk);
    }
    __syncthreads();                                             // w1l / w2l complete
    // ---- stage 2: C2 units, K = C3
    lp16x8_t gf1[C2 / 32];
#pragma unroll
    for (int kp = 0; kp < C2 / 32; ++kp) {
        f32x4_t a[2] = {f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int kk = 0; kk < C3 / 32; ++kk)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) {
                const lp16x8_t wf = *reinterpret_cast<const lp16x8_t*>(&w1l[(32 * kp + 16 * ut + i) * P1 + 32 * kk + 8 * q]);
                a[ut] = MFMA_LP16(wf, gf2[kk], a[ut]);
            }
        uint4 pk;
        MB_FINISH(a[0], a[1], pa2[kp], gz2 + b * C2 + 32 * kp + 8 * q, C3 + 32 * kp + 8 * q, pk)
        gf1[kp] = __builtin_bit_cast(lp16x8_t, pk);
    }
    // ---- stage 3: C1 units, K = C2
#pragma unroll
    for (int kp = 0; kp < C1 / 32; ++kp) {
        f32x4_t a[2] = {f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int kk = 0; kk < C2 / 32; ++kk)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) {
                const lp16x8_t wf = *reinterpret_cast<const lp16x8_t*>(&w2l[(32 * kp + 16 * ut + i) * P2 + 32 * kk + 8 * q]);
                a[ut] = MFMA_LP16(wf, gf1[kk], a[ut]);
            }
        uint4 pk;
        MB_FINISH(a[0], a[1], pa1[kp], gz1 + b * C1 + 32 * kp + 8 * q, C3 + C2 + 32 * kp + 8 * q, pk)
        (void)pk;
    }
#undef MB_FINISH
    // ---- column sums of the workgroup: the waves' rows added in wave order
    __syncthreads();
    for (int u = tid; u < C3 + C2 + C1; u += TH) {
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += red[w * (C3 + C2 + C1) + u];
        float* dst = u < C3 ? part3 + (long long)blockIdx.x * C3 + u
                            : (u < C3 + C2 ? part2 + (long long)blockIdx.x * C2 + (u - C3)
                                           : part1 + (long long)blockIdx.x * C1 + (u - C3 - C2));
        *dst = sum;
    }
}
template <int C3, int C2, int C1, int NCH, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 2))) void mlp3_bwd_elu_mfma_kernel(
    long long n, const lp16_t* __restrict__ G, long long ldg, const lp16_t* __restrict__ Wt0, long long ldw0,
    const lp16_t* __restrict__ Wt1, long long ldw1, const lp16_t* __restrict__ Wt2, long long ldw2,
    const lp16_t* __restrict__ a3, long long a3_stride, const lp16_t* __restrict__ a2, const lp16_t* __restrict__ a1,
    float alpha, lp16_t* __restrict__ gz3, lp16_t* __restrict__ gz2, lp16_t* __restrict__ gz1,
    float* __restrict__ part3, float* __restrict__ part2, float* __restrict__ part1) {
    mlp3_bwd_elu_mfma_body<C3, C2, C1, NCH, NW>(n, G, ldg, Wt0, ldw0, Wt1, ldw1, Wt2, ldw2, a3, a3_stride, a2, a1, alpha, gz3,
                                                gz2, gz1, part3, part2, part1);
}

// ---- Round 4: the LSTM backward pass and the MLP backward pass as TWO PHASES OF ONE LAUNCH.  Both kernels give workgroup
// b the same 128 samples -- sequences [32 b, 32 b + 32) x T = 4 steps are rows [128 b, 128 b + 128) of the sequence-major
// sample order -- and the MLP phase reads exactly the dG rows the LSTM phase of the SAME workgroup wrote, so nothing but a
// workgroup barrier separates them: no grid fill / drain between the two (phase-in-launch, MI355X_MICROARCH.md price
// list), and the MLP phase's K = 1024 stream of dG finds its rows in the CU's L2.  The weight-gradient kernels (reductions
// over ALL rows) stay launches of their own behind it.  LDS: the phases use the same dynamic block one after the other.
template <int RING, typename CT, typename GT>
__global__ __launch_bounds__(512) void lstm_seq_bwd_mlp3_bwd_kernel(
    int T, long long B, const GT* __restrict__ g_out, const uint4* __restrict__ Wt, const lp16_t* __restrict__ gates,
    const CT* __restrict__ c_all, const float* __restrict__ c0, const unsigned char* __restrict__ done,
    lp16_t* dG, float* __restrict__ bias_partial, int ablate, const float* __restrict__ c_last,
    const lp16_t* __restrict__ Wt0, long long ldw0, const lp16_t* __restrict__ Wt1, long long ldw1,
    const lp16_t* __restrict__ Wt2, long long ldw2, const lp16_t* __restrict__ a3, long long a3_stride,
    const lp16_t* __restrict__ a2, const lp16_t* __restrict__ a1, float alpha, lp16_t* __restrict__ gz3,
    lp16_t* __restrict__ gz2, lp16_t* __restrict__ gz1, float* __restrict__ part3, float* __restrict__ part2,
    float* __restrict__ part1) {
    lstm_seq_bwd_body<RING, CT, GT>(T, B, g_out, Wt, gates, c_all, c0, done, dG, bias_partial, ablate, c_last);
    __syncthreads();      // (all of this workgroup's dG rows are written and visible to its own waves)
    mlp3_bwd_elu_mfma_body<64, 128, 256, 8, 8>(B * T, dG, 4 * SEQ_H, Wt0, ldw0, Wt1, ldw1, Wt2, ldw2, a3, a3_stride, a2, a1,
                                              alpha, gz3, gz2, gz1, part3, part2, part1);
}

template <typename DG>
__global__ void lstm_bwd_kernel(long long B, int H, const float* __restrict__ g_out, long long g_stride,
                                const float* __restrict__ g_rec, const float* __restrict__ dc_next,
                                const unsigned char* __restrict__ done_next, long long done_next_stride,
                                const DG* __restrict__ gates_act, const float* __restrict__ c_new,
                                const float* __restrict__ c_prev, const unsigned char* __restrict__ done,
                                long long done_stride, DG* __restrict__ dgates, long long dg_stride,
                                float* __restrict__ dc_prev, float* __restrict__ bias_partial,
                                const float* __restrict__ bias_partial_prev) {
    const int H4 = H >> 2;
    const long long total = B * H4;
    // bias_partial (nullable, [gridDim.x, 4H]): requires blockDim.x % H4 == 0 so that a thread keeps its column quad
    float acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.0f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / H4;
        const int j = (int)(idx - b * H4) << 2;
        const float keep = done ? 1.0f - (float)done[b * done_stride] : 1.0f;
        const float keep_n = done_next ? 1.0f - (float)done_next[b * done_next_stride] : 1.0f;
        const float4 go4 = ld4(g_out + b * g_stride + j);
        float dh[4] = {go4.x, go4.y, go4.z, go4.w};
        float dc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (g_rec) {
            const float4 r = ld4(g_rec + b * H + j);
            dh[0] += keep_n * r.x; dh[1] += keep_n * r.y; dh[2] += keep_n * r.z; dh[3] += keep_n * r.w;
        }
        if (dc_next) {
            const float4 r = ld4(dc_next + b * H + j);
            dc[0] = keep_n * r.x; dc[1] = keep_n * r.y; dc[2] = keep_n * r.z; dc[3] = keep_n * r.w;
        }
        const DG* ga = gates_act + b * 4LL * H;
        const float4 i4 = ld4(ga + j), f4 = ld4(ga + H + j), g4 = ld4(ga + 2 * H + j), o4 = ld4(ga + 3 * H + j);
        const float4 cn4 = ld4(c_new + b * H + j), cp4 = ld4(c_prev + b * H + j);
        const float gi[4] = {i4.x, i4.y, i4.z, i4.w}, gf[4] = {f4.x, f4.y, f4.z, f4.w};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, go[4] = {o4.x, o4.y, o4.z, o4.w};
        const float cn[4] = {cn4.x, cn4.y, cn4.z, cn4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
        float di[4], df[4], dg[4], dout[4], dcp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float tc = tanhf_(cn[u]);
            const float d_o = dh[u] * tc;
            const float d_c = dc[u] + dh[u] * go[u] * (1.0f - tc * tc);
            di[u] = d_c * gg[u] * gi[u] * (1.0f - gi[u]);
            df[u] = d_c * (keep * cp[u]) * gf[u] * (1.0f - gf[u]);
            dg[u] = d_c * gi[u] * (1.0f - gg[u] * gg[u]);
            dout[u] = d_o * go[u] * (1.0f - go[u]);
            dcp[u] = d_c * gf[u];
        }
        DG* dgp = dgates + b * dg_stride;
        st4(dgp + 0 * H + j, make_float4(di[0], di[1], di[2], di[3]));
        st4(dgp + 1 * H + j, make_float4(df[0], df[1], df[2], df[3]));
        st4(dgp + 2 * H + j, make_float4(dg[0], dg[1], dg[2], dg[3]));
        st4(dgp + 3 * H + j, make_float4(dout[0], dout[1], dout[2], dout[3]));
        st4(dc_prev + b * H + j, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
        if (bias_partial) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] += di[u]; acc[4 + u] += df[u]; acc[8 + u] += dg[u]; acc[12 + u] += dout[u];
            }
        }
    }
    if (bias_partial) {
        // threads tid, tid + H4, tid + 2*H4 ... share a column quad: fold them through LDS, one row per block
        __shared__ float red[256 * 16];
#pragma unroll
        for (int u = 0; u < 16; ++u) red[u * 256 + threadIdx.x] = acc[u];
        __syncthreads();
        if ((int)threadIdx.x < H4) {
            float* row = bias_partial + (long long)blockIdx.x * 4 * H;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int col = (u >> 2) * H + ((int)threadIdx.x << 2) + (u & 3);
                // chained over the time steps: this step's row continues the row the previous launch wrote
                float sum = bias_partial_prev ? bias_partial_prev[(long long)blockIdx.x * 4 * H + col] : 0.0f;
                for (int t = threadIdx.x; t < (int)blockDim.x; t += H4) sum += red[u * 256 + t];
                row[col] = sum;
            }
        }
    }
}

// ---- LSTM step backward with the recurrent input gradient fused in (mixed precision): g_rec = dG_{t+1} W_hh is
// produced on the matrix cores exactly as in linear_bwd_elu_mfma_chunked_kernel (64 rows x 64 hidden units per
// workgroup, the 4H-long reduction streamed in 128-k chunks), handed over through LDS, and the pointwise backward of
// lstm_bwd_kernel runs on it with whole 256-B row segments per 16 lanes: g_rec never reaches HBM and the separate
// [B, 4H] x [4H, H] GEMM launch disappears.  NCH = 0: the last time step (no recurrent gradient).
// bias_partial: [B / 64, 4H] rows, chained over the time steps like lstm_bwd_kernel's.
template <int NCH>
__global__ __launch_bounds__(256) void lstm_bwd_mfma_kernel(
    long long B, int H, const float* __restrict__ g_out, long long g_stride, const lp16_t* __restrict__ G, long long ldg,
    const lp16_t* __restrict__ Wt, long long ldw, const float* __restrict__ dc_next,
    const unsigned char* __restrict__ done_next, long long done_next_stride, const lp16_t* __restrict__ gates_act,
    const float* __restrict__ c_new, const float* __restrict__ c_prev, const unsigned char* __restrict__ done,
    long long done_stride, lp16_t* __restrict__ dgates, long long dg_stride, float* __restrict__ dc_prev,
    float* __restrict__ bias_partial, const float* __restrict__ bias_partial_prev) {
    constexpr int CK = 128;
    constexpr int PITCH = CK + 8;
    constexpr int GP = 68;                                     // floats per row of the g_rec hand-over tile
    __shared__ __attribute__((aligned(16))) lp16_t wl[2][64 * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u0 = blockIdx.y * 64;
    float* gt = reinterpret_cast<float*>(&wl[0][0]);           // 64 x 68 floats == one weight buffer
    // pointwise part: thread -> unit quad q of rows rg, rg + 16, rg + 32, rg + 48.  The operands of its first pass are
    // requested here, ahead of the matrix work, so that their HBM round trip is not exposed after the hand-over (the
    // later passes' loads overlap with the arithmetic of the pass before them)
    const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j = u0 + 4 * q;
    const long long b0 = (long long)blockIdx.x * 64 + rg;
    const float4 pre_go = ld4(g_out + b0 * g_stride + j);
    const float4 pre_dc = dc_next ? ld4(dc_next + b0 * H + j) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 pre_i = ld4(gates_act + b0 * 4LL * H + j), pre_f = ld4(gates_act + b0 * 4LL * H + H + j);
    const float4 pre_g = ld4(gates_act + b0 * 4LL * H + 2 * H + j), pre_o = ld4(gates_act + b0 * 4LL * H + 3 * H + j);
    const float4 pre_cn = ld4(c_new + b0 * H + j), pre_cp = ld4(c_prev + b0 * H + j);
    float keep_pre[4], keepn_pre[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        keep_pre[p] = done ? 1.0f - (float)done[(b0 + 16 * p) * done_stride] : 1.0f;
        keepn_pre[p] = done_next ? 1.0f - (float)done_next[(b0 + 16 * p) * done_next_stride] : 1.0f;
    }
    if (NCH > 0) {
        const long long b = (long long)blockIdx.x * 64 + wave * 16 + (lane & 15);
        const lp16_t* grow = G + b * ldg + 8 * (lane >> 4);
        const int prow[4] = {(int)threadIdx.x >> 4, ((int)threadIdx.x + 256) >> 4, ((int)threadIdx.x + 512) >> 4,
                             ((int)threadIdx.x + 768) >> 4};
        const int pcol = threadIdx.x & 15;
        // three register sets {weight pieces, G fragments}: chunk c + 2 is requested while chunk c is multiplied
        uint4 wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3, wc0, wc1, wc2, wc3;
        lp16x8_t ga0, ga1, ga2, ga3, gb0, gb1, gb2, gb3, gc0, gc1, gc2, gc3;
#define LOAD_WG(S, c)                                                                                            \
    w##S##0 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[0]) * ldw + (c) * CK + pcol * 8);       \
    w##S##1 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[1]) * ldw + (c) * CK + pcol * 8);       \
    w##S##2 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[2]) * ldw + (c) * CK + pcol * 8);       \
    w##S##3 = *reinterpret_cast<const uint4*>(Wt + (long long)(u0 + prow[3]) * ldw + (c) * CK + pcol * 8);       \
    g##S##0 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK);                                               \
    g##S##1 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK + 32);                                          \
    g##S##2 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK + 64);                                          \
    g##S##3 = *reinterpret_cast<const lp16x8_t*>(grow + (c) * CK + 96)
#define STORE_W(S, c)                                                                                            \
    *reinterpret_cast<uint4*>(&wl[(c) & 1][prow[0] * PITCH + pcol * 8]) = w##S##0;                               \
    *reinterpret_cast<uint4*>(&wl[(c) & 1][prow[1] * PITCH + pcol * 8]) = w##S##1;                               \
    *reinterpret_cast<uint4*>(&wl[(c) & 1][prow[2] * PITCH + pcol * 8]) = w##S##2;                               \
    *reinterpret_cast<uint4*>(&wl[(c) & 1][prow[3] * PITCH + pcol * 8]) = w##S##3
        f32x4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#define BWD_STEP(c, SL, SC, SS)                                                                                  \
    if (NCH > (c)) {                                                                                             \
        if ((c) > 0 && (c) + 2 < NCH) { LOAD_WG(SL, (c) + 2); }                                                  \
        const lp16_t* wb_ = wl[(c) & 1];                                                                         \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                          \
            const lp16_t* wr = wb_ + (t * 16 + (lane & 15)) * PITCH + 8 * (lane >> 4);                           \
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr), g##SC##0, acc[t]);      \
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr + 32), g##SC##1, acc[t]); \
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr + 64), g##SC##2, acc[t]); \
            acc[t] = MFMA_LP16(*reinterpret_cast<const lp16x8_t*>(wr + 96), g##SC##3, acc[t]); \
        }                                                                                                        \
        if ((c) + 1 < NCH) { STORE_W(SS, (c) + 1); }                                                             \
        __syncthreads();                                                                                         \
    }
        LOAD_WG(a, 0);
        if (NCH > 1) { LOAD_WG(b, 1); }
        if (NCH > 2) { LOAD_WG(c, 2); }
        STORE_W(a, 0);
        __syncthreads();
        BWD_STEP(0, c, a, b) BWD_STEP(1, a, b, c) BWD_STEP(2, b, c, a) BWD_STEP(3, c, a, b)
        BWD_STEP(4, a, b, c) BWD_STEP(5, b, c, a) BWD_STEP(6, c, a, b) BWD_STEP(7, a, b, c)
        static_assert(NCH <= 8, "lstm_bwd_mfma_kernel: 4H <= 1024");
#undef BWD_STEP
#undef LOAD_WG
#undef STORE_W
        // accumulators -> [row][unit] tile (a lane holds 4 consecutive units of one row per 16-unit tile)
        const int r = wave * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<float4*>(&gt[r * GP + 16 * t + 4 * (lane >> 4)]) =
                make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        __syncthreads();
    }
    // ---- pointwise backward
    float bsum[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) bsum[u] = 0.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = rg + 16 * p;
        const long long b = (long long)blockIdx.x * 64 + r;
        const float keep = keep_pre[p], keep_n = keepn_pre[p];
        const float4 go4 = p == 0 ? pre_go : ld4(g_out + b * g_stride + j);
        float dh[4] = {go4.x, go4.y, go4.z, go4.w};
        float dc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (NCH > 0) {
            const float4 rr = *reinterpret_cast<const float4*>(&gt[r * GP + 4 * q]);
            dh[0] += keep_n * rr.x; dh[1] += keep_n * rr.y; dh[2] += keep_n * rr.z; dh[3] += keep_n * rr.w;
        }
        if (dc_next) {
            const float4 rr = p == 0 ? pre_dc : ld4(dc_next + b * H + j);
            dc[0] = keep_n * rr.x; dc[1] = keep_n * rr.y; dc[2] = keep_n * rr.z; dc[3] = keep_n * rr.w;
        }
        const lp16_t* ga = gates_act + b * 4LL * H;
        const float4 i4 = p == 0 ? pre_i : ld4(ga + j), f4 = p == 0 ? pre_f : ld4(ga + H + j);
        const float4 g4 = p == 0 ? pre_g : ld4(ga + 2 * H + j), o4 = p == 0 ? pre_o : ld4(ga + 3 * H + j);
        const float4 cn4 = p == 0 ? pre_cn : ld4(c_new + b * H + j), cp4 = p == 0 ? pre_cp : ld4(c_prev + b * H + j);
        const float gi[4] = {i4.x, i4.y, i4.z, i4.w}, gf[4] = {f4.x, f4.y, f4.z, f4.w};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, go[4] = {o4.x, o4.y, o4.z, o4.w};
        const float cn[4] = {cn4.x, cn4.y, cn4.z, cn4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
        float di[4], df[4], dg[4], dout[4], dcp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float tc = tanhf_(cn[u]);
            const float d_o = dh[u] * tc;
            const float d_c = dc[u] + dh[u] * go[u] * (1.0f - tc * tc);
            di[u] = d_c * gg[u] * gi[u] * (1.0f - gi[u]);
            df[u] = d_c * (keep * cp[u]) * gf[u] * (1.0f - gf[u]);
            dg[u] = d_c * gi[u] * (1.0f - gg[u] * gg[u]);
            dout[u] = d_o * go[u] * (1.0f - go[u]);
            dcp[u] = d_c * gf[u];
        }
        lp16_t* dgp = dgates + b * dg_stride;
        st4(dgp + 0 * H + j, make_float4(di[0], di[1], di[2], di[3]));
        st4(dgp + 1 * H + j, make_float4(df[0], df[1], df[2], df[3]));
        st4(dgp + 2 * H + j, make_float4(dg[0], dg[1], dg[2], dg[3]));
        st4(dgp + 3 * H + j, make_float4(dout[0], dout[1], dout[2], dout[3]));
        st4(dc_prev + b * H + j, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bsum[u] += di[u]; bsum[4 + u] += df[u]; bsum[8 + u] += dg[u]; bsum[12 + u] += dout[u];
        }
    }
    if (bias_partial) {
        // 16 row groups x (16 quads x 16 values): fold the row groups through LDS in a fixed order
        float* red = reinterpret_cast<float*>(&wl[1][0]);      // 16 x 256 floats = 16 KB <= one weight buffer
#pragma unroll
        for (int u = 0; u < 16; ++u) red[rg * 256 + u * 16 + q] = bsum[u];
        __syncthreads();
        const int u = threadIdx.x >> 4, qq = threadIdx.x & 15;              // value u = 4 gate + unit of quad qq
        const long long col = (long long)(u >> 2) * H + u0 + 4 * qq + (u & 3);
        float sum = bias_partial_prev ? bias_partial_prev[(long long)blockIdx.x * 4 * H + col] : 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) sum += red[g * 256 + threadIdx.x];
        bias_partial[(long long)blockIdx.x * 4 * H + col] = sum;
    }
}

// ---- Weight gradient on the matrix cores: part[s][m][n] = sum over the rows k of slice s of dy[k][m] * x[k][n]
// (dy [rows, M], x [rows, N] bf16, both with the reduction index as the SLOW dimension -- the layout the backward
// kernels produce).  Both MFMA operands want 8 consecutive k per lane, i.e. a column of the row-major tiles: the
// tiles are staged row-major in LDS ([32 rows][cols], pitch = cols * 2 + 32 B) and read with the gfx950 transposed
// LDS read (ds_read_b64_tr_b16: a 16-lane group fetches a 4-row x 16-column block and gets it column-major).
// The k <-> lane-group assignment of an MFMA is free as long as both operands agree: group g takes rows 4g .. 4g+3
// and 16+4g .. 16+4g+3 of the stage, so that a 32-lane half always reads 8 consecutive rows, which the pitch spreads
// over all 64 banks.  Workgroup = 4 waves along M: (64 MT) x (16 NT) outputs; stages of 32 rows, double-buffered in
// LDS behind a register prefetch, one barrier per stage.  The slices are summed by the column-sum kernel
// (deterministic, no atomics).
typedef __attribute__((__vector_size__(4 * sizeof(lp16_hw)))) lp16_hw lp16x4_t;
#define VINE_LDS __attribute__((address_space(3)))
__device__ __forceinline__ lp16x8_t tr_read8(const lp16_t* p, int second_block_elems) {
    // two transposed reads: rows (.., +3) at p and the block `second_block_elems` further on
    typedef __attribute__((__vector_size__(4 * sizeof(lp16_tr_hw)))) lp16_tr_hw tr4_t;
    const lp16x4_t lo = __builtin_bit_cast(lp16x4_t, DS_READ_TR16_B64((VINE_LDS tr4_t*)(p)));
    const lp16x4_t hi = __builtin_bit_cast(lp16x4_t, DS_READ_TR16_B64((VINE_LDS tr4_t*)(p + second_block_elems)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(int stages, const lp16_t* __restrict__ dy, long long ldy,
                                                         const lp16_t* __restrict__ x, long long ldx,
                                                         float* __restrict__ part, int M, int Nv) {
    constexpr int MTW = 64 * MT, NTW = 16 * NT;
    constexpr int PA = MTW + 16, PB = NTW + 16;                 // bf16 elements per LDS row
    constexpr int AC = MTW / 8, BC = NTW / 8;                   // 16-B pieces per row
    constexpr int AP = 32 * AC, BP = 32 * BC;
    constexpr int NA = (AP + 255) / 256, NB = (BP + 255) / 256;
    __shared__ __attribute__((aligned(16))) lp16_t al[2][32 * PA];
    __shared__ __attribute__((aligned(16))) lp16_t bl[2][32 * PB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * MTW, n0 = blockIdx.y * NTW;
    const long long k0 = (long long)blockIdx.z * stages * 32;
    // up to two 16-B pieces of each tile per thread and stage (named scalars: indexed arrays end up in scratch)
    const int pa0 = (int)threadIdx.x, pa1 = (int)threadIdx.x + 256;
    const bool va0 = pa0 < AP, va1 = NA > 1 && pa1 < AP, vb0 = pa0 < BP, vb1 = NB > 1 && pa1 < BP;
    const lp16_t* asrc0 = dy + (k0 + pa0 / AC) * ldy + m0 + 8 * (pa0 % AC);
    const lp16_t* asrc1 = dy + (k0 + pa1 / AC) * ldy + m0 + 8 * (pa1 % AC);
    const lp16_t* bsrc0 = x + (k0 + pa0 / BC) * ldx + n0 + 8 * (pa0 % BC);
    const lp16_t* bsrc1 = x + (k0 + pa1 / BC) * ldx + n0 + 8 * (pa1 % BC);
    const int aoff0 = (pa0 / AC) * PA + 8 * (pa0 % AC), aoff1 = (pa1 / AC) * PA + 8 * (pa1 % AC);
    const int boff0 = (pa0 / BC) * PB + 8 * (pa0 % BC), boff1 = (pa1 / BC) * PB + 8 * (pa1 % BC);
    uint4 wa0 = make_uint4(0, 0, 0, 0), wa1 = wa0, wb0 = wa0, wb1 = wa0;
#define WG_LOAD(it)                                                                              \
    if (va0) wa0 = *reinterpret_cast<const uint4*>(asrc0 + (long long)(it) * 32 * ldy);          \
    if (va1) wa1 = *reinterpret_cast<const uint4*>(asrc1 + (long long)(it) * 32 * ldy);          \
    if (vb0) wb0 = *reinterpret_cast<const uint4*>(bsrc0 + (long long)(it) * 32 * ldx);          \
    if (vb1) wb1 = *reinterpret_cast<const uint4*>(bsrc1 + (long long)(it) * 32 * ldx)
#define WG_STORE(buf)                                                                            \
    if (va0) *reinterpret_cast<uint4*>(&al[buf][aoff0]) = wa0;                                   \
    if (va1) *reinterpret_cast<uint4*>(&al[buf][aoff1]) = wa1;                                   \
    if (vb0) *reinterpret_cast<uint4*>(&bl[buf][boff0]) = wb0;                                   \
    if (vb1) *reinterpret_cast<uint4*>(&bl[buf][boff1]) = wb1
    WG_LOAD(0);
    WG_STORE(0);
    __syncthreads();
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    // transposed-read address of this lane inside a 16-column tile: row 4g + q, columns 4p .. 4p+3
    const int g = lane >> 4, il = lane & 15;
    const int ra = (4 * g + (il >> 2)) * PA + 4 * (il & 3) + wave * MT * 16;
    const int rb = (4 * g + (il >> 2)) * PB + 4 * (il & 3);
    for (int it = 0; it < stages; ++it) {
        const bool more = it + 1 < stages;
        if (more) { WG_LOAD(it + 1); }
        const lp16_t* ab = al[it & 1];
        const lp16_t* bb = bl[it & 1];
        lp16x8_t af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = tr_read8(ab + ra + 16 * mt, 16 * PA);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const lp16x8_t bf = tr_read8(bb + rb + 16 * nt, 16 * PB);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt][nt] = MFMA_LP16(af[mt], bf, acc[mt][nt]);
        }
        if (more) { WG_STORE((it + 1) & 1); }
        __syncthreads();
    }
#undef WG_LOAD
#undef WG_STORE
    // D[m = 4g + r][n = il] of tile (mt, nt)
    float* out = part + ((long long)blockIdx.z * M + m0 + (wave * MT) * 16 + 4 * g) * Nv + n0 + il;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (n0 + 16 * nt + il < Nv) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(long long)(16 * mt + r) * Nv + 16 * nt] = acc[mt][nt][r];
            }
}

// ---- Weight gradient(s) on the matrix cores, second generation: dy^T [x1 | x2] in ONE pass over dy,
//   part1[s][m][n] = sum_k dy[k][m] x1[k][n]  (n < Nv1),   part2[s][m][n] = sum_k dy[k][m] x2[k][n]  (n < Nv2)
// over the rows k of slice s; x1 may be empty (N1p = 0: a plain dy^T x2).  The LSTM's dW_ih and dW_hh come out of one
// launch -- dy = dG [n, 4H], the largest tensor of the backward pass, x1 = the step-input block of the operand buffer
// (92 + 4 pad columns), x2 = the masked hidden states -- and the three MLP weights use the same kernel.
// What the first generation (wgrad_mfma_kernel, and a 128 x 352-tile version of this one) got wrong was the SPLIT: few,
// large output tiles need many row slices to fill the chip, and every slice writes a full fp32 copy of the output
// (32 slices x 1.4 MB = 46 MB for the LSTM) that the column-sum kernel reads back -- as much HBM traffic as the
// operands.  Here the output tile is small, 64 x (16 NT) with NT <= 11, so that ~512 workgroups need only 8-64 slices:
//   workgroup = 4 waves along M (16 rows each), every wave NT column tiles: NT MFMAs against 2 (1 + NT) transposed LDS
//   reads per 32-row stage; stages double-buffered in LDS behind a register prefetch (one barrier per stage); 2+
//   workgroups per CU hide each other's barriers.  Tiles are staged row-major and read column-wise with
//   ds_read_b64_tr_b16 exactly as in wgrad_mfma_kernel; the LDS pitches are odd multiples of 32 B so that 8 consecutive
//   rows tile all 64 banks.  Block -> (slice, tile) keeps the tiles of a slice on one XCD (they share the operand rows:
//   blocks b and b + 8 share an L2).  Slices are summed by the column-sum kernel: fixed order, no atomics.
// One problem of a grouped launch (several independent products in ONE kernel: every node of the replayed update graph
// costs ~4.5 us of dispatch + drain, and the three MLP weight gradients are ready at the same time).
struct WgProblem {
    const lp16_t *dy, *x1, *x2;
    float *part1, *part2;
    long long ldy, ldx1, ldx2;
    int stages, mtiles, ntiles, slices, N1p, Nv1, Nv2, M, NT, first_block;
};
#define VINE_WGRAD_MAX_PROBLEMS 6
struct WgGroupArgs {
    WgProblem p[VINE_WGRAD_MAX_PROBLEMS];
    int n;
};
constexpr int WGC_LDS_BYTES = 2 * 32 * (64 + 16) * 2 + 2 * 32 * 176 * 2;      // dy tiles + the widest x tiles (NT = 11)

template <int NT>
__device__ __forceinline__ void wgrad_cat_body(const WgProblem& P, const int block, unsigned char* lds_raw) {
    const int stages = P.stages, mtiles = P.mtiles, ntiles = P.ntiles, slices = P.slices, N1p = P.N1p, Nv1 = P.Nv1,
              Nv2 = P.Nv2, M = P.M;
    const lp16_t* __restrict__ dy = P.dy;
    const lp16_t* __restrict__ x1 = P.x1;
    const lp16_t* __restrict__ x2 = P.x2;
    float* __restrict__ part1 = P.part1;
    float* __restrict__ part2 = P.part2;
    const long long ldy = P.ldy, ldx1 = P.ldx1, ldx2 = P.ldx2;
    constexpr int BM = 64, BN = 16 * NT;
    constexpr int PA = BM + 16;                                  // 160 B: 40 dwords = odd multiple of 8
    constexpr int PB = BN + (((8 * NT) % 16) == 8 ? 0 : 16);     // dwords per row = odd multiple of 8
    constexpr int BC = BN / 8, BP = 32 * BC;                     // 16-B pieces per B row / per B stage
    constexpr int NB = (BP + 255) / 256;
    static_assert(NB <= 3 && BP % 64 == 0, "at most 3 staging slots for the x tile; its end on a wave boundary");
    static_assert(2 * 32 * (PA + PB) * 2 <= WGC_LDS_BYTES, "LDS of the grouped kernel");
    lp16_t (*al)[32 * PA] = reinterpret_cast<lp16_t (*)[32 * PA]>(lds_raw);
    lp16_t (*bl)[32 * PB] = reinterpret_cast<lp16_t (*)[32 * PB]>(lds_raw + 2 * 32 * PA * sizeof(lp16_t));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // blocks b, b + 8, ... share an XCD: give one XCD all tiles of its slices
    const int xcd = block & 7, q = block >> 3, tps = mtiles * ntiles;
    const int slice = xcd * (slices >> 3) + q / tps, tile = q % tps;
    const int m0 = (tile / ntiles) * BM, n0 = (tile % ntiles) * BN;      // n0: column of the virtual [x1 | x2]
    const long long k0 = (long long)slice * stages * 32;
    // staging: piece p = tid + 256 i; the dy piece is one per thread; x pieces pick their operand by virtual column
    // (a pointer select, not a branch); named scalars (an indexed array lands in scratch memory); a slot past the last
    // piece re-reads piece 0 and is never stored
    const lp16_t* asrc = dy + (k0 + tid / 8) * ldy + m0 + 8 * (tid % 8);
    const int aoff = (tid / 8) * PA + 8 * (tid % 8);
    // two stages in flight in two named register sets (e / o), as in wgrad_cat_wide_kernel: with one stage of prefetch the
    // kernel was bound by the latency of its own loads (16 stages x ~1 us for the MLP weights)
    uint4 ra_e, rb0_e, rb1_e, rb2_e, ra_o, rb0_o, rb1_o, rb2_o;
    rb1_e = rb2_e = rb1_o = rb2_o = make_uint4(0, 0, 0, 0);
#define WGC_SRC(i)                                                                                             \
    const lp16_t* bsrc##i; long long bstep##i; int boff##i; bool bval##i;                                      \
    {                                                                                                          \
        const int p = tid + 256 * (i);                                                                         \
        bval##i = p < BP;                                                                                      \
        const int pp = bval##i ? p : 0, row = pp / BC, c = n0 + 8 * (pp % BC);                                 \
        const bool first = c < N1p;                                                                            \
        bsrc##i = first ? x1 + (k0 + row) * ldx1 + c : x2 + (k0 + row) * ldx2 + (c - N1p);                     \
        bstep##i = 32 * (first ? ldx1 : ldx2);                                                                 \
        boff##i = row * PB + 8 * (pp % BC);                                                                    \
    }
    WGC_SRC(0) WGC_SRC(1) WGC_SRC(2)
#undef WGC_SRC
#define WGC_LOAD(S, it)                                                                                        \
    {                                                                                                          \
        const long long t_ = (it) < stages ? (it) : stages - 1;   /* past the end: a harmless re-read */        \
        ra_##S = *reinterpret_cast<const uint4*>(asrc + t_ * 32 * ldy);                                        \
        rb0_##S = *reinterpret_cast<const uint4*>(bsrc0 + t_ * bstep0);                                        \
        if (NB > 1) rb1_##S = *reinterpret_cast<const uint4*>(bsrc1 + t_ * bstep1);                            \
        if (NB > 2) rb2_##S = *reinterpret_cast<const uint4*>(bsrc2 + t_ * bstep2);                            \
        __builtin_amdgcn_sched_barrier(0);   /* requests leave before the products, not after them */          \
    }
#define WGC_STORE(S, buf)                                                                                      \
    *reinterpret_cast<uint4*>(&al[buf][aoff]) = ra_##S;                                                        \
    if (bval0) *reinterpret_cast<uint4*>(&bl[buf][boff0]) = rb0_##S;                                           \
    if (NB > 1 && bval1) *reinterpret_cast<uint4*>(&bl[buf][boff1]) = rb1_##S;                                 \
    if (NB > 2 && bval2) *reinterpret_cast<uint4*>(&bl[buf][boff2]) = rb2_##S;
    f32x4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    const int g = lane >> 4, il = lane & 15;
    const int ra = (4 * g + (il >> 2)) * PA + 4 * (il & 3) + wave * 16;
    const int rb = (4 * g + (il >> 2)) * PB + 4 * (il & 3);
#define WGC_COMPUTE(buf)                                                                                       \
    {                                                                                                          \
        const lp16_t* ab = al[buf];                                                                            \
        const lp16_t* bb = bl[buf];                                                                            \
        const lp16x8_t af = tr_read8(ab + ra, 16 * PA);                                                        \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                    \
            const lp16x8_t bfr = tr_read8(bb + rb + 16 * nt, 16 * PB);                                         \
            acc[nt] = MFMA_LP16(af, bfr, acc[nt]);                                                             \
        }                                                                                                      \
    }
    // invariant at the top of an even stage `it`: LDS buffer 0 holds stage it, the odd set holds stage it + 1 (in flight)
    WGC_LOAD(e, 0)
    WGC_LOAD(o, 1)
    WGC_STORE(e, 0)
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < stages; it += 2) {
        WGC_LOAD(e, it + 2)
        WGC_COMPUTE(0)
        WGC_STORE(o, 1)
        __syncthreads();
        if (it + 1 < stages) {                                   // (stages may be odd here: wave-uniform)
            WGC_LOAD(o, it + 3)
            WGC_COMPUTE(1)
            WGC_STORE(e, 0)
        }
        __syncthreads();
    }
#undef WGC_LOAD
#undef WGC_STORE
#undef WGC_COMPUTE
    // D[m = 4g + r][n = il] of column tile nt; virtual column -> (part1 | part2)
    const long long mrow = (long long)slice * M + m0 + wave * 16 + 4 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int vc = n0 + 16 * nt;                            // wave-uniform
        float* base;
        int Nv, col;
        if (vc < N1p) { Nv = Nv1; col = vc + il; base = part1; }
        else { Nv = Nv2; col = vc - N1p + il; base = part2; }
        if (col < Nv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[(mrow + r) * Nv + col] = acc[nt][r];
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgGroupArgs G) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[WGC_LDS_BYTES];
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < G.n; ++k)
        if ((int)blockIdx.x >= G.p[k].first_block) j = k;
    const WgProblem& P = G.p[j];
    const int block = (int)blockIdx.x - P.first_block;
    if (P.NT == 11) wgrad_cat_body<11>(P, block, lds_raw);
    else if (P.NT == 8) wgrad_cat_body<8>(P, block, lds_raw);
    else wgrad_cat_body<2>(P, block, lds_raw);
}

// ---- The same product on ONE wide output tile per workgroup, for the LSTM ([x 96 | h 256] = 352 columns): 128 x 352
// outputs, 8 waves as 4 (M) x 2 (N), wave tile = 32 x 176 (2 x 11 MFMA tiles, 88 accumulator registers), one workgroup
// per CU.  Against the 64 x 176 tiles of wgrad_cat_mfma_kernel<11> this moves a quarter of the operand bytes through L2
// (245 MB instead of 1 GB per launch, which is what bounds the small tile: 59.9 us in the update) at the price of twice
// the row slices (32: 46 MB of partial sums).  With one stage of prefetch the kernel is bound by the latency of its
// own loads (32 stages x 1.4 us); two stages are kept in flight in two named register sets (the loop is unrolled by two
// so that the set is a compile-time choice; an indexed array lands in scratch memory).
// SEQ ("h once"): the second operand is formed on the fly from the LSTM's one copy of its hidden states -- row k = seq * T + t
// of the operand is (1 - done[k]) * h_{t-1} = slot t of x2 = the 16-bit h_out [rows / T, T + 1, ldx2] of
// vine_lstm_seq_forward_mfma, i.e. its row k + k / T.  T divides 32 and a stage starts on a multiple of 32, so a stage
// advances every piece by the same 32 + 32 / T rows: one base pointer per piece and a uniform stride, as before.
// The done flags of the workgroup's row slice are packed once, in the prologue, into one 32-bit word per stage (LDS);
// a piece is zeroed on its way to LDS when its row's bit is set (a byte load per piece and stage instead cost 4.5 us).
template <int NT1, int MT>      // MT = 16-row MFMA tiles per wave along M: workgroup tile = (64 MT) x 352
__global__ __launch_bounds__(512) void wgrad_cat_wide_kernel(int stages, int mtiles, int slices, const lp16_t* __restrict__ dy,
                                                             long long ldy, const lp16_t* __restrict__ x1, long long ldx1,
                                                             const lp16_t* __restrict__ x2, long long ldx2,
                                                             float* __restrict__ part1, int Nv1, float* __restrict__ part2,
                                                             int Nv2, int M, const unsigned char* __restrict__ done, int T) {
    constexpr int NT = 11, WM = 4, WN = 2, TH = 64 * WM * WN;
    constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN, N1 = 16 * NT1, N2 = BN - N1;
    constexpr int PA = BM + 16, PB = BN + 16;
    constexpr int APC = BM / 8, AP = 32 * APC;                  // 16-B pieces per A row / per A stage
    const bool SEQ = done != nullptr;
    constexpr int B1C = N1 / 8, B1P = 32 * B1C, B2C = N2 / 8, B2P = 32 * B2C;
    static_assert(AP <= TH && AP % 64 == 0 && B1P + B2P <= 3 * TH && B1P + B2P > 2 * TH && B1P % 64 == 0 &&
                      (B1P + B2P) % 64 == 0,
                  "staging slots: <= 1 piece of dy and 3 of [x1 | x2] per thread, operand boundaries on wave boundaries");
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    lp16_t (*al)[32 * PA] = reinterpret_cast<lp16_t (*)[32 * PA]>(wg_lds);
    lp16_t (*bl)[32 * PB] = reinterpret_cast<lp16_t (*)[32 * PB]>(wg_lds + 2 * 32 * PA * sizeof(lp16_t));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // blocks b, b + 8, ... share an XCD: give one XCD all m tiles of its slices
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int slice = xcd * (slices >> 3) + q / mtiles, mtile = q % mtiles;
    const int m0 = mtile * BM;
    const long long k0 = (long long)slice * stages * 32;
    // per-thread source of piece p = tid + TH i (the slot past the last piece re-reads piece 0 and is never stored)
    const bool aval = tid < AP;                                  // (wave-uniform; the other waves re-read piece 0)
    const int ap = aval ? tid : 0;
    const lp16_t* asrc = dy + (k0 + ap / APC) * ldy + m0 + 8 * (ap % APC);
    const int aoff = (ap / APC) * PA + 8 * (ap % APC);
#define WGW_SRC(i)                                                                                             \
    const lp16_t* bsrc##i; long long bstep##i; int boff##i; bool bval##i;                                      \
    unsigned dbit##i = 0;                         /* SEQ: this piece's row bit in a stage's done word (x2 pieces only) */ \
    {                                                                                                          \
        const int p = tid + TH * (i);                                                                          \
        bval##i = p < B1P + B2P;                                                                               \
        if (p < B1P) {                                                                                         \
            bsrc##i = x1 + (k0 + p / B1C) * ldx1 + 8 * (p % B1C); bstep##i = 32 * ldx1;                        \
            boff##i = (p / B1C) * PB + 8 * (p % B1C);                                                          \
        } else {                                                                                               \
            const int p2 = bval##i ? p - B1P : 0;                                                              \
            const int r2 = p2 / B2C;                                                                           \
            /* (the plain form passes T = 2^30: row k, 32 rows per stage) */                                    \
            if (SEQ && bval##i) dbit##i = 1u << r2;                                                            \
            bsrc##i = x2 + (k0 + r2 + (k0 + r2) / T) * ldx2 + 8 * (p2 % B2C); bstep##i = (32 + 32 / T) * ldx2; \
            boff##i = r2 * PB + N1 + 8 * (p2 % B2C);                                                           \
        }                                                                                                      \
    }
    WGW_SRC(0) WGW_SRC(1) WGW_SRC(2)
#undef WGW_SRC
    uint4 ra_e, rb0_e, rb1_e, rb2_e, ra_o, rb0_o, rb1_o, rb2_o;     // stages of even / odd index in flight
    __shared__ unsigned dwords[1024];                            // SEQ: bit r of word `it` = done[k0 + 32 it + r]
    if (SEQ) {
        for (int it = tid; it < stages; it += TH) {
            const uint4 a = *reinterpret_cast<const uint4*>(done + k0 + 32 * it);
            const uint4 b = *reinterpret_cast<const uint4*>(done + k0 + 32 * it + 16);
            const unsigned q[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            unsigned wbits = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) wbits |= (((q[c] >> (8 * e)) & 0xffu) ? 1u : 0u) << (4 * c + e);
            dwords[it] = wbits;
        }
    }
#define WGW_LOAD(S, it)                                                                                        \
    {                                                                                                          \
        const long long t_ = (it) < stages ? (it) : stages - 1;   /* past the end: a harmless re-read */        \
        ra_##S = *reinterpret_cast<const uint4*>(asrc + t_ * 32 * ldy);                                        \
        rb0_##S = *reinterpret_cast<const uint4*>(bsrc0 + t_ * bstep0);                                        \
        rb1_##S = *reinterpret_cast<const uint4*>(bsrc1 + t_ * bstep1);                                        \
        rb2_##S = *reinterpret_cast<const uint4*>(bsrc2 + t_ * bstep2);                                        \
        __builtin_amdgcn_sched_barrier(0);   /* requests leave before the products, not after them */          \
    }
    /* branch-free: an all-ones / all-zeros word per piece, ANDed in (a select on the uint4 became exec-masked moves) */ \
#define WGW_MASKED(v, bit)                                                                                     \
    ([&] { const unsigned m_ = (dw_##bit) ? 0u : 0xffffffffu; return make_uint4((v).x & m_, (v).y & m_, (v).z & m_, (v).w & m_); }())
#define WGW_STORE(S, buf, it)                                                                                  \
    {                                                                                                          \
        const unsigned dw_ = SEQ ? dwords[(it) < stages ? (it) : stages - 1] : 0u;                             \
        const unsigned dw_dbit0 = dw_ & dbit0, dw_dbit1 = dw_ & dbit1, dw_dbit2 = dw_ & dbit2;                 \
        if (aval) *reinterpret_cast<uint4*>(&al[buf][aoff]) = ra_##S;                                          \
        *reinterpret_cast<uint4*>(&bl[buf][boff0]) = WGW_MASKED(rb0_##S, dbit0);                               \
        *reinterpret_cast<uint4*>(&bl[buf][boff1]) = WGW_MASKED(rb1_##S, dbit1);                               \
        if (bval2) *reinterpret_cast<uint4*>(&bl[buf][boff2]) = WGW_MASKED(rb2_##S, dbit2);                    \
    }
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    const int g = lane >> 4, il = lane & 15;
    const int ra = (4 * g + (il >> 2)) * PA + 4 * (il & 3) + wm * MT * 16;
    const int rb = (4 * g + (il >> 2)) * PB + 4 * (il & 3) + wn * NT * 16;
#define WGW_COMPUTE(buf)                                                                                       \
    {                                                                                                          \
        const lp16_t* ab = al[buf];                                                                            \
        const lp16_t* bb = bl[buf];                                                                            \
        lp16x8_t af[MT], bfr[NT];                                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) af[mt] = tr_read8(ab + ra + 16 * mt, 16 * PA);       \
        /* every fragment of the stage requested before the first product (the scheduler, left alone, sometimes     \
           pairs each read with its own wait: 57 instead of 50 us) */                                            \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bfr[nt] = tr_read8(bb + rb + 16 * nt, 16 * PB);      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                    \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                  \
                acc[mt][nt] = MFMA_LP16(af[mt], bfr[nt], acc[mt][nt]);                                         \
        }                                                                                                      \
    }
    // invariant at the top of an even stage `it`: LDS buffer 0 holds stage it, the odd set holds stage it + 1 (in flight)
    WGW_LOAD(e, 0)
    WGW_LOAD(o, 1)
    if (SEQ) __syncthreads();                                    // the done words
    WGW_STORE(e, 0, 0)
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < stages; it += 2) {                     // stages is even (host check)
        WGW_LOAD(e, it + 2)
        WGW_COMPUTE(0)
        WGW_STORE(o, 1, it + 1)
        __syncthreads();
        WGW_LOAD(o, it + 3)
        WGW_COMPUTE(1)
        WGW_STORE(e, 0, it + 2)
        __syncthreads();
    }
#undef WGW_LOAD
#undef WGW_STORE
#undef WGW_MASKED
#undef WGW_COMPUTE
    // D[m = 4g + r][n = il] of tile (mt, nt); tiles below NT1 (global tile index) belong to part1, the rest to part2
    const long long mrow = (long long)slice * M + m0 + wm * MT * 16 + 4 * g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int tg = wn * NT + nt;                            // wave-uniform
        float* base;
        int Nv, col;
        if (tg < NT1) { Nv = Nv1; col = 16 * tg + il; base = part1; }
        else { Nv = Nv2; col = 16 * (tg - NT1) + il; base = part2; }
        if (col < Nv) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) base[(mrow + 16 * mt + r) * Nv + col] = acc[mt][nt][r];
        }
    }
}

// ---- LayerNorm over rows of H = 256 * NV floats: one wave per row, lane l owns columns [256 v + 4 l, +4) ----
// Sum over the 64 lanes, returned to every lane: DPP row shifts and row broadcasts (7 VALU adds with a lane-shifted
// operand, total in lane 63) + one v_readlane, instead of 6 ds_bpermute round trips through the LDS crossbar.
// Needs all 64 lanes active (every caller runs it from wave-uniform control flow).
__device__ __forceinline__ float dpp_i2f(int x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ int dpp_f2i(float x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ float wave_sum(float v) {
    const int x = dpp_f2i(v);
    // row_shr:1..3 of the input: lane i of a 16-lane row holds v[i-3 .. i] (lanes shifted in from outside the row: 0)
    float s = v + dpp_i2f(__builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, x, 0x113, 0xf, 0xf, true));
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(s), 0x114, 0xf, 0xe, true));    // row_shr:4 into lanes 4..15
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(s), 0x118, 0xf, 0xc, true));    // row_shr:8 into lanes 8..15
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(s), 0x142, 0xa, 0xf, true));    // row_bcast:15 into rows 1, 3
    s += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(s), 0x143, 0xc, 0xf, true));    // row_bcast:31 into rows 2, 3
    return dpp_i2f(__builtin_amdgcn_readlane(dpp_f2i(s), 63));
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(long long n, const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            float* __restrict__ y, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out) {
    constexpr int H = 256 * NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gm[NV], bt[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { gm[v] = ld4(gamma + 256 * v + 4 * lane); bt[v] = ld4(beta + 256 * v + 4 * lane); }
    for (long long r = (long long)blockIdx.x * 4 + wave; r < n; r += (long long)gridDim.x * 4) {
        float4 xv[NV];
        float s = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            xv[v] = ld4(x + r * H + 256 * v + 4 * lane);
            s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
        }
        const float mean = wave_sum(s) * (1.0f / H);
        float q = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float a = xv[v].x - mean, b = xv[v].y - mean, c = xv[v].z - mean, d = xv[v].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + eps);
#pragma unroll
        for (int v = 0; v < NV; ++v)
            st4(y + r * H + 256 * v + 4 * lane,
                make_float4((xv[v].x - mean) * rstd * gm[v].x + bt[v].x, (xv[v].y - mean) * rstd * gm[v].y + bt[v].y,
                            (xv[v].z - mean) * rstd * gm[v].z + bt[v].z, (xv[v].w - mean) * rstd * gm[v].w + bt[v].w));
        if (lane == 0 && mean_out) { mean_out[r] = mean; rstd_out[r] = rstd; }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  partial[block] = {sum dy * xhat | sum dy}
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(long long n, const float* __restrict__ dy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            float* __restrict__ partial) {
    constexpr int H = 256 * NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gm[NV];
    float dgm[NV][4], dbt[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        gm[v] = ld4(gamma + 256 * v + 4 * lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) { dgm[v][u] = 0.0f; dbt[v][u] = 0.0f; }
    }
    for (long long r = (long long)blockIdx.x * 4 + wave; r < n; r += (long long)gridDim.x * 4) {
        const float mean = mean_in[r], rstd = rstd_in[r];
        float xh[NV][4], g[NV][4];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 xv = ld4(x + r * H + 256 * v + 4 * lane), dv = ld4(dy + r * H + 256 * v + 4 * lane);
            const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, da[4] = {dv.x, dv.y, dv.z, dv.w};
            const float ga[4] = {gm[v].x, gm[v].y, gm[v].z, gm[v].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xh[v][u] = (xa[u] - mean) * rstd;
                g[v][u] = da[u] * ga[u];
                s1 += g[v][u];
                s2 += g[v][u] * xh[v][u];
                dgm[v][u] += da[u] * xh[v][u];
                dbt[v][u] += da[u];
            }
        }
        const float m1 = wave_sum(s1) * (1.0f / H), m2 = wave_sum(s2) * (1.0f / H);
#pragma unroll
        for (int v = 0; v < NV; ++v)
            st4(dx + r * H + 256 * v + 4 * lane,
                make_float4(rstd * (g[v][0] - m1 - xh[v][0] * m2), rstd * (g[v][1] - m1 - xh[v][1] * m2),
                            rstd * (g[v][2] - m1 - xh[v][2] * m2), rstd * (g[v][3] - m1 - xh[v][3] * m2)));
    }
    // the block's 4 waves own the same columns: fold through LDS, then one [2H] row per block
    __shared__ float red[4][2 * H];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            red[wave][256 * v + 4 * lane + u] = dgm[v][u];
            red[wave][H + 256 * v + 4 * lane + u] = dbt[v][u];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256)
        partial[(long long)blockIdx.x * 2 * H + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// RunningMeanStd in eval mode (rl_games: float64 statistics): out = clamp((x - mean) / sqrt(var + eps), +-clip), written
// into a column block of a wider buffer (rows out_stride apart), fp32 or bfloat16.  One thread per element.
template <typename OT>
__global__ void normalize_obs_kernel(long long n, int F, const float* __restrict__ x, const double* __restrict__ mean,
                                     const double* __restrict__ var, float eps, float clip, OT* __restrict__ out,
                                     long long out_stride) {
    const long long total = n * F;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / F;
        const int c = (int)(idx - r * F);
        // same arithmetic as the module: statistics cast to float first, then (x - mean) / sqrt(var + eps)
        const float m = (float)mean[c], sd = sqrtf((float)var[c] + eps);
        float y = (x[idx] - m) / sd;
        y = fminf(fmaxf(y, -clip), clip);
        if (sizeof(OT) == 2) reinterpret_cast<lp16_t*>(out)[r * out_stride + c] = f2lp(y);
        else reinterpret_cast<float*>(out)[r * out_stride + c] = y;
    }
}

// GAE in rl_games' next-nonterminal form over a [T, N] rollout, one env per lane, reverse scan in registers:
//   delta_t = r_t + gamma V_{t+1} nt_{t+1} - V_t;  A_t = delta_t + gamma tau nt_{t+1} A_{t+1};  R_t = A_t + V_t
// (V_T = last_values, nt_T = 1 - dones after the last step; nt_{t+1} = 1 - dones stored at step t+1).
__global__ void gae_kernel(int T, long long N, const float* __restrict__ rewards, const float* __restrict__ values,
                           const unsigned char* __restrict__ dones, const float* __restrict__ last_values,
                           const unsigned char* __restrict__ last_dones, float gamma, float tau,
                           float* __restrict__ advs, float* __restrict__ returns) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float next_v = last_values[e], nnt = 1.0f - (float)last_dones[e], lam = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const float v = values[(long long)t * N + e];
        const float delta = rewards[(long long)t * N + e] + gamma * next_v * nnt - v;
        lam = delta + gamma * tau * nnt * lam;
        advs[(long long)t * N + e] = lam;
        if (returns) returns[(long long)t * N + e] = lam + v;
        next_v = v;
        nnt = 1.0f - (float)dones[(long long)t * N + e];
    }
}

// ---- The rollout buffers -> the PPO dataset, in three launches (round 4; formerly ~40: GAE, eight transposing copies,
// two RunningMeanStd updates + normalisations, the advantage mean / std reductions and their elementwise arithmetic):
//   ds_gae_partial_kernel : GAE as gae_kernel (one env per lane, reverse scan in registers), its outputs written in DATASET
//       order (sample e T + t; 16-B stores) -- raw values, returns = A + V, advantages = returns - V (the stock
//       composition's two roundings) -- plus per-workgroup sums of x and x^2 of the three series in double
//   ds_finalize_kernel    : one wave folds the partial rows in a fixed order; RunningMeanStd (training mode, Chan merge,
//       unbiased batch variance) updated with the values, then with the returns, as value_mean_std(values) followed by
//       value_mean_std(returns) does; advantage mean and unbiased standard deviation; six floats for the last kernel
//   ds_assemble_kernel    : per 64 envs: the three series normalised in place ((x - mean) / sqrt(var + eps) clamped to
//       +-5 with the statistics AFTER their own update; (A - mean) / (std + 1e-8)), and every other rollout buffer
//       [T, N, W] transposed into dataset order [N T, W] through LDS (coalesced on both sides)
// No atomics, no memsets, fixed summation order: captured and replayed like every other kernel of the iteration.
#define DS_MAX_JOBS 8
struct DsJobs { int n; const void* src[DS_MAX_JOBS]; void* dst[DS_MAX_JOBS]; int width[DS_MAX_JOBS]; int elem[DS_MAX_JOBS]; };

__global__ __launch_bounds__(256) void ds_gae_partial_kernel(
    int T, long long N, const float* __restrict__ rewards, const float* __restrict__ values,
    const unsigned char* __restrict__ dones, const float* __restrict__ last_values,
    const unsigned char* __restrict__ last_dones, float gamma, float tau, float* __restrict__ ds_values,
    float* __restrict__ ds_returns, float* __restrict__ ds_adv, double* __restrict__ partial) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // sum / sum of squares of values, returns, advantages
    if (e < N) {
        float next_v = last_values[e], nnt = 1.0f - (float)last_dones[e], lam = 0.0f;
        const long long o = e * T;
#define DS_GAE_STEP(t_, V_, R_, A_)                                                  \
        {                                                                            \
            const float v = values[(long long)(t_) * N + e];                         \
            const float delta = rewards[(long long)(t_) * N + e] + gamma * next_v * nnt - v; \
            lam = delta + gamma * tau * nnt * lam;                                   \
            const float ret = lam + v, adv = ret - v;                                \
            V_ = v; R_ = ret; A_ = adv;                                              \
            acc[0] += (double)v; acc[1] += (double)v * (double)v;                    \
            acc[2] += (double)ret; acc[3] += (double)ret * (double)ret;              \
            acc[4] += (double)adv; acc[5] += (double)adv * (double)adv;              \
            next_v = v;                                                              \
            nnt = 1.0f - (float)dones[(long long)(t_) * N + e];                      \
        }
        if ((T & 3) == 0) {
            for (int t4 = T - 4; t4 >= 0; t4 -= 4) {
                float vv[4], rr[4], aa[4];
#pragma unroll
                for (int k = 3; k >= 0; --k) DS_GAE_STEP(t4 + k, vv[k], rr[k], aa[k])
                st4(ds_values + o + t4, make_float4(vv[0], vv[1], vv[2], vv[3]));
                st4(ds_returns + o + t4, make_float4(rr[0], rr[1], rr[2], rr[3]));
                st4(ds_adv + o + t4, make_float4(aa[0], aa[1], aa[2], aa[3]));
            }
        } else {
            for (int t = T - 1; t >= 0; --t) {
                float v1, r1, a1;
                DS_GAE_STEP(t, v1, r1, a1)
                ds_values[o + t] = v1; ds_returns[o + t] = r1; ds_adv[o + t] = a1;
            }
        }
#undef DS_GAE_STEP
    }
    __shared__ double red[6][256];
#pragma unroll
    for (int j = 0; j < 6; ++j) red[j][threadIdx.x] = acc[j];
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {                  // fixed-order tree
        if ((int)threadIdx.x < h)
#pragma unroll
            for (int j = 0; j < 6; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x < 6) partial[(long long)blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(64) void ds_finalize_kernel(int blocks, long long n, const double* __restrict__ partial,
                                                         const double* __restrict__ rmean,
                                                         const double* __restrict__ rvar,
                                                         const double* __restrict__ rcount, double* __restrict__ stats_out,
                                                         float eps, int normalize_value, float* __restrict__ scal) {
    // fold the partial rows: lane l sums rows l, l + 64, ... (independent loads: the serial form -- six lanes walking the
    // rows one dependent load at a time -- took 11.5 us for 64 rows), then the 64 lane sums in a fixed order
    __shared__ double lane_sum[64][6];
    __shared__ double tot[6];
    {
        double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int b = threadIdx.x; b < blocks; b += 64)
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] += partial[(long long)b * 6 + j];
#pragma unroll
        for (int j = 0; j < 6; ++j) lane_sum[threadIdx.x][j] = a[j];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double s = 0.0;
        for (int l = 0; l < 64; ++l) s += lane_sum[l][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double nb = (double)n;
    if (normalize_value) {
        double mean = rmean[0], var = rvar[0], cnt = rcount[0];
        for (int k = 0; k < 2; ++k) {                     // value_mean_std(values), then value_mean_std(returns)
            const double bmean = tot[2 * k] / nb;
            double bvar = n > 1 ? (tot[2 * k + 1] - nb * bmean * bmean) / (nb - 1.0) : 0.0;
            bvar = bvar > 0.0 ? bvar : 0.0;               // (sum of squares form: a constant series can round below zero; x.var() cannot)
            const double delta = bmean - mean, t = cnt + nb;
            const double m2 = var * cnt + bvar * nb + delta * delta * cnt * nb / t;
            mean += delta * nb / t;
            var = m2 / t;
            cnt = t;
            scal[2 * k] = (float)mean;
            scal[2 * k + 1] = sqrtf((float)var + eps);
        }
        stats_out[0] = mean; stats_out[1] = var; stats_out[2] = cnt;
    }
    const double amean = tot[4] / nb;
    const double avar = n > 1 ? (tot[5] - nb * amean * amean) / (nb - 1.0) : 0.0;
    scal[4] = (float)amean;
    scal[5] = (float)sqrt(avar > 0.0 ? avar : 0.0);
}

#define DS_LDS_WORDS 8192
#define DS_ENVS 16
__global__ __launch_bounds__(256) void ds_assemble_kernel(int T, long long N, const float* __restrict__ scal,
                                                          int normalize_value, int normalize_advantage,
                                                          float* __restrict__ ds_values, float* __restrict__ ds_returns,
                                                          float* __restrict__ ds_adv, const DsJobs J) {
    // DS_ENVS envs per workgroup (16: 1024 workgroups at 16384 envs = four per CU; with 64 envs and one workgroup per CU
    // the kernel took 57 us for its 2 x 33 MB: too few loads in flight)
    __shared__ unsigned int lds[DS_LDS_WORDS];
    const long long e0 = (long long)blockIdx.x * DS_ENVS;
    const int tid = threadIdx.x;
    {
        const float m1 = scal[0], sd1 = scal[1], m2 = scal[2], sd2 = scal[3], am = scal[4], as_ = scal[5] + 1e-8f;
        const long long base = e0 * T;
        for (int i = tid; i < DS_ENVS * T; i += 256) {
            if (normalize_value) {
                float y = (ds_values[base + i] - m1) / sd1;
                ds_values[base + i] = fminf(fmaxf(y, -5.0f), 5.0f);
                y = (ds_returns[base + i] - m2) / sd2;
                ds_returns[base + i] = fminf(fmaxf(y, -5.0f), 5.0f);
            }
            if (normalize_advantage) ds_adv[base + i] = (ds_adv[base + i] - am) / as_;
        }
    }
#pragma unroll 1
    for (int j = 0; j < J.n; ++j) {
        const int W = J.width[j];
        if (J.elem[j] == 4) {
            const unsigned int* src = reinterpret_cast<const unsigned int*>(J.src[j]);
            unsigned int* dst = reinterpret_cast<unsigned int*>(J.dst[j]);
            int EB = DS_ENVS;
            while (EB > 1 && EB * T * W > DS_LDS_WORDS) EB >>= 1;
            const int seg = EB * W, tw = T * W, total = T * seg;
            for (int sub = 0; sub < DS_ENVS; sub += EB) {
                const long long es = e0 + sub;
                __syncthreads();                          // (the previous sub-block's / job's readers are done)
#pragma unroll 4
                for (int i = tid; i < total; i += 256) {
                    const int t = i / seg, r = i - t * seg;
                    lds[i] = src[((long long)t * N + es) * W + r];
                }
                __syncthreads();
#pragma unroll 4
                for (int o = tid; o < total; o += 256) {
                    const int el = o / tw, rem = o - el * tw, t = rem / W, c = rem - t * W;
                    dst[es * tw + o] = lds[t * seg + el * W + c];
                }
            }
        } else {                                          // one byte per env and step (done flags)
            const unsigned int* src = reinterpret_cast<const unsigned int*>(J.src[j]);
            unsigned int* dst = reinterpret_cast<unsigned int*>(J.dst[j]);
            const unsigned char* lb = reinterpret_cast<const unsigned char*>(lds);
            constexpr int WPT = DS_ENVS / 4;                // words per step: DS_ENVS flags
            __syncthreads();
            for (int i = tid; i < WPT * T; i += 256) {
                const int t = i / WPT, r = i - t * WPT;
                lds[i] = src[((long long)t * N + e0) / 4 + r];
            }
            __syncthreads();
            for (int o = tid; o < WPT * T; o += 256) {
                unsigned int wv = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int b = 4 * o + k, el = b / T, t = b - el * T;
                    wv |= (unsigned int)lb[t * DS_ENVS + el] << (8 * k);
                }
                dst[(e0 * T) / 4 + o] = wv;
            }
        }
    }
}

// RunningMeanStd in training mode (rl_games: float64 statistics, Chan et al. merge of the batch moments), two launches:
//   partial : per-workgroup column sums of x and x^2 in double (64 column lanes x 4 row lanes)
//   finalize: batch mean / unbiased variance from the partials (fixed order), merged into the running moments
__global__ __launch_bounds__(256) void rms_partial_kernel(long long n, int F, const float* __restrict__ x,
                                                          double* __restrict__ partial) {
    // (vine_rms_update_multi: blockIdx.y = the batch, n rows each, its own rows of `partial`)
    x += (long long)blockIdx.y * n * F;
    partial += (long long)blockIdx.y * gridDim.x * 2 * F;
    // 256 / F row lanes x F columns (252 of 256 lanes busy at F = 28); consecutive lanes read consecutive floats
    const int rlanes = 256 / F;
    const int rl = threadIdx.x / F, c = threadIdx.x - rl * F;
    double s = 0.0, ss = 0.0;
    if (rl < rlanes) {
        // 8 rows in flight per thread (a rolled loop pays one memory round trip per row); same summation order
        const long long step = (long long)gridDim.x * rlanes;
        long long r = (long long)blockIdx.x * rlanes + rl;
        for (; r + 7 * step < n; r += 8 * step) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = x[(r + k * step) * F + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double d = (double)v[k];
                s += d;
                ss += d * d;
            }
        }
        for (; r < n; r += step) {
            const double v = (double)x[r * F + c];
            s += v;
            ss += v * v;
        }
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = rl < rlanes ? s : 0.0;
    red[1][threadIdx.x] = rl < rlanes ? ss : 0.0;
    __syncthreads();
    if ((int)threadIdx.x < F) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < rlanes; ++k) { a += red[0][k * F + threadIdx.x]; b += red[1][k * F + threadIdx.x]; }
        partial[(long long)blockIdx.x * 2 * F + threadIdx.x] = a;
        partial[(long long)blockIdx.x * 2 * F + F + threadIdx.x] = b;
    }
}

// the per-workgroup rows of rms_partial_kernel -> the batch's column sums of x and x^2 (valid in the threads rl == 0, c < F):
// 64 column lanes x 4 row lanes over the partial rows, folded through LDS in a fixed order
__device__ __forceinline__ void rms_fold(int blocks, int F, const double* __restrict__ partial, double& s, double& ss) {
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    s = 0.0; ss = 0.0;
    if (c < F) {
        int b = rl;
        for (; b + 28 < blocks; b += 32) {                  // 8 partial rows in flight per thread, same order
            double v[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = partial[(long long)(b + 4 * k) * 2 * F + c];
                w[k] = partial[(long long)(b + 4 * k) * 2 * F + F + c];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += v[k]; ss += w[k]; }
        }
        for (; b < blocks; b += 4) {
            s += partial[(long long)b * 2 * F + c];
            ss += partial[(long long)b * 2 * F + F + c];
        }
    }
    __shared__ double red[2][4][64];
    red[0][rl][c] = s;
    red[1][rl][c] = ss;
    __syncthreads();
    if (rl == 0 && c < F) {
        s = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        ss = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}
// Chan et al. merge of a batch's column sums into the running moments of column c (count: before the merge).
// (not inlined: ONE body for the single and the k-batch kernels, so that contraction into fmas cannot differ between them)
__device__ __attribute__((noinline)) void rms_merge(double s, double ss, long long n, double cnt, double& mean, double& var) {
    const double nb = (double)n;
    const double bmean = s / nb;
    const double bvar = n > 1 ? (ss - nb * bmean * bmean) / (nb - 1.0) : 0.0;      // unbiased, like x.var(0)
    const double delta = bmean - mean, tot = cnt + nb;
    const double m2 = var * cnt + bvar * nb + delta * delta * cnt * nb / tot;
    mean += delta * nb / tot;
    var = m2 / tot;
}

__global__ __launch_bounds__(256) void rms_finalize_kernel(int blocks, int F, long long n,
                                                           const double* __restrict__ partial,
                                                           double* __restrict__ running_mean,
                                                           double* __restrict__ running_var, double* __restrict__ count) {
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    double s, ss;
    rms_fold(blocks, F, partial, s, ss);
    const double cnt = count[0];
    if (rl == 0 && c < F) {
        double mean = running_mean[c], var = running_var[c];
        rms_merge(s, ss, n, cnt, mean, var);
        running_mean[c] = mean;
        running_var[c] = var;
    }
    __syncthreads();
    if (threadIdx.x == 0) count[0] = cnt + (double)n;
}

// vine_rms_update_multi: workgroup b folds the partial rows of batch b into sums[b][2F] ...
__global__ __launch_bounds__(256) void rms_fold_multi_kernel(int blocks, int F, const double* __restrict__ partial,
                                                             double* __restrict__ sums) {
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    double s, ss;
    rms_fold(blocks, F, partial + (long long)blockIdx.x * blocks * 2 * F, s, ss);
    if (rl == 0 && c < F) {
        sums[(long long)blockIdx.x * 2 * F + c] = s;
        sums[(long long)blockIdx.x * 2 * F + F + c] = ss;
    }
}
// ... and one wave merges the k batches IN ORDER into the running moments (the same arithmetic, batch after batch, as k
// calls of vine_rms_update), leaving the moments after batch b in snap_mean / snap_var [b][F]
__global__ __launch_bounds__(64) void rms_merge_multi_kernel(int k, int F, long long n, const double* __restrict__ sums,
                                                             double* __restrict__ running_mean, double* __restrict__ running_var,
                                                             double* __restrict__ count, double* __restrict__ snap_mean,
                                                             double* __restrict__ snap_var) {
    const int c = threadIdx.x;
    double cnt = count[0];
    if (c < F) {
        double mean = running_mean[c], var = running_var[c];
        for (int b = 0; b < k; ++b) {
            rms_merge(sums[(long long)b * 2 * F + c], sums[(long long)b * 2 * F + F + c], n, cnt, mean, var);
            cnt += (double)n;
            snap_mean[(long long)b * F + c] = mean;
            snap_var[(long long)b * F + c] = var;
        }
        running_mean[c] = mean;
        running_var[c] = var;
    }
    __syncthreads();
    if (c == 0) count[0] = count[0] + (double)k * (double)n;
}

// out = elu(z + bias): the activation of a Linear whose GEMM ran without an epilogue (bf16 operands, fp32 output)
template <typename OT>
__global__ __launch_bounds__(256) void bias_elu_kernel(long long n, int C, const float* __restrict__ z,
                                                       const float* __restrict__ bias, float alpha,
                                                       OT* __restrict__ out, long long out_stride) {
    const int C4 = C >> 2;
    const long long total = n * C4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / C4;
        const int j = (int)(idx - r * C4) << 2;
        const float4 zv = ld4(z + r * C + j), bv = ld4(bias + j);
        const float x[4] = {zv.x + bv.x, zv.y + bv.y, zv.z + bv.z, zv.w + bv.w};
        float y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = x[u] > 0.0f ? x[u] : alpha * (__expf(x[u]) - 1.0f);
        st4(out + r * out_stride + j, make_float4(y[0], y[1], y[2], y[3]));
    }
}

// ---- LayerNorm + output heads in one pass (H = 256: one float4 per lane; NH = actions + 1 head rows) ----
// forward: heads[r] = W (LN(x_r)) + b without materialising LN(x): the heads are 3 GEMV rows, three GEMMs over
// [n, 256] operands cost more in HBM traffic than the arithmetic is worth.
template <int NH>
__global__ __launch_bounds__(256) void ln_heads_fwd_kernel(long long n, const float* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           const float* __restrict__ w, const float* __restrict__ wb,
                                                           float* __restrict__ heads, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out) {
    constexpr int H = 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 gm = ld4(gamma + 4 * lane), bt = ld4(beta + 4 * lane);
    float4 wv[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) wv[h] = ld4(w + h * H + 4 * lane);
    for (long long r = (long long)blockIdx.x * 4 + wave; r < n; r += (long long)gridDim.x * 4) {
        const float4 xv = ld4(x + r * H + 4 * lane);
        const float mean = wave_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / H);
        const float a = xv.x - mean, b = xv.y - mean, c = xv.z - mean, d = xv.w - mean;
        const float rstd = rsqrtf(wave_sum((a * a + b * b) + (c * c + d * d)) * (1.0f / H) + eps);
        const float y0 = a * rstd * gm.x + bt.x, y1 = b * rstd * gm.y + bt.y, y2 = c * rstd * gm.z + bt.z,
                    y3 = d * rstd * gm.w + bt.w;
        float p[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) p[h] = wave_sum((y0 * wv[h].x + y1 * wv[h].y) + (y2 * wv[h].z + y3 * wv[h].w));
        if (lane == 0) {
#pragma unroll
            for (int h = 0; h < NH; ++h) heads[r * NH + h] = p[h] + wb[h];
            mean_out[r] = mean;
            rstd_out[r] = rstd;
        }
    }
}

// backward: dy = g W (never stored), LayerNorm backward, and per-workgroup partial sums
//   partial[block] = { d gamma [H] | d beta [H] | d W [NH, H] }   with  d W[h] = sum_r g[r, h] * LN(x_r)
template <int NH>
__global__ __launch_bounds__(256) void ln_heads_bwd_kernel(long long n, const float* __restrict__ g,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ w, float* __restrict__ dx,
                                                           float* __restrict__ partial) {
    constexpr int H = 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 gm4 = ld4(gamma + 4 * lane), bt4 = ld4(beta + 4 * lane);
    const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w}, bt[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
    float wv[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float4 t = ld4(w + h * H + 4 * lane);
        wv[h][0] = t.x; wv[h][1] = t.y; wv[h][2] = t.z; wv[h][3] = t.w;
    }
    float dgm[4] = {0, 0, 0, 0}, dbt[4] = {0, 0, 0, 0}, dw[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int u = 0; u < 4; ++u) dw[h][u] = 0.0f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < n; r += (long long)gridDim.x * 4) {
        const float mean = mean_in[r], rstd = rstd_in[r];
        float gh[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) gh[h] = g[r * NH + h];
        const float4 xv = ld4(x + r * H + 4 * lane);
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w};
        float xh[4], gg[4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xh[u] = (xa[u] - mean) * rstd;
            float dy = 0.0f;
#pragma unroll
            for (int h = 0; h < NH; ++h) dy += gh[h] * wv[h][u];
            const float y = xh[u] * gm[u] + bt[u];
#pragma unroll
            for (int h = 0; h < NH; ++h) dw[h][u] += gh[h] * y;
            gg[u] = dy * gm[u];
            s1 += gg[u];
            s2 += gg[u] * xh[u];
            dgm[u] += dy * xh[u];
            dbt[u] += dy;
        }
        const float m1 = wave_sum(s1) * (1.0f / H), m2 = wave_sum(s2) * (1.0f / H);
        st4(dx + r * H + 4 * lane, make_float4(rstd * (gg[0] - m1 - xh[0] * m2), rstd * (gg[1] - m1 - xh[1] * m2),
                                               rstd * (gg[2] - m1 - xh[2] * m2), rstd * (gg[3] - m1 - xh[3] * m2)));
    }
    constexpr int W = (2 + NH) * H;
    __shared__ float red[4][W];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        red[wave][4 * lane + u] = dgm[u];
        red[wave][H + 4 * lane + u] = dbt[u];
#pragma unroll
        for (int h = 0; h < NH; ++h) red[wave][(2 + h) * H + 4 * lane + u] = dw[h][u];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < W; c += 256)
        partial[(long long)blockIdx.x * W + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ELU backward from the OUTPUT a = elu(z): dz = g * (a > 0 ? 1 : a + alpha), plus per-block column sums of dz
// (= the bias gradient of the Linear that produced z).  C4 = C/4 threads per row, 256/C4 rows per block pass.
template <typename AT, typename OT>
__global__ __launch_bounds__(256) void elu_bwd_kernel(long long n, int C, const float* __restrict__ g, long long g_stride,
                                                      const AT* __restrict__ a, long long a_stride, float alpha,
                                                      OT* __restrict__ out, long long out_stride,
                                                      float* __restrict__ partial) {
    const int C4 = C >> 2;
    const int rows_per_pass = 256 / C4;
    const int rq = threadIdx.x / C4, j = (threadIdx.x - rq * C4) << 2;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (long long r = (long long)blockIdx.x * rows_per_pass + rq; r < n; r += (long long)gridDim.x * rows_per_pass) {
        const float4 gv = ld4(g + r * g_stride + j), av = ld4(a + r * a_stride + j);
        const float4 d = make_float4(gv.x * (av.x > 0.0f ? 1.0f : av.x + alpha), gv.y * (av.y > 0.0f ? 1.0f : av.y + alpha),
                                     gv.z * (av.z > 0.0f ? 1.0f : av.z + alpha), gv.w * (av.w > 0.0f ? 1.0f : av.w + alpha));
        st4(out + r * out_stride + j, d);
        acc[0] += d.x; acc[1] += d.y; acc[2] += d.z; acc[3] += d.w;
    }
    if (partial) {
        __shared__ float red[4 * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) red[u * 256 + threadIdx.x] = acc[u];
        __syncthreads();
        if ((int)threadIdx.x < C4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float sum = 0.0f;
                for (int t = threadIdx.x; t < 256; t += C4) sum += red[u * 256 + t];
                partial[(long long)blockIdx.x * C + ((int)threadIdx.x << 2) + u] = sum;
            }
        }
    }
}

// Column sums of src [R, C] (rows row_stride apart): 64 columns x 4 row-lanes per workgroup, no atomics, no
// zero-initialised scratch (ATen's multi-block reductions clear a semaphore buffer with a memset, and memset nodes did
// not replay reliably inside hipGraphs).  Columns [0, n0) go to out0, columns [n0, C) to out1 (out1 == NULL: all to
// out0); dup != 0 writes all C columns to both.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ src, long long R, long long C,
                                                     long long row_stride, float* __restrict__ out0, long long n0,
                                                     float* __restrict__ out1, int dup) {
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const long long c = (long long)blockIdx.x * 64 + cl;
    float acc = 0.0f;
    if (c < C)
        for (long long r = rl; r < R; r += 4) acc += src[r * row_stride + c];
    __shared__ float red[4][64];
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C) {
        const float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (dup) { out0[c] = v; out1[c] = v; }
        else if (out1 && c >= n0) out1[c - n0] = v;
        else out0[c] = v;
    }
}

// Tall-and-narrow variant (the [512..2048, <= 1024] partial-sum blocks): 16 columns x 64 row-lanes per workgroup so
// that a few hundred columns still spread over enough CUs; same output conventions, same determinism.
__global__ __launch_bounds__(1024) void colsum_tall_kernel(const float* __restrict__ src, long long R, long long C,
                                                           long long row_stride, float* __restrict__ out0,
                                                           long long n0, float* __restrict__ out1, int dup) {
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const long long c = (long long)blockIdx.x * 16 + cl;
    float acc = 0.0f;
    if (c < C) {
        long long r = rl;
        for (; r + 7 * 64 < R; r += 8 * 64) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(r + 64LL * k) * row_stride + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; r < R; r += 64) acc += src[r * row_stride + c];
    }
    __shared__ float red[64][17];
    red[rl][cl] = acc;
    __syncthreads();
    if (rl < 8) {                                 // 8 partial sums per column, fixed order
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[rl * 8 + k][cl];
        red[rl * 8][cl] = v;
    }
    __syncthreads();
    if (rl == 0 && c < C) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k * 8][cl];
        if (dup) { out0[c] = v; out1[c] = v; }
        else if (out1 && c >= n0) out1[c - n0] = v;
        else out0[c] = v;
    }
}

// Several column-sum jobs in ONE launch (the backward pass of the network ends with ~13 of them, each too small to
// fill the chip and each paying a launch): workgroup -> job through the block-offset table, then the same two
// geometries (64 columns x 4 row-lanes for short matrices, 16 x 16 for tall ones), 256 threads.
struct ColsumJob {
    const float* src;
    float* out0;
    float* out1;
    long long R, C, row_stride, n0;
    int dup, first_block, quad;
};
#define VINE_COLSUM_MAX_JOBS 16
struct ColsumBatch {
    ColsumJob job[VINE_COLSUM_MAX_JOBS];
    int njobs;
    float* found_inf;      // optional: set to 1 when a finished column sum is not finite (loss-scaled fp16 backward: an
                           // overflowed 16-bit gradient ends as inf / NaN in the weight gradients these jobs finish)
    VineLossFinalize fin;  // fin.partial != NULL: one more workgroup folds the loss kernel's per-workgroup rows
};
__device__ __forceinline__ bool not_finite(float v) { return !(fabsf(v) <= 3.0e38f); }
__device__ __forceinline__ void ppo_loss_finalize(int blocks, int A, long long n, const float* partial,
                                                  const float* __restrict__ logstd, float critic_coef, float entropy_coef,
                                                  float bounds_coef, float* __restrict__ stats,
                                                  float* __restrict__ grad_logstd, float* __restrict__ grad_mu_bias,
                                                  float* __restrict__ grad_value_bias, float* __restrict__ kl_out,
                                                  float* __restrict__ logstd_grad_accum, float S);
__global__ __launch_bounds__(256) void colsum_batched_kernel(ColsumBatch batch) {
    // (workgroup 0: it is the longest job of the launch -- two dependent passes over the rows -- so it starts first)
    const int fin_blocks = batch.fin.partial ? 1 : 0;
    if (fin_blocks && blockIdx.x == 0) {
        // the deferred last step of vine_ln_heads_loss (flag bit 2): loss statistics, KL slot, log-sigma and head-bias
        // gradients from the per-workgroup rows -- here it runs beside the other jobs instead of as the serial tail of
        // the loss kernel (ticket + fence behind 17 MB of stores + two dependent passes over the rows: 6 us)
        const VineLossFinalize& F = batch.fin;
        ppo_loss_finalize(F.blocks, F.A, F.n, F.partial, F.logstd, F.critic_coef, F.entropy_coef, F.bounds_coef, F.stats,
                          F.grad_logstd, F.grad_mu_bias, F.grad_value_bias, F.kl_out, F.logstd_grad_accum,
                          F.loss_scale ? *F.loss_scale : 1.0f);
        return;
    }
    const int bid = (int)blockIdx.x - fin_blocks;
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < batch.njobs; ++k)
        if (bid >= batch.job[k].first_block) j = k;
    const ColsumJob& J = batch.job[j];
    const int blk = bid - J.first_block;
    __shared__ __attribute__((aligned(16))) float red[1024];
    if (J.quad) {
        // short and wide (the slices of a split-K weight gradient): 64 column QUADS x 4 row-lanes, 16-B loads
        const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
        const long long c = ((long long)blk * 64 + cl) * 4;
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (c < J.C) {
#pragma unroll 8
            for (long long r = rl; r < J.R; r += 4) {
                const float4 v = ld4(J.src + r * J.row_stride + c);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        reinterpret_cast<float4*>(red)[rl * 64 + cl] = acc;
        __syncthreads();
        if (rl == 0 && c < J.C) {
            const float4* rq = reinterpret_cast<const float4*>(red);
            const float4 a = rq[cl], b = rq[64 + cl], cc = rq[128 + cl], d = rq[192 + cl];
            const float4 v = make_float4((a.x + cc.x) + (b.x + d.x), (a.y + cc.y) + (b.y + d.y), (a.z + cc.z) + (b.z + d.z),
                                         (a.w + cc.w) + (b.w + d.w));      // the order of the tree below
            if (batch.found_inf && (not_finite(v.x) || not_finite(v.y) || not_finite(v.z) || not_finite(v.w))) *batch.found_inf = 1.0f;
            if (J.dup) { st4(J.out0 + c, v); st4(J.out1 + c, v); }
            else if (J.out1 && c >= J.n0) st4(J.out1 + (c - J.n0), v);
            else st4(J.out0 + c, v);
        }
        return;
    }
    const bool tall = J.R >= 128;
    const int ct = tall ? 16 : 64, rlanes = tall ? 16 : 4;
    const int cl = threadIdx.x % ct, rl = threadIdx.x / ct;
    const long long c = (long long)blk * ct + cl;
    float acc = 0.0f;
    if (c < J.C) {
        // 8 independent loads in flight per thread: a rolled loop would pay one memory round trip per row
        const float* sp = J.src + c;
        long long r = rl;
        for (; r + 7 * rlanes < J.R; r += 8 * rlanes) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = sp[(r + (long long)k * rlanes) * J.row_stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; r < J.R; r += rlanes) acc += sp[r * J.row_stride];
    }
    red[rl * ct + cl] = acc;
    __syncthreads();
    // fixed-order tree over the row lanes (power of two)
    for (int stride = rlanes >> 1; stride > 0; stride >>= 1) {
        if (rl < stride) red[rl * ct + cl] += red[(rl + stride) * ct + cl];
        __syncthreads();
    }
    if (rl == 0 && c < J.C) {
        const float v = red[cl];
        if (batch.found_inf && not_finite(v)) *batch.found_inf = 1.0f;
        if (J.dup) { J.out0[c] = v; J.out1[c] = v; }
        else if (J.out1 && c >= J.n0) J.out1[c - J.n0] = v;
        else J.out0[c] = v;
    }
}

#define PPO_MAX_A 8
#define PPO_LOSS_ROW 32          // floats per workgroup row of the loss kernel's partial sums (22 used)
// rows of `partial` -> stats[8], grad_logstd[A] and (added into) the two head-bias gradients
// (run by the LAST workgroup of ppo_loss_kernel to finish: 256 threads)
__device__ __forceinline__ void ppo_loss_finalize(int blocks, int A, long long n, const float* partial,
                                                  const float* __restrict__ logstd, float critic_coef, float entropy_coef,
                                                  float bounds_coef, float* __restrict__ stats,
                                                  float* __restrict__ grad_logstd, float* __restrict__ grad_mu_bias,
                                                  float* __restrict__ grad_value_bias, float* __restrict__ kl_out,
                                                  float* __restrict__ logstd_grad_accum, float S) {
    // S: loss scale; the partial sums are unscaled, every GRADIENT written here is multiplied by it
    // (called by whole workgroups of 256 or 512 threads: the first 256 do the work, all take part in the barriers)
    const int q = threadIdx.x & (PPO_LOSS_ROW - 1), rl = threadIdx.x / PPO_LOSS_ROW;     // 32 columns x 8 row-lanes
    __shared__ float fred[8][PPO_LOSS_ROW];
    __shared__ float tot[PPO_LOSS_ROW];
    if (threadIdx.x < 256) {
        // this is the serial tail of the launch (one workgroup, every load an L2 round trip): 16 rows in flight per
        // thread, added in the order of one at a time
        float acc = 0.0f;
        int b = rl;
        for (; b + 120 < blocks; b += 128) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = partial[(long long)(b + 8 * k) * PPO_LOSS_ROW + q];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k];
        }
        for (; b + 24 < blocks; b += 32) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = partial[(long long)(b + 8 * k) * PPO_LOSS_ROW + q];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k];
        }
        for (; b < blocks; b += 8) acc += partial[(long long)b * PPO_LOSS_ROW + q];
        fred[rl][q] = acc;
    }
    __syncthreads();
    if (rl == 0) {
        float v = 0.0f;
        for (int k = 0; k < 8; ++k) v += fred[k][q];
        tot[q] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv_n = 1.0f / (float)n;
        float sum_ls = 0.0f;
        for (int k = 0; k < A; ++k) sum_ls += logstd[k];
        const float ent = A * (0.5f + 0.9189385332046727f) + sum_ls;
        const float a = tot[0] * inv_n, c = tot[1] * inv_n, b = tot[2] * inv_n, kl = tot[3] * inv_n;
        stats[0] = a; stats[1] = c; stats[2] = b; stats[3] = ent; stats[4] = kl;
        stats[5] = a + 0.5f * critic_coef * c + bounds_coef * b - entropy_coef * ent;
        stats[6] = 0.0f; stats[7] = 0.0f;
        for (int k = 0; k < A; ++k) {
            const float gl = (tot[5 + k] - entropy_coef) * S;
            grad_logstd[k] = gl;
            if (logstd_grad_accum) logstd_grad_accum[k] += gl;
        }
        if (kl_out) kl_out[0] = kl;
        if (grad_mu_bias) {
            for (int k = 0; k < A; ++k) grad_mu_bias[k] += tot[5 + PPO_MAX_A + k] * S;
            grad_value_bias[0] += tot[5 + 2 * PPO_MAX_A] * S;
        }
    }
}

// Ticket of the loss / Adam kernels' workgroups (the last one to finish runs the finalize step and resets it: a launch
// always finds 0).  The word is NOT a module global: the host wrappers hand every (device, stream) pair its own zeroed
// slot (ticket_slot below), so launches in flight on different streams -- two agents in one process, a warm-up on a side
// stream -- cannot corrupt each other's election; launches on one stream are ordered.

__global__ __launch_bounds__(256) void ppo_loss_kernel(long long n, int A, const float* __restrict__ mu,
                                                       const float* __restrict__ logstd, const float* __restrict__ value,
                                                       const float* __restrict__ actions, const float* __restrict__ old_neglogp,
                                                       const float* __restrict__ adv, const float* __restrict__ old_values,
                                                       const float* __restrict__ returns,
                                                       // old_mu / old_sigma may be the SAME buffers as mu_store /
                                                       // sigma_store (update_old: the dataset refresh in place), so
                                                       // none of the four is __restrict__
                                                       const float* old_mu, const float* old_sigma, float e_clip, int clip_value,
                                                       float critic_coef, float entropy_coef, float bounds_coef,
                                                       float soft_bound, float* __restrict__ grad_mu,
                                                       float* __restrict__ grad_value, float* __restrict__ grad_logstd,
                                                       float* __restrict__ stats, long long mu_stride,
                                                       long long value_stride, float* partial,
                                                       float* mu_store, float* sigma_store,
                                                       float* __restrict__ grad_mu_bias, float* __restrict__ grad_value_bias,
                                                       float* __restrict__ kl_out, float* __restrict__ logstd_grad_accum,
                                                       const float* __restrict__ loss_scale, unsigned int* ticket) {
    // mu / grad_mu rows are mu_stride floats apart, value / grad_value elements value_stride apart (A and 1 when the
    // heads are separate tensors; A+1 when one GEMM produced [mu | value] rows)
    // loss scale (GradScaler restated on the device: every gradient this kernel hands on is multiplied by it; the loss
    // statistics are not)
    const float S = loss_scale ? *loss_scale : 1.0f;
    const float inv_n = 1.0f / (float)n;
    float ls[PPO_MAX_A], sg[PPO_MAX_A], isg2[PPO_MAX_A];
    float sum_ls = 0.0f;
    for (int k = 0; k < A; ++k) {
        ls[k] = logstd[k];
        sg[k] = __expf(ls[k]);
        isg2[k] = 1.0f / (sg[k] * sg[k]);
        sum_ls += ls[k];
    }
    float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};                   // a, c, b, kl, (unused)
    float gls[PPO_MAX_A], gmb[PPO_MAX_A + 1];              // d/d logstd; column sums of the head gradients
    for (int k = 0; k < A; ++k) gls[k] = 0.0f;
    for (int k = 0; k <= PPO_MAX_A; ++k) gmb[k] = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float z2[PPO_MAX_A], dm[PPO_MAX_A], m[PPO_MAX_A];
        float nlp = 0.9189385332046727f * A + sum_ls;
        for (int k = 0; k < A; ++k) {
            m[k] = mu[i * mu_stride + k];
            dm[k] = actions[i * A + k] - m[k];
            z2[k] = dm[k] * dm[k] * isg2[k];
            nlp += 0.5f * z2[k];
        }
        const float a = adv[i];
        const float ratio = __expf(old_neglogp[i] - nlp);
        const float rc = fminf(fmaxf(ratio, 1.0f - e_clip), 1.0f + e_clip);
        const float s1 = -a * ratio, s2 = -a * rc;
        const bool first = s1 >= s2;                     // torch.max sends the tie's gradient to the first operand
        const float a_loss = first ? s1 : s2;
        const float inside = (ratio > 1.0f - e_clip && ratio < 1.0f + e_clip) ? 1.0f : 0.0f;
        const float dL_dratio = first ? -a : -a * inside;
        const float dL_dnlp = -ratio * dL_dratio * inv_n;    // d ratio / d nlp = -ratio
        // value loss
        const float v = value[i * value_stride], vp = old_values[i], R = returns[i];
        float c_loss, dL_dv;
        if (clip_value) {
            const float dv = v - vp;
            const float vc = vp + fminf(fmaxf(dv, -e_clip), e_clip);
            const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
            if (l1 >= l2) { c_loss = l1; dL_dv = 2.0f * (v - R); }
            else { c_loss = l2; dL_dv = (dv > -e_clip && dv < e_clip) ? 2.0f * (vc - R) : 0.0f; }
        } else {
            c_loss = (R - v) * (R - v);
            dL_dv = 2.0f * (v - R);
        }
        const float gval = 0.5f * critic_coef * dL_dv * inv_n;
        grad_value[i * value_stride] = gval * S;
        gmb[PPO_MAX_A] += gval;
        float b_loss = 0.0f, kl = 0.0f;
        for (int k = 0; k < A; ++k) {
            const float hi = fmaxf(m[k] - soft_bound, 0.0f), lo = fminf(m[k] + soft_bound, 0.0f);
            b_loss += hi * hi + lo * lo;
            // d nlp / d mu = -(a - mu)/sigma^2 ; d nlp / d logstd = 1 - z^2
            const float gm = dL_dnlp * (-dm[k] * isg2[k]) + bounds_coef * inv_n * 2.0f * (hi + lo);
            grad_mu[i * mu_stride + k] = gm * S;
            gmb[k] += gm;
            gls[k] += dL_dnlp * (1.0f - z2[k]);
            const float om = old_mu[i * A + k], os = old_sigma[i * A + k];
            const float c1 = __logf(os / sg[k] + 1e-5f);
            const float c2 = (sg[k] * sg[k] + (om - m[k]) * (om - m[k])) / (2.0f * (os * os + 1e-5f));
            kl += c1 + c2 - 0.5f;
            if (mu_store) {      // dataset.update_mu_sigma: may alias old_mu / old_sigma (read above by this thread)
                mu_store[i * A + k] = m[k];
                sigma_store[i * A + k] = sg[k];
            }
        }
        acc[0] += a_loss; acc[1] += c_loss; acc[2] += b_loss; acc[3] += kl;
    }
    // block reduction (wave shuffles, then LDS across the 4 waves) into one row of `partial` per workgroup; the
    // finalize kernel adds the rows in a fixed order: no float atomics, results are bit-reproducible
    constexpr int NRED = 5 + 2 * PPO_MAX_A + 1;
    __shared__ float red[4][NRED];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float vals[NRED];
    for (int q = 0; q < 4; ++q) vals[q] = acc[q];
    vals[4] = 0.0f;
    for (int k = 0; k < PPO_MAX_A; ++k) vals[5 + k] = (k < A) ? gls[k] : 0.0f;
    for (int k = 0; k <= PPO_MAX_A; ++k) vals[5 + PPO_MAX_A + k] = gmb[k];
#pragma unroll
    for (int q = 0; q < NRED; ++q) {
        // slots of actions k >= A hold zeros: skip their reductions (A is uniform, so is the branch)
        const bool live = q < 4 || (q >= 5 && q < 5 + A) || (q >= 5 + PPO_MAX_A && q < 5 + PPO_MAX_A + A) ||
                          q == 5 + 2 * PPO_MAX_A;
        const float x = live ? wave_sum(vals[q]) : 0.0f;
        if (lane == 0) red[wave][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < PPO_LOSS_ROW) {
        const int q = threadIdx.x;
        partial[(long long)blockIdx.x * PPO_LOSS_ROW + q] =
            q < NRED ? (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]) : 0.0f;
        __threadfence();                                    // the row is visible device-wide before the ticket is taken
    }
    // the last workgroup to get here adds the rows in a fixed order (formerly a second 1-workgroup launch)
    __shared__ bool is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(ticket, 1u);
        is_last = t == gridDim.x - 1;
        if (is_last) *ticket = 0;
    }
    __syncthreads();
    if (is_last) {
        __threadfence();
        ppo_loss_finalize((int)gridDim.x, A, n, partial, logstd, critic_coef, entropy_coef, bounds_coef, stats, grad_logstd,
                          grad_mu_bias, grad_value_bias, kl_out, logstd_grad_accum, S);
    }
}

// ---- LayerNorm + heads + PPO loss + their backward in ONE kernel (H = 256): ln_heads_fwd_kernel, ppo_loss_kernel and
// ln_heads_bwd_kernel were three nodes of the update graph (12 + 13 + 15 us) passing [n, NH] head values and gradients
// and re-reading the LSTM output.  The loss is a sum over samples, so its gradient w.r.t. a row's heads depends on that
// row alone: a wave takes 16 rows through all three steps with the rows held in registers.
//   phase 1 (row at a time, 64 lanes x float4): LayerNorm statistics and the NH head dot products; lane rr keeps row rr's
//   phase 2 (lanes 0..15, one row each): the loss terms and d loss / d heads exactly as ppo_loss_kernel computes them,
//            mu / sigma refresh of the dataset, heads written out for the caller;
//   phase 3 (row at a time): d heads -> d LN output -> LayerNorm backward -> dx, with the partial sums of d gamma,
//            d beta, d W accumulated per lane.
// Per-workgroup rows of partial sums (LayerNorm / head parameters: ln_partial; loss statistics: loss_partial) are
// folded by the column-sum kernel and, for the loss, by the last workgroup to finish (ppo_loss_finalize): fixed order,
// no atomics on floats.  Workgroup = 8 waves x 16 rows.
#ifdef SPLIT_TIMING
// (debug build, scripts/ubench/trunk_phases_clock.py) per wave: s_memtime / s_memrealtime at the phase boundaries
#define TRUNK_STAMP(i)                                                                                                  \
    if ((threadIdx.x & 63) == 0) {                                                                                      \
        mlp_split_t[((blockIdx.x * 8 + (threadIdx.x >> 6)) & 4095) * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime();      \
        mlp_split_t[((blockIdx.x * 8 + (threadIdx.x >> 6)) & 4095) * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define TRUNK_STAMP(i)
#endif
// sum over the 16 lanes of a DPP row, result in EVERY lane of the row: four rotate-and-add steps
__device__ __forceinline__ float row_allsum16(float v) {
    v += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(v), 0x128, 0xf, 0xf, true));      // row_ror:8
    v += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(v), 0x124, 0xf, 0xf, true));      // row_ror:4
    v += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(v), 0x122, 0xf, 0xf, true));      // row_ror:2
    v += dpp_i2f(__builtin_amdgcn_update_dpp(0, dpp_f2i(v), 0x121, 0xf, 0xf, true));      // row_ror:1
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
    return dpp_i2f(__builtin_amdgcn_ds_bpermute(src_lane << 2, dpp_f2i(v)));
}

// Lane mapping: SIXTEEN lanes per row (a wave works on 4 rows at a time; lane = 16 sub + cl holds the float4 columns
// cl + 16 j, j = 0..3, of row `sub`), so that a LayerNorm / head reduction is four DPP rotate-adds inside a 16-lane row
// instead of a wave-wide sum (9 DPP steps + a readlane, one row at a time: that version took 39 us).  NP passes of 4
// rows per wave; the rows are re-read in phase 3 (L2 hits) rather than kept in 64 registers.
// XT = lp16_t (the LSTM output as the 16-bit tensor an autocast LSTM hands its LayerNorm): a lane then owns 8
// consecutive columns twice (16-B loads, 16-B dx stores), ALL rows of the wave and the per-sample loss inputs are
// requested before anything is computed and the rows stay in registers (32 per lane) for phase 3 -- the fp32 form spends
// most of its life in nine dependent memory round trips (SQ_WAIT_ANY 64 % of its wave cycles).
template <int NH, int NP, typename DXT, typename XT>      // DXT: type of the gradient handed to the LSTM backward (float or lp16_t)
__device__ __forceinline__ void ln_heads_loss_body(
    float* lds_red, float* lds_lred,      // [8][(2 + NH) * 256] and [8][PPO_LOSS_ROW] floats of LDS (the caller's)
    long long n, const XT* __restrict__ x, int xT, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ w, const float* __restrict__ wb, const float* __restrict__ logstd,
    const float* __restrict__ actions, const float* __restrict__ old_neglogp, const float* __restrict__ adv,
    const float* __restrict__ old_values, const float* __restrict__ returns, const float* old_mu, const float* old_sigma,
    float e_clip, int clip_value, float critic_coef, float entropy_coef, float bounds_coef, float soft_bound,
    float* __restrict__ heads, DXT* __restrict__ dx, float* __restrict__ ln_partial, float* loss_partial,
    float* __restrict__ stats, float* __restrict__ grad_logstd, float* __restrict__ grad_mu_bias,
    float* __restrict__ grad_value_bias, float* __restrict__ kl_out, float* __restrict__ logstd_grad_accum, float* mu_store,
    float* sigma_store, const float* __restrict__ loss_scale, float* __restrict__ found_inf, unsigned int* ticket,
    int defer) {
    constexpr int H = 256, A = NH - 1, NWV = 8, W = (2 + NH) * H, RW = 4 * NP;
    // loss scale: applied where the per-row head gradients are formed, so dx and every parameter partial sum of this
    // kernel carry it; the bias / log-sigma gradients of the finalize step are multiplied there; statistics unscaled
    const float S = loss_scale ? *loss_scale : 1.0f;
    constexpr int NRED = 5 + 2 * PPO_MAX_A + 1;
    float (*red)[W] = reinterpret_cast<float (*)[W]>(lds_red);
    float (*lred)[PPO_LOSS_ROW] = reinterpret_cast<float (*)[PPO_LOSS_ROW]>(lds_lred);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, cl = lane & 15;
    const long long r0 = ((long long)blockIdx.x * NWV + wave) * RW;
    constexpr bool X16 = sizeof(XT) == 2;
    // first column of the lane's j-th group of 4: fp32 rows in 16-B pieces (columns 4 (cl + 16 j)), 16-bit rows in 16-B
    // pieces of 8 columns (groups 2 jj, 2 jj + 1 = columns 8 (cl + 16 jj) .. + 7)
#define LHL_COL(j) (X16 ? 8 * (cl + 16 * ((j) >> 1)) + 4 * ((j) & 1) : 4 * (cl + 16 * (j)))
    // 16-bit rows: every row of the wave and the per-sample loss inputs are requested here, ahead of the parameters
    uint4 xq[X16 ? NP : 1][2];
    float q_act[A], q_om[A], q_os[A], q_onl = 0.0f, q_adv = 0.0f, q_ov = 0.0f, q_ret = 0.0f;
    if (X16) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // xT > 0: x is the [n / xT, xT + 1, H] tensor of vine_lstm_seq_forward_mfma's "h once" form (slot 0 = h0):
            // sample r = seq * xT + t is its row r + seq + 1
            const long long r_ = r0 + 4 * p + sub;
            // (32-bit quotient: the 64-bit division is a ~100-instruction routine per lane and pass, in front of the loads)
            const XT* xr = x + (xT > 0 ? r_ + (long long)((unsigned)r_ / (unsigned)xT) + 1 : r_) * H;
            xq[p][0] = *reinterpret_cast<const uint4*>(xr + 8 * cl);
            xq[p][1] = *reinterpret_cast<const uint4*>(xr + 8 * (cl + 16));
        }
        if (cl < NP) {
            const long long i = r0 + 4 * cl + sub;
#pragma unroll
            for (int k = 0; k < A; ++k) { q_act[k] = actions[i * A + k]; q_om[k] = old_mu[i * A + k]; q_os[k] = old_sigma[i * A + k]; }
            q_onl = old_neglogp[i]; q_adv = adv[i]; q_ov = old_values[i]; q_ret = returns[i];
        }
    }
    float gm[4][4], bt[4][4], wv[NH][4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 g4 = ld4(gamma + LHL_COL(j)), b4 = ld4(beta + LHL_COL(j));
        gm[j][0] = g4.x; gm[j][1] = g4.y; gm[j][2] = g4.z; gm[j][3] = g4.w;
        bt[j][0] = b4.x; bt[j][1] = b4.y; bt[j][2] = b4.z; bt[j][3] = b4.w;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float4 t = ld4(w + h * H + LHL_COL(j));
            wv[h][j][0] = t.x; wv[h][j][1] = t.y; wv[h][j][2] = t.z; wv[h][j][3] = t.w;
        }
    }
    // the 16 values of a lane's share of a row, from the packed registers (16-bit) or from memory (fp32)
#define LHL_ROW(p, r, dst)                                                                                        \
    if (X16) {                                                                                                    \
        float lo_[8], hi_[8];                                                                                     \
        unpack_lp16x8(xq[X16 ? (p) : 0][0], lo_);                                                                 \
        unpack_lp16x8(xq[X16 ? (p) : 0][1], hi_);                                                                 \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                        \
            dst[0][u_] = lo_[u_]; dst[1][u_] = lo_[4 + u_]; dst[2][u_] = hi_[u_]; dst[3][u_] = hi_[4 + u_];       \
        }                                                                                                         \
    } else {                                                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                        \
            const float4 v_ = ld4(x + (r) * H + 4 * (cl + 16 * j_));                                              \
            dst[j_][0] = v_.x; dst[j_][1] = v_.y; dst[j_][2] = v_.z; dst[j_][3] = v_.w;                           \
        }                                                                                                         \
    }
    // ---- phase 1: LayerNorm statistics and heads of 4 rows per pass; lane (sub, cl = p) keeps row 4 p + sub
    float myp[NH], mymean = 0.0f, myrstd = 0.0f;
#pragma unroll
    for (int h = 0; h < NH; ++h) myp[h] = 0.0f;
    // (fp32 rows come from memory in both passes over them: their pass loops stay rolled -- unrolled, the compiler hoists
    // every pass's 4 x 16-B row loads to the front and the <5, 4, float, float> instantiation spilled 193 VGPRs to scratch,
    // VERDICT r4; the 16-bit form indexes its packed row registers by p and needs the unrolled loop)
    constexpr int PASS_UNROLL = X16 ? NP : 1;
#pragma unroll PASS_UNROLL
    for (int p = 0; p < NP; ++p) {
        float xa[4][4];
        LHL_ROW(p, r0 + 4 * p + sub, xa)
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (xa[j][0] + xa[j][1]) + (xa[j][2] + xa[j][3]);
        const float mean = row_allsum16(s) * (1.0f / H);
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xa[j][u] -= mean;
                q += xa[j][u] * xa[j][u];
            }
        const float rstd = rsqrtf(row_allsum16(q) * (1.0f / H) + eps);
        float ph[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) ph[h] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float y = xa[j][u] * rstd * gm[j][u] + bt[j][u];
#pragma unroll
                for (int h = 0; h < NH; ++h) ph[h] += y * wv[h][j][u];
            }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const float t = row_allsum16(ph[h]) + wb[h];
            if (cl == p) myp[h] = t;
        }
        if (cl == p) { mymean = mean; myrstd = rstd; }
    }
#ifndef BWD_STAMPS
    TRUNK_STAMP(5)
#endif
    // ---- phase 2: lanes with cl < NP hold one row each (row 4 cl + sub): ppo_loss_kernel's arithmetic
    float gh[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) gh[h] = 0.0f;
    float vals[NRED];
#pragma unroll
    for (int qq = 0; qq < NRED; ++qq) vals[qq] = 0.0f;
    if (cl < NP) {
        const long long i = r0 + 4 * cl + sub;
        const float inv_n = 1.0f / (float)n;
        float nlp = 0.9189385332046727f * A;
        float sg[A], isg2[A], z2[A], dm[A];
#pragma unroll
        for (int k = 0; k < A; ++k) {
            const float l = logstd[k];
            sg[k] = __expf(l);
            isg2[k] = 1.0f / (sg[k] * sg[k]);
            nlp += l;
        }
#pragma unroll
        for (int k = 0; k < A; ++k) {
            dm[k] = (X16 ? q_act[k] : actions[i * A + k]) - myp[k];
            z2[k] = dm[k] * dm[k] * isg2[k];
            nlp += 0.5f * z2[k];
        }
        const float a = X16 ? q_adv : adv[i];
        const float ratio = __expf((X16 ? q_onl : old_neglogp[i]) - nlp);
        const float rc = fminf(fmaxf(ratio, 1.0f - e_clip), 1.0f + e_clip);
        const float s1 = -a * ratio, s2 = -a * rc;
        const bool first = s1 >= s2;                     // torch.max sends the tie's gradient to the first operand
        const float a_loss = first ? s1 : s2;
        const float inside = (ratio > 1.0f - e_clip && ratio < 1.0f + e_clip) ? 1.0f : 0.0f;
        const float dL_dratio = first ? -a : -a * inside;
        const float dL_dnlp = -ratio * dL_dratio * inv_n;
        const float v = myp[A], vp = X16 ? q_ov : old_values[i], R = X16 ? q_ret : returns[i];
        float c_loss, dL_dv;
        if (clip_value) {
            const float dv = v - vp;
            const float vc = vp + fminf(fmaxf(dv, -e_clip), e_clip);
            const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
            if (l1 >= l2) { c_loss = l1; dL_dv = 2.0f * (v - R); }
            else { c_loss = l2; dL_dv = (dv > -e_clip && dv < e_clip) ? 2.0f * (vc - R) : 0.0f; }
        } else {
            c_loss = (R - v) * (R - v);
            dL_dv = 2.0f * (v - R);
        }
        const float gval = 0.5f * critic_coef * dL_dv * inv_n;
        gh[A] = gval * S;
        vals[5 + 2 * PPO_MAX_A] = gval;
        float b_loss = 0.0f, kl = 0.0f;
#pragma unroll
        for (int k = 0; k < A; ++k) {
            const float m = myp[k];
            const float hi = fmaxf(m - soft_bound, 0.0f), lo = fminf(m + soft_bound, 0.0f);
            b_loss += hi * hi + lo * lo;
            const float gmk = dL_dnlp * (-dm[k] * isg2[k]) + bounds_coef * inv_n * 2.0f * (hi + lo);
            gh[k] = gmk * S;
            vals[5 + PPO_MAX_A + k] = gmk;
            vals[5 + k] = dL_dnlp * (1.0f - z2[k]);
            const float om = X16 ? q_om[k] : old_mu[i * A + k], os = X16 ? q_os[k] : old_sigma[i * A + k];
            const float c1 = __logf(os / sg[k] + 1e-5f);
            const float c2 = (sg[k] * sg[k] + (om - m) * (om - m)) / (2.0f * (os * os + 1e-5f));
            kl += c1 + c2 - 0.5f;
            if (mu_store) {      // dataset.update_mu_sigma: may alias old_mu / old_sigma (read above by this lane)
                mu_store[i * A + k] = m;
                sigma_store[i * A + k] = sg[k];
            }
        }
        vals[0] = a_loss; vals[1] = c_loss; vals[2] = b_loss; vals[3] = kl;
#pragma unroll
        for (int h = 0; h < NH; ++h) heads[i * NH + h] = myp[h];
    }
    // the wave's rows added (inactive lanes hold zeros), one row of partial sums per wave
#pragma unroll
    for (int qq = 0; qq < NRED; ++qq) {
        const bool live = qq < 4 || (qq >= 5 && qq < 5 + A) || (qq >= 5 + PPO_MAX_A && qq < 5 + PPO_MAX_A + A) ||
                          qq == 5 + 2 * PPO_MAX_A;
        const float t = live ? wave_sum(vals[qq]) : 0.0f;
        if (lane == 0) lred[wave][qq] = t;
    }
    if (lane >= NRED && lane < PPO_LOSS_ROW) lred[wave][lane] = 0.0f;
#ifndef BWD_STAMPS
    TRUNK_STAMP(6)
#endif
    // ---- phase 3: d heads -> d LN output -> LayerNorm backward -> dx; parameter partial sums per lane
    float dgm[4][4], dbt[4][4], dw[NH][4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dgm[j][u] = 0.0f; dbt[j][u] = 0.0f;
#pragma unroll
            for (int h = 0; h < NH; ++h) dw[h][j][u] = 0.0f;
        }
#pragma unroll PASS_UNROLL
    for (int p = 0; p < NP; ++p) {
        const int src = (lane & 48) | p;                        // the lane of this row that holds pass p's values
        float ghr[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) ghr[h] = lane_bcast(gh[h], src);
        const float mean = lane_bcast(mymean, src), rstd = lane_bcast(myrstd, src);
        const long long r = r0 + 4 * p + sub;
        float xh[4][4], gg[4][4], s1 = 0.0f, s2 = 0.0f;
        float xrow[4][4];
        LHL_ROW(p, r, xrow)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xh[j][u] = (xrow[j][u] - mean) * rstd;
                float dy = 0.0f;
#pragma unroll
                for (int h = 0; h < NH; ++h) dy += ghr[h] * wv[h][j][u];
                const float y = xh[j][u] * gm[j][u] + bt[j][u];
#pragma unroll
                for (int h = 0; h < NH; ++h) dw[h][j][u] += ghr[h] * y;
                gg[j][u] = dy * gm[j][u];
                s1 += gg[j][u];
                s2 += gg[j][u] * xh[j][u];
                dgm[j][u] += dy * xh[j][u];
                dbt[j][u] += dy;
            }
        }
        const float m1 = row_allsum16(s1) * (1.0f / H), m2 = row_allsum16(s2) * (1.0f / H);
        bool bad = false;
        float4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = make_float4(rstd * (gg[j][0] - m1 - xh[j][0] * m2), rstd * (gg[j][1] - m1 - xh[j][1] * m2),
                               rstd * (gg[j][2] - m1 - xh[j][2] * m2), rstd * (gg[j][3] - m1 - xh[j][3] * m2));
            bad = bad || !(fabsf(o[j].x) <= LP16_MAX) || !(fabsf(o[j].y) <= LP16_MAX) || !(fabsf(o[j].z) <= LP16_MAX) ||
                  !(fabsf(o[j].w) <= LP16_MAX);
        }
        if (X16 && sizeof(DXT) == 2) {      // 8 consecutive columns per 16-B store
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float v8[8] = {o[2 * jj].x, o[2 * jj].y, o[2 * jj].z, o[2 * jj].w,
                                     o[2 * jj + 1].x, o[2 * jj + 1].y, o[2 * jj + 1].z, o[2 * jj + 1].w};
                *reinterpret_cast<uint4*>(dx + r * H + 8 * (cl + 16 * jj)) = pack_lp16x8(v8);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) st4(dx + r * H + LHL_COL(j), o[j]);
        }
        // a gradient that does not fit the 16-bit format (or is NaN) marks the optimiser step as overflowed: the Adam
        // kernel then skips it and backs the loss scale off, as torch's GradScaler does.  (Plain store of a constant by
        // whoever sees it: no atomic needed.  The column-sum kernel checks the final gradients as well.)
        if (sizeof(DXT) == 2 && found_inf && bad) *found_inf = 1.0f;
    }
#ifndef BWD_STAMPS
    TRUNK_STAMP(7)
#endif
    // the four 16-lane rows of the wave hold sums for the same columns: add them (lane l <- l ^ 16, l ^ 32), then one
    // row of W sums per wave in LDS, added over the waves in wave order
    // (gfx950's v_permlane16_swap / v_permlane32_swap: the partner row's value without a trip through the LDS crossbar --
    // 160 ds_bpermute per lane before)
    // (inline assembly: this compiler's __builtin_amdgcn_permlane{16,32}_swap hands back the first result register for
    // both elements of its result pair, so the partner rows' values are lost; the s_nop covers the VALU-write ->
    // permlane-read wait states the compiler would insert for the builtin)
#define LHL_FOLD(v)                                                                                             \
    {                                                                                                           \
        float lo_ = v, hi_ = v;                                                                                 \
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo_), "+v"(hi_));   /* {even row, odd row} of the pair */ \
        lo_ += hi_;                                                                                             \
        hi_ = lo_;                                                                                              \
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo_), "+v"(hi_));   /* {lower half, upper half} */ \
        v = lo_ + hi_;                                                                                          \
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            LHL_FOLD(dgm[j][u]) LHL_FOLD(dbt[j][u])
#pragma unroll
            for (int h = 0; h < NH; ++h) LHL_FOLD(dw[h][j][u])
        }
#undef LHL_FOLD
    if (sub == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = LHL_COL(j) + u;
                red[wave][c] = dgm[j][u];
                red[wave][H + c] = dbt[j][u];
#pragma unroll
                for (int h = 0; h < NH; ++h) red[wave][(2 + h) * H + c] = dw[h][j][u];
            }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < W; c += 512) {
        float sum = 0.0f;
#pragma unroll
        for (int wq = 0; wq < NWV; ++wq) sum += red[wq][c];
        ln_partial[(long long)blockIdx.x * W + c] = sum;
    }
    if (threadIdx.x < PPO_LOSS_ROW) {
        float sum = 0.0f;
#pragma unroll
        for (int wq = 0; wq < NWV; ++wq) sum += lred[wq][threadIdx.x];
        loss_partial[(long long)blockIdx.x * PPO_LOSS_ROW + threadIdx.x] = sum;
        if (!defer) __threadfence();
    }
    if (defer) return;      // the rows are folded later, by the batched column-sum launch (VineLossFinalize)
    __shared__ bool is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(ticket, 1u);
        is_last = t == gridDim.x - 1;
        if (is_last) *ticket = 0;
    }
    __syncthreads();
    if (is_last) {
        __threadfence();
        ppo_loss_finalize((int)gridDim.x, A, n, loss_partial, logstd, critic_coef, entropy_coef, bounds_coef, stats,
                          grad_logstd, grad_mu_bias, grad_value_bias, kl_out, logstd_grad_accum, S);
    }
}
template <int NH, int NP, typename DXT, typename XT>
__global__ __launch_bounds__(512) void ln_heads_loss_kernel(
    long long n, const XT* __restrict__ x, int xT, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const float* __restrict__ w, const float* __restrict__ wb, const float* __restrict__ logstd,
    const float* __restrict__ actions, const float* __restrict__ old_neglogp, const float* __restrict__ adv,
    const float* __restrict__ old_values, const float* __restrict__ returns, const float* old_mu, const float* old_sigma,
    float e_clip, int clip_value, float critic_coef, float entropy_coef, float bounds_coef, float soft_bound,
    float* __restrict__ heads, DXT* __restrict__ dx, float* __restrict__ ln_partial, float* loss_partial,
    float* __restrict__ stats, float* __restrict__ grad_logstd, float* __restrict__ grad_mu_bias,
    float* __restrict__ grad_value_bias, float* __restrict__ kl_out, float* __restrict__ logstd_grad_accum, float* mu_store,
    float* sigma_store, const float* __restrict__ loss_scale, float* __restrict__ found_inf, unsigned int* ticket,
    int defer) {
    __shared__ float red_s[8 * (2 + NH) * 256];
    __shared__ float lred_s[8 * PPO_LOSS_ROW];
    ln_heads_loss_body<NH, NP, DXT, XT>(red_s, lred_s, n, x, xT, gamma, beta, eps, w, wb, logstd, actions, old_neglogp, adv,
                                        old_values, returns, old_mu, old_sigma, e_clip, clip_value, critic_coef, entropy_coef,
                                        bounds_coef, soft_bound, heads, dx, ln_partial, loss_partial, stats, grad_logstd,
                                        grad_mu_bias, grad_value_bias, kl_out, logstd_grad_accum, mu_store, sigma_store,
                                        loss_scale, found_inf, ticket, defer);
}

// ---- Round 4: the row-local chain of the optimiser step as PHASES OF ONE LAUNCH -- LSTM forward (all T steps) ->
// LayerNorm + heads + PPO loss + their backward -> LSTM backward (all T steps) -> MLP backward.  Every one of the four
// kernels gives workgroup b the same 128 samples (sequences [32 b, 32 b + 32) x T = 4 = rows [128 b, 128 b + 128) of the
// sequence-major order) and reads, from the phase before it, only what the SAME workgroup wrote (hidden states -> loss;
// dh -> LSTM backward; dG -> MLP backward): a workgroup barrier is all that separates them -- three grid fills / drains
// fewer per optimiser step, and each phase finds its predecessor's rows in the CU's L2.  (The one-launch MLP forward
// stays in front as its own launch: its side job builds the weight operands ALL workgroups of this chain read, and the
// weight-gradient kernels -- reductions over all rows -- stay behind it.)  LDS: one dynamic block used phase after phase.
struct TrunkPhasesArgs {
    long long B; int T;
    // LSTM forward
    const lp16_t* x; long long ldx; const uint4* w_tiled; const float* bias; const float* c0; const float* h0;
    const unsigned char* done; lp16_t* h_out; lp16_t* c_all; lp16_t* gates; float* c_last; int ablate;
    // LayerNorm + heads + loss
    const float* gamma; const float* beta; float eps; const float* w; const float* wb; const float* logstd;
    const float* actions; const float* old_neglogp; const float* adv; const float* old_values; const float* returns;
    const float* old_mu; const float* old_sigma; float e_clip; int clip_value; float critic_coef, entropy_coef, bounds_coef,
        soft_bound;
    float* heads; lp16_t* dx; float* ln_partial; float* loss_partial; float* stats; float* grad_logstd; float* grad_mu_bias;
    float* grad_value_bias; float* kl_out; float* logstd_grad_accum; float* mu_store; float* sigma_store;
    const float* loss_scale; float* found_inf;
    // LSTM backward
    const uint4* w_hh_tiled; lp16_t* dG; float* bias_partial;
    // MLP backward
    const lp16_t* wt0; long long ldw0; const lp16_t* wt1; long long ldw1; const lp16_t* wt2; long long ldw2;
    const lp16_t* a3; long long a3_stride; const lp16_t* a2; const lp16_t* a1; float alpha;
    lp16_t* gz3; lp16_t* gz2; lp16_t* gz1; float* part3; float* part2; float* part1;
};
__global__ __launch_bounds__(512) void trunk_phases_kernel(const TrunkPhasesArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char trunk_lds[];
    TRUNK_STAMP(0)
    lstm_seq_fwd_body<3, 22, lp16_t, lp16_t, SEQ_FWD_NC>(a.T, a.B, a.x, a.ldx, nullptr, 0, a.w_tiled, a.bias, a.c0, a.done, a.h_out, a.c_all,
                                                 a.gates, a.ablate, a.c_last, a.h0);
    __syncthreads();      // this workgroup's hidden states are written (h_out slots 1 .. T of its 32 sequences)
    TRUNK_STAMP(1)
    ln_heads_loss_body<3, 4, lp16_t, lp16_t>(reinterpret_cast<float*>(trunk_lds),
                                             reinterpret_cast<float*>(trunk_lds) + 8 * 5 * 256, a.B * a.T, a.h_out, a.T, a.gamma,
                                             a.beta, a.eps, a.w, a.wb, a.logstd, a.actions, a.old_neglogp, a.adv, a.old_values,
                                             a.returns, a.old_mu, a.old_sigma, a.e_clip, a.clip_value, a.critic_coef,
                                             a.entropy_coef, a.bounds_coef, a.soft_bound, a.heads, a.dx, a.ln_partial,
                                             a.loss_partial, a.stats, a.grad_logstd, a.grad_mu_bias, a.grad_value_bias, a.kl_out,
                                             a.logstd_grad_accum, a.mu_store, a.sigma_store, a.loss_scale, a.found_inf, nullptr,
                                             1 /* the loss rows are folded by the column-sum launch (VineLossFinalize) */);
    __syncthreads();      // dh of this workgroup's rows
    TRUNK_STAMP(2)
    lstm_seq_bwd_body<SEQ_BWD_RING, lp16_t, lp16_t>(a.T, a.B, a.dx, a.w_hh_tiled, a.gates, a.c_all, a.c0, a.done, a.dG,
                                                    a.bias_partial, a.ablate, a.c_last);
    __syncthreads();      // dG of this workgroup's rows
    TRUNK_STAMP(3)
    mlp3_bwd_elu_mfma_body<64, 128, 256, 8, 8>(a.B * a.T, a.dG, 4 * SEQ_H, a.wt0, a.ldw0, a.wt1, a.ldw1, a.wt2, a.ldw2, a.a3,
                                              a.a3_stride, a.a2, a.a1, a.alpha, a.gz3, a.gz2, a.gz1, a.part3, a.part2, a.part1);
    TRUNK_STAMP(4)
}
#undef LHL_COL
#undef LHL_ROW

// The Adam kernel's last workgroup to finish bumps the step counter and applies the learning-rate schedule (every
// workgroup has read the old step / lr by then); it resets the ticket, so a launch always finds 0.
// amp = {loss scale, growth tracker, growth interval, -} and found_inf (both optional) restate torch.amp.GradScaler on the
// device (the reference's `mixed_precision: True` drives its fp16 update through one): gradients arrive multiplied by the
// loss scale and are unscaled here; when *found_inf is set the whole step is skipped (parameters, moments, step counter
// and the 16-bit parameter copies keep their values; the gradient block is still cleared) and the scale is halved;
// after `growth interval` consecutive good steps it is doubled.  The learning-rate schedule runs either way
// (rl_games updates it after scaler.step whatever that did).
__global__ void adam_kernel(long long n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float* lr_p, float* step_p,
                            float beta1, float beta2, float eps, float wd, float gscale, lp16_t* __restrict__ shadow,
                            const float* kl, float kl_scale, float kl_thr, float min_lr, float max_lr,
                            float* amp, float* found_inf, unsigned int* ticket, unsigned int* sub) {
    const float loss_scale = amp ? amp[0] : 1.0f;
    const bool skip = found_inf && *found_inf != 0.0f;
    if (amp) gscale = gscale / loss_scale;
    if (skip) {      // (uniform over the grid) clear the gradient block, touch nothing else
        const long long n4s = n >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4s; i += (long long)gridDim.x * blockDim.x)
            st4(g + 4 * i, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        const long long ts = (n4s << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (ts < n) g[ts] = 0.0f;
    }
    const float step = *step_p + 1.0f;                    // every thread reads the old values; the last workgroup
    const float lr = *lr_p;                               // to finish writes the new ones
    const float bc1 = 1.0f - __powf(beta1, step), bc2 = 1.0f - __powf(beta2, step);
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
    const long long n4 = skip ? 0 : n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 pp = ld4(p + 4 * i), gg = ld4(g + 4 * i), mm = ld4(m + 4 * i), vv = ld4(v + 4 * i);
        float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w};
        float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float gr = ga[u] * gscale + wd * pa[u];
            ma[u] = beta1 * ma[u] + (1.0f - beta1) * gr;
            va[u] = beta2 * va[u] + (1.0f - beta2) * gr * gr;
            pa[u] -= step_size * ma[u] / (sqrtf(va[u]) * inv_sqrt_bc2 + eps);
        }
        st4(p + 4 * i, make_float4(pa[0], pa[1], pa[2], pa[3]));
        st4(m + 4 * i, make_float4(ma[0], ma[1], ma[2], ma[3]));
        st4(v + 4 * i, make_float4(va[0], va[1], va[2], va[3]));
        st4(g + 4 * i, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        if (shadow) st4(shadow + 4 * i, make_float4(pa[0], pa[1], pa[2], pa[3]));   // bf16 copy for the GEMM operands
    }
    // tail (n not a multiple of 4)
    const long long t = ((n >> 2) << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && !skip) {
        float gr = g[t] * gscale + wd * p[t];
        float mt = beta1 * m[t] + (1.0f - beta1) * gr, vt = beta2 * v[t] + (1.0f - beta2) * gr * gr;
        p[t] -= step_size * mt / (sqrtf(vt) * inv_sqrt_bc2 + eps);
        m[t] = mt; v[t] = vt; g[t] = 0.0f;
        if (shadow) shadow[t] = f2lp(p[t]);
    }
    // step counter (+ rl_games' AdaptiveScheduler when `kl` is given) by the last workgroup to finish -- formerly two
    // 1-thread launches behind this one
    __syncthreads();
    if (threadIdx.x == 0) {
        // two-level election (returning atomics on one word serialise at ~23 ns each): the workgroups of a group -- blockIdx
        // modulo 8 -- queue on the group's word, the last of each group on the common word
        bool last;
        if (sub) {
            const unsigned grp = blockIdx.x & 7u;
            const unsigned members = (gridDim.x - grp + 7u) >> 3;             // workgroups with this blockIdx & 7
            unsigned int* sw = sub + grp * 32u;
            last = false;
            if (atomicAdd(sw, 1u) == members - 1) {
                *sw = 0;
                const unsigned groups = gridDim.x < 8u ? gridDim.x : 8u;
                last = atomicAdd(ticket, 1u) == groups - 1;
            }
        } else {
            last = atomicAdd(ticket, 1u) == gridDim.x - 1;
        }
        if (last) {
            *ticket = 0;
            if (!skip) *step_p = step;
            if (amp) {                                    // GradScaler.update()
                if (skip) { amp[0] = loss_scale * 0.5f; amp[1] = 0.0f; }
                else {
                    const float tr = amp[1] + 1.0f;
                    if (tr >= amp[2]) { amp[0] = loss_scale * 2.0f; amp[1] = 0.0f; }
                    else amp[1] = tr;
                }
            }
            if (found_inf) *found_inf = 0.0f;
            if (kl) {
                const float k = *kl * kl_scale;
                float out = lr;
                if (k > 2.0f * kl_thr) out = fmaxf(lr / 1.5f, min_lr);
                if (k < 0.5f * kl_thr) out = fminf(lr * 1.5f, max_lr);
                *lr_p = out;
            }
        }
    }
}

__global__ void adaptive_lr_kernel(float* lr, const float* kl, float kl_scale, float thr, float min_lr, float max_lr) {
    const float k = *kl * kl_scale, l = *lr;
    float out = l;
    if (k > 2.0f * thr) out = fmaxf(l / 1.5f, min_lr);
    if (k < 0.5f * thr) out = fminf(l * 1.5f, max_lr);
    *lr = out;
}

__device__ __forceinline__ void philox4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                        unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// One wave handles envs e = wave_id, wave_id + n_waves, ...; lane l owns hidden units 4l..4l+3 (H == 256) or a
// strided subset (general H): partial dot products in registers, butterfly reduction across the 64 lanes.
#define HEAD_MAX_A 8
__global__ __launch_bounds__(256) void policy_head_kernel(long long N, int A, int H, const float* __restrict__ y,
                                                          const float* __restrict__ w_mu, const float* __restrict__ b_mu,
                                                          const float* __restrict__ w_v, const float* __restrict__ b_v,
                                                          const float* __restrict__ logstd, const float* __restrict__ vmean,
                                                          const float* __restrict__ vstd, int normalize_value,
                                                          unsigned seed_lo, unsigned seed_hi,
                                                          const long long* __restrict__ counter, float* __restrict__ mu_out,
                                                          float* __restrict__ sigma_out, float* __restrict__ value_out,
                                                          float* __restrict__ action_out, float* __restrict__ neglogp_out,
                                                          const float* __restrict__ ln_gamma,
                                                          const float* __restrict__ ln_beta, float ln_eps, float value_eps) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const unsigned long long ctr = (unsigned long long)counter[0];
    for (long long e = wave; e < N; e += n_waves) {
        float acc[HEAD_MAX_A + 1];
#pragma unroll
        for (int k = 0; k <= HEAD_MAX_A; ++k) acc[k] = 0.0f;
        // ln_gamma != NULL (H == 256 only): y is the raw LSTM output and the LayerNorm in front of the heads is applied
        // here (same arithmetic as layernorm_fwd_kernel: two-pass mean / biased variance over the 256 units)
        float ln_mean = 0.0f, ln_rstd = 1.0f;
        if (ln_gamma) {
            const float4 xv = ld4(y + e * H + 4 * lane);
            ln_mean = wave_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / 256.0f);
            const float a = xv.x - ln_mean, b = xv.y - ln_mean, c = xv.z - ln_mean, d = xv.w - ln_mean;
            ln_rstd = rsqrtf(wave_sum((a * a + b * b) + (c * c + d * d)) * (1.0f / 256.0f) + ln_eps);
        }
        for (int j = lane * 4; j < H; j += 256) {
            float4 yy = ld4(y + e * H + j);
            if (ln_gamma) {
                const float4 gm = ld4(ln_gamma + j), bt = ld4(ln_beta + j);
                yy = make_float4((yy.x - ln_mean) * ln_rstd * gm.x + bt.x, (yy.y - ln_mean) * ln_rstd * gm.y + bt.y,
                                 (yy.z - ln_mean) * ln_rstd * gm.z + bt.z, (yy.w - ln_mean) * ln_rstd * gm.w + bt.w);
            }
#pragma unroll
            for (int k = 0; k < HEAD_MAX_A; ++k) {
                if (k < A) {
                    const float4 w = ld4(w_mu + (long long)k * H + j);
                    acc[k] += yy.x * w.x + yy.y * w.y + yy.z * w.z + yy.w * w.w;
                }
            }
            const float4 w = ld4(w_v + j);
            acc[HEAD_MAX_A] += yy.x * w.x + yy.y * w.y + yy.z * w.z + yy.w * w.w;
        }
#pragma unroll
        for (int k = 0; k <= HEAD_MAX_A; ++k) {
            if (k < A || k == HEAD_MAX_A) {
                acc[k] = wave_sum(acc[k]);
            }
        }
        if (lane == 0) {
            float v = acc[HEAD_MAX_A] + b_v[0];
            if (normalize_value == 2) {      // RunningMeanStd's own float64 statistics: mean.float(), sqrt(var.float() + eps)
                const float vm = (float)reinterpret_cast<const double*>(vmean)[0];
                const float vs = sqrtf((float)reinterpret_cast<const double*>(vstd)[0] + value_eps);
                v = fminf(fmaxf(v, -5.0f), 5.0f) * vs + vm;
            } else if (normalize_value) v = fminf(fmaxf(v, -5.0f), 5.0f) * vstd[0] + vmean[0];
            value_out[e] = v;
            float nlp = 0.9189385332046727f * A;
            unsigned r[4];
            for (int k = 0; k < A; k += 2) {
                philox4((unsigned)e, (unsigned)ctr, 0x504f4c59u | 0u, (unsigned)(k >> 1), seed_lo, seed_hi, r);
                const float u1 = 1.0f - (float)(r[0] >> 8) * (1.0f / 16777216.0f);
                const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
                const float rad = sqrtf(-2.0f * __logf(u1));
                float sn, cs;
                __sincosf(6.283185307179586f * u2, &sn, &cs);
                const float eps2[2] = {rad * cs, rad * sn};
                for (int q = 0; q < 2 && k + q < A; ++q) {
                    const int kk = k + q;
                    const float m = acc[kk] + b_mu[kk], ls = logstd[kk], sg = __expf(ls);
                    const float a = m + sg * eps2[q];
                    mu_out[e * A + kk] = m;
                    sigma_out[e * A + kk] = sg;
                    action_out[e * A + kk] = a;
                    nlp += 0.5f * eps2[q] * eps2[q] + ls;
                }
            }
            neglogp_out[e] = nlp;
        }
    }
}

// The same head for the rollout's default shape (H == 256, LayerNorm applied here): SIXTEEN lanes per env row, four rows
// per wave -- the reductions are DPP rotate-adds inside a 16-lane row (row_allsum16) instead of wave-wide sums, and four
// lanes per wave (one per row) sample in parallel where lane 0 sampled alone.  Same Philox keys, same formulas.
__global__ __launch_bounds__(256) void policy_head16_kernel(long long N, int A, const float* __restrict__ y,
                                                            const float* __restrict__ w_mu, const float* __restrict__ b_mu,
                                                            const float* __restrict__ w_v, const float* __restrict__ b_v,
                                                            const float* __restrict__ logstd, const float* __restrict__ vmean,
                                                            const float* __restrict__ vstd, int normalize_value,
                                                            unsigned seed_lo, unsigned seed_hi,
                                                            const long long* __restrict__ counter, float* __restrict__ mu_out,
                                                            float* __restrict__ sigma_out, float* __restrict__ value_out,
                                                            float* __restrict__ action_out, float* __restrict__ neglogp_out,
                                                            const float* __restrict__ ln_gamma,
                                                            const float* __restrict__ ln_beta, float ln_eps, float value_eps) {
    constexpr int H = 256;
    const int lane = threadIdx.x & 63, sub = lane >> 4, cl = lane & 15;
    const long long e = ((((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) << 2) + sub;      // N % 4 == 0 (host check)
    const unsigned long long ctr = (unsigned long long)counter[0];
    float xa[4][4], s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 v = ld4(y + e * H + 4 * (cl + 16 * j));
        xa[j][0] = v.x; xa[j][1] = v.y; xa[j][2] = v.z; xa[j][3] = v.w;
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = row_allsum16(s) * (1.0f / H);
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xa[j][u] -= mean;
            q += xa[j][u] * xa[j][u];
        }
    const float rstd = rsqrtf(row_allsum16(q) * (1.0f / H) + ln_eps);
    float acc[HEAD_MAX_A + 1];
#pragma unroll
    for (int k = 0; k <= HEAD_MAX_A; ++k) acc[k] = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * (cl + 16 * j);
        const float4 gm = ld4(ln_gamma + c), bt = ld4(ln_beta + c);
        const float yy[4] = {xa[j][0] * rstd * gm.x + bt.x, xa[j][1] * rstd * gm.y + bt.y, xa[j][2] * rstd * gm.z + bt.z,
                             xa[j][3] * rstd * gm.w + bt.w};
#pragma unroll
        for (int k = 0; k < HEAD_MAX_A; ++k) {
            if (k < A) {
                const float4 w = ld4(w_mu + (long long)k * H + c);
                acc[k] += yy[0] * w.x + yy[1] * w.y + yy[2] * w.z + yy[3] * w.w;
            }
        }
        const float4 w = ld4(w_v + c);
        acc[HEAD_MAX_A] += yy[0] * w.x + yy[1] * w.y + yy[2] * w.z + yy[3] * w.w;
    }
#pragma unroll
    for (int k = 0; k <= HEAD_MAX_A; ++k)
        if (k < A || k == HEAD_MAX_A) acc[k] = row_allsum16(acc[k]);
    if (cl == 0) {
        float v = acc[HEAD_MAX_A] + b_v[0];
        if (normalize_value == 2) {          // RunningMeanStd's own float64 statistics: mean.float(), sqrt(var.float() + eps)
            const float vm = (float)reinterpret_cast<const double*>(vmean)[0];
            const float vs = sqrtf((float)reinterpret_cast<const double*>(vstd)[0] + value_eps);
            v = fminf(fmaxf(v, -5.0f), 5.0f) * vs + vm;
        } else if (normalize_value) v = fminf(fmaxf(v, -5.0f), 5.0f) * vstd[0] + vmean[0];
        value_out[e] = v;
        float nlp = 0.9189385332046727f * A;
        unsigned r[4];
        for (int k = 0; k < A; k += 2) {
            philox4((unsigned)e, (unsigned)ctr, 0x504f4c59u | 0u, (unsigned)(k >> 1), seed_lo, seed_hi, r);
            const float u1 = 1.0f - (float)(r[0] >> 8) * (1.0f / 16777216.0f);
            const float u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * __logf(u1));
            float sn, cs;
            __sincosf(6.283185307179586f * u2, &sn, &cs);
            const float eps2[2] = {rad * cs, rad * sn};
            for (int qq = 0; qq < 2 && k + qq < A; ++qq) {
                const int kk = k + qq;
                const float m = acc[kk] + b_mu[kk], ls = logstd[kk], sg = __expf(ls);
                const float a = m + sg * eps2[qq];
                mu_out[e * A + kk] = m;
                sigma_out[e * A + kk] = sg;
                action_out[e * A + kk] = a;
                nlp += 0.5f * eps2[qq] * eps2[qq] + ls;
            }
        }
        neglogp_out[e] = nlp;
    }
}

#define ROLLOUT_POST_BLOCKS 1024
__global__ __launch_bounds__(256) void rollout_post_kernel(
    long long N, int H, const float* __restrict__ rew, const long long* __restrict__ reset,
    const unsigned char* __restrict__ timeouts, const float* __restrict__ values, float shift, float scale, float gamma_b,
    float* __restrict__ shaped, unsigned char* __restrict__ dones, float* __restrict__ cur_r, float* __restrict__ cur_l,
    float* __restrict__ h_state, float* __restrict__ c_state, float* __restrict__ partial, void* __restrict__ h_op,
    long long h_op_stride, int h_op_bf16) {
    // 16 threads per env: thread 0 of the group keeps the books, all 16 clear the LSTM state rows of a finished env
    // (256 contiguous bytes per group and pass).  The episode statistics leave as ONE {sum r, sum l, count} row per
    // workgroup (no atomics: with float atomicAdd on three shared words this kernel took 126 us when 30 % of the
    // envs finished in the same step, and the sum depended on the arrival order).
    float sr = 0.0f, sl = 0.0f, cnt = 0.0f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < N * 16;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long e = idx >> 4;
        const int k = (int)(idx & 15);
        const bool done = reset[e] != 0;
        if (k == 0) {
            const float r = rew[e];
            float s = (r + shift) * scale;
            if (gamma_b != 0.0f && timeouts[e]) s += gamma_b * values[e];
            shaped[e] = s;
            dones[e] = done ? 1 : 0;
            const float cr = cur_r[e] + r, cl = cur_l[e] + 1.0f;
            if (done) {
                sr += cr; sl += cl; cnt += 1.0f;
                cur_r[e] = 0.0f; cur_l[e] = 0.0f;
            } else {
                cur_r[e] = cr; cur_l[e] = cl;
            }
        }
        if (done) {
            const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int j = 4 * k; j < H; j += 64) {
                st4(h_state + e * H + j, z);
                st4(c_state + e * H + j, z);
                if (h_op) {   // the GEMM-operand copy of h (column block of the [x | h] buffer of the fused inference)
                    if (h_op_bf16) st4((lp16_t*)h_op + e * h_op_stride + j, z);
                    else st4((float*)h_op + e * h_op_stride + j, z);
                }
            }
        }
    }
    sr = wave_sum(sr); sl = wave_sum(sl); cnt = wave_sum(cnt);
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = sr; red[wave][1] = sl; red[wave][2] = cnt; }
    __syncthreads();
    if (threadIdx.x < 3)
        partial[blockIdx.x * 3 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// rl_games AverageMeter.update for both meters from the step's (sum, count) -- the per-workgroup rows of the kernel
// above summed in a fixed order by one 256-thread workgroup.
__device__ __forceinline__ void rollout_finalize_body(float* __restrict__ meter, float max_size,
                                                      long long* __restrict__ counter,
                                                      const float* __restrict__ partial, int blocks) {
    float v[3] = {0.0f, 0.0f, 0.0f};
    for (int b = threadIdx.x; b < blocks; b += 256) {
        v[0] += partial[b * 3]; v[1] += partial[b * 3 + 1]; v[2] += partial[b * 3 + 2];
    }
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) red[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float sum_r = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    const float sum_l = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const float size = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    if (size > 0.0f) {
        const float sc = fminf(size, max_size);
        const float sums[2] = {sum_r, sum_l};
        for (int m = 0; m < 2; ++m) {
            const float new_mean = sums[m] / size;
            const float cur = meter[2 * m + 1];
            const float old_size = fminf(max_size - sc, cur);
            const float sum = old_size + sc;
            meter[2 * m] = (meter[2 * m] * old_size + new_mean * sc) / sum;
            meter[2 * m + 1] = sum;
        }
    }
    meter[4] = sum_r; meter[5] = sum_l; meter[6] = size;      // this step's totals, for introspection
    counter[0] += 1;
}
__global__ __launch_bounds__(256) void rollout_finalize_kernel(float* __restrict__ meter, float max_size,
                                                               long long* __restrict__ counter,
                                                               const float* __restrict__ partial, int blocks) {
    rollout_finalize_body(meter, max_size, counter, partial, blocks);
}

int grid_for(long long work, int threads) {
    long long blocks = (work + threads - 1) / threads;
    if (blocks > 256 * 16) blocks = 256 * 16;   // 16 workgroups per CU, grid-stride beyond
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}


// ================================================================================================================
// Rollout inference at the reference's precision (rl_games runs play_steps in fp32; only the update is autocast):
// the policy trunk of one rollout step as TWO fp32 matrix-core kernels, v_mfma_f32_16x16x4_f32 (f32 operands, f32
// accumulate: bitwise an fmaf chain per output, 64 FLOP/clk/SIMD = the fp32 vector rate on gfx950).
// Both compute the transposed product D^T = W X^T like the 16-bit kernels above, so that a lane ends up with 4
// CONSECUTIVE units of ONE batch row (C/D map of the 16x16 tile: column = lane & 15 -> batch row, row = 4 (lane >> 4) + r
// -> unit).  The k index of an MFMA is free as long as both operands agree: lane group g = lane >> 4 takes
// k = 16 j + 4 g + i for the i-th MFMA of k-block j, i.e. ONE float4 per lane and operand feeds four MFMAs.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma_f32(float a, float b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- LSTM step, fp32: gates = [x | h] Wcat^T + bias on the matrix cores, pointwise cell update as the epilogue; the
// [N, 4H] pre-activations never reach memory.  Workgroup = 4 waves = 64 batch rows x 64 hidden units (x 4 gates); each
// wave owns 16 rows and all 16 (gate, unit-tile) accumulators.  The wave's X fragments (16 rows x K) are loaded ONCE
// into registers (KB float4 per lane); the weights arrive pre-tiled ([unit block][k-block][gate * 64 + unit][16], one
// contiguous 16 KB chunk per workgroup and k-block: vine_lstm_tile_weights_f32) and are streamed through two LDS buffers
// (row pitch 20 floats: the 16 lanes of a fragment read hit 16 disjoint 4-bank groups), the next chunk's global loads in
// flight under the current chunk's 64 MFMAs per wave.  MFMA-bound by construction: 64 MFMAs x 32 cycles per 16
// ds_read_b128 + 4 global loads per lane.
#define LSTM_F32_PITCH 20
template <int KB>      // K = 16 KB
__global__ __launch_bounds__(256, 4) void lstm_step_f32_kernel(long long N, const float* __restrict__ xh, long long ldx,
                                                               const float* __restrict__ wt, const float* __restrict__ bias,
                                                               const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                               long long ldh, float* __restrict__ c_out,
                                                               float* __restrict__ hp_next, long long ldhp) {
    // Register budget: 128 per lane (4 waves per SIMD = 4 workgroups per CU, 4 x 40 KB of LDS): all 1024 workgroups of the
    // rollout's 16384 rows are resident at once, and a SIMD always has another wave's MFMAs to issue while one waits at its
    // workgroup's barrier or for its LDS reads (2 waves per SIMD with the whole X fragment in registers: 125 us, the MFMA
    // pipe busy 60 % of the time).  X is therefore streamed like the weights: one float4 per lane and k-block, requested
    // two k-blocks ahead.
    constexpr int H = 256;
    __shared__ __attribute__((aligned(16))) float wl[2][256 * LSTM_F32_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane & 15, g = lane >> 4;
    // workgroups w, w + 8, w + 16, w + 24 (same XCD under round-robin dispatch) share one row block: its X is read from
    // HBM once per XCD-local L2 instead of four times
    const int w = blockIdx.x;
    const int rb = (w >> 5) * 8 + (w & 7), ub = (w >> 3) & 3;
    const long long row = (long long)rb * 64 + wave * 16 + u;
    const float* wsrc = wt + (long long)ub * KB * 4096;
    const float* xrow = xh + row * ldx + 4 * g;
    float4 sreg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sreg[i] = ld4(wsrc + 4 * (tid + 256 * i));
    float4 xq[3];                                                      // k-blocks j, j + 1, j + 2 (ring, static indices)
    xq[0] = ld4(xrow);
    xq[1] = ld4(xrow + 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;                                   // float4 index: row q >> 2, chunk q & 3
        st4(&wl[0][(q >> 2) * LSTM_F32_PITCH + 4 * (q & 3)], sreg[i]);
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[gg][t] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KB; ++j) {                      // fully unrolled: the rings stay register names
        const int buf = j & 1;
        if (j + 1 < KB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sreg[i] = ld4(wsrc + (long long)(j + 1) * 4096 + 4 * (tid + 256 * i));
        }
        if (j + 2 < KB) xq[(j + 2) % 3] = ld4(xrow + 16 * (j + 2));
        const float xa[4] = {xq[j % 3].x, xq[j % 3].y, xq[j % 3].z, xq[j % 3].w};
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            // four unit tiles at a time: consecutive MFMAs go to different accumulators (a dependent MFMA of this shape
            // waits 40 cycles, an independent one issues after 32)
            float wa[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 wv = ld4(&wl[buf][(gg * 64 + 16 * t + u) * LSTM_F32_PITCH + 4 * g]);
                wa[t][0] = wv.x; wa[t][1] = wv.y; wa[t][2] = wv.z; wa[t][3] = wv.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[gg][t] = mfma_f32(wa[t][i], xa[i], acc[gg][t]);
        }
        if (j + 1 < KB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + 256 * i;
                st4(&wl[buf ^ 1][(q >> 2) * LSTM_F32_PITCH + 4 * (q & 3)], sreg[i]);
            }
        }
        __syncthreads();
    }
    // ---- epilogue: lane holds, per unit tile t, the 4 gates of units ub*64 + 16 t + 4 g + {0..3} of its row
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int unit = ub * 64 + 16 * t + 4 * g;
        const float4 bi = ld4(bias + 0 * H + unit), bf = ld4(bias + 1 * H + unit), bg = ld4(bias + 2 * H + unit),
                     bo = ld4(bias + 3 * H + unit);
        const float4 cp = ld4(c_prev + row * H + unit);
        const float bia[4] = {bi.x, bi.y, bi.z, bi.w}, bfa[4] = {bf.x, bf.y, bf.z, bf.w}, bga[4] = {bg.x, bg.y, bg.z, bg.w},
                    boa[4] = {bo.x, bo.y, bo.z, bo.w}, cpa[4] = {cp.x, cp.y, cp.z, cp.w};
        float cn[4], hn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gi = sigmoidf_(acc[0][t][r] + bia[r]), gf = sigmoidf_(acc[1][t][r] + bfa[r]);
            const float gc = tanhf_(acc[2][t][r] + bga[r]), go = sigmoidf_(acc[3][t][r] + boa[r]);
            cn[r] = gf * cpa[r] + gi * gc;
            hn[r] = go * tanhf_(cn[r]);
        }
        st4(c_out + row * H + unit, make_float4(cn[0], cn[1], cn[2], cn[3]));
        st4(h_out + row * ldh + unit, make_float4(hn[0], hn[1], hn[2], hn[3]));
        if (hp_next) st4(hp_next + row * ldhp + unit, make_float4(hn[0], hn[1], hn[2], hn[3]));
    }
}

// [w_ih | 0 | w_hh] rows (4H x K, row stride ldw) -> the step kernel's tiles: dst[((ub * KB + j) * 256 + gate * 64 + uu) * 16 + c]
// = W[gate * H + ub * 64 + uu][16 j + c]
__global__ void lstm_tile_weights_f32_kernel(int KB, const float* __restrict__ w, long long ldw, float* __restrict__ dst) {
    const long long total = 4LL * KB * 256 * 4;                         // float4 elements
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(q & 3);
        const long long rest = q >> 2;
        const int r = (int)(rest & 255);
        const long long bj = rest >> 8;
        const int j = (int)(bj % KB), ub = (int)(bj / KB);
        const int gate = r >> 6, uu = r & 63;
        st4(dst + 4 * q, ld4(w + (long long)(gate * 256 + ub * 64 + uu) * ldw + 16 * j + 4 * c4));
    }
}

// ---- LSTM step, fp32 operands, EXACT products on the bf16 matrix cores ("split" form).  An fp32 number is the exact sum
// of three bfloat16 numbers: hi = bf16(v), mid = bf16(v - hi), lo = v - hi - mid (round to nearest: each piece takes 8 of
// the 24 significand bits; both subtractions are exact in fp32 and lo fits bf16's 8 bits).  A product of two bf16 pieces
// has 16 significand bits and is exact in the matrix core's fp32 accumulator, so
//     w x = sum over the 9 piece pairs (w_p x_q)
// holds exactly, term by term, and gates = [x | h] Wcat^T is formed as NT = 9 v_mfma_f32_16x16x32_bf16 per tile and
// k-step instead of 8 v_mfma_f32_16x16x4_f32: every bit of every fp32 product enters the fp32 accumulation (the native
// fp32 MFMA rounds each product into the running sum as well), at 16 / 9 of the fp32 matrix rate.  NT = 6 is the same
// without the three pairs below 2^-24 of the product (mid lo, lo mid, lo lo: less than the rounding of one fp32
// multiply); kept as a measured variant, not the default.
// Shapes: workgroup = 4 waves = 64 RT batch rows x 32 hidden units (x 4 gates = 8 weight tiles); a wave owns RT row tiles
// of 16 rows and all 8 x RT accumulator tiles (transposed product D^T = W X^T as above: a lane ends up with 4 consecutive
// units of one batch row per tile, all four gates of a unit on the same lane).  The weights arrive pre-split and
// pre-tiled in MFMA-fragment order ([unit block][k-step][piece][tile][lane][8], vine_lstm_tile_weights_split: one
// contiguous 24 KB chunk per workgroup and k-step) and are streamed through two LDS buffers, the next chunk's global loads
// in flight under the current chunk's 72 RT MFMAs per wave; a fragment read is one conflict-free ds_read_b128 per lane.
// X is loaded as fp32 (two float4 per lane, row tile and k-step, two k-steps ahead) and split in registers.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16_pack2(float a, float b) {          // v_cvt_pk_bf16_f32, round to nearest even
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// eight fp32 values -> the three bf16 pieces of each (packed pairs), v = hi + mid + lo exactly
__device__ __forceinline__ void split3_bf16x8(const float (&v)[8], uint4& hi, uint4& mid, uint4& lo) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        h[i] = bf16_pack2(a, b);
        const float ra = a - __uint_as_float(h[i] << 16), rb = b - __uint_as_float(h[i] & 0xFFFF0000u);
        m[i] = bf16_pack2(ra, rb);
        const float sa = ra - __uint_as_float(m[i] << 16), sb = rb - __uint_as_float(m[i] & 0xFFFF0000u);
        l[i] = bf16_pack2(sa, sb);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    mid = make_uint4(m[0], m[1], m[2], m[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ f32x4_t mfma_bf16(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

#define LSTM_SPLIT_CHUNK (3 * 8 * 64 * 8)          // bf16 elements of one (unit block, k-step) weight chunk: 24 KB
// K = 32 KS; RT row tiles per wave; NT = 9 (exact) or 6 piece pairs.  The split of the NEXT k-step's X sits inside the
// current k-step's MFMA stream (second set of piece registers): an MFMA holds the SIMD's vector issue for 8 of its 16
// cycles, so the ~44 RT conversion instructions per k-step ride in the gaps instead of standing in front of the k-step's
// first MFMA (99.6 -> 76.7 us at RT = 2, profiles/r03/rollout_f32_split.txt).
#ifdef SPLIT_TIMING
// (debug build, scripts/ubench/lstm_split_clock.py) per wave: shader clock (s_memtime) and 100 MHz real time (s_memrealtime)
// at kernel entry, after the prologue, after the last k-step, at the end
__device__ unsigned long long split_t[8192 * 8];
#define SPLIT_STAMP(i)                                                                                                  \
    if (lane == 0) {                                                                                                    \
        split_t[((blockIdx.x * 4 + wave) & 8191) * 8 + 2 * (i)] = __builtin_amdgcn_s_memtime();                          \
        split_t[((blockIdx.x * 4 + wave) & 8191) * 8 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();                  \
    }
#else
#define SPLIT_STAMP(i)
#endif
template <int KS, int RT, int NT>
__global__ __launch_bounds__(256, 2) void lstm_step_split_kernel(long long N, const float* __restrict__ xh, long long ldx,
                                                                  const unsigned short* __restrict__ wt,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                                  long long ldh, float* __restrict__ c_out,
                                                                  float* __restrict__ hp_next, long long ldhp, int xcd_map) {
    constexpr int H = 256;
    __shared__ __attribute__((aligned(16))) uint4 wl[2][3 * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane & 15, g = lane >> 4;
    const int w = blockIdx.x;
    const int rbs = xcd_map ? (w >> 6) * 8 + (w & 7) : (w >> 3);
    const int ub = xcd_map ? (w >> 3) & 7 : (w & 7);
    const long long row0 = (long long)rbs * (64 * RT) + wave * (16 * RT) + u;
    const uint4* wsrc = reinterpret_cast<const uint4*>(wt) + (long long)ub * KS * (LSTM_SPLIT_CHUNK / 8);
    const float* xrow = xh + row0 * ldx + 8 * g;
    SPLIT_STAMP(0)
    float4 xf[RT][2];
    uint4 xp[2][RT][3];
    {
        uint4 sreg[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) sreg[i] = wsrc[tid + 256 * i];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            xf[rt][0] = ld4(xrow + (long long)rt * 16 * ldx);
            xf[rt][1] = ld4(xrow + (long long)rt * 16 * ldx + 4);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) wl[0][tid + 256 * i] = sreg[i];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float v[8] = {xf[rt][0].x, xf[rt][0].y, xf[rt][0].z, xf[rt][0].w,
                                xf[rt][1].x, xf[rt][1].y, xf[rt][1].z, xf[rt][1].w};
            split3_bf16x8(v, xp[0][rt][0], xp[0][rt][1], xp[0][rt][2]);
            if (KS > 1) {
                xf[rt][0] = ld4(xrow + (long long)rt * 16 * ldx + 32);
                xf[rt][1] = ld4(xrow + (long long)rt * 16 * ldx + 32 + 4);
            }
        }
    }
    f32x4_t acc[8][RT];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[t][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    float4 cp[2][RT];
    __syncthreads();
    SPLIT_STAMP(1)
#pragma unroll
    for (int j = 0; j < KS; ++j) {                      // fully unrolled: buffers and piece sets stay register names
        const int buf = j & 1;
        uint4 sreg[6];
        if (j + 1 < KS) {
#pragma unroll
            for (int i = 0; i < 6; ++i) sreg[i] = wsrc[(long long)(j + 1) * (LSTM_SPLIT_CHUNK / 8) + tid + 256 * i];
        } else {                                                     // the epilogue's cell states, requested a k-step early
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    cp[ut][rt] = ld4(c_prev + (row0 + 16 * rt) * H + ub * 32 + 16 * ut + 4 * g);
        }
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            uint4 wf[2][3];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int p = 0; p < 3; ++p) wf[tt][p] = wl[buf][(p * 8 + 2 * tp + tt) * 64 + lane];
            constexpr int PP[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0}, QQ[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int n = 9 - NT; n < 9; ++n)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[2 * tp + tt][rt] = mfma_bf16(wf[tt][PP[n]], xp[buf][rt][QQ[n]], acc[2 * tp + tt][rt]);
            if (tp == 0 && j + 1 < KS) {
                // X of k-step j + 1 (requested during k-step j - 1) -> the other piece set; then request k-step j + 2
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float v[8] = {xf[rt][0].x, xf[rt][0].y, xf[rt][0].z, xf[rt][0].w,
                                        xf[rt][1].x, xf[rt][1].y, xf[rt][1].z, xf[rt][1].w};
                    split3_bf16x8(v, xp[buf ^ 1][rt][0], xp[buf ^ 1][rt][1], xp[buf ^ 1][rt][2]);
                    if (j + 2 < KS) {
                        xf[rt][0] = ld4(xrow + (long long)rt * 16 * ldx + 32 * (j + 2));
                        xf[rt][1] = ld4(xrow + (long long)rt * 16 * ldx + 32 * (j + 2) + 4);
                    }
                }
            }
        }
        if (j + 1 < KS) {
#pragma unroll
            for (int i = 0; i < 6; ++i) wl[buf ^ 1][tid + 256 * i] = sreg[i];
        }
        __syncthreads();
    }
    SPLIT_STAMP(2)
#pragma unroll
    for (int ut = 0; ut < 2; ++ut) {
        const int unit = ub * 32 + 16 * ut + 4 * g;
        const float4 bi = ld4(bias + 0 * H + unit), bf = ld4(bias + 1 * H + unit), bg = ld4(bias + 2 * H + unit),
                     bo = ld4(bias + 3 * H + unit);
        const float bia[4] = {bi.x, bi.y, bi.z, bi.w}, bfa[4] = {bf.x, bf.y, bf.z, bf.w}, bga[4] = {bg.x, bg.y, bg.z, bg.w},
                    boa[4] = {bo.x, bo.y, bo.z, bo.w};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long long row = row0 + 16 * rt;
            const float cpa[4] = {cp[ut][rt].x, cp[ut][rt].y, cp[ut][rt].z, cp[ut][rt].w};
            float cn[4], hn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gi = sigmoidf_(acc[0 + ut][rt][r] + bia[r]), gf = sigmoidf_(acc[2 + ut][rt][r] + bfa[r]);
                const float gc = tanhf_(acc[4 + ut][rt][r] + bga[r]), go = sigmoidf_(acc[6 + ut][rt][r] + boa[r]);
                cn[r] = gf * cpa[r] + gi * gc;
                hn[r] = go * tanhf_(cn[r]);
            }
            st4(c_out + row * H + unit, make_float4(cn[0], cn[1], cn[2], cn[3]));
            st4(h_out + row * ldh + unit, make_float4(hn[0], hn[1], hn[2], hn[3]));
            if (hp_next) st4(hp_next + row * ldhp + unit, make_float4(hn[0], hn[1], hn[2], hn[3]));
        }
    }
    SPLIT_STAMP(3)
}

// ---- The same step with the work of a workgroup split the other way round (round 4, as mlp3_elu_split_kernel): the four
// waves share the SAME 64 batch rows (4 row tiles) of one 32-unit block and each owns ONE GATE of it (2 of the 8 weight
// tiles).  A wave's weight fragments are then its own and come straight from global memory (same tiled array as above:
// 6 fragments per k-step, a k-step ahead), nothing of the weights touches LDS; LDS carries the operand rows instead:
// wave w loads and splits row tile w of the k-step ahead and writes its three pieces as B fragments (12 KB per k-step,
// two buffers, one barrier per k-step), every wave reads all four row tiles.  A wave holds 8 accumulator tiles instead
// of 32: ~150 registers, three workgroups per CU instead of two with a quarter of the rows each -- the memory stages of
// one wave (exposed in the lone-wave loop of the kernel above: 66 % of the matrix rate) stand under the products of the
// other two waves of its SIMD.  The gate pre-activations meet in LDS once, after the last k-step (32 KB, over the operand
// buffers): wave w then runs the cell update of row tile w with all four gates of a unit on one lane, as above.
// Price: every workgroup streams its 264 KB of weight pieces for 64 rows instead of 256 (540 MB instead of 135 MB through
// the L1s per launch at 16384 rows, a third of their bandwidth over the kernel's duration).
// DUAL (round 5, bit 17 of `terms`): TWO accumulators per tile -- the hi x hi pair, whose terms have the magnitude of the
// result, and all the other pairs (at most 2^-8 of it) -- added once at the end.  One accumulator takes 9 (6) rounded additions
// per 32-deep k-step at the result's magnitude: 99 (66) per output, the 4-deep native fp32 instruction takes 88, which is why
// the two have the same error (profiles/r05/split_terms_error.txt).  Split, the large one takes 11, the small one's roundings
// sit 8 bits lower: the error against float64 falls well below the native instruction's on every input, max and rms.
template <int KS, int NT, bool DUAL = false>
__global__ __launch_bounds__(256, 3) void lstm_step_nsplit_kernel(long long N, const float* __restrict__ xh, long long ldx,
                                                                   const unsigned short* __restrict__ wt,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                                   long long ldh, float* __restrict__ c_out,
                                                                   float* __restrict__ hp_next, long long ldhp, int xcd_map) {
    constexpr int H = 256, XB = 4 * 3 * 64;                      // uint4 per operand buffer: [row tile][piece][lane]
    __shared__ __attribute__((aligned(16))) uint4 lds[2048];      // 2 operand buffers (24 KB); the gate exchange (32 KB) over them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane & 15, g = lane >> 4;
    const int w = blockIdx.x;
    const int rb = xcd_map ? (w >> 6) * 8 + (w & 7) : (w >> 3);
    const int ub = xcd_map ? (w >> 3) & 7 : (w & 7);
    const long long row = (long long)rb * 64 + 16 * wave + u;    // this lane's row: operand staging and the cell update
    const uint4* asrc = reinterpret_cast<const uint4*>(wt) + (long long)ub * KS * (LSTM_SPLIT_CHUNK / 8) + (2 * wave) * 64 + lane;
    const float* xrow = xh + row * ldx + 8 * g;
    SPLIT_STAMP(0)
    uint4 af[2][2][3];
    float4 xf[2];
    xf[0] = ld4(xrow);
    xf[1] = ld4(xrow + 4);
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[0][ut][p] = asrc[(p * 8 + ut) * 64];
    {
        const float v[8] = {xf[0].x, xf[0].y, xf[0].z, xf[0].w, xf[1].x, xf[1].y, xf[1].z, xf[1].w};
        uint4 pc[3];
        split3_bf16x8(v, pc[0], pc[1], pc[2]);
        if (KS > 1) {
            xf[0] = ld4(xrow + 32);
            xf[1] = ld4(xrow + 32 + 4);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) lds[(wave * 3 + p) * 64 + lane] = pc[p];
    }
    f32x4_t acc[2][4], accs[DUAL ? 2 : 1][DUAL ? 4 : 1];      // accs: the pairs below hi x hi (DUAL)
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            acc[ut][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            if (DUAL) accs[ut][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        }
    float4 cp[2];
    __syncthreads();
    SPLIT_STAMP(1)
    constexpr int PP[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0}, QQ[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
#pragma unroll
    for (int j = 0; j < KS; ++j) {                      // fully unrolled: buffers and fragment sets stay register names
        const int buf = j & 1;
        if (j + 1 < KS) {
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[buf ^ 1][ut][p] = asrc[(long long)(j + 1) * (LSTM_SPLIT_CHUNK / 8) + (p * 8 + ut) * 64];
        } else {                                                     // the cell states of the epilogue, a k-step early
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) cp[ut] = ld4(c_prev + row * H + ub * 32 + 16 * ut + 4 * g);
        }
        const uint4* bl = lds + buf * XB + lane;
        uint4 bfr[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bfr[0][p] = bl[p * 64];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (rt + 1 < 4) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bfr[(rt + 1) & 1][p] = bl[((rt + 1) * 3 + p) * 64];
            }
#pragma unroll
            for (int n = 9 - NT; n < 9; ++n)
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) {
                    if (DUAL && n < 8) accs[ut][rt] = mfma_bf16(af[buf][ut][PP[n]], bfr[rt & 1][QQ[n]], accs[ut][rt]);
                    else acc[ut][rt] = mfma_bf16(af[buf][ut][PP[n]], bfr[rt & 1][QQ[n]], acc[ut][rt]);
                }
            if (rt == 0 && j + 1 < KS) {
                // row tile `wave` of k-step j + 1 (requested during k-step j - 1) -> pieces -> the other buffer; then
                // request k-step j + 2
                const float v[8] = {xf[0].x, xf[0].y, xf[0].z, xf[0].w, xf[1].x, xf[1].y, xf[1].z, xf[1].w};
                uint4 pc[3];
                split3_bf16x8(v, pc[0], pc[1], pc[2]);
                if (j + 2 < KS) {
                    xf[0] = ld4(xrow + 32 * (j + 2));
                    xf[1] = ld4(xrow + 32 * (j + 2) + 4);
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) lds[(buf ^ 1) * XB + (wave * 3 + p) * 64 + lane] = pc[p];
            }
        }
        __syncthreads();
    }
    SPLIT_STAMP(2)
    // ---- the gates meet: wave (= gate) writes its 8 tiles, wave w reads the four gates of row tile w
    float4* ex = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
            ex[((rt * 4 + wave) * 2 + ut) * 64 + lane] =
                DUAL ? make_float4(acc[ut][rt][0] + accs[ut][rt][0], acc[ut][rt][1] + accs[ut][rt][1],
                                   acc[ut][rt][2] + accs[ut][rt][2], acc[ut][rt][3] + accs[ut][rt][3])
                     : make_float4(acc[ut][rt][0], acc[ut][rt][1], acc[ut][rt][2], acc[ut][rt][3]);
    __syncthreads();
#pragma unroll
    for (int ut = 0; ut < 2; ++ut) {
        const int unit = ub * 32 + 16 * ut + 4 * g;
        const float4 bi = ld4(bias + 0 * H + unit), bf = ld4(bias + 1 * H + unit), bg = ld4(bias + 2 * H + unit),
                     bo = ld4(bias + 3 * H + unit);
        const float4 ai = ex[((wave * 4 + 0) * 2 + ut) * 64 + lane], af_ = ex[((wave * 4 + 1) * 2 + ut) * 64 + lane],
                     ag = ex[((wave * 4 + 2) * 2 + ut) * 64 + lane], ao = ex[((wave * 4 + 3) * 2 + ut) * 64 + lane];
        const float pi[4] = {ai.x + bi.x, ai.y + bi.y, ai.z + bi.z, ai.w + bi.w};
        const float pf[4] = {af_.x + bf.x, af_.y + bf.y, af_.z + bf.z, af_.w + bf.w};
        const float pg[4] = {ag.x + bg.x, ag.y + bg.y, ag.z + bg.z, ag.w + bg.w};
        const float po[4] = {ao.x + bo.x, ao.y + bo.y, ao.z + bo.z, ao.w + bo.w};
        const float cpa[4] = {cp[ut].x, cp[ut].y, cp[ut].z, cp[ut].w};
        float cn[4], hn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gi = sigmoidf_(pi[r]), gf = sigmoidf_(pf[r]);
            const float gc = tanhf_(pg[r]), go = sigmoidf_(po[r]);
            cn[r] = gf * cpa[r] + gi * gc;
            hn[r] = go * tanhf_(cn[r]);
        }
        st4(c_out + row * H + unit, make_float4(cn[0], cn[1], cn[2], cn[3]));
        st4(h_out + row * ldh + unit, make_float4(hn[0], hn[1], hn[2], hn[3]));
        if (hp_next) st4(hp_next + row * ldhp + unit, make_float4(hn[0], hn[1], hn[2], hn[3]));
    }
    SPLIT_STAMP(3)
}

// [w_ih | 0 | w_hh] rows (4H x K fp32, row stride ldw) -> the split step kernel's chunks of bf16 pieces:
// dst[((((ub * KS + j) * 3 + piece) * 8 + 2 gate + ut) * 64 + lane) * 8 + e] = piece of W[gate * H + ub * 32 + 16 ut + (lane & 15)]
// [32 j + 8 (lane >> 4) + e] -- one thread per (chunk, tile, lane): 8 consecutive k of one weight row
__global__ void lstm_tile_weights_split_kernel(int KS, const float* __restrict__ w, long long ldw, uint4* __restrict__ dst) {
    const long long total = 8LL * KS * 8 * 64;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(q & 63), tile = (int)((q >> 6) & 7);
        const long long chunk = q >> 9;                                // ub * KS + j
        const int j = (int)(chunk % KS), ub = (int)(chunk / KS);
        const int gate = tile >> 1, ut = tile & 1;
        const float* src = w + (long long)(gate * 256 + ub * 32 + 16 * ut + (lane & 15)) * ldw + 32 * j + 8 * (lane >> 4);
        const float4 a = ld4(src), b = ld4(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint4 hi, mid, lo;
        split3_bf16x8(v, hi, mid, lo);
        uint4* d = dst + chunk * (LSTM_SPLIT_CHUNK / 8) + tile * 64 + lane;
        d[0] = hi;
        d[8 * 64] = mid;
        d[2 * 8 * 64] = lo;
    }
}

// ---- observation normalisation + the three MLP layers (C1 = 256, C2 = 128, C3 = 64, ELU), fp32, ONE launch: a wave
// carries its 16 rows through the layers in registers -- the accumulator of unit tile t of layer L (lane: units
// 16 t + 4 g + {0..3} of its row) IS the B operand float4 of k-block t of layer L + 1 (k = 16 t + 4 g + i), no exchange.
// The weights of one layer at a time are staged in LDS (row pitch K + 4 floats: conflict-free fragment reads), the
// same region for all three (W2 alone is 130 KB in fp32).  Writes the normalised observations (zero-padded to 32
// columns) and the MLP output into the LSTM operand row: x[row] = [mlp(64) | obs_n(F) | 0 ...].
__global__ __launch_bounds__(256) void mlp3_elu_f32_kernel(long long n, float* __restrict__ x, long long ldx,
                                                           const float* __restrict__ raw, int F_in,
                                                           const double* __restrict__ mean, const double* __restrict__ var,
                                                           float eps, float clip, const float* __restrict__ w1, long long ldw1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           long long ldw2, const float* __restrict__ b2,
                                                           const float* __restrict__ w3, long long ldw3,
                                                           const float* __restrict__ b3, float alpha, float* fin_meter,
                                                           float fin_max_size, long long* fin_counter,
                                                           const float* fin_partial, int fin_blocks) {
    // (round 4) workgroup 0 first folds the PREVIOUS rollout step's episode sums into the meters and advances the rollout
    // counter (rollout_finalize_kernel's body: formerly a one-workgroup launch of its own per step; the policy head of THIS
    // step, two launches further on, is the first reader of the counter)
    if (fin_meter && blockIdx.x == 0) rollout_finalize_body(fin_meter, fin_max_size, fin_counter, fin_partial, fin_blocks);
    constexpr int C1 = 256, C2 = 128, C3 = 64, P1 = 36, P2 = C1 + 4, P3 = C2 + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw_f32[];
    float* wl = reinterpret_cast<float*>(lds_raw_f32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane & 15, g = lane >> 4;
    const long long row = ((long long)blockIdx.x * 4 + wave) * 16 + u;
    // ---- layer-1 operand: normalised observation columns 16 j + 4 g + i (j = 0, 1), zero beyond F_in; the same values go
    // into the LSTM operand's observation block (32 columns behind the MLP output)
    float xn[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * j + 4 * g + i;
            float y = 0.0f;
            if (c < F_in) {
                // same arithmetic as normalize_obs_kernel: statistics cast to float first
                const float m = (float)mean[c], sd = sqrtf((float)var[c] + eps);
                y = (raw[row * F_in + c] - m) / sd;
                y = fminf(fmaxf(y, -clip), clip);
            }
            xn[j][i] = y;
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) st4(x + row * ldx + C3 + 16 * j + 4 * g, make_float4(xn[j][0], xn[j][1], xn[j][2], xn[j][3]));
    // ---- weights -> LDS, one layer at a time.  The global loads of a layer's weights are issued BEFORE the previous
    // layer's arithmetic (W1 and W2 right here), so that every staging step finds its data in registers: one L2 round trip
    // at the start of the kernel, none between the layers.
    // W1: [C1][32] zero-padded on the host (rows ldw1 apart) -> [C1][P1]
    float4 s1[8], s2[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i;
        s1[i] = ld4(w1 + (long long)(q >> 3) * ldw1 + 4 * (q & 7));
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int q = tid + 256 * i, r = q / (C1 / 4), c4 = q - r * (C1 / 4);
        s2[i] = ld4(w2 + (long long)r * ldw2 + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i;
        st4(wl + (q >> 3) * P1 + 4 * (q & 7), s1[i]);
    }
    __syncthreads();
    // (unit tiles four at a time in all three layers: consecutive MFMAs then go to different accumulators -- a dependent
    // MFMA of this shape waits 40 cycles, an independent one issues after 32)
#define MLP_F32_LAYER(WPITCH, NJ, OPERAND, T0, ACC)                                                         \
    {                                                                                                       \
        _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) ACC[tt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};         \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                    \
            float wa[4][4];                                                                                 \
            _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) {                                              \
                const float4 wv = ld4(wl + (16 * (T0 + tt) + u) * WPITCH + 16 * j + 4 * g);                 \
                wa[tt][0] = wv.x; wa[tt][1] = wv.y; wa[tt][2] = wv.z; wa[tt][3] = wv.w;                     \
            }                                                                                               \
            const float xa[4] = OPERAND;                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
                _Pragma("unroll") for (int tt = 0; tt < 4; ++tt) ACC[tt] = mfma_f32(wa[tt][i], xa[i], ACC[tt]); \
        }                                                                                                   \
    }
#define MLP_F32_XN {xn[j][0], xn[j][1], xn[j][2], xn[j][3]}
#define MLP_F32_A1 {a1[j].x, a1[j].y, a1[j].z, a1[j].w}
#define MLP_F32_A2 {a2[j].x, a2[j].y, a2[j].z, a2[j].w}
    float4 a1[C1 / 16];                                                // layer-1 activations = layer-2 operand
#pragma unroll
    for (int t0 = 0; t0 < C1 / 16; t0 += 4) {
        f32x4_t a[4];
        MLP_F32_LAYER(P1, 2, MLP_F32_XN, t0, a)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float4 bb = ld4(b1 + 16 * (t0 + tt) + 4 * g);
            a1[t0 + tt] = make_float4(elu1(a[tt][0] + bb.x, alpha), elu1(a[tt][1] + bb.y, alpha), elu1(a[tt][2] + bb.z, alpha),
                                      elu1(a[tt][3] + bb.w, alpha));
        }
    }
    __syncthreads();                                                   // everyone is done with W1
    // ---- W2 (already in registers) -> LDS [C2][P2]; W3's loads leave now, under layer 2
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int q = tid + 256 * i, r = q / (C1 / 4), c4 = q - r * (C1 / 4);
        st4(wl + r * P2 + 4 * c4, s2[i]);
    }
    float4 s3[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, r = q / (C2 / 4), c4 = q - r * (C2 / 4);
        s3[i] = ld4(w3 + (long long)r * ldw3 + 4 * c4);
    }
    __syncthreads();
    float4 a2[C2 / 16];
#pragma unroll
    for (int t0 = 0; t0 < C2 / 16; t0 += 4) {
        f32x4_t a[4];
        MLP_F32_LAYER(P2, C1 / 16, MLP_F32_A1, t0, a)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float4 bb = ld4(b2 + 16 * (t0 + tt) + 4 * g);
            a2[t0 + tt] = make_float4(elu1(a[tt][0] + bb.x, alpha), elu1(a[tt][1] + bb.y, alpha), elu1(a[tt][2] + bb.z, alpha),
                                      elu1(a[tt][3] + bb.w, alpha));
        }
    }
    __syncthreads();
    // ---- W3 (in registers) -> LDS [C3][P3]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, r = q / (C2 / 4), c4 = q - r * (C2 / 4);
        st4(wl + r * P3 + 4 * c4, s3[i]);
    }
    __syncthreads();
    {
        f32x4_t a[4];
        MLP_F32_LAYER(P3, C2 / 16, MLP_F32_A2, 0, a)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float4 bb = ld4(b3 + 16 * tt + 4 * g);
            st4(x + row * ldx + 16 * tt + 4 * g,
                make_float4(elu1(a[tt][0] + bb.x, alpha), elu1(a[tt][1] + bb.y, alpha), elu1(a[tt][2] + bb.z, alpha),
                            elu1(a[tt][3] + bb.w, alpha)));
        }
    }
#undef MLP_F32_LAYER
#undef MLP_F32_XN
#undef MLP_F32_A1
#undef MLP_F32_A2
}

// ---- the same three layers with EXACT products from bf16 pieces (round 4; the arithmetic of lstm_step_split_kernel: an fp32
// operand is the exact sum of three bfloat16 pieces, a piece product is exact in the matrix core's fp32 accumulator, NT = 9
// piece pairs = every bit of every fp32 product, 9 v_mfma_f32_16x16x32_bf16 instead of 8 v_mfma_f32_16x16x4_f32 at twice the
// rate per instruction: 864 x 16 cycles per 16 rows instead of 768 x 32).
// The split of the work is the other way round than in mlp3_elu_f32_kernel: the four waves of a workgroup share the SAME
// 16 RT rows and each owns a QUARTER OF THE UNITS of every layer (layer 1: 4 unit tiles, layer 2: 2, layer 3: 1).  A wave's
// weight fragments are then its own: they come straight from global memory in fragment order (vine_mlp3_tile_weights_split:
// [layer][wave][k-block][tile][piece][lane][8], one 16-B load per lane and fragment, D k-blocks ahead), every weight byte
// enters the CU once and never touches LDS.  LDS carries the activations between the layers instead: the producing wave
// applies bias + ELU, splits each value ONCE into its three pieces and writes them in the consumer's fragment order
// ([row tile][k-block][piece][lane][8]: a B fragment is one conflict-free ds_read_b128 per lane); two barriers per
// launch.  With RT row tiles a weight fragment feeds 3 RT matrix instructions, an activation fragment 3 TILES.
// At 4096 rows (RT = 1) all 1024 SIMDs work on 256 row tiles, where the one-wave-per-row-tile kernel above fills a quarter
// of them with 768 dependent instructions each.
#ifdef SPLIT_TIMING
#define MLP_STAMP(i)                                                                                                    \
    if (lane == 0) {                                                                                                    \
        mlp_split_t[((blockIdx.x * 4 + wave) & 4095) * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime();                     \
        mlp_split_t[((blockIdx.x * 4 + wave) & 4095) * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();             \
    }
#else
#define MLP_STAMP(i)
#endif
__device__ __forceinline__ void split3_bf16x4(const float (&v)[4], uint2& hi, uint2& mid, uint2& lo) {
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // (two values at a time: the residual subtractions as one packed instruction each)
        const f32x2_t ab = {v[2 * i], v[2 * i + 1]};
        h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, bf16x2_t));
        const f32x2_t hf = {__uint_as_float(h[i] << 16), __uint_as_float(h[i] & 0xFFFF0000u)};
        const f32x2_t r = ab - hf;
        m[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
        const f32x2_t mf = {__uint_as_float(m[i] << 16), __uint_as_float(m[i] & 0xFFFF0000u)};
        const f32x2_t s2 = r - mf;
        l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(s2, bf16x2_t));
    }
    hi = make_uint2(h[0], h[1]);
    mid = make_uint2(m[0], m[1]);
    lo = make_uint2(l[0], l[1]);
}
// fragments (64 uint4 each) of the tiled weights: layer 1 [4 waves][1][4 tiles][3], layer 2 [4][8][2][3], layer 3 [4][4][1][3]
#define MLP_SPLIT_L2_BASE 48
#define MLP_SPLIT_L3_BASE (48 + 192)
#define MLP_SPLIT_FRAGS (48 + 192 + 48)
// the weight fragments of one layer in flight: a ring of D + 1 k-blocks (all indices are compile-time after unrolling)
template <int TILES, int D>
struct MlpSplitRing {
    uint4 f[D + 1][TILES][3];
    __device__ __forceinline__ void issue(const uint4* __restrict__ src, int kb) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) f[kb % (D + 1)][t][p] = src[((kb * TILES + t) * 3 + p) * 64];
    }
};
// one layer of one wave: acc[t][rt] += W(tile t) . act(row tile rt) over KB k-blocks.  `ring` holds k-blocks 0 .. min(D, KB) - 1
// on entry (requested by the caller ahead of the barrier in front of the layer).
// DUAL (round 5): the hi x hi pair into part[0], every other pair into part[1] (see lstm_step_nsplit_kernel)
template <int TILES, int KB, int RT, int NT, int D, bool DUAL = false>
__device__ __forceinline__ void mlp_split_layer(MlpSplitRing<TILES, D>& ring, const uint4* __restrict__ a_src,
                                                const uint4* b_lds, f32x4_t (&acc)[TILES][RT]) {
    constexpr int NC = (DUAL || TILES * RT == 1) ? 2 : 1;        // a lone accumulator: alternate between two (dependent-issue stall)
    constexpr int PP[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0}, QQ[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
    f32x4_t part[NC][TILES][RT];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) part[c][t][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    uint4 bfr[2][RT][3];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int p = 0; p < 3; ++p) bfr[0][rt][p] = b_lds[((rt * KB + 0) * 3 + p) * 64];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + D < KB) ring.issue(a_src, kb + D);
        if (kb + 1 < KB) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int p = 0; p < 3; ++p) bfr[(kb + 1) & 1][rt][p] = b_lds[((rt * KB + kb + 1) * 3 + p) * 64];
        }
#pragma unroll
        for (int n = 9 - NT; n < 9; ++n)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                {
                    const int c = DUAL ? (n == 8 ? 0 : 1) : n % NC;      // (compile-time after unrolling)
                    part[c][t][rt] = mfma_bf16(ring.f[kb % (D + 1)][t][PP[n]], bfr[kb & 1][rt][QQ[n]], part[c][t][rt]);
                }
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            acc[t][rt] = part[0][t][rt];
            if (NC == 2) acc[t][rt] += part[NC - 1][t][rt];
        }
}
// bias + ELU of one accumulator tile (units k0 + 4 g + {0..3} of the NEXT layer's K, batch row u of row tile rt), split into
// pieces and written where the next layer's B fragments are read: k-block k >> 5, lane slot ((k & 31) >> 3) * 16 + u,
// bytes 8 ((k >> 2) & 1) .. + 7 of the slot's 16
template <int KBN>
__device__ __forceinline__ void mlp_split_store(uint4* act, int rt, int k0, int lane, f32x4_t a, float4 bb, float alpha) {
    const int u = lane & 15, g = lane >> 4;
    const int k = k0 + 4 * g;
    const float v[4] = {elu1(a[0] + bb.x, alpha), elu1(a[1] + bb.y, alpha), elu1(a[2] + bb.z, alpha), elu1(a[3] + bb.w, alpha)};
    uint2 pc[3];
    split3_bf16x4(v, pc[0], pc[1], pc[2]);
    const int kb = k >> 5, slot = ((k & 31) >> 3) * 16 + u, half = (k >> 2) & 1;
#pragma unroll
    for (int p = 0; p < 3; ++p)
        reinterpret_cast<uint2*>(act + ((rt * KBN + kb) * 3 + p) * 64 + slot)[half] = pc[p];
}
template <int RT, int NT, bool DUAL = false>
__global__ __launch_bounds__(256) void mlp3_elu_split_kernel(long long n, float* __restrict__ x, long long ldx,
                                                             const float* __restrict__ raw, int F_in,
                                                             const double* __restrict__ mean, const double* __restrict__ var,
                                                             float eps, float clip, const uint4* __restrict__ wt,
                                                             const float* __restrict__ b1, const float* __restrict__ b2,
                                                             const float* __restrict__ b3, float alpha, float* fin_meter,
                                                             float fin_max_size, long long* fin_counter,
                                                             const float* fin_partial, int fin_blocks) {
    if (fin_meter && blockIdx.x == 0) rollout_finalize_body(fin_meter, fin_max_size, fin_counter, fin_partial, fin_blocks);
    constexpr int R = 16 * RT;
    constexpr int D = RT >= 4 ? 1 : (RT == 2 ? 2 : 3);       // k-blocks of weight fragments in flight ahead of the matrix loop
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw_split[];
    // activations as B fragments: A0 (observations, 1 k-block) and A2 (layer-2 output, 4 k-blocks) share the front region
    // (A0 is dead once layer 1 is through), A1 (layer-1 output, 8 k-blocks) sits behind it
    uint4* act02 = reinterpret_cast<uint4*>(lds_raw_split);
    uint4* act1 = act02 + RT * 4 * 3 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u = lane & 15, g = lane >> 4;
    const long long row_base = (long long)blockIdx.x * R;
    MLP_STAMP(0)
    // this wave's weight fragments; layer 1 and the head of layer 2 are requested before anything else
    const uint4* a1_src = wt + (long long)(wave * 1 * 4 * 3) * 64 + lane;
    const uint4* a2_src = wt + (long long)(MLP_SPLIT_L2_BASE + wave * 8 * 2 * 3) * 64 + lane;
    const uint4* a3_src = wt + (long long)(MLP_SPLIT_L3_BASE + wave * 4 * 1 * 3) * 64 + lane;
    // the observation loads go first (everything waits for them: one memory round trip), the weight requests behind
    const bool obs_thread = tid < 4 * R;
    const int rl = tid >> 2, gq = tid & 3;
    const long long row = row_base + rl;
    // (branch-free: clamped addresses, every load in flight at once -- per-column `if`s compile to one round trip each; the
    // float64 statistics -> (mean, sqrt(var + eps)) by 32 threads through LDS, as in mlp3_elu_mfma_kernel)
    __shared__ float stat_m[32], stat_sd[32];
    double stat_mean = 0.0, stat_var = 1.0;
    if (tid < 32) {
        stat_mean = mean[min(tid, F_in - 1)];
        stat_var = var[min(tid, F_in - 1)];
    }
    float rv[8];
    const float* rrow = raw + (obs_thread ? row : row_base) * F_in;
#pragma unroll
    for (int i = 0; i < 8; ++i) rv[i] = rrow[min(8 * gq + i, F_in - 1)];
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler moved the 25 weight-fragment loads in front of them otherwise)
    MlpSplitRing<4, 1> ring1;
    MlpSplitRing<2, D> ring2;
    MlpSplitRing<1, D> ring3;
    ring1.issue(a1_src, 0);
    // (the biases of this lane's units, all three layers: requested here, not in front of the epilogues)
    float4 bb1[4], bb2[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) bb1[t] = ld4(b1 + 64 * wave + 16 * t + 4 * g);
#pragma unroll
    for (int t = 0; t < 2; ++t) bb2[t] = ld4(b2 + 32 * wave + 16 * t + 4 * g);
    const float4 bb = ld4(b3 + 16 * wave + 4 * g);
    // ---- normalised observations: thread (row, 8-column group) -> the LSTM operand's observation block (fp32) and A0
    // (computed AND written to LDS by every thread -- threads without a row write into the still unused A1 region: with
    // every use under `if (obs_thread)` the compiler sinks the loads into the branch, behind the weight requests)
    if (tid < 32) {
        // same arithmetic as normalize_obs_kernel: statistics cast to float first
        stat_m[tid] = (float)stat_mean;
        stat_sd[tid] = sqrtf((float)stat_var + eps);
    }
    __syncthreads();
    {
        float y[8];
        const float4 m0 = ld4(stat_m + 8 * gq), m1 = ld4(stat_m + 8 * gq + 4), d0 = ld4(stat_sd + 8 * gq), d1 = ld4(stat_sd + 8 * gq + 4);
        const float ms[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const float sds[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = fminf(fmaxf((rv[i] - ms[i]) / sds[i], -clip), clip);
            y[i] = 8 * gq + i < F_in ? v : 0.0f;
        }
        uint4 pc[3];
        split3_bf16x8(y, pc[0], pc[1], pc[2]);
        if (obs_thread) {
            st4(x + row * ldx + 64 + 8 * gq, make_float4(y[0], y[1], y[2], y[3]));
            st4(x + row * ldx + 64 + 8 * gq + 4, make_float4(y[4], y[5], y[6], y[7]));
        }
        uint4* dst = obs_thread ? act02 + ((rl >> 4) * 3) * 64 + gq * 16 + (rl & 15) : act1 + tid;
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p * 64] = pc[p];
    }
    __syncthreads();
    // (layer 2's first weight fragments have all of layer 1 to arrive)
#pragma unroll
    for (int d = 0; d < D; ++d) ring2.issue(a2_src, d);
    MLP_STAMP(1)
    // ---- layer 1: units 64 wave + 16 t + ...
    {
        f32x4_t acc[4][RT];
        mlp_split_layer<4, 1, RT, NT, 1, DUAL>(ring1, a1_src, act02 + lane, acc);
        MLP_STAMP(2)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) mlp_split_store<8>(act1, rt, 64 * wave + 16 * t, lane, acc[t][rt], bb1[t], alpha);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) ring3.issue(a3_src, d);
    __syncthreads();
    MLP_STAMP(3)
    // ---- layer 2: units 32 wave + 16 t + ...
    {
        f32x4_t acc[2][RT];
        mlp_split_layer<2, 8, RT, NT, D, DUAL>(ring2, a2_src, act1 + lane, acc);
        MLP_STAMP(4)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) mlp_split_store<4>(act02, rt, 32 * wave + 16 * t, lane, acc[t][rt], bb2[t], alpha);
    }
    __syncthreads();
    MLP_STAMP(5)
    // ---- layer 3: units 16 wave + 4 g + {0..3} of batch row u
    {
        f32x4_t acc[1][RT];
        mlp_split_layer<1, 4, RT, NT, D, DUAL>(ring3, a3_src, act02 + lane, acc);
        MLP_STAMP(6)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4_t a = acc[0][rt];
            st4(x + (row_base + 16 * rt + u) * ldx + 16 * wave + 4 * g,
                make_float4(elu1(a[0] + bb.x, alpha), elu1(a[1] + bb.y, alpha), elu1(a[2] + bb.z, alpha),
                            elu1(a[3] + bb.w, alpha)));
        }
    }
    MLP_STAMP(7)
}

// W1 [256][F_in] (row stride ldw1, columns beyond F_in read as zero), W2 [128][256], W3 [64][128] fp32 -> the fragments of
// bf16 pieces mlp3_elu_split_kernel reads: one thread per (layer, wave, k-block, tile, lane) = 8 consecutive k of one unit
__global__ void mlp3_tile_weights_split_kernel(const float* __restrict__ w1, long long ldw1, int F_in,
                                               const float* __restrict__ w2, long long ldw2, const float* __restrict__ w3,
                                               long long ldw3, uint4* __restrict__ dst) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (MLP_SPLIT_FRAGS / 3) * 64) return;
    const int lane = q & 63;
    int trip = q >> 6;                                  // (wave, k-block, tile) triple of pieces, layers back to back
    const float* w;
    long long ldw;
    int tiles, kbs, kmax, upw;                          // unit tiles per wave, k-blocks, valid k, units per wave
    if (trip < 16) { w = w1; ldw = ldw1; tiles = 4; kbs = 1; kmax = F_in; upw = 64; }
    else if (trip < 16 + 64) { trip -= 16; w = w2; ldw = ldw2; tiles = 2; kbs = 8; kmax = 256; upw = 32; dst += MLP_SPLIT_L2_BASE * 64; }
    else { trip -= 80; w = w3; ldw = ldw3; tiles = 1; kbs = 4; kmax = 128; upw = 16; dst += MLP_SPLIT_L3_BASE * 64; }
    const int t = trip % tiles, kb = (trip / tiles) % kbs, wave = trip / (tiles * kbs);
    const int unit = upw * wave + 16 * t + (lane & 15), k0 = 32 * kb + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (k0 + e < kmax) ? w[(long long)unit * ldw + k0 + e] : 0.0f;
    uint4 hi, mid, lo;
    split3_bf16x8(v, hi, mid, lo);
    uint4* d = dst + (long long)(trip * 3) * 64 + lane;
    d[0] = hi;
    d[64] = mid;
    d[128] = lo;
}

}  // namespace

// One zeroed ticket word per (device, stream, kernel family) for the "last workgroup to finish" elections of the loss
// and Adam kernels, so that launches in flight on different streams never share a ticket.  The words come from a
// per-device pool that is allocated and zeroed ONCE (first use on that device, or vine_ppo_runtime_init): handing a new
// stream its slot later needs no allocation and no memset, so it is legal inside a stream capture (torch captures on a
// side stream the library has not seen before).  Returns nullptr when the pool is exhausted or cannot be allocated
// (the wrappers then report VINE_ERR_DEVICE).
#include <mutex>
namespace {
// hipFuncAttributeMaxDynamicSharedMemorySize applies to the CURRENT device: remember the size raised per (kernel,
// device) -- not once per process (a process driving a second GPU launched without the attribute) -- under a
// mutex.  Not a stream operation: legal inside a capture.
struct DynLdsEntry { const void* fn; int dev; size_t bytes; };
DynLdsEntry g_dyn_lds[256];
int g_dyn_lds_used = 0;
std::mutex g_dyn_lds_mutex;
bool ensure_dyn_lds(const void* fn, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_dyn_lds_mutex);
    DynLdsEntry* e = nullptr;
    for (int i = 0; i < g_dyn_lds_used; ++i)
        if (g_dyn_lds[i].fn == fn && g_dyn_lds[i].dev == dev) { e = &g_dyn_lds[i]; break; }
    if (e && e->bytes >= bytes) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    if (!e && g_dyn_lds_used < 256) { e = &g_dyn_lds[g_dyn_lds_used++]; e->fn = fn; e->dev = dev; e->bytes = 0; }
    if (e) e->bytes = bytes;        // (table full: the attribute is simply set again next time)
    return true;
}
}  // namespace
#define TICKET_SUB_GROUPS 8
#define TICKET_SUB_PITCH 32        // words: the group tickets of a stream sit 128 B apart
namespace {
enum { TICKET_LOSS = 0, TICKET_ADAM = 1, TICKET_KINDS = 2, TICKET_MAX_STREAMS = 64, TICKET_MAX_DEVICES = 16 };
struct TicketPool { unsigned int* words; unsigned int* sub; void* stream[TICKET_MAX_STREAMS]; int used; };
TicketPool g_ticket_pool[TICKET_MAX_DEVICES];
std::mutex g_ticket_mutex;
unsigned int* ticket_slot(void* stream, int kind) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TICKET_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_ticket_mutex);
    TicketPool& P = g_ticket_pool[dev];
    if (!P.words) {
        unsigned int* w = nullptr;
        const size_t bytes = (size_t)TICKET_MAX_STREAMS * TICKET_KINDS * sizeof(unsigned int);
        if (hipMalloc(&w, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(w, 0, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w); return nullptr; }
        // first-level tickets of two-level elections (Adam): TICKET_SUB_GROUPS words per stream
        unsigned int* sb = nullptr;
        const size_t sbytes = (size_t)TICKET_MAX_STREAMS * TICKET_SUB_GROUPS * TICKET_SUB_PITCH * sizeof(unsigned int);
        if (hipMalloc(&sb, sbytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w); return nullptr; }
        if (hipMemset(sb, 0, sbytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(w); (void)hipFree(sb); return nullptr; }
        P.words = w;
        P.sub = sb;
        P.used = 0;
    }
    for (int i = 0; i < P.used; ++i)
        if (P.stream[i] == stream) return P.words + i * TICKET_KINDS + kind;
    if (P.used == TICKET_MAX_STREAMS) return nullptr;
    P.stream[P.used] = stream;
    return P.words + (P.used++) * TICKET_KINDS + kind;
}
// the group tickets that belong to a slot returned by ticket_slot (same device, same stream)
unsigned int* ticket_sub_of(const unsigned int* slot) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TICKET_MAX_DEVICES) return nullptr;
    TicketPool& P = g_ticket_pool[dev];
    if (!P.words || !P.sub || slot < P.words) return nullptr;
    const long long i = (slot - P.words) / TICKET_KINDS;
    return i < TICKET_MAX_STREAMS ? P.sub + i * TICKET_SUB_GROUPS * TICKET_SUB_PITCH : nullptr;
}
}  // namespace

extern "C" {

int vine_lstm_cell_forward(int64_t B, int64_t H, const float* igates, int64_t ig_stride, const float* hgates,
                           const float* bias, const float* c_prev, const uint8_t* done, int64_t done_stride,
                           float* h_out, int64_t h_stride, float* c_out, void* gates_act, void* hp_next,
                           const uint8_t* done_next, int64_t done_next_stride, int32_t hp_bf16, int64_t hp_stride,
                           void* stream) {
    if (hp_stride <= 0) hp_stride = h_stride;
    if (B <= 0 || H <= 0 || (H & 3) || (ig_stride & 3) || (h_stride & 3) || (hp_stride & 3) || !igates || !bias ||
        !c_prev || !h_out || !c_out)
        return VINE_ERR_INVALID_ARG;
    const int threads = 256;
    const dim3 grid(grid_for(B * (H / 4), threads));
    if (hp_bf16)
        hipLaunchKernelGGL(lstm_fwd_kernel<lp16_t>, grid, dim3(threads), 0, (hipStream_t)stream, (long long)B, (int)H,
                           igates, (long long)ig_stride, hgates, bias, c_prev, done, (long long)done_stride, h_out,
                           (long long)h_stride, c_out, (lp16_t*)gates_act, (lp16_t*)hp_next, done_next,
                           (long long)done_next_stride,
                           (long long)hp_stride);
    else
        hipLaunchKernelGGL(lstm_fwd_kernel<float>, grid, dim3(threads), 0, (hipStream_t)stream, (long long)B, (int)H,
                           igates, (long long)ig_stride, hgates, bias, c_prev, done, (long long)done_stride, h_out,
                           (long long)h_stride, c_out, (float*)gates_act, (float*)hp_next, done_next,
                           (long long)done_next_stride,
                           (long long)hp_stride);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_lstm_step_mfma(int64_t B, int64_t H, int64_t K, const void* A, int64_t lda, const void* A2, int64_t lda2,
                        int64_t K1, const void* W, int64_t ldw, const float* igates, int64_t ig_stride,
                        const float* bias, const float* c_prev, const uint8_t* done, int64_t done_stride, float* h_out,
                        int64_t h_stride, float* c_out, void* gates_act, void* hp_next, const uint8_t* done_next,
                        int64_t done_next_stride, int64_t hp_stride, void* stream) {
    if (hp_stride <= 0) hp_stride = h_stride;
    if (!A2) K1 = 0;
    if (B <= 0 || H <= 0 || K <= 0 || !A || !W || !bias || !c_prev || !h_out || !c_out || (lda & 7) || (ldw & 7) ||
        (ig_stride & 3) || (h_stride & 3) || (hp_stride & 3) || (A2 && ((lda2 & 7) || K1 <= 0 || K1 >= K)))
        return VINE_ERR_INVALID_ARG;
    if ((B & 63) || (H & 15) || (K & 31) || (K1 & 31) || K > LSTM_MFMA_MAX_K) return VINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(B / 64), (unsigned)(H / 16)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)64 * (K + 8) * sizeof(lp16_t);            // 33 KB at K = 256, 45 KB at K = 352
#define VINE_LSTM_MFMA(KS, KS1)                                                                                         \
    do {                                                                                                                \
        if (lds > 65536 && /* K = 512 needs more than the 64 KB default */                                              \
            !ensure_dyn_lds(reinterpret_cast<const void*>(&lstm_step_mfma_kernel<KS, KS1>), lds))                       \
            return VINE_ERR_DEVICE;                                                                                     \
        hipLaunchKernelGGL((lstm_step_mfma_kernel<KS, KS1>), grid, block, lds, s, (long long)B, (int)H,                 \
                           (const lp16_t*)A, (long long)lda, (const lp16_t*)A2, (long long)lda2, (const lp16_t*)W,      \
                           (long long)ldw, igates, (long long)ig_stride, bias, c_prev, done, (long long)done_stride,    \
                           h_out, (long long)h_stride, c_out, (lp16_t*)gates_act, (lp16_t*)hp_next, done_next,          \
                           (long long)done_next_stride, (long long)hp_stride);                                          \
    } while (0)
    const int ks = (int)(K / 32), ks1 = (int)(K1 / 32);
    static const bool use64 = getenv("VINE_LSTM_MFMA_SLAB") == nullptr;       // resident-slab kernel for A/B runs
    // (measured at B = 8192: K = 352 two-source 28.9 us streamed vs 32.3 us slab; K = 256 + igates 35.5 vs 29.8: the
    //  streamed kernel only takes the K = 352 shapes the update and the rollout actually use)
    if (use64 && (H & 63) == 0 && ks == 11 && (ks1 == 0 || ks1 == 3)) {
        const dim3 grid64((unsigned)(B / 64), (unsigned)(H / 64));
#define VINE_LSTM_MFMA64(KS, KS1)                                                                                       \
        hipLaunchKernelGGL((lstm_step_mfma64_kernel<KS, KS1>), grid64, block, 0, s, (long long)B, (int)H,               \
                           (const lp16_t*)A, (long long)lda, (const lp16_t*)A2, (long long)lda2, (const lp16_t*)W,      \
                           (long long)ldw, igates, (long long)ig_stride, bias, c_prev, done, (long long)done_stride,    \
                           h_out, (long long)h_stride, c_out, (lp16_t*)gates_act, (lp16_t*)hp_next, done_next,          \
                           (long long)done_next_stride, (long long)hp_stride)
        if (ks1 == 0) VINE_LSTM_MFMA64(11, 0);
        else VINE_LSTM_MFMA64(11, 3);
#undef VINE_LSTM_MFMA64
        return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
    }
    if (ks1 == 0) {
        switch (ks) {
            case 4: VINE_LSTM_MFMA(4, 0); break;
            case 8: VINE_LSTM_MFMA(8, 0); break;
            case 9: VINE_LSTM_MFMA(9, 0); break;
            case 10: VINE_LSTM_MFMA(10, 0); break;
            case 11: VINE_LSTM_MFMA(11, 0); break;
            case 12: VINE_LSTM_MFMA(12, 0); break;
            case 16: VINE_LSTM_MFMA(16, 0); break;
            default: return VINE_ERR_UNSUPPORTED;
        }
    } else if (ks - ks1 == 8) {          // [x (32 * ks1 columns) | h (256)]: the update's two-operand form
        switch (ks1) {
            case 1: VINE_LSTM_MFMA(9, 1); break;
            case 2: VINE_LSTM_MFMA(10, 2); break;
            case 3: VINE_LSTM_MFMA(11, 3); break;
            case 4: VINE_LSTM_MFMA(12, 4); break;
            default: return VINE_ERR_UNSUPPORTED;
        }
    } else {
        return VINE_ERR_UNSUPPORTED;
    }
#undef VINE_LSTM_MFMA
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

#ifdef SEQ_TIMING
int vine_debug_seq_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(seq_t), sizeof(unsigned long long) * 256 * 8 * 16) == hipSuccess ? 0 : -1;
}
#endif

int vine_lstm_tile_weights(int64_t H, int64_t K, const void* src, int64_t ld, int32_t transposed, void* dst, void* stream) {
    if (!src || !dst || H <= 0 || K <= 0 || ld <= 0) return VINE_ERR_INVALID_ARG;
    if (H != SEQ_H || (K & 31) || (!transposed && (ld < K || (ld & 7))) || (transposed && ld < H)) return VINE_ERR_UNSUPPORTED;
    const int ksteps = (int)(K / 32), nj = transposed ? 2 : 8;
    const long long chunks = (long long)(H / 32) * ksteps * nj * 64;
    hipLaunchKernelGGL(lstm_tile_weights_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const lp16_t*)src, (long long)ld, (int)H, ksteps, nj, (int)transposed, (lp16_t*)dst);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

// experiment knob of the persistent kernels, VINE_SEQ_ABLATE: bit 0 = no global stores, bit 1 = no weight reloads
// (timing ablations only -- results are wrong; profiles/r02/lstm_seq_ablation.txt)
static int seq_ablate() {
    static int ab = -1;
    if (ab < 0) { const char* e = getenv("VINE_SEQ_ABLATE"); ab = e ? atoi(e) : 0; }
    return ab;
}

int vine_lstm_seq_forward_mfma(int64_t B, int64_t T, int64_t H, int64_t KX, const void* x, int64_t ldx, void* hp,
                               int64_t hp_stride, const void* w_tiled, const float* bias, const float* c0,
                               const uint8_t* done, void* h_out, void* c_all, void* gates, int32_t c_bf16, float* c_last,
                               const float* h0, void* stream) {
    const bool h16 = (c_bf16 & 2) != 0;      // bit 1: "h once" (h_out = 16-bit unmasked [B, T + 1, H]; hp unused)
    c_bf16 &= 1;
    if (B <= 0 || T <= 0 || !x || (!hp && !h16) || !w_tiled || !bias || !c0 || !h_out || !c_all || ldx < KX || (ldx & 7) ||
        (!h16 && (hp_stride < T * H || (hp_stride & 7))) || (c_bf16 && !c_last) || (h16 && !h0))
        return VINE_ERR_INVALID_ARG;
    if ((B % SEQ_ROWS) || H != SEQ_H || T > 8 || (KX != 32 && KX != 64 && KX != 96 && KX != 128)) return VINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(B / SEQ_ROWS)), block(512);
    hipStream_t s = (hipStream_t)stream;
    const int ablate = seq_ablate();
    static const bool wcache_on = [] { const char* e = getenv("VINE_SEQ_FWD_WCACHE"); return !(e && e[0] == '0'); }();      // A/B knob
#define VINE_SEQ_FWD_T(KS1, RING, CT, HT)                                                                               \
    do {                                                                                                                \
        const size_t rows_ = 4 * SEQ_H * sizeof(float) + ((size_t)2 * SEQ_ROWS * (SEQ_H + LDS_SKEW) +                    \
                                                          (size_t)T * SEQ_ROWS * (32 * KS1 + LDS_SKEW)) * sizeof(lp16_t); \
        /* the LDS weight cache (SEQ_FWD_NC fragments per wave = 88 KB) when the operand rows leave room for it */        \
        const bool cache_ = wcache_on && T > 1 && rows_ + (size_t)8 * SEQ_FWD_NC * 1024 <= 160 * 1024;                   \
        const size_t lds_ = rows_ + (cache_ ? (size_t)8 * SEQ_FWD_NC * 1024 : 0);                                        \
        if (cache_) {                                                                                                   \
            if (!ensure_dyn_lds(reinterpret_cast<const void*>(&lstm_seq_fwd_kernel<KS1, RING, CT, HT, SEQ_FWD_NC>), lds_)) \
                return VINE_ERR_DEVICE;                                                                                 \
            hipLaunchKernelGGL((lstm_seq_fwd_kernel<KS1, RING, CT, HT, SEQ_FWD_NC>), grid, block, lds_, s, (int)T, (long long)B, \
                               (const lp16_t*)x, (long long)ldx, (lp16_t*)hp, (long long)hp_stride,                     \
                               (const uint4*)w_tiled, bias, c0, done, (HT*)h_out, (CT*)c_all, (lp16_t*)gates, ablate,   \
                               c_last, h0);                                                                             \
        } else {                                                                                                        \
            if (!ensure_dyn_lds(reinterpret_cast<const void*>(&lstm_seq_fwd_kernel<KS1, RING, CT, HT, 0>), lds_))       \
                return VINE_ERR_DEVICE;                                                                                 \
            hipLaunchKernelGGL((lstm_seq_fwd_kernel<KS1, RING, CT, HT, 0>), grid, block, lds_, s, (int)T, (long long)B, \
                               (const lp16_t*)x, (long long)ldx, (lp16_t*)hp, (long long)hp_stride,                     \
                               (const uint4*)w_tiled, bias, c0, done, (HT*)h_out, (CT*)c_all, (lp16_t*)gates, ablate,   \
                               c_last, h0);                                                                             \
        }                                                                                                               \
    } while (0)
#define VINE_SEQ_FWD(KS1, RING)                                                                                         \
    {                                                                                                                   \
        if (h16 && !c_bf16) return VINE_ERR_UNSUPPORTED;                                                                \
        if (h16) VINE_SEQ_FWD_T(KS1, RING, lp16_t, lp16_t);                                                             \
        else if (c_bf16) VINE_SEQ_FWD_T(KS1, RING, lp16_t, float);                                                      \
        else VINE_SEQ_FWD_T(KS1, RING, float, float);                                                                   \
    }
    switch (KX / 32) {
        case 1: VINE_SEQ_FWD(1, 18) break;      // 72 fragments per step (24 in flight spills a few registers)
        case 2: VINE_SEQ_FWD(2, 20) break;      // 80
        case 3: VINE_SEQ_FWD(3, 22) break;      // 88: the update's [x (92 + 4 pad) | h] operand
        default: VINE_SEQ_FWD(4, 24) break;     // 96
    }
#undef VINE_SEQ_FWD
#undef VINE_SEQ_FWD_T
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_lstm_seq_backward_mfma(int64_t B, int64_t T, int64_t H, const void* g_out, const void* w_hh_tiled,
                                const void* gates, const void* c_all, const float* c0, const uint8_t* done,
                                void* dgates, float* bias_partial, int32_t c_bf16, const float* c_last, int32_t g_bf16,
                                void* stream) {
    if (B <= 0 || T <= 0 || !g_out || !w_hh_tiled || !gates || !c_all || !c0 || !dgates || (c_bf16 && !c_last))
        return VINE_ERR_INVALID_ARG;
    if ((B % SEQ_ROWS) || H != SEQ_H || T > 8) return VINE_ERR_UNSUPPORTED;
    constexpr int RING = SEQ_BWD_RING;
    const size_t lds = (size_t)2 * SEQ_ROWS * (4 * SEQ_H + LDS_SKEW) * sizeof(lp16_t);   // 130 KiB: one workgroup per CU
    const int ablate = seq_ablate();
#define VINE_SEQ_BWD(CT, GT)                                                                                              \
    {                                                                                                                     \
        if (!ensure_dyn_lds(reinterpret_cast<const void*>(&lstm_seq_bwd_kernel<RING, CT, GT>), lds))                      \
            return VINE_ERR_DEVICE;                                                                                       \
        hipLaunchKernelGGL((lstm_seq_bwd_kernel<RING, CT, GT>), dim3((unsigned)(B / SEQ_ROWS)), dim3(512), lds,           \
                           (hipStream_t)stream, (int)T, (long long)B, (const GT*)g_out, (const uint4*)w_hh_tiled,         \
                           (const lp16_t*)gates, (const CT*)c_all, c0, done, (lp16_t*)dgates, bias_partial, ablate,       \
                           c_last);                                                                                       \
    }
    if (c_bf16 && g_bf16) VINE_SEQ_BWD(lp16_t, lp16_t)
    else if (c_bf16) VINE_SEQ_BWD(lp16_t, float)
    else if (g_bf16) VINE_SEQ_BWD(float, lp16_t)
    else VINE_SEQ_BWD(float, float)
#undef VINE_SEQ_BWD
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_linear_elu_mfma(int64_t n, int64_t N, int64_t K, const void* A, int64_t lda, const void* W, int64_t ldw,
                         const float* bias, float alpha, void* out, int64_t out_stride, void* stream) {
    if (n <= 0 || N <= 0 || K <= 0 || !A || !W || !bias || !out || (lda & 7) || (ldw & 7) || (out_stride & 3))
        return VINE_ERR_INVALID_ARG;
    if ((n & 63) || (N & 63) || (K & 31) || K > 256) return VINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(n / 64), (unsigned)(N / 64)), block(256);
    const size_t lds = (size_t)64 * (K + 8) * sizeof(lp16_t);
    hipStream_t s = (hipStream_t)stream;
#define VINE_LIN_MFMA(KS)                                                                                             \
    hipLaunchKernelGGL(linear_elu_mfma_kernel<KS>, grid, block, lds, s, (long long)n, (int)N, (const lp16_t*)A,       \
                       (long long)lda, (const lp16_t*)W, (long long)ldw, bias, alpha, (lp16_t*)out, (long long)out_stride)
    switch (K / 32) {
        case 1: VINE_LIN_MFMA(1); break;
        case 2: VINE_LIN_MFMA(2); break;
        case 4: VINE_LIN_MFMA(4); break;
        case 8: VINE_LIN_MFMA(8); break;
        default: return VINE_ERR_UNSUPPORTED;
    }
#undef VINE_LIN_MFMA
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

static int copy_batch_pack(CopyBatchArgs& b, int& blocks, int32_t njobs, const int32_t* op, const int32_t* elem,
                           const void* const* src, const void* const* src2, void* const* dst, const int64_t* rows,
                           const int64_t* cols, const int64_t* src_stride, const int64_t* dst_stride, const int64_t* aux);

int vine_mlp3_elu_mfma(int64_t n, void* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean, const double* var,
                       float eps, float clip, const void* w1p, const float* b1, int64_t C1, const void* w2, int64_t ldw2,
                       const float* b2, int64_t C2, const void* w3, int64_t ldw3, const float* b3, int64_t C3, float alpha,
                       void* act1, void* act2, void* out, int64_t out_stride, void* stream) {
    return vine_mlp3_elu_mfma_prep(n, x, ldx, raw, F_in, mean, var, eps, clip, w1p, 32, b1, C1, w2, ldw2, b2, C2, w3, ldw3, b3, C3,
                                   alpha, act1, act2, out, out_stride, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, nullptr, nullptr, stream);
}

int vine_mlp3_elu_mfma_prep(int64_t n, void* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean,
                            const double* var, float eps, float clip, const void* w1p, int64_t ldw1, const float* b1,
                            int64_t C1, const void* w2, int64_t ldw2, const float* b2, int64_t C2, const void* w3, int64_t ldw3,
                            const float* b3, int64_t C3, float alpha, void* act1, void* act2, void* out, int64_t out_stride,
                            int32_t njobs, const int32_t* op, const int32_t* elem, const void* const* src,
                            const void* const* src2, void* const* dst, const int64_t* rows, const int64_t* cols,
                            const int64_t* src_stride, const int64_t* dst_stride, const int64_t* aux, void* stream) {
    if (n <= 0 || !x || !w1p || !b1 || !w2 || !b2 || !w3 || !b3 || !out || (ldx & 7) || (ldw2 & 7) || (ldw3 & 7) ||
        (out_stride & 3) || ldx < 32 || ldw2 < C1 || ldw3 < C2 || ((uintptr_t)x & 15) ||
        ((uintptr_t)w1p & (ldw1 == 32 ? 15 : 3)) || ldw1 <= 0 || ldw1 > 32 || (ldw1 & 1) ||
        ((uintptr_t)w2 & 15) || ((uintptr_t)w3 & 15) || ((uintptr_t)out & 7) || (raw && (!mean || !var || F_in <= 0 || F_in > 32)))
        return VINE_ERR_INVALID_ARG;
    if (C1 != 256 || C2 != 128 || C3 != 64 || (n & 63)) return VINE_ERR_UNSUPPORTED;
    CopyBatchArgs side;
    side.njobs = 0;
    int side_blocks = 0;
    if (njobs > 0) {
        const int rc = copy_batch_pack(side, side_blocks, njobs, op, elem, src, src2, dst, rows, cols, src_stride, dst_stride, aux);
        if (rc != VINE_OK) return rc;
        // the side job keeps every virtual block's elements in registers: only the moves it covers, and no more blocks
        // than MLP3_SIDE_ITEMS per 256 threads of this launch (the caller then runs vine_copy_batched itself)
        for (int k = 0; k < njobs; ++k)
            if (!side_job_supported(side.job[k])) return VINE_ERR_UNSUPPORTED;
        if ((long long)side_blocks > (long long)MLP3_SIDE_ITEMS * (n / 64)) return VINE_ERR_UNSUPPORTED;
    }
    const int threads = (n % 128 == 0 && n >= 32768) ? 512 : 256;        // 8 waves per CU when one round covers the chip
    const size_t lds = ((size_t)256 * (32 + LDS_SKEW) + 128 * (256 + LDS_SKEW) + 64 * (128 + LDS_SKEW)) * sizeof(lp16_t);
    if (!ensure_dyn_lds(threads == 512 ? reinterpret_cast<const void*>(&mlp3_elu_mfma_kernel<256, 128, 64, 8>)
                                       : reinterpret_cast<const void*>(&mlp3_elu_mfma_kernel<256, 128, 64, 4>), lds))
        return VINE_ERR_DEVICE;
#define VINE_MLP3(NW_)                                                                                                   \
    hipLaunchKernelGGL((mlp3_elu_mfma_kernel<256, 128, 64, NW_>), dim3((unsigned)(n / (16 * NW_))), dim3(64 * NW_), lds,  \
                       (hipStream_t)stream, (long long)n, (lp16_t*)x, (long long)ldx, raw, (int)F_in, mean, var, eps, clip, \
                       (const lp16_t*)w1p, b1, (const lp16_t*)w2, (long long)ldw2, b2, (const lp16_t*)w3, (long long)ldw3, \
                       b3, alpha, (lp16_t*)act1, (lp16_t*)act2, (lp16_t*)out, (long long)out_stride, (int)ldw1, side,     \
                       side_blocks)
    if (threads == 512) VINE_MLP3(8);
    else VINE_MLP3(4);
#undef VINE_MLP3
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_mlp3_bwd_elu_mfma(int64_t n, const void* dG, int64_t lddg, int64_t K0, const void* wt0, int64_t ldw0, const void* wt1,
                           int64_t ldw1, const void* wt2, int64_t ldw2, const void* a3, int64_t a3_stride, const void* a2,
                           const void* a1, int64_t C3, int64_t C2, int64_t C1, float alpha, void* gz3, void* gz2, void* gz1,
                           float* part3, float* part2, float* part1, void* stream) {
    if (n <= 0 || !dG || !wt0 || !wt1 || !wt2 || !a3 || !a2 || !a1 || !gz3 || !gz2 || !gz1 || !part3 || !part2 || !part1 ||
        (lddg & 7) || (ldw0 & 7) || (ldw1 & 7) || (ldw2 & 7) || (a3_stride & 7) || lddg < K0 || ldw0 < K0 || ldw1 < C3 ||
        ldw2 < C2 || ((uintptr_t)dG & 15) || ((uintptr_t)wt0 & 15) || ((uintptr_t)wt1 & 15) || ((uintptr_t)wt2 & 15) ||
        ((uintptr_t)a3 & 15) || ((uintptr_t)a2 & 15) || ((uintptr_t)a1 & 15) || ((uintptr_t)gz3 & 15) || ((uintptr_t)gz2 & 15) ||
        ((uintptr_t)gz1 & 15))
        return VINE_ERR_INVALID_ARG;
    if (C3 != 64 || C2 != 128 || C1 != 256 || K0 != 1024 || (n & 63)) return VINE_ERR_UNSUPPORTED;
    const int nw = (n % 128 == 0 && n >= 32768) ? 8 : 4;
    const size_t lds = ((size_t)2 * 64 * (128 + LDS_SKEW) + 128 * (64 + LDS_SKEW) + 256 * (128 + LDS_SKEW)) * sizeof(lp16_t) +
                       (size_t)nw * (64 + 128 + 256) * sizeof(float);
    if (!ensure_dyn_lds(nw == 8 ? reinterpret_cast<const void*>(&mlp3_bwd_elu_mfma_kernel<64, 128, 256, 8, 8>)
                                : reinterpret_cast<const void*>(&mlp3_bwd_elu_mfma_kernel<64, 128, 256, 8, 4>), 160 * 1024))
        return VINE_ERR_DEVICE;
#define VINE_MLP3B(NW_)                                                                                                   \
    hipLaunchKernelGGL((mlp3_bwd_elu_mfma_kernel<64, 128, 256, 8, NW_>), dim3((unsigned)(n / (16 * NW_))), dim3(64 * NW_), \
                       lds, (hipStream_t)stream, (long long)n, (const lp16_t*)dG, (long long)lddg, (const lp16_t*)wt0,     \
                       (long long)ldw0, (const lp16_t*)wt1, (long long)ldw1, (const lp16_t*)wt2, (long long)ldw2,          \
                       (const lp16_t*)a3, (long long)a3_stride, (const lp16_t*)a2, (const lp16_t*)a1, alpha, (lp16_t*)gz3, \
                       (lp16_t*)gz2, (lp16_t*)gz1, part3, part2, part1)
    if (nw == 8) VINE_MLP3B(8);
    else VINE_MLP3B(4);
#undef VINE_MLP3B
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_lstm_seq_backward_mlp3_mfma(int64_t B, int64_t T, int64_t H, const void* g_out, const void* w_hh_tiled,
                                     const void* gates, const void* c_all, const float* c0, const uint8_t* done,
                                     void* dgates, float* bias_partial, const float* c_last, const void* wt0, int64_t ldw0,
                                     const void* wt1, int64_t ldw1, const void* wt2, int64_t ldw2, const void* a3,
                                     int64_t a3_stride, const void* a2, const void* a1, float alpha, void* gz3, void* gz2,
                                     void* gz1, float* part3, float* part2, float* part1, void* stream) {
    if (B <= 0 || T <= 0 || !g_out || !w_hh_tiled || !gates || !c_all || !c0 || !dgates || !bias_partial || !c_last || !wt0 ||
        !wt1 || !wt2 || !a3 || !a2 || !a1 || !gz3 || !gz2 || !gz1 || !part3 || !part2 || !part1 || (ldw0 & 7) || (ldw1 & 7) ||
        (ldw2 & 7) || (a3_stride & 7) || ldw0 < 4 * SEQ_H || ldw1 < 64 || ldw2 < 128 || ((uintptr_t)dgates & 15) ||
        ((uintptr_t)wt0 & 15) || ((uintptr_t)wt1 & 15) || ((uintptr_t)wt2 & 15) || ((uintptr_t)a3 & 15) || ((uintptr_t)a2 & 15) ||
        ((uintptr_t)a1 & 15) || ((uintptr_t)gz3 & 15) || ((uintptr_t)gz2 & 15) || ((uintptr_t)gz1 & 15))
        return VINE_ERR_INVALID_ARG;
    // one workgroup = 32 sequences x T steps (the LSTM phase) = 128 rows (the MLP phase, 8 waves x 16 rows): T = 4 only;
    // 16-bit saved cell states and hidden-state gradient (the update's configuration)
    if ((B % SEQ_ROWS) || H != SEQ_H || T != 4) return VINE_ERR_UNSUPPORTED;
    constexpr int RING = SEQ_BWD_RING;
    const size_t lds_lstm = (size_t)2 * SEQ_ROWS * (4 * SEQ_H + LDS_SKEW) * sizeof(lp16_t);
    const size_t lds_mlp = ((size_t)2 * 64 * (128 + LDS_SKEW) + 128 * (64 + LDS_SKEW) + 256 * (128 + LDS_SKEW)) * sizeof(lp16_t) +
                           (size_t)8 * (64 + 128 + 256) * sizeof(float);
    const size_t lds = lds_lstm > lds_mlp ? lds_lstm : lds_mlp;
    if (!ensure_dyn_lds(reinterpret_cast<const void*>(&lstm_seq_bwd_mlp3_bwd_kernel<RING, lp16_t, lp16_t>), 160 * 1024))
        return VINE_ERR_DEVICE;
    hipLaunchKernelGGL((lstm_seq_bwd_mlp3_bwd_kernel<RING, lp16_t, lp16_t>), dim3((unsigned)(B / SEQ_ROWS)), dim3(512), lds,
                       (hipStream_t)stream, (int)T, (long long)B, (const lp16_t*)g_out, (const uint4*)w_hh_tiled,
                       (const lp16_t*)gates, (const lp16_t*)c_all, c0, done, (lp16_t*)dgates, bias_partial, seq_ablate(), c_last,
                       (const lp16_t*)wt0, (long long)ldw0, (const lp16_t*)wt1, (long long)ldw1, (const lp16_t*)wt2,
                       (long long)ldw2, (const lp16_t*)a3, (long long)a3_stride, (const lp16_t*)a2, (const lp16_t*)a1, alpha,
                       (lp16_t*)gz3, (lp16_t*)gz2, (lp16_t*)gz1, part3, part2, part1);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int64_t vine_trunk_args_size(void) { return (int64_t)sizeof(VineTrunkArgs); }

int vine_trunk_phases(const VineTrunkArgs* p, void* stream) {
    if (!p) return VINE_ERR_INVALID_ARG;
    const VineTrunkArgs& q = *p;
    const void* need[] = {q.x, q.w_tiled, q.bias, q.c0, q.h0, q.h_out, q.c_all, q.gates, q.c_last, q.ln_gamma, q.ln_beta, q.w_heads,
                          q.b_heads, q.logstd, q.actions, q.old_neglogp, q.advantages, q.old_values, q.returns, q.old_mu,
                          q.old_sigma, q.heads, q.d_out, q.ln_partial, q.loss_partial, q.stats, q.grad_logstd, q.w_hh_tiled,
                          q.dgates, q.bias_partial, q.wt0, q.wt1, q.wt2, q.a3, q.a2, q.a1, q.gz3, q.gz2, q.gz1, q.part3, q.part2,
                          q.part1};
    for (const void* v : need)
        if (!v) return VINE_ERR_INVALID_ARG;
    if (q.B <= 0 || ((q.grad_mu_bias == nullptr) != (q.grad_value_bias == nullptr)) || ((q.mu_store == nullptr) != (q.sigma_store == nullptr)) ||
        (q.ldx & 7) || (q.ldw0 & 7) || (q.ldw1 & 7) || (q.ldw2 & 7) || (q.a3_stride & 7) || q.ldw0 < 4 * SEQ_H || q.ldw1 < 64 ||
        q.ldw2 < 128)
        return VINE_ERR_INVALID_ARG;
    const void* al16[] = {q.x, q.w_tiled, q.h_out, q.c_all, q.gates, q.d_out, q.w_hh_tiled, q.dgates, q.wt0, q.wt1, q.wt2, q.a3, q.a2,
                          q.a1, q.gz3, q.gz2, q.gz1, q.c0, q.h0, q.c_last, q.bias};
    for (const void* v : al16)
        if ((uintptr_t)v & 15) return VINE_ERR_INVALID_ARG;
    // the update's default shapes only (the phases are instantiated for them); one workgroup = 32 sequences = 128 samples
    const long long n = q.B * q.T;
    if (q.T != 4 || (q.B % SEQ_ROWS) || q.ldx < 96 || n / 128 > VINE_PPO_LOSS_BLOCKS) return VINE_ERR_UNSUPPORTED;
    TrunkPhasesArgs a;
    a.B = q.B; a.T = (int)q.T;
    a.x = (const lp16_t*)q.x; a.ldx = q.ldx; a.w_tiled = (const uint4*)q.w_tiled; a.bias = q.bias; a.c0 = q.c0; a.h0 = q.h0;
    a.done = q.done; a.h_out = (lp16_t*)q.h_out; a.c_all = (lp16_t*)q.c_all; a.gates = (lp16_t*)q.gates; a.c_last = q.c_last;
    a.ablate = seq_ablate();
    a.gamma = q.ln_gamma; a.beta = q.ln_beta; a.eps = q.ln_eps; a.w = q.w_heads; a.wb = q.b_heads; a.logstd = q.logstd;
    a.actions = q.actions; a.old_neglogp = q.old_neglogp; a.adv = q.advantages; a.old_values = q.old_values; a.returns = q.returns;
    a.old_mu = q.old_mu; a.old_sigma = q.old_sigma; a.e_clip = q.e_clip; a.clip_value = q.clip_value; a.critic_coef = q.critic_coef;
    a.entropy_coef = q.entropy_coef; a.bounds_coef = q.bounds_coef; a.soft_bound = q.soft_bound;
    a.heads = q.heads; a.dx = (lp16_t*)q.d_out; a.ln_partial = q.ln_partial; a.loss_partial = q.loss_partial; a.stats = q.stats;
    a.grad_logstd = q.grad_logstd; a.grad_mu_bias = q.grad_mu_bias; a.grad_value_bias = q.grad_value_bias; a.kl_out = q.kl_out;
    a.logstd_grad_accum = q.logstd_grad_accum; a.mu_store = q.mu_store; a.sigma_store = q.sigma_store; a.loss_scale = q.loss_scale;
    a.found_inf = q.found_inf;
    a.w_hh_tiled = (const uint4*)q.w_hh_tiled; a.dG = (lp16_t*)q.dgates; a.bias_partial = q.bias_partial;
    a.wt0 = (const lp16_t*)q.wt0; a.ldw0 = q.ldw0; a.wt1 = (const lp16_t*)q.wt1; a.ldw1 = q.ldw1; a.wt2 = (const lp16_t*)q.wt2;
    a.ldw2 = q.ldw2; a.a3 = (const lp16_t*)q.a3; a.a3_stride = q.a3_stride; a.a2 = (const lp16_t*)q.a2; a.a1 = (const lp16_t*)q.a1;
    a.alpha = q.alpha; a.gz3 = (lp16_t*)q.gz3; a.gz2 = (lp16_t*)q.gz2; a.gz1 = (lp16_t*)q.gz1; a.part3 = q.part3; a.part2 = q.part2;
    a.part1 = q.part1;
    const size_t lds = 160 * 1024;      // the LSTM forward phase with its weight cache: 63 + 96 KB (the other phases need less)
    if (!ensure_dyn_lds(reinterpret_cast<const void*>(&trunk_phases_kernel), lds)) return VINE_ERR_DEVICE;
    hipLaunchKernelGGL(trunk_phases_kernel, dim3((unsigned)(q.B / SEQ_ROWS)), dim3(512), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_linear_bwd_elu_mfma(int64_t n, int64_t N, int64_t K, const void* G, int64_t ldg, const void* Wt, int64_t ldw,
                             const void* a, int64_t a_stride, float alpha, void* gz, int64_t gz_stride, float* partial,
                             void* stream) {
    if (n <= 0 || N <= 0 || K <= 0 || !G || !Wt || !a || !gz || (ldg & 7) || (ldw & 7) || (a_stride & 3) || (gz_stride & 3))
        return VINE_ERR_INVALID_ARG;
    if ((n & 63) || (N & 63) || (K != 64 && K != 128 && K != 256 && K != 512 && K != 1024)) return VINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(n / 64), (unsigned)(N / 64)), block(256);
    if (K >= 512) {          // long reduction: both operands stream in 128-wide k chunks
        hipStream_t s2 = (hipStream_t)stream;
        if (K == 512)
            hipLaunchKernelGGL(linear_bwd_elu_mfma_chunked_kernel<4>, grid, block, 0, s2, (long long)n, (int)N,
                               (const lp16_t*)G, (long long)ldg, (const lp16_t*)Wt, (long long)ldw, (const lp16_t*)a,
                               (long long)a_stride, alpha, (lp16_t*)gz, (long long)gz_stride, partial);
        else
            hipLaunchKernelGGL(linear_bwd_elu_mfma_chunked_kernel<8>, grid, block, 0, s2, (long long)n, (int)N,
                               (const lp16_t*)G, (long long)ldg, (const lp16_t*)Wt, (long long)ldw, (const lp16_t*)a,
                               (long long)a_stride, alpha, (lp16_t*)gz, (long long)gz_stride, partial);
        return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
    }
    size_t lds = (size_t)64 * (K + 8) * sizeof(lp16_t);
    if (lds < 64 * 65 * sizeof(float)) lds = 64 * 65 * sizeof(float);       // the column-sum stage reuses the slab
    hipStream_t s = (hipStream_t)stream;
#define VINE_LINB_MFMA(KS)                                                                                             \
    hipLaunchKernelGGL(linear_bwd_elu_mfma_kernel<KS>, grid, block, lds, s, (long long)n, (int)N, (const lp16_t*)G,    \
                       (long long)ldg, (const lp16_t*)Wt, (long long)ldw, (const lp16_t*)a, (long long)a_stride, alpha,  \
                       (lp16_t*)gz, (long long)gz_stride, partial)
    switch (K / 32) {
        case 2: VINE_LINB_MFMA(2); break;
        case 4: VINE_LINB_MFMA(4); break;
        case 8: VINE_LINB_MFMA(8); break;
        default: return VINE_ERR_UNSUPPORTED;
    }
#undef VINE_LINB_MFMA
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_lstm_cell_backward(int64_t B, int64_t H, const float* g_out, int64_t g_stride, const float* g_rec,
                            const float* dc_next, const uint8_t* done_next, int64_t done_next_stride,
                            const void* gates_act, const float* c_new, const float* c_prev, const uint8_t* done,
                            int64_t done_stride, void* dgates, int64_t dg_stride, float* dc_prev,
                            float* bias_partial, const float* bias_partial_prev, int32_t dgates_bf16, void* stream) {
    if (B <= 0 || H <= 0 || (H & 3) || (g_stride & 3) || (dg_stride & 3) || !g_out || !gates_act || !c_new || !c_prev ||
        !dgates || !dc_prev)
        return VINE_ERR_INVALID_ARG;
    const int threads = 256;
    if (bias_partial && (H > 1024 || threads % (int)(H / 4) != 0)) return VINE_ERR_UNSUPPORTED;
    // with partial sums the grid is fixed: the caller's buffer has VINE_PPO_PARTIAL_BLOCKS rows
    const int blocks = bias_partial ? VINE_PPO_PARTIAL_BLOCKS : grid_for(B * (H / 4), threads);
    if (dgates_bf16)
        hipLaunchKernelGGL(lstm_bwd_kernel<lp16_t>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (long long)B,
                           (int)H, g_out, (long long)g_stride, g_rec, dc_next, done_next, (long long)done_next_stride,
                           (const lp16_t*)gates_act, c_new, c_prev, done, (long long)done_stride, (lp16_t*)dgates,
                           (long long)dg_stride,
                           dc_prev, bias_partial, bias_partial_prev);
    else
        hipLaunchKernelGGL(lstm_bwd_kernel<float>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (long long)B,
                           (int)H, g_out, (long long)g_stride, g_rec, dc_next, done_next, (long long)done_next_stride,
                           (const float*)gates_act, c_new, c_prev, done, (long long)done_stride, (float*)dgates,
                           (long long)dg_stride,
                           dc_prev, bias_partial, bias_partial_prev);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_lstm_step_backward_mfma(int64_t B, int64_t H, const float* g_out, int64_t g_stride, const void* dgates_next,
                                 int64_t dgn_stride, const void* w_hh_t, int64_t ldw, const float* dc_next,
                                 const uint8_t* done_next, int64_t done_next_stride, const void* gates_act,
                                 const float* c_new, const float* c_prev, const uint8_t* done, int64_t done_stride,
                                 void* dgates, int64_t dg_stride, float* dc_prev, float* bias_partial,
                                 const float* bias_partial_prev, void* stream) {
    if (B <= 0 || H <= 0 || !g_out || !gates_act || !c_new || !c_prev || !dgates || !dc_prev ||
        (dgates_next != nullptr) != (w_hh_t != nullptr))
        return VINE_ERR_INVALID_ARG;
    if ((B & 63) || (H != 128 && H != 256) || (g_stride & 3) || (dg_stride & 3)) return VINE_ERR_UNSUPPORTED;
    if (dgates_next && ((dgn_stride & 7) || (ldw & 7) || ldw < 4 * H)) return VINE_ERR_INVALID_ARG;
    const dim3 grid((unsigned)(B / 64), (unsigned)(H / 64));
    hipStream_t s = (hipStream_t)stream;
#define VINE_LSTM_BWD_MFMA(NCH)                                                                                        \
    hipLaunchKernelGGL(lstm_bwd_mfma_kernel<NCH>, grid, dim3(256), 0, s, (long long)B, (int)H, g_out,                  \
                       (long long)g_stride, (const lp16_t*)dgates_next, (long long)dgn_stride, (const lp16_t*)w_hh_t,  \
                       (long long)ldw, dc_next, done_next, (long long)done_next_stride, (const lp16_t*)gates_act,      \
                       c_new, c_prev, done, (long long)done_stride, (lp16_t*)dgates, (long long)dg_stride, dc_prev,    \
                       bias_partial, bias_partial_prev)
    if (!dgates_next) VINE_LSTM_BWD_MFMA(0);
    else if (H == 256) VINE_LSTM_BWD_MFMA(8);
    else VINE_LSTM_BWD_MFMA(4);
#undef VINE_LSTM_BWD_MFMA
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_weight_grad_mfma(int64_t rows, int64_t M, int64_t Np, int64_t Nv, const void* dy, int64_t ldy, const void* x,
                          int64_t ldx, int64_t slices, float* part, void* stream) {
    if (rows <= 0 || M <= 0 || Np <= 0 || Nv <= 0 || Nv > Np || slices <= 0 || !dy || !x || !part || ldy < M || ldx < Np ||
        (ldy & 7) || (ldx & 7) || ((uintptr_t)dy & 15) || ((uintptr_t)x & 15))
        return VINE_ERR_INVALID_ARG;
    if ((M & 63) || rows % (slices * 32) || slices > 65535) return VINE_ERR_UNSUPPORTED;
    const int mt = (M & 127) ? 1 : 2;
    int nt;
    if (Np == 32) nt = 2;
    else if (Np == 96) nt = 6;
    else if ((Np & 127) == 0) nt = 8;
    else return VINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(M / (64 * mt)), (unsigned)(Np / (16 * nt)), (unsigned)slices);
    const int stages = (int)(rows / slices / 32);
    hipStream_t s = (hipStream_t)stream;
#define VINE_WGRAD(MT_, NT_)                                                                                          \
    hipLaunchKernelGGL((wgrad_mfma_kernel<MT_, NT_>), grid, dim3(256), 0, s, stages, (const lp16_t*)dy, (long long)ldy, \
                       (const lp16_t*)x, (long long)ldx, part, (int)M, (int)Nv)
    if (mt == 2) { if (nt == 2) VINE_WGRAD(2, 2); else if (nt == 6) VINE_WGRAD(2, 6); else VINE_WGRAD(2, 8); }
    else { if (nt == 2) VINE_WGRAD(1, 2); else if (nt == 6) VINE_WGRAD(1, 6); else VINE_WGRAD(1, 8); }
#undef VINE_WGRAD
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_weight_grad_group(int32_t nprob, const int64_t* rows, const int64_t* M, const void* const* dy, const int64_t* ldy,
                           const void* const* x1, const int64_t* ldx1, const int64_t* N1p, const int64_t* Nv1,
                           const void* const* x2, const int64_t* ldx2, const int64_t* N2p, const int64_t* Nv2, const int64_t* NT,
                           const int64_t* slices, float* const* part1, float* const* part2, void* stream) {
    if (nprob <= 0 || nprob > VINE_WGRAD_MAX_PROBLEMS || !rows || !M || !dy || !ldy || !x1 || !ldx1 || !N1p || !Nv1 || !x2 ||
        !ldx2 || !N2p || !Nv2 || !NT || !slices || !part1 || !part2)
        return VINE_ERR_INVALID_ARG;
    WgGroupArgs G;
    int blocks = 0;
    for (int k = 0; k < nprob; ++k) {
        const bool two = N1p[k] > 0;
        if (rows[k] <= 0 || M[k] <= 0 || slices[k] <= 0 || !dy[k] || !x2[k] || !part2[k] || ldy[k] < M[k] || ldx2[k] < N2p[k] ||
            (ldy[k] & 7) || (ldx2[k] & 7) || ((uintptr_t)dy[k] & 15) || ((uintptr_t)x2[k] & 15) || N2p[k] <= 0 || Nv2[k] <= 0 ||
            Nv2[k] > N2p[k] || N1p[k] < 0 ||
            (two && (!x1[k] || !part1[k] || ldx1[k] < N1p[k] || (ldx1[k] & 7) || ((uintptr_t)x1[k] & 15) || Nv1[k] <= 0 ||
                     Nv1[k] > N1p[k])))
            return VINE_ERR_INVALID_ARG;
        if ((NT[k] != 11 && NT[k] != 8 && NT[k] != 2) || (M[k] & 63) || (N1p[k] & 15) || ((N1p[k] + N2p[k]) % (16 * NT[k])) ||
            (slices[k] & 7) || rows[k] % (slices[k] * 32) || slices[k] > 8192)
            return VINE_ERR_UNSUPPORTED;
        WgProblem& P = G.p[k];
        P.dy = (const lp16_t*)dy[k]; P.ldy = ldy[k];
        P.x2 = (const lp16_t*)x2[k]; P.ldx2 = ldx2[k]; P.part2 = part2[k]; P.Nv2 = (int)Nv2[k];
        P.x1 = two ? (const lp16_t*)x1[k] : P.x2; P.ldx1 = two ? ldx1[k] : ldx2[k];
        P.part1 = two ? part1[k] : part2[k]; P.Nv1 = two ? (int)Nv1[k] : (int)Nv2[k];
        P.N1p = (int)N1p[k]; P.M = (int)M[k]; P.NT = (int)NT[k]; P.slices = (int)slices[k];
        P.mtiles = (int)(M[k] / 64); P.ntiles = (int)((N1p[k] + N2p[k]) / (16 * NT[k]));
        P.stages = (int)(rows[k] / slices[k] / 32);
        P.first_block = blocks;
        blocks += P.mtiles * P.ntiles * P.slices;
    }
    G.n = nprob;
    hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

// one (128 | 64) x 352 tile per workgroup over [x1 96 | x2 256]; seq: the "h once" form of the second operand
static int wgrad_cat_wide_launch(int64_t rows, int64_t M, const void* dy, int64_t ldy, const void* x1, int64_t ldx1,
                                 int64_t Nv1, const void* x2, int64_t ldx2, int64_t Nv2, int64_t NT, int64_t slices,
                                 float* part1, float* part2, const uint8_t* done, int64_t T, void* stream) {
    const int bm = NT == 22 ? 128 : 64;
    if ((M % bm) || (slices & 7) || rows % (slices * 64) || slices > 8192) return VINE_ERR_UNSUPPORTED;
    const int mtiles = (int)(M / bm), stages = (int)(rows / slices / 32);
    const size_t lds = (size_t)2 * 32 * ((bm + 16) + (352 + 16)) * sizeof(lp16_t);      // 64 / 56 KiB
#define VINE_WGW(MT_)                                                                                                     \
    {                                                                                                                     \
        if (!ensure_dyn_lds((const void*)wgrad_cat_wide_kernel<6, MT_>, lds)) return VINE_ERR_DEVICE;                     \
        hipLaunchKernelGGL((wgrad_cat_wide_kernel<6, MT_>), dim3((unsigned)(mtiles * slices)), dim3(512), lds,            \
                           (hipStream_t)stream, stages, mtiles, (int)slices, (const lp16_t*)dy, (long long)ldy,           \
                           (const lp16_t*)x1, (long long)ldx1, (const lp16_t*)x2, (long long)ldx2, part1, (int)Nv1,       \
                           part2, (int)Nv2, (int)M, done, done ? (int)T : (1 << 30));                                     \
    }
    if (NT == 22) VINE_WGW(2)
    else VINE_WGW(1)
#undef VINE_WGW
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_weight_grad_cat_seq_mfma(int64_t rows, int64_t M, const void* dy, int64_t ldy, const void* x1, int64_t ldx1,
                                  int64_t Nv1, const void* h_all, int64_t ldh, const uint8_t* done, int64_t T, int64_t Nv2,
                                  int64_t NT, int64_t slices, float* part1, float* part2, void* stream) {
    if (rows <= 0 || M <= 0 || slices <= 0 || T <= 0 || !dy || !x1 || !h_all || !done || !part1 || !part2 || ldy < M ||
        ldx1 < 96 || ldh < 256 || ((ldy | ldx1 | ldh) & 7) || (((uintptr_t)dy | (uintptr_t)x1 | (uintptr_t)h_all) & 15) ||
        Nv1 <= 0 || Nv1 > 96 || Nv2 <= 0 || Nv2 > 256 || rows % T)
        return VINE_ERR_INVALID_ARG;
    if ((NT != 22 && NT != 21) || (32 % T) || ((uintptr_t)done & 15) || rows / slices / 32 > 1024) return VINE_ERR_UNSUPPORTED;
    return wgrad_cat_wide_launch(rows, M, dy, ldy, x1, ldx1, Nv1, h_all, ldh, Nv2, NT, slices, part1, part2, done, T, stream);
}

int vine_weight_grad_cat_mfma(int64_t rows, int64_t M, const void* dy, int64_t ldy, const void* x1, int64_t ldx1, int64_t N1p,
                              int64_t Nv1, const void* x2, int64_t ldx2, int64_t N2p, int64_t Nv2, int64_t NT, int64_t slices,
                              float* part1, float* part2, void* stream) {
    if (rows <= 0 || M <= 0 || slices <= 0 || !dy || !x2 || !part2 || ldy < M || ldx2 < N2p || (ldy & 7) || (ldx2 & 7) ||
        ((uintptr_t)dy & 15) || ((uintptr_t)x2 & 15) || N2p <= 0 || Nv2 <= 0 || Nv2 > N2p || N1p < 0 ||
        (N1p > 0 && (!x1 || !part1 || ldx1 < N1p || (ldx1 & 7) || ((uintptr_t)x1 & 15) || Nv1 <= 0 || Nv1 > N1p)))
        return VINE_ERR_INVALID_ARG;
    if (NT == 22 || NT == 21) {
        if (N1p != 96 || N2p != 256) return VINE_ERR_UNSUPPORTED;
        return wgrad_cat_wide_launch(rows, M, dy, ldy, x1, ldx1, Nv1, x2, ldx2, Nv2, NT, slices, part1, part2, nullptr, 1,
                                     stream);
    }
    if ((NT != 11 && NT != 8 && NT != 2) || (M & 63) || (N1p & 15) || ((N1p + N2p) % (16 * NT)) || (slices & 7) ||
        rows % (slices * 32) || slices > 8192)
        return VINE_ERR_UNSUPPORTED;
    const void* dys[1] = {dy};
    const void* x1s[1] = {x1};
    const void* x2s[1] = {x2};
    float* p1s[1] = {part1};
    float* p2s[1] = {part2};
    return vine_weight_grad_group(1, &rows, &M, dys, &ldy, x1s, &ldx1, &N1p, &Nv1, x2s, &ldx2, &N2p, &Nv2, &NT, &slices, p1s, p2s,
                                  stream);
}

int vine_layernorm_forward(int64_t n, int64_t H, const float* x, const float* gamma, const float* beta, float eps,
                           float* y, float* mean, float* rstd, void* stream) {
    if (n <= 0 || !x || !gamma || !beta || !y || ((mean == nullptr) != (rstd == nullptr))) return VINE_ERR_INVALID_ARG;
    if (H != 256 && H != 512 && H != 1024) return VINE_ERR_UNSUPPORTED;
    const int blocks = (int)((n + 3) / 4 < 256 * 8 ? (n + 3) / 4 : 256 * 8);
    hipStream_t s = (hipStream_t)stream;
    if (H == 256) hipLaunchKernelGGL(layernorm_fwd_kernel<1>, dim3(blocks), dim3(256), 0, s, (long long)n, x, gamma, beta, eps, y, mean, rstd);
    else if (H == 512) hipLaunchKernelGGL(layernorm_fwd_kernel<2>, dim3(blocks), dim3(256), 0, s, (long long)n, x, gamma, beta, eps, y, mean, rstd);
    else hipLaunchKernelGGL(layernorm_fwd_kernel<4>, dim3(blocks), dim3(256), 0, s, (long long)n, x, gamma, beta, eps, y, mean, rstd);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_layernorm_backward(int64_t n, int64_t H, const float* dy, const float* x, const float* mean, const float* rstd,
                            const float* gamma, float* dx, float* partial, void* stream) {
    if (n <= 0 || !dy || !x || !mean || !rstd || !gamma || !dx || !partial) return VINE_ERR_INVALID_ARG;
    if (H != 256 && H != 512 && H != 1024) return VINE_ERR_UNSUPPORTED;
    const int blocks = VINE_PPO_PARTIAL_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    if (H == 256) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, (long long)n, dy, x, mean, rstd, gamma, dx, partial);
    else if (H == 512) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, (long long)n, dy, x, mean, rstd, gamma, dx, partial);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, (long long)n, dy, x, mean, rstd, gamma, dx, partial);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_layernorm_heads_forward(int64_t n, int64_t H, int64_t NH, const float* x, const float* gamma, const float* beta,
                                 float eps, const float* w, const float* wb, float* heads, float* mean, float* rstd,
                                 void* stream) {
    if (n <= 0 || !x || !gamma || !beta || !w || !wb || !heads || !mean || !rstd) return VINE_ERR_INVALID_ARG;
    if (H != 256 || NH < 2 || NH > 5) return VINE_ERR_UNSUPPORTED;
    const int blocks = (int)((n + 3) / 4 < 256 * 8 ? (n + 3) / 4 : 256 * 8);
    hipStream_t s = (hipStream_t)stream;
#define VINE_LNH_FWD(K) hipLaunchKernelGGL(ln_heads_fwd_kernel<K>, dim3(blocks), dim3(256), 0, s, (long long)n, x, gamma, beta, eps, w, wb, heads, mean, rstd)
    if (NH == 2) VINE_LNH_FWD(2); else if (NH == 3) VINE_LNH_FWD(3); else if (NH == 4) VINE_LNH_FWD(4); else VINE_LNH_FWD(5);
#undef VINE_LNH_FWD
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_layernorm_heads_backward(int64_t n, int64_t H, int64_t NH, const float* g, const float* x, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, const float* w, float* dx,
                                  float* partial, void* stream) {
    if (n <= 0 || !g || !x || !mean || !rstd || !gamma || !beta || !w || !dx || !partial) return VINE_ERR_INVALID_ARG;
    if (H != 256 || NH < 2 || NH > 5) return VINE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define VINE_LNH_BWD(K) hipLaunchKernelGGL(ln_heads_bwd_kernel<K>, dim3(VINE_PPO_PARTIAL_BLOCKS), dim3(256), 0, s, (long long)n, g, x, mean, rstd, gamma, beta, w, dx, partial)
    if (NH == 2) VINE_LNH_BWD(2); else if (NH == 3) VINE_LNH_BWD(3); else if (NH == 4) VINE_LNH_BWD(4); else VINE_LNH_BWD(5);
#undef VINE_LNH_BWD
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_elu_backward(int64_t n, int64_t C, const float* g, int64_t g_stride, const void* a, int64_t a_stride,
                      float alpha, void* out, int64_t out_stride, float* partial, int32_t a_bf16, int32_t out_bf16,
                      void* stream) {
    if (n <= 0 || C <= 0 || !g || !a || !out || (g_stride & 3) || (a_stride & 3) || (out_stride & 3))
        return VINE_ERR_INVALID_ARG;
    if ((C & 3) || C > 1024 || 256 % (C / 4) != 0) return VINE_ERR_UNSUPPORTED;
    const dim3 grid(VINE_PPO_PARTIAL_BLOCKS), block(256);
    hipStream_t s = (hipStream_t)stream;
#define VINE_ELU_BWD(AT, OT)                                                                                         \
    hipLaunchKernelGGL((elu_bwd_kernel<AT, OT>), grid, block, 0, s, (long long)n, (int)C, g, (long long)g_stride,    \
                       (const AT*)a, (long long)a_stride, alpha, (OT*)out, (long long)out_stride, partial)
    if (a_bf16 && out_bf16) VINE_ELU_BWD(lp16_t, lp16_t);
    else if (a_bf16) VINE_ELU_BWD(lp16_t, float);
    else if (out_bf16) VINE_ELU_BWD(float, lp16_t);
    else VINE_ELU_BWD(float, float);
#undef VINE_ELU_BWD
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_column_sums(int64_t R, int64_t C, const float* src, int64_t row_stride, float* out0, int64_t n0, float* out1,
                     int32_t dup, void* stream) {
    if (R <= 0 || C <= 0 || !src || !out0 || row_stride < C || (dup && !out1) || n0 < 0 || n0 > C)
        return VINE_ERR_INVALID_ARG;
    const long long split = out1 && !dup ? n0 : C;
    if (R >= 128 && C <= 8192)
        hipLaunchKernelGGL(colsum_tall_kernel, dim3((unsigned)((C + 15) / 16)), dim3(1024), 0, (hipStream_t)stream, src,
                           (long long)R, (long long)C, (long long)row_stride, out0, split, out1, (int)dup);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, src,
                           (long long)R, (long long)C, (long long)row_stride, out0, split, out1, (int)dup);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_gae(int32_t T, int64_t N, const float* rewards, const float* values, const uint8_t* dones,
             const float* last_values, const uint8_t* last_dones, float gamma, float tau, float* advs, float* returns,
             void* stream) {
    if (T <= 0 || N <= 0 || !rewards || !values || !dones || !last_values || !last_dones || !advs)
        return VINE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)T,
                       (long long)N, rewards, values, dones, last_values, last_dones, gamma, tau, advs, returns);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_dataset_assemble(int32_t T, int64_t N, const float* rewards, const float* values, const uint8_t* dones,
                          const float* last_values, const uint8_t* last_dones, float gamma, float tau, const double* vms_mean,
                          const double* vms_var, const double* vms_count, float vms_eps, int32_t normalize_value,
                          int32_t normalize_advantage, float* ds_values, float* ds_returns, float* ds_advantages,
                          int32_t njobs, const void* const* job_src, void* const* job_dst, const int32_t* job_width,
                          const int32_t* job_elem_bytes, double* scratch, double* vms_pending, void* stream) {
    if (T <= 0 || N <= 0 || !rewards || !values || !dones || !last_values || !last_dones || !ds_values || !ds_returns ||
        !ds_advantages || !scratch || njobs < 0 || (njobs > 0 && (!job_src || !job_dst || !job_width || !job_elem_bytes)) ||
        (normalize_value && (!vms_mean || !vms_var || !vms_count || !vms_pending)))
        return VINE_ERR_INVALID_ARG;
    if ((N & 63) || T > 256 || njobs > DS_MAX_JOBS) return VINE_ERR_UNSUPPORTED;
    DsJobs J;
    J.n = njobs;
    for (int j = 0; j < DS_MAX_JOBS; ++j) { J.src[j] = nullptr; J.dst[j] = nullptr; J.width[j] = 0; J.elem[j] = 0; }
    for (int j = 0; j < njobs; ++j) {
        if (!job_src[j] || !job_dst[j] || job_width[j] <= 0) return VINE_ERR_INVALID_ARG;
        if (job_elem_bytes[j] == 4) {
            if ((long long)T * job_width[j] > DS_LDS_WORDS || ((uintptr_t)job_src[j] & 3) || ((uintptr_t)job_dst[j] & 3))
                return VINE_ERR_UNSUPPORTED;
        } else if (job_elem_bytes[j] == 1) {
            if (job_width[j] != 1 || (DS_ENVS / 4) * T > DS_LDS_WORDS || ((uintptr_t)job_src[j] & 3) || ((uintptr_t)job_dst[j] & 3))
                return VINE_ERR_UNSUPPORTED;
        } else {
            return VINE_ERR_UNSUPPORTED;
        }
        J.src[j] = job_src[j]; J.dst[j] = job_dst[j]; J.width[j] = job_width[j]; J.elem[j] = job_elem_bytes[j];
    }
    if ((((uintptr_t)ds_values | (uintptr_t)ds_returns | (uintptr_t)ds_advantages) & 15)) return VINE_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((N + 255) / 256);
    float* scal = reinterpret_cast<float*>(scratch + (size_t)blocks * 6);
    hipLaunchKernelGGL(ds_gae_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (int)T, (long long)N, rewards, values,
                       dones, last_values, last_dones, gamma, tau, ds_values, ds_returns, ds_advantages, scratch);
    // the updated running statistics go to vms_pending {mean, var, count}; the caller commits them (rl_games updates
    // value_mean_std in prepare_dataset, not in the rollout: a rollout alone must leave the module as it was)
    hipLaunchKernelGGL(ds_finalize_kernel, dim3(1), dim3(64), 0, s, blocks, (long long)N * T, scratch, vms_mean, vms_var,
                       vms_count, vms_pending, vms_eps, (int)normalize_value, scal);
    hipLaunchKernelGGL(ds_assemble_kernel, dim3((unsigned)(N / DS_ENVS)), dim3(256), 0, s, (int)T, (long long)N, scal,
                       (int)normalize_value, (int)normalize_advantage, ds_values, ds_returns, ds_advantages, J);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_rms_update(int64_t n, int64_t F, const float* x, double* running_mean, double* running_var, double* count,
                    double* scratch, void* stream) {
    if (n <= 0 || F <= 0 || !x || !running_mean || !running_var || !count || !scratch) return VINE_ERR_INVALID_ARG;
    if (F > 64) return VINE_ERR_UNSUPPORTED;
    int blocks = (int)((n + 3) / 4);
    if (blocks > VINE_RMS_BLOCKS) blocks = VINE_RMS_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rms_partial_kernel, dim3(blocks), dim3(256), 0, s, (long long)n, (int)F, x, scratch);
    hipLaunchKernelGGL(rms_finalize_kernel, dim3(1), dim3(256), 0, s, blocks, (int)F, (long long)n, scratch, running_mean,
                       running_var, count);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_rms_update_multi(int32_t k, int64_t n, int64_t F, const float* x, double* running_mean, double* running_var,
                          double* count, double* scratch, double* snap_mean, double* snap_var, void* stream) {
    if (k <= 0 || n <= 0 || F <= 0 || !x || !running_mean || !running_var || !count || !scratch || !snap_mean || !snap_var)
        return VINE_ERR_INVALID_ARG;
    if (F > 64 || k > 65535) return VINE_ERR_UNSUPPORTED;
    int blocks = (int)((n + 3) / 4);
    if (blocks > VINE_RMS_BLOCKS) blocks = VINE_RMS_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    double* sums = scratch + (size_t)k * VINE_RMS_BLOCKS * 2 * F;
    hipLaunchKernelGGL(rms_partial_kernel, dim3(blocks, k), dim3(256), 0, s, (long long)n, (int)F, x, scratch);
    hipLaunchKernelGGL(rms_fold_multi_kernel, dim3(k), dim3(256), 0, s, blocks, (int)F, scratch, sums);
    hipLaunchKernelGGL(rms_merge_multi_kernel, dim3(1), dim3(64), 0, s, (int)k, (int)F, (long long)n, sums, running_mean,
                       running_var, count, snap_mean, snap_var);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_normalize_obs(int64_t n, int64_t F, const float* x, const double* mean, const double* var, float eps, float clip,
                       void* out, int64_t out_stride, int32_t out_bf16, void* stream) {
    if (n <= 0 || F <= 0 || !x || !mean || !var || !out || out_stride < F) return VINE_ERR_INVALID_ARG;
    const int threads = 256;
    const dim3 grid(grid_for(n * F, threads));
    if (out_bf16)
        hipLaunchKernelGGL(normalize_obs_kernel<lp16_t>, grid, dim3(threads), 0, (hipStream_t)stream, (long long)n, (int)F,
                           x, mean, var, eps, clip, (lp16_t*)out, (long long)out_stride);
    else
        hipLaunchKernelGGL(normalize_obs_kernel<float>, grid, dim3(threads), 0, (hipStream_t)stream, (long long)n, (int)F,
                           x, mean, var, eps, clip, (float*)out, (long long)out_stride);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_column_sums_batched(int32_t njobs, const int64_t* R, const int64_t* C, const float* const* src,
                             const int64_t* row_stride, float* const* out0, const int64_t* n0, float* const* out1,
                             const int32_t* dup, float* found_inf, void* stream) {
    return vine_column_sums_batched_fin(njobs, R, C, src, row_stride, out0, n0, out1, dup, found_inf, nullptr, stream);
}

int vine_column_sums_batched_fin(int32_t njobs, const int64_t* R, const int64_t* C, const float* const* src,
                                 const int64_t* row_stride, float* const* out0, const int64_t* n0, float* const* out1,
                                 const int32_t* dup, float* found_inf, const VineLossFinalize* fin, void* stream) {
    if (njobs <= 0 || njobs > VINE_COLSUM_MAX_JOBS || !R || !C || !src || !row_stride || !out0 || !n0 || !out1 || !dup)
        return VINE_ERR_INVALID_ARG;
    if (fin && (!fin->partial || fin->blocks <= 0 || fin->blocks > VINE_PPO_LOSS_BLOCKS || fin->A <= 0 || fin->A > PPO_MAX_A ||
                fin->n <= 0 || !fin->logstd || !fin->stats || !fin->grad_logstd ||
                ((fin->grad_mu_bias == nullptr) != (fin->grad_value_bias == nullptr))))
        return VINE_ERR_INVALID_ARG;
    ColsumBatch b;
    int blocks = 0;
    for (int k = 0; k < njobs; ++k) {
        if (R[k] <= 0 || C[k] <= 0 || !src[k] || !out0[k] || row_stride[k] < C[k] || (dup[k] && !out1[k]) || n0[k] < 0 ||
            n0[k] > C[k])
            return VINE_ERR_INVALID_ARG;
        // 16-B geometry for short, wide jobs whose rows, split point and outputs are all 16-B aligned
        const bool quad = R[k] < 128 && C[k] >= 4096 && !(C[k] & 3) && !(row_stride[k] & 3) && !(n0[k] & 3) &&
                          !((uintptr_t)src[k] & 15) && !((uintptr_t)out0[k] & 15) && !((uintptr_t)out1[k] & 15);
        const int ct = quad ? 256 : (R[k] >= 128 ? 16 : 64);
        b.job[k] = ColsumJob{src[k], out0[k], out1[k], (long long)R[k], (long long)C[k], (long long)row_stride[k],
                             (long long)(out1[k] && !dup[k] ? n0[k] : C[k]), (int)dup[k], blocks, (int)quad};
        blocks += (int)((C[k] + ct - 1) / ct);
    }
    b.njobs = njobs;
    b.found_inf = found_inf;
    if (fin) { b.fin = *fin; ++blocks; }
    else b.fin.partial = nullptr;
    hipLaunchKernelGGL(colsum_batched_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

static int copy_batch_pack(CopyBatchArgs& b, int& blocks, int32_t njobs, const int32_t* op, const int32_t* elem,
                           const void* const* src, const void* const* src2, void* const* dst, const int64_t* rows,
                           const int64_t* cols, const int64_t* src_stride, const int64_t* dst_stride, const int64_t* aux) {
    if (njobs <= 0 || njobs > VINE_COPY_MAX_JOBS || !op || !elem || !src || !src2 || !dst || !rows || !cols ||
        !src_stride || !dst_stride || !aux)
        return VINE_ERR_INVALID_ARG;
    blocks = 0;
    for (int k = 0; k < njobs; ++k) {
        if (op[k] < 0 || op[k] > 10 || rows[k] <= 0 || cols[k] <= 0 || !dst[k] || (op[k] != 1 && !src[k]) ||
            ((op[k] == 4 || op[k] == 6 || op[k] == 8) && !src2[k]) || (elem[k] != 2 && elem[k] != 4) || (op[k] >= 6 && elem[k] != 2))
            return VINE_ERR_INVALID_ARG;
        if (op[k] >= 8) {
            // scatter forms: 4 consecutive source elements = one aligned 8-B load, all index arithmetic in 32 bits
            const long long tot = rows[k] * cols[k];
            bool ok = tot < (1ll << 31) && !(tot & 3) && !(src_stride[k] & 3) && !((uintptr_t)src[k] & 7) && !((uintptr_t)dst[k] & 7);
            if (op[k] == 8) {
                const long long cols1 = aux[k] & 0xffff, K1 = aux[k] >> 16;
                ok = ok && !(cols1 & 3) && !(K1 & 3) && !((K1 + SEQ_H) & 31) && !(dst_stride[k] & 3) && !((uintptr_t)src2[k] & 7) &&
                     tot == 4ll * SEQ_H * (K1 + SEQ_H);
            } else if (op[k] == 9) {
                ok = ok && tot == 4ll * SEQ_H * SEQ_H;
            } else {
                ok = ok && !(rows[k] & 3);
            }
            if (!ok) return VINE_ERR_UNSUPPORTED;
        }
        // vector path: every group of 4 consecutive elements is one aligned 8-/16-B access on both sides
        const int src_elem = op[k] == 0 ? elem[k] : 4;
        const bool vec = op[k] != 2 && op[k] < 6 && !(cols[k] & 3) && !(dst_stride[k] & 3) && !((uintptr_t)dst[k] & (4 * elem[k] - 1)) &&
                         (op[k] == 1 || (!(src_stride[k] & 3) && !((uintptr_t)src[k] & (4 * src_elem - 1)))) &&
                         (op[k] != 4 || !((uintptr_t)src2[k] & 15));
        b.job[k] = CopyJob{src[k], src2[k], dst[k], (long long)rows[k], (long long)cols[k], (long long)src_stride[k],
                           (long long)dst_stride[k], (long long)aux[k], (int)op[k], (int)elem[k], blocks, (int)vec};
        blocks += (int)((rows[k] * cols[k] + 1023) / 1024);
    }
    b.njobs = njobs;
    return VINE_OK;
}

int vine_copy_batched(int32_t njobs, const int32_t* op, const int32_t* elem, const void* const* src,
                      const void* const* src2, void* const* dst, const int64_t* rows, const int64_t* cols,
                      const int64_t* src_stride, const int64_t* dst_stride, const int64_t* aux, void* stream) {
    CopyBatchArgs b;
    int blocks = 0;
    const int rc = copy_batch_pack(b, blocks, njobs, op, elem, src, src2, dst, rows, cols, src_stride, dst_stride, aux);
    if (rc != VINE_OK) return rc;
    hipLaunchKernelGGL(copy_batched_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_bias_elu(int64_t n, int64_t C, const float* z, const float* bias, float alpha, void* out, int64_t out_stride,
                  int32_t out_bf16, void* stream) {
    if (n <= 0 || C <= 0 || (C & 3) || (out_stride & 3) || !z || !bias || !out) return VINE_ERR_INVALID_ARG;
    const int threads = 256;
    const dim3 grid(grid_for(n * (C / 4), threads));
    if (out_bf16)
        hipLaunchKernelGGL(bias_elu_kernel<lp16_t>, grid, dim3(threads), 0, (hipStream_t)stream, (long long)n, (int)C, z,
                           bias, alpha, (lp16_t*)out, (long long)out_stride);
    else
        hipLaunchKernelGGL(bias_elu_kernel<float>, grid, dim3(threads), 0, (hipStream_t)stream, (long long)n, (int)C, z,
                           bias, alpha, (float*)out, (long long)out_stride);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_ppo_loss(int64_t n, int32_t A, const float* mu, const float* logstd, const float* value, const float* actions,
                  const float* old_neglogp, const float* advantages, const float* old_values, const float* returns,
                  const float* old_mu, const float* old_sigma, float e_clip, int32_t clip_value, float critic_coef,
                  float entropy_coef, float bounds_coef, float soft_bound, float* grad_mu, float* grad_value,
                  float* grad_logstd, float* stats, int64_t mu_stride, int64_t value_stride, float* grad_mu_bias,
                  float* grad_value_bias, float* scratch, float* kl_out, float* logstd_grad_accum, float* mu_store,
                  float* sigma_store, const float* loss_scale, void* stream) {
    if (n <= 0 || A <= 0 || A > PPO_MAX_A || !mu || !logstd || !value || !actions || !old_neglogp || !advantages ||
        !old_values || !returns || !old_mu || !old_sigma || !grad_mu || !grad_value || !grad_logstd || !stats ||
        ((grad_mu_bias == nullptr) != (grad_value_bias == nullptr)) || !scratch ||
        ((mu_store == nullptr) != (sigma_store == nullptr)))
        return VINE_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned int* ticket = ticket_slot(stream, TICKET_LOSS);
    if (!ticket) return VINE_ERR_DEVICE;
    const int threads = 256;
    int blocks = (int)((n + threads - 1) / threads);
    if (blocks > VINE_PPO_LOSS_BLOCKS) blocks = VINE_PPO_LOSS_BLOCKS;
    // no hipMemsetAsync and no float atomics: memset nodes of a few bytes did not survive repeated hipGraph replays
    // intact on ROCm 7.0, and atomics would make the gradients depend on the order workgroups retire in
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(blocks), dim3(threads), 0, s, (long long)n, (int)A, mu, logstd, value,
                       actions, old_neglogp, advantages, old_values, returns, old_mu, old_sigma, e_clip, (int)clip_value,
                       critic_coef, entropy_coef, bounds_coef, soft_bound, grad_mu, grad_value, grad_logstd, stats,
                       (long long)(mu_stride > 0 ? mu_stride : A), (long long)(value_stride > 0 ? value_stride : 1),
                       scratch, mu_store, sigma_store, grad_mu_bias, grad_value_bias, kl_out, logstd_grad_accum, loss_scale, ticket);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_ln_heads_loss(int64_t n, int64_t H, int32_t NH, const void* x, const float* gamma, const float* beta, float eps,
                       const float* w, const float* wb, const float* logstd, const float* actions, const float* old_neglogp,
                       const float* advantages, const float* old_values, const float* returns, const float* old_mu,
                       const float* old_sigma, float e_clip, int32_t clip_value, float critic_coef, float entropy_coef,
                       float bounds_coef, float soft_bound, float* heads, void* dx, int32_t dx_bf16, float* ln_partial,
                       float* stats, float* grad_logstd, float* grad_mu_bias, float* grad_value_bias, float* scratch,
                       float* kl_out, float* logstd_grad_accum, float* mu_store, float* sigma_store, const float* loss_scale,
                       float* found_inf, void* stream) {
    if (n <= 0 || !x || !gamma || !beta || !w || !wb || !logstd || !actions || !old_neglogp || !advantages || !old_values ||
        !returns || !old_mu || !old_sigma || !heads || !dx || !ln_partial || !stats || !grad_logstd || !scratch ||
        ((grad_mu_bias == nullptr) != (grad_value_bias == nullptr)) || ((mu_store == nullptr) != (sigma_store == nullptr)))
        return VINE_ERR_INVALID_ARG;
    // rows per wave (4 per pass): 16 keeps the workgroup count -- and with it the serial tail of the last workgroup,
    // which folds one row of loss sums per workgroup -- small (VINE_LHL_ROWS = 4 | 8 | 16 for experiments)
    static int rw = 0;
    if (!rw) {
        const char* e = getenv("VINE_LHL_ROWS");
        rw = e ? atoi(e) : 16;
        if (rw != 4 && rw != 8 && rw != 16) rw = 16;
    }
    const int rows_wg = 8 * rw;
    if (H != 256 || NH < 2 || NH > 5 || n % rows_wg || n / rows_wg > VINE_PPO_LOSS_BLOCKS) return VINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(n / rows_wg)), block(512);
    hipStream_t s = (hipStream_t)stream;
    unsigned int* ticket = ticket_slot(stream, TICKET_LOSS);
    if (!ticket) return VINE_ERR_DEVICE;
    const bool x16 = (dx_bf16 & 2) != 0;      // bit 1: x holds the library's 16-bit format (then dx does too)
    const int xT = (dx_bf16 >> 8) & 0xff;     // bits 8-15 (with bit 1): x is [n / T, T + 1, H], the samples in slots 1 .. T
    if ((x16 && !(dx_bf16 & 1)) || (xT && (!x16 || n % xT))) return VINE_ERR_UNSUPPORTED;
    const int defer = (dx_bf16 >> 2) & 1;     // bit 2: leave the per-workgroup loss rows in `scratch` (vine_column_sums_batched_fin)
    dx_bf16 &= 3;
#define VINE_LHL_T(K, R, DXT, XT)                                                                                         \
    hipLaunchKernelGGL((ln_heads_loss_kernel<K, R, DXT, XT>), grid, block, 0, s, (long long)n, (const XT*)x, xT, gamma, beta, \
                       eps, w, wb, logstd, actions, old_neglogp, advantages, old_values, returns, old_mu, old_sigma, e_clip,           \
                       (int)clip_value, critic_coef, entropy_coef, bounds_coef, soft_bound, heads, (DXT*)dx, ln_partial,   \
                       scratch, stats, grad_logstd, grad_mu_bias, grad_value_bias, kl_out, logstd_grad_accum, mu_store,    \
                       sigma_store, loss_scale, found_inf, ticket, defer)
#define VINE_LHL(K, R)                                                                                                    \
    {                                                                                                                     \
        if (x16) VINE_LHL_T(K, R, lp16_t, lp16_t);                                                                        \
        else if (dx_bf16) VINE_LHL_T(K, R, lp16_t, float);                                                                \
        else VINE_LHL_T(K, R, float, float);                                                                              \
    }
#define VINE_LHL_R(K)                                                                                                     \
    do {                                                                                                                  \
        if (rw == 4) VINE_LHL(K, 1)                                                                                       \
        else if (rw == 8) VINE_LHL(K, 2)                                                                                  \
        else VINE_LHL(K, 4)                                                                                               \
    } while (0)
    switch (NH) {
        case 2: VINE_LHL_R(2); break;
        case 3: VINE_LHL_R(3); break;
        case 4: VINE_LHL_R(4); break;
        default: VINE_LHL_R(5); break;
    }
#undef VINE_LHL_R
#undef VINE_LHL
#undef VINE_LHL_T
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_ln_heads_loss_rows(void) {
    const char* e = getenv("VINE_LHL_ROWS");
    const int rw = e ? atoi(e) : 16;
    return 8 * ((rw == 8 || rw == 4) ? rw : 16);
}

static int policy_head_launch(int64_t N, int32_t A, int64_t H, const float* y, const float* w_mu, const float* b_mu,
                              const float* w_v, const float* b_v, const float* logstd, const float* value_mean,
                              const float* value_std, int32_t normalize_value, float value_eps, uint64_t seed,
                              const int64_t* counter, float* mu_out, float* sigma_out, float* value_out, float* action_out,
                              float* neglogp_out, const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream);

int vine_policy_head(int64_t N, int32_t A, int64_t H, const float* y, const float* w_mu, const float* b_mu,
                     const float* w_v, const float* b_v, const float* logstd, const float* value_mean,
                     const float* value_std, int32_t normalize_value, uint64_t seed, const int64_t* counter,
                     float* mu_out, float* sigma_out, float* value_out, float* action_out, float* neglogp_out,
                     const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream) {
    return policy_head_launch(N, A, H, y, w_mu, b_mu, w_v, b_v, logstd, value_mean, value_std, normalize_value ? 1 : 0, 0.0f, seed,
                              counter, mu_out, sigma_out, value_out, action_out, neglogp_out, ln_gamma, ln_beta, ln_eps, stream);
}

int vine_policy_head_rms(int64_t N, int32_t A, int64_t H, const float* y, const float* w_mu, const float* b_mu,
                         const float* w_v, const float* b_v, const float* logstd, const double* running_mean,
                         const double* running_var, float value_eps, uint64_t seed, const int64_t* counter, float* mu_out,
                         float* sigma_out, float* value_out, float* action_out, float* neglogp_out, const float* ln_gamma,
                         const float* ln_beta, float ln_eps, void* stream) {
    if (!running_mean || !running_var) return VINE_ERR_INVALID_ARG;
    return policy_head_launch(N, A, H, y, w_mu, b_mu, w_v, b_v, logstd, reinterpret_cast<const float*>(running_mean),
                              reinterpret_cast<const float*>(running_var), 2, value_eps, seed, counter, mu_out, sigma_out,
                              value_out, action_out, neglogp_out, ln_gamma, ln_beta, ln_eps, stream);
}

static int policy_head_launch(int64_t N, int32_t A, int64_t H, const float* y, const float* w_mu, const float* b_mu,
                              const float* w_v, const float* b_v, const float* logstd, const float* value_mean,
                              const float* value_std, int32_t normalize_value, float value_eps, uint64_t seed,
                              const int64_t* counter, float* mu_out, float* sigma_out, float* value_out, float* action_out,
                              float* neglogp_out, const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream) {
    if (N <= 0 || A <= 0 || A > HEAD_MAX_A || H <= 0 || (H & 3) || !y || !w_mu || !b_mu || !w_v || !b_v || !logstd ||
        !counter || !mu_out || !sigma_out || !value_out || !action_out || !neglogp_out ||
        (normalize_value && (!value_mean || !value_std)) || ((ln_gamma == nullptr) != (ln_beta == nullptr)))
        return VINE_ERR_INVALID_ARG;
    if (ln_gamma && H != 256) return VINE_ERR_UNSUPPORTED;
    if (ln_gamma && (N & 15) == 0) {             // 16 lanes per env row: 16 rows per 256-thread workgroup
        hipLaunchKernelGGL(policy_head16_kernel, dim3((unsigned)(N / 16)), dim3(256), 0, (hipStream_t)stream, (long long)N,
                           (int)A, y, w_mu, b_mu, w_v, b_v, logstd, value_mean, value_std, (int)normalize_value,
                           (unsigned)seed, (unsigned)(seed >> 32), (const long long*)counter, mu_out, sigma_out, value_out,
                           action_out, neglogp_out, ln_gamma, ln_beta, ln_eps, value_eps);
        return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
    }
    const int threads = 256;                     // 4 waves per workgroup, one env per wave at a time
    long long blocks = (N + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(policy_head_kernel, dim3((int)blocks), dim3(threads), 0, (hipStream_t)stream, (long long)N, (int)A,
                       (int)H, y, w_mu, b_mu, w_v, b_v, logstd, value_mean, value_std, (int)normalize_value,
                       (unsigned)seed, (unsigned)(seed >> 32), (const long long*)counter, mu_out, sigma_out, value_out,
                       action_out, neglogp_out, ln_gamma, ln_beta, ln_eps, value_eps);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_rollout_post(int64_t N, int64_t H, const float* rew, const int64_t* reset, const uint8_t* timeouts,
                      const float* values, float reward_shift, float reward_scale, float gamma_bootstrap,
                      float* shaped_out, uint8_t* dones_out, float* cur_rewards, float* cur_lengths, float* h_state,
                      float* c_state, float* meter, float max_size, int64_t* counter, void* h_op, int64_t h_op_stride,
                      int32_t h_op_bf16, float* scratch, void* stream) {
    if (N <= 0 || H <= 0 || (H & 3) || !rew || !reset || !timeouts || !values || !shaped_out || !dones_out ||
        !cur_rewards || !cur_lengths || !h_state || !c_state || !meter || !counter || !scratch)
        return VINE_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int threads = 256;
    long long blocks = (N * 16 + threads - 1) / threads;
    if (blocks > ROLLOUT_POST_BLOCKS) blocks = ROLLOUT_POST_BLOCKS;
    hipLaunchKernelGGL(rollout_post_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, (long long)N, (int)H, rew,
                       (const long long*)reset, timeouts, values, reward_shift, reward_scale, gamma_bootstrap, shaped_out,
                       dones_out, cur_rewards, cur_lengths, h_state, c_state, scratch, h_op, (long long)h_op_stride,
                       (int)h_op_bf16);
    hipLaunchKernelGGL(rollout_finalize_kernel, dim3(1), dim3(256), 0, s, meter, max_size, (long long*)counter,
                       (const float*)scratch, (int)blocks);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_rollout_finalize(float* meter, float max_size, int64_t* counter, const float* scratch, int32_t blocks, void* stream) {
    if (!meter || !counter || !scratch || blocks <= 0 || blocks > ROLLOUT_POST_BLOCKS) return VINE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rollout_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, meter, max_size,
                       (long long*)counter, scratch, (int)blocks);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int32_t vine_rollout_post_blocks(int64_t N) {
    long long blocks = (N * 16 + 255) / 256;
    return (int32_t)(blocks > ROLLOUT_POST_BLOCKS ? ROLLOUT_POST_BLOCKS : blocks);
}

int vine_rollout_post_defer(int64_t N, int64_t H, const float* rew, const int64_t* reset, const uint8_t* timeouts,
                            const float* values, float reward_shift, float reward_scale, float gamma_bootstrap,
                            float* shaped_out, uint8_t* dones_out, float* cur_rewards, float* cur_lengths, float* h_state,
                            float* c_state, void* h_op, int64_t h_op_stride, int32_t h_op_bf16, float* scratch, void* stream) {
    if (N <= 0 || H <= 0 || (H & 3) || !rew || !reset || !timeouts || !values || !shaped_out || !dones_out ||
        !cur_rewards || !cur_lengths || !h_state || !c_state || !scratch)
        return VINE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rollout_post_kernel, dim3((unsigned)vine_rollout_post_blocks(N)), dim3(256), 0, (hipStream_t)stream,
                       (long long)N, (int)H, rew, (const long long*)reset, timeouts, values, reward_shift, reward_scale,
                       gamma_bootstrap, shaped_out, dones_out, cur_rewards, cur_lengths, h_state, c_state, scratch, h_op,
                       (long long)h_op_stride, (int)h_op_bf16);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_adam_step_amp(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* lr, float* step,
                       float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* lp16_shadow,
                       const float* kl, float kl_scale, float kl_threshold, float min_lr, float max_lr, float* amp_state,
                       float* found_inf, void* stream) {
    if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !lr || !step) return VINE_ERR_INVALID_ARG;
    unsigned int* ticket = ticket_slot(stream, TICKET_ADAM);
    if (!ticket) return VINE_ERR_DEVICE;
    const int threads = 256;
    // every workgroup ends with a returning atomic for the election of the one that advances the step counter and the
    // schedules.  On ONE ticket word these serialise at ~23 ns each: 400 workgroups spent longer queueing there than on their
    // 4 KB of parameters.  Default: a two-level election (8 group words 128 B apart, then the common word) on the full grid;
    // VINE_ADAM_TICKETS=1 = the single word, for which the grid is capped at 128 workgroups (VINE_ADAM_BLOCKS overrides)
    static const bool two_level = [] { const char* e = getenv("VINE_ADAM_TICKETS"); return !e || atoi(e) != 1; }();
    static const int max_blocks = [] { const char* e = getenv("VINE_ADAM_BLOCKS"); return e ? atoi(e) : (two_level ? 0 : 128); }();
    int blocks = grid_for((n + 3) / 4, threads);
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    unsigned int* sub = two_level ? ticket_sub_of(ticket) : nullptr;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, (long long)n,
                       params, grads, exp_avg, exp_avg_sq, lr, step, beta1, beta2, eps, weight_decay, grad_scale,
                       (lp16_t*)lp16_shadow, kl, kl_scale, kl_threshold, min_lr, max_lr, amp_state, found_inf, ticket, sub);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_adam_step_sched(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* lr, float* step,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* bf16_shadow,
                         const float* kl, float kl_scale, float kl_threshold, float min_lr, float max_lr, void* stream) {
    return vine_adam_step_amp(n, params, grads, exp_avg, exp_avg_sq, lr, step, beta1, beta2, eps, weight_decay, grad_scale,
                              bf16_shadow, kl, kl_scale, kl_threshold, min_lr, max_lr, nullptr, nullptr, stream);
}


int vine_lstm_step_f32(int64_t N, int64_t H, int64_t K, const float* xh, int64_t ldx, const float* w_tiled, const float* bias,
                       const float* c_prev, float* h_out, int64_t ldh, float* c_out, float* hp_next, int64_t ldhp,
                       void* stream) {
    if (N <= 0 || !xh || !w_tiled || !bias || !c_prev || !h_out || !c_out) return VINE_ERR_INVALID_ARG;
    if (H != 256 || K != 352 || (N & 63) || (ldx & 3) || ldx < K || (ldh & 3) || ldh < H || (hp_next && ((ldhp & 3) || ldhp < H)) ||
        ((uintptr_t)xh & 15) || ((uintptr_t)w_tiled & 15) || ((uintptr_t)h_out & 15) || ((uintptr_t)hp_next & 15))
        return VINE_ERR_UNSUPPORTED;
    const int rbs = (int)(N / 64);
    if (rbs & 7) return VINE_ERR_UNSUPPORTED;             // the XCD-aware block mapping walks row blocks in groups of 8
    hipLaunchKernelGGL(lstm_step_f32_kernel<22>, dim3(rbs * 4), dim3(256), 0, (hipStream_t)stream, (long long)N, xh,
                       (long long)ldx, w_tiled, bias, c_prev, h_out, (long long)ldh, c_out, hp_next, (long long)ldhp);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_lstm_tile_weights_f32(int64_t H, int64_t K, const float* wcat, int64_t ldw, float* dst, void* stream) {
    if (!wcat || !dst) return VINE_ERR_INVALID_ARG;
    if (H != 256 || (K & 15) || K <= 0 || (ldw & 3) || ldw < K || ((uintptr_t)wcat & 15) || ((uintptr_t)dst & 15))
        return VINE_ERR_UNSUPPORTED;
    const int KB = (int)(K / 16);
    hipLaunchKernelGGL(lstm_tile_weights_f32_kernel, dim3(grid_for(4LL * KB * 1024, 256)), dim3(256), 0, (hipStream_t)stream, KB,
                       wcat, (long long)ldw, dst);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_lstm_step_f32_split(int64_t N, int64_t H, int64_t K, const float* xh, int64_t ldx, const void* w_split,
                             const float* bias, const float* c_prev, float* h_out, int64_t ldh, float* c_out, float* hp_next,
                             int64_t ldhp, int terms, void* stream) {
    if (N <= 0 || !xh || !w_split || !bias || !c_prev || !h_out || !c_out) return VINE_ERR_INVALID_ARG;
    if ((terms & 255) != 9 && (terms & 255) != 6) return VINE_ERR_INVALID_ARG;
    if (H != 256 || K != 352 || (N & 127) || (ldx & 3) || ldx < K || (ldh & 3) || ldh < H || (hp_next && ((ldhp & 3) || ldhp < H)) ||
        ((uintptr_t)xh & 15) || ((uintptr_t)w_split & 15) || ((uintptr_t)h_out & 15) || ((uintptr_t)hp_next & 15) ||
        ((uintptr_t)c_prev & 15) || ((uintptr_t)c_out & 15) || ((uintptr_t)bias & 15))
        return VINE_ERR_UNSUPPORTED;
    // piece pairs in the low byte; second byte: row tiles per wave (tuning knob; 0 = 4 from 16384 rows on when N % 256 == 0:
    // 512 workgroups = two per CU in one round; else 2)
    const unsigned short* wt = (const unsigned short*)w_split;
    const int nt = terms & 255;
    int rt = (terms >> 8) & 255;
    if (!rt) rt = (N >= 16384 && N % 256 == 0) ? 4 : 2;
    if ((terms >> 16) & 1) {
        // bit 16: the one-gate-per-wave form (64 rows and one 32-unit block per workgroup; the second byte is not used);
        // bit 17 (with it): two accumulators per tile -- hi x hi apart from the smaller pairs
        if (((terms >> 8) & 255) || (terms >> 18)) return VINE_ERR_UNSUPPORTED;
        const bool dual = ((terms >> 17) & 1) != 0;
        const long long sets64 = N / 64;
        const int xmap = (sets64 & 7) == 0;
#define LAUNCH_NSPLIT(NT_, DUAL_)                                                                                            \
    hipLaunchKernelGGL((lstm_step_nsplit_kernel<11, NT_, DUAL_>), dim3((unsigned)sets64 * 8), dim3(256), 0, (hipStream_t)stream, \
                       (long long)N, xh, (long long)ldx, wt, bias, c_prev, h_out, (long long)ldh, c_out, hp_next,              \
                       (long long)ldhp, xmap)
        if (dual) { if (nt == 9) LAUNCH_NSPLIT(9, true); else LAUNCH_NSPLIT(6, true); }
        else { if (nt == 9) LAUNCH_NSPLIT(9, false); else LAUNCH_NSPLIT(6, false); }
#undef LAUNCH_NSPLIT
        return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
    }
    if ((rt != 2 && rt != 4) || (terms >> 16) || N % (64LL * rt)) return VINE_ERR_UNSUPPORTED;
    const long long sets = N / (64LL * rt);
    const int xcd_map = (sets & 7) == 0;
#define LAUNCH_SPLIT(RT_, NT_)                                                                                              \
    hipLaunchKernelGGL((lstm_step_split_kernel<11, RT_, NT_>), dim3((unsigned)sets * 8), dim3(256), 0, (hipStream_t)stream,   \
                       (long long)N, xh, (long long)ldx, wt, bias, c_prev, h_out, (long long)ldh, c_out, hp_next,              \
                       (long long)ldhp, xcd_map)
    if (rt == 2) { if (nt == 9) LAUNCH_SPLIT(2, 9); else LAUNCH_SPLIT(2, 6); }
    else { if (nt == 9) LAUNCH_SPLIT(4, 9); else LAUNCH_SPLIT(4, 6); }
#undef LAUNCH_SPLIT
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

#ifdef SPLIT_TIMING
int vine_debug_split_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(split_t), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : -1;
}
int vine_debug_mlp_split_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mlp_split_t), sizeof(unsigned long long) * 4096 * 16) == hipSuccess ? 0 : -1;
}
#endif

int vine_lstm_tile_weights_split(int64_t H, int64_t K, const float* wcat, int64_t ldw, void* dst, void* stream) {
    if (!wcat || !dst) return VINE_ERR_INVALID_ARG;
    if (H != 256 || (K & 31) || K <= 0 || (ldw & 3) || ldw < K || ((uintptr_t)wcat & 15) || ((uintptr_t)dst & 15))
        return VINE_ERR_UNSUPPORTED;
    const int KS = (int)(K / 32);
    hipLaunchKernelGGL(lstm_tile_weights_split_kernel, dim3(grid_for(8LL * KS * 512, 256)), dim3(256), 0, (hipStream_t)stream,
                       KS, wcat, (long long)ldw, (uint4*)dst);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_mlp3_elu_f32_fin(int64_t n, float* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean,
                          const double* var, float eps, float clip, const float* w1, int64_t ldw1, const float* b1, int64_t C1,
                          const float* w2, int64_t ldw2, const float* b2, int64_t C2, const float* w3, int64_t ldw3,
                          const float* b3, int64_t C3, float alpha, float* fin_meter, float fin_max_size, int64_t* fin_counter,
                          const float* fin_scratch, int32_t fin_blocks, void* stream) {
    if (fin_meter && (!fin_counter || !fin_scratch || fin_blocks <= 0)) return VINE_ERR_INVALID_ARG;
    if (n <= 0 || !x || !raw || !mean || !var || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) return VINE_ERR_INVALID_ARG;
    if (C1 != 256 || C2 != 128 || C3 != 64 || (n & 63) || F_in <= 0 || F_in > 32 || ldx < C3 + 32 || (ldx & 3) || (ldw2 & 3) ||
        ldw2 < C1 || (ldw3 & 3) || ldw3 < C2 || ldw1 < 32 || (ldw1 & 3) || ((uintptr_t)w1 & 15) || ((uintptr_t)x & 15) || ((uintptr_t)w2 & 15) ||
        ((uintptr_t)w3 & 15) || ((uintptr_t)b1 & 15) || ((uintptr_t)b2 & 15) || ((uintptr_t)b3 & 15))
        return VINE_ERR_UNSUPPORTED;
    const size_t lds = (size_t)128 * (256 + 4) * sizeof(float);          // W2 [C2][C1 + 4]: the largest of the three stages
    if (!ensure_dyn_lds(reinterpret_cast<const void*>(mlp3_elu_f32_kernel), lds)) return VINE_ERR_DEVICE;
    hipLaunchKernelGGL(mlp3_elu_f32_kernel, dim3((unsigned)(n / 64)), dim3(256), lds, (hipStream_t)stream, (long long)n, x,
                       (long long)ldx, raw, (int)F_in, mean, var, eps, clip, w1, (long long)ldw1, b1, w2, (long long)ldw2, b2, w3,
                       (long long)ldw3, b3, alpha, fin_meter, fin_max_size, (long long*)fin_counter, (const float*)fin_scratch,
                       fin_blocks);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_mlp3_tile_weights_split(const float* w1, int64_t ldw1, int64_t F_in, const float* w2, int64_t ldw2, const float* w3,
                                 int64_t ldw3, void* dst, void* stream) {
    if (!w1 || !w2 || !w3 || !dst) return VINE_ERR_INVALID_ARG;
    if (F_in <= 0 || F_in > 32 || ldw1 < F_in || ldw2 < 256 || ldw3 < 128 || ((uintptr_t)dst & 15)) return VINE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mlp3_tile_weights_split_kernel, dim3((MLP_SPLIT_FRAGS / 3) * 64 / 256), dim3(256), 0, (hipStream_t)stream,
                       w1, (long long)ldw1, (int)F_in, w2, (long long)ldw2, w3, (long long)ldw3, (uint4*)dst);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_mlp3_elu_f32_split(int64_t n, float* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean,
                            const double* var, float eps, float clip, const void* wt, const float* b1, const float* b2,
                            const float* b3, float alpha, int terms, float* fin_meter, float fin_max_size,
                            int64_t* fin_counter, const float* fin_scratch, int32_t fin_blocks, void* stream) {
    if (fin_meter && (!fin_counter || !fin_scratch || fin_blocks <= 0)) return VINE_ERR_INVALID_ARG;
    if (n <= 0 || !x || !raw || !mean || !var || !wt || !b1 || !b2 || !b3) return VINE_ERR_INVALID_ARG;
    if ((terms & 255) != 9 && (terms & 255) != 6) return VINE_ERR_INVALID_ARG;
    int rt = (terms >> 8) & 255;                     // row tiles per workgroup (0: chosen here), a tuning knob as in the LSTM step
    if (!rt) rt = n >= 32768 ? 4 : (n >= 16384 ? 2 : 1);       // two workgroups per CU from 16384 rows on (measured in situ)
    const bool dual = ((terms >> 16) & 1) != 0;      // bit 16: two accumulators per tile (hi x hi apart from the smaller pairs)
    if ((rt != 1 && rt != 2 && rt != 4) || (terms >> 17) || n % (16 * rt) || F_in <= 0 || F_in > 32 || ldx < 64 + 32 || (ldx & 3) ||
        ((uintptr_t)x & 15) || ((uintptr_t)wt & 15) || ((uintptr_t)b1 & 15) || ((uintptr_t)b2 & 15) || ((uintptr_t)b3 & 15))
        return VINE_ERR_UNSUPPORTED;
    const size_t lds = (size_t)rt * (4 + 8) * 3 * 64 * sizeof(uint4);
    const int nt = terms & 255;
#define LAUNCH_MLP_SPLIT_D(RT_, NT_, DU_)                                                                                     \
    {                                                                                                                        \
        if (!ensure_dyn_lds(reinterpret_cast<const void*>(mlp3_elu_split_kernel<RT_, NT_, DU_>), lds)) return VINE_ERR_DEVICE; \
        hipLaunchKernelGGL((mlp3_elu_split_kernel<RT_, NT_, DU_>), dim3((unsigned)(n / (16 * RT_))), dim3(256), lds,         \
                           (hipStream_t)stream, (long long)n, x, (long long)ldx, raw, (int)F_in, mean, var, eps, clip,        \
                           (const uint4*)wt, b1, b2, b3, alpha, fin_meter, fin_max_size, (long long*)fin_counter,             \
                           (const float*)fin_scratch, fin_blocks);                                                           \
    }
#define LAUNCH_MLP_SPLIT(RT_, NT_) { if (dual) LAUNCH_MLP_SPLIT_D(RT_, NT_, true) else LAUNCH_MLP_SPLIT_D(RT_, NT_, false) }
    if (rt == 4) { if (nt == 9) LAUNCH_MLP_SPLIT(4, 9) else LAUNCH_MLP_SPLIT(4, 6) }
    else if (rt == 2) { if (nt == 9) LAUNCH_MLP_SPLIT(2, 9) else LAUNCH_MLP_SPLIT(2, 6) }
    else { if (nt == 9) LAUNCH_MLP_SPLIT(1, 9) else LAUNCH_MLP_SPLIT(1, 6) }
#undef LAUNCH_MLP_SPLIT
#undef LAUNCH_MLP_SPLIT_D
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

int vine_mlp3_elu_f32(int64_t n, float* x, int64_t ldx, const float* raw, int64_t F_in, const double* mean, const double* var,
                      float eps, float clip, const float* w1, int64_t ldw1, const float* b1, int64_t C1, const float* w2,
                      int64_t ldw2, const float* b2, int64_t C2, const float* w3, int64_t ldw3, const float* b3, int64_t C3,
                      float alpha, void* stream) {
    return vine_mlp3_elu_f32_fin(n, x, ldx, raw, F_in, mean, var, eps, clip, w1, ldw1, b1, C1, w2, ldw2, b2, C2, w3, ldw3, b3, C3,
                                 alpha, nullptr, 0.0f, nullptr, nullptr, 0, stream);
}

const char* vine_lp16_format(void) { return VINE_LP16_NAME; }

int vine_ppo_runtime_init(void) { return ticket_slot(nullptr, TICKET_LOSS) ? VINE_OK : VINE_ERR_DEVICE; }

int vine_adam_step(int64_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* lr,
                   float* step, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                   void* bf16_shadow, void* stream) {
    return vine_adam_step_sched(n, params, grads, exp_avg, exp_avg_sq, const_cast<float*>(lr), step, beta1, beta2, eps,
                                weight_decay, grad_scale, bf16_shadow, nullptr, 0.0f, 0.0f, 0.0f, 0.0f, stream);
}

int vine_adaptive_lr(float* lr, const float* kl, float kl_scale, float kl_threshold, float min_lr, float max_lr,
                     void* stream) {
    if (!lr || !kl) return VINE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(adaptive_lr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, lr, kl, kl_scale, kl_threshold, min_lr,
                       max_lr);
    return hipGetLastError() == hipSuccess ? VINE_OK : VINE_ERR_DEVICE;
}

}  // extern "C"
