// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths vine_step_kernel uses
// (MI355X_MICROARCH.md, HBM section: only 16-B-per-lane streaming is calibrated there; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel moves exactly BYTES bytes.
//   hipcc -O3 --offload-arch=gfx950 -o pmc_calib pmc_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -- ./pmc_calib ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -- ./pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr size_t BYTES = 256ull << 20;

template <typename T>
__global__ void read_kernel(const T* __restrict__ src, size_t n, float* __restrict__ sink) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = src[i];
        acc += *reinterpret_cast<float*>(&v);
    }
    if (acc == 12345.678f) sink[0] = acc;     // never true: keeps the loads alive without a store
}
template <typename T>
__global__ void write_kernel(T* __restrict__ dst, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
// SoA field access like vine_step_kernel: lane e reads field f at base[f * n + e] (4 B per lane, one wave = 256 B)
__global__ void soa_read_kernel(const float* __restrict__ base, size_t n_env, int fields, float* __restrict__ sink) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    float acc = 0.0f;
    for (int f = 0; f < fields; ++f) acc += base[(size_t)f * n_env + e];
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void soa_write_kernel(float* __restrict__ base, size_t n_env, int fields) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    for (int f = 0; f < fields; ++f) base[(size_t)f * n_env + e] = (float)f;
}
// the four-lanes-per-env kernel's pattern: the 4 lanes of a quad read the SAME word (one wave = 16 envs = a 64-B segment
// per field), and lane t of the quad stores field 4 j + t (again 64-B segments per wave)
__global__ void soa_quad_read_kernel(const float* __restrict__ base, size_t n_env, int fields, float* __restrict__ sink) {
    size_t e = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    if (e >= n_env) return;
    float acc = 0.0f;
    for (int f = 0; f < fields; ++f) acc += base[(size_t)f * n_env + e];
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void soa_quad_write_kernel(float* __restrict__ base, size_t n_env, int fields) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x, e = g >> 2;
    int t = (int)(g & 3);
    if (e >= n_env) return;
    for (int f = 0; f < fields / 4; ++f) base[(size_t)(4 * f + t) * n_env + e] = (float)f;
}
// row-major 28-float rows written as 7 float4 per lane (the observation row store): 112 B per lane, strided by lane
__global__ void rows_write_kernel(float4* __restrict__ dst, size_t rows) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows) return;
    for (int k = 0; k < 7; ++k) dst[e * 7 + k] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    void *a, *s;
    hipMalloc(&a, BYTES);
    hipMalloc(&s, 64);
    hipMemset(a, 0, BYTES);
    const int blocks = 256 * 8, threads = 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_kernel<float>, dim3(blocks), dim3(threads), 0, 0, (const float*)a, BYTES / 4, (float*)s);
        hipLaunchKernelGGL(read_kernel<float2>, dim3(blocks), dim3(threads), 0, 0, (const float2*)a, BYTES / 8, (float*)s);
        hipLaunchKernelGGL(read_kernel<float4>, dim3(blocks), dim3(threads), 0, 0, (const float4*)a, BYTES / 16, (float*)s);
        hipLaunchKernelGGL(write_kernel<float>, dim3(blocks), dim3(threads), 0, 0, (float*)a, BYTES / 4, 1.0f);
        hipLaunchKernelGGL(write_kernel<float2>, dim3(blocks), dim3(threads), 0, 0, (float2*)a, BYTES / 8, make_float2(1.f, 2.f));
        hipLaunchKernelGGL(write_kernel<float4>, dim3(blocks), dim3(threads), 0, 0, (float4*)a, BYTES / 16, make_float4(1.f, 2.f, 3.f, 4.f));
        // 16384-env shaped SoA block (60 fields x 16384 envs = 3.9 MB) and a 2^20-env one (252 MB)
        hipLaunchKernelGGL(soa_read_kernel, dim3(64), dim3(256), 0, 0, (const float*)a, (size_t)16384, 60, (float*)s);
        hipLaunchKernelGGL(soa_write_kernel, dim3(64), dim3(256), 0, 0, (float*)a, (size_t)16384, 60);
        hipLaunchKernelGGL(soa_read_kernel, dim3(4096), dim3(256), 0, 0, (const float*)a, (size_t)1 << 20, 60, (float*)s);
        hipLaunchKernelGGL(soa_write_kernel, dim3(4096), dim3(256), 0, 0, (float*)a, (size_t)1 << 20, 60);
        hipLaunchKernelGGL(soa_quad_read_kernel, dim3(256), dim3(256), 0, 0, (const float*)a, (size_t)16384, 60, (float*)s);
        hipLaunchKernelGGL(soa_quad_write_kernel, dim3(256), dim3(256), 0, 0, (float*)a, (size_t)16384, 60);
        hipLaunchKernelGGL(rows_write_kernel, dim3(64), dim3(256), 0, 0, (float4*)a, (size_t)16384);
        hipLaunchKernelGGL(rows_write_kernel, dim3(4096), dim3(256), 0, 0, (float4*)a, (size_t)1 << 20);
    }
    hipDeviceSynchronize();
    printf("bytes per streaming kernel: %zu; soa 16384: %d; soa 2^20: %zu; rows 16384: %d; rows 2^20: %zu\n", BYTES,
           60 * 16384 * 4, (size_t)60 * 4 << 20, 16384 * 112, (size_t)112 << 20);
    return 0;
}
