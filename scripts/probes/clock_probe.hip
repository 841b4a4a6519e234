// Shader clock under load: s_memtime (shader cycles) against s_memrealtime (100 MHz) around a busy loop, one wave per SIMD on every CU,
// (a) idle-ish integer loop, (b) back-to-back bf16 MFMAs.  Build: hipcc --offload-arch=gfx950 -O2 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(long long* out, int iters, int mode) {
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc = {0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    int x = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (mode == 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) x = x * 1664525 + 1013904223;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = r1 - r0;
        out[blockIdx.x * 4 + 2] = (long long)(acc[0] + x);
    }
}
int main() {
    long long* d;
    hipMalloc(&d, 4096 * 4 * sizeof(long long));
    long long h[4096 * 4];
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d, mode ? 20000 : 20000, mode);
            hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0;
            for (int i = 0; i < 1024; ++i) { cyc += h[i * 4]; rt += h[i * 4 + 1]; }
            printf("mode %s rep %d: %.0f memtime ticks per workgroup over %.1f us -> %.3f GHz\n", mode ? "mfma" : "int ", rep, cyc / 1024, rt / 1024 / 100.0,
                   (cyc / 1024) / (rt / 1024 / 100.0) / 1e3);
        }
    return 0;
}
