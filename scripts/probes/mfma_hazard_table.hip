// Probe kernels whose compiler output (hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only) shows the wait states hipcc ROCm 7.2
// inserts around v_mfma_f32_16x16x32_bf16: VALU -> SrcA 2 (k1), MFMA -> VALU / store / ds_write 8 (k3, k6, k7), MFMA -> MFMA SrcA 8 (k5),
// MFMA -> MFMA SrcC (same registers) 0 (k4, k7).  scripts/isa_lint.py rule R2 is this table.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
#define MF(a,b,c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0)
// 1: VALU write -> MFMA srcA
extern "C" __global__ void k1_valu_to_srcA(const i32x4* in, f32x4* out) {
    i32x4 x = in[threadIdx.x]; i32x4 y = in[threadIdx.x + 64];
    f32x4 acc = {0,0,0,0};
    x += 1;  // VALU
    acc = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc);
    out[threadIdx.x] = acc;
}
// 3: MFMA -> VALU read
extern "C" __global__ void k3_mfma_to_valu(const i32x4* in, f32x4* out, float s) {
    i32x4 x = in[threadIdx.x]; i32x4 y = in[threadIdx.x + 64];
    f32x4 acc = out[threadIdx.x + 64];
    acc = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc);
    acc = acc * s;
    out[threadIdx.x] = acc;
}
// 4: MFMA -> MFMA srcC, different dst
extern "C" __global__ void k4_mfma_to_srcC(const i32x4* in, f32x4* out) {
    i32x4 x = in[threadIdx.x]; i32x4 y = in[threadIdx.x + 64];
    f32x4 c0 = out[threadIdx.x + 64];
    f32x4 d1 = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c0);
    f32x4 d2 = MF(__builtin_bit_cast(bf16x8, y), __builtin_bit_cast(bf16x8, x), d1);
    out[threadIdx.x] = d1; out[threadIdx.x + 128] = d2;
}
// 5: MFMA -> MFMA srcA
extern "C" __global__ void k5_mfma_to_srcA(const i32x4* in, f32x4* out) {
    i32x4 x = in[threadIdx.x]; i32x4 y = in[threadIdx.x + 64];
    f32x4 c0 = out[threadIdx.x + 64];
    f32x4 d1 = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c0);
    f32x4 d2 = MF(__builtin_bit_cast(bf16x8, d1), __builtin_bit_cast(bf16x8, x), c0);
    out[threadIdx.x] = d2;
}
// 6: MFMA -> store / ds_write
extern "C" __global__ void k6_mfma_to_store(const i32x4* in, f32x4* out) {
    __shared__ f32x4 sm[64];
    i32x4 x = in[threadIdx.x]; i32x4 y = in[threadIdx.x + 64];
    f32x4 c0 = out[threadIdx.x + 64];
    f32x4 d1 = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c0);
    sm[threadIdx.x] = d1;
    __syncthreads();
    out[threadIdx.x] = sm[63 - threadIdx.x];
}
// 7: in-place chain then read
extern "C" __global__ void k7_chain(const i32x4* in, f32x4* out) {
    i32x4 x = in[threadIdx.x]; i32x4 y = in[threadIdx.x + 64];
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) acc = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc);
    out[threadIdx.x] = acc;
}
// 8: WAR: MFMA reads A, then VALU overwrites A
extern "C" __global__ void k8_war(const i32x4* in, f32x4* out, i32x4* out2) {
    i32x4 x = in[threadIdx.x]; i32x4 y = in[threadIdx.x + 64];
    f32x4 acc = out[threadIdx.x + 64];
    acc = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc);
    x = x * 3;
    out2[threadIdx.x] = x;
    out[threadIdx.x] = acc;
}
// 9: ds_read -> MFMA, and MFMA followed by ds_read into its A regs
extern "C" __global__ void k9_lds(const i32x4* in, f32x4* out) {
    __shared__ i32x4 sm[128];
    sm[threadIdx.x] = in[threadIdx.x]; sm[threadIdx.x + 64] = in[threadIdx.x + 64];
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        i32x4 x = sm[(threadIdx.x + i) & 127], y = sm[(threadIdx.x + 2 * i + 64) & 127];
        acc = MF(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc);
    }
    out[threadIdx.x] = acc;
}
