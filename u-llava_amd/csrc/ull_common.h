// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of u-llava_amd.
// Written for wave64 + MFMA only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all HBM activations / weights are bf16

using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;  // one MFMA A/B operand (4 VGPRs)
using f32x4_t = __attribute__((ext_vector_type(4))) float;    // 16x16 MFMA accumulator
using f32x16_t = __attribute__((ext_vector_type(16))) float;  // 32x32 MFMA accumulator

#define ULL_DEV __device__ __forceinline__

// ---- bf16 <-> f32 (round-to-nearest-even, identical to torch's c10::BFloat16) -------------
ULL_DEV float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16 goes through the native __bf16 type so hipcc emits gfx950's v_cvt_pk_bf16_f32 (IEEE round-to-nearest-even,
// two values per instruction) instead of a ~8-instruction integer sequence; bf16 -> f32 is a 16-bit shift.
typedef __bf16 bf16x2_native_t __attribute__((ext_vector_type(2)));
ULL_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
// Round an fp32 value through bf16: this is how the kernels reproduce the rounding points of the
// reference's bf16 PyTorch graph (every torch op boundary stores bf16) inside fused epilogues.
ULL_DEV float rbf(float f) { return (float)(__bf16)f; }

ULL_DEV uint32_t pack2bf(float lo, float hi) {
    const bf16x2_native_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
ULL_DEV void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
ULL_DEV uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
    v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// ---- wave64 reductions ------------------------------------------------------------------
ULL_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
ULL_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce across the `w` lanes (power of two <= 64) that share the same (lane / w)
ULL_DEV float group_sum(float v, int w) {
    for (int o = w >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
ULL_DEV float group_max(float v, int w) {
    for (int o = w >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- MFMA wrappers ------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16: D[16x16] += A[16x32] * B[32x16].
//   operand A: lane l holds A[row = l&15][k = 8*(l>>4) + j], j = 0..7
//   operand B: lane l holds B[k = 8*(l>>4) + j][col = l&15]
//   C/D      : lane l, reg r holds D[row = 4*(l>>4) + r][col = l&15]
// (the k <-> (lane group, j) map is the same for A and B, so any K-contiguous 16-byte load that is
//  identical for both operands is correct by construction.)
ULL_DEV f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_bf16: D[32x32] += A[32x16] * B[16x32].
//   operand A: lane l holds A[row = l&31][k = 8*(l>>5) + j];  operand B: lane l holds B[k = 8*(l>>5) + j][col = l&31]
//   C/D      : lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
ULL_DEV f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ---- activations (computed in fp32 on a bf16-rounded input, like torch's bf16 CPU/GPU kernels)
ULL_DEV float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
ULL_DEV float act_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
ULL_DEV float act_silu(float x) { return x / (1.0f + __expf(-x)); }
// transformers QuickGELUActivation on a bf16 tensor: input * sigmoid(1.702 * input) has THREE bf16
// roundings (the scaled input, the sigmoid, the product).
ULL_DEV float act_quick_gelu_bf16(float t) {
    float u = rbf(1.702f * t);
    float s = rbf(act_sigmoid(u));
    return rbf(t * s);
}

// C-ABI status codes (mirrored in include/ullava_hip.h)
#define ULL_OK 0
#define ULL_ERR_ARG (-1)
#define ULL_ERR_SHAPE (-2)
#define ULL_ERR_LAUNCH (-3)
#define ULL_ERR_LDS (-4)

// hipFuncSetAttribute is per device: one of these per kernel instantiation remembers which devices already have it
// (a racing first call repeats an idempotent attribute call; nothing else is shared between host threads or streams).
struct UllOncePerDevice {
    unsigned long long seen = 0;
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        const unsigned long long bit = 1ull << dev;
        if (seen & bit) return false;
        seen |= bit;
        return true;
    }
};

static inline int ull_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ULL_OK : ULL_ERR_LAUNCH;
}
