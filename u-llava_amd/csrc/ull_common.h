// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of u-llava_amd.
// Written for wave64 + MFMA only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- element type of the 16-bit tensors: a BUILD-TIME parameter of the translation unit --------------------------------------
// Every dtype-dependent .hip file is compiled twice (Makefile): as-is for bfloat16 (entry points ull_*_bf16) and with
// -DULL_ELEM_F16 for IEEE binary16 (entry points ull_*_f16; the reference's `--dtype fp16`, inference_ullava.py:26,164-168).
// The kernels are written against elem_t / e2f / f2e / rnd / pack2e / pk_lo / pk_hi / mfma16 only, so both builds keep the same
// rounding points (every torch op boundary of the reference's 16-bit graph) -- only the 16-bit format differs.
typedef uint16_t elem_t;   // raw bits of one 16-bit element (bfloat16 or binary16); all HBM activations / weights

using f32x4_t = __attribute__((ext_vector_type(4))) float;    // 16x16 MFMA accumulator
using f32x16_t = __attribute__((ext_vector_type(16))) float;  // 32x32 MFMA accumulator

#define ULL_DEV __device__ __forceinline__
#define ULL_CAT_(a, b) a##b
#define ULL_CAT(a, b) ULL_CAT_(a, b)

// both formats are always available by name (the byte-level pre/post-processing and loss kernels take a dtype code)
#define ULL_DT_F32 0
#define ULL_DT_BF16 1
#define ULL_DT_F16 2
ULL_DEV float bf16_bits_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
ULL_DEV float f16_bits_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// f32 -> 16 bit goes through the native types so hipcc emits gfx950's v_cvt_pk_bf16_f32 / v_cvt_f16_f32 (IEEE round-to-nearest-even,
// identical to torch's c10::BFloat16 / c10::Half conversions)
ULL_DEV uint16_t f32_to_bf16_bits(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
ULL_DEV uint16_t f32_to_f16_bits(float f) {
    asm("" : "+v"(f));             // keep a preceding multiply out of the conversion (v_fma_mixlo_f16 would round the product once; see rnd)
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// dtype-coded access for the kernels that are compiled once and take ULL_DT_* at run time (bilinear, box losses, u8 -> CHW)
template <int DT> ULL_DEV float load_dt(const void* p, long i) {
    if constexpr (DT == ULL_DT_F32) return ((const float*)p)[i];
    else if constexpr (DT == ULL_DT_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
    else return f16_bits_to_f32(((const uint16_t*)p)[i]);
}
template <int DT> ULL_DEV void store_dt(void* p, long i, float v) {
    if constexpr (DT == ULL_DT_F32) ((float*)p)[i] = v;
    else if constexpr (DT == ULL_DT_BF16) ((uint16_t*)p)[i] = f32_to_bf16_bits(v);
    else ((uint16_t*)p)[i] = f32_to_f16_bits(v);
}

#ifdef ULL_ELEM_F16
// ------------------------------------------------------------------ IEEE binary16 build
#define ULL_FN(base) ULL_CAT(base, f16)                        // ULL_FN(ull_gemm_) -> ull_gemm_f16
using mfma_ab_t = __attribute__((ext_vector_type(8))) _Float16;   // one MFMA A/B operand (4 VGPRs)
typedef _Float16 elem2_native_t __attribute__((ext_vector_type(2)));
constexpr uint16_t ELEM_NEG_INF = 0xFC00;
constexpr uint16_t ELEM_MIN = 0xFBFF;                          // torch.finfo(torch.float16).min = -65504
#define ELEM_MIN_F (-65504.0f)
ULL_DEV float e2f(elem_t v) { return f16_bits_to_f32(v); }
ULL_DEV elem_t f2e(float f) { return f32_to_f16_bits(f); }
// The rounded value is made opaque to the optimiser: with -ffp-contract=fast LLVM folds fpext(fptrunc(a16 * b16)) + c into a
// mixed-precision v_fma_mix_f32, which keeps the product UNROUNDED (measured: 19 % of RoPE outputs off by one fp16 ulp).
// The INPUT is opaque too: fptrunc(a * b) otherwise becomes one v_fma_mixlo_f16 that rounds the exact product once, while torch rounds
// the fp32 product and then the half (a tie after the first rounding goes the other way: QuickGELU's 1.702 * x at x = -2.9297, found
// by the G6 fixture).
ULL_DEV float rnd(float f) {
    asm("" : "+v"(f));
    _Float16 h = (_Float16)f;
    asm("" : "+v"(h));
    return (float)h;
}
ULL_DEV uint32_t pack2e(float lo, float hi) {
    asm("" : "+v"(lo));
    asm("" : "+v"(hi));
    const elem2_native_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
// low / high element of a packed pair as fp32
ULL_DEV float pk_lo(uint32_t v) { return (float)__builtin_bit_cast(elem2_native_t, v)[0]; }
ULL_DEV float pk_hi(uint32_t v) { return (float)__builtin_bit_cast(elem2_native_t, v)[1]; }
#else
// ------------------------------------------------------------------ bfloat16 build
#define ULL_FN(base) ULL_CAT(base, bf16)                       // ULL_FN(ull_gemm_) -> ull_gemm_bf16
using mfma_ab_t = __attribute__((ext_vector_type(8))) __bf16;  // one MFMA A/B operand (4 VGPRs)
typedef __bf16 elem2_native_t __attribute__((ext_vector_type(2)));
constexpr uint16_t ELEM_NEG_INF = 0xFF80;
constexpr uint16_t ELEM_MIN = 0xFF7F;                          // torch.finfo(torch.bfloat16).min: the eager additive mask value
#define ELEM_MIN_F (-3.3895313892515355e38f)
ULL_DEV float e2f(elem_t v) { return bf16_bits_to_f32(v); }
ULL_DEV elem_t f2e(float f) { return f32_to_bf16_bits(f); }
// Round an fp32 value through the element type: this is how the kernels reproduce the rounding points of the reference's 16-bit
// PyTorch graph (every torch op boundary stores a 16-bit tensor) inside fused epilogues.
ULL_DEV float rnd(float f) { return (float)(__bf16)f; }
ULL_DEV uint32_t pack2e(float lo, float hi) {
    const elem2_native_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
ULL_DEV float pk_lo(uint32_t v) { return __uint_as_float(v << 16); }           // bf16 -> fp32 is a 16-bit shift
ULL_DEV float pk_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
#endif

ULL_DEV void unpack8(const uint4& v, float* f) {
    f[0] = pk_lo(v.x); f[1] = pk_hi(v.x);
    f[2] = pk_lo(v.y); f[3] = pk_hi(v.y);
    f[4] = pk_lo(v.z); f[5] = pk_hi(v.z);
    f[6] = pk_lo(v.w); f[7] = pk_hi(v.w);
}
ULL_DEV uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2e(f[0], f[1]); v.y = pack2e(f[2], f[3]);
    v.z = pack2e(f[4], f[5]); v.w = pack2e(f[6], f[7]);
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
// v_mfma_f32_16x16x32_{bf16,f16}: D[16x16] += A[16x32] * B[32x16].
//   operand A: lane l holds A[row = l&15][k = 8*(l>>4) + j], j = 0..7
//   operand B: lane l holds B[k = 8*(l>>4) + j][col = l&15]
//   C/D      : lane l, reg r holds D[row = 4*(l>>4) + r][col = l&15]
// (the k <-> (lane group, j) map is the same for A and B, so any K-contiguous 16-byte load that is
//  identical for both operands is correct by construction.)
ULL_DEV f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
#ifdef ULL_ELEM_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_ab_t, a), __builtin_bit_cast(mfma_ab_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_ab_t, a), __builtin_bit_cast(mfma_ab_t, b), c, 0, 0, 0);
#endif
}

// v_mfma_f32_32x32x16_{bf16,f16}: D[32x32] += A[32x16] * B[16x32].
//   operand A: lane l holds A[row = l&31][k = 8*(l>>5) + j];  operand B: lane l holds B[k = 8*(l>>5) + j][col = l&31]
//   C/D      : lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
ULL_DEV f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
#ifdef ULL_ELEM_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mfma_ab_t, a), __builtin_bit_cast(mfma_ab_t, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_ab_t, a), __builtin_bit_cast(mfma_ab_t, b), c, 0, 0, 0);
#endif
}

// ---- activations (computed in fp32 on a 16-bit-rounded input, like torch's bf16 / fp16 CPU and GPU kernels)
// GELU(erf) = 0.5 x (1 + erf(x / sqrt 2)), the formula torch evaluates in fp32 (ATen GeluKernel).  erf through the complementary
// function of |z| in the Chebyshev form of Numerical Recipes' erfcc (fractional error < 1.2e-7 everywhere): branch-free, one rcp,
// nine FMAs and one exp -- the library erff is two divergent branches and ~3x the instructions, which made the
// SAM MLP epilogue longer than its K-loop.  The inputs are 16-bit values, so the test is exhaustive: over all 51 022 normal bf16
// inputs this differs from torch's CPU bf16 GELU in ~24 results, all in the tail x in [-5.4, -3.1] where |GELU| < 3e-3 and the
// reference's own 1 + erf has lost its digits (torch's fp32 formula differs from torch's bf16 kernel in 27): tests/test_kernels_gpu.py.
ULL_DEV float act_gelu_erf(float x) {
    // a = |z| sqrt(log2 e) with z = x / sqrt 2, and every polynomial coefficient times log2 e: the exponential is then a bare v_exp_f32
    const float a = fabsf(x) * 0.8493218002880191f;
    const float d = fmaf(0.41627730557884884f, a, 1.0f);
    const float t = __builtin_amdgcn_rcpf(d);            // 1 / (1 + |z| / 2): the 1-ulp reciprocal is enough (a Newton step changes no result)
    float p = 0.24651729790196045f;
    p = fmaf(t, p, -1.1861149450768025f);
    p = fmaf(t, p, 2.147474463933521f);
    p = fmaf(t, p, -1.637753152343414f);
    p = fmaf(t, p, 0.40232158165127635f);
    p = fmaf(t, p, -0.2687568603388257f);
    p = fmaf(t, p, 0.1396300565225048f);
    p = fmaf(t, p, 0.5397006155284324f);
    p = fmaf(t, p, 1.4427292039075317f);
    const float ec = t * __builtin_amdgcn_exp2f(fmaf(-a, a, fmaf(t, p, -1.825748218405333f)));   // erfc(|z|)
    const float erf = __builtin_copysignf(1.0f - ec, x);  // 1 - ec >= 0: one v_bfi instead of compare + two subtractions + select
    return 0.5f * x * (1.0f + erf);
}
// The same on two values at a time, written on 2-vectors so that the multiplies, adds and FMAs become packed fp32 instructions
// (v_pk_mul / v_pk_add / v_pk_fma_f32: two lanes of work per issue slot); the reciprocal, the exponential and the sign transfer
// stay scalar.  Same operations in the same order as act_gelu_erf on each element: identical results.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
ULL_DEV f32x2_t act_gelu_erf2(f32x2_t x) {
    const f32x2_t ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2_t a = ax * 0.8493218002880191f;
    const f32x2_t d = __builtin_elementwise_fma(f32x2_t{0.41627730557884884f, 0.41627730557884884f}, a, f32x2_t{1.0f, 1.0f});
    const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2_t p = {0.24651729790196045f, 0.24651729790196045f};
#define ULL_H2(c) p = __builtin_elementwise_fma(t, p, f32x2_t{c, c})
    ULL_H2(-1.1861149450768025f); ULL_H2(2.147474463933521f); ULL_H2(-1.637753152343414f); ULL_H2(0.40232158165127635f);
    ULL_H2(-0.2687568603388257f); ULL_H2(0.1396300565225048f); ULL_H2(0.5397006155284324f); ULL_H2(1.4427292039075317f);
#undef ULL_H2
    const f32x2_t g = __builtin_elementwise_fma(-a, a, __builtin_elementwise_fma(t, p, f32x2_t{-1.825748218405333f, -1.825748218405333f}));
    const f32x2_t ec = t * f32x2_t{__builtin_amdgcn_exp2f(g.x), __builtin_amdgcn_exp2f(g.y)};
    const f32x2_t om = f32x2_t{1.0f, 1.0f} - ec;
    const f32x2_t erf = {__builtin_copysignf(om.x, x.x), __builtin_copysignf(om.y, x.y)};
    return (x * 0.5f) * (f32x2_t{1.0f, 1.0f} + erf);
}

// 1 / d for d in [1, inf): the hardware reciprocal (1 ulp) plus one Newton step, three instructions where the IEEE division sequence
// (v_div_scale / v_rcp / 4 FMAs / v_div_fmas / v_div_fixup) is ten -- the SwiGLU and QuickGELU epilogues are VALU-bound.  The results
// go through a 16-bit rounding right away; the fixtures and the exhaustive activation tests pin that they do not move.
ULL_DEV float rcp_newton(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
ULL_DEV float act_sigmoid(float x) { return rcp_newton(1.0f + __expf(-x)); }
ULL_DEV float act_silu(float x) { return x * rcp_newton(1.0f + __expf(-x)); }
// transformers QuickGELUActivation on a 16-bit tensor: input * sigmoid(1.702 * input) has THREE roundings
// (the scaled input, the sigmoid, the product).
ULL_DEV float act_quick_gelu_e(float t) {
    float u = rnd(1.702f * t);
    float s = rnd(act_sigmoid(u));
    return rnd(t * s);
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

// compute units of the current device (sizes the grids of the kernels that walk their work list: one workgroup per CU)
static inline int ull_cu_count() {
    static int n_cu[64];                 // 0 = not yet asked
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n_cu[dev]) {
        hipDeviceProp_t prop;
        n_cu[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n_cu[dev];
}

static inline int ull_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ULL_OK : ULL_ERR_LAUNCH;
}
