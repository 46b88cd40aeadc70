// Issue cost of the vector instructions the attention kernels are made of (gfx950): one wave per SIMD (256 threads, 1 block per CU on a few CUs),
// 8 independent chains of one instruction, 4096 issues per wave, s_memtime around the loop -> cycles per wave-instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_rate tools/probes/valu_rate.hip ; run: tools/probes/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(name, BODY)                                                                                  \
    __global__ void name(unsigned long long* out, float seed) {                                             \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;   \
        float b0 = a0 * 2, b1 = a1 * 2, b2 = a2 * 2, b3 = a3 * 2, b4 = a4 * 2, b5 = a5 * 2, b6 = a6 * 2, b7 = a7 * 2;                 \
        const unsigned long long t0 = __builtin_readcyclecounter();                                         \
        for (int i = 0; i < 512; ++i) { BODY }                                                               \
        const unsigned long long t1 = __builtin_readcyclecounter();                                         \
        float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;            \
        if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }                                                    \
        if (s == 12345.678f) out[3000] = 1;                                                                 \
    }
#define X_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
#define X_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
#define X_AND(i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a##i));
#define X_SHL(i) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a##i));
#define X_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
#define X_CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
#define X_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define X_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p##i));
#define X_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p##i));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p##i));
KERNEL(k_exp, REP8(X_EXP))
KERNEL(k_add, REP8(X_ADD))
KERNEL(k_fma, REP8(X_FMA))
KERNEL(k_and, REP8(X_AND))
KERNEL(k_shl, REP8(X_SHL))
KERNEL(k_max3, REP8(X_MAX3))
KERNEL(k_cvt, REP8(X_CVT))
KERNEL(k_rcp, REP8(X_RCP))
#define X_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a##i) : "v"(b##i));
#define X_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
#define X_SDWA(i) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(a##i));
#define X_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a##i) : "v"(b##i));
#define X_MULU24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
#define X_BFI(i) asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b##i));
#define X_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
#define X_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
#define X_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b##i));
#define X_LSHL1(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a##i) : "v"(b##i));
#define X_ADDSDWA(i) asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a##i) : "v"(b##i));
KERNEL(k_align, REP8(X_ALIGN))
KERNEL(k_perm, REP8(X_PERM))
KERNEL(k_sdwa, REP8(X_SDWA))
KERNEL(k_lshlor, REP8(X_LSHLOR))
KERNEL(k_mulu24, REP8(X_MULU24))
KERNEL(k_bfi, REP8(X_BFI))
KERNEL(k_mul, REP8(X_MUL))
KERNEL(k_max, REP8(X_MAX))
KERNEL(k_addu, REP8(X_ADDU))
KERNEL(k_lshl1, REP8(X_LSHL1))
KERNEL(k_addsdwa, REP8(X_ADDSDWA))
typedef float f2 __attribute__((ext_vector_type(2)));
#define PKKERNEL(name, BODY)                                                                                \
    __global__ void name(unsigned long long* out, float seed) {                                             \
        f2 p0 = {seed + threadIdx.x, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f; \
        const unsigned long long t0 = __builtin_readcyclecounter();                                         \
        for (int i = 0; i < 512; ++i) { BODY }                                                               \
        const unsigned long long t1 = __builtin_readcyclecounter();                                         \
        f2 s = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;                                                       \
        if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }                                                    \
        if (s.x + s.y == 12345.678f) out[3000] = 1;                                                         \
    }
PKKERNEL(k_pkmul, REP8(X_PKMUL))
PKKERNEL(k_pkadd, REP8(X_PKADD))
PKKERNEL(k_pkfma, REP8(X_PKFMA))
// exp interleaved with plain VALU: does a transcendental run beside ordinary vector instructions?
#define X_MIX(i) asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %1, %1, %1" : "+v"(a##i), "+v"(b##i));
KERNEL(k_mix_exp_3add, REP8(X_MIX))

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int NV>
__global__ void k_mfma_valu(unsigned long long* out, float seed) {
    f4 c0 = {seed, seed, seed, seed}, c1 = c0, c2 = c0, c3 = c0;
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x + i); b[i] = (__bf16)(seed * 0.5f + i); }
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 512; ++i) {
#define ONE(C, R0, R1, R2, R3) \
        C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, C, 0, 0, 0); \
        if (NV >= 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(R0)); \
        if (NV >= 2) asm volatile("v_add_f32 %0, %0, %0" : "+v"(R1)); \
        if (NV >= 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(R2)); \
        if (NV >= 4) asm volatile("v_add_f32 %0, %0, %0" : "+v"(R3)); \
        if (NV >= 6) { asm volatile("v_add_f32 %0, %0, %0" : "+v"(R0)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(R1)); } \
        if (NV >= 8) { asm volatile("v_add_f32 %0, %0, %0" : "+v"(R2)); asm volatile("v_add_f32 %0, %0, %0" : "+v"(R3)); }
        ONE(c0, a0, a1, a2, a3) ONE(c1, a4, a5, a6, a7) ONE(c2, a0, a1, a2, a3) ONE(c3, a4, a5, a6, a7)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3];
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
    if (s == 12345.678f) out[3000] = 1;
}

// ---- round 6: the OTHER matrix instruction.  v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD) with 0..8 independent vector fillers behind it, at
// 1 and 2 waves per SIMD: MI355X_MICROARCH.md reports <= 5 single-issue fillers HIDDEN per gap at one wave per SIMD and packed fp32 VALU as an
// anti-lever there.  KIND: 0 v_add_f32, 1 v_fma_f32, 2 v_pk_add_f32, 3 v_pk_fma_f32, 4 v_exp_f32, 5 v_cvt_pk_bf16_f32, 6 v_max3_f32,
// 7 v_perm_b32, 8 ds_read_b64 (LDS), 9 v_mul_f32 + v_add_f32 alternating
typedef float f16v __attribute__((ext_vector_type(16)));
template <int KIND>
__device__ __forceinline__ void filler(float& r, f2& p, float& q, unsigned lds_addr) {
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r));
    if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r));
    if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p));
    if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p));
    if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
    if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(q));
    if (KIND == 6) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r) : "v"(q));
    if (KIND == 7) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r) : "v"(q));
    if (KIND == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(p) : "v"(lds_addr));
}
template <int NV, int KIND, int BIG>
__global__ void k_mfma2_valu(unsigned long long* out, float seed) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = seed;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(threadIdx.x & 63) * 8u;
    f16v C0, C1, C2, C3;
    f4 c0 = {seed, seed, seed, seed}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < 16; ++i) { C0[i] = seed + i; C1[i] = seed - i; C2[i] = seed * i; C3[i] = seed; }
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x + i); b[i] = (__bf16)(seed * 0.5f + i); }
    float r[8]; f2 p[8]; float q = seed * 3.f;
    for (int i = 0; i < 8; ++i) { r[i] = seed + threadIdx.x + i; p[i] = f2{seed + i, seed - i}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 512; ++i) {
#define GRP(CB, CS)                                                                                       \
        if (BIG) CB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, CB, 0, 0, 0);                          \
        else CS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, CS, 0, 0, 0);                              \
        _Pragma("unroll") for (int f = 0; f < NV; ++f) {                                                   \
            if (KIND == 9) { if (f & 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[f & 7])); else asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[f & 7])); } \
            else filler<KIND>(r[f & 7], p[f & 7], q, lds_addr);                                            \
        }
        GRP(C0, c0) GRP(C1, c1) GRP(C2, c2) GRP(C3, c3)
    }
    if (KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = c0[0] + c1[1] + c2[2] + c3[3] + C0[0] + C1[5] + C2[9] + C3[15] + q;
    for (int i = 0; i < 8; ++i) s += r[i] + p[i].x + p[i].y;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
    if (s == 12345.678f) out[3000] = 1;
}
static const char* KIND_NAME[] = {"v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_perm_b32",
                                  "ds_read_b64", "v_mul_f32/v_add_f32"};
template <int NV, int KIND, int BIG>
void run_mfma2(unsigned long long* d, unsigned long long* h) {
    for (int waves = 1; waves <= 3; ++waves) {
        hipMemset(d, 0, 8 * 4096);
        hipLaunchKernelGGL((k_mfma2_valu<NV, KIND, BIG>), dim3(8), dim3(256 * waves), 0, 0, d, 1.0f);
        hipLaunchKernelGGL((k_mfma2_valu<NV, KIND, BIG>), dim3(8), dim3(256 * waves), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 8 * 8 * 32, hipMemcpyDeviceToHost);
        double c = 0;
        for (int b = 0; b < 8; ++b) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < 4 * waves; ++w) { lo = h[(b * 16 + w) * 2] < lo ? h[(b * 16 + w) * 2] : lo; hi = h[(b * 16 + w) * 2 + 1] > hi ? h[(b * 16 + w) * 2 + 1] : hi; }
            c += (double)(hi - lo);
        }
        c /= 8;
        printf("1 MFMA %s + %d x %-20s %d wave(s)/SIMD: %7.2f cycles per group per wave, %6.2f per SIMD\n", BIG ? "32x32x16" : "16x16x32", NV, KIND_NAME[KIND], waves,
               c / 2048.0, c / 2048.0 / waves);
    }
}
template <int KIND, int BIG>
void sweep_kind(unsigned long long* d, unsigned long long* h) {
    run_mfma2<1, KIND, BIG>(d, h); run_mfma2<2, KIND, BIG>(d, h); run_mfma2<3, KIND, BIG>(d, h); run_mfma2<4, KIND, BIG>(d, h); run_mfma2<5, KIND, BIG>(d, h);
    run_mfma2<6, KIND, BIG>(d, h); run_mfma2<8, KIND, BIG>(d, h);
}
#define RUNM(NV)                                                                              \
    for (int waves = 1; waves <= 4; ++waves) {                                                  \
        hipMemset(d, 0, 8 * 4096);                                                              \
        hipLaunchKernelGGL(k_mfma_valu<NV>, dim3(8), dim3(256 * waves), 0, 0, d, 1.0f);         \
        hipLaunchKernelGGL(k_mfma_valu<NV>, dim3(8), dim3(256 * waves), 0, 0, d, 1.0f);         \
        hipDeviceSynchronize();                                                                 \
        hipMemcpy(h, d, 8 * 8 * 32, hipMemcpyDeviceToHost);                                          \
        double c = 0; for (int b = 0; b < 8; ++b) { unsigned long long lo = ~0ull, hi = 0; for (int w = 0; w < 4 * waves; ++w) { lo = h[(b * 16 + w) * 2] < lo ? h[(b * 16 + w) * 2] : lo; hi = h[(b * 16 + w) * 2 + 1] > hi ? h[(b * 16 + w) * 2 + 1] : hi; } c += (double)(hi - lo); }                            \
        c /= 8;                                                                                 \
        printf("1 MFMA 16x16x32 + %d v_add   %d wave(s)/SIMD: %7.2f cycles per group per wave, %6.2f per SIMD\n", NV, waves, c / 2048.0, c / 2048.0 / waves); \
    }
#define RUN(k, n_inst, label)                                                                  \
    for (int waves = 1; waves <= 4; ++waves) {                                                  \
        hipMemset(d, 0, 8 * 4096);                                                              \
        hipLaunchKernelGGL(k, dim3(8), dim3(256 * waves), 0, 0, d, 1.0f);                       \
        hipLaunchKernelGGL(k, dim3(8), dim3(256 * waves), 0, 0, d, 1.0f);                       \
        hipDeviceSynchronize();                                                                 \
        hipMemcpy(h, d, 8 * 8 * 32, hipMemcpyDeviceToHost);                                          \
        double c = 0; for (int b = 0; b < 8; ++b) { unsigned long long lo = ~0ull, hi = 0; for (int w = 0; w < 4 * waves; ++w) { lo = h[(b * 16 + w) * 2] < lo ? h[(b * 16 + w) * 2] : lo; hi = h[(b * 16 + w) * 2 + 1] > hi ? h[(b * 16 + w) * 2 + 1] : hi; } c += (double)(hi - lo); }                            \
        c /= 8;                                                                                 \
        printf("%-22s %d wave(s)/SIMD: %7.2f cycles per wave-instruction (per SIMD: %6.2f)\n", label, waves, c / (512.0 * (n_inst)), c / (512.0 * (n_inst)) / waves); \
    }
int main() {
    unsigned long long *d, h[8 * 32];
    hipMalloc(&d, 8 * 4096);
    RUN(k_add, 8, "v_add_f32") RUN(k_fma, 8, "v_fma_f32") RUN(k_and, 8, "v_and_b32") RUN(k_shl, 8, "v_lshlrev_b32") RUN(k_max3, 8, "v_max3_f32")
    RUN(k_cvt, 8, "v_cvt_pk_bf16_f32") RUN(k_exp, 8, "v_exp_f32") RUN(k_rcp, 8, "v_rcp_f32") RUN(k_pkmul, 8, "v_pk_mul_f32") RUN(k_pkadd, 8, "v_pk_add_f32")
    RUN(k_pkfma, 8, "v_pk_fma_f32") RUN(k_mix_exp_3add, 32, "exp + 3 add (per inst)")
    RUN(k_align, 8, "v_alignbit_b32") RUN(k_perm, 8, "v_perm_b32") RUN(k_sdwa, 8, "v_mov_b32_sdwa W1<-W0") RUN(k_lshlor, 8, "v_lshl_or_b32") RUN(k_mulu24, 8, "v_mul_u32_u24")
    RUN(k_bfi, 8, "v_bfi_b32") RUN(k_mul, 8, "v_mul_f32") RUN(k_max, 8, "v_max_f32") RUN(k_addu, 8, "v_add_u32") RUN(k_lshl1, 8, "v_lshlrev_b32 (2 regs)") RUN(k_addsdwa, 8, "v_add_f32_sdwa")
    RUNM(0) RUNM(2) RUNM(4) RUNM(8)
    printf("---- round 6: fillers behind v_mfma_f32_32x32x16_bf16 (BIG) and, same harness, v_mfma_f32_16x16x32_bf16\n");
    run_mfma2<0, 0, 1>(d, h); run_mfma2<0, 0, 0>(d, h);
    sweep_kind<0, 1>(d, h); sweep_kind<1, 1>(d, h); sweep_kind<2, 1>(d, h); sweep_kind<3, 1>(d, h); sweep_kind<4, 1>(d, h); sweep_kind<5, 1>(d, h);
    sweep_kind<6, 1>(d, h); sweep_kind<7, 1>(d, h); sweep_kind<8, 1>(d, h); sweep_kind<9, 1>(d, h);
    sweep_kind<0, 0>(d, h); sweep_kind<2, 0>(d, h); sweep_kind<4, 0>(d, h); sweep_kind<5, 0>(d, h); sweep_kind<8, 0>(d, h);
    return 0;
}
