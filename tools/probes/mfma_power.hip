// Sustained (power-capped) rate of the two bf16 MFMA shapes with register-resident operands: does v_mfma_f32_32x32x16_bf16 (half the
// operand-register reads per flop) sustain a higher rate than v_mfma_f32_16x16x32_bf16 on random data?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <initializer_list>
typedef __attribute__((ext_vector_type(8))) __bf16 ab_t;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) _Float16 ah_t;

template <int SHAPE> __global__ __launch_bounds__(256) void k(const uint4* in, float* out, int iters) {
    const int lane = threadIdx.x;
    uint4 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(lane * 8 + i) & 4095]; b[i] = in[(lane * 8 + i + 2048) & 4095]; }
    float s = 0.f;
    if (SHAPE == 17) {                // 16x16x32 in fp16 (round 4: the fp16 build of every GEMM launch runs 8-12 % below the bf16 one)
        f4 acc[8][8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ah_t, a[i]), __builtin_bit_cast(ah_t, b[j]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else if (SHAPE == 16) {
        f4 acc[8][8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ab_t, a[i]), __builtin_bit_cast(ab_t, b[j]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f16v acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ab_t, a[i + 4 * kk]), __builtin_bit_cast(ab_t, b[j + 4 * kk]), acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    uint4* in; float* out;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, 64);
    uint16_t* h = (uint16_t*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
    const int iters = 2000;
    // argv[2] = "f16": operands are fp16 values of the same distribution, shapes 16x16x32 bf16 vs fp16
    const bool f16 = argc > 2 && !strcmp(argv[2], "f16");
    if (f16) {
        for (int i = 0; i < 4096 * 8; ++i) { _Float16 v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f); memcpy(&h[i], &v, 2); }
    }
    uint4* in16; hipMalloc(&in16, 4096 * 16); hipMemcpy(in16, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int shape : (f16 ? std::initializer_list<int>{16, 17, 16, 17} : std::initializer_list<int>{16, 32, 16, 32})) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double total_ms = 0; long launches = 0; float last = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) {
                if (shape == 17) hipLaunchKernelGGL(k<17>, dim3(256 * 4), dim3(256), 0, 0, in16, out, iters);
                else if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256 * 4), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<32>, dim3(256 * 4), dim3(256), 0, 0, in, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&last, e0, e1); total_ms += last; launches += 10;
        }
        // flops per launch: blocks * waves * iters * mfmas * flops
        const double fl = shape != 32 ? 1024.0 * 4 * iters * 64 * (2.0 * 16 * 16 * 32) : 1024.0 * 4 * iters * 32 * (2.0 * 32 * 32 * 16);
        printf("mfma %s%dx%d: last-10-launch rate %.0f TF/s, mean %.0f TF/s over %.1f s\n", shape == 17 ? "fp16 " : "", shape == 17 ? 16 : shape, shape == 17 ? 16 : shape, fl * 10 / last / 1e9, fl * launches / total_ms / 1e9, total_ms / 1e3);
    }
    return 0;
}
