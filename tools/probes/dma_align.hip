// Probe: does global_load_lds_dwordx4 accept source addresses that are only 4-byte aligned?  (fused patchify needs 28-byte strides)
// build: hipcc --offload-arch=gfx950 -O2 -o dma_align dma_align.hip ; run: ./dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k(const uint8_t* src, uint32_t* out, int stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const void* a = src + (long)lane * stride;           // 16 bytes per lane from byte offset lane * stride
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(a), "s"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = ((uint32_t*)smem)[lane * 4 + i];
}

int main() {
    const int N = 64 * 64;
    std::vector<uint8_t> h(N);
    for (int i = 0; i < N; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t* d; uint32_t* o;
    hipMalloc(&d, N); hipMalloc(&o, 64 * 16);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    for (int stride : {16, 28, 4, 12, 2}) {
        hipMemset(o, 0, 64 * 16);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, o, stride);
        hipError_t e = hipDeviceSynchronize();
        std::vector<uint8_t> r(64 * 16);
        hipMemcpy(r.data(), o, 64 * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int b = 0; b < 16; ++b) if (r[l * 16 + b] != h[l * stride + b]) ++bad;
        printf("stride %2d: err=%d mismatching bytes=%d\n", stride, (int)e, bad);
    }
    return 0;
}
