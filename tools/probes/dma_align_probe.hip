// Does the alignment of the 16-byte lanes of an LDS-DMA request (global_load_lds_dwordx4) change what a CU can pull from the L2?
// Every wave streams the same 1-MB window into its own LDS KiB -- the 8 waves of a CU ask for the same lines at about the same time, so
// this measures the CU's address-coalescing / L1 path (64 B per clock when aligned), not the L2: lane l of request i reads 16 bytes at
//   base + (i * 64 + l) * STRIDE   with STRIDE = 16 (aligned, contiguous), 32 (aligned, every other chunk) or 28 (the patchify gather: a
//   patch's 14 bf16 pixels; 4-byte alignment, neighbouring lanes overlap by 4 bytes).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_align_probe tools/probes/dma_align_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int STRIDE>
__global__ __launch_bounds__(512) void probe(const char* src, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    const char* p = src + (size_t)lane * STRIDE + (size_t)(blockIdx.x & 7) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const char* a = p + ((size_t)((it * 8 + i) & 1023) * 64) * STRIDE;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(a), "s"(lds) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(uint32_t*)smem;
}

// L2 -> LDS: every wave walks its OWN 12-KiB window (96 KiB per CU: three times the L1, 3 MiB per XCD: inside its 4-MiB L2), DEPTH
// requests in flight per wave.  What a CU gets when all of them stream from their L2 at once -- the GEMM K-loop's situation.
template <int DEPTH>
__global__ __launch_bounds__(512) void probe_l2(const char* src, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    // consecutive block ids sit on different XCDs: blocks b, b + 8, ... share one L2 -> windows laid out per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const char* p = src + ((size_t)xcd * 32 + slot) * (96 * 1024) + (size_t)wave * (12 * 1024) + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const char* a = p + (size_t)((it * DEPTH + i) % 12) * 1024;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(a), "s"(lds) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(uint32_t*)smem;
}

template <int DEPTH>
void run_l2(const char* buf, uint32_t* sink, int n_cu) {
    const int iters = 16000 / DEPTH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe_l2<DEPTH>, dim3(n_cu), dim3(512), 8192, 0, buf, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)n_cu * 8 * iters * DEPTH * 1024;
        if (rep == 1)
            printf("L2 -> LDS, %2d requests in flight per wave (8 waves): %7.3f ms, %6.1f GB/s per CU, %5.2f TB/s over %d CUs\n", DEPTH, ms,
                   bytes / n_cu / ms / 1e6, bytes / ms / 1e9, n_cu);
    }
}

template <int STRIDE>
void run(const char* buf, uint32_t* sink, int n_cu) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<STRIDE>, dim3(n_cu), dim3(512), 8192, 0, buf, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)n_cu * 8 /*waves*/ * iters * 8 * 1024;
        if (rep == 1)
            printf("lane stride %2d B: %7.3f ms, %6.1f GB/s per CU (%5.2f TB/s over %d CUs), %.1f B per clock and CU at 2.1 GHz\n", STRIDE, ms,
                   bytes / n_cu / ms / 1e6, bytes / ms / 1e9, n_cu, bytes / n_cu / (ms * 1e-3) / 2.1e9);
    }
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount;
    char* buf; uint32_t* sink;
    hipMalloc(&buf, 32 << 20); hipMemset(buf, 1, 32 << 20);
    hipMalloc(&sink, n_cu * 4);
    run<16>(buf, sink, n_cu);
    run<32>(buf, sink, n_cu);
    run<28>(buf, sink, n_cu);
    run<16>(buf, sink, n_cu);
    run_l2<4>(buf, sink, n_cu);
    run_l2<8>(buf, sink, n_cu);
    run_l2<16>(buf, sink, n_cu);
    run_l2<32>(buf, sink, n_cu);
    return 0;
}
