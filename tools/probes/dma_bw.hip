// Probe: peak global->LDS (LDS-DMA) throughput per CU as a function of bytes in flight and of the source's residency.
// build: hipcc --offload-arch=gfx950 -O3 -o build/dma_bw tools/probes/dma_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__device__ __forceinline__ void glds16(const void* g, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
// each wave issues DEPTH 1-KiB pieces, then waits for the oldest before issuing the next (queue depth DEPTH per wave)
template <int DEPTH>
__global__ __launch_bounds__(512) void stream(const char* src, size_t region_bytes, size_t per_block_stride, int iters, int shared_by) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    // blocks in groups of `shared_by` read the same stream
    const size_t base = (size_t)(blockIdx.x / shared_by) * per_block_stride;
    size_t off = (size_t)wave * 1024 + lane * 16;
    for (int d = 0; d < DEPTH; ++d) {
        glds16(src + (base + off) % region_bytes, lds_base + (wave * DEPTH + d) * 1024);
        off += 8 * 1024;
    }
    for (int i = 0; i < iters; ++i) {
        for (int d = 0; d < DEPTH; ++d) {
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            glds16(src + (base + off) % region_bytes, lds_base + (wave * DEPTH + d) * 1024);
            off += 8 * 1024;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int DEPTH>
void run(const char* name, const char* src, size_t region, size_t stride, int shared_by) {
    const int iters = 2000 / DEPTH, blocks = 256;
    hipFuncSetAttribute((const void*)stream<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream<DEPTH><<<blocks, 512, 8 * DEPTH * 1024>>>(src, region, stride, 10, shared_by);
    hipEventRecord(e0);
    stream<DEPTH><<<blocks, 512, 8 * DEPTH * 1024>>>(src, region, stride, iters, shared_by);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * 8 * 1024.0 * DEPTH * (iters + 1);
    printf("%-34s depth %2d (%3d KiB in flight/CU): %7.2f TB/s  %6.1f GB/s/CU\n", name, DEPTH, 8 * DEPTH, bytes / ms / 1e9, bytes / ms / 1e6 / blocks);
}
int main() {
    size_t total = (size_t)2 << 30;
    char* buf; hipMalloc(&buf, total); hipMemset(buf, 1, total);
    // (a) every block re-reads one 1-MiB region: pure L2-hit path
    run<1>("L2-resident 1 MiB, all blocks", buf, 1 << 20, 0, 1); run<2>("L2-resident 1 MiB, all blocks", buf, 1 << 20, 0, 1);
    run<4>("L2-resident 1 MiB, all blocks", buf, 1 << 20, 0, 1); run<8>("L2-resident 1 MiB, all blocks", buf, 1 << 20, 0, 1);
    run<16>("L2-resident 1 MiB, all blocks", buf, 1 << 20, 0, 1);
    // (b) every block streams its own 8 MiB slice of a 2 GiB buffer: HBM streaming
    run<2>("HBM stream, private per block", buf, total, 8 << 20, 1); run<8>("HBM stream, private per block", buf, total, 8 << 20, 1);
    run<16>("HBM stream, private per block", buf, total, 8 << 20, 1);
    // (c) groups of 8 blocks (one per XCD, so no L2 sharing) and of 64 consecutive blocks stream the same data in lock step
    run<8>("stream shared by 8 consecutive blocks", buf, total, 8 << 20, 8);
    run<8>("stream shared by 64 consecutive blocks", buf, total, 8 << 20, 64);
    run<16>("stream shared by 64 consecutive blocks", buf, total, 8 << 20, 64);
    return 0;
}
