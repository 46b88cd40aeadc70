// What a CU's store path takes, by the shape of one wave-wide global_store_dwordx4 (64 lanes x 16 B = 1 KiB):
//   A  1 KiB contiguous                      (8 full 128-byte lines)
//   B  16 rows x 64 B, row pitch 24 KiB       (the 4-wave GEMM's direct epilogue: half lines; the other half comes with the NEXT instruction)
//   C  8 rows x 128 B, row pitch 24 KiB       (full lines: what a lane-pair exchange in the epilogue would give)
//   D  B, but the two halves of a line in two different launches' worth of distance (never merged)
// 4 waves per CU (one per SIMD, like the kernel), every block writes its own 128 KiB tile region (256 rows x 512 B) `rounds` times to
// different tiles.  build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/store_pattern_probe tools/probes/store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr long PITCH = 24576;            // bytes per output row (N = 12288 bf16)
template <int MODE>
__global__ __launch_bounds__(256) void probe(char* out, int rounds, int tiles_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 1);
    for (int r = 0; r < rounds; ++r) {
        const int tile = (blockIdx.x + r * gridDim.x);
        const long row0 = (long)(tile / tiles_n) * 256 + wm * 128, col0 = (long)(tile % tiles_n) * 512 + wn * 256;   // bytes
        char* base = out + row0 * PITCH + col0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {        // q = (h, pp): column offsets h * 128 + pp * 64 bytes
                const int h = q >> 1, pp = q & 1;
                char* p;
                if (MODE == 0) p = base + (long)(j * 4 + q) * 1024 + lane * 16;                                   // contiguous KiB (layout ignored)
                else if (MODE == 1) p = base + (long)(j * 16 + fr) * PITCH + h * 128 + pp * 64 + fg * 16;            // 16 rows x 64 B
                else p = base + (long)(j * 16 + (q & 1) * 8 + (fr >> 1)) * PITCH + h * 128 + ((fr & 1) * 4 + fg) * 16;   // 8 rows x 128 B
                *(uint4*)p = v;
            }
        }
    }
}

template <int MODE>
void run(const char* name, char* buf, int n_cu) {
    const int rounds = 16, tiles_n = 24;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(n_cu), dim3(256), 0, 0, buf, rounds, tiles_n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)n_cu * rounds * 131072.0;
        if (rep == 2) printf("%-44s: %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s  (%.2f us per 128-KiB tile)\n", name, ms, bytes / n_cu / ms / 1e6, bytes / ms / 1e9, ms * 1e3 / rounds);
    }
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount;
    char* buf;
    const size_t bytes = (size_t)((n_cu * 16 + 23) / 24 + 1) * 256 * PITCH;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    run<0>("A  1 KiB contiguous per instruction", buf, n_cu);
    run<1>("B  16 rows x 64 B (shipped epilogue)", buf, n_cu);
    run<2>("C  8 rows x 128 B (full lines)", buf, n_cu);
    run<1>("B  again", buf, n_cu);
    // the same with ONE block per 8 CUs' worth of time, i.e. 32 blocks: is it the CU or the memory system?
    run<1>("B  32 blocks only", buf, 32);
    run<2>("C  32 blocks only", buf, 32);
    return 0;
}
