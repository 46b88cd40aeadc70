// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = its own element index; every lane reads with its own address and
// the host prints what each lane received.  Build: hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint32_t* out, int pitch_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // lane i of a 16-lane group g: row = 4 * g + i / 4, 4 consecutive elements at column 4 * (i % 4)
    const int g = lane >> 4, i = lane & 15;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + ((4 * g + i / 4) * pitch_elems + 4 * (i % 4)) * 2;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 2] = v.x;
    out[lane * 2 + 1] = v.y;
}
int main() {
    uint32_t* d;
    hipMalloc(&d, 64 * 2 * 4);
    const int pitch = 80;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch);
    uint32_t h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        const int e[4] = {(int)(h[2 * l] & 0xffff), (int)(h[2 * l] >> 16), (int)(h[2 * l + 1] & 0xffff), (int)(h[2 * l + 1] >> 16)};
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  (r%2d,c%2d)", e[j] / pitch, e[j] % pitch);
        printf("\n");
    }
    return 0;
}
