// Probe: what does one grid-wide barrier cost on MI355X (256 blocks of 1024 threads, one per CU, co-resident by a cooperative launch)?
// Monotonic counter in device memory: release fence, one atomic per block, spin on the counter (bounded), acquire fence.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int FENCE, int SLEEP>
__global__ __launch_bounds__(1024) void barrier_loop(unsigned* counter, int iters, unsigned* fail, float* sink) {
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        acc += __sinf((float)(threadIdx.x + i));                 // a little work between barriers
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCE) __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * nb;
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (SLEEP) __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 24)) { *fail = 1; break; }
            }
            if (FENCE) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
    }
    if (acc == 12345.f) sink[0] = acc;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int nb = prop.multiProcessorCount;
    unsigned *counter, *fail;
    float* sink;
    hipMalloc(&counter, 4); hipMalloc(&fail, 4); hipMalloc(&sink, 4);
    const void* kernels[4] = {(const void*)barrier_loop<1, 1>, (const void*)barrier_loop<1, 0>, (const void*)barrier_loop<0, 1>, (const void*)barrier_loop<0, 0>};
    const char* names[4] = {"fences + s_sleep", "fences, busy spin", "no fences, s_sleep", "no fences, busy spin"};
    for (int rep = 0; rep < 8; ++rep) {
        hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        void* args[] = {&counter, &iters, &fail, &sink};
        hipEventRecord(e0);
        hipError_t rc = hipLaunchCooperativeKernel(kernels[rep & 3], dim3(nb), dim3(1024), args, 0, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned f = 0;
        hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("[%s] launch rc %d, %d blocks x 1024 threads, %d barriers: %.3f ms = %.2f us per barrier, fail %u\n", names[rep & 3], (int)rc, nb, iters, ms, ms * 1e3 / iters, f);
    }
    return 0;
}
