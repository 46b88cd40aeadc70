// AdamW on a shard of the flattened parameters, and the squared-gradient-norm reduction for clipping (training path).
//
// reference: the optimizer transformers.Trainer builds for train_ullava.py:273-293 (torch.optim.AdamW, decoupled weight decay, bias
// correction) under DeepSpeed ZeRO stage 2 with bf16 (configs/deepspeed/bf16_zero2.json): fp32 master weights and fp32 first / second
// moments live ONLY on the rank that owns the shard; gradients arrive reduce-scattered, updated 16-bit parameters leave by all-gather
// (u-llava_amd/optim.py).  Arithmetic follows torch's single-tensor AdamW in fp32, statement by statement:
//     p *= 1 - lr * wd;  m = m + (g - m) (1 - beta1) [lerp];  v = beta2 v + (1 - beta2) g g;
//     denom = sqrt(v) / sqrt(1 - beta2^t) + eps;  p -= (lr / (1 - beta1^t)) * m / denom
// HBM-bound: 16 bytes of state read + written per parameter and step.
#include "ull_common.h"

namespace {

template <int GDT, int PDT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v, const void* __restrict__ grad,
                                                    void* __restrict__ param_out, long n, float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float grad_scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float g = load_dt<GDT>(grad, i) * grad_scale;
        float p = master[i];
        p *= 1.0f - lr * wd;
        const float m0 = m[i];
        const float mi = m0 + (g - m0) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = beta2 * v[i] + ((1.0f - beta2) * g) * g;        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p -= (lr / bc1) * (mi / denom);
        master[i] = p; m[i] = mi; v[i] = vi;
        if (param_out != nullptr) store_dt<PDT>(param_out, i, p);
    }
}

// out[0] += sum g[i]^2 (fp32 partial sums per block, one atomic per block)
template <int GDT>
__global__ __launch_bounds__(256) void sumsq_kernel(const void* __restrict__ g, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = load_dt<GDT>(g, i);
        s += x * x;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

unsigned grid_for(long n) {
    const long b = (n + 255) / 256;
    return (unsigned)(b < 8192 ? (b > 0 ? b : 1) : 8192);
}

}  // namespace

// One AdamW step on n elements.  master / m / v: fp32 state (updated in place); grad: dtype code grad_dtype (ULL_DT_*), multiplied by
// grad_scale first (1 / world for an un-averaged sum, the clipping coefficient, or both); param_out: the updated parameters in dtype
// param_dtype (ULL_DT_BF16 / ULL_DT_F16 / ULL_DT_F32), or NULL; step >= 1 = the step count AFTER this update (bias correction).
extern "C" int ull_adamw_step_f32(void* master, void* m, void* v, const void* grad, int grad_dtype, void* param_out, int param_dtype, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    if (!master || !m || !v || !grad || n <= 0 || step < 1) return ULL_ERR_ARG;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipStream_t st = (hipStream_t)stream;
#define ULL_AW(G, P)                                                                                                                         \
    hipLaunchKernelGGL((adamw_kernel<G, P>), dim3(grid_for(n)), dim3(256), 0, st, (float*)master, (float*)m, (float*)v, grad, param_out, (long)n, lr,   \
                       beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale)
    const int key = grad_dtype * 4 + param_dtype;
    switch (key) {
        case ULL_DT_F32 * 4 + ULL_DT_F32: ULL_AW(ULL_DT_F32, ULL_DT_F32); break;
        case ULL_DT_F32 * 4 + ULL_DT_BF16: ULL_AW(ULL_DT_F32, ULL_DT_BF16); break;
        case ULL_DT_F32 * 4 + ULL_DT_F16: ULL_AW(ULL_DT_F32, ULL_DT_F16); break;
        case ULL_DT_BF16 * 4 + ULL_DT_F32: ULL_AW(ULL_DT_BF16, ULL_DT_F32); break;
        case ULL_DT_BF16 * 4 + ULL_DT_BF16: ULL_AW(ULL_DT_BF16, ULL_DT_BF16); break;
        case ULL_DT_F16 * 4 + ULL_DT_F32: ULL_AW(ULL_DT_F16, ULL_DT_F32); break;
        case ULL_DT_F16 * 4 + ULL_DT_F16: ULL_AW(ULL_DT_F16, ULL_DT_F16); break;
        default: return ULL_ERR_SHAPE;
    }
#undef ULL_AW
    return ull_check_launch();
}

// out[0] (fp32, device, caller-zeroed) += sum of squares of n gradient elements (torch.nn.utils.clip_grad_norm_'s total norm is the
// square root of this over all parameters and ranks).
extern "C" int ull_sumsq_f32(const void* g, int grad_dtype, int64_t n, void* out, void* stream) {
    if (!g || !out || n <= 0) return ULL_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = grid_for(n) < 1024 ? grid_for(n) : 1024;
    if (grad_dtype == ULL_DT_F32) hipLaunchKernelGGL(sumsq_kernel<ULL_DT_F32>, dim3(grid), dim3(256), 0, st, g, (long)n, (float*)out);
    else if (grad_dtype == ULL_DT_BF16) hipLaunchKernelGGL(sumsq_kernel<ULL_DT_BF16>, dim3(grid), dim3(256), 0, st, g, (long)n, (float*)out);
    else if (grad_dtype == ULL_DT_F16) hipLaunchKernelGGL(sumsq_kernel<ULL_DT_F16>, dim3(grid), dim3(256), 0, st, g, (long)n, (float*)out);
    else return ULL_ERR_SHAPE;
    return ull_check_launch();
}
