// Forward values of the RES / REC training losses.  HBM-bound reductions over the post-processed mask
// logits; everything in fp32 like the reference (postprocess_masks returns fp32, sam.py:159-172).
//
// reference: models/loss.py:45-69 dice_loss, :72-89 sigmoid_ce_loss, :92-110 bbox_l1_loss / bbox_giou_loss (+ :6-42 the IoU
// helpers and torchvision.ops.boxes.box_area), combined per sample by models/ullava.py:283-312.
#include "ull_common.h"

namespace {

constexpr int LOSS_CHUNKS = 64;       // partial sums per mask; the host adds them (fixed order -> deterministic result)

// part[m][chunk] = { sum bce(x, t), sum (sigmoid(x)/scale) * t, sum sigmoid(x)/scale, sum t/scale } over the chunk's pixels
__global__ __launch_bounds__(256) void mask_loss_kernel(const float* __restrict__ logits, const float* __restrict__ target, long hw,
                                                        float inv_scale, float* __restrict__ part) {
    __shared__ float red[4][4];
    const int m = blockIdx.y, chunk = blockIdx.x;
    const float* x = logits + (long)m * hw;
    const float* t = target + (long)m * hw;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)chunk * 256 + threadIdx.x; i < hw; i += (long)LOSS_CHUNKS * 256) {
        const float xv = x[i], tv = t[i];
        // F.binary_cross_entropy_with_logits: (1 - t) * x - log_sigmoid(x),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
        s[0] += (1.f - tv) * xv - (fminf(xv, 0.f) - log1pf(expf(-fabsf(xv))));
        const float sg = (1.f / (1.f + expf(-xv))) * inv_scale;          // inputs.sigmoid() / scale
        s[1] += sg * tv;
        s[2] += sg;
        s[3] += tv * inv_scale;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j] = wave_sum(s[j]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < 4) part[((long)m * LOSS_CHUNKS + chunk) * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0] = sum_i sum_c |p_ic - g_ic|;  out[1] = sum over boxes with x1 >= x0 and y1 >= y0 of (1 - GIoU(p_i, g_i))
template <int DT>
__global__ void box_loss_kernel(const void* __restrict__ pred, const float* __restrict__ gt, int n, float* __restrict__ out) {
    float l1 = 0.f, gl = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) {
        float p[4], g[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            p[c] = load_dt<DT>(pred, i * 4 + c);
            g[c] = gt[i * 4 + c];
            l1 += fabsf(p[c] - g[c]);
        }
        if (p[2] >= p[0] && p[3] >= p[1]) {
            const float a1 = (p[2] - p[0]) * (p[3] - p[1]), a2 = (g[2] - g[0]) * (g[3] - g[1]);
            const float iw = fmaxf(fminf(p[2], g[2]) - fmaxf(p[0], g[0]), 0.f), ih = fmaxf(fminf(p[3], g[3]) - fmaxf(p[1], g[1]), 0.f);
            const float inter = iw * ih, uni = a1 + a2 - inter;
            const float cw = fmaxf(fmaxf(p[2], g[2]) - fminf(p[0], g[0]), 0.f), ch = fmaxf(fmaxf(p[3], g[3]) - fminf(p[1], g[1]), 0.f);
            const float area = cw * ch;
            gl += 1.f - (inter / uni - (area - uni) / area);
        }
    }
    l1 = wave_sum(l1);
    gl = wave_sum(gl);
    if (threadIdx.x == 0) { out[0] = l1; out[1] = gl; }
}

}  // namespace

// part: float [n_masks, 64, 4] (see mask_loss_kernel); scale = dice_loss's `scale` (1000).
extern "C" int ull_mask_loss_sums_f32(const void* logits, const void* target, int64_t n_masks, int64_t hw, float scale, void* part, void* stream) {
    if (!logits || !target || !part || n_masks <= 0 || hw <= 0 || scale == 0.f) return ULL_ERR_ARG;
    hipLaunchKernelGGL(mask_loss_kernel, dim3(LOSS_CHUNKS, (unsigned)n_masks), dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                       (const float*)target, hw, 1.0f / scale, (float*)part);
    return ull_check_launch();
}

// pred [n, 4] of dtype pred_dtype (ULL_DT_*), gt [n, 4] fp32, xyxy; out float[2] = {L1 sum, sum of (1 - GIoU) over well-formed predictions}.
extern "C" int ull_box_losses_f32(const void* pred, int pred_dtype, const void* gt, int64_t n, void* out, void* stream) {
    if (!pred || !gt || !out || n <= 0 || pred_dtype < 0 || pred_dtype > 2) return ULL_ERR_ARG;
    const hipStream_t st = (hipStream_t)stream;
    if (pred_dtype == ULL_DT_BF16) hipLaunchKernelGGL(box_loss_kernel<ULL_DT_BF16>, dim3(1), dim3(64), 0, st, pred, (const float*)gt, (int)n, (float*)out);
    else if (pred_dtype == ULL_DT_F16) hipLaunchKernelGGL(box_loss_kernel<ULL_DT_F16>, dim3(1), dim3(64), 0, st, pred, (const float*)gt, (int)n, (float*)out);
    else hipLaunchKernelGGL(box_loss_kernel<ULL_DT_F32>, dim3(1), dim3(64), 0, st, pred, (const float*)gt, (int)n, (float*)out);
    return ull_check_launch();
}
