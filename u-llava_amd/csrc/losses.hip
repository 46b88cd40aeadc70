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

// ---- backward of the two reductions above and of the bilinear resize (fp32 throughout, like the forward) -----------------------------
// dlogits[m][i] = g[m][0] * (sigmoid(x) - t) + (g[m][1] * t + g[m][2]) * sigmoid(x) * (1 - sigmoid(x)) / scale
__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ target, const float* __restrict__ g,
                                                            long hw, float inv_scale, float* __restrict__ dl) {
    const int m = blockIdx.y;
    const float g0 = g[m * 4], g1 = g[m * 4 + 1], g2 = g[m * 4 + 2];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
        const float xv = logits[(long)m * hw + i], tv = target[(long)m * hw + i];
        const float sg = 1.f / (1.f + expf(-xv));
        dl[(long)m * hw + i] = g0 * (sg - tv) + (g1 * tv + g2) * sg * (1.f - sg) * inv_scale;
    }
}

// d/dpred of {sum |p - g|, sum (1 - GIoU)} weighted by gw[0], gw[1]; sub-gradients of min / max / clamp follow torch's conventions.
template <int DT>
__global__ void box_loss_bwd_kernel(const void* __restrict__ pred, const float* __restrict__ gt, int n, const float* __restrict__ gw,
                                    float* __restrict__ dp) {
    for (int i = threadIdx.x; i < n; i += 64) {
        float p[4], g[4], d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            p[c] = load_dt<DT>(pred, i * 4 + c);
            g[c] = gt[i * 4 + c];
            d[c] = gw[0] * ((p[c] > g[c]) ? 1.f : ((p[c] < g[c]) ? -1.f : 0.f));
        }
        if (p[2] >= p[0] && p[3] >= p[1]) {
            const float w1 = p[2] - p[0], h1 = p[3] - p[1];
            const float a1 = w1 * h1, a2 = (g[2] - g[0]) * (g[3] - g[1]);
            const float iwr = fminf(p[2], g[2]) - fmaxf(p[0], g[0]), ihr = fminf(p[3], g[3]) - fmaxf(p[1], g[1]);
            const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
            const float inter = iw * ih, uni = a1 + a2 - inter;
            const float cwr = fmaxf(p[2], g[2]) - fminf(p[0], g[0]), chr = fmaxf(p[3], g[3]) - fminf(p[1], g[1]);
            const float cw = fmaxf(cwr, 0.f), ch = fmaxf(chr, 0.f);
            const float area = cw * ch;
            // giou = inter / uni - 1 + uni / area
            const float d_inter = 1.f / uni, d_uni = -inter / (uni * uni) + 1.f / area, d_area = -uni / (area * area);
            // partials of iw, ih, cw, ch, a1 wrt p0..p3
            float diw[4] = {0.f, 0.f, 0.f, 0.f}, dih[4] = {0.f, 0.f, 0.f, 0.f}, dcw[4] = {0.f, 0.f, 0.f, 0.f}, dch[4] = {0.f, 0.f, 0.f, 0.f};
            if (iwr > 0.f) { diw[0] = (p[0] > g[0]) ? -1.f : 0.f; diw[2] = (p[2] < g[2]) ? 1.f : 0.f; }
            if (ihr > 0.f) { dih[1] = (p[1] > g[1]) ? -1.f : 0.f; dih[3] = (p[3] < g[3]) ? 1.f : 0.f; }
            if (cwr > 0.f) { dcw[0] = (p[0] < g[0]) ? -1.f : 0.f; dcw[2] = (p[2] > g[2]) ? 1.f : 0.f; }
            if (chr > 0.f) { dch[1] = (p[1] < g[1]) ? -1.f : 0.f; dch[3] = (p[3] > g[3]) ? 1.f : 0.f; }
            const float da1[4] = {-h1, -w1, h1, w1};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float dinter = diw[c] * ih + iw * dih[c];
                const float duni = da1[c] - dinter;
                const float darea = dcw[c] * ch + cw * dch[c];
                const float dgiou = d_inter * dinter + d_uni * duni + d_area * darea;
                d[c] -= gw[1] * dgiou;                       // loss term is (1 - giou)
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) dp[i * 4 + c] = d[c];
    }
}

// adjoint of bilinear_kernel (sam.hip): din[n][y][x] += weights * dout, din fp32 zeroed by the caller (atomics)
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, long in_img_stride, long in_row_stride,
                                                           int in_h, int in_w, int out_h, int out_w, long total) {
    const float sh = (float)in_h / (float)out_h, sw = (float)in_w / (float)out_w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % out_w), oy = (int)((i / out_w) % out_h);
        const long n = i / ((long)out_w * out_h);
        float fy = sh * ((float)oy + 0.5f) - 0.5f, fx = sw * ((float)ox + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < in_h - 1 ? 1 : 0), x1 = x0 + (x0 < in_w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float g = dout[i];
        float* b = din + n * in_img_stride;
        atomicAdd(b + (long)y0 * in_row_stride + x0, hy * hx * g);
        atomicAdd(b + (long)y0 * in_row_stride + x1, hy * lx * g);
        atomicAdd(b + (long)y1 * in_row_stride + x0, ly * hx * g);
        atomicAdd(b + (long)y1 * in_row_stride + x1, ly * lx * g);
    }
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

// Backward of ull_mask_loss_sums_f32: g float [n_masks, 4] = gradients of the four per-mask sums; dlogits float [n_masks, hw].
extern "C" int ull_mask_loss_sums_bwd_f32(const void* logits, const void* target, const void* g, int64_t n_masks, int64_t hw, float scale,
                                          void* dlogits, void* stream) {
    if (!logits || !target || !g || !dlogits || n_masks <= 0 || hw <= 0 || scale == 0.f) return ULL_ERR_ARG;
    hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3((unsigned)((hw + 255) / 256 < 1024 ? (hw + 255) / 256 : 1024), (unsigned)n_masks), dim3(256), 0,
                       (hipStream_t)stream, (const float*)logits, (const float*)target, (const float*)g, hw, 1.0f / scale, (float*)dlogits);
    return ull_check_launch();
}

// Backward of ull_box_losses_f32: gw float[2] = gradients of {L1 sum, GIoU-loss sum}; dpred float [n, 4].
extern "C" int ull_box_losses_bwd_f32(const void* pred, int pred_dtype, const void* gt, int64_t n, const void* gw, void* dpred, void* stream) {
    if (!pred || !gt || !gw || !dpred || n <= 0 || pred_dtype < 0 || pred_dtype > 2) return ULL_ERR_ARG;
    const hipStream_t st = (hipStream_t)stream;
    if (pred_dtype == ULL_DT_BF16) hipLaunchKernelGGL(box_loss_bwd_kernel<ULL_DT_BF16>, dim3(1), dim3(64), 0, st, pred, (const float*)gt, (int)n, (const float*)gw, (float*)dpred);
    else if (pred_dtype == ULL_DT_F16) hipLaunchKernelGGL(box_loss_bwd_kernel<ULL_DT_F16>, dim3(1), dim3(64), 0, st, pred, (const float*)gt, (int)n, (const float*)gw, (float*)dpred);
    else hipLaunchKernelGGL(box_loss_bwd_kernel<ULL_DT_F32>, dim3(1), dim3(64), 0, st, pred, (const float*)gt, (int)n, (const float*)gw, (float*)dpred);
    return ull_check_launch();
}

// Adjoint of ull_bilinear_f32: din float32 (zeroed by the caller) [n] images of stride in_img_stride / rows in_row_stride, crop in_h x in_w.
extern "C" int ull_bilinear_bwd_f32(const void* dout, void* din, int64_t in_img_stride, int64_t in_row_stride, int64_t in_h, int64_t in_w,
                                    int64_t n, int64_t out_h, int64_t out_w, void* stream) {
    if (!dout || !din || n <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return ULL_ERR_ARG;
    const long total = n * out_h * out_w;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dout, (float*)din, in_img_stride, in_row_stride, (int)in_h, (int)in_w, (int)out_h, (int)out_w, total);
    return ull_check_launch();
}
