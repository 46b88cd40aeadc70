// Image pre/post-processing either side of the forward path, on the device.  HBM-bound byte/integer
// work: every result is bit-exact against the host libraries the reference calls.
//
// reference (all on the host CPU, per image):
//   dataset/processors/clip_processor.py:82-95   CLIPImageProcessor: PIL bicubic resize -> center crop -> /255 -> normalize
//   models/segment_anything/utils/transforms.py:27-35 + dataset/tools/mask_toolbox.py:15-25   PIL bilinear resize of the
//       uint8 image to longest side 1024, (x - mean) / std, zero pad to 1024 x 1024
//   trainers/ullava_trainer.py:40-52 + evaluation/tools.py:29-41   (logits > 0), intersection / union pixel counts
//
// Resampling = Pillow libImaging/Resample.c: two separable passes (horizontal, then vertical) with 22-bit fixed-point taps, a
// uint8 image between the passes.  The taps (bounds + integer coefficients) are computed by the host with Pillow's exact
// double-precision recipe (u-llava_amd/preprocess.py) -- a few KB per image size -- so the kernels are pure integer arithmetic.
#include "ull_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// One pass along `axis` of a uint8 [H, W, C] image.  One thread per output byte; the fastest-varying thread index walks
// (x, c) of an output row, so the vertical pass reads whole input rows coalesced and the horizontal pass re-reads its
// (overlapping) tap windows from L1/L2.
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int C, int axis, int out_size,
                                                          const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                          uint8_t* __restrict__ dst) {
    const int OW = axis == 1 ? out_size : W, OH = axis == 0 ? out_size : H;
    const long total = (long)OH * OW * C;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const int x = (int)((i / C) % OW);
    const int y = (int)(i / ((long)C * OW));
    const int o = axis == 1 ? x : y;
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int32_t* k = kk + (long)o * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    if (axis == 1) {
        const uint8_t* p = src + ((long)y * W + first) * C + c;
        for (int t = 0; t < n; ++t) acc += (int)p[(long)t * C] * k[t];
    } else {
        const uint8_t* p = src + ((long)first * W + x) * C + c;
        for (int t = 0; t < n; ++t) acc += (int)p[(long)t * W * C] * k[t];
    }
    int v = acc >> PRECISION_BITS;                     // clip8: arithmetic shift, then clamp
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[i] = (uint8_t)v;
}

// dst[c][y][x] = lut[c][src[top + y][left + x][c]] inside the copy_h x copy_w window, 0 outside (SAM's zero padding).
template <int DT>
__global__ __launch_bounds__(256) void u8_lut_chw_kernel(const uint8_t* __restrict__ src, int W, int C, int top, int left,
                                                         const float* __restrict__ lut, void* __restrict__ dst, int OH, int OW, int copy_h,
                                                         int copy_w) {
    __shared__ float tab[3 * 256];
    for (int i = threadIdx.x; i < C * 256; i += blockDim.x) tab[i] = lut[i];
    __syncthreads();
    const long total = (long)C * OH * OW;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % OW), y = (int)((i / OW) % OH), c = (int)(i / ((long)OW * OH));
    float v = 0.f;
    if (y < copy_h && x < copy_w) v = tab[c * 256 + src[((long)(top + y) * W + left + x) * C + c]];
    store_dt<DT>(dst, i, v);
}

// counts[m][0..5] += {inter0, inter1, out0, out1, tgt0, tgt1} of mask m, with out = logits > 0 and pixels whose target is
// `ignore` dropped from all three histograms (evaluation/tools.py:35-40).
__global__ __launch_bounds__(256) void mask_iou_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ target, long hw,
                                                       int ignore, int32_t* __restrict__ counts) {
    const int m = blockIdx.y;
    const float* lp = logits + (long)m * hw;
    const uint8_t* tp = target + (long)m * hw;
    int c[6] = {0, 0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
        const int t = tp[i];
        if (t == ignore) continue;
        const int o = lp[i] > 0.f ? 1 : 0;
        c[2 + o] += 1;
        if (t < 2) c[4 + t] += 1;
        if (o == t) c[o] += 1;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        int v = c[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(counts + m * 6 + j, v);
    }
}

}  // namespace

extern "C" int ull_resample_u8(const void* src, int64_t H, int64_t W, int64_t C, int axis, int64_t out_size, const void* bounds,
                               const void* coeffs, int64_t ksize, void* dst, void* stream) {
    if (!src || !dst || !bounds || !coeffs || H <= 0 || W <= 0 || C <= 0 || out_size <= 0 || ksize <= 0) return ULL_ERR_ARG;
    if (axis != 0 && axis != 1) return ULL_ERR_ARG;
    const long total = (axis == 1 ? H * out_size : out_size * W) * C;
    hipLaunchKernelGGL(resample_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (int)H,
                       (int)W, (int)C, axis, (int)out_size, (const int32_t*)bounds, (const int32_t*)coeffs, (int)ksize, (uint8_t*)dst);
    return ull_check_launch();
}

extern "C" int ull_u8_lut_chw(const void* src, int64_t H, int64_t W, int64_t C, int64_t top, int64_t left, const void* lut, void* dst,
                              int64_t OH, int64_t OW, int64_t copy_h, int64_t copy_w, int out_dtype, void* stream) {
    if (!src || !lut || !dst || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || out_dtype < 0 || out_dtype > 2) return ULL_ERR_ARG;
    if (C != 3 || top < 0 || left < 0 || copy_h > OH || copy_w > OW || top + copy_h > H || left + copy_w > W) return ULL_ERR_SHAPE;
    const long total = C * OH * OW;
    const dim3 grid((unsigned)((total + 255) / 256));
#define ULL_LAUNCH_LUT(DT)                                                                                                              \
    hipLaunchKernelGGL(u8_lut_chw_kernel<DT>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (int)W, (int)C, (int)top, \
                       (int)left, (const float*)lut, dst, (int)OH, (int)OW, (int)copy_h, (int)copy_w)
    if (out_dtype == ULL_DT_BF16) ULL_LAUNCH_LUT(ULL_DT_BF16);
    else if (out_dtype == ULL_DT_F16) ULL_LAUNCH_LUT(ULL_DT_F16);
    else ULL_LAUNCH_LUT(ULL_DT_F32);
#undef ULL_LAUNCH_LUT
    return ull_check_launch();
}

// counts: int32 [n_masks, 6], zeroed by the caller.
extern "C" int ull_mask_iou_counts(const void* logits, const void* target, int64_t n_masks, int64_t hw, int ignore_index, void* counts,
                                   void* stream) {
    if (!logits || !target || !counts || n_masks <= 0 || hw <= 0) return ULL_ERR_ARG;
    long bx = (hw + 256 * 8 - 1) / (256 * 8);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(mask_iou_kernel, dim3((unsigned)bx, (unsigned)n_masks), dim3(256), 0, (hipStream_t)stream, (const float*)logits,
                       (const uint8_t*)target, hw, ignore_index, (int32_t*)counts);
    return ull_check_launch();
}
