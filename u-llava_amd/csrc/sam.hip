// SAM-specific data-movement and small-math kernels (token-major / channels-last layouts throughout):
//   window partition / unpartition(+residual)      segment_anything/modeling/image_encoder.py:263-318, 177-193
//   decomposed relative-position bias tables        image_encoder.py:321-392 (get_rel_pos + the two einsums)
//   LayerNorm2d (bf16 op chain) (+GELU)             common.py:31-43, mask_decoder.py:53-64
//   3x3 im2col for the neck conv                    image_encoder.py:92-108
//   hyper-network mask product                      mask_decoder.py:150-158
//   bilinear resize (postprocess_masks)             sam.py:137-172
// All HBM-bound or tiny; 16-byte accesses where rows are contiguous.
#include "ull_common.h"

namespace {

// x [B, H, W, C] -> windows [B * nWh * nWw, ws * ws, C]; tokens beyond H/W are zero rows (F.pad in the reference)
__global__ __launch_bounds__(256) void window_partition_kernel(const elem_t* __restrict__ x, elem_t* __restrict__ out, int H, int W, int C,
                                                               int ws, int nWh, int nWw, long total_chunks) {
    const int cpr = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_chunks; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cpr);
        long row = i / cpr;                              // output token index
        const int t = (int)(row % (ws * ws));
        long win = row / (ws * ws);
        const int ww = (int)(win % nWw), wh = (int)((win / nWw) % nWh);
        const long b = win / ((long)nWw * nWh);
        const int y = wh * ws + t / ws, xx = ww * ws + t % ws;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (y < H && xx < W) v = *(const uint4*)(x + ((b * H + y) * (long)W + xx) * C + c * 8);
        *(uint4*)(out + row * C + c * 8) = v;
    }
}

// out[b, y, x, :] = bf16(shortcut[b, y, x, :] + win[window(b, y, x), :])
__global__ __launch_bounds__(256) void window_unpartition_add_kernel(const elem_t* __restrict__ win, const elem_t* __restrict__ shortcut,
                                                                     elem_t* __restrict__ out, int H, int W, int C, int ws, int nWh, int nWw,
                                                                     long total_chunks) {
    const int cpr = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_chunks; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cpr);
        const long row = i / cpr;                        // b*H*W + y*W + x
        const int xx = (int)(row % W), y = (int)((row / W) % H);
        const long b = row / ((long)W * H);
        const long wrow = ((b * nWh + y / ws) * nWw + xx / ws) * (long)(ws * ws) + (y % ws) * ws + xx % ws;
        float a[8], s[8];
        unpack8(*(const uint4*)(win + wrow * C + c * 8), a);
        unpack8(*(const uint4*)(shortcut + row * C + c * 8), s);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += a[j];
        *(uint4*)(out + row * C + c * 8) = pack8(s);
    }
}

// One block per (head-instance bh, query s = (qy, qx)): rel_h[bh, s, kh] = bf16(sum_c q[c] * Rh[qy - kh + KH - 1][c]),
// rel_w[bh, s, kw] = bf16(sum_c q[c] * Rw[qx - kw + KW - 1][c])   (q and k grids have the same size: no interpolation)
__global__ __launch_bounds__(128) void relpos_kernel(const elem_t* __restrict__ q, long q_bs, long q_hs, long q_ss,
                                                     const elem_t* __restrict__ rph, const elem_t* __restrict__ rpw,
                                                     elem_t* __restrict__ out_h, elem_t* __restrict__ out_w, int nH, int KH, int KW, int hd) {
    __shared__ float qs[256];
    const int S = KH * KW;
    const long blk = blockIdx.x;
    const int s = (int)(blk % S);
    const long bh = blk / S;
    const long b = bh / nH;
    const int h = (int)(bh % nH);
    const elem_t* qp = q + b * q_bs + (long)h * q_hs + (long)s * q_ss;
    for (int c = threadIdx.x; c < hd; c += 128) qs[c] = e2f(qp[c]);
    __syncthreads();
    const int qy = s / KW, qx = s % KW;
    for (int t = threadIdx.x; t < KH + KW; t += 128) {
        const elem_t* r = (t < KH) ? rph + (long)(qy - t + KH - 1) * hd : rpw + (long)(qx - (t - KH) + KW - 1) * hd;
        float acc = 0.f;
        for (int c = 0; c < hd; ++c) acc += qs[c] * e2f(r[c]);
        if (t < KH) out_h[(bh * S + s) * KH + t] = f2e(acc);
        else out_w[(bh * S + s) * KW + (t - KH)] = f2e(acc);
    }
}

// LayerNorm2d on channels-last rows, reproducing the reference's bf16 tensor-op chain:
//   u = bf16(mean(x)); d = bf16(x - u); s = bf16(mean(bf16(d*d))); y = bf16(d / bf16(sqrt(bf16(s + eps))));
//   out = bf16(bf16(w * y) + b)  [; out = bf16(gelu(out))]
// One row per `lpr` lanes (C/8 chunks, C <= 512).
__global__ __launch_bounds__(256) void layernorm2d_cl_kernel(const elem_t* __restrict__ x, const elem_t* __restrict__ w,
                                                             const elem_t* __restrict__ b, elem_t* __restrict__ y, long rows, int C,
                                                             float eps, int lpr, int gelu) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lpr - 1);
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / lpr) + lane / lpr;
    const bool ok = row < rows && sub * 8 < C;
    float v[8];
    if (ok) unpack8(*(const uint4*)(x + row * C + sub * 8), v);
    else
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s1 += v[j];
    const float u = rnd(group_sum(s1, lpr) / (float)C);
    float d[8], s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d[j] = ok ? rnd(v[j] - u) : 0.f;
        s2 += rnd(d[j] * d[j]);
    }
    const float var = rnd(group_sum(s2, lpr) / (float)C);
    const float den = rnd(sqrtf(rnd(var + eps)));
    if (!ok) return;
    float wv[8], bv[8], o[8];
    unpack8(*(const uint4*)(w + sub * 8), wv);
    unpack8(*(const uint4*)(b + sub * 8), bv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = rnd(rnd(wv[j] * rnd(d[j] / den)) + bv[j]);
        if (gelu) t = act_gelu_erf(t);
        o[j] = t;
    }
    *(uint4*)(y + row * C + sub * 8) = pack8(o);
}

#ifdef ULL_ELEM_F16
// image_encoder.py:117-124: an fp16 model runs its neck under autocast(float32) -- LayerNorm2d (common.py:31-43) then sees the fp32
// output of a convolution and promotes: everything below is fp32 arithmetic in the reference's order (u = mean(x); d = x - u;
// s = mean(d*d); y = w * (d / sqrt(s + eps)) + b with the fp16 w / b promoted), separate multiply and add (no contraction).
//   x[row] = xa[row] (+ xb[row] * xb_scale when xb != null): fp32 channels-last rows (xb: the second term of the split GEMM below).
//   y_lo == null: y_hi[row] = fp16(y)                       -- the `x.to(float16)` that ends the neck.
//   y_lo != null: y_hi = fp16(y), y_lo = fp16((y - y_hi) * 2^11) -- a two-term fp16 split of the fp32 activations (22 significand
//                 bits), so that the 3x3 convolution that follows can run as fp16 MFMA GEMMs on both terms with fp32 accumulation and
//                 still see its fp32 input: conv(y) = conv(y_hi) + 2^-11 conv(y_lo); the weights are fp16 values, exact in either.
// One row per `lpr` lanes, 8 channels per lane (C <= 512, C % 8 == 0).
__global__ __launch_bounds__(256) void neck_ln2d_f32_kernel(const float* __restrict__ xa, const float* __restrict__ xb, float xb_scale,
                                                            const elem_t* __restrict__ w, const elem_t* __restrict__ b,
                                                            elem_t* __restrict__ y_hi, elem_t* __restrict__ y_lo, long rows, int C, float eps,
                                                            int lpr) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lpr - 1);
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / lpr) + lane / lpr;
    const bool ok = row < rows && sub * 8 < C;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (ok) {
        const float4 a0 = *(const float4*)(xa + row * C + sub * 8), a1 = *(const float4*)(xa + row * C + sub * 8 + 4);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
        if (xb) {
            const float4 b0 = *(const float4*)(xb + row * C + sub * 8), b1 = *(const float4*)(xb + row * C + sub * 8 + 4);
            const float t[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __fadd_rn(v[j], __fmul_rn(t[j], xb_scale));   // 2^-11 scaling is exact
        }
    }
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s1 += v[j];
    const float u = group_sum(s1, lpr) / (float)C;
    float d[8], s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        d[j] = ok ? v[j] - u : 0.f;
        s2 = __fadd_rn(s2, __fmul_rn(d[j], d[j]));
    }
    const float var = group_sum(s2, lpr) / (float)C;
    const float den = __fsqrt_rn(var + eps);
    if (!ok) return;
    float wv[8], bv[8], hi[8], lo[8];
    unpack8(*(const uint4*)(w + sub * 8), wv);
    unpack8(*(const uint4*)(b + sub * 8), bv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float y = __fadd_rn(__fmul_rn(wv[j], __fdiv_rn(d[j], den)), bv[j]);
        hi[j] = rnd(y);
        lo[j] = (y - hi[j]) * 2048.0f;
    }
    *(uint4*)(y_hi + row * C + sub * 8) = pack8(hi);
    if (y_lo) *(uint4*)(y_lo + row * C + sub * 8) = pack8(lo);
}
#endif

// x [B, H, W, C] channels-last -> cols [B*H*W, 9*C], column block (ky*3+kx) holds the C channels of the neighbour
// (y+ky-1, x+kx-1), zeros outside the image (conv padding=1).  Weight is packed host-side in the same (ky,kx,ci) order.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const elem_t* __restrict__ x, elem_t* __restrict__ out, int H, int W, int C,
                                                        long total_chunks) {
    const int cpr = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_chunks; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cpr);
        const int tap = (int)((i / cpr) % 9);
        const long pix = i / ((long)cpr * 9);
        const int xx = (int)(pix % W), y = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        const int sy = y + tap / 3 - 1, sx = xx + tap % 3 - 1;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = *(const uint4*)(x + ((b * H + sy) * (long)W + sx) * C + c * 8);
        *(uint4*)(out + (pix * 9 + tap) * C + c * 8) = v;
    }
}

// masks[n, t, Y, X] = bf16( sum_c hyper[n, t, c] * up[n, pixel(Y, X), c] ), up given in the blocked layout produced by
// the two transposed convolutions run as GEMMs:  up[n][y][x][d1 = dy*2+dx][d2 = dy2*2+dx2][c], Y = 4y + 2dy + dy2, X = 4x + 2dx + dx2.
__global__ __launch_bounds__(256) void mask_matmul_kernel(const elem_t* __restrict__ hyper, const elem_t* __restrict__ up,
                                                          elem_t* __restrict__ masks, int T, int Cc, int G, long total) {
    // total = n * G*G*16 output pixels; one thread per output pixel, all T masks
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d2 = (int)(i & 3), d1 = (int)((i >> 2) & 3);
        const long cell = i >> 4;                        // n*G*G + y*G + x
        const int xx = (int)(cell % G), y = (int)((cell / G) % G);
        const long n = cell / ((long)G * G);
        const elem_t* u = up + i * Cc;
        float uv[32];
        for (int c0 = 0; c0 < Cc; c0 += 8) unpack8(*(const uint4*)(u + c0), uv + c0);
        const int Y = 4 * y + 2 * (d1 >> 1) + (d2 >> 1), X = 4 * xx + 2 * (d1 & 1) + (d2 & 1);
        const int HW = 4 * G;
        for (int t = 0; t < T; ++t) {
            const elem_t* hp = hyper + (n * T + t) * Cc;
            float acc = 0.f;
            for (int c = 0; c < Cc; ++c) acc += e2f(hp[c]) * uv[c];
            masks[((n * T + t) * HW + Y) * (long)HW + X] = f2e(acc);
        }
    }
}

// F.interpolate(mode="bilinear", align_corners=False) in fp32: src = max((dst + 0.5) * (in / out) - 0.5, 0).
// Input image i: elements of dtype DT (ULL_DT_*) at `in + i * in_img_stride`, rows `in_row_stride` apart, logical size in_h x in_w (a crop).
// (dtype-coded, so it exists once: only the bf16 build of this file defines it)
#ifndef ULL_ELEM_F16
template <int DT>
__global__ __launch_bounds__(256) void bilinear_kernel(const void* __restrict__ in, long in_img_stride, long in_row_stride, int in_h, int in_w,
                                                       float* __restrict__ out, int out_h, int out_w, long total) {
    const float sh = (float)in_h / (float)out_h, sw = (float)in_w / (float)out_w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % out_w), oy = (int)((i / out_w) % out_h);
        const long n = i / ((long)out_w * out_h);
        float fy = sh * ((float)oy + 0.5f) - 0.5f, fx = sw * ((float)ox + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < in_h - 1 ? 1 : 0), x1 = x0 + (x0 < in_w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const long base = n * in_img_stride;
        auto ld = [&](int yy, int xx) -> float { return load_dt<DT>(in, base + (long)yy * in_row_stride + xx); };
        const float t0 = __fadd_rn(__fmul_rn(hx, ld(y0, x0)), __fmul_rn(lx, ld(y0, x1)));
        const float t1 = __fadd_rn(__fmul_rn(hx, ld(y1, x0)), __fmul_rn(lx, ld(y1, x1)));
        out[i] = __fadd_rn(__fmul_rn(hy, t0), __fmul_rn(ly, t1));
    }
}
#endif

inline unsigned nblocks(long total, long cap = 16384) {
    long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int ULL_FN(ull_window_partition_)(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ws, void* stream) {
    if (!x || !out || B <= 0 || ws <= 0) return ULL_ERR_ARG;
    if (C & 7) return ULL_ERR_SHAPE;
    const int nWh = (int)((H + ws - 1) / ws), nWw = (int)((W + ws - 1) / ws);
    const long total = B * nWh * nWw * ws * ws * (C >> 3);
    hipLaunchKernelGGL(window_partition_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (elem_t*)out, (int)H,
                       (int)W, (int)C, (int)ws, nWh, nWw, total);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_window_unpartition_add_)(const void* win, const void* shortcut, void* out, int64_t B, int64_t H, int64_t W, int64_t C,
                                               int64_t ws, void* stream) {
    if (!win || !shortcut || !out || B <= 0 || ws <= 0) return ULL_ERR_ARG;
    if (C & 7) return ULL_ERR_SHAPE;
    const int nWh = (int)((H + ws - 1) / ws), nWw = (int)((W + ws - 1) / ws);
    const long total = B * H * W * (C >> 3);
    hipLaunchKernelGGL(window_unpartition_add_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)win,
                       (const elem_t*)shortcut, (elem_t*)out, (int)H, (int)W, (int)C, (int)ws, nWh, nWw, total);
    return ull_check_launch();
}

// image_encoder.py:336-343: F.interpolate(rel_pos[L, C] as [1, C, L], size = M, mode = "linear") -> [M, C], the resize get_rel_pos applies to
// a table whose length is not 2 * size - 1 (a checkpoint trained at another input size).  Rounding as ATen's CPU kernel, which the
// oracle runs: source index in fp32 (half-pixel centres, clamped at 0), the weight lambda rounded to the element type, 1 - lambda
// rounded again, products and sum in fp32, one rounding of the result.
namespace {
__global__ __launch_bounds__(256) void interp_rows_linear_kernel(const elem_t* __restrict__ x, elem_t* __restrict__ y, int L, int M, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * C) return;
    const int m = (int)(i / C), c = (int)(i % C);
    const float scale = (float)L / (float)M;
    float src = __fsub_rn(__fmul_rn(scale, (float)m + 0.5f), 0.5f);
    if (src < 0.f) src = 0.f;
    const int i0 = min((int)src, L - 1), i1 = min(i0 + 1, L - 1);
    const float l1 = rnd(__fsub_rn(src, (float)i0));
    const float l0 = rnd(__fsub_rn(1.0f, l1));
    y[i] = f2e(__fadd_rn(__fmul_rn(l0, e2f(x[(long)i0 * C + c])), __fmul_rn(l1, e2f(x[(long)i1 * C + c]))));
}
}  // namespace

extern "C" int ULL_FN(ull_interp_rows_linear_)(const void* x, void* y, int64_t L, int64_t M, int64_t C, void* stream) {
    if (!x || !y || L <= 0 || M <= 0 || C <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(interp_rows_linear_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x,
                       (elem_t*)y, (int)L, (int)M, (int)C);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_sam_relpos_)(const void* q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* rel_pos_h, const void* rel_pos_w,
                                   void* out_h, void* out_w, int64_t NB, int64_t nH, int64_t KH, int64_t KW, int64_t hd, void* stream) {
    if (!q || !rel_pos_h || !rel_pos_w || !out_h || !out_w || NB <= 0) return ULL_ERR_ARG;
    if (hd > 256 || KH + KW > 4096) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(relpos_kernel, dim3((unsigned)(NB * nH * KH * KW)), dim3(128), 0, (hipStream_t)stream, (const elem_t*)q, q_bs, q_hs, q_ss,
                       (const elem_t*)rel_pos_h, (const elem_t*)rel_pos_w, (elem_t*)out_h, (elem_t*)out_w, (int)nH, (int)KH, (int)KW, (int)hd);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_layernorm2d_cl_)(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t C, float eps, int gelu,
                                       void* stream) {
    if (!x || !w || !b || !y || rows <= 0) return ULL_ERR_ARG;
    if ((C & 7) || C > 512) return ULL_ERR_SHAPE;
    int lpr = 1;
    while (lpr < (C >> 3)) lpr <<= 1;
    const long rows_per_block = 4 * (64 / lpr);
    hipLaunchKernelGGL(layernorm2d_cl_kernel, dim3((unsigned)((rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)x, (const elem_t*)w, (const elem_t*)b, (elem_t*)y, rows, (int)C, eps, lpr, gelu);
    return ull_check_launch();
}

#ifdef ULL_ELEM_F16
extern "C" int ull_neck_layernorm2d_f32in_f16(const void* xa, const void* xb, float xb_scale, const void* w, const void* b, void* y_hi, void* y_lo,
                                              int64_t rows, int64_t C, float eps, void* stream) {
    if (!xa || !w || !b || !y_hi || rows <= 0) return ULL_ERR_ARG;
    if ((C & 7) || C > 512) return ULL_ERR_SHAPE;
    int lpr = 1;
    while (lpr < (C >> 3)) lpr <<= 1;
    const long rows_per_block = 4 * (64 / lpr);
    hipLaunchKernelGGL(neck_ln2d_f32_kernel, dim3((unsigned)((rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)xa, (const float*)xb, xb_scale, (const elem_t*)w, (const elem_t*)b, (elem_t*)y_hi, (elem_t*)y_lo, rows, (int)C,
                       eps, lpr);
    return ull_check_launch();
}
#endif

extern "C" int ULL_FN(ull_im2col3x3_)(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    if (!x || !out || B <= 0) return ULL_ERR_ARG;
    if (C & 7) return ULL_ERR_SHAPE;
    const long total = B * H * W * 9 * (C >> 3);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (elem_t*)out, (int)H, (int)W,
                       (int)C, total);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_mask_matmul_)(const void* hyper, const void* up, void* masks, int64_t n, int64_t T, int64_t C, int64_t G, void* stream) {
    if (!hyper || !up || !masks || n <= 0) return ULL_ERR_ARG;
    if (C != 32 && C != 16 && C != 8) return ULL_ERR_SHAPE;
    const long total = n * G * G * 16;
    hipLaunchKernelGGL(mask_matmul_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)hyper, (const elem_t*)up,
                       (elem_t*)masks, (int)T, (int)C, (int)G, total);
    return ull_check_launch();
}

#ifndef ULL_ELEM_F16
extern "C" int ull_bilinear_f32(const void* in, int in_dtype, int64_t in_img_stride, int64_t in_row_stride, int64_t in_h, int64_t in_w,
                                void* out, int64_t n, int64_t out_h, int64_t out_w, void* stream) {
    if (!in || !out || n <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || in_dtype < 0 || in_dtype > 2) return ULL_ERR_ARG;
    const long total = n * out_h * out_w;
#define ULL_LAUNCH_BIL(DT)                                                                                                             \
    hipLaunchKernelGGL(bilinear_kernel<DT>, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, in, in_img_stride, in_row_stride, \
                       (int)in_h, (int)in_w, (float*)out, (int)out_h, (int)out_w, total)
    if (in_dtype == ULL_DT_BF16) ULL_LAUNCH_BIL(ULL_DT_BF16);
    else if (in_dtype == ULL_DT_F16) ULL_LAUNCH_BIL(ULL_DT_F16);
    else ULL_LAUNCH_BIL(ULL_DT_F32);
#undef ULL_LAUNCH_BIL
    return ull_check_launch();
}
#endif
