// The fp32 build of the inference path: `--dtype fp32` of the reference's entry points (inference_ullava.py:25,164-168: fp32 | bf16 | fp16).
//
// Correctness path, no performance bar: every tensor is float32 and there are NO intermediate roundings to reproduce (the reference's fp32
// graph keeps fp32 between ops), so these are plain kernels -- the contractions run on gfx950's exact fp32 matrix instruction
// (v_mfma_f32_16x16x4_f32: products and sums are IEEE fp32, 1/16 of the bf16 rate), everything else is one thread per element or one
// wave-reduction per row.  Same C signatures, layouts and flags as the ull_*_bf16 entries of the same name (include/ullava_hip.h), so the
// host layer (u-llava_amd/ops.py) dispatches by tensor dtype alone.  Pure data movement (embedding splice, row gather, window partition)
// has no fp32 entry: the host hands the 16-bit kernels the same bytes as rows of twice as many 16-bit elements.
// Not covered (the host never routes fp32 tensors there): the fused / tiled fast paths (ull_gemm_qkv_rope, ull_patchify, ull_sam_window_attention,
// the fused mask-decoder kernels, the coarse layer-stack entries), rel_mode 2 of the attention, and the backward kernels.
#include "ull_common.h"

#define ULL_EPI_BIAS 1
#define ULL_EPI_ACT_MASK (3 << 1)
#define ULL_EPI_ACT_QUICK_GELU (1 << 1)
#define ULL_EPI_ACT_GELU (2 << 1)
#define ULL_EPI_ACT_RELU (3 << 1)
#define ULL_EPI_RESID 8
#define ULL_EPI_SWIGLU 16
#define ULL_EPI_W_TILED 64
#define ULL_EPI_X_TILED 128

namespace {

ULL_DEV float gelu_erf32(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }       // torch.nn.GELU() in fp32
ULL_DEV float sigmoid32(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline unsigned grid_for(long n, long cap = 65535) { const long b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b < cap ? b : cap)); }

// ---- C[M, N] = epilogue(X[M, K] W[N, K]^T): 64 x 64 tile per 256-thread block, K in steps of 16 through the LDS, one 16 x 64 strip per wave on
// v_mfma_f32_16x16x4_f32.  Any M / N / K / strides (scalar loads, zero fill at the edges).
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W, long ldw, float* __restrict__ C,
                                                       long ldc, const float* __restrict__ bias, const float* __restrict__ R, long ldr, int M, int N,
                                                       int K, int flags) {
    __shared__ float Xs[64][17], Ws[64][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.y * 64, n0 = (long)blockIdx.x * 64;
    const int lr = tid >> 2, lk = (tid & 3) * 4;            // this thread stages row lr, columns lk .. lk + 3 of both tiles
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + lk + j;
            Xs[lr][lk + j] = (m0 + lr < M && k < K) ? X[(m0 + lr) * ldx + k] : 0.f;
            Ws[lr][lk + j] = (n0 + lr < N && k < K) ? W[(n0 + lr) * ldw + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float a = Xs[wave * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Ws[j * 16 + (lane & 15)][kk * 4 + (lane >> 4)], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int act = flags & ULL_EPI_ACT_MASK;
    if (flags & ULL_EPI_SWIGLU) {
        // W rows are gate / up interleaved in groups of 16: strips j = 0, 2 are gate rows, j = 1, 3 the matching up rows; out has N / 2 columns
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const long oc = (n0 / 32 + pr) * 16 + (lane & 15);
            if (n0 + pr * 32 + 16 + (lane & 15) >= N) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = m0 + wave * 16 + 4 * (lane >> 4) + i;
                if (m >= M) continue;
                const float g = acc[2 * pr][i], u = acc[2 * pr + 1][i];
                float v = (g * sigmoid32(g)) * u;                       // F.silu(gate) * up
                if (R) v = R[m * ldr + oc] + v;
                C[m * ldc + oc] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long n = n0 + j * 16 + (lane & 15);
        if (n >= N) continue;
        const float bv = (flags & ULL_EPI_BIAS) ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + wave * 16 + 4 * (lane >> 4) + i;
            if (m >= M) continue;
            float v = acc[j][i] + bv;
            if (act == ULL_EPI_ACT_QUICK_GELU) v = v * sigmoid32(1.702f * v);
            else if (act == ULL_EPI_ACT_GELU) v = gelu_erf32(v);
            else if (act == ULL_EPI_ACT_RELU) v = fmaxf(v, 0.f);
            if (flags & ULL_EPI_RESID) v = R[m * ldr + n] + v;
            C[m * ldc + n] = v;
        }
    }
}

// ---- attention: one block per (16 queries, batch x head); keys streamed in tiles of 64 with a running (max, sum) -- softmax(S) V in fp32.
struct AttnF32 {
    const float *Q, *K, *V;
    float* O;
    const int32_t* key_mask;
    const float *rel_h, *rel_w;
    long q_bs, q_hs, q_ss, k_bs, k_hs, k_ss, v_bs, v_hs, v_ds, v_len, o_bs, o_hs, o_ss;
    int B, H, Sq, Sk, hd, causal, scale_mode, KH, KW;
    float scale, q_scale;
};

ULL_DEV long vt_slot(long key) {          // ull_transpose_v's key permutation inside a 32-key block
    const int w = (int)(key & 31);
    return (key & ~31L) + 8 * ((w >> 2) & 3) + 4 * (w >> 4) + (w & 3);
}

__global__ __launch_bounds__(256) void attn_f32_kernel(AttnF32 p) {
    constexpr int HDP = 128;
    __shared__ float Qs[16][HDP + 1], KVs[64][HDP + 1], Ss[16][65], alpha[16], rowm[16], rowl[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * 16;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int hd = p.hd, hd4 = (hd + 3) / 4, ndb = (hd + 15) / 16;
    const int koff = p.Sk - p.Sq;
    const float* qb = p.Q + (long)b * p.q_bs + (long)h * p.q_hs;
    const float* kb = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
    const float* vb = p.V + (long)b * p.v_bs + (long)h * p.v_hs;
    for (int i = tid; i < 16 * HDP; i += 256) {
        const int r = i / HDP, d = i % HDP;
        const int q = min(q0 + r, p.Sq - 1);
        Qs[r][d] = d < hd ? qb[(long)q * p.q_ss + d] * p.q_scale : 0.f;
    }
    if (tid < 16) { rowm[tid] = -INFINITY; rowl[tid] = 0.f; }
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    int kend = p.Sk;
    if (p.causal) kend = max(1, min(p.Sk, q0 + 16 + koff));
    __syncthreads();
    for (int k0 = 0; k0 < kend; k0 += 64) {
        for (int i = tid; i < 64 * HDP; i += 256) {                       // K tile
            const int r = i / HDP, d = i % HDP;
            KVs[r][d] = (k0 + r < p.Sk && d < hd) ? kb[(long)(k0 + r) * p.k_ss + d] : 0.f;
        }
        __syncthreads();
        {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
            for (int k4 = 0; k4 < hd4; ++k4)
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[lane & 15][k4 * 4 + (lane >> 4)], KVs[wave * 16 + (lane & 15)][k4 * 4 + (lane >> 4)], s, 0, 0, 0);
            const int kl = wave * 16 + (lane & 15), key = k0 + kl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * (lane >> 4) + i, q = min(q0 + r, p.Sq - 1);
                float v = s[i];
                if (p.scale_mode == 1) v *= p.scale;
                else if (p.scale_mode == 2) v /= p.scale;
                if (key >= p.Sk) v = -INFINITY;
                else {
                    if (p.rel_h) {
                        const long row = ((long)b * p.H + h) * p.Sq + q;
                        v = (v + p.rel_h[row * p.KH + key / p.KW]) + p.rel_w[row * p.KW + key % p.KW];
                    }
                    // hf eager mask: finfo.min is ADDED to the score -- in fp32 the sum is finfo.min itself
                    if ((p.causal && key > q + koff) || (p.key_mask && p.key_mask[(long)b * p.Sk + key] == 0)) v = -3.4028234663852886e38f;
                }
                Ss[r][kl] = v;
            }
        }
        __syncthreads();
        {
            const int r = tid >> 4, c = tid & 15;
            const float s0 = Ss[r][c], s1 = Ss[r][c + 16], s2 = Ss[r][c + 32], s3 = Ss[r][c + 48];
            const float mo = rowm[r];
            const float mn = fmaxf(mo, group_max(fmaxf(fmaxf(s0, s1), fmaxf(s2, s3)), 16));
            const float e0 = expf(s0 - mn), e1 = expf(s1 - mn), e2 = expf(s2 - mn), e3 = expf(s3 - mn);
            Ss[r][c] = e0; Ss[r][c + 16] = e1; Ss[r][c + 32] = e2; Ss[r][c + 48] = e3;
            const float sum = group_sum((e0 + e1) + (e2 + e3), 16);
            if (c == 0) {
                const float al = expf(mo - mn);
                alpha[r] = al;
                rowm[r] = mn;
                rowl[r] = rowl[r] * al + sum;
            }
        }
        for (int i = tid; i < 64 * HDP; i += 256) {                       // V tile over the K tile (every wave is past its reads of it)
            const int r = i / HDP, d = i % HDP;
            float v = 0.f;
            if (k0 + r < p.Sk && d < hd) v = p.v_len == 0 ? vb[(long)(k0 + r) * p.v_ds + d] : vb[(long)d * p.v_ds + vt_slot(k0 + r)];
            KVs[r][d] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int db = wave * 2 + j;
            if (db >= ndb) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] *= alpha[4 * (lane >> 4) + i];
            for (int k4 = 0; k4 < 16; ++k4)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ss[lane & 15][k4 * 4 + (lane >> 4)], KVs[k4 * 4 + (lane >> 4)][db * 16 + (lane & 15)], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    float* ob = p.O + (long)b * p.o_bs + (long)h * p.o_hs;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int d = (wave * 2 + j) * 16 + (lane & 15);
        if (d >= hd) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * (lane >> 4) + i;
            if (q0 + r < p.Sq) ob[(long)(q0 + r) * p.o_ss + d] = acc[j][i] / rowl[r];
        }
    }
}

// ---- row norms: one wave per row ------------------------------------------------------------------------------------------------------------
// MODE 0: LlamaRMSNorm  y = w * (x * rsqrt(mean(x^2) + eps));  MODE 1: nn.LayerNorm;  cls != null: CLIPVisionEmbeddings + pre_layrnorm
// (row (img, t): t == 0 ? class_embedding : patch[img * (tokens - 1) + t - 1], plus position_embedding[t]).
template <int MODE>
__global__ __launch_bounds__(256) void rownorm_f32_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ y, long ldy, long rows, int D, float eps, const float* __restrict__ cls,
                                                          const float* __restrict__ pos, int tokens) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    const float* pr = nullptr;
    if (cls) {
        const long img = row / tokens;
        const int t = (int)(row % tokens);
        xr = t == 0 ? cls : x + (img * (tokens - 1) + t - 1) * ldx;
        pr = pos + (long)t * D;
    }
    float s = 0.f;
    for (int i = lane; i < D; i += 64) {
        const float v = pr ? xr[i] + pr[i] : xr[i];
        s += MODE == 0 ? v * v : v;
    }
    s = wave_sum(s);
    float* yr = y + row * ldy;
    if (MODE == 0) {
        const float r = rsqrtf(s / (float)D + eps);
        for (int i = lane; i < D; i += 64) yr[i] = w[i] * (xr[i] * r);
    } else {
        const float mean = s / (float)D;
        float v2 = 0.f;
        for (int i = lane; i < D; i += 64) {
            const float d = (pr ? xr[i] + pr[i] : xr[i]) - mean;
            v2 += d * d;
        }
        const float rstd = rsqrtf(wave_sum(v2) / (float)D + eps);
        for (int i = lane; i < D; i += 64) yr[i] = ((pr ? xr[i] + pr[i] : xr[i]) - mean) * rstd * w[i] + b[i];
    }
}

// common.py:31-43 LayerNorm2d on channels-last rows: u = mean(x); s = mean((x - u)^2); y = w * ((x - u) / sqrt(s + eps)) + b  [; GELU]
__global__ __launch_bounds__(256) void layernorm2d_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                              float* __restrict__ y, long rows, int C, float eps, int gelu) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += xr[i];
    const float u = wave_sum(s) / (float)C;
    float v2 = 0.f;
    for (int i = lane; i < C; i += 64) v2 += (xr[i] - u) * (xr[i] - u);
    const float den = sqrtf(wave_sum(v2) / (float)C + eps);
    for (int i = lane; i < C; i += 64) {
        float t = w[i] * ((xr[i] - u) / den) + b[i];
        if (gelu) t = gelu_erf32(t);
        y[row * C + i] = t;
    }
}

// ---- RoPE (hf apply_rotary_pos_emb in fp32: q * cos + rotate_half(q) * sin, cos / sin = fp32 cos / sin of pos * inv_freq) --------------------
__global__ __launch_bounds__(256) void rope_inplace_f32_kernel(float* __restrict__ x, long row_stride, const int64_t* __restrict__ pos,
                                                               const float* __restrict__ inv_freq, int n_heads, int hd) {
    const long tok = blockIdx.x;
    const int half = hd >> 1;
    const float pf = (float)pos[tok];
    float* xr = x + tok * row_stride;
    for (int i = threadIdx.x; i < n_heads * half; i += 256) {
        const int hh = i / half, c = i % half;
        const float a = pf * inv_freq[c], cs = cosf(a), sn = sinf(a);
        float* p1 = xr + hh * hd + c;
        const float u = p1[0], v = p1[half];
        p1[0] = u * cs + (-v) * sn;
        p1[half] = v * cs + u * sn;
    }
}

__global__ __launch_bounds__(128) void rope_append_f32_kernel(float* __restrict__ qkv, long row_stride, const int64_t* __restrict__ pos,
                                                              const float* __restrict__ inv_freq, int S, int H, int hd, float* __restrict__ kc,
                                                              float* __restrict__ vtc, int smax, int past) {
    const long tok = blockIdx.x;
    const int h = blockIdx.y;
    const int b = (int)(tok / S), s = (int)(tok % S);
    const int half = hd >> 1;
    float* xr = qkv + tok * row_stride;
    const int slot_k = past + s;
    const float pf = (float)pos[tok];
    for (int i = threadIdx.x; i < 2 * half; i += 128) {
        const bool is_k = i >= half;
        const int c = i % half;
        const float a = pf * inv_freq[c], cs = cosf(a), sn = sinf(a);
        float* p1 = xr + ((is_k ? H : 0) + h) * hd + c;
        const float u = p1[0], v = p1[half];
        const float o1 = u * cs + (-v) * sn, o2 = v * cs + u * sn;
        if (!is_k) { p1[0] = o1; p1[half] = o2; }
        else {
            float* kp = kc + (((long)b * H + h) * smax + slot_k) * hd + c;
            kp[0] = o1;
            kp[half] = o2;
        }
    }
    const long slot_v = vt_slot(slot_k);
    const float* vr = xr + 2 * H * hd + h * hd;
    for (int i = threadIdx.x; i < hd; i += 128) vtc[(((long)b * H + h) * hd + i) * smax + slot_v] = vr[i];
}

// V [B, S, H, hd] -> Vt [B, H, hd, pitch] in ull_transpose_v's key-permuted layout, zeros for keys >= S
__global__ __launch_bounds__(256) void transpose_v_f32_kernel(const float* __restrict__ v, long v_bs, long v_ss, float* __restrict__ vt, int S, int H,
                                                              int hd, int pitch, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long slot = i % pitch;
        long t = i / pitch;
        const int d = (int)(t % hd);
        t /= hd;
        const int h = (int)(t % H);
        const long b = t / H;
        const int w = (int)(slot & 31);
        const long key = (slot & ~31L) + 16 * ((w >> 2) & 1) + 4 * (w >> 3) + (w & 3);        // inverse of vt_slot
        vt[i] = key < S ? v[b * v_bs + key * v_ss + (long)h * hd + d] : 0.f;
    }
}

// ---- gathers / elementwise --------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int C, int Hh, int Ww, int ps, int gh,
                                                         int gw, int Kp, long total) {
    const int K = C * ps * ps;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const long row = i / Kp;
        float v = 0.f;
        if (k < K) {
            const int kx = k % ps, ky = (k / ps) % ps, c = k / (ps * ps);
            const int px = (int)(row % gw), py = (int)((row / gw) % gh);
            const long b = row / ((long)gw * gh);
            v = img[((b * C + c) * Hh + (py * ps + ky)) * (long)Ww + px * ps + kx];
        }
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void im2col3x3_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int tap = (int)((i / C) % 9);
        const long pix = i / ((long)C * 9);
        const int xx = (int)(pix % W), y = (int)((pix / W) % H);
        const long b = pix / ((long)W * H);
        const int sy = y + tap / 3 - 1, sx = xx + tap % 3 - 1;
        out[i] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? x[((b * H + sy) * (long)W + sx) * C + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void video_pool_f32_kernel(const float* __restrict__ f, float* __restrict__ out, int T, int N, int D, int pitch, int off) {
    const int b = blockIdx.y, r = blockIdx.x;
    const float* fb = f + ((long)b * T * pitch + off) * D;
    float* o = out + ((long)b * (T + N) + r) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        if (r < T) {
            for (int n = 0; n < N; ++n) acc += fb[((long)r * pitch + n) * D + d];
            o[d] = acc / (float)N;
        } else {
            for (int t = 0; t < T; ++t) acc += fb[((long)t * pitch + (r - T)) * D + d];
            o[d] = acc / (float)T;
        }
    }
}

__global__ __launch_bounds__(256) void add_rows_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long rows,
                                                           int D, long b_rows) {
    const long total = rows * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) out[i] = a[i] + b[((i / D) % b_rows) * D + i % D];
}

__global__ __launch_bounds__(256) void window_unpartition_add_f32_kernel(const float* __restrict__ win, const float* __restrict__ shortcut,
                                                                         float* __restrict__ out, int H, int W, int C, int ws, int nWh, int nWw, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long row = i / C;
        const int xx = (int)(row % W), y = (int)((row / W) % H);
        const long b = row / ((long)W * H);
        const long wrow = ((b * nWh + y / ws) * nWw + xx / ws) * (long)(ws * ws) + (y % ws) * ws + xx % ws;
        out[i] = shortcut[i] + win[wrow * C + c];
    }
}

// image_encoder.py:321-392: rel_h[bh, s, kh] = sum_c q[c] * Rh[qy - kh + KH - 1][c], rel_w[bh, s, kw] = sum_c q[c] * Rw[qx - kw + KW - 1][c]
__global__ __launch_bounds__(128) void relpos_f32_kernel(const float* __restrict__ q, long q_bs, long q_hs, long q_ss, const float* __restrict__ rph,
                                                         const float* __restrict__ rpw, float* __restrict__ out_h, float* __restrict__ out_w, int nH, int KH,
                                                         int KW, int hd) {
    __shared__ float qs[256];
    const int S = KH * KW;
    const long blk = blockIdx.x;
    const int s = (int)(blk % S);
    const long bh = blk / S;
    const float* qp = q + (bh / nH) * q_bs + (bh % nH) * q_hs + (long)s * q_ss;
    for (int c = threadIdx.x; c < hd; c += 128) qs[c] = qp[c];
    __syncthreads();
    const int qy = s / KW, qx = s % KW;
    for (int t = threadIdx.x; t < KH + KW; t += 128) {
        const float* r = (t < KH) ? rph + (long)(qy - t + KH - 1) * hd : rpw + (long)(qx - (t - KH) + KW - 1) * hd;
        float acc = 0.f;
        for (int c = 0; c < hd; ++c) acc += qs[c] * r[c];
        if (t < KH) out_h[(bh * S + s) * KH + t] = acc;
        else out_w[(bh * S + s) * KW + (t - KH)] = acc;
    }
}

// F.interpolate(x as [1, C, L], size = M, mode = "linear") -> [M, C] (half-pixel centres, source index clamped at 0)
__global__ __launch_bounds__(256) void interp_rows_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int L, int M, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * C) return;
    const int m = (int)(i / C), c = (int)(i % C);
    float src = __fsub_rn(__fmul_rn((float)L / (float)M, (float)m + 0.5f), 0.5f);
    if (src < 0.f) src = 0.f;
    const int i0 = min((int)src, L - 1), i1 = min(i0 + 1, L - 1);
    const float l1 = __fsub_rn(src, (float)i0), l0 = __fsub_rn(1.0f, l1);
    y[i] = __fadd_rn(__fmul_rn(l0, x[(long)i0 * C + c]), __fmul_rn(l1, x[(long)i1 * C + c]));
}

// mask_decoder.py:150-158 on the blocked up-scaling layout (see ull_mask_matmul_bf16)
__global__ __launch_bounds__(256) void mask_matmul_f32_kernel(const float* __restrict__ hyper, const float* __restrict__ up, float* __restrict__ masks, int T,
                                                              int Cc, int G, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d2 = (int)(i & 3), d1 = (int)((i >> 2) & 3);
        const long cell = i >> 4;
        const int xx = (int)(cell % G), y = (int)((cell / G) % G);
        const long n = cell / ((long)G * G);
        const float* u = up + i * Cc;
        const int Y = 4 * y + 2 * (d1 >> 1) + (d2 >> 1), X = 4 * xx + 2 * (d1 & 1) + (d2 & 1);
        const int HW = 4 * G;
        for (int t = 0; t < T; ++t) {
            const float* hp = hyper + (n * T + t) * Cc;
            float acc = 0.f;
            for (int c = 0; c < Cc; ++c) acc += hp[c] * u[c];
            masks[((n * T + t) * HW + Y) * (long)HW + X] = acc;
        }
    }
}

__global__ __launch_bounds__(1024) void greedy_step_f32_kernel(const float* __restrict__ logits, long row_stride, int V, int32_t* __restrict__ unfinished,
                                                               const int64_t* __restrict__ eos, int n_eos, long pad, int has_pad, int64_t* __restrict__ seq,
                                                               long seq_ld, int pos, int32_t* __restrict__ alive) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int b = blockIdx.x;
    const float* row = logits + (long)b * row_stride;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += 1024) {
        const float v = row[c];
        if (v > best || (v == best && c < idx)) { best = v; idx = c; }
    }
    if (idx == 0x7fffffff) idx = threadIdx.x < V ? threadIdx.x : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        const int live = unfinished[b];
        long tok = idx;
        if (!live && has_pad) tok = pad;
        seq[(long)b * seq_ld + pos] = tok;
        int still = live;
        if (live)
            for (int e = 0; e < n_eos; ++e)
                if (eos[e] == tok) still = 0;
        unfinished[b] = still;
        if (still) atomicAdd(alive, 1);
    }
}

__global__ __launch_bounds__(256) void shifted_ce_f32_kernel(const float* __restrict__ logits, long ld, const int64_t* __restrict__ labels, int B, int S, int V,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * (S - 1)) return;
    const int b = (int)(row / (S - 1)), t = (int)(row % (S - 1));
    const int64_t lab = labels[(long)b * S + t + 1];
    if (lab < 0 || lab >= V) return;
    const float* lp = logits + ((long)b * S + t) * ld;
    float m = -INFINITY;
    for (int i = lane; i < V; i += 64) m = fmaxf(m, lp[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += expf(lp[i] - m);
    s = wave_sum(s);
    if (lane == 0) {
        atomicAdd(out, m + logf(s) - lp[lab]);
        atomicAdd(out + 1, 1.0f);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int ull_gemm_f32(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr,
                            int64_t M, int64_t N, int64_t K, int flags, void* ws, int64_t ws_bytes, void* stream) {
    (void)ws; (void)ws_bytes;
    if (!X || !W || !C || M <= 0 || N <= 0 || K <= 0) return ULL_ERR_ARG;
    if (((flags & ULL_EPI_BIAS) && !bias) || ((flags & ULL_EPI_RESID) && !R)) return ULL_ERR_ARG;
    if (flags & (ULL_EPI_W_TILED | ULL_EPI_X_TILED)) return ULL_ERR_SHAPE;
    if ((flags & ULL_EPI_SWIGLU) && (N % 32)) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), dim3(256), 0, ST, (const float*)X, (long)ldx,
                       (const float*)W, (long)ldw, (float*)C, (long)ldc, (const float*)bias, (flags & ULL_EPI_RESID) ? (const float*)R : nullptr, (long)ldr,
                       (int)M, (int)N, (int)K, flags);
    return ull_check_launch();
}

extern "C" int ull_attention_f32(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* K, int64_t k_bs, int64_t k_hs, int64_t k_ss,
                                 const void* Vt, int64_t vt_bs, int64_t vt_hs, int64_t vt_ds, int64_t vt_len, void* O, int64_t o_bs, int64_t o_hs,
                                 int64_t o_ss, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal, int scale_mode,
                                 float scale, float q_scale, const void* rel_h, const void* rel_w, int64_t rel_kh, int64_t rel_kw, int rel_mode,
                                 const void* zeros, void* stream) {
    (void)zeros;
    if (!Q || !K || !Vt || !O || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || hd <= 0) return ULL_ERR_ARG;
    if (hd > 128 || (rel_h && rel_mode != 1) || (rel_h && (!rel_w || rel_kh * rel_kw != Sk))) return ULL_ERR_SHAPE;
    AttnF32 p;
    p.Q = (const float*)Q; p.K = (const float*)K; p.V = (const float*)Vt; p.O = (float*)O;
    p.key_mask = (const int32_t*)key_mask;
    p.rel_h = (const float*)rel_h; p.rel_w = (const float*)rel_w;
    p.q_bs = q_bs; p.q_hs = q_hs; p.q_ss = q_ss; p.k_bs = k_bs; p.k_hs = k_hs; p.k_ss = k_ss;
    p.v_bs = vt_bs; p.v_hs = vt_hs; p.v_ds = vt_ds; p.v_len = vt_len; p.o_bs = o_bs; p.o_hs = o_hs; p.o_ss = o_ss;
    p.B = (int)B; p.H = (int)H; p.Sq = (int)Sq; p.Sk = (int)Sk; p.hd = (int)hd; p.causal = causal; p.scale_mode = scale_mode;
    p.KH = (int)rel_kh; p.KW = (int)(rel_kw > 0 ? rel_kw : 1);
    p.scale = scale; p.q_scale = q_scale;
    hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)((Sq + 15) / 16), (unsigned)(B * H)), dim3(256), 0, ST, p);
    return ull_check_launch();
}

extern "C" int ull_transpose_v_f32(const void* v, int64_t v_bs, int64_t v_ss, void* vt, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t pitch,
                                   void* stream) {
    if (!v || !vt || B <= 0 || S <= 0 || H <= 0 || hd <= 0) return ULL_ERR_ARG;
    if (pitch % 64 || pitch < S) return ULL_ERR_SHAPE;
    const long total = B * H * hd * pitch;
    hipLaunchKernelGGL(transpose_v_f32_kernel, dim3(grid_for(total)), dim3(256), 0, ST, (const float*)v, (long)v_bs, (long)v_ss, (float*)vt, (int)S, (int)H,
                       (int)hd, (int)pitch, total);
    return ull_check_launch();
}

extern "C" int ull_rmsnorm_f32(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t rows, int64_t D, float eps, void* stream) {
    if (!x || !w || !y || rows <= 0 || D <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL((rownorm_f32_kernel<0>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, (const float*)x, (long)ldx, (const float*)w, nullptr,
                       (float*)y, (long)ldy, (long)rows, (int)D, eps, nullptr, nullptr, 0);
    return ull_check_launch();
}

extern "C" int ull_layernorm_f32(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy, int64_t rows, int64_t D, float eps,
                                 void* stream) {
    if (!x || !w || !b || !y || rows <= 0 || D <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL((rownorm_f32_kernel<1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, (const float*)x, (long)ldx, (const float*)w,
                       (const float*)b, (float*)y, (long)ldy, (long)rows, (int)D, eps, nullptr, nullptr, 0);
    return ull_check_launch();
}

extern "C" int ull_clip_embed_ln_f32(const void* patch, int64_t ldp, const void* cls, const void* pos, const void* w, const void* b, void* y, int64_t ldy,
                                     int64_t n_img, int64_t tokens, int64_t D, float eps, void* stream) {
    if (!patch || !cls || !pos || !w || !b || !y || tokens < 2 || n_img <= 0) return ULL_ERR_ARG;
    const long rows = n_img * tokens;
    hipLaunchKernelGGL((rownorm_f32_kernel<1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, (const float*)patch, (long)ldp, (const float*)w,
                       (const float*)b, (float*)y, (long)ldy, rows, (int)D, eps, (const float*)cls, (const float*)pos, (int)tokens);
    return ull_check_launch();
}

extern "C" int ull_layernorm2d_cl_f32(const void* x, const void* w, const void* b, void* y, int64_t rows, int64_t C, float eps, int gelu, void* stream) {
    if (!x || !w || !b || !y || rows <= 0 || C <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(layernorm2d_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, (const float*)x, (const float*)w, (const float*)b,
                       (float*)y, (long)rows, (int)C, eps, gelu);
    return ull_check_launch();
}

extern "C" int ull_rope_inplace_f32(void* x, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens, int64_t n_heads, int64_t hd,
                                    void* stream) {
    if (!x || !positions || !inv_freq || tokens <= 0 || n_heads <= 0) return ULL_ERR_ARG;
    if (hd & 1) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(rope_inplace_f32_kernel, dim3((unsigned)tokens), dim3(256), 0, ST, (float*)x, (long)row_stride, (const int64_t*)positions,
                       (const float*)inv_freq, (int)n_heads, (int)hd);
    return ull_check_launch();
}

extern "C" int ull_rope_append_f32(void* qkv, int64_t row_stride, const void* positions, const void* inv_freq, int64_t B, int64_t S, int64_t H, int64_t hd,
                                   void* k_cache, void* vt_cache, int64_t smax, int64_t past, void* stream) {
    if (!qkv || !positions || !inv_freq || !k_cache || !vt_cache || B <= 0 || S <= 0 || H <= 0) return ULL_ERR_ARG;
    if ((hd & 1) || past + S > smax) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(rope_append_f32_kernel, dim3((unsigned)(B * S), (unsigned)H), dim3(128), 0, ST, (float*)qkv, (long)row_stride,
                       (const int64_t*)positions, (const float*)inv_freq, (int)S, (int)H, (int)hd, (float*)k_cache, (float*)vt_cache, (int)smax, (int)past);
    return ull_check_launch();
}

extern "C" int ull_im2col_f32(const void* img, void* out, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, int64_t Kp, void* stream) {
    if (!img || !out || n_img <= 0 || ps <= 0) return ULL_ERR_ARG;
    if (H % ps || W % ps || Kp < C * ps * ps) return ULL_ERR_SHAPE;
    const int gh = (int)(H / ps), gw = (int)(W / ps);
    const long total = n_img * gh * gw * Kp;
    hipLaunchKernelGGL(im2col_f32_kernel, dim3(grid_for(total)), dim3(256), 0, ST, (const float*)img, (float*)out, (int)C, (int)H, (int)W, (int)ps, gh, gw,
                       (int)Kp, total);
    return ull_check_launch();
}

extern "C" int ull_im2col3x3_f32(const void* x, void* out, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return ULL_ERR_ARG;
    const long total = B * H * W * 9 * C;
    hipLaunchKernelGGL(im2col3x3_f32_kernel, dim3(grid_for(total)), dim3(256), 0, ST, (const float*)x, (float*)out, (int)H, (int)W, (int)C, total);
    return ull_check_launch();
}

extern "C" int ull_video_pool_f32(const void* f, void* out, int64_t B, int64_t T, int64_t N, int64_t D, int64_t tok_pitch, int64_t tok_off, void* stream) {
    if (!f || !out || B <= 0 || T <= 0 || N <= 0 || D <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(video_pool_f32_kernel, dim3((unsigned)(T + N), (unsigned)B), dim3(256), 0, ST, (const float*)f, (float*)out, (int)T, (int)N, (int)D,
                       (int)tok_pitch, (int)tok_off);
    return ull_check_launch();
}

extern "C" int ull_add_rows_f32(const void* a, const void* b, void* out, int64_t rows, int64_t D, int64_t b_rows, void* stream) {
    if (!a || !b || !out || rows <= 0 || D <= 0 || b_rows <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(add_rows_f32_kernel, dim3(grid_for(rows * D)), dim3(256), 0, ST, (const float*)a, (const float*)b, (float*)out, (long)rows, (int)D,
                       (long)b_rows);
    return ull_check_launch();
}

extern "C" int ull_window_unpartition_add_f32(const void* win, const void* shortcut, void* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ws,
                                              void* stream) {
    if (!win || !shortcut || !out || B <= 0 || ws <= 0) return ULL_ERR_ARG;
    const int nWh = (int)((H + ws - 1) / ws), nWw = (int)((W + ws - 1) / ws);
    const long total = B * H * W * C;
    hipLaunchKernelGGL(window_unpartition_add_f32_kernel, dim3(grid_for(total)), dim3(256), 0, ST, (const float*)win, (const float*)shortcut, (float*)out,
                       (int)H, (int)W, (int)C, (int)ws, nWh, nWw, total);
    return ull_check_launch();
}

extern "C" int ull_sam_relpos_f32(const void* q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* rel_pos_h, const void* rel_pos_w, void* out_h,
                                  void* out_w, int64_t NB, int64_t nH, int64_t KH, int64_t KW, int64_t hd, void* stream) {
    if (!q || !rel_pos_h || !rel_pos_w || !out_h || !out_w || NB <= 0) return ULL_ERR_ARG;
    if (hd > 256) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(relpos_f32_kernel, dim3((unsigned)(NB * nH * KH * KW)), dim3(128), 0, ST, (const float*)q, (long)q_bs, (long)q_hs, (long)q_ss,
                       (const float*)rel_pos_h, (const float*)rel_pos_w, (float*)out_h, (float*)out_w, (int)nH, (int)KH, (int)KW, (int)hd);
    return ull_check_launch();
}

extern "C" int ull_interp_rows_linear_f32(const void* x, void* y, int64_t L, int64_t M, int64_t C, void* stream) {
    if (!x || !y || L <= 0 || M <= 0 || C <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(interp_rows_f32_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, ST, (const float*)x, (float*)y, (int)L, (int)M, (int)C);
    return ull_check_launch();
}

extern "C" int ull_mask_matmul_f32(const void* hyper, const void* up, void* masks, int64_t n, int64_t T, int64_t C, int64_t G, void* stream) {
    if (!hyper || !up || !masks || n <= 0 || T <= 0 || C <= 0 || G <= 0) return ULL_ERR_ARG;
    const long total = n * G * G * 16;
    hipLaunchKernelGGL(mask_matmul_f32_kernel, dim3(grid_for(total)), dim3(256), 0, ST, (const float*)hyper, (const float*)up, (float*)masks, (int)T, (int)C,
                       (int)G, total);
    return ull_check_launch();
}

extern "C" int ull_greedy_step_f32(const void* logits, int64_t row_stride, int64_t B, int64_t V, void* unfinished, const void* eos, int64_t n_eos,
                                   int64_t pad, int has_pad, void* seq, int64_t seq_ld, int64_t pos, void* alive, void* stream) {
    if (!logits || !unfinished || !seq || !alive || B <= 0 || V <= 0 || pos < 0 || pos >= seq_ld || (n_eos > 0 && !eos)) return ULL_ERR_ARG;
    hipLaunchKernelGGL(greedy_step_f32_kernel, dim3((unsigned)B), dim3(1024), 0, ST, (const float*)logits, (long)row_stride, (int)V, (int32_t*)unfinished,
                       (const int64_t*)eos, (int)n_eos, (long)pad, has_pad, (int64_t*)seq, (long)seq_ld, (int)pos, (int32_t*)alive);
    return ull_check_launch();
}

extern "C" int ull_shifted_cross_entropy_f32(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V, void* out, void* stream) {
    if (!logits || !labels || !out || B <= 0 || S <= 1 || V <= 0) return ULL_ERR_ARG;
    const long rows = B * (S - 1);
    hipLaunchKernelGGL(shifted_ce_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST, (const float*)logits, (long)ld, (const int64_t*)labels,
                       (int)B, (int)S, (int)V, (float*)out);
    return ull_check_launch();
}
