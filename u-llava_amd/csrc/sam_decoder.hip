// Fused kernels of SAM's two-way mask decoder (north_star: "SAM's MaskDecoder cross-attention fused into one LDS-resident kernel").
//
// reference: models/segment_anything/modeling/transformer.py:62-106 (TwoWayTransformer), :151-182 (TwoWayAttentionBlock),
// :220-242 (Attention), mask_decoder.py:137-164 (hyper-network MLPs, IoU head).  One decode used to be ~70 launches of generic
// kernels (3 Linears + add + transpose + attention + Linear + LayerNorm per attention); here every attention of the block is one or
// two launches that keep the projected tiles in LDS:
//   sam_token_self_attn_ln : q/k/v projections, 8-head attention over the T tokens, out-projection, (+residual), LayerNorm
//   sam_t2i_kv_scores      : k = (keys + pe) Wk^T, v = keys Wv^T on the MFMA for a 64-key tile kept in LDS, q projection, scores
//   sam_t2i_softmax_out_ln : exact fp32 softmax over all 4096 keys, P V, out-projection, residual, LayerNorm
//   sam_token_mlp_ln       : 256 -> 2048 -> 256 MLP (ReLU), residual, LayerNorm
//   sam_i2t_fused          : q = (keys + pe) Wq^T on the MFMA, k / v of the T tokens, softmax over T, P V, out-projection on the MFMA,
//                            residual, LayerNorm -- the image->token attention in ONE launch, nothing but the new keys written
//   sam_small_mlps         : the four hyper-network MLPs and the IoU head (3-layer MLPs on single token rows)
// Every 16-bit rounding point of the reference's graph is kept (rnd()): Linear output (bias fused in fp32, or added after the rounding
// where at::linear sees a non-contiguous input -- `late_bias`), tensor adds, scores / sqrt(hd), softmax output, P V, residual adds,
// LayerNorm output.  Fixed dims: embedding 256, internal dim 128 (cross) / 256 (self), 8 heads, T <= 8 tokens per prompt.
#include "ull_common.h"

namespace {

constexpr int D = 256;          // transformer width
constexpr int DI = 128;         // cross-attention internal width (downsample rate 2)
constexpr int NH = 8;           // heads
constexpr int TMAX = 8;         // tokens per prompt (iou + 4 mask tokens + 1 text prompt = 6)
constexpr int KT = 64;          // image rows per block in the cross-attention kernels
constexpr int PX = D + 8;       // LDS pitch (elements) of a [rows][256] tile: 16-byte shift per row -> conflict-free ds_read_b128
constexpr int PI = DI + 8;      // LDS pitch of a [rows][128] tile

struct LinW { const elem_t* w; const elem_t* b; };          // nn.Linear: w [out, in] row-major, b [out]
struct LnW { const elem_t* w; const elem_t* b; };

// fp32 dot of an LDS float vector with a 16-bit weight row (K % 8 == 0), 16-byte weight loads
ULL_DEV float dot_row(const float* __restrict__ x, const elem_t* __restrict__ w, int K) {
    float acc = 0.f;
#pragma unroll 4
    for (int c = 0; c < K; c += 8) {
        float wv[8];
        unpack8(*(const uint4*)(w + c), wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += x[c + j] * wv[j];
    }
    return acc;
}

// LayerNorm of T rows of 256 floats in LDS (already rounded values), one wave per row (rows t = wave, wave + nwaves, ...), output 16-bit
ULL_DEV void layernorm_rows(const float* __restrict__ xs, int T, LnW ln, float eps, elem_t* __restrict__ out, int wave, int nwaves, int lane) {
    for (int t = wave; t < T; t += nwaves) {
        float v[4], s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = xs[t * D + lane + 64 * i]; s1 += v[i]; }
        const float mean = wave_sum(s1) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            out[(long)t * D + c] = f2e(((v[i] - mean) * rstd) * e2f(ln.w[c]) + e2f(ln.b[c]));
        }
    }
}

// ---- token self attention + LayerNorm (transformer.py:151-160) ---------------------------------------------------------------------
// first layer: q = k = v = queries, queries <- attn (no residual); else q = k = queries + query_pe, v = queries, queries <- queries + attn.
__global__ __launch_bounds__(256) void sam_token_self_attn_ln_kernel(const elem_t* __restrict__ queries, const elem_t* __restrict__ qpe, int T, int first,
                                                                     LinW wq, LinW wk, LinW wv, LinW wo, LnW ln, float eps, elem_t* __restrict__ out) {
    __shared__ float xq[TMAX * D], xv[TMAX * D], qs[TMAX * D], ks[TMAX * D], vs[TMAX * D], pr[NH * TMAX * TMAX];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const elem_t* qr = queries + (long)n * T * D;
    const elem_t* pe = qpe + (long)n * T * D;
    for (int e = tid; e < T * D; e += 256) {
        const float a = e2f(qr[e]);
        xv[e] = a;
        xq[e] = first ? a : rnd(a + e2f(pe[e]));
    }
    __syncthreads();
    {   // projections: thread o owns output feature o of q, k, v for every token
        const int o = tid;
        float aq[TMAX], ak[TMAX], av[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) aq[t] = ak[t] = av[t] = 0.f;
        for (int c = 0; c < D; c += 8) {
            float a[8], b[8], cc[8];
            unpack8(*(const uint4*)(wq.w + (long)o * D + c), a);
            unpack8(*(const uint4*)(wk.w + (long)o * D + c), b);
            unpack8(*(const uint4*)(wv.w + (long)o * D + c), cc);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t < T) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        aq[t] += xq[t * D + c + j] * a[j];
                        ak[t] += xq[t * D + c + j] * b[j];
                        av[t] += xv[t * D + c + j] * cc[j];
                    }
                }
            }
        }
        const float bq = e2f(wq.b[o]), bk = e2f(wk.b[o]), bv = e2f(wv.b[o]);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) { qs[t * D + o] = rnd(aq[t] + bq); ks[t * D + o] = rnd(ak[t] + bk); vs[t * D + o] = rnd(av[t] + bv); }
    }
    __syncthreads();
    constexpr int HD = D / NH;                                // 32
    const float sq = sqrtf((float)HD);
    for (int e = tid; e < NH * T * T; e += 256) {             // scores: rnd(rnd(q k) / sqrt(hd))
        const int u = e % T, t = (e / T) % T, h = e / (T * T);
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc += qs[t * D + h * HD + d] * ks[u * D + h * HD + d];
        pr[(h * TMAX + t) * TMAX + u] = rnd(rnd(acc) / sq);
    }
    __syncthreads();
    if (tid < NH * T) {                                       // softmax over the T keys of one (head, query)
        const int t = tid % T, h = tid / T;
        float* row = pr + (h * TMAX + t) * TMAX;
        float m = -INFINITY, l = 0.f;
        for (int u = 0; u < T; ++u) m = fmaxf(m, row[u]);
        for (int u = 0; u < T; ++u) l += __expf(row[u] - m);
        const float inv = 1.0f / l;
        for (int u = 0; u < T; ++u) row[u] = rnd(__expf(row[u] - m) * inv);
    }
    __syncthreads();
    for (int e = tid; e < T * D; e += 256) {                  // o = P V -> xq (reused as the attention output)
        const int c = e % D, t = e / D, h = c / HD;
        float acc = 0.f;
        for (int u = 0; u < T; ++u) acc += pr[(h * TMAX + t) * TMAX + u] * vs[u * D + c];
        xq[e] = rnd(acc);
    }
    __syncthreads();
    {   // out projection (+ residual) -> qs (reused)
        const int o = tid;
        float acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
        for (int c = 0; c < D; c += 8) {
            float a[8];
            unpack8(*(const uint4*)(wo.w + (long)o * D + c), a);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t] += xq[t * D + c + j] * a[j];
                }
        }
        const float bo = e2f(wo.b[o]);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
                const float y = rnd(acc[t] + bo);
                qs[t * D + o] = first ? y : rnd(xv[t * D + o] + y);
            }
    }
    __syncthreads();
    layernorm_rows(qs, T, ln, eps, out + (long)n * T * D, wave, 4, lane);
}

// ---- MLP block + LayerNorm (transformer.py:168-171): queries <- LN(queries + lin2(relu(lin1(queries)))) ------------------------------
__global__ __launch_bounds__(1024) void sam_token_mlp_ln_kernel(const elem_t* __restrict__ queries, int T, int HID, LinW w1, LinW w2, LnW ln, float eps,
                                                                elem_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;                                 // [T][256]
    float* hs = xs + TMAX * D;                                // [T][HID]
    float* part = hs + TMAX * HID;                            // [4][T][256] partial sums of lin2
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const elem_t* qr = queries + (long)n * T * D;
    for (int e = tid; e < T * D; e += 1024) xs[e] = e2f(qr[e]);
    __syncthreads();
    for (int j = tid; j < HID; j += 1024) {
        float acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
        for (int c = 0; c < D; c += 8) {
            float a[8];
            unpack8(*(const uint4*)(w1.w + (long)j * D + c), a);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[t] += xs[t * D + c + k] * a[k];
                }
        }
        const float b = e2f(w1.b[j]);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) hs[t * HID + j] = fmaxf(rnd(acc[t] + b), 0.f);
    }
    __syncthreads();
    {   // lin2: output o = tid & 255, K split in 4 quarters (tid >> 8)
        const int o = tid & 255, qd = tid >> 8;
        const int k0 = qd * (HID / 4), k1 = k0 + HID / 4;
        float acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
        for (int c = k0; c < k1; c += 8) {
            float a[8];
            unpack8(*(const uint4*)(w2.w + (long)o * HID + c), a);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[t] += hs[t * HID + c + k] * a[k];
                }
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) part[(qd * TMAX + t) * D + o] = acc[t];
    }
    __syncthreads();
    for (int e = tid; e < T * D; e += 1024) {
        const int o = e % D, t = e / D;
        const float y = rnd(part[(0 * TMAX + t) * D + o] + part[(1 * TMAX + t) * D + o] + part[(2 * TMAX + t) * D + o] + part[(3 * TMAX + t) * D + o] +
                            e2f(w2.b[o]));
        hs[e] = rnd(xs[e] + y);                               // residual; hs reused as [T][256]
    }
    __syncthreads();
    layernorm_rows(hs, T, ln, eps, out + (long)n * T * D, wave, 16, lane);
}

// ---- the four hyper-network MLPs + the IoU head (mask_decoder.py:137-164): y = L3(relu(L2(relu(L1(hs[n, row]))))) ----------------------
struct Mlp3 { LinW l[3]; int row; int n_out; elem_t* out; int out_stride; };   // out[n * out_stride + o]
struct Mlp3Set { Mlp3 m[5]; };
__global__ __launch_bounds__(256) void sam_small_mlps_kernel(const elem_t* __restrict__ hs, int T, Mlp3Set set) {
    __shared__ float a0[D], a1[D];
    const int n = blockIdx.x, tid = threadIdx.x;
    const Mlp3& m = set.m[blockIdx.y];
    a0[tid] = e2f(hs[((long)n * T + m.row) * D + tid]);
    __syncthreads();
    a1[tid] = fmaxf(rnd(dot_row(a0, m.l[0].w + (long)tid * D, D) + e2f(m.l[0].b[tid])), 0.f);
    __syncthreads();
    a0[tid] = fmaxf(rnd(dot_row(a1, m.l[1].w + (long)tid * D, D) + e2f(m.l[1].b[tid])), 0.f);
    __syncthreads();
    if (tid < m.n_out) m.out[(long)n * m.out_stride + tid] = f2e(dot_row(a0, m.l[2].w + (long)tid * D, D) + e2f(m.l[2].b[tid]));
}

// ---- helpers of the two cross-attention kernels --------------------------------------------------------------------------------------
// stage `rows` image rows [*, 256] (+ positional rows) into an LDS tile [rows][PX] of 16-bit elements: tile = rnd(a + pe) or a
ULL_DEV void stage_rows(const elem_t* __restrict__ a, const elem_t* __restrict__ pe, elem_t* __restrict__ tile, int rows, int tid, int nthr) {
    for (int c = tid; c < rows * (D / 8); c += nthr) {
        const int r = c / (D / 8), ch = c % (D / 8);
        uint4 v = *(const uint4*)(a + (long)r * D + ch * 8);
        if (pe != nullptr) {
            float x[8], y[8];
            unpack8(v, x);
            unpack8(*(const uint4*)(pe + (long)r * D + ch * 8), y);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] += y[j];
            v = pack8(x);
        }
        *(uint4*)(tile + r * PX + ch * 8) = v;
    }
}

// Linear on the MFMA for one wave: out[feature f][row m] = sum_k W[f][k] * X[m][k] for the wave's 16 rows (LDS tile, pitch PXs) and
// NF * 16 features; K in steps of 32.  acc[i][r] = out[feature i*16 + 4*(lane>>4) + r][row (lane & 15)].
template <int NF, int KDIM, int PXs>
ULL_DEV void mfma_linear(const elem_t* __restrict__ W, const elem_t* __restrict__ xt /* wave's first row */, f32x4_t (&acc)[NF], int lane) {
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kk = 0; kk < KDIM / 32; ++kk) {
        const uint4 xf = *(const uint4*)(xt + fr * PXs + kk * 32 + fg * 8);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const uint4 wf = *(const uint4*)(W + (long)(i * 16 + fr) * KDIM + kk * 32 + fg * 8);
            acc[i] = mfma16(wf, xf, acc[i]);
        }
    }
}

ULL_DEV float lin_out(float acc, float bias, bool late_bias) { return late_bias ? rnd(rnd(acc) + bias) : rnd(acc + bias); }

// ---- token -> image attention, part 1: k / v projections of a 64-key tile + scores of the T tokens against it --------------------------
// scores [n, 8, TMAX, P] (16-bit, already scaled), vproj [n, P, 128].
__global__ __launch_bounds__(256) void sam_t2i_kv_scores_kernel(const elem_t* __restrict__ queries, const elem_t* __restrict__ qpe, int T,
                                                                const elem_t* __restrict__ keys, const elem_t* __restrict__ pos, int P, LinW wq, LinW wk,
                                                                LinW wv, int late_bias_kv, elem_t* __restrict__ scores, elem_t* __restrict__ vproj) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem_t* kin = (elem_t*)smem;                              // [KT][PX]  keys + pe
    elem_t* vin = kin + KT * PX;                              // [KT][PX]  keys
    elem_t* kt = vin + KT * PX;                               // [KT][PI]  projected k tile
    float* xq = (float*)(kt + KT * PI);                       // [T][256]  queries + pe
    float* qs = xq + TMAX * D;                                // [T][128]  projected q
    const int n = blockIdx.y, p0 = blockIdx.x * KT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const elem_t* kr = keys + ((long)n * P + p0) * D;
    stage_rows(kr, pos + (long)p0 * D, kin, KT, tid, 256);
    stage_rows(kr, nullptr, vin, KT, tid, 256);
    for (int e = tid; e < T * D; e += 256) xq[e] = rnd(e2f(queries[(long)n * T * D + e]) + e2f(qpe[(long)n * T * D + e]));
    __syncthreads();
    if (tid < DI) {                                           // q projection (T x 128, fused bias): thread o = tid
        float acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
        for (int c = 0; c < D; c += 8) {
            float a[8];
            unpack8(*(const uint4*)(wq.w + (long)tid * D + c), a);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t] += xq[t * D + c + j] * a[j];
                }
        }
        const float b = e2f(wq.b[tid]);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) qs[t * DI + tid] = rnd(acc[t] + b);
    }
    const int fr = lane & 15, fg = lane >> 4;
    {   // k tile -> LDS, v tile -> global; wave w owns keys 16w .. 16w+15
        f32x4_t acc[DI / 16];
        mfma_linear<DI / 16, D, PX>(wk.w, kin + wave * 16 * PX, acc, lane);
#pragma unroll
        for (int i = 0; i < DI / 16; ++i) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = lin_out(acc[i][r], e2f(wk.b[i * 16 + fg * 4 + r]), late_bias_kv);
            uint2 pk;
            pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
            *(uint2*)(kt + (wave * 16 + fr) * PI + i * 16 + fg * 4) = pk;
        }
        mfma_linear<DI / 16, D, PX>(wv.w, vin + wave * 16 * PX, acc, lane);
        elem_t* vo = vproj + ((long)n * P + p0 + wave * 16 + fr) * DI;
#pragma unroll
        for (int i = 0; i < DI / 16; ++i) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = lin_out(acc[i][r], e2f(wv.b[i * 16 + fg * 4 + r]), late_bias_kv);
            uint2 pk;
            pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
            *(uint2*)(vo + i * 16 + fg * 4) = pk;
        }
    }
    __syncthreads();
    constexpr int HD = DI / NH;                               // 16
    for (int e = tid; e < KT * NH; e += 256) {                // scores of (key j, head h) against every token
        const int j = e % KT, h = e / KT;
        float kv[HD];
        unpack8(*(const uint4*)(kt + j * PI + h * HD), kv);
        unpack8(*(const uint4*)(kt + j * PI + h * HD + 8), kv + 8);
        for (int t = 0; t < T; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc += qs[t * DI + h * HD + d] * kv[d];
            scores[(((long)n * NH + h) * TMAX + t) * P + p0 + j] = f2e(rnd(acc) * 0.25f);      // / sqrt(16): exact scaling
        }
    }
}

// ---- token -> image attention, part 2: exact softmax over all P keys, P V, out-projection, residual, LayerNorm --------------------------
__global__ __launch_bounds__(512) void sam_t2i_softmax_out_ln_kernel(const elem_t* __restrict__ scores, const elem_t* __restrict__ vproj,
                                                                     const elem_t* __restrict__ queries, int T, int P, LinW wo, LnW ln, float eps,
                                                                     elem_t* __restrict__ out) {
    __shared__ float os[TMAX * DI], xs[TMAX * D], mst[NH][TMAX], ist[NH][TMAX];
    constexpr int HD = DI / NH;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, h = tid >> 6;          // one wave per head
    const elem_t* sh = scores + ((long)n * NH + h) * TMAX * P;
    float* m = mst[h];
    float* inv = ist[h];
#pragma unroll 1
    for (int t = 0; t < T; ++t) {                             // statistics of every row (fp32 over the 16-bit scores)
        float mx = -INFINITY;
#pragma unroll 1
        for (int j = lane * 8; j < P; j += 512) {
            float s[8];
            unpack8(*(const uint4*)(sh + (long)t * P + j), s);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, s[e]);
        }
        mx = wave_max(mx);
        float l = 0.f;
#pragma unroll 1
        for (int j = lane * 8; j < P; j += 512) {
            float s[8];
            unpack8(*(const uint4*)(sh + (long)t * P + j), s);
#pragma unroll
            for (int e = 0; e < 8; ++e) l += __expf(s[e] - mx);
        }
        l = wave_sum(l);
        if (lane == 0) { m[t] = mx; inv[t] = 1.0f / l; }
    }
    __syncthreads();
    const elem_t* vh = vproj + (long)n * P * DI + h * HD;
    constexpr int TG = 4;                                     // tokens per pass (register budget: acc 64 + probabilities 32 VGPRs)
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += TG) {
        float acc[TG][HD];
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[t][d] = 0.f;
        float mg[TG], ig[TG];
#pragma unroll
        for (int t = 0; t < TG; ++t) { mg[t] = m[(t0 + t) & (TMAX - 1)]; ig[t] = inv[(t0 + t) & (TMAX - 1)]; }   // (rows >= T: unused)
#pragma unroll 2
        for (int j = lane; j < P; j += 64) {                  // one key per lane and iteration (keeps the V row + 4 probabilities in registers)
            float v[HD];
            unpack8(*(const uint4*)(vh + (long)j * DI), v);
            unpack8(*(const uint4*)(vh + (long)j * DI + 8), v + 8);
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const float pr = (t0 + t < T) ? rnd(__expf(e2f(sh[(long)(t0 + t) * P + j]) - mg[t]) * ig[t]) : 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) acc[t][d] += pr * v[d];
            }
        }
#pragma unroll
        for (int t = 0; t < TG; ++t)
            if (t0 + t < T) {
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    const float v = wave_sum(acc[t][d]);
                    if (lane == 0) os[(t0 + t) * DI + h * HD + d] = rnd(v);
                }
            }
    }
    __syncthreads();
#pragma unroll 1
    for (int e = tid; e < T * D; e += 512) {                  // out projection + residual
        const int o = e % D, t = e / D;
        const float y = rnd(dot_row(os + t * DI, wo.w + (long)o * DI, DI) + e2f(wo.b[o]));
        xs[e] = rnd(e2f(queries[(long)n * T * D + e]) + y);
    }
    __syncthreads();
    layernorm_rows(xs, T, ln, eps, out + (long)n * T * D, h, 8, lane);
}

// ---- image -> token attention, fused (transformer.py:173-180): keys <- LN(keys + attn(q = keys + pe, k = queries + qpe, v = queries)) --
__global__ __launch_bounds__(256) void sam_i2t_fused_kernel(const elem_t* __restrict__ keys, const elem_t* __restrict__ pos, int P,
                                                            const elem_t* __restrict__ queries, const elem_t* __restrict__ qpe, int T, LinW wq, LinW wk,
                                                            LinW wv, LinW wo, int late_bias_q, LnW ln, float eps, elem_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem_t* kin = (elem_t*)smem;                              // [KT][PX]  keys + pe; later the pre-LayerNorm rows
    elem_t* qt = kin + KT * PX;                               // [KT][PI]  projected q tile
    elem_t* ot = qt + KT * PI;                                // [KT][PI]  attention output tile
    float* xq = (float*)(ot + KT * PI);                       // [T][256]  queries + qpe
    float* xv = xq + TMAX * D;                                // [T][256]  queries
    float* ks = xv + TMAX * D;                                // [T][128]
    float* vs = ks + TMAX * DI;                               // [T][128]
    const int n = blockIdx.y, p0 = blockIdx.x * KT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const elem_t* kr = keys + ((long)n * P + p0) * D;
    stage_rows(kr, pos + (long)p0 * D, kin, KT, tid, 256);
    for (int e = tid; e < T * D; e += 256) {
        const float a = e2f(queries[(long)n * T * D + e]);
        xv[e] = a;
        xq[e] = rnd(a + e2f(qpe[(long)n * T * D + e]));
    }
    __syncthreads();
    {   // k / v of the tokens: thread -> (matrix, feature)
        const int o = tid & 127, which = tid >> 7;            // 0: k from xq, 1: v from xv
        const LinW& w = which ? wv : wk;
        const float* x = which ? xv : xq;
        float acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
        for (int c = 0; c < D; c += 8) {
            float a[8];
            unpack8(*(const uint4*)(w.w + (long)o * D + c), a);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t] += x[t * D + c + j] * a[j];
                }
        }
        const float b = e2f(w.b[o]);
        float* dst = which ? vs : ks;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) dst[t * DI + o] = rnd(acc[t] + b);
    }
    {   // q tile on the MFMA
        f32x4_t acc[DI / 16];
        mfma_linear<DI / 16, D, PX>(wq.w, kin + wave * 16 * PX, acc, lane);
#pragma unroll
        for (int i = 0; i < DI / 16; ++i) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = lin_out(acc[i][r], e2f(wq.b[i * 16 + fg * 4 + r]), late_bias_q);
            uint2 pk;
            pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
            *(uint2*)(qt + (wave * 16 + fr) * PI + i * 16 + fg * 4) = pk;
        }
    }
    __syncthreads();
    constexpr int HD = DI / NH;
    for (int e = tid; e < KT * NH; e += 256) {                // (image row j, head h): softmax over the T tokens, P V
        const int j = e % KT, h = e / KT;
        float q[HD], s[TMAX];
        unpack8(*(const uint4*)(qt + j * PI + h * HD), q);
        unpack8(*(const uint4*)(qt + j * PI + h * HD + 8), q + 8);
        float mx = -INFINITY;
        for (int t = 0; t < T; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc += q[d] * ks[t * DI + h * HD + d];
            s[t] = rnd(rnd(acc) * 0.25f);
            mx = fmaxf(mx, s[t]);
        }
        float l = 0.f;
        for (int t = 0; t < T; ++t) l += __expf(s[t] - mx);
        const float inv = 1.0f / l;
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
        for (int t = 0; t < T; ++t) {
            const float pr = rnd(__expf(s[t] - mx) * inv);
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] += pr * vs[t * DI + h * HD + d];
        }
        *(uint4*)(ot + j * PI + h * HD) = pack8(o);
        *(uint4*)(ot + j * PI + h * HD + 8) = pack8(o + 8);
    }
    __syncthreads();
    {   // out projection on the MFMA (256 features x 16 rows per wave), + residual -> kin (reused), then LayerNorm
        f32x4_t acc[D / 16];
        mfma_linear<D / 16, DI, PI>(wo.w, ot + wave * 16 * PI, acc, lane);
        const elem_t* res = kr + (long)(wave * 16 + fr) * D;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = i * 16 + fg * 4 + r;
                o[r] = rnd(e2f(res[f]) + rnd(acc[i][r] + e2f(wo.b[f])));
            }
            uint2 pk;
            pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
            *(uint2*)(kin + (wave * 16 + fr) * PX + i * 16 + fg * 4) = pk;      // each wave rewrites only its own 16 rows
        }
    }
    __syncthreads();
    for (int r = wave; r < KT; r += 4) {                      // LayerNorm, one wave per row
        float v[4], s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = e2f(kin[r * PX + lane + 64 * i]); s1 += v[i]; }
        const float mean = wave_sum(s1) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
        elem_t* orow = out + ((long)n * P + p0 + r) * D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            orow[c] = f2e(((v[i] - mean) * rstd) * e2f(ln.w[c]) + e2f(ln.b[c]));
        }
    }
}

inline LinW lw(const void* w, const void* b) { return LinW{(const elem_t*)w, (const elem_t*)b}; }

}  // namespace

// queries / qpe [n, T, 256]; weights nn.Linear layout; out [n, T, 256].  first != 0: layer 0 (no positional add, no residual).
extern "C" int ULL_FN(ull_sam_token_self_attn_ln_)(const void* queries, const void* qpe, int64_t n, int64_t T, int first, const void* wq, const void* bq,
                                               const void* wk, const void* bk, const void* wv, const void* bv, const void* wo, const void* bo,
                                               const void* ln_w, const void* ln_b, float eps, void* out, void* stream) {
    if (!queries || !qpe || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !ln_w || !ln_b || !out || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(sam_token_self_attn_ln_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (const elem_t*)queries, (const elem_t*)qpe,
                       (int)T, first, lw(wq, bq), lw(wk, bk), lw(wv, bv), lw(wo, bo), LnW{(const elem_t*)ln_w, (const elem_t*)ln_b}, eps, (elem_t*)out);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_sam_token_mlp_ln_)(const void* queries, int64_t n, int64_t T, int64_t hidden, const void* w1, const void* b1, const void* w2,
                                         const void* b2, const void* ln_w, const void* ln_b, float eps, void* out, void* stream) {
    if (!queries || !w1 || !b1 || !w2 || !b2 || !ln_w || !ln_b || !out || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX || hidden <= 0 || (hidden & 31) || hidden > 2048) return ULL_ERR_SHAPE;
    const size_t lds = (size_t)(TMAX * D + TMAX * hidden + 4 * TMAX * D) * sizeof(float);
    static UllOncePerDevice once;
    if (once.first() && hipFuncSetAttribute((const void*)sam_token_mlp_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return ULL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sam_token_mlp_ln_kernel, dim3((unsigned)n), dim3(1024), lds, (hipStream_t)stream, (const elem_t*)queries, (int)T, (int)hidden,
                       lw(w1, b1), lw(w2, b2), LnW{(const elem_t*)ln_w, (const elem_t*)ln_b}, eps, (elem_t*)out);
    return ull_check_launch();
}

// ptrs: 5 MLPs x {w1, b1, w2, b2, w3, b3} (hyper-network MLPs 0..3, then the IoU head) = 30 device pointers; hs [n, T, 256] ->
// hyper [n, 4, 32], iou [n, n_iou].
extern "C" int ULL_FN(ull_sam_small_mlps_)(const void* hs, int64_t n, int64_t T, const void* const* ptrs, int64_t n_mask_tokens, int64_t hyper_out,
                                       int64_t n_iou, void* hyper, void* iou, void* stream) {
    if (!hs || !ptrs || !hyper || !iou || n <= 0) return ULL_ERR_ARG;
    if (n_mask_tokens != 4 || hyper_out > 256 || n_iou > 256 || T < 1 + n_mask_tokens || T > TMAX) return ULL_ERR_SHAPE;
    Mlp3Set set;
    for (int i = 0; i < 5; ++i) {
        for (int l = 0; l < 3; ++l) set.m[i].l[l] = lw(ptrs[i * 6 + 2 * l], ptrs[i * 6 + 2 * l + 1]);
        if (i < 4) { set.m[i].row = 1 + i; set.m[i].n_out = (int)hyper_out; set.m[i].out = (elem_t*)hyper + i * hyper_out; set.m[i].out_stride = (int)(4 * hyper_out); }
        else { set.m[i].row = 0; set.m[i].n_out = (int)n_iou; set.m[i].out = (elem_t*)iou; set.m[i].out_stride = (int)n_iou; }
    }
    hipLaunchKernelGGL(sam_small_mlps_kernel, dim3((unsigned)n, 5), dim3(256), 0, (hipStream_t)stream, (const elem_t*)hs, (int)T, set);
    return ull_check_launch();
}

// token -> image attention of TwoWayAttentionBlock / final_attn_token_to_image, two launches on `stream`.  keys [n, P, 256], pos [P, 256],
// queries / qpe [n, T, 256]; scratch: scores [n, 8, 8, P] + vproj [n, P, 128] elements (caller-owned); out [n, T, 256] = new queries.
extern "C" int ULL_FN(ull_sam_t2i_attention_ln_)(const void* queries, const void* qpe, const void* keys, const void* pos, int64_t n, int64_t T, int64_t P,
                                             const void* wq, const void* bq, const void* wk, const void* bk, const void* wv, const void* bv, const void* wo,
                                             const void* bo, int late_bias_kv, const void* ln_w, const void* ln_b, float eps, void* scores_ws,
                                             void* vproj_ws, void* out, void* stream) {
    if (!queries || !qpe || !keys || !pos || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !ln_w || !ln_b || !scores_ws || !vproj_ws || !out ||
        n <= 0)
        return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX || P <= 0 || (P % 512)) return ULL_ERR_SHAPE;
    const size_t lds = (size_t)(2 * KT * PX + KT * PI) * sizeof(elem_t) + (size_t)(TMAX * D + TMAX * DI) * sizeof(float);
    static UllOncePerDevice once;
    if (once.first() && hipFuncSetAttribute((const void*)sam_t2i_kv_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return ULL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sam_t2i_kv_scores_kernel, dim3((unsigned)(P / KT), (unsigned)n), dim3(256), lds, (hipStream_t)stream, (const elem_t*)queries,
                       (const elem_t*)qpe, (int)T, (const elem_t*)keys, (const elem_t*)pos, (int)P, lw(wq, bq), lw(wk, bk), lw(wv, bv), late_bias_kv,
                       (elem_t*)scores_ws, (elem_t*)vproj_ws);
    hipLaunchKernelGGL(sam_t2i_softmax_out_ln_kernel, dim3((unsigned)n), dim3(512), 0, (hipStream_t)stream, (const elem_t*)scores_ws,
                       (const elem_t*)vproj_ws, (const elem_t*)queries, (int)T, (int)P, lw(wo, bo), LnW{(const elem_t*)ln_w, (const elem_t*)ln_b}, eps,
                       (elem_t*)out);
    return ull_check_launch();
}

// image -> token attention of TwoWayAttentionBlock in one launch: out [n, P, 256] = LN(keys + attn(keys + pos, queries + qpe, queries)).
extern "C" int ULL_FN(ull_sam_i2t_attention_ln_)(const void* keys, const void* pos, const void* queries, const void* qpe, int64_t n, int64_t T, int64_t P,
                                             const void* wq, const void* bq, const void* wk, const void* bk, const void* wv, const void* bv, const void* wo,
                                             const void* bo, int late_bias_q, const void* ln_w, const void* ln_b, float eps, void* out, void* stream) {
    if (!keys || !pos || !queries || !qpe || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !ln_w || !ln_b || !out || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX || P <= 0 || (P % KT) || keys == out) return ULL_ERR_SHAPE;
    const size_t lds = (size_t)(KT * PX + 2 * KT * PI) * sizeof(elem_t) + (size_t)(2 * TMAX * D + 2 * TMAX * DI) * sizeof(float);
    static UllOncePerDevice once;
    if (once.first() && hipFuncSetAttribute((const void*)sam_i2t_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return ULL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sam_i2t_fused_kernel, dim3((unsigned)(P / KT), (unsigned)n), dim3(256), lds, (hipStream_t)stream, (const elem_t*)keys,
                       (const elem_t*)pos, (int)P, (const elem_t*)queries, (const elem_t*)qpe, (int)T, lw(wq, bq), lw(wk, bk), lw(wv, bv), lw(wo, bo),
                       late_bias_q, LnW{(const elem_t*)ln_w, (const elem_t*)ln_b}, eps, (elem_t*)out);
    return ull_check_launch();
}
