// Fused kernels of SAM's two-way mask decoder (north_star: "SAM's MaskDecoder cross-attention fused into one LDS-resident kernel").
//
// reference: models/segment_anything/modeling/transformer.py:62-106 (TwoWayTransformer), :151-182 (TwoWayAttentionBlock),
// :220-242 (Attention), mask_decoder.py:137-164 (hyper-network MLPs, IoU head).  One decode used to be ~70 launches of generic
// kernels (3 Linears + add + transpose + attention + Linear + LayerNorm per attention); here every attention of the block is one or
// a few launches that keep the projected tiles in LDS and spread the weight streams over many CUs:
//   sam_self_attn_heads  : (prompt, head) blocks: q/k/v projections of the head, attention over the T tokens
//   sam_out_ln           : out-projection (+ residual) + LayerNorm of the T token rows (+ the next attention's token-side projection)
//   sam_t2i_kv_scores    : 128-key tiles: v = keys Wv^T and k = (keys + pe) Wk^T on the MFMA from ONE LDS-resident tile (weights
//                          streamed through LDS in K-slices), scaled scores of the T tokens against the tile
//   sam_t2i_softmax_pv   : (prompt, head) blocks of 16 waves: exact fp32 softmax over all 4096 keys, P V
//   sam_mlp_partial / sam_mlp_reduce_ln : the 256 -> 2048 -> 256 MLP split over 8 hidden chunks per prompt, then sum + residual +
//                          LayerNorm (+ the token-side k / v projections of the image->token attention that follows)
//   sam_i2t_fused        : 128 image rows per block: q = (keys + pe) Wq^T on the MFMA, softmax over the T tokens, P V, out-projection on
//                          the MFMA, residual, LayerNorm -- the image->token attention in ONE launch, only the new keys are written
//   sam_small_mlps       : the four hyper-network MLPs and the IoU head (3-layer MLPs on single token rows)
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

// Cooperative version for the token-side kernels (a few rows, weight-latency bound): LPO consecutive lanes share one output feature, lane
// `sub` sums its K / LPO slice for every token row (x rows XS floats apart), xor-shuffles add the slices -- a 256-long dot becomes 8
// dependent 16-byte weight loads per lane instead of 32.  acc[t] holds the full sum in all LPO lanes.
constexpr int LPO = 4;
ULL_DEV void dot_rows_coop(const float* __restrict__ x, int XS, const elem_t* __restrict__ w, int K, int T, int sub, float (&acc)[TMAX]) {
#pragma unroll
    for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
    const int kc = K / LPO;
    for (int c = sub * kc; c < (sub + 1) * kc; c += 8) {
        float a[8];
        unpack8(*(const uint4*)(w + c), a);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[t] += x[t * XS + c + j] * a[j];
            }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        acc[t] += __shfl_xor(acc[t], 1, 64);
        acc[t] += __shfl_xor(acc[t], 2, 64);
    }
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

// LayerNorm of T rows of 256 floats in LDS (already rounded values), one wave per row; writes the 16-bit output AND leaves the rounded
// output in LDS (for projections that follow)
ULL_DEV void layernorm_rows_keep(float* __restrict__ xs, int T, LnW ln, float eps, elem_t* __restrict__ out, int wave, int nwaves, int lane) {
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
            const float y = rnd(((v[i] - mean) * rstd) * e2f(ln.w[c]) + e2f(ln.b[c]));
            out[(long)t * D + c] = f2e(y);
            xs[t * D + c] = y;
        }
    }
}

// token-side projections emitted by the kernels that produce new queries: p[t][o] = rnd(dot(x_t (+ qpe_t), W[o]) + b[o]), o < 128
struct Proj { LinW w; int add_pe; elem_t* out; };          // out [n, T, 128]; w.w == nullptr: unused
struct ProjSet { Proj p[3]; };
ULL_DEV void token_projections(const float* __restrict__ xs /* LDS [T][256], rounded */, float* __restrict__ tmp /* LDS [T][256] */,
                               const elem_t* __restrict__ qpe, int T, const ProjSet& ps, long n, int tid, int nthr) {
    for (int k = 0; k < 3; ++k) {
        const Proj& pr = ps.p[k];
        if (pr.w.w == nullptr) continue;                      // (uniform across the block)
        __syncthreads();
        for (int e = tid; e < T * D; e += nthr) tmp[e] = pr.add_pe ? rnd(xs[e] + e2f(qpe[n * T * D + e])) : xs[e];
        __syncthreads();
        for (int e = tid; e < DI * LPO; e += nthr) {          // (nthr is a multiple of LPO; DI * LPO = 512 work items)
            const int o = e / LPO, sub = e % LPO;
            float acc[TMAX];
            dot_rows_coop(tmp, D, pr.w.w + (long)o * D, D, T, sub, acc);
            const float b = e2f(pr.w.b[o]);
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
                if (t < T && sub == (t & (LPO - 1))) pr.out[(n * T + t) * DI + o] = f2e(acc[t] + b);
        }
    }
}

// ---- token self attention, one (prompt, head) per block (transformer.py:151-160, 220-242; head dim 32) ---------------------------------
// first layer: q = k = v = queries; else q = k = queries + query_pe, v = queries.  o [n, T, 256] = concatenated heads (rounded).
__global__ __launch_bounds__(384) void sam_self_attn_heads_kernel(const elem_t* __restrict__ queries, const elem_t* __restrict__ qpe, int T, int first,
                                                                  LinW wq, LinW wk, LinW wv, elem_t* __restrict__ o) {
    constexpr int HD = D / NH;                                // 32
    __shared__ float xq[TMAX * D], xv[TMAX * D], q[TMAX * HD], k[TMAX * HD], v[TMAX * HD], pr[TMAX * TMAX];
    const int h = blockIdx.x, tid = threadIdx.x;
    const long n = blockIdx.y;
    for (int e = tid; e < T * D; e += 384) {
        const float a = e2f(queries[n * T * D + e]);
        xv[e] = a;
        xq[e] = first ? a : rnd(a + e2f(qpe[n * T * D + e]));
    }
    __syncthreads();
    {   // 3 x 32 output features, LPO lanes each
        const int oi = tid / LPO, sub = tid % LPO;
        const int which = oi / HD, f = h * HD + oi % HD;
        const LinW& w = which == 0 ? wq : (which == 1 ? wk : wv);
        const float* x = which == 2 ? xv : xq;
        float acc[TMAX];
        dot_rows_coop(x, D, w.w + (long)f * D, D, T, sub, acc);
        float* dst = which == 0 ? q : (which == 1 ? k : v);
        const float b = e2f(w.b[f]);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T && sub == (t & (LPO - 1))) dst[t * HD + oi % HD] = rnd(acc[t] + b);
    }
    __syncthreads();
    const float sq = sqrtf((float)HD);
    if (tid < T * T) {                                        // scores: rnd(rnd(q k) / sqrt(hd))
        const int u = tid % T, t = tid / T;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc += q[t * HD + d] * k[u * HD + d];
        pr[t * TMAX + u] = rnd(rnd(acc) / sq);
    }
    __syncthreads();
    if (tid < T) {
        float* row = pr + tid * TMAX;
        float m = -INFINITY, l = 0.f;
        for (int u = 0; u < T; ++u) m = fmaxf(m, row[u]);
        for (int u = 0; u < T; ++u) l += __expf(row[u] - m);
        const float inv = 1.0f / l;
        for (int u = 0; u < T; ++u) row[u] = rnd(__expf(row[u] - m) * inv);
    }
    __syncthreads();
    for (int e = tid; e < T * HD; e += 384) {
        const int d = e % HD, t = e / HD;
        float acc = 0.f;
        for (int u = 0; u < T; ++u) acc += pr[t * TMAX + u] * v[u * HD + d];
        o[(n * T + t) * D + h * HD + d] = f2e(acc);
    }
}

// ---- out-projection (+ residual) + LayerNorm of the T token rows (+ projections for the next attention) --------------------------------
// att [n, T, DIN] -> out = LN(res + (att Wo^T + bo)) (res == nullptr: no residual: layer 0's self attention replaces the queries)
template <int DIN>
__global__ __launch_bounds__(1024) void sam_out_ln_kernel(const elem_t* __restrict__ att, const elem_t* __restrict__ res, const elem_t* __restrict__ qpe, int T,
                                                         LinW wo, LnW ln, float eps, elem_t* __restrict__ out, ProjSet ps) {
    __shared__ float as[TMAX * DIN], xs[TMAX * D], tmp[TMAX * D];
    const long n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < T * DIN; e += 1024) as[e] = e2f(att[n * T * DIN + e]);
    __syncthreads();
    {
        const int o = tid / LPO, sub = tid % LPO;             // 256 output features x LPO lanes
        float acc[TMAX];
        dot_rows_coop(as, DIN, wo.w + (long)o * DIN, DIN, T, sub, acc);
        const float bo = e2f(wo.b[o]);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T && sub == (t & (LPO - 1))) {
                const float y = rnd(acc[t] + bo);
                xs[t * D + o] = res != nullptr ? rnd(e2f(res[(n * T + t) * D + o]) + y) : y;
            }
    }
    __syncthreads();
    layernorm_rows_keep(xs, T, ln, eps, out + n * T * D, wave, 16, lane);
    token_projections(xs, tmp, qpe, T, ps, n, tid, 1024);
}

// ---- MLP block (transformer.py:168-171), hidden dimension split over gridDim.x chunks of 256 units ---------------------------------------
// part [n, chunks, T, 256] fp32 = partial lin2 sums of the chunk's hidden units (lin1 + ReLU computed in LDS)
__global__ __launch_bounds__(1024) void sam_mlp_partial_kernel(const elem_t* __restrict__ queries, int T, int HID, LinW w1, LinW w2, float* __restrict__ part) {
    __shared__ float xs[TMAX * D], hs[TMAX * 256];
    const int ch = blockIdx.x, tid = threadIdx.x;
    const long n = blockIdx.y;
    for (int e = tid; e < T * D; e += 1024) xs[e] = e2f(queries[n * T * D + e]);
    __syncthreads();
    const int oi = tid / LPO, sub = tid % LPO;                // 256 units x LPO lanes
    {
        const int j = ch * 256 + oi;
        float acc[TMAX];
        dot_rows_coop(xs, D, w1.w + (long)j * D, D, T, sub, acc);
        const float b = e2f(w1.b[j]);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T && sub == (t & (LPO - 1))) hs[t * 256 + oi] = fmaxf(rnd(acc[t] + b), 0.f);
    }
    __syncthreads();
    {
        float acc[TMAX];
        dot_rows_coop(hs, 256, w2.w + (long)oi * HID + ch * 256, 256, T, sub, acc);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T && sub == (t & (LPO - 1))) part[((n * gridDim.x + ch) * TMAX + t) * D + oi] = acc[t];
    }
}

// out = LN(queries + (sum of the chunk partials + b2)) (+ projections for the attention that follows)
__global__ __launch_bounds__(1024) void sam_mlp_reduce_ln_kernel(const float* __restrict__ part, int chunks, const elem_t* __restrict__ queries,
                                                                const elem_t* __restrict__ qpe, int T, const elem_t* __restrict__ b2, LnW ln, float eps,
                                                                elem_t* __restrict__ out, ProjSet ps) {
    __shared__ float xs[TMAX * D], tmp[TMAX * D];
    const long n = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < T * D; e += 1024) {
        const int o = e % D, t = e / D;
        float acc = 0.f;
        for (int c = 0; c < chunks; ++c) acc += part[((n * chunks + c) * TMAX + t) * D + o];
        xs[e] = rnd(e2f(queries[n * T * D + e]) + rnd(acc + e2f(b2[o])));
    }
    __syncthreads();
    layernorm_rows_keep(xs, T, ln, eps, out + n * T * D, wave, 16, lane);
    token_projections(xs, tmp, qpe, T, ps, n, tid, 1024);
}

// ---- the four hyper-network MLPs + the IoU head (mask_decoder.py:137-164): y = L3(relu(L2(relu(L1(hs[n, row]))))) ----------------------
struct Mlp3 { LinW l[3]; int row; int n_out; elem_t* out; int out_stride; };   // out[n * out_stride + o]
struct Mlp3Set { Mlp3 m[5]; };
__global__ __launch_bounds__(1024) void sam_small_mlps_kernel(const elem_t* __restrict__ hs, int T, Mlp3Set set) {
    __shared__ float a0[D], a1[D];
    const int n = blockIdx.x, tid = threadIdx.x, o = tid / LPO, sub = tid % LPO;
    const Mlp3& m = set.m[blockIdx.y];
    if (tid < D) a0[tid] = e2f(hs[((long)n * T + m.row) * D + tid]);
    __syncthreads();
    float acc[TMAX];
    dot_rows_coop(a0, D, m.l[0].w + (long)o * D, D, 1, sub, acc);
    if (sub == 0) a1[o] = fmaxf(rnd(acc[0] + e2f(m.l[0].b[o])), 0.f);
    __syncthreads();
    dot_rows_coop(a1, D, m.l[1].w + (long)o * D, D, 1, sub, acc);
    __syncthreads();
    if (sub == 0) a0[o] = fmaxf(rnd(acc[0] + e2f(m.l[1].b[o])), 0.f);
    __syncthreads();
    const int oc = o < m.n_out ? o : 0;                       // (all lanes run the shuffles)
    dot_rows_coop(a0, D, m.l[2].w + (long)oc * D, D, 1, sub, acc);
    if (sub == 0 && o < m.n_out) m.out[(long)n * m.out_stride + o] = f2e(acc[0] + e2f(m.l[2].b[o]));
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

// Block-cooperative Linear on the MFMA: out[feature f][row m] = sum_k W[f][k] * X[m][k] for the block's X tile in LDS (pitch PXs); every wave
// owns RT row tiles of 16 rows starting at `xt`; W [NF*16, KDIM] is streamed through the LDS buffer `wsl` ([NF*16][WP]) in 32-wide
// K-slices, so each weight byte is read from L2 once per block.  acc[rt][i][r] = out[feature i*16 + 4*(lane>>4) + r][row rt*16 + (lane&15)].
constexpr int WP = 40;                                        // slice pitch (elements): 80-byte rows -> conflict-free ds_read_b128
template <int NF, int KDIM, int PXs, int RT>
ULL_DEV void block_linear(const elem_t* __restrict__ W, elem_t* __restrict__ wsl, const elem_t* __restrict__ xt, f32x4_t (&acc)[RT][NF], int tid, int nthr,
                          int lane) {
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < NF; ++i) acc[rt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < KDIM / 32; ++kk) {
        __syncthreads();                                      // the previous slice has been consumed by every wave
        for (int c = tid; c < NF * 16 * 4; c += nthr) {
            const int r = c >> 2, ch = c & 3;
            *(uint4*)(wsl + r * WP + ch * 8) = *(const uint4*)(W + (long)r * KDIM + kk * 32 + ch * 8);
        }
        __syncthreads();
        uint4 xf[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xf[rt] = *(const uint4*)(xt + (rt * 16 + fr) * PXs + kk * 32 + fg * 8);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const uint4 wf = *(const uint4*)(wsl + (i * 16 + fr) * WP + fg * 8);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][i] = mfma16(wf, xf[rt], acc[rt][i]);
        }
    }
}

ULL_DEV float lin_out(float acc, float bias, bool late_bias) { return late_bias ? rnd(rnd(acc) + bias) : rnd(acc + bias); }

constexpr int KTB = 128;                                      // image rows per block in the cross-attention kernels
constexpr int RT = 2;                                         // 16-row MFMA tiles per wave (4 waves x 2 x 16 = 128 rows)

// ---- token -> image attention, part 1: v / k projections of a 128-key tile + scores of the T tokens against it -------------------------
// qproj [n, T, 128] (from sam_out_ln / sam_mlp_reduce_ln); scores [n, 8, TMAX, P] (16-bit, already scaled); vproj [n, P, 128].
__global__ __launch_bounds__(256) void sam_t2i_kv_scores_kernel(const elem_t* __restrict__ qproj, int T, const elem_t* __restrict__ keys,
                                                                const elem_t* __restrict__ pos, int P, LinW wk, LinW wv, int late_bias_kv,
                                                                elem_t* __restrict__ scores, elem_t* __restrict__ vproj) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem_t* tile = (elem_t*)smem;                             // [KTB][PX]  keys, then keys + pe
    elem_t* kt = tile + KTB * PX;                             // [KTB][PI]  projected k tile
    elem_t* wsl = kt + KTB * PI;                              // [128][WP]  weight K-slice
    float* qs = (float*)(wsl + DI * WP);                      // [T][128]
    const long n = blockIdx.y;
    const int p0 = blockIdx.x * KTB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const elem_t* kr = keys + (n * P + p0) * D;
    stage_rows(kr, nullptr, tile, KTB, tid, 256);
    for (int e = tid; e < T * DI; e += 256) qs[e] = e2f(qproj[n * T * DI + e]);
    f32x4_t acc[RT][DI / 16];
    block_linear<DI / 16, D, PX, RT>(wv.w, wsl, tile + wave * (RT * 16) * PX, acc, tid, 256, lane);       // v = keys Wv^T
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        elem_t* vo = vproj + (n * P + p0 + wave * (RT * 16) + rt * 16 + fr) * DI;
#pragma unroll
        for (int i = 0; i < DI / 16; ++i) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = lin_out(acc[rt][i][r], e2f(wv.b[i * 16 + fg * 4 + r]), late_bias_kv);
            uint2 pk;
            pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
            *(uint2*)(vo + i * 16 + fg * 4) = pk;
        }
    }
    __syncthreads();                                          // every wave is done with the plain keys tile
    for (int c = tid; c < KTB * (D / 8); c += 256) {          // tile <- rnd(keys + pe), in place
        const int r = c / (D / 8), ch = c % (D / 8);
        float x[8], y[8];
        unpack8(*(const uint4*)(tile + r * PX + ch * 8), x);
        unpack8(*(const uint4*)(pos + (long)(p0 + r) * D + ch * 8), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        *(uint4*)(tile + r * PX + ch * 8) = pack8(x);
    }
    block_linear<DI / 16, D, PX, RT>(wk.w, wsl, tile + wave * (RT * 16) * PX, acc, tid, 256, lane);       // k = (keys + pe) Wk^T
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < DI / 16; ++i) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = lin_out(acc[rt][i][r], e2f(wk.b[i * 16 + fg * 4 + r]), late_bias_kv);
            uint2 pk;
            pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
            *(uint2*)(kt + (wave * (RT * 16) + rt * 16 + fr) * PI + i * 16 + fg * 4) = pk;
        }
    __syncthreads();
    constexpr int HD = DI / NH;                               // 16
    for (int e = tid; e < KTB * NH; e += 256) {               // scores of (key j, head h) against every token
        const int j = e % KTB, h = e / KTB;
        float kv[HD];
        unpack8(*(const uint4*)(kt + j * PI + h * HD), kv);
        unpack8(*(const uint4*)(kt + j * PI + h * HD + 8), kv + 8);
        for (int t = 0; t < T; ++t) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a += qs[t * DI + h * HD + d] * kv[d];
            scores[((n * NH + h) * TMAX + t) * P + p0 + j] = f2e(rnd(a) * 0.25f);       // / sqrt(16): exact scaling
        }
    }
}

// ---- token -> image attention, part 2: one (prompt, head) per block of 16 waves: exact fp32 softmax over all P keys, P V ------------------
// att [n, T, 128] (the head's 16 columns).  P = 16 waves x 64 lanes x KPL keys.
template <int KPL>
__global__ __launch_bounds__(1024) void sam_t2i_softmax_pv_kernel(const elem_t* __restrict__ scores, const elem_t* __restrict__ vproj, int T, int P,
                                                                  elem_t* __restrict__ att) {
    constexpr int HD = DI / NH, NW = 16;
    __shared__ float red[TMAX][NW], stat[2][TMAX], part[NW][TMAX * HD];
    const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n = blockIdx.y;
    const elem_t* sh = scores + (n * NH + h) * TMAX * P;
    const int j0 = (wave * 64 + lane) * KPL;                  // this lane's KPL consecutive keys
    float sc[TMAX][KPL];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
#pragma unroll
        for (int e = 0; e < KPL; ++e) sc[t][e] = (t < T) ? e2f(sh[(long)t * P + j0 + e]) : 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        float mx = sc[t][0];
#pragma unroll
        for (int e = 1; e < KPL; ++e) mx = fmaxf(mx, sc[t][e]);
        mx = wave_max(mx);
        if (lane == 0) red[t][wave] = mx;
    }
    __syncthreads();
    if (tid < TMAX) {
        float mx = red[tid][0];
        for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[tid][w]);
        stat[0][tid] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const float mx = stat[0][t];
        float l = 0.f;
#pragma unroll
        for (int e = 0; e < KPL; ++e) l += __expf(sc[t][e] - mx);
        l = wave_sum(l);
        if (lane == 0) red[t][wave] = l;
    }
    __syncthreads();
    if (tid < TMAX) {
        float l = 0.f;
        for (int w = 0; w < NW; ++w) l += red[tid][w];
        stat[1][tid] = 1.0f / l;
    }
    __syncthreads();
    const elem_t* vh = vproj + n * P * DI + h * HD;
    float v[KPL][HD];
#pragma unroll
    for (int e = 0; e < KPL; ++e) {
        unpack8(*(const uint4*)(vh + (long)(j0 + e) * DI), v[e]);
        unpack8(*(const uint4*)(vh + (long)(j0 + e) * DI + 8), v[e] + 8);
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {                                          // (uniform)
            const float mx = stat[0][t], inv = stat[1][t];
            float acc[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
            for (int e = 0; e < KPL; ++e) {
                const float pr = rnd(__expf(sc[t][e] - mx) * inv);
#pragma unroll
                for (int d = 0; d < HD; ++d) acc[d] += pr * v[e][d];
            }
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                const float s_ = wave_sum(acc[d]);
                if (lane == 0) part[wave][t * HD + d] = s_;
            }
        }
    }
    __syncthreads();
    if (tid < T * HD) {
        float a = 0.f;
        for (int w = 0; w < NW; ++w) a += part[w][tid];
        att[(n * T + tid / HD) * DI + h * HD + tid % HD] = f2e(a);
    }
}

// ---- image -> token attention, fused (transformer.py:173-180): keys <- LN(keys + attn(q = keys + pe, k = queries + qpe, v = queries)) --
// kproj / vproj [n, T, 128]: the token-side projections (from sam_mlp_reduce_ln).
__global__ __launch_bounds__(256) void sam_i2t_fused_kernel(const elem_t* __restrict__ keys, const elem_t* __restrict__ pos, int P,
                                                            const elem_t* __restrict__ kproj, const elem_t* __restrict__ vproj, int T, LinW wq, LinW wo,
                                                            int late_bias_q, LnW ln, float eps, elem_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem_t* kin = (elem_t*)smem;                              // [KTB][PX]  keys + pe; later the pre-LayerNorm rows
    elem_t* qt = kin + KTB * PX;                              // [KTB][PI]  projected q tile; later Wo's K-slices
    elem_t* ot = qt + KTB * PI;                               // [KTB][PI]  Wq's K-slices, then the attention output tile
    float* ks = (float*)(ot + KTB * PI);                      // [T][128]
    float* vs = ks + TMAX * DI;                               // [T][128]
    const long n = blockIdx.y;
    const int p0 = blockIdx.x * KTB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const elem_t* kr = keys + (n * P + p0) * D;
    stage_rows(kr, pos + (long)p0 * D, kin, KTB, tid, 256);
    for (int e = tid; e < T * DI; e += 256) { ks[e] = e2f(kproj[n * T * DI + e]); vs[e] = e2f(vproj[n * T * DI + e]); }
    {
        f32x4_t acc[RT][DI / 16];
        block_linear<DI / 16, D, PX, RT>(wq.w, ot, kin + wave * (RT * 16) * PX, acc, tid, 256, lane);     // q = (keys + pe) Wq^T
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < DI / 16; ++i) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = lin_out(acc[rt][i][r], e2f(wq.b[i * 16 + fg * 4 + r]), late_bias_q);
                uint2 pk;
                pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
                *(uint2*)(qt + (wave * (RT * 16) + rt * 16 + fr) * PI + i * 16 + fg * 4) = pk;
            }
    }
    __syncthreads();
    constexpr int HD = DI / NH;
    for (int e = tid; e < KTB * NH; e += 256) {               // (image row j, head h): softmax over the T tokens, P V
        const int j = e % KTB, h = e / KTB;
        float q[HD], s[TMAX];
        unpack8(*(const uint4*)(qt + j * PI + h * HD), q);
        unpack8(*(const uint4*)(qt + j * PI + h * HD + 8), q + 8);
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            s[t] = -INFINITY;
            if (t < T) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) a += q[d] * ks[t * DI + h * HD + d];
                s[t] = rnd(a) * 0.25f;
                mx = fmaxf(mx, s[t]);
            }
        }
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) l += __expf(s[t] - mx);
        const float inv = 1.0f / l;
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
                const float pr = rnd(__expf(s[t] - mx) * inv);
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] += pr * vs[t * DI + h * HD + d];
            }
        *(uint4*)(ot + j * PI + h * HD) = pack8(o);
        *(uint4*)(ot + j * PI + h * HD + 8) = pack8(o + 8);
    }
    {   // out projection on the MFMA (256 features), + bias + residual -> kin rows of this wave, then LayerNorm
        f32x4_t acc[RT][D / 16];
        block_linear<D / 16, DI, PI, RT>(wo.w, qt, ot + wave * (RT * 16) * PI, acc, tid, 256, lane);     // (its first barrier orders the ot writes)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = wave * (RT * 16) + rt * 16 + fr;
            const elem_t* res = kr + (long)row * D;
#pragma unroll
            for (int i = 0; i < D / 16; ++i) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = i * 16 + fg * 4 + r;
                    o[r] = rnd(e2f(res[f]) + rnd(acc[rt][i][r] + e2f(wo.b[f])));
                }
                uint2 pk;
                pk.x = pack2e(o[0], o[1]); pk.y = pack2e(o[2], o[3]);
                *(uint2*)(kin + row * PX + i * 16 + fg * 4) = pk;          // kin was last read by the q projection: each wave rewrites its own rows
            }
        }
    }
    __syncthreads();
    for (int r = wave; r < KTB; r += 4) {                     // LayerNorm, one wave per row
        float v[4], s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = e2f(kin[r * PX + lane + 64 * i]); s1 += v[i]; }
        const float mean = wave_sum(s1) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
        elem_t* orow = out + (n * P + p0 + r) * D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            orow[c] = f2e(((v[i] - mean) * rstd) * e2f(ln.w[c]) + e2f(ln.b[c]));
        }
    }
}

inline LinW lw(const void* w, const void* b) { return LinW{(const elem_t*)w, (const elem_t*)b}; }

// projs: HOST array of 3 x {w, b, out} device pointers (w == NULL: unused) + add_pe flags
inline ProjSet make_projs(const void* const* projs, const int* add_pe) {
    ProjSet ps;
    for (int i = 0; i < 3; ++i) {
        ps.p[i].w = projs ? lw(projs[3 * i], projs[3 * i + 1]) : LinW{nullptr, nullptr};
        ps.p[i].out = projs ? (elem_t*)projs[3 * i + 2] : nullptr;
        ps.p[i].add_pe = add_pe ? add_pe[i] : 0;
    }
    return ps;
}

}  // namespace

// (prompt, head) blocks: q/k/v projections of one head + attention over the T tokens; att [n, T, 256].
extern "C" int ULL_FN(ull_sam_self_attn_heads_)(const void* queries, const void* qpe, int64_t n, int64_t T, int first, const void* wq, const void* bq,
                                            const void* wk, const void* bk, const void* wv, const void* bv, void* att, void* stream) {
    if (!queries || !qpe || !wq || !bq || !wk || !bk || !wv || !bv || !att || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(sam_self_attn_heads_kernel, dim3(NH, (unsigned)n), dim3(384), 0, (hipStream_t)stream, (const elem_t*)queries, (const elem_t*)qpe,
                       (int)T, first, lw(wq, bq), lw(wk, bk), lw(wv, bv), (elem_t*)att);
    return ull_check_launch();
}

// out [n, T, 256] = LayerNorm(res + att Wo^T + bo) (res NULL: no residual), att [n, T, din] with din = 256 or 128; then up to three
// token-side projections for the attention that follows: projs = HOST array of 3 x {w [128, 256], b [128], out [n, T, 128]} device
// pointers (w NULL = unused, projs NULL = none), add_pe[i] != 0: the projection's input is out + qpe.
extern "C" int ULL_FN(ull_sam_out_ln_)(const void* att, int64_t din, const void* res, const void* qpe, int64_t n, int64_t T, const void* wo, const void* bo,
                                   const void* ln_w, const void* ln_b, float eps, void* out, const void* const* projs, const int* add_pe, void* stream) {
    if (!att || !qpe || !wo || !bo || !ln_w || !ln_b || !out || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX || (din != D && din != DI)) return ULL_ERR_SHAPE;
    const ProjSet ps = make_projs(projs, add_pe);
    const LnW ln{(const elem_t*)ln_w, (const elem_t*)ln_b};
    if (din == D)
        hipLaunchKernelGGL(sam_out_ln_kernel<D>, dim3((unsigned)n), dim3(1024), 0, (hipStream_t)stream, (const elem_t*)att, (const elem_t*)res,
                           (const elem_t*)qpe, (int)T, lw(wo, bo), ln, eps, (elem_t*)out, ps);
    else
        hipLaunchKernelGGL(sam_out_ln_kernel<DI>, dim3((unsigned)n), dim3(1024), 0, (hipStream_t)stream, (const elem_t*)att, (const elem_t*)res,
                           (const elem_t*)qpe, (int)T, lw(wo, bo), ln, eps, (elem_t*)out, ps);
    return ull_check_launch();
}

// MLP block + LayerNorm in two launches: part_ws float32 [n * (hidden / 256) * 8 * 256] (caller-owned); projections as in ull_sam_out_ln.
extern "C" int ULL_FN(ull_sam_token_mlp_ln_)(const void* queries, const void* qpe, int64_t n, int64_t T, int64_t hidden, const void* w1, const void* b1,
                                         const void* w2, const void* b2, const void* ln_w, const void* ln_b, float eps, void* part_ws, void* out,
                                         const void* const* projs, const int* add_pe, void* stream) {
    if (!queries || !qpe || !w1 || !b1 || !w2 || !b2 || !ln_w || !ln_b || !part_ws || !out || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX || hidden <= 0 || (hidden & 255)) return ULL_ERR_SHAPE;
    const int chunks = (int)(hidden / 256);
    hipLaunchKernelGGL(sam_mlp_partial_kernel, dim3(chunks, (unsigned)n), dim3(1024), 0, (hipStream_t)stream, (const elem_t*)queries, (int)T, (int)hidden,
                       lw(w1, b1), lw(w2, b2), (float*)part_ws);
    hipLaunchKernelGGL(sam_mlp_reduce_ln_kernel, dim3((unsigned)n), dim3(1024), 0, (hipStream_t)stream, (const float*)part_ws, chunks, (const elem_t*)queries,
                       (const elem_t*)qpe, (int)T, (const elem_t*)b2, LnW{(const elem_t*)ln_w, (const elem_t*)ln_b}, eps, (elem_t*)out,
                       make_projs(projs, add_pe));
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
    hipLaunchKernelGGL(sam_small_mlps_kernel, dim3((unsigned)n, 5), dim3(1024), 0, (hipStream_t)stream, (const elem_t*)hs, (int)T, set);
    return ull_check_launch();
}

// token -> image attention core (transformer.py:162-166, :100-105), two launches: (1) 128-key tiles: v / k projections on the MFMA from one
// LDS-resident tile + scaled scores against qproj [n, T, 128]; (2) (prompt, head) blocks: exact fp32 softmax over all P keys and P V ->
// att [n, T, 128] (out-projection / residual / LayerNorm: ull_sam_out_ln).  Caller-owned scratch: scores_ws [n*8*8*P], vproj_ws [n*P*128].
extern "C" int ULL_FN(ull_sam_t2i_attention_)(const void* qproj, const void* keys, const void* pos, int64_t n, int64_t T, int64_t P, const void* wk,
                                          const void* bk, const void* wv, const void* bv, int late_bias_kv, void* scores_ws, void* vproj_ws, void* att,
                                          void* stream) {
    if (!qproj || !keys || !pos || !wk || !bk || !wv || !bv || !scores_ws || !vproj_ws || !att || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX || (P != 4096 && P != 1024)) return ULL_ERR_SHAPE;
    const size_t lds = (size_t)(KTB * PX + KTB * PI + DI * WP) * sizeof(elem_t) + (size_t)(TMAX * DI) * sizeof(float);
    static UllOncePerDevice once;
    if (once.first() && hipFuncSetAttribute((const void*)sam_t2i_kv_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return ULL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sam_t2i_kv_scores_kernel, dim3((unsigned)(P / KTB), (unsigned)n), dim3(256), lds, (hipStream_t)stream, (const elem_t*)qproj, (int)T,
                       (const elem_t*)keys, (const elem_t*)pos, (int)P, lw(wk, bk), lw(wv, bv), late_bias_kv, (elem_t*)scores_ws, (elem_t*)vproj_ws);
    if (P == 4096)
        hipLaunchKernelGGL(sam_t2i_softmax_pv_kernel<4>, dim3(NH, (unsigned)n), dim3(1024), 0, (hipStream_t)stream, (const elem_t*)scores_ws,
                           (const elem_t*)vproj_ws, (int)T, (int)P, (elem_t*)att);
    else
        hipLaunchKernelGGL(sam_t2i_softmax_pv_kernel<1>, dim3(NH, (unsigned)n), dim3(1024), 0, (hipStream_t)stream, (const elem_t*)scores_ws,
                           (const elem_t*)vproj_ws, (int)T, (int)P, (elem_t*)att);
    return ull_check_launch();
}

// image -> token cross attention + norm4 (transformer.py:173-180) in ONE launch per 128 image rows; kproj / vproj [n, T, 128] = the
// token-side projections (ull_sam_token_mlp_ln); out [n, P, 256] (must not alias keys).
extern "C" int ULL_FN(ull_sam_i2t_attention_ln_)(const void* keys, const void* pos, const void* kproj, const void* vproj, int64_t n, int64_t T, int64_t P,
                                             const void* wq, const void* bq, const void* wo, const void* bo, int late_bias_q, const void* ln_w,
                                             const void* ln_b, float eps, void* out, void* stream) {
    if (!keys || !pos || !kproj || !vproj || !wq || !bq || !wo || !bo || !ln_w || !ln_b || !out || n <= 0) return ULL_ERR_ARG;
    if (T <= 0 || T > TMAX || P <= 0 || (P % KTB) || keys == out) return ULL_ERR_SHAPE;
    const size_t lds = (size_t)(KTB * PX + 2 * KTB * PI) * sizeof(elem_t) + (size_t)(2 * TMAX * DI) * sizeof(float);
    static UllOncePerDevice once;
    if (once.first() && hipFuncSetAttribute((const void*)sam_i2t_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return ULL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sam_i2t_fused_kernel, dim3((unsigned)(P / KTB), (unsigned)n), dim3(256), lds, (hipStream_t)stream, (const elem_t*)keys,
                       (const elem_t*)pos, (int)P, (const elem_t*)kproj, (const elem_t*)vproj, (int)T, lw(wq, bq), lw(wo, bo), late_bias_q,
                       LnW{(const elem_t*)ln_w, (const elem_t*)ln_b}, eps, (elem_t*)out);
    return ull_check_launch();
}
