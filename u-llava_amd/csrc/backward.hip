// Backward kernels of the u-LLaVA path for gfx950 (what `train_ullava.py` / `train_ullava_core.py` need beyond the
// forward): RMSNorm, SwiGLU, RoPE, attention, shifted cross-entropy and the embedding splice, plus the column sums of bias
// gradients.  Linear backward needs no kernel of its own: dX = dY W and dW = dY^T X are the forward GEMM on transposed operands.
//
// Reference semantics: torch autograd of the ops cited at each kernel (hf modeling_llama.py LlamaRMSNorm / LlamaMLP /
// apply_rotary_pos_emb / eager_attention_forward, models/ullava_core.py:182-277,327-338).  Gradients are computed in fp32 from the
// 16-bit tensors the forward stored and rounded once to the element type on output (autograd's bf16 backward rounds after every
// elementary op; the fp32 evaluation here is closer to the exact gradient -- tests compare both against an fp32 reference).
// Correctness-first kernels (VALU, one block per row / row group): the training step is not on the benchmarked forward path yet.
#include "ull_common.h"

namespace {

ULL_DEV float block_sum(float v, float* red) {          // 256 threads; red: >= 4 floats of LDS
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
ULL_DEV float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- LlamaRMSNorm backward: y = w * rnd(x * r), r = rsqrt(mean(x^2) + eps) --------------------------------------------------------
// dx = r * (g - xh * mean(g * xh)) with g = dy * w, xh = x * r;  dw[c] += dy[c] * rnd(xh[c]).  dw: fp32 [D], zeroed by the caller.
constexpr int RN_MAXC = 32;                              // columns per thread: D <= 8192
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const elem_t* __restrict__ x, long ldx, const elem_t* __restrict__ w,
                                                          const elem_t* __restrict__ dy, long lddy, elem_t* __restrict__ dx, long lddx,
                                                          float* __restrict__ dw, long rows, int D, float eps) {
    __shared__ float red[4];
    float dwp[RN_MAXC];
#pragma unroll
    for (int i = 0; i < RN_MAXC; ++i) dwp[i] = 0.f;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const elem_t* xr = x + row * ldx;
        const elem_t* gr = dy + row * lddy;
        float s2 = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) { const float v = e2f(xr[c]); s2 += v * v; }
        s2 = block_sum(s2, red);
        const float r = rsqrtf(s2 / (float)D + eps);
        float dot = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) dot += e2f(gr[c]) * e2f(w[c]) * (e2f(xr[c]) * r);
        dot = block_sum(dot, red) / (float)D;
#pragma unroll
        for (int i = 0; i < RN_MAXC; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < D) {
                const float xh = e2f(xr[c]) * r, g = e2f(gr[c]);
                dx[row * lddx + c] = f2e(r * (g * e2f(w[c]) - xh * dot));
                dwp[i] += g * rnd(xh);
            }
        }
    }
    if (dw != nullptr) {
#pragma unroll
        for (int i = 0; i < RN_MAXC; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < D) atomicAdd(dw + c, dwp[i]);
        }
    }
}

// The same for rows of D <= 4096 elements (D % 8 == 0, 16-byte aligned rows): one WAVE per row, the row and its gradient held in
// registers (16-byte loads, one pass over HBM, wave reductions instead of block barriers).  The scalar kernel above read 2 bytes
// per lane in three passes with two block reductions per row: 87 us for the 2584 x 4096 rows of the training step, this one ~15.
template <bool DW>
__global__ __launch_bounds__(256) void rmsnorm_bwd_wave_kernel(const elem_t* __restrict__ x, long ldx, const elem_t* __restrict__ w,
                                                               const elem_t* __restrict__ dy, long lddy, elem_t* __restrict__ dx, long lddx,
                                                               float* __restrict__ dw, long rows, int D, float eps) {
    constexpr int NCH = 8;                                   // 16-byte chunks per lane: 64 * 8 * 8 = 4096 columns
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int nchunk = D >> 3;
    float wv[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) unpack8(*(const uint4*)(w + c * 8), wv[i]);
    }
    float dwp[DW ? NCH : 1][8];
    if constexpr (DW) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) dwp[i][j] = 0.f;
    }
    for (long row = wave; row < rows; row += nwaves) {
        uint4 xq[NCH], gq[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            xq[i] = c < nchunk ? *(const uint4*)(x + row * ldx + c * 8) : make_uint4(0, 0, 0, 0);
            gq[i] = c < nchunk ? *(const uint4*)(dy + row * lddy + c * 8) : make_uint4(0, 0, 0, 0);
        }
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float xv[8];
            unpack8(xq[i], xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) s2 += xv[j] * xv[j];
        }
        const float r = rsqrtf(wave_sum(s2) / (float)D + eps);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (lane + 64 * i < nchunk) {
                float xv[8], gv[8];
                unpack8(xq[i], xv); unpack8(gq[i], gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) dot += gv[j] * wv[i][j] * (xv[j] * r);
            }
        }
        dot = wave_sum(dot) / (float)D;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float xv[8], gv[8], o[8];
                unpack8(xq[i], xv); unpack8(gq[i], gv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = xv[j] * r;
                    o[j] = r * (gv[j] * wv[i][j] - xh * dot);
                    if constexpr (DW) dwp[i][j] += gv[j] * rnd(xh);
                }
                *(uint4*)(dx + row * lddx + c * 8) = pack8(o);
            }
        }
    }
    if constexpr (DW) {
        // the block's four waves meet in the LDS first, then ONE fp32 atomic per column and block reaches dw (the launch keeps the grid at
        // one block per CU for this form): every wave adding its 4096 partial sums to the same 16 KB of global memory serialised 21 M
        // atomics in the L2 -- 1.36 ms per call, 23 % of a full-parameter training step.
        __shared__ float red[4096];
        for (int c = threadIdx.x; c < D; c += 256) red[c] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk)
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(&red[c * 8 + j], dwp[i][j]);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += 256) atomicAdd(dw + c, red[c]);
    }
}

// ---- LlamaMLP activation on the interleaved gate/up layout (groups of 16 gate | 16 up columns, ULL_EPI_SWIGLU's weight order) -----
// forward: a[m, 16g + j] = rnd(rnd(silu(gate)) * up);  backward: d_gate = da * up * silu'(gate), d_up = da * silu(gate).
// (16-byte accesses: a chunk of 8 outputs 16g + 8h .. +7 reads the gate chunk at column 32g + 8h and the up chunk 16 columns on)
// halves: gu = [gate (I columns) | up (I columns)] (the training path's gate_proj / up_proj share one [2I, K] buffer as row slices);
// otherwise the 16-column interleave of the inference pack.
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const elem_t* __restrict__ gu, elem_t* __restrict__ a, long M, int I, int halves) {
    const int cpr = I >> 3;                                  // output chunks per row
    const long total = M * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cpr;
        const int oc = (int)(i % cpr), g = oc >> 1, hh = oc & 1;
        const elem_t* src = gu + m * 2 * I + (halves ? oc * 8 : g * 32 + hh * 8);
        const int up_off = halves ? I : 16;
        float gate[8], up[8], o[8];
        unpack8(*(const uint4*)src, gate);
        unpack8(*(const uint4*)(src + up_off), up);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rnd(act_silu(gate[e])) * up[e];
        *(uint4*)(a + m * I + oc * 8) = pack8(o);
    }
}
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const elem_t* __restrict__ gu, const elem_t* __restrict__ da, elem_t* __restrict__ dgu,
                                                         long M, int I, int halves) {
    const int cpr = I >> 3;
    const long total = M * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cpr;
        const int oc = (int)(i % cpr), g = oc >> 1, hh = oc & 1;
        const long ig = m * 2 * I + (halves ? oc * 8 : g * 32 + hh * 8);
        const int up_off = halves ? I : 16;
        float gate[8], up[8], d[8], og[8], ou[8];
        unpack8(*(const uint4*)(gu + ig), gate);
        unpack8(*(const uint4*)(gu + ig + up_off), up);
        unpack8(*(const uint4*)(da + m * I + oc * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = act_sigmoid(gate[e]);
            og[e] = d[e] * up[e] * (s * (1.0f + gate[e] * (1.0f - s)));
            ou[e] = d[e] * (gate[e] * s);
        }
        *(uint4*)(dgu + ig) = pack8(og);
        *(uint4*)(dgu + ig + up_off) = pack8(ou);
    }
}

// ---- attention backward -----------------------------------------------------------------------------------------------------------
// O = softmax(mult * Q K^T + mask) V per (batch, head); causal: key j visible to query i iff j <= i + (Sk - Sq); key_mask int32 [B, Sk].
// Two kernels without atomics: (1) one block per 8 query rows: scores in LDS, softmax statistics (lse), delta = rowsum(dO * O), dQ;
// (2) one block per 8 keys: recomputes P from lse, accumulates dK / dV over all queries.
struct AttnBwdArgs {
    const elem_t *Q, *K, *V, *O, *dO;
    elem_t *dQ, *dK, *dV;
    long q_bs, q_hs, q_ss, k_bs, k_hs, k_ss, v_bs, v_hs, v_ss, o_bs, o_hs, o_ss, g_bs, g_hs, g_ss;      // element strides (batch, head, seq)
    long dq_bs, dq_hs, dq_ss, dk_bs, dk_hs, dk_ss, dv_bs, dv_hs, dv_ss;
    const int32_t* key_mask;
    float* lse; float* delta;              // [B, H, Sq] scratch
    int B, H, Sq, Sk, hd, causal;
    float mult;                            // S = mult * Q K^T
};
constexpr int AB_R = 8;                    // query rows (kernel 1) / keys (kernel 2) per block

ULL_DEV bool attn_visible(const AttnBwdArgs& p, int b, int i, int j) {
    if (p.causal && j > i + (p.Sk - p.Sq)) return false;
    return p.key_mask == nullptr || p.key_mask[(long)b * p.Sk + j] != 0;
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = (float*)smem;                            // [AB_R][Sk]
    float* qs = sc + (long)AB_R * p.Sk;                  // [AB_R][hd]
    float* gs = qs + AB_R * p.hd;                        // [AB_R][hd]  dO rows
    float* st = gs + AB_R * p.hd;                        // [AB_R][4]: max, sum, delta, -
    __shared__ float red[4];
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int i0 = blockIdx.x * AB_R;
    const int nr = min(AB_R, p.Sq - i0);
    const int hd = p.hd, Sk = p.Sk, tid = threadIdx.x;
    for (int e = tid; e < AB_R * hd; e += 256) {
        const int r = e / hd, d = e % hd;
        float q = 0.f, g = 0.f;
        if (r < nr) {
            q = e2f(p.Q[b * p.q_bs + h * p.q_hs + (long)(i0 + r) * p.q_ss + d]);
            g = e2f(p.dO[b * p.g_bs + h * p.g_hs + (long)(i0 + r) * p.g_ss + d]);
        }
        qs[e] = q; gs[e] = g;
    }
    __syncthreads();
    for (int r = 0; r < AB_R; ++r) {                     // delta_r = sum_d dO * O
        float v = 0.f;
        if (r < nr)
            for (int d = tid; d < hd; d += 256) v += gs[r * hd + d] * e2f(p.O[b * p.o_bs + h * p.o_hs + (long)(i0 + r) * p.o_ss + d]);
        v = block_sum(v, red);
        if (tid == 0) st[r * 4 + 2] = v;
    }
    // scores
    for (int j = tid; j < Sk; j += 256) {
        const elem_t* kp = p.K + b * p.k_bs + h * p.k_hs + (long)j * p.k_ss;
        float acc[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) acc[r] = 0.f;
        for (int d = 0; d < hd; ++d) {
            const float kv = e2f(kp[d]);
#pragma unroll
            for (int r = 0; r < AB_R; ++r) acc[r] += qs[r * hd + d] * kv;
        }
#pragma unroll
        for (int r = 0; r < AB_R; ++r) sc[(long)r * Sk + j] = (r < nr && attn_visible(p, b, i0 + r, j)) ? acc[r] * p.mult : -INFINITY;
    }
    __syncthreads();
    for (int r = 0; r < AB_R; ++r) {
        float m = -INFINITY;
        for (int j = tid; j < Sk; j += 256) m = fmaxf(m, sc[(long)r * Sk + j]);
        m = block_max(m, red);
        float l = 0.f;
        for (int j = tid; j < Sk; j += 256) {
            const float s = sc[(long)r * Sk + j];
            const float e = (s == -INFINITY) ? 0.f : __expf(s - m);
            sc[(long)r * Sk + j] = e;
            l += e;
        }
        l = block_sum(l, red);
        if (tid == 0) { st[r * 4] = m; st[r * 4 + 1] = l; }
        __syncthreads();
    }
    if (tid < nr) {
        p.lse[(long)bh * p.Sq + i0 + tid] = st[tid * 4] + __logf(st[tid * 4 + 1]);
        p.delta[(long)bh * p.Sq + i0 + tid] = st[tid * 4 + 2];
    }
    // dS = P * (dP - delta) * mult, dP = dO V^T
    for (int j = tid; j < Sk; j += 256) {
        const elem_t* vp = p.V + b * p.v_bs + h * p.v_hs + (long)j * p.v_ss;
        float acc[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) acc[r] = 0.f;
        for (int d = 0; d < hd; ++d) {
            const float vv = e2f(vp[d]);
#pragma unroll
            for (int r = 0; r < AB_R; ++r) acc[r] += gs[r * hd + d] * vv;
        }
#pragma unroll
        for (int r = 0; r < AB_R; ++r) {
            const float l = st[r * 4 + 1];
            const float pr = l > 0.f ? sc[(long)r * Sk + j] / l : 0.f;
            sc[(long)r * Sk + j] = pr * (acc[r] - st[r * 4 + 2]) * p.mult;
        }
    }
    __syncthreads();
    // dQ[r][d] = sum_j dS[r][j] K[j][d]
    for (int e = tid; e < AB_R * hd; e += 256) {
        const int r = e / hd, d = e % hd;
        if (r >= nr) continue;
        const elem_t* kp = p.K + b * p.k_bs + h * p.k_hs + d;
        float acc = 0.f;
        for (int j = 0; j < Sk; ++j) acc += sc[(long)r * Sk + j] * e2f(kp[(long)j * p.k_ss]);
        p.dQ[b * p.dq_bs + h * p.dq_hs + (long)(i0 + r) * p.dq_ss + d] = f2e(acc);
    }
}

constexpr int AB_QC = 32;                                // queries staged per iteration of kernel 2
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int hd = p.hd;
    float* ks = (float*)smem;                            // [AB_R][hd]
    float* vs = ks + AB_R * hd;                          // [AB_R][hd]
    float* qs = vs + AB_R * hd;                          // [AB_QC][hd]
    float* gs = qs + AB_QC * hd;                         // [AB_QC][hd]
    float* pb = gs + AB_QC * hd;                         // [AB_QC][AB_R]  P
    float* db = pb + AB_QC * AB_R;                       // [AB_QC][AB_R]  dS
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int j0 = blockIdx.x * AB_R;
    const int nk = min(AB_R, p.Sk - j0);
    const int tid = threadIdx.x;
    for (int e = tid; e < AB_R * hd; e += 256) {
        const int r = e / hd, d = e % hd;
        ks[e] = r < nk ? e2f(p.K[b * p.k_bs + h * p.k_hs + (long)(j0 + r) * p.k_ss + d]) : 0.f;
        vs[e] = r < nk ? e2f(p.V[b * p.v_bs + h * p.v_hs + (long)(j0 + r) * p.v_ss + d]) : 0.f;
    }
    // each thread owns up to 4 (key, dim) outputs of dK and dV
    float dk[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
    const int nout = AB_R * hd;                          // <= 1024
    for (int i0 = 0; i0 < p.Sq; i0 += AB_QC) {
        const int nq = min(AB_QC, p.Sq - i0);
        __syncthreads();
        for (int e = tid; e < AB_QC * hd; e += 256) {
            const int r = e / hd, d = e % hd;
            qs[e] = r < nq ? e2f(p.Q[b * p.q_bs + h * p.q_hs + (long)(i0 + r) * p.q_ss + d]) : 0.f;
            gs[e] = r < nq ? e2f(p.dO[b * p.g_bs + h * p.g_hs + (long)(i0 + r) * p.g_ss + d]) : 0.f;
        }
        __syncthreads();
        {   // one (query, key) pair per thread
            const int qi = tid / AB_R, kj = tid % AB_R;
            float pr = 0.f, ds = 0.f;
            if (qi < nq && kj < nk && attn_visible(p, b, i0 + qi, j0 + kj)) {
                float s = 0.f, dp = 0.f;
                for (int d = 0; d < hd; ++d) { s += qs[qi * hd + d] * ks[kj * hd + d]; dp += gs[qi * hd + d] * vs[kj * hd + d]; }
                pr = __expf(s * p.mult - p.lse[(long)bh * p.Sq + i0 + qi]);
                ds = pr * (dp - p.delta[(long)bh * p.Sq + i0 + qi]) * p.mult;
            }
            pb[qi * AB_R + kj] = pr;
            db[qi * AB_R + kj] = ds;
        }
        __syncthreads();
        for (int o = 0; o < 4; ++o) {
            const int e = tid + o * 256;
            if (e >= nout) break;
            const int kj = e / hd, d = e % hd;
            float a = 0.f, c = 0.f;
            for (int qi = 0; qi < nq; ++qi) { a += db[qi * AB_R + kj] * qs[qi * hd + d]; c += pb[qi * AB_R + kj] * gs[qi * hd + d]; }
            dk[o] += a; dv[o] += c;
        }
    }
    for (int o = 0; o < 4; ++o) {
        const int e = tid + o * 256;
        if (e >= nout) break;
        const int kj = e / hd, d = e % hd;
        if (kj >= nk) continue;
        p.dK[b * p.dk_bs + h * p.dk_hs + (long)(j0 + kj) * p.dk_ss + d] = f2e(dk[o]);
        p.dV[b * p.dv_bs + h * p.dv_hs + (long)(j0 + kj) * p.dv_ss + d] = f2e(dv[o]);
    }
}

// ---- the same backward on the matrix cores (head dim 64 / 128: the LLaMA block) ------------------------------------------------------
// Mirrors the forward kernel's "swapped" products: an MFMA of A = key rows and B = query rows leaves, in lane (fr, fg), the scores of
// query fr against keys 4fg + r of a 16-key block -- and eight of those values, packed, ARE the B operand of the next product over the
// keys of a 32-key block, provided the other operand comes with its keys permuted the same way (slot 8g + 4a + r <- key 16a + 4g + r:
// the layout transpose_v_kernel writes).  So:
//   dQ kernel  (block = 64 queries, wave = 16): per 64-key tile  S = K Q^T, dP = V dO^T, dS = P o (dP - delta) mult,
//              dQ^T += Kt_perm dS        (Kt = transpose_v(K));  first a pass over the keys for the row statistics (lse), delta = dO . O
//   dKV kernel (block = 64 keys, wave = 16):    per 64-query tile S^T = Q K^T, dP^T = dO V^T (A = query rows, B = key rows),
//              dV^T += dOt_perm P^T, dK^T += Qt_perm dS^T        (Qt / dOt = transpose_v(Q / dO))
// Operand fragments come straight from global memory (a (batch, head) is 82-164 KB per tensor: L2-resident); no atomics, fp32 softmax
// recomputation exactly like the scalar kernels above, P and dS rounded to the element type where they enter an MFMA (the reference's
// 16-bit autograd rounds them at the same place).
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(AttnBwdArgs p, const elem_t* __restrict__ Kt, int pitch, int sqp) {
    constexpr int NKS = HD / 32, NDB = HD / 16;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int q = blockIdx.x * 64 + wave * 16 + fr;
    const int qc = min(q, p.Sq - 1);
    const int shift = p.Sk - p.Sq;
    uint4 qf[NKS], gf[NKS];
    float delta = 0.f;
    {
        const elem_t* qp = p.Q + b * p.q_bs + h * p.q_hs + (long)qc * p.q_ss;
        const elem_t* gp = p.dO + b * p.g_bs + h * p.g_hs + (long)qc * p.g_ss;
        const elem_t* op = p.O + b * p.o_bs + h * p.o_hs + (long)qc * p.o_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = *(const uint4*)(qp + ks * 32 + fg * 8);
            gf[ks] = *(const uint4*)(gp + ks * 32 + fg * 8);
            float g8[8], o8[8];
            unpack8(gf[ks], g8);
            unpack8(*(const uint4*)(op + ks * 32 + fg * 8), o8);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += g8[e] * o8[e];
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
    }
    // keys this block can see at all (block-uniform)
    const int kend = p.causal ? min(p.Sk, blockIdx.x * 64 + 63 + shift + 1) : p.Sk;
    const int nkt = (kend + 63) / 64;
    const elem_t* kbase = p.K + b * p.k_bs + h * p.k_hs;
    const elem_t* vbase = p.V + b * p.v_bs + h * p.v_hs;
    const int32_t* km = p.key_mask ? p.key_mask + (long)b * p.Sk : nullptr;
    auto visible = [&](int key) { return key < p.Sk && (!p.causal || key <= q + shift) && (km == nullptr || km[key] != 0); };
    // ---- pass 1: log-sum-exp of the row ------------------------------------------------------------------------------------------
    float m = -INFINITY, l = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        float sv[16];
        float mt = -INFINITY;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int kr = min(kt * 64 + cb * 16 + fr, p.Sk - 1);
            const elem_t* kp = kbase + (long)kr * p.k_ss;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) acc = mfma16(*(const uint4*)(kp + ks * 32 + fg * 8), qf[ks], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = visible(kt * 64 + cb * 16 + fg * 4 + r) ? acc[r] * p.mult : -INFINITY;
                sv[cb * 4 + r] = sc;
                mt = fmaxf(mt, sc);
            }
        }
        const float mn = fmaxf(m, mt);
        if (mn > -INFINITY) {
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) add += __expf(sv[e] - mn);
            l = l * __expf(m - mn) + add;
            m = mn;
        }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, m2);
        if (mn > -INFINITY) l = l * __expf(m - mn) + l2 * __expf(m2 - mn);
        m = mn;
    }
    const float lse = l > 0.f ? m + __logf(l) : INFINITY;          // a fully masked row: every probability is exp(s - inf) = 0
    if (fg == 0 && q < p.Sq) {
        p.lse[(long)bh * sqp + q] = lse;
        p.delta[(long)bh * sqp + q] = delta;
    }
    // ---- pass 2: dS and dQ^T ---------------------------------------------------------------------------------------------------------
    f32x4_t dq[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) dq[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const elem_t* ktb = Kt + (long)bh * HD * pitch;
    for (int kt = 0; kt < nkt; ++kt) {
        uint32_t dsp[8];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int kr = min(kt * 64 + cb * 16 + fr, p.Sk - 1);
            const elem_t* kp = kbase + (long)kr * p.k_ss;
            const elem_t* vp = vbase + (long)kr * p.v_ss;
            f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                sa = mfma16(*(const uint4*)(kp + ks * 32 + fg * 8), qf[ks], sa);
                da = mfma16(*(const uint4*)(vp + ks * 32 + fg * 8), gf[ks], da);
            }
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = visible(kt * 64 + cb * 16 + fg * 4 + r) ? __expf(sa[r] * p.mult - lse) : 0.f;
                ds[r] = pr * (da[r] - delta) * p.mult;
            }
            dsp[cb * 2] = pack2e(ds[0], ds[1]);
            dsp[cb * 2 + 1] = pack2e(ds[2], ds[3]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint4 bf = make_uint4(dsp[4 * a], dsp[4 * a + 1], dsp[4 * a + 2], dsp[4 * a + 3]);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
                dq[db] = mfma16(*(const uint4*)(ktb + (long)(db * 16 + fr) * pitch + kt * 64 + a * 32 + fg * 8), bf, dq[db]);
        }
    }
    // dq[db][r] = dQ^T[d = db*16 + 4fg + r][query fr]
    if (q < p.Sq) {
        elem_t* dst = p.dQ + b * p.dq_bs + h * p.dq_hs + (long)q * p.dq_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            uint2 o;
            o.x = pack2e(dq[db][0], dq[db][1]);
            o.y = pack2e(dq[db][2], dq[db][3]);
            *(uint2*)(dst + db * 16 + fg * 4) = o;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(AttnBwdArgs p, const elem_t* __restrict__ Qt, const elem_t* __restrict__ dOt,
                                                                int pitch, int sqp) {
    constexpr int NKS = HD / 32, NDB = HD / 16;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int j = blockIdx.x * 64 + wave * 16 + fr;           // this lane's key (B-operand row)
    const int jc = min(j, p.Sk - 1);
    const int shift = p.Sk - p.Sq;
    uint4 kf[NKS], vf[NKS];
    {
        const elem_t* kp = p.K + b * p.k_bs + h * p.k_hs + (long)jc * p.k_ss;
        const elem_t* vp = p.V + b * p.v_bs + h * p.v_hs + (long)jc * p.v_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *(const uint4*)(kp + ks * 32 + fg * 8);
            vf[ks] = *(const uint4*)(vp + ks * 32 + fg * 8);
        }
    }
    const bool key_ok = j < p.Sk && (p.key_mask == nullptr || p.key_mask[(long)b * p.Sk + j] != 0);
    const int qbeg = p.causal ? max(0, blockIdx.x * 64 - shift) : 0;        // first query that can see a key of this block
    const int nqt = (p.Sq + 63) / 64;
    const elem_t* qbase = p.Q + b * p.q_bs + h * p.q_hs;
    const elem_t* gbase = p.dO + b * p.g_bs + h * p.g_hs;
    const elem_t* qtb = Qt + (long)bh * HD * pitch;
    const elem_t* gtb = dOt + (long)bh * HD * pitch;
    const float* lsep = p.lse + (long)bh * sqp;
    const float* delp = p.delta + (long)bh * sqp;
    f32x4_t dk[NDB], dv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) { dk[db] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int qt = qbeg / 64; qt < nqt; ++qt) {
        uint32_t pp[8], dsp[8];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const int ir = min(qt * 64 + qb * 16 + fr, p.Sq - 1);        // A-operand row of this lane: a query
            const elem_t* qp = qbase + (long)ir * p.q_ss;
            const elem_t* gp = gbase + (long)ir * p.g_ss;
            f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                sa = mfma16(*(const uint4*)(qp + ks * 32 + fg * 8), kf[ks], sa);
                da = mfma16(*(const uint4*)(gp + ks * 32 + fg * 8), vf[ks], da);
            }
            // sa[r] = S[query i = qt*64 + qb*16 + 4fg + r][key j]
            const int i0 = qt * 64 + qb * 16 + fg * 4;
            const f32x4_t l4 = *(const f32x4_t*)(lsep + i0), d4 = *(const f32x4_t*)(delp + i0);
            float pr[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + r;
                const bool vis = key_ok && i < p.Sq && (!p.causal || j <= i + shift);
                pr[r] = vis ? __expf(sa[r] * p.mult - l4[r]) : 0.f;
                ds[r] = vis ? pr[r] * (da[r] - d4[r]) * p.mult : 0.f;
            }
            pp[qb * 2] = pack2e(pr[0], pr[1]); pp[qb * 2 + 1] = pack2e(pr[2], pr[3]);
            dsp[qb * 2] = pack2e(ds[0], ds[1]); dsp[qb * 2 + 1] = pack2e(ds[2], ds[3]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint4 bp = make_uint4(pp[4 * a], pp[4 * a + 1], pp[4 * a + 2], pp[4 * a + 3]);
            const uint4 bd = make_uint4(dsp[4 * a], dsp[4 * a + 1], dsp[4 * a + 2], dsp[4 * a + 3]);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const long off = (long)(db * 16 + fr) * pitch + qt * 64 + a * 32 + fg * 8;
                dv[db] = mfma16(*(const uint4*)(gtb + off), bp, dv[db]);
                dk[db] = mfma16(*(const uint4*)(qtb + off), bd, dk[db]);
            }
        }
    }
    if (j < p.Sk) {
        elem_t* dkp = p.dK + b * p.dk_bs + h * p.dk_hs + (long)j * p.dk_ss;
        elem_t* dvp = p.dV + b * p.dv_bs + h * p.dv_hs + (long)j * p.dv_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            uint2 o;
            o.x = pack2e(dk[db][0], dk[db][1]); o.y = pack2e(dk[db][2], dk[db][3]);
            *(uint2*)(dkp + db * 16 + fg * 4) = o;
            o.x = pack2e(dv[db][0], dv[db][1]); o.y = pack2e(dv[db][2], dv[db][3]);
            *(uint2*)(dvp + db * 16 + fg * 4) = o;
        }
    }
}

// ---- the same two kernels for hd = 128 with the streamed operand tiles shared through the LDS -------------------------------------
// The direct form above gives every wave its own 16-byte fragment loads from global memory: four waves of a block fetch the same Q / dO
// (or K / V) rows and their transposed images, one vector-memory instruction per MFMA -- the texture path, not the matrix pipe, is
// the bound (64 loads against 64 MFMAs per wave and tile: ~4x the MFMA time per CU).  Here a block DMAs each 64-row tile once
// ([64 rows][256 B], double-buffered), all four waves read their fragments from it, and the transposed operands of the dV / dK / dQ
// products come out of the SAME row-major tile through ds_read_b64_tr_b16 -- no Q^T / K^T / dO^T images at all.
// Chunk position inside a row: 32-byte pairs XOR (row & 7), the chunk inside a pair XOR bit 3 of the row: conflict-free for the
// 16-row ds_read_b128 of the "row" operands and for the 4-rows-by-32-bytes pattern of the transposing read.
ULL_DEV void bw_glds16(const void* gsrc, uint32_t lds_byte_addr /* wave-uniform */) {
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
typedef uint32_t bw_u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
ULL_DEV bw_u32x2 bw_tr(uint32_t addr) {      // lane i of a 16-lane group: address of row i / 4, 4 elements at column 4 * (i % 4); gets column i
    bw_u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
ULL_DEV void bw_tr_wait(bw_u32x2 (&a)[4], bw_u32x2 (&b)[4]) {     // the compiler does not know the reads above are LDS loads
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) :: "memory");
}
constexpr int BW_TILE = 64 * 256;            // one [64 rows][128 x 16-bit] tile
ULL_DEV int bw_pos(int c, int r) { return (((c >> 1) ^ (r & 7)) << 1) | ((c & 1) ^ ((r >> 3) & 1)); }      // (its own inverse in c)
// this wave's quarter (4 of 16 one-KiB pieces) of tile rows first .. first + 63 (clamped to last) of a [rows][hd = 128] operand
ULL_DEV void bw_stage(const elem_t* base, long stride, int first, int last, uint32_t dst, int wave, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = wave * 4 + k;
        const int row = i * 4 + (lane >> 4);
        const int c = bw_pos(lane & 15, row);
        bw_glds16(base + (long)min(first + row, last) * stride + c * 8, dst + i * 1024);
    }
}
// "row" fragment (A operand, rows = tile rows, k = 8 head dims): row blk * 16 + fr, dims ks * 32 + fg * 8 .. +7
ULL_DEV uint4 bw_rowfrag(const char* tile, int blk, int ks, int fr, int fg) {
    const int row = blk * 16 + fr;
    return *(const uint4*)(tile + row * 256 + bw_pos(ks * 4 + fg, row) * 16);
}
// per-lane base address for the transposed fragments of a tile: head dim 16 * db + fr of rows half * 32 + 4 * fg + {0..3} (+ 16)
ULL_DEV uint32_t bw_tr_base(uint32_t tile, int fr, int fg) {
    return tile + (4 * fg + (fr >> 2)) * 256 + ((((fr & 3) >> 1) ^ (fg >> 1)) << 4) + ((fr & 1) << 3);
}
// acc[d0 + j] += (tile^T fragment of head-dim block d0 + j, rows half * 32 ..) x bfrag, j < 4
template <int HALF>
ULL_DEV void bw_tr_mma4(uint32_t trb, int swr, int d0, const uint4& bfrag, f32x4_t* acc) {
    bw_u32x2 va[4], vc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t ad = trb + (((d0 + j) ^ swr) << 5);
        va[j] = bw_tr<HALF * 32 * 256>(ad);
        vc[j] = bw_tr<HALF * 32 * 256 + 16 * 256>(ad);
    }
    bw_tr_wait(va, vc);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[d0 + j] = mfma16(make_uint4(va[j].x, va[j].y, vc[j].x, vc[j].y), bfrag, acc[d0 + j]);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_tiles_kernel(AttnBwdArgs p, int sqp) {
    extern __shared__ __attribute__((aligned(16))) char bsm[];                  // 2 x [K tile | V tile]
    constexpr int NKS = 4, NDB = 8;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, fg = lane >> 4;
    const int q = blockIdx.x * 64 + wave * 16 + fr;
    const int qc = min(q, p.Sq - 1);
    const int shift = p.Sk - p.Sq;
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)bsm);
    uint4 qf[NKS], gf[NKS];
    float delta = 0.f;
    {
        const elem_t* qp = p.Q + b * p.q_bs + h * p.q_hs + (long)qc * p.q_ss;
        const elem_t* gp = p.dO + b * p.g_bs + h * p.g_hs + (long)qc * p.g_ss;
        const elem_t* op = p.O + b * p.o_bs + h * p.o_hs + (long)qc * p.o_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            qf[ks] = *(const uint4*)(qp + ks * 32 + fg * 8);
            gf[ks] = *(const uint4*)(gp + ks * 32 + fg * 8);
            float g8[8], o8[8];
            unpack8(gf[ks], g8);
            unpack8(*(const uint4*)(op + ks * 32 + fg * 8), o8);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += g8[e] * o8[e];
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
    }
    const int kend = p.causal ? min(p.Sk, blockIdx.x * 64 + 63 + shift + 1) : p.Sk;
    const int nkt = (kend + 63) / 64;
    const elem_t* kbase = p.K + b * p.k_bs + h * p.k_hs;
    const elem_t* vbase = p.V + b * p.v_bs + h * p.v_hs;
    const int32_t* km = p.key_mask ? p.key_mask + (long)b * p.Sk : nullptr;
    auto visible = [&](int key) { return key < p.Sk && (!p.causal || key <= q + shift) && (km == nullptr || km[key] != 0); };
    // ---- pass 1: log-sum-exp of the row (K tiles only) ---------------------------------------------------------------------------
    float m = -INFINITY, l = 0.f;
    bw_stage(kbase, p.k_ss, 0, p.Sk - 1, lds, wave, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nkt) bw_stage(kbase, p.k_ss, (kt + 1) * 64, p.Sk - 1, lds + ((kt + 1) & 1) * 2 * BW_TILE, wave, lane);
        const char* tk = bsm + (kt & 1) * 2 * BW_TILE;
        float sv[16];
        float mt = -INFINITY;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) acc = mfma16(bw_rowfrag(tk, cb, ks, fr, fg), qf[ks], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = visible(kt * 64 + cb * 16 + fg * 4 + r) ? acc[r] * p.mult : -INFINITY;
                sv[cb * 4 + r] = sc;
                mt = fmaxf(mt, sc);
            }
        }
        const float mn = fmaxf(m, mt);
        if (mn > -INFINITY) {
            float add = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) add += __expf(sv[e] - mn);
            l = l * __expf(m - mn) + add;
            m = mn;
        }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, m2);
        if (mn > -INFINITY) l = l * __expf(m - mn) + l2 * __expf(m2 - mn);
        m = mn;
    }
    const float lse = l > 0.f ? m + __logf(l) : INFINITY;
    if (fg == 0 && q < p.Sq) {
        p.lse[(long)bh * sqp + q] = lse;
        p.delta[(long)bh * sqp + q] = delta;
    }
    // ---- pass 2: dS and dQ^T (K and V tiles) ------------------------------------------------------------------------------------------
    f32x4_t dq[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) dq[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int swr = 4 * (fg & 1) + (fr >> 2);
    __builtin_amdgcn_s_barrier();                                   // pass 1's last tile has been read by every wave
    bw_stage(kbase, p.k_ss, 0, p.Sk - 1, lds, wave, lane);
    bw_stage(vbase, p.v_ss, 0, p.Sk - 1, lds + BW_TILE, wave, lane);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nkt) {
            const uint32_t nb = lds + ((kt + 1) & 1) * 2 * BW_TILE;
            bw_stage(kbase, p.k_ss, (kt + 1) * 64, p.Sk - 1, nb, wave, lane);
            bw_stage(vbase, p.v_ss, (kt + 1) * 64, p.Sk - 1, nb + BW_TILE, wave, lane);
        }
        const char* tk = bsm + (kt & 1) * 2 * BW_TILE;
        const char* tv = tk + BW_TILE;
        uint32_t dsp[8];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                sa = mfma16(bw_rowfrag(tk, cb, ks, fr, fg), qf[ks], sa);
                da = mfma16(bw_rowfrag(tv, cb, ks, fr, fg), gf[ks], da);
            }
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = visible(kt * 64 + cb * 16 + fg * 4 + r) ? __expf(sa[r] * p.mult - lse) : 0.f;
                ds[r] = pr * (da[r] - delta) * p.mult;
            }
            dsp[cb * 2] = pack2e(ds[0], ds[1]);
            dsp[cb * 2 + 1] = pack2e(ds[2], ds[3]);
        }
        const uint32_t trk = bw_tr_base(lds + (kt & 1) * 2 * BW_TILE, fr, fg);
        {
            const uint4 bf = make_uint4(dsp[0], dsp[1], dsp[2], dsp[3]);
            bw_tr_mma4<0>(trk, swr, 0, bf, dq);
            bw_tr_mma4<0>(trk, swr, 4, bf, dq);
        }
        {
            const uint4 bf = make_uint4(dsp[4], dsp[5], dsp[6], dsp[7]);
            bw_tr_mma4<1>(trk, swr, 0, bf, dq);
            bw_tr_mma4<1>(trk, swr, 4, bf, dq);
        }
    }
    if (q < p.Sq) {
        elem_t* dst = p.dQ + b * p.dq_bs + h * p.dq_hs + (long)q * p.dq_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            uint2 o;
            o.x = pack2e(dq[db][0], dq[db][1]);
            o.y = pack2e(dq[db][2], dq[db][3]);
            *(uint2*)(dst + db * 16 + fg * 4) = o;
        }
    }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_tiles_kernel(AttnBwdArgs p, int sqp) {
    extern __shared__ __attribute__((aligned(16))) char bsm[];                  // 2 x [Q tile | dO tile]
    constexpr int NKS = 4, NDB = 8;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, fg = lane >> 4;
    const int j = blockIdx.x * 64 + wave * 16 + fr;           // this lane's key (B-operand row)
    const int jc = min(j, p.Sk - 1);
    const int shift = p.Sk - p.Sq;
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)bsm);
    uint4 kf[NKS], vf[NKS];
    {
        const elem_t* kp = p.K + b * p.k_bs + h * p.k_hs + (long)jc * p.k_ss;
        const elem_t* vp = p.V + b * p.v_bs + h * p.v_hs + (long)jc * p.v_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            kf[ks] = *(const uint4*)(kp + ks * 32 + fg * 8);
            vf[ks] = *(const uint4*)(vp + ks * 32 + fg * 8);
        }
    }
    const bool key_ok = j < p.Sk && (p.key_mask == nullptr || p.key_mask[(long)b * p.Sk + j] != 0);
    const int qbeg = p.causal ? max(0, blockIdx.x * 64 - shift) : 0;        // first query that can see a key of this block
    const int nqt = (p.Sq + 63) / 64, qt0 = qbeg / 64;
    const elem_t* qbase = p.Q + b * p.q_bs + h * p.q_hs;
    const elem_t* gbase = p.dO + b * p.g_bs + h * p.g_hs;
    const float* lsep = p.lse + (long)bh * sqp;
    const float* delp = p.delta + (long)bh * sqp;
    f32x4_t dk[NDB], dv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) { dk[db] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const int swr = 4 * (fg & 1) + (fr >> 2);
    if (qt0 < nqt) {
        bw_stage(qbase, p.q_ss, qt0 * 64, p.Sq - 1, lds, wave, lane);
        bw_stage(gbase, p.g_ss, qt0 * 64, p.Sq - 1, lds + BW_TILE, wave, lane);
    }
    for (int qt = qt0; qt < nqt; ++qt) {
        const int buf = (qt - qt0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (qt + 1 < nqt) {
            const uint32_t nb = lds + (buf ^ 1) * 2 * BW_TILE;
            bw_stage(qbase, p.q_ss, (qt + 1) * 64, p.Sq - 1, nb, wave, lane);
            bw_stage(gbase, p.g_ss, (qt + 1) * 64, p.Sq - 1, nb + BW_TILE, wave, lane);
        }
        const char* tq = bsm + buf * 2 * BW_TILE;
        const char* tg = tq + BW_TILE;
        uint32_t pp[8], dsp[8];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                sa = mfma16(bw_rowfrag(tq, qb, ks, fr, fg), kf[ks], sa);
                da = mfma16(bw_rowfrag(tg, qb, ks, fr, fg), vf[ks], da);
            }
            // sa[r] = S[query i = qt*64 + qb*16 + 4fg + r][key j]
            const int i0 = qt * 64 + qb * 16 + fg * 4;
            const f32x4_t l4 = *(const f32x4_t*)(lsep + i0), d4 = *(const f32x4_t*)(delp + i0);
            float pr[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + r;
                const bool vis = key_ok && i < p.Sq && (!p.causal || j <= i + shift);
                pr[r] = vis ? __expf(sa[r] * p.mult - l4[r]) : 0.f;
                ds[r] = vis ? pr[r] * (da[r] - d4[r]) * p.mult : 0.f;
            }
            pp[qb * 2] = pack2e(pr[0], pr[1]); pp[qb * 2 + 1] = pack2e(pr[2], pr[3]);
            dsp[qb * 2] = pack2e(ds[0], ds[1]); dsp[qb * 2 + 1] = pack2e(ds[2], ds[3]);
        }
        const uint32_t trq = bw_tr_base(lds + buf * 2 * BW_TILE, fr, fg), trg = trq + BW_TILE;
        {
            const uint4 bp = make_uint4(pp[0], pp[1], pp[2], pp[3]), bd = make_uint4(dsp[0], dsp[1], dsp[2], dsp[3]);
            bw_tr_mma4<0>(trg, swr, 0, bp, dv); bw_tr_mma4<0>(trg, swr, 4, bp, dv);
            bw_tr_mma4<0>(trq, swr, 0, bd, dk); bw_tr_mma4<0>(trq, swr, 4, bd, dk);
        }
        {
            const uint4 bp = make_uint4(pp[4], pp[5], pp[6], pp[7]), bd = make_uint4(dsp[4], dsp[5], dsp[6], dsp[7]);
            bw_tr_mma4<1>(trg, swr, 0, bp, dv); bw_tr_mma4<1>(trg, swr, 4, bp, dv);
            bw_tr_mma4<1>(trq, swr, 0, bd, dk); bw_tr_mma4<1>(trq, swr, 4, bd, dk);
        }
    }
    if (j < p.Sk) {
        elem_t* dkp = p.dK + b * p.dk_bs + h * p.dk_hs + (long)j * p.dk_ss;
        elem_t* dvp = p.dV + b * p.dv_bs + h * p.dv_hs + (long)j * p.dv_ss;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            uint2 o;
            o.x = pack2e(dk[db][0], dk[db][1]); o.y = pack2e(dk[db][2], dk[db][3]);
            *(uint2*)(dkp + db * 16 + fg * 4) = o;
            o.x = pack2e(dv[db][0], dv[db][1]); o.y = pack2e(dv[db][2], dv[db][3]);
            *(uint2*)(dvp + db * 16 + fg * 4) = o;
        }
    }
}

// ---- models/ullava_core.py:327-338 backward: d/dlogits of mean CE(logits[:, :-1], labels[:, 1:]) ----------------------------------
// dlogits[b, s, :] = (softmax(logits[b, s]) - onehot(labels[b, s + 1])) * g / count for counted positions, 0 elsewhere.
// stats float[2] = {sum of token losses, counted tokens} from the forward; gout = pointer to the upstream gradient (one float).
__global__ __launch_bounds__(256) void ce_bwd_kernel(const elem_t* __restrict__ logits, long ld, const int64_t* __restrict__ labels, int S, int V,
                                                     const float* __restrict__ stats, const float* __restrict__ gout, elem_t* __restrict__ dl) {
    __shared__ float red[4];
    const long row = blockIdx.x;                         // (b, s)
    const int s = (int)(row % S);
    const elem_t* lr = logits + row * ld;
    elem_t* dr = dl + row * ld;
    const long lab = (s + 1 < S) ? labels[row + 1] : -100;
    if (lab < 0 || lab >= V) {
        for (int c = threadIdx.x; c < V; c += 256) dr[c] = f2e(0.f);
        return;
    }
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, e2f(lr[c]));
    m = block_max(m, red);
    float l = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) l += __expf(e2f(lr[c]) - m);
    l = block_sum(l, red);
    const float sc = gout[0] / stats[1];
    for (int c = threadIdx.x; c < V; c += 256) {
        const float pr = __expf(e2f(lr[c]) - m) / l;
        dr[c] = f2e((pr - (c == lab ? 1.f : 0.f)) * sc);
    }
}

// ---- models/ullava_core.py:191,230-269 backward of the embedding lookup + visual-token splice --------------------------------------
// rows that came from the token table add into d_table (fp32 [vocab, D], atomics: ids repeat); rows of the spliced span are the
// gradient of the projected visual features (each written once) and NEVER reach the table: the reference's torch.cat drops the
// placeholder rows of the looked-up embeddings (:243-245), whether or not the projector asks for a gradient.
// detach_text (projector_from_scratch, :230-240 / :255-264): in a sample that carries an image / video only the start-token row
// and the end-token row (position start + tokens + 1) keep their gradient, every other text row is .detach()ed; a text-only sample
// (:213-220) keeps all of its rows.
__global__ __launch_bounds__(256) void embed_splice_bwd_kernel(const int64_t* __restrict__ ids, const elem_t* __restrict__ demb, float* __restrict__ d_table,
                                                               elem_t* __restrict__ d_img, int n_img_tok, int img_pitch, int img_off,
                                                               elem_t* __restrict__ d_vid, int n_vid_tok, const int32_t* __restrict__ spans, int S,
                                                               int D, long vocab, int detach_text) {
    const long row = blockIdx.x;
    const int b = (int)(row / S), s = (int)(row % S);
    const elem_t* g = demb + row * D;
    if (spans != nullptr) {
        const int kind = spans[b * 4], pos = spans[b * 4 + 1], idx = spans[b * 4 + 2];
        const int ntok = kind == 1 ? n_img_tok : (kind == 2 ? n_vid_tok : 0);
        if (ntok > 0 && s > pos && s <= pos + ntok) {
            elem_t* base = kind == 1 ? d_img : d_vid;
            if (base != nullptr) {
                elem_t* o = kind == 1 ? base + ((long)idx * img_pitch + img_off + (s - pos - 1)) * D : base + ((long)idx * n_vid_tok + (s - pos - 1)) * D;
                for (int c = threadIdx.x; c < D; c += 256) o[c] = g[c];
            }
            return;
        }
        if (detach_text && ntok > 0 && s != pos && s != pos + ntok + 1) return;
    }
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    if (d_table != nullptr)
        for (int c = threadIdx.x; c < D; c += 256) atomicAdd(d_table + id * D + c, e2f(g[c]));
}

// out[c] = sum_r x[r, c]  (bias gradients), fp32, one block per 64 columns
__global__ __launch_bounds__(256) void colsum_kernel(const elem_t* __restrict__ x, long ld, long rows, int N, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < N)
        for (long r = w; r < rows; r += 4) acc += e2f(x[r * ld + c]);
    part[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && c < N) out[c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// out[i] = rnd(scale * sum_r x[r, i]) over R contiguous slabs of n elements, fp32 accumulation: the local reduction of the
// direct-exchange reduce-scatter of the gradient buckets (every rank receives its shard from every peer, dist.py).
__global__ __launch_bounds__(256) void sum_slabs_kernel(const elem_t* __restrict__ x, elem_t* __restrict__ out, int R, long n, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc += e2f(x[(long)r * n + i]);
        out[i] = f2e(acc * scale);
    }
}

// ---- torch.nn.LayerNorm backward (SAM TwoWayAttentionBlock.norm1-4, norm_final_attn): y = ((x - mean) * rstd) * w + b ---------------
// dx = rstd * (g - mean(g) - xh * mean(g * xh)), g = dy * w;  dw += dy * xh, db += dy (fp32 [D], zeroed by the caller).
constexpr int LN_MAXC = 8;                               // columns per thread: D <= 2048
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const elem_t* __restrict__ x, long ldx, const elem_t* __restrict__ w,
                                                            const elem_t* __restrict__ dy, long lddy, elem_t* __restrict__ dx, long lddx,
                                                            float* __restrict__ dw, float* __restrict__ db, long rows, int D, float eps) {
    __shared__ float red[4];
    float dwp[LN_MAXC], dbp[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) dwp[i] = dbp[i] = 0.f;
    const float invD = 1.0f / (float)D;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const elem_t* xr = x + row * ldx;
        const elem_t* gr = dy + row * lddy;
        float s1 = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) s1 += e2f(xr[c]);
        const float mean = block_sum(s1, red) * invD;
        float s2 = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) { const float d = e2f(xr[c]) - mean; s2 += d * d; }
        const float rstd = 1.0f / sqrtf(block_sum(s2, red) * invD + eps);
        float a = 0.f, bsum = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) {
            const float g = e2f(gr[c]) * e2f(w[c]);
            a += g;
            bsum += g * (e2f(xr[c]) - mean) * rstd;
        }
        a = block_sum(a, red) * invD;
        bsum = block_sum(bsum, red) * invD;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < D) {
                const float xh = (e2f(xr[c]) - mean) * rstd, g0 = e2f(gr[c]);
                dx[row * lddx + c] = f2e(rstd * (g0 * e2f(w[c]) - a - xh * bsum));
                dwp[i] += g0 * xh;
                dbp[i] += g0;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < D) {
            if (dw != nullptr) atomicAdd(dw + c, dwp[i]);
            if (db != nullptr) atomicAdd(db + c, dbp[i]);
        }
    }
}

ULL_DEV float gelu_grad(float x) {                       // d/dx [0.5 x (1 + erf(x / sqrt 2))]
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// ---- common.py:31-43 LayerNorm2d (+ the nn.GELU that follows it in output_upscaling) backward on channels-last rows [rows, C] --------
// one wave per row, C <= 512; the forward's LN output is recomputed in fp32.
__global__ __launch_bounds__(256) void layernorm2d_cl_bwd_kernel(const elem_t* __restrict__ x, const elem_t* __restrict__ w, const elem_t* __restrict__ b,
                                                                 const elem_t* __restrict__ dy, elem_t* __restrict__ dx, float* __restrict__ dw,
                                                                 float* __restrict__ db, long rows, int C, float eps, int gelu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dwp[8], dbp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dwp[i] = dbp[i] = 0.f;
    const float invC = 1.0f / (float)C;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const elem_t* xr = x + row * C;
        float s1 = 0.f;
        for (int c = lane; c < C; c += 64) s1 += e2f(xr[c]);
        const float mean = wave_sum(s1) * invC;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = e2f(xr[c]) - mean; s2 += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(s2) * invC + eps);
        float gv[8], xh[8], a = 0.f, bs = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + i * 64;
            gv[i] = xh[i] = 0.f;
            if (c < C) {
                xh[i] = (e2f(xr[c]) - mean) * rstd;
                float g0 = e2f(dy[row * C + c]);
                if (gelu) g0 *= gelu_grad(e2f(w[c]) * xh[i] + e2f(b[c]));
                gv[i] = g0;
                a += g0 * e2f(w[c]);
                bs += g0 * e2f(w[c]) * xh[i];
            }
        }
        a = wave_sum(a) * invC;
        bs = wave_sum(bs) * invC;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                dx[row * C + c] = f2e(rstd * (gv[i] * e2f(w[c]) - a - xh[i] * bs));
                dwp[i] += gv[i] * xh[i];
                dbp[i] += gv[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + i * 64;
        if (c < C) {
            if (dw != nullptr) atomicAdd(dw + c, dwp[i]);
            if (db != nullptr) atomicAdd(db + c, dbp[i]);
        }
    }
}

// ---- torch.nn.GELU (erf) as its own op on the training path (the inference path fuses it into the GEMM epilogue) ------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const elem_t* __restrict__ x, elem_t* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = f2e(act_gelu_erf(e2f(x[i])));
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const elem_t* __restrict__ x, const elem_t* __restrict__ dy, elem_t* __restrict__ dx, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = f2e(e2f(dy[i]) * gelu_grad(e2f(x[i])));
}

// ---- mask_decoder.py:150-158 backward of masks = hyper_in @ upscaled (blocked `up` layout of mask_matmul_kernel) ------------------------
// dup[i][c] = sum_t dm[n, t, Y, X] * hyper[n, t, c];  dhyper[n, t, c] += sum_pixels dm * up  (fp32, zeroed by the caller).
__global__ __launch_bounds__(256) void mask_matmul_bwd_kernel(const elem_t* __restrict__ hyper, const elem_t* __restrict__ up, const elem_t* __restrict__ dm,
                                                              float* __restrict__ dhyper, elem_t* __restrict__ dup, int T, int Cc, int G, long per_n) {
    __shared__ float acc[8 * 32];                            // [T <= 8][Cc <= 32]
    const long n = blockIdx.y;
    for (int e = threadIdx.x; e < T * Cc; e += 256) acc[e] = 0.f;
    __syncthreads();
    const int HW = 4 * G;
    for (long il = (long)blockIdx.x * 256 + threadIdx.x; il < per_n; il += (long)gridDim.x * 256) {
        const long i = n * per_n + il;
        const int d2 = (int)(il & 3), d1 = (int)((il >> 2) & 3);
        const long cell = il >> 4;
        const int xx = (int)(cell % G), y = (int)(cell / G);
        const int Y = 4 * y + 2 * (d1 >> 1) + (d2 >> 1), X = 4 * xx + 2 * (d1 & 1) + (d2 & 1);
        float g[8];
        for (int t = 0; t < T; ++t) g[t] = e2f(dm[((n * T + t) * HW + Y) * (long)HW + X]);
        for (int c = 0; c < Cc; ++c) {
            const float u = e2f(up[i * Cc + c]);
            float d = 0.f;
            for (int t = 0; t < T; ++t) {
                d += g[t] * e2f(hyper[(n * T + t) * Cc + c]);
                if (g[t] != 0.f) atomicAdd(&acc[t * Cc + c], g[t] * u);
            }
            dup[i * Cc + c] = f2e(d);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * Cc; e += 256) atomicAdd(dhyper + n * T * Cc + e, acc[e]);
}

inline unsigned nblk(long total, long cap = 16384) {
    long b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int ULL_FN(ull_rmsnorm_bwd_)(const void* x, int64_t ldx, const void* w, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw,
                                    int64_t rows, int64_t D, float eps, void* stream) {
    if (!x || !w || !dy || !dx || rows <= 0 || D <= 0) return ULL_ERR_ARG;
    if (D > 256 * RN_MAXC) return ULL_ERR_SHAPE;
    if (D <= 4096 && (D & 7) == 0 && (ldx & 7) == 0 && (lddy & 7) == 0 && (lddx & 7) == 0 && ((uintptr_t)w & 15) == 0) {
        const unsigned wb = (unsigned)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
        if (dw)
            hipLaunchKernelGGL(rmsnorm_bwd_wave_kernel<true>, dim3(wb < 512 ? wb : 512), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, ldx, (const elem_t*)w,
                               (const elem_t*)dy, lddy, (elem_t*)dx, lddx, (float*)dw, (long)rows, (int)D, eps);
        else
            hipLaunchKernelGGL(rmsnorm_bwd_wave_kernel<false>, dim3(wb), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, ldx, (const elem_t*)w,
                               (const elem_t*)dy, lddy, (elem_t*)dx, lddx, (float*)dw, (long)rows, (int)D, eps);
        return ull_check_launch();
    }
    const unsigned blocks = (unsigned)(rows < 1024 ? rows : 1024);
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, ldx, (const elem_t*)w,
                       (const elem_t*)dy, lddy, (elem_t*)dx, lddx, (float*)dw, (long)rows, (int)D, eps);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_swiglu_fwd_)(const void* gu, void* a, int64_t M, int64_t I, int halves, void* stream) {
    if (!gu || !a || M <= 0 || I <= 0) return ULL_ERR_ARG;
    if (I & 15) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(nblk(M * I)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)gu, (elem_t*)a, (long)M, (int)I, halves);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_swiglu_bwd_)(const void* gu, const void* da, void* dgu, int64_t M, int64_t I, int halves, void* stream) {
    if (!gu || !da || !dgu || M <= 0 || I <= 0) return ULL_ERR_ARG;
    if (I & 15) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(nblk(M * I)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)gu, (const elem_t*)da, (elem_t*)dgu,
                       (long)M, (int)I, halves);
    return ull_check_launch();
}

// ReLU backward as a selection: dx = y > 0 ? dy : 0 (y = the Linear's activated output).  n % 8 == 0 elements are taken 8 at a time.
namespace {
__global__ __launch_bounds__(256) void relu_mask_kernel(const elem_t* __restrict__ y, const elem_t* __restrict__ dy, elem_t* __restrict__ dx, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long)gridDim.x * 256 * 8) {
        if (i + 8 <= n) {
            float a[8], d[8];
            unpack8(*(const uint4*)(y + i), a);
            const uint4 dv = *(const uint4*)(dy + i);
            const uint16_t* dh = (const uint16_t*)&dv;
            uint4 o;
            uint16_t* oh = (uint16_t*)&o;
#pragma unroll
            for (int e = 0; e < 8; ++e) oh[e] = a[e] > 0.f ? dh[e] : (uint16_t)0;
            *(uint4*)(dx + i) = o;
        } else {
            for (long j = i; j < n; ++j) dx[j] = e2f(y[j]) > 0.f ? dy[j] : (elem_t)0;
        }
    }
}
}  // namespace
extern "C" int ULL_FN(ull_relu_mask_)(const void* y, const void* dy, void* dx, int64_t n, void* stream) {
    if (!y || !dy || !dx || n <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(nblk(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)y, (const elem_t*)dy, (elem_t*)dx, (long)n);
    return ull_check_launch();
}

// strides: 5 x (batch, head, seq) for Q, K, V, O, dO then 3 x for dQ, dK, dV = int64[24]; scratch: float [2, B, H, Sq].
extern "C" int ULL_FN(ull_attention_bwd_)(const void* Q, const void* K, const void* V, const void* O, const void* dO, void* dQ, void* dK, void* dV,
                                      const int64_t* strides, const void* key_mask, int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd,
                                      int causal, float mult, void* scratch, void* stream) {
    if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || !strides || !scratch || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return ULL_ERR_ARG;
    if (hd <= 0 || hd > 128) return ULL_ERR_SHAPE;
    AttnBwdArgs p;
    p.Q = (const elem_t*)Q; p.K = (const elem_t*)K; p.V = (const elem_t*)V; p.O = (const elem_t*)O; p.dO = (const elem_t*)dO;
    p.dQ = (elem_t*)dQ; p.dK = (elem_t*)dK; p.dV = (elem_t*)dV;
    const int64_t* s = strides;
    p.q_bs = s[0]; p.q_hs = s[1]; p.q_ss = s[2]; p.k_bs = s[3]; p.k_hs = s[4]; p.k_ss = s[5]; p.v_bs = s[6]; p.v_hs = s[7]; p.v_ss = s[8];
    p.o_bs = s[9]; p.o_hs = s[10]; p.o_ss = s[11]; p.g_bs = s[12]; p.g_hs = s[13]; p.g_ss = s[14];
    p.dq_bs = s[15]; p.dq_hs = s[16]; p.dq_ss = s[17]; p.dk_bs = s[18]; p.dk_hs = s[19]; p.dk_ss = s[20]; p.dv_bs = s[21]; p.dv_hs = s[22]; p.dv_ss = s[23];
    p.key_mask = (const int32_t*)key_mask;
    p.lse = (float*)scratch; p.delta = (float*)scratch + B * H * Sq;
    p.B = (int)B; p.H = (int)H; p.Sq = (int)Sq; p.Sk = (int)Sk; p.hd = (int)hd; p.causal = causal; p.mult = mult;
    const size_t lds1 = ((size_t)AB_R * Sk + 2 * AB_R * hd + AB_R * 4) * sizeof(float);
    constexpr int MAX_DYN_LDS = 160 * 1024 - 256;        // the kernels also hold a few static LDS words
    if (lds1 > MAX_DYN_LDS) return ULL_ERR_LDS;
    static UllOncePerDevice once;
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_DYN_LDS) != hipSuccess) {
            (void)hipGetLastError();
            return ULL_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)((Sq + AB_R - 1) / AB_R), (unsigned)(B * H)), dim3(256), lds1, (hipStream_t)stream, p);
    const size_t lds2 = ((size_t)2 * AB_R * hd + 2 * AB_QC * hd + 2 * AB_QC * AB_R) * sizeof(float);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)((Sk + AB_R - 1) / AB_R), (unsigned)(B * H)), dim3(256), lds2, (hipStream_t)stream, p);
    return ull_check_launch();
}

// MFMA form (head dim 64 or 128, hd contiguous, 8-byte aligned rows).  Qt / Kt / dOt: transpose_v images [B, H, hd, pitch] of Q, K and dO
// (keys permuted inside 32-blocks, zero-filled behind the sequence); scratch: float [2, B, H, ceil64(Sq)].
extern "C" int ULL_FN(ull_attention_bwd_mfma_)(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* Qt, const void* Kt,
                                           const void* dOt, int64_t pitch, void* dQ, void* dK, void* dV, const int64_t* strides, const void* key_mask,
                                           int64_t B, int64_t H, int64_t Sq, int64_t Sk, int64_t hd, int causal, float mult, void* scratch, void* stream) {
    if (!Q || !K || !V || !O || !dO || !dQ || !dK || !dV || !strides || !scratch || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return ULL_ERR_ARG;
    if (hd != 64 && hd != 128) return ULL_ERR_SHAPE;
    const bool tiles = hd == 128;              // LDS-tile kernels: transposed operands through the transposing LDS read, no images needed
    if (!tiles && (!Qt || !Kt || !dOt)) return ULL_ERR_ARG;
    const int64_t need = ((Sq > Sk ? Sq : Sk) + 63) / 64 * 64;
    if (!tiles && (pitch < need || (pitch & 7))) return ULL_ERR_SHAPE;
    for (int i = 0; i < 24; ++i)
        if (strides[i] & 3) return ULL_ERR_SHAPE;              // 8-byte fragment stores / 16-byte loads need aligned rows
    AttnBwdArgs p;
    p.Q = (const elem_t*)Q; p.K = (const elem_t*)K; p.V = (const elem_t*)V; p.O = (const elem_t*)O; p.dO = (const elem_t*)dO;
    p.dQ = (elem_t*)dQ; p.dK = (elem_t*)dK; p.dV = (elem_t*)dV;
    const int64_t* s = strides;
    p.q_bs = s[0]; p.q_hs = s[1]; p.q_ss = s[2]; p.k_bs = s[3]; p.k_hs = s[4]; p.k_ss = s[5]; p.v_bs = s[6]; p.v_hs = s[7]; p.v_ss = s[8];
    p.o_bs = s[9]; p.o_hs = s[10]; p.o_ss = s[11]; p.g_bs = s[12]; p.g_hs = s[13]; p.g_ss = s[14];
    p.dq_bs = s[15]; p.dq_hs = s[16]; p.dq_ss = s[17]; p.dk_bs = s[18]; p.dk_hs = s[19]; p.dk_ss = s[20]; p.dv_bs = s[21]; p.dv_hs = s[22]; p.dv_ss = s[23];
    for (int i = 2; i < 24; i += 3)
        if (s[i] & 7) return ULL_ERR_SHAPE;                    // token strides: 16-byte fragment loads
    p.key_mask = (const int32_t*)key_mask;
    const int sqp = (int)((Sq + 63) / 64 * 64);
    p.lse = (float*)scratch; p.delta = (float*)scratch + B * H * sqp;
    p.B = (int)B; p.H = (int)H; p.Sq = (int)Sq; p.Sk = (int)Sk; p.hd = (int)hd; p.causal = causal; p.mult = mult;
    const dim3 gq((unsigned)((Sq + 63) / 64), (unsigned)(B * H)), gk((unsigned)((Sk + 63) / 64), (unsigned)(B * H));
    if (tiles) {
        for (int i = 1; i < 15; i += 3)
            if (s[i] & 7) return ULL_ERR_SHAPE;                // head strides of Q, K, V, O, dO: the DMA copies 16-byte chunks
        hipLaunchKernelGGL(attn_bwd_dq_tiles_kernel, gq, dim3(256), 4 * BW_TILE, (hipStream_t)stream, p, sqp);
        hipLaunchKernelGGL(attn_bwd_dkv_tiles_kernel, gk, dim3(256), 4 * BW_TILE, (hipStream_t)stream, p, sqp);
    } else if (hd == 128) {
        hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<128>, gq, dim3(256), 0, (hipStream_t)stream, p, (const elem_t*)Kt, (int)pitch, sqp);
        hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<128>, gk, dim3(256), 0, (hipStream_t)stream, p, (const elem_t*)Qt, (const elem_t*)dOt, (int)pitch, sqp);
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<64>, gq, dim3(256), 0, (hipStream_t)stream, p, (const elem_t*)Kt, (int)pitch, sqp);
        hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<64>, gk, dim3(256), 0, (hipStream_t)stream, p, (const elem_t*)Qt, (const elem_t*)dOt, (int)pitch, sqp);
    }
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_shifted_cross_entropy_bwd_)(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V,
                                                  const void* stats, const void* gout, void* dlogits, void* stream) {
    if (!logits || !labels || !stats || !gout || !dlogits || B <= 0 || S <= 0 || V <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)(B * S)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)logits, ld, (const int64_t*)labels,
                       (int)S, (int)V, (const float*)stats, (const float*)gout, (elem_t*)dlogits);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_embed_splice_bwd_)(const void* ids, const void* demb, void* d_table, void* d_img, int64_t n_img_tok, int64_t img_pitch,
                                         int64_t img_off, void* d_vid, int64_t n_vid_tok, const void* spans, int64_t B, int64_t S, int64_t D,
                                         int64_t vocab, int detach_text, void* stream) {
    if (!ids || !demb || B <= 0 || S <= 0 || D <= 0 || vocab <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(embed_splice_bwd_kernel, dim3((unsigned)(B * S)), dim3(256), 0, (hipStream_t)stream, (const int64_t*)ids, (const elem_t*)demb,
                       (float*)d_table, (elem_t*)d_img, (int)n_img_tok, (int)img_pitch, (int)img_off, (elem_t*)d_vid, (int)n_vid_tok,
                       (const int32_t*)spans, (int)S, (int)D, (long)vocab, detach_text);
    return ull_check_launch();
}

// out[c][r] = x[r][c]: 64 x 64 tiles through LDS, 16-byte accesses on both sides when the shape allows (the transposed operands of the
// Linear backward: dW = dY^T X wants dY^T and X^T row-major, dX = dY W wants W^T; a strided elementwise copy ran at 0.1-0.5 TB/s).
namespace {
__global__ __launch_bounds__(256) void transpose2d_kernel(const elem_t* __restrict__ x, long ldx, elem_t* __restrict__ y, long ldy, int R, int C) {
    __shared__ __attribute__((aligned(16))) elem_t t[64][72];              // [c][r]; 144-byte rows keep the 16-byte reads aligned
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const bool fast = ((ldx | ldy) & 7) == 0 && r0 + 64 <= R && c0 + 64 <= C &&
                      (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    if (fast) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int r = i >> 3, cc = (i & 7) * 8;
            const uint4 v = *(const uint4*)(x + (long)(r0 + r) * ldx + c0 + cc);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[cc + 2 * j][r] = (elem_t)(w[j] & 0xffff);
                t[cc + 2 * j + 1][r] = (elem_t)(w[j] >> 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int c = i >> 3, rr = (i & 7) * 8;
            *(uint4*)(y + (long)(c0 + c) * ldy + r0 + rr) = *(const uint4*)&t[c][rr];
        }
    } else {
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            if (r0 + r < R && c0 + c < C) t[c][r] = x[(long)(r0 + r) * ldx + c0 + c];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int c = i >> 6, r = i & 63;
            if (r0 + r < R && c0 + c < C) y[(long)(c0 + c) * ldy + r0 + r] = t[c][r];
        }
    }
}
}  // namespace

extern "C" int ULL_FN(ull_transpose2d_)(const void* x, int64_t ldx, void* y, int64_t ldy, int64_t R, int64_t C, void* stream) {
    if (!x || !y || R <= 0 || C <= 0 || ldx < C || ldy < R) return ULL_ERR_ARG;
    if (R > (1 << 30) || C > (1 << 30)) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const elem_t*)x, (long)ldx, (elem_t*)y, (long)ldy, (int)R, (int)C);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_colsum_)(const void* x, int64_t ld, int64_t rows, int64_t N, void* out, void* stream) {
    if (!x || !out || rows <= 0 || N <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, ld, (long)rows, (int)N,
                       (float*)out);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_sum_slabs_)(const void* x, void* out, int64_t R, int64_t n, float scale, void* stream) {
    if (!x || !out || R <= 0 || n <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (elem_t*)out, (int)R, (long)n, scale);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_layernorm_bwd_)(const void* x, int64_t ldx, const void* w, const void* dy, int64_t lddy, void* dx, int64_t lddx, void* dw,
                                      void* db, int64_t rows, int64_t D, float eps, void* stream) {
    if (!x || !w || !dy || !dx || rows <= 0 || D <= 0) return ULL_ERR_ARG;
    if (D > 256 * LN_MAXC) return ULL_ERR_SHAPE;
    const unsigned blocks = (unsigned)(rows < 2048 ? rows : 2048);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, ldx, (const elem_t*)w,
                       (const elem_t*)dy, lddy, (elem_t*)dx, lddx, (float*)dw, (float*)db, (long)rows, (int)D, eps);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_layernorm2d_cl_bwd_)(const void* x, const void* w, const void* b, const void* dy, void* dx, void* dw, void* db, int64_t rows,
                                           int64_t C, float eps, int gelu, void* stream) {
    if (!x || !w || !b || !dy || !dx || rows <= 0 || C <= 0) return ULL_ERR_ARG;
    if (C > 512) return ULL_ERR_SHAPE;
    const long want = (rows + 3) / 4;
    hipLaunchKernelGGL(layernorm2d_cl_bwd_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x,
                       (const elem_t*)w, (const elem_t*)b, (const elem_t*)dy, (elem_t*)dx, (float*)dw, (float*)db, (long)rows, (int)C, eps, gelu);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_gelu_fwd_)(const void* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (elem_t*)y, (long)n);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_gelu_bwd_)(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
    if (!x || !dy || !dx || n <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (const elem_t*)dy, (elem_t*)dx, (long)n);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_mask_matmul_bwd_)(const void* hyper, const void* up, const void* dmasks, void* dhyper, void* dup, int64_t n, int64_t T,
                                        int64_t C, int64_t G, void* stream) {
    if (!hyper || !up || !dmasks || !dhyper || !dup || n <= 0 || G <= 0) return ULL_ERR_ARG;
    if (T > 8 || C > 32 || T <= 0 || C <= 0) return ULL_ERR_SHAPE;
    const long per_n = G * G * 16;
    hipLaunchKernelGGL(mask_matmul_bwd_kernel, dim3((unsigned)((per_n + 255) / 256 < 256 ? (per_n + 255) / 256 : 256), (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, (const elem_t*)hyper, (const elem_t*)up, (const elem_t*)dmasks, (float*)dhyper, (elem_t*)dup, (int)T, (int)C,
                       (int)G, per_n);
    return ull_check_launch();
}
