// Row normalisations for gfx950: RMSNorm (LLaMA), LayerNorm (CLIP / SAM), CLIP embedding assembly +
// pre-LayerNorm.  HBM-bound streaming kernels: 16-byte bf16x8 loads, the whole row held in registers
// (read once, written once), fp32 statistics reduced across the sub-wave that owns the row.
//
// reference semantics:
//   RMSNorm  : transformers LlamaRMSNorm.forward  -> w * bf16(x * rsqrt(mean(x^2) + eps))   (two roundings)
//   LayerNorm: torch.nn.LayerNorm on bf16          -> bf16(((x - mean) * rstd) * g + b)       (one rounding)
//   CLIP embed: transformers CLIPVisionEmbeddings.forward + CLIPVisionModel.pre_layrnorm
#include "ull_common.h"

namespace {

// A row of D elements is owned by LPR lanes (power of two, <= 64); lane s of the group loads the
// 8-element chunks s, s+LPR, s+2*LPR, ... (NCH of them).
template <int NCH, int MODE>  // MODE 0 rmsnorm, 1 layernorm
__global__ __launch_bounds__(256) void rownorm_kernel(const elem_t* __restrict__ x, long ldx, const elem_t* __restrict__ w,
                                                      const elem_t* __restrict__ b, elem_t* __restrict__ y, long ldy,
                                                      long rows, int D, float eps, int lpr,
                                                      // optional CLIP-embedding gather (MODE 1 only): row = (img, tok)
                                                      const elem_t* __restrict__ cls, const elem_t* __restrict__ pos, int tokens) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lpr - 1);
    const int rows_per_wave = 64 / lpr;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave + lane / lpr;
    const bool row_ok = row < rows;
    const int nchunk = D >> 3;
    float v[NCH][8];
    const elem_t* xr;
    const elem_t* pr = nullptr;
    if (cls != nullptr) {
        // CLIP: token 0 is the class embedding, token t>0 is patch-embedding row (img*(tokens-1) + t-1);
        // the position embedding is added as a bf16 tensor op (one rounding) before the LayerNorm.
        const long img = row / tokens;
        const int t = (int)(row - img * tokens);
        xr = (t == 0) ? cls : x + (img * (tokens - 1) + (t - 1)) * ldx;
        pr = pos + (long)t * D;
    } else {
        xr = x + row * ldx;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = sub + i * lpr;
        if (row_ok && c < nchunk) {
            unpack8(*(const uint4*)(xr + c * 8), v[i]);
            if (pr != nullptr) {
                float pv[8];
                unpack8(*(const uint4*)(pr + c * 8), pv);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = rnd(v[i][j] + pv[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1 += v[i][j]; s2 += v[i][j] * v[i][j]; }
    }
    const float invD = 1.0f / (float)D;
    float mean = 0.f, rstd;
    if (MODE == 0) {
        s2 = group_sum(s2, lpr);
        rstd = rsqrtf(s2 * invD + eps);
    } else {
        s1 = group_sum(s1, lpr);
        mean = s1 * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = sub + i * lpr;
            if (c < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        }
        q = group_sum(q, lpr);
        rstd = 1.0f / sqrtf(q * invD + eps);
    }
    if (!row_ok) return;
    elem_t* yr = y + row * ldy;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = sub + i * lpr;
        if (c < nchunk) {
            float wv[8], o[8];
            unpack8(*(const uint4*)(w + c * 8), wv);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = wv[j] * rnd(v[i][j] * rstd);
            } else {
                float bv[8];
                unpack8(*(const uint4*)(b + c * 8), bv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = ((v[i][j] - mean) * rstd) * wv[j] + bv[j];
            }
            *(uint4*)(yr + c * 8) = pack8(o);
        }
    }
}

template <int MODE>
int launch_rownorm(const elem_t* x, long ldx, const elem_t* w, const elem_t* b, elem_t* y, long ldy, long rows, int D, float eps,
                   const elem_t* cls, const elem_t* pos, int tokens, hipStream_t st) {
    if (D <= 0 || (D & 7) || (ldx & 7) || (ldy & 7) || rows <= 0) return ULL_ERR_SHAPE;
    const int nchunk = D >> 3;
    int lpr = 1;
    while (lpr < 64 && lpr < nchunk) lpr <<= 1;
    const int nch = (nchunk + lpr - 1) / lpr;
    const long rows_per_block = 4 * (64 / lpr);
    const dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
#define ULL_RN(N) hipLaunchKernelGGL((rownorm_kernel<N, MODE>), grid, dim3(256), 0, st, x, ldx, w, b, y, ldy, rows, D, eps, lpr, cls, pos, tokens)
    if (nch <= 1) ULL_RN(1);
    else if (nch <= 2) ULL_RN(2);
    else if (nch <= 4) ULL_RN(4);
    else if (nch <= 8) ULL_RN(8);
    else if (nch <= 16) ULL_RN(16);
    else return ULL_ERR_SHAPE;   // D > 8192 never occurs on this path
#undef ULL_RN
    return ull_check_launch();
}

}  // namespace

extern "C" int ULL_FN(ull_rmsnorm_)(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t rows, int64_t D, float eps,
                                void* stream) {
    if (!x || !w || !y) return ULL_ERR_ARG;
    return launch_rownorm<0>((const elem_t*)x, ldx, (const elem_t*)w, nullptr, (elem_t*)y, ldy, rows, (int)D, eps, nullptr, nullptr, 0,
                             (hipStream_t)stream);
}

extern "C" int ULL_FN(ull_layernorm_)(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy, int64_t rows, int64_t D,
                                  float eps, void* stream) {
    if (!x || !w || !b || !y) return ULL_ERR_ARG;
    return launch_rownorm<1>((const elem_t*)x, ldx, (const elem_t*)w, (const elem_t*)b, (elem_t*)y, ldy, rows, (int)D, eps, nullptr, nullptr,
                             0, (hipStream_t)stream);
}

// out[(img, t), :] = LayerNorm( (t == 0 ? class_embedding : patch[img, t-1]) + position_embedding[t] )
extern "C" int ULL_FN(ull_clip_embed_ln_)(const void* patch, int64_t ldp, const void* cls, const void* pos, const void* w, const void* b,
                                      void* y, int64_t ldy, int64_t n_img, int64_t tokens, int64_t D, float eps, void* stream) {
    if (!patch || !cls || !pos || !w || !b || !y || tokens < 2) return ULL_ERR_ARG;
    return launch_rownorm<1>((const elem_t*)patch, ldp, (const elem_t*)w, (const elem_t*)b, (elem_t*)y, ldy, n_img * tokens, (int)D, eps,
                             (const elem_t*)cls, (const elem_t*)pos, (int)tokens, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Shifted next-token cross entropy (reference models/ullava_core.py:327-338: CrossEntropyLoss over logits[..., :-1, :] vs
// labels[..., 1:], ignore_index = -100, mean over the counted tokens).  One wave per (b, t) row: fp32 log-sum-exp over the
// bf16 logits row, minus the label logit; block partials are accumulated with one atomic pair per wave.
namespace {
__global__ __launch_bounds__(256) void shifted_ce_kernel(const elem_t* __restrict__ logits, long ld, const int64_t* __restrict__ labels, int B,
                                                         int S, int V, float* __restrict__ out /* [loss_sum, count] */) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);       // over B * (S - 1)
    if (row >= (long)B * (S - 1)) return;
    const int b = (int)(row / (S - 1)), t = (int)(row % (S - 1));
    const int64_t lab = labels[(long)b * S + t + 1];
    if (lab < 0 || lab >= V) return;                                  // ignore_index (-100)
    const elem_t* lp = logits + ((long)b * S + t) * ld;
    float m = -INFINITY;
    for (int i = lane; i < V; i += 64) m = fmaxf(m, e2f(lp[i]));
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += __expf(e2f(lp[i]) - m);
    s = wave_sum(s);
    if (lane == 0) {
        atomicAdd(out, m + __logf(s) - e2f(lp[lab]));
        atomicAdd(out + 1, 1.0f);
    }
}
}  // namespace

// out: float[2] = {sum of per-token losses, number of counted tokens}, must be zeroed by the caller.
extern "C" int ULL_FN(ull_shifted_cross_entropy_)(const void* logits, int64_t ld, const void* labels, int64_t B, int64_t S, int64_t V, void* out,
                                              void* stream) {
    if (!logits || !labels || !out || B <= 0 || S <= 1 || V <= 0) return ULL_ERR_ARG;
    const long rows = B * (S - 1);
    hipLaunchKernelGGL(shifted_ce_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const elem_t*)logits, ld,
                       (const int64_t*)labels, (int)B, (int)S, (int)V, (float*)out);
    return ull_check_launch();
}
