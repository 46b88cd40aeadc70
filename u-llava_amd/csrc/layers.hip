// Coarse C-ABI entries: ONE call enqueues a whole stack of transformer layers on the stream (round 6).
//
// Host code only -- no kernel lives here.  Every function below calls the per-op entry points of this library (the same kernels, the same
// dispatch rules as u-llava_amd/ops.py applies call by call), so the results are bit-identical to the per-op path; what changes is the host
// cost: a ctypes round trip + Python argument marshalling per LAUNCH (~15 us, profiles/r05_decode.txt) becomes one per FORWARD.  With eight
// Python ranks sharing one host that is the margin the >= 6x target has.
//
// Reference lines replaced: hf LlamaModel.forward's layer loop (modeling_llama.py:347-419 through models/ullava_core.py:312-322), hf
// CLIPEncoder.forward's layer loop (modeling_clip.py:353-384 through models/ullava_core.py:146-158), ImageEncoderViT.forward's block loop
// (models/segment_anything/modeling/image_encoder.py:110-116, Block.forward :165-193).
#include <stdint.h>
#include <math.h>

#include "../../include/ullava_hip.h"

#ifdef ULL_ELEM_F16
#define FN(base) base##f16
#else
#define FN(base) base##bf16
#endif
#define TRY(call)                 \
    do {                          \
        const int rc_ = (call);   \
        if (rc_ != ULL_OK) return rc_; \
    } while (0)

namespace {

struct SK {                      // the caller's stream-K policy (ops.streamk_policy) and workspace
    void* ws;
    int64_t bytes;
    int64_t min_k;               // < 0: never split
};

// ops.linear for M > 16, K % 64 == 0: the tiled GEMM; 256 x 256 kernel + tile-major weights + stream-K tail when the shape is "big".
inline int lin(const void* x, int64_t ldx, const ull_linear* L, void* out, int64_t ldc, const void* R, int64_t ldr, int64_t M, int flags,
               const SK& sk, void* stream) {
    const bool big = M >= 1024 && L->n >= 512 && L->k >= 128;
    void* ws = nullptr;
    int64_t wsb = 0;
    if (big && sk.min_k >= 0 && L->k >= sk.min_k) { ws = sk.ws; wsb = sk.bytes; }
    if (L->bias) flags |= ULL_EPI_BIAS;
    if (R) flags |= ULL_EPI_RESID;
    if (big && L->w_tiled)
        return FN(ull_gemm_)(x, ldx, L->w_tiled, L->k, out, ldc, L->bias, R, ldr, M, L->n, L->k, flags | ULL_EPI_W_TILED, ws, wsb, stream);
    return FN(ull_gemm_)(x, ldx, L->w, L->ldw, out, ldc, L->bias, R, ldr, M, L->n, L->k, flags, ws, wsb, stream);
}

// ops.linear for M <= 4 (decode steps): the skinny MFMA GEMM from M = 3 on against LLaMA-sized weights, the weight-streaming GEMV otherwise;
// a preceding LlamaRMSNorm is fused into the GEMV where its LDS staging allows it and is a launch of its own otherwise.
inline int lin_decode(const void* x, int64_t ldx, const void* rms_w, float eps, void* xn_scratch, const ull_linear* L, void* out, int64_t ldc,
                      const void* R, int64_t ldr, int64_t M, int flags, void* stream) {
    const bool skinny = M >= 3 && L->k % 32 == 0 && L->n * L->k >= ((int64_t)1 << 22) && L->ldw % 8 == 0;
    if (L->bias) flags |= ULL_EPI_BIAS;
    if (R) flags |= ULL_EPI_RESID;
    if (rms_w && (skinny || !(L->k % 8 == 0 && M * L->k <= 16384))) {
        TRY(FN(ull_rmsnorm_)(x, ldx, rms_w, xn_scratch, L->k, M, L->k, eps, stream));
        x = xn_scratch;
        ldx = L->k;
        rms_w = nullptr;
    }
    if (skinny) return FN(ull_gemm_skinny_)(x, ldx, L->w, L->ldw, out, ldc, L->bias, R, ldr, M, L->n, L->k, flags, stream);
    if (rms_w) return FN(ull_gemv_rmsnorm_)(x, ldx, rms_w, eps, L->w, L->ldw, out, ldc, L->bias, R, ldr, M, L->n, L->k, flags, stream);
    return FN(ull_gemv_)(x, ldx, L->w, L->ldw, out, ldc, L->bias, R, ldr, M, L->n, L->k, flags, stream);
}

}  // namespace

extern "C" int FN(ull_llama_prefill_layers_)(const ull_llama_layer* layers, int64_t n_layers, const void* x_in, void* const* x_out, void* x_mid,
                                             void* xn, void* qkv, void* att, void* act, const void* rope_cos, const void* rope_sin,
                                             const void* key_mask, int64_t B, int64_t S, int64_t H, int64_t hd, int64_t I, float eps, void* ws,
                                             int64_t ws_bytes, int64_t sk_min_k, const void* zeros, void* stream) {
    if (!layers || !x_in || !x_out || !x_mid || !xn || !qkv || !att || !act || !rope_cos || !rope_sin || !zeros || n_layers <= 0) return ULL_ERR_ARG;
    const int64_t D = H * hd, T = B * S;
    if (hd != 128 || T <= 16 || S <= 16 || S > 1024 || D % 64 || I % 64) return ULL_ERR_SHAPE;       // the fused-RoPE prefill form only
    const SK sk{ws, ws_bytes, sk_min_k};
    const float scale = 1.0f / sqrtf((float)hd);
    const char* q = (const char*)qkv;
    const void* x = x_in;
    for (int64_t l = 0; l < n_layers; ++l) {
        const ull_llama_layer& w = layers[l];
        if (w.qkv.n != 3 * D || w.qkv.k != D || w.o.n != D || w.o.k != D || w.gu.n != 2 * I || w.gu.k != D || w.down.n != D || w.down.k != I || !x_out[l])
            return ULL_ERR_ARG;
        TRY(FN(ull_rmsnorm_)(x, D, w.ln1, xn, D, T, D, eps, stream));                                   // input_layernorm
        {                                                                                               // q|k|v projection + RoPE epilogue
            const bool big = T >= 1024 && w.qkv.n >= 512 && w.qkv.k >= 128;
            void* wsp = nullptr;
            int64_t wsb = 0;
            if (big && sk.min_k >= 0 && D >= sk.min_k) { wsp = ws; wsb = ws_bytes; }
            if (big && w.qkv.w_tiled)
                TRY(FN(ull_gemm_qkv_rope_)(xn, D, w.qkv.w_tiled, D, qkv, 3 * D, T, 3 * D, D, rope_cos, rope_sin, 2 * D, hd, ULL_EPI_W_TILED, wsp, wsb, stream));
            else
                TRY(FN(ull_gemm_qkv_rope_)(xn, D, w.qkv.w, w.qkv.ldw, qkv, 3 * D, T, 3 * D, D, rope_cos, rope_sin, 2 * D, hd, 0, wsp, wsb, stream));
        }
        // causal attention, V read as rows of the fused q|k|v buffer (vt_len = 0)
        TRY(FN(ull_attention_)(q, S * 3 * D, hd, 3 * D, q + D * 2, S * 3 * D, hd, 3 * D, q + 2 * D * 2, S * 3 * D, hd, 3 * D, 0, att, S * D, hd, D, key_mask,
                               B, H, S, S, hd, 1, 1, scale, 1.0f, nullptr, nullptr, 0, 0, 0, zeros, stream));
        TRY(lin(att, D, &w.o, x_mid, D, x, D, T, 0, sk, stream));                                       // o_proj + residual
        TRY(FN(ull_rmsnorm_)(x_mid, D, w.ln2, xn, D, T, D, eps, stream));                               // post_attention_layernorm
        TRY(lin(xn, D, &w.gu, act, I, nullptr, 0, T, ULL_EPI_SWIGLU, sk, stream));                      // gate|up + SwiGLU
        TRY(lin(act, I, &w.down, x_out[l], D, x_mid, D, T, 0, sk, stream));                             // down_proj + residual
        x = x_out[l];
    }
    return ULL_OK;
}

extern "C" int FN(ull_llama_decode_layers_)(const ull_llama_layer* layers, int64_t n_layers, const void* x_in, void* const* x_out, void* x_mid,
                                            void* xn, void* q, void* att, void* act, const void* rope_cos, const void* rope_sin,
                                            const void* key_mask, void* const* k_cache, void* const* vt_cache, int64_t B, int64_t S, int64_t H,
                                            int64_t hd, int64_t I, int64_t smax, int64_t past, float eps, const void* zeros, void* stream) {
    if (!layers || !x_in || !x_out || !x_mid || !xn || !q || !att || !act || !rope_cos || !rope_sin || !k_cache || !vt_cache || !zeros || n_layers <= 0)
        return ULL_ERR_ARG;
    const int64_t D = H * hd, T = B * S;
    if (T > 4 || T <= 0 || past <= 0 || (hd & 1) || D % 8 || T * D * 2 > 32768 || past + S > smax) return ULL_ERR_SHAPE;
    const float scale = 1.0f / sqrtf((float)hd);
    const void* x = x_in;
    for (int64_t l = 0; l < n_layers; ++l) {
        const ull_llama_layer& w = layers[l];
        if (w.qkv.n != 3 * D || w.qkv.k != D || !k_cache[l] || !vt_cache[l] || !x_out[l]) return ULL_ERR_ARG;
        TRY(FN(ull_gemv_qkv_rope_append_)(x, D, w.ln1, eps, w.qkv.w, w.qkv.ldw, q, D, rope_cos, rope_sin, k_cache[l], vt_cache[l], B, S, H, hd, D, smax,
                                          past, stream));
        TRY(FN(ull_attention_)(q, S * D, hd, D, k_cache[l], H * smax * hd, smax * hd, hd, vt_cache[l], H * hd * smax, hd * smax, smax, smax, att, S * D, hd, D,
                               key_mask, B, H, S, past + S, hd, 1, 1, scale, 1.0f, nullptr, nullptr, 0, 0, 0, zeros, stream));
        TRY(lin_decode(att, D, nullptr, 0.f, xn, &w.o, x_mid, D, x, D, T, 0, stream));
        TRY(lin_decode(x_mid, D, w.ln2, eps, xn, &w.gu, act, I, nullptr, 0, T, ULL_EPI_SWIGLU, stream));
        TRY(lin_decode(act, I, nullptr, 0.f, xn, &w.down, x_out[l], D, x_mid, D, T, 0, stream));
        x = x_out[l];
    }
    return ULL_OK;
}

extern "C" int FN(ull_clip_layers_)(const ull_clip_layer* layers, int64_t n_layers, void* h, void* h_mid, void* y, void* qkv, void* att, void* f,
                                    int64_t n_img, int64_t S, int64_t H, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes,
                                    int64_t sk_min_k, const void* zeros, void* stream) {
    if (!layers || !h || !h_mid || !y || !qkv || !att || !f || !zeros || n_layers < 0) return ULL_ERR_ARG;
    const int64_t D = H * hd, T = n_img * S;
    if (hd != 64 || T <= 16 || S <= 16 || S > 704 || D % 64 || I % 64) return ULL_ERR_SHAPE;
    const SK sk{ws, ws_bytes, sk_min_k};
    const float scale = 1.0f / sqrtf((float)hd);
    const char* q = (const char*)qkv;
    for (int64_t l = 0; l < n_layers; ++l) {
        const ull_clip_layer& w = layers[l];
        if (w.qkv.n != 3 * D || w.qkv.k != D || w.out.n != D || w.out.k != D || w.fc1.n != I || w.fc1.k != D || w.fc2.n != D || w.fc2.k != I)
            return ULL_ERR_ARG;
        TRY(FN(ull_layernorm_)(h, D, w.ln1_w, w.ln1_b, y, D, T, D, eps, stream));
        TRY(lin(y, D, &w.qkv, qkv, 3 * D, nullptr, 0, T, 0, sk, stream));
        TRY(FN(ull_attention_)(q, S * 3 * D, hd, 3 * D, q + D * 2, S * 3 * D, hd, 3 * D, q + 2 * D * 2, S * 3 * D, hd, 3 * D, 0, att, S * D, hd, D, nullptr,
                               n_img, H, S, S, hd, 0, 1, scale, 1.0f, nullptr, nullptr, 0, 0, 0, zeros, stream));
        TRY(lin(att, D, &w.out, h_mid, D, h, D, T, 0, sk, stream));
        TRY(FN(ull_layernorm_)(h_mid, D, w.ln2_w, w.ln2_b, y, D, T, D, eps, stream));
        TRY(lin(y, D, &w.fc1, f, I, nullptr, 0, T, ULL_EPI_ACT_QUICK_GELU, sk, stream));
        TRY(lin(f, I, &w.fc2, h, D, h_mid, D, T, 0, sk, stream));
    }
    return ULL_OK;
}

extern "C" int FN(ull_sam_blocks_)(const ull_sam_block* blocks, int64_t n_blocks, void* x, void* x_mid, void* y, void* qkv, void* att, void* f,
                                   int64_t B, int64_t g, int64_t nH, int64_t hd, int64_t I, float eps, void* ws, int64_t ws_bytes, int64_t sk_min_k,
                                   const void* zeros, void* stream) {
    if (!blocks || !x || !x_mid || !y || !qkv || !att || !f || !zeros || n_blocks <= 0) return ULL_ERR_ARG;
    const int64_t C = nH * hd, T = B * g * g, S = g * g;
    if (hd != 80 || g != 64 || C % 64 || I % 64) return ULL_ERR_SHAPE;                                  // SAM ViT-H / L / B at 1024 x 1024
    const SK sk{ws, ws_bytes, sk_min_k};
    const float q_scale = 1.0f / sqrtf((float)hd);
    const char* q = (const char*)qkv;
    for (int64_t i = 0; i < n_blocks; ++i) {
        const ull_sam_block& w = blocks[i];
        if (w.qkv.n != 3 * C || w.qkv.k != C || !w.qkv.bias || w.proj.n != C || w.proj.k != C || w.lin1.n != I || w.lin1.k != C || w.lin2.n != C ||
            w.lin2.k != I || !w.rel_pos_h || !w.rel_pos_w || (w.window != 0 && w.window != 14))
            return ULL_ERR_ARG;
        TRY(FN(ull_layernorm_)(x, C, w.n1_w, w.n1_b, y, C, T, C, eps, stream));
        TRY(lin(y, C, &w.qkv, qkv, 3 * C, nullptr, 0, T, 0, sk, stream));
        if (w.window)          // 14 x 14 windows on image-order tokens; the padded positions' q|k|v = the qkv bias
            TRY(FN(ull_sam_window_attention_)(qkv, 3 * C, w.qkv.bias, w.rel_pos_h, w.rel_pos_w, att, C, B, g, g, nH, hd, w.window, q_scale, zeros, stream));
        else                   // global attention over the 64 x 64 grid, rel-pos tables built in the kernel (rel_mode 2), V rows read in place
            TRY(FN(ull_attention_)(q, S * 3 * C, hd, 3 * C, q + C * 2, S * 3 * C, hd, 3 * C, q + 2 * C * 2, S * 3 * C, hd, 3 * C, 0, att, S * C, hd, C,
                                   nullptr, B, nH, S, S, hd, 0, 0, 1.0f, q_scale, w.rel_pos_h, w.rel_pos_w, g, g, 2, zeros, stream));
        TRY(lin(att, C, &w.proj, x_mid, C, x, C, T, 0, sk, stream));
        TRY(FN(ull_layernorm_)(x_mid, C, w.n2_w, w.n2_b, y, C, T, C, eps, stream));
        TRY(lin(y, C, &w.lin1, f, I, nullptr, 0, T, ULL_EPI_ACT_GELU, sk, stream));
        TRY(lin(f, I, &w.lin2, x, C, x_mid, C, T, 0, sk, stream));
    }
    return ULL_OK;
}
