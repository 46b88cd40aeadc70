// Decode-shape Linear for gfx950: C[M,N] = epilogue(X[M,K] * W[N,K]^T) with M <= 4 (token-by-token generation).
//
// reference: the same nn.Linear modules as gemm_bf16.hip, reached from `generate()` steps after the prefill
// (models/ullava_core.py:357-395 prepare_inputs_for_generation keeps only the last token when a KV cache exists).
//
// At M <= 4 the op is a pure weight stream (LLaMA-7B: 13.5 GB per token), so there is no LDS staging and no MFMA: one wave
// owns one output feature at a time, its 64 lanes stream that weight row with 16-byte loads (1 KiB per wave-instruction,
// read exactly once), X comes from L1/L2, fp32 accumulation, wave reduction, same epilogues/rounding points as the GEMM.
#include "ull_common.h"

namespace {

constexpr int EPI_BIAS = 1, EPI_ACT_SHIFT = 1, EPI_ACT_MASK = 3 << 1, EPI_RESID = 8, EPI_SWIGLU = 16, EPI_OUT_F32 = 32;
constexpr int EPI_BIAS_ROUNDED = 256;     // bias added to the already rounded product (at::linear's unfused matmul + add_ path)
constexpr int EPI_ROPE_APPEND = 1 << 20;  // (internal) q|k|v projection of a decode step: RoPE + KV-cache append in the epilogue, see RopeAppend
constexpr int MAXM = 4;

// Decode-step q|k|v projection (hf LlamaAttention.forward: q/k/v_proj, apply_rotary_pos_emb, cache update) in ONE launch: a wave owns the
// output pair (i, i + hd/2) of one head -- the two elements a rotation mixes -- so after the dot products
//   q: both rotated elements go to the query buffer;  k: to row `past + s` of the K cache;  v (no rotation): to its V^T cache columns,
// with exactly the operations of rope_append_kernel on the rounded projection outputs (same bits), from the cos / sin table of the
// step's positions (rope_table: one launch per step instead of 32 x 16 lanes evaluating sinf / cosf per layer).
struct RopeAppend {
    const elem_t* cs; const elem_t* sn;      // [tokens][hd / 2], 16-bit rounded
    elem_t* kc; elem_t* vtc;                 // K cache [B, H, smax, hd]; V^T cache [B, H, hd, smax] (32-key permutation of transpose_v_kernel)
    int S, H, hd, smax, past;
};

typedef uint32_t gv_u32x4_t __attribute__((ext_vector_type(4)));
// The weight stream is read exactly once per token, by exactly one CU: non-temporal loads (global_load_dwordx4 ... nt; MI355X_MICROARCH.md
// "nt-weights") keep it from evicting X and the KV cache from the L2.  Measured (tools/gemv_bw.py, tools/decode_bench.py): 1-GB stream
// 5.82 -> 6.29 TB/s, decode step 3.77 -> 3.61 ms.  (Software-pipelining the batches of a wave across output rows was also tried: 168
// registers, 3 waves per SIMD instead of 4, 3.85 TB/s on the q|k|v shape against 4.9 -- not shipped.  Requesting the wave's first weight
// batch before the activations are staged -- the weights do not depend on them -- was tried in two forms in the decode loop: held in the
// real buffers across the staging code (151 registers, the fourth wave per SIMD gone: 3.79 vs 3.38 ms per token) and as loads into one
// scratch register quad that only warm the L2 (3.40 vs 3.37): neither shipped; the launch's first memory round trip is not what a
// 20-us GEMV loses against the 1-GB stream rate.)
ULL_DEV uint4 w_load16(const elem_t* p) {
    const gv_u32x4_t v = __builtin_nontemporal_load((const gv_u32x4_t*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

constexpr int XS_MAX_BYTES = 32 * 1024;     // X (optionally RMS-normalised) is staged in LDS when M * K * 2 fits in this

// X staged in LDS (`staged`): every block first copies -- or, with norm_w, RMS-normalises (transformers LlamaRMSNorm:
// w * bf16(x * rsqrt(mean(x^2) + eps)), the op that precedes the q/k/v and gate/up projections) -- the M activation rows,
// which removes one tiny latency-bound kernel per projection from the decode step.  The weight stream keeps U 16-byte loads
// per lane in flight (U KiB per wave).
template <int M, int U>
__global__ __launch_bounds__(256) void gemv_kernel(const elem_t* __restrict__ X, long ldx, const elem_t* __restrict__ W, long ldw, void* C,
                                                   long ldc, const elem_t* __restrict__ bias, const elem_t* __restrict__ R, long ldr, int N, int K,
                                                   int flags, int n_out, const elem_t* __restrict__ norm_w, float eps, int staged, RopeAppend ra) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem_t* xs = (elem_t*)smem;                                   // [M][K] when staged
    __shared__ float red[4][MAXM];
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv_id;                        // global wave id
    const int nwaves = gridDim.x * 4;
    const bool rope = flags & EPI_ROPE_APPEND;
    const bool swiglu = (flags & EPI_SWIGLU) || rope;             // two weight rows per output unit
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
    const int nchunk = K >> 3;
    if (staged) {
        float rstd[M];
#pragma unroll
        for (int m = 0; m < M; ++m) rstd[m] = 1.f;
        if (norm_w != nullptr) {
            float ss[M];
#pragma unroll
            for (int m = 0; m < M; ++m) ss[m] = 0.f;
            for (int c = threadIdx.x; c < nchunk; c += 256) {
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    float xv[8];
                    unpack8(*(const uint4*)(X + (long)m * ldx + c * 8), xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss[m] += xv[j] * xv[j];
                }
            }
#pragma unroll
            for (int m = 0; m < M; ++m) {
                ss[m] = wave_sum(ss[m]);
                if (lane == 0) red[wv_id][m] = ss[m];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < M; ++m) rstd[m] = rsqrtf((red[0][m] + red[1][m] + red[2][m] + red[3][m]) / (float)K + eps);
        }
        for (int c = threadIdx.x; c < nchunk; c += 256) {
            float wn[8];
            if (norm_w != nullptr) unpack8(*(const uint4*)(norm_w + c * 8), wn);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                uint4 raw = *(const uint4*)(X + (long)m * ldx + c * 8);
                if (norm_w != nullptr) {
                    float xv[8];
                    unpack8(raw, xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = wn[j] * rnd(xv[j] * rstd[m]);
                    raw = pack8(xv);
                }
                *(uint4*)(xs + (long)m * K + c * 8) = raw;
            }
        }
        __syncthreads();
    }
    for (int o = gw; o < n_out; o += nwaves) {
        // SwiGLU pack: output o <- gate row (o/16)*32 + o%16 and up row 16 below it
        int row0 = swiglu ? (o >> 4) * 32 + (o & 15) : o;
        int row1_off = 16;
        int r_sec = 0, r_head = 0, r_i = 0;                       // RoPE unit o -> section (q / k / v), head, element i < hd / 2
        if (rope) {
            const int half = ra.hd >> 1, per_sec = ra.H * half;
            r_sec = o / per_sec;
            const int rem = o - r_sec * per_sec;
            r_head = rem / half;
            r_i = rem - r_head * half;
            row0 = (r_sec * ra.H + r_head) * ra.hd + r_i;
            row1_off = half;
        }
        const elem_t* w0 = W + (long)row0 * ldw;
        const elem_t* w1 = w0 + (long)row1_off * ldw;
        float a0[M], a1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) a0[m] = a1[m] = 0.f;
        for (int c0 = lane; c0 < nchunk; c0 += 64 * U) {
            uint4 wq[U], uq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 64 * u;
                wq[u] = c < nchunk ? w_load16(w0 + c * 8) : make_uint4(0, 0, 0, 0);
                if (swiglu) uq[u] = c < nchunk ? w_load16(w1 + c * 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 64 * u;
                if (c < nchunk) {
                    float wv[8], uv[8];
                    unpack8(wq[u], wv);
                    if (swiglu) unpack8(uq[u], uv);
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        float xv[8];
                        if (staged) unpack8(*(const uint4*)(xs + (long)m * K + c * 8), xv);
                        else unpack8(*(const uint4*)(X + (long)m * ldx + c * 8), xv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            a0[m] += wv[j] * xv[j];
                            if (swiglu) a1[m] += uv[j] * xv[j];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            a0[m] = wave_sum(a0[m]);
            if (swiglu) a1[m] = wave_sum(a1[m]);
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                float t;
                if (rope) {
                    const int half = ra.hd >> 1;
                    const float x0 = rnd(a0[m]), x1 = rnd(a1[m]);              // what the q | k | v buffer would hold
                    const int b = m / ra.S, slot = ra.past + (m - b * ra.S);
                    if (r_sec < 2) {
                        const float cs = e2f(ra.cs[(long)m * half + r_i]), sn = e2f(ra.sn[(long)m * half + r_i]);
                        const float o1 = rnd(rnd(x0 * cs) + rnd(-x1 * sn)), o2 = rnd(rnd(x1 * cs) + rnd(x0 * sn));
                        elem_t* dst = r_sec == 0 ? (elem_t*)C + (long)m * ldc + r_head * ra.hd + r_i
                                                 : ra.kc + (((long)b * ra.H + r_head) * ra.smax + slot) * ra.hd + r_i;
                        dst[0] = f2e(o1);
                        dst[half] = f2e(o2);
                    } else {
                        const int w = slot & 31;
                        const int slot_v = (slot & ~31) + 8 * ((w >> 2) & 3) + 4 * (w >> 4) + (w & 3);
                        elem_t* dst = ra.vtc + (((long)b * ra.H + r_head) * ra.hd + r_i) * ra.smax + slot_v;
                        dst[0] = f2e(x0);
                        dst[(long)half * ra.smax] = f2e(x1);
                    }
                    continue;
                }
                if (swiglu) {
                    t = rnd(rnd(act_silu(rnd(a0[m]))) * rnd(a1[m]));
                } else {
                    t = a0[m];
                    if ((flags & EPI_BIAS) && (flags & EPI_BIAS_ROUNDED)) t = rnd(t);
                    if (flags & EPI_BIAS) t += e2f(bias[o]);
                    if (!(flags & EPI_OUT_F32) || act || (flags & EPI_RESID)) t = rnd(t);
                    if (act == 1) t = act_quick_gelu_e(t);
                    else if (act == 2) t = rnd(act_gelu_erf(t));
                    else if (act == 3) t = fmaxf(t, 0.f);
                }
                if (flags & EPI_RESID) t = rnd(e2f(R[(long)m * ldr + o]) + t);
                if (flags & EPI_OUT_F32) ((float*)C)[(long)m * ldc + o] = t;
                else ((elem_t*)C)[(long)m * ldc + o] = f2e(t);
            }
        }
    }
}

// ---- 2 <= M <= 16 (batched decode steps): the same weight stream on the matrix cores -------------------------------------------------
// At M = 4 the GEMV above already spends more time on its 4 x 8 FMAs per weight chunk than on the stream (2.2 TB/s), and M > 4 fell
// to the 128 x 128 GEMM whose grid is a few dozen blocks.  Here 16 output features x 16 (padded) rows of X are one 16x16x32 MFMA per
// 64 bytes of each weight row: a block of 8 waves owns 16 features (SwiGLU: 16 gate + 16 up rows), every wave one eighth of K,
// weight fragments straight from HBM (8 in flight per lane), X fragments from L2 (M x K x 2 bytes, shared by every block), partial
// sums through LDS, then the GEMV's epilogue (same flags, same rounding points).  MFMA work is 16 / M times the useful flops and
// still far below the stream's time.
template <bool SW>
__global__ __launch_bounds__(512) void skinny_gemm_kernel(const elem_t* __restrict__ X, long ldx, const elem_t* __restrict__ W, long ldw, void* C,
                                                          long ldc, const elem_t* __restrict__ bias, const elem_t* __restrict__ R, long ldr, int M,
                                                          int N, int K, int flags, int n_out) {
    __shared__ f32x4_t part[SW ? 2 : 1][8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int o0 = blockIdx.x * 16;                              // first output feature of the block
    // weight row of A-operand row fr: plain: o0 + fr; SwiGLU pack: gate rows (o0/16)*32 + fr, up rows 16 below
    const int wr0 = SW ? (o0 >> 4) * 32 + fr : min(o0 + fr, N - 1);
    const elem_t* w0 = W + (long)wr0 * ldw + fg * 8;
    const elem_t* w1 = w0 + 16 * ldw;
    const elem_t* xr = X + (long)min(fr, M - 1) * ldx + fg * 8;  // rows >= M repeat the last row: those accumulator columns are not stored
    const int nks = K >> 5;                                      // 32-wide k-steps
    const int ks0 = (int)((long)nks * wave / 8), ks1 = (int)((long)nks * (wave + 1) / 8);
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = SW ? 4 : 8;              // (SwiGLU streams two weight rows per fragment: 8 would cost the second block per CU its registers)
    for (int k0 = ks0; k0 < ks1; k0 += U) {
        uint4 wq[U], uq[U], xq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = min(k0 + u, ks1 - 1);                 // past the end: a repeated, unused fragment
            wq[u] = *(const uint4*)(w0 + ks * 32);
            if constexpr (SW) uq[u] = *(const uint4*)(w1 + ks * 32);
            xq[u] = *(const uint4*)(xr + ks * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < ks1) {
                a0 = mfma16(wq[u], xq[u], a0);
                if constexpr (SW) a1 = mfma16(uq[u], xq[u], a1);
            }
        }
    }
    part[0][wave][lane] = a0;
    if constexpr (SW) part[1][wave][lane] = a1;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < 8; ++w) {
        a0 += part[0][w][lane];
        if constexpr (SW) a1 += part[1][w][lane];
    }
    // a0[r] = (x_m . w_o) for m = fr, o = o0 + 4 fg + r
    const int m = fr;
    if (m >= M) return;
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
    float t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + fg * 4 + r;
        float v;
        if constexpr (SW) {
            v = rnd(rnd(act_silu(rnd(a0[r]))) * rnd(a1[r]));
        } else {
            v = a0[r];
            if (o < n_out) {
                if ((flags & EPI_BIAS) && (flags & EPI_BIAS_ROUNDED)) v = rnd(v);
                if (flags & EPI_BIAS) v += e2f(bias[o]);
                if (!(flags & EPI_OUT_F32) || act || (flags & EPI_RESID)) v = rnd(v);
                if (act == 1) v = act_quick_gelu_e(v);
                else if (act == 2) v = rnd(act_gelu_erf(v));
                else if (act == 3) v = fmaxf(v, 0.f);
            }
        }
        if ((flags & EPI_RESID) && o < n_out) v = rnd(e2f(R[(long)m * ldr + o]) + v);
        t[r] = v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + fg * 4 + r;
        if (o < n_out) {
            if (flags & EPI_OUT_F32) ((float*)C)[(long)m * ldc + o] = t[r];
            else ((elem_t*)C)[(long)m * ldc + o] = f2e(t[r]);
        }
    }
}

int launch_skinny(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr,
                  int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    if (!X || !W || !C || M <= 0 || N <= 0 || K <= 0) return ULL_ERR_ARG;
    if (M > 16 || (K & 31) || (ldx & 7) || (ldw & 7)) return ULL_ERR_SHAPE;
    if ((flags & EPI_BIAS) && !bias) return ULL_ERR_ARG;
    if ((flags & EPI_RESID) && !R) return ULL_ERR_ARG;
    if ((flags & EPI_SWIGLU) && ((N & 31) || (flags & (EPI_BIAS | EPI_ACT_MASK)))) return ULL_ERR_SHAPE;
    const int n_out = (int)((flags & EPI_SWIGLU) ? N / 2 : N);
    const unsigned blocks = (unsigned)((n_out + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
    if (flags & EPI_SWIGLU)
        hipLaunchKernelGGL(skinny_gemm_kernel<true>, dim3(blocks), dim3(512), 0, st, (const elem_t*)X, ldx, (const elem_t*)W, ldw, C, ldc,
                           (const elem_t*)bias, (const elem_t*)R, ldr, (int)M, (int)N, (int)K, flags, n_out);
    else
        hipLaunchKernelGGL(skinny_gemm_kernel<false>, dim3(blocks), dim3(512), 0, st, (const elem_t*)X, ldx, (const elem_t*)W, ldw, C, ldc,
                           (const elem_t*)bias, (const elem_t*)R, ldr, (int)M, (int)N, (int)K, flags, n_out);
    return ull_check_launch();
}

int launch_gemv(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr,
                int64_t M, int64_t N, int64_t K, int flags, const void* norm_w, float eps, void* stream, const RopeAppend* rope = nullptr) {
    if (!X || !W || !C || M <= 0 || N <= 0 || K <= 0) return ULL_ERR_ARG;
    if ((flags & EPI_ROPE_APPEND) && (!rope || flags != EPI_ROPE_APPEND || (N & 1))) return ULL_ERR_ARG;   // (not part of the public flags)
    const RopeAppend ra = rope ? *rope : RopeAppend{};
    if (M > MAXM || (K & 7) || (ldx & 7) || (ldw & 7)) return ULL_ERR_SHAPE;
    if ((flags & EPI_BIAS) && !bias) return ULL_ERR_ARG;
    if ((flags & EPI_RESID) && !R) return ULL_ERR_ARG;
    if ((flags & EPI_SWIGLU) && ((N & 31) || (flags & (EPI_BIAS | EPI_ACT_MASK)))) return ULL_ERR_SHAPE;
    const int staged = M * K * 2 <= XS_MAX_BYTES;
    if (norm_w && !staged) return ULL_ERR_SHAPE;
    const int lds = staged ? (int)(M * K * 2) : 0;
    const int n_out = (int)((flags & (EPI_SWIGLU | EPI_ROPE_APPEND)) ? N / 2 : N);
    // every block pays the X staging once, so give a block several output rows per wave: ~2 blocks per CU
    int blocks = (n_out + 3) / 4;
    if (staged && blocks > 1024) blocks = 1024;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
#define ULL_GV(MM, UU)                                                                                                                    \
    hipLaunchKernelGGL((gemv_kernel<MM, UU>), dim3(blocks), dim3(256), lds, st, (const elem_t*)X, ldx, (const elem_t*)W, ldw, C, ldc,     \
                       (const elem_t*)bias, (const elem_t*)R, ldr, (int)N, (int)K, flags, n_out, (const elem_t*)norm_w, eps, staged, ra)
    switch ((int)M) {
        case 1: ULL_GV(1, 8); break;
        case 2: ULL_GV(2, 4); break;
        case 3: ULL_GV(3, 4); break;
        default: ULL_GV(4, 4); break;
    }
#undef ULL_GV
    return ull_check_launch();
}

}  // namespace

// Same contract as ull_gemm_bf16 (flags, layouts) for M <= 4; K % 8 == 0.
extern "C" int ULL_FN(ull_gemv_)(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R,
                             int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    return launch_gemv(X, ldx, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, nullptr, 0.f, stream);
}

// The same with the preceding LlamaRMSNorm fused in: C = epilogue(rmsnorm(X; norm_w, eps) * W^T).  M * K <= 16384.
extern "C" int ULL_FN(ull_gemv_rmsnorm_)(const void* X, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* C,
                                     int64_t ldc, const void* bias, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, int flags,
                                     void* stream) {
    if (!norm_w) return ULL_ERR_ARG;
    return launch_gemv(X, ldx, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, norm_w, eps, stream);
}

// Decode-step q | k | v projection with RoPE and the KV-cache append in its epilogue (see RopeAppend): W = [3 * H * hd, K] (q | k | v rows),
// M = B * S tokens (<= 4), optional fused RMSNorm (norm_w may be null).  Q_out [M, H * hd] receives the rotated queries; the rotated keys
// go to k_cache[b, h, past + s, :], the values to vt_cache[b, h, :, slot(past + s)].  Same bits as ull_gemv_rmsnorm_ + ull_rope_append_.
extern "C" int ULL_FN(ull_gemv_qkv_rope_append_)(const void* X, int64_t ldx, const void* norm_w, float eps, const void* W, int64_t ldw, void* Q_out,
                                             int64_t ldq, const void* cos_tab, const void* sin_tab, void* k_cache, void* vt_cache, int64_t B,
                                             int64_t S, int64_t H, int64_t hd, int64_t K, int64_t smax, int64_t past, void* stream) {
    if (!cos_tab || !sin_tab || !k_cache || !vt_cache || B <= 0 || S <= 0 || H <= 0) return ULL_ERR_ARG;
    if (hd <= 0 || (hd & 1) || past < 0 || past + S > smax || ldq < H * hd) return ULL_ERR_SHAPE;
    RopeAppend ra;
    ra.cs = (const elem_t*)cos_tab; ra.sn = (const elem_t*)sin_tab; ra.kc = (elem_t*)k_cache; ra.vtc = (elem_t*)vt_cache;
    ra.S = (int)S; ra.H = (int)H; ra.hd = (int)hd; ra.smax = (int)smax; ra.past = (int)past;
    return launch_gemv(X, ldx, W, ldw, Q_out, ldq, nullptr, nullptr, 0, B * S, 3 * H * hd, K, EPI_ROPE_APPEND, norm_w, norm_w ? eps : 0.f, stream,
                       &ra);
}

// The same contract for 2 <= M <= 16 on the matrix cores (batched decode steps); K % 32 == 0.
extern "C" int ULL_FN(ull_gemm_skinny_)(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R,
                                    int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    return launch_skinny(X, ldx, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, stream);
}
