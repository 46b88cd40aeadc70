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
constexpr int MAXM = 4;

constexpr int XS_MAX_BYTES = 32 * 1024;     // X (optionally RMS-normalised) is staged in LDS when M * K * 2 fits in this

// X staged in LDS (`staged`): every block first copies -- or, with norm_w, RMS-normalises (transformers LlamaRMSNorm:
// w * bf16(x * rsqrt(mean(x^2) + eps)), the op that precedes the q/k/v and gate/up projections) -- the M activation rows,
// which removes one tiny latency-bound kernel per projection from the decode step.  The weight stream keeps U 16-byte loads
// per lane in flight (U KiB per wave).
template <int M, int U>
__global__ __launch_bounds__(256) void gemv_kernel(const elem_t* __restrict__ X, long ldx, const elem_t* __restrict__ W, long ldw, void* C,
                                                   long ldc, const elem_t* __restrict__ bias, const elem_t* __restrict__ R, long ldr, int N, int K,
                                                   int flags, int n_out, const elem_t* __restrict__ norm_w, float eps, int staged) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem_t* xs = (elem_t*)smem;                                   // [M][K] when staged
    __shared__ float red[4][MAXM];
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wv_id;                        // global wave id
    const int nwaves = gridDim.x * 4;
    const bool swiglu = flags & EPI_SWIGLU;
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
        const int row0 = swiglu ? (o >> 4) * 32 + (o & 15) : o;
        const elem_t* w0 = W + (long)row0 * ldw;
        const elem_t* w1 = w0 + 16 * ldw;
        float a0[M], a1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) a0[m] = a1[m] = 0.f;
        for (int c0 = lane; c0 < nchunk; c0 += 64 * U) {
            uint4 wq[U], uq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 64 * u;
                wq[u] = c < nchunk ? *(const uint4*)(w0 + c * 8) : make_uint4(0, 0, 0, 0);
                if (swiglu) uq[u] = c < nchunk ? *(const uint4*)(w1 + c * 8) : make_uint4(0, 0, 0, 0);
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

int launch_gemv(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R, int64_t ldr,
                int64_t M, int64_t N, int64_t K, int flags, const void* norm_w, float eps, void* stream) {
    if (!X || !W || !C || M <= 0 || N <= 0 || K <= 0) return ULL_ERR_ARG;
    if (M > MAXM || (K & 7) || (ldx & 7) || (ldw & 7)) return ULL_ERR_SHAPE;
    if ((flags & EPI_BIAS) && !bias) return ULL_ERR_ARG;
    if ((flags & EPI_RESID) && !R) return ULL_ERR_ARG;
    if ((flags & EPI_SWIGLU) && ((N & 31) || (flags & (EPI_BIAS | EPI_ACT_MASK)))) return ULL_ERR_SHAPE;
    const int staged = M * K * 2 <= XS_MAX_BYTES;
    if (norm_w && !staged) return ULL_ERR_SHAPE;
    const int lds = staged ? (int)(M * K * 2) : 0;
    const int n_out = (int)((flags & EPI_SWIGLU) ? N / 2 : N);
    // every block pays the X staging once, so give a block several output rows per wave: ~2 blocks per CU
    int blocks = (n_out + 3) / 4;
    if (staged && blocks > 1024) blocks = 1024;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
#define ULL_GV(MM, UU)                                                                                                                    \
    hipLaunchKernelGGL((gemv_kernel<MM, UU>), dim3(blocks), dim3(256), lds, st, (const elem_t*)X, ldx, (const elem_t*)W, ldw, C, ldc,     \
                       (const elem_t*)bias, (const elem_t*)R, ldr, (int)N, (int)K, flags, n_out, (const elem_t*)norm_w, eps, staged)
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
