// Attention for gfx950 with the score strip resident in LDS.
//
// The reference's attention (transformers eager path used by LlamaAttention / CLIPAttention, and SAM's
// Attention modules) materialises   S = bf16(Q K^T) -> bf16(S * scale) -> (+mask) -> fp32 softmax -> bf16 P
// -> bf16(P V)   as separate bf16 tensors in HBM.  On MI355X a 64-query strip of that matrix
// (64 x 1024 bf16 = 128 KiB) fits in the CU's 160 KiB LDS, so this kernel keeps the *same rounding points*
// as the reference but never writes S or P to HBM:
//   phase 1  S strip  = K-tile x Q^T on MFMA (operands swapped so a lane owns 4 consecutive keys of one
//            query -> one 8-byte LDS store), rounded/scaled/masked exactly like the eager graph;
//   phase 2  exact row softmax in fp32 over the bf16 strip (true row max, no online rescaling), P
//            written back over S as bf16;
//   phase 3  O^T = V^T-tile x P^T on MFMA (V^T is produced K-contiguous by ull_transpose_v_bf16 so both
//            operands are plain 16-byte LDS reads), 8-byte stores into [token, head*hd] row-major O.
// Causal tiles beyond the diagonal are skipped in all three phases.
//
// Also here: the QKV post-processing kernels (RoPE in place on q|k, V -> V^T with zero padding).
#include "ull_common.h"

namespace {

constexpr int KT = 64;                 // keys per tile
constexpr uint16_t BF16_NEG_INF = 0xFF80;
constexpr uint16_t BF16_MIN = 0xFF7F;  // torch.finfo(torch.bfloat16).min: the eager additive mask value

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
    const int32_t* key_mask;            // [B, Sk] nonzero = may be attended, or null
    long q_bs, q_hs, q_ss, k_bs, k_hs, k_ss, vt_bs, vt_hs, vt_ds, o_bs, o_hs, o_ss;
    int B, H, Sq, Sk, hd, vt_len;
    int causal, scale_mode;
    float scale;
    int pitch;                          // strip row pitch in bytes (multiple of 16)
};

template <int HDP, int NW>
__global__ __launch_bounds__(NW * 64) void attn_strip_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BQ = 16 * NW;
    constexpr int KP = HDP * 2 + 16;     // K-tile row pitch (bytes): +16 breaks the power-of-two stride
    constexpr int VP = KT * 2 + 16;      // V^T-tile row pitch (bytes)
    constexpr int NKS = HDP / 32;        // MFMA k-steps over the head dim
    constexpr int NDS = HDP / 16;        // 16-wide d sub-tiles of the output
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * BQ;
    char* strip = smem;
    char* tile = smem + BQ * p.pitch;
    const int koff = p.Sk - p.Sq;        // query i sees key j <= i + koff when causal

    // ---- Q fragments (B operand: lane -> query fr, dims 8*fg.. of k-step ks) --------------------
    uint4 qf[NKS];
    {
        const int qi = q0 + wave * 16 + fr;
        const bf16_t* qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qi * p.q_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + fg * 8;
            qf[ks] = (qi < p.Sq && d < p.hd) ? *(const uint4*)(qp + d) : make_uint4(0, 0, 0, 0);
        }
    }
    int kend = p.Sk;
    if (p.causal) kend = min(p.Sk, q0 + BQ + koff);
    if (kend < 1) kend = 1;
    const int nkt = (kend + KT - 1) / KT;

    // ---- phase 1: score strip -------------------------------------------------------------
    const bf16_t* kbase = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        for (int i = tid; i < KT * (HDP / 8); i += NW * 64) {
            const int row = i / (HDP / 8), ch = i % (HDP / 8);
            const int key = kt * KT + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (key < p.Sk && ch * 8 < p.hd) v = *(const uint4*)(kbase + (long)key * p.k_ss + ch * 8);
            *(uint4*)(tile + row * KP + ch * 16) = v;
        }
        __syncthreads();
        const int qi = q0 + wave * 16 + fr;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const uint4 kf = *(const uint4*)(tile + (ns * 16 + fr) * KP + (ks * 4 + fg) * 16);
                acc = mfma16(kf, qf[ks], acc);
            }
            // acc[r] = S[key = kt*64 + ns*16 + 4*fg + r][query = fr]
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kt * KT + ns * 16 + fg * 4 + r;
                float s = rbf(acc[r]);
                if (p.scale_mode == 1) s = rbf(s * p.scale);
                else if (p.scale_mode == 2) s = rbf(s / p.scale);
                bool allowed = true;
                if (p.causal) allowed = j <= qi + koff;
                if (allowed && p.key_mask != nullptr && j < p.Sk) allowed = p.key_mask[(long)b * p.Sk + j] != 0;
                o[r] = (j >= p.Sk) ? BF16_NEG_INF : (allowed ? f2bf(s) : BF16_MIN);
            }
            uint2 pk;
            pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *(uint2*)(strip + (wave * 16 + fr) * p.pitch + (kt * KT + ns * 16 + fg * 4) * 2) = pk;
        }
    }
    __syncthreads();

    // ---- phase 2: row softmax (4 lanes per row, this wave's 16 rows) ------------------------------
    {
        const int row = lane >> 2, sub = lane & 3;
        char* rp = strip + (wave * 16 + row) * p.pitch;
        const int nch = nkt * (KT / 8);
        float m = -INFINITY;
        for (int c = sub; c < nch; c += 4) {
            float f[8];
            unpack8(*(const uint4*)(rp + c * 16), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) m = fmaxf(m, f[j]);
        }
        m = group_max(m, 4);
        float sum = 0.f;
        for (int c = sub; c < nch; c += 4) {
            float f[8];
            unpack8(*(const uint4*)(rp + c * 16), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += __expf(f[j] - m);
        }
        sum = group_sum(sum, 4);
        const float inv = 1.0f / sum;
        for (int c = sub; c < nch; c += 4) {
            float f[8];
            unpack8(*(const uint4*)(rp + c * 16), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = __expf(f[j] - m) * inv;
            *(uint4*)(rp + c * 16) = pack8(f);
        }
    }

    // ---- phase 3: O^T = V^T P^T --------------------------------------------------------------
    f32x4_t oacc[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) oacc[ds] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16_t* vbase = p.Vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        for (int i = tid; i < p.hd * (KT / 8); i += NW * 64) {
            const int d = i >> 3, ch = i & 7;
            const int key0 = kt * KT + ch * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (key0 + 8 <= p.vt_len) v = *(const uint4*)(vbase + (long)d * p.vt_ds + key0);
            *(uint4*)(tile + d * VP + ch * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint4 pf = *(const uint4*)(strip + (wave * 16 + fr) * p.pitch + (kt * KT + kk * 32 + fg * 8) * 2);
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                if (ds * 16 < p.hd) {
                    const uint4 vf = *(const uint4*)(tile + (ds * 16 + fr) * VP + (kk * 4 + fg) * 16);
                    oacc[ds] = mfma16(vf, pf, oacc[ds]);
                }
            }
        }
    }
    // oacc[ds][r] = O[d = ds*16 + 4*fg + r][query = fr]
    const int qi = q0 + wave * 16 + fr;
    if (qi < p.Sq) {
        bf16_t* op = p.O + (long)b * p.o_bs + (long)h * p.o_hs + (long)qi * p.o_ss;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            if (ds * 16 < p.hd) {
                uint2 pk;
                pk.x = pack2bf(oacc[ds][0], oacc[ds][1]);
                pk.y = pack2bf(oacc[ds][2], oacc[ds][3]);
                *(uint2*)(op + ds * 16 + fg * 4) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// RoPE in place on the q|k part of a fused QKV buffer (transformers apply_rotary_pos_emb on bf16
// tensors: q*cos -> bf16, rotate_half(q)*sin -> bf16, sum -> bf16; cos/sin are fp32 values cast to bf16).
// One block per token; thread t owns the 8-wide dim chunk (t % (hd/16)) of head-instances t / (hd/16), ...
__global__ __launch_bounds__(256) void rope_inplace_kernel(bf16_t* __restrict__ x, long row_stride, const int64_t* __restrict__ pos,
                                                           const float* __restrict__ inv_freq, int n_heads, int hd) {
    const long tok = blockIdx.x;
    const int half = hd >> 1;
    const int cpr = half >> 3;                   // 8-wide chunks per half head (hd % 16 == 0)
    const int c = threadIdx.x % cpr;
    const float pf = (float)pos[tok];
    float cs[8], sn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = pf * inv_freq[c * 8 + j];
        cs[j] = rbf(cosf(a));
        sn[j] = rbf(sinf(a));
    }
    bf16_t* xr = x + tok * row_stride;
    for (int hh = threadIdx.x / cpr; hh < n_heads; hh += blockDim.x / cpr) {
        bf16_t* p1 = xr + hh * hd + c * 8;
        bf16_t* p2 = p1 + half;
        float a[8], bb[8], o1[8], o2[8];
        unpack8(*(const uint4*)p1, a);
        unpack8(*(const uint4*)p2, bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o1[j] = rbf(a[j] * cs[j]) + rbf(-bb[j] * sn[j]);
            o2[j] = rbf(bb[j] * cs[j]) + rbf(a[j] * sn[j]);
        }
        *(uint4*)p1 = pack8(o1);
        *(uint4*)p2 = pack8(o2);
    }
}

// V [B, S, H, hd] (token stride row_stride, heads contiguous) -> Vt [B, H, hd, pitch], zero-filled for
// s in [S, pitch).  64(s) x 64(d) tiles through LDS.
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ v, long v_bs, long v_ss, bf16_t* __restrict__ vt, int S,
                                                          int H, int hd, int pitch) {
    __shared__ bf16_t t[64][66];
    const int b = blockIdx.z / H, h = blockIdx.z % H;
    const int s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const bf16_t* vp = v + (long)b * v_bs + (long)h * hd;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int s = i >> 6, d = i & 63;
        bf16_t val = 0;
        if (s0 + s < S && d0 + d < hd) val = vp[(long)(s0 + s) * v_ss + d0 + d];
        t[s][d] = val;
    }
    __syncthreads();
    bf16_t* op = vt + ((long)b * H + h) * hd * pitch;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int d = i >> 6, s = i & 63;
        if (d0 + d < hd && s0 + s < pitch) op[(long)(d0 + d) * pitch + s0 + s] = t[s][d];
    }
}

template <int HDP>
int launch_attn(const AttnArgs& a, hipStream_t st) {
    constexpr int NW = 4;
    constexpr int BQ = 16 * NW;
    constexpr int tile_bytes = (KT * (HDP * 2 + 16) > HDP * (KT * 2 + 16)) ? KT * (HDP * 2 + 16) : HDP * (KT * 2 + 16);
    const int lds = BQ * a.pitch + tile_bytes;
    if (lds > 160 * 1024) return ULL_ERR_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_strip_kernel<HDP, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const dim3 grid((a.Sq + BQ - 1) / BQ, a.H, a.B);
    hipLaunchKernelGGL((attn_strip_kernel<HDP, NW>), grid, dim3(NW * 64), lds, st, a);
    return ull_check_launch();
}

}  // namespace

// Strides are in elements.  Q/K rows are head_dim-contiguous; Vt rows (one per head dim) are key-contiguous with
// `vt_len` readable, finite columns (multiple of 8; keys >= Sk must be zero).  key_mask: int32 [B, Sk] or null.
// scale_mode 0: S = bf16(QK^T); 1: bf16(bf16(QK^T) * scale); 2: bf16(bf16(QK^T) / scale).
extern "C" int ull_attention_bf16(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* K, int64_t k_bs, int64_t k_hs,
                                  int64_t k_ss, const void* Vt, int64_t vt_bs, int64_t vt_hs, int64_t vt_ds, int64_t vt_len, void* O,
                                  int64_t o_bs, int64_t o_hs, int64_t o_ss, const void* key_mask, int64_t B, int64_t H, int64_t Sq,
                                  int64_t Sk, int64_t hd, int causal, int scale_mode, float scale, void* stream) {
    if (!Q || !K || !Vt || !O || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return ULL_ERR_ARG;
    if (hd <= 0 || hd > 128 || (hd & 15) || (vt_len & 7) || vt_len < ((Sk + 7) & ~7)) return ULL_ERR_SHAPE;
    if ((q_ss & 7) || (k_ss & 7) || (vt_ds & 7) || (q_hs & 7) || (k_hs & 7) || (q_bs & 7) || (k_bs & 7) || (vt_hs & 7) || (vt_bs & 7) ||
        (o_ss & 3) || (o_hs & 3) || (o_bs & 3))
        return ULL_ERR_SHAPE;
    AttnArgs a;
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.Vt = (const bf16_t*)Vt; a.O = (bf16_t*)O;
    a.key_mask = (const int32_t*)key_mask;
    a.q_bs = q_bs; a.q_hs = q_hs; a.q_ss = q_ss; a.k_bs = k_bs; a.k_hs = k_hs; a.k_ss = k_ss;
    a.vt_bs = vt_bs; a.vt_hs = vt_hs; a.vt_ds = vt_ds; a.o_bs = o_bs; a.o_hs = o_hs; a.o_ss = o_ss;
    a.B = (int)B; a.H = (int)H; a.Sq = (int)Sq; a.Sk = (int)Sk; a.hd = (int)hd; a.vt_len = (int)vt_len;
    a.causal = causal; a.scale_mode = scale_mode; a.scale = scale;
    const int lmax = (int)((Sk + KT - 1) / KT) * KT;
    a.pitch = lmax * 2 + 16;
    hipStream_t st = (hipStream_t)stream;
    if (hd <= 32) return launch_attn<32>(a, st);
    if (hd <= 64) return launch_attn<64>(a, st);
    if (hd <= 96) return launch_attn<96>(a, st);
    return launch_attn<128>(a, st);
}

// x: first of `n_heads` consecutive heads (q heads then k heads of a fused QKV row); positions int64 [tokens];
// inv_freq fp32 [hd/2] (host-computed exactly like LlamaRotaryEmbedding).
extern "C" int ull_rope_inplace_bf16(void* x, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens,
                                     int64_t n_heads, int64_t hd, void* stream) {
    if (!x || !positions || !inv_freq || tokens <= 0) return ULL_ERR_ARG;
    const int64_t cpr = hd >> 4;
    if ((hd & 15) || hd > 256 || (cpr & (cpr - 1)) || (row_stride & 7)) return ULL_ERR_SHAPE;   // 256 % (hd/16) == 0
    hipLaunchKernelGGL(rope_inplace_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, row_stride,
                       (const int64_t*)positions, (const float*)inv_freq, (int)n_heads, (int)hd);
    return ull_check_launch();
}

extern "C" int ull_transpose_v_bf16(const void* v, int64_t v_bs, int64_t v_ss, void* vt, int64_t B, int64_t S, int64_t H, int64_t hd,
                                    int64_t pitch, void* stream) {
    if (!v || !vt || B <= 0 || S <= 0) return ULL_ERR_ARG;
    if (pitch < S || (pitch & 7)) return ULL_ERR_SHAPE;
    const dim3 grid((unsigned)((pitch + 63) / 64), (unsigned)((hd + 63) / 64), (unsigned)(B * H));
    hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v, v_bs, v_ss, (bf16_t*)vt, (int)S, (int)H,
                       (int)hd, (int)pitch);
    return ull_check_launch();
}
