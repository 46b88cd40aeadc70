// Attention for gfx950 with the whole score row resident in registers.
//
// The reference's attention (transformers eager path used by LlamaAttention / CLIPAttention, and SAM's
// Attention modules) materialises   S = bf16(Q K^T) -> bf16(S * scale) -> (+mask) -> fp32 softmax -> bf16 P
// -> bf16(P V)   as separate bf16 tensors in HBM.  Sequences on this path are <= 1024 tokens, so one wave can
// keep its 16 queries' complete score rows in VGPRs as packed bf16 (8 registers per 64 keys, 128 for 1024 keys;
// gfx950 gives a wave 256 at two waves/SIMD).  The kernel therefore keeps the *same rounding points* as the
// reference but S and P never leave the register file:
//   phase 1  S = K-tile x Q^T on MFMA (operands swapped so a lane owns 4 consecutive keys of one query),
//            rounded / scaled / masked exactly like the eager graph, packed to bf16 in registers;
//   phase 2  exact row softmax in fp32 (true row max, no online rescaling), two shuffles per reduction;
//   phase 3  O^T = V^T-tile x P^T on MFMA: the P registers ARE the B operand (V^T is stored with the matching
//            key permutation by ull_transpose_v_bf16), 8-byte stores into [token, head*hd] row-major O.
// K and V^T tiles stream HBM/L2 -> LDS by LDS-DMA, double-buffered through all three phases (the first V^T tile
// lands while the softmax runs).  Causal tiles beyond a wave's diagonal are skipped.  All query tiles of one head are
// placed on the same XCD so K/V are fetched into one L2 only.
//
// Also here: the QKV post-processing kernels (RoPE in place on q|k, V -> V^T with zero padding).
#include "ull_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KT = 64;                 // keys per tile

struct AttnArgs {
    const elem_t* Q; const elem_t* K; const elem_t* Vt; elem_t* O;
    const int32_t* key_mask;            // [B, Sk] nonzero = may be attended, or null
    long q_bs, q_hs, q_ss, k_bs, k_hs, k_ss, vt_bs, vt_hs, vt_ds, o_bs, o_hs, o_ss;
    int B, H, Sq, Sk, hd, vt_len;
    int causal, scale_mode;
    float scale;
    const elem_t* zeros;                // >= 16 readable zero bytes (source of the head-dim padding chunks)
    // SAM decomposed relative-position bias (image_encoder.py:354-392): S += rel_h[q, key / KW]; S += rel_w[q, key % KW]
    const elem_t* rel_h; const elem_t* rel_w;   // [B*H, Sq, KH] / [B*H, Sq, KW] or null
    int KH, KW;
    int rel_mode;                       // 1: rel_h/rel_w are per-query tables [B*H,Sq,KH|KW]; 2: they are the raw rel_pos_h/w parameters
    int win16;                          // ull_sam_window_attention: sam_window_kernel, see below
    uint32_t mg_h, mg_nwx, mg_nwy, mg_nw;   // win16: ceil(2^32 / d) for d = H, nwx, nwy, nwx * nwy (udiv_magic: no run-time integer division in the kernel)
    int v_rows;                         // Vt is V itself, [B,H,S,hd] by (vt_bs, vt_hs, vt_ds = token stride): kernels with a VROW form
                                        //    [2KH-1,hd] / [2KW-1,hd] and the tables are built in the kernel prologue on the MFMA
    float inv_kw;                       // 1 / KW
    float q_scale;                      // != 1: Q is consumed as bf16(q * q_scale)  (SAM: (q * scale) @ k^T)
    // win16 (ull_sam_window_attention): Q / K / V / O rows are tokens of [img, img_h, img_w] grids in image order, "batch" b is
    // window (img, wy, wx) of the 14 x 14 partition, Vt points at the V part of the rows (same strides as K), and window positions
    // outside the grid are the reference's zero padding (image_encoder.py:262-289 pads AFTER norm1, so a padded token's q|k|v is
    // the qkv bias): K / V rows of such keys come from k_pad / v_pad.
    int img_h, img_w, nwy, nwx;
    const elem_t* k_pad; const elem_t* v_pad;   // K / V part of the pad token's row (+ h * k_hs)
};
#define ULL_ATT_ACC(slot, t0) ((void)0)

// ull_sam_window_attention: window b = (img * nwy + wy) * nwx + wx -> (image, first token row, first token column).  The divisors are
// run-time values; a / d goes through the host-computed m = ceil(2^32 / d): exact while a * d < 2^32 (the dispatcher checks), m = 0
// encodes d = 1.  (As three integer divisions, ~30 scalar instructions each and repeated per DMA piece, this was ~1200 scalar
// instructions in every wave's prologue.)
struct WinOrigin { int img, iy0, ix0; };
ULL_DEV int udiv_magic(int a, uint32_t m) { return m ? (int)__umulhi((uint32_t)a, m) : a; }
ULL_DEV WinOrigin win_origin(const AttnArgs& p, int b, int ws) {
    const int t = udiv_magic(b, p.mg_nwx), wx = b - t * p.nwx;
    const int img = udiv_magic(b, p.mg_nw), wy = t - img * p.nwy;
    return WinOrigin{img, wy * ws, wx * ws};
}

// Compile-time "flavors" of the score epilogue.  The runtime-flag version (FL_RUNTIME) costs ~6 wave-uniform branches per
// score element, which fragments the schedule (measured: SAM global attention 12 ms -> see profiles/); the hot callers
// get straight-line code instead.
constexpr int FL_RUNTIME = -1;   // every switch read from AttnArgs at run time (any combination)
constexpr int FL_LLAMA = 0;      // S*scale, causal + key-padding mask            (hf llama eager_attention_forward)
constexpr int FL_CLIP = 1;       // S*scale                                        (hf clip eager_attention_forward)
constexpr int FL_SAM_ENC = 2;    // (q*scale) pre-scaled, + rel_h, + rel_w         (SAM image_encoder.py Attention)
constexpr int FL_SAM_DEC = 3;    // S / sqrt(hd)                                   (SAM transformer.py Attention)

// One lane's 4 consecutive scores of one query -> the reference's rounding chain -> two packed bf16 pairs.
//   acc[r] = raw fp32 dot product for key j0 + r;  mk = 4 mask bytes (1 attend, 0 masked, 2 out of range)
//   brow   = this query's bias row in LDS, rel_h(kh) = brow[bh_off - kh], rel_w(kw) = brow[bw_off - kw]; or null
template <int FL>
ULL_DEV void score_quad(const AttnArgs& p, const f32x4_t& acc, int j0, uint32_t mk, int qi, int koff, const elem_t* brow, int bh_off,
                        int bw_off, uint32_t& lo, uint32_t& hi, float* row_max = nullptr) {
    const bool do_mul = FL == FL_RUNTIME ? p.scale_mode == 1 : (FL == FL_LLAMA || FL == FL_CLIP);
    const bool do_div = FL == FL_RUNTIME ? p.scale_mode == 2 : (FL == FL_SAM_DEC);
    const bool do_bias = FL == FL_RUNTIME ? brow != nullptr : (FL == FL_SAM_ENC);
    const bool do_causal = FL == FL_RUNTIME ? p.causal != 0 : (FL == FL_LLAMA);
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + r;
        float sv = rnd(acc[r]);
        if (do_mul) sv = rnd(sv * p.scale);
        if (do_div) sv = rnd(sv / p.scale);
        if (do_bias) {
            // j / KW without the integer-division sequence: exact for j < 2^16, KW <= 256 (|err| << 0.5 / KW)
            const int kh = min((int)(((float)j + 0.5f) * p.inv_kw), p.KH - 1), kw = j - kh * p.KW;
            sv = rnd(rnd(sv + e2f(brow[bh_off - kh])) + e2f(brow[bw_off - kw]));
        }
        const uint32_t mb = (mk >> (8 * r)) & 0xff;
        const bool allowed = (mb == 1) && (!do_causal || j <= qi + koff);
        o[r] = (mb == 2) ? -INFINITY : (allowed ? sv : ELEM_MIN_F);   // -inf / finfo(bf16).min, exact in bf16
    }
    if (row_max) *row_max = fmaxf(fmaxf(*row_max, fmaxf(o[0], o[1])), fmaxf(o[2], o[3]));   // values are already 16-bit exact
    lo = pack2e(o[0], o[1]);
    hi = pack2e(o[2], o[3]);
}

// The same for a quad whose four keys are all attendable for every lane of the wave (no padding, below the causal diagonal, no
// bias): only the scale + the two roundings remain.  ~85 % of the LLaMA / CLIP score quads take this path.
// LLaMA / CLIP (S * scale): rnd(acc) two at a time through one packed convert, the scale as one packed multiply, and -- rounding to 16
// bits is monotone, so max_i rnd(x_i) = rnd(max_i x_i) -- the row maximum is fed with the UNROUNDED products (one v_max3 per pair instead
// of two unpacks and two v_max); the caller rounds the row maximum once.  11 vector instructions per quad instead of 26, same bits.
template <int FL>
ULL_DEV void score_quad_clean(const AttnArgs& p, const f32x4_t& acc, uint32_t& lo, uint32_t& hi, float* row_max = nullptr) {
    if constexpr (FL == FL_LLAMA || FL == FL_CLIP) {
        // (the packed pairs are made opaque: seeing through pack -> unpack, the compiler converts every value on its own again --
        //  4 single conversions + 4 shifts instead of 2 packed conversions + 2 shifts + 2 ands; census in profiles/r05_attn_prefill_census.txt)
        uint32_t a01 = pack2e(acc[0], acc[1]), a23 = pack2e(acc[2], acc[3]);
        asm volatile("" : "+v"(a01), "+v"(a23));
        const f32x2_t x01 = f32x2_t{pk_lo(a01), pk_hi(a01)} * p.scale, x23 = f32x2_t{pk_lo(a23), pk_hi(a23)} * p.scale;
        if (row_max) *row_max = fmaxf(fmaxf(fmaxf(*row_max, x01.x), x01.y), fmaxf(x23.x, x23.y));
        lo = pack2e(x01.x, x01.y);
        hi = pack2e(x23.x, x23.y);
        return;
    }
    const bool do_mul = FL == FL_RUNTIME ? p.scale_mode == 1 : false;
    const bool do_div = FL == FL_RUNTIME ? p.scale_mode == 2 : (FL == FL_SAM_DEC);
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sv = rnd(acc[r]);
        if (do_mul) sv = rnd(sv * p.scale);
        if (do_div) sv = rnd(sv / p.scale);
        o[r] = sv;
    }
    if (row_max) *row_max = fmaxf(fmaxf(*row_max, fmaxf(o[0], o[1])), fmaxf(o[2], o[3]));
    lo = pack2e(o[0], o[1]);
    hi = pack2e(o[2], o[3]);
}


// SAM 14 x 14 windows (sam_window_kernel): one lane's scores of window row kh for its four window columns kw = 4 fg + r:
// rnd(rnd(rnd(acc) + rel_h[kh]) + rel_w[kw]), the reference's three roundings, two values at a time: one packed convert rounds a pair, the
// two adds are one packed add, the third rounding IS the packed pair that is kept, and the row maximum takes the unrounded sums (rounding
// is monotone; the caller rounds the maximum once).  wv23 = -inf in the lanes whose columns are the padding slots kw = 14, 15.
// 21 vector instructions per window row instead of 45, same bits (the kernel is bound by its vector-issue slots: docs/experiments.md).
ULL_DEV void score_quad_win(const f32x4_t& acc, float hb, const f32x2_t& wv01, const f32x2_t& wv23, uint32_t& lo, uint32_t& hi, float& row_max) {
    // (the packed pairs are made opaque: seeing through pack -> unpack, the compiler converts every value on its own again)
    uint32_t a01 = pack2e(acc[0], acc[1]), a23 = pack2e(acc[2], acc[3]);
    asm volatile("" : "+v"(a01), "+v"(a23));
    const f32x2_t x01 = f32x2_t{pk_lo(a01), pk_hi(a01)} + hb, x23 = f32x2_t{pk_lo(a23), pk_hi(a23)} + hb;
    uint32_t b01 = pack2e(x01.x, x01.y), b23 = pack2e(x23.x, x23.y);
    asm volatile("" : "+v"(b01), "+v"(b23));
    const f32x2_t y01 = f32x2_t{pk_lo(b01), pk_hi(b01)} + wv01, y23 = f32x2_t{pk_lo(b23), pk_hi(b23)} + wv23;
    row_max = fmaxf(fmaxf(fmaxf(row_max, y01.x), y01.y), fmaxf(y23.x, y23.y));
    lo = pack2e(y01.x, y01.y);
    hi = pack2e(y23.x, y23.y);
    asm volatile("" : "+v"(lo), "+v"(hi));
}

// ... and the exact fp32 softmax over the lane's 56 scores (14 window rows x 4 columns; the row's other 168 sit in the lanes fr, fr + 16,
// fr + 32, fr + 48): every exponential is evaluated once and kept, subtraction / log2(e) / normalisation are packed fp32 operations.
// mrow = the lane's running maximum from score_quad_win.  P = 16-bit softmax, in place.
ULL_DEV void softmax_win(uint32_t (&sp)[4][8], float mrow) {
    float m = rnd(mrow);
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
    f32x2_t e[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) {
        const f32x2_t t = (f32x2_t{pk_lo(sp[i / 8][i % 8]), pk_hi(sp[i / 8][i % 8])} - m) * 1.4426950408889634f;   // __expf(x) = exp2(x * log2 e)
        e[i] = f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        sum += e[i].x;
        sum += e[i].y;
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 28; ++i) {
        const f32x2_t q = e[i] * inv;
        sp[i / 8][i % 8] = pack2e(q.x, q.y);
    }
}

// Stage this wave's 16 relative-position bias rows in LDS (`dst`, row pitch `bp` elements, see bias_pitch()).
//   rel_mode 1: copy the precomputed per-query tables (stored reversed so both modes index the same way);
//   rel_mode 2: build them here: G[q][t] = bf16(q . rel_pos[t]) for every table row t on the MFMA (the reference's
//               einsum("bhwc,hkc->bhwk") is a Toeplitz slice of exactly this product: rel_h[q][kh] = Gh[q][qy - kh + KH - 1]).
// qf0 = the wave's UNSCALED query fragments.  Returns the two lookup offsets of this lane's query.
template <int NKS>
ULL_DEV void stage_rel_bias(const AttnArgs& p, elem_t* dst, int bp, const uint4 (&qf0)[NKS], int q_first, long head, int lane,
                            int& bh_off, int& bw_off, int hd) {
    const int fr = lane & 15, fg = lane >> 4;
    const int qi = min(q_first + fr, p.Sq - 1);
    if (p.rel_mode == 1) {
        const int bw = p.KH + p.KW;
        for (int i = lane; i < 16 * bw; i += 64) {
            const int r = i / bw, c = i % bw;
            const long row = head * p.Sq + min(q_first + r, p.Sq - 1);
            if (c < p.KH) dst[r * bp + (p.KH - 1 - c)] = p.rel_h[row * p.KH + c];
            else dst[r * bp + p.KH + (p.KW - 1 - (c - p.KH))] = p.rel_w[row * p.KW + (c - p.KH)];
        }
        bh_off = p.KH - 1;
        bw_off = p.KH + p.KW - 1;
    } else {
        const int nth = 2 * p.KH - 1, ntw = 2 * p.KW - 1;
        if (nth <= 32 && ntw <= 32) {
            // window-sized tables (14 x 14 -> 27 rows each): all 4 x NKS table fragments are requested before the first MFMA, so the
            // block pays one memory round trip here instead of four dependent ones (this prologue was ~1/4 of a window block's time)
            uint4 a[2][2][NKS];
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const elem_t* tab = which ? p.rel_w : p.rel_h;
                    const int nt = which ? ntw : nth;
                    const int t = min(st * 16 + fr, nt - 1);
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const int d = ks * 32 + fg * 8;
                        a[which][st][ks] = (d < hd) ? *(const uint4*)(tab + (long)t * hd + d) : make_uint4(0, 0, 0, 0);
                    }
                }
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int nt = which ? ntw : nth, base = which ? nth : 0;
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks)
                        if (ks * 32 < hd) acc = mfma16(a[which][st][ks], qf0[ks], acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int tt = st * 16 + fg * 4 + r;
                        if (tt < nt) dst[fr * bp + base + tt] = f2e(acc[r]);
                    }
                }
        } else
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const elem_t* tab = which ? p.rel_w : p.rel_h;
            const int nt = which ? ntw : nth, base = which ? nth : 0;
            for (int st = 0; st * 16 < nt; ++st) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const int t = min(st * 16 + fr, nt - 1);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const int d = ks * 32 + fg * 8;
                    if (ks * 32 < hd) {
                        const uint4 a = (d < hd) ? *(const uint4*)(tab + (long)t * hd + d) : make_uint4(0, 0, 0, 0);
                        acc = mfma16(a, qf0[ks], acc);
                    }
                }
                // acc[r] = G[t = st*16 + 4*fg + r][query fr]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tt = st * 16 + fg * 4 + r;
                    if (tt < nt) dst[fr * bp + base + tt] = f2e(acc[r]);
                }
            }
        }
        bh_off = qi / p.KW + p.KH - 1;
        bw_off = nth + qi % p.KW + p.KW - 1;
    }
}

ULL_DEV int bias_pitch(const AttnArgs& p) {       // elements; odd so the 16 query rows start in different LDS banks
    const int n = p.rel_mode == 2 ? (2 * p.KH - 1) + (2 * p.KW - 1) : p.KH + p.KW;
    return n | 1;
}

ULL_DEV uint4 scale_q8(const uint4& v, float sc) {
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= sc;
    return pack8(f);
}

// LDS-DMA of 64 x 16 B (see gemm_bf16.hip: issued via inline asm so hipcc does not drain it before the next ds_read).
// the same with a wave-uniform base and a 32-bit per-lane byte offset (saddr form): the per-lane part is computed once per kernel
ULL_DEV void glds16s(const void* sbase /* wave-uniform */, uint32_t voff, uint32_t lds_byte_addr /* wave-uniform */) {
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);   // make uniformity provable to the compiler
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
ULL_DEV void glds16(const void* gsrc, uint32_t lds_byte_addr /* wave-uniform */) {
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);   // make uniformity provable to the compiler
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// XOR swizzle of the 16-byte chunk index inside an LDS tile row (CPR chunks per row): chosen so that the 16 rows a
// ds_read_b128 lane group touches land on 16 different bank slots.
template <int CPR>
ULL_DEV int swz(int row) { return CPR >= 16 ? (row & 15) : CPR == 8 ? (row & 7) : ((row >> 2) & 3); }

// Head dim as a compile-time constant where the flavor pins it (flavor_of() checks the argument): the `ks * 32 < hd` /
// `ds * 16 < hd` tests that skip pure-padding MFMAs then fold away.  With a run-time hd every MFMA sits in its own basic
// block behind an s_waitcnt (seen in the ISA of the first version of these kernels).
template <int HDP, int FL>
ULL_DEV int head_dim_of(const AttnArgs& p) {
    if constexpr (FL == FL_LLAMA || FL == FL_CLIP) return HDP;
    else if constexpr (FL == FL_SAM_ENC && HDP == 128) return 80;
    else return p.hd;
}

// Block = NWV waves = 16*NWV queries of one (batch, head); wave w owns queries q0+16w .. +16 against ALL keys.
// (NWV = 4 lets two blocks share a CU so one block's barrier / DMA waits overlap the other's MFMAs.)
//   HDP : head dim padded to 32/64/128 (K-tile row = HDP bf16);  NT : max number of 64-key tiles held in registers.
// Per lane the whole score/probability row segment lives in registers as packed bf16 (8 VGPRs per 64 keys).
//   EXACT (SAM 14 x 14 windows): a non-causal call with exactly NT key tiles and ALL of them resident: the NT K tiles are
//   DMA'd up front, the NT V^T tiles right after the K barrier (they land during the score / softmax phases), so a block
//   passes 2 barriers instead of 2*NT and exposes two memory round trips instead of 2*NT -- the streaming form measured
//   46 us per block for ~10 us of work.  One block of NWV = 13 waves covers all 196 queries of a (window, head).
//   (The 14 x 14 windows on image-order tokens -- ull_sam_window_attention -- have their own kernel: sam_window_kernel below.)
// gfx950's transposing LDS read: within each group of 16 lanes, lane i passes the address of 4 consecutive 16-bit elements -- row i / 4,
// columns 4 * (i % 4) .. +3 of a 4 x 16 block -- and receives column i of the block (rows 0..3).  (Probed: tools/debug/tr_read_probe.hip.)
// The compiler does not know this is an LDS load: lds_tr_wait() below must sit between the reads and their first use.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF>
ULL_DEV u32x2_t lds_tr_b64(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N, int LEFT = 0>     // LEFT: LDS operations issued AFTER these reads that may still be in flight
ULL_DEV void lds_tr_wait(u32x2_t (&a)[N], u32x2_t (&b)[N]) {      // ties the values to the wait so that no use can move above it
    static_assert(N == 1 || N == 4 || N == 5 || N == 8, "head-dim blocks of hd = 64 / 80 / 128");
    if constexpr (N == 1)
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(LEFT) : "memory");
    else if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                     : "n"(LEFT) : "memory");
    else if constexpr (N == 5)
        asm volatile("s_waitcnt lgkmcnt(%10)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4])
                     : "n"(LEFT) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%16)"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]),
                       "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                     : "n"(LEFT) : "memory");
}

// tile buffers of attn_reg_kernel: EXACT keeps every K and V tile; the prefill form of LLaMA / CLIP (VROW, every wave issuing the same number
// of DMA pieces per tile) streams through a ring of ULL_ATTN_RING (2: measured best); everything else through two
template <int HDP, int NT, int NWV, bool EXACT, bool VROW>
constexpr int attn_reg_nbuf() {
    constexpr int ULL_ATTN_RING = 2;
    return EXACT ? 2 * NT : ((VROW && (HDP / 8) % NWV == 0 && NT >= 2) ? ULL_ATTN_RING : 2);
}

//   VROW (LLaMA / CLIP prefill): the V tiles are DMA'd ROW-major from V itself ([64 keys][head dim], like the K tiles) and the V^T
//   operand of P*V comes out of them through ds_read_b64_tr_b16: no V^T pass in front of the attention.
template <int HDP, int NT, int FL, int NWV, bool EXACT = false, bool VROW = false>
__global__ __launch_bounds__(NWV * 64, (EXACT || (VROW && HDP == 64 && NWV == 4)) ? 4 : (VROW && HDP == 128 && NT <= 11 && NWV <= 4) ? 3 :
                                       2)
void attn_reg_kernel(AttnArgs p) {                               // (LLaMA prefill: three blocks per CU = at most 168 registers)
    extern __shared__ __attribute__((aligned(256))) char smem[];     // (256: the V fragment addresses below XOR bits 5..7)
    static_assert(!VROW || ((FL == FL_LLAMA || FL == FL_CLIP) && HDP >= 64), "VROW: flavors that pin hd = HDP");
    constexpr int PM = HDP / 16 >= 8 ? 7 : HDP / 16 - 1;      // VROW: XOR mask of the 32-byte pair index (pairs per row - 1, at most 7)
    constexpr int BQ = 16 * NWV;
    constexpr int CPR = HDP / 8;          // 16-byte chunks per K-tile row
    constexpr int KROW = HDP * 2;         // K-tile row bytes
    constexpr int NKS = HDP / 32, NDS = HDP / 16;
    constexpr int TILE = 64 * KROW > HDP * 128 ? 64 * KROW : HDP * 128;   // bytes per tile buffer (K: 64 x HDP, V^T: HDP x 64)
    const int tid = threadIdx.x, lane = tid & 63;
    const int hd = head_dim_of<HDP, FL>(p);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    // ---- block -> (head, query tile): all query tiles of a head run on ONE XCD (K/V stay in that L2) ----
    const int nq = (p.Sq + BQ - 1) / BQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = (slot / nq) * 8 + xcd;
    if (head >= p.B * p.H) return;
    const int qt = nq - 1 - slot % nq;    // longest (most keys under the causal mask) first
    const int b = head / p.H, h = head % p.H;
    const int q0 = qt * BQ;
    const int koff = p.Sk - p.Sq;

    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    constexpr int NBUF = attn_reg_nbuf<HDP, NT, NWV, EXACT, VROW>();   // EXACT: [K tiles 0..NT) [V^T tiles 0..NT); else a ring of tile buffers
    constexpr int PPW = (CPR + NWV - 1) / NWV;                   // DMA pieces a wave issues per streamed tile
    char* maskb = smem + NBUF * TILE;     // one byte per key: 1 attend, 0 masked (finfo.min), 2 out of range (-inf)
    elem_t* biasb = (elem_t*)(smem + NBUF * TILE + NT * KT);

    int kend = p.Sk;
    if (p.causal) kend = min(p.Sk, q0 + BQ + koff);
    if (kend < 1) kend = 1;
    const int nkt = EXACT ? NT : (kend + KT - 1) / KT;          // tiles this block streams
    int kend_w = p.Sk;
    if (p.causal) kend_w = min(p.Sk, q0 + wave * 16 + 16 + koff);
    int nkt_w = EXACT ? NT : (q0 + wave * 16 < p.Sq) ? max(1, (kend_w + KT - 1) / KT) : 0;   // tiles this wave computes on

    const elem_t* kbase = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
    const elem_t* vbase = p.Vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
    // VROW flavors (LLaMA / CLIP prefill: hd == HDP, token strides that fit 32 bits): the per-lane part of every DMA source address -- row
    // within the tile x token stride + swizzled chunk -- is the same for every tile and is computed ONCE; a tile then costs one scalar
    // base and PPW loads.  (Recomputed per piece it was ~15 vector instructions of 64-bit address arithmetic, a quarter of the
    // kernel's VALU work, which is what bounds it.)  Only the tile that holds the last key clamps its rows and takes the general path.
    constexpr bool FASTDMA = VROW && (HDP / 8) % NWV == 0;
    uint32_t kvo[FASTDMA ? 2 * PPW : 1];
    if constexpr (FASTDMA) {
#pragma unroll
        for (int i0 = 0; i0 < PPW; ++i0) {
            const int i = i0 * NWV + wave;
            const int row = i * (64 / CPR) + lane / CPR;
            const int ck = (lane % CPR) ^ swz<CPR>(row);
            const int cpos = lane % CPR;
            const int cv = ((((cpos >> 1) ^ (row & PM)) << 1) | (cpos & 1));
            kvo[i0] = (uint32_t)(((long)row * p.k_ss + ck * 8) * 2);
            kvo[PPW + i0] = (uint32_t)(((long)row * p.vt_ds + cv * 8) * 2);
        }
    }
    const bool fast_dma_ok = FASTDMA && 64 * p.k_ss * 2 < (1L << 31) && 64 * p.vt_ds * 2 < (1L << 31);
    // stream step s: s < nkt -> K tile s ; else V^T tile s - nkt.   Buffer = s & 1.
    auto issue = [&](int s) {
        const uint32_t dst = lds_base + (EXACT ? s : (s % NBUF)) * TILE;
        if constexpr (FASTDMA) {
            const int kt = s < nkt ? s : s - nkt;
            if (fast_dma_ok && kt * KT + KT <= p.Sk) {              // every row of the tile is a real key: no clamp
                const elem_t* tb = s < nkt ? kbase + (long)kt * KT * p.k_ss : vbase + (long)kt * KT * p.vt_ds;
#pragma unroll
                for (int i0 = 0; i0 < PPW; ++i0) glds16s(tb, s < nkt ? kvo[i0] : kvo[PPW + i0], dst + (i0 * NWV + wave) * 1024);
                return;
            }
        }
        if (s < nkt) {
            const int kt = s;
#pragma unroll
            for (int i0 = 0; i0 < CPR; i0 += NWV) {
                const int i = i0 + wave;                        // one 1-KiB piece = 64/CPR rows
                if (i < CPR) {
                    const int row = i * (64 / CPR) + lane / CPR;
                    const int c = (lane % CPR) ^ swz<CPR>(row);
                    const int key = min(kt * KT + row, p.Sk - 1);
                    const elem_t* krow = kbase + (long)key * p.k_ss;
                    const elem_t* src = (c * 8 < hd) ? krow + c * 8 : p.zeros;
                    glds16(src, dst + i * 1024);
                }
            }
        } else if constexpr (VROW) {
            // V tile kt, row-major: [64 keys][KROW bytes]; 32-byte pairs of chunks XOR-swizzled with the row (the 16 rows one transposing read touches spread over all banks)
            const int kt = s - nkt;
#pragma unroll
            for (int i0 = 0; i0 < CPR; i0 += NWV) {
                const int i = i0 + wave;
                if (i < CPR) {
                    const int row = i * (64 / CPR) + lane / CPR;
                    const int cpos = lane % CPR;
                    const int c = ((((cpos >> 1) ^ (row & PM)) << 1) | (cpos & 1));
                    const int key = min(kt * KT + row, p.Sk - 1);          // rows past the last key: P is exactly 0 there
                    glds16(vbase + (long)key * p.vt_ds + c * 8, dst + i * 1024);
                }
            }
        } else {
            const int kt = s - nkt;
            const int npieces = hd >> 3;                      // 8 V^T rows (head dims) per 1-KiB piece
#pragma unroll
            for (int i0 = 0; i0 < HDP / 8; i0 += NWV) {
                const int i = i0 + wave;
                if (i < npieces) {
                    const int row = i * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ (row & 7);
                    glds16(vbase + (long)row * p.vt_ds + kt * KT + c * 8, dst + i * 1024);
                }
            }
        }
    };

    if constexpr (EXACT) {                // all K tiles in flight while Q / mask / bias tables are prepared
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) issue(kt);
    } else {
        // the ring runs NBUF - 1 tiles ahead; its first tiles are requested BEFORE the Q rows so that a block pays one memory latency at its
        // start, not two in a row (a block lives for ~12 tile steps: the start-up is a visible share of it)
#pragma unroll
        for (int s0 = 0; s0 < NBUF - 1; ++s0)
            if (s0 < 2 * nkt) issue(s0);
    }

    // ---- Q fragments + key-mask bytes (ordinary loads, drained together with the first DMA'd tiles) -------
    uint4 qf[NKS];
    const int qi = q0 + wave * 16 + fr;
    {
        const elem_t* qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qi * p.q_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + fg * 8;
            qf[ks] = (qi < p.Sq && d < hd) ? *(const uint4*)(qp + d) : make_uint4(0, 0, 0, 0);
        }
        for (int j = tid; j < nkt * KT; j += NWV * 64) {
            unsigned char m = 2;
            if (j < p.Sk) m = (p.key_mask == nullptr || p.key_mask[(long)b * p.Sk + j] != 0) ? 1 : 0;
            maskb[j] = m;
        }
    }
    const elem_t* brow = nullptr;
    int bh_off = 0, bw_off = 0;
    if (p.rel_h != nullptr) {
        const int bp = bias_pitch(p);
        stage_rel_bias<NKS>(p, biasb + wave * 16 * bp, bp, qf, q0 + wave * 16, head, lane, bh_off, bw_off, hd);
        brow = biasb + (wave * 16 + fr) * bp;
    }
    if (p.q_scale != 1.0f) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = scale_q8(qf[ks], p.q_scale);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // The compiler does not see the wait above (nor the DMA's share of vmcnt): left alone it guards the first use of the Q fragments in
    // EVERY tile (the tiles sit behind run-time branches) with its own s_waitcnt vmcnt(0) -- which also waits for the tile that was
    // just requested, i.e. no DMA ever overlapped with the MFMAs.  Redefining the fragments here ends its bookkeeping of those loads.
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks].x), "+v"(qf[ks].y), "+v"(qf[ks].z), "+v"(qf[ks].w));
    float mrow = -INFINITY;               // running maximum of this lane's scores
    ULL_ATT_ACC(0, stamp_t);

    uint32_t sp[NT][8];                   // [tile][2*ns + half]: bf16 pairs for keys kt*64 + ns*16 + 4*fg + {0,1 | 2,3}
    // LDS fragment addressing, per lane and ONCE: the K fragment of k-step ks sits at kfo[ks] inside rows fr, fr + 16, ... of a tile (the
    // swizzle term of row 16 * ns + fr does not depend on ns), so tile, buffer and ns are immediate offsets of the ds_read; the V fragment
    // of head-dim block d sits at (tile base + vfo) ^ (d << 5).  Recomputed per read these were 3 vector instructions each, 48 + 48 per
    // key tile of a kernel that is bound by its vector-issue slots.
    uint32_t kfo[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        kfo[ks] = fr * KROW + (((ks * 4 + fg) ^ swz<CPR>(fr)) << 4);
        asm volatile("" : "+v"(kfo[ks]));
    }
    uint32_t vfo = (4 * fg + (fr >> 2)) * KROW + ((fr & 2) << 3) + ((fr & 1) << 3) + (((4 * (fg & 1) + (fr >> 2)) & PM) << 5);
    asm volatile("" : "+v"(vfo));
    if constexpr (EXACT) {
        __builtin_amdgcn_s_barrier();     // (vmcnt(0) above) every K tile, the mask bytes and the bias rows are in LDS
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) issue(NT + kt);          // V^T tiles land during phases 1 and 2
    }
    // Streamed tiles (not EXACT): step s = K tile s, then V tile s - nkt, in buffer s % NBUF.  Before step s is consumed: wait until only
    // the pieces of the (at most NBUF - 2) later steps are still in flight, barrier (tile s visible to every wave, buffer (s - 1) % NBUF
    // free), then issue step s + NBUF - 1 into that buffer.  Two buffers = one tile ahead is what ships: rings of three and four tiles
    // (-DULL_ATTN_RING) measured 362 / 433 us against 364 on the LLaMA C4 shape (tools/attn_prefill_bench.py) -- the DMA latency is
    // already covered by the four blocks a CU holds at 34 KB of LDS each, and a fourth buffer costs two of them.
    auto stream_step = [&](int s) {
        const int total = 2 * nkt;
        const int later = min(NBUF - 2, total - 1 - s);
        if (NBUF >= 4 && later >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
        else if (NBUF >= 3 && later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (timing-only ablations, wrong results: -DULL_ATTN_ABL_NO_BARRIER 302 -> 296 us, -DULL_ATTN_ABL_NO_DMA 308 -> 243 us on the C4 shape: the
        //  step barriers cost 2 %, REQUESTING the tiles 21 % -- profiles/r05_attn_prefill_census.txt)
        __builtin_amdgcn_s_barrier();
        if (s + NBUF - 1 < total) issue(s + NBUF - 1);
    };

    // ---- phase 1: S = bf16(K Q^T) (+scale, +mask), kept in registers -----------------------------------
#pragma clang loop unroll(full)
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
            if constexpr (!EXACT) stream_step(kt);
            ULL_ATT_ACC(1, stamp_t);
            if (kt < nkt_w) {
                const char* tb = smem + (EXACT ? kt : (kt % NBUF)) * TILE;
                if constexpr (VROW) {
                    // a whole tile of real keys, no key mask, entirely below the diagonal for each of the wave's 16 queries (all but the last
                    // tile or two of a wave): ONE wave-uniform branch per tile, then straight-line code -- the four 16-key groups' accumulator
                    // chains interleave (no MFMA waits for the one before it) and the four score epilogues follow without look-ups
                    if (p.key_mask == nullptr && kt * KT + KT <= p.Sk && (FL != FL_LLAMA || kt * KT + KT - 1 <= q0 + wave * 16 + koff)) {
constexpr int ULL_ATTN_CHAINS = 2;               // accumulator chains in flight (4: 12 more registers, spills at the 168 of three blocks per CU)
#pragma unroll
                        for (int n0 = 0; n0 < 4; n0 += ULL_ATTN_CHAINS) {
                            f32x4_t accn[ULL_ATTN_CHAINS];
#pragma unroll
                            for (int c = 0; c < ULL_ATTN_CHAINS; ++c) accn[c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                                for (int c = 0; c < ULL_ATTN_CHAINS; ++c) {
                                    const uint4 kf = *(const uint4*)(tb + (n0 + c) * 16 * KROW + kfo[ks]);
                                    accn[c] = mfma16(kf, qf[ks], accn[c]);
                                }
#pragma unroll
                            for (int c = 0; c < ULL_ATTN_CHAINS; ++c)
                                score_quad_clean<FL>(p, accn[c], sp[kt][(n0 + c) * 2], sp[kt][(n0 + c) * 2 + 1], &mrow);
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int ns = 0; ns < 4; ++ns) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                    const int row = ns * 16 + fr;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        if (ks * 32 < hd) {                    // k-steps that are pure head-dim padding are skipped
                            const uint4 kf = *(const uint4*)(tb + ns * 16 * KROW + kfo[ks]);
                            acc = mfma16(kf, qf[ks], acc);
                        }
                    }
                    if constexpr (FL == FL_LLAMA || FL == FL_CLIP) {
                        // a whole tile of real keys, no key mask, entirely below the diagonal for each of the wave's queries: nothing to look up
                        if (p.key_mask == nullptr && kt * KT + KT <= p.Sk && (FL != FL_LLAMA || kt * KT + KT - 1 <= q0 + wave * 16 + koff)) {
                            score_quad_clean<FL>(p, acc, sp[kt][ns * 2], sp[kt][ns * 2 + 1], &mrow);
                            continue;
                        }
                    }
                    const uint32_t mk = *(const uint32_t*)(maskb + kt * KT + ns * 16 + fg * 4);
                    const int j0 = kt * KT + ns * 16 + fg * 4;
                    bool fast = false;
                    if constexpr (FL == FL_LLAMA || FL == FL_CLIP) {
                        const bool dirty = mk != 0x01010101u || (FL == FL_LLAMA && j0 + 3 > qi + koff);
                        fast = __builtin_amdgcn_ballot_w64(dirty) == 0;     // wave-uniform
                    }
                    // the row maximum is taken here, on the fp32 copies of the 16-bit scores (phase 2 would unpack them again)
                    if (fast) score_quad_clean<FL>(p, acc, sp[kt][ns * 2], sp[kt][ns * 2 + 1], &mrow);
                    else score_quad<FL>(p, acc, j0, mk, qi, koff, brow, bh_off, bw_off, sp[kt][ns * 2], sp[kt][ns * 2 + 1], &mrow);
                    if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- phase 2: exact fp32 row softmax over the bf16 scores, P = bf16(softmax) (registers only) --------
    {
        float m = rnd(mrow);                  // (score_quad_clean feeds it unrounded products: see the note there)
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma clang loop unroll(full)
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt_w) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    // __expf(x) = v_exp_f32(x * log2(e)), written out so that the subtraction and the multiply pair up (v_pk_*_f32)
                    const f32x2_t t = (f32x2_t{pk_lo(sp[kt][i]), pk_hi(sp[kt][i])} - m) * 1.4426950408889634f;
                    sum += __builtin_amdgcn_exp2f(t.x);
                    sum += __builtin_amdgcn_exp2f(t.y);
                }
                if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma clang loop unroll(full)
        for (int kt = 0; kt < NT; ++kt)
            if (kt < nkt_w) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2_t t = (f32x2_t{pk_lo(sp[kt][i]), pk_hi(sp[kt][i])} - m) * 1.4426950408889634f;
                    const f32x2_t e = f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} * inv;
                    sp[kt][i] = pack2e(e.x, e.y);
                }
                if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
            }
    }

    // ---- phase 3: O^T = V^T P^T; P feeds the MFMA B operand straight from registers ---------------------
    // V^T tiles are stored with keys permuted inside every 32-key block (slot 8g+4a+r <- key 16a+4g+r, see
    // transpose_v_kernel) so that the k-slot <-> key map of the A operand equals the one the P registers already have.
    f32x4_t oacc[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) oacc[ds] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (EXACT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma clang loop unroll(full)
    for (int kt = 0; kt < NT; ++kt) {
        if (kt < nkt) {
            if constexpr (!EXACT) stream_step(nkt + kt);
            ULL_ATT_ACC(4, stamp_t);
            if constexpr (VROW) {
                if (kt < nkt_w) {
                    const uint32_t vb = lds_base + (EXACT ? NT + kt : ((nkt + kt) % NBUF)) * TILE + vfo;
                    // four head-dim blocks at a time (8 reads in flight, 16 registers): more would cost the third wave per SIMD
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const uint4 pf = make_uint4(sp[kt][4 * kk], sp[kt][4 * kk + 1], sp[kt][4 * kk + 2], sp[kt][4 * kk + 3]);
#pragma unroll
                        for (int d0 = 0; d0 < NDS; d0 += 4) {
                            u32x2_t va[4], vc[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t ad = vb ^ ((d0 + j) << 5);
                                if (kk == 0) { va[j] = lds_tr_b64<0>(ad); vc[j] = lds_tr_b64<16 * KROW>(ad); }
                                else { va[j] = lds_tr_b64<32 * KROW>(ad); vc[j] = lds_tr_b64<48 * KROW>(ad); }
                            }
                            lds_tr_wait<4>(va, vc);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                oacc[d0 + j] = mfma16(make_uint4(va[j].x, va[j].y, vc[j].x, vc[j].y), pf, oacc[d0 + j]);
                        }
                    }
                }
            } else
            if (kt < nkt_w) {
                const char* tb = smem + (EXACT ? NT + kt : ((nkt + kt) % NBUF)) * TILE;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const uint4 pf = make_uint4(sp[kt][4 * kk], sp[kt][4 * kk + 1], sp[kt][4 * kk + 2], sp[kt][4 * kk + 3]);
#pragma unroll
                    for (int ds = 0; ds < NDS; ++ds) {
                        if (ds * 16 < hd) {
                            const int row = ds * 16 + fr;
                            const uint4 vf = *(const uint4*)(tb + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                            oacc[ds] = mfma16(vf, pf, oacc[ds]);
                        }
                    }
                    if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if constexpr (VROW && !EXACT && NBUF == 2 && NWV * 16 * KROW <= TILE) {    // (the block's 16 NWV rows must fit ONE ring buffer: NWV <= 4)
        // O through LDS, stored as whole rows.  A lane holds 4 head dims of ONE query per block of 16: stored from the registers that is 8-byte
        // pieces at a row stride (every store instruction touches 16 rows x 4 x 32 B).  Instead each wave writes its 16 x hd block into
        // the ring buffer nobody reads any more (the last step lives in the other one; every wave has passed that step's barrier) and
        // reads it back as 16-byte pieces of consecutive rows: CPR / 4 stores of 4 (hd 128) or 8 (hd 64) full rows each.
        char* ob = smem + ((2 * nkt - 2) % NBUF) * TILE + wave * 16 * KROW;
        constexpr int SM = CPR >= 16 ? 15 : CPR - 1;
        int ln = lane;                                    // (opaque: the per-lane addresses below are computed HERE, not hoisted to the
        asm volatile("" : "+v"(ln));                      //  kernel entry and kept -- or spilled -- through the three phases)
        const int er = ln & 15, eg = ln >> 4;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            uint2 pk;
            pk.x = pack2e(oacc[ds][0], oacc[ds][1]);
            pk.y = pack2e(oacc[ds][2], oacc[ds][3]);
            *(uint2*)(ob + er * KROW + ((((ds * 2 + (eg >> 1)) ^ (er & SM))) << 4) + ((eg & 1) << 3)) = pk;
        }
        constexpr int RPI = 64 / CPR;                     // rows per store instruction
        elem_t* ow = p.O + (long)b * p.o_bs + (long)h * p.o_hs;
#pragma unroll
        for (int it = 0; it < 16 / RPI; ++it) {
            const int r = it * RPI + ln / CPR, c = ln % CPR;
            const uint4 v = *(const uint4*)(ob + r * KROW + ((c ^ (r & SM)) << 4));
            const int q = q0 + wave * 16 + r;
            if (q < p.Sq) *(uint4*)(ow + (long)q * p.o_ss + c * 8) = v;
        }
    } else
    if (qi < p.Sq) {
        elem_t* op = p.O + (long)b * p.o_bs + (long)h * p.o_hs + (long)qi * p.o_ss;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            if (ds * 16 < hd) {
                uint2 pk;
                pk.x = pack2e(oacc[ds][0], oacc[ds][1]);
                pk.y = pack2e(oacc[ds][2], oacc[ds][3]);
                *(uint2*)(op + ds * 16 + fg * 4) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SAM 14 x 14 window attention (ull_sam_window_attention), one workgroup per CU walking a run of consecutive (window, head) items.
//
// The arithmetic is attn_reg_kernel's exact form (all 196 keys of a window resident: 4 K tiles + 4 V tiles of 64 slots, slot 16 kh + kw =
// window position (kh, kw), 13 waves of 16 queries, scores -> rel-pos bias -> exact fp32 softmax -> P*V from registers), statement for
// statement; what differs is everything around it.  That kernel ran one item per block -- 154 KB of LDS, so one block per CU -- and
// was bound by its instruction issue (~4100 instructions per wave and item, 13 waves on 4 SIMDs), 40 % of them a prologue that an item
// repeats for nothing: six integer divisions, the window -> token addressing of every DMA piece, the rel-pos tables fetched from memory.
// Here the per-lane pieces of all addressing are computed once per block, the two rel-pos tables sit in LDS for the block's life, and
// the NEXT item's prologue is spread under the current item's phases, still with two barriers per item and the same tile buffers:
//   * the K tiles of the next item are requested behind the barrier that ends everybody's scores (they land under softmax and P*V),
//     the V tiles behind the barrier that ends everybody's P*V (they land under the next scores);
//   * its Q rows are requested after the softmax into the registers of the current ones (dead since the scores) and land under P*V;
//   * a wave's bias rows in LDS are private to it: it rebuilds them for the next item once its P*V is stored.
// (Nothing may spill: a reload is a VMEM load, and the wait the compiler puts in front of its use also waits for every DMA in flight.)
// Items of a block are consecutive: mostly the 16 heads of one window (same token rows, neighbouring 160-byte columns, one XCD's L2).
// LDS bias row of a query: [rel_h products t = 0..26][pad][rel_w products t = 0..26][pad] = 56 elements: both halves start 8-byte aligned, so
// the four products a lane holds (t = 16 st + 4 fg + r) go out as one 64-bit write
constexpr int SW_NWV = 13, SW_NT = 4, SW_WS = 14, SW_HD = 80, SW_NTAB = 2 * SW_WS - 1, SW_BW = SW_NTAB + 1, SW_BP = 2 * SW_BW, SW_TILE = 64 * 256;
constexpr int SW_LDS = 2 * SW_NT * SW_TILE + SW_NWV * 16 * SW_BP * 2 + 2 * SW_NTAB * SW_HD * 2;      // 163 008 bytes
static_assert(SW_LDS <= 160 * 1024, "K + V tiles, 208 bias rows and the two rel-pos tables fit one CU's LDS");
__global__ __launch_bounds__(SW_NWV * 64) void sam_window_kernel(AttnArgs p, int n_items) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    constexpr int NWV = SW_NWV, NT = SW_NT, WS = SW_WS, NBLK = SW_WS, KROW = 256, TILE = SW_TILE, NKS = 3, HD = SW_HD, NTAB = SW_NTAB, BW = SW_BW, BP = SW_BP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    elem_t* biasb = (elem_t*)(smem + 2 * NT * TILE);
    elem_t* tabs = biasb + NWV * 16 * BP;             // [2][27][80]: rel_pos_h, rel_pos_w
    elem_t* brow = biasb + (wave * 16 + fr) * BP;     // this lane's query's bias row (the wave's 16 rows are private to it)

    int item = (int)((uint64_t)blockIdx.x * (uint64_t)n_items / gridDim.x);            // 64-bit: blockIdx * n_items passes 2^32 from 2^24 items on
    const int last = (int)((uint64_t)(blockIdx.x + 1) * (uint64_t)n_items / gridDim.x);
    if (item >= last) return;

    // ---- per-lane state ---------------------------------------------------------------------------------------------------------------
    // Registers are the scarce thing here (13 waves: 128 per lane, 56 of them the softmax's exponentials), so only what the inner loops
    // read stays resident -- the K / V fragment offsets and the lane's bias row.  Everything the per-item address arithmetic needs is
    // re-derived from the lane id where it is used (a few vector instructions per item), behind an opaque copy of the id so that the
    // compiler cannot hoist it back out of the item loop: hoisted, it spills, and a spill reload in the middle of the DMA requests waits
    // for all of them.
    const int qi = wave * 16 + fr;                    // this lane's query: window position (qi / 14, qi % 14); >= 196: none
    const int qy = min(qi, WS * WS - 1) / WS;
    const elem_t* brow_h = biasb + (wave * 16 + fr) * BP + qy + WS - 1;       // rel_h(kh) = brow_h[-kh]
    uint32_t kfo[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        kfo[ks] = fr * KROW + (((ks * 4 + fg) ^ (fr & 15)) << 4);
        asm volatile("" : "+v"(kfo[ks]));
    }
    // V fragment of head-dim block ds: (tile base + vfx) ^ (ds << 5) -- the 32-byte pair of chunks is XOR-swizzled with (slot & 7)
    uint32_t vfx = (lds_base + NT * TILE + (4 * fg + (fr >> 2)) * 256 + ((fr & 2) << 3) + ((fr & 1) << 3)) ^ ((4 * (fg & 1) + (fr >> 2)) << 5);
    asm volatile("" : "+v"(vfx));
    const uint32_t row_pitch = (uint32_t)p.img_w * (uint32_t)p.k_ss * 2u;      // bytes between window rows (dispatcher: 14 rows < 2^31)

    // ---- per-item scalar state ----------------------------------------------------------------------------------------------------------
    struct Item { int h, iy0, ix0; long tok0; };       // head, window origin, token index of the window's first position
    auto locate = [&](int it) {
        const int b = udiv_magic(it, p.mg_h);
        const WinOrigin wo = win_origin(p, b, WS);
        return Item{it - b * p.H, wo.iy0, wo.ix0, ((long)wo.img * p.img_h + wo.iy0) * p.img_w + wo.ix0};
    };
    // The wave's pieces of the four K (which = 0) or V (which = 1) tiles of an item.  One LDS-DMA piece = 1 KiB = 4 window columns x 16
    // chunks of ONE window row: piece i of tile kt holds window row 4 kt + i / 4, columns 4 (i % 4) + lane / 16; a wave's pieces are
    // i = wave and, in waves 0..2, i = wave + 13.  Everything a lane needs except the row is the same for the four tiles, so per tile
    // the source is one 64-bit add of a wave-uniform row offset and one select (per piece -- token index, its 64-bit multiply by the
    // row pitch, three nested selects -- it was ~40 instructions x 10 pieces in every wave).
    //   in  = this lane's chunk of window row 0 (used when row and column are inside the image and the chunk holds real head dims)
    //   out = its source otherwise: the padded position's row (= the qkv bias: what the reference's F.pad + Linear leaves there), or
    //         zeros for the chunks past the head dim
    auto issue_tiles = [&](const Item& t, int which) {
        int ln = tid;
        asm volatile("" : "+v"(ln));
        const char* in_base = (const char*)((which ? p.Vt : p.K) + t.tok0 * p.k_ss + (long)t.h * p.k_hs);
        const char* pad_base = (const char*)((which ? p.v_pad : p.k_pad) + (long)t.h * p.k_hs);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = wave + j * NWV;
            if (i < 16) {
                const int l4 = (ln >> 4) & 3, cpos = ln & 15, row = i * 4 + l4;
                const uint32_t c = 16u * (uint32_t)(which ? ((((cpos >> 1) ^ (row & 7)) << 1) | (cpos & 1)) : (cpos ^ (row & 15)));
                const int lx = min((i & 3) * 4 + l4, WS - 1);                     // slots kw = 14, 15 read a real key and are masked
                const bool real = c < HD * 2;                                     // chunks past the head dim are zeros
                const bool col_ok = real && t.ix0 + lx < p.img_w;
                const char* in = in_base + ((uint32_t)lx * (uint32_t)p.k_ss * 2u + c);
                const char* out = real ? pad_base + c : (const char*)p.zeros;   // outside the image: the padded position's row (the qkv bias)
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    if (kt * 64 + i * 4 >= NBLK * 16) continue;                   // slots past the last window row are never read
                    const int ly = kt * 4 + (i >> 2);                             // the piece's window row (wave-uniform)
                    const bool row_ok = t.iy0 + ly < p.img_h;
                    glds16((row_ok && col_ok) ? in + (uint32_t)ly * row_pitch : out, lds_base + (which * NT + kt) * TILE + i * 1024);
                }
            }
        }
    };
    // byte offset of this lane's query's row inside the window (pitch = the row pitch of Q or O in elements) and whether it is in the image
    auto query_pos = [&](const Item& t, int pitch, uint32_t& off, bool& ok) {
        int ln = tid;
        asm volatile("" : "+v"(ln));
        const int q = (ln >> 6) * 16 + (ln & 15), qc = min(q, WS * WS - 1), y = qc / WS, x = qc - y * WS;
        ok = q < WS * WS && t.iy0 + y < p.img_h && t.ix0 + x < p.img_w;
        off = (uint32_t)(y * p.img_w + x) * (uint32_t)pitch * 2u;
    };
    auto load_q = [&](const Item& t, uint4 (&q)[NKS], bool& ok) {
        uint32_t off;
        query_pos(t, (int)p.q_ss, off, ok);
        int ln = tid;
        asm volatile("" : "+v"(ln));
        const int g = (ln >> 4) & 3;
        const char* qb = (const char*)(p.Q + t.tok0 * p.q_ss + (long)t.h * p.q_hs) + (off + g * 16);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            q[ks] = (ok && ks * 32 + g * 8 < HD) ? *(const uint4*)(qb + ks * 64) : make_uint4(0, 0, 0, 0);
    };
    float wv[4];
    // unscaled q -> the wave's bias rows in LDS (stage_rel_bias's rel_mode-2 product with the table fragments read from LDS, one
    // (table, 16-row step) at a time: same fragments, MFMAs and rounding), then the scaled q and this lane's four rel_w values
    auto finish_prologue = [&](uint4 (&q)[NKS]) {
        int ln = tid;
        asm volatile("" : "+v"(ln));
        const int fr = ln & 15, fg = (ln >> 4) & 3;   // (shadow the block-level copies: see "per-lane state")
        elem_t* dst = biasb + wave * 16 * BP;
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int tr = min(st * 16 + fr, NTAB - 1);
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const int d = ks * 32 + fg * 8;
                    const uint4 tf = (d < HD) ? *(const uint4*)(tabs + (which * NTAB + tr) * HD + d) : make_uint4(0, 0, 0, 0);
                    acc = mfma16(tf, q[ks], acc);
                }
                // acc[r] = G[t = 16 st + 4 fg + r][query fr]; t = 27 lands in the half's pad element, t >= 28 does not exist
                if (st == 0 || fg < 3) {
                    uint2 pk;
                    pk.x = pack2e(acc[0], acc[1]);
                    pk.y = pack2e(acc[2], acc[3]);
                    *(uint2*)(dst + fr * BP + which * BW + st * 16 + fg * 4) = pk;
                }
            }
        if (p.q_scale != 1.0f) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {        // scale_q8, two values per multiply
                uint32_t* w = (uint32_t*)&q[ks];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_t v = f32x2_t{pk_lo(w[j]), pk_hi(w[j])} * p.q_scale;
                    w[j] = pack2e(v.x, v.y);
                }
            }
        }
        const elem_t* brow_w = brow_h - qy + BW + (min(qi, WS * WS - 1) - qy * WS);         // row + 28 + qx + 13: rel_w(kw) = brow_w[-kw]
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[r] = e2f(*(brow_w - min(fg * 4 + r, WS - 1)));
        if (fg == 3) wv[2] = wv[3] = -INFINITY;       // kw = 14, 15: padding slots
    };

    // ---- the block's first item: its whole prologue is exposed ------------------------------------------------------------------------
    Item cur = locate(item);
    uint4 qf[NKS];
    bool q_ok;
    issue_tiles(cur, 0);
    load_q(cur, qf, q_ok);
    for (int i = tid; i < 2 * NTAB * HD / 8; i += NWV * 64)
        ((uint4*)tabs)[i] = ((const uint4*)(i < NTAB * HD / 8 ? p.rel_h : p.rel_w))[i < NTAB * HD / 8 ? i : i - NTAB * HD / 8];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (the compiler does not see that wait: redefining the fragments ends its own bookkeeping of the loads, see attn_reg_kernel)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks].x), "+v"(qf[ks].y), "+v"(qf[ks].z), "+v"(qf[ks].w));
    __syncthreads();                                  // tables and every K piece are in LDS
    issue_tiles(cur, 1);                              // V tiles land during the scores
    finish_prologue(qf);

    while (true) {
        const bool has_next = item + 1 < last;
        const bool live = __builtin_amdgcn_ballot_w64(q_ok) != 0;             // 16 padding positions: barriers and DMA only

        // ---- phase 1: scores of window row kh = 4 kt + ns, this lane's columns kw = 4 fg + r --------------------------------------------
        uint32_t sp[NT][8];
        float mrow = -INFINITY;
        if (live) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const char* tb = smem + kt * TILE;
#pragma unroll
                for (int ns = 0; ns < 4; ++ns) {
                    if (kt * 4 + ns >= NBLK) continue;
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const uint4 kf = *(const uint4*)(tb + ns * 16 * KROW + kfo[ks]);
                        acc = mfma16(kf, qf[ks], acc);
                    }
                    const float hb = e2f(*(brow_h - (kt * 4 + ns)));
                    score_quad_win(acc, hb, f32x2_t{wv[0], wv[1]}, f32x2_t{wv[2], wv[3]}, sp[kt][ns * 2], sp[kt][ns * 2 + 1], mrow);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's V pieces, requested a phase ago
        __builtin_amdgcn_s_barrier();                 // V tiles visible; every wave has left the scores: the K tiles are free
        Item nxt = cur;
        if (has_next) {
            if (cur.h + 1 < p.H) nxt.h = cur.h + 1;   // same window, next head
            else nxt = locate(item + 1);
            issue_tiles(nxt, 0);                      // the next item's K tiles land under the softmax, P*V and its prologue
        }
        // ---- phase 2: exact fp32 softmax over the 196 scores of the row (registers only) ------------------------------------------------
        if (live) softmax_win(sp, mrow);
        __builtin_amdgcn_sched_barrier(0);
        const bool q_ok_cur = q_ok;
        if (has_next) load_q(nxt, qf, q_ok);          // (the current Q fragments had their last use in phase 1) lands under P*V

        // ---- phase 3: O^T = V^T P^T ------------------------------------------------------------------------------------------------------
        f32x4_t oacc[5];
#pragma unroll
        for (int ds = 0; ds < 5; ++ds) oacc[ds] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (live) {
            uint32_t vad[5];                          // re-derived per item: kept across the loop they would be spilled
            {
                uint32_t vf = vfx;
                asm volatile("" : "+v"(vf));
#pragma unroll
                for (int ds = 0; ds < 5; ++ds) vad[ds] = vf ^ (ds << 5);
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if (2 * (kt * 2 + kk) >= NBLK) continue;
                    const uint4 pf = make_uint4(sp[kt][4 * kk], sp[kt][4 * kk + 1], sp[kt][4 * kk + 2], sp[kt][4 * kk + 3]);
                    u32x2_t va[5], vc[5];
#pragma unroll
                    for (int ds = 0; ds < 5; ++ds) {                          // (tile and slot block are immediate offsets of the read)
                        if (kt == 0) { if (kk == 0) { va[ds] = lds_tr_b64<0>(vad[ds]); vc[ds] = lds_tr_b64<16 * 256>(vad[ds]); }
                                       else { va[ds] = lds_tr_b64<32 * 256>(vad[ds]); vc[ds] = lds_tr_b64<48 * 256>(vad[ds]); } }
                        if (kt == 1) { if (kk == 0) { va[ds] = lds_tr_b64<TILE>(vad[ds]); vc[ds] = lds_tr_b64<TILE + 16 * 256>(vad[ds]); }
                                       else { va[ds] = lds_tr_b64<TILE + 32 * 256>(vad[ds]); vc[ds] = lds_tr_b64<TILE + 48 * 256>(vad[ds]); } }
                        if (kt == 2) { if (kk == 0) { va[ds] = lds_tr_b64<2 * TILE>(vad[ds]); vc[ds] = lds_tr_b64<2 * TILE + 16 * 256>(vad[ds]); }
                                       else { va[ds] = lds_tr_b64<2 * TILE + 32 * 256>(vad[ds]); vc[ds] = lds_tr_b64<2 * TILE + 48 * 256>(vad[ds]); } }
                        if (kt == 3) { va[ds] = lds_tr_b64<3 * TILE>(vad[ds]); vc[ds] = lds_tr_b64<3 * TILE + 16 * 256>(vad[ds]); }
                    }
                    lds_tr_wait(va, vc);              // (the K pieces in flight are VMEM: these reads are counted by lgkmcnt alone)
#pragma unroll
                    for (int ds = 0; ds < 5; ++ds)
                        oacc[ds] = mfma16(make_uint4(va[ds].x, va[ds].y, vc[ds].x, vc[ds].y), pf, oacc[ds]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (q_ok_cur) {
            uint32_t o_off;
            bool unused;
            query_pos(cur, (int)p.o_ss, o_off, unused);
            int ln = tid;
            asm volatile("" : "+v"(ln));
            char* ob = (char*)(p.O + cur.tok0 * p.o_ss + (long)cur.h * p.o_hs) + (o_off + ((ln >> 4) & 3) * 8);
#pragma unroll
            for (int ds = 0; ds < 5; ++ds) {
                uint2 pk;
                pk.x = pack2e(oacc[ds][0], oacc[ds][1]);
                pk.y = pack2e(oacc[ds][2], oacc[ds][3]);
                *(uint2*)(ob + ds * 32) = pk;
            }
        }
        if (!has_next) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next Q rows, this wave's pieces of the next K tiles, its stores
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks].x), "+v"(qf[ks].y), "+v"(qf[ks].z), "+v"(qf[ks].w));
        finish_prologue(qf);                          // this wave's bias rows and wv are its own: their last use was phase 1
        __builtin_amdgcn_s_barrier();                 // next K tiles visible; every wave has left P*V: the V tiles are free
        issue_tiles(nxt, 1);
        cur = nxt; ++item;
    }
}

// ---------------------------------------------------------------------------------------------
// Any number of keys, exact form (mask-decoder token->image cross attention: 4096 keys; any other > 1024-key call; the SAM
// encoder's global attention takes the single-pass kernel further down unless ULL_ATTN_TWO_PASS is set).
// Same rounding points as above, but the score row no longer fits in registers, so the kernel streams the keys TWICE:
//   pass 1  S tiles -> running row max and sum of exp (fp32; combined across the 4 lanes of a query at the end);
//   pass 2  the same S tiles again (bit-identical), P = bf16(exp(S - max) / sum) straight into the P*V MFMA.
// Nothing but O is written; K is read twice from L2 instead of S/P being materialised in HBM like the reference does.
template <int HDP, int FL>
__global__ __launch_bounds__(512) void attn_long_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWV = 8, BQ = 16 * NWV;
    constexpr int CPR = HDP / 8, KROW = HDP * 2;
    constexpr int NKS = HDP / 32, NDS = HDP / 16;
    constexpr int TILE = 64 * KROW;       // == HDP * 128: a K tile (64 x HDP) and a V^T tile (HDP x 64) have the same size
    const int tid = threadIdx.x, lane = tid & 63;
    const int hd = head_dim_of<HDP, FL>(p);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nq = (p.Sq + BQ - 1) / BQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = (slot / nq) * 8 + xcd;
    if (head >= p.B * p.H) return;
    const int qt = nq - 1 - slot % nq;
    const int b = head / p.H, h = head % p.H;
    const int q0 = qt * BQ;
    const int koff = p.Sk - p.Sq;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    int kend = p.Sk;
    if (p.causal) kend = min(p.Sk, q0 + BQ + koff);
    if (kend < 1) kend = 1;
    const int nkt = (kend + KT - 1) / KT;
    char* maskb = smem + 4 * TILE;
    elem_t* biasb = (elem_t*)(smem + 4 * TILE + ((nkt * KT + 15) & ~15));

    uint4 qf[NKS];
    const int qi = q0 + wave * 16 + fr;
    {
        const elem_t* qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qi * p.q_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + fg * 8;
            qf[ks] = (qi < p.Sq && d < hd) ? *(const uint4*)(qp + d) : make_uint4(0, 0, 0, 0);
        }
        for (int j = tid; j < nkt * KT; j += 512) {
            unsigned char m = 2;
            if (j < p.Sk) m = (p.key_mask == nullptr || p.key_mask[(long)b * p.Sk + j] != 0) ? 1 : 0;
            maskb[j] = m;
        }
    }
    const elem_t* brow = nullptr;
    int bh_off = 0, bw_off = 0;
    if (p.rel_h != nullptr) {
        const int bp = bias_pitch(p);
        stage_rel_bias<NKS>(p, biasb + wave * 16 * bp, bp, qf, q0 + wave * 16, head, lane, bh_off, bw_off, hd);
        brow = biasb + (wave * 16 + fr) * bp;
    }
    if (p.q_scale != 1.0f) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = scale_q8(qf[ks], p.q_scale);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const elem_t* kbase = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
    const elem_t* vbase = p.Vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
    // stream step s: s < nkt -> K tile s (pass 1); else K tile + V^T tile s - nkt (pass 2).  Slot = s & 1, [K | V^T].
    auto issue = [&](int s) {
        if (s >= 2 * nkt) return;
        const uint32_t dst = lds_base + (s & 1) * (2 * TILE);
        const int kt = s < nkt ? s : s - nkt;
#pragma unroll
        for (int i0 = 0; i0 < CPR; i0 += NWV) {
            const int i = i0 + wave;
            if (i < CPR) {
                const int row = i * (64 / CPR) + lane / CPR;
                const int c = (lane % CPR) ^ swz<CPR>(row);
                const int key = min(kt * KT + row, p.Sk - 1);
                const elem_t* src = (c * 8 < hd) ? kbase + (long)key * p.k_ss + c * 8 : p.zeros;
                glds16(src, dst + i * 1024);
            }
        }
        if (s >= nkt) {
            const int npieces = hd >> 3;
#pragma unroll
            for (int i0 = 0; i0 < HDP / 8; i0 += NWV) {
                const int i = i0 + wave;
                if (i < npieces) {
                    const int row = i * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ (row & 7);
                    glds16(vbase + (long)row * p.vt_ds + kt * KT + c * 8, dst + TILE + i * 1024);
                }
            }
        }
    };
    // scores of one 64-key tile for this lane's query: sq[2*ns + half]
    auto scores = [&](const char* tb, int kt, uint32_t (&sq)[8]) {
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const int row = ns * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks * 32 < hd) {
                    const uint4 kf = *(const uint4*)(tb + row * KROW + (((ks * 4 + fg) ^ swz<CPR>(row)) << 4));
                    acc = mfma16(kf, qf[ks], acc);
                }
            }
            const uint32_t mk = *(const uint32_t*)(maskb + kt * KT + ns * 16 + fg * 4);
            score_quad<FL>(p, acc, kt * KT + ns * 16 + fg * 4, mk, qi, koff, brow, bh_off, bw_off, sq[ns * 2], sq[ns * 2 + 1]);
        }
    };

    issue(0);
    float m = -INFINITY, l = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {                       // ---- pass 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(kt + 1);
        uint32_t sq[8];
        scores(smem + (kt & 1) * (2 * TILE), kt, sq);
        float tm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            tm = fmaxf(tm, pk_lo(sq[i]));
            tm = fmaxf(tm, pk_hi(sq[i]));
        }
        if (tm > m) { l *= __expf(m - tm); m = tm; }
        if (m > -INFINITY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                l += __expf(pk_lo(sq[i]) - m);
                l += __expf(pk_hi(sq[i]) - m);
            }
        }
    }
    {   // combine the 4 lanes (fg = 0..3) that share a query
        float mo = __shfl_xor(m, 16, 64), lo = __shfl_xor(l, 16, 64);
        float mn = fmaxf(m, mo);
        l = (m > -INFINITY ? l * __expf(m - mn) : 0.f) + (mo > -INFINITY ? lo * __expf(mo - mn) : 0.f);
        m = mn;
        mo = __shfl_xor(m, 32, 64); lo = __shfl_xor(l, 32, 64);
        mn = fmaxf(m, mo);
        l = (m > -INFINITY ? l * __expf(m - mn) : 0.f) + (mo > -INFINITY ? lo * __expf(mo - mn) : 0.f);
        m = mn;
    }
    const float inv = 1.0f / l;

    f32x4_t oacc[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) oacc[ds] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {                       // ---- pass 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(nkt + kt + 1);
        const char* tb = smem + ((nkt + kt) & 1) * (2 * TILE);
        uint32_t sq[8];
        scores(tb, kt, sq);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float lo = __expf(pk_lo(sq[i]) - m) * inv;
            const float hi = __expf(pk_hi(sq[i]) - m) * inv;
            sq[i] = pack2e(lo, hi);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint4 pf = make_uint4(sq[4 * kk], sq[4 * kk + 1], sq[4 * kk + 2], sq[4 * kk + 3]);
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                if (ds * 16 < hd) {
                    const int row = ds * 16 + fr;
                    const uint4 vf = *(const uint4*)(tb + TILE + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                    oacc[ds] = mfma16(vf, pf, oacc[ds]);
                }
            }
        }
    }
    if (qi < p.Sq) {
        elem_t* op = p.O + (long)b * p.o_bs + (long)h * p.o_hs + (long)qi * p.o_ss;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            if (ds * 16 < hd) {
                uint2 pk;
                pk.x = pack2e(oacc[ds][0], oacc[ds][1]);
                pk.y = pack2e(oacc[ds][2], oacc[ds][3]);
                *(uint2*)(op + ds * 16 + fg * 4) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Single-pass form of the long-key kernel: one sweep over the keys with a running row max (the "online" softmax).
// Per 64-key tile: S tile (same rounding points as everywhere else: bf16 scores, bf16 bias adds), tile max combined over the 4
// lanes of a query, P = bf16(exp(S - m)) into the P*V MFMA, O rescaled by exp(m_old - m_new) only when some query of the
// wave saw a new maximum; O / sum(exp) at the end.  Against the two-pass form it drops the second QK^T and the second
// round of score arithmetic -- the kernel is VALU-bound (~17 VALU-instruction equivalents per score against 22 MFMAs per
// tile), not MFMA-bound -- at the price of rounding P before instead of after the normalisation (same 2^-9 relative error
// per probability as the reference's bf16 softmax, but not the identical bits).
//   HOIST (SAM global attention, 64 x 64 key grid, KW == 64 == tile width, per-query tables from relpos_kernel): a key tile
//   is one grid row, so rel_h is ONE value per (query, tile) and the 16 rel_w values of a lane are the same for every tile:
//   they live in registers and the per-score table lookups / index arithmetic / LDS bias rows disappear; without the mask
//   bytes and bias rows a block needs 64 KiB of LDS and two blocks share a CU.
//   VROW: V is handed over as rows (AttnArgs::v_rows); the V tile is DMA'd row-major and read through ds_read_b64_tr_b16 (see
//   attn_reg_kernel): the global SAM blocks then run without a V^T pass as well.
template <int HDP, int FL, int HOIST, bool VROW = false>
__global__ __launch_bounds__(512, 4) void attn_stream_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NWV = 8, BQ = 16 * NWV;
    constexpr int CPR = HDP / 8, KROW = HDP * 2;
    constexpr int NKS = HDP / 32, NDS = HDP / 16;
    constexpr int TILE = 64 * KROW;       // a K tile (64 x HDP) and a V^T tile (HDP x 64) have the same size
    const int tid = threadIdx.x, lane = tid & 63;
    const int hd = head_dim_of<HDP, FL>(p);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nq = (p.Sq + BQ - 1) / BQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = (slot / nq) * 8 + xcd;
    if (head >= p.B * p.H) return;
    const int qt = nq - 1 - slot % nq;
    const int b = head / p.H, h = head % p.H;
    const int q0 = qt * BQ;
    const int koff = p.Sk - p.Sq;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    int kend = p.Sk;
    if (p.causal) kend = min(p.Sk, q0 + BQ + koff);
    if (kend < 1) kend = 1;
    const int nkt = (kend + KT - 1) / KT;
    char* maskb = smem + 4 * TILE;
    elem_t* biasb = (elem_t*)(smem + 4 * TILE + ((nkt * KT + 15) & ~15));

    uint4 qf[NKS];
    const int qi = q0 + wave * 16 + fr;
    const int qic = min(qi, p.Sq - 1);
    {
        const elem_t* qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qi * p.q_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + fg * 8;
            qf[ks] = (qi < p.Sq && d < hd) ? *(const uint4*)(qp + d) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (HOIST == 0) {
            for (int j = tid; j < nkt * KT; j += 512) {
                unsigned char m = 2;
                if (j < p.Sk) m = (p.key_mask == nullptr || p.key_mask[(long)b * p.Sk + j] != 0) ? 1 : 0;
                maskb[j] = m;
            }
        }
    }
    const elem_t* brow = nullptr;
    int bh_off = 0, bw_off = 0;
    float rw[16];                                              // HOIST: rel_w of this lane's 16 key columns
    const elem_t* rh_row = nullptr;                            // HOIST: rel_h row of this lane's query
    float gh[4][4];                                            // HOIST 2: rel_h of this wave's 16 queries, see below
    if constexpr (HOIST == 1) {
        const long row = (long)head * p.Sq + qic;
        const elem_t* wrow = p.rel_w + row * p.KW;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
            const uint2 v = *(const uint2*)(wrow + ns * 16 + fg * 4);
            rw[ns * 4 + 0] = pk_lo(v.x); rw[ns * 4 + 1] = pk_hi(v.x);
            rw[ns * 4 + 2] = pk_lo(v.y); rw[ns * 4 + 3] = pk_hi(v.y);
        }
        rh_row = p.rel_h + row * p.KH;
    } else if constexpr (HOIST == 2) {
        // raw rel_pos_h / rel_pos_w [127, hd] (64 x 64 grid): G[q][t] = bf16(q . rel_pos[t]) on the MFMA (A = table rows, B = the
        // UNSCALED query fragments), rel_w[q][kw] = Gw[q][qx - kw + 63], rel_h[q][kh] = Gh[q][qy - kh + 63] (image_encoder.py:354-392).
        // Gw: all 127 rows -> this wave's 4-KiB LDS pad (aliases the tile buffers, which are not in use yet) -> 16 registers.
        elem_t* gw = (elem_t*)(smem + wave * 4096);            // [16 queries][128]
#pragma unroll 2
        for (int st = 0; st < 8; ++st) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const elem_t* tr = p.rel_w + (long)min(st * 16 + fr, 126) * hd + fg * 8;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks * 32 < hd) {
                    const uint4 a = (ks * 32 + fg * 8 < hd) ? *(const uint4*)(tr + ks * 32) : make_uint4(0, 0, 0, 0);
                    acc = mfma16(a, qf[ks], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gw[fr * 128 + st * 16 + fg * 4 + r] = f2e(acc[r]);      // G[t = st*16 + 4*fg + r][query fr]
        }
        // Gh: the 16 queries of a wave share qy, so the 64 table rows they need are qy - kh + 63; group g (kh = 16g .. 16g+15) is
        // rows base_g .. base_g + 15 with base_g = qy - 16g + 48, one MFMA chain per group, kept in registers: lane (fr, fg')
        // holds gh[g][r] = Gh[query fr][base_g + 4*fg' + r]; tile kh = 16g + j needs row index 15 - j.
        const int qy = (q0 + wave * 16) >> 6;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const elem_t* tr = p.rel_h + (long)min(max(qy - 16 * g + 48 + fr, 0), 126) * hd + fg * 8;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks * 32 < hd) {
                    const uint4 a = (ks * 32 + fg * 8 < hd) ? *(const uint4*)(tr + ks * 32) : make_uint4(0, 0, 0, 0);
                    acc = mfma16(a, qf[ks], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gh[g][r] = rnd(acc[r]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // own pad only
        const int qx = qic & 63;
#pragma unroll
        for (int i = 0; i < 16; ++i) rw[i] = e2f(gw[fr * 128 + qx - ((i >> 2) * 16 + fg * 4 + (i & 3)) + 63]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // every wave is done with its pad before tile 0 is DMA'd over it
    } else if (p.rel_h != nullptr) {
        const int bp = bias_pitch(p);
        stage_rel_bias<NKS>(p, biasb + wave * 16 * bp, bp, qf, q0 + wave * 16, head, lane, bh_off, bw_off, hd);
        brow = biasb + (wave * 16 + fr) * bp;
    }
    if (p.q_scale != 1.0f) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = scale_q8(qf[ks], p.q_scale);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const elem_t* kbase = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
    const elem_t* vbase = p.Vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
    // SAM global attention (HOIST 2 + VROW: 4096 keys = whole tiles, hd 80 in 256-byte rows): the per-lane part of every DMA source is
    // the same for all 64 tiles -- one 32-bit byte offset per piece, computed once (per tile it was ~20 vector instructions x 4 pieces,
    // a third of the tile's vector work) -- and the six chunks of a row that are head-dim padding are not transferred at all: the two
    // the last QK k-step reads (dims 80..95) are zeroed once per tile slot below, the other four and V's six are never read.
    constexpr bool FASTDMA = HOIST == 2 && VROW;
    uint32_t dko[FASTDMA ? 2 : 1], dvo[FASTDMA ? 2 : 1];
    bool dk_real[FASTDMA ? 2 : 1], dv_real[FASTDMA ? 2 : 1];
    if constexpr (FASTDMA) {
        static_assert(!FASTDMA || (CPR == 16 && NWV == 8), "two 1-KiB pieces per wave, K and V");
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = wave + j * NWV, row = i * 4 + (lane >> 4), cpos = lane & 15;
            const int ck = cpos ^ (row & 15), cv = ((((cpos >> 1) ^ (row & 7)) << 1) | (cpos & 1));
            dko[j] = (uint32_t)(((long)row * p.k_ss + ck * 8) * 2);
            dvo[j] = (uint32_t)(((long)row * p.vt_ds + cv * 8) * 2);
            dk_real[j] = ck * 8 < hd;
            dv_real[j] = cv * 8 < hd;
        }
        for (int t = tid; t < 2 * 64 * 2; t += 512) {          // chunks 10, 11 of every K row of both slots (after the barrier above)
            const int slot = t >> 7, row = (t & 127) >> 1, c = 10 + (t & 1);
            *(uint4*)(smem + slot * (2 * TILE) + row * KROW + ((c ^ (row & 15)) << 4)) = make_uint4(0, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the first tile's barrier orders these writes before every read)
    }
    auto issue = [&](int kt) {                                // K tile kt and V^T tile kt -> slot kt & 1 = [K | V^T]
        if (kt >= nkt) return;
        const uint32_t dst = lds_base + (kt & 1) * (2 * TILE);
        if constexpr (FASTDMA) {
            const elem_t* tk = kbase + (long)kt * KT * p.k_ss;
            const elem_t* tv = vbase + (long)kt * KT * p.vt_ds;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (dk_real[j]) glds16s(tk, dko[j], dst + (wave + j * NWV) * 1024);
                if (dv_real[j]) glds16s(tv, dvo[j], dst + TILE + (wave + j * NWV) * 1024);
            }
            return;
        }
#pragma unroll
        for (int i0 = 0; i0 < CPR; i0 += NWV) {
            const int i = i0 + wave;
            if (i < CPR) {
                const int row = i * (64 / CPR) + lane / CPR;
                const int c = (lane % CPR) ^ swz<CPR>(row);
                const int key = min(kt * KT + row, p.Sk - 1);
                const elem_t* src = (c * 8 < hd) ? kbase + (long)key * p.k_ss + c * 8 : p.zeros;
                glds16(src, dst + i * 1024);
            }
        }
        if constexpr (VROW) {
            static_assert(HDP == 128, "row-major V tiles: 256-byte rows");
#pragma unroll
            for (int i0 = 0; i0 < CPR; i0 += NWV) {
                const int i = i0 + wave;
                if (i < CPR) {
                    const int row = i * (64 / CPR) + lane / CPR;
                    const int cpos = lane % CPR;
                    const int c = ((((cpos >> 1) ^ (row & 7)) << 1) | (cpos & 1));      // 32-byte pairs XOR-swizzled with the row
                    const int key = min(kt * KT + row, p.Sk - 1);
                    glds16((c * 8 < hd) ? vbase + (long)key * p.vt_ds + c * 8 : p.zeros, dst + TILE + i * 1024);
                }
            }
        } else {
        const int npieces = hd >> 3;
#pragma unroll
        for (int i0 = 0; i0 < HDP / 8; i0 += NWV) {
            const int i = i0 + wave;
            if (i < npieces) {
                const int row = i * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (row & 7);
                glds16(vbase + (long)row * p.vt_ds + kt * KT + c * 8, dst + TILE + i * 1024);
            }
        }
        }
    };

    f32x4_t oacc[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) oacc[ds] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY;
    f32x4_t lacc = {0.f, 0.f, 0.f, 0.f};     // sum of the (rounded) probabilities of this lane's query: P^T times a row of ones on the matrix pipe
    // one 64-key tile: scores (+ bias), running max, P, P*V
    auto tile = [&](int kt, float rh) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(kt + 1);
        const char* tb = smem + (kt & 1) * (2 * TILE);
        // scores as 16-bit pairs sq[2 ns + half] (keys 16 ns + 4 fg + {0,1 | 2,3}) and the tile maximum of this lane's 16
        uint32_t sq[8];
        float tm = -INFINITY;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const int row = ns * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks * 32 < hd) {
                    const uint4 kf = *(const uint4*)(tb + row * KROW + (((ks * 4 + fg) ^ swz<CPR>(row)) << 4));
                    acc = mfma16(kf, qf[ks], acc);
                }
            }
            if constexpr (HOIST != 0) {
                // rnd(rnd(rnd(acc) + rel_h) + rel_w) two values at a time (score_quad_win: packed converts and adds, 21 vector
                // instructions per quad instead of 45; the maximum is taken on the unrounded sums and rounded once below)
                score_quad_win(acc, rh, f32x2_t{rw[ns * 4], rw[ns * 4 + 1]}, f32x2_t{rw[ns * 4 + 2], rw[ns * 4 + 3]}, sq[ns * 2], sq[ns * 2 + 1], tm);
            } else {
                const uint32_t mk = *(const uint32_t*)(maskb + kt * KT + ns * 16 + fg * 4);
                score_quad<FL>(p, acc, kt * KT + ns * 16 + fg * 4, mk, qi, koff, brow, bh_off, bw_off, sq[ns * 2], sq[ns * 2 + 1], &tm);
            }
        }
        tm = rnd(tm);
        tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float mn = fmaxf(m, tm);                         // finite from tile 0 on (key 0 is always in range)
        if (__builtin_amdgcn_ballot_w64(mn > m) != 0) {        // some query of this wave has a new maximum
            const float alpha = __expf(m - mn);                // exp(-inf) = 0 on the first tile
#pragma unroll
            for (int r = 0; r < 4; ++r) lacc[r] *= alpha;
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[ds][r] *= alpha;
            m = mn;
        }
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // __expf(x) = v_exp_f32(x * log2(e)), written out so that the subtraction and the multiply pair up (v_pk_*_f32)
            const f32x2_t t = (f32x2_t{pk_lo(sq[i]), pk_hi(sq[i])} - m) * 1.4426950408889634f;
            pk[i] = pack2e(__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y));
        }
        // The row sum runs over the ROUNDED probabilities, so that O / l is a true weighted mean of V rows: one more "V^T row" of ones
        // through the MFMA (2 per tile) instead of unpacking and adding every probability again on the VALU, which is the bound here.
#ifdef ULL_ELEM_F16
        const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
#else
        const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
#endif
        lacc = mfma16(ones, make_uint4(pk[0], pk[1], pk[2], pk[3]), lacc);
        lacc = mfma16(ones, make_uint4(pk[4], pk[5], pk[6], pk[7]), lacc);
        if constexpr (VROW) {
            static_assert(FL == FL_SAM_ENC, "hd = 80: five head-dim blocks");
            const int swr = 4 * (fg & 1) + (fr >> 2);
            const uint32_t vb = lds_base + (kt & 1) * (2 * TILE) + TILE + (4 * fg + (fr >> 2)) * KROW + ((fr & 2) << 3) + ((fr & 1) << 3);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint4 pf = make_uint4(pk[4 * kk], pk[4 * kk + 1], pk[4 * kk + 2], pk[4 * kk + 3]);
                u32x2_t va[4], vc[4], wa[1], wc[1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t ad = vb + ((j ^ swr) << 5);
                    if (kk == 0) { va[j] = lds_tr_b64<0>(ad); vc[j] = lds_tr_b64<16 * KROW>(ad); }
                    else { va[j] = lds_tr_b64<32 * KROW>(ad); vc[j] = lds_tr_b64<48 * KROW>(ad); }
                }
                {
                    const uint32_t ad = vb + ((4 ^ swr) << 5);
                    if (kk == 0) { wa[0] = lds_tr_b64<0>(ad); wc[0] = lds_tr_b64<16 * KROW>(ad); }
                    else { wa[0] = lds_tr_b64<32 * KROW>(ad); wc[0] = lds_tr_b64<48 * KROW>(ad); }
                }
                lds_tr_wait<4, 2>(va, vc);
#pragma unroll
                for (int j = 0; j < 4; ++j) oacc[j] = mfma16(make_uint4(va[j].x, va[j].y, vc[j].x, vc[j].y), pf, oacc[j]);
                lds_tr_wait<1, 0>(wa, wc);
                oacc[4] = mfma16(make_uint4(wa[0].x, wa[0].y, wc[0].x, wc[0].y), pf, oacc[4]);
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint4 pf = make_uint4(pk[4 * kk], pk[4 * kk + 1], pk[4 * kk + 2], pk[4 * kk + 3]);
#pragma unroll
            for (int ds = 0; ds < NDS; ++ds) {
                if (ds * 16 < hd) {
                    const int row = ds * 16 + fr;
                    const uint4 vf = *(const uint4*)(tb + TILE + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                    oacc[ds] = mfma16(vf, pf, oacc[ds]);
                }
            }
        }
        }
    };
    issue(0);
    if constexpr (HOIST == 2) {
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            float cur[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cur[r] = g == 0 ? gh[0][r] : g == 1 ? gh[1][r] : g == 2 ? gh[2][r] : gh[3][r];
#pragma unroll 1
            for (int a4 = 0; a4 < 4; ++a4) {
                const int src = fr + 16 * (3 - a4);            // the lane group that holds row index 15 - j, j = 4*a4 + b
#pragma unroll
                for (int b4 = 0; b4 < 4; ++b4) tile(16 * g + 4 * a4 + b4, __shfl(cur[3 - b4], src, 64));
            }
        }
    } else if constexpr (HOIST == 1) {
        float rh = e2f(rh_row[0]);
        for (int kt = 0; kt < nkt; ++kt) {
            const float rh_next = e2f(rh_row[min(kt + 1, p.KH - 1)]);
            tile(kt, rh);
            rh = rh_next;
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) tile(kt, 0.f);
    }
    const float inv = 1.0f / lacc[0];             // every accumulator row holds the same sum over all keys for query fr
    if (qi < p.Sq) {
        elem_t* op = p.O + (long)b * p.o_bs + (long)h * p.o_hs + (long)qi * p.o_ss;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            if (ds * 16 < hd) {
                uint2 o;
                o.x = pack2e(oacc[ds][0] * inv, oacc[ds][1] * inv);
                o.y = pack2e(oacc[ds][2] * inv, oacc[ds][3] * inv);
                *(uint2*)(op + ds * 16 + fg * 4) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SAM global attention (image_encoder.py:196-260 with the decomposed rel-pos bias :354-392 on the 64 x 64 grid: 4096 keys, hd 80, raw
// rel_pos_h / rel_pos_w [127, 80], q | k | v consumed in place, V as rows) -- attn_stream_kernel<128, SAM, HOIST 2, VROW>'s arithmetic,
// statement for statement per query (same rounding points, same fp32 summation orders: the outputs are bit-identical, tools/global_attn_ab.py),
// on a different decomposition: a wave owns 32 queries (two groups of 16, one MFMA column block each) instead of 16, a block is 4 waves
// (128 queries as before) instead of 8, and the tile loop is software-pipelined (S of tile kt + 1 beside the softmax of tile kt, P V of
// tile kt beside the score epilogue of tile kt + 1; K runs two tiles ahead of V in its own ring).
//   * every K fragment (ds_read_b128) and every V fragment (two ds_read_b64_tr_b16) feeds TWO MFMAs: the LDS array is busy 1.41e8 cycles per
//     launch instead of 2.35e8, waves wait half as long (profiles/r05_sam_global_attention.txt);
//   * four waves meet at the tile barrier instead of eight;
//   * the rel_h values of a wave's 32 queries (all in one grid row) are a [64][32] table in LDS, one 2-byte read per (group, tile), instead
//     of 16 registers and a shuffle ladder; rel_w stays in registers (16 per group: the lane's key columns are the same in every tile).
// 80 KB of LDS per block (K ring + V ring + the rel_h tables): two blocks per CU, two waves per SIMD.
// What it did NOT buy is time (1.17-1.22 ms per layer at B = 8 for either form): the kernel is bound by the vector + matrix ISSUE of one SIMD,
// which on gfx950 add up instead of overlapping (tools/probes/valu_rate.hip: one 16x16x32 MFMA 16.1 cycles, + 2.4 per v_add beside it; cvt_pk /
// shifts / max / packed fp32 4.3 cycles, v_exp 8.2) -- 24 MFMAs + ~190 vector instructions per 16 queries x 64 keys is ~1000 cycles = 0.98 ms
// per layer, and the instruction count is pinned by the reference's three roundings per score.
constexpr int ULL_SAMG_VA = 5;
constexpr int ULL_SAMG_VB = 8;
__global__ __launch_bounds__(256, 2) void sam_global_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    constexpr int NWV = 4, NG = 2, BQ = 16 * NG * NWV, HDP = 128, hd = 80;
    constexpr int KROW = HDP * 2, NKS = 3 /* 96 = 3 x 32 >= hd */, NDS = 5 /* 80 = 5 x 16 */;
    constexpr int TILE = 64 * KROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nq = p.Sq / BQ;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = (slot / nq) * 8 + xcd;
    if (head >= p.B * p.H) return;
    const int qt = slot % nq;
    const int b = head / p.H, h = head % p.H;
    const int q0 = qt * BQ + wave * 16 * NG;                   // this wave's first query; its 32 queries share the grid row qy
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    elem_t* ghl = (elem_t*)(smem + 4 * TILE) + wave * (64 * 16 * NG);     // [64 key rows][32 queries]: rel_h, 16-bit (a tile's read = 64 B)

    uint4 qf[NG][NKS];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const elem_t* qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs + (long)(q0 + g * 16 + fr) * p.q_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + fg * 8;
            qf[g][ks] = d < hd ? *(const uint4*)(qp + d) : make_uint4(0, 0, 0, 0);
        }
    }
    // G[q][t] = 16-bit(q . rel_pos[t]) on the MFMA (A = table rows, B = the UNSCALED query fragments): rel_w[q][kw] = Gw[q][qx - kw + 63],
    // rel_h[q][kh] = Gh[q][qy - kh + 63].  Gw: all 127 rows -> this wave's pad (aliases the tile slots, not in use yet) -> 16 registers per
    // group; Gh: the 64 rows qy - kh + 63 -> the wave's LDS table.
    float rw[NG][16];
    {
        elem_t* gw = (elem_t*)(smem + wave * (NG * 4096));      // [group][16 queries][128]
        const int qy = q0 >> 6;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll 2
            for (int st = 0; st < 8; ++st) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const elem_t* tr = p.rel_w + (long)min(st * 16 + fr, 126) * hd + fg * 8;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const uint4 a = (ks * 32 + fg * 8 < hd) ? *(const uint4*)(tr + ks * 32) : make_uint4(0, 0, 0, 0);
                    acc = mfma16(a, qf[g][ks], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[g * 2048 + fr * 128 + st * 16 + fg * 4 + r] = f2e(acc[r]);   // G[t = st*16 + 4*fg + r][query fr]
            }
#pragma unroll 2
            for (int st = 0; st < 4; ++st) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const elem_t* tr = p.rel_h + (long)(qy - (st * 16 + fr) + 63) * hd + fg * 8;               // row of key-grid row kh = st*16 + fr
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const uint4 a = (ks * 32 + fg * 8 < hd) ? *(const uint4*)(tr + ks * 32) : make_uint4(0, 0, 0, 0);
                    acc = mfma16(a, qf[g][ks], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) ghl[(st * 16 + fg * 4 + r) * (16 * NG) + g * 16 + fr] = f2e(acc[r]);  // Gh[kh = st*16 + 4*fg + r][query fr]
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // own pad only
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int qx = (q0 + g * 16 + fr) & 63;
#pragma unroll
            for (int i = 0; i < 16; ++i) rw[g][i] = e2f(gw[g * 2048 + fr * 128 + qx - ((i >> 2) * 16 + fg * 4 + (i & 3)) + 63]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (p.q_scale != 1.0f) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[g][ks] = scale_q8(qf[g][ks], p.q_scale);
    }
    __builtin_amdgcn_s_barrier();                              // every wave is done with its pad before tile 0 is DMA'd over it

    // LDS: K ring [2][64 x 256 B] at 0, V ring [2][64 x 256 B] at 2 TILE, rel_h tables behind.  DMA: a tile is 16 pieces of 1 KiB (4 key rows x
    // 16 chunks); a wave issues 4 K + 4 V pieces per step.  The per-lane byte offset of every piece is the same for all 64 tiles; the six
    // chunks of a row that are head-dim padding are not transferred: chunks 10 / 11 of every K row (dims 80..95, read by the last QK k-step)
    // are zeroed once per buffer, the rest is never read.
    const elem_t* kbase = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
    const elem_t* vbase = p.Vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
    uint32_t dko[4], dvo[4];
    bool dk_real[4], dv_real[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = wave + j * NWV, row = i * 4 + (lane >> 4), cpos = lane & 15;
        const int ck = cpos ^ (row & 15), cv = ((((cpos >> 1) ^ (row & 7)) << 1) | (cpos & 1));
        dko[j] = (uint32_t)(((long)row * p.k_ss + ck * 8) * 2);
        dvo[j] = (uint32_t)(((long)row * p.vt_ds + cv * 8) * 2);
        dk_real[j] = ck * 8 < hd;
        dv_real[j] = cv * 8 < hd;
    }
    for (int t = tid; t < 2 * 64 * 2; t += 64 * NWV) {
        const int sl = t >> 7, row = (t & 127) >> 1, c = 10 + (t & 1);
        *(uint4*)(smem + sl * TILE + row * KROW + ((c ^ (row & 15)) << 4)) = make_uint4(0, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the first barrier below orders these writes before every read)
    auto issue_k = [&](int kt) {
        if (kt >= 64) return;
        const uint32_t dst = lds_base + (kt & 1) * TILE;
        const elem_t* tk = kbase + (long)kt * KT * p.k_ss;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (dk_real[j]) glds16s(tk, dko[j], dst + (wave + j * NWV) * 1024);
    };
    auto issue_v = [&](int kt) {
        if (kt >= 64) return;
        const uint32_t dst = lds_base + (2 + (kt & 1)) * TILE;
        const elem_t* tv = vbase + (long)kt * KT * p.vt_ds;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (dv_real[j]) glds16s(tv, dvo[j], dst + (wave + j * NWV) * 1024);
    };
    // LDS fragment addresses, per lane and once: K fragment of k-step ks in rows fr, fr + 16, ... (the swizzle term of row 16 ns + fr is fr);
    // V fragment base of the transposing reads (see attn_reg_kernel)
    uint32_t kfo[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) kfo[ks] = fr * KROW + (((ks * 4 + fg) ^ fr) << 4);
    const int swr = 4 * (fg & 1) + (fr >> 2);
    const uint32_t vfo = 2 * TILE + (4 * fg + (fr >> 2)) * KROW + ((fr & 2) << 3) + ((fr & 1) << 3);

    f32x4_t oacc[NG][NDS], lacc[NG];
    float m[NG], mn[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        m[g] = -INFINITY;
        lacc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) oacc[g][ds] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#ifdef ULL_ELEM_F16
    const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
#else
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
#endif
    // S tile kt on the matrix pipe: 4 x 3 K fragments, each into both groups' chains (24 MFMAs)
    auto qk = [&](int kt, f32x4_t (&acc)[NG][4]) {
        const char* tb = smem + (kt & 1) * TILE;
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][ns] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const uint4 kf = *(const uint4*)(tb + ns * 16 * KROW + kfo[ks]);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g][ns] = mfma16(kf, qf[g][ks], acc[g][ns]);
            }
        }
    };
    // score epilogue of tile kt: rnd(rnd(rnd(acc) + rel_h) + rel_w) as 16-bit pairs sq[g][2 ns + half] (keys 16 ns + 4 fg + {0,1 | 2,3}), the
    // tile maximum folded into mn[g] = the row maximum including tile kt; returns whether some query of the wave saw a new maximum
    auto scores = [&](int kt, const f32x4_t (&acc)[NG][4], uint32_t (&sq)[NG][8]) -> bool {
        bool grew = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float rh = e2f(ghl[kt * (16 * NG) + g * 16 + fr]);
            float tm = -INFINITY;
#pragma unroll
            for (int ns = 0; ns < 4; ++ns)
                score_quad_win(acc[g][ns], rh, f32x2_t{rw[g][ns * 4], rw[g][ns * 4 + 1]}, f32x2_t{rw[g][ns * 4 + 2], rw[g][ns * 4 + 3]}, sq[g][ns * 2],
                               sq[g][ns * 2 + 1], tm);
            tm = rnd(tm);
            tm = fmaxf(tm, __shfl_xor(tm, 16, 64));
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            mn[g] = fmaxf(m[g], tm);                           // finite from tile 0 on
            grew = grew || mn[g] > m[g];
        }
        return __builtin_amdgcn_ballot_w64(grew) != 0;
    };
    // The tile loop is software-pipelined by hand: while the matrix pipe computes S of tile kt + 1 the vector pipe turns the scores of tile
    // kt into probabilities, and while it multiplies P(kt) into V(kt) the vector pipe does the score epilogue of tile kt + 1 -- the two
    // halves of a step are independent instruction streams inside one wave (the 16-query form left that overlap to chance between waves,
    // and its phases added up).  Step kt therefore needs K(kt + 1) and V(kt): K runs two tiles ahead of V in its own ring.
    issue_k(0);
    issue_k(1);
    issue_v(0);
    f32x4_t acc[NG][4];
    uint32_t sq[NG][8];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    qk(0, acc);
    bool grew = scores(0, acc, sq);
    // one step; LAST: tile 63 (nothing left to prefetch, no S(kt + 1))
    auto step = [&](int kt, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (this wave's pieces of K(kt + 1) and V(kt))
        __builtin_amdgcn_s_barrier();                          // ... and everybody's; every wave is done with K(kt) and V(kt - 1)
        if constexpr (!LAST) {
            issue_k(kt + 2);
            issue_v(kt + 1);
        }
        if (grew) {                                            // some query of this wave has a new maximum (alpha = 1 exactly for the others)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float alpha = __expf(m[g] - mn[g]);      // exp(-inf) = 0 on the first tile
#pragma unroll
                for (int r = 0; r < 4; ++r) lacc[g][r] *= alpha;
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[g][ds][r] *= alpha;
                m[g] = mn[g];
            }
        }
        // ---- first half: S(kt + 1) on the MFMA, P(kt) on the VALU (one basic block, independent streams)
        if constexpr (!LAST) qk(kt + 1, acc);
        uint32_t pk[NG][8];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x2_t t = (f32x2_t{pk_lo(sq[g][i]), pk_hi(sq[g][i])} - m[g]) * 1.4426950408889634f;
                pk[g][i] = pack2e(__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y));
            }
        }
        // ---- second half: l += sum P(kt), O += V(kt)^T P(kt) on the MFMA  ||  score epilogue of tile kt + 1 on the VALU
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // the row sum runs over the ROUNDED probabilities: one more "V^T row" of ones through the MFMA
            lacc[g] = mfma16(ones, make_uint4(pk[g][0], pk[g][1], pk[g][2], pk[g][3]), lacc[g]);
            lacc[g] = mfma16(ones, make_uint4(pk[g][4], pk[g][5], pk[g][6], pk[g][7]), lacc[g]);
        }
        const uint32_t vb = lds_base + (kt & 1) * TILE + vfo;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x2_t va[4], vc[4], wa[1], wc[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t ad = vb + ((j ^ swr) << 5);
                if (kk == 0) { va[j] = lds_tr_b64<0>(ad); vc[j] = lds_tr_b64<16 * KROW>(ad); }
                else { va[j] = lds_tr_b64<32 * KROW>(ad); vc[j] = lds_tr_b64<48 * KROW>(ad); }
            }
            {
                const uint32_t ad = vb + ((4 ^ swr) << 5);
                if (kk == 0) { wa[0] = lds_tr_b64<0>(ad); wc[0] = lds_tr_b64<16 * KROW>(ad); }
                else { wa[0] = lds_tr_b64<32 * KROW>(ad); wc[0] = lds_tr_b64<48 * KROW>(ad); }
            }
            lds_tr_wait<4, 2>(va, vc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 vf = make_uint4(va[j].x, va[j].y, vc[j].x, vc[j].y);
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    oacc[g][j] = mfma16(vf, make_uint4(pk[g][4 * kk], pk[g][4 * kk + 1], pk[g][4 * kk + 2], pk[g][4 * kk + 3]), oacc[g][j]);
            }
            lds_tr_wait<1, 0>(wa, wc);
            {
                const uint4 vf = make_uint4(wa[0].x, wa[0].y, wc[0].x, wc[0].y);
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    oacc[g][4] = mfma16(vf, make_uint4(pk[g][4 * kk], pk[g][4 * kk + 1], pk[g][4 * kk + 2], pk[g][4 * kk + 3]), oacc[g][4]);
            }
        }
        if constexpr (!LAST) {
            grew = scores(kt + 1, acc, sq);
        }
    };
#pragma unroll 2
    for (int kt = 0; kt < 63; ++kt) step(kt, std::false_type{});
    step(63, std::true_type{});
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float inv = 1.0f / lacc[g][0];      // every accumulator row holds the same sum over all keys for query fr
        elem_t* op = p.O + (long)b * p.o_bs + (long)h * p.o_hs + (long)(q0 + g * 16 + fr) * p.o_ss;
#pragma unroll
        for (int ds = 0; ds < NDS; ++ds) {
            uint2 o;
            o.x = pack2e(oacc[g][ds][0] * inv, oacc[g][ds][1] * inv);
            o.y = pack2e(oacc[g][ds][2] * inv, oacc[g][ds][3] * inv);
            *(uint2*)(op + ds * 16 + fg * 4) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// At most 16 queries per (batch, head): KV-cached decoding (1 query against the whole cache) and the mask decoder's
// token -> image attention (7..16 queries against 4096 keys).  The kernels above give such a call ONE working wave per head
// that walks the key tiles one barrier at a time (measured 32 us per LLaMA decode layer, 234 us per mask-decoder call); here the
// key tiles are split over up to 16 waves of the block (wave w owns tiles w, w + nwv, ...; <= 4 per wave, kept in registers),
// K / V^T fragments come straight from global memory into the MFMA operands (every byte is used by exactly one wave, so LDS
// staging would buy nothing), and the waves meet three times in LDS: row max, row sum, partial O.  Same rounding points as the
// register kernel (bf16 scores, exact two-step fp32 softmax, bf16 P); only the fp32 summation order of sum(exp) and of the
// P*V partials differs.
//   TPW = tiles per wave held in registers: 1 for <= 1024 keys (decode; the wave's V^T fragments are then requested together
//   with its K fragments, ahead of the softmax barriers), 4 for up to 4096 keys.
template <int HDP, int FL, int TPW>
__global__ __launch_bounds__(1024) void attn_fewq_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NKS = HDP / 32, NDS = HDP / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int hd = head_dim_of<HDP, FL>(p);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwv = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int head = blockIdx.x;
    const int b = head / p.H, h = head % p.H;
    const int koff = p.Sk - p.Sq;
    const int nkt = (p.Sk + KT - 1) / KT;
    float* red = (float*)smem;                                  // [2][16 waves][16 queries]
    float* obuf = red + 2 * 16 * 16;                            // [nwv][NDS * 4][64 lanes]

    uint4 qf[NKS];
    const int qi = fr;
    {
        const elem_t* qp = p.Q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qi * p.q_ss;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + fg * 8;
            qf[ks] = (qi < p.Sq && d < hd) ? *(const uint4*)(qp + d) : make_uint4(0, 0, 0, 0);
        }
        if (p.q_scale != 1.0f) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qf[ks] = scale_q8(qf[ks], p.q_scale);
        }
    }
    const elem_t* kbase = p.K + (long)b * p.k_bs + (long)h * p.k_hs;
    const elem_t* vbase = p.Vt + (long)b * p.vt_bs + (long)h * p.vt_hs;

    // ---- scores of this wave's tiles -> registers (packed bf16), running max -------------------------------
    uint32_t sp[TPW][8];
    float m = -INFINITY;
#pragma clang loop unroll(full)
    for (int t = 0; t < TPW; ++t) {
        const int kt = wave + t * nwv;
        if (kt < nkt) {
#pragma unroll
            for (int ns = 0; ns < 4; ++ns) {
                const int key = min(kt * KT + ns * 16 + fr, p.Sk - 1);
                const elem_t* kp = kbase + (long)key * p.k_ss + fg * 8;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    if (ks * 32 < hd) {
                        const uint4 kf = (ks * 32 + fg * 8 < hd) ? *(const uint4*)(kp + ks * 32) : make_uint4(0, 0, 0, 0);
                        acc = mfma16(kf, qf[ks], acc);
                    }
                }
                const int j0 = kt * KT + ns * 16 + fg * 4;
                uint32_t mk = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + r;
                    uint32_t mb = 2;
                    if (j < p.Sk) mb = (p.key_mask == nullptr || p.key_mask[(long)b * p.Sk + j] != 0) ? 1 : 0;
                    mk |= mb << (8 * r);
                }
                score_quad<FL>(p, acc, j0, mk, qi, koff, nullptr, 0, 0, sp[t][ns * 2], sp[t][ns * 2 + 1]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                m = fmaxf(m, pk_lo(sp[t][i]));
                m = fmaxf(m, pk_hi(sp[t][i]));
            }
        }
    }
    uint4 vpre[2][NDS];                                         // TPW == 1: this wave's V^T fragments, in flight across the barriers
    if constexpr (TPW == 1) {
        if (wave < nkt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds)
                    if (ds * 16 < hd) vpre[kk][ds] = *(const uint4*)(vbase + (long)(ds * 16 + fr) * p.vt_ds + wave * KT + (kk * 4 + fg) * 8);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (fg == 0) red[wave * 16 + fr] = m;
    __syncthreads();
    for (int w = 0; w < nwv; ++w) m = fmaxf(m, red[w * 16 + fr]);

    // ---- exact fp32 softmax over the bf16 scores of ALL waves ----------------------------------------------
    float sum = 0.f;
#pragma clang loop unroll(full)
    for (int t = 0; t < TPW; ++t)
        if (wave + t * nwv < nkt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                sum += __expf(pk_lo(sp[t][i]) - m);
                sum += __expf(pk_hi(sp[t][i]) - m);
            }
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (fg == 0) red[256 + wave * 16 + fr] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < nwv; ++w) sum += red[256 + w * 16 + fr];
    const float inv = 1.0f / sum;

    // ---- partial O^T = V^T P^T over this wave's tiles --------------------------------------------------------
    f32x4_t oacc[NDS];
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) oacc[ds] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(full)
    for (int t = 0; t < TPW; ++t) {
        const int kt = wave + t * nwv;
        if (kt < nkt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float lo = __expf(pk_lo(sp[t][i]) - m) * inv;
                const float hi = __expf(pk_hi(sp[t][i]) - m) * inv;
                sp[t][i] = pack2e(lo, hi);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint4 pf = make_uint4(sp[t][4 * kk], sp[t][4 * kk + 1], sp[t][4 * kk + 2], sp[t][4 * kk + 3]);
#pragma unroll
                for (int ds = 0; ds < NDS; ++ds) {
                    if (ds * 16 < hd) {
                        uint4 vf;
                        if constexpr (TPW == 1) vf = vpre[kk][ds];
                        else vf = *(const uint4*)(vbase + (long)(ds * 16 + fr) * p.vt_ds + kt * KT + (kk * 4 + fg) * 8);
                        oacc[ds] = mfma16(vf, pf, oacc[ds]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds)
#pragma unroll
        for (int r = 0; r < 4; ++r) obuf[(wave * NDS * 4 + ds * 4 + r) * 64 + lane] = oacc[ds][r];
    __syncthreads();
    // element e = (ds*4 + r)*64 + lane  <->  O[query lane & 15][d = ds*16 + 4*(lane >> 4) + r]
    for (int e = tid; e < NDS * 4 * 64; e += blockDim.x) {
        float acc = 0.f;
        for (int w = 0; w < nwv; ++w) acc += obuf[w * NDS * 4 * 64 + e];
        const int reg = e >> 6, ln = e & 63;
        const int q = ln & 15, d = (reg >> 2) * 16 + (ln >> 4) * 4 + (reg & 3);
        if (q < p.Sq && d < hd) p.O[(long)b * p.o_bs + (long)h * p.o_hs + (long)q * p.o_ss + d] = f2e(acc);
    }
}

// ---------------------------------------------------------------------------------------------
// RoPE in place on the q|k part of a fused QKV buffer (transformers apply_rotary_pos_emb on bf16
// tensors: q*cos -> bf16, rotate_half(q)*sin -> bf16, sum -> bf16; cos/sin are fp32 values cast to bf16).
// One block per token; thread t owns the 8-wide dim chunk (t % (hd/16)) of head-instances t / (hd/16), ...
// sin_sign = -1 applies the transposed rotation: the backward of apply_rotary_pos_emb (the rotation is orthogonal).
__global__ __launch_bounds__(256) void rope_inplace_kernel(elem_t* __restrict__ x, long row_stride, const int64_t* __restrict__ pos,
                                                           const float* __restrict__ inv_freq, int n_heads, int hd, float sin_sign) {
    const long tok = blockIdx.x;
    const int half = hd >> 1;
    const int cpr = half >> 3;                   // 8-wide chunks per half head (hd % 16 == 0)
    const int c = threadIdx.x % cpr;
    const float pf = (float)pos[tok];
    float cs[8], sn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = pf * inv_freq[c * 8 + j];
        cs[j] = rnd(cosf(a));
        sn[j] = rnd(sinf(a)) * sin_sign;
    }
    elem_t* xr = x + tok * row_stride;
    for (int hh = threadIdx.x / cpr; hh < n_heads; hh += blockDim.x / cpr) {
        elem_t* p1 = xr + hh * hd + c * 8;
        elem_t* p2 = p1 + half;
        float a[8], bb[8], o1[8], o2[8];
        unpack8(*(const uint4*)p1, a);
        unpack8(*(const uint4*)p2, bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o1[j] = rnd(a[j] * cs[j]) + rnd(-bb[j] * sn[j]);
            o2[j] = rnd(bb[j] * cs[j]) + rnd(a[j] * sn[j]);
        }
        *(uint4*)p1 = pack8(o1);
        *(uint4*)p2 = pack8(o2);
    }
}

// cos / sin tables of LlamaRotaryEmbedding for a list of positions: fp32 cos / sin of pos * inv_freq cast to the element type
// (hf modeling_llama.py:72-126).  One table serves every layer of a forward (the fused QKV epilogue reads it).
__global__ __launch_bounds__(256) void rope_table_kernel(const int64_t* __restrict__ pos, const float* __restrict__ inv_freq, long tokens,
                                                         int half, elem_t* __restrict__ cs, elem_t* __restrict__ sn) {
    const long total = tokens * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float a = (float)pos[i / half] * inv_freq[i % half];
        cs[i] = f2e(cosf(a));
        sn[i] = f2e(sinf(a));
    }
}

// Decode-step variant: RoPE on q (in place) and k, and the append of the new token's k / v to the KV cache, in one launch
// (was rope + two strided copies per layer).  qkv rows = tokens (b, s) of a step, [q | k | v] heads contiguous; the token at
// step position s goes to cache slot past + s: K cache [B, H, smax, hd] row past + s, V^T cache [B, H, hd, smax] column
// vt_slot(past + s) (the key-permuted layout of transpose_v_kernel).  One block per token.
__global__ __launch_bounds__(64) void rope_append_kernel(elem_t* __restrict__ qkv, long row_stride, const int64_t* __restrict__ pos,
                                                         const float* __restrict__ inv_freq, int S, int H, int hd, elem_t* __restrict__ kc,
                                                         elem_t* __restrict__ vtc, int smax, int past) {
    // one 64-lane block per (token, head): a decode step is ONE token, and a single block walking all 2H heads and scattering
    // H * hd cache columns was a 10-us latency chain per layer; H blocks do it in a third of that
    const long tok = blockIdx.x;
    const int h = blockIdx.y;
    const int b = (int)(tok / S), s = (int)(tok % S);
    const int half = hd >> 1;
    const int cpr = half >> 3;                                   // 16-byte chunks per half head
    elem_t* xr = qkv + tok * row_stride;
    const int slot_k = past + s;
    if ((int)threadIdx.x < 2 * cpr) {                            // lanes [0, cpr): q head h; [cpr, 2 cpr): k head h
        const int c = threadIdx.x % cpr;
        const bool is_k = (int)threadIdx.x >= cpr;
        const float pf = (float)pos[tok];
        float cs[8], sn[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = pf * inv_freq[c * 8 + j];
            cs[j] = rnd(cosf(a));
            sn[j] = rnd(sinf(a));
        }
        elem_t* p1 = xr + ((is_k ? H : 0) + h) * hd + c * 8;
        elem_t* p2 = p1 + half;
        float a[8], bb[8], o1[8], o2[8];
        unpack8(*(const uint4*)p1, a);
        unpack8(*(const uint4*)p2, bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o1[j] = rnd(a[j] * cs[j]) + rnd(-bb[j] * sn[j]);
            o2[j] = rnd(bb[j] * cs[j]) + rnd(a[j] * sn[j]);
        }
        const uint4 r1 = pack8(o1), r2 = pack8(o2);
        if (!is_k) {
            *(uint4*)p1 = r1;
            *(uint4*)p2 = r2;
        } else {
            elem_t* kp = kc + (((long)b * H + h) * smax + slot_k) * hd + c * 8;
            *(uint4*)kp = r1;
            *(uint4*)(kp + half) = r2;
        }
    }
    const int w = slot_k & 31;
    const int slot_v = (slot_k & ~31) + 8 * ((w >> 2) & 3) + 4 * (w >> 4) + (w & 3);
    const elem_t* vr = xr + 2 * H * hd + h * hd;
    for (int i = threadIdx.x; i < hd; i += 64) vtc[(((long)b * H + h) * hd + i) * smax + slot_v] = vr[i];
}

// V [B, S, H, hd] (token stride v_ss, heads contiguous) -> Vt [B, H, hd, pitch], zero-filled for keys >= S.
// Inside every 32-key block the keys are stored permuted: slot 8g + 4a + r holds key 16a + 4g + r (a<2, g<4, r<4),
// which is the (lane group g, element j = 4a + r) <-> key map that the attention kernel's probability registers
// have after the swapped QK^T MFMA -- so P*V needs no cross-lane movement.  64(s) x 64(d) tiles through LDS.
__global__ __launch_bounds__(256) void transpose_v_kernel(const elem_t* __restrict__ v, long v_bs, long v_ss, elem_t* __restrict__ vt, int S,
                                                          int H, int hd, int pitch) {
    __shared__ __attribute__((aligned(16))) elem_t t[64][68];            // [d][key]; 136-byte rows keep the 8-byte reads aligned
    const int b = blockIdx.z / H, h = blockIdx.z % H;
    const int s0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const elem_t* vp = v + (long)b * v_bs + (long)h * hd;
    // 16-byte loads along d (coalesced: 8 lanes cover one token's 64 dims), scattered 2-byte LDS writes
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = threadIdx.x + 256 * k;
        const int sl = c >> 3, dc = (c & 7) * 8;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (s0 + sl < S && d0 + dc < hd) val = *(const uint4*)(vp + (long)(s0 + sl) * v_ss + d0 + dc);
        const uint32_t w[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[dc + 2 * j][sl] = (elem_t)(w[j] & 0xffff);
            t[dc + 2 * j + 1][sl] = (elem_t)(w[j] >> 16);
        }
    }
    __syncthreads();
    elem_t* op = vt + ((long)b * H + h) * hd * pitch;
    // output chunk oc = 8 consecutive slots 8g .. 8g+7 of a 32-key block = keys {4g..4g+3} and {16+4g..16+4g+3}: two 8-byte LDS reads
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = threadIdx.x + 256 * k;
        const int d = c >> 3, oc = c & 7;
        const int kb = (oc >> 2) * 32 + (oc & 3) * 4;
        const uint2 lo = *(const uint2*)&t[d][kb], hi = *(const uint2*)&t[d][kb + 16];
        if (d0 + d < hd) *(uint4*)(op + (long)(d0 + d) * pitch + s0 + oc * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

template <int HDP, int NT, int FL, int NWV = 8, bool EXACT = false, bool VROW = false>
int launch_attn(const AttnArgs& a, hipStream_t st) {
    constexpr int TILE = 64 * HDP * 2 > HDP * 128 ? 64 * HDP * 2 : HDP * 128;
    const int lds = attn_reg_nbuf<HDP, NT, NWV, EXACT, VROW>() * TILE + NT * KT +
                    (a.rel_h ? NWV * 16 * (((a.rel_mode == 2 ? 2 * (a.KH + a.KW) - 2 : a.KH + a.KW) | 1)) * 2 + 16 : 0);
    if (lds > 160 * 1024) return ULL_ERR_LDS;
    static UllOncePerDevice once;
    if (lds > 64 * 1024 && once.first())
        (void)hipFuncSetAttribute((const void*)attn_reg_kernel<HDP, NT, FL, NWV, EXACT, VROW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nq = (a.Sq + 16 * NWV - 1) / (16 * NWV);
    const int nheads = a.B * a.H;
    const dim3 grid(((nheads + 7) / 8) * 8 * nq);
    hipLaunchKernelGGL((attn_reg_kernel<HDP, NT, FL, NWV, EXACT, VROW>), grid, dim3(NWV * 64), lds, st, a);
    return ull_check_launch();
}

template <int HDP, int FL>
int launch_long(const AttnArgs& a, hipStream_t st) {
    constexpr int TILE = 64 * HDP * 2;
    const int nt = (a.Sk + KT - 1) / KT;
    const int lds = 4 * TILE + ((nt * KT + 15) & ~15) + (a.rel_h ? 8 * 16 * (((a.rel_mode == 2 ? 2 * (a.KH + a.KW) - 2 : a.KH + a.KW) | 1)) * 2 + 16 : 0);
    if (lds > 160 * 1024) return ULL_ERR_LDS;
    static UllOncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)attn_long_kernel<HDP, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nq = (a.Sq + 127) / 128;
    const dim3 grid(((a.B * a.H + 7) / 8) * 8 * nq);
    hipLaunchKernelGGL((attn_long_kernel<HDP, FL>), grid, dim3(512), lds, st, a);
    return ull_check_launch();
}

template <int HDP, int FL, int HOIST, bool VROW = false>
int launch_stream(const AttnArgs& a, hipStream_t st) {
    constexpr int TILE = 64 * HDP * 2;
    static_assert(!VROW || HOIST == 2, "row-major V: the SAM global-attention form");
    const int nt = (a.Sk + KT - 1) / KT;
    int lds = 4 * TILE;
    if (HOIST == 0) lds += ((nt * KT + 15) & ~15) + (a.rel_h ? 8 * 16 * (((a.rel_mode == 2 ? 2 * (a.KH + a.KW) - 2 : a.KH + a.KW) | 1)) * 2 + 16 : 0);
    if (lds > 160 * 1024) return ULL_ERR_LDS;
    static UllOncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)attn_stream_kernel<HDP, FL, HOIST, VROW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nq = (a.Sq + 127) / 128;
    const dim3 grid(((a.B * a.H + 7) / 8) * 8 * nq);
    hipLaunchKernelGGL((attn_stream_kernel<HDP, FL, HOIST, VROW>), grid, dim3(512), lds, st, a);
    return ull_check_launch();
}

int launch_sam_global(const AttnArgs& a, hipStream_t st) {
    constexpr int LDS = 4 * 64 * 256 + 4 * 32 * 64 * 2;        // K ring + V ring + the rel_h tables = 80 KB: two blocks per CU
    static UllOncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)sam_global_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nq = a.Sq / 128;
    const dim3 grid(((a.B * a.H + 7) / 8) * 8 * nq);
    hipLaunchKernelGGL(sam_global_kernel, grid, dim3(256), LDS, st, a);
    return ull_check_launch();
}

template <int HDP, int FL, int TPW>
int launch_fewq_t(const AttnArgs& a, int nwv, hipStream_t st) {
    const int lds = 2 * 16 * 16 * 4 + nwv * HDP * 16 * 4;
    static UllOncePerDevice once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)attn_fewq_kernel<HDP, FL, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((attn_fewq_kernel<HDP, FL, TPW>), dim3(a.B * a.H), dim3(nwv * 64), lds, st, a);
    return ull_check_launch();
}

template <int HDP, int FL>
int launch_fewq(const AttnArgs& a, hipStream_t st) {
    const int nt = (a.Sk + KT - 1) / KT;
    const int nwv = nt < 16 ? nt : 16;
    return nt <= 16 ? launch_fewq_t<HDP, FL, 1>(a, nwv, st) : launch_fewq_t<HDP, FL, 4>(a, nwv, st);
}

// Which straight-line flavor (if any) the arguments correspond to.
int flavor_of(const AttnArgs& a) {
    if (a.scale_mode == 1 && a.causal && !a.rel_h && a.q_scale == 1.0f) return FL_LLAMA;
    if (a.scale_mode == 1 && !a.causal && !a.rel_h && !a.key_mask && a.q_scale == 1.0f) return FL_CLIP;
    if (a.scale_mode == 0 && !a.causal && a.rel_h && !a.key_mask) return FL_SAM_ENC;
    if (a.scale_mode == 2 && !a.causal && !a.rel_h && !a.key_mask && a.q_scale == 1.0f) return FL_SAM_DEC;
    return FL_RUNTIME;
}

template <int HDP>
int dispatch_nt(const AttnArgs& a, hipStream_t st) {
    const int nt = (a.Sk + KT - 1) / KT;
    int fl = flavor_of(a);
    // the flavored kernels take the head dim as a compile-time constant (head_dim_of)
    if ((fl == FL_LLAMA || fl == FL_CLIP) && a.hd != HDP) fl = FL_RUNTIME;
    if (fl == FL_SAM_ENC && HDP == 128 && a.hd != 80) fl = FL_RUNTIME;
    if (a.v_rows) {                      // V handed over row-major: the kernels that transpose on the fly (see the C entry)
        if constexpr (HDP == 128) {
            constexpr int ULL_ATTN_NWV = 4;
            if (fl == FL_LLAMA && a.Sq > 16 && nt <= 11) return launch_attn<128, 11, FL_LLAMA, ULL_ATTN_NWV, false, true>(a, st);
            if (fl == FL_LLAMA && a.Sq > 16 && nt <= 16) return launch_attn<128, 16, FL_LLAMA, 8, false, true>(a, st);
            if (fl == FL_SAM_ENC && a.rel_mode == 2 && a.KW == 64 && a.KH == 64 && a.Sk == 4096 && (a.Sq & 15) == 0 && a.Sq > 16 &&
                64 * a.k_ss * 2 < (1L << 31) && 64 * a.vt_ds * 2 < (1L << 31)) {       // (32-bit per-lane DMA offsets inside a tile)
                if ((a.Sq & 127) == 0 && a.Sq == a.Sk && a.hd == 80 && !a.key_mask && !a.causal) return launch_sam_global(a, st);   // (queries = the key grid)
                return launch_stream<128, FL_SAM_ENC, 2, true>(a, st);
            }
        }
        if constexpr (HDP == 64) {
            if (fl == FL_CLIP && a.Sq > 16 && nt <= 5) return launch_attn<64, 5, FL_CLIP, 8, false, true>(a, st);
            if (fl == FL_CLIP && a.Sq > 16 && nt <= 11) return launch_attn<64, 11, FL_CLIP, 4, false, true>(a, st);
        }
        return ULL_ERR_SHAPE;
    }
    // <= 16 queries (decode steps, mask-decoder tokens): split the keys over the waves of one block per head
    if (a.Sq <= 16 && !a.rel_h && nt >= 2 && nt <= 64) {
        if constexpr (HDP == 128) {
            if (fl == FL_LLAMA) return launch_fewq<HDP, FL_LLAMA>(a, st);
        }
        if constexpr (HDP == 32) {
            if (fl == FL_SAM_DEC) return launch_fewq<HDP, FL_SAM_DEC>(a, st);
        }
        return launch_fewq<HDP, FL_RUNTIME>(a, st);
    }
    // specialised instantiations exist for the shapes on the u-LLaVA path; everything else takes the run-time-flag kernels
    if constexpr (HDP == 128) {
        if (fl == FL_LLAMA && nt <= 11) return launch_attn<128, 11, FL_LLAMA, 4>(a, st);
        if (fl == FL_LLAMA && nt <= 16) return launch_attn<128, 16, FL_LLAMA>(a, st);
        if (a.win16) {                                                                   // 14 x 14 windows on image-order tokens
            if (fl != FL_SAM_ENC || a.KH != 14 || a.KW != 14 || a.Sk != 196 || a.Sq != 196 || a.key_mask) return ULL_ERR_SHAPE;
            const int n_items = a.B * a.H, n_cu = ull_cu_count();
            static UllOncePerDevice once;
            if (once.first()) (void)hipFuncSetAttribute((const void*)sam_window_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            // one workgroup per CU (two / four shorter runs per CU: 165 / 188 us against 152 standalone, RES step 86.1 / 86.5 ms against 85.8)
            hipLaunchKernelGGL(sam_window_kernel, dim3(n_items < n_cu ? n_items : n_cu), dim3(SW_NWV * 64), SW_LDS, st, a, n_items);
            return ull_check_launch();
        }
        if (fl == FL_SAM_ENC && nt == 4 && a.Sq <= 208) return launch_attn<128, 4, FL_SAM_ENC, 13, true>(a, st);   // 14 x 14 windows
        if (fl == FL_SAM_ENC && nt <= 11) return launch_attn<128, 11, FL_SAM_ENC>(a, st);
        if (fl == FL_SAM_ENC && nt > 16) {
            if (a.rel_mode == 1 && a.KW == KT && (a.Sk % KT) == 0) return launch_stream<128, FL_SAM_ENC, 1>(a, st);
            if (a.rel_mode == 2 && a.KW == 64 && a.KH == 64 && a.Sk == 4096 && (a.Sq & 15) == 0) return launch_stream<128, FL_SAM_ENC, 2>(a, st);
            return launch_stream<128, FL_SAM_ENC, 0>(a, st);
        }
    }
    if constexpr (HDP == 64) {
        if (fl == FL_CLIP && nt <= 5) return launch_attn<64, 5, FL_CLIP>(a, st);
        if (fl == FL_CLIP && nt <= 11) return launch_attn<64, 11, FL_CLIP, 4>(a, st);
    }
    if constexpr (HDP == 32) {
        if (fl == FL_SAM_DEC && nt <= 5) return launch_attn<32, 5, FL_SAM_DEC>(a, st);
        if (fl == FL_SAM_DEC && nt > 16) return launch_long<32, FL_SAM_DEC>(a, st);      // mask decoder: the exact two-pass form
    }
    if constexpr (HDP < 128) {          // (the <128, 5> instantiation spills; hd=128 starts at the 11-tile variant)
        if (nt <= 5) return launch_attn<HDP, 5, FL_RUNTIME>(a, st);
    }
    if (nt <= 11) return launch_attn<HDP, 11, FL_RUNTIME>(a, st);
    if (nt <= 16) return launch_attn<HDP, 16, FL_RUNTIME>(a, st);
    return launch_long<HDP, FL_RUNTIME>(a, st);     // > 1024 keys: two-pass streaming kernel
}

}  // namespace

// Strides are in elements.  Q/K rows are head_dim-contiguous; Vt rows (one per head dim) are key-contiguous with
// `vt_len` readable, finite columns (multiple of 8; keys >= Sk must be zero).  key_mask: int32 [B, Sk] or null.
// scale_mode 0: S = bf16(QK^T); 1: bf16(bf16(QK^T) * scale); 2: bf16(bf16(QK^T) / scale).

extern "C" int ULL_FN(ull_attention_)(const void* Q, int64_t q_bs, int64_t q_hs, int64_t q_ss, const void* K, int64_t k_bs, int64_t k_hs,
                                  int64_t k_ss, const void* Vt, int64_t vt_bs, int64_t vt_hs, int64_t vt_ds, int64_t vt_len, void* O,
                                  int64_t o_bs, int64_t o_hs, int64_t o_ss, const void* key_mask, int64_t B, int64_t H, int64_t Sq,
                                  int64_t Sk, int64_t hd, int causal, int scale_mode, float scale, float q_scale, const void* rel_h,
                                  const void* rel_w, int64_t rel_kh, int64_t rel_kw, int rel_mode, const void* zeros, void* stream) {
    if (!Q || !K || !Vt || !O || !zeros || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return ULL_ERR_ARG;
    if (hd <= 0 || hd > 128 || (hd & 15)) return ULL_ERR_SHAPE;
    if (vt_len != 0 && ((vt_len & 63) || vt_len < ((Sk + 63) & ~63))) return ULL_ERR_SHAPE;
    if ((q_ss & 7) || (k_ss & 7) || (vt_ds & 7) || (q_hs & 7) || (k_hs & 7) || (q_bs & 7) || (k_bs & 7) || (vt_hs & 7) || (vt_bs & 7) ||
        (o_ss & 3) || (o_hs & 3) || (o_bs & 3))
        return ULL_ERR_SHAPE;
    // V-as-rows forms (LLaMA / CLIP prefill, SAM global): the epilogue stores whole O rows through the LDS as 16-byte pieces -- the output needs
    // 8-element strides and a 16-byte base, not only the 8 bytes of the register-direct epilogue (no kernel of that form otherwise: -2, and
    // the host falls back to the V^T path)
    if (vt_len == 0 && (((o_ss | o_hs | o_bs) & 7) || ((uintptr_t)O & 15))) return ULL_ERR_SHAPE;
    AttnArgs a;
    a.Q = (const elem_t*)Q; a.K = (const elem_t*)K; a.Vt = (const elem_t*)Vt; a.O = (elem_t*)O;
    a.key_mask = (const int32_t*)key_mask;
    a.q_bs = q_bs; a.q_hs = q_hs; a.q_ss = q_ss; a.k_bs = k_bs; a.k_hs = k_hs; a.k_ss = k_ss;
    a.vt_bs = vt_bs; a.vt_hs = vt_hs; a.vt_ds = vt_ds; a.o_bs = o_bs; a.o_hs = o_hs; a.o_ss = o_ss;
    a.B = (int)B; a.H = (int)H; a.Sq = (int)Sq; a.Sk = (int)Sk; a.hd = (int)hd; a.vt_len = (int)vt_len;
    a.causal = causal; a.scale_mode = scale_mode; a.scale = scale;
    a.zeros = (const elem_t*)zeros;
    a.rel_h = (const elem_t*)rel_h; a.rel_w = (const elem_t*)rel_w; a.KH = (int)rel_kh; a.KW = (int)rel_kw; a.q_scale = q_scale;
    a.inv_kw = rel_kw > 0 ? 1.0f / (float)rel_kw : 0.f;
    a.rel_mode = rel_h ? rel_mode : 0;
    a.win16 = 0; a.mg_h = a.mg_nwx = a.mg_nwy = a.mg_nw = 0;
    a.v_rows = vt_len == 0;
    a.img_h = a.img_w = a.nwy = a.nwx = 0; a.k_pad = a.v_pad = nullptr;
    if (rel_h && rel_mode != 1 && rel_mode != 2) return ULL_ERR_ARG;
    if ((rel_h == nullptr) != (rel_w == nullptr)) return ULL_ERR_ARG;
    if (rel_h && (rel_kh <= 0 || rel_kw <= 0 || rel_kh + rel_kw > 256 || rel_kh * rel_kw < Sk)) return ULL_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (hd <= 32) return dispatch_nt<32>(a, st);
    if (hd <= 64) return dispatch_nt<64>(a, st);
    return dispatch_nt<128>(a, st);
}

// x: first of `n_heads` consecutive heads (q heads then k heads of a fused QKV row); positions int64 [tokens];
// inv_freq fp32 [hd/2] (host-computed exactly like LlamaRotaryEmbedding).
extern "C" int ULL_FN(ull_rope_inplace_)(void* x, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens,
                                     int64_t n_heads, int64_t hd, void* stream) {
    if (!x || !positions || !inv_freq || tokens <= 0) return ULL_ERR_ARG;
    const int64_t cpr = hd >> 4;
    if ((hd & 15) || hd > 256 || (cpr & (cpr - 1)) || (row_stride & 7)) return ULL_ERR_SHAPE;   // 256 % (hd/16) == 0
    hipLaunchKernelGGL(rope_inplace_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (elem_t*)x, row_stride,
                       (const int64_t*)positions, (const float*)inv_freq, (int)n_heads, (int)hd, 1.0f);
    return ull_check_launch();
}

// Backward of ull_rope_inplace: dx <- R(pos)^T dx (same kernel with the sine negated), in place on the gradient of the q|k heads.
extern "C" int ULL_FN(ull_rope_bwd_inplace_)(void* dx, int64_t row_stride, const void* positions, const void* inv_freq, int64_t tokens,
                                         int64_t n_heads, int64_t hd, void* stream) {
    if (!dx || !positions || !inv_freq || tokens <= 0) return ULL_ERR_ARG;
    const int64_t cpr = hd >> 4;
    if ((hd & 15) || hd > 256 || (cpr & (cpr - 1)) || (row_stride & 7)) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(rope_inplace_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (elem_t*)dx, row_stride,
                       (const int64_t*)positions, (const float*)inv_freq, (int)n_heads, (int)hd, -1.0f);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_rope_table_)(const void* positions, const void* inv_freq, int64_t tokens, int64_t half, void* cos_out, void* sin_out,
                                   void* stream) {
    if (!positions || !inv_freq || !cos_out || !sin_out || tokens <= 0 || half <= 0) return ULL_ERR_ARG;
    const long total = tokens * half;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rope_table_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const int64_t*)positions, (const float*)inv_freq,
                       (long)tokens, (int)half, (elem_t*)cos_out, (elem_t*)sin_out);
    return ull_check_launch();
}

// Decode step: RoPE on the q and k heads of the fused QKV rows + append of k (roped) and v to the KV cache (see the kernel).
extern "C" int ULL_FN(ull_rope_append_)(void* qkv, int64_t row_stride, const void* positions, const void* inv_freq, int64_t B, int64_t S,
                                    int64_t H, int64_t hd, void* k_cache, void* vt_cache, int64_t smax, int64_t past, void* stream) {
    if (!qkv || !positions || !inv_freq || !k_cache || !vt_cache || B <= 0 || S <= 0) return ULL_ERR_ARG;
    const int64_t cpr = hd >> 4;
    if ((hd & 15) || hd > 256 || (cpr & (cpr - 1)) || (row_stride & 7) || past + S > smax) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(rope_append_kernel, dim3((unsigned)(B * S), (unsigned)H), dim3(64), 0, (hipStream_t)stream, (elem_t*)qkv, row_stride,
                       (const int64_t*)positions, (const float*)inv_freq, (int)S, (int)H, (int)hd, (elem_t*)k_cache, (elem_t*)vt_cache,
                       (int)smax, (int)past);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_transpose_v_)(const void* v, int64_t v_bs, int64_t v_ss, void* vt, int64_t B, int64_t S, int64_t H, int64_t hd,
                                    int64_t pitch, void* stream) {
    if (!v || !vt || B <= 0 || S <= 0) return ULL_ERR_ARG;
    if (pitch < S || (pitch & 63)) return ULL_ERR_SHAPE;
    const dim3 grid((unsigned)((pitch + 63) / 64), (unsigned)((hd + 63) / 64), (unsigned)(B * H));
    hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const elem_t*)v, v_bs, v_ss, (elem_t*)vt, (int)S, (int)H,
                       (int)hd, (int)pitch);
    return ull_check_launch();
}

// image_encoder.py Block.forward :176-190 between norm1 and the projection, for the 14 x 14 windows: window_partition (zero padding
// included) + Attention (decomposed rel-pos) + window_unpartition, on tokens that stay in IMAGE order.  qkv [B*H*W, 3*nH*hd] rows
// q|k|v (ld elements apart), out [B*H*W, nH*hd]; pad_row = the q|k|v row of a padded token = the qkv bias (the reference pads the
// normalised activations with zeros, so Linear gives exactly its bias there); rel_pos_h / rel_pos_w [27, hd].  The GEMMs either
// side then run on H*W rows per image instead of the padded 25 * 196 (+19.6 % at 64 x 64); the two re-ordering passes and the V^T
// pass do not exist (one launch).
extern "C" int ULL_FN(ull_sam_window_attention_)(const void* qkv, int64_t ld, const void* pad_row, const void* rel_pos_h, const void* rel_pos_w,
                                             void* out, int64_t ldo, int64_t B, int64_t Hh, int64_t Ww, int64_t nH, int64_t hd, int64_t ws,
                                             float q_scale, const void* zeros, void* stream) {
    if (!qkv || !pad_row || !rel_pos_h || !rel_pos_w || !out || !zeros || B <= 0 || Hh <= 0 || Ww <= 0 || nH <= 0) return ULL_ERR_ARG;
    if (ws != 14 || hd != 80 || (ld & 7) || (ldo & 3) || ld < 3 * nH * hd || ldo < nH * hd) return ULL_ERR_SHAPE;
    const int nwy = (int)((Hh + ws - 1) / ws), nwx = (int)((Ww + ws - 1) / ws);
    const int C = (int)(nH * hd), S = (int)(ws * ws);
    const elem_t* base = (const elem_t*)qkv;
    AttnArgs a;
    a.Q = base; a.K = base + C; a.Vt = base + 2 * C; a.O = (elem_t*)out;
    a.key_mask = nullptr;
    a.q_bs = 0; a.q_hs = hd; a.q_ss = ld; a.k_bs = 0; a.k_hs = hd; a.k_ss = ld;
    a.vt_bs = 0; a.vt_hs = 0; a.vt_ds = 0; a.o_bs = 0; a.o_hs = hd; a.o_ss = ldo;
    a.B = (int)(B * nwy * nwx); a.H = (int)nH; a.Sq = S; a.Sk = S; a.hd = (int)hd; a.vt_len = 0;
    a.causal = 0; a.scale_mode = 0; a.scale = 1.0f;
    a.zeros = (const elem_t*)zeros;
    a.rel_h = (const elem_t*)rel_pos_h; a.rel_w = (const elem_t*)rel_pos_w; a.KH = (int)ws; a.KW = (int)ws; a.q_scale = q_scale;
    a.inv_kw = 1.0f / (float)ws;
    a.rel_mode = 2; a.win16 = 1; a.v_rows = 0;
    a.img_h = (int)Hh; a.img_w = (int)Ww; a.nwy = nwy; a.nwx = nwx;
    // the kernel's index arithmetic: (window, head) -> window -> image by multiply-high, row offsets inside a window in 32 bits
    const int64_t items = (int64_t)a.B * nH, dmax = nH > (int64_t)nwy * nwx ? nH : (int64_t)nwy * nwx;
    if (items * dmax >= (1LL << 32) || Ww * ld * 2 * ws >= (1LL << 31)) return ULL_ERR_SHAPE;
    auto magic = [](int64_t d) { return d <= 1 ? 0u : (uint32_t)(((1ULL << 32) + (uint64_t)d - 1) / (uint64_t)d); };
    a.mg_h = magic(nH); a.mg_nwx = magic(nwx); a.mg_nwy = magic(nwy); a.mg_nw = magic((int64_t)nwy * nwx);
    a.k_pad = (const elem_t*)pad_row + C; a.v_pad = (const elem_t*)pad_row + 2 * C;
    return dispatch_nt<128>(a, (hipStream_t)stream);
}
