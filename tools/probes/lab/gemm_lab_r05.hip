// MFMA GEMM for gfx950:  C[M,N] = epilogue( X[M,K] * W[N,K]^T )      (bf16 in, fp32 accumulate)
//
// This is every nn.Linear on the u-LLaVA forward path (reference: transformers LlamaAttention /
// LlamaMLP / CLIPAttention / CLIPMLP projections, models/ullava_core.py:117-129 vision_projector,
// :325 lm_head, models/ullava.py:86-118 seg/det projectors, SAM MLPBlock/Attention linears).
// Both operands are K-contiguous (activations [tokens, K]; nn.Linear weights [out, K]), so both
// are staged with the same code path.
//
// Design (MI355X-first, see DESIGN.md "GEMM"):
//   * 128x128x64 block tile, 256 threads = 4 waves in a 2x2 grid, 64x64 per wave, 16
//     v_mfma_f32_16x16x32_bf16 accumulators (64 VGPR) per wave.
//   * HBM -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip), double-buffered;
//     the next K-tile's DMA is in flight while the current one feeds the MFMAs (counted vmcnt +
//     raw s_barrier, never a drain inside the loop).
//   * LDS image is lane-linear (DMA constraint); bank conflicts are removed by an XOR swizzle of
//     the 16-byte chunk index with (row & 7), applied on the DMA *source* address and on the
//     ds_read_b128 address (same involution both sides).
//   * MFMA operands are swapped (A-operand = W rows, B-operand = X rows) so each lane ends up with 4
//     consecutive output features of one token -> 8-byte bf16 stores into row-major C.
//   * Fused epilogues reproduce the rounding points of the reference's bf16 graph (rnd()).
//   * 1-D grid with XCD-aware remap (block b runs on XCD b%8; each XCD gets a contiguous chunk of the
//     tile space, walked in GROUP_M-row groups so co-resident blocks share X / W panels in that L2).
#include "ull_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;           // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;         // X tile + W tile
constexpr int GEMM_LDS = 2 * BUF_BYTES;           // double buffered: 64 KiB -> 2 blocks / CU
constexpr int GROUP_M = 8;

// epilogue flag bits (mirrored in include/ullava_hip.h)
constexpr int EPI_BIAS = 1, EPI_ACT_SHIFT = 1, EPI_ACT_MASK = 3 << 1;  // act: 0 none 1 quick_gelu 2 gelu(erf) 3 relu
constexpr int EPI_RESID = 8, EPI_SWIGLU = 16, EPI_OUT_F32 = 32;
// operand stored tile-major [rows/256][K/64][256][64] (every 256 x 64 K-tile = 32 contiguous KiB); big kernel only
constexpr int EPI_W_TILED = 64, EPI_X_TILED = 128;
// The bias is added to the ROUNDED product: out = rnd(rnd(X W^T) + b).  at::linear fuses the bias (one rounding) only for 2-D and
// contiguous n-D inputs; a non-contiguous 3-D input goes through matmul + add_ (two roundings).  On the path that is layer 0 of
// SAM's TwoWayTransformer, whose `keys` are still the permuted NCHW view: cross_attn_token_to_image.{k,v}_proj and
// cross_attn_image_to_token.q_proj (transformer.py:92-93,151-182).
constexpr int EPI_BIAS_ROUNDED = 256;
constexpr int ULL_W4_FORCE_STAGED = 512;          // tools / tests: the 4-wave kernel's LDS-staged epilogue where the direct one would run

struct GemmArgs {
    const elem_t* X; const elem_t* W; void* C;
    const elem_t* bias; const elem_t* R;
    long ldx, ldw, ldc, ldr;
    int M, N, K, flags;
    int nbm, nbn;
    // stream-K tail (big kernel only): tiles [t_full, nbm*nbn) are each computed by `sk` blocks over disjoint K ranges that
    // dump raw fp32 accumulators into `ws` ([tile - t_full][slice][256][256]); gemm_splitk_finalize sums them + epilogue.
    int t_full, sk;
    float* ws;
    int group_m;                     // tile raster: blocks walk `group_m` M-tiles before stepping to the next N-tile
    // fused RoPE epilogue (ull_gemm_qkv_rope_*): output columns [0, rope_cols) are heads of 128 dims that get
    // transformers' apply_rotary_pos_emb with the per-token tables rope_cos / rope_sin [M, 64] (element type, already rounded)
    const elem_t* rope_cos; const elem_t* rope_sin;
    int rope_cols;
#ifdef ULL_GEMM_STAMPS
    unsigned long long* stamps;
#endif
};

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS[m0 .. m0+1024) (lane-linear).
// Issued through inline asm on purpose: hipcc models the builtin form as a pending LDS write and
// drains it with `s_waitcnt vmcnt(0)` in front of the next ds_read, which would serialise the DMA of
// tile k+1 against the MFMAs of tile k.  The kernel counts these loads itself (one vmcnt(0) at the top
// of each K-step, when the only DMA in flight is the tile about to be consumed).  M0 is saved/restored
// inside the statement because the compiler reserves it.
ULL_DEV void glds16(const void* gsrc, uint32_t lds_byte_addr /* wave-uniform */) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// Same with a scalar base and a 32-bit per-lane byte offset (saddr form).
ULL_DEV void glds16s(const void* sbase /* wave-uniform */, uint32_t voff, uint32_t lds_byte_addr /* wave-uniform */) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// Flag-dependent tail of every epilogue: finish 8 consecutive outputs of row m starting at column n (late bias, activation, residual,
// store).  The values arrive as the rounded Linear output (or the bare fp32 accumulator on the raw_f32 path).
struct EpiCtx {
    const GemmArgs& p;
    int act, n_out;
    bool out_f32, has_res, bias_late, c_al, r_al;
};
ULL_DEV EpiCtx epi_ctx(const GemmArgs& p, bool swiglu) {
    const int flags = p.flags;
    const bool out_f32 = flags & EPI_OUT_F32;
    return EpiCtx{p, (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT, swiglu ? p.N / 2 : p.N, out_f32, (flags & EPI_RESID) != 0,
                  !swiglu && (flags & EPI_BIAS) && (flags & EPI_BIAS_ROUNDED), out_f32 ? (p.ldc & 3) == 0 : (p.ldc & 7) == 0, (p.ldr & 7) == 0};
}
ULL_DEV void big_finish8(const EpiCtx& c, float (&a)[8], int m, int n, bool has_pre = false, uint4 res_pre = uint4{0, 0, 0, 0}) {
        if (c.bias_late) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < c.n_out) a[e] = rnd(a[e] + e2f(c.p.bias[n + e]));
        }
        if (c.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = act_quick_gelu_e(a[e]);
        } else if (c.act == 2) {
#pragma unroll                                           // (rolled, the dynamic index into a[] cost 32 us per tile: 3x the erf itself)
            for (int e = 0; e < 8; e += 2) {
                const f32x2_t r = act_gelu_erf2(f32x2_t{a[e], a[e + 1]});        // packed fp32 math, same results
                a[e] = rnd(r.x); a[e + 1] = rnd(r.y);
            }
        } else if (c.act == 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], 0.f);
        }
        const bool full = n + 8 <= c.n_out;
        if (c.has_res) {
            const elem_t* rp = c.p.R + (long)m * c.p.ldr + n;
            if (full && c.r_al) {
                float b[8];
                unpack8(has_pre ? res_pre : *(const uint4*)rp, b);       // res_pre: the caller fetched these 16 bytes ahead of time
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = rnd(b[e] + a[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < c.n_out) a[e] = rnd(e2f(rp[e]) + a[e]);
            }
        }
        if (c.out_f32) {
            float* cp = (float*)c.p.C + (long)m * c.p.ldc + n;
            if (full && c.c_al) {
                *(float4*)cp = make_float4(a[0], a[1], a[2], a[3]);
                *(float4*)(cp + 4) = make_float4(a[4], a[5], a[6], a[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < c.n_out) cp[e] = a[e];
            }
        } else {
            elem_t* cp = (elem_t*)c.p.C + (long)m * c.p.ldc + n;
            if (full && c.c_al) {
#if defined(ULL_ABL_NOSTORE)
                const uint4 v_ = pack8(a);
                asm volatile("" :: "v"(v_.x), "v"(v_.y), "v"(v_.z), "v"(v_.w), "v"(cp));
#else
                *(uint4*)cp = pack8(a);
#endif
            } else if (full) {
                // rows that are only 2-byte aligned (lm_head: V = 32011): 4-byte stores where the address allows, two 2-byte ends otherwise
                const uint4 pk = pack8(a);
                if ((((long)m * c.p.ldc + n) & 1) == 0) {
                    uint32_t* c4 = (uint32_t*)cp;
                    c4[0] = pk.x; c4[1] = pk.y; c4[2] = pk.z; c4[3] = pk.w;
                } else {
                    cp[0] = (elem_t)(pk.x & 0xffff);
                    uint32_t* c4 = (uint32_t*)(cp + 1);
                    c4[0] = (pk.x >> 16) | (pk.y << 16);
                    c4[1] = (pk.y >> 16) | (pk.z << 16);
                    c4[2] = (pk.z >> 16) | (pk.w << 16);
                    cp[7] = (elem_t)(pk.w >> 16);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < c.n_out) cp[e] = f2e(a[e]);
            }
        }
    }

// ---- LDS-staged epilogue shared by both kernels ------------------------------------------------------------------------
// A wave owns JT*16 rows (m) x 64 W-rows (n): acc[i][j][r] = D[n = nw0 + i*16 + 4*(lane>>4) + r][m = mrow0 + j*16 + (lane&15)].
// Two things were measured on the earlier register-direct form (profiles/r01_gemm_notes.md): scattered 16-row x 32-B stores,
// and -- much worse -- ~150 KiB of fully unrolled bias/activation/residual code per kernel (erf inlined 128 times), which
// streamed through the 64-KiB instruction cache once per tile (~15 us per 256x256 tile).  So the unrolled part (register-
// indexed accumulators) only adds the bias, rounds and parks the sub-tile in the wave's own LDS region `reg`
// (JT*16 rows x 144 B); everything flag-dependent runs in a small rolled loop that writes whole rows.
// The caller has passed a block barrier after its last LDS fragment read.
// ROPE (fused q|k|v projection): a head of 128 columns is parked by two neighbouring waves (64 columns each); after a block barrier
// every wave rotates its half against the partner's: out = rnd(x * cos) + rnd(-+x_partner * sin), rounded once more by the store --
// the three roundings of apply_rotary_pos_emb on 16-bit tensors (hf modeling_llama.py:129-159), same as rope_inplace_kernel.
// PHASE: 0 = park + finish (with the block barrier between them when ROPE); 1 = park only, 2 = finish only -- for a wave that owns both
// halves of a head (the 4-wave kernel) and parks them in two regions before finishing either.
// RAW1 (a region of JT*16 rows x 272 B): the fp32 path parks all rows in one pass, so that it splits into the two phases as well.
// UNR: unroll of the finish loop -- with one wave per SIMD nothing else hides the LDS / residual latencies of an iteration.
// SW2 (SwiGLU, wave tile 128 accumulator columns wide): the two 64-column halves park their 32 outputs side by side in ONE region
// (sw_half = 0 / 1 picks the side) and a single finish pass stores 64 outputs = whole 128-byte lines per row; finishing the halves
// separately wrote every output line in two 64-byte pieces at different times (WRITE_SIZE +30 %).  nw0 = the tile's first column.
// PERM (the 4-wave kernel's accumulator <-> column maps, see gemm256w4_kernel): 0 = acc[i][j][r] is column i*16 + 4*fg + r of the wave's 64;
// 1 = column 32*(i>>1) + 8*fg + 4*(i&1) + r; with SWIGLU, 1 = gate acc[2pp] / up acc[2pp+1] give output 8*fg + 4*pp + r of the half's 32.
template <bool SWIGLU, int JT, bool ROPE = false, int PHASE = 0, bool RAW1 = false, int UNR = 2, bool SW2 = false, int PERM = 0>
ULL_DEV void staged_epilogue(const GemmArgs& p, f32x4_t (&acc)[4][JT], char* reg, int lane, int mrow0, int nw0, const char* reg_partner = nullptr,
                             int sw_half = 0) {
    constexpr int ROWS = JT * 16;
    const int fr = lane & 15, fg = lane >> 4;
    const int flags = p.flags;
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
    const bool out_f32 = flags & EPI_OUT_F32;
    const bool has_res = flags & EPI_RESID;
    const int n_out = SWIGLU ? p.N / 2 : p.N;
    const bool bias_late = !SWIGLU && (flags & EPI_BIAS) && (flags & EPI_BIAS_ROUNDED);   // added in finish8, after the first rounding
    // fp32 result of the accumulator, never rounded.  (A caller that splits into phases without RAW1 routes this case to PHASE 0 itself.)
    const bool raw_f32 = (PHASE == 0 || RAW1) && !SWIGLU && out_f32 && !act && !has_res && !bias_late;
    const bool c_al = out_f32 ? (p.ldc & 3) == 0 : (p.ldc & 7) == 0;   // 16-byte row alignment of C / R
    const bool r_al = (p.ldr & 7) == 0;
    const int ncol0 = SWIGLU ? nw0 / 2 : nw0;

    const EpiCtx ectx = epi_ctx(p, SWIGLU);
    auto finish8 = [&](float (&a)[8], int m, int n) { big_finish8(ectx, a, m, n); };
    auto col_of = [&](int i) { return PERM ? 32 * (i >> 1) + 8 * fg + 4 * (i & 1) : i * 16 + fg * 4; };   // first of the lane's 4 columns
    float bias_v[4][4];                                    // -0.0f: x + (-0.0f) == x bit-for-bit when there is no bias
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nw0 + col_of(i) + r;
            bias_v[i][r] = (!SWIGLU && (flags & EPI_BIAS) && !bias_late && n < p.N) ? e2f(p.bias[n]) : -0.0f;
        }

    if (!raw_f32) {
        constexpr int WCOLS = (SWIGLU && !SW2) ? 32 : 64;  // output columns of the region
        constexpr int PITCH = WCOLS * 2 + 16;              // bytes; +16 keeps 16-byte alignment and spreads banks
        if constexpr (PHASE != 2) {
        // (two copies: without a bias the 4 adds per accumulator -- 256 instructions per wave of the 4-wave kernel -- are not issued)
        auto park_all = [&](auto with_bias) {
        constexpr bool WB = decltype(with_bias)::value;
#pragma clang loop unroll(full)
        for (int j = 0; j < JT; ++j) {
            if constexpr (SWIGLU) {
                // W rows are interleaved in 16-row groups: [gate 16g..16g+15][up 16g..16g+15]
#pragma clang loop unroll(full)
                for (int ip = 0; ip < 2; ++ip) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = rnd(acc[2 * ip][j][r]);          // gate_proj output (bf16 tensor)
                        const float u = rnd(acc[2 * ip + 1][j][r]);      // up_proj output (bf16 tensor)
                        v[r] = rnd(act_silu(g)) * u;                     // silu -> bf16, product -> bf16 (by the pack)
                    }
                    uint2 o;
                    o.x = pack2e(v[0], v[1]);
                    o.y = pack2e(v[2], v[3]);
                    *(uint2*)(reg + (j * 16 + fr) * PITCH + (SW2 ? sw_half * 64 : 0) + (PERM ? 8 * fg + 4 * ip : ip * 16 + fg * 4) * 2) = o;
                }
            } else {
#pragma clang loop unroll(full)
                for (int i = 0; i < 4; ++i) {
                    uint2 o;                                             // the Linear's bf16 output
                    if constexpr (WB) {
                        o.x = pack2e(acc[i][j][0] + bias_v[i][0], acc[i][j][1] + bias_v[i][1]);
                        o.y = pack2e(acc[i][j][2] + bias_v[i][2], acc[i][j][3] + bias_v[i][3]);
                    } else {
                        o.x = pack2e(acc[i][j][0], acc[i][j][1]);
                        o.y = pack2e(acc[i][j][2], acc[i][j][3]);
                    }
                    *(uint2*)(reg + (j * 16 + fr) * PITCH + col_of(i) * 2) = o;
                }
            }
        }
        };
        if (!SWIGLU && (flags & EPI_BIAS) && !bias_late) park_all(std::true_type{});
        else park_all(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own region only: no block barrier needed
        }
        if constexpr (PHASE == 1) return;
        if constexpr (ROPE && PHASE == 0) __builtin_amdgcn_s_barrier();   // ... unless the partner wave's half of the head is read below
        constexpr int LPR = WCOLS / 8;                         // lanes per row (16 B each)
        constexpr int RPI = 64 / LPR;                          // rows per wave-instruction
        // UNR iterations at a time: first every load of the group (LDS rows, RoPE table rows, residual rows -- from clamped, always
        // valid addresses, so nothing branches), then the arithmetic and the stores.  With one wave per SIMD an iteration that loads,
        // waits, computes and stores exposes every latency in turn (the RoPE epilogue cost 116 us per qkv launch that way).
        constexpr int NIT = ROWS / RPI;
        static_assert(NIT % UNR == 0, "finish loop groups");
        const int c8 = lane % LPR;
        const int n = ncol0 + c8 * 8;
        const bool rope_on = ROPE && n < p.rope_cols;
        const bool res_pre = has_res && r_al && n + 8 <= n_out;
        // The common cases -- a 16-bit store of whole 16-byte groups with no activation: plain, + residual, + RoPE -- run in a loop of
        // their own: the general tail (big_finish8) tests six run-time flags per group of 8 outputs, and those scalar branches, not the
        // stores, were most of the 5.9 us (8 waves) / 10.7 us (4 waves) an epilogue cost without its stores (tools/gemm_fixed.py with
        // the NOEPI / NOSTORE ablations).
        // (the residual takes this loop on the 4-wave kernel only: with groups of 2 the 8-wave kernel measured 2 us per tile round slower
        // here than on the general path)
        // An activation takes it on the 8-wave kernel (where the ViT MLPs run), without a residual (no Linear on the path has both).
        const bool fast_act = act && UNR <= 2 && !has_res && !ROPE && !SWIGLU;
        const bool fast = (!act || fast_act) && !bias_late && !out_f32 && c_al && (!has_res || (r_al && UNR >= 4)) && ncol0 + WCOLS <= n_out;
        if (fast) {
            elem_t* cbase = (elem_t*)p.C + n;
            const elem_t* rbase = p.R + n;
            // compiled twice (with / without the residual) so that the residual loads are unconditional statements of the load phase:
            // under a run-time `if (has_res)` the compiler merged them with their use and every group waited out its own HBM latency
            auto fast_loop = [&](auto with_res, auto act_kind) {
                constexpr bool RES = decltype(with_res)::value;
                constexpr int ACT = decltype(act_kind)::value;          // 0 none, 1 QuickGELU, 2 GELU(erf), 3 ReLU
#pragma unroll 1
                for (int it0 = 0; it0 < NIT; it0 += UNR) {
                    uint4 va[UNR], vb[UNR], vc[UNR], vs[UNR], vr[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {              // the long-latency loads first
                        const long mc = min(mrow0 + (it0 + u) * RPI + lane / LPR, p.M - 1);
                        if constexpr (RES) vr[u] = *(const uint4*)(rbase + mc * p.ldr);
                        if constexpr (ROPE) {
                            vc[u] = *(const uint4*)(p.rope_cos + mc * 64 + c8 * 8);
                            vs[u] = *(const uint4*)(p.rope_sin + mc * 64 + c8 * 8);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int row = (it0 + u) * RPI + lane / LPR;
                        va[u] = *(const uint4*)(reg + row * PITCH + c8 * 16);
                        if constexpr (ROPE) vb[u] = *(const uint4*)(reg_partner + row * PITCH + c8 * 16);
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int m = mrow0 + (it0 + u) * RPI + lane / LPR;
                        uint4 o = va[u];
                        if (ROPE ? rope_on : false) {
                            float a[8], b[8], cs[8], sn[8];
                            unpack8(va[u], a); unpack8(vb[u], b); unpack8(vc[u], cs); unpack8(vs[u], sn);
                            const bool first_half = (n & 64) == 0;
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = rnd(a[e] * cs[e]) + rnd((first_half ? -b[e] : b[e]) * sn[e]);
                            o = pack8(a);
                        }
                        if constexpr (ACT != 0) {
                            float a[8];
                            unpack8(o, a);
                            if constexpr (ACT == 2) {
#pragma unroll
                                for (int e = 0; e < 8; e += 2) {
                                    const f32x2_t r = act_gelu_erf2(f32x2_t{a[e], a[e + 1]});
                                    a[e] = r.x; a[e + 1] = r.y;          // (pack8 rounds)
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) a[e] = ACT == 1 ? act_quick_gelu_e(a[e]) : fmaxf(a[e], 0.f);
                            }
                            o = pack8(a);
                        }
                        if constexpr (RES) {
                            float a[8], b[8];
                            unpack8(o, a); unpack8(vr[u], b);
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = rnd(b[e] + a[e]);
                            o = pack8(a);
                        }
#if defined(ULL_ABL_NOSTORE)
                        asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w), "v"(cbase));
#else
                        if (m < p.M) *(uint4*)(cbase + (long)m * p.ldc) = o;
#endif
                    }
                }
            };
            using I0 = std::integral_constant<int, 0>;
            if (has_res) fast_loop(std::true_type{}, I0{});
            else if constexpr (UNR <= 2 && !ROPE && !SWIGLU) {
                if (act == 1) fast_loop(std::false_type{}, std::integral_constant<int, 1>{});
                else if (act == 2) fast_loop(std::false_type{}, std::integral_constant<int, 2>{});
                else if (act == 3) fast_loop(std::false_type{}, std::integral_constant<int, 3>{});
                else fast_loop(std::false_type{}, I0{});
            } else fast_loop(std::false_type{}, I0{});
        } else
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += UNR) {
            uint4 va[UNR], vb[UNR], vc[UNR], vs[UNR], vr[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int row = (it0 + u) * RPI + lane / LPR;
                const long mc = min(mrow0 + row, p.M - 1);
                va[u] = *(const uint4*)(reg + row * PITCH + c8 * 16);
                if constexpr (ROPE) {
                    vb[u] = *(const uint4*)(reg_partner + row * PITCH + c8 * 16);
                    vc[u] = *(const uint4*)(p.rope_cos + mc * 64 + c8 * 8);     // always a valid address (64 columns, clamped row): the
                    vs[u] = *(const uint4*)(p.rope_sin + mc * 64 + c8 * 8);     // v columns load and ignore them rather than branch
                }
                if (res_pre) vr[u] = *(const uint4*)(p.R + mc * p.ldr + n);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int m = mrow0 + (it0 + u) * RPI + lane / LPR;
                if (m >= p.M || n >= n_out) continue;
                float a[8];
                unpack8(va[u], a);
                if constexpr (ROPE) {
                    if (rope_on) {
                        float b[8], cs[8], sn[8];
                        unpack8(vb[u], b);
                        unpack8(vc[u], cs);
                        unpack8(vs[u], sn);
                        const bool first_half = (n & 64) == 0;         // dims 0..63 of the head: rotate_half contributes -x[d + 64]
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] = rnd(a[e] * cs[e]) + rnd((first_half ? -b[e] : b[e]) * sn[e]);
                    }
                }
                big_finish8(ectx, a, m, n, res_pre, vr[u]);
            }
        }
    } else if constexpr (PHASE == 0 || RAW1) {
        // fp32 output of the bare accumulator (+bias): two passes of ROWS/2 rows x 64 fp32 columns through the same region
        constexpr int PITCH = 64 * 4 + 16;
        constexpr int NPASS = RAW1 ? 1 : 2;
        constexpr int JH = JT / NPASS;
        static_assert(RAW1 || PHASE == 0, "the two-pass fp32 path parks and finishes in one call");
#pragma clang loop unroll(full)
        for (int h = 0; h < NPASS; ++h) {
            if constexpr (PHASE != 2) {
#pragma clang loop unroll(full)
                for (int jj = 0; jj < JH; ++jj) {
#pragma clang loop unroll(full)
                    for (int i = 0; i < 4; ++i) {
                        f32x4_t o = acc[i][h * JH + jj];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] += bias_v[i][r];
                        *(f32x4_t*)(reg + (jj * 16 + fr) * PITCH + col_of(i) * 4) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (PHASE == 1) return;
#pragma unroll UNR
            for (int it = 0; it < JH * 2; ++it) {
                const int row = it * 8 + (lane >> 3), c8 = lane & 7;
                const int m = mrow0 + h * (JH * 16) + row, n = ncol0 + c8 * 8;
                if (m >= p.M || n >= n_out) continue;
                const f32x4_t lo = *(const f32x4_t*)(reg + row * PITCH + c8 * 32);
                const f32x4_t hi = *(const f32x4_t*)(reg + row * PITCH + c8 * 32 + 16);
                float a[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                finish8(a, m, n);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads of this pass done before the next pass overwrites
        }
    }
}

template <bool SWIGLU, bool ROPE = false>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;

    // ---- block -> (tile, K slice): XCD-contiguous chunks, grouped raster inside --------------
    // split-K (p.sk > 1: few tiles and a long K -- the o_proj / down_proj of a single-image prefill are 96 tiles of 64 / 172 K-steps on a
    // chip that holds 512 blocks): slice s = blockIdx / tiles works on K-steps [s nk / sk, (s + 1) nk / sk) and leaves its raw fp32
    // accumulators in p.ws [tile][slice][128][128]; gemm128_splitk_finalize sums the slices in order and applies the epilogue.
    const int nwg = p.nbm * p.nbn;
    const int slice = blockIdx.x / nwg;
    int bid = blockIdx.x - slice * nwg;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective for any nwg
    }
    const int per_group = GROUP_M * p.nbn;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(p.nbm - first_m, GROUP_M);
    const int bm = first_m + (bid % per_group) % gsz;
    const int bn = (bid % per_group) / gsz;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- DMA source pointers (loop invariant apart from the K offset) --------------------------
    // wave w, step i stages rows (w*4+i)*8 .. +8 of each tile; lane -> (row l>>3, physical chunk l&7),
    // which must fetch LOGICAL chunk (l&7) ^ (row&7) so that reads can undo the swizzle.
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const elem_t* xsrc[4];
    const elem_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;
        xsrc[i] = p.X + (long)min(m0 + r, p.M - 1) * p.ldx + schunk * 8;
        wsrc[i] = p.W + (long)min(n0 + r, p.N - 1) * p.ldw + schunk * 8;
    }
    const int stage_off = wave * 4 * 8 * (BK * 2);   // byte offset of this wave's first 1-KiB piece

    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    auto stage = [&](int buf, int kt) {
        const uint32_t bx = lds_base + buf * BUF_BYTES + stage_off;
        const uint32_t bw = bx + TILE_BYTES;
        const long ko = (long)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(xsrc[i] + ko, bx + i * 1024);
            glds16(wsrc[i] + ko, bw + i * 1024);
        }
    };

    // ---- fragment read offsets ------------------------------------------------------------
    const int frow = lane & 15, fgrp = lane >> 4;
    int swz[2];
    swz[0] = ((0 + fgrp) ^ (lane & 7)) << 4;
    swz[1] = ((4 + fgrp) ^ (lane & 7)) << 4;
    const int xrow_off = (wm * 64 + frow) * (BK * 2);
    const int wrow_off = TILE_BYTES + (wn * 64 + frow) * (BK * 2);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk_all = p.K / BK;
    const int k_first = p.sk > 1 ? (int)((long)nk_all * slice / p.sk) : 0;
    const int nk = p.sk > 1 ? (int)((long)nk_all * (slice + 1) / p.sk) : nk_all;
    stage(k_first & 1, k_first);
    for (int kt = k_first; kt < nk; ++kt) {
        const int cur = kt & 1;
        // Tile kt was issued one iteration ago and is the only DMA in flight for this wave.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // One barrier per K-step: (a) every wave's pieces of tile kt are in LDS, (b) every wave has
        // finished reading buffer cur^1 (compute of tile kt-1), so it can be overwritten below.
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);          // DMA of tile kt+1 overlaps the MFMAs of tile kt
        const char* base = smem + cur * BUF_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *(const uint4*)(base + wrow_off + i * 16 * (BK * 2) + swz[kk]);
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = *(const uint4*)(base + xrow_off + j * 16 * (BK * 2) + swz[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
        }
    }

    // ---- epilogue: acc[i][j][r] = D[n = n0 + wn*64 + i*16 + 4*(l>>4) + r][m = m0 + wm*64 + j*16 + (l&15)] --------------
    if (p.sk > 1) {                                        // a K slice: raw accumulators to the workspace, row-major [m][n] per (tile, slice)
        float* slab = p.ws + ((long)(bm * p.nbn + bn) * p.sk + slice) * (BM * BN);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4_t*)(slab + (wm * 64 + j * 16 + (lane & 15)) * BN + wn * 64 + i * 16 + 4 * (lane >> 4)) = acc[i][j];
        return;
    }
    __builtin_amdgcn_s_barrier();                          // every wave has consumed the last K-tile: LDS is free
    staged_epilogue<SWIGLU, 4, ROPE>(p, acc, smem + wave * (64 * 144), lane, m0 + wm * 64, n0 + wn * 64, smem + (wave ^ 1) * (64 * 144));
}


// ---- fused ViT patchify: Conv2d(C, N, kernel = stride = ps) as a GEMM whose A tile is DMA'd straight from the image ----------
// reference: transformers CLIPVisionEmbeddings.patch_embedding (models/ullava_core.py:131-159 encode_image) and SAM PatchEmbed
// (segment_anything/modeling/image_encoder.py:395-427).  The old path materialised im2col(image) in HBM first (+24 us and +47 MB
// at B=32, 336^2).  Here K is ordered (c, ky, kx) with kx padded to 16: K' = C*ps*16 (+ zero segments up to a multiple of 64).
// One (c, ky) segment of a patch is 16 bf16 = two 16-byte LDS-DMA chunks read from img[b][c][py*ps+ky][px*ps .. +16): for ps = 14
// the last two elements are the next patch's first pixels -- the packed weight W' is zero there, so they do not matter.
// LDS-DMA takes 2-byte-aligned source addresses (tools/probes/dma_align.hip), so no thread ever touches the pixels.
// The one chunk that would read past the end of the buffer (last image, last channel, last pixel row, last patch: 12 real bytes +
// 4 beyond) goes through registers instead of the DMA.
struct PatchArgs {
    const elem_t* img; const elem_t* zeros; const elem_t* img_end;
    int C, H, W, ps, gw, gh;          // image geometry; gw x gh patches per image
    int nseg;                         // C * ps real (c, ky) segments; segments >= nseg read zeros
    int xcd_raster;                   // patchify_strip_kernel: all N-tiles of a strip on one XCD (needs nbm % 8 == 0)
};

template <int DUMMY>
__global__ __launch_bounds__(256, 2) void patchify_gemm_kernel(GemmArgs p, PatchArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int nwg = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + k;
    }
    const int per_group = GROUP_M * p.nbn;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(p.nbm - first_m, GROUP_M);
    const int bm = first_m + (bid % per_group) % gsz;
    const int bn = (bid % per_group) / gsz;
    const int m0 = bm * BM, n0 = bn * BN;

    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;                 // logical 16-byte chunk of the 128-byte K-tile row this lane fetches
    const int seg = schunk >> 1, half = schunk & 1;       // (c, ky) segment within the K-tile, first / second 8 kx
    const elem_t* xbase[4];                               // pixel (b, c = 0, y = py*ps, x = px*ps) of the patch this lane stages
    const elem_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;
        const int m = min(m0 + r, p.M - 1);
        const int px = m % q.gw, py = (m / q.gw) % q.gh, b = m / (q.gw * q.gh);
        xbase[i] = q.img + (((long)b * q.C * q.H + (long)py * q.ps) * q.W + px * q.ps) + half * 8;
        wsrc[i] = p.W + (long)min(n0 + r, p.N - 1) * p.ldw + schunk * 8;
    }
    const int stage_off = wave * 4 * 8 * (BK * 2);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    auto stage = [&](int buf, int kt) {
        const uint32_t bx = lds_base + buf * BUF_BYTES + stage_off;
        const uint32_t bw = bx + TILE_BYTES;
        const int idx = kt * 4 + seg;                     // global (c, ky) segment index of this lane's chunk
        const int c = idx / q.ps, ky = idx - c * q.ps;
        const long xoff = ((long)c * q.H + ky) * q.W;
        const bool real = idx < q.nseg;
        const long ko = (long)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const elem_t* sp = real ? xbase[i] + xoff : q.zeros;
            if (real && sp + 8 > q.img_end) {
                // the one chunk that would read past the buffer (last image / channel / pixel row / patch, second half): its 6 real
                // pixels come through registers, the 2 pad slots are zero (W' is zero there anyway, but garbage could be NaN bits)
                const uint32_t* s32 = (const uint32_t*)sp;
                const uint4 v = make_uint4(s32[0], s32[1], s32[2], 0u);
                *(uint4*)(smem + buf * BUF_BYTES + stage_off + i * 1024 + lane * 16) = v;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                glds16(sp, bx + i * 1024);
            }
            glds16(wsrc[i] + ko, bw + i * 1024);
        }
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    int swz[2];
    swz[0] = ((0 + fgrp) ^ (lane & 7)) << 4;
    swz[1] = ((4 + fgrp) ^ (lane & 7)) << 4;
    const int xrow_off = (wm * 64 + frow) * (BK * 2);
    const int wrow_off = TILE_BYTES + (wn * 64 + frow) * (BK * 2);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* base = smem + cur * BUF_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *(const uint4*)(base + wrow_off + i * 16 * (BK * 2) + swz[kk]);
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = *(const uint4*)(base + xrow_off + j * 16 * (BK * 2) + swz[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
        }
    }
    __builtin_amdgcn_s_barrier();
    staged_epilogue<false, 4>(p, acc, smem + wave * (64 * 144), lane, m0 + wm * 64, n0 + wn * 64);
}

// =============================================================================================================
// Large-shape kernel: 256x256 block tile, BK = 64, two 64-KiB LDS slots (128 KiB), 512 threads = 8 waves (2 x 4),
// 128(m) x 64(n) per wave = 32 accumulators (128 VGPR).  Measured motivation (profiles/r01_gemm_notes.md): ablating the
// MFMAs out of either kernel leaves the run time almost unchanged -- the GEMM is bound by the HBM/L2 -> LDS staging path
// (~85 GB/s per CU peak through the vector L1, ~0.2 us L2-hit and ~1-2 us first-touch latency), so this kernel doubles
// the FLOPs per staged byte (128 vs 64 FLOP/B) and keeps a whole K-step of DMA in flight behind the MFMAs.
//   * MFMA fragments are double-buffered in REGISTERS at half-K-step granularity: while the MFMAs of (tile k, half h)
//     issue, the fragments of the next half are read from LDS.  Tile k therefore lives in registers while tile k+1 is
//     being read from one LDS slot and tile k+2 is being DMA'd into the other: two slots give a prefetch distance of 2.
//   * two barriers per BK=64 step and a hand-dealt slot schedule (at the loop): one MFMA per slot, the fragment reads and the DMA
//     pieces of tile k+2 spread between them so that the memory pipe sees an even stream and is never drained.
#ifdef ULL_GEMM_STAMPS      // debug build only (tools/gemm_tile_phases.py): per-block timestamps of the 4-wave kernel's phases
unsigned long long* ull_stamp_host_ptr = nullptr;            // set by ull_debug_gemm_stamps_*, copied into GemmArgs::stamps at launch
ULL_DEV void ull_stamp(unsigned long long* buf, int slot) {
    if (buf && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) {     // wave 0, every lane the same store (a wave-uniform branch)
        unsigned long long t = __builtin_amdgcn_s_memrealtime();            // 100 MHz
        if (slot == 7) {                                                    // where the block ran: XCC id << 32 | HW_ID
            uint32_t hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(4, 0, 32)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(20, 0, 32)" : "=s"(xcc));
            t = ((unsigned long long)(xcc & 0xf) << 32) | hw;
        }
        buf[(size_t)blockIdx.x * 8 + slot] = t;
    }
}
#define ULL_STAMP(slot) ull_stamp(p.stamps, slot)
#else
#define ULL_STAMP(slot) ((void)0)
#endif

namespace big {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OP_BYTES = BM * BK * 2;             // 32 KiB per operand per slot
constexpr int SLOT_BYTES = 2 * OP_BYTES;          // 64 KiB
constexpr int LDS_BYTES = 8 * 128 * (64 * 2 + 16);   // 144 KiB: two 64-KiB K-tile slots, re-used by the 8 x 18-KiB epilogue regions
constexpr int LDS_BYTES_W4 = LDS_BYTES + 16 * 1024;   // + the dump of the last two steps' prefetches: all 160 KiB of the CU

struct Frags { uint4 w[4]; uint4 x[8]; };

template <bool SWIGLU, bool ROPE = false>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    // Blocks [0, t_full) own one whole tile each (t_full is a multiple of the CU count, so those rounds are full); the
    // remaining tiles -- which would otherwise occupy a few CUs for one more whole round -- are split `sk` ways along K.
    int bid = blockIdx.x;
    int slice = 0;
    const bool split = bid >= p.t_full;
    if (!split) {
        const int nwg = p.t_full;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective for any nwg
    } else {
        const int r = bid - p.t_full;
        bid = p.t_full + r / p.sk;
        slice = r % p.sk;
    }
    const int per_group = p.group_m * p.nbn;
    const int gid = bid / per_group;
    const int first_m = gid * p.group_m;
    const int gsz = min(p.nbm - first_m, p.group_m);
    const int bm = first_m + (bid % per_group) % gsz;
    const int bn = (bid % per_group) / gsz;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk_total = p.K / BK;
    const bool w_tiled = p.flags & EPI_W_TILED, x_tiled = p.flags & EPI_X_TILED;

    // DMA: a 1-KiB piece = 8 rows x 128 B; wave w stages pieces 4w..4w+3 of X and of W.
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const elem_t* xsrc[4];
    const elem_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;
        xsrc[i] = x_tiled ? p.X + (((long)bm * nk_total) * BM + r) * BK + schunk * 8
                          : p.X + (long)min(m0 + r, p.M - 1) * p.ldx + schunk * 8;
        wsrc[i] = w_tiled ? p.W + (((long)bn * nk_total) * BN + r) * BK + schunk * 8
                          : p.W + (long)min(n0 + r, p.N - 1) * p.ldw + schunk * 8;
    }
    const long xstep = x_tiled ? BM * BK : BK, wstep = w_tiled ? BN * BK : BK;    // elements between consecutive K-tiles
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const uint32_t piece_off = wave * 4 * 1024;
    auto stage = [&](int kt) {
        const uint32_t bx = lds_base + (kt & 1) * SLOT_BYTES + piece_off;
        const long kox = (long)kt * xstep, kow = (long)kt * wstep;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(xsrc[i] + kox, bx + i * 1024);
            glds16(wsrc[i] + kow, bx + OP_BYTES + i * 1024);
        }
    };

    const int fr = lane & 15, fg = lane >> 4;
    int swz[2];
    swz[0] = ((0 + fg) ^ (lane & 7)) << 4;
    swz[1] = ((4 + fg) ^ (lane & 7)) << 4;
    const int xoff = (wm * 128 + fr) * (BK * 2);
    const int woff = OP_BYTES + (wn * 64 + fr) * (BK * 2);
    auto read_frags = [&](int kt, int kk, Frags& f) {
        const char* base = smem + (kt & 1) * SLOT_BYTES + swz[kk];
#pragma unroll
        for (int i = 0; i < 4; ++i) f.w[i] = *(const uint4*)(base + woff + i * 16 * (BK * 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) f.x[j] = *(const uint4*)(base + xoff + j * 16 * (BK * 2));
    };

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int nk = p.K / BK;                                // >= 2 (host dispatch, also per K-slice)
    if (split) {
        const int kt0 = (int)((long)nk * slice / p.sk), kt1 = (int)((long)nk * (slice + 1) / p.sk);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xsrc[i] += (long)kt0 * xstep; wsrc[i] += (long)kt0 * wstep; }
        nk = kt1 - kt0;
    }
    // Same step schedule as the 4-wave kernel (see there), 64 MFMA slots per wave and step: the 12 fragment reads of the tile's second
    // half first, barrier A, the 8 DMA pieces of tile kt+2 dealt out one per 6 slots (a burst of all 64 pieces of a CU right behind
    // the barrier measured 3-7 % slower end to end: the memory pipe wants an even stream), barrier B with the newest pieces still in
    // flight, then the first-half fragments of tile kt+1.  The last two steps re-fetch the last tile into a 16-KiB dump.
#ifndef ULL_W8_BAR_A
#define ULL_W8_BAR_A 14
#endif
#ifndef ULL_W8_DMA_STRIDE
#define ULL_W8_DMA_STRIDE 6
#endif
#ifndef ULL_W8_BAR_B
#define ULL_W8_BAR_B 40
#endif
#ifndef ULL_W8_FA_STRIDE
#define ULL_W8_FA_STRIDE 2
#endif
    constexpr int W8_BAR_A = ULL_W8_BAR_A, W8_DMA_STRIDE = ULL_W8_DMA_STRIDE, W8_BAR_B = ULL_W8_BAR_B, W8_FA_STRIDE = ULL_W8_FA_STRIDE;
    constexpr int W8_ISSUED = (W8_BAR_B - W8_BAR_A + W8_DMA_STRIDE - 1) / W8_DMA_STRIDE;
    constexpr int W8_INFLIGHT = W8_ISSUED > 8 ? 8 : W8_ISSUED;
    static_assert(W8_BAR_A >= 12 && W8_BAR_B >= W8_BAR_A && W8_BAR_A + 7 * W8_DMA_STRIDE < 64 && W8_BAR_B + 11 * W8_FA_STRIDE < 64, "schedule");
    constexpr int USE_ORDER[12] = {0, 4, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11};    // w0, x0, w1..w3, x1..x7
    auto read1 = [&](int kt, int kk, Frags& f, int r) {
        const char* base = smem + (kt & 1) * SLOT_BYTES + swz[kk];
        if (r < 4) f.w[r] = *(const uint4*)(base + woff + r * 16 * (BK * 2));
        else f.x[r - 4] = *(const uint4*)(base + xoff + (r - 4) * 16 * (BK * 2));
    };
    auto mma1 = [&](const Frags& f, int s) {
        const int j = s >> 2, i = s & 3;
#if !defined(ULL_ABL_NOMMA)
        acc[i][j] = mfma16(f.w[i], f.x[j], acc[i][j]);
#endif
    };
    stage(0);
    stage(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile 0 landed, tile 1 in flight
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;                                     // fa: half 0 of the current tile, fb: half 1
    read_frags(0, 0, fa);
    const uint32_t dump = lds_base + LDS_BYTES;
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nk; ++kt) {
        const bool with_dma = kt + 2 < nk;
        const int ktd = with_dma ? kt + 2 : nk - 1;
        const long kox = (long)ktd * xstep, kow = (long)ktd * wstep;
        const uint32_t bx = with_dma ? lds_base + (kt & 1) * SLOT_BYTES + piece_off : dump;
        const uint32_t bw = with_dma ? bx + OP_BYTES : dump + 4096;
#pragma clang loop unroll(full)
        for (int s = 0; s < 64; ++s) {
            if (s == W8_BAR_A) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (s == W8_BAR_B) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W8_INFLIGHT) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (s < 12) read1(kt, 1, fb, s);
            if (s >= W8_BAR_B && (s - W8_BAR_B) % W8_FA_STRIDE == 0 && (s - W8_BAR_B) / W8_FA_STRIDE < 12)
                read1(kt + 1, 0, fa, USE_ORDER[(s - W8_BAR_B) / W8_FA_STRIDE]);
#if !defined(ULL_ABL_NODMA)
            if (s >= W8_BAR_A && (s - W8_BAR_A) % W8_DMA_STRIDE == 0 && (s - W8_BAR_A) / W8_DMA_STRIDE < 8) {
                const int pc = (s - W8_BAR_A) / W8_DMA_STRIDE;
                if (pc < 4) glds16(xsrc[pc] + kox, bx + pc * 1024);
                else glds16(wsrc[pc - 4] + kow, bw + (pc - 4) * 1024);
            }
#endif
            mma1(s < 32 ? fa : fb, s & 31);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // (the dump's pieces may still be landing: they touch nothing the epilogue uses, and s_endpgm waits for them)

#if defined(ULL_ABL_NOEPI)
    if (p.M > 0) return;
#endif
    // ---- epilogue: acc[i][j][r] = D[n = n0 + wn*64 + i*16 + 4*fg + r][m = m0 + wm*128 + j*16 + fr] ----------
    if (split) {
        float* slab = p.ws + ((long)(bid - p.t_full) * p.sk + slice) * (BM * BN);
#pragma clang loop unroll(full)
        for (int j = 0; j < 8; ++j)
#pragma clang loop unroll(full)
            for (int i = 0; i < 4; ++i)
                *(f32x4_t*)(slab + (wm * 128 + j * 16 + fr) * BN + wn * 64 + i * 16 + fg * 4) = acc[i][j];
        return;
    }
    __builtin_amdgcn_s_barrier();                          // every wave has consumed the last K-tile: LDS is free
    staged_epilogue<SWIGLU, 8, ROPE>(p, acc, smem + wave * (128 * 144), lane, m0 + wm * 128, n0 + wn * 64, smem + (wave ^ 1) * (128 * 144));
}

// (A persistent form of this kernel -- one workgroup per CU walking its tiles, K-tile 0 of the next tile prefetched by step nk-2, the
// epilogue staged through the slot the last step frees, K-tile 1 issued behind it -- was built and measured twice, rounds 1 and 2:
// bit-identical results, 5-8 % SLOWER on every shape from K = 1024 to K = 11008.  The hardware's block hand-over plus a cold
// prologue costs less than an in-kernel tile turn-around with its extra barrier and branchy DMA slots; profiles/r02_gemm_notes.md.)

// ---- 4-wave form of the same tile ---------------------------------------------------------------------------------------------
// Same 256x256x64 block tile, LDS layout, raster and stream-K tail, but 256 threads = 4 waves (2 x 2), 128(m) x 128(n) per wave:
// 64 accumulators (256 registers; one wave per SIMD owns the SIMD's whole 512-entry register file) and 16 fragment reads per 64 MFMAs
// instead of 12 per 32 -- a K-step reads 128 KiB of fragments from LDS per CU instead of 192 KiB.  Counters on the gate/up launch
// (profiles/r02_gemm_notes.md) put the 8-wave kernel and the vendor BLAS's kernel at the same L2 requests, hit rate and fabric bytes,
// and differ in exactly this: LDS instructions (8.6e7 vs 5.8e7), VALU/SALU instructions and the share of cycles the MFMA pipe is busy
// (68 % vs 84 %), so the staging path was never the bound -- the CU-side instruction stream was.
// With a single wave per SIMD nothing hides an issue stall, so the K-step is scheduled by hand in chunks of 4 MFMAs with a
// sched_barrier between chunks (schedule: at the loop).
// DMA addressing: scalar base (advanced per K-step) + one 32-bit VGPR offset per piece.
struct Frags4 { uint4 w[8]; uint4 x[8]; };

// The accumulators are pinned to the AGPR half of the register file and accumulated in place through inline asm: with the builtin,
// hipcc's allocator splits the 64 accumulators between VGPRs and AGPRs under the 512-register pressure and shuffles them with
// ~340 v_accvgpr moves per K-step.  (The asm is opaque to the hazard recognizer: the K-loop never re-reads an accumulator sooner than
// 64 MFMAs later, and the epilogue waits out the last MFMA explicitly.)
#ifdef ULL_ELEM_F16
#define ULL_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#else
#define ULL_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#endif
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
ULL_DEV void mfma16_inplace(f32x4_t& c, const uint4& a, const uint4& b) {
    const u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
#if !defined(ULL_ABL_NOMMA)
    asm volatile(ULL_MFMA16_ASM " %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
#else
    asm volatile("" : "+a"(c) : "v"(av), "v"(bv));
#endif
}


// ---- the 4-wave kernel's W-row permutation and its register-direct epilogue ---------------------------------------------------
// Which physical W row feeds MFMA row q = 4*a + b (a = q >> 2, b = q & 3) of block i of a 64-row half is a free choice: the MFMA does
// not care, only the epilogue has to know.  The natural choice (row 16*i + q) leaves a lane with four SCATTERED groups of 4 output
// columns per token, which is why the epilogue had to go through the LDS to build whole rows (park, barrier-free wait, row loop:
// 7.8 of the 13.5 us a tile round cost at K = 128, profiles/r02_gemm_epilogue_costs.txt).  With
//     plain / RoPE:  row = 32*(i>>1) + 8*a + 4*(i&1) + b          SwiGLU ([16 gate | 16 up] packs):  row = 32*(a>>1) + 16*(i&1) + 8*(a&1) + 4*(i>>1) + b
// a lane (fg = a, r = b) owns 8 CONSECUTIVE output columns per block pair (i>>1) -- columns 32*(i>>1) + 8*fg + 4*(i&1) + r -- resp.
// gate (i even) and up (i odd) of the 8 consecutive SwiGLU outputs 8*fg + 4*(i>>1) + r: the four lanes of a token write 64 contiguous
// bytes with one 16-byte store each, straight from the accumulators, and both halves of a RoPE head (columns n and n + 64) sit in the
// same lane.  No LDS, no barrier: the epilogue is ~130 conversions and 32 stores per wave.
// The rows a ds_read_b128 fetches are then no longer 16 consecutive ones, so the XOR swizzle of the W tile uses the row bits that
// vary across the 16 lanes of a fragment read instead of row & 7: s(row) = bit1 | bits 3,4 << 1 (plain), bit1 | bit3 << 1 | bit5 << 2
// (SwiGLU) -- for the reading lane both are simply fr >> 1, and every 16-lane service group of the read hits 64 distinct banks.
template <bool SWIGLU> ULL_DEV int w4_row_swizzle(int row) {
    return SWIGLU ? (((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 5) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 3) << 1));
}
// physical W row (within the wave's 128) read by fragment c = 4*h + i, minus the lane-dependent part
template <bool SWIGLU> ULL_DEV constexpr int w4_frag_row(int c) {
    return SWIGLU ? 64 * (c >> 2) + 16 * (c & 1) + 4 * ((c >> 1) & 1) : 32 * (c >> 1) + 4 * (c & 1);
}
template <bool SWIGLU> ULL_DEV int w4_lane_row(int fr) {
    return SWIGLU ? 32 * (fr >> 3) + 8 * ((fr >> 2) & 1) + (fr & 3) : 8 * (fr >> 2) + (fr & 3);
}
// first of the 4 consecutive physical rows (= slab columns of the stream-K tail) that acc[h][i][.][0..3] of lane group fg holds
template <bool SWIGLU> ULL_DEV int w4_acc_row(int h, int i, int fg) {
    return SWIGLU ? h * 64 + 32 * (fg >> 1) + 16 * (i & 1) + 8 * (fg & 1) + 4 * (i >> 1) : h * 64 + 32 * (i >> 1) + 8 * fg + 4 * (i & 1);
}

// Register-direct epilogue of the 4-wave kernel for 16-bit outputs with 16-byte-aligned rows and no activation: plain, + bias,
// + residual, SwiGLU, RoPE.  Same rounding points as the staged epilogue (bit-identical results: tests/test_kernels_gpu.py forces both
// kernel forms through every epilogue): rnd(acc + bias); SwiGLU rnd(rnd(silu(rnd(g))) * rnd(u)); residual rnd(res + rnd(.)); RoPE
// rnd(rnd(x cos) + rnd(-+x' sin)) on x = rnd(acc).
// The accumulators live in AGPRs (tied operands of the MFMA asm).  Left to the allocator, the epilogue's reads become 16-byte scratch
// stores of whole AGPR tuples and scratch loads into VGPRs ("folded spills": 168 of the 256 values went through memory); an opaque
// asm on the copy makes the AGPR -> VGPR move happen HERE, four v_accvgpr_read at the point of use.
ULL_DEV f32x4_t w4_acc_to_vgpr(const f32x4_t& a) {
    f32x4_t t = a;
    asm volatile("" : "+v"(t));
    return t;
}

// NH: 64-column halves a wave owns (2: the 4-wave kernel; 1: the 8-wave direct form, gemm256d_kernel)
// RPF: groups of residual rows requested ahead (8 = all up front; 2 = one group ahead, for callers short of registers)
template <bool SWIGLU, bool ROPE, int NH = 2, int RPF = 8>
ULL_DEV void w4_direct_epilogue(const GemmArgs& p, f32x4_t (&acc)[NH][4][8], int lane, int mrow0, int nw0) {
    static_assert(NH == 2 || (!SWIGLU && !ROPE), "SwiGLU and RoPE pair columns across the two halves");
    // One group of 16 token rows (j) at a time, fenced by sched_barriers: left to itself the scheduler hoists every accumulator read
    // and every load of the 8 groups to the top (~250 live registers, spilled around the K-loop).  Loads of group j + 1 (residual rows,
    // RoPE table rows) are issued before the arithmetic of group j.
    const int fr = lane & 15, fg = lane >> 4;
    if constexpr (SWIGLU) {
        const int n_out = p.N / 2, nb = nw0 / 2 + 8 * fg;
        elem_t* cb = (elem_t*)p.C + nb;
#pragma clang loop unroll(full)
        for (int j = 0; j < 8; ++j) {
            const int m = mrow0 + j * 16 + fr;
#pragma clang loop unroll(full)
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma clang loop unroll(full)
                for (int pp = 0; pp < 2; ++pp) {
                    const f32x4_t ag = w4_acc_to_vgpr(acc[h][2 * pp][j]), au = w4_acc_to_vgpr(acc[h][2 * pp + 1][j]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = rnd(ag[r]);                          // gate_proj output (16-bit tensor)
                        const float u = rnd(au[r]);                          // up_proj output
                        v[4 * pp + r] = rnd(act_silu(g)) * u;                // silu -> 16 bit, product -> 16 bit (by the pack)
                    }
                }
                if (m < p.M && nb + h * 32 + 8 <= n_out) *(uint4*)(cb + (long)m * p.ldc + h * 32) = pack8(v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const int flags = p.flags;
        const bool has_bias = flags & EPI_BIAS, has_res = flags & EPI_RESID;
        const bool rope_on = ROPE && nw0 < p.rope_cols;          // the wave's 128 columns are one head
        const int nb = nw0 + 8 * fg;                             // + h*64 + pp*32: the lane's 8 columns
        uint4 bvp[NH][2];                                         // the lane's 32 bias values, packed (unpacked at use: 16 registers held, not 32)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int n = nb + h * 64 + pp * 32;
                bvp[h][pp] = (has_bias && n + 8 <= p.N) ? *(const uint4*)(p.bias + n) : uint4{0, 0, 0, 0};
            }
        elem_t* cb = (elem_t*)p.C + nb;
        const elem_t* rb = p.R + nb;
        auto body = [&](auto with_bias, auto with_res, auto act_kind) {
            constexpr bool WB = decltype(with_bias)::value, WR = decltype(with_res)::value;
            constexpr int ACT = decltype(act_kind)::value;       // 0 none, 1 QuickGELU, 2 GELU(erf), 3 ReLU: on the rounded Linear output
            // residual rows / RoPE table rows of ALL eight groups are requested up front (128 registers: the fragments are dead, the
            // accumulators sit in AGPRs): one burst with every load in flight instead of eight round trips to HBM
            // (the RoPE tables keep a one-group-ahead double buffer: all eight groups up front spilled 424 B per lane in that kernel)
            constexpr int NB_ = ROPE ? 2 : RPF;
            uint4 rv[NB_][NH][2], tc[NB_][2], ts[NB_][2];         // [group][h][pp] residual rows, [group][pp] RoPE table rows
            auto fetch = [&](int j, int buf) {
                const long mc = min(mrow0 + j * 16 + fr, p.M - 1);
                if constexpr (WR) {
#pragma unroll
                    for (int h = 0; h < NH; ++h)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) rv[buf][h][pp] = *(const uint4*)(rb + mc * p.ldr + h * 64 + pp * 32);   // (columns past N: masked at the store)
                }
                if constexpr (ROPE) {
                    if (rope_on) {
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) {
                            tc[buf][pp] = *(const uint4*)(p.rope_cos + mc * 64 + pp * 32 + 8 * fg);
                            ts[buf][pp] = *(const uint4*)(p.rope_sin + mc * 64 + pp * 32 + 8 * fg);
                        }
                    }
                }
            };
            if constexpr (ROPE || NB_ < 8) fetch(0, 0);
            else if constexpr (WR) {
#pragma clang loop unroll(full)
                for (int j = 0; j < 8; ++j) fetch(j, j);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma clang loop unroll(full)
            for (int j = 0; j < 8; ++j) {
                const int m = mrow0 + j * 16 + fr;
                if constexpr (ROPE || NB_ < 8) { if (j + 1 < 8) fetch(j + 1, (j + 1) % NB_); }
                uint4 o[NH][2];
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        float v[8], bv[8];
                        if constexpr (WB) unpack8(bvp[h][pp], bv);
                        const f32x4_t a0 = w4_acc_to_vgpr(acc[h][2 * pp][j]), a1 = w4_acc_to_vgpr(acc[h][2 * pp + 1][j]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a = e < 4 ? a0[e & 3] : a1[e & 3];
                            v[e] = WB ? a + bv[e] : a;
                        }
                        o[h][pp] = pack8(v);                                 // the Linear's 16-bit output
                        if constexpr (ACT != 0) {
                            float a[8];
                            unpack8(o[h][pp], a);
                            if constexpr (ACT == 2) {
#pragma unroll
                                for (int e = 0; e < 8; e += 2) {
                                    const f32x2_t r = act_gelu_erf2(f32x2_t{a[e], a[e + 1]});
                                    a[e] = r.x; a[e + 1] = r.y;              // (pack8 rounds)
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) a[e] = ACT == 1 ? act_quick_gelu_e(a[e]) : fmaxf(a[e], 0.f);
                            }
                            o[h][pp] = pack8(a);
                        }
                    }
                if constexpr (ROPE) {
                    if (rope_on) {
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) {
                            float x0[8], x1[8], cs[8], sn[8];
                            unpack8(o[0][pp], x0); unpack8(o[1][pp], x1);
                            unpack8(tc[j % NB_][pp], cs); unpack8(ts[j % NB_][pp], sn);
                            float y0[8], y1[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                y0[e] = rnd(x0[e] * cs[e]) + rnd(-x1[e] * sn[e]);   // dims 0..63 of the head: rotate_half contributes -x[d + 64]
                                y1[e] = rnd(x1[e] * cs[e]) + rnd(x0[e] * sn[e]);
                            }
                            o[0][pp] = pack8(y0); o[1][pp] = pack8(y1);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        if constexpr (WR) {
                            float a[8], b[8];
                            unpack8(o[h][pp], a); unpack8(rv[j % NB_][h][pp], b);
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = rnd(b[e] + a[e]);
                            o[h][pp] = pack8(a);
                        }
                        // (A lane's 16 bytes: four lanes make 64 contiguous bytes per token row, half a 128-byte line; the other half goes
                        // with the next instruction.  Full-line stores -- 8 rows x 128 B per instruction after a lane-pair exchange -- drain
                        // a tile in 2.1 instead of 4.4 us per CU (tools/probes/store_pattern_probe.hip), and a timing-only build with that
                        // address pattern shortened the epilogue from 5.8 to 3.8 us per tile WITHOUT shortening a single launch: these
                        // launches run against the power cap, and time taken out of a low-power phase comes back as a lower clock in the
                        // K-loop (docs/experiments.md, round 4).  Not built.)
                        if (m < p.M && nb + h * 64 + pp * 32 + 8 <= p.N) *(uint4*)(cb + (long)m * p.ldc + h * 64 + pp * 32) = o[h][pp];
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
        if constexpr (!ROPE) {
            if (act) {                                           // (no Linear on the path has an activation AND a residual: host dispatch)
                if (act == 1) { if (has_bias) body(std::true_type{}, std::false_type{}, std::integral_constant<int, 1>{}); else body(std::false_type{}, std::false_type{}, std::integral_constant<int, 1>{}); }
                else if (act == 2) { if (has_bias) body(std::true_type{}, std::false_type{}, std::integral_constant<int, 2>{}); else body(std::false_type{}, std::false_type{}, std::integral_constant<int, 2>{}); }
                else { if (has_bias) body(std::true_type{}, std::false_type{}, std::integral_constant<int, 3>{}); else body(std::false_type{}, std::false_type{}, std::integral_constant<int, 3>{}); }
                return;
            }
        }
        if (has_bias) { if (has_res) body(std::true_type{}, std::true_type{}, I0{}); else body(std::true_type{}, std::false_type{}, I0{}); }
        else { if (has_res) body(std::false_type{}, std::true_type{}, I0{}); else body(std::false_type{}, std::false_type{}, I0{}); }
    }
}

template <bool SWIGLU, bool ROPE = false>
__global__ __launch_bounds__(256) void gemm256w4_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ULL_STAMP(0); ULL_STAMP(7);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    int bid = blockIdx.x;
    int slice = 0;
    const bool split = bid >= p.t_full;
    if (!split) {
        const int nwg = p.t_full;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective for any nwg
    } else {
        const int r = bid - p.t_full;
        bid = p.t_full + r / p.sk;
        slice = r % p.sk;
    }
    const int per_group = p.group_m * p.nbn;
    const int gid = bid / per_group;
    const int first_m = gid * p.group_m;
    const int gsz = min(p.nbm - first_m, p.group_m);
    const int bm = first_m + (bid % per_group) % gsz;
    const int bn = (bid % per_group) / gsz;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk_total = p.K / BK;
    const bool w_tiled = p.flags & EPI_W_TILED, x_tiled = p.flags & EPI_X_TILED;

    // DMA: a 1-KiB piece = 8 rows x 128 B; wave w stages pieces 8w..8w+7 of X and of W.
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    uint32_t xo[8], wo[8];                            // byte offsets from the tile's first row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (wave * 8 + i) * 8 + srow;
        const int wchunk = (lane & 7) ^ w4_row_swizzle<SWIGLU>(r);          // the W tile's swizzle follows the permuted fragment reads
        xo[i] = x_tiled ? (uint32_t)(r * BK + schunk * 8) * 2 : (uint32_t)(((long)(min(m0 + r, p.M - 1) - m0) * p.ldx + schunk * 8) * 2);
        wo[i] = w_tiled ? (uint32_t)(r * BK + wchunk * 8) * 2 : (uint32_t)(((long)(min(n0 + r, p.N - 1) - n0) * p.ldw + wchunk * 8) * 2);
    }
    const elem_t* xbase = x_tiled ? p.X + ((long)bm * nk_total) * BM * BK : p.X + (long)m0 * p.ldx;
    const elem_t* wbase = w_tiled ? p.W + ((long)bn * nk_total) * BN * BK : p.W + (long)n0 * p.ldw;
    const long xstep = x_tiled ? BM * BK : BK, wstep = w_tiled ? BN * BK : BK;    // elements between consecutive K-tiles
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const uint32_t piece_off = wave * 8 * 1024;
    // piece c of tile kt: c < 8 -> X piece, else W piece
    auto dma_piece = [&](const elem_t* xk, const elem_t* wk, uint32_t slot_addr, int c) {
#if defined(ULL_ABL_DMASAME)     // every piece re-reads the same KiB: the issue cost of the DMA without its memory traffic
        glds16s(p.X, (uint32_t)lane * 16, slot_addr + c * 1024);
        return;
#endif
        if (c < 8) glds16s(xk, xo[c], slot_addr + c * 1024);
        else glds16s(wk, wo[c - 8], slot_addr + OP_BYTES + (c - 8) * 1024);
    };
    auto stage_all = [&](int kt) {
        const elem_t* xk = xbase + (long)kt * xstep;
        const elem_t* wk = wbase + (long)kt * wstep;
        const uint32_t sa = lds_base + (kt & 1) * SLOT_BYTES + piece_off;
#pragma unroll
        for (int c = 0; c < 16; ++c) dma_piece(xk, wk, sa, c);
    };

    const int fr = lane & 15, fg = lane >> 4;
    int swz[2];
    swz[0] = ((0 + fg) ^ (lane & 7)) << 4;
    swz[1] = ((4 + fg) ^ (lane & 7)) << 4;
    int wswz[2];                                      // W tile: permuted rows, swizzle = fr >> 1 for the reading lane (see w4_row_swizzle)
    wswz[0] = ((0 + fg) ^ (fr >> 1)) << 4;
    wswz[1] = ((4 + fg) ^ (fr >> 1)) << 4;
    const int xoff = (wm * 128 + fr) * (BK * 2);
    const int woff = OP_BYTES + (wn * 128 + w4_lane_row<SWIGLU>(fr)) * (BK * 2);
    // c-th fragment read of a half step: the 8 W fragments first (all of them feed the first two chunks), then X in use order
    auto read_one = [&](int kt, int kk, Frags4& f, int c) {
        const char* base = smem + (kt & 1) * SLOT_BYTES;
        if (c < 8) f.w[c] = *(const uint4*)(base + wswz[kk] + woff + w4_frag_row<SWIGLU>(c) * (BK * 2));
        else f.x[c - 8] = *(const uint4*)(base + swz[kk] + xoff + (c - 8) * 16 * (BK * 2));
    };

    f32x4_t acc[2][4][8];                             // [half of the 128 n-columns][i][j]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // chunk c of a half step: MFMAs (j = c / 2, i = 4 (c % 2) .. + 3)
    auto chunk_mma = [&](const Frags4& f, int c) {
        const int j = c >> 1, h = c & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) mfma16_inplace(acc[h][i][j], f.w[h * 4 + i], f.x[j]);
    };

    int nk = p.K / BK;                                // >= 2 (host dispatch, also per K-slice)
    if (split) {
        const int kt0 = (int)((long)nk * slice / p.sk), kt1 = (int)((long)nk * (slice + 1) / p.sk);
        xbase += (long)kt0 * xstep; wbase += (long)kt0 * wstep;
        nk = kt1 - kt0;
    }
    ULL_STAMP(1);                                     // index arithmetic done
    stage_all(0);
    stage_all(1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); // tile 0 landed, tile 1 in flight
    __builtin_amdgcn_s_barrier();
    ULL_STAMP(2);                                     // first K-tile in LDS
    Frags4 fa, fb;                                    // fa: half 0 of the current tile, fb: half 1
#pragma unroll
    for (int c = 0; c < 16; ++c) read_one(0, 0, fa, c);
    // One loop over all K-steps and nothing between it and the epilogue that touches an accumulator: the register allocator puts any
    // accumulator copy it wants (loop exit, a second loop's entry) directly behind the last MFMA, which is an opaque asm to it and so
    // gets no wait states.  Hence: the last two steps (nothing left to prefetch) skip their DMA pieces through wave-uniform branches
    // instead of living in a loop of their own, the fragment reads of "tile nk" in the very last step fetch stale LDS that is never
    // used, and the wait for the matrix pipe sits inside the loop body, at the end of the last step.
    // A K-step is 128 slots of one MFMA each (0..63 on fa = k-half 0 of tile kt, 64..127 on fb = k-half 1), every slot fenced by a
    // sched_barrier.  A SIMD with one wave issues one instruction per 4 clocks and an MFMA holds the matrix pipe for 16, so at most
    // three other instructions fit behind an MFMA for free, and a group of them anywhere idles the pipe: everything else is dealt out
    // one or two instructions per slot.
    //   slots   0..15  read fb <- (kt, half 1), one fragment per slot
    //   slot   32      lgkmcnt(0) + barrier A: every wave is done with slot kt
    //   slots  32..95  one DMA piece of tile kt+2 per 4 slots into that slot (m0 write, the MFMA as its wait state, the load: one asm)
    //   slot   96      vmcnt(16): tile kt+1 (issued a step ago) has landed, tile kt+2 stays in flight; barrier B: visible to all
    //   slots  96..127 read fa <- (kt+1, half 0), one fragment per 2 slots, in first-use order
    // so the memory pipe is never drained (1.0-1.5 steps for a piece to land) and never sees a burst.  The last two steps have nothing
    // to prefetch: they re-fetch the last tile (L2 hits) into a 16-KiB dump behind the epilogue regions instead of branching.
    // schedule knobs (tools/build_ablations.sh overrides them with -D for sweeps)
#ifndef ULL_W4_BAR_A
#define ULL_W4_BAR_A 20
#endif
#ifndef ULL_W4_BAR_B
#define ULL_W4_BAR_B 96
#endif
#ifndef ULL_W4_FB_STRIDE
#define ULL_W4_FB_STRIDE 1
#endif
#ifndef ULL_W4_FA_FIRST
#define ULL_W4_FA_FIRST 96
#endif
#ifndef ULL_W4_FA_STRIDE
#define ULL_W4_FA_STRIDE 2
#endif
#ifndef ULL_W4_DMA_FIRST
#define ULL_W4_DMA_FIRST 20
#endif
#ifndef ULL_W4_DMA_STRIDE
#define ULL_W4_DMA_STRIDE 7
#endif
    constexpr int W4_BAR_A = ULL_W4_BAR_A, W4_BAR_B = ULL_W4_BAR_B, W4_FB_STRIDE = ULL_W4_FB_STRIDE, W4_FA_FIRST = ULL_W4_FA_FIRST,
                  W4_FA_STRIDE = ULL_W4_FA_STRIDE, W4_DMA_FIRST = ULL_W4_DMA_FIRST, W4_DMA_STRIDE = ULL_W4_DMA_STRIDE;
    constexpr int W4_ISSUED = (W4_BAR_B - W4_DMA_FIRST + W4_DMA_STRIDE - 1) / W4_DMA_STRIDE;   // pieces of tile kt+2 issued before barrier B
    constexpr int W4_INFLIGHT = W4_ISSUED < 0 ? 0 : W4_ISSUED > 16 ? 16 : W4_ISSUED;
    static_assert(W4_BAR_A >= 15 * W4_FB_STRIDE + 1 && W4_DMA_FIRST >= W4_BAR_A && W4_FA_FIRST >= W4_BAR_B, "schedule order");
    static_assert(W4_DMA_FIRST + 15 * W4_DMA_STRIDE < 128 && W4_FA_FIRST + 15 * W4_FA_STRIDE < 128, "schedule fits the step");
    constexpr int USE_ORDER[16] = {0, 8, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15};   // read_one index: w0, x0, w1..w7, x1..x7
    // slot s of a half: accumulator (j = s / 8, h = (s / 4) % 2, i = s % 4)
    auto slot_mma = [&](const Frags4& f, int s) {
        const int j = s >> 3, h = (s >> 2) & 1, i = s & 3;
        mfma16_inplace(acc[h][i][j], f.w[h * 4 + i], f.x[j]);
    };
    auto slot_mma_dma = [&](const Frags4& f, int s, const void* sbase, uint32_t voff, uint32_t lds_addr) {
        const int j = s >> 3, h = (s >> 2) & 1, i = s & 3;
        const uint4 &a = f.w[h * 4 + i], &b = f.x[j];
        const u32x4_t av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
#if defined(ULL_ABL_NODMA)
        asm volatile(ULL_MFMA16_ASM " %0, %1, %2, %0" : "+a"(acc[h][i][j]) : "v"(av), "v"(bv));
#elif defined(ULL_ABL_NOMMA)
        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4"
                     : "+a"(acc[h][i][j]) : "v"(av), "v"(bv), "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
#else
        asm volatile("s_mov_b32 m0, %5\n\t" ULL_MFMA16_ASM " %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %3, %4"
                     : "+a"(acc[h][i][j]) : "v"(av), "v"(bv), "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
#endif
    };
    uint32_t m0_keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(m0_keep));        // the loop writes m0 without saving it (the compiler emits nothing that reads it there)
    const uint32_t dump = lds_base + LDS_BYTES;               // 16 KiB, all four waves
#pragma clang loop unroll(disable)                      // also keeps the unroller from peeling the first step off into a copy of the loop
    for (int kt = 0; kt < nk; ++kt) {
        const bool with_dma = kt + 2 < nk;
        const int ktd = with_dma ? kt + 2 : nk - 1;
        const elem_t* xk = xbase + (long)ktd * xstep;
        const elem_t* wk = wbase + (long)ktd * wstep;
        const uint32_t sa = with_dma ? lds_base + (kt & 1) * SLOT_BYTES + piece_off : dump;
        const uint32_t sw = with_dma ? sa + OP_BYTES : dump;
#pragma clang loop unroll(full)
        for (int s = 0; s < 128; ++s) {
            if (s == W4_BAR_A) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !defined(ULL_ABL_NOBAR)
                __builtin_amdgcn_s_barrier();
#endif
            }
            if (s == W4_BAR_B) {
#if defined(ULL_W4_NODUMP)
                if (!with_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
#endif
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W4_INFLIGHT) : "memory");
#if !defined(ULL_ABL_NOBAR)
                __builtin_amdgcn_s_barrier();
#endif
            }
#if !defined(ULL_ABL_NOREAD)
            if (s < 16 * W4_FB_STRIDE && s % W4_FB_STRIDE == 0) read_one(kt, 1, fb, s / W4_FB_STRIDE);
            if (s >= W4_FA_FIRST && (s - W4_FA_FIRST) % W4_FA_STRIDE == 0 && (s - W4_FA_FIRST) / W4_FA_STRIDE < 16)
                read_one(kt + 1, 0, fa, USE_ORDER[(s - W4_FA_FIRST) / W4_FA_STRIDE]);
#endif
            const Frags4& f = s < 64 ? fa : fb;
            const int pc = (s - W4_DMA_FIRST) / W4_DMA_STRIDE;   // piece: 0..7 of X, 8..15 of W
            if (s >= W4_DMA_FIRST && (s - W4_DMA_FIRST) % W4_DMA_STRIDE == 0 && pc < 16) {
#if defined(ULL_W4_NODUMP)
                if (!with_dma) slot_mma(f, s & 63); else
#endif
                if (pc < 8) slot_mma_dma(f, s & 63, xk, xo[pc], sa + pc * 1024);
                else slot_mma_dma(f, s & 63, wk, wo[pc - 8], sw + (pc - 8) * 1024);
            } else slot_mma(f, s & 63);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt == nk - 1) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    }
    ULL_STAMP(3);                                     // K-loop done
    asm volatile("s_mov_b32 m0, %0" :: "s"(m0_keep) : "memory");   // (the dump's pieces may still be landing: they touch nothing
                                                                    // the epilogue uses, and s_endpgm waits for them)

#if defined(ULL_ABL_NOEPI)
    if (p.M > 0) return;
#endif
    // ---- epilogue: acc[h][i][j][r] = D[n = n0 + wn*128 + w4_acc_row(h, i, fg) + r][m = m0 + wm*128 + j*16 + fr] (physical W row n) ------
    if (split) {
        float* slab = p.ws + ((long)(bid - p.t_full) * p.sk + slice) * (BM * BN);
#pragma clang loop unroll(full)
        for (int j = 0; j < 8; ++j)
#pragma clang loop unroll(full)
            for (int h = 0; h < 2; ++h)
#pragma clang loop unroll(full)
                for (int i = 0; i < 4; ++i)
                    *(f32x4_t*)(slab + (wm * 128 + j * 16 + fr) * BN + wn * 128 + w4_acc_row<SWIGLU>(h, i, fg)) = acc[h][i][j];
        return;
    }
    const int mrow0 = m0 + wm * 128, nw0 = n0 + wn * 128;
    {
        // 16-bit output, 16-byte-aligned rows, whole groups of 8 columns, no activation: straight from the accumulators (wave-uniform test)
        const int fl = p.flags;
        const bool direct = ROPE || (!(fl & (EPI_OUT_F32 | EPI_BIAS_ROUNDED)) && (!(fl & EPI_ACT_MASK) || !(fl & EPI_RESID)) && (p.ldc & 7) == 0 &&
                                     ((SWIGLU ? p.N / 2 : p.N) & 7) == 0 && (!(fl & EPI_RESID) || (p.ldr & 7) == 0) && !(fl & ULL_W4_FORCE_STAGED));
        if (direct) {
            w4_direct_epilogue<SWIGLU, ROPE>(p, acc, lane, mrow0, nw0);
            ULL_STAMP(4);                             // stores issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ULL_STAMP(5);                             // stores retired (this wave's)
            return;
        }
    }
    __builtin_amdgcn_s_barrier();                          // every wave has consumed the last K-tile: LDS is free
    char* reg0 = smem + wave * (2 * 128 * 144);      // 36 KiB per wave: two bf16 regions, or one fp32 region of 128 rows x 272 B
    // The flag-dependent half of the epilogue (finish) never touches an accumulator, so it is compiled once and run per 64-column half.
    // (RoPE launches always qualify for the direct form: the entry point insists on aligned rows and whole heads.)
    if constexpr (ROPE) {
        return;
    } else if constexpr (SWIGLU) {
        staged_epilogue<true, 8, false, 1, true, 4, true, 1>(p, acc[0], reg0, lane, mrow0, nw0, nullptr, 0);
        staged_epilogue<true, 8, false, 1, true, 4, true, 1>(p, acc[1], reg0, lane, mrow0, nw0, nullptr, 1);
        staged_epilogue<true, 8, false, 2, true, 4, true, 1>(p, acc[0], reg0, lane, mrow0, nw0);
    } else {
#pragma clang loop unroll(disable)
        for (int h = 0; h < 2; ++h) {
            if (h == 0) staged_epilogue<false, 8, false, 1, true, 4, false, 1>(p, acc[0], reg0, lane, mrow0, nw0);
            else staged_epilogue<false, 8, false, 1, true, 4, false, 1>(p, acc[1], reg0, lane, mrow0, nw0 + 64);
            staged_epilogue<false, 8, false, 2, true, 4, false, 1>(p, acc[0], reg0, lane, mrow0, nw0 + 64 * h);
        }
    }
}

// (Round 3 built the persistent form a third time, now on top of the register-direct epilogue -- one workgroup per CU chaining its
// items into one K-step stream, the last two steps of an item prefetching K-tiles 0 / 1 of the next, the epilogue's stores left in
// flight under the next item's first step; no LDS juggling, no extra barrier, zero spills, bit-identical results.  Measured against
// the one-tile-per-block launch of the SAME kernel body on the same box (tools/gemm_shapes.py, profiles/r03_gemm_notes.md): qkv+RoPE
// 1519 vs 1461 us, gate/up+SwiGLU 2568 vs 2538, o+residual 524 vs 518, down 1294 vs 1271, K = 1280 shapes equal: 1-4 % SLOWER again.
// 256 accumulator clears, the re-read of the first fragments and the item decode cost what the hidden prologue saves, and the wait
// for the epilogue's stores moves to barrier B of the next item's first step instead of disappearing.  Removed; the block hand-over
// of the hardware stays.
// A fourth build fixed the one real flaw of the third -- stores and loads share the vmcnt counter and complete out of order with respect
// to each other, so the next item's first step had been waiting for the epilogue's 32 stores: the item now drained its prefetched K-tiles
// BEFORE issuing the stores and the next first step passed its barrier without a vmcnt wait -- and measured the same: SAM qkv 267.6 vs
// 270.7 us, CLIP qkv 111 vs 115, gate/up 2558 vs 2519, down 1283 vs 1251.  The K-loop-only ablation explains it: what a round of tiles
// pays behind its last MFMA is not instructions but a 33-MB store burst from 256 CUs that finish together (~5 us at HBM write rate);
// in flight under the next item's loop it delays that loop's operand loads instead.  Only CUs that are OUT OF PHASE would hide it
// (a stream-K split of every CU's first tile), and the fp32 slabs + finalize that costs are about what it would save.)

// ---- fused ViT patchify, one round of strips (round 3) -------------------------------------------------------------------------
// The 128x128 form above runs B = 32 at 336^2 (M = 18432 patches, N = 1024, K' = 704) as 1152 tiles = 2.25 rounds of two blocks per CU and
// moves 415 MB from L2 to LDS for 61 MB of algorithmic bytes: 47 us, and every plain GEMM kernel of this library and the vendor's lands at
// 40-50 us on the same M x N x K' (profiles/r02_patchify_ceiling.txt): the op is bound by tile quantisation and L2 -> LDS bytes, not HBM.
// This form fits the TILE to the problem instead: (32 * WMB) x 256 with 8 waves of (16 * WMB) x 64 -- WMB = 9: 288 x 256, exactly
// 64 x 4 = 256 tiles for the C4 batch (one per CU, one round, no padded rows); WMB = 4: 128 x 256 for 224^2 -- which also halves the
// L2 -> LDS traffic (196 MB).  Same A-tile DMA from the pixels, permuted W rows + register-direct stores as in the 4-wave GEMM.
// Measured (tools/patchify_bench.py, same box): 38.5 us against 42.4 at 336^2 = 19.7 % of the 8 TB/s the north-star target counts
// against.  Per CU the tile needs 765 KB through the L1 (~9 us at the ~85 GB/s a CU sustains) and 12.7 us of MFMA time, so ~15-19 us
// would be the floor of ANY single-round GEMM form of this op (= 40-50 % of HBM peak: the 60 % target, 12.6 us, is below the MFMA time
// alone).  Round 4 took the kernel apart (profiles/r04_patchify_variants.txt): (1) the DMA pieces' address arithmetic, ~420
// instructions per K-step and wave, moved into a table (segtab): 38.2 -> 33.4-33.7 us, shipped; (2) a ring of four 34-KiB stages of
// 32-wide K-steps, three steps in flight instead of one: 36.8 us (tools/probes/patchify_bk32_ring_r04.hip.inc) -- the loop does not wait
// for memory latency; (3) fragment reads software-pipelined in groups of three pixel-row blocks under the previous group's MFMAs:
// 33.4 us, no change; (4) ablations of the shipped form: without the MFMAs 27.0 us, without any DMA after the first stage 27.8 us.
// A misaligned 16-byte-lane gather (28-byte lane stride) is not the limit either: a CU's address / L1 path takes it at 98 GB/s
// (tools/probes/dma_align_probe.hip; 140 aligned), three times what the loop asks for.  What the ablations leave is the single round itself: launch + per-lane patch
// addressing + first stage (~5 us), then all 256 blocks finish together and write 37.7 MB at once (~6 us at the 6.9 TB/s fill rate)
// with nothing left to overlap it, around a K-loop of ~21 us against 12.7 us of MFMA time.  0.02 % of a C4 step.
template <int WMB>
__global__ __launch_bounds__(512) void patchify_strip_kernel(GemmArgs p, PatchArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = 32 * WMB, TN = 256;
    constexpr int A_BYTES = TM * 128, W_BYTES = TN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int NPIECE = (TM + TN) / 8;                  // 1-KiB DMA pieces per stage (8 rows each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int fr = lane & 15, fg = lane >> 4;
    // The nbn N-tiles of a strip read the SAME pixels: they belong on one XCD (one L2), not on nbn neighbouring block ids -- which the
    // hardware deals round-robin over the 8 XCDs, so that every strip was fetched from the fabric nbn times (PMC, round 4: 88 MB of
    // fabric reads for 21.7 MB of pixels, L2 hit rate 51 %).  Block id -> (XCD, slot); an XCD owns nbm / 8 consecutive strips.
    int bn = blockIdx.x % p.nbn, bm = blockIdx.x / p.nbn;
    if (q.xcd_raster) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;     // slot 0 .. nbm * nbn / 8
        bm = xcd * (p.nbm >> 3) + slot / p.nbn;
        bn = slot % p.nbn;
    }
    const int m0 = bm * TM, n0 = bn * TN;
    const int srow = lane >> 3;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    constexpr int PPW = (NPIECE + 7) / 8;                  // pieces per wave
    // Element offset of (c, ky) segment idx = c * ps + ky inside an image, -1 past the last real segment: a table of K / 16 <= 64 entries in
    // LDS, filled once.  (Computed per DMA piece and K-step -- an integer division by ps, a 64-bit multiply, three selects -- the nine
    // pieces of a wave cost ~420 instructions per K-step against its 72 MFMAs, and the barrier of every step lined the two waves of a
    // SIMD up so that both did their address arithmetic first and their MFMAs second: 38.2 us, 33.7 with the table; round 4.)
    int* segtab = (int*)(smem + 2 * STAGE);
    if (tid < 64) {
        const int c = tid / q.ps, ky = tid - c * q.ps;
        segtab[tid] = tid < q.nseg ? (c * q.H + ky) * q.W : -1;
    }
    // piece pc (0 .. NPIECE): pc < TM / 8 -> A rows 8 pc .. + 8, else W rows
    const elem_t* src[PPW];                                // A: pixel (b, c = 0, y = py * ps, x = px * ps) + half * 8;  W: row start + chunk
    const int schunk = (lane & 7) ^ srow;                  // X tile swizzle: row & 7 (the same for every A piece of the lane)
    const int aseg = schunk >> 1;                          // (c, ky) segment of this lane's chunk within the K-tile (0..3)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + 8 * i;
        src[i] = nullptr;
        if (pc < TM / 8) {
            const int r = pc * 8 + srow;
            const int m = min(m0 + r, p.M - 1);
            const int px = m % q.gw, py = (m / q.gw) % q.gh, b = m / (q.gw * q.gh);
            src[i] = q.img + (((long)b * q.C * q.H + (long)py * q.ps) * q.W + px * q.ps) + (schunk & 1) * 8;
        } else if (pc < NPIECE) {
            const int r = (pc - TM / 8) * 8 + srow;
            const int wchunk = (lane & 7) ^ w4_row_swizzle<false>(r);
            src[i] = p.W + (long)min(n0 + r, p.N - 1) * p.ldw + wchunk * 8;
        }
    }
    // the 16-byte chunk of the last patch column runs 2 pixels into the next image row (zero weight columns): for the very last row of
    // the image buffer that is past its end, and only the strip that holds the last patches can get there (two code paths: the guard
    // costs a 64-bit compare and an exec-masked branch per piece)
    const bool tail_strip = __builtin_amdgcn_readfirstlane(m0 + TM >= p.M ? 1 : 0) != 0;
    __syncthreads();                                       // segtab
    auto stage_t = [&](int buf, int kt, auto guarded) {
        const uint32_t sb = lds_base + buf * STAGE;
        const int off = segtab[kt * 4 + aseg];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + 8 * i;
            if (pc >= NPIECE) continue;
            if (pc < TM / 8) {
                const elem_t* sp = off >= 0 ? src[i] + off : q.zeros;
                if (decltype(guarded)::value && off >= 0 && sp + 8 > q.img_end) {          // through registers
                    const uint32_t* s32 = (const uint32_t*)sp;
                    *(uint4*)(smem + buf * STAGE + pc * 1024 + lane * 16) = make_uint4(s32[0], s32[1], s32[2], 0u);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    glds16(sp, sb + pc * 1024);
                }
            } else {
                glds16(src[i] + (long)kt * BK, sb + A_BYTES + (pc - TM / 8) * 1024);
            }
        }
    };
    auto stage = [&](int buf, int kt) {
        if (tail_strip) stage_t(buf, kt, std::true_type{});
        else stage_t(buf, kt, std::false_type{});
    };
    int swz[2], wswz[2];
    swz[0] = ((0 + fg) ^ (lane & 7)) << 4;
    swz[1] = ((4 + fg) ^ (lane & 7)) << 4;
    wswz[0] = ((0 + fg) ^ (fr >> 1)) << 4;
    wswz[1] = ((4 + fg) ^ (fr >> 1)) << 4;
    const int xoff = (wm * 16 * WMB + fr) * 128;
    const int woff = A_BYTES + (wn * 64 + w4_lane_row<false>(fr)) * 128;

    f32x4_t acc[4][WMB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* base = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 wf[4], xf[WMB];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *(const uint4*)(base + wswz[kk] + woff + w4_frag_row<false>(i) * 128);
#pragma unroll
            for (int j = 0; j < WMB; ++j) xf[j] = *(const uint4*)(base + swz[kk] + xoff + j * 16 * 128);
#pragma unroll
            for (int j = 0; j < WMB; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
        }
    }
    // epilogue: acc[i][j][r] = D[n = n0 + wn*64 + 32*(i>>1) + 8*fg + 4*(i&1) + r][m = m0 + wm*16*WMB + j*16 + fr]: 16-byte stores
    const int nb = n0 + wn * 64 + 8 * fg;
    float bv[2][8];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        if ((p.flags & EPI_BIAS) && nb + pp * 32 + 8 <= p.N) unpack8(*(const uint4*)(p.bias + nb + pp * 32), bv[pp]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[pp][e] = -0.0f;
        }
    }
    elem_t* cb = (elem_t*)p.C + nb;
#pragma unroll
    for (int j = 0; j < WMB; ++j) {
        const int m = m0 + wm * 16 * WMB + j * 16 + fr;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[2 * pp + (e >> 2)][j][e & 3] + bv[pp][e];
            if (m < p.M && nb + pp * 32 + 8 <= p.N) *(uint4*)(cb + (long)m * p.ldc + pp * 32) = pack8(v);
        }
    }
}

// ---- 8 waves, permuted W rows, register-direct epilogue ----------------------------------------------------------------------------
// The 8-wave K-loop with the 4-wave kernel's W-row permutation and direct epilogue (one 64-column half per wave).  For launches whose
// epilogue is VALU-heavy -- QuickGELU / erf-GELU -- two waves per SIMD issue the activation at the full VALU rate (one wave per SIMD gets
// every other issue slot), which outweighs this form's 4 % slower K-loop at K = 1024 / 1280.
__global__ __launch_bounds__(512) void gemm256d_kernel(GemmArgs p) {
    constexpr bool SWIGLU = false, ROPE = false;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    // Blocks [0, t_full) own one whole tile each (t_full is a multiple of the CU count, so those rounds are full); the
    // remaining tiles -- which would otherwise occupy a few CUs for one more whole round -- are split `sk` ways along K.
    int bid = blockIdx.x;
    int slice = 0;
    const bool split = bid >= p.t_full;
    if (!split) {
        const int nwg = p.t_full;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective for any nwg
    } else {
        const int r = bid - p.t_full;
        bid = p.t_full + r / p.sk;
        slice = r % p.sk;
    }
    const int per_group = p.group_m * p.nbn;
    const int gid = bid / per_group;
    const int first_m = gid * p.group_m;
    const int gsz = min(p.nbm - first_m, p.group_m);
    const int bm = first_m + (bid % per_group) % gsz;
    const int bn = (bid % per_group) / gsz;
    const int m0 = bm * BM, n0 = bn * BN;
    const int nk_total = p.K / BK;
    const bool w_tiled = p.flags & EPI_W_TILED, x_tiled = p.flags & EPI_X_TILED;

    // DMA: a 1-KiB piece = 8 rows x 128 B; wave w stages pieces 4w..4w+3 of X and of W.
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    const elem_t* xsrc[4];
    const elem_t* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;
        xsrc[i] = x_tiled ? p.X + (((long)bm * nk_total) * BM + r) * BK + schunk * 8
                          : p.X + (long)min(m0 + r, p.M - 1) * p.ldx + schunk * 8;
        const int wchunk = (lane & 7) ^ w4_row_swizzle<false>(r);           // permuted W rows: the W tile's own swizzle
        wsrc[i] = w_tiled ? p.W + (((long)bn * nk_total) * BN + r) * BK + wchunk * 8
                          : p.W + (long)min(n0 + r, p.N - 1) * p.ldw + wchunk * 8;
    }
    const long xstep = x_tiled ? BM * BK : BK, wstep = w_tiled ? BN * BK : BK;    // elements between consecutive K-tiles
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const uint32_t piece_off = wave * 4 * 1024;
    auto stage = [&](int kt) {
        const uint32_t bx = lds_base + (kt & 1) * SLOT_BYTES + piece_off;
        const long kox = (long)kt * xstep, kow = (long)kt * wstep;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(xsrc[i] + kox, bx + i * 1024);
            glds16(wsrc[i] + kow, bx + OP_BYTES + i * 1024);
        }
    };

    const int fr = lane & 15, fg = lane >> 4;
    int swz[2];
    swz[0] = ((0 + fg) ^ (lane & 7)) << 4;
    swz[1] = ((4 + fg) ^ (lane & 7)) << 4;
    int wswz[2];
    wswz[0] = ((0 + fg) ^ (fr >> 1)) << 4;
    wswz[1] = ((4 + fg) ^ (fr >> 1)) << 4;
    const int xoff = (wm * 128 + fr) * (BK * 2);
    const int woff = OP_BYTES + (wn * 64 + w4_lane_row<false>(fr)) * (BK * 2);
    auto read_frags = [&](int kt, int kk, Frags& f) {
        const char* base = smem + (kt & 1) * SLOT_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) f.w[i] = *(const uint4*)(base + wswz[kk] + woff + w4_frag_row<false>(i) * (BK * 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) f.x[j] = *(const uint4*)(base + swz[kk] + xoff + j * 16 * (BK * 2));
    };

    f32x4_t acc[1][4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int nk = p.K / BK;                                // >= 2 (host dispatch, also per K-slice)
    if (split) {
        const int kt0 = (int)((long)nk * slice / p.sk), kt1 = (int)((long)nk * (slice + 1) / p.sk);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xsrc[i] += (long)kt0 * xstep; wsrc[i] += (long)kt0 * wstep; }
        nk = kt1 - kt0;
    }
    // Same step schedule as the 4-wave kernel (see there), 64 MFMA slots per wave and step: the 12 fragment reads of the tile's second
    // half first, barrier A, the 8 DMA pieces of tile kt+2 dealt out one per 6 slots (a burst of all 64 pieces of a CU right behind
    // the barrier measured 3-7 % slower end to end: the memory pipe wants an even stream), barrier B with the newest pieces still in
    // flight, then the first-half fragments of tile kt+1.  The last two steps re-fetch the last tile into a 16-KiB dump.
#ifndef ULL_W8_BAR_A
#define ULL_W8_BAR_A 14
#endif
#ifndef ULL_W8_DMA_STRIDE
#define ULL_W8_DMA_STRIDE 6
#endif
#ifndef ULL_W8_BAR_B
#define ULL_W8_BAR_B 40
#endif
#ifndef ULL_W8_FA_STRIDE
#define ULL_W8_FA_STRIDE 2
#endif
    constexpr int W8_BAR_A = ULL_W8_BAR_A, W8_DMA_STRIDE = ULL_W8_DMA_STRIDE, W8_BAR_B = ULL_W8_BAR_B, W8_FA_STRIDE = ULL_W8_FA_STRIDE;
    constexpr int W8_ISSUED = (W8_BAR_B - W8_BAR_A + W8_DMA_STRIDE - 1) / W8_DMA_STRIDE;
    constexpr int W8_INFLIGHT = W8_ISSUED > 8 ? 8 : W8_ISSUED;
    static_assert(W8_BAR_A >= 12 && W8_BAR_B >= W8_BAR_A && W8_BAR_A + 7 * W8_DMA_STRIDE < 64 && W8_BAR_B + 11 * W8_FA_STRIDE < 64, "schedule");
    constexpr int USE_ORDER[12] = {0, 4, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11};    // w0, x0, w1..w3, x1..x7
    auto read1 = [&](int kt, int kk, Frags& f, int r) {
        const char* base = smem + (kt & 1) * SLOT_BYTES;
        if (r < 4) f.w[r] = *(const uint4*)(base + wswz[kk] + woff + w4_frag_row<false>(r) * (BK * 2));
        else f.x[r - 4] = *(const uint4*)(base + swz[kk] + xoff + (r - 4) * 16 * (BK * 2));
    };
    auto mma1 = [&](const Frags& f, int s) {
        const int j = s >> 2, i = s & 3;
#if !defined(ULL_ABL_NOMMA)
        acc[0][i][j] = mfma16(f.w[i], f.x[j], acc[0][i][j]);
#endif
    };
    stage(0);
    stage(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile 0 landed, tile 1 in flight
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;                                     // fa: half 0 of the current tile, fb: half 1
    read_frags(0, 0, fa);
    const uint32_t dump = lds_base + LDS_BYTES;
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nk; ++kt) {
        const bool with_dma = kt + 2 < nk;
        const int ktd = with_dma ? kt + 2 : nk - 1;
        const long kox = (long)ktd * xstep, kow = (long)ktd * wstep;
        const uint32_t bx = with_dma ? lds_base + (kt & 1) * SLOT_BYTES + piece_off : dump;
        const uint32_t bw = with_dma ? bx + OP_BYTES : dump + 4096;
#pragma clang loop unroll(full)
        for (int s = 0; s < 64; ++s) {
            if (s == W8_BAR_A) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (s == W8_BAR_B) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W8_INFLIGHT) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (s < 12) read1(kt, 1, fb, s);
            if (s >= W8_BAR_B && (s - W8_BAR_B) % W8_FA_STRIDE == 0 && (s - W8_BAR_B) / W8_FA_STRIDE < 12)
                read1(kt + 1, 0, fa, USE_ORDER[(s - W8_BAR_B) / W8_FA_STRIDE]);
#if !defined(ULL_ABL_NODMA)
            if (s >= W8_BAR_A && (s - W8_BAR_A) % W8_DMA_STRIDE == 0 && (s - W8_BAR_A) / W8_DMA_STRIDE < 8) {
                const int pc = (s - W8_BAR_A) / W8_DMA_STRIDE;
                if (pc < 4) glds16(xsrc[pc] + kox, bx + pc * 1024);
                else glds16(wsrc[pc - 4] + kow, bw + (pc - 4) * 1024);
            }
#endif
            mma1(s < 32 ? fa : fb, s & 31);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // (the dump's pieces may still be landing: they touch nothing the epilogue uses, and s_endpgm waits for them)

#if defined(ULL_ABL_NOEPI)
    if (p.M > 0) return;
#endif
    // ---- epilogue: acc[i][j][r] = D[n = n0 + wn*64 + i*16 + 4*fg + r][m = m0 + wm*128 + j*16 + fr] ----------
    if (split) {
        float* slab = p.ws + ((long)(bid - p.t_full) * p.sk + slice) * (BM * BN);
#pragma clang loop unroll(full)
        for (int j = 0; j < 8; ++j)
#pragma clang loop unroll(full)
            for (int i = 0; i < 4; ++i)
                *(f32x4_t*)(slab + (wm * 128 + j * 16 + fr) * BN + wn * 64 + w4_acc_row<false>(0, i, fg)) = acc[0][i][j];
        return;
    }
    {
        const int fl = p.flags;
        const bool direct = !(fl & (EPI_OUT_F32 | EPI_BIAS_ROUNDED)) && (!(fl & EPI_ACT_MASK) || !(fl & EPI_RESID)) && (p.ldc & 7) == 0 && (p.N & 7) == 0 &&
                            (!(fl & EPI_RESID) || (p.ldr & 7) == 0) && !(fl & ULL_W4_FORCE_STAGED);
        if (direct) {
            w4_direct_epilogue<false, false, 1>(p, acc, lane, m0 + wm * 128, n0 + wn * 64);
            return;
        }
    }
    __builtin_amdgcn_s_barrier();                          // every wave has consumed the last K-tile: LDS is free
    staged_epilogue<false, 8, false, 0, false, 2, false, 1>(p, acc[0], smem + wave * (128 * 144), lane, m0 + wm * 128, n0 + wn * 64);
}



// (A v_mfma_f32_32x32x16_bf16 variant of this kernel measured 6-11 % slower in this structure and was removed;
// numbers in profiles/r01_gemm_notes.md.)

// Sum the K-slices of the stream-K tail tiles and apply the same epilogue as the main kernel.  One thread per 4 output
// columns (8 accumulator columns for SwiGLU).
template <int BM, int BN, bool LINEAR>
__global__ __launch_bounds__(256) void splitk_finalize_any_kernel(GemmArgs p) {
    const int flags = p.flags;
    const bool swiglu = flags & EPI_SWIGLU;
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
    const bool out_f32 = flags & EPI_OUT_F32;
    const int n_out_total = swiglu ? p.N / 2 : p.N;
    const int tile_r = blockIdx.y;
    int m0, n0;
    if constexpr (LINEAR) {                                      // the 128x128 kernel's split-K: slabs indexed by (M-tile, N-tile)
        m0 = (tile_r / p.nbn) * BM;
        n0 = (tile_r % p.nbn) * BN;
    } else {                                                     // stream-K tail of the 256x256 kernels: the tiles behind the full rounds
        const int pid = p.t_full + tile_r;
        const int per_group = p.group_m * p.nbn;
        const int gid = pid / per_group;
        const int first_m = gid * p.group_m;
        const int gsz = min(p.nbm - first_m, p.group_m);
        m0 = (first_m + (pid % per_group) % gsz) * BM;
        n0 = ((pid % per_group) / gsz) * BN;
    }
    const float* slab0 = p.ws + (long)tile_r * p.sk * (BM * BN);
    const int quads_per_row = swiglu ? BN / 8 : BN / 4;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < BM * quads_per_row; q += gridDim.x * 256) {
        const int ml = q / quads_per_row, qc = q % quads_per_row;
        const int m = m0 + ml;
        if (m >= p.M) continue;
        float v[4];
        int n;
        if (swiglu) {
            // accumulator columns: 32-wide groups of [16 gate | 16 up]; this thread: 4 gate + the matching 4 up columns
            const int grp = qc / 4, off = (qc % 4) * 4;
            float g[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < p.sk; ++s) {
                const float* row = slab0 + (long)s * (BM * BN) + ml * BN + grp * 32 + off;
                const f32x4_t a = *(const f32x4_t*)row, b = *(const f32x4_t*)(row + 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) { g[r] += a[r]; u[r] += b[r]; }
            }
            n = n0 / 2 + grp * 16 + off;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rnd(rnd(act_silu(rnd(g[r]))) * rnd(u[r]));
        } else {
            float a4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < p.sk; ++s) {
                const f32x4_t a = *(const f32x4_t*)(slab0 + (long)s * (BM * BN) + ml * BN + qc * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) a4[r] += a[r];
            }
            n = n0 + qc * 4;
            if (p.rope_cos != nullptr && n < p.rope_cols) {
                // fused RoPE (see staged_epilogue): the partner half of the head sits 64 columns away in the same tile
                const bool first_half = (n & 64) == 0;
                const int qp = first_half ? qc + 16 : qc - 16;
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int s = 0; s < p.sk; ++s) {
                    const f32x4_t b = *(const f32x4_t*)(slab0 + (long)s * (BM * BN) + ml * BN + qp * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[r] += b[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float c = e2f(p.rope_cos[(long)m * 64 + ((n + r) & 63)]), sn = e2f(p.rope_sin[(long)m * 64 + ((n + r) & 63)]);
                    const float x = rnd(a4[r]), xp = rnd(b4[r]);
                    a4[r] = rnd(x * c) + rnd((first_half ? -xp : xp) * sn);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = a4[r];
                if ((flags & EPI_BIAS) && (flags & EPI_BIAS_ROUNDED)) t = rnd(t);
                if ((flags & EPI_BIAS) && n + r < p.N) t += e2f(p.bias[n + r]);
                if (!out_f32 || act || (flags & EPI_RESID)) t = rnd(t);
                if (act == 1) t = act_quick_gelu_e(t);
                else if (act == 2) t = rnd(act_gelu_erf(t));
                else if (act == 3) t = fmaxf(t, 0.f);
                v[r] = t;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= n_out_total) continue;
            float t = v[r];
            if (flags & EPI_RESID) t = rnd(e2f(p.R[(long)m * p.ldr + n + r]) + t);
            if (out_f32) ((float*)p.C)[(long)m * p.ldc + n + r] = t;
            else ((elem_t*)p.C)[(long)m * p.ldc + n + r] = f2e(t);
        }
    }
}

}  // namespace big

}  // namespace

// Per-device launch state (function attributes are per device; the CU count sizes the stream-K tail).  No allocation, no stream
// state: the only scratch the GEMM needs -- the fp32 slabs of the stream-K tail -- is the caller's (`ws`), so two streams
// never share anything.
static int gemm_device_state(int* n_cu_out) {
    constexpr int MAX_DEV = 64;
    static int n_cu[MAX_DEV];            // 0 = not yet initialised
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return ULL_ERR_LAUNCH;
    if (!n_cu[dev]) {
        hipDeviceProp_t prop;
        const int n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        (void)hipFuncSetAttribute((const void*)big::gemm256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES_W4);
        (void)hipFuncSetAttribute((const void*)big::gemm256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES_W4);
        (void)hipFuncSetAttribute((const void*)big::gemm256_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES_W4);
        (void)hipFuncSetAttribute((const void*)big::gemm256d_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES_W4);
        (void)hipFuncSetAttribute((const void*)big::gemm256w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES_W4);
        (void)hipFuncSetAttribute((const void*)big::gemm256w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES_W4);
        (void)hipFuncSetAttribute((const void*)big::gemm256w4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big::LDS_BYTES_W4);
        (void)hipFuncSetAttribute((const void*)gemm128_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)gemm128_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)gemm128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)patchify_gemm_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)big::patchify_strip_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (288 + 256) * 128 + 256);
        (void)hipFuncSetAttribute((const void*)big::patchify_strip_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 256) * 128 + 256);
        n_cu[dev] = n;                   // last: a racing first call on another thread repeats the (idempotent) attribute calls
    }
    *n_cu_out = n_cu[dev];
    return ULL_OK;
}

#ifndef ULL_ELEM_F16
extern "C" int64_t ull_gemm_streamk_ws_bytes(void) { return (int64_t)256 * big::BM * big::BN * sizeof(float); }
#endif

static int gemm_dispatch(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* R,
                         int64_t ldr, int64_t M, int64_t N, int64_t K, int flags, void* ws, int64_t ws_bytes, void* stream,
                         const elem_t* rope_cos, const elem_t* rope_sin, int rope_cols) {
    if (!X || !W || !C || M <= 0 || N <= 0 || K <= 0 || ws_bytes < 0) return ULL_ERR_ARG;
    if (K % BK != 0 || (ldx & 7) || (ldw & 7)) return ULL_ERR_SHAPE;          // 16-byte DMA pieces
    if ((flags & EPI_BIAS) && !bias) return ULL_ERR_ARG;
    if ((flags & EPI_RESID) && !R) return ULL_ERR_ARG;
    if ((flags & EPI_SWIGLU) && ((N & 31) || (flags & (EPI_BIAS | EPI_ACT_MASK)))) return ULL_ERR_SHAPE;
    if (M > (1 << 30) || N > (1 << 30)) return ULL_ERR_SHAPE;
    int n_cu = 0;
    if (const int rc = gemm_device_state(&n_cu)) return rc;
    // per-call tuning overrides (tools/ only; 0 = the shipped policy)
    const int tune_group_m = (flags >> 16) & 15;
    const bool force_small = flags & (1 << 20);
    const bool force_waves8 = flags & (1 << 21);
    const bool force_waves4 = (flags & (1 << 22)) && ldx < (1 << 21) && ldw < (1 << 21);
    flags &= 0xffff;
    if ((flags & EPI_BIAS_ROUNDED) && (flags & EPI_SWIGLU)) return ULL_ERR_SHAPE;
    GemmArgs a;
    a.X = (const elem_t*)X; a.W = (const elem_t*)W; a.C = C;
    a.bias = (const elem_t*)bias; a.R = (const elem_t*)R;
    a.ldx = ldx; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = flags;
    a.rope_cos = rope_cos; a.rope_sin = rope_sin; a.rope_cols = rope_cols;
#ifdef ULL_GEMM_STAMPS
    a.stamps = ull_stamp_host_ptr;
#endif
    const bool rope = rope_cos != nullptr;
    // short K and fewer than two rounds of 256x256 tiles (ViT patchify: K = 640, 128 / 288 tiles): the 128x128 kernel's 4x finer
    // tiles fill the chip better (measured 61 vs 70 us at B=32, 336^2)
    const bool short_and_few = K <= 768 && ((M + 255) / 256) * ((N + 255) / 256) < 512 && !(flags & (EPI_W_TILED | EPI_X_TILED));
    if (!force_small && !short_and_few && M >= 1024 && N >= 512 && K >= 128) {   // nk >= 2
        a.nbm = (int)((M + big::BM - 1) / big::BM); a.nbn = (int)((N + big::BN - 1) / big::BN);
        // stream-K tail: whole rounds of one tile per CU, the remainder split along K (needs >= 2 K-steps per slice) into fp32
        // slabs in the CALLER's workspace (ws == NULL: no split; the caller owns the policy -- see ops.py -- and one workspace per
        // stream keeps concurrent GEMMs on different streams independent)
        constexpr long SLAB = (long)big::BM * big::BN * sizeof(float);
        const int max_slabs = ws ? (int)(ws_bytes / SLAB < 4096 ? ws_bytes / SLAB : 4096) : 0;
        const int T = a.nbm * a.nbn;
        const int rem = T % n_cu;
        int sk = 1;
        if (max_slabs && T > n_cu && rem > 0 && rem <= n_cu / 2 && rem <= max_slabs) {
            sk = n_cu / rem;
            if (sk > 16) sk = 16;
            const int nk = (int)(K / big::BK);
            while (sk > 1 && nk / sk < 2) --sk;
            if (rem * sk > max_slabs) sk = max_slabs / rem;
        }
        a.sk = sk > 1 ? sk : 1;
        a.t_full = sk > 1 ? T - rem : T;
        a.ws = (float*)ws;
        // measured sweep at M=20576 (profiles/r01_gemm_notes.md): 4 M-tiles x 8 N-tiles per XCD wave beats 8 x 4 by ~5 %
        a.group_m = tune_group_m ? tune_group_m : 4;
        const int grid = a.sk > 1 ? a.t_full + rem * a.sk : T;
        // The 4-wave form has the faster K-loop and the slower tile turn-around (one wave per SIMD: nothing overlaps the epilogue's and
        // the prologue's latencies, and the last two steps of every tile re-fetch a K-tile), so it takes the long-K launches -- the
        // LLaMA layer, SAM's lin2 -- and the 8-wave form the K = 1024 / 1280 shapes of the ViTs (measured crossover, tools/gemm_bench.py:
        // K = 3072 +4 %, K = 2048 -2 %, K = 1280 -9 %; with few tiles -- M = 1024-3032 tokens -- the 4-wave form is equal or up to 7 %
        // ahead at K >= 4096: tools/gemm_smallm.py).  It addresses its DMA pieces with
        // 32-bit offsets from the tile's first row.
        // (An output whose rows are not 16-byte aligned -- lm_head, V = 32011 -- is stored element by element: latency again, 8 waves.)
        // Which form: the 4-wave kernel whenever its epilogue is register-direct (16-bit output, 16-byte-aligned rows, whole groups of 8
        // columns, no activation: same test as in the kernel) -- it has the faster K-loop and, since round 3, the cheaper turn-around too
        // (tools/gemm_shapes.py: +4.5-7 % on the LLaMA layer, +3-13 % on the K = 1024 / 1280 ViT shapes over the 8-wave form).  With an
        // activation, an fp32 output or unaligned rows (lm_head, V = 32011: element-wise stores) the epilogue goes through the LDS and one
        // wave per SIMD hides none of its latencies: those keep the 8-wave form below K = 3072.  The 4-wave kernel addresses its DMA
        // pieces with 32-bit offsets from the tile's first row.
        const bool c_rows_aligned = (flags & EPI_OUT_F32) ? (ldc & 3) == 0 : (ldc & 7) == 0;
        const int n_out_cols = (flags & EPI_SWIGLU) ? (int)(N / 2) : (int)N;
        const bool direct = rope || (!(flags & (EPI_OUT_F32 | EPI_BIAS_ROUNDED | ULL_W4_FORCE_STAGED)) && (!(flags & EPI_ACT_MASK) || !(flags & EPI_RESID)) &&
                                     (ldc & 7) == 0 && (n_out_cols & 7) == 0 && (!(flags & EPI_RESID) || (ldr & 7) == 0));
        const bool fits32 = ldx < (1 << 21) && ldw < (1 << 21);
        const bool waves4 = force_waves4 || (!force_waves8 && fits32 && (direct || (K >= 3072 && c_rows_aligned)));
        const bool waves8_direct = !force_waves4 && !force_waves8 && direct && !rope && !(flags & EPI_SWIGLU) && (flags & EPI_ACT_MASK) &&
                                   ((flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT) != 3 && K < 3072;
        if (waves8_direct) {
            hipLaunchKernelGGL(big::gemm256d_kernel, dim3(grid), dim3(512), big::LDS_BYTES_W4, (hipStream_t)stream, a);
        } else if (waves4) {
            if (flags & EPI_SWIGLU)
                hipLaunchKernelGGL(big::gemm256w4_kernel<true>, dim3(grid), dim3(256), big::LDS_BYTES_W4, (hipStream_t)stream, a);
            else if (rope)
                hipLaunchKernelGGL((big::gemm256w4_kernel<false, true>), dim3(grid), dim3(256), big::LDS_BYTES_W4, (hipStream_t)stream, a);
            else
                hipLaunchKernelGGL(big::gemm256w4_kernel<false>, dim3(grid), dim3(256), big::LDS_BYTES_W4, (hipStream_t)stream, a);
        } else if (flags & EPI_SWIGLU)
            hipLaunchKernelGGL(big::gemm256_kernel<true>, dim3(grid), dim3(512), big::LDS_BYTES_W4, (hipStream_t)stream, a);
        else if (rope)
            hipLaunchKernelGGL((big::gemm256_kernel<false, true>), dim3(grid), dim3(512), big::LDS_BYTES_W4, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(big::gemm256_kernel<false>, dim3(grid), dim3(512), big::LDS_BYTES_W4, (hipStream_t)stream, a);
        if (a.sk > 1) hipLaunchKernelGGL((big::splitk_finalize_any_kernel<big::BM, big::BN, false>), dim3(32, rem), dim3(256), 0, (hipStream_t)stream, a);
        return ull_check_launch();
    }
    if (flags & (EPI_W_TILED | EPI_X_TILED)) return ULL_ERR_SHAPE;      // tile-major operands: 256x256 kernel only
    a.nbm = (int)((M + BM - 1) / BM); a.nbn = (int)((N + BN - 1) / BN);
    a.t_full = 0; a.sk = 1; a.ws = nullptr; a.group_m = GROUP_M;
    // split-K: the chip holds 2 blocks per CU; a launch of at most half that many tiles with a long K (o_proj / down_proj of a single-image
    // prefill: 96 tiles x 64 / 172 K-steps, each step one memory round trip in this kernel) is cut into sk slices of >= 8 K-steps that fill
    // the slots, fp32 slabs in the caller's workspace (none: no split), summed in slice order by the finalize launch.
    const int T = a.nbm * a.nbn, nk = (int)(K / BK), slots = 2 * n_cu;
    if (ws && !force_small && T <= slots / 2 && nk >= 16) {
        constexpr long SLAB = (long)BM * BN * sizeof(float);
        int sk = slots / T;
        if (sk > nk / 8) sk = nk / 8;
        if (sk > 8) sk = 8;
        if ((long)T * sk * SLAB > ws_bytes) sk = (int)(ws_bytes / (T * SLAB));
        if (sk > 1) { a.sk = sk; a.ws = (float*)ws; }
    }
    const int grid = T * a.sk;
    if (flags & EPI_SWIGLU)
        hipLaunchKernelGGL(gemm128_kernel<true>, dim3(grid), dim3(256), GEMM_LDS, (hipStream_t)stream, a);
    else if (rope)
        hipLaunchKernelGGL((gemm128_kernel<false, true>), dim3(grid), dim3(256), GEMM_LDS, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(gemm128_kernel<false>, dim3(grid), dim3(256), GEMM_LDS, (hipStream_t)stream, a);
    if (a.sk > 1) hipLaunchKernelGGL((big::splitk_finalize_any_kernel<BM, BN, true>), dim3(16, T), dim3(256), 0, (hipStream_t)stream, a);
    return ull_check_launch();
}

#ifdef ULL_GEMM_STAMPS
extern "C" int ULL_FN(ull_debug_gemm_stamps_)(void* buf) { ull_stamp_host_ptr = (unsigned long long*)buf; return ULL_OK; }
#endif
extern "C" int ULL_FN(ull_gemm_)(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc,
                             const void* bias, const void* R, int64_t ldr,
                             int64_t M, int64_t N, int64_t K, int flags, void* ws, int64_t ws_bytes, void* stream) {
    return gemm_dispatch(X, ldx, W, ldw, C, ldc, bias, R, ldr, M, N, K, flags, ws, ws_bytes, stream, nullptr, nullptr, 0);
}

// The fused q|k|v projection of LlamaAttention (hf modeling_llama.py:214-277): C[M, N] = X W^T (no bias) with
// apply_rotary_pos_emb applied in the epilogue to the output columns [0, rope_cols) = the q and k heads (head_dim must be 128; the v
// columns pass through).  rope_cos / rope_sin: [M, 64] tables of the element type from ull_rope_table_*.  flags: only
// ULL_EPI_W_TILED / ULL_EPI_X_TILED / tuning bits.
extern "C" int ULL_FN(ull_gemm_qkv_rope_)(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                                      int64_t K, const void* rope_cos, const void* rope_sin, int64_t rope_cols, int64_t head_dim, int flags,
                                      void* ws, int64_t ws_bytes, void* stream) {
    if (!rope_cos || !rope_sin) return ULL_ERR_ARG;
    if (head_dim != 128 || rope_cols <= 0 || rope_cols > N || (rope_cols & 127) || (N & 127) || (ldc & 7)) return ULL_ERR_SHAPE;
    if (flags & (EPI_BIAS | EPI_ACT_MASK | EPI_RESID | EPI_SWIGLU | EPI_OUT_F32 | EPI_BIAS_ROUNDED)) return ULL_ERR_ARG;
    return gemm_dispatch(X, ldx, W, ldw, C, ldc, nullptr, nullptr, 0, M, N, K, flags, ws, ws_bytes, stream, (const elem_t*)rope_cos,
                         (const elem_t*)rope_sin, (int)rope_cols);
}

// out[(b, py, px), n] = bias[n] + sum_{c,ky,kx} img[b, c, py*ps+ky, px*ps+kx] * w[n, c, ky, kx]   (bf16 in / out, fp32 accumulate)
// Wp: packed weight [N, Kp], Kp = ceil(C*ps*16 / 64) * 64, Wp[n][(c*ps+ky)*16 + kx] = w[n][c][ky][kx] for kx < ps, zero elsewhere.
// img: contiguous [n_img, C, H, W] bf16 (nothing behind it is read); zeros: >= 16 zero bytes.  ps even, <= 16, H % ps == W % ps == 0,
// W >= 16.
extern "C" int ULL_FN(ull_patchify_)(const void* img, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, const void* Wp, int64_t Kp,
                                 const void* bias, void* out, int64_t ldc, int64_t N, const void* zeros, void* stream) {
    if (!img || !Wp || !out || !zeros || n_img <= 0 || C <= 0 || H <= 0 || W < 16 || N <= 0) return ULL_ERR_ARG;
    if (ps <= 0 || ps > 16 || (ps & 1) || H % ps || W % ps || Kp % BK || Kp < C * ps * 16 || (ldc & 7)) return ULL_ERR_SHAPE;
    GemmArgs a;
    PatchArgs q;
    q.img = (const elem_t*)img; q.zeros = (const elem_t*)zeros;
    q.img_end = (const elem_t*)img + n_img * C * H * W;
    q.C = (int)C; q.H = (int)H; q.W = (int)W; q.ps = (int)ps; q.gw = (int)(W / ps); q.gh = (int)(H / ps); q.nseg = (int)(C * ps);
    q.xcd_raster = 0;
    const int64_t M = n_img * q.gw * q.gh;
    if (M > (1 << 30)) return ULL_ERR_SHAPE;
    a.X = nullptr; a.W = (const elem_t*)Wp; a.C = out; a.bias = (const elem_t*)bias; a.R = nullptr;
    a.ldx = 0; a.ldw = Kp; a.ldc = ldc; a.ldr = 0;
    a.M = (int)M; a.N = (int)N; a.K = (int)Kp; a.flags = bias ? EPI_BIAS : 0;
    a.nbm = (int)((M + BM - 1) / BM); a.nbn = (int)((N + BN - 1) / BN);
    a.t_full = 0; a.sk = 1; a.ws = nullptr; a.group_m = GROUP_M;
    a.rope_cos = a.rope_sin = nullptr; a.rope_cols = 0;
#ifdef ULL_GEMM_STAMPS
    a.stamps = nullptr;
#endif
    int n_cu = 0;
    if (const int rc = gemm_device_state(&n_cu)) return rc;
    // one round of problem-sized strips when the batch allows it (C4: 32 x 576 patches = 64 strips of 288 x 4 column tiles = 256 blocks)
    if ((N & 255) == 0 && (ldc & 7) == 0 && Kp <= 1024 && C * H * W < (1LL << 31)) {
        const int nbn = (int)(N / 256);
        for (const int wmb : {9}) {          // (WMB = 4, 128 x 256 strips for 224^2 at B = 32, measured 20.9 us against 19.6 for the 128x128 form: not used)
            const int tm = 32 * wmb;
            if (M % tm) continue;
            const long tiles = (M / tm) * nbn;
            if (tiles > n_cu || tiles * 2 <= n_cu) continue;
            a.nbm = (int)(M / tm); a.nbn = nbn;
            q.xcd_raster = a.nbm % 8 == 0;
            const int lds = 2 * (tm + 256) * 128 + 256;          // two stages + the segment table
            if (wmb == 9) hipLaunchKernelGGL(big::patchify_strip_kernel<9>, dim3((unsigned)tiles), dim3(512), lds, (hipStream_t)stream, a, q);
            else hipLaunchKernelGGL(big::patchify_strip_kernel<4>, dim3((unsigned)tiles), dim3(512), lds, (hipStream_t)stream, a, q);
            return ull_check_launch();
        }
    }
    hipLaunchKernelGGL(patchify_gemm_kernel<0>, dim3(a.nbm * a.nbn), dim3(256), GEMM_LDS, (hipStream_t)stream, a, q);
    return ull_check_launch();
}
