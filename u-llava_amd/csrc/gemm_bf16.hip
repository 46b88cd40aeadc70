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
//   * Fused epilogues reproduce the rounding points of the reference's bf16 graph (rbf()).
//   * 1-D grid with XCD-aware remap (block b runs on XCD b%8; each XCD gets a contiguous chunk of the
//     tile space, walked in GROUP_M-row groups so co-resident blocks share X / W panels in that L2).
#include "ull_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;           // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;         // X tile + W tile
constexpr int GEMM_LDS = 2 * BUF_BYTES;           // double buffered: 64 KiB -> 2 blocks / CU
constexpr int GROUP_M = 8;

// epilogue flag bits (mirrored in include/ullava_hip.h)
constexpr int EPI_BIAS = 1, EPI_ACT_SHIFT = 1, EPI_ACT_MASK = 3 << 1;  // act: 0 none 1 quick_gelu 2 gelu(erf) 3 relu
constexpr int EPI_RESID = 8, EPI_SWIGLU = 16, EPI_OUT_F32 = 32;

struct GemmArgs {
    const bf16_t* X; const bf16_t* W; void* C;
    const bf16_t* bias; const bf16_t* R;
    long ldx, ldw, ldc, ldr;
    int M, N, K, flags;
    int nbm, nbn;
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

template <bool SWIGLU>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;

    // ---- block -> tile: XCD-contiguous chunks, grouped raster inside ------------------------
    const int nwg = p.nbm * p.nbn;
    int bid = blockIdx.x;
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
    const bf16_t* xsrc[4];
    const bf16_t* wsrc[4];
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

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
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

    // ---- epilogue -------------------------------------------------------------------------
    // acc[i][j][r] = D[n = n0 + wn*64 + i*16 + 4*(l>>4) + r][m = m0 + wm*64 + j*16 + (l&15)]
    const int flags = p.flags;
    const int act = (flags & EPI_ACT_MASK) >> EPI_ACT_SHIFT;
    const bool out_f32 = flags & EPI_OUT_F32;
    const int n_out_total = SWIGLU ? p.N / 2 : p.N;
    const bool vec_ok = ((p.ldc & 3) == 0) && ((n_out_total & 3) == 0) && (!(flags & EPI_RESID) || (p.ldr & 3) == 0);

    auto emit = [&](int m, int n, float (&v)[4]) {
        if (n >= n_out_total) return;
        if (flags & EPI_RESID) {
            const bf16_t* rp = p.R + (long)m * p.ldr + n;
            if (vec_ok) {
                const uint2 rv = *(const uint2*)rp;
                v[0] = rbf(bf2f((bf16_t)(rv.x & 0xffff)) + v[0]);
                v[1] = rbf(bf2f((bf16_t)(rv.x >> 16)) + v[1]);
                v[2] = rbf(bf2f((bf16_t)(rv.y & 0xffff)) + v[2]);
                v[3] = rbf(bf2f((bf16_t)(rv.y >> 16)) + v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < n_out_total) v[r] = rbf(bf2f(rp[r]) + v[r]);
            }
        }
        if (out_f32) {
            float* cp = (float*)p.C + (long)m * p.ldc + n;
            if (vec_ok) *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < n_out_total) cp[r] = v[r];
            }
        } else {
            bf16_t* cp = (bf16_t*)p.C + (long)m * p.ldc + n;
            if (vec_ok) {
                uint2 o;
                o.x = pack2bf(v[0], v[1]);
                o.y = pack2bf(v[2], v[3]);
                *(uint2*)cp = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < n_out_total) cp[r] = f2bf(v[r]);
            }
        }
    };

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + frow;
        if (m < p.M) {
            if constexpr (SWIGLU) {
                // W rows are interleaved in 16-row groups: [gate 16g..16g+15][up 16g..16g+15]
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const int n = (n0 + wn * 64) / 2 + ip * 16 + fgrp * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float g = rbf(acc[2 * ip][j][r]);          // gate_proj output (bf16 tensor)
                        const float u = rbf(acc[2 * ip + 1][j][r]);      // up_proj output (bf16 tensor)
                        v[r] = rbf(rbf(act_silu(g)) * u);                // silu -> bf16, product -> bf16
                    }
                    emit(m, n, v);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + wn * 64 + i * 16 + fgrp * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = acc[i][j][r];
                        if ((flags & EPI_BIAS) && n + r < p.N) t += bf2f(p.bias[n + r]);
                        if (!out_f32 || act || (flags & EPI_RESID)) t = rbf(t);   // the Linear's bf16 output
                        if (act == 1) t = act_quick_gelu_bf16(t);
                        else if (act == 2) t = rbf(act_gelu_erf(t));
                        else if (act == 3) t = fmaxf(t, 0.f);
                        v[r] = t;
                    }
                    emit(m, n, v);
                }
            }
        }
    }
}

}  // namespace

extern "C" int ull_gemm_bf16(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc,
                             const void* bias, const void* R, int64_t ldr,
                             int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    if (!X || !W || !C || M <= 0 || N <= 0 || K <= 0) return ULL_ERR_ARG;
    if (K % BK != 0 || (ldx & 7) || (ldw & 7)) return ULL_ERR_SHAPE;          // 16-byte DMA pieces
    if ((flags & EPI_BIAS) && !bias) return ULL_ERR_ARG;
    if ((flags & EPI_RESID) && !R) return ULL_ERR_ARG;
    if ((flags & EPI_SWIGLU) && ((N & 31) || (flags & (EPI_BIAS | EPI_ACT_MASK)))) return ULL_ERR_SHAPE;
    if (M > (1 << 30) || N > (1 << 30)) return ULL_ERR_SHAPE;
    GemmArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.C = C;
    a.bias = (const bf16_t*)bias; a.R = (const bf16_t*)R;
    a.ldx = ldx; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.M = (int)M; a.N = (int)N; a.K = (int)K; a.flags = flags;
    a.nbm = (int)((M + BM - 1) / BM); a.nbn = (int)((N + BN - 1) / BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_set = true;
    }
    if (flags & EPI_SWIGLU)
        hipLaunchKernelGGL(gemm_bf16_nt_kernel<true>, dim3(a.nbm * a.nbn), dim3(256), GEMM_LDS, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(gemm_bf16_nt_kernel<false>, dim3(a.nbm * a.nbn), dim3(256), GEMM_LDS, (hipStream_t)stream, a);
    return ull_check_launch();
}
