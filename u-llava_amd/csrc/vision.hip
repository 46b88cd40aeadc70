// Gather / scatter / pooling kernels around the GEMMs (all HBM-bound streaming, 16-byte accesses where
// the layout allows):
//   * patch im2col            (CLIPVisionEmbeddings.patch_embedding / SAM PatchEmbed: conv k = stride = patch)
//   * multimodal span finder + token-embedding lookup with visual-token splice
//                             (reference models/ullava_core.py:182-277 embed_images_videos)
//   * video spatio-temporal pooling (models/ullava_core.py:160-180 encode_video)
//   * row gather, bf16 add
#include "ull_common.h"

namespace {

// out[(img, py, px)][k], k = (c*ps + ky)*ps + kx  (the flattening of conv weight [D, C, ps, ps]); columns
// [C*ps*ps, Kp) are zero so the GEMM can use K = Kp (multiple of 64).
// Fast path (even patch size, W % 8 == 0): one thread per 8-pixel chunk of an image row -> one coalesced 16-byte load and four
// 4-byte stores (a dword never straddles a patch because ps and the chunk start are even).
__global__ __launch_bounds__(256) void im2col_vec_kernel(const elem_t* __restrict__ img, elem_t* __restrict__ out, int C, int Hh, int Ww, int ps,
                                                         int gw, int Kp, long total_chunks, int K, long pad_dwords) {
    const int cpr = Ww >> 3;
    const int padw = (Kp - K) >> 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_chunks + pad_dwords; i += (long)gridDim.x * 256) {
        if (i >= total_chunks) {                             // zero tail [K, Kp) of every row (same launch as the copy)
            const long z = i - total_chunks;
            *(uint32_t*)(out + (z / padw) * Kp + K + (z % padw) * 2) = 0u;
            continue;
        }
        const int x0 = (int)(i % cpr) * 8;
        long t = i / cpr;
        const int y = (int)(t % Hh);
        t /= Hh;
        const int c = (int)(t % C);
        const long b = t / C;
        const uint4 v = *(const uint4*)(img + ((b * C + c) * Hh + y) * (long)Ww + x0);
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        const int py = y / ps, ky = y - py * ps;
        const long rowbase = (b * (Hh / ps) + py) * (long)gw;
        const int kbase = (c * ps + ky) * ps;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + 2 * j;
            const int px = x / ps, kx = x - px * ps;
            *(uint32_t*)(out + (rowbase + px) * Kp + kbase + kx) = d[j];
        }
    }
}

__global__ __launch_bounds__(256) void im2col_kernel(const elem_t* __restrict__ img, elem_t* __restrict__ out, int C, int Hh, int Ww, int ps,
                                                     int gh, int gw, int Kp, long total) {
    const int K = C * ps * ps;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        const long row = i / Kp;
        elem_t v = 0;
        if (k < K) {
            const int kx = k % ps, ky = (k / ps) % ps, c = k / (ps * ps);
            const int px = (int)(row % gw), py = (int)((row / gw) % gh);
            const long b = row / ((long)gw * gh);
            v = img[((b * C + c) * Hh + (py * ps + ky)) * (long)Ww + px * ps + kx];
        }
        out[i] = v;
    }
}

// Per sample: count start/end tokens, first start position, running index into the feature batches.
// spans[b] = {kind (0 text, 1 image, 2 video), first start pos, feature index, error flag}
__global__ void mm_spans_kernel(const int64_t* __restrict__ ids, int B, int S, int img_start, int img_end, int vid_start, int vid_end,
                                long vocab, int32_t* __restrict__ spans) {
    // single block of 1024 threads; wave w scans samples w, w + 16, ... (64 ids per load, coalesced: one thread per sample took 0.29 ms
    // at B = 32, S = 643), then an in-block prefix count assigns feature indices
    __shared__ int kind_s[1024], pos_s[1024], err_s[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int b = wave; b < B; b += nwave) {
        int nis = 0, nie = 0, nvs = 0, nve = 0, pi = 0x7fffffff, pv = 0x7fffffff, err = 0;
        for (int s = lane; s < S; s += 64) {
            const int64_t t = ids[(long)b * S + s];
            if (t == img_start) { pi = min(pi, s); ++nis; }
            if (t == img_end) ++nie;
            if (t == vid_start) { pv = min(pv, s); ++nvs; }
            if (t == vid_end) ++nve;
            if (vocab > 0 && (t < 0 || t >= vocab)) err |= 2;   // nn.Embedding raises IndexError (ullava_core.py:191)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nis += __shfl_xor(nis, o, 64); nie += __shfl_xor(nie, o, 64); nvs += __shfl_xor(nvs, o, 64); nve += __shfl_xor(nve, o, 64);
            pi = min(pi, __shfl_xor(pi, o, 64)); pv = min(pv, __shfl_xor(pv, o, 64)); err |= __shfl_xor(err, o, 64);
        }
        if (lane == 0) {
            if (nis != nie || nvs != nve) err |= 1;    // reference asserts (ullava_core.py:209-211)
            int kind = 0, pos = -1;
            if (nis > 0) { kind = 1; pos = pi; }
            else if (nvs > 0) { kind = 2; pos = pv; }
            kind_s[b] = kind; pos_s[b] = pos; err_s[b] = err;
        }
    }
    __syncthreads();
    const int b = threadIdx.x;
    if (b < B) {
        const int kind = kind_s[b];
        int idx = 0;
        for (int j = 0; j < b; ++j) idx += (kind_s[j] == kind);
        spans[b * 4 + 0] = kind;
        spans[b * 4 + 1] = pos_s[b];
        spans[b * 4 + 2] = idx;
        spans[b * 4 + 3] = err_s[b];
    }
}

// out[b, s, :] = image/video feature row if s lies in the placeholder span of sample b, else table[ids[b, s]]
__global__ __launch_bounds__(256) void embed_splice_kernel(const int64_t* __restrict__ ids, const elem_t* __restrict__ table,
                                                           const elem_t* __restrict__ img_feat, int n_img_tok, int img_pitch, int img_off,
                                                           const elem_t* __restrict__ vid_feat, int n_vid_tok,
                                                           const int32_t* __restrict__ spans, elem_t* __restrict__ out, int S, int D,
                                                           long rows, long vocab) {
    const int cpr = D >> 3;                      // 16-byte chunks per row
    const int rows_per_block = 256 / min(cpr, 256);
    const int lanes_per_row = min(cpr, 256);
    const long row = (long)blockIdx.x * rows_per_block + threadIdx.x / lanes_per_row;
    if (row >= rows || (int)(threadIdx.x / lanes_per_row) >= rows_per_block) return;
    const int b = (int)(row / S), s = (int)(row % S);
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);     // never read outside the table (ull_mm_spans reports such ids)
    const elem_t* src = table + id * D;
    if (spans != nullptr) {
        const int kind = spans[b * 4], pos = spans[b * 4 + 1], idx = spans[b * 4 + 2];
        if (kind == 1 && img_feat != nullptr && s > pos && s <= pos + n_img_tok) src = img_feat + ((long)idx * img_pitch + img_off + (s - pos - 1)) * D;
        if (kind == 2 && vid_feat != nullptr && s > pos && s <= pos + n_vid_tok) src = vid_feat + ((long)idx * n_vid_tok + (s - pos - 1)) * D;
    }
    elem_t* dst = out + row * D;
    for (int c = threadIdx.x % lanes_per_row; c < cpr; c += lanes_per_row) *(uint4*)(dst + c * 8) = *(const uint4*)(src + c * 8);
}

// f [b, t, pitch, d] (tokens off..off+n of every frame are the patches) -> out [b, t + n, d]: rows [0, t) = mean over patches (temporal), rows [t, t+n) = mean over
// frames (spatial); fp32 accumulate, one bf16 rounding (torch.mean on a bf16 tensor).
__global__ __launch_bounds__(256) void video_pool_kernel(const elem_t* __restrict__ f, elem_t* __restrict__ out, int T, int N, int D,
                                                         int pitch, int off) {
    const int b = blockIdx.y;
    const int r = blockIdx.x;                    // output row in [0, T + N)
    const elem_t* fb = f + ((long)b * T * pitch + off) * D;
    elem_t* o = out + ((long)b * (T + N) + r) * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        if (r < T) {
            for (int n = 0; n < N; ++n) acc += e2f(fb[((long)r * pitch + n) * D + d]);
            o[d] = f2e(acc / (float)N);
        } else {
            const int n = r - T;
            for (int t = 0; t < T; ++t) acc += e2f(fb[((long)t * pitch + n) * D + d]);
            o[d] = f2e(acc / (float)T);
        }
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const elem_t* __restrict__ src, long lds_, const int64_t* __restrict__ idx,
                                                          elem_t* __restrict__ dst, long ldd, int D) {
    const long r = blockIdx.x;
    const elem_t* s = src + idx[r] * lds_;
    elem_t* d = dst + r * ldd;
    // long rows (the mask decoder gathers whole 2-MB image embeddings) are split over gridDim.y blocks
    for (int c = blockIdx.y * 256 + threadIdx.x; c < (D >> 3); c += 256 * gridDim.y) *(uint4*)(d + c * 8) = *(const uint4*)(s + c * 8);
}

// out = bf16(a + b[row % b_rows])   (row-broadcast add: residual adds, + positional tables)
__global__ __launch_bounds__(256) void add_rows_kernel(const elem_t* __restrict__ a, const elem_t* __restrict__ b, elem_t* __restrict__ out,
                                                       long rows, int D, long b_rows) {
    const int cpr = D >> 3;
    const long total = rows * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / cpr;
        const int c = (int)(i % cpr);
        float x[8], y[8];
        unpack8(*(const uint4*)(a + r * D + c * 8), x);
        unpack8(*(const uint4*)(b + (r % b_rows) * D + c * 8), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        *(uint4*)(out + r * D + c * 8) = pack8(x);
    }
}

}  // namespace

extern "C" int ULL_FN(ull_im2col_)(const void* img, void* out, int64_t n_img, int64_t C, int64_t H, int64_t W, int64_t ps, int64_t Kp,
                               void* stream) {
    if (!img || !out || n_img <= 0 || ps <= 0) return ULL_ERR_ARG;
    if (H % ps || W % ps || Kp < C * ps * ps) return ULL_ERR_SHAPE;
    const int gh = (int)(H / ps), gw = (int)(W / ps);
    const long total = n_img * gh * gw * Kp;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if ((ps & 1) == 0 && (W & 7) == 0 && (Kp & 1) == 0) {
        const long chunks = n_img * C * H * (W >> 3);
        const long rows = n_img * gh * gw;
        const int K = (int)(C * ps * ps);
        const long pad_dwords = rows * ((Kp - K) >> 1);                      // K and Kp are even: the zero tail is whole dwords
        const long work = chunks + pad_dwords;
        hipLaunchKernelGGL(im2col_vec_kernel, dim3((unsigned)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384)), dim3(256), 0,
                           (hipStream_t)stream, (const elem_t*)img, (elem_t*)out, (int)C, (int)H, (int)W, (int)ps, gw, (int)Kp, chunks, K,
                           pad_dwords);
        return ull_check_launch();
    }
    hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const elem_t*)img, (elem_t*)out, (int)C, (int)H, (int)W,
                       (int)ps, gh, gw, (int)Kp, total);
    return ull_check_launch();
}


namespace {
// ---- one greedy decoding step's bookkeeping (HF GenerationMixin greedy search: argmax, pad fill of finished rows, EOS tracking, append) ----
// reference: the `generate` the models inherit (models/ullava.py:350-361, inference_ullava_core.py:73-80).  One block per batch row:
// next = argmax(logits[b]) (first index on ties, as torch.argmax); finished rows get `pad` instead; the token is written to
// seq[b][pos]; a row that just emitted one of the EOS ids becomes finished; alive[0] += rows still unfinished (the host reads it every
// few steps instead of synchronising on every token).  Was: argmax, where, cat, isin, bitwise-not, and, any -- seven launches and a
// device -> host read per token.
__global__ __launch_bounds__(1024) void greedy_step_kernel(const elem_t* __restrict__ logits, long row_stride, int V, int32_t* __restrict__ unfinished,
                                                           const int64_t* __restrict__ eos, int n_eos, long pad, int has_pad,
                                                           int64_t* __restrict__ seq, long seq_ld, int pos, int32_t* __restrict__ alive) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int b = blockIdx.x;
    const elem_t* row = logits + (long)b * row_stride;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    // 16-byte chunks, four of them in flight per thread (a 32 064-entry row is 4 008 chunks: one round of 1024 threads); the scalar form
    // of this loop -- 125 dependent 2-byte loads per thread -- took 39 us per token
    const int nchunk = (((uintptr_t)row & 15) == 0) ? (V >> 3) : 0;
    for (int c0 = threadIdx.x; c0 < nchunk; c0 += 4 * 1024) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = c0 + u * 1024 < nchunk ? *(const uint4*)(row + (long)(c0 + u * 1024) * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u * 1024 < nchunk) {
                float f[8];
                unpack8(q[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (f[j] > best) { best = f[j]; idx = (c0 + u * 1024) * 8 + j; }      // increasing index per thread: the first maximum stays
            }
        }
    }
    for (int c = nchunk * 8 + threadIdx.x; c < V; c += 1024) {
        const float v = e2f(row[c]);
        if (v > best || (v == best && c < idx)) { best = v; idx = c; }
    }
    if (idx == 0x7fffffff) idx = threadIdx.x < V ? threadIdx.x : 0;   // a row of NaNs: any valid index (torch returns the first NaN)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        int live = unfinished[b];
        long tok = idx;
        if (!live && has_pad) tok = pad;
        seq[(long)b * seq_ld + pos] = tok;
        if (live) {
            for (int e = 0; e < n_eos; ++e)
                if (eos[e] == tok) live = 0;
            unfinished[b] = live;
        }
        if (live) atomicAdd(alive, 1);
    }
}

}  // namespace

extern "C" int ULL_FN(ull_greedy_step_)(const void* logits, int64_t row_stride, int64_t B, int64_t V, void* unfinished, const void* eos, int64_t n_eos,
                                    int64_t pad, int has_pad, void* seq, int64_t seq_ld, int64_t pos, void* alive, void* stream) {
    if (!logits || !unfinished || !seq || !alive || B <= 0 || V <= 0 || pos < 0 || pos >= seq_ld || (n_eos > 0 && !eos)) return ULL_ERR_ARG;
    hipLaunchKernelGGL(greedy_step_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, (const elem_t*)logits, (long)row_stride, (int)V,
                       (int32_t*)unfinished, (const int64_t*)eos, (int)n_eos, (long)pad, has_pad, (int64_t*)seq, (long)seq_ld, (int)pos, (int32_t*)alive);
    return ull_check_launch();
}

#ifndef ULL_ELEM_F16      // integer work: exists once (bf16 build of this file)
extern "C" int ull_mm_spans(const void* ids, int64_t B, int64_t S, int64_t img_start, int64_t img_end, int64_t vid_start, int64_t vid_end,
                            int64_t vocab, void* spans, void* stream) {
    if (!ids || !spans || B <= 0 || S <= 0) return ULL_ERR_ARG;
    if (B > 1024) return ULL_ERR_SHAPE;
    hipLaunchKernelGGL(mm_spans_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int64_t*)ids, (int)B, (int)S, (int)img_start,
                       (int)img_end, (int)vid_start, (int)vid_end, (long)vocab, (int32_t*)spans);
    return ull_check_launch();
}
#endif

extern "C" int ULL_FN(ull_embed_splice_)(const void* ids, const void* table, const void* img_feat, int64_t n_img_tok, int64_t img_pitch,
                                     int64_t img_off, const void* vid_feat, int64_t n_vid_tok, const void* spans, void* out, int64_t B,
                                     int64_t S, int64_t D, int64_t vocab, void* stream) {
    if (!ids || !table || !out || B <= 0 || S <= 0 || vocab <= 0) return ULL_ERR_ARG;
    if (D & 7) return ULL_ERR_SHAPE;
    const long rows = B * S;
    const int cpr = (int)(D >> 3);
    const int rows_per_block = 256 / (cpr < 256 ? cpr : 256);
    const unsigned blocks = (unsigned)((rows + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(embed_splice_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const int64_t*)ids, (const elem_t*)table,
                       (const elem_t*)img_feat, (int)n_img_tok, (int)img_pitch, (int)img_off, (const elem_t*)vid_feat, (int)n_vid_tok, (const int32_t*)spans,
                       (elem_t*)out, (int)S, (int)D, rows, (long)vocab);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_video_pool_)(const void* f, void* out, int64_t B, int64_t T, int64_t N, int64_t D, int64_t tok_pitch,
                                   int64_t tok_off, void* stream) {
    if (!f || !out || B <= 0 || T <= 0 || N <= 0 || D <= 0) return ULL_ERR_ARG;
    hipLaunchKernelGGL(video_pool_kernel, dim3((unsigned)(T + N), (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const elem_t*)f,
                       (elem_t*)out, (int)T, (int)N, (int)D, (int)tok_pitch, (int)tok_off);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_gather_rows_)(const void* src, int64_t lds_, const void* idx, void* dst, int64_t ldd, int64_t n, int64_t D,
                                    void* stream) {
    if (!src || !idx || !dst) return ULL_ERR_ARG;
    if (n == 0) return ULL_OK;
    if ((D & 7) || (lds_ & 7) || (ldd & 7)) return ULL_ERR_SHAPE;
    const long chunks = D >> 3;
    const unsigned ny = (unsigned)(chunks > 4096 ? (chunks / 2048 < 512 ? chunks / 2048 : 512) : 1);       // >= 8 chunks (128 B) per thread
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n, ny), dim3(256), 0, (hipStream_t)stream, (const elem_t*)src, lds_, (const int64_t*)idx,
                       (elem_t*)dst, ldd, (int)D);
    return ull_check_launch();
}

// nn.Dropout in training mode (PEFT's lora_dropout in front of lora_A, train_ullava.py:222): y = keep ? rnd(x * scale) : 0 with
// scale = 1 / (1 - p); the keep mask (one byte per element) comes from the caller's RNG.  The backward is the same map on dy.
namespace {
__global__ __launch_bounds__(256) void dropout_apply_kernel(const elem_t* __restrict__ x, const uint8_t* __restrict__ keep, elem_t* __restrict__ y,
                                                            long n, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = keep[i] ? f2e(e2f(x[i]) * scale) : f2e(0.f);
}
}  // namespace
extern "C" int ULL_FN(ull_dropout_apply_)(const void* x, const void* keep, void* y, int64_t n, float scale, void* stream) {
    if (!x || !keep || !y || n <= 0) return ULL_ERR_ARG;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const elem_t*)x, (const uint8_t*)keep, (elem_t*)y,
                       (long)n, scale);
    return ull_check_launch();
}

extern "C" int ULL_FN(ull_add_rows_)(const void* a, const void* b, void* out, int64_t rows, int64_t D, int64_t b_rows, void* stream) {
    if (!a || !b || !out || rows <= 0 || b_rows <= 0) return ULL_ERR_ARG;
    if (D & 7) return ULL_ERR_SHAPE;
    const long total = rows * (D >> 3);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(add_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const elem_t*)a, (const elem_t*)b, (elem_t*)out, rows,
                       (int)D, b_rows);
    return ull_check_launch();
}
