"""Thin torch-tensor wrappers over the C-ABI (device pointers + sizes + current HIP stream).

torch is used for device memory and streams only; every arithmetic op on the path is a HIP kernel from
libullava_hip.so.  Activations are 16-bit (bf16 or fp16: every kernel exists in both builds, picked by the dtype of the op's first
operand; the other operands must match), row-major, last dim contiguous.
"""
import weakref
from typing import Optional


import torch

from . import _lib

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32
# entry-point suffix by element type: the two 16-bit builds of every dtype-dependent kernel, and the fp32 build of the inference path
# (csrc/f32.hip: `--dtype fp32`, plain kernels on the exact fp32 matrix instruction -- a correctness path, no fused / tiled fast paths)
_SFX = {BF16: "bf16", F16: "f16", F32: "f32"}
DT_CODE = {torch.float32: 0, BF16: 1, F16: 2}               # ULL_DT_* of the dtype-coded entry points
EPI_BIAS, EPI_QGELU, EPI_GELU, EPI_RELU, EPI_RESID, EPI_SWIGLU, EPI_F32 = 1, 1 << 1, 2 << 1, 3 << 1, 8, 16, 32
EPI_BIAS_ROUNDED = 256
ACTS = {None: 0, "quick_gelu": EPI_QGELU, "gelu": EPI_GELU, "relu": EPI_RELU}


_ZEROS = {}


def _zeros(device):
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(256, device=device, dtype=BF16)
    return z


def vt_unpermute_index(pitch: int) -> torch.Tensor:
    """index so that vt[..., idx] is in natural key order (inverse of the 32-key block permutation of transpose_v)."""
    key = torch.arange(pitch)
    w = key % 32
    a, g, r = w // 16, (w % 16) // 4, w % 4
    return (key // 32) * 32 + 8 * g + 4 * a + r


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _copy_sfx(dtype, D: int):
    """(entry suffix, row width in that entry's elements) of a PURE data-movement kernel: fp32 rows go through the 16-bit kernel as rows of
    twice as many 16-bit elements (the same bytes; csrc/f32.hip has no copy kernels of its own)."""
    if dtype == F32:
        if D % 4:
            raise RuntimeError("u-llava_amd: fp32 rows must be a multiple of 4 elements wide")
        return "bf16", 2 * D
    return _SFX[dtype], D


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=None):
    """dtype None: any 16-bit element type with a kernel build (bf16 / fp16); otherwise exactly `dtype` (the op's other operands
    must match its first one -- there are no mixed-dtype kernels)."""
    if not t.is_cuda:
        raise RuntimeError(f"u-llava_amd: `{name}` must live on the GPU (no CPU path exists)")
    if dtype is None:
        if t.dtype not in _SFX:
            raise RuntimeError(f"u-llava_amd: `{name}` must be torch.bfloat16, torch.float16 or torch.float32, got {t.dtype}")
    elif t.dtype != dtype:
        raise RuntimeError(f"u-llava_amd: `{name}` must be {dtype}, got {t.dtype}")
    if t.dim() and t.stride(-1) != 1:
        raise RuntimeError(f"u-llava_amd: `{name}` must be contiguous in its last dim")


def _rows(t: torch.Tensor):
    """View [*, D] as rows with one row stride; returns (rows, ld)."""
    if t.dim() == 1:
        return 1, t.shape[0]
    if t.dim() == 2:
        return t.shape[0], t.stride(0)
    if not t.is_contiguous():
        raise RuntimeError("u-llava_amd: >2-D operands must be contiguous")
    return t.numel() // t.shape[-1], t.shape[-1]


EPI_W_TILED = 64
_TILED = {}          # data_ptr of a row-major weight -> (weakref to it, its tile-major copy, the weight's ._version the copy was made from)


def tile_major(w: torch.Tensor) -> torch.Tensor:
    """[N, K] (K % 64 == 0) -> [ceil(N/256), K/64, 256, 64] contiguous: the ULL_EPI_W_TILED layout of ull_gemm_bf16."""
    N, K = w.shape
    Np = (N + 255) // 256 * 256
    if Np != N:
        w = torch.cat([w, w.new_zeros(Np - N, K)])
    return w.view(Np // 256, 256, K // 64, 64).permute(0, 2, 1, 3).contiguous()


def register_tiled(w: torch.Tensor) -> None:
    """Keep a tile-major copy of a weight for the prefill-shape GEMM (the row-major original still feeds the decode GEMV)."""
    if w.dim() == 2 and w.shape[1] % 64 == 0 and w.shape[0] >= 512 and w.shape[1] >= 128 and w.is_contiguous() and w.dtype != F32:
        ptr = w.data_ptr()
        _TILED[ptr] = [weakref.ref(w, lambda _r, _p=ptr: _TILED.pop(_p, None) if (_TILED.get(_p) or (None,))[0] is _r else None),
                       tile_major(w.detach()), w._version]    # the copy is dropped when the weight tensor dies


def _tiled_of(w: torch.Tensor):
    """The tile-major copy of `w`, or None.  The copy describes the weight AS IT WAS when the copy was made: an in-place update since
    (an optimizer step's `p.copy_`, `load_state_dict`, `merge_lora`) bumps `w._version`, and the copy is then re-made from the live
    values before it is handed out -- a training run must never multiply by step-0 weights."""
    e = _TILED.get(w.data_ptr())
    if e is None:
        return None
    if e[0]() is not w:
        _TILED.pop(w.data_ptr(), None)       # the address was recycled by another tensor
        return None
    if e[2] != w._version:
        # a NEW tensor, not copy_: a copy first made under torch.inference_mode() (the reference wraps generate / evaluate in it,
        # inference_ullava_core.py:72, models/ullava.py:349) is an inference tensor and may not be updated in place outside that mode
        # ... in place where that is allowed (a training run re-tiles every weight after every optimizer step: no allocator churn)
        N, K = w.shape
        if not e[1].is_inference() and not torch.is_inference_mode_enabled() and N % 256 == 0 and tuple(e[1].shape) == (N // 256, K // 64, 256, 64):
            with torch.no_grad():
                e[1].copy_(w.detach().view(N // 256, 256, K // 64, 64).permute(0, 2, 1, 3))
        else:
            e[1] = tile_major(w.detach())
        e[2] = w._version
    return e[1]


# ---- stream-K workspace: one fp32 scratch buffer per (device, stream) ---------------------------------------------------
# ull_gemm_bf16 splits a partial last round of 256x256 tiles along K into fp32 slabs in a CALLER-owned workspace.  Each HIP
# stream gets its own buffer here, so GEMMs running concurrently on two streams (RES forward / evaluate: SAM encoder beside
# CLIP + LLaMA) never share scratch.  Whether to split at all is host policy: K >= streamk_min_k (default 2048: below that the
# slab round trip costs more than the partial round it replaces); `streamk_policy(None)` turns the split off, which the model
# does while a second stream is filling the idle CUs of a partial round anyway.
_SK_WS = {}
_SK_MIN_K = [2048]


def _streamk_ws(device: torch.device, stream_id: int):
    key = (device.index, stream_id)
    ws = _SK_WS.get(key)
    if ws is None:
        with torch.cuda.device(device):
            ws = _SK_WS[key] = torch.empty(_lib.query("ull_gemm_streamk_ws_bytes"), device=device, dtype=torch.uint8)
    return ws


# Latency mode for single-image prefills (M < 1024): split-K in the 128x128 kernel.  OFF by default: it makes a sample's result depend on
# the batch it is computed in (the fp32 summation order of K changes with the split), and the path's invariant -- sample b of a batch of
# 32 equals the single-sample run bit for bit through every layer (tests/test_full_depth_gpu.py) -- is worth more than 5 ms of prefill.
# `with ops.small_m_split_k(True):` turns it on (C2 shape at batch 1: 16.9 -> 11.9 ms, 59 -> 84 images/s).  (Round 6: no environment variable
# reads this any more -- the product has no environment switches.)
_SMALL_M_SPLIT_K = [False]


class small_m_split_k:
    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        self.prev = _SMALL_M_SPLIT_K[0]
        _SMALL_M_SPLIT_K[0] = self.on
        return self

    def __exit__(self, *exc):
        _SMALL_M_SPLIT_K[0] = self.prev
        return False


class streamk_policy:
    """with streamk_policy(min_k): ... -- K threshold of the stream-K tail inside the block (None = never split)."""

    def __init__(self, min_k: Optional[int]):
        self.min_k = min_k

    def __enter__(self):
        self.prev = _SK_MIN_K[0]
        _SK_MIN_K[0] = self.min_k
        return self

    def __exit__(self, *exc):
        _SK_MIN_K[0] = self.prev
        return False


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: Optional[str] = None,
           residual: Optional[torch.Tensor] = None, swiglu: bool = False, out: Optional[torch.Tensor] = None,
           out_f32: bool = False, rms_w: Optional[torch.Tensor] = None, rms_eps: float = 0.0, tune: int = 0,
           bias_after_rounding: bool = False) -> torch.Tensor:
    """y = epilogue(x @ w.T).  x [..., K]; w [N, K] (nn.Linear layout).  swiglu: w is the 16-row interleaved gate/up pack.
    rms_w/rms_eps: apply LlamaRMSNorm to x first (fused into the GEMV prologue at decode shapes, a separate kernel otherwise).
    tune: ULL_GEMM_TUNE_* bits (tools/ only).  bias_after_rounding: y = round(round(x @ w.T) + bias) -- what at::linear computes
    for a NON-contiguous 3-D input (matmul + add_ instead of the fused addmm)."""
    _chk(x, "x"); _chk(w, "w", x.dtype)
    M, ldx = _rows(x)
    N, K = w.shape
    lead = tuple(x.shape[:-1])
    if x.shape[-1] != K:
        raise RuntimeError(f"u-llava_amd.linear: K mismatch {x.shape[-1]} vs {K}")
    if x.dtype == F32:
        # fp32 build: one plain GEMM kernel for every shape (any M / N / K / strides), the preceding RMSNorm as its own launch
        if rms_w is not None:
            x = rmsnorm(x, rms_w, rms_eps)
            M, ldx = _rows(x)
        if w.stride(1) != 1:
            w = w.contiguous()
        n_out = N // 2 if swiglu else N
        if out is None:
            out = torch.empty(*lead, n_out, device=x.device, dtype=F32)
        flags = ACTS[act] | (EPI_BIAS if bias is not None else 0) | (EPI_RESID if residual is not None else 0) | (EPI_SWIGLU if swiglu else 0)
        if bias is not None:
            _chk(bias, "bias", F32)
        ldr = 0
        if residual is not None:
            _chk(residual, "residual", F32)
            ldr = _rows(residual)[1]
        _lib.call("ull_gemm_f32", _p(x), ldx, _p(w), w.stride(0), _p(out), _rows(out)[1], _p(bias), _p(residual), ldr, M, N, K, flags, None, 0, _stream())
        return out
    # batched decode steps against LLaMA-sized weights: the weight stream on the matrix cores (the GEMV is FMA-bound from M = 4 on and
    # measured slower from M = 3 on (decode step at batch 3: 4.47 vs 4.24 ms; at batch 2 the GEMV wins, 4.00 vs 4.08); the tiled GEMM's grid
    # is a few dozen blocks at these M).  Small weights stay where they were.
    skinny = 3 <= M <= 16 and K % 32 == 0 and N * K >= (1 << 22) and w.stride(0) % 8 == 0 and w.stride(1) == 1 and tune == 0
    if rms_w is not None and (skinny or not (M <= 4 and K % 8 == 0 and M * K <= 16384)):
        x = rmsnorm(x, rms_w, rms_eps)
        rms_w = None
        M, ldx = _rows(x)
    if skinny:
        n_out = N // 2 if swiglu else N
        if out is None:
            out = torch.empty(*lead, n_out, device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
        flags = ACTS[act] | (EPI_BIAS if bias is not None else 0) | (EPI_RESID if residual is not None else 0) | \
            (EPI_SWIGLU if swiglu else 0) | (EPI_F32 if out_f32 else 0) | (EPI_BIAS_ROUNDED if bias_after_rounding else 0)
        if bias is not None:
            _chk(bias, "bias", x.dtype)
        ldr = 0
        if residual is not None:
            _chk(residual, "residual", x.dtype)
            ldr = _rows(residual)[1]
        _lib.call("ull_gemm_skinny_" + _SFX[x.dtype], _p(x), ldx, _p(w), w.stride(0), _p(out), _rows(out)[1], _p(bias), _p(residual), ldr, M, N, K,
                  flags, _stream())
        return out
    if M <= 4 and K % 8 == 0:
        # decode shape: weight-streaming GEMV (no padding, no MFMA)
        n_out = N // 2 if swiglu else N
        if out is None:
            out = torch.empty(*lead, n_out, device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
        flags = ACTS[act] | (EPI_BIAS if bias is not None else 0) | (EPI_RESID if residual is not None else 0) | \
            (EPI_SWIGLU if swiglu else 0) | (EPI_F32 if out_f32 else 0) | (EPI_BIAS_ROUNDED if bias_after_rounding else 0)
        ldr = _rows(residual)[1] if residual is not None else 0
        if rms_w is not None:
            _chk(rms_w, "rms_w", x.dtype)
            _lib.call("ull_gemv_rmsnorm_" + _SFX[x.dtype], _p(x), ldx, _p(rms_w), float(rms_eps), _p(w), w.stride(0), _p(out), _rows(out)[1], _p(bias),
                      _p(residual), ldr, M, N, K, flags, _stream())
        else:
            _lib.call("ull_gemv_" + _SFX[x.dtype], _p(x), ldx, _p(w), w.stride(0), _p(out), _rows(out)[1], _p(bias), _p(residual), ldr, M, N, K, flags,
                      _stream())
        return out
    if K % 64:
        # The MFMA kernel consumes K in 64-wide DMA tiles.  Every real width on the path (1024, 1280, 4096, 5120,
        # 11008, 256, 128, 2048, 64, patch K padded by im2col) is a multiple of 64; only the tiny test models are not.
        # Zero-padding K is exact (adds 0*0 terms).
        Kp = ((K + 63) // 64) * 64
        x = torch.nn.functional.pad(x.reshape(M, K) if x.dim() != 2 else x, (0, Kp - K))
        w = torch.nn.functional.pad(w, (0, Kp - K))
        K, ldx = Kp, Kp
    n_out = N // 2 if swiglu else N
    if out is None:
        out = torch.empty(*lead, n_out, device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    flags = ACTS[act] | (EPI_BIAS if bias is not None else 0) | (EPI_RESID if residual is not None else 0) | \
        (EPI_SWIGLU if swiglu else 0) | (EPI_F32 if out_f32 else 0) | (EPI_BIAS_ROUNDED if bias_after_rounding else 0)
    if bias is not None:
        _chk(bias, "bias", x.dtype)
    ldr = 0
    if residual is not None:
        _chk(residual, "residual", x.dtype)
        _, ldr = _rows(residual)
    _, ldc = _rows(out)
    big = M >= 1024 and N >= 512 and K >= 128
    wt = _tiled_of(w) if big else None
    st = _stream()
    ws_ptr, ws_bytes = None, 0
    min_k = _SK_MIN_K[0]
    if big and min_k is not None and K >= min_k:
        ws = _streamk_ws(x.device, st)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    elif _SMALL_M_SPLIT_K[0] and not big and min_k is not None and K >= 1024 and M >= 128 and N * K >= (1 << 22) and -(-M // 128) * -(-N // 128) <= 256:
        # opt-in latency mode (small_m_split_k): few 128x128 tiles, a long K and a weight of at least 8 MB (single-image prefill: o_proj /
        # down_proj, CLIP's fc2): the kernel splits K over the idle CUs (376 -> 256 us per LLaMA layer at M = 323)
        ws = _streamk_ws(x.device, st)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    if wt is not None:
        _lib.call("ull_gemm_" + _SFX[x.dtype], _p(x), ldx, _p(wt), K, _p(out), ldc, _p(bias), _p(residual), ldr, M, N, K, flags | EPI_W_TILED | tune,
                  ws_ptr, ws_bytes, st)
    else:
        _lib.call("ull_gemm_" + _SFX[x.dtype], _p(x), ldx, _p(w), w.stride(0), _p(out), ldc, _p(bias), _p(residual), ldr, M, N, K, flags | tune,
                  ws_ptr, ws_bytes, st)
    return out


def rope_table(positions: torch.Tensor, inv_freq: torch.Tensor, dtype) -> tuple:
    """(cos, sin) [tokens, hd/2] of `dtype` for int64 positions [tokens]: LlamaRotaryEmbedding's fp32 cos / sin cast to the model dtype."""
    _chk(positions, "positions", torch.int64); _chk(inv_freq, "inv_freq", torch.float32)
    if dtype not in _SFX:
        raise RuntimeError("u-llava_amd.rope_table: dtype must be bf16 or fp16")
    T, half = positions.numel(), inv_freq.numel()
    cs = torch.empty(T, half, device=positions.device, dtype=dtype)
    sn = torch.empty(T, half, device=positions.device, dtype=dtype)
    _lib.call("ull_rope_table_" + _SFX[dtype], _p(positions), _p(inv_freq), T, half, _p(cs), _p(sn), _stream())
    return cs, sn


GEMM_TUNE_WAVES8, GEMM_TUNE_WAVES4 = 1 << 21, 1 << 22     # ULL_GEMM_TUNE_*: force one form of the 256x256 kernel (tests / tools)


def linear_qkv_rope(x: torch.Tensor, w: torch.Tensor, rope_cos: torch.Tensor, rope_sin: torch.Tensor, rope_cols: int, head_dim: int,
                    out: Optional[torch.Tensor] = None, tune: int = 0) -> torch.Tensor:
    """Fused q|k|v projection + RoPE on the first `rope_cols` output columns (q and k heads), head_dim 128, K % 64 == 0, M > 4.
    Bit-identical to linear() followed by rope_inplace()."""
    _chk(x, "x"); _chk(w, "w", x.dtype); _chk(rope_cos, "rope_cos", x.dtype); _chk(rope_sin, "rope_sin", x.dtype)
    M, ldx = _rows(x)
    N, K = w.shape
    if x.shape[-1] != K or K % 64 or head_dim != 128 or rope_cos.shape != (M, 64) or not rope_cos.is_contiguous() or not rope_sin.is_contiguous():
        raise RuntimeError("u-llava_amd.linear_qkv_rope: needs K % 64 == 0, head_dim == 128 and [M, 64] contiguous tables")
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=x.dtype)
    _, ldc = _rows(out)
    big = M >= 1024 and N >= 512 and K >= 128
    wt = _tiled_of(w) if big else None
    st = _stream()
    ws_ptr, ws_bytes = None, 0
    min_k = _SK_MIN_K[0]
    if big and min_k is not None and K >= min_k:
        ws = _streamk_ws(x.device, st)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    if wt is not None:
        _lib.call("ull_gemm_qkv_rope_" + _SFX[x.dtype], _p(x), ldx, _p(wt), K, _p(out), ldc, M, N, K, _p(rope_cos), _p(rope_sin), rope_cols,
                  head_dim, EPI_W_TILED | tune, ws_ptr, ws_bytes, st)
    else:
        _lib.call("ull_gemm_qkv_rope_" + _SFX[x.dtype], _p(x), ldx, _p(w), w.stride(0), _p(out), ldc, M, N, K, _p(rope_cos), _p(rope_sin),
                  rope_cols, head_dim, tune, ws_ptr, ws_bytes, st)
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, "x"); _chk(w, "w", x.dtype)
    rows, ldx = _rows(x)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    _, ldy = _rows(out)
    _lib.call("ull_rmsnorm_" + _SFX[x.dtype], _p(x), ldx, _p(w), _p(out), ldy, rows, x.shape[-1], float(eps), _stream())
    return out


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, "x"); _chk(w, "w", x.dtype); _chk(b, "b", x.dtype)
    rows, ldx = _rows(x)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    _, ldy = _rows(out)
    _lib.call("ull_layernorm_" + _SFX[x.dtype], _p(x), ldx, _p(w), _p(b), _p(out), ldy, rows, x.shape[-1], float(eps), _stream())
    return out


def clip_embed_ln(patch: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, w: torch.Tensor, b: torch.Tensor, n_img: int, tokens: int,
                  eps: float) -> torch.Tensor:
    """patch [n_img*(tokens-1), D] -> pre-LN hidden states [n_img, tokens, D]."""
    _chk(patch, "patch")
    for t, n in ((cls, "cls"), (pos, "pos"), (w, "w"), (b, "b")):
        _chk(t, n, patch.dtype)
    D = patch.shape[-1]
    out = torch.empty(n_img, tokens, D, device=patch.device, dtype=patch.dtype)
    _lib.call("ull_clip_embed_ln_" + _SFX[patch.dtype], _p(patch), patch.stride(0), _p(cls), _p(pos), _p(w), _p(b), _p(out), D, n_img, tokens, D,
              float(eps), _stream())
    return out


_REL_POS_CACHE: dict = {}          # id(table) -> (weakref(table), version, n, resized): only the very same tensor object may hit


def fit_rel_pos(rel_pos: torch.Tensor, size: int) -> torch.Tensor:
    """get_rel_pos's resize (image_encoder.py:333-345): a table whose length is not 2 * size - 1 is linearly interpolated to that length.
    The resized copy is kept for the tensor OBJECT it was made from (a module parameter), never by address: a temporary's storage
    is reused by the allocator."""
    n = 2 * size - 1
    if rel_pos.shape[0] == n:
        return rel_pos
    _chk(rel_pos, "rel_pos")
    hit = _REL_POS_CACHE.get(id(rel_pos))
    if hit is not None and hit[0]() is rel_pos and hit[1] == rel_pos._version and hit[2] == n:
        return hit[3]
    out = torch.empty(n, rel_pos.shape[1], device=rel_pos.device, dtype=rel_pos.dtype)
    _lib.call("ull_interp_rows_linear_" + _SFX[rel_pos.dtype], _p(rel_pos.contiguous()), _p(out), rel_pos.shape[0], n, rel_pos.shape[1], _stream())
    for k in [k for k, v in _REL_POS_CACHE.items() if v[0]() is None]:
        del _REL_POS_CACHE[k]
    _REL_POS_CACHE[id(rel_pos)] = (weakref.ref(rel_pos), rel_pos._version, n, out)
    return out


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, B: int, H: int, Sq: int, Sk: int, hd: int,
              q_strides, k_strides, o_strides, key_mask: Optional[torch.Tensor] = None, causal: bool = False, scale_mode: int = 1,
              scale: float = 1.0, q_scale: float = 1.0, rel_h: Optional[torch.Tensor] = None, rel_w: Optional[torch.Tensor] = None,
              rel_pos_hw: Optional[tuple] = None, v_strides=None):
    """rel_h/rel_w: either per-query bias tables [B*H, Sq, KH|KW] (from sam_relpos), or -- with rel_pos_hw=(KH, KW) -- the raw
    rel_pos_h / rel_pos_w parameters [2KH-1, hd] / [2KW-1, hd], in which case the kernel builds the tables itself.
    v_strides = (batch, head, token) strides: `vt` is V ITSELF ([B,H,Sk,hd] by strides), not its transposed image; the kernels that
    transpose on the fly (LLaMA / CLIP prefill) take it directly, for any other shape the V^T image is made here first."""
    _chk(q, "q"); _chk(k, "k", q.dtype); _chk(vt, "vt", q.dtype); _chk(out, "out", q.dtype)
    rel_mode, kh, kw = 0, 0, 0
    if rel_h is not None and rel_pos_hw is not None and q.dtype == F32:
        # fp32 build: the per-query bias tables come from their own kernel (rel_mode 1); the 16-bit kernels build them in place (rel_mode 2)
        kh, kw = rel_pos_hw
        rel_h, rel_w = sam_relpos(q, q_strides, fit_rel_pos(rel_h, kh), fit_rel_pos(rel_w, kw), B, H, kh, kw, hd)
        rel_pos_hw = None
    if rel_h is not None:
        _chk(rel_h, "rel_h", q.dtype); _chk(rel_w, "rel_w", q.dtype)
        if rel_pos_hw is not None:
            rel_mode, (kh, kw) = 2, rel_pos_hw
            rel_h, rel_w = fit_rel_pos(rel_h, kh), fit_rel_pos(rel_w, kw)
            if rel_h.shape[1] != hd or rel_w.shape[1] != hd:
                raise ValueError(f"rel_pos tables of width {rel_h.shape[1]} / {rel_w.shape[1]} for head_dim {hd}")
        else:
            rel_mode, kh, kw = 1, rel_h.shape[-1], rel_w.shape[-1]
    if key_mask is not None:
        _chk(key_mask, "key_mask", torch.int32)
    name = "ull_attention_" + _SFX[q.dtype]
    tail = (_p(out), *o_strides, _p(key_mask), B, H, Sq, Sk, hd, int(causal), scale_mode, float(scale), float(q_scale), _p(rel_h), _p(rel_w),
            kh, kw, rel_mode, _zeros(q.device).data_ptr(), _stream())
    if v_strides is not None:
        rc = _lib.query(name, _p(q), *q_strides, _p(k), *k_strides, _p(vt), *v_strides, 0, *tail)
        if rc == 0:
            return out
        if rc != -2:                               # anything but "no kernel of that form for this shape"
            raise RuntimeError(f"u-llava_amd: {name} failed: {_lib.ERRORS.get(rc, rc)}")
        if v_strides[1] != hd:
            raise ValueError("V rows must have their heads contiguous (head stride == head_dim) to be transposed here")
        vt = transpose_v(vt, v_strides[0], v_strides[2], B, Sk, H, hd)
    pitch = vt.shape[-1]
    _lib.call(name, _p(q), *q_strides, _p(k), *k_strides, _p(vt), H * hd * pitch, hd * pitch, pitch, pitch, *tail)
    return out


def dropout_apply(x: torch.Tensor, keep: torch.Tensor, p: float) -> torch.Tensor:
    """y = keep ? x / (1 - p) : 0 (keep: uint8, same number of elements)."""
    _chk(x, "x"); _chk(keep, "keep", torch.uint8)
    x = x.contiguous()
    y = torch.empty_like(x)
    _lib.call("ull_dropout_apply_" + _SFX[x.dtype], _p(x), _p(keep), _p(y), x.numel(), float(1.0 / (1.0 - p)), _stream())
    return y


def sam_window_attention(qkv: torch.Tensor, pad_row: torch.Tensor, rel_pos_h: torch.Tensor, rel_pos_w: torch.Tensor, B: int, H: int, W: int,
                         nH: int, hd: int, ws: int) -> torch.Tensor:
    """Block.forward's window_partition + Attention + window_unpartition (image_encoder.py:176-190) on image-order tokens:
    qkv [B*H*W, 3*nH*hd] -> [B*H*W, nH*hd].  pad_row: the qkv bias (= the q|k|v of a zero-padded token)."""
    _chk(qkv, "qkv"); _chk(pad_row, "pad_row", qkv.dtype); _chk(rel_pos_h, "rel_pos_h", qkv.dtype); _chk(rel_pos_w, "rel_pos_w", qkv.dtype)
    rel_pos_h, rel_pos_w = fit_rel_pos(rel_pos_h, ws), fit_rel_pos(rel_pos_w, ws)
    C = nH * hd
    if qkv.shape != (B * H * W, 3 * C) or pad_row.numel() != 3 * C:
        raise ValueError(f"qkv {tuple(qkv.shape)} / pad_row {tuple(pad_row.shape)} for B={B} H={H} W={W} C={C}")
    out = torch.empty(B * H * W, C, device=qkv.device, dtype=qkv.dtype)
    _lib.call("ull_sam_window_attention_" + _SFX[qkv.dtype], _p(qkv), 3 * C, _p(pad_row), _p(rel_pos_h), _p(rel_pos_w), _p(out), C,
              B, H, W, nH, hd, ws, float(hd ** -0.5), _zeros(qkv.device).data_ptr(), _stream())
    return out


def rope_inplace(x: torch.Tensor, row_stride: int, positions: torch.Tensor, inv_freq: torch.Tensor, tokens: int, n_heads: int, hd: int):
    _chk(x, "x"); _chk(positions, "positions", torch.int64); _chk(inv_freq, "inv_freq", torch.float32)
    _lib.call("ull_rope_inplace_" + _SFX[x.dtype], _p(x), row_stride, _p(positions), _p(inv_freq), tokens, n_heads, hd, _stream())


def rope_append(qkv: torch.Tensor, row_stride: int, positions: torch.Tensor, inv_freq: torch.Tensor, B: int, S: int, H: int, hd: int,
                k_cache: torch.Tensor, vt_cache: torch.Tensor, smax: int, past: int):
    """decode step: RoPE on q (in place) and k + append of k / v to the KV cache (K [B,H,smax,hd], V^T [B,H,hd,smax] permuted)."""
    _chk(qkv, "qkv"); _chk(positions, "positions", torch.int64); _chk(inv_freq, "inv_freq", torch.float32)
    _chk(k_cache, "k_cache", qkv.dtype); _chk(vt_cache, "vt_cache", qkv.dtype)
    _lib.call("ull_rope_append_" + _SFX[qkv.dtype], _p(qkv), row_stride, _p(positions), _p(inv_freq), B, S, H, hd, _p(k_cache), _p(vt_cache), smax, past,
              _stream())


def linear_qkv_rope_append(x: torch.Tensor, w_qkv: torch.Tensor, rope_cos: torch.Tensor, rope_sin: torch.Tensor, B: int, S: int, H: int, hd: int,
                           k_cache: torch.Tensor, vt_cache: torch.Tensor, smax: int, past: int, rms_w: Optional[torch.Tensor] = None,
                           rms_eps: float = 0.0) -> torch.Tensor:
    """decode step (B * S <= 4 tokens): q | k | v projection of x (optionally RMS-normalised first) with RoPE and the KV-cache append in the
    GEMV's epilogue.  Returns the rotated queries [B * S, H * hd]; the rotated keys / the values land in the caches.  Same bits as
    `linear(x, w_qkv, rms_w=...)` followed by `rope_append`."""
    _chk(x, "x"); _chk(w_qkv, "w_qkv", x.dtype); _chk(rope_cos, "rope_cos", x.dtype); _chk(rope_sin, "rope_sin", x.dtype)
    _chk(k_cache, "k_cache", x.dtype); _chk(vt_cache, "vt_cache", x.dtype)
    T, K = x.shape
    if T != B * S or T > 4 or w_qkv.shape != (3 * H * hd, K) or rope_cos.shape != (T, hd // 2) or rope_sin.shape != (T, hd // 2):
        raise RuntimeError("u-llava_amd.linear_qkv_rope_append: shapes (at most 4 tokens; w [3 * H * hd, K]; cos / sin [tokens, hd / 2])")
    if x.stride(1) != 1 or w_qkv.stride(1) != 1 or not (rope_cos.is_contiguous() and rope_sin.is_contiguous()):
        raise RuntimeError("u-llava_amd.linear_qkv_rope_append: rows must be contiguous")
    if rms_w is not None:
        _chk(rms_w, "rms_w", x.dtype)
    q = torch.empty(T, H * hd, device=x.device, dtype=x.dtype)
    _lib.call("ull_gemv_qkv_rope_append_" + _SFX[x.dtype], _p(x), x.stride(0), _p(rms_w), float(rms_eps), _p(w_qkv), w_qkv.stride(0), _p(q),
              q.stride(0), _p(rope_cos), _p(rope_sin), _p(k_cache), _p(vt_cache), B, S, H, hd, K, smax, past, _stream())
    return q


def transpose_v(v: torch.Tensor, v_bs: int, v_ss: int, B: int, S: int, H: int, hd: int, pitch: Optional[int] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(v, "v")
    pitch = pitch or ((S + 63) // 64) * 64
    vt = out if out is not None else torch.empty(B, H, hd, pitch, device=v.device, dtype=v.dtype)
    _lib.call("ull_transpose_v_" + _SFX[v.dtype], _p(v), v_bs, v_ss, _p(vt), B, S, H, hd, pitch, _stream())
    return vt


def im2col(img: torch.Tensor, ps: int, Kp: int) -> torch.Tensor:
    _chk(img, "img")
    if not img.is_contiguous():
        raise RuntimeError("u-llava_amd.im2col: image batch must be contiguous NCHW")
    n, C, H, W = img.shape
    out = torch.empty(n * (H // ps) * (W // ps), Kp, device=img.device, dtype=img.dtype)
    _lib.call("ull_im2col_" + _SFX[img.dtype], _p(img), _p(out), n, C, H, W, ps, Kp, _stream())
    return out


def pack_patch_weight(w: torch.Tensor) -> torch.Tensor:
    """Conv2d weight [N, C, ps, ps] -> the [N, Kp] layout of ull_patchify_bf16: (c, ky) segments of 16 with kx >= ps zero, zero
    segments up to a multiple of 64."""
    N, C, ps, ps2 = w.shape
    if ps != ps2 or ps > 16 or ps % 2:
        raise NotImplementedError("patch sizes on the path: 14 (CLIP) and 16 (SAM)")
    K = C * ps * 16
    Kp = (K + 63) // 64 * 64
    out = torch.zeros(N, Kp, device=w.device, dtype=w.dtype)
    out[:, :K].view(N, C * ps, 16)[:, :, :ps] = w.reshape(N, C * ps, ps)
    return out


def patchify(img: torch.Tensor, wp: torch.Tensor, ps: int, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Conv2d(kernel = stride = ps) patch embedding straight from the pixels: img [n, C, H, W] bf16 -> [n*(H/ps)*(W/ps), N]."""
    _chk(img, "img"); _chk(wp, "packed patch weight", img.dtype)
    if not img.is_contiguous():
        raise RuntimeError("u-llava_amd.patchify: image batch must be contiguous NCHW")
    n, C, H, W = img.shape
    N, Kp = wp.shape
    out = torch.empty(n * (H // ps) * (W // ps), N, device=img.device, dtype=img.dtype)
    _lib.call("ull_patchify_" + _SFX[img.dtype], _p(img), n, C, H, W, ps, _p(wp), Kp, _p(bias), _p(out), N, N, _zeros(img.device).data_ptr(), _stream())
    return out


def mm_spans(ids: torch.Tensor, img_start: int, img_end: int, vid_start: int, vid_end: int, vocab: int = 0) -> torch.Tensor:
    """int32 [B, 4] = {kind, first start pos, feature index, error bits (1: start/end counts differ, 2: id outside [0, vocab))}."""
    _chk(ids, "input_ids", torch.int64)
    B, S = ids.shape
    spans = torch.empty(B, 4, device=ids.device, dtype=torch.int32)
    _lib.call("ull_mm_spans", _p(ids), B, S, img_start, img_end, vid_start, vid_end, vocab, _p(spans), _stream())
    return spans


def embed_splice(ids: torch.Tensor, table: torch.Tensor, img_feat: Optional[torch.Tensor], vid_feat: Optional[torch.Tensor],
                 spans: Optional[torch.Tensor], img_tokens: int = 0, img_pitch: int = 0, img_off: int = 0) -> torch.Tensor:
    """img_feat [n_img, img_pitch, D] (rows img_off..img_off+img_tokens of each image are spliced); vid_feat [n_vid, n_tok, D]."""
    _chk(ids, "input_ids", torch.int64); _chk(table, "embed_tokens")
    B, S = ids.shape
    D = table.shape[1]
    out = torch.empty(B, S, D, device=ids.device, dtype=table.dtype)
    if img_feat is not None:
        _chk(img_feat, "img_feat", table.dtype)
        img_pitch = img_pitch or img_feat.shape[-2]
        img_tokens = img_tokens or img_pitch - img_off
    n_vid = 0
    if vid_feat is not None:
        _chk(vid_feat, "vid_feat", table.dtype)
        n_vid = vid_feat.shape[-2]
    sfx, De = _copy_sfx(table.dtype, D)
    _lib.call("ull_embed_splice_" + sfx, _p(ids), _p(table), _p(img_feat), img_tokens, img_pitch, img_off, _p(vid_feat), n_vid, _p(spans),
              _p(out), B, S, De, table.shape[0], _stream())
    return out


def greedy_step(logits_last: torch.Tensor, unfinished: torch.Tensor, eos_ids: Optional[torch.Tensor], pad: Optional[int], seq: torch.Tensor,
                pos: int, alive: torch.Tensor) -> None:
    """One greedy step's bookkeeping in one launch (see ull_greedy_step_*): logits_last [B, V] (rows may be strided), unfinished int32 [B]
    (updated), eos_ids int64 [n] or None, seq int64 [B, L] (column `pos` written), alive int32 [1] (+= rows still unfinished)."""
    _chk(logits_last, "logits"); _chk(unfinished, "unfinished", torch.int32); _chk(seq, "seq", torch.int64); _chk(alive, "alive", torch.int32)
    B, V = logits_last.shape
    if seq.stride(1) != 1 or not unfinished.is_contiguous():
        raise RuntimeError("u-llava_amd.greedy_step: seq rows / unfinished must be contiguous")
    n_eos = 0
    if eos_ids is not None:
        _chk(eos_ids, "eos_ids", torch.int64)
        n_eos = eos_ids.numel()
    _lib.call("ull_greedy_step_" + _SFX[logits_last.dtype], _p(logits_last), logits_last.stride(0), B, V, _p(unfinished), _p(eos_ids), n_eos,
              int(pad) if pad is not None else 0, int(pad is not None), _p(seq), seq.stride(0), int(pos), _p(alive), _stream())


def video_pool(f: torch.Tensor, B: int, T: int, N: int, tok_pitch: Optional[int] = None, tok_off: int = 0) -> torch.Tensor:
    """f [B*T, tok_pitch, D]; patches are tokens tok_off..tok_off+N of every frame."""
    _chk(f, "f")
    D = f.shape[-1]
    out = torch.empty(B, T + N, D, device=f.device, dtype=f.dtype)
    _lib.call("ull_video_pool_" + _SFX[f.dtype], _p(f), _p(out), B, T, N, D, tok_pitch or N, tok_off, _stream())
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _chk(src, "src"); _chk(idx, "idx", torch.int64)
    n, D = idx.numel(), src.shape[-1]
    out = torch.empty(n, D, device=src.device, dtype=src.dtype)
    if n:
        sfx, De = _copy_sfx(src.dtype, D)
        k = De // D
        _lib.call("ull_gather_rows_" + sfx, _p(src), src.stride(-2) * k, _p(idx), _p(out), De, n, De, _stream())
    return out


def add_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bf16(a + b) with b broadcast over leading rows (b has b_rows rows, a has k*b_rows)."""
    _chk(a, "a"); _chk(b, "b", a.dtype)
    D = a.shape[-1]
    rows, b_rows = a.numel() // D, b.numel() // D
    out = torch.empty_like(a)
    _lib.call("ull_add_rows_" + _SFX[a.dtype], _p(a), _p(b), _p(out), rows, D, b_rows, _stream())
    return out


# ---- SAM --------------------------------------------------------------------------------------------------------------
def window_partition(x: torch.Tensor, B: int, H: int, W: int, ws: int) -> torch.Tensor:
    """x [B*H*W, C] token-major -> [B*nW*ws*ws, C]."""
    _chk(x, "x")
    C = x.shape[-1]
    nW = ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    out = torch.empty(B * nW * ws * ws, C, device=x.device, dtype=x.dtype)
    sfx, Ce = _copy_sfx(x.dtype, C)
    _lib.call("ull_window_partition_" + sfx, _p(x), _p(out), B, H, W, Ce, ws, _stream())
    return out


def window_unpartition_add(win: torch.Tensor, shortcut: torch.Tensor, B: int, H: int, W: int, ws: int) -> torch.Tensor:
    _chk(win, "win"); _chk(shortcut, "shortcut", win.dtype)
    out = torch.empty_like(shortcut)
    _lib.call("ull_window_unpartition_add_" + _SFX[win.dtype], _p(win), _p(shortcut), _p(out), B, H, W, shortcut.shape[-1], ws, _stream())
    return out


def sam_relpos(q: torch.Tensor, q_strides, rel_pos_h: torch.Tensor, rel_pos_w: torch.Tensor, NB: int, nH: int, KH: int, KW: int, hd: int):
    _chk(q, "q"); _chk(rel_pos_h, "rel_pos_h", q.dtype); _chk(rel_pos_w, "rel_pos_w", q.dtype)
    rel_pos_h, rel_pos_w = fit_rel_pos(rel_pos_h, KH), fit_rel_pos(rel_pos_w, KW)
    oh = torch.empty(NB * nH, KH * KW, KH, device=q.device, dtype=q.dtype)
    ow = torch.empty(NB * nH, KH * KW, KW, device=q.device, dtype=q.dtype)
    _lib.call("ull_sam_relpos_" + _SFX[q.dtype], _p(q), *q_strides, _p(rel_pos_h), _p(rel_pos_w), _p(oh), _p(ow), NB, nH, KH, KW, hd, _stream())
    return oh, ow


def layernorm2d_cl(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6, gelu: bool = False) -> torch.Tensor:
    _chk(x, "x"); _chk(w, "w", x.dtype); _chk(b, "b", x.dtype)
    if not x.is_contiguous():
        raise RuntimeError("u-llava_amd.layernorm2d_cl: x must be contiguous channels-last rows")
    C = x.shape[-1]
    out = torch.empty_like(x)
    _lib.call("ull_layernorm2d_cl_" + _SFX[x.dtype], _p(x), _p(w), _p(b), _p(out), x.numel() // C, C, float(eps), int(gelu), _stream())
    return out


def neck_layernorm2d_f32(xa: torch.Tensor, xb: Optional[torch.Tensor], xb_scale: float, w: torch.Tensor, b: torch.Tensor, eps: float,
                         split: bool) -> torch.Tensor:
    """LayerNorm2d of the fp16 model's fp32 neck (image_encoder.py:117-124): fp32 rows xa (+ xb * xb_scale) -> fp16.
    split=False: [rows, C] = fp16(y).  split=True: [2, rows, C] = (fp16(y), fp16((y - fp16(y)) * 2^11)), the two-term image of y."""
    _chk(xa, "xa", torch.float32); _chk(w, "w", torch.float16); _chk(b, "b", torch.float16)
    C = xa.shape[-1]
    rows = xa.numel() // C
    if not xa.is_contiguous() or (xb is not None and (not xb.is_contiguous() or xb.shape != xa.shape or xb.dtype != torch.float32)):
        raise RuntimeError("u-llava_amd.neck_layernorm2d_f32: xa / xb must be contiguous fp32 rows of the same shape")
    out = torch.empty((2, rows, C) if split else (rows, C), device=xa.device, dtype=torch.float16)
    _lib.call("ull_neck_layernorm2d_f32in_f16", _p(xa), _p(xb), float(xb_scale), _p(w), _p(b), _p(out), _p(out[1]) if split else None, rows, C,
              float(eps), _stream())
    return out


def im2col3x3(x: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    _chk(x, "x")
    C = x.shape[-1]
    out = torch.empty(B * H * W, 9 * C, device=x.device, dtype=x.dtype)
    _lib.call("ull_im2col3x3_" + _SFX[x.dtype], _p(x), _p(out), B, H, W, C, _stream())
    return out


def mask_matmul(hyper: torch.Tensor, up: torch.Tensor, n: int, T: int, C: int, G: int) -> torch.Tensor:
    _chk(hyper, "hyper"); _chk(up, "up", hyper.dtype)
    out = torch.empty(n, T, 4 * G, 4 * G, device=up.device, dtype=hyper.dtype)
    _lib.call("ull_mask_matmul_" + _SFX[hyper.dtype], _p(hyper), _p(up), _p(out), n, T, C, G, _stream())
    return out


def bilinear(x: torch.Tensor, in_h: int, in_w: int, out_h: int, out_w: int) -> torch.Tensor:
    """x [n, Hfull, Wfull] (bf16 or fp32, contiguous); the top-left in_h x in_w crop of every image is resized -> fp32 [n, out_h, out_w]."""
    if not x.is_cuda or x.dtype not in DT_CODE or not x.is_contiguous():
        raise RuntimeError("u-llava_amd.bilinear: contiguous bf16/fp16/fp32 GPU tensor required")
    n, Hf, Wf = x.shape
    out = torch.empty(n, out_h, out_w, device=x.device, dtype=torch.float32)
    _lib.call("ull_bilinear_f32", _p(x), DT_CODE[x.dtype], Hf * Wf, Wf, in_h, in_w, _p(out), n, out_h, out_w, _stream())
    return out


def shifted_cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """mean CE of logits[:, :-1] against labels[:, 1:] (ignore_index -100) -> 0-dim tensor of the logits' dtype like the reference's loss."""
    _chk(logits, "logits"); _chk(labels, "labels", torch.int64)
    B, S, V = logits.shape
    acc = torch.zeros(2, device=logits.device, dtype=torch.float32)
    _lib.call("ull_shifted_cross_entropy_" + _SFX[logits.dtype], _p(logits), logits.stride(1), _p(labels.contiguous()), B, S, V, _p(acc), _stream())
    return (acc[0] / acc[1]).to(logits.dtype)


def mask_loss_sums(logits: torch.Tensor, target: torch.Tensor, scale: float = 1000.0) -> torch.Tensor:
    """fp32 [n, 4] = per mask {sum BCE-with-logits, sum (sigmoid/scale)*t, sum sigmoid/scale, sum t/scale} (models/loss.py)."""
    _chk(logits, "mask logits", torch.float32); _chk(target, "mask target", torch.float32)
    if logits.shape != target.shape or not logits.is_contiguous() or not target.is_contiguous():
        raise RuntimeError("u-llava_amd.mask_loss_sums: logits / target must be contiguous and of the same shape")
    n = logits.shape[0]
    part = torch.empty(n, 64, 4, device=logits.device, dtype=torch.float32)
    _lib.call("ull_mask_loss_sums_f32", _p(logits), _p(target), n, logits[0].numel(), float(scale), _p(part), _stream())
    return part.sum(1)


def box_losses(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """fp32 [2] = {sum |pred - gt|, sum (1 - GIoU) over well-formed predictions}; pred [n,4] bf16/fp32, gt [n,4] fp32."""
    if pred.dtype not in DT_CODE:
        raise RuntimeError("u-llava_amd.box_losses: pred must be bf16, fp16 or fp32")
    _chk(pred, "pred boxes", pred.dtype); _chk(gt, "gt boxes", torch.float32)
    out = torch.empty(2, device=pred.device, dtype=torch.float32)
    _lib.call("ull_box_losses_f32", _p(pred.contiguous()), DT_CODE[pred.dtype], _p(gt.contiguous()), pred.shape[0], _p(out), _stream())
    return out


# ---- backward kernels (training path; see autograd_ops.py) ----------------------------------------------------------------------------
def rmsnorm_bwd(x: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, eps: float, need_dw: bool = True):
    """-> (dx like x, dw float32 [D] or None)."""
    _chk(x, "x"); _chk(w, "w", x.dtype); _chk(dy, "dy", x.dtype)
    rows, ldx = _rows(x)
    D = x.shape[-1]
    dx = torch.empty_like(x)
    dw = torch.zeros(D, device=x.device, dtype=torch.float32) if need_dw else None
    _lib.call("ull_rmsnorm_bwd_" + _SFX[x.dtype], _p(x), ldx, _p(w), _p(dy), _rows(dy)[1], _p(dx), _rows(dx)[1], _p(dw), rows, D, float(eps), _stream())
    return dx, dw


def swiglu_fwd(gu: torch.Tensor, halves: bool = False) -> torch.Tensor:
    """gu [M, 2I] -> silu(gate) * up [M, I].  Columns: the 16-wide gate/up interleave of the inference pack, or (halves) [gate | up]."""
    _chk(gu, "gu")
    M, I = gu.numel() // gu.shape[-1], gu.shape[-1] // 2
    a = torch.empty(*gu.shape[:-1], I, device=gu.device, dtype=gu.dtype)
    _lib.call("ull_swiglu_fwd_" + _SFX[gu.dtype], _p(gu.contiguous()), _p(a), M, I, int(halves), _stream())
    return a


def swiglu_bwd(gu: torch.Tensor, da: torch.Tensor, halves: bool = False) -> torch.Tensor:
    _chk(gu, "gu"); _chk(da, "da", gu.dtype)
    M, I = gu.numel() // gu.shape[-1], gu.shape[-1] // 2
    dgu = torch.empty_like(gu)
    _lib.call("ull_swiglu_bwd_" + _SFX[gu.dtype], _p(gu.contiguous()), _p(da.contiguous()), _p(dgu), M, I, int(halves), _stream())
    return dgu


def relu_mask(y: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dx = y > 0 ? dy : 0 (ReLU backward as a selection)."""
    _chk(y, "y"); _chk(dy, "dy", y.dtype)
    y, dy = y.contiguous(), dy.contiguous()
    dx = torch.empty_like(dy)
    _lib.call("ull_relu_mask_" + _SFX[y.dtype], _p(y), _p(dy), _p(dx), y.numel(), _stream())
    return dx


def rope_bwd_inplace(dx: torch.Tensor, row_stride: int, positions: torch.Tensor, inv_freq: torch.Tensor, tokens: int, n_heads: int, hd: int):
    _chk(dx, "dx"); _chk(positions, "positions", torch.int64); _chk(inv_freq, "inv_freq", torch.float32)
    _lib.call("ull_rope_bwd_inplace_" + _SFX[dx.dtype], _p(dx), row_stride, _p(positions), _p(inv_freq), tokens, n_heads, hd, _stream())


def attention_bwd(q, k, v, o, do, dq, dk, dv, strides, key_mask, B: int, H: int, Sq: int, Sk: int, hd: int, causal: bool, mult: float):
    """strides: 8 triples (batch, head, seq) of element strides for q, k, v, o, do, dq, dk, dv (hd contiguous everywhere)."""
    import ctypes
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o"), (do, "do"), (dq, "dq"), (dk, "dk"), (dv, "dv")):
        _chk(t, n, q.dtype if t is not q else None)
    if key_mask is not None:
        _chk(key_mask, "key_mask", torch.int32)
    flat = [int(x) for tr in strides for x in tr]
    assert len(flat) == 24
    arr = (ctypes.c_int64 * 24)(*flat)
    (q_bs, q_hs, q_ss), (k_bs, k_hs, k_ss), _, _, (g_bs, g_hs, g_ss) = strides[:5]
    if hd in (64, 128) and q_hs == hd and k_hs == hd and g_hs == hd and all(x % 4 == 0 for x in flat) and all(flat[i] % 8 == 0 for i in range(2, 24, 3)):
        # matrix-core kernels.  hd 128 (the LLaMA block): operand tiles staged through the LDS, transposed fragments from the
        # transposing LDS read.  hd 64: fragments straight from global memory; Q, K and dO also as transpose_v images.
        pitch = ((max(Sq, Sk) + 63) // 64) * 64
        qt = kt = gt = None
        if hd != 128:
            qt = transpose_v(q, q_bs, q_ss, B, Sq, H, hd, pitch)
            kt = transpose_v(k, k_bs, k_ss, B, Sk, H, hd, pitch)
            gt = transpose_v(do, g_bs, g_ss, B, Sq, H, hd, pitch)
        scratch = torch.empty(2 * B * H * (((Sq + 63) // 64) * 64), device=q.device, dtype=torch.float32)
        _lib.call("ull_attention_bwd_mfma_" + _SFX[q.dtype], _p(q), _p(k), _p(v), _p(o), _p(do), _p(qt), _p(kt), _p(gt), pitch, _p(dq), _p(dk), _p(dv),
                  arr, _p(key_mask), B, H, Sq, Sk, hd, int(causal), float(mult), _p(scratch), _stream())
        return
    scratch = torch.empty(2 * B * H * Sq, device=q.device, dtype=torch.float32)
    _lib.call("ull_attention_bwd_" + _SFX[q.dtype], _p(q), _p(k), _p(v), _p(o), _p(do), _p(dq), _p(dk), _p(dv), arr, _p(key_mask), B, H, Sq, Sk, hd,
              int(causal), float(mult), _p(scratch), _stream())


def shifted_cross_entropy_stats(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """float32 [2] = {sum of token losses, counted tokens} of the shifted CE (the loss is stats[0] / stats[1])."""
    _chk(logits, "logits"); _chk(labels, "labels", torch.int64)
    B, S, V = logits.shape
    acc = torch.zeros(2, device=logits.device, dtype=torch.float32)
    _lib.call("ull_shifted_cross_entropy_" + _SFX[logits.dtype], _p(logits), logits.stride(1), _p(labels.contiguous()), B, S, V, _p(acc), _stream())
    return acc


def shifted_cross_entropy_bwd(logits: torch.Tensor, labels: torch.Tensor, stats: torch.Tensor, gout: torch.Tensor) -> torch.Tensor:
    _chk(logits, "logits"); _chk(labels, "labels", torch.int64); _chk(stats, "stats", torch.float32); _chk(gout, "gout", torch.float32)
    B, S, V = logits.shape
    dl = torch.empty_like(logits)
    _lib.call("ull_shifted_cross_entropy_bwd_" + _SFX[logits.dtype], _p(logits), logits.stride(1), _p(labels.contiguous()), B, S, V, _p(stats),
              _p(gout), _p(dl), _stream())
    return dl


def embed_splice_bwd(ids, demb, vocab: int, img_shape=None, vid_shape=None, spans=None, img_tokens: int = 0, img_pitch: int = 0, img_off: int = 0,
                     need_table: bool = True, vid_tokens: int = 0, detach_text: bool = False):
    """-> (d_table float32 [vocab, D] or None, d_img [n_img, pitch, D] or None, d_vid or None); rows not written stay zero.
    img_tokens / vid_tokens: span lengths (needed even when d_img / d_vid are not: span rows never reach the table).
    detach_text: projector_from_scratch -- only the start / end token rows of samples with an image / video reach the table."""
    _chk(ids, "input_ids", torch.int64); _chk(demb, "demb")
    B, S = ids.shape
    D = demb.shape[-1]
    d_table = torch.zeros(vocab, D, device=demb.device, dtype=torch.float32) if need_table else None
    d_img = torch.zeros(img_shape, device=demb.device, dtype=demb.dtype) if img_shape is not None else None
    d_vid = torch.zeros(vid_shape, device=demb.device, dtype=demb.dtype) if vid_shape is not None else None
    n_vid = vid_shape[-2] if vid_shape is not None else vid_tokens
    _lib.call("ull_embed_splice_bwd_" + _SFX[demb.dtype], _p(ids), _p(demb.contiguous()), _p(d_table), _p(d_img), img_tokens, img_pitch, img_off,
              _p(d_vid), n_vid, _p(spans), B, S, D, vocab, int(detach_text), _stream())
    return d_table, d_img, d_vid


def transpose2d(x: torch.Tensor) -> torch.Tensor:
    """[R, C] (last dim contiguous, any row stride) -> fresh contiguous [C, R]."""
    _chk(x, "x")
    if x.dim() != 2:
        raise RuntimeError("u-llava_amd.transpose2d: needs a 2-D tensor")
    R, C = x.shape
    y = torch.empty(C, R, device=x.device, dtype=x.dtype)
    _lib.call("ull_transpose2d_" + _SFX[x.dtype], _p(x), x.stride(0) if R > 1 else C, _p(y), R, R, C, _stream())
    return y


def colsum(x: torch.Tensor) -> torch.Tensor:
    """float32 [N] column sums of x [rows, N]."""
    _chk(x, "x")
    rows, ld = _rows(x)
    out = torch.empty(x.shape[-1], device=x.device, dtype=torch.float32)
    _lib.call("ull_colsum_" + _SFX[x.dtype], _p(x), ld, rows, x.shape[-1], _p(out), _stream())
    return out


def sum_slabs(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """x [R, n] -> scale * sum over R (fp32 accumulation), same dtype."""
    _chk(x, "x")
    if not x.is_contiguous() or x.dim() != 2:
        raise RuntimeError("u-llava_amd.sum_slabs: contiguous [R, n] required")
    out = torch.empty(x.shape[1], device=x.device, dtype=x.dtype)
    _lib.call("ull_sum_slabs_" + _SFX[x.dtype], _p(x), _p(out), x.shape[0], x.shape[1], float(scale), _stream())
    return out


def layernorm_bwd(x, w, dy, eps: float, need_wb: bool = True):
    """-> (dx, dw float32 [D] or None, db float32 [D] or None)."""
    _chk(x, "x"); _chk(w, "w", x.dtype); _chk(dy, "dy", x.dtype)
    rows, ldx = _rows(x)
    D = x.shape[-1]
    dx = torch.empty_like(x)
    dw = torch.zeros(D, device=x.device, dtype=torch.float32) if need_wb else None
    db = torch.zeros(D, device=x.device, dtype=torch.float32) if need_wb else None
    _lib.call("ull_layernorm_bwd_" + _SFX[x.dtype], _p(x), ldx, _p(w), _p(dy), _rows(dy)[1], _p(dx), _rows(dx)[1], _p(dw), _p(db), rows, D, float(eps),
              _stream())
    return dx, dw, db


def layernorm2d_cl_bwd(x, w, b, dy, eps: float, gelu: bool, need_wb: bool = True):
    _chk(x, "x"); _chk(w, "w", x.dtype); _chk(b, "b", x.dtype); _chk(dy, "dy", x.dtype)
    C = x.shape[-1]
    dx = torch.empty_like(x)
    dw = torch.zeros(C, device=x.device, dtype=torch.float32) if need_wb else None
    db = torch.zeros(C, device=x.device, dtype=torch.float32) if need_wb else None
    _lib.call("ull_layernorm2d_cl_bwd_" + _SFX[x.dtype], _p(x.contiguous()), _p(w), _p(b), _p(dy.contiguous()), _p(dx), _p(dw), _p(db), x.numel() // C, C,
              float(eps), int(gelu), _stream())
    return dx, dw, db


def gelu_fwd(x):
    _chk(x, "x")
    y = torch.empty_like(x)
    _lib.call("ull_gelu_fwd_" + _SFX[x.dtype], _p(x.contiguous()), _p(y), x.numel(), _stream())
    return y


def gelu_bwd(x, dy):
    _chk(x, "x"); _chk(dy, "dy", x.dtype)
    dx = torch.empty_like(x)
    _lib.call("ull_gelu_bwd_" + _SFX[x.dtype], _p(x.contiguous()), _p(dy.contiguous()), _p(dx), x.numel(), _stream())
    return dx


def mask_matmul_bwd(hyper, up, dmasks, n: int, T: int, C: int, G: int):
    """-> (dhyper float32 [n, T, C], dup like up)."""
    _chk(hyper, "hyper"); _chk(up, "up", hyper.dtype); _chk(dmasks, "dmasks", hyper.dtype)
    dh = torch.zeros(n, T, C, device=up.device, dtype=torch.float32)
    dup = torch.empty_like(up)
    _lib.call("ull_mask_matmul_bwd_" + _SFX[hyper.dtype], _p(hyper.contiguous()), _p(up), _p(dmasks.contiguous()), _p(dh), _p(dup), n, T, C, G, _stream())
    return dh, dup


def mask_loss_sums_bwd(logits, target, g, scale: float = 1000.0):
    _chk(logits, "mask logits", torch.float32); _chk(target, "mask target", torch.float32); _chk(g, "g", torch.float32)
    n = logits.shape[0]
    dl = torch.empty_like(logits)
    _lib.call("ull_mask_loss_sums_bwd_f32", _p(logits.contiguous()), _p(target.contiguous()), _p(g.contiguous()), n, logits[0].numel(), float(scale),
              _p(dl), _stream())
    return dl


def box_losses_bwd(pred, gt, gw):
    _chk(pred, "pred boxes", pred.dtype if pred.dtype in DT_CODE else None); _chk(gt, "gt boxes", torch.float32); _chk(gw, "gw", torch.float32)
    dp = torch.empty(pred.shape[0], 4, device=pred.device, dtype=torch.float32)
    _lib.call("ull_box_losses_bwd_f32", _p(pred.contiguous()), DT_CODE[pred.dtype], _p(gt.contiguous()), pred.shape[0], _p(gw.contiguous()), _p(dp),
              _stream())
    return dp


def bilinear_bwd(dout, full_hw, in_h: int, in_w: int):
    """adjoint of bilinear(): dout fp32 [n, out_h, out_w] -> din fp32 [n, Hfull, Wfull] (zero outside the in_h x in_w crop)."""
    _chk(dout, "dout", torch.float32)
    n, oh, ow = dout.shape
    Hf, Wf = full_hw
    din = torch.zeros(n, Hf, Wf, device=dout.device, dtype=torch.float32)
    _lib.call("ull_bilinear_bwd_f32", _p(dout.contiguous()), _p(din), Hf * Wf, Wf, in_h, in_w, n, oh, ow, _stream())
    return din


# ---- fused two-way mask decoder (csrc/sam_decoder.hip) ------------------------------------------------------------------------------
def _lin_ptrs(*lins):
    out = []
    for l in lins:
        out += [_p(l.weight), _p(l.bias)]
    return out


def _proj_args(projs, n: int, T: int, like: torch.Tensor):
    """projs: list of up to 3 (linear holder, add_pe) -> (ctypes pointer array or None, ctypes int array or None, output tensors)."""
    import ctypes
    if not projs:
        return None, None, []
    outs, ptrs, flags = [], [], []
    for lin, add_pe in projs:
        o = torch.empty(n, T, lin.weight.shape[0], device=like.device, dtype=like.dtype)
        outs.append(o)
        ptrs += [_p(lin.weight), _p(lin.bias), _p(o)]
        flags.append(int(add_pe))
    while len(flags) < 3:
        ptrs += [None, None, None]
        flags.append(0)
    return (ctypes.c_void_p * 9)(*ptrs), (ctypes.c_int * 3)(*flags), outs


def sam_self_attn_heads(queries, qpe, a, first: bool):
    """queries / qpe [n, T, 256] -> concatenated head outputs [n, T, 256] of the token self attention (before out_proj)."""
    _chk(queries, "queries"); _chk(qpe, "qpe", queries.dtype)
    n, T, _ = queries.shape
    att = torch.empty_like(queries)
    _lib.call("ull_sam_self_attn_heads_" + _SFX[queries.dtype], _p(queries), _p(qpe), n, T, int(first), *_lin_ptrs(a.q_proj, a.k_proj, a.v_proj), _p(att),
              _stream())
    return att


def sam_out_ln(att, res, qpe, out_proj, ln, projs=None, eps: float = 1e-5):
    """LayerNorm(res + out_proj(att)) (res None: no residual) + token-side projections for the next attention -> (out, [proj outputs])."""
    _chk(att, "att"); _chk(qpe, "qpe", att.dtype)
    n, T, din = att.shape
    out = torch.empty(n, T, 256, device=att.device, dtype=att.dtype)
    pa, fa, outs = _proj_args(projs, n, T, att)
    _lib.call("ull_sam_out_ln_" + _SFX[att.dtype], _p(att), din, _p(res), _p(qpe), n, T, _p(out_proj.weight), _p(out_proj.bias), _p(ln.weight), _p(ln.bias),
              float(eps), _p(out), pa, fa, _stream())
    return out, outs


def sam_token_mlp_ln(queries, qpe, lin1, lin2, ln, projs=None, eps: float = 1e-5):
    _chk(queries, "queries"); _chk(qpe, "qpe", queries.dtype)
    n, T, _ = queries.shape
    hidden = lin1.weight.shape[0]
    ws = torch.empty(n * (hidden // 256) * 8 * 256, device=queries.device, dtype=torch.float32)
    out = torch.empty_like(queries)
    pa, fa, outs = _proj_args(projs, n, T, queries)
    _lib.call("ull_sam_token_mlp_ln_" + _SFX[queries.dtype], _p(queries), _p(qpe), n, T, hidden, *_lin_ptrs(lin1, lin2), _p(ln.weight), _p(ln.bias),
              float(eps), _p(ws), _p(out), pa, fa, _stream())
    return out, outs


def sam_small_mlps(hs, hyper_mlps, iou_head):
    """hs [n, T, 256] -> (hyper [n, 4, C], iou [n, n_iou]): the 4 hyper-network MLPs on rows 1..4 and the IoU head on row 0."""
    import ctypes
    _chk(hs, "hs")
    n, T, _ = hs.shape
    ptrs = []
    for m in list(hyper_mlps) + [iou_head]:
        ptrs += _lin_ptrs(*m.layers)
    arr = (ctypes.c_void_p * 30)(*ptrs)
    C, n_iou = hyper_mlps[0].layers[2].weight.shape[0], iou_head.layers[2].weight.shape[0]
    hyper = torch.empty(n, 4, C, device=hs.device, dtype=hs.dtype)
    iou = torch.empty(n, n_iou, device=hs.device, dtype=hs.dtype)
    _lib.call("ull_sam_small_mlps_" + _SFX[hs.dtype], _p(hs), n, T, arr, 4, C, n_iou, _p(hyper), _p(iou), _stream())
    return hyper, iou


def sam_t2i_attention(qproj, keys, pos, a, late_bias_kv: bool):
    """token -> image attention core: qproj [n, T, 128], keys [n, P, 256], pos [P, 256] -> attention output [n, T, 128] (before out_proj)."""
    _chk(qproj, "qproj"); _chk(keys, "keys", qproj.dtype); _chk(pos, "pos", qproj.dtype)
    n, T, _ = qproj.shape
    P = keys.shape[1]
    ws_s = torch.empty(n * 8 * 8 * P, device=keys.device, dtype=keys.dtype)
    ws_v = torch.empty(n * P * 128, device=keys.device, dtype=keys.dtype)
    att = torch.empty(n, T, 128, device=keys.device, dtype=keys.dtype)
    _lib.call("ull_sam_t2i_attention_" + _SFX[qproj.dtype], _p(qproj), _p(keys), _p(pos), n, T, P, *_lin_ptrs(a.k_proj, a.v_proj), int(late_bias_kv),
              _p(ws_s), _p(ws_v), _p(att), _stream())
    return att


def sam_i2t_attention_ln(keys, pos, kproj, vproj, a, ln, late_bias_q: bool, eps: float = 1e-5):
    """image -> token attention + residual + LayerNorm in one launch -> new keys [n, P, 256]."""
    _chk(keys, "keys"); _chk(pos, "pos", keys.dtype); _chk(kproj, "kproj", keys.dtype); _chk(vproj, "vproj", keys.dtype)
    n, T, _ = kproj.shape
    P = keys.shape[1]
    out = torch.empty_like(keys)
    _lib.call("ull_sam_i2t_attention_ln_" + _SFX[keys.dtype], _p(keys), _p(pos), _p(kproj), _p(vproj), n, T, P, *_lin_ptrs(a.q_proj, a.out_proj),
              int(late_bias_q), _p(ln.weight), _p(ln.bias), float(eps), _p(out), _stream())
    return out


# ---- optimizer kernels (csrc/optim.hip; host: optim.py) --------------------------------------------------------------------------------
def adamw_step(master: torch.Tensor, m: torch.Tensor, v: torch.Tensor, grad: torch.Tensor, param_out: Optional[torch.Tensor], lr: float,
               beta1: float, beta2: float, eps: float, weight_decay: float, step: int, grad_scale: float = 1.0) -> None:
    """One AdamW step on a flat shard: master / m / v fp32 (in place), grad any of fp32 / bf16 / fp16, param_out 16-bit or fp32 or None."""
    for t, n in ((master, "master"), (m, "m"), (v, "v")):
        _chk(t, n, torch.float32)
    if not grad.is_cuda or grad.dtype not in DT_CODE or not grad.is_contiguous() or grad.numel() != master.numel():
        raise RuntimeError("u-llava_amd.adamw_step: grad must be a contiguous GPU tensor of the shard's size (fp32 / bf16 / fp16)")
    pd = 0
    if param_out is not None:
        if not param_out.is_cuda or param_out.dtype not in DT_CODE or not param_out.is_contiguous() or param_out.numel() != master.numel():
            raise RuntimeError("u-llava_amd.adamw_step: param_out must be a contiguous GPU tensor of the shard's size")
        pd = DT_CODE[param_out.dtype]
    _lib.call("ull_adamw_step_f32", _p(master), _p(m), _p(v), _p(grad), DT_CODE[grad.dtype], _p(param_out), pd, master.numel(), float(lr), float(beta1),
              float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _stream())


def sumsq(g: torch.Tensor, out: torch.Tensor) -> None:
    """out[0] (fp32) += sum of squares of g."""
    _chk(out, "out", torch.float32)
    if not g.is_cuda or g.dtype not in DT_CODE or not g.is_contiguous():
        raise RuntimeError("u-llava_amd.sumsq: contiguous GPU tensor (fp32 / bf16 / fp16) required")
    if g.numel():
        _lib.call("ull_sumsq_f32", _p(g), DT_CODE[g.dtype], g.numel(), _p(out), _stream())


# ---- coarse entries (include/ullava_hip.h "coarse entries", csrc/layers.hip): one C call enqueues a whole stack of layers ---------------------
# The per-op wrappers above cost a ctypes round trip + Python marshalling per LAUNCH (~15 us); a C4 step has ~400 launches, a decode step ~160.
# A LayerStack holds the ctypes array of per-layer structs (weight / bias / norm pointers) and refreshes it when a pointer it recorded has
# moved (a re-made tile-major copy after an optimizer step, a re-pack).  Results are bit-identical to the per-op path: same entries, same
# dispatch rules (layers.hip `lin` / `lin_decode` mirror `linear` above).
COARSE = [True]     # `with ops.per_op_layers():` = the per-op path (tests A/B the two)


class per_op_layers:
    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev = COARSE[0]
        COARSE[0] = not self.on
        return self

    def __exit__(self, *exc):
        COARSE[0] = self.prev
        return False


def coarse_ok() -> bool:
    return COARSE[0] and not _SMALL_M_SPLIT_K[0]


class LayerStack:
    """`kind`: _lib.LlamaLayer / ClipLayer / SamBlock; `layers`: one dict per layer, field name -> tensor (pointer fields) or
    (weight, bias-or-None) (ull_linear fields) or int (plain fields)."""

    def __init__(self, kind, layers):
        self.kind, self.layers = kind, layers
        self.arr = (kind * len(layers))()
        self._lin_fields = [n for n, t in kind._fields_ if t is _lib.Linear]
        self._seen = None
        self.refresh()

    def _fingerprint(self):
        # (address, version) of every weight: an in-place update (optimizer step, load_state_dict) bumps the version, and only then is the
        # tile-major copy re-made (`_tiled_of`, in refresh below) -- between updates the copy's address cannot change
        return [v for d in self.layers for n in self._lin_fields for v in (d[n][0].data_ptr(), d[n][0]._version)]

    def refresh(self):
        fp = self._fingerprint()
        if fp == self._seen:
            return self.arr
        for i, d in enumerate(self.layers):
            s = self.arr[i]
            for n, t in self.kind._fields_:
                v = d[n]
                if t is _lib.Linear:
                    w, b = v
                    if w.dim() != 2 or w.stride(1) != 1:
                        raise RuntimeError("u-llava_amd: coarse entries need row-major 2-D weights")
                    wt = _tiled_of(w)
                    setattr(s, n, _lib.Linear(w.data_ptr(), None if wt is None else wt.data_ptr(), _p(b), w.shape[0], w.shape[1], w.stride(0)))
                elif isinstance(v, int):
                    setattr(s, n, v)
                else:
                    setattr(s, n, v.data_ptr())
        self._seen = fp
        return self.arr


def _sk_args(device):
    """(ws pointer, ws bytes, min_k) of the current stream-K policy for a coarse entry."""
    min_k = _SK_MIN_K[0]
    if min_k is None:
        return None, 0, -1
    ws = _streamk_ws(device, _stream())
    return ws.data_ptr(), ws.numel(), int(min_k)


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def llama_prefill_layers(stack: LayerStack, x_in: torch.Tensor, x_out, rope_cos, rope_sin, key_mask, B: int, S: int, H: int, hd: int, I: int,
                         eps: float):
    """x_in [B*S, D]; x_out: list of n_layers [B*S, D] tensors (may repeat; x_out[0] must not be x_in).  See ull_llama_prefill_layers_bf16."""
    _chk(x_in, "x_in")
    T, D = x_in.shape
    dev, dt = x_in.device, x_in.dtype
    x_mid = torch.empty(T, D, device=dev, dtype=dt)
    xn = torch.empty(T, D, device=dev, dtype=dt)
    qkv = torch.empty(T, 3 * D, device=dev, dtype=dt)
    att = torch.empty(T, D, device=dev, dtype=dt)
    act = torch.empty(T, I, device=dev, dtype=dt)
    ws, wsb, mk = _sk_args(dev)
    _lib.call("ull_llama_prefill_layers_" + _SFX[dt], stack.refresh(), len(stack.layers), _p(x_in), _ptr_array(x_out), _p(x_mid), _p(xn), _p(qkv), _p(att),
              _p(act), _p(rope_cos), _p(rope_sin), _p(key_mask), B, S, H, hd, I, float(eps), ws, wsb, mk, _zeros(dev).data_ptr(), _stream())


def llama_decode_layers(stack: LayerStack, x_in: torch.Tensor, x_out, rope_cos, rope_sin, key_mask, k_ptrs, vt_ptrs, B: int, S: int, H: int, hd: int,
                        I: int, smax: int, past: int, eps: float):
    """One generation step through all layers (T = B*S <= 4).  k_ptrs / vt_ptrs: ctypes void* arrays of the per-layer caches."""
    _chk(x_in, "x_in")
    T, D = x_in.shape
    dev, dt = x_in.device, x_in.dtype
    scratch = torch.empty(T * (4 * D + 2 * max(D, I)), device=dev, dtype=dt)
    x_mid, q, att = scratch[:T * D], scratch[T * D:2 * T * D], scratch[2 * T * D:3 * T * D]
    xn = scratch[3 * T * D:3 * T * D + T * max(D, I)]
    act = scratch[3 * T * D + T * max(D, I):3 * T * D + T * max(D, I) + T * I]
    _lib.call("ull_llama_decode_layers_" + _SFX[dt], stack.refresh(), len(stack.layers), _p(x_in), _ptr_array(x_out), _p(x_mid), _p(xn), _p(q), _p(att),
              _p(act), _p(rope_cos), _p(rope_sin), _p(key_mask), k_ptrs, vt_ptrs, B, S, H, hd, I, smax, past, float(eps), _zeros(dev).data_ptr(), _stream())


def clip_layers(stack: LayerStack, n_layers: int, h: torch.Tensor, n_img: int, S: int, H: int, hd: int, I: int, eps: float):
    """The first n_layers CLIP encoder layers on h [n_img*S, D], in place."""
    _chk(h, "h")
    T, D = h.shape
    dev, dt = h.device, h.dtype
    h_mid = torch.empty(T, D, device=dev, dtype=dt)
    y = torch.empty(T, D, device=dev, dtype=dt)
    qkv = torch.empty(T, 3 * D, device=dev, dtype=dt)
    att = torch.empty(T, D, device=dev, dtype=dt)
    f = torch.empty(T, I, device=dev, dtype=dt)
    ws, wsb, mk = _sk_args(dev)
    _lib.call("ull_clip_layers_" + _SFX[dt], stack.refresh(), n_layers, _p(h), _p(h_mid), _p(y), _p(qkv), _p(att), _p(f), n_img, S, H, hd, I, float(eps),
              ws, wsb, mk, _zeros(dev).data_ptr(), _stream())


def sam_blocks(stack: LayerStack, x: torch.Tensor, B: int, g: int, nH: int, hd: int, I: int, eps: float = 1e-6):
    """All SAM encoder blocks on image-order tokens x [B*g*g, C], in place."""
    _chk(x, "x")
    T, C = x.shape
    dev, dt = x.device, x.dtype
    x_mid = torch.empty(T, C, device=dev, dtype=dt)
    y = torch.empty(T, C, device=dev, dtype=dt)
    qkv = torch.empty(T, 3 * C, device=dev, dtype=dt)
    att = torch.empty(T, C, device=dev, dtype=dt)
    f = torch.empty(T, I, device=dev, dtype=dt)
    ws, wsb, mk = _sk_args(dev)
    _lib.call("ull_sam_blocks_" + _SFX[dt], stack.refresh(), len(stack.layers), _p(x), _p(x_mid), _p(y), _p(qkv), _p(att), _p(f), B, g, nH, hd, I, float(eps),
              ws, wsb, mk, _zeros(dev).data_ptr(), _stream())
