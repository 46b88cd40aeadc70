"""torch.autograd.Function wrappers that give the HIP forward ops a HIP backward.

The inference path calls `ops.*` directly and never builds a graph.  When gradients are enabled and a parameter requires them
(reference: `train_ullava.py:207-261`, `train_ullava_core.py`), the models route through these functions instead: every forward is the
same HIP kernel as in inference (un-fused where the backward needs an intermediate: the SwiGLU pre-activations), every backward is a
HIP kernel from csrc/backward.hip or the forward GEMM on transposed operands.  torch supplies only the tape, the tensor plumbing
(cat / transpose / pad / where, which are data movement) and `.grad` accumulation.
"""
from typing import Optional

import weakref

import torch

from . import ops


def _t(x: torch.Tensor) -> torch.Tensor:
    """[R, C] -> contiguous [C, R] (data movement for the transposed-operand GEMMs of Linear backward)."""
    if x.stride(-1) != 1:
        x = x.contiguous()
    return ops.transpose2d(x)


_T_CACHE: dict = {}           # id(weight) -> (weakref(weight), version, W^T): only the very same tensor object may hit


def _t_frozen(w: torch.Tensor, tag=None) -> torch.Tensor:
    """Transposed copy of a FROZEN weight (dX = dY @ W needs W^T as the GEMM's row-major operand), kept across steps: the reference's
    stage-2 recipe freezes every LLaMA Linear but the LoRA targets, so the 13 GB of transposes are paid once, not per backward.
    Kept per tensor OBJECT (weak reference + version counter), never per address: the allocator hands a freed temporary's storage
    to the next one, so an address says nothing about the contents.  Callers therefore pass persistent tensors (parameters, or the
    shared q|k|v / gate|up buffers of modeling_core._alias_pack)."""
    if w.requires_grad or w.grad_fn is not None:
        return _t(w)
    ver = (w._version, tag)          # tag: version counters of parameters that alias `w`'s storage (their updates do not bump w._version)
    hit = _T_CACHE.get(id(w))
    if hit is not None and hit[0]() is w and hit[1] == ver:
        return hit[2]
    wt = _t(w)
    for k in [k for k, v in _T_CACHE.items() if v[0]() is None]:
        del _T_CACHE[k]
    _T_CACHE[id(w)] = (weakref.ref(w), ver, wt)
    return wt


def clear_transpose_cache():
    _T_CACHE.clear()


class _Linear(torch.autograd.Function):
    """y = x @ w.T (+ bias) (relu) (+ residual).  dX = dY @ W, dW = dY^T @ X, db = colsum(dY): the forward GEMM on transposed operands."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, relu, bias_after_rounding):
        if relu and residual is not None:
            raise NotImplementedError("relu + residual in one epilogue hides the pre-residual sign needed by the backward")
        y = ops.linear(x, w, bias, act="relu" if relu else None, residual=residual, bias_after_rounding=bias_after_rounding)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias, ctx.has_res, ctx.relu = bias is not None, residual is not None, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        g = ops.relu_mask(y, dy) if ctx.relu else dy                                   # selection (HIP kernel)
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(g2, _t_frozen(w)).view(x.shape)                              # [M, N] x [K, N]^T
        if ctx.needs_input_grad[1]:
            dw = ops.linear(_t(g2), _t(x2))                                              # [N, M] x [K, M]^T
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(g2).to(w.dtype)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None


def linear(x, w, bias=None, residual=None, relu: bool = False, bias_after_rounding: bool = False):
    return _Linear.apply(x, w, bias, residual, relu, bias_after_rounding)


class _LinearPacked(torch.autograd.Function):
    """y = x @ packed.T where `packed` [sum(N_i), K] is ONE buffer whose row slices ARE the parameters ws (q|k|v, gate|up: storage shared,
    see modeling_core._alias_for_training).  One forward GEMM, one dX GEMM, one dW GEMM for the whole pack; the parameters' gradients are
    row slices (views) of that one dW -- no torch.cat of the weights per step and no split / copy of the gradient."""

    @staticmethod
    def forward(ctx, x, packed, *ws):
        y = ops.linear(x, packed)
        ctx.save_for_backward(x, packed)
        ctx.rows = [w.shape[0] for w in ws]
        ctx.frozen = not any(w.requires_grad for w in ws)
        ctx.wver = tuple(w._version for w in ws)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, packed = ctx.saved_tensors
        g2, x2 = dy.contiguous().reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(g2, _t_frozen(packed, ctx.wver) if ctx.frozen else _t(packed)).view(x.shape)
        dws = [None] * len(ctx.rows)
        if any(ctx.needs_input_grad[2:]):
            dwp = ops.linear(_t(g2), _t(x2))                                             # [sum(N_i), K]
            o = 0
            for i, n in enumerate(ctx.rows):
                if ctx.needs_input_grad[2 + i]:
                    dws[i] = dwp[o:o + n]
                o += n
        return (dx, None, *dws)


def linear_packed(x, packed, *ws):
    return _LinearPacked.apply(x, packed, *ws)


class _Fork(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (the residual stream: the norm and the residual add).  Autograd would sum the two
    gradients with a torch add; here the sum is the HIP add (rnd(a + b), the same 16-bit arithmetic)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, d1, d2):
        if d1 is None:
            return d2
        if d2 is None:
            return d1
        return ops.add_rows(d1.contiguous(), d2.contiguous())


def fork(x):
    return _Fork.apply(x)


class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return ops.rmsnorm(x, w, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw = ops.rmsnorm_bwd(x, w, dy.contiguous(), ctx.eps, need_dw=ctx.needs_input_grad[1])
        return dx, (dw.to(w.dtype) if dw is not None else None), None


def rmsnorm(x, w, eps):
    return _RMSNorm.apply(x, w, eps)


class _SwiGLU(torch.autograd.Function):
    """gu [M, 2I] (interleaved gate/up columns, or with halves [gate | up]) -> silu(gate) * up."""

    @staticmethod
    def forward(ctx, gu, halves):
        ctx.save_for_backward(gu)
        ctx.halves = halves
        return ops.swiglu_fwd(gu, halves)

    @staticmethod
    def backward(ctx, da):
        (gu,) = ctx.saved_tensors
        return ops.swiglu_bwd(gu, da.contiguous(), ctx.halves), None


def swiglu(gu, halves: bool = False):
    return _SwiGLU.apply(gu, bool(halves))


class _Rope(torch.autograd.Function):
    """apply_rotary_pos_emb on the first n_heads heads of every row of a fused [T, 3D] q|k|v buffer (functional: returns a new tensor)."""

    @staticmethod
    def forward(ctx, qkv, positions, inv_freq, n_heads, hd):
        out = qkv.clone()
        T = out.shape[0]
        ops.rope_inplace(out, out.stride(0), positions, inv_freq, T, n_heads, hd)
        ctx.save_for_backward(positions, inv_freq)
        ctx.n_heads, ctx.hd = n_heads, hd
        return out

    @staticmethod
    def backward(ctx, dout):
        positions, inv_freq = ctx.saved_tensors
        d = dout.contiguous().clone()
        ops.rope_bwd_inplace(d, d.stride(0), positions, inv_freq, d.shape[0], ctx.n_heads, ctx.hd)
        return d, None, None, None, None


def rope(qkv, positions, inv_freq, n_heads, hd):
    return _Rope.apply(qkv, positions, inv_freq, n_heads, hd)


class _SelfAttention(torch.autograd.Function):
    """Causal / masked self attention on a fused, already rotated [B*S, 3D] q|k|v buffer (hf eager_attention_forward: scores * hd^-0.5)."""

    @staticmethod
    def forward(ctx, qkv, key_mask, B, S, H, hd, causal):
        D = H * hd
        T = B * S
        att = torch.empty(T, D, device=qkv.device, dtype=qkv.dtype)
        st = (S * 3 * D, hd, 3 * D)                 # V as rows of the q|k|v buffer (the wrapper makes the V^T image where no kernel takes rows)
        ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], att, B, H, S, S, hd, st, st, (S * D, hd, D), key_mask, causal=causal, scale_mode=1,
                      scale=hd ** -0.5, v_strides=st)
        ctx.save_for_backward(qkv, att, key_mask)
        ctx.dims = (B, S, H, hd, causal)
        return att

    @staticmethod
    def backward(ctx, datt):
        qkv, att, key_mask = ctx.saved_tensors
        B, S, H, hd, causal = ctx.dims
        D = H * hd
        datt = datt.contiguous()
        dqkv = torch.empty_like(qkv)
        s3, s1 = (S * 3 * D, hd, 3 * D), (S * D, hd, D)
        ops.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], att, datt, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], (s3, s3, s3, s1, s1, s3, s3, s3), key_mask,
                          B, H, S, S, hd, causal, hd ** -0.5)
        return dqkv, None, None, None, None, None, None


def self_attention(qkv, key_mask, B, S, H, hd, causal=True):
    return _SelfAttention.apply(qkv, key_mask, B, S, H, hd, causal)


class _ShiftedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        stats = ops.shifted_cross_entropy_stats(logits, labels)
        ctx.save_for_backward(logits, labels, stats)
        return (stats[0] / stats[1]).to(logits.dtype)

    @staticmethod
    def backward(ctx, g):
        logits, labels, stats = ctx.saved_tensors
        return ops.shifted_cross_entropy_bwd(logits, labels, stats, g.float().reshape(1).contiguous()), None


def shifted_cross_entropy(logits, labels):
    return _ShiftedCE.apply(logits, labels)


class _EmbedSplice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, img_feat, vid_feat, ids, spans, img_tokens, img_pitch, img_off, detach_text):
        ctx.save_for_backward(ids, spans)
        ctx.meta = (table.shape[0], None if img_feat is None else tuple(img_feat.shape), None if vid_feat is None else tuple(vid_feat.shape),
                    img_tokens, img_pitch, img_off, table.dtype, detach_text)
        return ops.embed_splice(ids, table, img_feat, vid_feat, spans, img_tokens, img_pitch, img_off)

    @staticmethod
    def backward(ctx, demb):
        ids, spans = ctx.saved_tensors
        vocab, img_shape, vid_shape, img_tokens, img_pitch, img_off, dt, detach_text = ctx.meta
        # span lengths travel even when the features need no gradient (frozen projector): span rows never reach the table
        d_table, d_img, d_vid = ops.embed_splice_bwd(ids, demb.contiguous(), vocab, img_shape if ctx.needs_input_grad[1] else None,
                                                     vid_shape if ctx.needs_input_grad[2] else None, spans, img_tokens, img_pitch, img_off,
                                                     need_table=ctx.needs_input_grad[0], vid_tokens=vid_shape[-2] if vid_shape is not None else 0,
                                                     detach_text=detach_text)
        return (d_table.to(dt) if d_table is not None else None), d_img, d_vid, None, None, None, None, None, None


def embed_splice(table, img_feat, vid_feat, ids, spans, img_tokens=0, img_pitch=0, img_off=0, detach_text=False):
    """detach_text: reference `projector_from_scratch` (ullava_core.py:230-240): text rows of image / video samples are detached,
    except the start / end token rows."""
    return _EmbedSplice.apply(table, img_feat, vid_feat, ids, spans, img_tokens, img_pitch, img_off, bool(detach_text))


# ---- SAM mask decoder / postprocess / losses (training of `mask_decoder`, seg / det heads: train_ullava.py:248-261) ------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return ops.layernorm(x, w, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = ops.layernorm_bwd(x, w, dy.contiguous(), ctx.eps, need_wb=ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, (dw.to(w.dtype) if dw is not None else None), (db.to(w.dtype) if db is not None else None), None


def layernorm(x, w, b, eps):
    return _LayerNorm.apply(x, w, b, eps)


class _LayerNorm2dCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, gelu):
        ctx.save_for_backward(x, w, b)
        ctx.eps, ctx.gelu = eps, gelu
        return ops.layernorm2d_cl(x, w, b, eps, gelu)

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        dx, dw, db = ops.layernorm2d_cl_bwd(x, w, b, dy, ctx.eps, ctx.gelu)
        return dx, dw.to(w.dtype), db.to(w.dtype), None, None


def layernorm2d_cl(x, w, b, eps=1e-6, gelu=False):
    return _LayerNorm2dCL.apply(x, w, b, eps, gelu)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x, dy)


def gelu(x):
    return _Gelu.apply(x)


class _Dropout(torch.autograd.Function):
    """nn.Dropout(p) in training mode; the keep mask is drawn from torch's generator of the device (the reference's RNG), the scaling
    and the zeroing run in the HIP kernel.  p == 0 never gets here."""

    @staticmethod
    def forward(ctx, x, p):
        keep = (torch.rand(x.shape, device=x.device) >= p).to(torch.uint8)
        ctx.save_for_backward(keep)
        ctx.p = p
        return ops.dropout_apply(x, keep, p)

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        return ops.dropout_apply(dy.contiguous(), keep, ctx.p), None


def dropout(x, p: float):
    if p <= 0.0:
        return x
    if p >= 1.0:
        raise ValueError("dropout probability must be < 1")
    return _Dropout.apply(x, p)


class _Add(torch.autograd.Function):
    """rnd(a + b[row % b_rows]) (ops.add_rows); gradients pass through (b's only when it is not broadcast)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.same = a.shape == b.shape
        return ops.add_rows(a, b)

    @staticmethod
    def backward(ctx, d):
        if ctx.needs_input_grad[1] and not ctx.same:
            raise NotImplementedError("gradient of a broadcast operand of add_rows")
        return d, (d if ctx.needs_input_grad[1] else None)


def add(a, b):
    return _Add.apply(a, b)


class _Attention(torch.autograd.Function):
    """SAM decoder attention (transformer.py:220-242): separate projected q [n*Sq, Di], k / v [n*Sk, Di], softmax(q k^T / sqrt(hd)) v."""

    @staticmethod
    def forward(ctx, q, k, v, n, H, Sq, Sk):
        import math
        Di = q.shape[-1]
        hd = Di // H
        vt = ops.transpose_v(v, Sk * Di, Di, n, Sk, H, hd)
        att = torch.empty(n * Sq, Di, device=q.device, dtype=q.dtype)
        ops.attention(q, k, vt, att, n, H, Sq, Sk, hd, (Sq * Di, hd, Di), (Sk * Di, hd, Di), (Sq * Di, hd, Di), None, causal=False, scale_mode=2,
                      scale=math.sqrt(hd))
        ctx.save_for_backward(q, k, v, att)
        ctx.dims = (n, H, Sq, Sk, hd)
        return att

    @staticmethod
    def backward(ctx, datt):
        q, k, v, att = ctx.saved_tensors
        n, H, Sq, Sk, hd = ctx.dims
        Di = H * hd
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        sq, sk = (Sq * Di, hd, Di), (Sk * Di, hd, Di)
        ops.attention_bwd(q, k, v, att, datt.contiguous(), dq, dk, dv, (sq, sk, sk, sq, sq, sq, sk, sk), None, n, H, Sq, Sk, hd, False, hd ** -0.5)
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, n, H, Sq, Sk):
    return _Attention.apply(q, k, v, n, H, Sq, Sk)


class _MaskMatmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hyper, up, n, T, C, G):
        ctx.save_for_backward(hyper, up)
        ctx.dims = (n, T, C, G)
        return ops.mask_matmul(hyper, up, n, T, C, G)

    @staticmethod
    def backward(ctx, dm):
        hyper, up = ctx.saved_tensors
        n, T, C, G = ctx.dims
        dh, dup = ops.mask_matmul_bwd(hyper, up, dm, n, T, C, G)
        return dh.to(hyper.dtype), dup, None, None, None, None


def mask_matmul(hyper, up, n, T, C, G):
    return _MaskMatmul.apply(hyper, up, n, T, C, G)


class _Bilinear(torch.autograd.Function):
    """ops.bilinear (fp32 output) with its adjoint; the gradient is returned in the input's dtype."""

    @staticmethod
    def forward(ctx, x, in_h, in_w, out_h, out_w):
        ctx.meta = (tuple(x.shape[-2:]), in_h, in_w, x.dtype)
        return ops.bilinear(x, in_h, in_w, out_h, out_w)

    @staticmethod
    def backward(ctx, dout):
        full, in_h, in_w, dt = ctx.meta
        return ops.bilinear_bwd(dout.contiguous(), full, in_h, in_w).to(dt), None, None, None, None


def bilinear(x, in_h, in_w, out_h, out_w):
    return _Bilinear.apply(x, in_h, in_w, out_h, out_w)


class _MaskLossSums(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, scale):
        ctx.save_for_backward(logits, target)
        ctx.scale = scale
        return ops.mask_loss_sums(logits, target, scale)

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        return ops.mask_loss_sums_bwd(logits, target, g.float().contiguous(), ctx.scale), None, None


def mask_loss_sums(logits, target, scale=1000.0):
    return _MaskLossSums.apply(logits, target, scale)


class _BoxLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        ctx.save_for_backward(pred, gt)
        return ops.box_losses(pred, gt)

    @staticmethod
    def backward(ctx, g):
        pred, gt = ctx.saved_tensors
        return ops.box_losses_bwd(pred, gt, g.float().contiguous()).to(pred.dtype), None


def box_losses(pred, gt):
    return _BoxLosses.apply(pred, gt)
