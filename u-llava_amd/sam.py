"""SAM (ViT-H image encoder, prompt encoder, two-way mask decoder, postprocess) on the HIP path.

Host-side mirror of reference `models/segment_anything/modeling/{image_encoder,prompt_encoder,mask_decoder,transformer,
common,sam}.py` + `build_sam.py`: the module tree reproduces the reference's state-dict key names; the arithmetic is
HIP kernels (ops.py).  Internal activation layout is token-major / channels-last ([tokens, C] rows) everywhere, so
every conv on the path is a GEMM: patch embed (im2col), neck 1x1 (plain), neck 3x3 (9-tap im2col), the two
ConvTranspose2d(k=2,s=2) of the mask decoder (GEMM whose 4 column blocks are the 2x2 output pixels -- the result is kept
in that blocked order and only un-blocked by the final mask product kernel).
"""
import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import autograd_ops as A
from . import ops
from .configuration import SamConfig
from .modeling_core import BF16, Conv2dHolder, Embedding, Linear, Norm, _Holder, _param


def _attn_holder(dim, internal, device, dtype):
    m = _Holder()
    m.q_proj = Linear(dim, internal, device=device, dtype=dtype)
    m.k_proj = Linear(dim, internal, device=device, dtype=dtype)
    m.v_proj = Linear(dim, internal, device=device, dtype=dtype)
    m.out_proj = Linear(internal, dim, device=device, dtype=dtype)
    return m


def _mlp_holder(i, h, o, n, device, dtype):
    m = _Holder()
    dims = [i] + [h] * (n - 1) + [o]
    m.layers = nn.ModuleList([Linear(a, b, device=device, dtype=dtype) for a, b in zip(dims[:-1], dims[1:])])
    return m


def build_sam_holder(cfg: SamConfig, device=None, dtype=BF16) -> nn.Module:
    """Parameter tree with the key names of reference build_sam._build_sam (build_sam.py:56-102)."""
    sam = _Holder()
    C, D = cfg.embed_dim, cfg.out_chans
    g = cfg.img_size // cfg.patch_size
    enc = _Holder()
    enc.patch_embed = _Holder()
    enc.patch_embed.proj = Conv2dHolder(3, C, cfg.patch_size, bias=True, device=device, dtype=dtype)
    enc.pos_embed = _param(1, g, g, C, device=device, dtype=dtype)
    blocks = []
    for i in range(cfg.depth):
        b = _Holder()
        size = g if i in cfg.global_attn_indexes else cfg.window_size
        b.norm1 = Norm(C, eps=1e-6, device=device, dtype=dtype)
        b.attn = _Holder()
        b.attn.qkv = Linear(C, 3 * C, device=device, dtype=dtype)
        b.attn.proj = Linear(C, C, device=device, dtype=dtype)
        b.attn.rel_pos_h = _param(2 * size - 1, C // cfg.num_heads, device=device, dtype=dtype)
        b.attn.rel_pos_w = _param(2 * size - 1, C // cfg.num_heads, device=device, dtype=dtype)
        b.norm2 = Norm(C, eps=1e-6, device=device, dtype=dtype)
        b.mlp = _Holder()
        b.mlp.lin1 = Linear(C, int(C * cfg.mlp_ratio), device=device, dtype=dtype)
        b.mlp.lin2 = Linear(int(C * cfg.mlp_ratio), C, device=device, dtype=dtype)
        blocks.append(b)
    enc.blocks = nn.ModuleList(blocks)
    enc.neck = nn.ModuleList([Conv2dHolder(C, D, 1, bias=False, device=device, dtype=dtype), Norm(D, eps=1e-6, device=device, dtype=dtype),
                              Conv2dHolder(D, D, 3, bias=False, device=device, dtype=dtype), Norm(D, eps=1e-6, device=device, dtype=dtype)])
    sam.image_encoder = enc

    pe = _Holder()
    pe.pe_layer = _Holder()
    pe.pe_layer.register_buffer("positional_encoding_gaussian_matrix", torch.empty(2, D // 2, device=device, dtype=dtype))
    pe.point_embeddings = nn.ModuleList([Embedding(1, D, device=device, dtype=dtype) for _ in range(4)])
    pe.not_a_point_embed = Embedding(1, D, device=device, dtype=dtype)
    mic = cfg.mask_in_chans
    pe.mask_downscaling = nn.ModuleList([Conv2dHolder(1, mic // 4, 2, True, device, dtype), Norm(mic // 4, eps=1e-6, device=device, dtype=dtype),
                                         nn.Identity(), Conv2dHolder(mic // 4, mic, 2, True, device, dtype),
                                         Norm(mic, eps=1e-6, device=device, dtype=dtype), nn.Identity(),
                                         Conv2dHolder(mic, D, 1, True, device, dtype)])
    pe.no_mask_embed = Embedding(1, D, device=device, dtype=dtype)
    sam.prompt_encoder = pe

    md = _Holder()
    tr = _Holder()
    layers = []
    for _ in range(cfg.decoder_depth):
        l = _Holder()
        l.self_attn = _attn_holder(D, D, device, dtype)
        l.norm1 = Norm(D, device=device, dtype=dtype)
        l.cross_attn_token_to_image = _attn_holder(D, D // 2, device, dtype)
        l.norm2 = Norm(D, device=device, dtype=dtype)
        l.mlp = _Holder()
        l.mlp.lin1 = Linear(D, cfg.decoder_mlp_dim, device=device, dtype=dtype)
        l.mlp.lin2 = Linear(cfg.decoder_mlp_dim, D, device=device, dtype=dtype)
        l.norm3 = Norm(D, device=device, dtype=dtype)
        l.norm4 = Norm(D, device=device, dtype=dtype)
        l.cross_attn_image_to_token = _attn_holder(D, D // 2, device, dtype)
        layers.append(l)
    tr.layers = nn.ModuleList(layers)
    tr.final_attn_token_to_image = _attn_holder(D, D // 2, device, dtype)
    tr.norm_final_attn = Norm(D, device=device, dtype=dtype)
    md.transformer = tr
    nm = cfg.num_multimask_outputs + 1
    md.iou_token = Embedding(1, D, device=device, dtype=dtype)
    md.mask_tokens = Embedding(nm, D, device=device, dtype=dtype)
    up0, up3 = _Holder(), _Holder()           # ConvTranspose2d weights are [in, out, kh, kw]
    up0.weight, up0.bias = _param(D, D // 4, 2, 2, device=device, dtype=dtype), _param(D // 4, device=device, dtype=dtype)
    up3.weight, up3.bias = _param(D // 4, D // 8, 2, 2, device=device, dtype=dtype), _param(D // 8, device=device, dtype=dtype)
    md.output_upscaling = nn.ModuleList([up0, Norm(D // 4, eps=1e-6, device=device, dtype=dtype), nn.Identity(), up3, nn.Identity()])
    md.output_hypernetworks_mlps = nn.ModuleList([_mlp_holder(D, D, D // 8, 3, device, dtype) for _ in range(nm)])
    md.iou_prediction_head = _mlp_holder(D, cfg.iou_head_hidden_dim, nm, cfg.iou_head_depth, device, dtype)
    sam.mask_decoder = md
    return sam


class SamEngine:
    """Runs the SAM sub-models of a `build_sam_holder` tree on the HIP path (weights re-laid out once)."""

    def __init__(self, sam: nn.Module, cfg: SamConfig):
        self.sam, self.cfg = sam, cfg
        self._pk = None
        self._dense_pe = None

    def invalidate(self):
        self._pk = None
        self._dense_pe = None

    # -- weight re-layout ------------------------------------------------------------------------------------------
    def pack(self):
        cfg, enc, md = self.cfg, self.sam.image_encoder, self.sam.mask_decoder
        C, D = cfg.embed_dim, cfg.out_chans
        pk = {}
        w = enc.patch_embed.proj.weight
        pk["patch_w"] = w.reshape(C, -1).contiguous()                        # K = 3*16*16 = 768 (multiple of 64)
        pk["patch_wp"] = ops.pack_patch_weight(w) if (cfg.patch_size <= 16 and cfg.patch_size % 2 == 0 and w.dtype != torch.float32) else None
        pk["pos"] = enc.pos_embed.reshape(-1, C).contiguous()
        pk["neck0"] = enc.neck[0].weight.reshape(D, C).contiguous()
        pk["neck2"] = enc.neck[2].weight.permute(0, 2, 3, 1).reshape(D, 9 * D).contiguous()   # (ky,kx,ci) columns
        u0, u3 = md.output_upscaling[0], md.output_upscaling[3]
        # ConvTranspose2d(k=2,s=2) as a GEMM: row (dy*2+dx)*Cout + co of W^T holds W[:, co, dy, dx]
        pk["up0_w"] = u0.weight.permute(2, 3, 1, 0).reshape(-1, u0.weight.shape[0]).contiguous()
        pk["up0_b"] = u0.bias.repeat(4).contiguous()
        pk["up3_w"] = u3.weight.permute(2, 3, 1, 0).reshape(-1, u3.weight.shape[0]).contiguous()
        pk["up3_b"] = u3.bias.repeat(4).contiguous()
        # tile-major copies of the encoder's big Linears for the prefill-shape GEMM (see ops.register_tiled)
        for blk in enc.blocks:
            for t in (blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.lin1.weight, blk.mlp.lin2.weight):
                ops.register_tiled(t)
        ops.register_tiled(pk["patch_w"])
        self._pk = pk
        return pk

    def pk(self):
        return self._pk if self._pk is not None else self.pack()

    # -- image encoder (image_encoder.py:110-125) ------------------------------------------------------------------
    def encode(self, pixel_values: torch.Tensor, trace: Optional[dict] = None) -> torch.Tensor:
        """[B,3,S,S] -> token-major embeddings [B, g*g, out_chans].  trace: optional dict receiving the stage outputs [B, g, g, C]."""
        cfg, enc, pk = self.cfg, self.sam.image_encoder, self.pk()
        B = pixel_values.shape[0]
        C, nH = cfg.embed_dim, cfg.num_heads
        hd = C // nH
        g = cfg.img_size // cfg.patch_size
        if pk["patch_wp"] is not None:                                       # fused: A tiles DMA'd straight from the pixels
            x = ops.patchify(pixel_values.to(enc.pos_embed.dtype).contiguous(), pk["patch_wp"], cfg.patch_size, enc.patch_embed.proj.bias)
        else:
            cols = ops.im2col(pixel_values.to(enc.pos_embed.dtype).contiguous(), cfg.patch_size, pk["patch_w"].shape[1])
            x = ops.linear(cols, pk["patch_w"], enc.patch_embed.proj.bias)       # [B*g*g, C]
        x = ops.add_rows(x, pk["pos"])
        if trace is not None:
            trace["embed"] = x.view(B, g, g, C)
        I = enc.blocks[0].mlp.lin1.weight.shape[0] if len(enc.blocks) else 0
        coarse = (trace is None and ops.coarse_ok() and len(enc.blocks) > 0 and hd == 80 and g == 64 and cfg.window_size == 14 and C % 64 == 0
                  and I % 64 == 0 and all(b.attn.qkv.bias is not None for b in enc.blocks) and x.dtype != torch.float32)
        if coarse:
            # one C call for all blocks (csrc/layers.hip): the same launches as the loop below, bit-identical results
            stack = pk.get("_c_blocks")
            if stack is None:
                def rel(t, glob):
                    return ops.fit_rel_pos(t, g if glob else cfg.window_size)
                stack = pk["_c_blocks"] = ops.LayerStack(_lib.SamBlock, [dict(
                    n1_w=b.norm1.weight, n1_b=b.norm1.bias, n2_w=b.norm2.weight, n2_b=b.norm2.bias, qkv=(b.attn.qkv.weight, b.attn.qkv.bias),
                    proj=(b.attn.proj.weight, b.attn.proj.bias), lin1=(b.mlp.lin1.weight, b.mlp.lin1.bias), lin2=(b.mlp.lin2.weight, b.mlp.lin2.bias),
                    rel_pos_h=rel(b.attn.rel_pos_h, i in cfg.global_attn_indexes), rel_pos_w=rel(b.attn.rel_pos_w, i in cfg.global_attn_indexes),
                    window=0 if i in cfg.global_attn_indexes else cfg.window_size) for i, b in enumerate(enc.blocks)])
            ops.sam_blocks(stack, x, B, g, nH, hd, I, 1e-6)
        for i, blk in enumerate(() if coarse else enc.blocks):
            glob = i in cfg.global_attn_indexes
            ws = 0 if glob else cfg.window_size
            y = ops.layernorm(x, blk.norm1.weight, blk.norm1.bias, 1e-6)
            if ws == 14 and hd == 80 and blk.attn.qkv.bias is not None and x.dtype != torch.float32:
                # the path's own window shape: the tokens stay in image order, the attention kernel does the window addressing and
                # takes the q|k|v of the reference's zero-padded positions from the qkv bias -- no partition / unpartition passes
                # and no GEMM rows for the padding (25 windows x 196 = 4900 positions for 4096 tokens), and reads V through the
                # transposing LDS load, so there is no V^T pass either
                qkv = ops.linear(y, blk.attn.qkv.weight, blk.attn.qkv.bias)
                att = ops.sam_window_attention(qkv, blk.attn.qkv.bias, blk.attn.rel_pos_h, blk.attn.rel_pos_w, B, g, g, nH, hd, ws)
                x = ops.linear(att, blk.attn.proj.weight, blk.attn.proj.bias, residual=x)
            else:
                if ws:
                    y = ops.window_partition(y, B, g, g, ws)
                    side = ws
                    NB = B * ((g + ws - 1) // ws) ** 2
                else:
                    side, NB = g, B
                S = side * side
                qkv = ops.linear(y, blk.attn.qkv.weight, blk.attn.qkv.bias)          # [NB*S, 3C]: q | k | v, heads contiguous
                strides = (S * 3 * C, hd, 3 * C)
                att = torch.empty(NB * S, C, device=x.device, dtype=x.dtype)
                if glob and side != 64:
                    # other grid sizes: per-query bias tables from their own kernel, looked up by the attention kernel
                    rel_h, rel_w = ops.sam_relpos(qkv, strides, blk.attn.rel_pos_h, blk.attn.rel_pos_w, NB, nH, side, side, hd)
                    ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False,
                                  scale_mode=0, q_scale=hd ** -0.5, rel_h=rel_h, rel_w=rel_w, v_strides=strides)
                else:
                    # the attention kernels build the rel-pos bias themselves from the raw rel_pos_h / rel_pos_w parameters (Toeplitz
                    # product on the MFMA); V goes in as rows of the q|k|v buffer (the 64 x 64 global kernel reads it through the
                    # transposing LDS load, other shapes get their V^T image made by the wrapper)
                    ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False,
                                  scale_mode=0, q_scale=hd ** -0.5, rel_h=blk.attn.rel_pos_h, rel_w=blk.attn.rel_pos_w,
                                  rel_pos_hw=(side, side), v_strides=strides)
                if ws:
                    o = ops.linear(att, blk.attn.proj.weight, blk.attn.proj.bias)
                    x = ops.window_unpartition_add(o, x, B, g, g, ws)
                else:
                    x = ops.linear(att, blk.attn.proj.weight, blk.attn.proj.bias, residual=x)
            if trace is not None:
                trace[f"block{i}.attn"] = x.view(B, g, g, C)
            y = ops.layernorm(x, blk.norm2.weight, blk.norm2.bias, 1e-6)
            f = ops.linear(y, blk.mlp.lin1.weight, blk.mlp.lin1.bias, act="gelu")
            x = ops.linear(f, blk.mlp.lin2.weight, blk.mlp.lin2.bias, residual=x)
            if trace is not None:
                trace[f"block{i}"] = x.view(B, g, g, C)
        if x.dtype == torch.float16:
            # image_encoder.py:117-124: the fp16 model's neck runs in fp32 ("prevent overflow") and only its result is cast back.
            # 1x1 conv: fp16 MFMA products are exact in fp32, fp32 accumulators written out unrounded; LayerNorm2d in fp32; the 3x3
            # conv sees its fp32 input as a two-term fp16 split (hi + 2^-11 lo, 22 significand bits) stacked as 2B images: one GEMM,
            # fp32 out, the halves recombined by the last LayerNorm2d, which also does the `.to(float16)`.
            T = x.shape[0]
            x32 = ops.linear(x, pk["neck0"], out_f32=True)
            hl = ops.neck_layernorm2d_f32(x32, None, 0.0, enc.neck[1].weight, enc.neck[1].bias, 1e-6, split=True)
            o = ops.linear(ops.im2col3x3(hl.view(2 * T, -1), 2 * B, g, g), pk["neck2"], out_f32=True)
            x = ops.neck_layernorm2d_f32(o[:T], o[T:], 2.0 ** -11, enc.neck[3].weight, enc.neck[3].bias, 1e-6, split=False)
            return x.view(B, g * g, cfg.out_chans)
        x = ops.linear(x, pk["neck0"])
        x = ops.layernorm2d_cl(x, enc.neck[1].weight, enc.neck[1].bias, 1e-6)
        x = ops.linear(ops.im2col3x3(x, B, g, g), pk["neck2"])
        x = ops.layernorm2d_cl(x, enc.neck[3].weight, enc.neck[3].bias, 1e-6)
        return x.view(B, g * g, cfg.out_chans)

    # -- prompt encoder (prompt_encoder.py:67-76,216-229) ----------------------------------------------------------
    def dense_pe(self) -> torch.Tensor:
        """Token-major [g*g, D] dense positional encoding.  A constant of the weights: evaluated once on the host with the
        reference's exact bf16 op sequence (the Gaussian buffer is cast with the model, so the PE is computed in bf16)."""
        if self._dense_pe is None:
            gm = self.sam.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix
            g = self.cfg.img_size // self.cfg.patch_size
            m = gm.detach().cpu()
            grid = torch.ones((g, g), dtype=m.dtype)
            y = (grid.cumsum(dim=0) - 0.5) / g
            x = (grid.cumsum(dim=1) - 0.5) / g
            c = torch.stack([x, y], dim=-1)
            c = 2 * c - 1
            c = c @ m
            c = 2 * math.pi * c
            pe = torch.cat([torch.sin(c), torch.cos(c)], dim=-1)           # [g, g, D]
            self._dense_pe = pe.reshape(g * g, -1).contiguous().to(gm.device)
        return self._dense_pe

    # -- mask decoder (mask_decoder.py:75-164, transformer.py) -----------------------------------------------------
    def _attn(self, a, q_in, k_in, v_in, n, Sq, Sk, residual=None, trace=None, tag="", unfused_bias=()):
        """transformer.py:220-242.  unfused_bias: which of "q", "k", "v" projections see a NON-contiguous 3-D input in the reference
        (the image keys while they are still the permuted NCHW view, i.e. in layer 0): at::linear then runs matmul + add_ and the
        bias is added to the already rounded product."""
        H = self.cfg.decoder_heads
        Di = a.q_proj.weight.shape[0]
        hd = Di // H
        q = ops.linear(q_in, a.q_proj.weight, a.q_proj.bias, bias_after_rounding="q" in unfused_bias)
        k = ops.linear(k_in, a.k_proj.weight, a.k_proj.bias, bias_after_rounding="k" in unfused_bias)
        v = ops.linear(v_in, a.v_proj.weight, a.v_proj.bias, bias_after_rounding="v" in unfused_bias)
        vt = ops.transpose_v(v, Sk * Di, Di, n, Sk, H, hd)
        att = torch.empty(n * Sq, Di, device=q.device, dtype=q.dtype)
        ops.attention(q, k, vt, att, n, H, Sq, Sk, hd, (Sq * Di, hd, Di), (Sk * Di, hd, Di), (Sq * Di, hd, Di), None, causal=False,
                      scale_mode=2, scale=math.sqrt(hd))
        if trace is not None:
            trace.update({tag + ".q": q, tag + ".k": k, tag + ".v": v, tag + ".att": att})
        return ops.linear(att, a.out_proj.weight, a.out_proj.bias, residual=residual)

    # -- fused two-way transformer (csrc/sam_decoder.hip) ------------------------------------------------------------------------
    fused_decoder = True        # False: the op-by-op chain below (what `trace=` and the training path use)

    def _fusable(self, D, P, T, nm) -> bool:
        c = self.cfg
        return D == 256 and c.decoder_heads == 8 and c.decoder_mlp_dim % 256 == 0 and P in (4096, 1024) and T <= 8 and nm == 4 \
            and self.sam.mask_decoder.transformer.layers[0].cross_attn_token_to_image.q_proj.weight.shape[0] == 128

    def _decode_fused(self, image_embedding_tm, tokens, image_index, n, D, P, g, nm, T):
        """decode() on the fused kernels of csrc/sam_decoder.hip: 8 launches per TwoWayAttentionBlock (self-attention heads, out+LN,
        k/v tile projections + scores, softmax + PV, out+LN, MLP partials, reduce+LN, image->token attention in one launch) -- 27 per
        decode instead of ~85, key tiles LDS-resident, weights streamed through LDS once per block.  Same rounding points as the
        op-by-op chain."""
        md, tr, pk = self.sam.mask_decoder, self.sam.mask_decoder.transformer, self.pk()
        src = ops.add_rows(image_embedding_tm.reshape(-1, D), self.sam.prompt_encoder.no_mask_embed.weight)
        if image_index is None:
            keys = src.view(1, P, D).expand(n, -1, -1).contiguous()
        else:
            keys = ops.gather_rows(src.view(-1, P * D), image_index).view(n, P, D)
        pos = self.dense_pe()
        queries = tokens
        nl = len(tr.layers)
        for i, l in enumerate(tr.layers):
            att = ops.sam_self_attn_heads(queries, tokens, l.self_attn, first=(i == 0))
            queries, (q_t2i,) = ops.sam_out_ln(att, None if i == 0 else queries, tokens, l.self_attn.out_proj, l.norm1,
                                               projs=[(l.cross_attn_token_to_image.q_proj, True)])
            # layer 0: the image keys are still non-contiguous views in the reference -> at::linear's unfused-bias path (see _attn)
            att = ops.sam_t2i_attention(q_t2i, keys, pos, l.cross_attn_token_to_image, late_bias_kv=(i == 0))
            queries, _ = ops.sam_out_ln(att, queries, tokens, l.cross_attn_token_to_image.out_proj, l.norm2)
            projs = [(l.cross_attn_image_to_token.k_proj, True), (l.cross_attn_image_to_token.v_proj, False)]
            if i == nl - 1:
                projs.append((tr.final_attn_token_to_image.q_proj, True))
            queries, pr = ops.sam_token_mlp_ln(queries, tokens, l.mlp.lin1, l.mlp.lin2, l.norm3, projs=projs)
            keys = ops.sam_i2t_attention_ln(keys, pos, pr[0], pr[1], l.cross_attn_image_to_token, l.norm4, late_bias_q=(i == 0))
        att = ops.sam_t2i_attention(pr[2], keys, pos, tr.final_attn_token_to_image, late_bias_kv=False)
        hs, _ = ops.sam_out_ln(att, queries, tokens, tr.final_attn_token_to_image.out_proj, tr.norm_final_attn)
        ln = md.output_upscaling[1]
        y1 = ops.linear(keys.view(n * P, D), pk["up0_w"], pk["up0_b"])
        y1 = ops.layernorm2d_cl(y1.view(-1, D // 4), ln.weight, ln.bias, 1e-6, gelu=True)
        y2 = ops.linear(y1, pk["up3_w"], pk["up3_b"], act="gelu")
        hyper, iou = ops.sam_small_mlps(hs, md.output_hypernetworks_mlps, md.iou_prediction_head)
        return ops.mask_matmul(hyper, y2, n, nm, D // 8, g), iou

    def _mlp3(self, m, x):
        nl = len(m.layers)
        for i, l in enumerate(m.layers):
            x = ops.linear(x, l.weight, l.bias, act="relu" if i < nl - 1 else None)
        return x

    def decode(self, image_embedding_tm: torch.Tensor, text_embeds: torch.Tensor,
               image_index: Optional[torch.Tensor] = None, trace: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """image_embedding_tm [g*g, D] (one image), text_embeds [n, D] -> (masks [n, nm, 4g, 4g] bf16, iou [n, nm]).
        Covers PromptEncoder.forward with text_embeds only (sparse = text, dense = no_mask_embed) + MaskDecoder.predict_masks.
        With image_index (int64 [n]) image_embedding_tm is [B, g*g, D] and prompt j decodes against image image_index[j]: the
        prompts of a whole batch go through ONE chain of launches (every op below is independent per prompt, so the results are
        the per-image results).  trace: optional dict that receives the stage outputs (same names as the oracle's trace)."""
        def rec(name, t, shape=None):
            if trace is not None:
                trace[name] = t if shape is None else t.view(*shape)
            return t
        md, tr, pk = self.sam.mask_decoder, self.sam.mask_decoder.transformer, self.pk()
        n, D = text_embeds.shape
        P = image_embedding_tm.shape[-2]
        g = int(math.isqrt(P))
        nm = md.mask_tokens.weight.shape[0]
        out_tok = torch.cat([md.iou_token.weight, md.mask_tokens.weight], dim=0)                  # [1+nm, D]
        tokens = torch.cat([out_tok.unsqueeze(0).expand(n, -1, -1), text_embeds.unsqueeze(1)], dim=1).contiguous()   # [n, T, D]
        T = tokens.shape[1]
        if trace is None and self.fused_decoder and self._fusable(D, P, T, nm) and tokens.dtype != torch.float32:
            return self._decode_fused(image_embedding_tm, tokens, image_index, n, D, P, g, nm, T)
        src = ops.add_rows(image_embedding_tm.reshape(-1, D), self.sam.prompt_encoder.no_mask_embed.weight)   # + dense (no-mask) embedding
        if image_index is None:
            keys = src.unsqueeze(0).expand(n, -1, -1).contiguous().view(n * P, D)                # repeat_interleave over prompts
        else:
            keys = ops.gather_rows(src.view(-1, P * D), image_index).view(n * P, D)              # each prompt's own image
        pos = self.dense_pe()
        qpe = tokens.view(n * T, D)
        queries = qpe
        for i, l in enumerate(tr.layers):
            if i == 0:
                queries = self._attn(l.self_attn, queries, queries, queries, n, T, T)
            else:
                q = ops.add_rows(queries, qpe)
                queries = self._attn(l.self_attn, q, q, queries, n, T, T, residual=queries)
            queries = rec(f"l{i}.norm1", ops.layernorm(queries, l.norm1.weight, l.norm1.bias, 1e-5), (n, T, D))
            q = ops.add_rows(queries, qpe)
            k = ops.add_rows(keys, pos)                                   # keys + key_pe feeds both cross attentions of the block
            # layer 0: `keys` / `keys + key_pe` are still non-contiguous views in the reference (image_embedding.flatten(2).permute(0, 2, 1),
            # transformer.py:92-93) -> their projections take at::linear's unfused-bias path; norm4 makes keys contiguous afterwards
            queries = self._attn(l.cross_attn_token_to_image, q, k, keys, n, T, P, residual=queries, trace=trace, tag=f"l{i}.t2i",
                                 unfused_bias=("k", "v") if i == 0 else ())
            queries = rec(f"l{i}.norm2", ops.layernorm(queries, l.norm2.weight, l.norm2.bias, 1e-5), (n, T, D))
            m = ops.linear(queries, l.mlp.lin1.weight, l.mlp.lin1.bias, act="relu")
            queries = ops.linear(m, l.mlp.lin2.weight, l.mlp.lin2.bias, residual=queries)
            queries = rec(f"l{i}.norm3", ops.layernorm(queries, l.norm3.weight, l.norm3.bias, 1e-5), (n, T, D))
            q = ops.add_rows(queries, qpe)
            keys = self._attn(l.cross_attn_image_to_token, k, q, queries, n, P, T, residual=keys, trace=trace, tag=f"l{i}.i2t",
                              unfused_bias=("q",) if i == 0 else ())
            keys = rec(f"l{i}.norm4", ops.layernorm(keys, l.norm4.weight, l.norm4.bias, 1e-5), (n, P, D))
        q = ops.add_rows(queries, qpe)
        k = ops.add_rows(keys, pos)
        queries = self._attn(tr.final_attn_token_to_image, q, k, keys, n, T, P, residual=queries)
        hs = rec("final.norm", ops.layernorm(queries, tr.norm_final_attn.weight, tr.norm_final_attn.bias, 1e-5).view(n, T, D))
        # upscaling: ConvT(k2,s2) -> LN2d -> GELU -> ConvT(k2,s2) -> GELU, kept in blocked [cell][d1][d2][c] order
        ln = md.output_upscaling[1]
        y1 = ops.linear(keys, pk["up0_w"], pk["up0_b"])                                         # [n*P, 4*D/4]
        y1 = ops.layernorm2d_cl(y1.view(-1, D // 4), ln.weight, ln.bias, 1e-6, gelu=True)        # [n*P*4, D/4]
        y2 = ops.linear(y1, pk["up3_w"], pk["up3_b"], act="gelu")                                # [n*P*4, 4*D/8]
        hyper = rec("hyper", torch.stack([self._mlp3(md.output_hypernetworks_mlps[t], hs[:, 1 + t, :]) for t in range(nm)], dim=1).contiguous())
        if trace is not None:       # blocked [n][cell][d1][c] / [n][cell][d1][d2][c] -> NCHW like the reference's tensors
            trace["up1"] = y1.view(n, g, g, 2, 2, D // 4).permute(0, 5, 1, 3, 2, 4).reshape(n, D // 4, 2 * g, 2 * g)
            trace["up2"] = y2.view(n, g, g, 2, 2, 2, 2, D // 8).permute(0, 7, 1, 3, 5, 2, 4, 6).reshape(n, D // 8, 4 * g, 4 * g)
        masks = ops.mask_matmul(hyper, y2, n, nm, D // 8, g)
        iou = self._mlp3(md.iou_prediction_head, hs[:, 0, :])
        return masks, iou

    # -- training path of the mask decoder (train_ullava.py:248-261 makes `mask_decoder` trainable) -----------------------------------
    def decode_train(self, image_embedding_tm: torch.Tensor, text_embeds: torch.Tensor, image_index: Optional[torch.Tensor] = None):
        """decode() with an autograd graph: same HIP forward kernels (GELU un-fused from the second up-scaling GEMM so that its input is
        kept, ConvTranspose2d weights re-packed under autograd), HIP backward kernels (autograd_ops.py).  The image embedding, the
        dense (no-mask) embedding and the positional encoding are constants (SAM encoder and prompt encoder are frozen)."""
        md, tr = self.sam.mask_decoder, self.sam.mask_decoder.transformer
        n, D = text_embeds.shape
        P = image_embedding_tm.shape[-2]
        g = int(math.isqrt(P))
        nm = md.mask_tokens.weight.shape[0]
        H = self.cfg.decoder_heads
        u0, u3 = md.output_upscaling[0], md.output_upscaling[3]
        up0_w = u0.weight.permute(2, 3, 1, 0).reshape(-1, u0.weight.shape[0])
        up3_w = u3.weight.permute(2, 3, 1, 0).reshape(-1, u3.weight.shape[0])
        up0_b, up3_b = u0.bias.repeat(4), u3.bias.repeat(4)
        out_tok = torch.cat([md.iou_token.weight, md.mask_tokens.weight], dim=0)
        tokens = torch.cat([out_tok.unsqueeze(0).expand(n, -1, -1), text_embeds.unsqueeze(1)], dim=1).contiguous()
        T = tokens.shape[1]
        with torch.no_grad():
            src = ops.add_rows(image_embedding_tm.reshape(-1, D), self.sam.prompt_encoder.no_mask_embed.weight)
            if image_index is None:
                keys = src.unsqueeze(0).expand(n, -1, -1).contiguous().view(n * P, D)
            else:
                keys = ops.gather_rows(src.view(-1, P * D), image_index).view(n * P, D)
            pos = self.dense_pe()

        def attn(a, q_in, k_in, v_in, Sq, Sk, residual=None, unfused=()):
            q = A.linear(q_in, a.q_proj.weight, a.q_proj.bias, bias_after_rounding="q" in unfused)
            k = A.linear(k_in, a.k_proj.weight, a.k_proj.bias, bias_after_rounding="k" in unfused)
            v = A.linear(v_in, a.v_proj.weight, a.v_proj.bias, bias_after_rounding="v" in unfused)
            return A.linear(A.attention(q, k, v, n, H, Sq, Sk), a.out_proj.weight, a.out_proj.bias, residual=residual)

        def ln(x, m):
            return A.layernorm(x, m.weight, m.bias, 1e-5)
        qpe = tokens.view(n * T, D)
        queries = qpe
        for i, l in enumerate(tr.layers):
            if i == 0:
                queries = attn(l.self_attn, queries, queries, queries, T, T)
            else:
                q = A.add(queries, qpe)
                queries = attn(l.self_attn, q, q, queries, T, T, residual=queries)
            queries = ln(queries, l.norm1)
            q = A.add(queries, qpe)
            k = A.add(keys, pos)
            queries = ln(attn(l.cross_attn_token_to_image, q, k, keys, T, P, residual=queries, unfused=("k", "v") if i == 0 else ()), l.norm2)
            m = A.linear(queries, l.mlp.lin1.weight, l.mlp.lin1.bias, relu=True)
            queries = ln(A.linear(m, l.mlp.lin2.weight, l.mlp.lin2.bias, residual=queries), l.norm3)
            q = A.add(queries, qpe)
            keys = ln(attn(l.cross_attn_image_to_token, k, q, queries, P, T, residual=keys, unfused=("q",) if i == 0 else ()), l.norm4)
        q = A.add(queries, qpe)
        k = A.add(keys, pos)
        hs = ln(attn(tr.final_attn_token_to_image, q, k, keys, T, P, residual=queries), tr.norm_final_attn).view(n, T, D)
        lnu = md.output_upscaling[1]
        y1 = A.linear(keys, up0_w, up0_b)
        y1 = A.layernorm2d_cl(y1.view(-1, D // 4), lnu.weight, lnu.bias, 1e-6, True)
        y2 = A.gelu(A.linear(y1, up3_w, up3_b))

        def mlp3(mm, x):
            nl = len(mm.layers)
            for j, lyr in enumerate(mm.layers):
                x = A.linear(x, lyr.weight, lyr.bias, relu=j < nl - 1)
            return x
        hyper = torch.stack([mlp3(md.output_hypernetworks_mlps[t], hs[:, 1 + t, :].contiguous()) for t in range(nm)], dim=1).contiguous()
        return A.mask_matmul(hyper, y2, n, nm, D // 8, g)

    def postprocess_train(self, masks: torch.Tensor, input_size, original_size) -> torch.Tensor:
        s_ = self.cfg.img_size
        up = A.bilinear(masks, masks.shape[-2], masks.shape[-1], s_, s_)
        return A.bilinear(up, int(input_size[0]), int(input_size[1]), int(original_size[0]), int(original_size[1]))

    # -- Sam.postprocess_masks (sam.py:137-172) --------------------------------------------------------------------
    def postprocess(self, masks: torch.Tensor, input_size, original_size) -> torch.Tensor:
        """masks [n, H, W] (bf16/fp32, contiguous) -> fp32 [n, orig_h, orig_w]."""
        s = self.cfg.img_size
        up = ops.bilinear(masks, masks.shape[-2], masks.shape[-1], s, s)
        return ops.bilinear(up, int(input_size[0]), int(input_size[1]), int(original_size[0]), int(original_size[1]))
