"""CPU oracle for the u-LLaVA multimodal forward path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU) functional restatement of the reference's
algorithm for the hot path named in BASELINE.json (CLIP ViT-L -> projector ->
LLaMA -> [SEG] -> SAM prompt-encoder + MaskDecoder -> postprocess).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it; the product path (`u-llava_amd/`) never does and fails loudly when the HIP
extension is missing.

Parity pin: every function here is checked bit-for-bit (torch.equal, fp32 and
bf16) against the reference itself imported from /root/reference in the build
container -- see `tests/golden/gen_golden.py`, which also writes the committed
fixtures `tests/golden/*.pt` that `tests/test_oracle_golden.py` replays without
the reference present.

The arithmetic that lives in the reference's third-party dependency
(`transformers` -- pinned ==4.29.1 in shells/requirements.txt:19, NOT vendored
under /root/reference) is restated from the version importable here,
transformers 5.15.0 with attn_implementation="eager" (explicit matmul -> scale
-> +mask -> fp32 softmax -> matmul, the same op order 4.29.1 uses).  Parity is
therefore declared against "reference glue + transformers 5.15.0 eager".

All weights come in as a flat ``sd`` dict using the reference's state-dict key
names:  core ``model.* lm_head.* vision_encoder.*
vision_projector.*``; full model ``llm.* seg_projector.* det_projector.*
det_decoder.* visual_model.{image_encoder,prompt_encoder,mask_decoder}.*``.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# small helpers
# --------------------------------------------------------------------------- #
def linear(x: Tensor, sd: Dict[str, Tensor], name: str) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def layer_norm(x: Tensor, sd: Dict[str, Tensor], name: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def layer_norm_2d(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """reference models/segment_anything/modeling/common.py:31-43 (channel LN on NCHW,
    computed in the input dtype, biased variance)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


# --------------------------------------------------------------------------- #
# LLaMA (transformers LlamaModel, eager) -- SURVEY 8(a) row a8
# --------------------------------------------------------------------------- #
def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """transformers llama/modeling_llama.py LlamaRMSNorm.forward: fp32 x*rsqrt(mean(x^2)+eps)
    -> cast to input dtype -> * weight."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


def rope_tables(position_ids: Tensor, head_dim: int, theta: float, dtype) -> Tuple[Tensor, Tensor]:
    """LlamaRotaryEmbedding.forward: inv_freq = theta^(-2i/d); fp32 cos/sin of pos*inv_freq,
    halves duplicated, cast to activation dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    inv = inv_freq[None, :, None].expand(position_ids.shape[0], -1, 1).to(torch.float)
    pos = position_ids[:, None, :].float()
    freqs = (inv @ pos).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * 1.0).to(dtype), (emb.sin() * 1.0).to(dtype)


def rotate_half(x: Tensor) -> Tensor:
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def causal_additive_mask(attention_mask: Optional[Tensor], B: int, S: int, past: int, dtype) -> Tensor:
    """transformers masking_utils eager mask: 0 where key j may be seen by query i
    (j <= i + past and attention_mask[b, j] != 0), finfo(dtype).min elsewhere."""
    kv = S + past
    i = torch.arange(S)[:, None] + past
    j = torch.arange(kv)[None, :]
    allowed = (j <= i)[None, None].expand(B, 1, S, kv)
    if attention_mask is not None:
        allowed = allowed & (attention_mask[:, None, None, :kv] != 0)
    zero = torch.zeros((), dtype=dtype)
    return torch.where(allowed, zero, torch.full((), torch.finfo(dtype).min, dtype=dtype))


def llama_attention(sd, pfx: str, x: Tensor, cos, sin, mask, n_heads: int, past_kv=None):
    """LlamaAttention.forward + eager_attention_forward (no GQA in LLaMA-7B)."""
    B, S, D = x.shape
    hd = D // n_heads
    q = linear(x, sd, pfx + "q_proj").view(B, S, n_heads, hd).transpose(1, 2)
    k = linear(x, sd, pfx + "k_proj").view(B, S, n_heads, hd).transpose(1, 2)
    v = linear(x, sd, pfx + "v_proj").view(B, S, n_heads, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2)
        v = torch.cat([past_kv[1], v], dim=2)
    w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
    if mask is not None:
        w = w + mask
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(B, S, D).contiguous()
    return linear(o, sd, pfx + "o_proj"), (k, v)


def llama_model(sd, cfg: dict, inputs_embeds: Tensor, attention_mask: Optional[Tensor] = None,
                position_ids: Optional[Tensor] = None, past=None, pfx: str = "model."):
    """LlamaModel.forward.  Returns (hidden_states tuple of L+1 entries with the LAST entry
    post-final-RMSNorm, as HF records them; new_past list)."""
    B, S, _ = inputs_embeds.shape
    L, H = cfg["num_hidden_layers"], cfg["num_attention_heads"]
    eps = cfg.get("rms_norm_eps", 1e-6)
    past_len = 0 if not past else past[0][0].shape[2]
    if position_ids is None:
        position_ids = (torch.arange(S) + past_len).unsqueeze(0)
    hd = cfg["hidden_size"] // H
    cos, sin = rope_tables(position_ids, hd, cfg.get("rope_theta", 10000.0), inputs_embeds.dtype)
    mask = causal_additive_mask(attention_mask, B, S, past_len, inputs_embeds.dtype)
    h = inputs_embeds
    all_h = []
    new_past = []
    for l in range(L):
        all_h.append(h)
        p = f"{pfx}layers.{l}."
        r = h
        a, kv = llama_attention(sd, p + "self_attn.", rms_norm(h, sd[p + "input_layernorm.weight"], eps),
                                cos, sin, mask, H, None if not past else past[l])
        new_past.append(kv)
        h = r + a
        r = h
        y = rms_norm(h, sd[p + "post_attention_layernorm.weight"], eps)
        y = linear(F.silu(linear(y, sd, p + "mlp.gate_proj")) * linear(y, sd, p + "mlp.up_proj"), sd, p + "mlp.down_proj")
        h = r + y
    h = rms_norm(h, sd[pfx + "norm.weight"], eps)
    all_h.append(h)
    return tuple(all_h), new_past


# --------------------------------------------------------------------------- #
# CLIP vision tower (transformers CLIPVisionModel, eager) -- row a6
# --------------------------------------------------------------------------- #
def quick_gelu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(1.702 * x)


def clip_vision_hidden_states(sd, vcfg: dict, pixel_values: Tensor, pfx: str = "vision_encoder.",
                              n_layers_to_run: Optional[int] = None) -> List[Tensor]:
    """CLIPVisionModel.forward(output_hidden_states=True).hidden_states:
    [pre-LN embeddings, out of layer 0, ..., out of layer L-1]."""
    D, H = vcfg["hidden_size"], vcfg["num_attention_heads"]
    eps = vcfg.get("layer_norm_eps", 1e-5)
    L = vcfg["num_hidden_layers"] if n_layers_to_run is None else n_layers_to_run
    w = sd[pfx + "embeddings.patch_embedding.weight"]
    B = pixel_values.shape[0]
    pe = F.conv2d(pixel_values.to(w.dtype), w, None, stride=vcfg["patch_size"])
    pe = pe.flatten(2).transpose(1, 2)
    cls = sd[pfx + "embeddings.class_embedding"].expand(B, 1, -1)
    h = torch.cat([cls, pe], dim=1)
    h = h + sd[pfx + "embeddings.position_embedding.weight"][None, : h.shape[1]]
    h = layer_norm(h, sd, pfx + "pre_layrnorm", eps)
    hs = [h]
    hd = D // H
    for l in range(L):
        p = f"{pfx}encoder.layers.{l}."
        r = h
        x = layer_norm(h, sd, p + "layer_norm1", eps)
        S = x.shape[1]
        q = linear(x, sd, p + "self_attn.q_proj").view(B, S, H, hd).transpose(1, 2)
        k = linear(x, sd, p + "self_attn.k_proj").view(B, S, H, hd).transpose(1, 2)
        v = linear(x, sd, p + "self_attn.v_proj").view(B, S, H, hd).transpose(1, 2)
        a = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
        a = F.softmax(a, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(a, v).transpose(1, 2).contiguous().reshape(B, S, D).contiguous()
        h = r + linear(o, sd, p + "self_attn.out_proj")
        r = h
        x = layer_norm(h, sd, p + "layer_norm2", eps)
        x = linear(quick_gelu(linear(x, sd, p + "mlp.fc1")), sd, p + "mlp.fc2")
        h = r + x
        hs.append(h)
    return hs


# --------------------------------------------------------------------------- #
# u-LLaVA core glue -- rows a4, a5, a6, a7  (reference models/ullava_core.py)
# --------------------------------------------------------------------------- #
def _select_layer_count(vcfg: dict, vision_hidden_layer: int) -> Tuple[int, int]:
    L = vcfg["num_hidden_layers"]
    idx = vision_hidden_layer if vision_hidden_layer >= 0 else L + 1 + vision_hidden_layer
    return idx, idx  # hidden_states[idx] needs `idx` layers to have run


def encode_image(sd, cfg: dict, images: Tensor) -> Tensor:
    """models/ullava_core.py:146-158: hidden_states[vision_hidden_layer][:, 1:]."""
    idx, nrun = _select_layer_count(cfg["vision_config"], cfg["vision_hidden_layer"])
    hs = clip_vision_hidden_states(sd, cfg["vision_config"], images, n_layers_to_run=nrun)
    return hs[idx][:, 1:]


def encode_video(sd, cfg: dict, videos: Tensor) -> Tensor:
    """models/ullava_core.py:160-180: (b t) frames -> CLIP -> mean over t (spatial) and over
    patches (temporal) -> concat([temporal, spatial], dim=1)."""
    b, c, t, hh, ww = videos.shape
    frames = videos.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
    idx, nrun = _select_layer_count(cfg["vision_config"], cfg["vision_hidden_layer"])
    f = clip_vision_hidden_states(sd, cfg["vision_config"], frames, n_layers_to_run=nrun)[idx][:, 1:]
    f = f.reshape(b, t, f.shape[1], f.shape[2])
    spatial = f.mean(dim=1)
    temporal = f.mean(dim=2)
    return torch.concat([temporal, spatial], dim=1)


def vision_projector(sd, cfg: dict, x: Tensor) -> Tensor:
    """models/ullava_core.py:117-129."""
    if cfg.get("projector_type", "mlp") == "mlp":
        return linear(x, sd, "vision_projector")
    if cfg["projector_type"] == "mlp2x":
        return linear(F.gelu(linear(x, sd, "vision_projector.0")), sd, "vision_projector.2")
    raise NotImplementedError


def embed_images_videos(sd, cfg: dict, input_ids: Tensor, images: Optional[Tensor], videos: Optional[Tensor]):
    """models/ullava_core.py:182-277.  Returns inputs_embeds [B,S,D] (or None when S==1)."""
    if input_ids.shape[1] == 1:
        return None
    emb = F.embedding(input_ids, sd["model.embed_tokens.weight"])
    img_f = encode_image(sd, cfg, images) if images is not None else None
    vid_f = encode_video(sd, cfg, videos) if videos is not None else None
    ids = cfg["mm_token_ids"]
    out = []
    ii = vi = 0
    for cur_ids, cur in zip(input_ids, emb):
        n_is, n_ie = int((cur_ids == ids["IMG_START"]).sum()), int((cur_ids == ids["IMG_END"]).sum())
        n_vs, n_ve = int((cur_ids == ids["VID_START"]).sum()), int((cur_ids == ids["VID_END"]).sum())
        assert n_is == n_ie and n_vs == n_ve
        if n_is == 0 and n_vs == 0:
            dummy = torch.zeros(256, 1024, dtype=emb.dtype)
            if sd["vision_projector.weight" if cfg.get("projector_type", "mlp") == "mlp" else "vision_projector.0.weight"].shape[1] != 1024:
                raise RuntimeError("reference hard-codes zeros(256,1024) for text-only samples (ullava_core.py:216)")
            cur = cur + (0.0 * vision_projector(sd, cfg, dummy)).sum()
            new = cur
        elif n_is > 0:
            pos = int(torch.where(cur_ids == ids["IMG_START"])[0][0])
            feat = vision_projector(sd, cfg, img_f[ii])
            n = feat.shape[0]
            if cfg.get("projector_from_scratch", False):     # :230-240: text rows detached except the start / end token rows
                new = torch.cat((cur[:pos].detach(), cur[pos:pos + 1], feat, cur[pos + n + 1:pos + n + 2], cur[pos + n + 2:].detach()), dim=0)
            else:
                new = torch.cat((cur[: pos + 1], feat, cur[pos + n + 1:]), dim=0)
            ii += 1
        else:
            pos = int(torch.where(cur_ids == ids["VID_START"])[0][0])
            feat = vision_projector(sd, cfg, vid_f[vi])
            n = feat.shape[0]
            if cfg.get("projector_from_scratch", False):     # :255-264
                new = torch.cat((cur[:pos].detach(), cur[pos:pos + 1], feat, cur[pos + n + 1:pos + n + 2], cur[pos + n + 2:].detach()), dim=0)
            else:
                new = torch.cat((cur[: pos + 1], feat, cur[pos + n + 1:]), dim=0)
            vi += 1
        out.append(new)
    return torch.stack(out, dim=0)


def core_forward(sd, cfg: dict, input_ids: Tensor, attention_mask: Optional[Tensor] = None,
                 images: Optional[Tensor] = None, videos: Optional[Tensor] = None,
                 position_ids: Optional[Tensor] = None, past=None, labels: Optional[Tensor] = None):
    """UllavaCoreForCausalLM.forward (models/ullava_core.py:279-355).
    Returns dict(logits, hidden_states, past, inputs_embeds, loss)."""
    emb = embed_images_videos(sd, cfg, input_ids, images, videos)
    if emb is None:
        emb = F.embedding(input_ids, sd["model.embed_tokens.weight"])
    hs, new_past = llama_model(sd, cfg, emb, attention_mask, position_ids, past)
    logits = F.linear(hs[-1], sd["lm_head.weight"])
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits[..., :-1, :].reshape(-1, logits.shape[-1]), labels[..., 1:].reshape(-1))
    return dict(logits=logits, hidden_states=hs, past=new_past, inputs_embeds=emb, loss=loss)


def greedy_generate(sd, cfg: dict, input_ids: Tensor, images=None, videos=None, max_new_tokens: int = 8,
                    eos_token_id: Optional[int] = None, attention_mask: Optional[Tensor] = None):
    """Greedy decode through core_forward WITHOUT a KV cache (the configuration the reference
    checkpoints run: use_cache=False).  Returns (sequences, last-step hidden_states[-1]).
    With an attention_mask (left-padded batches) every step derives position_ids = cumsum(mask) - 1 with
    pad positions set to 1, as prepare_inputs_for_generation does (models/ullava_core.py:371-377)."""
    seq = input_ids.clone()
    last_h = None
    for _ in range(max_new_tokens):
        if attention_mask is None:
            mask, pos = torch.ones_like(seq), None
        else:
            mask = torch.cat([attention_mask, attention_mask.new_ones(seq.shape[0], seq.shape[1] - attention_mask.shape[1])], dim=1)
            pos = mask.long().cumsum(-1) - 1
            pos.masked_fill_(mask == 0, 1)
        o = core_forward(sd, cfg, seq, mask, images, videos, position_ids=pos)
        last_h = o["hidden_states"][-1]
        nxt = o["logits"][:, -1].float().argmax(-1, keepdim=True)
        seq = torch.cat([seq, nxt], dim=1)
        if eos_token_id is not None and bool((nxt == eos_token_id).all()):
            break
    return seq, last_h


# --------------------------------------------------------------------------- #
# SAM image encoder (ViT-H) -- row a3  (segment_anything/modeling/image_encoder.py)
# --------------------------------------------------------------------------- #
def window_partition(x: Tensor, ws: int):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph > 0 or pw > 0:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(w: Tensor, ws: int, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    if Hp > H or Wp > W:
        x = x[:, :H, :W, :].contiguous()
    return x


def get_rel_pos(q_size: int, k_size: int, rel_pos: Tensor) -> Tensor:
    """image_encoder.py:321-351."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel, mode="linear")
        r = r.reshape(-1, max_rel).permute(1, 0)
    else:
        r = rel_pos
    qc = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    kc = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (qc - kc) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def sam_attention(sd, p: str, x: Tensor, n_heads: int) -> Tensor:
    """image_encoder.py:235-260 + add_decomposed_rel_pos :354-392."""
    B, H, W, C = x.shape
    qkv = linear(x, sd, p + "qkv").reshape(B, H * W, 3, n_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * n_heads, H * W, -1).unbind(0)
    hd = q.shape[-1]
    attn = (q * (hd ** -0.5)) @ k.transpose(-2, -1)
    Rh = get_rel_pos(H, H, sd[p + "rel_pos_h"])
    Rw = get_rel_pos(W, W, sd[p + "rel_pos_w"])
    r_q = q.reshape(B * n_heads, H, W, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(B * n_heads, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(
        B * n_heads, H * W, H * W)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).view(B, n_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return linear(o, sd, p + "proj")


def sam_image_encoder(sd, scfg: dict, x: Tensor, pfx: str = "visual_model.image_encoder.", trace: Optional[dict] = None) -> Tensor:
    """ImageEncoderViT.forward (image_encoder.py:110-125); the neck of an fp16 model runs in fp32 (:117-124), bf16/fp32 go
    straight through it.
    trace: optional dict receiving the stage outputs ([B, H, W, C] after the patch embedding and after every block)."""
    ps = scfg["patch_size"]
    x = F.conv2d(x, sd[pfx + "patch_embed.proj.weight"], sd[pfx + "patch_embed.proj.bias"], stride=ps).permute(0, 2, 3, 1)
    x = x + sd[pfx + "pos_embed"]
    if trace is not None:
        trace["embed"] = x
    for i in range(scfg["depth"]):
        p = f"{pfx}blocks.{i}."
        ws = 0 if i in scfg["global_attn_indexes"] else scfg["window_size"]
        sc = x
        y = layer_norm(x, sd, p + "norm1", 1e-6)
        if ws > 0:
            Hh, Ww = y.shape[1], y.shape[2]
            y, pad_hw = window_partition(y, ws)
        y = sam_attention(sd, p + "attn.", y, scfg["num_heads"])
        if ws > 0:
            y = window_unpartition(y, ws, pad_hw, (Hh, Ww))
        x = sc + y
        if trace is not None:
            trace[f"block{i}.attn"] = x
        y = layer_norm(x, sd, p + "norm2", 1e-6)
        x = x + linear(F.gelu(linear(y, sd, p + "mlp.lin1")), sd, p + "mlp.lin2")
        if trace is not None:
            trace[f"block{i}"] = x
    x = x.permute(0, 3, 1, 2)
    dtype = x.dtype
    # image_encoder.py:117-124: an fp16 model runs the neck under torch.autocast("cuda", dtype=float32) ("prevent overflow"):
    # both convolutions get fp32 casts of their input and of their fp16-valued weights, LayerNorm2d's arithmetic promotes
    # (fp16 weight * fp32 tensor -> fp32), and only the neck's result is cast back to fp16.  bf16 / fp32 go straight through.
    # (The pinned torch==1.13.1 honours fast_dtype=float32; torch >= 2.x warns "target dtype is not supported" and disables
    # the context, and a CPU-only host disables it as well -- the oracle restates the branch the reference's authors ran.)
    up = (lambda t: t.float()) if dtype == torch.float16 else (lambda t: t)
    x = F.conv2d(up(x), up(sd[pfx + "neck.0.weight"]))
    x = layer_norm_2d(x, up(sd[pfx + "neck.1.weight"]), up(sd[pfx + "neck.1.bias"]))
    x = F.conv2d(x, up(sd[pfx + "neck.2.weight"]), padding=1)
    x = layer_norm_2d(x, up(sd[pfx + "neck.3.weight"]), up(sd[pfx + "neck.3.bias"]))
    return x.to(dtype)


# --------------------------------------------------------------------------- #
# SAM prompt encoder / mask decoder / postprocess -- rows a11-a15
# --------------------------------------------------------------------------- #
def dense_pe(sd, emb_hw: Tuple[int, int], pfx: str = "visual_model.prompt_encoder.") -> Tensor:
    """PromptEncoder.get_dense_pe -> PositionEmbeddingRandom.forward (prompt_encoder.py:67-76,216-229).
    Computed in the dtype of the Gaussian buffer (bf16 when the model is bf16)."""
    g = sd[pfx + "pe_layer.positional_encoding_gaussian_matrix"]
    h, w = emb_hw
    grid = torch.ones((h, w), dtype=g.dtype)
    y = (grid.cumsum(dim=0) - 0.5) / h
    x = (grid.cumsum(dim=1) - 0.5) / w
    c = torch.stack([x, y], dim=-1)
    c = 2 * c - 1
    c = c @ g
    c = 2 * math.pi * c   # np.pi == math.pi
    pe = torch.cat([torch.sin(c), torch.cos(c)], dim=-1)
    return pe.permute(2, 0, 1).unsqueeze(0)


def prompt_encoder_text(sd, text_embeds: Tensor, emb_hw: Tuple[int, int], pfx: str = "visual_model.prompt_encoder."):
    """PromptEncoder.forward with points=boxes=masks=None (prompt_encoder.py:140-186): sparse is the
    fp32 concat of an empty fp32 tensor with text_embeds; dense is no_mask_embed broadcast."""
    bs = text_embeds.shape[0]
    d = sd[pfx + "no_mask_embed.weight"].shape[1]
    sparse = torch.cat([torch.empty((bs, 0, d)), text_embeds], dim=1)
    dense = sd[pfx + "no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(bs, -1, emb_hw[0], emb_hw[1])
    return sparse, dense


def _sam_attn(sd, p: str, q: Tensor, k: Tensor, v: Tensor, n_heads: int, trace: Optional[dict] = None, tag: str = "") -> Tensor:
    """transformer.py:220-242 (Attention.forward; softmax in the input dtype)."""
    q, k, v = linear(q, sd, p + "q_proj"), linear(k, sd, p + "k_proj"), linear(v, sd, p + "v_proj")
    if trace is not None:
        trace.update({tag + ".q": q, tag + ".k": k, tag + ".v": v})

    def sep(t):
        b, n, c = t.shape
        return t.reshape(b, n, n_heads, c // n_heads).transpose(1, 2)
    q, k, v = sep(q), sep(k), sep(v)
    c = q.shape[-1]
    a = q @ k.permute(0, 1, 3, 2)
    a = a / math.sqrt(c)
    a = torch.softmax(a, dim=-1)
    o = a @ v
    b, nh, nt, ch = o.shape
    o = o.transpose(1, 2).reshape(b, nt, nh * ch)
    if trace is not None:
        trace[tag + ".att"] = o
    return linear(o, sd, p + "out_proj")


def two_way_transformer(sd, p: str, image_embedding: Tensor, image_pe: Tensor, point_embedding: Tensor,
                        depth: int = 2, n_heads: int = 8, trace: Optional[dict] = None):
    """transformer.py:62-106 + TwoWayAttentionBlock :151-182.  trace: optional dict that receives the stage outputs
    (stage-level parity tests of the HIP path)."""
    def tr(name, t):
        if trace is not None:
            trace[name] = t
        return t
    image_embedding = image_embedding.flatten(2).permute(0, 2, 1)
    image_pe = image_pe.flatten(2).permute(0, 2, 1)
    queries, keys = point_embedding, image_embedding
    for i in range(depth):
        lp = f"{p}layers.{i}."
        if i == 0:
            queries = _sam_attn(sd, lp + "self_attn.", queries, queries, queries, n_heads)
        else:
            q = queries + point_embedding
            queries = queries + _sam_attn(sd, lp + "self_attn.", q, q, queries, n_heads)
        queries = tr(f"l{i}.norm1", layer_norm(queries, sd, lp + "norm1", 1e-5))
        q = queries + point_embedding
        k = keys + image_pe
        queries = queries + tr(f"l{i}.t2i", _sam_attn(sd, lp + "cross_attn_token_to_image.", q, k, keys, n_heads, trace, f"l{i}.t2i"))
        queries = tr(f"l{i}.norm2", layer_norm(queries, sd, lp + "norm2", 1e-5))
        m = linear(F.relu(linear(queries, sd, lp + "mlp.lin1")), sd, lp + "mlp.lin2")
        queries = tr(f"l{i}.norm3", layer_norm(queries + m, sd, lp + "norm3", 1e-5))
        q = queries + point_embedding
        k = keys + image_pe
        keys = keys + tr(f"l{i}.i2t", _sam_attn(sd, lp + "cross_attn_image_to_token.", k, q, queries, n_heads, trace, f"l{i}.i2t"))
        keys = tr(f"l{i}.norm4", layer_norm(keys, sd, lp + "norm4", 1e-5))
    q = queries + point_embedding
    k = keys + image_pe
    queries = queries + tr("final.t2i", _sam_attn(sd, p + "final_attn_token_to_image.", q, k, keys, n_heads))
    queries = tr("final.norm", layer_norm(queries, sd, p + "norm_final_attn", 1e-5))
    return queries, keys


def _mlp3(sd, p: str, x: Tensor, n: int = 3) -> Tensor:
    for i in range(n):
        x = linear(x, sd, f"{p}layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def mask_decoder(sd, image_embeddings: Tensor, image_pe: Tensor, sparse: Tensor, dense: Tensor,
                 multimask_output: bool = False, pfx: str = "visual_model.mask_decoder.", trace: Optional[dict] = None):
    """MaskDecoder.forward/predict_masks (mask_decoder.py:75-164).  trace: optional dict receiving the stage outputs."""
    n_mask_tokens = sd[pfx + "mask_tokens.weight"].shape[0]
    out_tok = torch.cat([sd[pfx + "iou_token.weight"], sd[pfx + "mask_tokens.weight"]], dim=0)
    out_tok = out_tok.unsqueeze(0).expand(sparse.size(0), -1, -1)
    tokens = torch.cat((out_tok, sparse), dim=1)
    src = torch.repeat_interleave(image_embeddings, tokens.shape[0], dim=0)
    src = src + dense
    pos_src = torch.repeat_interleave(image_pe, tokens.shape[0], dim=0)
    b, c, h, w = src.shape
    hs, src = two_way_transformer(sd, pfx + "transformer.", src, pos_src, tokens, trace=trace)
    iou_tok = hs[:, 0, :]
    mask_toks = hs[:, 1:(1 + n_mask_tokens), :]
    src = src.transpose(1, 2).view(b, c, h, w)
    up = F.conv_transpose2d(src, sd[pfx + "output_upscaling.0.weight"], sd[pfx + "output_upscaling.0.bias"], stride=2)
    if trace is not None:
        trace["up0"] = up
    up = layer_norm_2d(up, sd[pfx + "output_upscaling.1.weight"], sd[pfx + "output_upscaling.1.bias"])
    up = F.gelu(up)
    if trace is not None:
        trace["up1"] = up
    up = F.conv_transpose2d(up, sd[pfx + "output_upscaling.3.weight"], sd[pfx + "output_upscaling.3.bias"], stride=2)
    up = F.gelu(up)
    hyper = torch.stack([_mlp3(sd, f"{pfx}output_hypernetworks_mlps.{i}.", mask_toks[:, i, :]) for i in range(n_mask_tokens)], dim=1)
    if trace is not None:
        trace["up2"], trace["hyper"] = up, hyper
    b, c, h, w = up.shape
    masks = (hyper @ up.view(b, c, h * w)).view(b, n_mask_tokens, h, w)
    iou = _mlp3(sd, pfx + "iou_prediction_head.", iou_tok)
    sl = slice(1, None) if multimask_output else slice(0, 1)
    return masks[:, sl, :, :], iou[:, sl]


def postprocess_masks(masks: Tensor, input_size, original_size, img_size: int = 1024) -> Tensor:
    """Sam.postprocess_masks (sam.py:137-172): fp32 bilinear -> crop -> bilinear."""
    masks = F.interpolate(masks.float(), (img_size, img_size), mode="bilinear", align_corners=False)
    masks = masks[..., : int(input_size[0]), : int(input_size[1])]
    return F.interpolate(masks, tuple(int(v) for v in original_size), mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------- #
# full u-LLaVA forward / evaluate -- rows a1, a2, a10 (reference models/ullava.py)
# --------------------------------------------------------------------------- #
def _sub(sd, prefix: str) -> Dict[str, Tensor]:
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def _proj_mlp(sd, p: str, x: Tensor) -> Tensor:
    """seg_projector / det_projector: Linear-ReLU-Linear-Dropout(0) (ullava.py:86-91,113-118)."""
    return linear(F.relu(linear(x, sd, p + "0")), sd, p + "2")


def det_decoder(sd, x: Tensor) -> Tensor:
    """ullava.py:96-102."""
    x = F.relu(linear(x, sd, "det_decoder.0"))
    x = F.relu(linear(x, sd, "det_decoder.2"))
    return linear(x, sd, "det_decoder.4")


def _decode_masks(sd, cfg, image_embeddings, pred_embeddings: Sequence[Tensor], resize_list, size_list):
    emb_hw = tuple(image_embeddings.shape[-2:])
    img_size = cfg["sam"]["img_size"]
    out, low = [], []
    for i, pe_i in enumerate(pred_embeddings):
        sparse, dense = prompt_encoder_text(sd, pe_i.unsqueeze(1), emb_hw)
        sparse = sparse.to(pe_i.dtype)
        lr, _iou = mask_decoder(sd, image_embeddings[i].unsqueeze(0), dense_pe(sd, emb_hw), sparse, dense, False)
        pm = postprocess_masks(lr, resize_list[i], size_list[i], img_size)
        out.append(pm[:, 0])
        low.append(lr)
    return out, low


def _gather_tokens(h_proj: Tensor, mask: Tensor) -> List[Tensor]:
    sel = h_proj[mask]
    off = torch.cat([torch.zeros(1).long(), mask.int().sum(-1).cumsum(-1)], dim=0)
    return [sel[int(off[i]): int(off[i + 1])] for i in range(mask.shape[0])]


def ullava_forward(sd, cfg: dict, images_sam: Tensor, images: Tensor, input_ids: Tensor,
                   attention_mask: Tensor, size_list, resize_list, labels: Optional[Tensor] = None):
    """UllavaForCausalLM.forward(inference=True) (models/ullava.py:152-268).
    cfg: dict(llm=<core cfg>, sam=<sam cfg>, seg_token_idx, loc_token_idx)."""
    llm_sd = _sub(sd, "llm.")
    B = input_ids.shape[0]
    image_embeddings = torch.cat([sam_image_encoder(sd, cfg["sam"], images_sam[i].unsqueeze(0)) for i in range(B)], 0)
    pad = torch.zeros((B, 1)).bool()
    seg_mask = torch.cat([input_ids[:, 1:] == cfg["seg_token_idx"], pad], dim=1)
    loc_mask = torch.cat([input_ids[:, 1:] == cfg["loc_token_idx"], pad], dim=1)
    out = core_forward(llm_sd, cfg["llm"], input_ids, attention_mask, images, labels=labels)
    last = out["hidden_states"][-1]
    seg_emb = _gather_tokens(_proj_mlp(sd, "seg_projector.", last), seg_mask)
    loc_emb = _gather_tokens(_proj_mlp(sd, "det_projector.", last), loc_mask)
    pred_masks, low_res = _decode_masks(sd, cfg, image_embeddings, seg_emb, resize_list, size_list)
    pred_boxes = [det_decoder(sd, e) for e in loc_emb]
    return dict(pred_masks=pred_masks, pred_boxes=pred_boxes, logits=out["logits"], low_res_masks=low_res,
                image_embeddings=image_embeddings, last_hidden_state=last, ce_loss=out["loss"])


# ---- training losses (forward values), models/loss.py + models/ullava.py:268-333 ------------------------------------------------
def dice_loss(inputs: Tensor, targets: Tensor, num_masks: float, scale=1000, eps=1e-6) -> Tensor:
    """models/loss.py:45-69."""
    inputs = inputs.sigmoid().flatten(1, 2)
    targets = targets.flatten(1, 2)
    numerator = 2 * (inputs / scale * targets).sum(-1)
    denominator = (inputs / scale).sum(-1) + (targets / scale).sum(-1)
    loss = 1 - (numerator + eps) / (denominator + eps)
    return loss.sum() / (num_masks + 1e-8)


def sigmoid_ce_loss(inputs: Tensor, targets: Tensor, num_masks: float) -> Tensor:
    """models/loss.py:72-89."""
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)


def _box_area(b: Tensor) -> Tensor:
    """torchvision.ops.boxes.box_area (third-party, absent here): (x1 - x0) * (y1 - y0)."""
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def generalized_box_iou(b1: Tensor, b2: Tensor) -> Tensor:
    """models/loss.py:6-42 (box_iou + generalized_box_iou), pairwise [N, M]."""
    area1, area2 = _box_area(b1), _box_area(b2)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    iou = inter / union
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


def bbox_l1_loss(src: Tensor, tgt: Tensor, num_boxes: float) -> Tensor:
    """models/loss.py:92-95."""
    return F.l1_loss(src, tgt, reduction="none").sum() / (num_boxes + 1e-8)


def bbox_giou_loss(src: Tensor, tgt: Tensor, num_boxes: float) -> Tensor:
    """models/loss.py:98-110: boxes whose x1 < x0 or y1 < y0 are dropped first."""
    keep = (src[:, 2:] >= src[:, :2]).all(-1)
    src, tgt = src[keep], tgt[keep]
    return (1 - torch.diag(generalized_box_iou(src, tgt))).sum() / (num_boxes + 1e-8)


def ullava_losses(pred_masks, pred_boxes, gt_masks, gt_boxes, ce_loss: Tensor, weights: dict) -> Dict[str, Tensor]:
    """models/ullava.py:268-333: the dict UllavaForCausalLM.forward(inference=False) returns.
    weights: ce_weight, bce_weight, dice_weight, l1_weight, iou_weight (UllavaConfig)."""
    ce_loss = ce_loss * weights["ce_weight"]
    loss = ce_loss
    mask_bce, mask_dice, num_masks = 0, 0, 0
    box_l1, box_giou, num_boxes = 0, 0, 0
    for i in range(len(pred_masks)):
        gm, pm = gt_masks[i], pred_masks[i]
        assert gm.shape[0] == pm.shape[0]
        mask_bce = mask_bce + sigmoid_ce_loss(pm, gm, num_masks=gm.shape[0]) * gm.shape[0]
        mask_dice = mask_dice + dice_loss(pm, gm, num_masks=gm.shape[0]) * gm.shape[0]
        num_masks += gm.shape[0]
        gb, pb = gt_boxes[i], pred_boxes[i]
        assert gb.shape[0] == pb.shape[0]
        box_l1 = box_l1 + bbox_l1_loss(pb, gb, gb.shape[0])
        box_giou = box_giou + bbox_giou_loss(pb, gb, gb.shape[0])
        num_boxes += gb.shape[0]
    mask_bce = weights["bce_weight"] * mask_bce / (num_masks + 1e-8)
    mask_dice = weights["dice_weight"] * mask_dice / (num_masks + 1e-8)
    mask_loss = mask_bce + mask_dice
    box_l1 = weights["l1_weight"] * box_l1 / (num_boxes + 1e-8)
    box_giou = weights["iou_weight"] * box_giou / (num_boxes + 1e-8)
    bbox_loss = box_l1 + box_giou
    # the reference accumulates IN PLACE into the tensor that `ce_loss` also names (ullava.py:271-272,323-324), so the returned
    # "ce_loss" is the total loss, not the weighted next-token loss: reproduced, not corrected
    loss += mask_loss
    loss += bbox_loss
    return dict(loss=loss, ce_loss=ce_loss, mask_bce_loss=mask_bce, mask_dice_loss=mask_dice, mask_loss=mask_loss, bbox_loss=bbox_loss)


def ullava_evaluate(sd, cfg: dict, images_sam: Tensor, images: Tensor, input_ids: Tensor, raw_size_list,
                    resize_list, max_new_tokens: int = 32, eos_token_id: Optional[int] = None):
    """UllavaForCausalLM.evaluate with temperature=0 (greedy) (models/ullava.py:335-434).  The hidden
    states used are those of the last no-cache generation step, i.e. a forward over sequences[:, :-1]
    (SURVEY 3.2); row t is selected iff output_ids[t+1] is [SEG]/[LOC]."""
    llm_sd = _sub(sd, "llm.")
    seq, last_h = greedy_generate(llm_sd, cfg["llm"], input_ids, images, None, max_new_tokens, eos_token_id)
    seg_mask = seq[:, 1:] == cfg["seg_token_idx"]
    loc_mask = seq[:, 1:] == cfg["loc_token_idx"]
    seg_emb = _gather_tokens(_proj_mlp(sd, "seg_projector.", last_h), seg_mask)
    loc_emb = _gather_tokens(_proj_mlp(sd, "det_projector.", last_h), loc_mask)
    B = images_sam.shape[0]
    image_embeddings = torch.cat([sam_image_encoder(sd, cfg["sam"], images_sam[i].unsqueeze(0)) for i in range(B)], 0)
    pred_masks, _ = _decode_masks(sd, cfg, image_embeddings, seg_emb, resize_list, raw_size_list)
    pred_boxes = [det_decoder(sd, e) for e in loc_emb]
    return seq, pred_masks, pred_boxes
