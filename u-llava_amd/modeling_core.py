"""UllavaCoreForCausalLM on MI355X: CLIP ViT-L -> vision_projector -> LLaMA -> lm_head.

Host-side mirror of reference `models/ullava_core.py:78-395` (same constructor, attribute names, state-dict keys,
`forward` / `encode_image` / `encode_video` / `embed_images_videos` / `prepare_inputs_for_generation` / `generate`
signatures and outputs).  The module tree exists to carry the reference's parameter names; no nn.Module.forward
of a leaf is ever used: every arithmetic op is a HIP kernel reached through `ops.py` -> C-ABI.

MI355X-specific host design:
  * weights are re-laid out once (`pack_weights`): q|k|v fused into one [3D, D] matrix (one GEMM, one pass over
    the activations), gate/up interleaved in 16-row groups so SwiGLU is a register-level GEMM epilogue, the CLIP
    patch conv flattened and K-padded to a multiple of 64 for the MFMA GEMM;
  * the residual stream is a flat [B*S, D] bf16 matrix; q/k are consumed in place from the fused QKV buffer by
    strides, V is written once as V^T (K-contiguous for the P*V MFMA);
  * the per-sample python splice loop of the reference (with its device->host syncs) is one gather kernel.
"""
from typing import List, Optional

import weakref

import torch
import torch.nn as nn

from . import _lib
from . import autograd_ops as A
from . import ops
from .configuration import UllavaCoreConfig

BF16 = torch.bfloat16


# ----------------------------------------------------------------------------------------------------------
# parameter holders (names match transformers' modules; forward intentionally absent)
# ----------------------------------------------------------------------------------------------------------
def _param(*shape, device=None, dtype=BF16):
    return nn.Parameter(torch.empty(*shape, device=device, dtype=dtype), requires_grad=False)



def _clear_transposes():
    from .autograd_ops import clear_transpose_cache
    clear_transpose_cache()


_ALIAS_WARNED = [False]


def _warn_alias_fallback(why: str) -> None:
    """one line, once per process: the training path fell back from aliased q|k|v / gate|up buffers to a per-step torch.cat (correct, slower)."""
    if not _ALIAS_WARNED[0]:
        _ALIAS_WARNED[0] = True
        import warnings
        warnings.warn("u-llava_amd: training forward concatenates q|k|v / gate|up under autograd on every step instead of aliasing them: " + why +
                      ".  Results are identical; call model._apply(lambda t: t) (or .to(device)) after taking ownership back to re-enable aliasing.",
                      RuntimeWarning, stacklevel=3)


def _dealias_state_dict(module, state_dict, prefix, local_metadata):
    """state-dict hook: the training path makes q|k|v and gate|up row slices of one buffer each (_alias_pack); a state dict must hold
    tensors that own their storage (safetensors refuses shared memory; HF Trainer._save writes state_dict() as it is)."""
    for k in list(state_dict.keys()):
        v = state_dict[k]
        if isinstance(v, nn.Parameter):
            continue                 # state_dict(keep_vars=True): the caller asked for the parameters themselves, aliased or not
        if k.startswith(prefix) and isinstance(v, torch.Tensor) and v.untyped_storage().nbytes() > v.numel() * v.element_size() + 64:
            state_dict[k] = v.clone()
    return state_dict


class Linear(nn.Linear):
    """An `nn.Linear` by type and by state-dict layout -- `isinstance(m, torch.nn.Linear)` is how the reference picks its LoRA targets
    (train_ullava.py:88-113 `find_linear_layers`) and how HF utilities classify parameters -- whose storage is left UNINITIALISED at
    construction (nn.Linear.__init__ would run a kaiming init over 7 B parameters on the host; weights arrive from a checkpoint or a
    seeded generator) and whose `forward` is the HIP GEMM.  The model code never calls it (the projections are operands of fused launches);
    it exists for callers that treat the module as a layer."""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=BF16):
        nn.Module.__init__(self)
        self.in_features, self.out_features = in_features, out_features
        self.weight = _param(out_features, in_features, device=device, dtype=dtype)
        if bias:
            self.bias = _param(out_features, device=device, dtype=dtype)
        else:
            self.register_parameter("bias", None)

    def reset_parameters(self) -> None:
        raise RuntimeError("u-llava_amd: parameters are loaded, never re-initialised in place")

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return A.linear(x, self.weight, self.bias)
        return ops.linear(x, self.weight, self.bias)


class Embedding(nn.Module):
    def __init__(self, n, d, device=None, dtype=BF16):
        super().__init__()
        self.weight = _param(n, d, device=device, dtype=dtype)


class Norm(nn.Module):
    def __init__(self, d, bias=True, eps=1e-5, device=None, dtype=BF16):
        super().__init__()
        self.eps = eps
        self.weight = _param(d, device=device, dtype=dtype)
        self.bias = _param(d, device=device, dtype=dtype) if bias else None


class Conv2dHolder(nn.Module):
    def __init__(self, cin, cout, k, bias, device=None, dtype=BF16):
        super().__init__()
        self.weight = _param(cout, cin, k, k, device=device, dtype=dtype)
        self.bias = _param(cout, device=device, dtype=dtype) if bias else None


class _Holder(nn.Module):
    pass


def _llama_layer(cfg, device, dtype):
    m = _Holder()
    D, I = cfg.hidden_size, cfg.intermediate_size
    m.self_attn = _Holder()
    for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
        setattr(m.self_attn, n, Linear(D, D, bias=False, device=device, dtype=dtype))
    m.mlp = _Holder()
    m.mlp.gate_proj = Linear(D, I, bias=False, device=device, dtype=dtype)
    m.mlp.up_proj = Linear(D, I, bias=False, device=device, dtype=dtype)
    m.mlp.down_proj = Linear(I, D, bias=False, device=device, dtype=dtype)
    m.input_layernorm = Norm(D, bias=False, eps=cfg.rms_norm_eps, device=device, dtype=dtype)
    m.post_attention_layernorm = Norm(D, bias=False, eps=cfg.rms_norm_eps, device=device, dtype=dtype)
    return m


def _clip_tower(vc, device, dtype):
    D, I = vc.hidden_size, vc.intermediate_size
    m = _Holder()
    m.embeddings = _Holder()
    m.embeddings.class_embedding = _param(D, device=device, dtype=dtype)
    m.embeddings.patch_embedding = Conv2dHolder(vc.num_channels, D, vc.patch_size, bias=False, device=device, dtype=dtype)
    m.embeddings.position_embedding = Embedding((vc.image_size // vc.patch_size) ** 2 + 1, D, device=device, dtype=dtype)
    m.pre_layrnorm = Norm(D, eps=vc.layer_norm_eps, device=device, dtype=dtype)
    m.encoder = _Holder()
    layers = []
    for _ in range(vc.num_hidden_layers):
        l = _Holder()
        l.self_attn = _Holder()
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            setattr(l.self_attn, n, Linear(D, D, device=device, dtype=dtype))
        l.layer_norm1 = Norm(D, eps=vc.layer_norm_eps, device=device, dtype=dtype)
        l.mlp = _Holder()
        l.mlp.fc1 = Linear(D, I, device=device, dtype=dtype)
        l.mlp.fc2 = Linear(I, D, device=device, dtype=dtype)
        l.layer_norm2 = Norm(D, eps=vc.layer_norm_eps, device=device, dtype=dtype)
        layers.append(l)
    m.encoder.layers = nn.ModuleList(layers)
    m.post_layernorm = Norm(D, eps=vc.layer_norm_eps, device=device, dtype=dtype)
    return m


class CausalLMOutputWithPast(dict):
    """Attribute + key + index access, like transformers.modeling_outputs.CausalLMOutputWithPast."""
    _order = ("loss", "logits", "past_key_values", "hidden_states", "attentions")

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            return [self.get(n) for n in self._order if self.get(n) is not None][k]
        return dict.__getitem__(self, k)


class GenerateOutput(dict):
    __getattr__ = dict.__getitem__


class KVCache:
    """Per-layer K [B,H,Smax,hd] (post-RoPE) and V^T [B,H,hd,Smax] in the attention kernel's key-permuted layout, plus the
    last-layer hidden states of every position seen so far (what `evaluate()` reads for [SEG]/[LOC] rows).  Truthy once a
    prefill has been stored, like a non-empty HF past_key_values tuple."""

    def __init__(self, n_layers, B, H, hd, smax, device, dtype=BF16):
        self.smax = ((smax + 63) // 64) * 64
        self.k = [torch.empty(B, H, self.smax, hd, device=device, dtype=dtype) for _ in range(n_layers)]
        self.vt = [torch.zeros(B, H, hd, self.smax, device=device, dtype=dtype) for _ in range(n_layers)]
        self.length = 0
        self.last_hidden = []

    def __bool__(self):
        return self.length > 0

    def c_ptrs(self):
        """(void* array of the K buffers, void* array of the V^T buffers) for the coarse decode entry; built once per cache."""
        if getattr(self, "_c_ptrs", None) is None:
            import ctypes
            self._c_ptrs = ((ctypes.c_void_p * len(self.k))(*[t.data_ptr() for t in self.k]), (ctypes.c_void_p * len(self.vt))(*[t.data_ptr() for t in self.vt]))
        return self._c_ptrs

    def __len__(self):
        return len(self.k)

    def to_legacy_cache(self):
        """The HF view of this cache (what `outputs.past_key_values` is in the reference under transformers 4.29.1, models/ullava_core.py:349-355):
        a tuple of per-layer (key, value) pairs, each [B, H, length, hd] with post-RoPE keys.  Pure data movement (a slice of the K buffer;
        the key-permuted V^T image gathered back into natural key order and transposed); `KVCache.from_hf` is the inverse."""
        n = self.length
        idx = ops.vt_unpermute_index(self.smax).to(self.k[0].device)[:n]
        return tuple((k[:, :, :n].contiguous(), vt[..., idx].transpose(-1, -2).contiguous()) for k, vt in zip(self.k, self.vt))

    def __iter__(self):
        return iter(self.to_legacy_cache())

    def __getitem__(self, i):
        """layer i's (key, value) pair -- so that code written against a tuple of pairs (`past_key_values[0][0].shape[2]`, the idiom of
        4.29-era callers) reads this object too."""
        n = self.length
        idx = ops.vt_unpermute_index(self.smax).to(self.k[i].device)[:n]
        return self.k[i][:, :, :n], self.vt[i][..., idx].transpose(-1, -2)

    @classmethod
    def from_hf(cls, past, headroom: int = 512):
        """A caller-supplied HF `past_key_values` -> KVCache (reference models/ullava_core.py:279-292 takes `past_key_values:
        Optional[List[torch.FloatTensor]]`): the legacy tuple of per-layer (key, value) pairs, each [B, H, S, hd] with post-RoPE keys, or a
        transformers Cache object exposing `key_cache` / `value_cache` lists (or `.layers[i].keys / .values`).  Keys are copied into the
        K buffers, values go through `ull_transpose_v` into the permuted V^T layout; `last_hidden` starts empty (hidden states of the
        cached positions are not part of an HF cache)."""
        if hasattr(past, "key_cache") and hasattr(past, "value_cache"):
            pairs = list(zip(past.key_cache, past.value_cache))
        elif hasattr(past, "layers"):
            pairs = [(l.keys, l.values) for l in past.layers]
        else:
            pairs = [(kv[0], kv[1]) for kv in past]
        if not pairs:
            raise ValueError("empty past_key_values")
        k0 = pairs[0][0]
        if k0.dim() != 4:
            raise ValueError("past_key_values entries must be [batch, heads, seq, head_dim] tensors")
        B, H, S, hd = k0.shape
        c = cls(len(pairs), B, H, hd, S + headroom, k0.device, k0.dtype)
        for li, (k, v) in enumerate(pairs):
            if k.shape != (B, H, S, hd) or v.shape != (B, H, S, hd):
                raise ValueError("past_key_values layers disagree in shape")
            c.k[li][:, :, :S].copy_(k)
            vr = v.permute(0, 2, 1, 3).contiguous()                              # [B, S, H, hd]: heads contiguous per token (data movement)
            ops.transpose_v(vr, S * H * hd, H * hd, B, S, H, hd, pitch=c.smax, out=c.vt[li])
        c.length = S
        return c

    @staticmethod
    def vt_slot(pos: int) -> int:
        """column of key `pos` inside V^T (32-key blocks stored as slot 8g+4a+r <- key 16a+4g+r, see transpose_v)."""
        w = pos & 31
        return (pos & ~31) + 8 * ((w >> 2) & 3) + 4 * (w >> 4) + (w & 3)


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I, K] x2 -> [2I, K] with rows [gate 16g..16g+15 | up 16g..16g+15] per group g (ULL_EPI_SWIGLU layout)."""
    I, K = gate.shape
    assert I % 16 == 0
    return torch.stack((gate.view(I // 16, 16, K), up.view(I // 16, 16, K)), dim=1).reshape(2 * I, K).contiguous()


def no_repeat_ngram_banned_tokens(rows, ngram_size: int):
    """transformers' NoRepeatNGramLogitsProcessor rule (generation/logits_process.py `_calc_banned_ngram_tokens`; reference
    models/ullava.py:360 forwards `no_repeat_ngram_size` to HF generate): for every row (list of ids so far, prompt included), the tokens
    that followed an earlier occurrence of the row's last ngram_size - 1 tokens.  Host-side integer bookkeeping, like HF's."""
    out = []
    for toks in rows:
        cur = len(toks)
        if ngram_size <= 0 or cur + 1 < ngram_size:
            out.append([])
            continue
        prefix = tuple(toks[cur + 1 - ngram_size:cur])
        banned = []
        for i in range(cur - ngram_size + 1):
            if tuple(toks[i:i + ngram_size - 1]) == prefix:
                banned.append(toks[i + ngram_size - 1])
        out.append(banned)
    return out


def sampling_probs(logits: torch.Tensor, temperature: float, top_k: Optional[int] = 50, top_p: Optional[float] = None) -> torch.Tensor:
    """The distribution HF `GenerationMixin._sample` draws from, in its own arithmetic and order (transformers generation/logits_process.py:
    TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper on fp32 scores, then softmax): the reference's callers pass `temperature`
    (and optionally `top_p`) and inherit `top_k = 50` from GenerationConfig's defaults (models/ullava.py:350-361, inference_ullava_core.py:73-80).
    One `torch.multinomial(probs, 1)` per step on these probabilities consumes the RNG exactly as HF does, so a seeded run draws HF's tokens
    from the same logits.  Integer / fp32 bookkeeping on [B, V] scores (the logits themselves come from the HIP lm_head)."""
    scores = logits.float()
    if temperature is not None and temperature != 1.0:
        scores = scores / temperature
    if top_k is not None and top_k > 0:
        k = min(int(top_k), scores.size(-1))
        scores = scores.masked_fill(scores < torch.topk(scores, k)[0][..., -1, None], float("-inf"))
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(scores, descending=False)
        cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cumulative <= (1 - top_p)
        remove[..., -1:] = 0                                     # min_tokens_to_keep = 1
        scores = scores.masked_fill(remove.scatter(1, sorted_indices, remove), float("-inf"))
    return torch.softmax(scores, dim=-1)


# ----------------------------------------------------------------------------------------------------------
class UllavaCoreForCausalLM(nn.Module):
    config_class = UllavaCoreConfig

    def __init__(self, config: UllavaCoreConfig, device=None, dtype=BF16):
        super().__init__()
        if dtype not in (BF16, torch.float16, torch.float32):
            raise NotImplementedError("the MI355X path has bf16 (reference configs: bf16: true), fp16 and fp32 (inference_ullava.py --dtype "
                                      "fp16 / fp32) kernel builds")
        # (fp32: csrc/f32.hip -- plain kernels for the inference path, no fused / tiled fast paths, no backward)
        self.config = config
        D = config.hidden_size
        self.model = _Holder()
        self.model.embed_tokens = Embedding(config.vocab_size, D, device=device, dtype=dtype)
        self.model.layers = nn.ModuleList([_llama_layer(config, device, dtype) for _ in range(config.num_hidden_layers)])
        self.model.norm = Norm(D, bias=False, eps=config.rms_norm_eps, device=device, dtype=dtype)
        self.lm_head = Linear(D, config.vocab_size, bias=False, device=device, dtype=dtype)
        self.vision_encoder = _clip_tower(config.vision_config, device, dtype)
        self.vision_projector = self.build_vision_projector(config.vision_config.hidden_size, D, config.projector_type, device, dtype)
        self.vision_hidden_layer = config.vision_hidden_layer
        self.projector_from_scratch = config.projector_from_scratch
        self.mm_token_ids = config.mm_token_ids
        self.strict_checks = True          # reproduce the reference's start/end-count assert (one tiny D2H read)
        self._packed = None
        self._inv_freq = None
        self._register_state_dict_hook(_dealias_state_dict)

    # -- construction helpers ----------------------------------------------------------------------------
    @staticmethod
    def build_vision_projector(in_dim, hidden_dim, name="mlp", device=None, dtype=BF16):
        """reference models/ullava_core.py:117-129."""
        if name == "mlp":
            return Linear(in_dim, hidden_dim, device=device, dtype=dtype)
        if name == "mlp2x":
            return nn.Sequential(Linear(in_dim, hidden_dim, device=device, dtype=dtype), nn.GELU(),
                                 Linear(hidden_dim, hidden_dim, device=device, dtype=dtype))
        raise NotImplementedError

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    @property
    def device(self):
        return self.lm_head.weight.device

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .half() / .bfloat16(): the MI355X re-layouts (fused QKV, interleaved gate/up, tile-major copies) are
        derived from the parameters and are rebuilt from the moved / cast ones on the next forward."""
        out = super()._apply(fn, *args, **kwargs)
        self._packed = None
        _clear_transposes()          # cached W^T copies of the training path describe the old weights
        self._inv_freq = None
        self._pos_cache = None
        self._reset_alias_slots()    # every parameter is a fresh tensor: the training path may alias them anew
        return out

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def resize_token_embeddings(self, new_num_tokens=None, pad_to_multiple_of=None):
        """PreTrainedModel.resize_token_embeddings as the reference's callers use it (train_ullava.py:212, models/tools.py:45,72,
        102,108): grow (or shrink) embed_tokens and lm_head to `new_num_tokens` rows, old rows kept, new rows ~ N(0, 0.02)
        (LlamaPreTrainedModel._init_weights; the callers then overwrite them with the mean embedding), config.vocab_size updated."""
        emb = self.model.embed_tokens
        if new_num_tokens is None:
            return emb
        if pad_to_multiple_of:
            new_num_tokens = -(-new_num_tokens // pad_to_multiple_of) * pad_to_multiple_of
        old = emb.weight.shape[0]
        if new_num_tokens != old:
            keep = min(old, new_num_tokens)
            for holder in (emb, self.lm_head):
                w = holder.weight
                nw = torch.empty(new_num_tokens, w.shape[1], device=w.device, dtype=w.dtype)
                nw.normal_(0.0, 0.02)
                nw[:keep] = w.data[:keep]
                holder.weight = nn.Parameter(nw, requires_grad=w.requires_grad)
            self.lm_head.out_features = new_num_tokens
            self.config.vocab_size = new_num_tokens
            self._packed = None
            _clear_transposes()          # cached W^T copies of the training path describe the old weights
        return emb

    def get_output_embeddings(self):
        return self.lm_head

    def init_mm_tokens(self, tokenizer, mm_tokens):
        ids = {k: tokenizer.convert_tokens_to_ids(v) for k, v in mm_tokens.items()}
        self.config.mm_token_ids = ids
        self.mm_token_ids = ids

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, device=None, **kwargs):
        """reference: inference_ullava_core.py:35 / train_ullava_core.py:94 (PreTrainedModel.from_pretrained on a local directory)."""
        from .checkpoint import core_from_pretrained
        return core_from_pretrained(cls, pretrained_model_name_or_path, torch_dtype, device, **kwargs)

    def save_pretrained(self, save_directory, **kwargs):
        from .checkpoint import save_pretrained
        return save_pretrained(self, save_directory, **kwargs)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        # transformers 4.29.1 checkpoints nest the CLIP tower under `vision_encoder.vision_model.`
        sd = {k.replace("vision_encoder.vision_model.", "vision_encoder."): v for k, v in state_dict.items()}
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids") and "rotary_emb.inv_freq" not in k}
        self._packed = None
        _clear_transposes()          # cached W^T copies of the training path describe the old weights
        return super().load_state_dict(sd, strict=strict, assign=assign)

    # -- weight re-layout ----------------------------------------------------------------------------------
    def pack_weights(self, free_originals: bool = False):
        """Build the MI355X layouts.  Call again after changing parameters."""
        cfg = self.config
        pk = {"llama": [], "clip": []}
        lora = getattr(self, "_lora", None)

        def eff(lin):
            """the projection the inference kernels see: with an (un-merged) LoRA adapter attached, W + (alpha / r) B A rounded once --
            PeftModel.merge_adapter's arithmetic -- WITHOUT touching the parameters (generate() / evaluate() under no_grad while
            the adapters train: train_ullava.py's evaluation loop; the pack is rebuilt when the adapters change, see _pk)."""
            if lora is None or not hasattr(lin, "lora_A"):
                return lin.weight
            from .checkpoint import lora_merged_weight
            return lora_merged_weight(lin.weight, lin.lora_A.weight, lin.lora_B.weight, lora["lora_alpha"] / lora["r"])
        for l in self.model.layers:
            a, m = l.self_attn, l.mlp
            pk["llama"].append(dict(
                w_qkv=torch.cat([eff(a.q_proj), eff(a.k_proj), eff(a.v_proj)], dim=0).contiguous(),
                w_o=a.o_proj.weight, w_gu=interleave_gate_up(m.gate_proj.weight, m.up_proj.weight), w_down=m.down_proj.weight,
                ln1=l.input_layernorm.weight, ln2=l.post_attention_layernorm.weight))
        for l in self.vision_encoder.encoder.layers:
            a = l.self_attn
            pk["clip"].append(dict(
                w_qkv=torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], dim=0).contiguous(),
                b_qkv=torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], dim=0).contiguous(),
                w_out=a.out_proj.weight, b_out=a.out_proj.bias, fc1=l.mlp.fc1, fc2=l.mlp.fc2, ln1=l.layer_norm1, ln2=l.layer_norm2))
        vc = cfg.vision_config
        w = self.vision_encoder.embeddings.patch_embedding.weight
        K = vc.num_channels * vc.patch_size * vc.patch_size
        Kp = ((K + 63) // 64) * 64
        wp = torch.zeros(w.shape[0], Kp, device=w.device, dtype=w.dtype)
        wp[:, :K] = w.reshape(w.shape[0], K)
        pk["patch_w"], pk["patch_Kp"] = wp, Kp
        # fused patchify (A tiles DMA'd from the pixels): (c, ky) segments of 16, see ops.pack_patch_weight
        pk["patch_wp"] = ops.pack_patch_weight(w) if (vc.patch_size <= 16 and vc.patch_size % 2 == 0 and vc.image_size >= 16 and
                                                      w.dtype != torch.float32) else None
        # tile-major copies for the prefill-shape GEMM (+13.5 GB for LLaMA-7B; sized for 288 GB of HBM).  The row-major
        # tensors stay: they feed the decode GEMV, which streams whole rows.
        for d in pk["llama"]:
            for k in ("w_qkv", "w_o", "w_gu", "w_down"):
                ops.register_tiled(d[k])
        for d in pk["clip"]:
            for t in (d["w_qkv"], d["w_out"], d["fc1"].weight, d["fc2"].weight):
                ops.register_tiled(t)
        ops.register_tiled(self.lm_head.weight)
        pk["lora_versions"] = self._lora_versions()
        pk["llama_versions"], pk["clip_versions"] = self._llama_versions(), self._clip_versions()
        pk["freed"] = bool(free_originals)
        self._packed = pk
        if free_originals:
            for l in self.model.layers:
                for mod in (l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj, l.mlp.gate_proj, l.mlp.up_proj):
                    mod.weight.data = torch.empty(0, device=mod.weight.device, dtype=mod.weight.dtype)
        return self

    # The version tuples are read on EVERY forward (a decode step too): walking the module tree for them cost 0.5 ms of host time per step
    # (nn.Module.__getattr__ is Python).  The tensors are collected once per pack -- whatever re-binds a parameter (`_apply`, load_state_dict,
    # add_lora / merge_lora) also drops the pack, and with it these lists.
    def _ver_tensors(self, which: str):
        cache = self.__dict__.setdefault("_ver_cache", {})
        pk = self._packed
        if cache.get("pack") is not pk or pk is None:
            cache.clear()
            cache["pack"] = pk
        lst = cache.get(which)
        if lst is None:
            if which == "lora":
                lst = [p_ for l in self.model.layers for t in ("q_proj", "k_proj", "v_proj") if hasattr(getattr(l.self_attn, t), "lora_A")
                       for p_ in (getattr(l.self_attn, t).lora_A.weight, getattr(l.self_attn, t).lora_B.weight)]
            elif which == "llama":
                lst = [getattr(h, n).weight for l in self.model.layers
                       for h, ns in ((l.self_attn, ("q_proj", "k_proj", "v_proj")), (l.mlp, ("gate_proj", "up_proj"))) for n in ns]
            else:
                ve = self.vision_encoder
                lst = [ve.embeddings.patch_embedding.weight] + [p_ for l in ve.encoder.layers for n in ("q_proj", "k_proj", "v_proj")
                                                                  for p_ in (getattr(l.self_attn, n).weight, getattr(l.self_attn, n).bias)]
            if pk is not None:
                cache[which] = lst
        return lst

    def _lora_versions(self):
        if getattr(self, "_lora", None) is None:
            return None
        return tuple(p_._version for p_ in self._ver_tensors("lora"))

    def _llama_versions(self):
        """version counters of the LLaMA weights the packs hold COPIES of (q|k|v concatenated, gate|up interleaved)."""
        return tuple(p_._version for p_ in self._ver_tensors("llama"))

    def _clip_versions(self):
        return tuple(p_._version for p_ in self._ver_tensors("clip"))

    def _pk(self, for_llama: bool = False):
        """The packed weights, re-made when a parameter they were copied from has been written since (`._version`: an optimizer step's
        in-place update, load_state_dict, merge_lora ...).  for_llama (the inference LLaMA path) also checks the LLaMA packs: the q|k|v
        concatenations / gate|up interleaves of the base weights, and with un-merged LoRA adapters attached the adapters too (the packs
        then hold W + s B A).  The CLIP tower asks without that check: a TRAINING forward needs only the CLIP packs (its LLaMA half reads
        the parameters through _alias_pack), and re-packing 13.5 GB of LLaMA weights on every step because the parameters moved cost
        94 ms of device copies per step (profile of bench.py --workload train --train-config lora)."""
        pk = self._packed
        if pk is None:
            return self.pack_weights()._packed
        if pk.get("freed"):
            return pk                        # pack_weights(free_originals=True): the packs are the only copy, nothing to compare with
        if pk["clip_versions"] != self._clip_versions() or (for_llama and (
                pk["llama_versions"] != self._llama_versions() or pk.get("lora_versions") != self._lora_versions())):
            self.pack_weights()
        return self._packed

    # -- CLIP ----------------------------------------------------------------------------------------------
    def _selected_layer_count(self) -> int:
        L = self.config.vision_config.num_hidden_layers
        idx = self.vision_hidden_layer if self.vision_hidden_layer >= 0 else L + 1 + self.vision_hidden_layer
        if not 0 <= idx <= L:
            raise ValueError("vision_hidden_layer out of range")
        return idx          # hidden_states[idx] = output after `idx` encoder layers

    def _clip_hidden(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """hidden_states[vision_hidden_layer] of CLIPVisionModel, shape [n, 1 + patches, Dv] (CLS row kept).
        Layers past the selected one (and post_layernorm) are dead compute in the reference and are not run."""
        pk = self._pk()
        vc = self.config.vision_config
        ve = self.vision_encoder
        n = pixel_values.shape[0]
        if pixel_values.shape[-1] != vc.image_size or pixel_values.shape[-2] != vc.image_size:
            raise ValueError(f"Input image size ({pixel_values.shape[-2]}*{pixel_values.shape[-1]}) doesn't match model "
                             f"({vc.image_size}*{vc.image_size}).")
        x = pixel_values.to(self.dtype).contiguous()
        P = (vc.image_size // vc.patch_size) ** 2
        Dv, H = vc.hidden_size, vc.num_attention_heads
        hd = Dv // H
        if pk["patch_wp"] is not None:
            patches = ops.patchify(x, pk["patch_wp"], vc.patch_size)
        else:
            patches = ops.linear(ops.im2col(x, vc.patch_size, pk["patch_Kp"]), pk["patch_w"])
        S = P + 1
        h = ops.clip_embed_ln(patches, ve.embeddings.class_embedding, ve.embeddings.position_embedding.weight,
                              ve.pre_layrnorm.weight, ve.pre_layrnorm.bias, n, S, vc.layer_norm_eps).view(n * S, Dv)
        nsel = self._selected_layer_count()
        I = vc.intermediate_size
        coarse = ops.coarse_ok() and nsel > 0 and hd == 64 and n * S > 16 and 16 < S <= 704 and Dv % 64 == 0 and I % 64 == 0 and \
            h.dtype != torch.float32
        if coarse:
            # one C call for the tower's layers (csrc/layers.hip): the same launches as the loop below, bit-identical results
            stack = pk.get("_c_clip")
            if stack is None:
                stack = pk["_c_clip"] = ops.LayerStack(_lib.ClipLayer, [dict(ln1_w=w["ln1"].weight, ln1_b=w["ln1"].bias, ln2_w=w["ln2"].weight,
                                                                              ln2_b=w["ln2"].bias, qkv=(w["w_qkv"], w["b_qkv"]), out=(w["w_out"], w["b_out"]),
                                                                              fc1=(w["fc1"].weight, w["fc1"].bias), fc2=(w["fc2"].weight, w["fc2"].bias))
                                                                         for w in pk["clip"]])
            ops.clip_layers(stack, nsel, h, n, S, H, hd, I, vc.layer_norm_eps)
        for li in range(0 if coarse else nsel):
            w = pk["clip"][li]
            y = ops.layernorm(h, w["ln1"].weight, w["ln1"].bias, vc.layer_norm_eps)
            qkv = ops.linear(y, w["w_qkv"], w["b_qkv"])
            att = torch.empty(n * S, Dv, device=h.device, dtype=h.dtype)
            st = (S * 3 * Dv, hd, 3 * Dv)                       # V goes in as rows of the fused q|k|v buffer: no V^T pass
            ops.attention(qkv, qkv[:, Dv:], qkv[:, 2 * Dv:], att, n, H, S, S, hd, st, st, (S * Dv, hd, Dv), None, causal=False, scale_mode=1,
                          scale=hd ** -0.5, v_strides=st)
            h = ops.linear(att, w["w_out"], w["b_out"], residual=h)
            y = ops.layernorm(h, w["ln2"].weight, w["ln2"].bias, vc.layer_norm_eps)
            f = ops.linear(y, w["fc1"].weight, w["fc1"].bias, act="quick_gelu")
            h = ops.linear(f, w["fc2"].weight, w["fc2"].bias, residual=h)
        return h.view(n, S, Dv)

    def encode_image(self, image_tensors: torch.Tensor) -> torch.Tensor:
        """reference :146-158 -> [bs, num_patches, Dv] (CLS removed)."""
        return self._clip_hidden(image_tensors)[:, 1:]

    def encode_video(self, video_clip_tensors: torch.Tensor) -> torch.Tensor:
        """reference :160-180 -> [bs, n_frm + num_patches, Dv]."""
        b, c, t, hh, ww = video_clip_tensors.shape
        frames = video_clip_tensors.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
        h = self._clip_hidden(frames)                       # [b*t, P+1, Dv]
        return ops.video_pool(h, b, t, h.shape[1] - 1, tok_pitch=h.shape[1], tok_off=1)

    def _project(self, x: torch.Tensor) -> torch.Tensor:
        vp = self.vision_projector
        if self._training_graph():
            if not isinstance(vp, Linear):
                raise NotImplementedError("training path: projector_type 'mlp' (the configured type, configs/train/ullava_core.yaml:5)")
            return A.linear(x, vp.weight, vp.bias)
        if isinstance(vp, Linear):
            return ops.linear(x, vp.weight, vp.bias)
        return ops.linear(ops.linear(x, vp[0].weight, vp[0].bias, act="gelu"), vp[2].weight, vp[2].bias)

    # -- LoRA (train_ullava.py:219-237: get_peft_model(model.llm, LoraConfig(r, lora_alpha, target_modules, lora_dropout))) ---------
    def add_lora(self, r: int, lora_alpha: float = 16.0, lora_dropout: float = 0.0, target_modules=("q_proj", "v_proj")):
        """Attach LoRA adapters to the attention projections named in `target_modules` (the reference's default: q_proj, v_proj) of every
        LLaMA layer: `lora_A.weight` [r, in] (kaiming-uniform, PEFT's init), `lora_B.weight` [out, r] (zeros), the base weights and
        every other language-model weight frozen, the adapters trainable -- what get_peft_model leaves.  The forward adds
        (lora_alpha / r) * B(A(dropout(x))) to the projection as PEFT's Linear does; save_pretrained writes the adapter in PEFT's
        file layout and from_pretrained merges such files (checkpoint.merge_lora_adapter)."""
        import math
        targets = tuple(target_modules)
        if not targets or any(t not in ("q_proj", "k_proj", "v_proj") for t in targets):
            raise NotImplementedError("LoRA targets on this path: q_proj, k_proj, v_proj (the reference's configuration: q_proj, v_proj)")
        for p_ in self.model.parameters():
            p_.requires_grad = False
        for p_ in self.lm_head.parameters():
            p_.requires_grad = False
        for l in self.model.layers:
            for t in targets:
                lin = getattr(l.self_attn, t)
                dev, dt = lin.weight.device, lin.weight.dtype
                lin.lora_A, lin.lora_B = _Holder(), _Holder()
                a = torch.empty(r, lin.in_features, dtype=torch.float32)
                torch.nn.init.kaiming_uniform_(a, a=math.sqrt(5))
                lin.lora_A.weight = nn.Parameter(a.to(dt).to(dev), requires_grad=True)
                lin.lora_B.weight = nn.Parameter(torch.zeros(lin.out_features, r, dtype=dt, device=dev), requires_grad=True)
        self._lora = {"r": int(r), "lora_alpha": float(lora_alpha), "lora_dropout": float(lora_dropout), "target_modules": targets}
        self._packed = None
        return self

    def _lora_delta_operands(self, attn):
        """(A_cat [n*r, D], B_cat [3D, n*r] already scaled by lora_alpha / r) for the fused q|k|v projection of one layer: B_cat is block
        structured, so z = x A_cat^T followed by z B_cat^T adds each target's own B(A(x)) to its own third of the q|k|v row."""
        cfg = self._lora
        r, s_ = cfg["r"], cfg["lora_alpha"] / cfg["r"]
        D = self.config.hidden_size
        As, Bs = [], []
        present = [t for t in ("q_proj", "k_proj", "v_proj") if hasattr(getattr(attn, t), "lora_A")]
        for i, t in enumerate(present):
            lin = getattr(attn, t)
            As.append(lin.lora_A.weight)
            blk = torch.zeros(3 * D, r, device=lin.weight.device, dtype=lin.weight.dtype)
            off = ("q_proj", "k_proj", "v_proj").index(t) * D
            blk = torch.cat([blk[:off], lin.lora_B.weight * s_, blk[off + D:]], dim=0)      # (weight preparation; s_ = 2 in the reference's config)
            Bs.append(blk)
        # the GEMM kernels take K in 64-wide tiles: the rank axis is zero-padded here, once per operand pair, so that neither the forward
        # nor the backward GEMMs pad their activations (zero rows of A_cat / zero columns of B_cat add exact zeros)
        nr = len(present) * r
        pad = (-nr) % 64
        if pad:
            w0 = As[0]
            As.append(torch.zeros(pad, D, device=w0.device, dtype=w0.dtype))
            Bs.append(torch.zeros(3 * D, pad, device=w0.device, dtype=w0.dtype))
        return torch.cat(As, dim=0), torch.cat(Bs, dim=1)

    def merge_lora(self):
        """Fold the adapters into the base weights (PeftModel.merge_and_unload) and drop them: the inference kernels read plain weights."""
        cfg = getattr(self, "_lora", None)
        if cfg is None:
            return self
        s_ = cfg["lora_alpha"] / cfg["r"]
        from .checkpoint import lora_merged_weight
        with torch.no_grad():
            for l in self.model.layers:
                for t in cfg["target_modules"]:
                    lin = getattr(l.self_attn, t)
                    lin.weight.copy_(lora_merged_weight(lin.weight, lin.lora_A.weight, lin.lora_B.weight, s_))
                    del lin.lora_A, lin.lora_B
        self._lora = None
        self._packed = None
        _clear_transposes()
        return self

    # -- training path ---------------------------------------------------------------------------------------
    def _training_graph(self) -> bool:
        """True when this call must build an autograd graph: gradients enabled and some parameter of the language model or the
        projector asks for one (the reference freezes / unfreezes by `requires_grad`, train_ullava.py:207-261).  The CLIP tower
        is frozen by both training scripts and always runs the inference kernels without a graph."""
        if not torch.is_grad_enabled():
            return False                     # no_grad / eval with adapters attached: inference kernels on a temporarily merged pack (_pk)
        if self.dtype == torch.float32:
            # the fp32 build (csrc/f32.hip) is the INFERENCE path (`--dtype fp32` of inference_ullava.py): there are no fp32 backward kernels, so an
            # fp32 model never builds a graph -- its outputs carry no grad_fn, like the reference's under torch.inference_mode()
            return False
        if getattr(self, "_lora", None) is not None:
            return True
        return any(p.requires_grad for p in self.lm_head.parameters()) or any(p.requires_grad for p in self.model.parameters()) or \
            any(p.requires_grad for p in self.vision_projector.parameters())

    def enable_input_require_grads(self):
        """PreTrainedModel API used by train_ullava.py:214 (makes checkpointed segments differentiable in HF); the HIP training path
        needs nothing here."""
        return None

    def gradient_checkpointing_enable(self, *args, **kwargs):
        """train_ullava.py:215.  Activations are kept (288 GB of HBM per GPU): accepted and recorded, not acted upon."""
        self.config.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.config.gradient_checkpointing = False

    alias_packed_weights = True          # training path: q|k|v and gate|up parameters as row slices of one buffer each (see _alias_pack)

    @staticmethod
    def _alias_pack(holder, names, slot, enabled: bool = True):
        """Make the parameters `names` of `holder` row slices of ONE [sum(N_i), K] buffer (storage shared; values, shapes, names and
        state-dict entries unchanged) and return (that buffer, the parameters).

        The aliasing is established ONCE per holder (and again after .to() / .half(), which re-create every parameter: _apply clears the
        slots) -- and only if every parameter still owns its storage at that moment.  If a parameter is a view into somebody's flat
        buffer already, or the aliasing is found broken later -- a parameter's `.data` no longer points into the buffer -- somebody else has taken
        ownership of the parameter storage (DeepSpeed ZeRO-1/2 rebinds p.data to its flat bit16 partition, FSDP to its flat parameter):
        rebinding `.data` again would detach the parameters from that owner and its updates would never be seen.  The holder is then
        marked and this function returns (None, params): the caller concatenates the live parameters under autograd instead (one
        device copy per step, always correct).  `model.alias_packed_weights = False` turns the aliasing off altogether."""
        ws = [getattr(holder, n).weight for n in names]
        if not enabled or getattr(holder, slot + "_external", False):
            return None, ws
        packed = getattr(holder, slot, None)
        if packed is not None:
            ok = packed.device == ws[0].device and packed.dtype == ws[0].dtype
            o = 0
            for w in ws:
                ok = ok and w.data_ptr() == packed.data_ptr() + o * packed.shape[1] * packed.element_size() and w.is_contiguous()
                o += w.shape[0]
            if ok and o == packed.shape[0]:
                return packed, ws
            object.__setattr__(holder, slot, None)
            object.__setattr__(holder, slot + "_external", True)          # storage re-bound by an external owner: leave it alone
            _warn_alias_fallback("the parameters no longer point into the packed buffer (an external owner re-bound their storage)")
            return None, ws
        # First call on this holder.  A parameter that is already a VIEW into a larger storage has an owner: deepspeed.initialize
        # (ZeRO-1/2: p.data = a slice of the flat bit16 group) or an FSDP wrap (flat parameter) ran before the first training forward --
        # the order HF Trainer uses.  Taking the storage away here would leave the owner updating a buffer nobody reads.
        if any(w.storage_offset() != 0 or w.untyped_storage().nbytes() > w.numel() * w.element_size() + 64 for w in ws):
            object.__setattr__(holder, slot + "_external", True)
            _warn_alias_fallback("a parameter is already a view into a larger storage (DeepSpeed / FSDP flat buffer, load_state_dict(assign=True) "
                                 "from an mmap, or a user-side flat parameter)")
            return None, ws
        with torch.no_grad():
            packed = torch.cat([w.data for w in ws], dim=0).contiguous()
            o = 0
            for w in ws:
                w.data = packed[o:o + w.shape[0]]
                o += w.shape[0]
        object.__setattr__(holder, slot, packed)             # a plain attribute: not a Parameter, not a buffer, not in the state dict
        return packed, ws

    def _reset_alias_slots(self):
        """after _apply (.to / .cuda / .half): where the move re-created the parameters the shared buffers describe dead tensors -- drop them
        (and any `external owner` mark) so that the next training forward aliases the new parameters; a no-op move keeps the aliasing."""
        for l in self.model.layers:
            for h, slot, names in ((l.self_attn, "_qkv_pack", ("q_proj", "k_proj", "v_proj")), (l.mlp, "_gu_pack", ("gate_proj", "up_proj"))):
                packed = h.__dict__.get(slot)
                if packed is not None:
                    ws, o, ok = [getattr(h, n).weight for n in names], 0, True
                    for w in ws:
                        ok = ok and w.device == packed.device and w.dtype == packed.dtype and \
                            w.data_ptr() == packed.data_ptr() + o * packed.shape[1] * packed.element_size()
                        o += w.shape[0]
                    if ok:
                        continue
                for a in (slot, slot + "_external"):
                    if a in h.__dict__:
                        object.__delattr__(h, a)

    def _llama_train(self, inputs_embeds, attention_mask, position_ids, output_hidden_states):
        """LlamaModel.forward with an autograd graph: the same HIP forward kernels (SwiGLU un-fused, RoPE stand-alone, weights
        re-packed under autograd so gradients reach q/k/v/gate/up_proj), HIP backward kernels (autograd_ops.py)."""
        cfg = self.config
        B, S, D = inputs_embeds.shape
        H = cfg.num_attention_heads
        hd = D // H
        dev = inputs_embeds.device
        if position_ids is None:
            pos = torch.arange(S, device=dev, dtype=torch.int64).repeat(B)
        else:
            pos = position_ids.to(torch.int64).expand(B, S).reshape(-1).contiguous()
        key_mask = None if attention_mask is None else attention_mask.to(torch.int32).contiguous()
        inv_freq = self._rope_inv_freq(dev)
        x = inputs_embeds.reshape(B * S, D)
        all_h = []
        for l in self.model.layers:
            if output_hidden_states:
                all_h.append(x.view(B, S, D))
            a, m = l.self_attn, l.mlp
            # q|k|v and gate|up are each ONE buffer whose row slices are the parameters (no per-step torch.cat / interleave of trainable
            # weights, one dW GEMM per pack whose row slices are the parameters' gradients)
            w_qkv, qkv_w = self._alias_pack(a, ("q_proj", "k_proj", "v_proj"), "_qkv_pack", self.alias_packed_weights)
            x, x_res = A.fork(x)                                 # two consumers: the norm and the residual add of o_proj
            h = A.rmsnorm(x, l.input_layernorm.weight, cfg.rms_norm_eps)
            # (no shared buffer -- an external owner holds the parameter storage, see _alias_pack: concatenate under autograd)
            qkv_lin = A.linear_packed(h, w_qkv, *qkv_w) if w_qkv is not None else A.linear(h, torch.cat(qkv_w, dim=0))
            if getattr(self, "_lora", None) is not None:
                # PEFT's Linear.forward: result += lora_B(lora_A(dropout(x))) * scaling -- here as two skinny GEMMs for the whole q|k|v row,
                # the base projection riding along as the second one's residual operand
                a_cat, b_cat = self._lora_delta_operands(a)
                hd_in = A.dropout(h, self._lora["lora_dropout"]) if (self.training and self._lora["lora_dropout"] > 0) else h
                qkv_lin = A.linear(A.linear(hd_in, a_cat), b_cat, residual=qkv_lin)
            qkv = A.rope(qkv_lin, pos, inv_freq, 2 * H, hd)
            att = A.self_attention(qkv, key_mask, B, S, H, hd, True)
            x = A.linear(att, a.o_proj.weight, residual=x_res)
            x, x_res = A.fork(x)
            h = A.rmsnorm(x, l.post_attention_layernorm.weight, cfg.rms_norm_eps)
            w_gu, gu_w = self._alias_pack(m, ("gate_proj", "up_proj"), "_gu_pack", self.alias_packed_weights)
            gu = A.linear_packed(h, w_gu, *gu_w) if w_gu is not None else A.linear(h, torch.cat(gu_w, dim=0))      # columns [gate | up]
            x = A.linear(A.swiglu(gu, halves=True), m.down_proj.weight, residual=x_res)
        x = A.rmsnorm(x, self.model.norm.weight, cfg.rms_norm_eps)
        last = x.view(B, S, D)
        if output_hidden_states:
            all_h.append(last)
        return last, (tuple(all_h) if output_hidden_states else None)

    # -- embedding + splice ----------------------------------------------------------------------------------
    def embed_images_videos(self, input_ids=None, images=None, videos=None):
        """reference :182-277.  Returns (input_ids, None) when S == 1, else (None, inputs_embeds [B,S,D])."""
        if input_ids.shape[1] == 1:
            return input_ids, None
        ids = input_ids.contiguous()
        mm = self.mm_token_ids
        img_feat = vid_feat = None
        img_tokens = img_pitch = img_off = 0
        if images is not None:
            with torch.no_grad():                           # the CLIP tower is frozen (train_ullava.py:207-208, ullava_core.py:148)
                h = self._clip_hidden(images)               # [n, P+1, Dv]; the projector also runs on the CLS row,
            img_feat = self._project(h.view(-1, h.shape[-1])).view(h.shape[0], h.shape[1], -1)   # which the splice skips
            img_pitch, img_off, img_tokens = h.shape[1], 1, h.shape[1] - 1
        if videos is not None:
            with torch.no_grad():
                v = self.encode_video(videos)
            vid_feat = self._project(v.view(-1, v.shape[-1])).view(v.shape[0], v.shape[1], -1)
        spans = None
        if mm is not None or self.strict_checks:
            m_ = mm or {"IMG_START": -1, "IMG_END": -1, "VID_START": -1, "VID_END": -1}
            spans = ops.mm_spans(ids, m_["IMG_START"], m_["IMG_END"], m_["VID_START"], m_["VID_END"], self.model.embed_tokens.weight.shape[0])
            if self.strict_checks:
                sp = spans.cpu()
                if int((sp[:, 3] & 2).sum()):
                    raise IndexError("index out of range in self")         # nn.Embedding on an out-of-vocabulary id
                assert int((sp[:, 3] & 1).sum()) == 0, "Number of image/video start and end tokens should be the same."
                if int((sp[:, 0] == 1).sum()) > (0 if img_feat is None else img_feat.shape[0]) or \
                        int((sp[:, 0] == 2).sum()) > (0 if vid_feat is None else vid_feat.shape[0]):
                    raise IndexError("fewer images/videos than samples that reference one")
            if mm is None:
                spans = None
        if self._training_graph():
            # projector_from_scratch (stage 1, configs/train/ullava_core.yaml): reference :230-240 detaches every text row of a
            # sample that carries an image / video except its start / end token rows
            emb = A.embed_splice(self.model.embed_tokens.weight, img_feat, vid_feat, ids, spans, img_tokens, img_pitch, img_off,
                                 detach_text=bool(self.projector_from_scratch))
        else:
            emb = ops.embed_splice(ids, self.model.embed_tokens.weight, img_feat, vid_feat, spans, img_tokens, img_pitch, img_off)
        return None, emb

    # -- LLaMA ---------------------------------------------------------------------------------------------
    def _rope_inv_freq(self, device):
        if self._inv_freq is None or self._inv_freq.device != device:
            hd = self.config.hidden_size // self.config.num_attention_heads
            inv = 1.0 / (self.config.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))   # LlamaRotaryEmbedding
            self._inv_freq = inv.to(device)
        return self._inv_freq

    def _pos_arange(self, device, n: int) -> torch.Tensor:
        ar = getattr(self, "_pos_cache", None)
        if ar is None or ar.device != device or ar.numel() < n:
            ar = self._pos_cache = torch.arange(max(4096, 2 * n), device=device, dtype=torch.int64)
        return ar

    def _llama(self, inputs_embeds, attention_mask, position_ids, output_hidden_states, cache: Optional[KVCache] = None):
        """LlamaModel.forward.  cache=None: plain prefill.  cache empty: prefill that also fills the cache.  cache filled:
        incremental step(s) -- the new tokens' q attend to cached K / V^T plus their own."""
        cfg = self.config
        pk = self._pk(for_llama=True)
        B, S, D = inputs_embeds.shape
        H = cfg.num_attention_heads
        hd = D // H
        dev = inputs_embeds.device
        past = cache.length if cache is not None else 0
        if position_ids is None:
            # positions past .. past + S - 1 of every row: a slice of one arange kept per device (a decode step at batch 1 then costs no
            # torch kernel for its positions -- it was an arange and an add per step; larger batches pay one repeat)
            ar = self._pos_arange(dev, past + S)
            pos = ar[past:past + S] if B == 1 else ar[past:past + S].repeat(B)
        else:
            pos = position_ids.to(torch.int64).expand(B, S).reshape(-1).contiguous()
        key_mask = None if attention_mask is None else attention_mask.to(torch.int32).contiguous()
        if cache is not None and past + S > cache.smax:
            raise RuntimeError(f"KV cache too small: {past + S} > {cache.smax}")
        inv_freq = self._rope_inv_freq(dev)
        x = inputs_embeds.reshape(B * S, D)
        T = B * S
        # prefill with LLaMA's head_dim: RoPE runs inside the QKV GEMM's epilogue from one cos / sin table per forward (the
        # positions are the same for every layer); other shapes (tiny test models, decode steps) use the stand-alone kernels
        f32 = x.dtype == torch.float32       # fp32 build: stand-alone RoPE kernels, one generic GEMM / attention (csrc/f32.hip)
        fuse_rope = hd == 128 and T > 4 and D % 64 == 0 and not (cache is not None and past > 0) and not f32
        # generation steps of at most 4 tokens (the GEMV shapes): RoPE and the cache append run in the q|k|v GEMV's epilogue, from the
        # same per-forward table
        # (that kernel stages the T x D activations in 32 KB of LDS: LLaMA-7B at T = 4 is exactly the limit; wider models / more rows take the
        # stand-alone RMSNorm + RoPE-append kernels below)
        fuse_append = cache is not None and past > 0 and T <= 4 and hd % 2 == 0 and D % 8 == 0 and T * D * 2 <= 32768 and not f32
        rope_cs = ops.rope_table(pos, inv_freq, x.dtype) if (fuse_rope or fuse_append) else None
        all_h = []
        I = cfg.intermediate_size
        coarse = None
        if ops.coarse_ok() and pk["llama"]:
            if fuse_append and I % 8 == 0:
                coarse = "decode"
            elif fuse_rope and cache is None and T > 16 and 16 < S <= 1024 and I % 64 == 0:
                coarse = "prefill"
        if coarse is not None:
            # one C call for the whole layer stack (csrc/layers.hip): the same launches as the loop below, bit-identical results
            stack = pk.get("_c_llama")
            if stack is None:
                stack = pk["_c_llama"] = ops.LayerStack(_lib.LlamaLayer, [dict(ln1=w["ln1"], ln2=w["ln2"], qkv=(w["w_qkv"], None), o=(w["w_o"], None),
                                                                                gu=(w["w_gu"], None), down=(w["w_down"], None)) for w in pk["llama"]])
            L = len(pk["llama"])
            if output_hidden_states:
                outs = [torch.empty(T, D, device=dev, dtype=x.dtype) for _ in range(L)]
                all_h = [x.view(B, S, D)] + [o.view(B, S, D) for o in outs[:-1]]
            else:
                outs = [torch.empty(T, D, device=dev, dtype=x.dtype)] * L
            if coarse == "decode":
                ops.llama_decode_layers(stack, x, outs, rope_cs[0], rope_cs[1], key_mask, cache.c_ptrs()[0], cache.c_ptrs()[1], B, S, H, hd, I, cache.smax,
                                        past, cfg.rms_norm_eps)
            else:
                ops.llama_prefill_layers(stack, x, outs, rope_cs[0], rope_cs[1], key_mask, B, S, H, hd, I, cfg.rms_norm_eps)
            x = outs[-1]
        for li, w in enumerate(pk["llama"] if coarse is None else ()):
            if output_hidden_states:
                all_h.append(x.view(B, S, D))
            decode = cache is not None and past > 0
            if fuse_append:
                kc, vtc = cache.k[li], cache.vt[li]
                q = ops.linear_qkv_rope_append(x, w["w_qkv"], rope_cs[0], rope_cs[1], B, S, H, hd, kc, vtc, cache.smax, past, rms_w=w["ln1"],
                                               rms_eps=cfg.rms_norm_eps)
                att = torch.empty(T, D, device=dev, dtype=x.dtype)
                ops.attention(q, kc, vtc, att, B, H, S, past + S, hd, (S * D, hd, D), (H * cache.smax * hd, cache.smax * hd, hd), (S * D, hd, D),
                              key_mask, causal=True, scale_mode=1, scale=hd ** -0.5)
                x = ops.linear(att, w["w_o"], residual=x)
                a = ops.linear(x, w["w_gu"], swiglu=True, rms_w=w["ln2"], rms_eps=cfg.rms_norm_eps)
                x = ops.linear(a, w["w_down"], residual=x)
                continue
            if fuse_rope:
                qkv = ops.linear_qkv_rope(ops.rmsnorm(x, w["ln1"], cfg.rms_norm_eps), w["w_qkv"], rope_cs[0], rope_cs[1], 2 * D, hd)
            else:
                qkv = ops.linear(x, w["w_qkv"], rms_w=w["ln1"], rms_eps=cfg.rms_norm_eps)      # input_layernorm -> q|k|v
            att = torch.empty(T, D, device=dev, dtype=x.dtype)
            if decode:
                # generation step: RoPE + cache append in one launch, then the split-key attention over the cache
                kc, vtc = cache.k[li], cache.vt[li]
                ops.rope_append(qkv, 3 * D, pos, inv_freq, B, S, H, hd, kc, vtc, cache.smax, past)
                ops.attention(qkv, kc, vtc, att, B, H, S, past + S, hd, (S * 3 * D, hd, 3 * D), (H * cache.smax * hd, cache.smax * hd, hd),
                              (S * D, hd, D), key_mask, causal=True, scale_mode=1, scale=hd ** -0.5)
            else:
                if not fuse_rope:
                    ops.rope_inplace(qkv, 3 * D, pos, inv_freq, T, 2 * H, hd)
                st = (S * 3 * D, hd, 3 * D)
                if cache is None:                                # V goes in as rows of the fused q|k|v buffer: no V^T pass
                    ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], att, B, H, S, S, hd, st, st, (S * D, hd, D), key_mask, causal=True,
                                  scale_mode=1, scale=hd ** -0.5, v_strides=st)
                else:                                            # prefill that also fills the cache (K rows, V^T image)
                    kc, vt = cache.k[li], cache.vt[li]
                    kc[:, :, :S].copy_(qkv.view(B, S, 3, H, hd)[:, :, 1].permute(0, 2, 1, 3))
                    ops.transpose_v(qkv[:, 2 * D:], S * 3 * D, 3 * D, B, S, H, hd, pitch=cache.smax, out=vt)
                    ops.attention(qkv, qkv[:, D:], vt, att, B, H, S, S, hd, st, st, (S * D, hd, D), key_mask, causal=True, scale_mode=1,
                                  scale=hd ** -0.5)
            x = ops.linear(att, w["w_o"], residual=x)
            a = ops.linear(x, w["w_gu"], swiglu=True, rms_w=w["ln2"], rms_eps=cfg.rms_norm_eps)  # post_attention_layernorm -> gate|up
            x = ops.linear(a, w["w_down"], residual=x)
        x = ops.rmsnorm(x, self.model.norm.weight, cfg.rms_norm_eps)
        last = x.view(B, S, D)
        if cache is not None:
            cache.length = past + S
            cache.last_hidden.append(last)
        if output_hidden_states:
            all_h.append(last)
        return last, (tuple(all_h) if output_hidden_states else None)

    def forward(self, input_ids: torch.LongTensor = None, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_values: Optional[List[torch.FloatTensor]] = None,
                inputs_embeds: Optional[torch.FloatTensor] = None, labels: Optional[torch.LongTensor] = None,
                use_cache: Optional[bool] = None, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, images: Optional[torch.FloatTensor] = None,
                videos: Optional[torch.FloatTensor] = None, return_dict: Optional[bool] = None):
        """reference :279-355 (same argument list)."""
        if output_attentions:
            raise NotImplementedError("attention probabilities never leave LDS on this path")
        cache = None
        if past_key_values is not None or use_cache:
            if past_key_values is not None and not isinstance(past_key_values, KVCache):
                # an HF-style cache handed in by the caller (legacy tuple of (key, value) per layer, or a transformers Cache object)
                past_key_values = KVCache.from_hf(past_key_values, headroom=getattr(self, "_cache_headroom", 512))
            cache = past_key_values
        output_hidden_states = output_hidden_states if output_hidden_states is not None else self.config.output_hidden_states
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        if inputs_embeds is None:
            ids, inputs_embeds = self.embed_images_videos(input_ids, images, videos)
            if inputs_embeds is None:
                inputs_embeds = ops.embed_splice(ids.contiguous(), self.model.embed_tokens.weight, None, None, None)
        if use_cache and cache is None:
            B_, S_ = inputs_embeds.shape[:2]
            cache = KVCache(self.config.num_hidden_layers, B_, self.config.num_attention_heads,
                            self.config.hidden_size // self.config.num_attention_heads,
                            max(S_ + getattr(self, "_cache_headroom", 512), 64), inputs_embeds.device, inputs_embeds.dtype)
        if self._training_graph():
            if cache is not None:
                raise NotImplementedError("use_cache / past_key_values are inference features (the training scripts set use_cache=False)")
            last, all_h = self._llama_train(inputs_embeds, attention_mask, position_ids, output_hidden_states)
            logits = A.linear(last, self.lm_head.weight)
            loss = A.shifted_cross_entropy(logits, labels) if labels is not None else None
        else:
            last, all_h = self._llama(inputs_embeds, attention_mask, position_ids, output_hidden_states, cache)
            logits = ops.linear(last, self.lm_head.weight)
            loss = ops.shifted_cross_entropy(logits, labels) if labels is not None else None
        if not return_dict:
            # reference :346-348: (loss,) + (logits,) + LlamaModel outputs[1:] = past_key_values / hidden_states when requested
            out = (logits,) + ((cache,) if cache is not None else ()) + ((all_h,) if all_h is not None else ())
            return ((loss,) + out) if loss is not None else out
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=cache, hidden_states=all_h, attentions=None)

    def prepare_inputs_for_generation(self, input_ids=None, inputs_embeds=None, attention_mask=None, images=None, videos=None,
                                      labels=None, past_key_values=None, **kwargs):
        """reference :357-395."""
        if past_key_values:
            input_ids = input_ids[:, -1:]
        position_ids = kwargs.get("position_ids", None)
        if attention_mask is not None and position_ids is None:
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            if past_key_values:
                position_ids = position_ids[:, -1].unsqueeze(-1)
        model_inputs = {"inputs_embeds": inputs_embeds} if inputs_embeds is not None and past_key_values is None else {"input_ids": input_ids}
        model_inputs.update({"position_ids": position_ids, "past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                             "attention_mask": attention_mask, "images": images, "videos": videos})
        return model_inputs

    @torch.no_grad()
    def generate(self, input_ids=None, images=None, videos=None, attention_mask=None, max_new_tokens=32, do_sample=False,
                 temperature=1.0, top_p=None, top_k=50, num_beams=1, no_repeat_ngram_size=None, stopping_criteria=None, eos_token_id=None,
                 pad_token_id=None, output_hidden_states=False, return_dict_in_generate=False, use_cache=None,
                 keep_last_step_only=False, **kwargs):
        """Token-by-token decoding with HF GenerationMixin's greedy / sampling semantics (the reference inherits `generate`):
        every step goes through `prepare_inputs_for_generation` (position_ids = cumsum(attention_mask) - 1, so left-padded batches
        get the reference's RoPE positions); `eos_token_id` defaults to `config.eos_token_id` (int or list); rows that have emitted
        EOS are tracked per row, keep receiving `pad_token_id` (default: the EOS id, as HF does for open-ended generation) and the
        loop ends when every row is finished or a stopping criterion fires.
        use_cache=False reproduces the reference checkpoints' behaviour (config.use_cache=False: every step re-runs the multimodal
        prefill); use_cache=True prefills once and then streams the weights once per token through the GEMV kernels
        with a KV cache.  Greedy (`do_sample=False`) is deterministic; sampling draws from torch's RNG on
        softmax(logits / temperature) with optional nucleus filtering."""
        if num_beams != 1:
            raise NotImplementedError("beam search is not used by the reference callers (num_beams=1)")
        if kwargs:
            # HF generate() validates its model kwargs and raises on unknown ones; options this loop does not implement must not be
            # swallowed (a silently ignored `repetition_penalty` changes the ids)
            raise TypeError(f"generate() got unsupported keyword arguments {sorted(kwargs)} (supported: greedy / sampling with temperature, "
                            "top_k, top_p, no_repeat_ngram_size, stopping_criteria, eos_token_id, pad_token_id, max_new_tokens)")
        ngram = int(no_repeat_ngram_size) if no_repeat_ngram_size else 0
        if ngram < 0:
            raise ValueError(f"`no_repeat_ngram_size` has to be a positive integer, but is {no_repeat_ngram_size}")
        use_cache = self.config.use_cache if use_cache is None else use_cache
        seq = input_ids
        B = seq.shape[0]
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        eos_ids = None
        if eos is not None:
            eos_ids = torch.as_tensor([eos] if isinstance(eos, int) else list(eos), device=seq.device, dtype=seq.dtype)
        pad = pad_token_id if pad_token_id is not None else getattr(self.config, "pad_token_id", None)
        if pad is None and eos_ids is not None:
            pad = int(eos_ids[0])
        steps_hidden = []
        cache = None
        self._cache_headroom = max_new_tokens + 1
        # No padding anywhere (no mask given, or a mask of ones -- one host read, here): the steps run without a key mask.  The results
        # are the same (position_ids = cumsum(ones) - 1 = arange; every key attended) and the attention kernels skip their per-key
        # mask loads, which sit on the latency chain of every decode step.
        no_pad = attention_mask is None or bool(attention_mask.ne(0).all())
        sampling = bool(do_sample and temperature and temperature > 0)
        crit = [] if stopping_criteria is None else (list(stopping_criteria) if isinstance(stopping_criteria, (list, tuple)) else [stopping_criteria])
        # Greedy decoding keeps the whole step on the device: the tokens land in a pre-sized [B, L0 + max_new_tokens] buffer, the
        # argmax / pad-fill / EOS bookkeeping is ONE kernel (ops.greedy_step), and the host looks at the "rows still unfinished"
        # counters only every CHECK steps (every step when a stopping criterion needs the ids on the host anyway).  Steps that run past
        # the point where every row had finished only produce pad tokens; the result is trimmed to what a per-step check returns.
        fused = not sampling and ngram == 0 and seq.is_cuda
        L0 = seq.shape[1]
        CHECK = 1 if crit else 8
        if fused:
            buf = torch.empty(B, L0 + max_new_tokens, dtype=torch.int64, device=seq.device)
            buf[:, :L0] = seq
            live = torch.ones(B, dtype=torch.int32, device=seq.device)
            alive = torch.zeros(max_new_tokens, dtype=torch.int32, device=seq.device)
        else:
            unfinished = torch.ones(B, dtype=torch.bool, device=seq.device)
        n_done = 0                                              # tokens appended so far
        stop_at = None                                          # fused: number of tokens after which every row had finished
        for step in range(max_new_tokens):
            if fused:
                seq = buf[:, :L0 + step]
            mask = None if no_pad else torch.cat(
                [attention_mask, attention_mask.new_ones(B, seq.shape[1] - attention_mask.shape[1])], dim=1)
            inputs = self.prepare_inputs_for_generation(input_ids=seq, attention_mask=mask, images=images, videos=videos,
                                                        past_key_values=cache if use_cache else None, use_cache=use_cache)
            if use_cache:
                # KV-cached decoding: prefill once, then one token per step; vision runs once
                if step > 0:
                    inputs["images"] = inputs["videos"] = None
                out = self.forward(**inputs)
                cache = out.past_key_values
            else:
                # the reference's released checkpoints run with use_cache=False: every step re-runs the multimodal prefill
                out = self.forward(**inputs, output_hidden_states=output_hidden_states)
                if output_hidden_states:
                    if keep_last_step_only:
                        steps_hidden = [out.hidden_states]      # evaluate() only reads hidden_states[-1] (the last step)
                    else:
                        steps_hidden.append(out.hidden_states)
            if fused:
                ops.greedy_step(out.logits[:, -1], live, eos_ids, pad, buf, L0 + step, alive[step:step + 1])
                n_done = step + 1
                if crit:
                    if any(bool(torch.as_tensor(c(buf[:, :L0 + n_done], None)).all()) for c in crit):
                        break
                if eos_ids is not None and (n_done % CHECK == 0 or n_done == max_new_tokens):
                    a_host = alive[:n_done].cpu()               # the only device -> host read: every CHECK steps
                    dead = (a_host == 0).nonzero()
                    if dead.numel():
                        stop_at = int(dead[0]) + 1
                        break
                continue
            logits = out.logits[:, -1].float()
            if ngram > 0:
                # HF NoRepeatNGramLogitsProcessor (logits processors run before the sampling warpers): tokens that would complete an
                # n-gram already present in the row (prompt included) get -inf
                for b_, banned in enumerate(no_repeat_ngram_banned_tokens(seq.tolist(), ngram)):
                    if banned:
                        logits[b_, torch.as_tensor(banned, device=logits.device)] = float("-inf")
            if sampling:
                nxt = torch.multinomial(sampling_probs(logits, temperature, top_k, top_p), 1)
            else:
                nxt = logits.argmax(-1, keepdim=True)
            if pad is not None:
                nxt = torch.where(unfinished.unsqueeze(1), nxt, torch.full_like(nxt, pad))      # finished rows: pad fill
            seq = torch.cat([seq, nxt], dim=1)
            if eos_ids is not None:
                unfinished = unfinished & ~torch.isin(nxt.squeeze(1), eos_ids)
                if not bool(unfinished.any()):
                    break
            if crit:
                if any(bool(torch.as_tensor(c(seq, None)).all()) for c in crit):
                    break
        if fused:
            n_tok = n_done if stop_at is None else stop_at
            seq = buf[:, :L0 + n_tok].contiguous()
            if stop_at is not None and stop_at < n_done:
                # steps that ran after every row had finished (at most CHECK - 1): drop their states so that the outputs equal a per-step check's
                if cache is not None:
                    del cache.last_hidden[n_tok:]
                    cache.length = L0 + n_tok - 1
                if steps_hidden and not keep_last_step_only:
                    del steps_hidden[n_tok:]
                elif steps_hidden and keep_last_step_only and not use_cache:
                    # the kept step saw L0 + n_done - 1 positions; the reference's last step L0 + n_tok - 1
                    steps_hidden = [tuple(h[:, :L0 + n_tok - 1] for h in steps_hidden[0])]
        if output_hidden_states and use_cache and cache is not None:
            # same tensor the no-cache path returns at its last step: last-layer states of ALL positions fed so far, joined once
            steps_hidden = [(torch.cat(cache.last_hidden, dim=1),)]
        if return_dict_in_generate:
            return GenerateOutput(sequences=seq, hidden_states=tuple(steps_hidden) if output_hidden_states else None)
        return seq
