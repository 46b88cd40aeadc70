"""HF-format checkpoint directories <-> the MI355X modules.

reference: `UllavaCoreForCausalLM.from_pretrained(path, torch_dtype=...)` (inference_ullava_core.py:35, train_ullava_core.py:94) and
`UllavaForCausalLM.from_pretrained(...)` (inference_ullava.py:37, evaluation/eval_ullava.py:135, webui/gradio_chat.py:26) go through
transformers' PreTrainedModel loader: `config.json` + either one weight file or an index json naming shards, safetensors or
torch-pickle.  This module reads the same directory layouts without importing transformers:

    config.json
    model.safetensors                | pytorch_model.bin
    model.safetensors.index.json     | pytorch_model.bin.index.json   -> {"weight_map": {param_name: shard_file}}

Shards are streamed one at a time (a LLaMA-7B checkpoint is 13.5 GB; the host never holds more than one shard), parameters are
copied into the already-allocated device tensors, and the key-name drift between transformers 4.29 checkpoints and the module tree
(`vision_encoder.vision_model.*`, `rotary_emb.inv_freq`, `position_ids`) is handled by the modules' own
`load_state_dict`.  `save_pretrained` writes the reference's key names back, so a directory written here loads in the reference.
"""
import json
import os
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import torch

WEIGHT_FILES = ("model.safetensors", "pytorch_model.bin")
INDEX_FILES = ("model.safetensors.index.json", "pytorch_model.bin.index.json")
_SKIP = ("position_ids", "rotary_emb.inv_freq")


def read_config(path: str) -> dict:
    with open(os.path.join(path, "config.json")) as f:
        return json.load(f)


def shard_files(path: str) -> List[str]:
    """Weight files of a checkpoint directory, in a deterministic order."""
    for idx in INDEX_FILES:
        p = os.path.join(path, idx)
        if os.path.exists(p):
            with open(p) as f:
                wm = json.load(f)["weight_map"]
            return [os.path.join(path, n) for n in sorted(set(wm.values()))]
    for w in WEIGHT_FILES:
        p = os.path.join(path, w)
        if os.path.exists(p):
            return [p]
    raise FileNotFoundError(f"u-llava_amd: no {' / '.join(WEIGHT_FILES + INDEX_FILES)} under {path}")


def _load_file(fn: str) -> Dict[str, torch.Tensor]:
    if fn.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(fn, device="cpu")
    return torch.load(fn, map_location="cpu", weights_only=True, mmap=True)


def iter_shards(path: str) -> Iterator[Dict[str, torch.Tensor]]:
    for fn in shard_files(path):
        yield _load_file(fn)


def _canon(key: str, has_vm_prefix: bool) -> str:
    """checkpoint key -> module key (the inverse of `_reference_key`)."""
    return key.replace("vision_encoder.vision_model.", "vision_encoder.") if has_vm_prefix else key


def load_into(model: torch.nn.Module, path: str, strict: bool = True) -> Tuple[List[str], List[str]]:
    """Copy every tensor of the checkpoint at `path` into `model`'s parameters/buffers, shard by shard.
    Returns (missing, unexpected); raises on either when strict (like load_state_dict)."""
    own = dict(model.state_dict())
    seen, unexpected = set(), []
    for shard in iter_shards(path):
        for k, v in shard.items():
            if any(s in k for s in _SKIP):
                continue
            ck = _canon(k, True)
            if ck not in own:
                unexpected.append(k)
                continue
            dst = own[ck]
            if tuple(dst.shape) != tuple(v.shape):
                raise RuntimeError(f"u-llava_amd: size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(dst.shape)}")
            with torch.no_grad():
                dst.copy_(v.to(dst.dtype))
            seen.add(ck)
        del shard
    missing = [k for k in own if k not in seen and not any(s in k for s in _SKIP)]
    if strict and (missing or unexpected):
        raise RuntimeError(f"u-llava_amd: checkpoint {path} does not match the model: missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                           f"unexpected {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
    return missing, unexpected


def _reference_key(key: str) -> str:
    """module key -> the name the reference's module tree gives the same tensor (CLIPVisionModel nests `.vision_model.`)."""
    for pre in ("llm.vision_encoder.", "vision_encoder."):
        if key.startswith(pre) and not key.startswith(pre + "vision_model."):
            return pre + "vision_model." + key[len(pre):]
    return key


def save_pretrained(model: torch.nn.Module, path: str, max_shard_bytes: int = 5 << 30, safe_serialization: bool = True):
    """config.json + (sharded) weights under the reference's parameter names."""
    os.makedirs(path, exist_ok=True)
    cfg = model.config.to_dict()
    cfg["architectures"] = [type(model).__name__]
    cfg["torch_dtype"] = str(next(model.parameters()).dtype).replace("torch.", "")
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)
    full = model.state_dict()
    sd = {_reference_key(k): v.detach().to("cpu").contiguous() for k, v in full.items() if ".lora_A." not in k and ".lora_B." not in k}
    lora = {k: v.detach().to("cpu").contiguous() for k, v in full.items() if ".lora_A." in k or ".lora_B." in k}
    if lora:
        # train_ullava.py:287-289 with lora_r > 0: the base / head weights as above, the adapter beside them the way
        # PeftModel.save_pretrained writes it (keys relative to the wrapped language model)
        save_lora_adapter(getattr(model, "llm", model), path)
    shards, cur, size = [], {}, 0
    for k in sorted(sd):
        n = sd[k].numel() * sd[k].element_size()
        if cur and size + n > max_shard_bytes:
            shards.append(cur)
            cur, size = {}, 0
        cur[k] = sd[k]
        size += n
    shards.append(cur)
    ext = "safetensors" if safe_serialization else "bin"
    stem = "model" if safe_serialization else "pytorch_model"

    def write(d, fn):
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(d, fn, metadata={"format": "pt"})
        else:
            torch.save(d, fn)

    if len(shards) == 1:
        write(shards[0], os.path.join(path, f"{stem}.{ext}"))
        return
    wm = {}
    for i, d in enumerate(shards):
        name = f"{stem}-{i + 1:05d}-of-{len(shards):05d}.{ext}"
        write(d, os.path.join(path, name))
        wm.update({k: name for k in d})
    total = sum(v.numel() * v.element_size() for v in sd.values())
    with open(os.path.join(path, f"{stem}.{ext}.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": wm}, f, indent=2, sort_keys=True)


def save_lora_adapter(core: torch.nn.Module, path: str) -> None:
    """PeftModel.save_pretrained (train_ullava.py:291 `model.llm.save_pretrained(output_dir)`): adapter_config.json + adapter_model.safetensors,
    keys `base_model.model.<module path relative to the language model>.lora_A|lora_B.weight`."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    lc = getattr(core, "_lora", None)
    if lc is None:
        raise RuntimeError("u-llava_amd: no LoRA adapter is attached to this model")
    sd = {"base_model.model." + k: v.detach().to("cpu").contiguous() for k, v in core.state_dict().items() if ".lora_A." in k or ".lora_B." in k}
    save_file(sd, os.path.join(path, "adapter_model.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": lc.get("r"), "lora_alpha": lc.get("lora_alpha"),
                   "lora_dropout": lc.get("lora_dropout", 0.0), "target_modules": list(lc.get("target_modules", ())), "bias": "none",
                   "fan_in_fan_out": False, "inference_mode": True}, f, indent=2, sort_keys=True)


def _dtype_arg(torch_dtype):
    """reference: inference_ullava.py:26,164-168 `--dtype {fp32,bf16,fp16}` -> torch_dtype.  The MI355X path has bf16 (default, the
    reference's training dtype), fp16 and -- round 6, inference only -- fp32 kernel builds."""
    if torch_dtype in (None, torch.bfloat16, "bfloat16", "bf16"):
        return torch.bfloat16
    if torch_dtype in (torch.float16, "float16", "fp16", "half"):
        return torch.float16
    if torch_dtype in (torch.float32, "float32", "fp32", "float"):
        return torch.float32
    raise NotImplementedError(f"torch_dtype {torch_dtype!r}: the MI355X path has bf16, fp16 and fp32 kernel builds")


def _merge_config(raw: dict, config_overrides: dict) -> dict:
    """`config=` (a config object handed over by AutoModelForCausalLM.from_pretrained, or one of ours) replaces config.json's
    fields; other keyword overrides are applied on top; loader-only keywords of PreTrainedModel.from_pretrained are dropped."""
    cfg_obj = config_overrides.pop("config", None)
    for k in ("low_cpu_mem_usage", "device_map", "cache_dir", "trust_remote_code", "use_safetensors", "attn_implementation",
              "local_files_only", "revision", "token", "use_auth_token"):
        config_overrides.pop(k, None)
    if cfg_obj is not None:
        d = cfg_obj.to_dict() if hasattr(cfg_obj, "to_dict") else dict(cfg_obj)
        raw.update({k: v for k, v in d.items() if k in raw or k in ("llm_config", "sam_config", "vision_config")})
    raw.update(config_overrides)
    return raw


def core_from_pretrained(cls, path: str, torch_dtype=None, device=None, strict: bool = True, **config_overrides):
    from .configuration import UllavaCoreConfig
    raw = _merge_config(read_config(path), config_overrides)
    model = cls(UllavaCoreConfig(**raw), device=device, dtype=_dtype_arg(torch_dtype))
    load_into(model, path, strict=strict)
    if has_lora_adapter(path):                               # inference_ullava_core.py with a LoRA checkpoint directory
        merge_lora_adapter(model, path)
    model._packed = None
    return model


def ullava_from_pretrained(cls, path: str, torch_dtype=None, device=None, strict: bool = True, **config_overrides):
    from .configuration import UllavaConfig
    raw = _merge_config(read_config(path), config_overrides)
    model = cls(UllavaConfig(**raw), device=device, dtype=_dtype_arg(torch_dtype))
    # stage-2 checkpoints may ship without the frozen SAM encoder (it is loaded from sam_vit_h.pth by load_visual_checkpoint,
    # reference ullava.py:134-137): tolerate exactly that family of missing keys
    missing, unexpected = load_into(model, path, strict=False)
    hard_missing = [k for k in missing if not k.startswith("visual_model.image_encoder.")]
    if strict and (hard_missing or unexpected):
        raise RuntimeError(f"u-llava_amd: checkpoint {path} does not match the model: missing {hard_missing[:8]}, unexpected {unexpected[:8]}")
    if has_lora_adapter(path):                               # inference_ullava.py:41-43 (PeftModel.from_pretrained(model.llm, llm_path))
        merge_lora_adapter(model.llm, path)
    model.llm._packed = None
    model._sam.invalidate()
    return model


# -- LoRA adapters (inference_ullava.py:41-43: `PeftModel.from_pretrained(model.llm, path)` when --lora_r > 0) ---------------------------
# PEFT wraps nn.Linear.forward; this implementation reads the weights into fused GEMM operands and never calls that forward, so an
# adapter is MERGED into the base weights at load time instead: W += (lora_alpha / r) * B @ A for every target module -- what
# `PeftModel.merge_and_unload()` leaves behind.  Reads the files PEFT writes (adapter_config.json + adapter_model.safetensors / .bin,
# keys `base_model.model.<module path>.lora_A[.<adapter>].weight` [r, in] and `.lora_B[...].weight` [out, r]).  Host-side weight
# preparation like the rest of this file, with PEFT's own rounding points for the adapter's stored dtype (lora_merged_weight).
def lora_merged_weight(w: torch.Tensor, A: torch.Tensor, B: torch.Tensor, scaling: float, live_adapter_dtype: bool = False) -> torch.Tensor:
    """PEFT 0.4.0 (the reference's pin, shells/requirements.txt:25) tuners/lora.py `Linear.merge`:
    `self.weight.data += (lora_B.weight @ lora_A.weight) * scaling`.
    Adapter FILES (default): `PeftModel.from_pretrained` creates lora_A / lora_B as fp32 nn.Linear, moves them by device only, and
    `load_state_dict` upcasts a 16-bit adapter file into those fp32 parameters; the reference does no `.half()` afterwards
    (inference_ullava.py:43, eval_ullava.py:138) -- so whatever dtype the file stores, the delta is an fp32 matrix of the (upcast) values and
    the in-place add rounds ONCE, to the weight's dtype.
    The model's own attached adapters (merge_lora, the inference packs of a model under training) take the same arithmetic: under PEFT 0.4.0
    `get_peft_model` creates fp32 adapter parameters as well, so whatever dtype THIS path stores its adapters in, the merge is the fp32 delta
    of their values, rounded once.
    live_adapter_dtype=True is an explicit opt-in to the other reading -- adapter parameters that were cast with the model (`model.half()` after
    get_peft_model) make PEFT's `merge` a 16-bit computation with three roundings (the product B A, the scaling, the sum); nothing in the
    package passes it."""
    dt = w.dtype
    adt = torch.promote_types(A.dtype, B.dtype)
    d = B.detach().float() @ A.detach().float()
    if live_adapter_dtype and adt in (torch.float16, torch.bfloat16):
        d = (d.to(adt).float() * float(scaling)).to(adt).float()
    else:
        d = d * float(scaling)
    return (w.detach().float() + d.to(w.device)).to(dt)


def has_lora_adapter(path: str) -> bool:
    return os.path.isfile(os.path.join(path, "adapter_config.json")) and any(
        os.path.isfile(os.path.join(path, f)) for f in ("adapter_model.safetensors", "adapter_model.bin"))


def merge_lora_adapter(llm: torch.nn.Module, path: str, adapter_name: str = "default") -> List[str]:
    """Merge the LoRA adapter stored under `path` into `llm` (an UllavaCoreForCausalLM).  Returns the merged module paths."""
    with open(os.path.join(path, "adapter_config.json")) as f:
        cfg = json.load(f)
    if cfg.get("peft_type", "LORA") != "LORA":
        raise RuntimeError(f"u-llava_amd: adapter type {cfg.get('peft_type')} is not LoRA")
    r, alpha = int(cfg["r"]), float(cfg.get("lora_alpha", cfg["r"]))
    fan_in_fan_out = bool(cfg.get("fan_in_fan_out", False))
    fn = os.path.join(path, "adapter_model.safetensors")
    sd = _load_file(fn) if os.path.isfile(fn) else _load_file(os.path.join(path, "adapter_model.bin"))
    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in sd.items():
        for tag in ("lora_A", "lora_B"):
            marker = f".{tag}."
            if marker in k and k.endswith(".weight"):
                mod = k[:k.index(marker)]
                for pre in ("base_model.model.", "base_model."):
                    if mod.startswith(pre):
                        mod = mod[len(pre):]
                        break
                pairs.setdefault(mod, {})[tag] = v
    if not pairs:
        raise RuntimeError(f"u-llava_amd: no lora_A / lora_B tensors in {path}")
    modules = dict(llm.named_modules())
    merged = []
    with torch.no_grad():
        for mod, ab in sorted(pairs.items()):
            if "lora_A" not in ab or "lora_B" not in ab:
                raise RuntimeError(f"u-llava_amd: adapter for {mod} lacks lora_A or lora_B")
            target = modules.get(mod)
            if target is None or not hasattr(target, "weight"):
                raise RuntimeError(f"u-llava_amd: adapter target {mod} is not a module of the model")
            A, B = ab["lora_A"], ab["lora_B"]
            if A.shape[0] != r or B.shape[1] != r:
                raise RuntimeError(f"u-llava_amd: adapter rank mismatch on {mod}: A {tuple(A.shape)}, B {tuple(B.shape)}, r = {r}")
            w = target.weight
            if fan_in_fan_out:
                raise NotImplementedError("fan_in_fan_out adapters (Conv1D targets) do not occur on this path: every target is an nn.Linear")
            if (B.shape[0], A.shape[1]) != tuple(w.shape):
                raise RuntimeError(f"u-llava_amd: adapter delta {(B.shape[0], A.shape[1])} does not fit {mod}.weight {tuple(w.shape)}")
            w.copy_(lora_merged_weight(w.detach().cpu(), A, B, alpha / r).to(w.device))      # fp32 delta of the (upcast) file values, one rounding
            merged.append(mod)
    if hasattr(llm, "_packed"):
        llm._packed = None                                   # fused q|k|v / tile-major copies describe the old weights
    return merged
