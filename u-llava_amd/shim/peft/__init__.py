"""Drop-in `peft` for the reference's three LoRA call sites, over the MI355X model's native adapters.

The reference (peft==0.4.0, shells/requirements.txt:25) uses exactly this surface:

    from peft import LoraConfig, get_peft_model                     # train_ullava.py:27
    model.llm = get_peft_model(model.llm, LoraConfig(r=, lora_alpha=, target_modules=[...], lora_dropout=, bias="none",
                                                     task_type="CAUSAL_LM"))                          # train_ullava.py:228-236
    model.llm.print_trainable_parameters()                          # train_ullava.py:237
    model.llm.save_pretrained(output_dir)                           # train_ullava.py:291 (adapter files only)
    from peft import PeftModel                                      # inference_ullava.py:12, evaluation/eval_ullava.py:25
    model.llm = PeftModel.from_pretrained(model.llm, llm_path, torch_dtype=dtype)                    # inference_ullava.py:43, eval_ullava.py:138

With `<repo>/u-llava_amd/shim` in front on sys.path (the same switch that makes `from models import ...` resolve to the MI355X
implementation) those lines run unchanged.  PEFT proper rewrites `nn.Linear.forward`; the projections here are operands of fused HIP GEMMs, so
the adapters are the model's own (`UllavaCoreForCausalLM.add_lora`: `lora_A` / `lora_B` parameters on the attention projections, the low-rank
branch as two skinny GEMMs on the training path) and this package is the thin naming layer on top:

  * `get_peft_model` attaches them and returns a `PeftModel` whose module tree is `base_model.model.<the language model>` -- PEFT's nesting, so
    `state_dict()` keys carry `.base_model.model` exactly where `safe_save_model_for_hf_trainer(is_peft=True)` (train_ullava.py:71-79) strips it;
  * attribute reads AND writes fall through to the wrapped model (PEFT forwards reads only; the MI355X model invalidates weight packs by
    assignment), `forward` / `generate` call it;
  * `PeftModel.from_pretrained` folds the adapter stored at `path` into the base weights with PEFT's merge arithmetic
    (`checkpoint.merge_lora_adapter`) -- the inference kernels read plain weights.
"""
import dataclasses
import importlib as _il
import os as _os
import sys as _sys
from typing import List, Optional, Union

import torch
import torch.nn as nn

_REPO = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _REPO not in _sys.path:
    _sys.path.insert(0, _REPO)

__version__ = "0.4.0+ullava_amd"
_PROJ = ("q_proj", "k_proj", "v_proj")


@dataclasses.dataclass
class LoraConfig:
    """The fields of peft.LoraConfig the reference sets (train_ullava.py:228-235) plus PEFT's defaults for the rest."""
    r: int = 8
    target_modules: Optional[Union[List[str], str]] = None
    lora_alpha: float = 8
    lora_dropout: float = 0.0
    fan_in_fan_out: bool = False
    bias: str = "none"
    modules_to_save: Optional[List[str]] = None
    task_type: Optional[str] = None
    inference_mode: bool = False
    peft_type: str = "LORA"
    base_model_name_or_path: Optional[str] = None

    def to_dict(self):
        return dataclasses.asdict(self)


def _core_of(model):
    core = model
    while isinstance(core, (PeftModel, _LoraModel)):
        core = core.base_model if isinstance(core, PeftModel) else core.model
    if not hasattr(core, "add_lora"):
        raise TypeError(f"peft shim: {type(core).__name__} is not the MI355X language model (UllavaCoreForCausalLM)")
    return core


def _targets(core, target_modules) -> tuple:
    """PEFT 0.4.0 `_find_and_replace`: a module is a target when its dotted name ends with one of `target_modules` (a string is a
    full-match regex).  The reference passes the full names `find_linear_layers` returned (every layer's q_proj and v_proj); the adapters here
    are per projection KIND over all layers, so the match must select the same kinds in every layer."""
    import re
    names = [n for n, m in core.named_modules() if isinstance(m, nn.Linear)]
    if isinstance(target_modules, str):
        hit = [n for n in names if re.fullmatch(target_modules, n)]
    else:
        hit = [n for n in names if any(n.endswith(t) for t in (target_modules or ()))]
    if not hit:
        raise ValueError(f"Target modules {target_modules} not found in the base model. Please check the target modules and try again.")
    kinds = sorted({n.rsplit(".", 1)[-1] for n in hit}, key=lambda k: _PROJ.index(k) if k in _PROJ else 99)
    bad = [n for n in hit if ".self_attn." not in n or n.rsplit(".", 1)[-1] not in _PROJ]
    if bad:
        raise NotImplementedError(f"peft shim: LoRA targets on the MI355X path are the LLaMA attention projections q_proj / k_proj / v_proj "
                                  f"(the reference's configuration: q_proj, v_proj); got {bad[:4]}")
    n_layers = len(core.model.layers)
    for k in kinds:
        if sum(n.endswith("." + k) for n in hit) != n_layers:
            raise NotImplementedError(f"peft shim: `{k}` must be targeted in every one of the {n_layers} layers")
    return tuple(kinds)


class _LoraModel(nn.Module):
    """peft.tuners.lora.LoraModel's place in the module tree: `.model` is the adapted language model."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self._modules["model"], name)

    def forward(self, *args, **kwargs):
        return self.model.forward(*args, **kwargs)


class PeftModel(nn.Module):
    def __init__(self, model, peft_config: LoraConfig, adapter_name: str = "default"):
        super().__init__()
        self.base_model = _LoraModel(model)
        self.peft_config = {adapter_name: peft_config}
        self.active_adapter = adapter_name

    # -- PEFT forwards attribute reads to the base model; writes of attributes the wrapped model owns go there too ------------------------
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self._modules["base_model"], name)

    def __setattr__(self, name, value):
        if name in ("base_model", "peft_config", "active_adapter", "training") or "base_model" not in self.__dict__.get("_modules", {}):
            return super().__setattr__(name, value)
        core = self._modules["base_model"]._modules["model"]
        if hasattr(core, name) and not isinstance(value, nn.Module):
            return setattr(core, name, value)
        return super().__setattr__(name, value)

    def get_base_model(self):
        return self.base_model.model

    def forward(self, *args, **kwargs):
        return self.base_model.model.forward(*args, **kwargs)

    __call__ = nn.Module.__call__

    def generate(self, *args, **kwargs):
        return self.base_model.model.generate(*args, **kwargs)

    def get_nb_trainable_parameters(self):
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return trainable, sum(p.numel() for p in self.parameters())

    def print_trainable_parameters(self):
        t, a = self.get_nb_trainable_parameters()
        print(f"trainable params: {t} || all params: {a} || trainable%: {100 * t / a}")

    def save_pretrained(self, save_directory, **kwargs):
        """PeftModel.save_pretrained: the adapter files only (adapter_config.json + adapter_model.*), keys relative to the wrapped model."""
        C = _il.import_module("u-llava_amd.checkpoint")
        C.save_lora_adapter(self.get_base_model(), save_directory)

    def merge_and_unload(self):
        return self.get_base_model().merge_lora()

    @classmethod
    def from_pretrained(cls, model, model_id, adapter_name: str = "default", is_trainable: bool = False, **kwargs):
        """inference_ullava.py:43 / eval_ullava.py:138 (which spells the keyword `torch_type`): loader keywords are accepted and unused -- the
        adapter takes the dtype of the weights it is folded into."""
        C = _il.import_module("u-llava_amd.checkpoint")
        core = _core_of(model)
        if is_trainable:
            raise NotImplementedError("peft shim: from_pretrained(is_trainable=True) -- resume LoRA training through get_peft_model + load_state_dict")
        if not C.has_lora_adapter(model_id):
            raise ValueError(f"Can't find 'adapter_config.json' at '{model_id}'")
        with open(_os.path.join(model_id, "adapter_config.json")) as f:
            import json
            raw = json.load(f)
        C.merge_lora_adapter(core, model_id, adapter_name)
        known = {f_.name for f_ in dataclasses.fields(LoraConfig)}
        cfg = LoraConfig(**{k: v for k, v in raw.items() if k in known})
        cfg.inference_mode = True
        return cls(core, cfg, adapter_name)


PeftModelForCausalLM = PeftModel


def get_peft_model(model, peft_config: LoraConfig, adapter_name: str = "default") -> PeftModel:
    """train_ullava.py:236."""
    if getattr(peft_config, "peft_type", "LORA") != "LORA":
        raise NotImplementedError("peft shim: LoRA only (the reference's configuration)")
    if peft_config.bias != "none" or peft_config.fan_in_fan_out or peft_config.modules_to_save:
        raise NotImplementedError("peft shim: bias='none', fan_in_fan_out=False, no modules_to_save (the reference's configuration)")
    core = _core_of(model)
    core.add_lora(int(peft_config.r), float(peft_config.lora_alpha), float(peft_config.lora_dropout), _targets(core, peft_config.target_modules))
    return PeftModel(core, peft_config, adapter_name)


__all__ = ["LoraConfig", "PeftModel", "PeftModelForCausalLM", "get_peft_model"]
