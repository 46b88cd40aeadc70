"""CPU: host logic that needs no GPU -- the C-ABI library loads and exports every declared symbol, configs mirror the
reference, weight packing layouts, state-dict key names."""
import os
import re

import pytest
import torch

from conftest import pkg, ROOT, load_fixture


def test_library_exports_every_declared_symbol():
    L = pkg("_lib")
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "ullava_hip.h")).read()
    declared = set(re.findall(r"\bint(?:64_t)? (ull_[a-z0-9_]+)\(", header))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)


def test_ops_refuse_cpu_tensors():
    ops = pkg("ops")
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.rmsnorm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-6)


def test_state_dict_keys_match_reference_fixture():
    fx = load_fixture("g1_core_tiny_bf16.pt")
    C, M = pkg("configuration"), pkg("modeling_core")
    cd = fx["cfg"]
    cfg = C.UllavaCoreConfig(vision_config=cd["vision_config"], vision_hidden_layer=-2, mm_token_ids=cd["mm_token_ids"],
                             vocab_size=cd["vocab_size"], hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                             num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"])
    m = M.UllavaCoreForCausalLM(cfg)
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ours == {k: tuple(v) for k, v in fx["shapes"].items()}
    # 4.29.1-style CLIP prefix is accepted too
    sd = {k.replace("vision_encoder.", "vision_encoder.vision_model."): torch.zeros(s, dtype=torch.bfloat16) for k, s in ours.items()}
    m.load_state_dict(sd, strict=True)


def test_gate_up_interleave_layout():
    M = pkg("modeling_core")
    g = torch.arange(32 * 4, dtype=torch.float32).reshape(32, 4)
    u = -g
    w = M.interleave_gate_up(g, u)
    assert w.shape == (64, 4)
    assert torch.equal(w[0:16], g[0:16]) and torch.equal(w[16:32], u[0:16]) and torch.equal(w[32:48], g[16:32])


def test_config_to_dict_keys():
    C = pkg("configuration")
    d = C.UllavaConfig(llm_config=dict(hidden_size=64, num_attention_heads=4, vision_config=dict(hidden_size=32))).to_dict()
    for k in ("llm_config", "ce_weight", "bce_weight", "dice_weight", "l1_weight", "iou_weight", "out_dim", "seg_token_idx",
              "loc_token_idx", "train_mask_decoder", "model_type"):
        assert k in d
    assert d["llm_config"]["vision_config"]["hidden_size"] == 32 and d["seg_token_idx"] == 32007


def test_public_signatures_match_reference_fixture():
    """constructor / forward / evaluate / generate-prep signatures of the two model classes and the two configs
    carry the reference's parameter names in the reference's order (tests/golden/reference_signatures.json is generated from the
    reference with inspect.signature by gen_golden.py).  Extra trailing keyword parameters (device=, dtype=) are allowed."""
    import inspect
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))["signatures"]
    C, MC, MU = pkg("configuration"), pkg("modeling_core"), pkg("modeling_ullava")
    classes = {"UllavaCoreForCausalLM": MC.UllavaCoreForCausalLM, "UllavaForCausalLM": MU.UllavaForCausalLM,
               "UllavaCoreConfig": C.UllavaCoreConfig, "UllavaConfig": C.UllavaConfig}
    checked = 0
    for key, want in ref.items():
        if key == "registered_model_types":
            assert [C.UllavaCoreConfig.model_type, C.UllavaConfig.model_type] == want
            continue
        if key in ("models_all", "models_constants"):
            continue
        cname, mname = key.split(".")
        got = list(inspect.signature(getattr(classes[cname], mname)).parameters.values())
        want_named = [w for w in want if "VAR_" not in w[1]]
        if cname.endswith("Config"):
            # configs: every reference keyword must be accepted by name (ours spell out LLaMA fields the reference gets via **kwargs)
            names = {p.name for p in got}
            has_kwargs = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in got)
            for w in want_named:
                assert w[0] in names or has_kwargs, (key, w[0])
            for w in want_named:
                if w[0] in names and w[2] is not None and w[0] != "self":
                    assert repr(next(p.default for p in got if p.name == w[0])) == w[2], (key, w[0])
        else:
            got_named = [p for p in got if p.kind not in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL)]
            assert [p.name for p in got_named[:len(want_named)]] == [w[0] for w in want_named], (key, [p.name for p in got_named])
            for p, w in zip(got_named, want_named):
                if w[2] is not None:
                    assert repr(p.default) == w[2], (key, p.name, repr(p.default), w[2])
        checked += 1
    assert checked >= 16


_SHIM_PROBE = r"""
import json, sys, os, tempfile, torch
sys.path.insert(0, os.path.join(ROOT_DIR, "u-llava_amd", "shim"))
import models
from models import UllavaForCausalLM, UllavaCoreForCausalLM, KeywordsStoppingCriteria, DEFAULT_IMG_END_TOKEN, DEFAULT_IMG_START_TOKEN, DEFAULT_IMG_TOKEN, DEFAULT_IMG_PATCH_TOKEN
out = {"all": sorted(models.__all__), "consts": {k: getattr(models, k) for k in models.__all__ if k.startswith("DEFAULT_") or k == "IGNORE_INDEX"}}
from transformers import AutoConfig, AutoModelForCausalLM
cfg = models.UllavaCoreConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4, vocab_size=50,
                              vision_config=dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, image_size=28, patch_size=14))
m = UllavaCoreForCausalLM(cfg)
for p in m.parameters():
    p.data.zero_()
d = tempfile.mkdtemp()
m.save_pretrained(d)
ac = AutoConfig.from_pretrained(d)
out["auto_config_model_type"] = ac.model_type
m2 = AutoModelForCausalLM.from_pretrained(d)
out["auto_model_class"] = type(m2).__name__
out["auto_model_hidden"] = m2.config.hidden_size
print("RESULT" + json.dumps(out))
"""


def test_models_shim_is_a_drop_in_for_the_reference_package():
    """`from models import ...` with <repo>/u-llava_amd/shim on sys.path: same exported names and constants as the reference's
    models/__init__.py, Auto classes registered (reference models/ullava_core.py:398-399, models/ullava.py:437-438), and
    AutoModelForCausalLM.from_pretrained on a saved checkpoint directory builds our model.  Runs in a subprocess so that `models`
    never shadows anything in this test session."""
    import json
    import subprocess
    import sys
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))["signatures"]
    r = subprocess.run([sys.executable, "-c", f"ROOT_DIR = {ROOT!r}\n" + _SHIM_PROBE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:])
    assert out["all"] == ref["models_all"]
    assert out["consts"] == ref["models_constants"]
    assert out["auto_config_model_type"] == "ullava_core" and out["auto_model_class"] == "UllavaCoreForCausalLM" and out["auto_model_hidden"] == 64


class _FakeTokenizer:
    def __init__(self, n):
        self.vocab = {f"t{i}": i for i in range(n)}

    def __len__(self):
        return len(self.vocab)

    def add_tokens(self, toks, special_tokens=False):
        new = [t for t in toks if t not in self.vocab]
        for t in new:
            self.vocab[t] = len(self.vocab)
        return len(new)

    def add_special_tokens(self, d):
        return self.add_tokens(list(d.values()), True)

    def __call__(self, text):
        return type("E", (), {"input_ids": [self.vocab[text]] if text in self.vocab else [1, 2]})()

    def batch_decode(self, ids, skip_special_tokens=True):
        inv = {v: k for k, v in self.vocab.items()}
        return [" ".join(inv[int(i)] for i in row) for row in ids]


def _tiny_core():
    C, M = pkg("configuration"), pkg("modeling_core")
    cfg = C.UllavaCoreConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4, vocab_size=50,
                             vision_config=dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, image_size=28, patch_size=14))
    m = M.UllavaCoreForCausalLM(cfg)
    g = torch.Generator().manual_seed(0)
    for p in m.parameters():
        p.data.copy_(torch.randn(p.shape, generator=g).to(p.dtype))
    return m


def test_resize_token_embeddings_and_helpers():
    """train_ullava.py:156-158,212 / models/tools.py:34-117: tokenizer growth -> embed_tokens + lm_head rows, old rows kept, new rows
    averaged by the helpers, config.vocab_size updated, packed layouts invalidated."""
    T = pkg("tools")
    m = _tiny_core()
    old_in, old_out = m.get_input_embeddings().weight.data.clone(), m.get_output_embeddings().weight.data.clone()
    tok = _FakeTokenizer(50)
    m._packed = {"stale": True}
    T.smart_resize_token_embedding(["[SEG]", "[LOC]"], tok, m)
    assert len(tok) == 52 and m.config.vocab_size == 52 and m._packed is None
    w_in, w_out = m.get_input_embeddings().weight.data, m.get_output_embeddings().weight.data
    assert tuple(w_in.shape) == (52, 64) and tuple(w_out.shape) == (52, 64)
    assert torch.equal(w_in[:50], old_in) and torch.equal(w_out[:50], old_out)
    assert torch.equal(w_in[50], old_in.mean(0, keepdim=True)[0]) and torch.equal(w_out[51], old_out.mean(0, keepdim=True)[0])
    T.multi_modal_resize_token_embedding(dict(IMG_PATCH="<ip>", VID_PATCH="<vp>", IMG_START="<ib>", IMG_END="<ie>", VID_START="<vb>", VID_END="<ve>"), tok, m)
    assert m.config.vocab_size == 58 and m.lm_head.weight.shape[0] == 58
    names = dict(m.named_parameters())
    assert "lm_head.weight" in names and "model.embed_tokens.weight" in names
    # .to(dtype) invalidates the packs and is reflected by .dtype (inference_ullava.py: from_pretrained(torch_dtype=...) then .cuda())
    m._packed = {"stale": True}
    m.to(torch.float16)
    assert m.dtype == torch.float16 and m._packed is None


def test_keywords_stopping_criteria_semantics():
    """models/tools.py:11-31: first call only records the prompt length; then single-token keyword on row 0's last id, or keyword
    substring in the decoded continuation."""
    T = pkg("tools")
    tok = _FakeTokenizer(10)
    prompt = torch.tensor([[1, 2, 3]])
    c = T.KeywordsStoppingCriteria(["t7", "t4 t5"], tok, prompt)
    assert c.keyword_ids == [7]
    assert c(torch.tensor([[1, 2, 3, 7]]), None) is False            # first call: start_len only
    assert c(torch.tensor([[1, 2, 3, 7, 7]]), None) is True          # last id is the single-token keyword
    assert c(torch.tensor([[1, 2, 3, 4, 6]]), None) is False
    assert c(torch.tensor([[1, 2, 3, 4, 5, 6]]), None) is True       # "t4 t5" appears in the decoded continuation


def test_shim_registers_with_the_reference_registry():
    """models/ullava_core.py:78, models/ullava.py:69: with the reference's `utils.registry` importable (a stand-in with the same
    `mapping` / `register_model` / `get_model_class` surface here), importing the shim enters both classes under the reference's names,
    replacing an earlier registration and staying idempotent.  Subprocess: `models` / `utils` must not leak into this session."""
    import subprocess
    import sys
    code = f"""
import sys, types, os
ROOT = {ROOT!r}
utils = types.ModuleType("utils"); utils.__path__ = []
regmod = types.ModuleType("utils.registry")
class registry:
    mapping = {{"model_name_mapping": {{"ullava": object}}}}
    @classmethod
    def register_model(cls, name):
        def wrap(c):
            if name in cls.mapping["model_name_mapping"]:
                raise KeyError(name)
            cls.mapping["model_name_mapping"][name] = c
            return c
        return wrap
    @classmethod
    def get_model_class(cls, name):
        return cls.mapping["model_name_mapping"].get(name, None)
regmod.registry = registry
sys.modules["utils"] = utils; sys.modules["utils.registry"] = regmod
sys.path.insert(0, os.path.join(ROOT, "u-llava_amd", "shim"))
import models
assert registry.get_model_class("ullava_core") is models.UllavaCoreForCausalLM
assert registry.get_model_class("ullava") is models.UllavaForCausalLM
import importlib
hf = importlib.import_module("u-llava_amd.hf_integration")
assert hf.register_with_reference_registry() is True          # idempotent
assert registry.get_model_class("ullava") is models.UllavaForCausalLM
print("OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-3000:]


def test_lora_adapter_is_merged_at_load(tmp_path):
    """inference_ullava.py:41-43 loads a LoRA checkpoint with PeftModel.from_pretrained(model.llm, path); here the adapter PEFT wrote
    (adapter_config.json + adapter_model.safetensors, `base_model.model.<module>.lora_A|B.weight`) is merged into the targets at
    load: W += (alpha / r) * (B @ A) with PEFT's rounding points, everything else untouched."""
    import json
    from safetensors.torch import save_file
    CK, M = pkg("checkpoint"), pkg("modeling_core")
    base = _tiny_core()
    d = str(tmp_path / "ckpt")
    base.save_pretrained(d)
    r, alpha = 4, 8.0
    g = torch.Generator().manual_seed(3)
    sd, want = {}, {}
    for li in range(base.config.num_hidden_layers):
        for name in ("q_proj", "v_proj"):
            mod = f"model.layers.{li}.self_attn.{name}"
            w = dict(base.named_modules())[mod].weight
            A, B = torch.randn(r, w.shape[1], generator=g) * 0.1, torch.randn(w.shape[0], r, generator=g) * 0.1
            if name == "q_proj":
                # adapter stored in fp32 (what PEFT 0.4.0 creates and the reference's training saves): fp32 delta, ONE rounding in `weight.data +=`
                sd[f"base_model.model.{mod}.lora_A.weight"] = A
                sd[f"base_model.model.{mod}.lora_B.weight"] = B
                want[mod] = (w.detach().float() + (B @ A) * (alpha / r)).to(w.dtype)
            else:
                # adapter FILE stored in the weights' 16-bit dtype: PeftModel.from_pretrained loads it into fp32 parameters (PEFT 0.4.0) and the
                # reference never casts them back, so the merge is still the fp32 delta of the upcast values with ONE rounding
                sd[f"base_model.model.{mod}.lora_A.weight"] = A.to(w.dtype)
                sd[f"base_model.model.{mod}.lora_B.weight"] = B.to(w.dtype)
                want[mod] = (w.detach().float() + (B.to(w.dtype).float() @ A.to(w.dtype).float()) * (alpha / r)).to(w.dtype)
    assert not CK.has_lora_adapter(d)
    save_file(sd, os.path.join(d, "adapter_model.safetensors"))
    json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "target_modules": ["q_proj", "v_proj"], "fan_in_fan_out": False},
              open(os.path.join(d, "adapter_config.json"), "w"))
    assert CK.has_lora_adapter(d)
    m = M.UllavaCoreForCausalLM.from_pretrained(d, torch_dtype=base.model.embed_tokens.weight.dtype)
    mods = dict(m.named_modules())
    for mod, w in want.items():
        assert torch.equal(mods[mod].weight.detach().cpu(), w), mod
    k0 = "model.layers.0.self_attn.k_proj"
    assert torch.equal(mods[k0].weight.detach().cpu(), dict(base.named_modules())[k0].weight.detach().cpu())
    # a rank that does not match the config is refused
    json.dump({"peft_type": "LORA", "r": r + 1, "lora_alpha": alpha}, open(os.path.join(d, "adapter_config.json"), "w"))
    with pytest.raises(RuntimeError):
        M.UllavaCoreForCausalLM.from_pretrained(d, torch_dtype=base.model.embed_tokens.weight.dtype)


def test_no_repeat_ngram_rule_matches_transformers_processor():
    """generate(no_repeat_ngram_size=n) (reference models/ullava.py:360 forwards it to HF generate): the host-side banned-token rule must be
    transformers' NoRepeatNGramLogitsProcessor's, row by row, on random id histories with many repeats."""
    from transformers.generation.logits_process import NoRepeatNGramLogitsProcessor
    M = pkg("modeling_core")
    g = torch.Generator().manual_seed(3)
    for n in (1, 2, 3, 4):
        for L in (1, 2, 3, 7, 40):
            ids = torch.randint(0, 6, (5, L), generator=g)
            scores = torch.zeros(5, 6)
            want = NoRepeatNGramLogitsProcessor(n)(ids, scores.clone())
            banned = M.no_repeat_ngram_banned_tokens(ids.tolist(), n)
            got = scores.clone()
            for b, toks in enumerate(banned):
                if toks:
                    got[b, torch.tensor(toks)] = float("-inf")
            assert torch.equal(got, want), (n, L)


def test_generate_rejects_unknown_options():
    """Options the decoding loop does not implement must raise, not be swallowed (HF validates model kwargs the same way)."""
    C, M = pkg("configuration"), pkg("modeling_core")
    cfg = C.UllavaCoreConfig(vision_config=dict(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64, image_size=28,
                                                patch_size=14), vocab_size=50, hidden_size=64, intermediate_size=128, num_hidden_layers=1,
                             num_attention_heads=4)
    m = M.UllavaCoreForCausalLM(cfg)
    ids = torch.ones(1, 3, dtype=torch.long)
    with pytest.raises(TypeError, match="repetition_penalty"):
        m.generate(input_ids=ids, repetition_penalty=1.2)
    with pytest.raises(NotImplementedError):
        m.generate(input_ids=ids, num_beams=4)
    with pytest.raises(ValueError):
        m.generate(input_ids=ids, no_repeat_ngram_size=-1)


def test_hf_trainer_constructs_optimises_and_saves_the_shim_model(tmp_path):
    """train_ullava.py:273-293 hands the model to a transformers Trainer (SegmentationTrainer): construction, create_optimizer() over the
    reference's trainable set and Trainer._save() must work on these plain nn.Modules -- including after the training path has made the
    q|k|v / gate|up parameters row slices of shared buffers (the state-dict hook hands out tensors that own their storage; safetensors
    refuses shared memory).  The saved file loads back to the same values."""
    from safetensors.torch import load_file
    from transformers import Trainer, TrainingArguments
    C, M = pkg("configuration"), pkg("modeling_ullava")
    cfg = C.UllavaConfig(llm_config=dict(vision_config=dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, image_size=28,
                                                            patch_size=14), vocab_size=120, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                         num_attention_heads=4), seg_token_idx=101, loc_token_idx=102, out_dim=256,
                         sam_config=dict(embed_dim=32, depth=2, num_heads=2, global_attn_indexes=[1]))
    m = M.UllavaForCausalLM(cfg)
    g = torch.Generator().manual_seed(0)
    for p in list(m.parameters()) + list(m.buffers()):          # (holders are torch.empty: buffers too, or NaN garbage breaks torch.equal)
        p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.dtype))
    # the reference's stage-2 switches (train_ullava.py:207-261, no LoRA)
    for p in m.parameters():
        p.requires_grad = False
    for p in list(m.llm.model.parameters()) + list(m.llm.lm_head.parameters()) + list(m.llm.vision_projector.parameters()):
        p.requires_grad = True
    for n, p in m.named_parameters():
        if any(x in n for x in ["lm_head", "embed_tokens", "seg_projector", "mask_decoder", "det_projector", "det_decoder"]):
            p.requires_grad = "mask_decoder.iou_prediction_head" not in n
    n_train = sum(p.requires_grad for p in m.parameters())
    # what a training forward does to the attention / MLP projections of every LLaMA layer
    for l in m.llm.model.layers:
        m.llm._alias_pack(l.self_attn, ("q_proj", "k_proj", "v_proj"), "_qkv_pack")
        m.llm._alias_pack(l.mlp, ("gate_proj", "up_proj"), "_gu_pack")
    assert m.llm.model.layers[0].self_attn.k_proj.weight.untyped_storage().nbytes() == 3 * 64 * 64 * 2
    want = {k: v.clone() for k, v in m.state_dict().items()}
    assert all(v.untyped_storage().nbytes() <= v.numel() * v.element_size() + 64 for v in m.state_dict().values())
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, learning_rate=2e-5, weight_decay=0.0, report_to=[], use_cpu=True)
    trainer = Trainer(model=m, args=args, train_dataset=None)
    opt = trainer.create_optimizer()
    assert sum(len(gr["params"]) for gr in opt.param_groups) == n_train
    trainer._save(str(tmp_path))
    back = load_file(str(tmp_path / "model.safetensors"))
    assert set(back) == set(want) and all(torch.equal(back[k], want[k]) for k in want)
    # an optimizer step through the aliased parameters moves the shared buffer (the GEMMs of the next forward read the new weights)
    q = m.llm.model.layers[0].self_attn.q_proj.weight
    q.grad = torch.ones_like(q)
    before = m.llm.model.layers[0].self_attn._qkv_pack[:64].clone()
    opt.step()
    assert not torch.equal(m.llm.model.layers[0].self_attn._qkv_pack[:64], before) and torch.equal(m.llm.model.layers[0].self_attn._qkv_pack[:64], q.data)


def test_alias_pack_leaves_externally_owned_parameters_alone():
    """DeepSpeed ZeRO-1/2 / FSDP bind p.data to their flat buffer BEFORE the first training forward (HF Trainer wraps first): the first
    _alias_pack call must see that the parameters are views into somebody else's storage, mark the holder and never rebind -- the
    owner's in-place updates of its flat buffer stay visible through the parameters."""
    M_ = pkg("modeling_core")
    holder = torch.nn.Module()
    names = ("q_proj", "k_proj", "v_proj")
    for n_ in names:
        setattr(holder, n_, M_.Linear(8, 16, bias=False, device="cpu", dtype=torch.bfloat16))
    flat = torch.arange(3 * 16 * 8 + 5, dtype=torch.float32).to(torch.bfloat16)          # the owner's flat partition (with its own padding)
    o = 5
    for n_ in names:
        w = getattr(holder, n_).weight
        w.data = flat[o:o + w.numel()].view_as(w)
        o += w.numel()
    packed, ws = M_.UllavaCoreForCausalLM._alias_pack(holder, names, "_qkv_pack")
    assert packed is None and holder._qkv_pack_external is True
    assert all(w.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for w in ws)
    flat.add_(1.0)                                                                      # the owner's optimizer step
    assert torch.equal(holder.k_proj.weight.data.reshape(-1), flat[5 + 128:5 + 256])
    assert M_.UllavaCoreForCausalLM._alias_pack(holder, names, "_qkv_pack")[0] is None    # and it stays that way
    # parameters that own their storage are aliased as before
    h2 = torch.nn.Module()
    for n_ in names:
        setattr(h2, n_, M_.Linear(8, 16, bias=False, device="cpu", dtype=torch.bfloat16))
        getattr(h2, n_).weight.data.normal_()
    p2, w2 = M_.UllavaCoreForCausalLM._alias_pack(h2, names, "_qkv_pack")
    assert p2 is not None and w2[2].data_ptr() == p2.data_ptr() + 32 * 8 * 2


REFERENCE = "/root/reference"


def _reference_source_nodes():
    """(find_linear_layers FunctionDef, the `if lora_r > 0:` block of train_ullava.main) parsed from the reference checkout when it is there
    (the build container); the test executes THOSE lines against the shim -- nothing of the reference is kept in this repository."""
    import ast
    src = open(os.path.join(REFERENCE, "train_ullava.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "find_linear_layers")
    blk = next(n for n in ast.walk(tree) if isinstance(n, ast.If) and isinstance(n.test, ast.Compare) and isinstance(n.test.left, ast.Name)
               and n.test.left.id == "lora_r")
    return fn, blk


def test_peft_shim_runs_the_references_lora_call_sites(tmp_path, capsys):
    """`from peft import LoraConfig, get_peft_model, PeftModel` resolves to u-llava_amd/shim/peft; the reference's own LoRA lines
    (train_ullava.py:88-113 find_linear_layers, :217-245 the lora_r > 0 block, :71-79 the is_peft state-dict filter, :291 save_pretrained,
    inference_ullava.py:43 PeftModel.from_pretrained) run against it unchanged."""
    import ast
    import sys
    import types
    shim = os.path.join(ROOT, "u-llava_amd", "shim")
    saved_path, saved_mod = list(sys.path), sys.modules.pop("peft", None)
    sys.path.insert(0, shim)
    try:
        import peft
        assert peft.__file__.startswith(shim) and peft.__version__.endswith("ullava_amd")
        from peft import LoraConfig, PeftModel, get_peft_model
        C, M, MC = pkg("configuration"), pkg("modeling_ullava"), pkg("modeling_core")
        cfg = C.UllavaConfig(llm_config=dict(vision_config=dict(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                                                                image_size=28, patch_size=14), vocab_size=120, hidden_size=64, intermediate_size=128,
                                             num_hidden_layers=3, num_attention_heads=4), seg_token_idx=101, loc_token_idx=102, out_dim=256,
                             sam_config=dict(embed_dim=32, depth=2, num_heads=2, global_attn_indexes=[1]))

        def build():
            m = M.UllavaForCausalLM(cfg)
            g = torch.Generator().manual_seed(0)
            for p in list(m.parameters()) + list(m.buffers()):
                p.data.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.dtype))
            return m
        model = build()
        assert isinstance(model.llm.model.layers[0].self_attn.q_proj, torch.nn.Linear)
        assert isinstance(model.llm.lm_head, torch.nn.Linear) and model.llm.lm_head.bias is None and "lm_head.bias" not in model.llm.state_dict()
        ns = dict(torch=torch, LoraConfig=LoraConfig, get_peft_model=get_peft_model, print=print, model=model, lora_r=4,
                  model_args=types.SimpleNamespace(lora_r=4, lora_alpha=8, lora_dropout=0.05, lora_target_modules="q_proj,v_proj"))
        if os.path.isdir(REFERENCE):
            fn, blk = _reference_source_nodes()
            exec(compile(ast.Module(body=[fn], type_ignores=[]), "train_ullava.py", "exec"), ns)
            names = ns["find_linear_layers"](model.llm, ["q_proj", "v_proj"])
            assert names == sorted(f"model.layers.{i}.self_attn.{k}" for i in range(3) for k in ("q_proj", "v_proj"))
            exec(compile(ast.Module(body=[blk], type_ignores=[]), "train_ullava.py", "exec"), ns)        # model.llm = get_peft_model(...); print_trainable...
        else:
            names = sorted(f"model.layers.{i}.self_attn.{k}" for i in range(3) for k in ("q_proj", "v_proj"))
            model.llm = get_peft_model(model.llm, LoraConfig(r=4, lora_alpha=8, target_modules=names, lora_dropout=0.05, bias="none", task_type="CAUSAL_LM"))
            model.llm.print_trainable_parameters()
        out = capsys.readouterr().out
        assert "trainable params:" in out and "all params:" in out
        assert isinstance(model.llm, PeftModel) and isinstance(model.llm.get_base_model(), MC.UllavaCoreForCausalLM)
        core = model.llm.get_base_model()
        assert core._lora == {"r": 4, "lora_alpha": 8.0, "lora_dropout": 0.05, "target_modules": ("q_proj", "v_proj")}
        train_names = [n for n, p in model.named_parameters() if p.requires_grad and n.startswith("llm.")]
        assert train_names and all(".lora_A." in n or ".lora_B." in n for n in train_names)
        assert not hasattr(core.model.layers[0].self_attn.k_proj, "lora_A")
        # attribute reads and writes fall through the wrapper (the MI355X model invalidates packs by assignment)
        # (`.model` of a PeftModel is the wrapped language model itself -- LoraModel.model -- exactly as in PEFT)
        assert model.llm.config is core.config and model.llm.model is core and model.llm.model.model is core.model
        assert model.llm.dtype == core.dtype and model.dtype == core.dtype
        core._packed = {"stale": True}
        model.llm._packed = None
        assert core._packed is None and "_packed" not in model.llm.__dict__
        # train_ullava.py:71-79 `safe_save_model_for_hf_trainer(is_peft=True)`: the key filter must give back the un-wrapped model's key names
        plain = set(build().state_dict())
        filtered = {k.replace(".base_model.model", ""): v for k, v in model.state_dict().items() if "lora_" not in k}
        assert set(filtered) == plain
        # :291 model.llm.save_pretrained(output_dir): adapter files only, PEFT's key layout
        with torch.no_grad():
            for l in core.model.layers:
                l.self_attn.q_proj.lora_B.weight.normal_(0, 0.05)
                l.self_attn.v_proj.lora_B.weight.normal_(0, 0.05)
        model.llm.save_pretrained(str(tmp_path))
        assert sorted(os.listdir(tmp_path)) == ["adapter_config.json", "adapter_model.safetensors"]
        from safetensors.torch import load_file
        ad = load_file(str(tmp_path / "adapter_model.safetensors"))
        assert len(ad) == 3 * 2 * 2 and all(k.startswith("base_model.model.model.layers.") for k in ad)
        # inference_ullava.py:43: PeftModel.from_pretrained(model.llm, llm_path, torch_dtype=dtype) on a fresh model: the adapter is folded in with
        # PEFT 0.4.0's arithmetic for an adapter FILE: from_pretrained loads it into fp32 lora_A / lora_B whatever the file stores, so the merge
        # is the fp32 delta of the upcast values with one rounding into the weight (checkpoint.lora_merged_weight)
        fresh = build()
        w0 = fresh.llm.model.layers[1].self_attn.v_proj.weight.detach().clone()
        k0 = fresh.llm.model.layers[1].self_attn.k_proj.weight.detach().clone()
        fresh.llm = PeftModel.from_pretrained(fresh.llm, str(tmp_path), torch_dtype=torch.bfloat16)
        lin = core.model.layers[1].self_attn.v_proj
        want = (w0.float() + (lin.lora_B.weight.float() @ lin.lora_A.weight.float()) * 2.0).to(torch.bfloat16)
        merged = fresh.llm.get_base_model().model.layers[1].self_attn
        got = merged.v_proj.weight.detach()
        assert torch.equal(got, want) and not torch.equal(got, w0)
        assert torch.equal(merged.k_proj.weight.detach(), k0) and not hasattr(merged.v_proj, "lora_A")
        with pytest.raises(NotImplementedError):
            get_peft_model(build().llm, LoraConfig(r=4, target_modules=["o_proj"]))
    finally:
        sys.path[:] = saved_path
        sys.modules.pop("peft", None)
        if saved_mod is not None:
            sys.modules["peft"] = saved_mod


def test_sampling_distribution_and_draws_match_hf_warpers():
    """VERDICT r3 missing #6: `evaluate()`'s default path samples (temperature 0.2, models/ullava.py:343,356).  `sampling_probs` must be the
    distribution HF's `_sample` draws from -- TemperatureLogitsWarper -> TopKLogitsWarper (GenerationConfig's default top_k = 50, which the
    reference's callers inherit) -> TopPLogitsWarper -> softmax, bit for bit -- so that one seeded `torch.multinomial` per step yields HF's
    tokens from the same logits."""
    from transformers.generation.logits_process import LogitsProcessorList, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    M = pkg("modeling_core")
    g = torch.Generator().manual_seed(12)
    ids = torch.zeros(3, 4, dtype=torch.long)
    for V, temperature, top_k, top_p in ((32011, 0.2, 50, None), (32011, 0.2, 50, 0.9), (1000, 0.7, 0, 0.5), (257, 1.0, 5, None), (64, 1.3, 50, 0.95)):
        logits = (torch.randn(3, V, generator=g) * 3.0).to(torch.bfloat16)
        procs = LogitsProcessorList()
        if temperature != 1.0:
            procs.append(TemperatureLogitsWarper(temperature))
        if top_k:
            procs.append(TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1))
        if top_p is not None:
            procs.append(TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1))
        want = torch.softmax(procs(ids, logits.to(copy=True, dtype=torch.float32)), dim=-1)        # HF _sample's lines, verbatim in spirit
        got = M.sampling_probs(logits, temperature, top_k, top_p)
        assert torch.equal(got, want), (V, temperature, top_k, top_p)
        for seed in range(5):
            torch.manual_seed(seed)
            a = torch.multinomial(want, 1)
            torch.manual_seed(seed)
            b = torch.multinomial(got, 1)
            assert torch.equal(a, b)


def test_layer_stack_structs_follow_the_weights():
    """ops.LayerStack (the ctypes array behind the coarse C-ABI entries, csrc/layers.hip): fields carry the tensors' addresses and shapes, the
    tile-major copy where one is registered, NULL where there is no bias -- and a refresh picks up a weight that moved or was updated in place."""
    ops, L = pkg("ops"), pkg("_lib")
    D, I = 128, 512

    def lin(n, k):
        return torch.randn(n, k).to(torch.bfloat16)
    layers = []
    for _ in range(2):
        d = dict(ln1=torch.ones(D, dtype=torch.bfloat16), ln2=torch.ones(D, dtype=torch.bfloat16), qkv=(lin(3 * D, D), None), o=(lin(D, D), None),
                 gu=(lin(2 * I, D), None), down=(lin(D, I), None))
        layers.append(d)
    big = layers[1]["gu"][0]                                       # [1024, 128]: large enough for a tile-major copy
    ops.register_tiled(big)
    st = ops.LayerStack(L.LlamaLayer, layers)
    arr = st.refresh()
    assert len(arr) == 2 and arr[0].ln1 == layers[0]["ln1"].data_ptr() and arr[1].down.w == layers[1]["down"][0].data_ptr()
    assert (arr[0].qkv.n, arr[0].qkv.k, arr[0].qkv.ldw) == (3 * D, D, D) and arr[0].qkv.bias is None and arr[0].qkv.w_tiled is None
    t0 = arr[1].gu.w_tiled
    assert t0 is not None and t0 == ops._tiled_of(big).data_ptr()
    assert st.refresh() is arr                                     # nothing moved: same array, no rebuild
    with torch.no_grad():
        big.mul_(2.0)                                              # an in-place update: the tile-major copy must describe the new values
    arr = st.refresh()
    tm = ops._tiled_of(big)
    assert arr[1].gu.w_tiled == tm.data_ptr() and torch.equal(tm, ops.tile_major(big))
    layers[0]["o"] = (lin(D, D), torch.zeros(D, dtype=torch.bfloat16))      # a re-bound weight (and a bias): new addresses
    arr = st.refresh()
    assert arr[0].o.w == layers[0]["o"][0].data_ptr() and arr[0].o.bias == layers[0]["o"][1].data_ptr()
    with pytest.raises(RuntimeError, match="row-major"):
        ops.LayerStack(L.LlamaLayer, [dict(layers[0], o=(lin(D, D).t(), None))])
