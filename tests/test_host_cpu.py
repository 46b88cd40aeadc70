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
