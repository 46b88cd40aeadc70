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
    declared = set(re.findall(r"\bint (ull_[a-z0-9_]+)\(", header))
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
