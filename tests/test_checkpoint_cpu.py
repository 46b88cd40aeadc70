"""CPU: HF-format checkpoint directories (config.json + single / sharded safetensors or torch-pickle) load into the module trees
with the reference's key names, and `save_pretrained` writes them back."""
import json
import os

import pytest
import torch

from conftest import pkg, load_fixture


def _tiny_core_cfg():
    fx = load_fixture("g1_core_tiny_bf16.pt")
    cd = fx["cfg"]
    return dict(vision_config=cd["vision_config"], vision_hidden_layer=-2, mm_token_ids=cd["mm_token_ids"], vocab_size=cd["vocab_size"],
                hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"], num_hidden_layers=cd["num_hidden_layers"],
                num_attention_heads=cd["num_attention_heads"], projector_type=cd["projector_type"])


def _randomize(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.state_dict().values():                   # parameters AND buffers (module buffers are allocated uninitialised)
            p.copy_(torch.randn(p.shape, generator=g).to(p.dtype))


@pytest.mark.parametrize("safe,shard_bytes", [(True, 5 << 30), (True, 20000), (False, 5 << 30), (False, 20000)])
def test_core_save_load_roundtrip(tmp_path, safe, shard_bytes):
    C, M, K = pkg("configuration"), pkg("modeling_core"), pkg("checkpoint")
    m = M.UllavaCoreForCausalLM(C.UllavaCoreConfig(**_tiny_core_cfg()))
    _randomize(m, 1)
    d = str(tmp_path / "ckpt")
    m.save_pretrained(d, max_shard_bytes=shard_bytes, safe_serialization=safe)
    files = K.shard_files(d)
    assert (len(files) > 1) == (shard_bytes < 1 << 20)
    # the directory uses the reference's parameter names: the CLIP tower nests under `.vision_model.`
    keys = set()
    for sh in K.iter_shards(d):
        keys |= set(sh)
    assert any(k.startswith("vision_encoder.vision_model.encoder.layers.0.") for k in keys)
    assert not any(k.startswith("vision_encoder.encoder.") for k in keys)
    m2 = M.UllavaCoreForCausalLM.from_pretrained(d, torch_dtype=torch.bfloat16)
    a, b = m.state_dict(), m2.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert m2.config.to_dict() == m.config.to_dict()


def test_core_load_reports_mismatches(tmp_path):
    C, M, K = pkg("configuration"), pkg("modeling_core"), pkg("checkpoint")
    m = M.UllavaCoreForCausalLM(C.UllavaCoreConfig(**_tiny_core_cfg()))
    d = str(tmp_path / "ckpt")
    m.save_pretrained(d, safe_serialization=False)
    sd = torch.load(os.path.join(d, "pytorch_model.bin"), weights_only=True)
    sd.pop("lm_head.weight")
    sd["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.zeros(4)         # ignored, like the reference's persistent=False buffer
    sd["bogus.weight"] = torch.zeros(1)
    torch.save(sd, os.path.join(d, "pytorch_model.bin"))
    with pytest.raises(RuntimeError, match="lm_head.weight"):
        M.UllavaCoreForCausalLM.from_pretrained(d)
    m3 = M.UllavaCoreForCausalLM(C.UllavaCoreConfig(**_tiny_core_cfg()))
    missing, unexpected = K.load_into(m3, d, strict=False)
    assert missing == ["lm_head.weight"] and unexpected == ["bogus.weight"]
    # `--dtype fp32` (inference_ullava.py:25,164-168): round 6 has an fp32 kernel build -- the checkpoint's 16-bit values load exactly
    m32 = M.UllavaCoreForCausalLM.from_pretrained(d, torch_dtype=torch.float32, strict=False)
    assert m32.dtype == torch.float32 and torch.equal(m32.model.norm.weight, m3.model.norm.weight.float())
    with pytest.raises(NotImplementedError):
        M.UllavaCoreForCausalLM.from_pretrained(d, torch_dtype=torch.float64)


def test_ullava_roundtrip_and_missing_sam_encoder(tmp_path):
    C, M, K = pkg("configuration"), pkg("modeling_ullava"), pkg("checkpoint")
    sam = dict(embed_dim=32, depth=2, num_heads=2, global_attn_indexes=[1], img_size=64, patch_size=16, window_size=2)
    cfg = C.UllavaConfig(llm_config=_tiny_core_cfg(), seg_token_idx=90, loc_token_idx=91, out_dim=256, sam_config=sam)
    m = M.UllavaForCausalLM(cfg)
    _randomize(m, 2)
    d = str(tmp_path / "ckpt")
    m.save_pretrained(d, max_shard_bytes=1 << 18)
    m2 = M.UllavaForCausalLM.from_pretrained(d)
    a, b = m.state_dict(), m2.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    # a stage-2 directory without the frozen SAM image encoder still loads (load_visual_checkpoint supplies it)
    for fn in K.shard_files(d):
        from safetensors.torch import load_file, save_file
        sh = {k: v for k, v in load_file(fn).items() if not k.startswith("visual_model.image_encoder.")}
        if sh:
            save_file(sh, fn, metadata={"format": "pt"})
        else:
            os.remove(fn)
            idx = os.path.join(d, "model.safetensors.index.json")
            j = json.load(open(idx))
            j["weight_map"] = {k: v for k, v in j["weight_map"].items() if v != os.path.basename(fn)}
            json.dump(j, open(idx, "w"))
    m3 = M.UllavaForCausalLM.from_pretrained(d)
    c = m3.state_dict()
    assert all(torch.equal(a[k], c[k]) for k in a if not k.startswith("visual_model.image_encoder."))
