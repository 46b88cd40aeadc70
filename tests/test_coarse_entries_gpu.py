"""GPU: the coarse C-ABI entries (include/ullava_hip.h "coarse entries", csrc/layers.hip: one call enqueues a whole stack of layers) against
the per-op path they replace -- the same launches with the same dispatch rules, so every output must be BIT-IDENTICAL.  Full widths (LLaMA-7B
4096 / 11008 / 32 heads, ViT-L/14 1024 / 4096 / 16 heads, SAM ViT-H 1280 / 5120 / 16 heads at 1024 x 1024), few layers."""
import os
import sys

import pytest
import torch

from helpers import pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _core(image, n_llama=3, n_clip=4, dtype=torch.bfloat16):
    import bench
    C, M = pkg("configuration"), pkg("modeling_core")
    cfg = C.UllavaCoreConfig(vision_config=dict(image_size=image, patch_size=14, num_hidden_layers=n_clip), vision_hidden_layer=-2, projector_type="mlp",
                             mm_token_ids=dict(bench.MM), vocab_size=32011, num_hidden_layers=n_llama)
    model = M.UllavaCoreForCausalLM(cfg, device=DEV)
    bench.init_random_(model, 21)
    if dtype != torch.bfloat16:
        model = model.to(dtype)
    model.strict_checks = False
    return model, cfg, bench


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,image", [(3, 336), (1, 224)])
def test_prefill_and_clip_coarse_equal_per_op(batch, image, dtype):
    """B = 3 x S = 643 (256 x 256 kernel, tile-major weights, stream-K tail) and B = 1 x S = 323 (128 x 128 kernel): logits and every hidden
    state equal bit for bit; ragged right padding in the larger batch."""
    ops = pkg("ops")
    model, cfg, bench = _core(image, dtype=dtype)
    images, ids, mask = bench.make_inputs(cfg, batch, 64, torch.device(DEV), 7, ragged=batch > 1)
    images = images.to(dtype)
    with torch.no_grad():
        assert ops.coarse_ok()
        a = model.forward(input_ids=ids, attention_mask=mask, images=images, output_hidden_states=True)
        a2 = model.forward(input_ids=ids, attention_mask=mask, images=images)
        feat_a = model.encode_image(images)
        with ops.per_op_layers():
            assert not ops.coarse_ok()
            b = model.forward(input_ids=ids, attention_mask=mask, images=images, output_hidden_states=True)
            feat_b = model.encode_image(images)
    assert torch.equal(feat_a, feat_b), "CLIP tower: coarse entry != per-op path"
    assert len(a.hidden_states) == len(b.hidden_states) == cfg.num_hidden_layers + 1
    for i, (x, y) in enumerate(zip(a.hidden_states, b.hidden_states)):
        assert torch.equal(x, y), f"hidden state {i}: coarse entry != per-op path"
    assert torch.equal(a.logits, b.logits) and torch.equal(a2.logits, b.logits)
    assert bool(torch.isfinite(a.logits.float()).all())


@pytest.mark.parametrize("batch", [1, 2, 3])
def test_decode_coarse_equal_per_op(batch):
    """KV-cached greedy generation, 6 steps at batch 1 / 2 (GEMV) / 3 (skinny MFMA GEMM): token ids and the last-step hidden states equal."""
    ops = pkg("ops")
    model, cfg, bench = _core(224)
    images, ids, mask = bench.make_inputs(cfg, batch, 32, torch.device(DEV), 9)
    kw = dict(input_ids=ids, images=images, max_new_tokens=6, do_sample=False, use_cache=True, eos_token_id=-1, output_hidden_states=True,
              return_dict_in_generate=True, keep_last_step_only=True)
    with torch.no_grad():
        a = model.generate(**kw)
        with ops.per_op_layers():
            b = model.generate(**kw)
    assert torch.equal(a.sequences, b.sequences) and a.sequences.shape[1] == ids.shape[1] + 6
    assert torch.equal(a.hidden_states[-1][-1], b.hidden_states[-1][-1]), "decode steps: coarse entry != per-op path"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sam_blocks_coarse_equal_per_op(dtype):
    """SAM ViT-H width, 3 blocks (window, global, window), 2 images of 1024 x 1024: image embeddings equal bit for bit."""
    ops, S, C = pkg("ops"), pkg("sam"), pkg("configuration")
    scfg = C.SamConfig(embed_dim=1280, depth=3, num_heads=16, global_attn_indexes=[1])
    holder = S.build_sam_holder(scfg, device=DEV, dtype=dtype)
    g = torch.Generator(device="cuda").manual_seed(31)
    with torch.no_grad():
        for n, p in holder.named_parameters():
            if p.dim() == 1 and "norm" in n and n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, device=DEV, generator=g) * (0.02 if p.dim() > 1 else 0.1)).to(dtype))
        sam = S.SamEngine(holder, scfg)
        x = torch.randn(2, 3, 1024, 1024, device=DEV, generator=g).to(dtype)
        a = sam.encode(x)
        with ops.per_op_layers():
            b = sam.encode(x)
    assert tuple(a.shape) == (2, 4096, 256) and bool(torch.isfinite(a.float()).all())
    assert torch.equal(a, b), "SAM blocks: coarse entry != per-op path"
