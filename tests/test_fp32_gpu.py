"""GPU: the fp32 build of the inference path (`--dtype fp32` of inference_ullava.py:25,164-168; csrc/f32.hip) -- plain fp32 kernels on the exact
fp32 matrix instruction, against torch fp32 references per kernel and against the REFERENCE's fp32 fixtures (G1 / G7 / G8: tiny models run by
/root/reference in fp32).  fp32 has no rounding points to reproduce: the tolerance is the fp32 summation-order noise (1e-5 of the tensor's
maximum; token ids, [SEG] / [LOC] bookkeeping and shapes exact)."""
import os
import sys

import pytest
import torch

from helpers import pkg, load_fixture, fixture_sd, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32 = torch.float32
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _r(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale)


@pytest.mark.parametrize("M,N,K,act,bias,res", [(300, 1031, 72, "quick_gelu", True, False), (1, 777, 72, None, True, False), (5, 520, 64, "gelu", True, True),
                                                (129, 40, 73, "relu", False, True), (1500, 8, 192, None, True, False), (64, 64, 4, None, False, False),
                                                (257, 4096, 1024, None, True, True)])
def test_gemm_f32_against_torch(M, N, K, act, bias, res):
    ops = pkg("ops")
    x, w = _r(M, K, seed=1), _r(N, K, seed=2, scale=K ** -0.5)
    b = _r(N, seed=3) if bias else None
    r = _r(M, N, seed=4) if res else None
    ref = x.double() @ w.double().T
    if bias:
        ref = ref + b.double()
    ref = {None: lambda t: t, "relu": torch.relu, "gelu": torch.nn.functional.gelu, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t)}[act](ref)
    if res:
        ref = ref + r.double()
    out = ops.linear(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), act=act, residual=None if r is None else r.to(DEV))
    assert out.dtype == F32 and tuple(out.shape) == (M, N)
    e = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert e <= 2e-6, e


def test_gemm_f32_strided_rows_and_swiglu():
    ops, MC = pkg("ops"), pkg("modeling_core")
    x_full = _r(70, 3, 96, seed=5)
    x = x_full[:, 1, :].to(DEV)                                   # rows 288 elements apart
    gate, up = _r(64, 96, seed=6, scale=0.1), _r(64, 96, seed=7, scale=0.1)
    out = ops.linear(x_full.to(DEV)[:, 1, :], MC.interleave_gate_up(gate, up).to(DEV), swiglu=True)
    ref = torch.nn.functional.silu(x_full[:, 1, :].double() @ gate.double().T) * (x_full[:, 1, :].double() @ up.double().T)
    assert tuple(out.shape) == (70, 64)
    assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) <= 2e-6


def _attn_ref(q, k, v, causal, key_mask, scale):
    # hf eager_attention_forward in fp64: finfo(float32).min ADDED to masked scores, softmax, P V
    B, H, Sq, hd = q.shape
    Sk = k.shape[2]
    s = (q.double() @ k.double().transpose(-1, -2)) * scale
    allowed = torch.ones(B, 1, Sq, Sk, dtype=torch.bool)
    if causal:
        allowed = allowed & (torch.arange(Sk)[None, :] <= torch.arange(Sq)[:, None] + (Sk - Sq))[None, None]
    if key_mask is not None:
        allowed = allowed & (key_mask[:, None, None, :] != 0)
    s = torch.where(allowed, s, torch.full_like(s, float(torch.finfo(torch.float32).min)))
    return torch.softmax(s, dim=-1) @ v.double()


@pytest.mark.parametrize("B,H,Sq,Sk,hd,causal,masked", [(2, 3, 65, 65, 64, True, False), (1, 2, 17, 17, 16, False, False), (2, 2, 320, 320, 128, True, True),
                                                         (1, 2, 1, 130, 80, True, False), (2, 1, 3, 1025, 128, True, True), (1, 3, 130, 70, 32, False, True)])
def test_attention_f32_against_torch(B, H, Sq, Sk, hd, causal, masked):
    """V as rows of a fused q|k|v-style buffer (vt_len = 0) AND as the key-permuted V^T image (the KV cache's layout): both equal fp64 softmax(QK^T)V."""
    ops = pkg("ops")
    q, k, v = _r(B, H, Sq, hd, seed=1), _r(B, H, Sk, hd, seed=2), _r(B, H, Sk, hd, seed=3)
    km = None
    if masked:
        km = torch.ones(B, Sk, dtype=torch.int32)
        km[0, Sk - 5:] = 0
        if B > 1:
            km[1, :3] = 0
    ref = _attn_ref(q, k, v, causal, km, hd ** -0.5).float()
    qd, kd, vd = (t.permute(0, 2, 1, 3).contiguous().to(DEV) for t in (q, k, v))            # [B, S, H, hd]: token-major like the model's buffers
    D = H * hd
    for form in ("rows", "vt"):
        out = torch.empty(B * Sq, D, device=DEV, dtype=F32)
        kw = dict(key_mask=None if km is None else km.to(DEV), causal=causal, scale_mode=1, scale=hd ** -0.5)
        if form == "rows":
            ops.attention(qd, kd, vd, out, B, H, Sq, Sk, hd, (Sq * D, hd, D), (Sk * D, hd, D), (Sq * D, hd, D), v_strides=(Sk * D, hd, D), **kw)
        else:
            vt = ops.transpose_v(vd, Sk * D, D, B, Sk, H, hd)
            assert vt.dtype == F32 and vt.shape[-1] % 64 == 0
            ops.attention(qd, kd, vt, out, B, H, Sq, Sk, hd, (Sq * D, hd, D), (Sk * D, hd, D), (Sq * D, hd, D), **kw)
        got = out.view(B, Sq, H, hd).permute(0, 2, 1, 3).cpu()
        # rows with NO attendable key (a left-padded position under the causal mask) are don't-care: every score is finfo.min and the
        # reference's uniform average runs over all Sk keys, the kernels' over the keys up to the causal limit -- nothing reads those rows
        allowed = torch.ones(B, Sq, Sk, dtype=torch.bool)
        if causal:
            allowed &= (torch.arange(Sk)[None, :] <= torch.arange(Sq)[:, None] + (Sk - Sq))[None]
        if km is not None:
            allowed &= (km[:, None, :] != 0)
        valid = allowed.any(-1)[:, None, :, None].expand_as(ref)
        e = float(((got - ref).abs() * valid).max() / ref.abs().max())
        assert e <= 5e-6, (form, e)


def _core_model(fx, dtype=F32):
    C, M = pkg("configuration"), pkg("modeling_core")
    cd = fx["cfg"]
    cfg = C.UllavaCoreConfig(vision_config=cd["vision_config"], vision_hidden_layer=cd["vision_hidden_layer"], projector_type=cd["projector_type"],
                             projector_from_scratch=bool(cd.get("projector_from_scratch", False)), mm_token_ids=cd["mm_token_ids"],
                             vocab_size=cd["vocab_size"], hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                             num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"], rms_norm_eps=cd["rms_norm_eps"],
                             rope_theta=cd["rope_theta"])
    model = M.UllavaCoreForCausalLM(cfg, device=DEV, dtype=dtype)
    model.load_state_dict(fixture_sd(fx, dtype), strict=True)
    return model


def test_core_fixture_g1_fp32():
    """G1 / G2 in fp32: the reference's own fp32 run of the tiny core model (UllavaCoreForCausalLM.forward, models/ullava_core.py:279-355)."""
    fx = load_fixture("g1_core_tiny_fp32.pt")
    assert fx["dtype"] == "torch.float32"
    model = _core_model(fx)
    assert model.dtype == F32
    ids, mask, images = fx["input_ids"].to(DEV), fx["attention_mask"].to(DEV), fx["images"].to(DEV)
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask, images=images, output_hidden_states=True)
    assert out.logits.dtype == F32
    valid = fx["attention_mask"].bool()
    e = rel_err(out.logits.cpu()[valid], fx["logits"][valid])
    eh = [rel_err(h.cpu()[valid], r[valid]) for h, r in zip(out.hidden_states, fx["hidden_states"])]
    ef = rel_err(model.encode_image(images), fx["image_features"])
    print(f"G1 fp32: logits err {e:.2e}, hidden states {[f'{x:.1e}' for x in eh]}, CLIP features {ef:.2e}")
    assert e <= 1e-5 and max(eh) <= 1e-5 and ef <= 1e-5
    with torch.no_grad():
        nc = model.generate(input_ids=fx["greedy_prompt"].to(DEV), images=images[:1], max_new_tokens=8, do_sample=False, use_cache=False)
        kv = model.generate(input_ids=fx["greedy_prompt"].to(DEV), images=images[:1], max_new_tokens=8, do_sample=False, use_cache=True)
        lp = model.generate(input_ids=fx["leftpad_ids"].to(DEV), attention_mask=fx["leftpad_mask"].to(DEV), images=images[:1], max_new_tokens=6,
                            do_sample=False, use_cache=True)
    assert torch.equal(nc.cpu(), fx["greedy_sequences"]), "fp32 greedy ids (no KV cache) differ from the reference's"
    assert torch.equal(kv.cpu(), fx["greedy_sequences"]), "fp32 greedy ids (KV cache) differ from the reference's"
    assert torch.equal(lp.cpu(), fx["leftpad_sequences"]), "fp32 greedy ids on a left-padded prompt differ from the reference's"


def test_mask_decoder_fixture_g7_fp32():
    """G7 in fp32: prompt encoder + two-way MaskDecoder + postprocess (mask_decoder.py:75-164, transformer.py:62-242, sam.py:137-172)."""
    from oracle import ullava_oracle as O
    C, S = pkg("configuration"), pkg("sam")
    fx = load_fixture("g7_sam_decoder_fp32.pt")
    cfg = C.SamConfig(depth=0)
    holder = S.build_sam_holder(cfg, device=DEV, dtype=F32)
    sd_full = fixture_sd(fx, F32)
    missing = holder.load_state_dict({k[len("visual_model."):]: v for k, v in sd_full.items()}, strict=False)
    assert not missing.unexpected_keys
    eng = S.SamEngine(holder, cfg)
    pe = eng.dense_pe().cpu()
    ref_pe = O.dense_pe(sd_full, (64, 64))[0].permute(1, 2, 0).reshape(4096, 256)
    assert float((pe - ref_pe).abs().max()) <= 1e-6
    g = torch.Generator().manual_seed(fx["image_embedding_seed"])
    emb = torch.randn(1, 256, 64, 64, generator=g)
    emb_tm = emb[0].permute(1, 2, 0).reshape(4096, 256).contiguous().to(DEV)
    for case in fx["cases"]:
        with torch.no_grad():
            masks, iou = eng.decode(emb_tm, case["text_embeds"][:, 0].to(DEV))
        assert masks.dtype == F32
        st = case["low_res_stride"]
        e = float((masks[:, 0:1, ::st, ::st].cpu() - case["low_res_masks"].float()).abs().max()) / case["low_res_max"]
        e_iou = rel_err(iou[:, 0:1], case["iou"])
        print(f"G7 fp32 n={case['n']}: mask-logit err vs the reference {e:.2e}, iou err {e_iou:.2e}")
        assert e <= 2e-5 and e_iou <= 2e-5
        post = eng.postprocess(masks[:, 0].contiguous(), (768, 1024), (480, 640)).cpu()
        ref_post = O.postprocess_masks(masks[:, 0:1].cpu(), (768, 1024), (480, 640))[:, 0]
        assert post.dtype == F32 and float((post - ref_post).abs().max()) <= 1e-5 * float(ref_post.abs().max())


def test_full_forward_fixture_g8_fp32():
    """G8 in fp32: UllavaForCausalLM.forward(inference=True) (models/ullava.py:152-268) with the shrunk SAM encoder: [SEG] / [LOC] bookkeeping exact,
    logits / SAM embedding / masks / boxes at fp32 summation-order noise."""
    C, M = pkg("configuration"), pkg("modeling_ullava")
    fx = load_fixture("g8_full_tiny_fp32.pt")
    cfg, cd = fx["cfg"], fx["cfg"]["llm"]
    ucfg = C.UllavaConfig(llm_config=dict(vision_config=cd["vision_config"], vision_hidden_layer=cd["vision_hidden_layer"], projector_type="mlp",
                                          projector_from_scratch=bool(cd.get("projector_from_scratch", False)), mm_token_ids=cd["mm_token_ids"],
                                          vocab_size=cd["vocab_size"], hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                                          num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"]),
                          seg_token_idx=cfg["seg_token_idx"], loc_token_idx=cfg["loc_token_idx"], sam_config=dict(cfg["sam"]))
    model = M.UllavaForCausalLM(ucfg, device=DEV, dtype=F32)
    model.load_state_dict(fixture_sd(fx, F32), strict=True)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g)
    with torch.no_grad():
        out = model(images_sam=images_sam.to(DEV), images=fx["images"].to(DEV), input_ids=fx["input_ids"].to(DEV), labels=None,
                    attention_mask=fx["attention_mask"].to(DEV), mask_list=[None, None], size_list=fx["size_list"], resize_list=fx["resize_list"],
                    bbox_list=[None, None], inference=True)
        emb = model.get_visual_embs(images_sam.to(DEV)).cpu()
    assert sorted(out.keys()) == fx["dict_keys"]
    assert [m.shape[0] for m in out["pred_masks"]] == [2, 1] and [b.shape[0] for b in out["pred_boxes"]] == [1, 2]
    valid = fx["attention_mask"].bool()
    el = rel_err(out["logits"].cpu()[valid], fx["logits"][valid])
    ee = rel_err(emb[:, ::16, ::4, ::4], fx["image_embeddings_sample"])
    print(f"G8 fp32: logits err {el:.2e}, SAM embedding err {ee:.2e}")
    assert out["logits"].dtype == F32 and el <= 1e-5 and ee <= 2e-5
    for i in range(2):
        assert out["pred_masks"][i].dtype == F32 and tuple(out["pred_masks"][i].shape) == tuple(fx["pred_mask_shapes"][i])
        em = float((out["pred_masks"][i].cpu()[:, ::8, ::8] - fx["pred_mask_samples"][i]).abs().max()) / float(fx["low_res_masks"][i].float().abs().max())
        eb = rel_err(out["pred_boxes"][i], fx["pred_boxes"][i])
        print(f"G8 fp32 sample {i}: mask err {em:.2e} box err {eb:.2e}")
        assert em <= 5e-5 and eb <= 2e-5


def test_evaluate_fp32_against_the_oracle():
    """`evaluate(temperature=0)` of an fp32 model (models/ullava.py:335-434: generate -> [SEG] / [LOC] states -> SAM decode -> postprocess), with and
    without the KV cache, against the oracle's fp32 evaluate on the same weights (the G11 model and inputs, weights promoted from bf16): ids equal,
    masks / boxes at fp32 noise."""
    from oracle import ullava_oracle as O
    C, M = pkg("configuration"), pkg("modeling_ullava")
    fx = load_fixture("g11_evaluate_bf16.pt")
    cfg, cd = fx["cfg"], fx["cfg"]["llm"]
    ucfg = C.UllavaConfig(llm_config=dict(vision_config=cd["vision_config"], vision_hidden_layer=cd["vision_hidden_layer"], projector_type="mlp",
                                          projector_from_scratch=bool(cd.get("projector_from_scratch", False)), mm_token_ids=cd["mm_token_ids"],
                                          vocab_size=cd["vocab_size"], hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                                          num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"]),
                          seg_token_idx=cfg["seg_token_idx"], loc_token_idx=cfg["loc_token_idx"], sam_config=dict(cfg["sam"]))
    sd = {k: v.float() for k, v in fixture_sd(fx, torch.bfloat16).items()}
    model = M.UllavaForCausalLM(ucfg, device=DEV, dtype=F32)
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g).to(torch.bfloat16)[:1].float()
    images = fx["images"].float()
    want_seq, want_masks, want_boxes = O.ullava_evaluate(sd, cfg, images_sam, images, fx["input_ids"], [fx["size"]], [fx["resize"]], max_new_tokens=6)
    for use_cache in (False, True):
        model.llm.config.use_cache = use_cache
        seq, masks, boxes = model.evaluate(images_sam.to(DEV), images.to(DEV), fx["input_ids"].to(DEV), [fx["size"]], [fx["resize"]], max_new_tokens=6,
                                           temperature=0)
        assert torch.equal(seq.cpu(), want_seq), (seq.tolist(), want_seq.tolist())
        assert masks[0].dtype == F32 and tuple(masks[0].shape) == tuple(want_masks[0].shape)
        em, eb = rel_err(masks[0], want_masks[0]), rel_err(boxes[0], want_boxes[0]) if want_boxes[0].numel() else 0.0
        print(f"fp32 evaluate(use_cache={use_cache}): mask err {em:.2e}, box err {eb:.2e}, ids {seq[0, fx['input_ids'].shape[1]:].tolist()}")
        assert em <= 5e-5 and eb <= 5e-5


@pytest.mark.parametrize("name", ["g3_video_bf16.pt", "g4_mixed_bf16.pt"])
def test_video_and_mixed_batches_fp32_against_the_oracle(name):
    """The video branch (per-frame ViT, temporal + spatial pooling, models/ullava_core.py:160-180,248-269) and a mixed batch with a text-only row and
    right padding (:205-226) in fp32: the G3 / G4 models and inputs with the weights promoted to fp32, against the oracle's fp32 forward, plus the
    shifted CE loss (:327-338)."""
    from oracle import ullava_oracle as O
    fx = load_fixture(name)
    model = _core_model(fx)
    sd = {k: v.float() for k, v in fixture_sd(fx, torch.bfloat16).items()}
    model.load_state_dict(sd, strict=True)
    ids = fx["input_ids"]
    mask = fx.get("attention_mask", torch.ones_like(ids))
    kw_o = dict(videos=fx["videos"].float()) if "videos" in fx else dict(images=fx["images"].float())
    labels = ids.clone()
    labels[mask == 0] = -100
    with torch.no_grad():
        out = model(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), labels=labels.to(DEV), output_hidden_states=True,
                    **{k: v.to(DEV) for k, v in kw_o.items()})
    want = O.core_forward(sd, fx["cfg"], ids, mask, kw_o.get("images"), kw_o.get("videos"), labels=labels)
    valid = mask.bool()
    e = rel_err(out.logits.cpu()[valid], want["logits"][valid])
    eh = rel_err(out.hidden_states[-1].cpu()[valid], want["hidden_states"][-1][valid])
    el = abs(float(out.loss) - float(want["loss"])) / abs(float(want["loss"]))
    print(f"{name} in fp32: logits err {e:.2e}, last hidden {eh:.2e}, loss {float(out.loss):.6f} vs {float(want['loss']):.6f}")
    assert out.logits.dtype == F32 and e <= 1e-5 and eh <= 1e-5 and el <= 1e-5
