#!/usr/bin/env python
"""G15 / G16: the REFERENCE ITSELF at full size on this container's CPU (build container only; needs /root/reference).

    python tests/golden/gen_golden_full_depth.py [g15_bf16 g15_fp16 g16_bf16 g16_fp16]

  G15 = BASELINE.json configs[0] (C1): ViT-L/14-224 (24 layers, 23 used) + LLaMA-7B (32 layers), V = 32011, one 224 x 224 image + 32-token
        prompt, S = 291 -- `UllavaCoreForCausalLM.forward` (models/ullava_core.py:279-355).
  G16 = batch-1 C3: the same LLM + SAM ViT-H (32 blocks, d = 1280, 1024 x 1024) + prompt encoder + two-way MaskDecoder + postprocess, three
        [SEG] / [LOC] rounds, S = 379 -- `UllavaForCausalLM.forward(inference=True)` (models/ullava.py:152-268, image_encoder.py:110-125).

Weights: `u-llava_amd/weights.seeded_tensor(name, shape, seed, hf_init=True)` per tensor -- the CPU-keyed generator, so every machine regenerates
the same 7 B parameters from (name, shape, seed); rounded ONCE fp32 -> bf16 / fp16 (`llm.*` tensors of the full model are generated under the core model's names,
`weights.seeded_state_dict(strip_prefix="llm.")`: one generated LLM serves G15 and G16).  The fp32 "truth" is the reference run in fp32 on those
16-bit-rounded weights.  For every run (16-bit and fp32) the oracle is run on the same weights and inputs and must be torch.equal with the
reference on every output -- this pins the oracle to the reference at REAL depth and width, not only on the tiny fixtures.

A fixture holds: config, seed, shape table, inputs (or their seeds), sha256 digests of the reference's full outputs, and SAMPLES of the reference's
16-bit outputs together with the fp32 truth at the same places (whole logit rows at a few positions, strided columns at all positions, strided
hidden-state / embedding / mask samples), plus per-position top-k / noise / margin tables for the margin-gated token-id comparison.  The HIP
path cannot be bit-equal to a CPU at this depth (fp32 summation order), so the GPU tests compare its error against the fp32 truth with the
REFERENCE's own 16-bit error on the same samples, and token ids / mask signs wherever the margin clears the 16-bit noise.
"""
import gc
import hashlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G          # noqa: E402  (sets up the torchvision stub, imports the reference, the oracle and the weight generator)
import torch                    # noqa: E402

O, W = G.O, G.W
SEED = 15
MM = dict(IMG_START=32001, IMG_END=32002, IMG_PATCH=32003, VID_START=32004, VID_END=32005, VID_PATCH=32006)     # = bench.MM
SEG, LOC = 32007, 32008
V = 32011
ROWS = 12               # whole logit rows kept
COL_STRIDE = 64         # logit columns kept at every position (V / 64 = 501 columns)
HID_LAYERS = (0, 8, 16, 24, 32)
HID_STRIDE = 32


def log(*a):
    print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)


def digest(t):
    flat = t.detach().contiguous().view(-1)
    raw = flat.view(torch.int16 if t.element_size() == 2 else torch.int32).numpy().tobytes()
    return dict(shape=list(t.shape), dtype=str(t.dtype), sha256=hashlib.sha256(raw).hexdigest())


def llm_cfg_dict():
    return dict(hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, intermediate_size=11008, vocab_size=V, rms_norm_eps=1e-6,
                rope_theta=10000.0, vision_hidden_layer=-2, projector_type="mlp", mm_token_ids=dict(MM),
                vision_config=dict(hidden_size=1024, num_attention_heads=16, num_hidden_layers=24, intermediate_size=4096, image_size=224,
                                   patch_size=14, num_channels=3, layer_norm_eps=1e-5))


SAM_H = dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=[7, 15, 23, 31], window_size=14, patch_size=16, img_size=1024,
             out_chans=256)


def build_reference(full, dtype):
    """The reference model built on the meta device (HF's random init of 7 B parameters on this CPU takes minutes and is thrown away anyway),
    materialised in `dtype`, every parameter / persistent buffer filled from the seeded generator, every NON-persistent buffer rebuilt the way
    the constructors build it."""
    cd = llm_cfg_dict()
    with torch.device("meta"):
        if full:
            llm_cfg = dict(vision_config=dict(cd["vision_config"]), vision_hidden_layer=-2, projector_type="mlp", projector_from_scratch=False,
                           mm_token_ids=dict(MM), hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                           num_key_value_heads=32, vocab_size=V, rms_norm_eps=1e-6, attn_implementation="eager")
            cfg = G.UllavaConfig(llm_config=llm_cfg, seg_token_idx=SEG, loc_token_idx=LOC, out_dim=256)
            cfg.llm_config.vision_config._attn_implementation = "eager"
            m = G.UllavaForCausalLM(cfg)
            core = m.llm
        else:
            m = core = G.build_ref_core(cd)
    m = m.eval().to(dtype).to_empty(device="cpu")
    assert core.config._attn_implementation == "eager" and core.vision_encoder.config._attn_implementation == "eager"
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    n = 0
    with torch.no_grad():
        for k, v in m.state_dict().items():                     # parameters AND persistent buffers (SAM's Gaussian PE matrix)
            # (the full model's llm.* tensors are generated under the core model's names: G15 and G16 hold the SAME LLaMA-7B + CLIP weights)
            v.copy_(W.seeded_tensor(k[4:] if k.startswith("llm.") else k, v.shape, SEED, torch.float32, hf_init=True).to(dtype))
            n += v.numel()
    persistent = set(shapes)
    for name, buf in list(m.named_buffers()):
        if name in persistent:
            continue
        mod = m.get_submodule(name.rsplit(".", 1)[0]) if "." in name else m
        leaf = name.rsplit(".", 1)[-1]
        if leaf in ("inv_freq", "original_inv_freq"):
            inv, _ = mod.compute_default_rope_parameters(mod.config)
            setattr(mod, leaf, inv.float().clone())            # fp32, as a from_pretrained(torch_dtype=...) load leaves it (see gen_golden.load_seeded)
        elif leaf == "position_ids":
            setattr(mod, leaf, torch.arange(buf.shape[-1]).expand((1, -1)))
        elif leaf in ("pixel_mean", "pixel_std"):               # Sam's input normalisation constants (build_sam.py:100-101; unused on this path)
            vals = [123.675, 116.28, 103.53] if leaf == "pixel_mean" else [58.395, 57.12, 57.375]
            setattr(mod, leaf, torch.tensor(vals).view(-1, 1, 1).to(buf.dtype))
        else:
            raise RuntimeError(f"non-persistent buffer {name} has no rebuild rule")
    log(f"reference built: {n / 1e9:.3f} B parameters in {dtype}")
    return m, shapes


def c1_inputs(dtype):
    g = torch.Generator().manual_seed(SEED + 100)
    ids = torch.tensor([[1, MM["IMG_START"]] + [MM["IMG_PATCH"]] * 256 + [MM["IMG_END"]] + torch.randint(5, 32000, (32,), generator=g).tolist()])
    img = torch.randn(1, 3, 224, 224, generator=g).to(dtype)
    return ids, torch.ones_like(ids), img


def c3_inputs(dtype):
    g = torch.Generator().manual_seed(SEED + 200)
    ids = torch.tensor([[1, MM["IMG_START"]] + [MM["IMG_PATCH"]] * 256 + [MM["IMG_END"]] + torch.randint(5, 32000, (120,), generator=g).tolist()])
    S = ids.shape[1]
    for r in range(3):                                          # three RES rounds: ... [SEG] .... [LOC] ...
        ids[0, S - 10 - 40 * r] = SEG
        ids[0, S - 5 - 40 * r] = LOC
    img = torch.randn(1, 3, 224, 224, generator=g).to(dtype)
    images_sam = torch.randn(1, 3, 1024, 1024, generator=g).to(dtype)
    return ids, torch.ones_like(ids), img, images_sam, [(480, 640)], [(768, 1024)]


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def logits_record(ref16, truth):
    """ref16 / truth: [S, V].  Per-position tables for the margin gate + samples."""
    r, t = ref16.float(), truth.float()
    S = r.shape[0]
    sigma = (r - t).pow(2).mean(-1).sqrt() * 2.0 ** 0.5        # noise of a logit DIFFERENCE at that position (bench.parity_stats)
    top_t = t.topk(8, dim=-1)
    top_r = r.topk(8, dim=-1)
    rows = torch.linspace(0, S - 1, ROWS).round().long()
    rec = dict(sigma=sigma, truth_top_values=top_t.values, truth_top_ids=top_t.indices.int(), ref_top_values=ref16.topk(8, dim=-1).values,
               ref_top_ids=top_r.indices.int(), ref_argmax=r.argmax(-1).int(), truth_argmax=t.argmax(-1).int(),
               rows=rows, ref_rows=ref16[rows], truth_rows=truth[rows], col_stride=COL_STRIDE, ref_cols=ref16[:, ::COL_STRIDE],
               truth_cols=truth[:, ::COL_STRIDE], truth_absmax=float(t.abs().max()),
               full_stats=dict(ref_err_vs_fp32=rel(r, t), ref_rms_vs_fp32=float((r - t).pow(2).mean().sqrt()),
                               argmax_agree_ref_fp32=float((r.argmax(-1) == t.argmax(-1)).float().mean())))
    gap = top_t.values[:, 0] - top_t.values[:, 1]
    gated = gap > 4.0 * sigma
    rec["positions_gated_k4"] = int(gated.sum())
    assert bool((r.argmax(-1) == t.argmax(-1))[gated].all()), "the reference's own 16-bit ids differ from its fp32 ids at a gated position"
    log(f"   logits: ref-16-bit err vs fp32 {rec['full_stats']['ref_err_vs_fp32']:.5f}, gated {int(gated.sum())}/{S}, "
        f"argmax agree {rec['full_stats']['argmax_agree_ref_fp32']:.3f}")
    return rec


def run_core(m, sd, ids, mask, img, what):
    t0 = time.time()
    with torch.no_grad():
        r = m(input_ids=ids, attention_mask=mask, images=img, output_hidden_states=True)
    t1 = time.time()
    o = O.core_forward(sd, llm_cfg_dict(), ids, mask, img)
    log(f"   {what}: reference {t1 - t0:.1f} s, oracle {time.time() - t1:.1f} s")
    G.eq(o["logits"], r.logits, f"{what} logits")
    assert len(o["hidden_states"]) == len(r.hidden_states) == 33
    for i, (x, y) in enumerate(zip(o["hidden_states"], r.hidden_states)):
        G.eq(x, y, f"{what} hidden_states[{i}]")
    return r.logits, [h for h in r.hidden_states]


def gen_g15(tag, dtype):
    name = f"g15_c1_full_depth_{tag}.pt"
    log(f"[{name}]")
    m, shapes = build_reference(False, dtype)
    ids, mask, img = c1_inputs(dtype)
    sd = m.state_dict()
    logits16, hid16 = run_core(m, sd, ids, mask, img, f"C1 {tag}")
    dig = dict(logits=digest(logits16), **{f"hidden_{i}": digest(hid16[i]) for i in HID_LAYERS})
    m.float()                                                   # fp32 arithmetic on the 16-bit-rounded weights = the truth
    gc.collect()
    sd = m.state_dict()
    logits32, hid32 = run_core(m, sd, ids, mask, img.float(), "C1 fp32 truth")
    fx = dict(cfg=llm_cfg_dict(), seed=SEED, hf_init=True, dtype=str(dtype), shapes=shapes, input_ids=ids, attention_mask=mask, images=img,
              digests=dig, logits=logits_record(logits16[0], logits32[0]), hid_layers=list(HID_LAYERS), hid_stride=HID_STRIDE,
              ref_hidden={i: hid16[i][0, :, ::HID_STRIDE] for i in HID_LAYERS}, truth_hidden={i: hid32[i][0, :, ::HID_STRIDE] for i in HID_LAYERS},
              truth_hidden_absmax={i: float(hid32[i].abs().max()) for i in HID_LAYERS},
              ref_hidden_err_full={i: rel(hid16[i], hid32[i]) for i in HID_LAYERS})
    G.META_EXTRA["what"] = ("reference UllavaCoreForCausalLM.forward at FULL size (ViT-L/14-224 + LLaMA-7B) on the build container's CPU; "
                            "reference == oracle torch.equal on logits and all 33 hidden states, in this dtype and in fp32")
    G.save(name, fx)
    del m, sd
    gc.collect()


def gen_g16(tag, dtype):
    name = f"g16_res_full_depth_{tag}.pt"
    log(f"[{name}]")
    m, shapes = build_reference(True, dtype)
    if dtype == torch.float16:
        G._emulate_fp16_neck_autocast(m.visual_model.image_encoder)
    ids, mask, img, images_sam, sizes, resizes = c3_inputs(dtype)
    ocfg = dict(llm=llm_cfg_dict(), sam=SAM_H, seg_token_idx=SEG, loc_token_idx=LOC)

    def run(mod, sd, a_sam, a_img, what):
        t0 = time.time()
        with torch.no_grad():
            r = mod(images_sam=a_sam, images=a_img, input_ids=ids, labels=None, attention_mask=mask, mask_list=[None], size_list=sizes,
                    resize_list=resizes, bbox_list=[None], inference=True)
            emb = mod.get_visual_embs(a_sam)
        t1 = time.time()
        o = O.ullava_forward(sd, ocfg, a_sam, a_img, ids, mask, sizes, resizes)
        log(f"   {what}: reference {t1 - t0:.1f} s, oracle {time.time() - t1:.1f} s")
        G.eq(o["logits"], r["logits"], f"{what} logits")
        G.eq(o["pred_masks"][0], r["pred_masks"][0], f"{what} pred_masks")
        G.eq(o["pred_boxes"][0], r["pred_boxes"][0], f"{what} pred_boxes")
        G.eq(o["image_embeddings"], emb, f"{what} SAM image embedding")
        assert sorted(r.keys()) == ["gt_boxes", "gt_masks", "logits", "pred_boxes", "pred_masks"]
        return r["logits"], r["pred_masks"][0], r["pred_boxes"][0], emb, o["low_res_masks"]
    # the state dict BEFORE the fp16 neck wrapper re-keys it: the oracle wants the reference's key names
    sd = {k.replace("neck.neck32.", "neck."): v for k, v in m.state_dict().items()} if dtype == torch.float16 else m.state_dict()
    if dtype == torch.float16:                                  # (the wrapper holds an fp32 COPY of the fp16 neck: hand the oracle the fp16 values)
        sd = {k: (v.to(dtype) if "image_encoder.neck." in k else v) for k, v in sd.items()}
    lg16, pm16, pb16, emb16, low16 = run(m, sd, images_sam, img, f"C3 {tag}")
    dig = dict(logits=digest(lg16), pred_masks=digest(pm16), pred_boxes=digest(pb16), image_embeddings=digest(emb16))
    if dtype == torch.float16:
        m.visual_model.image_encoder.neck = m.visual_model.image_encoder.neck.neck32      # already fp32-valued; .float() below is a no-op on it
    m.float()
    gc.collect()
    sd = m.state_dict()
    lg32, pm32, pb32, emb32, low32 = run(m, sd, images_sam.float(), img.float(), "C3 fp32 truth")
    assert pm16.dtype == torch.float32 and tuple(pm16.shape) == (3, 480, 640)
    e_emb, e_pm, e_pb = rel(emb16, emb32), rel(pm16, pm32), rel(pb16, pb32)
    clear = pm32.abs() > 4.0 * float((pm16 - pm32).abs().max())
    assert bool(((pm16 > 0) == (pm32 > 0))[clear].all())
    log(f"   ref-16-bit err vs fp32: SAM embedding {e_emb:.5f}, masks {e_pm:.5f}, boxes {e_pb:.5f}; mask pixels decided with margin "
        f"{float(clear.float().mean()) * 100:.1f} %")
    fx = dict(cfg=ocfg, seed=SEED, hf_init=True, dtype=str(dtype), shapes=shapes, input_ids=ids, attention_mask=mask, images=img,
              inputs_seed=SEED + 200, images_sam_digest=digest(images_sam), size_list=sizes, resize_list=resizes, digests=dig,
              logits=logits_record(lg16[0], lg32[0]),
              emb_sample_index="[:, ::8, ::2, ::2]", ref_emb=emb16[:, ::8, ::2, ::2], truth_emb=emb32[:, ::8, ::2, ::2],
              truth_emb_absmax=float(emb32.abs().max()),
              mask_sample_index="[:, ::4, ::4]", ref_masks=pm16[:, ::4, ::4], truth_masks=pm32[:, ::4, ::4], truth_masks_absmax=float(pm32.abs().max()),
              ref_low_res_masks=low16[0] if isinstance(low16, (list, tuple)) else low16,
              truth_low_res_masks=low32[0] if isinstance(low32, (list, tuple)) else low32,
              ref_boxes=pb16, truth_boxes=pb32,
              ref_err_full=dict(sam_image_embedding=e_emb, pred_masks=e_pm, pred_boxes=e_pb, mask_margin_share=float(clear.float().mean())),
              mask_sign_margin=4.0 * float((pm16 - pm32).abs().max()))
    G.META_EXTRA["what"] = ("reference UllavaForCausalLM.forward(inference=True) at FULL size (LLaMA-7B + ViT-L/14-224 + SAM ViT-H, 3 prompts) on "
                            "the build container's CPU; reference == oracle torch.equal on logits, masks, boxes, SAM embedding, in this dtype and fp32")
    G.save(name, fx)
    del m, sd
    gc.collect()


if __name__ == "__main__":
    which = set(sys.argv[1:])
    jobs = [("g15_bf16", gen_g15, "bf16", torch.bfloat16), ("g15_fp16", gen_g15, "fp16", torch.float16),
            ("g16_bf16", gen_g16, "bf16", torch.bfloat16), ("g16_fp16", gen_g16, "fp16", torch.float16)]
    for key, fn, tag, dt in jobs:
        if not which or key in which:
            fn(tag, dt)
            print(f"{key}: reference == oracle bit-exact at full depth ({tag} and fp32)", flush=True)
