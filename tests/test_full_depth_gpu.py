"""GPU: the EXACT workloads bench.py times (C4: ViT-L/14-336 + 32-layer LLaMA-7B, B = 32, S = 643; C3 / RES: + SAM ViT-H with 32 blocks
on a second HIP stream, B = 8), checked through size-independent properties -- the CPU oracle cannot run these sizes in test time:

  * finite outputs, and two runs of the same step give identical bits (no race between the two streams, no atomics in the forward);
  * prefix property: the hidden states after LLaMA layers 0 and 1 of the 32-layer model are bit-identical to those of a 3-layer model
    holding the same weights at the same batch -- and at single-sample size such shallow full-width models are what
    tests/test_model_gpu.py::test_full_width_forward_against_oracle compares with the oracle;
  * batch independence: sample b inside the batch vs the same sample alone.  The two sizes take different tile kernels / stream-K
    splits; measured on MI355X they nevertheless agree bit for bit at every depth (both accumulate K in the same order), which the
    asserts pin (exact for embeddings / layer 0, the 1.5-ulp rule of test_model_gpu.py for logits; the record is printed on every run).

Also here: the never-before-executed RCCL surfaces at world size 1 (process-group init, barrier, the two scalar all-reduces of the bench
aggregation, the direct-exchange gradient reduction), run in a subprocess under a timeout.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import load_fixture, pkg

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stats(a, b):
    a, b = a.float(), b.float()
    d = (a - b).abs()
    return dict(max=float(d.max()), mean=float(d.mean()), ref_max=float(b.abs().max()), ref_std=float(b.std()),
                frac_equal=float((a == b).float().mean()))


def test_c4_bench_workload_full_depth():
    import bench
    M, C = pkg("modeling_core"), pkg("configuration")
    with torch.no_grad():
        step, batch, S, cfg, desc, flops, model = bench.workload_step("c4", DEV, 0)
        assert (batch, S, cfg.num_hidden_layers, cfg.vision_config.image_size) == (32, 643, 32, 336)
        a = step().logits
        b = step().logits
        assert tuple(a.shape) == (32, 643, 32011) and bool(torch.isfinite(a.float()).all())
        assert torch.equal(a, b), "two runs of the C4 step differ"
        bench.check_finite({"logits": a})
        vis, ids, mask = bench.make_inputs(cfg, batch, 64, DEV, 0)          # the very inputs workload_step built (same seed)
        full = model.forward(input_ids=ids, attention_mask=mask, images=vis, output_hidden_states=True)
        assert torch.equal(full.logits, a)
        # --- prefix property against a 3-layer model with the same weights ------------------------------------------------------
        cfg3 = C.UllavaCoreConfig(vision_config=dict(image_size=336, patch_size=14), vision_hidden_layer=-2, projector_type="mlp",
                                  mm_token_ids=dict(bench.MM), vocab_size=32011, num_hidden_layers=3)
        m3 = M.UllavaCoreForCausalLM(cfg3, device=DEV)
        sd = model.state_dict()
        m3.load_state_dict({k: v for k, v in sd.items() if not k.startswith("model.layers.") or int(k.split(".")[2]) < 3}, strict=True)
        m3.strict_checks = False
        o3 = m3.forward(input_ids=ids, attention_mask=mask, images=vis, output_hidden_states=True)
        for li in (0, 1, 2):
            assert torch.equal(o3.hidden_states[li], full.hidden_states[li]), f"hidden state {li} of the 32-layer model != 3-layer model"
        # --- batch independence at full depth -----------------------------------------------------------------------------------
        bsel = 5
        one = model.forward(input_ids=ids[bsel:bsel + 1], attention_mask=mask[bsel:bsel + 1], images=vis[bsel:bsel + 1], output_hidden_states=True)
        one3 = m3.forward(input_ids=ids[bsel:bsel + 1], attention_mask=mask[bsel:bsel + 1], images=vis[bsel:bsel + 1])
    s_emb = _stats(full.hidden_states[0][bsel], one.hidden_states[0][0])
    s_l1 = _stats(full.hidden_states[1][bsel], one.hidden_states[1][0])
    s3 = _stats(o3.logits[bsel], one3.logits[0])
    s32 = _stats(full.logits[bsel], one.logits[0])
    agree = float((full.logits[bsel].argmax(-1) == one.logits[0].argmax(-1)).float().mean())
    print("C4 full depth, sample 5 in B=32 vs alone:", json.dumps(dict(embeds=s_emb, layer0=s_l1, logits_3_layers=s3, logits_32_layers=s32, argmax_agree=agree)))
    # measured on MI355X: every one of these is bit-identical (frac_equal 1.0) -- the 128x128 kernel of the single-sample run and the
    # 256x256 kernel (+ stream-K tail) of the batch accumulate K in the same order.  Embeddings and layer 0 are asserted exact; the
    # logits get the 1.5-ulp rule of test_model_gpu.py so that a legitimate change of the K-split policy does not read as a failure
    assert s_emb["max"] == 0.0 and s_l1["max"] == 0.0
    for s_ in (s3, s32):
        assert s_["max"] <= 2.0 ** -6 * s_["ref_max"] and s_["mean"] <= 2.0 ** -8 * s_["ref_std"]
    assert agree >= 0.99


@pytest.mark.parametrize("name", ["c2", "c5"])
def test_c2_c5_bench_workloads_full_depth(name):
    """BASELINE.json configs[1] (C2: VQA, batch 16, prompts of 32..64 tokens right-padded to S = 323) and configs[4] (C5: 8 clips of 8 frames per
    GPU, S = 299) exactly as `bench.py` times them (`extra.c2` / `extra.c5` of the bench line): 23 CLIP + 32 LLaMA-7B layers at the bench batch.
    The oracle cannot run these sizes in test time, so -- as for C4 above -- size-independent properties: finite and deterministic; the hidden
    states after LLaMA layers 0 / 1 equal those of a 3-layer model with the same weights (which test_model_gpu.py compares with the oracle at
    these very batch shapes); a sample inside the batch equals the same sample alone; and for the ragged C2 batch a row truncated to its own
    length gives the valid positions' logits (right padding is invisible under the causal mask; reference collator base_collator.py:27-42)."""
    import bench
    M, C = pkg("modeling_core"), pkg("configuration")
    video = name == "c5"
    kw = "videos" if video else "images"
    with torch.no_grad():
        step, batch, S, cfg, desc, flops, model = bench.workload_step(name, DEV, 0)
        assert (batch, S, cfg.num_hidden_layers, cfg.vision_config.image_size) == ((16, 323, 32, 224) if name == "c2" else (8, 299, 32, 224))
        a = step().logits
        b = step().logits
        assert tuple(a.shape) == (batch, S, 32011) and bool(torch.isfinite(a.float()).all())
        assert torch.equal(a, b), f"two runs of the {name} step differ"
        vis, ids, mask = bench.make_inputs(cfg, batch, bench.WORKLOADS[name][1], DEV, 0, ragged=(name == "c2"), video=video)
        if name == "c2":
            lens = mask.sum(1).tolist()
            assert min(lens) == 259 + 32 and max(lens) == 259 + 64 and len(set(lens)) > 8            # really ragged
        else:
            assert tuple(vis.shape) == (8, 3, 8, 224, 224)
        full = model.forward(input_ids=ids, attention_mask=mask, output_hidden_states=True, **{kw: vis})
        assert torch.equal(full.logits, a)
        cfg3 = C.UllavaCoreConfig(vision_config=dict(image_size=224, patch_size=14), vision_hidden_layer=-2, projector_type="mlp",
                                  mm_token_ids=dict(bench.MM), vocab_size=32011, num_hidden_layers=3)
        m3 = M.UllavaCoreForCausalLM(cfg3, device=DEV)
        sd = model.state_dict()
        m3.load_state_dict({k: v for k, v in sd.items() if not k.startswith("model.layers.") or int(k.split(".")[2]) < 3}, strict=True)
        m3.strict_checks = False
        o3 = m3.forward(input_ids=ids, attention_mask=mask, output_hidden_states=True, **{kw: vis})
        for li in (0, 1, 2):
            assert torch.equal(o3.hidden_states[li], full.hidden_states[li]), f"hidden state {li} of the 32-layer model != 3-layer model"
        bsel = 3                                                                                   # (C2: a row with padding)
        one = model.forward(input_ids=ids[bsel:bsel + 1], attention_mask=mask[bsel:bsel + 1], output_hidden_states=True, **{kw: vis[bsel:bsel + 1]})
        s_emb = _stats(full.hidden_states[0][bsel], one.hidden_states[0][0])
        s_l1 = _stats(full.hidden_states[1][bsel], one.hidden_states[1][0])
        valid = mask[bsel].bool()
        s32 = _stats(full.logits[bsel][valid], one.logits[0][valid])
        agree = float((full.logits[bsel][valid].argmax(-1) == one.logits[0][valid].argmax(-1)).float().mean())
        rec = dict(embeds=s_emb, layer0=s_l1, logits_32_layers=s32, argmax_agree=agree)
        if name == "c2":
            n = int(mask[bsel].sum())
            assert n < S
            cut = model.forward(input_ids=ids[bsel:bsel + 1, :n], attention_mask=mask[bsel:bsel + 1, :n], images=vis[bsel:bsel + 1])
            rec["truncated_row"] = st = _stats(full.logits[bsel, :n], cut.logits[0])
            assert st["max"] <= 2.0 ** -6 * st["ref_max"] and st["mean"] <= 2.0 ** -8 * st["ref_std"]
    print(f"{name} full depth, sample {bsel} in B={batch} vs alone:", json.dumps(rec))
    assert s_emb["max"] == 0.0 and s_l1["max"] == 0.0
    assert s32["max"] <= 2.0 ** -6 * s32["ref_max"] and s32["mean"] <= 2.0 ** -8 * s32["ref_std"]
    assert agree >= 0.99


@pytest.mark.parametrize("case", ["c4_336", "c5_video"])
def test_c4_c5_shapes_full_depth_against_oracle(case):
    """The other two core-path configs of BASELINE.json at full depth and batch 1 against the oracle (bf16 + fp32 truth), same rule as
    test_c1_full_depth_against_oracle: C4's shape (336 x 336 image -> 576 visual tokens, 64-token prompt, S = 643; a 577-row position table) and
    C5's (an 8-frame 224 x 224 clip -> per-frame ViT-L, 8 + 256 pooled tokens, S = 299)."""
    import bench
    from oracle import ullava_oracle as O
    sd, model = _c1_fixture()
    llm_cfg = bench.c1_case()[0]
    g = torch.Generator().manual_seed(17)
    if case == "c4_336":
        sd = dict(sd)
        sd["vision_encoder.embeddings.position_embedding.weight"] = (torch.randn(577, 1024, generator=g) * 0.02).to(torch.bfloat16)
        llm_cfg = dict(llm_cfg, vision_config=dict(llm_cfg["vision_config"], image_size=336))
        C = pkg("configuration")
        M = pkg("modeling_core")
        with torch.no_grad():
            model = M.UllavaCoreForCausalLM(C.UllavaCoreConfig(vision_config=dict(image_size=336, patch_size=14), vision_hidden_layer=-2, projector_type="mlp",
                                                               mm_token_ids=dict(bench.MM), vocab_size=32011), device=DEV)
            model.load_state_dict(sd, strict=True)
        model.strict_checks = False
        vis = torch.randn(1, 3, 336, 336, generator=g).to(torch.bfloat16)
        head = [1, bench.MM["IMG_START"]] + [bench.MM["IMG_PATCH"]] * 576 + [bench.MM["IMG_END"]]
        ids = torch.tensor([head + torch.randint(5, 32000, (64,), generator=g).tolist()])
        kw_h, kw_o = dict(images=vis.to(DEV)), dict(images=vis)
        kw_t = dict(images=vis.float())
    else:
        vis = torch.randn(1, 3, 8, 224, 224, generator=g).to(torch.bfloat16)
        head = [1, bench.MM["VID_START"]] + [bench.MM["VID_PATCH"]] * (8 + 256) + [bench.MM["VID_END"]]
        ids = torch.tensor([head + torch.randint(5, 32000, (32,), generator=g).tolist()])
        kw_h, kw_o, kw_t = dict(videos=vis.to(DEV)), dict(videos=vis), dict(videos=vis.float())
    mask = torch.ones_like(ids)
    torch.set_num_threads(min(os.cpu_count(), 64))
    with torch.no_grad():
        out = model.forward(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), output_hidden_states=True, **kw_h)
        ref = O.core_forward(sd, llm_cfg, ids, mask, **kw_o)
        truth = O.core_forward(bench.F32View(sd), llm_cfg, ids, mask, **kw_t)
    assert tuple(out.logits.shape) == (1, ids.shape[1], 32011)
    e0_ref, e0_hip = _rel(ref["hidden_states"][0], truth["hidden_states"][0]), _rel(out.hidden_states[0], truth["hidden_states"][0])
    st = bench.parity_stats(out.logits[0], ref["logits"][0], truth["logits"][0])
    print(f"{case} full depth vs oracle:", json.dumps(dict(S=int(ids.shape[1]), spliced_embeds=dict(oracle_bf16_err=round(e0_ref, 5), hip_err=round(e0_hip, 5)), logits=st)))
    assert e0_hip <= max(1.5 * e0_ref, 2.0 ** -7)                # the CLIP tower + projector + splice (hidden state 0)
    assert st["hip_err_vs_fp32"] <= max(1.5 * st["oracle_err_vs_fp32"], 2.0 ** -6)
    assert st["gated_exact"] and st["positions_gated"] >= st["positions"] // 10, st
    assert st["hip_rms_vs_fp32"] <= 1.25 * st["oracle_rms_vs_fp32"], st
    if case == "c4_336":
        del model
        torch.cuda.empty_cache()


def test_res_bench_workload_full_depth():
    import bench
    with torch.no_grad():
        step, batch, S, cfg, desc, flops, model = bench.workload_step("res", DEV, 0)
        assert batch == 8 and len(model.visual_model.image_encoder.blocks) == 32 and model.overlap_sam_encoder
        o1 = step()
        o2 = step()
        bench.check_finite(o1)
        assert len(o1["pred_masks"]) == 8 and all(tuple(m.shape) == (3, 480, 640) and m.dtype == torch.float32 for m in o1["pred_masks"])
        assert all(tuple(b.shape) == (3, 4) for b in o1["pred_boxes"])
        assert torch.equal(o1["logits"], o2["logits"])
        for m1, m2 in zip(o1["pred_masks"], o2["pred_masks"]):
            assert torch.equal(m1, m2), "two runs of the RES step differ (two-stream race?)"
        for b1, b2 in zip(o1["pred_boxes"], o2["pred_boxes"]):
            assert torch.equal(b1, b2)
        # the SAM encoder on its own stream vs on the main stream: same kernels; only the stream-K policy differs (K >= 8192 vs 2048)
        emb_side = model._visual_embs_tm
        g = torch.Generator(device="cuda").manual_seed(2000)
        images_sam = torch.randn(8, 3, 1024, 1024, device=DEV, generator=g).to(torch.bfloat16)
        e_all = emb_side(images_sam)
        e_one = emb_side(images_sam[3:4])
        assert bool(torch.isfinite(e_all.float()).all())
        s = _stats(e_all[3], e_one[0])
        print("SAM ViT-H 32 blocks, image 3 in B=8 vs alone:", json.dumps(s))
        # batch 8 and batch 1 take the same 256x256 tiles (4096 rows per image) except for stream-K tails; measured: bit-identical
        assert s["max"] <= 2.0 ** -7 * s["ref_max"] and s["frac_equal"] >= 0.999
        model.overlap_sam_encoder = False
        o3 = step()
        model.overlap_sam_encoder = True
        sm = _stats(torch.stack(o3["pred_masks"]), torch.stack(o1["pred_masks"]))
        print("RES masks, SAM encoder on the main stream vs on the side stream:", json.dumps(sm))
        assert sm["max"] <= 2.0 ** -7 * sm["ref_max"]              # measured: bit-identical


C1_GREEDY_SEED = 2362       # prompt seed of the greedy comparison: the best of 3000 searched by tools/c1_greedy_seed_search.py on the GPU box
C1_GREEDY_K = 3.5           # ... its smallest step margin measured 4.14 noise standard deviations (profiles/r04_c1_seed_search.json); asserted >= 3.5


def _c1_fixture():
    """(state dict on the host, HIP model on the GPU) of the full-depth C1 model, shared by the two tests below (13.9 GB each side)."""
    import bench
    if not hasattr(_c1_fixture, "v"):
        sd = bench._c1_state_dict(DEV)
        with torch.no_grad():
            _c1_fixture.v = (sd, bench.c1_hip_model(sd, DEV))
    return _c1_fixture.v


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_c1_full_depth_against_oracle(dt):
    """BASELINE.json configs[0] at FULL depth -- 1 x 224x224 image + 32-token prompt, S = 291, 23 CLIP + 32 LLaMA-7B layers, V = 32011, the
    very weights on both sides -- HIP forward vs the CPU oracle (same 16-bit dtype) vs the oracle in fp32, in bf16 (the reference's training
    dtype) and fp16 (its `--dtype fp16` option, inference_ullava.py:26,164-168; the default there is bf16):
      * hidden states 0 / 8 / 16 / 24 / 32 and the logits are as close to the fp32 truth as the oracle's own 16-bit run is (x1.5), the rule of
        tests/test_model_gpu.py, now through all 32 layers;
      * margin-gated exact token ids (bench.parity_stats): wherever the fp32 top-1 / top-2 gap exceeds 4 standard deviations of that
        position's 16-bit noise on a logit difference, argmax(HIP) == argmax(oracle 16-bit) == argmax(fp32) -- and a meaningful share of the
        291 positions passes the gate (a random-init 7 B model has near-flat logits: median gap 0.21 against a bf16 noise sigma of ~0.09)."""
    import bench
    from oracle import ullava_oracle as O
    sd, model = _c1_fixture()
    if dt != torch.bfloat16:                                       # the fp16 twin: the same generated weights rounded to fp16 on both sides
        sd = {k: v.to(dt) for k, v in sd.items()}
        with torch.no_grad():
            model = bench.c1_hip_model(sd, DEV, dtype=dt)
    cfg, ids, mask, img = bench.c1_case()
    img = img.to(dt)
    torch.set_num_threads(min(os.cpu_count(), 64))
    with torch.no_grad():
        out = model.forward(input_ids=ids.to(DEV), attention_mask=mask.to(DEV), images=img.to(DEV), output_hidden_states=True)
        ref = O.core_forward(sd, cfg, ids, mask, img)
        truth = O.core_forward(bench.F32View(sd), cfg, ids, mask, img.float())
    assert tuple(out.logits.shape) == (1, 291, 32011) and len(out.hidden_states) == 33 and out.logits.dtype == dt
    rec = {}
    for li in (0, 8, 16, 24, 32):
        e_ref, e_hip = _rel(ref["hidden_states"][li], truth["hidden_states"][li]), _rel(out.hidden_states[li], truth["hidden_states"][li])
        rec[f"hidden_{li}"] = dict(oracle_16bit_err=round(e_ref, 5), hip_err=round(e_hip, 5), hip_vs_oracle=round(_rel(out.hidden_states[li], ref["hidden_states"][li]), 5))
        assert e_hip <= max(1.5 * e_ref, 2.0 ** -7), (li, e_hip, e_ref)
    st = bench.parity_stats(out.logits[0], ref["logits"][0], truth["logits"][0])
    rec["logits"] = st
    print(f"C1 full depth vs oracle ({dt}):", json.dumps(rec))
    assert st["hip_err_vs_fp32"] <= max(1.5 * st["oracle_err_vs_fp32"], 2.0 ** -6)
    assert st["gated_exact"], st                                   # token ids EQUAL wherever the margin clears the 16-bit noise
    assert st["positions_gated"] >= 29, st                         # ... which is not a vacuous set (>= 10 % of the positions)
    # (all positions, gated or not: the HIP run agrees with the fp32 truth at least as often as the oracle's own 16-bit run does, minus 3 %)
    assert st["argmax_agree_hip_fp32"] >= st["argmax_agree_oracle_fp32"] - 0.03, st
    assert st["hip_rms_vs_fp32"] <= 1.25 * st["oracle_rms_vs_fp32"], st
    if dt != torch.bfloat16:
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_c1_greedy_ids_match_oracle_where_gated(dt):
    """8-token greedy generate() on the full-depth C1 model vs the oracle's greedy loop (no KV cache: the reference checkpoints' configuration,
    and with the KV cache): ids torch.equal.  The prompt (seed C1_GREEDY_SEED, searched offline on the GPU box: tools/c1_greedy_seed_search.py,
    profiles/r04_c1_seed_search.json) is one where EVERY step clears the noise gate in the oracle's own numbers -- fp32 top-1 / top-2 gap
    > C1_GREEDY_K standard deviations of the bf16 noise on a logit difference at that step -- so equality is a requirement, not luck; the
    test asserts the gate too.  (A random-init 7 B model has near-flat logits -- median top-2 gap 0.21 against a noise sigma of ~0.09 --
    which is why the prompt has to be searched for.)"""
    import bench
    from oracle import ullava_oracle as O
    sd, model = _c1_fixture()
    if dt != torch.bfloat16:                                       # fp16 (the reference's --dtype fp16 option): same weights rounded to fp16
        sd = {k: v.to(dt) for k, v in sd.items()}
        with torch.no_grad():
            model = bench.c1_hip_model(sd, DEV, dtype=dt)
    cfg, ids, mask, img = bench.c1_case(seed=C1_GREEDY_SEED)
    img = img.to(dt)
    L0 = ids.shape[1]
    torch.set_num_threads(min(os.cpu_count(), 64))
    with torch.no_grad():
        got_nc = model.generate(input_ids=ids.to(DEV), images=img.to(DEV), max_new_tokens=8, do_sample=False, use_cache=False, eos_token_id=-1)
        got_kv = model.generate(input_ids=ids.to(DEV), images=img.to(DEV), max_new_tokens=8, do_sample=False, use_cache=True, eos_token_id=-1)
        if dt == torch.bfloat16:
            want, _ = O.greedy_generate(sd, cfg, ids, images=img, max_new_tokens=8)       # the oracle's own greedy loop: 8 forwards without cache
        else:
            # fp16 twin (the CPU's fp16 forward is 3x slower): ONE oracle forward teacher-forced on the HIP ids -- the oracle's argmax after every
            # prefix equals the next HIP id (asserted below through `want`), which by induction is what its greedy loop produces
            want = got_nc.cpu()
        seq = want[:, :-1]
        o = O.core_forward(sd, cfg, seq, torch.ones_like(seq), img)["logits"][0, L0 - 1:].float()
        t = O.core_forward(bench.F32View(sd), cfg, seq, torch.ones_like(seq), img.float())["logits"][0, L0 - 1:]
        hip = model.forward(input_ids=seq.to(DEV), images=img.to(DEV)).logits[0, L0 - 1:].float().cpu()      # teacher-forced on the oracle's ids
    sigma = (o - t).pow(2).mean(-1).sqrt() * 2.0 ** 0.5
    t2 = t.topk(2, dim=-1).values
    gaps = t2[:, 0] - t2[:, 1]
    print(f"C1 greedy ({dt}):", json.dumps(dict(seed=C1_GREEDY_SEED, oracle=want[0, L0:].tolist(), hip_no_cache=got_nc[0, L0:].tolist(),
                                         hip_kv_cache=got_kv[0, L0:].tolist(), fp32_gaps=[round(float(x), 4) for x in gaps],
                                         diff_sigma=[round(float(x), 4) for x in sigma], min_gap_over_sigma=round(float((gaps / sigma).min()), 3))))
    assert bool((gaps > C1_GREEDY_K * sigma).all()), "the committed prompt no longer clears the noise gate at every step: re-run tools/c1_greedy_seed_search.py"
    assert torch.equal(o.argmax(-1), want[0, L0:])                 # the oracle's 16-bit argmax after every prefix IS the sequence (its greedy loop)
    assert torch.equal(t.argmax(-1), want[0, L0:])                 # ... and the fp32 ids at every (gated) step
    assert torch.equal(hip.argmax(-1), want[0, L0:]), "teacher-forced HIP argmax differs from the oracle's ids at a gated step"
    assert torch.equal(got_nc.cpu(), want), "generate() without a KV cache: token ids differ from the oracle's greedy loop"
    assert torch.equal(got_kv.cpu(), want), "generate() with the KV cache: token ids differ from the oracle's greedy loop"


def test_c1_kv_cached_generate_returns_the_no_cache_last_step_states_full_depth():
    """`evaluate()` reads `generate(...).hidden_states[-1][-1]` -- under the reference checkpoints' use_cache=False that is the
    last-layer (post-norm) state of EVERY position of `sequences[:, :-1]`, i.e. `forward(sequences[:, :-1]).hidden_states[-1]`
    (models/ullava.py:350-371).  The HIP generate() keeps a KV cache and must hand back the same tensor: on the 32-layer 7 B C1 model,
    8 greedy steps, `keep_last_step_only=True` (what evaluate() passes):
      * shape [1, L - 1, 4096]; the prefill rows are BIT-identical to a HIP forward over sequences[:, :-1];
      * the decode rows (GEMV kernels, a different fp32 summation order) and the whole tensor are as close to the oracle's fp32 forward over
        the same ids as the oracle's own bf16 forward is (x1.5 rule);
      * the no-cache generate() returns the forward's tensor bit for bit."""
    import bench
    from oracle import ullava_oracle as O
    sd, model = _c1_fixture()
    cfg, ids, mask, img = bench.c1_case(seed=C1_GREEDY_SEED)
    L0 = ids.shape[1]
    torch.set_num_threads(min(os.cpu_count(), 64))
    with torch.no_grad():
        kv = model.generate(input_ids=ids.to(DEV), images=img.to(DEV), max_new_tokens=8, do_sample=False, use_cache=True, eos_token_id=-1,
                            output_hidden_states=True, return_dict_in_generate=True, keep_last_step_only=True)
        nc = model.generate(input_ids=ids.to(DEV), images=img.to(DEV), max_new_tokens=8, do_sample=False, use_cache=False, eos_token_id=-1,
                            output_hidden_states=True, return_dict_in_generate=True, keep_last_step_only=True)
        assert torch.equal(kv.sequences, nc.sequences) and kv.sequences.shape[1] == L0 + 8
        seq = kv.sequences[:, :-1]
        fwd = model.forward(input_ids=seq, images=img.to(DEV), output_hidden_states=True).hidden_states[-1]
        h_kv, h_nc = kv.hidden_states[-1][-1], nc.hidden_states[-1][-1]
        assert tuple(h_kv.shape) == tuple(h_nc.shape) == tuple(fwd.shape) == (1, L0 + 7, 4096)
        assert torch.equal(h_nc, fwd), "no-cache generate(): last-step states != forward(sequences[:, :-1])"
        assert torch.equal(h_kv[:, :L0], fwd[:, :L0]), "KV-cached generate(): prefill rows differ from forward(sequences[:, :-1])"
        sc = seq.cpu()
        ref = O.core_forward(sd, cfg, sc, torch.ones_like(sc), img)["hidden_states"][-1]
        truth = O.core_forward(bench.F32View(sd), cfg, sc, torch.ones_like(sc), img.float())["hidden_states"][-1]
    e_ref, e_kv, e_fwd = _rel(ref, truth), _rel(h_kv, truth), _rel(fwd, truth)
    e_dec_ref, e_dec_kv = _rel(ref[:, L0:], truth[:, L0:]), _rel(h_kv[:, L0:], truth[:, L0:])
    print("C1 evaluate() states, KV cache vs forward vs oracle:", json.dumps(dict(
        oracle_bf16_err=round(e_ref, 5), hip_kv_err=round(e_kv, 5), hip_forward_err=round(e_fwd, 5), decode_rows_oracle_err=round(e_dec_ref, 5),
        decode_rows_hip_kv_err=round(e_dec_kv, 5), kv_vs_forward_decode_rows=round(_rel(h_kv[:, L0:], fwd[:, L0:]), 5))))
    assert e_kv <= max(1.5 * e_ref, 2.0 ** -7) and e_dec_kv <= max(1.5 * e_dec_ref, 2.0 ** -7)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_res_full_depth_against_oracle(dt):
    """BASELINE.json configs[2] at FULL depth and batch 1 (bf16, and fp16 = the reference's --dtype fp16 option): ViT-L/14-224 + 32 LLaMA-7B layers + SAM ViT-H (32 blocks, d = 1280, 1024 x 1024) +
    prompt encoder + two-way MaskDecoder + postprocess, three [SEG] / [LOC] rounds, the very weights on both sides (random init made on the
    GPU, copied to the host for the oracle): `UllavaForCausalLM.forward(inference=True)` vs `O.ullava_forward` in bf16 vs the oracle in fp32.
    SAM image embedding, LLaMA logits, mask logits (fp32 [3, 480, 640]) and boxes must be as close to the fp32 truth as the oracle's own
    bf16 run is (x1.5, with a floor of a few bf16 ulps) -- the C3 path against the oracle at the size bench.py times it, not only through
    properties."""
    import bench
    from oracle import ullava_oracle as O
    if dt == torch.float16:
        # fp16 (the reference's --dtype fp16 option): the CPU oracle's fp16 SAM ViT-H + LLaMA-7B forward takes ~5 min on the GPU box's host, so
        # this case reads the committed G16 fixture instead -- the REFERENCE ITSELF run once at this size on the build container (fp16 with the
        # fp32 neck, and fp32), reference == oracle asserted bit for bit there (tests/golden/gen_golden_full_depth.py).  Round 6: no skip.
        return _res_against_reference_fixture("fp16", dt)
    with torch.no_grad():
        model, cfg = bench.build_model(224, DEV, seed=3, with_sam=True)
        if dt != torch.bfloat16:
            model = model.to(dt)                                  # .half(): the packed copies are re-made from the cast parameters
            model.llm.strict_checks = False
        vis, ids, mask = bench.make_inputs(cfg, 1, 120, DEV, 3)
        vis = vis.to(dt)
        S = ids.shape[1]
        for r in range(3):
            ids[:, S - 10 - 40 * r] = bench.SEG
            ids[:, S - 5 - 40 * r] = bench.LOC
        g = torch.Generator(device="cuda").manual_seed(2003)
        images_sam = torch.randn(1, 3, 1024, 1024, device=DEV, generator=g).to(dt)
        sizes, resizes = [(480, 640)], [(768, 1024)]
        out = model.forward(images_sam=images_sam, images=vis, input_ids=ids, labels=None, attention_mask=mask, mask_list=[None], size_list=sizes,
                            resize_list=resizes, bbox_list=[None], inference=True)
        emb = model.get_visual_embs(images_sam)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    llm_cfg = bench.c1_case()[0]
    ocfg = dict(llm=llm_cfg, seg_token_idx=bench.SEG, loc_token_idx=bench.LOC,
                sam=dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=[7, 15, 23, 31], window_size=14, patch_size=16, img_size=1024,
                         out_chans=256))
    torch.set_num_threads(min(os.cpu_count(), 64))
    a = (images_sam.cpu(), vis.cpu(), ids.cpu(), mask.cpu(), sizes, resizes)
    with torch.no_grad():
        ref = O.ullava_forward(sd, ocfg, *a)
        truth = O.ullava_forward(bench.F32View(sd), ocfg, a[0].float(), a[1].float(), *a[2:])
    rec = {}
    for name, hip, r_, t_ in (("sam_image_embedding", emb, ref["image_embeddings"], truth["image_embeddings"]),
                              ("logits", out["logits"], ref["logits"], truth["logits"]),
                              ("pred_masks", out["pred_masks"][0], ref["pred_masks"][0], truth["pred_masks"][0]),
                              ("pred_boxes", out["pred_boxes"][0], ref["pred_boxes"][0], truth["pred_boxes"][0])):
        e_ref, e_hip = _rel(r_, t_), _rel(hip, t_)
        rec[name] = dict(oracle_bf16_err=round(e_ref, 5), hip_err=round(e_hip, 5), hip_vs_oracle=round(_rel(hip, r_), 5), shape=list(t_.shape))
    print(f"RES full depth vs oracle ({dt}):", json.dumps(rec))
    assert tuple(out["pred_masks"][0].shape) == (3, 480, 640) and out["pred_masks"][0].dtype == torch.float32
    for name, floor in (("sam_image_embedding", 2.0 ** -5), ("logits", 2.0 ** -6), ("pred_masks", 2.0 ** -5), ("pred_boxes", 2.0 ** -5)):
        assert rec[name]["hip_err"] <= max(1.5 * rec[name]["oracle_bf16_err"], floor), (name, rec[name])
    # mask SIGNS (the segmentation itself): where the fp32 logit is at least 4 x the oracle's bf16 error away from zero, all three agree
    hm, rm, tm = out["pred_masks"][0].cpu(), ref["pred_masks"][0].float(), truth["pred_masks"][0]
    clear = tm.abs() > 4.0 * float((rm - tm).abs().max())
    assert bool(((hm > 0) == (tm > 0))[clear].all()) and bool(((rm > 0) == (tm > 0))[clear].all())
    print(f"mask pixels decided with margin: {float(clear.float().mean()) * 100:.1f} %, sign agreement there: exact")


# ---- G15 / G16: the REFERENCE ITSELF at full depth (tests/golden/gen_golden_full_depth.py ran /root/reference on the build container's CPU, in
# bf16, fp16 and fp32, and asserted reference == oracle bit for bit on every output) -------------------------------------------------------------
_REF = {}


def _ref_models():
    """{dtype: HIP UllavaForCausalLM} holding the fixtures' weights: `weights.seeded_tensor(name, shape, 15, hf_init=True)` per tensor (`llm.*`
    under the core model's names), generated ONCE in fp32 on the host by a thread pool and rounded once into the bf16 and the fp16 model."""
    if _REF:
        return _REF
    import bench
    from concurrent.futures import ThreadPoolExecutor
    C, MU, W = pkg("configuration"), pkg("modeling_ullava"), pkg("weights")
    fx = load_fixture("g16_res_full_depth_bf16.pt")
    llm = dict(vision_config=dict(image_size=224, patch_size=14), vision_hidden_layer=-2, projector_type="mlp", projector_from_scratch=False,
               mm_token_ids=dict(bench.MM), vocab_size=32011)
    models = {}
    with torch.no_grad():
        for dt in (torch.bfloat16, torch.float16, torch.float32):          # (float32: the fp32 build, holding the bf16-ROUNDED values -- the fixtures' "truth" model)
            m = MU.UllavaForCausalLM(C.UllavaConfig(llm_config=llm, seg_token_idx=bench.SEG, loc_token_idx=bench.LOC), device=DEV, dtype=dt)
            m.llm.strict_checks = False
            models[dt] = m
        sds = {dt: m.state_dict(keep_vars=True) for dt, m in models.items()}       # the parameters / buffers themselves
        shapes = {k: tuple(v) for k, v in fx["shapes"].items()}
        assert {k: tuple(v.shape) for k, v in sds[torch.bfloat16].items()} == shapes, "state-dict keys / shapes differ from the reference's"
        keys = list(shapes)

        def gen(k):
            return k, W.seeded_tensor(k[4:] if k.startswith("llm.") else k, shapes[k], fx["seed"], torch.float32, hf_init=True)
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
            for i in range(0, len(keys), 48):
                for k, t in ex.map(gen, keys[i:i + 48]):
                    for dt in models:
                        sds[dt][k].data.copy_(t.to(dt) if dt != torch.float32 else t.to(torch.bfloat16).float())
        for m in models.values():
            m.llm._packed = None                                   # (the re-layouts are made from the filled parameters on first use)
            if hasattr(m, "_sam"):
                m._sam.invalidate()
    _REF.update(models)
    return _REF


def _sample_err(hip, ref, truth, absmax):
    h, r, t = (x.detach().float().cpu() for x in (hip, ref, truth))
    return float((h - t).abs().max()) / absmax, float((r - t).abs().max()) / absmax, float((h - t).pow(2).mean().sqrt()), float((r - t).pow(2).mean().sqrt())


def _check_logits_against_reference(logits, rec, what):
    """HIP logits [S, V] against the fixture's record of the reference's 16-bit logits and its fp32 logits: errors on the SAME samples (whole rows at
    12 positions, every 64th column at all positions), and token ids EQUAL to the reference's (16-bit and fp32, which agree there) at every position
    whose fp32 top-1 / top-2 gap exceeds 4 standard deviations of the reference's own 16-bit noise on a logit difference."""
    rows = rec["rows"]
    e_hip_r, e_ref_r, rms_hip_r, rms_ref_r = _sample_err(logits[rows], rec["ref_rows"], rec["truth_rows"], rec["truth_absmax"])
    cs = rec["col_stride"]
    e_hip_c, e_ref_c, rms_hip_c, rms_ref_c = _sample_err(logits[:, ::cs], rec["ref_cols"], rec["truth_cols"], rec["truth_absmax"])
    gap = rec["truth_top_values"][:, 0] - rec["truth_top_values"][:, 1]
    gated = gap > 4.0 * rec["sigma"]
    ah = logits.float().argmax(-1).cpu()
    at, ar = rec["truth_argmax"].long(), rec["ref_argmax"].long()
    mism = int(((ah != at) | (ah != ar))[gated].sum())
    out = dict(rows=dict(hip_err=round(e_hip_r, 5), ref_err=round(e_ref_r, 5), hip_rms=round(rms_hip_r, 5), ref_rms=round(rms_ref_r, 5)),
               cols=dict(hip_err=round(e_hip_c, 5), ref_err=round(e_ref_c, 5), hip_rms=round(rms_hip_c, 5), ref_rms=round(rms_ref_c, 5)),
               positions=int(gap.numel()), positions_gated=int(gated.sum()), gated_mismatches=mism,
               argmax_agree_hip_fp32=round(float((ah == at).float().mean()), 4), argmax_agree_ref_fp32=round(float((ar == at).float().mean()), 4),
               argmax_agree_hip_ref=round(float((ah == ar).float().mean()), 4), reference_full_tensor=rec["full_stats"])
    print(f"{what} logits vs the reference:", json.dumps(out))
    for k in ("rows", "cols"):
        assert out[k]["hip_err"] <= max(1.5 * out[k]["ref_err"], 2.0 ** -6), (k, out[k])
        assert out[k]["hip_rms"] <= 1.25 * out[k]["ref_rms"], (k, out[k])
    assert mism == 0, f"{what}: token ids differ from the reference's at {mism} margin-gated positions"
    assert int(gated.sum()) >= gap.numel() // 10
    assert out["argmax_agree_hip_fp32"] >= out["argmax_agree_ref_fp32"] - 0.03
    return out


@pytest.mark.parametrize("tag,dt", [("bf16", torch.bfloat16), ("fp16", torch.float16)])
def test_c1_full_depth_against_reference_fixture(tag, dt):
    """G15: BASELINE.json configs[0] at FULL size against the reference itself (`UllavaCoreForCausalLM.forward`, models/ullava_core.py:279-355, run
    on the build container's CPU): hidden states 0 / 8 / 16 / 24 / 32 and the logits as close to the reference's fp32 run as the reference's own
    16-bit run is (x1.5) on the committed samples, and margin-gated token ids equal to the reference's."""
    fx = load_fixture(f"g15_c1_full_depth_{tag}.pt")
    model = _ref_models()[dt].llm
    with torch.no_grad():
        out = model.forward(input_ids=fx["input_ids"].to(DEV), attention_mask=fx["attention_mask"].to(DEV), images=fx["images"].to(DEV),
                            output_hidden_states=True)
    assert tuple(out.logits.shape) == (1, 291, 32011) and out.logits.dtype == dt and len(out.hidden_states) == 33
    rec = {}
    st = fx["hid_stride"]
    for li in fx["hid_layers"]:
        e_hip, e_ref, _, _ = _sample_err(out.hidden_states[li][0, :, ::st], fx["ref_hidden"][li], fx["truth_hidden"][li], fx["truth_hidden_absmax"][li])
        rec[f"hidden_{li}"] = dict(hip_err=round(e_hip, 5), ref_err=round(e_ref, 5))
        assert e_hip <= max(1.5 * e_ref, 2.0 ** -7), (li, e_hip, e_ref)
    print(f"G15 ({tag}) hidden states vs the reference:", json.dumps(rec))
    _check_logits_against_reference(out.logits[0], fx["logits"], f"G15 ({tag})")


def _res_against_reference_fixture(tag, dt):
    """G16: batch-1 C3 at FULL size against the reference itself (`UllavaForCausalLM.forward(inference=True)`, models/ullava.py:152-268; SAM ViT-H
    image_encoder.py:110-125 -- in fp16 with its fp32 neck, :117-124): SAM image embedding, mask logits, boxes and LLaMA logits as close to the
    reference's fp32 run as the reference's own 16-bit run is (x1.5) on the committed samples; mask SIGNS equal wherever the fp32 logit clears
    4x the reference's own 16-bit error; margin-gated token ids equal."""
    from helpers import digest_matches
    fx = load_fixture(f"g16_res_full_depth_{tag}.pt")
    model = _ref_models()[dt]
    g = torch.Generator().manual_seed(fx["inputs_seed"])                      # gen_golden_full_depth.c3_inputs: ids, image, then the SAM image
    torch.randint(5, 32000, (120,), generator=g)
    torch.randn(1, 3, 224, 224, generator=g)
    images_sam = torch.randn(1, 3, 1024, 1024, generator=g).to(dt)
    assert digest_matches(images_sam, fx["images_sam_digest"]), "the regenerated SAM input differs from the one the reference saw"
    sizes, resizes = [tuple(x) for x in fx["size_list"]], [tuple(x) for x in fx["resize_list"]]
    with torch.no_grad():
        out = model.forward(images_sam=images_sam.to(DEV), images=fx["images"].to(DEV), input_ids=fx["input_ids"].to(DEV), labels=None,
                            attention_mask=fx["attention_mask"].to(DEV), mask_list=[None], size_list=sizes, resize_list=resizes, bbox_list=[None],
                            inference=True)
        emb = model.get_visual_embs(images_sam.to(DEV))
    pm = out["pred_masks"][0]
    assert tuple(pm.shape) == (3, 480, 640) and pm.dtype == torch.float32 and tuple(out["pred_boxes"][0].shape) == (3, 4)
    rec = {}
    for name, hip, r_, t_, amax, floor in (
            ("sam_image_embedding", emb[:, ::8, ::2, ::2], fx["ref_emb"], fx["truth_emb"], fx["truth_emb_absmax"], 2.0 ** -5 if dt == torch.bfloat16 else 2.0 ** -8),
            ("pred_masks", pm[:, ::4, ::4], fx["ref_masks"], fx["truth_masks"], fx["truth_masks_absmax"], 2.0 ** -5 if dt == torch.bfloat16 else 2.0 ** -8),
            ("pred_boxes", out["pred_boxes"][0], fx["ref_boxes"], fx["truth_boxes"], float(fx["truth_boxes"].float().abs().max()),
             2.0 ** -5 if dt == torch.bfloat16 else 2.0 ** -8)):
        e_hip, e_ref, rms_hip, rms_ref = _sample_err(hip, r_, t_, amax)
        rec[name] = dict(hip_err=round(e_hip, 5), ref_err=round(e_ref, 5), hip_rms=round(rms_hip, 6), ref_rms=round(rms_ref, 6))
        assert e_hip <= max(1.5 * e_ref, floor), (name, rec[name])
    tm, hm, rm = fx["truth_masks"], pm[:, ::4, ::4].cpu(), fx["ref_masks"]
    clear = tm.abs() > fx["mask_sign_margin"]
    assert bool(((hm > 0) == (tm > 0))[clear].all()) and bool(((rm > 0) == (tm > 0))[clear].all())
    rec["mask_pixels_decided_with_margin"] = round(float(clear.float().mean()), 4)
    rec["reference_full_tensor"] = fx["ref_err_full"]
    print(f"G16 ({tag}) RES full depth vs the reference:", json.dumps(rec))
    _check_logits_against_reference(out["logits"][0], fx["logits"], f"G16 ({tag})")


@pytest.mark.parametrize("tag,dt", [("bf16", torch.bfloat16), ("fp16", torch.float16)])
def test_res_full_depth_against_reference_fixture(tag, dt):
    _res_against_reference_fixture(tag, dt)


def test_full_depth_fp32_against_the_references_fp32_runs():
    """`--dtype fp32` (inference_ullava.py:25,164-168) at FULL size: the fp32 build (csrc/f32.hip) holding the bf16-rounded weights against the
    reference's own fp32 runs on those weights -- the "truth" halves of the G15 / G16 bf16 fixtures (UllavaCoreForCausalLM.forward on C1,
    UllavaForCausalLM.forward(inference=True) on batch-1 C3).  fp32 has no rounding points: the bound is summation-order noise through 32 LLaMA +
    32 SAM blocks (1e-4 of each tensor's maximum), token ids equal wherever the fp32 top-1 / top-2 gap exceeds 1e-3, mask signs equal where
    |logit| > 1e-3 of the maximum."""
    from helpers import digest_matches
    model = _ref_models()[torch.float32]
    assert model.dtype == torch.float32
    rec = {}
    fx = load_fixture("g15_c1_full_depth_bf16.pt")
    with torch.no_grad():
        out = model.llm.forward(input_ids=fx["input_ids"].to(DEV), attention_mask=fx["attention_mask"].to(DEV), images=fx["images"].float().to(DEV),
                                output_hidden_states=True)
    assert out.logits.dtype == torch.float32 and tuple(out.logits.shape) == (1, 291, 32011)
    st = fx["hid_stride"]
    for li in fx["hid_layers"]:
        e = float((out.hidden_states[li][0, :, ::st].cpu() - fx["truth_hidden"][li]).abs().max()) / fx["truth_hidden_absmax"][li]
        rec[f"c1_hidden_{li}"] = e
        assert e <= 1e-4, (li, e)

    def logits_check(lg, lrec, what):
        e_r = float((lg[lrec["rows"]].cpu() - lrec["truth_rows"]).abs().max()) / lrec["truth_absmax"]
        e_c = float((lg[:, ::lrec["col_stride"]].cpu() - lrec["truth_cols"]).abs().max()) / lrec["truth_absmax"]
        gap = lrec["truth_top_values"][:, 0] - lrec["truth_top_values"][:, 1]
        am = lg.argmax(-1).cpu()
        clear = gap > 1e-3
        rec[what] = dict(rows=e_r, cols=e_c, ids_equal_all=float((am == lrec["truth_argmax"].long()).float().mean()), positions_clear=int(clear.sum()))
        assert e_r <= 1e-4 and e_c <= 1e-4, rec[what]
        assert bool((am == lrec["truth_argmax"].long())[clear].all()), f"{what}: fp32 token ids differ from the reference's fp32 ids"
        assert int(clear.sum()) >= int(0.9 * gap.numel())
    logits_check(out.logits[0], fx["logits"], "c1_logits")
    fx = load_fixture("g16_res_full_depth_bf16.pt")
    g = torch.Generator().manual_seed(fx["inputs_seed"])
    torch.randint(5, 32000, (120,), generator=g)
    torch.randn(1, 3, 224, 224, generator=g)
    images_sam = torch.randn(1, 3, 1024, 1024, generator=g).to(torch.bfloat16)
    assert digest_matches(images_sam, fx["images_sam_digest"])
    sizes, resizes = [tuple(x) for x in fx["size_list"]], [tuple(x) for x in fx["resize_list"]]
    with torch.no_grad():
        o = model.forward(images_sam=images_sam.float().to(DEV), images=fx["images"].float().to(DEV), input_ids=fx["input_ids"].to(DEV), labels=None,
                          attention_mask=fx["attention_mask"].to(DEV), mask_list=[None], size_list=sizes, resize_list=resizes, bbox_list=[None],
                          inference=True)
        emb = model.get_visual_embs(images_sam.float().to(DEV))
    pm = o["pred_masks"][0]
    assert pm.dtype == torch.float32 and tuple(pm.shape) == (3, 480, 640)
    for name, hip, t_, amax in (("res_sam_embedding", emb[:, ::8, ::2, ::2], fx["truth_emb"], fx["truth_emb_absmax"]),
                                ("res_masks", pm[:, ::4, ::4], fx["truth_masks"], fx["truth_masks_absmax"]),
                                ("res_boxes", o["pred_boxes"][0], fx["truth_boxes"], float(fx["truth_boxes"].abs().max()))):
        rec[name] = float((hip.cpu().float() - t_.float()).abs().max()) / amax
        assert rec[name] <= 1e-4, (name, rec[name])
    tm = fx["truth_masks"]
    clear = tm.abs() > 1e-3 * fx["truth_masks_absmax"]
    assert bool(((pm[:, ::4, ::4].cpu() > 0) == (tm > 0))[clear].all())
    logits_check(o["logits"][0], fx["logits"], "res_logits")
    print("fp32 build at full depth vs the reference's fp32 runs:", json.dumps(rec))


RCCL_SCRIPT = r"""
import importlib, os, sys, json
sys.path.insert(0, os.environ["ULL_ROOT"])
import torch, torch.distributed as dist
D = importlib.import_module("u-llava_amd.dist")
ops = importlib.import_module("u-llava_amd.ops")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)                       # RCCL, world size 1
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.barrier()
rate, units, tmax = D.global_rate(64.0, 2.0, device=dev)            # the two scalar all-reduces (MAX, SUM) through RCCL
assert (rate, units, tmax) == (32.0, 64.0, 2.0), (rate, units, tmax)
g = torch.Generator(device="cuda").manual_seed(3)
shapes = [(1000, 37), (4099,), (64, 64), (5,)]
params = [torch.nn.Parameter(torch.zeros(s, device=dev, dtype=torch.bfloat16)) for s in shapes]
params.append(torch.nn.Parameter(torch.zeros(33, 3, device=dev, dtype=torch.float32)))       # fp32 master gradient: all_reduce branch
for p in params:
    p.grad = torch.randn(p.shape, device=dev, generator=g).to(p.dtype)
nograd = torch.nn.Parameter(torch.zeros(17, device=dev, dtype=torch.bfloat16))                # a head this rank's batch never touched
params.insert(2, nograd)
before = [None if p.grad is None else p.grad.clone() for p in params]
assert D.allreduce_gradients(params) == 0                            # world 1 without force_direct: nothing to do
nb = D.allreduce_gradients(params, bucket_bytes=50000, force_direct=True)
assert nb >= 3, nb
for p, b in zip(params, before):
    if b is None:
        assert p.grad is not None and float(p.grad.abs().max()) == 0.0
    else:
        assert torch.equal(p.grad, b), "world-1 direct exchange must be the identity"
# ZeRO-2 optimizer cycle through RCCL at world 1 (all_to_all_single -> sum_slabs -> AdamW on the shard -> all_gather_into_tensor): must equal
# the collective-free path bit for bit
O_ = importlib.import_module("u-llava_amd.optim")
def run(force):
    gg = torch.Generator(device="cuda").manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(s, device=dev, generator=gg).to(torch.bfloat16)) for s in [(300, 41), (77,), (64, 64)]]
    opt = O_.ShardedAdamW(ps, lr=1e-2, weight_decay=0.01, max_grad_norm=1.0, bucket_bytes=30000, force_collectives=force)
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn(p.shape, device=dev, generator=gg).to(torch.bfloat16)
        opt.step()
    return [p.detach().clone() for p in ps]
a_, b_ = run(True), run(False)
assert all(torch.equal(x, y) for x, y in zip(a_, b_)), "ZeRO-2 step through RCCL at world 1 differs from the local path"
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"rccl_world1": "ok", "buckets": nb}))
"""


def _env():
    env = dict(os.environ, ULL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_rccl_world1_gradient_exchange_and_aggregation():
    """RCCL has never run in this project's GPU tests (no multi-GPU box): at world size 1 every call still goes through the library --
    init, barrier, all_reduce(MAX / SUM), all_to_all_single, all_gather_into_tensor -- and the direct-exchange branch of
    `allreduce_gradients` (with `ull_sum_slabs`) must be the identity."""
    r = subprocess.run([sys.executable, "-c", RCCL_SCRIPT], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert '"rccl_world1": "ok"' in r.stdout


def test_bench_runs_with_rccl_process_group_at_one_gpu():
    """`bench.py --gpus 1 --init-pg`: the driver's launch path with the process group up -- barrier-bracketed timing and the scalar
    aggregation execute on RCCL -- on a reduced batch so that the test stays short; the JSON line must carry the same fields."""
    env = _env()
    for k in ("MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--init-pg", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-res", "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["outputs_finite"] is True
    assert rec["process_group"] == {"backend": "nccl", "rccl_world_size": 1, "rank": 0}          # what RCCL itself reports, not what was asked for
    assert rec["probes"]["order"] == "after destroy_process_group" and rec["per_rank_ms_per_step"]["min"] <= rec["ms_per_step"] + 0.01
    assert rec["config"]["per_gpu_batch"] == 4 and rec["value"] > 0 and abs(rec["value"] - 4 * 2 / (rec["ms_per_step"] * 2e-3)) < 0.05 * rec["value"]


def test_bench_two_ranks_with_real_steps_on_one_gpu():
    """The N > 1 protocol of `bench.py` with REAL HIP steps: two ranks launched by the script itself (`--gpus 2`), both computing on cuda:0
    (`--share-gpu`; RCCL refuses two ranks on one device -- profiles/r04_rccl_two_ranks_one_gpu.txt -- so gloo carries the barriers and the
    scalar aggregation), C2 at batch 4, timed regions for C2 only.  The line must report two ranks seen by the collective layer, the SUM of
    the ranks' images over the MAX of their times, the per-rank range, and the probes after the process group is gone."""
    env = _env()
    for k in ("MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--workload", "c2",
                        "--batch", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-pin"],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["process_group"] == {"backend": "gloo", "rccl_world_size": 2, "rank": 0}
    assert rec["config"]["per_gpu_batch"] == 4 and rec["config"]["global_batch"] == 8 and rec["outputs_finite"] is True
    sp = rec["per_rank_ms_per_step"]
    assert 0 < sp["min"] <= sp["max"] <= rec["ms_per_step"] + 0.01
    assert abs(rec["value"] - 8 * 3 / (rec["ms_per_step"] * 3e-3)) < 0.05 * rec["value"]            # SUM(images) / MAX(elapsed)
    assert rec["probes"]["order"] == "after destroy_process_group"
