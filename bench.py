#!/usr/bin/env python
"""bench.py -- images/sec of the u-LLaVA multimodal forward on MI355X through the HIP path, with the roofline of the dominant
kernel and a CPU-oracle baseline.

  python bench.py [--gpus N --steps K --warmup W]

N > 1: this script launches N ranks itself (re-exec under `torch.distributed.run`, one rank per GPU, RCCL over xGMI) unless it
is already running inside such a launch (WORLD_SIZE set by the driver's own `python -m torch.distributed.run ... bench.py`).

A "step" = one forward of the per-GPU batch.  Headline workload = BASELINE.json config C4 (336x336 synthetic images, 64-token
prompts, S = 643, per-GPU batch 32, weak scaling); the same JSON line carries a `res` sub-record = config C3 (full RES forward:
+ SAM ViT-H encoder at 1024x1024, 3 [SEG]/[LOC] per image, MaskDecoder, postprocess; batch 8) timed with the same protocol, so
both halves of BASELINE.json's metric ("ViT-L-336 + LLaMA-7B + SAM RES forward") are measured in the driver's run, and `extra.c2` /
`extra.c5` = BASELINE.json configs[1] (VQA, batch 16, ragged prompts) and configs[4] (8-frame video clips, 8 per GPU) at their own sizes
(`config.also_timed` repeats the three in brief).  Inputs and random-init weights are resident in HBM before the timed region.
Every timed region is bracketed by barriers; the process group is destroyed after the last one and rank 0's single-rank probes
(roofline, patchify, CPU baseline) run afterwards, on a node where no other rank is working.  Rank 0 prints ONE JSON line, which also
carries what the collective layer itself saw (`process_group.rccl_world_size`) and the per-rank step time range.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "images/sec/GPU (ViT-L-336 + LLaMA-7B + SAM RES forward) at 1/2/4/8 MI355X"      # BASELINE.json, verbatim
MM = dict(IMG_START=32001, IMG_END=32002, IMG_PATCH=32003, VID_START=32004, VID_END=32005, VID_PATCH=32006)
WORKLOADS = {
    # name: (image_size, prompt_tokens, per_gpu_batch, description)
    "c4": (336, 64, 32, "C4: ViT-L/14-336 + LLaMA-7B instruct forward, 336x336 image + 64-token prompt (S=643), batch 32/GPU"),
    # the reference's own CPU-runnable case (BASELINE.json configs[0]; the shape `cpu_baseline` times the oracle on): one image, latency-bound
    "c1": (224, 32, 1, "C1: ViT-L/14-224 + LLaMA-7B forward, one 224x224 image + 32-token prompt (S=291), batch 1"),
    "c2": (224, 64, 16, "C2: ViT-L/14-224 + LLaMA-7B VQA forward, 224x224 image + 32..64-token prompts right-padded to S=323, batch 16"),
    # full RES path: + SAM ViT-H encoder on 1024x1024, 3 [SEG]+[LOC] rounds per sample, prompt-encoder + mask decoder + postprocess
    "res": (224, 120, 8, "C3: full RES forward (ViT-L/14-224 + LLaMA-7B + SAM ViT-H 1024x1024 + MaskDecoder, 3 [SEG]/[LOC] per image), batch 8"),
    "c5": (224, 32, 8, "C5: video forward, 8-frame 224x224 clips (per-frame ViT-L, 8+256 pooled tokens) + 32-token prompt (S=299), 8 clips/GPU"),
    # forward with labels + loss.backward() + gradient exchange, the reference's stage-2 trainable set (train_ullava.py:229-261)
    # forward with labels + loss.backward() + gradient exchange at the reference's stage-2 settings; --train-config picks
    # full (train_ullava.py:239-245, configs/train/ullava.yaml: every llm.model / lm_head / projector weight trainable, 16 samples per GPU),
    # lora (configs/train/ullava_lora.yaml: r = 8 adapters on q_proj / v_proj + lm_head + embed_tokens, 32 per GPU) or qv (round 2's set)
    "train": (224, 64, 16, "F4: training step (forward + backward + gradient exchange), ViT-L/14-224 + LLaMA-7B (S=323)"),
}
PEAK_BF16_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
SEG, LOC = 32007, 32008


def llama_flops(S, V, D=4096, I=11008, L=32):
    """32*[2S(4D^2 + 3DI) + 4S^2 D] + 2SDV per sample (2*MAC, undiscounted attention)."""
    return L * (2 * S * (4 * D * D + 3 * D * I) + 4 * S * S * D) + 2 * S * D * V


def clip_flops(P, D=1024, I=4096, L=23, K=588):
    T = P + 1
    return L * (2 * T * (4 * D * D + 2 * D * I) + 4 * T * T * D) + 2 * P * K * D


def init_random_(model, seed):
    """HF-style random init directly on the GPU (N(0, 0.02) matrices, ones for norm weights, zero biases)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() == 1 and ("norm" in name or "layrnorm" in name) and name.endswith("weight"):
            p.data.fill_(1.0)
        elif p.dim() == 1 and name.endswith("bias"):
            p.data.zero_()
        else:
            p.data.normal_(0.0, 0.02, generator=g)


DTYPE = torch.bfloat16            # --dtype fp16: the reference's --dtype fp16 option (a side check; its default and the contract line are bf16)


def build_model(image_size, device, seed=0, with_sam=False):
    C = importlib.import_module("u-llava_amd.configuration")
    llm = dict(vision_config=dict(image_size=image_size, patch_size=14), vision_hidden_layer=-2, projector_type="mlp",
               projector_from_scratch=False,            # train_ullava.py:162 (stage 2); forward-identical either way
               mm_token_ids=dict(MM), vocab_size=32011)
    if with_sam:
        M = importlib.import_module("u-llava_amd.modeling_ullava")
        model = M.UllavaForCausalLM(C.UllavaConfig(llm_config=llm, seg_token_idx=SEG, loc_token_idx=LOC), device=device, dtype=DTYPE)
        init_random_(model, seed)
        g = torch.Generator(device="cuda").manual_seed(seed + 1)
        model.visual_model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix.normal_(0.0, 1.0, generator=g)
        core, cfg = model.llm, model.config.llm_config
    else:
        M = importlib.import_module("u-llava_amd.modeling_core")
        cfg = C.UllavaCoreConfig(**llm)
        model = core = M.UllavaCoreForCausalLM(cfg, device=device, dtype=DTYPE)
        init_random_(model, seed)
    core.strict_checks = False        # no host sync inside the timed region (the check itself is covered by tests)
    core.pack_weights()
    return model, cfg


def make_inputs(cfg, batch, prompt_tokens, device, seed, ragged=False, video=False):
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    isz = cfg.vision_config.image_size
    P = (isz // cfg.vision_config.patch_size) ** 2
    txt = torch.randint(5, 32000, (batch, prompt_tokens), device=device, generator=g)
    if video:
        T = 8
        vis = torch.randn(batch, 3, T, isz, isz, device=device, generator=g).to(DTYPE)
        head = torch.tensor([1, MM["VID_START"]] + [MM["VID_PATCH"]] * (T + P) + [MM["VID_END"]], device=device).expand(batch, -1)
    else:
        vis = torch.randn(batch, 3, isz, isz, device=device, generator=g).to(DTYPE)
        head = torch.tensor([1, MM["IMG_START"]] + [MM["IMG_PATCH"]] * P + [MM["IMG_END"]], device=device).expand(batch, -1)
    ids = torch.cat([head, txt], dim=1).contiguous()
    mask = torch.ones_like(ids)
    if ragged:        # C2: prompts of 32..64 tokens, right-padded with the pad id (0) and masked, like the reference's collator
        lens = torch.linspace(prompt_tokens // 2, prompt_tokens, batch).round().long().tolist()
        for b, n in enumerate(lens):
            ids[b, head.shape[1] + n:] = 0
            mask[b, head.shape[1] + n:] = 0
    return vis, ids, mask


def gemm_sources_sha():
    """sha256 (first 16 hex digits) of the GEMM kernel source: ties a committed PMC traffic record to the kernel it measured."""
    h = hashlib.sha256()
    for n in ("gemm.hip", "ull_common.h"):
        p = os.path.join(ROOT, "u-llava_amd", "csrc", n)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def in_situ_kernel_times():
    """Average duration of the GEMM launches whose kernel instantiation names them (gate/up + SwiGLU = gemm256w4_kernel<true, false>, q|k|v + RoPE
    = <false, true>) inside the C4 step, from the newest committed `profiles/rNN_c4_kernel_stats.md` (rocprofv3 --kernel-trace --stats of
    `bench.py --steps 3`, tools/prof_r05.sh).  o_proj / down share one instantiation with the CLIP launches and cannot be told apart by name."""
    out = {}
    try:
        fns = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_c4_kernel_stats.md"))
        if not fns:
            return out
        for ln in open(os.path.join(ROOT, "profiles", fns[-1])):
            c = [t.strip() for t in ln.split("|")]
            if len(c) < 6:
                continue
            for key, tag in (("gate_up+swiglu", "gemm256w4_kernel<true, false>"), ("qkv+rope", "gemm256w4_kernel<false, true>")):
                if tag in c[1]:
                    out[key] = {"avg_us": float(c[4]), "calls": int(c[2]), "record": "profiles/" + fns[-1]}
    except (OSError, ValueError):
        pass
    return out


def gemm_roofline(cfg, tokens, device, iters=40, warm=10):
    """Time the four GEMM launches of one LLaMA layer at the benchmark's token count with HIP events on the launch stream.
    The chip runs these kernels against its socket power cap (1.4 kW: tools/hot_power.sh), and the clock needs a few launches to settle
    after the host-side set-up of each shape, so every shape gets `warm` untimed launches and the average is taken over `iters`."""
    ops = importlib.import_module("u-llava_amd.ops")
    D, I = cfg.hidden_size, cfg.intermediate_size
    shapes = [("qkv+rope", 3 * D, D, False), ("o_proj", D, D, False), ("gate_up+swiglu", 2 * I, D, True), ("down", D, I, False)]
    g = torch.Generator(device="cuda").manual_seed(7)
    per = []
    tot_t = tot_f = 0.0
    in_situ = in_situ_kernel_times()
    for name, N, K, sw in shapes:
        x = (torch.randn(tokens, K, device=device, generator=g)).to(torch.bfloat16)
        w = (torch.randn(N, K, device=device, generator=g) * 0.02).to(torch.bfloat16)
        ops.register_tiled(w)
        out = torch.empty(tokens, N // 2 if sw else N, device=device, dtype=torch.bfloat16)
        if name == "qkv+rope":
            # the launch the model makes: q | k | v projection with RoPE in the epilogue (the plain projection the probe timed up to round 4
            # is 4-5 % shorter than what runs in the step)
            pos = torch.arange(tokens, device=device, dtype=torch.int64) % 643
            inv = (1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).to(device)
            cs, sn = ops.rope_table(pos, inv, torch.bfloat16)
            fn = lambda: ops.linear_qkv_rope(x, w, cs, sn, 2 * D, 128, out=out)
        else:
            fn = lambda: ops.linear(x, w, swiglu=sw, out=out)
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * tokens * N * K
        rec = dict(gemm=name, M=tokens, N=N, K=K, ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1))
        situ = in_situ.get(name) if tokens == 32 * 643 else None
        if situ:                                                   # the same launch inside the C4 step, from the committed rocprofv3 kernel trace
            # nested under "archived_profile": the committed rocprofv3 kernel trace of an EARLIER run of this bench -- not a measurement of this run
            rec["archived_profile"] = {"in_step_us": situ["avg_us"], "in_step_tflops": round(fl / situ["avg_us"] / 1e6, 1), "record": situ["record"],
                                       "calls": situ["calls"]}
        per.append(rec)
        tot_t += ms
        tot_f += fl
    achieved = tot_f / tot_t / 1e9
    # HBM-side traffic comes from separate rocprofv3 --pmc passes (they cannot run inside this process); the committed record is
    # only quoted while it describes THIS kernel source (sha of csrc/gemm.hip), otherwise traffic is null
    traffic, detail = None, None
    sha = gemm_sources_sha()
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if fn.endswith("gemm_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
            except Exception:
                continue
            if tj.get("kernel_source_sha") == sha:
                traffic = tj["traffic_bytes_per_launch"]
                detail = {"record": "profiles/" + fn, "kernel": tj["kernel"], "algorithmic_bytes_per_launch": tj["algorithmic_bytes_per_launch"],
                          "l2_hit_rate": tj.get("l2", {}).get("hit_rate"), "note": tj["note"],
                          "fabric_read_floor_model_bytes": tj.get("fabric_read_floor_model_bytes"), "ea_request_counters": tj.get("ea"),
                          "hbm_bytes_bounds": tj.get("hbm_bytes_bounds"),
                          "mfma_pipe_busy_fraction": tj.get("mfma", {}).get("mfma_pipe_busy_fraction_of_simd_cycles")}
            else:
                detail = {"record": "profiles/" + fn, "stale": True,
                          "note": f"PMC record was taken on kernel source {tj.get('kernel_source_sha')}, current source is {sha}: not quoted"}
            break
    return dict(bound="mfma", kernel="big::gemm256w4_kernel (LLaMA-7B layer: qkv, o, gate/up+SwiGLU, down; 2*M*N*K flop per launch)",
                achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(achieved / PEAK_BF16_TFLOPS, 4),
                traffic=traffic, traffic_detail=detail, per_launch=per)


def _time_launches(fn, iters=40, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def sam_gemm_roofline(device, batch=8):
    """The RES half of the metric: the four Linear launches of one SAM ViT-H block (image_encoder.py:196-260, common.py:13-26) at the RES
    workload's token count (batch x 4096 tokens, d = 1280, MLP 5120) with their real epilogues -- qkv + bias, proj + bias + residual,
    lin1 + bias + erf-GELU, lin2 + bias + residual -- timed like gemm_roofline (HIP events on the launch stream), 2 * M * N * K flop per
    launch.  These K = 1280 launches are 44 % of a RES step and the kernels furthest below the LLaMA layer's rate."""
    ops = importlib.import_module("u-llava_amd.ops")
    M, D, I = batch * 4096, 1280, 5120
    g = torch.Generator(device="cuda").manual_seed(8)
    shapes = [("qkv+bias", 3 * D, D, None, False), ("proj+bias+residual", D, D, None, True), ("lin1+bias+gelu", I, D, "gelu", False),
              ("lin2+bias+residual", D, I, None, True)]
    per, tot_t, tot_f = [], 0.0, 0.0
    for name, N, K, act, res in shapes:
        x = torch.randn(M, K, device=device, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=device, generator=g) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=device, generator=g).to(torch.bfloat16)
        r = torch.randn(M, N, device=device, generator=g).to(torch.bfloat16) if res else None
        ops.register_tiled(w)
        out = torch.empty(M, N, device=device, dtype=torch.bfloat16)
        ms = _time_launches(lambda: ops.linear(x, w, b, act=act, residual=r, out=out))
        fl = 2.0 * M * N * K
        per.append(dict(gemm=name, M=M, N=N, K=K, ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1)))
        tot_t += ms
        tot_f += fl
    ach = tot_f / tot_t / 1e9
    return dict(bound="mfma", kernel="big::gemm256w4_kernel / gemm256d_kernel (SAM ViT-H block: qkv, proj, lin1 + GELU, lin2; 2*M*N*K flop per launch)",
                achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4), traffic=None,
                us_per_block=round(tot_t * 1e3, 1), per_launch=per)


def patchify_record(device, batch=32, image=336):
    """north_star's HBM target line: ViT patchify at batch 32 -- Conv2d(3, 1024, 14, stride 14) straight from the pixels (`ull_patchify_*`) --
    algorithmic bytes (pixels in + patches out + packed weights, BASELINE.md section 2: 60.6 MB at 336^2) over the average launch time."""
    ops = importlib.import_module("u-llava_amd.ops")
    g = torch.Generator(device="cuda").manual_seed(9)
    img = torch.randn(batch, 3, image, image, device=device, generator=g).to(torch.bfloat16)
    w = (torch.randn(1024, 3, 14, 14, device=device, generator=g) * 0.02).to(torch.bfloat16)
    wp = ops.pack_patch_weight(w)
    ms = _time_launches(lambda: ops.patchify(img, wp, 14), iters=100, warm=20)
    P = (image // 14) ** 2
    nbytes = img.numel() * 2 + batch * P * 1024 * 2 + wp.numel() * 2
    gbps = nbytes / ms / 1e6
    return dict(kernel="big::patchify_strip_kernel (A tiles DMA'd from the pixels)", batch=batch, image=image, us=round(ms * 1e3, 2),
                algorithmic_bytes=nbytes, GBps=round(gbps, 1), peak_GBps=8000.0, frac_hbm=round(gbps / 8000.0, 4),
                mfma_tflops=round(2.0 * batch * P * 1024 * wp.shape[1] / ms / 1e9, 1),
                note="north_star target 0.60; a GEMM-shaped patchify is MFMA-time bound below ~40 % of HBM peak (DESIGN.md, patchify)")


def _cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _c1_state_dict(device, n_llama=32, n_clip=24, P=256, V=32011):
    """Random-init weights of ViT-L/14-224 + LLaMA-7B in the reference's state-dict layout, generated on the GPU (seconds, instead of
    minutes of host RNG: BASELINE.md section 3 "do not use HF random init") and copied to host memory as bf16 (13.9 GB)."""
    D, I, Dv, Iv = 4096, 11008, 1024, 4096
    g = torch.Generator(device=device).manual_seed(11)
    sd = {}

    def mat(name, *shape):
        sd[name] = (torch.randn(*shape, device=device, generator=g) * 0.02).to(torch.bfloat16).cpu()

    def one(name, n):
        sd[name] = torch.ones(n, dtype=torch.bfloat16)

    def zero(name, n):
        sd[name] = torch.zeros(n, dtype=torch.bfloat16)
    mat("model.embed_tokens.weight", V, D); mat("lm_head.weight", V, D); one("model.norm.weight", D)
    mat("vision_projector.weight", D, Dv); zero("vision_projector.bias", D)
    for l in range(n_llama):
        p = f"model.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            mat(p + f"self_attn.{n}.weight", D, D)
        mat(p + "mlp.gate_proj.weight", I, D); mat(p + "mlp.up_proj.weight", I, D); mat(p + "mlp.down_proj.weight", D, I)
        one(p + "input_layernorm.weight", D); one(p + "post_attention_layernorm.weight", D)
    ve = "vision_encoder."
    mat(ve + "embeddings.class_embedding", Dv); mat(ve + "embeddings.patch_embedding.weight", Dv, 3, 14, 14)
    mat(ve + "embeddings.position_embedding.weight", P + 1, Dv); one(ve + "pre_layrnorm.weight", Dv); zero(ve + "pre_layrnorm.bias", Dv)
    one(ve + "post_layernorm.weight", Dv); zero(ve + "post_layernorm.bias", Dv)       # (dead on this path: vision_hidden_layer = -2)
    for l in range(n_clip):
        p = f"{ve}encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            mat(p + f"self_attn.{n}.weight", Dv, Dv); zero(p + f"self_attn.{n}.bias", Dv)
        for n in ("layer_norm1", "layer_norm2"):
            one(p + n + ".weight", Dv); zero(p + n + ".bias", Dv)
        mat(p + "mlp.fc1.weight", Iv, Dv); zero(p + "mlp.fc1.bias", Iv); mat(p + "mlp.fc2.weight", Dv, Iv); zero(p + "mlp.fc2.bias", Dv)
    return sd


C1_VCFG = dict(hidden_size=1024, num_attention_heads=16, num_hidden_layers=24, intermediate_size=4096, image_size=224, patch_size=14,
               num_channels=3, layer_norm_eps=1e-5)


def c1_case(seed=5, prompt_tokens=32, V=32011):
    """The C1 inputs (BASELINE.json configs[0]: one 224x224 image + 32-token prompt, S = 291) and the oracle's config dict for
    ViT-L/14-224 + LLaMA-7B.  Host tensors from a CPU generator: the same on every box."""
    cfg = dict(hidden_size=4096, num_hidden_layers=32, num_attention_heads=32, intermediate_size=11008, vocab_size=V, rms_norm_eps=1e-6,
               rope_theta=10000.0, vision_hidden_layer=-2, projector_type="mlp", mm_token_ids=dict(MM), vision_config=dict(C1_VCFG))
    g = torch.Generator().manual_seed(seed)
    P = 256
    ids = torch.tensor([[1, MM["IMG_START"]] + [MM["IMG_PATCH"]] * P + [MM["IMG_END"]] + torch.randint(5, 32000, (prompt_tokens,), generator=g).tolist()])
    mask = torch.ones_like(ids)
    img = torch.randn(1, 3, 224, 224, generator=g).to(torch.bfloat16)
    return cfg, ids, mask, img


def c1_hip_model(sd, device, n_llama=32, dtype=torch.bfloat16):
    """The MI355X model holding the very weights of `sd` (the oracle's state dict, host 16-bit): same values on both sides of a parity check."""
    C = importlib.import_module("u-llava_amd.configuration")
    M = importlib.import_module("u-llava_amd.modeling_core")
    cfg = C.UllavaCoreConfig(vision_config=dict(image_size=224, patch_size=14), vision_hidden_layer=-2, projector_type="mlp",
                             projector_from_scratch=False, mm_token_ids=dict(MM), vocab_size=sd["lm_head.weight"].shape[0], num_hidden_layers=n_llama)
    model = M.UllavaCoreForCausalLM(cfg, device=device, dtype=dtype)
    model.load_state_dict(sd, strict=True)
    model.strict_checks = False
    return model


class F32View(dict):
    """The oracle's fp32 "truth" run on 16-bit-rounded weights without a second 28 GB copy: every read hands out an fp32 copy of ONE tensor."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()

    def get(self, k, default=None):
        return dict.__getitem__(self, k).float() if k in self else default

    def items(self):                                    # (oracle helpers that re-key a sub-tree iterate: they get fp32 copies too)
        return ((k, dict.__getitem__(self, k).float()) for k in self.keys())

    def values(self):
        return (dict.__getitem__(self, k).float() for k in self.keys())


def parity_stats(hip_logits, oracle_logits, truth_logits, k=4.0):
    """HIP logits vs the oracle's bf16 logits vs the oracle's fp32 logits on the same inputs and weights ([S, V] each).  `gated_exact`: at every
    position whose fp32 top-1 / top-2 gap exceeds k x the measured bf16 noise of a logit DIFFERENCE at that position -- sqrt(2) x the rms of
    (oracle_bf16 - fp32) over the vocabulary, i.e. the decision is k standard deviations away from flipping -- the three argmax token ids
    are EQUAL: north_star's "token ids bit-exact" wherever 16-bit rounding cannot legitimately flip the decision.  (The max over the 32011
    entries of a position is ~4.5 sigma of that noise and exceeds the MEDIAN top-2 gap of a random-init model: reported, not the gate.)"""
    h, o, t = (x.detach().float().cpu().reshape(-1, x.shape[-1]) for x in (hip_logits, oracle_logits, truth_logits))
    tmax = float(t.abs().max())
    sigma = (o - t).pow(2).mean(-1).sqrt() * 2.0 ** 0.5
    top2 = t.topk(2, dim=-1).values
    gap = top2[:, 0] - top2[:, 1]
    gated = gap > k * sigma
    ah, ao, at = h.argmax(-1), o.argmax(-1), t.argmax(-1)
    exact = (ah == ao) & (ao == at)
    return dict(positions=int(h.shape[0]), max_rel=round(float((h - o).abs().max()) / tmax, 6),
                hip_err_vs_fp32=round(float((h - t).abs().max()) / tmax, 6), oracle_err_vs_fp32=round(float((o - t).abs().max()) / tmax, 6),
                hip_rms_vs_fp32=round(float((h - t).pow(2).mean().sqrt()), 5), oracle_rms_vs_fp32=round(float((o - t).pow(2).mean().sqrt()), 5),
                argmax_agree=round(float((ah == ao).float().mean()), 4), argmax_agree_hip_fp32=round(float((ah == at).float().mean()), 4),
                argmax_agree_oracle_fp32=round(float((ao == at).float().mean()), 4),
                gate=f"fp32 top-1/top-2 gap > {k:g} x sqrt(2) x rms_v(oracle_16bit - fp32) at that position", positions_gated=int(gated.sum()),
                gated_exact=bool(exact[gated].all()), gated_mismatches=int((~exact[gated]).sum()),
                median_gap=round(float(gap.median()), 4), median_diff_sigma=round(float(sigma.median()), 5),
                max_abs_noise=round(float((o - t).abs().max()), 5))


def cpu_baseline(device, c4_image=336, c4_prompt=64):
    """the oracle (torch-CPU bf16 restatement of the reference path) on ONE whole C1 forward -- 224x224 image +
    32-token prompt, S = 291, random-init ViT-L/14 (23 of 24 layers feed the projector) + all 32 LLaMA-7B layers + lm_head -- 1 warm-up +
    3 timed forwards, median; thread count and CPU model stated.  The logits of that forward are then compared with the HIP path's on the
    same weights and inputs (`parity_vs_gpu`; the oracle as the CHECKER of the product, never the other way round), with the oracle's own
    fp32 run as the truth that sets the noise gate.  Secondary record: the same oracle at the C4 shape (the GPU headline's shape) from ONE and
    TWO CLIP / LLaMA layers, extrapolated linearly in layer count, with the 1- vs 2-layer check of that linearity."""
    from oracle import ullava_oracle as O
    t_build = time.perf_counter()
    sd = _c1_state_dict(device)
    t_build = time.perf_counter() - t_build
    # thread count: whatever runs TWO LLaMA layers of the oracle itself fastest (a bare matmul probe picked 128 threads on a 256-thread
    # EPYC where the whole forward then ran 2.5x slower than on 64: the oracle is many mid-sized ops, not one GEMM)
    probe_emb = torch.randn(1, 291, 4096).to(torch.bfloat16)
    probe_cfg = dict(hidden_size=4096, num_hidden_layers=2, num_attention_heads=32, rms_norm_eps=1e-6, rope_theta=10000.0)
    best_n, best_t, probe = 1, float("inf"), {}
    with torch.no_grad():
        for n in sorted({min(os.cpu_count(), c) for c in (8, 16, 32, 64, 128, os.cpu_count())}):
            torch.set_num_threads(n)
            O.llama_model(sd, probe_cfg, probe_emb)
            t0 = time.perf_counter()
            O.llama_model(sd, probe_cfg, probe_emb)
            dt = time.perf_counter() - t0
            probe[n] = round(dt * 1e3, 1)
            if dt < best_t:
                best_n, best_t = n, dt
    torch.set_num_threads(best_n)
    V, D = sd["lm_head.weight"].shape
    cfg, ids, mask, img = c1_case(V=V)
    vcfg = cfg["vision_config"]
    g = torch.Generator().manual_seed(6)

    def t(fn, n=3):
        fn()                                           # 1 warm-up
        ts = []
        for _ in range(n):                             # 3 timed, median
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts)
    with torch.no_grad():
        out = O.core_forward(sd, cfg, ids, mask, img)
        finite = bool(torch.isfinite(out["logits"].float()).all())
        o_logits = out["logits"][0]
        del out
        t_c1 = t(lambda: O.core_forward(sd, cfg, ids, mask, img))
        # parity of the product against this very forward: HIP model on the same weights + inputs, fp32 oracle run as the truth
        parity = None
        if device.type == "cuda":
          try:
            t0 = time.perf_counter()
            truth = O.core_forward(F32View(sd), cfg, ids, mask, img.float())["logits"][0]
            t_truth = time.perf_counter() - t0
            hip_model = c1_hip_model(sd, device)
            hip = hip_model.forward(input_ids=ids.to(device), attention_mask=mask.to(device), images=img.to(device)).logits[0]
            parity = parity_stats(hip, o_logits, truth)
            parity["fp32_truth_seconds"] = round(t_truth, 2)
            parity["what"] = "HIP forward (cuda) vs this oracle forward vs the oracle in fp32: same 7.05 B random-init weights, same C1 inputs, logits [291, 32011]"
            del hip_model, hip, truth
            torch.cuda.empty_cache()
          except Exception as e:                     # the comparison must never cost the bench line; a failure is REPORTED in its place
            parity = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        # secondary: C4 shape from 1 and 2 layers (the 336-px position table is a fresh random one: timing only)
        P4 = (c4_image // 14) ** 2
        S4 = 2 + P4 + 1 + c4_prompt
        sd4 = dict(sd)
        sd4["vision_encoder.embeddings.position_embedding.weight"] = (torch.randn(P4 + 1, 1024, generator=g) * 0.02).to(torch.bfloat16)
        img4 = torch.randn(1, 3, c4_image, c4_image, generator=g).to(torch.bfloat16)
        emb4 = torch.randn(1, S4, D, generator=g).to(torch.bfloat16)
        v4 = dict(vcfg, image_size=c4_image)
        tc = [t(lambda n=n: O.clip_vision_hidden_states(sd4, v4, img4, n_layers_to_run=n)) for n in (0, 1, 2)]
        tl = [t(lambda n=n: O.llama_model(sd4, dict(cfg, num_hidden_layers=n), emb4)) for n in (0, 1, 2)]
        hs = O.llama_model(sd4, dict(cfg, num_hidden_layers=0), emb4)[0][-1]
        t_head = t(lambda: torch.nn.functional.linear(hs, sd4["lm_head.weight"]))
    per_clip, per_llm = max(tc[2] - tc[0], 0.0) / 2, max(tl[2] - tl[0], 0.0) / 2
    c4_s = tc[0] + per_clip * 23 + tl[0] + per_llm * 32 + t_head
    lin = dict(clip_layer_s_1=round(tc[1] - tc[0], 4), clip_layer_s_2nd=round(tc[2] - tc[1], 4),
               llama_layer_s_1=round(tl[1] - tl[0], 4), llama_layer_s_2nd=round(tl[2] - tl[1], 4))
    return dict(value=round(1.0 / t_c1, 4), unit="images/sec", cores=best_n, kind="port", cpu_model=_cpu_model_name(),
                host_threads=os.cpu_count(), seconds_per_image=round(t_c1, 3), logits_finite=finite, parity_vs_gpu=parity,
                sample=f"oracle (torch-CPU bf16 restatement of the reference path), WHOLE C1 forward: 1 image 224x224 + 32-token prompt, S=291, "
                       f"ViT-L/14 (23 layers used) + projector + 32 LLaMA-7B layers + lm_head (V={V}), random-init weights, 1 warm-up + 3 timed "
                       f"forwards, median {t_c1:.2f} s; {best_n} of {os.cpu_count()} host threads (fastest on two LLaMA layers of the oracle: {probe} ms); "
                       f"weights generated on `{device.type}` and held in host memory, {t_build:.0f} s (not timed)",
                c4_shape_extrapolated=dict(value=round(1.0 / c4_s, 4), unit="images/sec", seconds_per_image=round(c4_s, 3),
                                           sample=f"same oracle, C4 shape ({c4_image}x{c4_image}, S={S4}): (2-layer - 0-layer)/2 per CLIP / LLaMA "
                                                  "layer x 23 / 32 + embeddings + lm_head", linearity_check=lin))


def _parse_cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def numa_topology():
    """{node: [cpus]} from sysfs, or None."""
    base = "/sys/devices/system/node"
    try:
        nodes = sorted(int(d[4:]) for d in os.listdir(base) if d.startswith("node") and d[4:].isdigit())
        topo = {n: _parse_cpulist(open(f"{base}/node{n}/cpulist").read()) for n in nodes}
        return {n: c for n, c in topo.items() if c} or None
    except OSError:
        return None


def gpu_numa_node(index):
    """NUMA node of GPU `index` (its PCI function's sysfs entry), or None."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        n = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        return n if n >= 0 else None
    except Exception:
        return None


def rank_cpu_set(rank, world, allowed, numa, gpu_nodes=None):
    """CPUs for local rank `rank` of `world`: disjoint, contiguous shares of the allowed set.  With a NUMA map, a rank stays inside one
    node: the node of its GPU when `gpu_nodes` (node per local rank) is known -- ranks whose GPUs hang off the same node split that node's
    CPUs -- else nodes dealt out in order when the rank count is a multiple of the node count.  Fewer CPUs than ranks: no pinning."""
    allowed = sorted(allowed)
    if world <= 1 or len(allowed) < world:
        return allowed
    aset = set(allowed)
    if numa:
        nodes = {n: [c for c in cpus if c in aset] for n, cpus in sorted(numa.items())}
        nodes = {n: c for n, c in nodes.items() if c}
        if gpu_nodes and all(g in nodes for g in gpu_nodes):
            peers = [r for r in range(world) if gpu_nodes[r] == gpu_nodes[rank]]
            cpus, k, j = nodes[gpu_nodes[rank]], len(peers), peers.index(rank)
            if len(cpus) >= k:
                return cpus[j * len(cpus) // k:(j + 1) * len(cpus) // k]
        elif nodes and world % len(nodes) == 0:
            per = world // len(nodes)
            cpus = nodes[sorted(nodes)[rank // per]]
            j = rank % per
            if len(cpus) >= per:
                return cpus[j * len(cpus) // per:(j + 1) * len(cpus) // per]
    return allowed[rank * len(allowed) // world:(rank + 1) * len(allowed) // world]


def pin_rank_to_cpus(local, world, use_gpu_topology):
    """One process per GPU, each on its own CPUs (with 8 Python launch threads on one host, migration and shared caches
    are what separates 8 GPUs from 8x one GPU).  Returns a small record for the JSON line, or None when nothing was pinned."""
    if world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        numa = numa_topology()
        gpu_nodes = None
        if use_gpu_topology and torch.cuda.is_available() and torch.cuda.device_count() >= world:
            gn = [gpu_numa_node(i) for i in range(world)]
            gpu_nodes = gn if all(g is not None for g in gn) else None
        cpus = rank_cpu_set(local, world, allowed, numa, gpu_nodes)
        if not cpus or len(cpus) == len(allowed):
            return None
        os.sched_setaffinity(0, cpus)
        torch.set_num_threads(max(1, min(len(cpus), 16)))
        return {"cpus_per_rank": len(cpus), "rank0_cpus": f"{cpus[0]}-{cpus[-1]}", "numa_nodes": len(numa) if numa else None,
                "gpu_numa_aware": gpu_nodes is not None}
    except OSError:
        return None


def check_finite(out):
    """The timed steps must have computed something: every floating-point tensor the last step returned (logits; for RES also the
    masks and boxes; for training the loss) is finite.  Raises otherwise -- a bench line is only printed for a sane forward."""
    if out is None:
        return
    todo, n = [out], 0
    while todo:
        o = todo.pop()
        if isinstance(o, torch.Tensor):
            if o.is_floating_point() and o.numel():
                if not bool(torch.isfinite(o.float() if o.dtype != torch.float32 else o).all()):
                    raise SystemExit("bench.py: non-finite values in the outputs of the last timed step")
                n += 1
        elif isinstance(o, dict):
            todo += [v for k, v in o.items() if k not in ("past_key_values", "hidden_states", "gt_masks", "gt_boxes")]
        elif isinstance(o, (list, tuple)):
            todo += list(o)
    if n == 0:
        raise SystemExit("bench.py: the last timed step returned no tensor to check")


def timed_steps(step, steps, warmup, dist, batch, device):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX elapsed over ranks and
    SUM of images over ranks through the path's only collective (u-llava_amd.dist.global_rate)."""
    D = importlib.import_module("u-llava_amd.dist")
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    for _ in range(warmup):
        step()
    sync()
    if dist:
        dist.barrier()
    sync()
    last = None
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    sync()
    busy = time.perf_counter() - t0                                           # this rank's own K steps, before it waits for the others
    if dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    check_finite(last)                                                        # outside the timed region: the last step's outputs
    host_ms = None
    if device.type == "cuda":
        # host time to ENQUEUE one step (idle GPU, no synchronize inside): what a rank's Python thread costs per step -- eight of them share a
        # host at N = 8.  `coarse` = the shipped path (one C call per layer stack, csrc/layers.hip), `per_op` = one ctypes call per launch.
        ops = importlib.import_module("u-llava_amd.ops")
        host_ms = {}
        for label, ctx in (("coarse", ops.per_op_layers(False)), ("per_op", ops.per_op_layers(True))):
            best = None
            with ctx:
                for _ in range(3):
                    sync()
                    h0 = time.perf_counter()
                    step()
                    h = time.perf_counter() - h0
                    best = h if best is None else min(best, h)
            host_ms[label] = round(best * 1e3, 3)
        sync()
    agg_dev = torch.device("cpu") if (dist is not None and dist.get_backend() == "gloo") else device
    rate, total, t_max = D.global_rate(float(batch * steps), elapsed, device=agg_dev)    # (images/s whole job, images, max elapsed)
    lo, hi = D.elapsed_spread(busy, device=agg_dev)                           # fastest / slowest rank's own time: where skew comes from
    spread = {"min": round(lo / steps * 1e3, 3), "max": round(hi / steps * 1e3, 3)}
    if host_ms is not None:
        spread["host_enqueue_ms"] = host_ms
    return rate, total, t_max, spread


TRAIN_CONFIGS = {"full": 16, "lora": 32, "qv": 8}          # per-GPU batch (configs/train/ullava.yaml:148, ullava_lora.yaml:148)


def workload_step(name, dev, rank, batch_override=None, train_config="full", model=None):
    """Build model + inputs of a workload; returns (step callable, per-GPU batch, S, cfg, description, flops per image, model).
    model: an already built UllavaCoreForCausalLM of the workload's image size to run it on (C2 / C5 after RES), else one is built."""
    image_size, prompt, batch, desc = WORKLOADS[name]
    if name == "train":
        batch = TRAIN_CONFIGS[train_config]
    batch = batch_override or batch
    res, video = name == "res", name == "c5"
    if model is None:
        model, cfg = build_model(image_size, dev, seed=rank, with_sam=res)
    else:
        cfg = model.config
        assert not res and cfg.vision_config.image_size == image_size
    vis, ids, mask = make_inputs(cfg, batch, prompt, dev, rank, ragged=(name == "c2"), video=video)
    S = ids.shape[1]
    P = (image_size // 14) ** 2
    flops_img = llama_flops(S, cfg.vocab_size) + (8 if video else 1) * clip_flops(P) + 2 * ((8 + P) if video else (P + 1)) * 1024 * 4096
    if res:
        # three rounds per sample, each ending "... [SEG] ... [LOC]" (RefCOCO-shaped: valid region 768x1024 -> original 480x640)
        for r in range(3):
            ids[:, S - 10 - 40 * r] = SEG
            ids[:, S - 5 - 40 * r] = LOC
        g = torch.Generator(device="cuda").manual_seed(2000 + rank)
        images_sam = torch.randn(batch, 3, 1024, 1024, device=dev, generator=g).to(DTYPE)
        sizes, resizes = [(480, 640)] * batch, [(768, 1024)] * batch
        flops_img += 5.96e12 + 3 * 3.61e9          # SAM ViT-H encoder + 3 mask-decoder passes

        def step():
            return model.forward(images_sam=images_sam, images=vis, input_ids=ids, labels=None, attention_mask=mask,
                                 mask_list=[None] * batch, size_list=sizes, resize_list=resizes, bbox_list=[None] * batch, inference=True)
    elif name == "train":
        Dm = importlib.import_module("u-llava_amd.dist")
        labels = ids.clone()
        labels[:, :P + 3] = -100
        trainable = []
        if train_config == "lora":                               # train_ullava.py:217-237 + :248-261
            model.add_lora(8, 16.0, 0.05, ("q_proj", "v_proj"))
            model.train()
        for n, p_ in model.named_parameters():
            if train_config == "lora":
                p_.requires_grad = (n.startswith("lm_head") or "embed_tokens" in n or ".lora_" in n)
            elif train_config == "full":                         # train_ullava.py:239-245: llm.model, lm_head, vision_projector
                p_.requires_grad = n.startswith("model.") or n.startswith("lm_head") or n.startswith("vision_projector")
            else:
                p_.requires_grad = (n.startswith("lm_head") or "embed_tokens" in n or n.startswith("vision_projector") or
                                    (n.startswith("model.") and (".q_proj." in n or ".v_proj." in n)))
            if p_.requires_grad:
                trainable.append(p_)
        desc += f", --train-config {train_config} ({sum(p_.numel() for p_ in trainable) / 1e9:.2f} G trainable parameters), batch {batch}/GPU"
        flops_img *= 3 if train_config == "full" else 2.0        # forward + dX (+ dW for every Linear with the full set)
        # the optimizer step is part of a training step: ZeRO-2 AdamW (configs/deepspeed/bf16_zero2.json; lr / weight decay of
        # configs/train/ullava.yaml:141,156 resp. ullava_lora.yaml), gradient reduce-scatter and parameter all-gather inside it
        O_ = importlib.import_module("u-llava_amd.optim")
        opt = O_.ShardedAdamW(trainable, lr=2e-4 if train_config == "lora" else 2e-5, weight_decay=0.0, max_grad_norm=1.0)
        desc += f", AdamW with ZeRO-2 sharded fp32 state ({opt.state_bytes() / 2**30:.1f} GiB on this rank)"

        def step():
            opt.zero_grad()
            out = model.forward(input_ids=ids, attention_mask=mask, images=vis, labels=labels)
            out.loss.backward()
            opt.step()
            return out.loss
    elif video:
        def step():
            return model.forward(input_ids=ids, attention_mask=mask, videos=vis)
    else:
        def step():
            return model.forward(input_ids=ids, attention_mask=mask, images=vis)
    return step, batch, S, cfg, desc, flops_img, model


def self_launch(a):
    """`python bench.py --gpus N` outside a distributed launch: start N ranks (one per GPU) under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvp(cmd[0], cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="fp16 / fp32 = the reference's --dtype options (side checks; default bf16 like inference_ullava.py:165).  fp32 is the plain-kernel "
                         "correctness build (csrc/f32.hip, ~1/50 of the bf16 speed): use it with --batch 1 or 2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-res", action="store_true", help="skip the C3 RES sub-record of the default (c4) run")
    ap.add_argument("--no-extra", action="store_true", help="skip the C2 / C5 sub-records of the default (c4) run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU ranks (only with --stub)")
    ap.add_argument("--stub", action="store_true", help="replace the model step by a fixed host-side delay (tests the N-rank protocol without a GPU)")
    ap.add_argument("--init-pg", action="store_true", help="initialise the process group (RCCL) even at --gpus 1, so that the barrier and the "
                    "two scalar all-reduces of the aggregation run through RCCL on a single-GPU box")
    ap.add_argument("--no-pin", action="store_true", help="do not pin ranks to CPU sets")
    ap.add_argument("--share-gpu", action="store_true", help="tests only: rank r computes on cuda:(r %% device_count); needs --backend gloo")
    ap.add_argument("--train-config", default="full", choices=sorted(TRAIN_CONFIGS), help="--workload train: which trainable set / batch")
    a = ap.parse_args()
    global DTYPE
    DTYPE = {"fp16": torch.float16, "fp32": torch.float32}.get(a.dtype, torch.bfloat16)
    if a.dtype != "bf16":                                         # the roofline / parity legs are bf16 records: not part of a side check
        a.no_roofline = a.no_cpu_baseline = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)                                           # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    if a.share_gpu and a.backend != "gloo":
        raise SystemExit("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    if a.stub:
        dev = torch.device("cpu")
    else:
        # --share-gpu (tests only, with --backend gloo): the ranks run their real HIP steps on the devices there are (a 1-GPU box: all on
        # cuda:0; RCCL refuses two ranks on one device, gloo carries the barrier and the scalar aggregation instead)
        gpu = local % torch.cuda.device_count() if a.share_gpu else local
        torch.cuda.set_device(gpu)
        dev = torch.device("cuda", gpu)
    affinity = None if a.no_pin else pin_rank_to_cpus(local, world, use_gpu_topology=not a.stub)
    dist = None
    if world > 1 or a.init_pg:
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                      # --init-pg outside a launcher
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist_.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist_.init_process_group("gloo")
        dist = dist_

    pg = None
    if dist is not None:
        # what the collective layer itself reports: a SCALE record must show that RCCL saw N ranks, not what the launcher was asked for
        pg = {"backend": dist.get_backend(), "rccl_world_size": dist.get_world_size(), "rank": dist.get_rank()}

    def close_pg():
        """barrier + destroy: every timed region is over.  Rank 0's single-rank probes run AFTER this, on an otherwise idle node (in an
        N > 1 run ranks 1..N-1 have left; before round 5 they sat in a barrier while rank 0 probed between the C4 and RES timings)."""
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if a.stub:
        image_size, prompt, batch, desc = WORKLOADS[a.workload]
        batch = a.batch or batch

        def step():
            time.sleep(0.01 * (1 + rank))                        # rank r is (r+1)x slower: the MAX over ranks must show
            return torch.zeros(1)
        value, total_images, elapsed, spread = timed_steps(step, a.steps, a.warmup, dist, batch, dev)
        close_pg()
        if rank == 0:
            import torch.distributed as dist_chk
            probes = {"order": "after destroy_process_group", "process_group_alive": bool(dist_chk.is_initialized())}
            time.sleep(0.02)                                     # stands for the roofline / patchify / cpu_baseline probes
            print(json.dumps({"metric": METRIC, "value": round(value, 3), "unit": "images/sec (whole job)", "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "stub", "total_images": total_images,
                              "cpu_affinity": affinity, "process_group": pg, "per_rank_ms_per_step": spread, "probes": probes,
                              "config": {"workload": "stub step (host-side delay), " + desc, "per_gpu_batch": batch,
                                         "global_batch": batch * world, "parallelism": f"dp{world}"}}), flush=True)
        return

    def sub_record(svalue, ssteps, selapsed, sspread, sbatch, sS, sdesc, sflops, **cfg_extra):
        return {"value": round(svalue, 3), "unit": "images/sec (whole job)", "steps": ssteps, "ms_per_step": round(selapsed / ssteps * 1e3, 3),
                "host_enqueue_ms_per_step": sspread.pop("host_enqueue_ms", None),
                "per_rank_ms_per_step": sspread, "images_per_sec_per_gpu": round(svalue / world, 3),
                "config": dict({"workload": sdesc, "per_gpu_batch": sbatch, "global_batch": sbatch * world, "seq_len": sS}, **cfg_extra),
                "model_tflops_per_image": round(sflops / 1e12, 3),
                "frac_of_bf16_peak_end_to_end": round(sflops * svalue / world / 1e12 / PEAK_BF16_TFLOPS, 4)}

    with (torch.enable_grad() if a.workload == "train" else torch.no_grad()):
        step, batch, S, cfg, desc, flops_img, model = workload_step(a.workload, dev, rank, a.batch, a.train_config)
        value, total_images, elapsed, spread = timed_steps(step, a.steps, a.warmup, dist, batch, dev)
        image_size, prompt = WORKLOADS[a.workload][:2]
        res_rec = None
        extra = {}
        if a.workload == "c4" and not a.no_res:
            # second half of BASELINE.json's metric: the full RES forward (C3), same timing protocol, same process
            del step, model
            torch.cuda.empty_cache()
            rstep, rbatch, rS, rcfg, rdesc, rflops, rmodel = workload_step("res", dev, rank)
            rsteps = max(3, min(a.steps, 10))
            rvalue, rimgs, relapsed, rspread = timed_steps(rstep, rsteps, max(1, min(a.warmup, 2)), dist, rbatch, dev)
            res_rec = sub_record(rvalue, rsteps, relapsed, rspread, rbatch, rS, rdesc, rflops,
                                 sam_input="1024x1024 (valid 768x1024 -> 480x640 masks)", prompts_per_image=3)
            if not a.no_extra:
                # the other two BASELINE.json configurations, timed at their own sizes on the RES model's language model (the very
                # ViT-L/14-224 + LLaMA-7B of C2 / C5; SAM stays idle): C2 = configs[1] (VQA, batch 16, ragged prompts), C5 = configs[4] (video, 8 clips / GPU)
                del rstep
                for wname in ("c2", "c5"):
                    wstep, wbatch, wS, _, wdesc, wflops, _ = workload_step(wname, dev, rank, model=rmodel.llm)
                    wsteps = max(3, min(a.steps, 5))
                    wvalue, _, welapsed, wspread = timed_steps(wstep, wsteps, 2, dist, wbatch, dev)
                    extra[wname] = sub_record(wvalue, wsteps, welapsed, wspread, wbatch, wS, wdesc, wflops)
                    del wstep
            del rmodel
            torch.cuda.empty_cache()
        close_pg()
        # ---- single-rank probes: after the process group is gone (see close_pg) -------------------------------------------------------
        roof = patch_rec = res_roof = None
        if rank == 0 and not a.no_roofline and a.workload != "train":
            roof = gemm_roofline(cfg, batch * S, dev)
            if a.workload == "c4":
                patch_rec = patchify_record(dev)
                if not a.no_res:
                    res_roof = sam_gemm_roofline(dev)

    if rank == 0:
        line = {"metric": METRIC, "value": round(value, 3), "unit": "images/sec (whole job, all GPUs)", "cpu_affinity": affinity,
                "outputs_finite": True, "process_group": pg, "host_enqueue_ms_per_step": spread.pop("host_enqueue_ms", None),
                "per_rank_ms_per_step": spread,
                "probes": {"order": "after destroy_process_group", "ranks_running": 1},
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                "config": {"workload": desc, "per_gpu_batch": batch, "global_batch": batch * world, "seq_len": S, "image": image_size,
                           "parallelism": f"dp{world}", "weights": "random-init N(0,0.02), ViT-L/14 + LLaMA-7B (V=32011) shapes"},
                "images_per_sec_per_gpu": round(value / world, 3),
                "model_tflops_per_image": round(flops_img / 1e12, 3),
                "model_tflops_per_sec_per_gpu": round(flops_img * value / world / 1e12, 1),
                "frac_of_bf16_peak_end_to_end": round(flops_img * value / world / 1e12 / PEAK_BF16_TFLOPS, 4)}
        also = {}
        if res_rec is not None:
            if res_roof is not None:
                res_rec["roofline"] = res_roof
            line["res"] = res_rec
            also["res"] = {"images_per_sec": res_rec["value"], "ms_per_step": res_rec["ms_per_step"], "per_gpu_batch": res_rec["config"]["per_gpu_batch"]}
        for k, v in extra.items():
            also[k] = {"images_per_sec": v["value"], "ms_per_step": v["ms_per_step"], "per_gpu_batch": v["config"]["per_gpu_batch"]}
        if extra:
            line["extra"] = extra
        if also:
            # the other BASELINE.json configurations timed in this run, in brief (full records: `res`, `extra.c2`, `extra.c5`)
            line["config"]["also_timed"] = also
        if patch_rec is not None:
            line["patchify"] = patch_rec
        if roof is not None:
            line["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(dev)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
