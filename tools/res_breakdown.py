#!/usr/bin/env python
"""Where a RES (C3) step goes: SAM ViT-H encoder alone, CLIP + LLaMA alone, heads + mask decoder alone, the whole step with the SAM
encoder on the side stream and on the main stream.  HIP-event timing, 3 warm-up + 10 timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def t(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


dev = torch.device("cuda", 0)
with torch.no_grad():
    step, batch, S, cfg, desc, flops, model = bench.workload_step("res", dev, 0)
    vis, ids, mask = bench.make_inputs(cfg, batch, 120, dev, 0)
    for r in range(3):
        ids[:, S - 10 - 40 * r] = bench.SEG
        ids[:, S - 5 - 40 * r] = bench.LOC
    g = torch.Generator(device="cuda").manual_seed(2000)
    images_sam = torch.randn(batch, 3, 1024, 1024, device=dev, generator=g).to(torch.bfloat16)
    print(f"whole step, SAM encoder on the side stream : {t(step):8.2f} ms")
    model.overlap_sam_encoder = False
    print(f"whole step, everything on one stream        : {t(step):8.2f} ms")
    model.overlap_sam_encoder = True
    print(f"SAM ViT-H encoder alone (B=8)               : {t(lambda: model._visual_embs_tm(images_sam)):8.2f} ms")
    print(f"CLIP + LLaMA + lm_head alone (B=8, S={S})   : {t(lambda: model.llm.forward(images=vis, attention_mask=mask, input_ids=ids, output_hidden_states=True)):8.2f} ms")
    print(f"CLIP tower alone                            : {t(lambda: model.llm._clip_hidden(vis)):8.2f} ms")
    emb = model._visual_embs_tm(images_sam)
    out = model.llm.forward(images=vis, attention_mask=mask, input_ids=ids, output_hidden_states=True)
    last = out.hidden_states[-1]
    pad = torch.zeros((batch, 1), dtype=torch.bool, device=dev)
    seg_mask = torch.cat([ids[:, 1:] == bench.SEG, pad], dim=1)
    loc_mask = torch.cat([ids[:, 1:] == bench.LOC, pad], dim=1)

    def heads():
        pe = model._select(last, seg_mask, model.seg_projector)
        pl = model._select(last, loc_mask, model.det_projector)
        pm = model._decode(emb, pe, [(768, 1024)] * batch, [(480, 640)] * batch)
        pb = [model._run_mlp(model.det_decoder, e) for e in pl]
        return pm, pb
    print(f"gather + projectors + mask decoder + postprocess + boxes: {t(heads):8.2f} ms")
