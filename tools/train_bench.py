#!/usr/bin/env python
"""f4 at full width: one training step (forward with labels + loss.backward()) of UllavaCoreForCausalLM, ViT-L/14-224 + LLaMA-7B,
the reference's stage-2 trainable set (train_ullava.py:229-261): lm_head, embed_tokens, vision projector and the q_proj / v_proj
weights (what LoRA adapts; here their full gradients).  Times the step and lists the backward kernels' share."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = "cuda:0"
B = int(os.environ.get("BATCH", 16))
model, cfg = bench.build_model(224, dev, seed=0, with_sam=False)
vis, ids, mask = bench.make_inputs(cfg, B, 64, dev, 0)
labels = ids.clone(); labels[:, :259] = -100
CFG = os.environ.get("TRAIN_CONFIG", "full")          # full (train_ullava.py:239-245) | qv (round 2's set)
for n, p in model.named_parameters():
    if CFG == "full":
        p.requires_grad = n.startswith("model.") or n.startswith("lm_head") or n.startswith("vision_projector")
    else:
        p.requires_grad = (n.startswith("lm_head") or "embed_tokens" in n or n.startswith("vision_projector") or ".q_proj." in n or ".v_proj." in n)
ntrain = sum(p.numel() for p in model.parameters() if p.requires_grad)
print(f"trainable parameters: {ntrain / 1e6:.1f} M; batch {B}, S = {ids.shape[1]}")
def step():
    for p in model.parameters():
        p.grad = None
    out = model(input_ids=ids, attention_mask=mask, images=vis, labels=labels)
    out.loss.backward()
    return out.loss
for i in range(2):
    l = step(); torch.cuda.synchronize()
print("loss", float(l), " peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30))
t0 = time.time(); n = 3
for i in range(n): step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
with torch.no_grad():
    for i in range(2): model(input_ids=ids, attention_mask=mask, images=vis, labels=labels)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n): model(input_ids=ids, attention_mask=mask, images=vis, labels=labels)
    torch.cuda.synchronize(); df = (time.time() - t0) / n
S = ids.shape[1]
fl = bench.llama_flops(S, cfg.vocab_size) + bench.clip_flops(256)
print(f"training step {dt * 1e3:.1f} ms = {B / dt:.1f} samples/s; inference forward {df * 1e3:.1f} ms; backward+graph overhead = {dt / df:.2f}x forward; "
      f"model flops fwd {fl * B / 1e12:.1f} TF -> {3 * fl * B / dt / 1e12:.0f} TF/s if backward = 2x forward")
g = [p.grad for n_, p in model.named_parameters() if p.requires_grad]
print("gradients present:", sum(x is not None for x in g), "of", len(g), "; finite:", all(bool(torch.isfinite(x).all()) for x in g if x is not None))
