"""Stage-by-stage comparison of the HIP mask decoder against the oracle (run on the GPU box: oracle on the host CPU)."""
import importlib, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pkg, load_fixture, fixture_sd
from oracle import ullava_oracle as O
dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
name = "g7_sam_decoder_fp16.pt" if dt == torch.float16 else "g7_sam_decoder_bf16.pt"
fx = load_fixture(name)
C, S = pkg("configuration"), pkg("sam")
cfg = C.SamConfig(depth=0)
holder = S.build_sam_holder(cfg, device="cuda:0", dtype=dt)
sdv = {k[len("visual_model."):]: v for k, v in fixture_sd(fx, dt).items()}
holder.load_state_dict(sdv, strict=False)
eng = S.SamEngine(holder, cfg)
sd = fixture_sd(fx, dt)
g = torch.Generator().manual_seed(fx["image_embedding_seed"])
emb = torch.randn(1, 256, 64, 64, generator=g).to(dt)
emb_tm = emb[0].permute(1, 2, 0).reshape(4096, 256).contiguous().cuda()
case = fx["cases"][0]
th, to = {}, {}
masks, iou = eng.decode(emb_tm, case["text_embeds"][:, 0].cuda(), trace=th)
sp, de = O.prompt_encoder_text(sd, case["text_embeds"], (64, 64))
lr, oiou = O.mask_decoder(sd, emb, O.dense_pe(sd, (64, 64)), sp.to(dt), de, False, trace=to)
st = case["low_res_stride"]
print("oracle-on-this-host == fixture:", torch.equal(lr[:, :, ::st, ::st], case["low_res_masks"]),
      float((lr[:, :, ::st, ::st].float() - case["low_res_masks"].float()).abs().max()) / case["low_res_max"])
ref = torch.load(os.path.join(ROOT, "tools", "debug", f"g7_trace_{str(dt).split('.')[-1]}.pt"), weights_only=True)
print("---- HIP vs REFERENCE trace (build container CPU)   |   box-oracle vs REFERENCE trace")
for k in ref:
    if k not in th and k != "masks":
        continue
    a = (masks[:, 0:1] if k == "masks" else th[k]).float().cpu().reshape(ref[k].shape)
    b = ref[k].float(); c = (lr if k == "masks" else to[k]).float()
    d, d2 = (a - b).abs(), (c - b).abs()
    print(f"{k:12s} HIP: max {float(d.max() / b.abs().max()):.2e} frac {float((d > 0).float().mean()):.3f}   box-oracle: max {float(d2.max() / b.abs().max()):.2e} frac {float((d2 > 0).float().mean()):.3f}")
print("---- HIP vs box oracle")
for k in to:
    if k not in th:
        continue
    a, b = th[k].float().cpu().reshape(to[k].shape), to[k].float()
    d = (a - b).abs()
    print(f"{k:12s} max|d|/max {float(d.max() / b.abs().max()):.2e}  mean|d|/mean|b| {float(d.mean() / b.abs().mean()):.2e}  frac differing {float((d > 0).float().mean()):.3f}")
a, b = masks[:, 0:1].float().cpu(), lr.float()
d = (a - b).abs()
print(f"masks        max|d|/max {float(d.max() / b.abs().max()):.2e}  mean {float(d.mean() / b.abs().mean()):.2e} frac differing {float((d > 0).float().mean()):.3f}")

if "l0.t2i.v" in th:
    import torch.nn.functional as F
    p_ = "visual_model.mask_decoder.transformer.layers.0.cross_attn_token_to_image."
    w = sd["visual_model.prompt_encoder.no_mask_embed.weight"]
    keys_ref = (emb + w.reshape(1, -1, 1, 1)).flatten(2).permute(0, 2, 1)[0].contiguous()
    v_host = F.linear(keys_ref, sd[p_ + "v_proj.weight"], sd[p_ + "v_proj.bias"])
    print("ref trace v == host recompute:", float((ref["l0.t2i.v"][0] != v_host).float().mean()))
    print("HIP trace v vs host recompute:", float((th["l0.t2i.v"].cpu() != v_host).float().mean()))
    a = holder.mask_decoder.transformer.layers[0].cross_attn_token_to_image
    print("weights equal:", torch.equal(a.v_proj.weight.cpu(), sd[p_ + "v_proj.weight"]), torch.equal(a.v_proj.bias.cpu(), sd[p_ + "v_proj.bias"]))
    ops = pkg("ops")
    v2 = ops.linear(keys_ref.cuda(), a.v_proj.weight, a.v_proj.bias)
    print("HIP linear now vs host:", float((v2.cpu() != v_host).float().mean()), " vs HIP trace:", float((v2 != th["l0.t2i.v"]).float().mean()))
