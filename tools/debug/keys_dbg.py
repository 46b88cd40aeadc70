import importlib, sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pkg, load_fixture, fixture_sd
ops = pkg("ops")
dt = torch.bfloat16
fx = load_fixture("g7_sam_decoder_bf16.pt"); sd = fixture_sd(fx, dt)
g = torch.Generator().manual_seed(fx["image_embedding_seed"])
emb = torch.randn(1, 256, 64, 64, generator=g).to(dt)
w = sd["visual_model.prompt_encoder.no_mask_embed.weight"]
ref = (emb + w.reshape(1, -1, 1, 1)).flatten(2).permute(0, 2, 1)[0]
emb_tm = emb[0].permute(1, 2, 0).reshape(4096, 256).contiguous().cuda()
got = ops.add_rows(emb_tm, w.cuda())
print("add_rows(b_rows=1):", float((got.cpu() != ref).float().mean()), w.shape, w.is_contiguous())
p = "visual_model.mask_decoder.transformer.layers.0.cross_attn_token_to_image."
import torch.nn.functional as F
v_ref = F.linear(ref, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
for name, x in (("from add_rows", got), ("host keys", ref.cuda())):
    v = ops.linear(x, sd[p + "v_proj.weight"].cuda(), sd[p + "v_proj.bias"].cuda())
    print("v_proj", name, float((v.cpu() != v_ref).float().mean()))
x2 = got.unsqueeze(0).expand(1, -1, -1).contiguous().view(4096, 256)
v = ops.linear(x2, sd[p + "v_proj.weight"].cuda(), sd[p + "v_proj.bias"].cuda())
print("v_proj expand path", float((v.cpu() != v_ref).float().mean()), x2.data_ptr() == got.data_ptr())
