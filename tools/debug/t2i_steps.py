import importlib, sys, os, math, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pkg, load_fixture, fixture_sd
from oracle import ullava_oracle as O
ops = pkg("ops")
dt = torch.bfloat16
fx = load_fixture("g7_sam_decoder_bf16.pt"); sd = fixture_sd(fx, dt)
ref = torch.load(os.path.join(ROOT, "tools", "debug", "g7_trace_bfloat16.pt"), weights_only=True)
g = torch.Generator().manual_seed(fx["image_embedding_seed"])
emb = torch.randn(1, 256, 64, 64, generator=g).to(dt)
case = fx["cases"][0]
pfx = "visual_model.mask_decoder."
sp, de = O.prompt_encoder_text(sd, case["text_embeds"], (64, 64))
out_tok = torch.cat([sd[pfx + "iou_token.weight"], sd[pfx + "mask_tokens.weight"]], 0).unsqueeze(0)
tokens = torch.cat((out_tok, sp.to(dt)), dim=1)                     # [1, 6, 256]
keys = (emb + de).flatten(2).permute(0, 2, 1).contiguous()           # [1, 4096, 256]
pe = O.dense_pe(sd, (64, 64)).flatten(2).permute(0, 2, 1).contiguous()
queries = ref["l0.norm1"]
p = pfx + "transformer.layers.0.cross_attn_token_to_image."
q_in, k_in = queries + tokens, keys + pe
def frac(a, b): a, b = a.float().cpu().reshape(b.shape), b.float(); return f"frac {float((a != b).float().mean()):.4f} max {float((a-b).abs().max()/b.abs().max()):.2e}"
D = "cuda:0"
W = lambda n: sd[p + n].to(D)
# host-side references
q_r, k_r, v_r = O.linear(q_in, sd, p + "q_proj"), O.linear(k_in, sd, p + "k_proj"), O.linear(keys, sd, p + "v_proj")
print("k_in add_rows:", frac(ops.add_rows(keys[0].to(D), pe[0].to(D)), k_in[0]))
q_h = ops.linear(q_in[0].to(D), W("q_proj.weight"), W("q_proj.bias"))
k_h = ops.linear(k_in[0].to(D), W("k_proj.weight"), W("k_proj.bias"))
v_h = ops.linear(keys[0].to(D), W("v_proj.weight"), W("v_proj.bias"))
print("q_proj:", frac(q_h, q_r[0]), "| k_proj:", frac(k_h, k_r[0]), "| v_proj:", frac(v_h, v_r[0]))
def sep(t): b, n, c = t.shape; return t.reshape(b, n, 8, c // 8).transpose(1, 2)
qs, ks, vs = sep(q_r), sep(k_r), sep(v_r)
a = qs @ ks.permute(0, 1, 3, 2); a = a / 4.0; a = torch.softmax(a, dim=-1); o = (a @ vs)
o_r = o.transpose(1, 2).reshape(1, 6, 128)
# HIP attention on the REFERENCE q/k/v
Sq, Sk, Di, hd = 6, 4096, 128, 16
vt = ops.transpose_v(v_r[0].to(D), Sk * Di, Di, 1, Sk, 8, hd)
att = torch.empty(Sq, Di, device=D, dtype=dt)
ops.attention(q_r[0].to(D), k_r[0].to(D), vt, att, 1, 8, Sq, Sk, hd, (Sq * Di, hd, Di), (Sk * Di, hd, Di), (Sq * Di, hd, Di), None, causal=False, scale_mode=2, scale=4.0)
print("attention (ref q,k,v):", frac(att, o_r[0]))
o64 = (torch.softmax((qs.double() @ ks.double().permute(0,1,3,2)).to(dt).double()/4.0, -1).to(dt).double() @ vs.double()).to(dt).transpose(1,2).reshape(1,6,128)
print("   torch bf16 attention vs fp64-accumulated same-rounding-points:", frac(o_r[0], o64[0]), "| HIP vs that:", frac(att, o64[0]))
out_r = O.linear(o_r, sd, p + "out_proj")
out_h = ops.linear(o_r[0].to(D), W("out_proj.weight"), W("out_proj.bias"))
print("out_proj (ref input):", frac(out_h, out_r[0]), " ref t2i trace equal:", torch.equal(out_r, ref["l0.t2i"]))
