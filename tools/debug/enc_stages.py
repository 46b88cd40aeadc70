"""SAM encoder blocks (G9): HIP and the oracle run on THIS host's CPU, both against the committed reference trace."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pkg, load_fixture, fixture_sd
from oracle import ullava_oracle as O
C, S = pkg("configuration"), pkg("sam")
fx = load_fixture("g9_sam_blocks_bf16.pt")
cfg = C.SamConfig(depth=2, global_attn_indexes=[1])
holder = S.build_sam_holder(cfg, device="cuda:0")
sd = fixture_sd(fx, torch.bfloat16)
holder.load_state_dict({k[len("visual_model."):]: v for k, v in sd.items()}, strict=False)
eng = S.SamEngine(holder, cfg)
g = torch.Generator().manual_seed(fx["image_seed"])
img = torch.randn(1, 3, 1024, 1024, generator=g).to(torch.bfloat16)
tr, to = {}, {}
eng.encode(img.cuda(), trace=tr)
torch.set_num_threads(32)
O.sam_image_encoder(sd, fx["cfg"], img, trace=to)
for k, ref in fx["trace"].items():
    a, b = tr[k].cpu()[:, ::4, ::4, ::8], to[k][:, ::4, ::4, ::8]
    print(f"{k:12s} HIP vs reference: {float((a != ref).float().mean()):.5f}   oracle-on-this-host vs reference: {float((b != ref).float().mean()):.5f}   HIP vs host-oracle {float((a != b).float().mean()):.5f}")
