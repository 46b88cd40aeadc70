"""(build container) full stage trace of the oracle (== reference, bit-exact here) for G7 case n=1 -> tools/debug/g7_trace_<dt>.pt"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_fixture, fixture_sd
from oracle import ullava_oracle as O
torch.set_num_threads(8)
for dt, name in ((torch.float16, "g7_sam_decoder_fp16.pt"), (torch.bfloat16, "g7_sam_decoder_bf16.pt")):
    fx = load_fixture(name); sd = fixture_sd(fx, dt)
    g = torch.Generator().manual_seed(fx["image_embedding_seed"])
    emb = torch.randn(1, 256, 64, 64, generator=g).to(dt)
    case = fx["cases"][0]
    to = {}
    sp, de = O.prompt_encoder_text(sd, case["text_embeds"], (64, 64))
    lr, oiou = O.mask_decoder(sd, emb, O.dense_pe(sd, (64, 64)), sp.to(dt), de, False, trace=to)
    assert torch.equal(lr, case["low_res_masks"])
    to["masks"] = lr
    torch.save({k: v.contiguous().clone() for k, v in to.items()}, os.path.join(ROOT, "tools", "debug", f"g7_trace_{str(dt).split('.')[-1]}.pt"))
    print(dt, {k: tuple(v.shape) for k, v in to.items()})
