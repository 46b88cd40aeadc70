"""Shared test plumbing: build the HIP-backed model from a golden fixture and compare with the reference
outputs stored in the fixture and with the CPU oracle.  (Imports `oracle/` -- tests only.)"""
import importlib
import os

import torch

from oracle import ullava_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pkg(sub=""):
    return importlib.import_module("u-llava_amd" + ("." + sub if sub else ""))


def load_fixture(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=True)


def fixture_sd(fx, dtype=None):
    W = pkg("weights")
    dt = dtype or getattr(torch, fx["dtype"].split(".")[-1])
    return {k: v.to(dt) for k, v in W.seeded_state_dict(fx["shapes"], fx["seed"], torch.float32).items()}


def rel_err(a, b):
    """max |a-b| / max |b| in fp32."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def assert_close_bf16(a, b, ulps=2.0, floor=None, what="", outlier_frac=0.0, outlier_floor=None):
    """See below; with `outlier_frac` > 0 that fraction of elements may miss the tight bound as long as ALL elements meet the
    same bound evaluated with `outlier_floor` as the absolute floor (e.g. attention: one bf16 flip of a dominant probability
    p ~ 0.5 moves the output by 2^-9 * max|v| whatever the output's own magnitude)."""
    if outlier_frac > 0.0:
        a_, b_ = a.detach().float().cpu(), b.detach().float().cpu()
        fl = float(b_.abs().max()) * 0.02 if floor is None else floor
        bad = (a_ - b_).abs() > ulps * 2.0 ** -7 * torch.maximum(b_.abs(), torch.full_like(b_, fl))
        print(f"{what}: {float(bad.float().mean()):.2e} of the elements beyond the tight bound (allowed {outlier_frac:.1e})")
        assert float(bad.float().mean()) <= outlier_frac, f"{what}: {int(bad.sum())}/{bad.numel()} elements beyond the tight bound"
        return assert_close_bf16(a, b, ulps, outlier_floor, what)
    """|a-b| <= ulps * 2^-7 * max(|b|, floor).  One bf16 ulp is between 2^-8 and 2^-7 of the value, so `ulps=1` admits a
    single rounding flip anywhere; `floor` (default 2% of max|b|) sets the absolute tolerance for near-zero entries."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    fl = float(b.abs().max()) * 0.02 if floor is None else floor
    tol = ulps * 2.0 ** -7 * torch.maximum(b.abs(), torch.full_like(b, fl))
    bad = (a - b).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max|d|={float((a - b).abs().max()):.4g} " \
                                f"max|ref|={float(b.abs().max()):.4g}"


def core_model_from_fixture(fx, device):
    M = pkg("modeling_core")
    C = pkg("configuration")
    cd = fx["cfg"]
    cfg = C.UllavaCoreConfig(vision_config=cd["vision_config"], vision_hidden_layer=cd["vision_hidden_layer"],
                             projector_type=cd["projector_type"], projector_from_scratch=bool(cd.get("projector_from_scratch", False)),
                             mm_token_ids=cd["mm_token_ids"],
                             vocab_size=cd["vocab_size"], hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                             num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"],
                             rms_norm_eps=cd["rms_norm_eps"], rope_theta=cd["rope_theta"])
    model = M.UllavaCoreForCausalLM(cfg, device=device)
    sd = fixture_sd(fx, torch.bfloat16)
    missing = model.load_state_dict(sd, strict=True)
    return model, sd


def run_core_fixture(name, device="cuda:0"):
    """HIP forward on a G1-style fixture -> error statistics (used by smoke() and the gpu tests)."""
    fx = load_fixture(name)
    model, sd = core_model_from_fixture(fx, device)
    ids, mask, images = fx["input_ids"].to(device), fx["attention_mask"].to(device), fx["images"].to(device)
    out = model(input_ids=ids, attention_mask=mask, images=images, output_hidden_states=True)
    torch.cuda.synchronize()
    # fp32 "truth": same bf16-rounded weights/inputs, fp32 arithmetic
    sd32 = {k: v.float() for k, v in sd.items()}
    truth = O.core_forward(sd32, fx["cfg"], fx["input_ids"], fx["attention_mask"], fx["images"].float())
    e_ref = rel_err(fx["logits"], truth["logits"])
    e_hip = rel_err(out.logits, truth["logits"])
    stats = dict(logits_err_vs_ref=rel_err(out.logits, fx["logits"]), ref_err_vs_fp32=e_ref, hip_err_vs_fp32=e_hip,
                 hidden_err_vs_ref=[round(rel_err(h, r), 5) for h, r in zip(out.hidden_states, fx["hidden_states"])],
                 tol=max(1.5 * e_ref, 2.0 ** -6))
    if "greedy_prompt" in fx:
        seq = model.generate(input_ids=fx["greedy_prompt"].to(device), images=images[:1], max_new_tokens=8, do_sample=False)
        stats["greedy_equal"] = bool(torch.equal(seq.cpu(), fx["greedy_sequences"]))
        stats["greedy"] = seq[0, fx["greedy_prompt"].shape[1]:].tolist()
    return stats


def per_op_inputs(seed, dtype):
    """Seeded inputs / weights of the G6 ops (shared with the tests: they regenerate these, the fixture only holds outputs)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dtype)
    return dict(rms_x=r(96, 4096, sc=1.5), rms_w=(torch.randn(4096, generator=g) * 0.1 + 1.0).to(dtype),
                rope_q=r(1, 2, 1024, 128), rope_k=r(1, 2, 1024, 128),
                gate=r(160, 1024, sc=2.0), up=r(160, 1024), act_x=r(160, 1024, sc=2.5),
                px224=r(2, 3, 224, 224), px336=r(1, 3, 336, 336),
                patch_w=r(1024, 3, 14, 14, sc=0.02), cls=r(1024, sc=0.02), pos224=r(257, 1024, sc=0.02), pos336=r(577, 1024, sc=0.02),
                ln_w=(torch.randn(1024, generator=g) * 0.1 + 1.0).to(dtype), ln_b=r(1024, sc=0.1))


def digest_matches(t, d):
    """Does tensor t reproduce a gen_golden._digest record bit for bit?  (sha256 of the raw bytes)"""
    import hashlib
    flat = t.detach().cpu().contiguous().view(-1)
    raw = flat.view(torch.int16 if t.element_size() == 2 else torch.int32).numpy().tobytes()
    return tuple(t.shape) == tuple(d["shape"]) and hashlib.sha256(raw).hexdigest() == d["sha256"]
