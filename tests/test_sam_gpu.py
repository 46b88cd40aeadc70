"""GPU parity of the SAM kernels / sub-models against the CPU oracle and the committed reference fixtures (G7, G8)."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import pkg, load_fixture, fixture_sd, assert_close_bf16, rel_err
from oracle import ullava_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
RESULTS = []          # measured parity numbers, written to gpurun_out/parity_sam.json at session end (copied to profiles/)


@pytest.fixture(scope="module", autouse=True)
def _dump_results():
    yield
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_sam.json"), "w") as f:
        json.dump(RESULTS, f, indent=1)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def test_window_partition_roundtrip_and_zero_pad():
    ops = pkg("ops")
    B, H, W, C, ws = 2, 10, 10, 64, 4
    x = _rand(B, H, W, C, seed=1)
    win = ops.window_partition(x.view(-1, C).to(DEV), B, H, W, ws).cpu()
    ref, pad_hw = O.window_partition(x, ws)
    assert torch.equal(win.view(ref.shape), ref)
    sc = _rand(B, H, W, C, seed=2)
    out = ops.window_unpartition_add(win.to(DEV), sc.view(-1, C).to(DEV), B, H, W, ws).cpu()
    assert torch.equal(out.view(B, H, W, C), sc + O.window_unpartition(ref, ws, pad_hw, (H, W)))


@pytest.mark.parametrize("side,hd,nH,NB", [(14, 80, 2, 3), (8, 16, 2, 1)])
def test_relpos_tables(side, hd, nH, NB):
    ops = pkg("ops")
    S, C = side * side, nH * hd
    qkv = _rand(NB * S, 3 * C, seed=3)
    rph, rpw = _rand(2 * side - 1, hd, seed=4), _rand(2 * side - 1, hd, seed=5)
    oh, ow = ops.sam_relpos(qkv.to(DEV), (S * 3 * C, hd, 3 * C), rph.to(DEV), rpw.to(DEV), NB, nH, side, side, hd)
    q = qkv[:, :C].view(NB, S, nH, hd).permute(0, 2, 1, 3).reshape(NB * nH, S, hd)
    r_q = q.reshape(NB * nH, side, side, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, O.get_rel_pos(side, side, rph))
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, O.get_rel_pos(side, side, rpw))
    assert_close_bf16(oh.view(rel_h.shape), rel_h, ulps=1.0, what="rel_h")
    assert_close_bf16(ow.view(rel_w.shape), rel_w, ulps=1.0, what="rel_w")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L,size", [(27, 64), (127, 14), (27, 20), (63, 64), (27, 14)])
def test_rel_pos_interpolation_is_bit_exact(L, size, dtype):
    """get_rel_pos's resize of a table of another length (image_encoder.py:336-343), against the oracle's F.interpolate."""
    ops = pkg("ops")
    g = torch.Generator().manual_seed(L * 100 + size)
    tab = torch.randn(L, 80, generator=g).to(dtype)
    ref = O.get_rel_pos(size, size, tab)                       # [size, size, C] gather of the resized table
    got = ops.fit_rel_pos(tab.to(DEV), size).cpu()
    assert got.shape == (2 * size - 1, 80)
    idx = (torch.arange(size)[:, None] - torch.arange(size)[None, :]) + (size - 1)
    assert torch.equal(got[idx], ref)


def test_sam_attention_with_tables_of_another_length():
    """A SAM block whose rel_pos tables were trained at another window size: the attention path resizes them like the reference."""
    ops = pkg("ops")
    side, hd, nH, NB = 14, 80, 2, 2
    S, C = side * side, nH * hd
    x = _rand(NB, side, side, C, seed=16)
    sd = {"qkv.weight": _rand(3 * C, C, seed=17, scale=C ** -0.5), "qkv.bias": _rand(3 * C, seed=18, scale=0.1),
          "proj.weight": torch.eye(C).to(BF), "proj.bias": torch.zeros(C).to(BF),
          "rel_pos_h": _rand(39, hd, seed=19, scale=0.3), "rel_pos_w": _rand(39, hd, seed=20, scale=0.3)}
    ref = O.sam_attention(sd, "", x, nH).reshape(NB * S, C)
    qkv = F.linear(x.reshape(-1, C), sd["qkv.weight"], sd["qkv.bias"]).to(DEV)
    strides = (S * 3 * C, hd, 3 * C)
    vmax = float(qkv[:, 2 * C:].float().abs().max())
    vt = ops.transpose_v(qkv[:, 2 * C:], S * 3 * C, 3 * C, NB, S, nH, hd)
    att = torch.empty(NB * S, C, device=DEV, dtype=BF)
    ops.attention(qkv, qkv[:, C:], vt, att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False, scale_mode=0,
                  q_scale=hd ** -0.5, rel_h=sd["rel_pos_h"].to(DEV), rel_w=sd["rel_pos_w"].to(DEV), rel_pos_hw=(side, side))
    assert_close_bf16(att, ref, ulps=2.0, what="sam attention, resized tables", outlier_frac=2e-3, outlier_floor=vmax)
    # the image-order entry resizes them too (one window, no padding: the same tokens)
    att2 = ops.sam_window_attention(qkv[:S], sd["qkv.bias"].to(DEV), sd["rel_pos_h"].to(DEV), sd["rel_pos_w"].to(DEV), 1, side, side, nH, hd, side)
    assert_close_bf16(att2, ref[:S], ulps=2.0, what="window attention entry, resized tables", outlier_frac=2e-3, outlier_floor=vmax)


@pytest.mark.parametrize("H,W,B,nH", [(20, 20, 2, 2), (28, 14, 2, 2), (31, 17, 2, 2), (50, 37, 5, 16), (64, 64, 2, 16)])
def test_window_attention_on_image_order_tokens(H, W, B, nH):
    """ull_sam_window_attention (tokens stay in image order; the kernel does the window addressing, reads the zero-padded positions'
    q|k|v from the qkv bias and V through the transposing LDS load) against window_partition -> qkv Linear -> V^T pass -> attention ->
    window_unpartition with the generic kernels, and against the oracle's Block-level arithmetic."""
    # (the last two cases give every workgroup of sam_window_kernel a run of several (window, head) items: 960 and 800 items on 256 CUs,
    # runs that cross from head 15 of one window to head 0 of the next, windows that hang over the right and bottom edges)
    ops = pkg("ops")
    hd, ws = 80, 14
    C = nH * hd
    y = _rand(B, H, W, C, seed=31).to(DEV)                      # norm1 output
    w, b = _rand(3 * C, C, seed=32, scale=C ** -0.5).to(DEV), _rand(3 * C, seed=33, scale=0.3).to(DEV)
    rph, rpw = _rand(27, hd, seed=34, scale=0.3).to(DEV), _rand(27, hd, seed=35, scale=0.3).to(DEV)
    # the window-major chain with the generic kernels
    yw = ops.window_partition(y.view(-1, C), B, H, W, ws)
    NB, S = yw.shape[0] // (ws * ws), ws * ws
    qkv_w = ops.linear(yw, w, b)
    strides = (S * 3 * C, hd, 3 * C)
    vt = ops.transpose_v(qkv_w[:, 2 * C:], S * 3 * C, 3 * C, NB, S, nH, hd)
    att_w = torch.empty(NB * S, C, device=DEV, dtype=BF)
    ops.attention(qkv_w, qkv_w[:, C:], vt, att_w, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False, scale_mode=0,
                  q_scale=hd ** -0.5, rel_h=rph, rel_w=rpw, rel_pos_hw=(ws, ws))
    ref = ops.window_unpartition_add(att_w, torch.zeros(B * H * W, C, device=DEV, dtype=BF), B, H, W, ws)
    # image order
    qkv = ops.linear(y.view(-1, C), w, b)
    got = ops.sam_window_attention(qkv, b, rph, rpw, B, H, W, nH, hd, ws)
    # scores, softmax and P are the same numbers in both kernels; the fp32 P*V sum runs over the keys in another order
    d = (got.float() - ref.float()).abs()
    nd = float((d > 0).float().mean())
    print(f"image-order vs window-major chain {H}x{W}: {nd:.4%} of outputs differ, max {float(d.max()):.3g}")
    assert nd < 0.01 and float(d.max()) <= 2.0 ** -7 * float(ref.float().abs().max())
    # and the oracle (proj = identity)
    sd = {"qkv.weight": w.cpu(), "qkv.bias": b.cpu(), "proj.weight": torch.eye(C).to(BF), "proj.bias": torch.zeros(C).to(BF),
          "rel_pos_h": rph.cpu(), "rel_pos_w": rpw.cpu()}
    xw, pad_hw = O.window_partition(y.cpu(), ws)
    o = O.window_unpartition(O.sam_attention(sd, "", xw, nH), ws, pad_hw, (H, W)).reshape(B * H * W, C)
    vmax = float(qkv[:, 2 * C:].float().abs().max())
    assert_close_bf16(got, o, ulps=2.0, what="window attention on image-order tokens", outlier_frac=2e-3, outlier_floor=vmax)


@pytest.mark.parametrize("dt", [BF, torch.float16])
def test_window_attention_is_batch_independent_bit_for_bit(dt):
    """sam_window_kernel walks runs of (window, head) items, one workgroup per CU: which workgroup computes an item, and next to which
    other items, depends on the batch.  The result must not: image b of a batch of 6 equals the single-image call bit for bit
    (6 x 16 windows x 16 heads = 1536 items: six per workgroup; 1 x 256 items: one per workgroup)."""
    ops = pkg("ops")
    B, H, W, nH, hd, ws = 6, 50, 45, 16, 80, 14
    C = nH * hd
    qkv = _rand(B * H * W, 3 * C, seed=61).to(dt).to(DEV)
    b = _rand(3 * C, seed=62, scale=0.3).to(dt).to(DEV)
    rph, rpw = _rand(27, hd, seed=63, scale=0.3).to(dt).to(DEV), _rand(27, hd, seed=64, scale=0.3).to(dt).to(DEV)
    full = ops.sam_window_attention(qkv, b, rph, rpw, B, H, W, nH, hd, ws)
    assert bool(torch.isfinite(full.float()).all())
    for i in (0, 3, 5):
        one = ops.sam_window_attention(qkv[i * H * W:(i + 1) * H * W].contiguous(), b, rph, rpw, 1, H, W, nH, hd, ws)
        assert torch.equal(one.view(torch.int16), full[i * H * W:(i + 1) * H * W].view(torch.int16)), f"image {i}"


@pytest.mark.parametrize("side,hd,nH,NB", [(14, 80, 2, 3), (64, 80, 2, 1), (64, 32, 2, 1)])
def test_sam_encoder_attention(side, hd, nH, NB):
    """windowed (196 keys, register kernel + bias) and global (4096 keys, single-pass streaming kernel + bias) SAM attention.
    The streaming kernel rounds P = bf16(exp(s - running max)) before the normalisation, the reference after it: the two
    bf16 results carry independent rounding noise of the same size, so (a) a slightly larger share of elements may sit
    beyond the 2-ulp bound and (b) the kernel must be as close to the fp32 truth as the reference's own bf16 path is."""
    ops = pkg("ops")
    S, C = side * side, nH * hd
    x = _rand(NB, side, side, C, seed=6)
    sd = {"qkv.weight": _rand(3 * C, C, seed=7, scale=C ** -0.5), "qkv.bias": _rand(3 * C, seed=8, scale=0.1),
          "proj.weight": torch.eye(C).to(BF), "proj.bias": torch.zeros(C).to(BF),
          "rel_pos_h": _rand(2 * side - 1, hd, seed=9, scale=0.3), "rel_pos_w": _rand(2 * side - 1, hd, seed=10, scale=0.3)}
    ref = O.sam_attention(sd, "", x, nH).reshape(NB * S, C)           # proj = identity -> the attention output itself
    qkv = F.linear(x.reshape(-1, C), sd["qkv.weight"], sd["qkv.bias"]).to(DEV)
    strides = (S * 3 * C, hd, 3 * C)
    rel_h, rel_w = ops.sam_relpos(qkv, strides, sd["rel_pos_h"].to(DEV), sd["rel_pos_w"].to(DEV), NB, nH, side, side, hd)
    vt = ops.transpose_v(qkv[:, 2 * C:], S * 3 * C, 3 * C, NB, S, nH, hd)
    att = torch.empty(NB * S, C, device=DEV, dtype=BF)
    ops.attention(qkv, qkv[:, C:], vt, att, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False, scale_mode=0,
                  q_scale=hd ** -0.5, rel_h=rel_h, rel_w=rel_w)
    vmax = float(qkv[:, 2 * C:].float().abs().max())
    # measured on MI355X (round 5, printed on every run): windows 0, global hd 80: 3.3e-3 of the elements beyond the 2-ulp bound, global hd 32: 0
    frac = 5e-4 if S <= 1024 else 4.5e-3
    assert_close_bf16(att, ref, ulps=2.0, what=f"sam attention side={side}", outlier_frac=frac, outlier_floor=vmax)
    truth = O.sam_attention({k: v.float() for k, v in sd.items()}, "", x.float(), nH).reshape(NB * S, C)
    e_ours = float((att.float().cpu() - truth).pow(2).mean().sqrt())
    e_ref = float((ref.float() - truth).pow(2).mean().sqrt())
    assert e_ours <= 1.15 * e_ref, f"rms error vs fp32 truth: kernel {e_ours:.4g}, reference bf16 path {e_ref:.4g}"
    # fused path: the kernel builds the bias tables from the raw rel_pos parameters (Toeplitz product on the MFMA)
    att2 = torch.empty_like(att)
    ops.attention(qkv, qkv[:, C:], vt, att2, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False, scale_mode=0,
                  q_scale=hd ** -0.5, rel_h=sd["rel_pos_h"].to(DEV), rel_w=sd["rel_pos_w"].to(DEV), rel_pos_hw=(side, side))
    # (MFMA vs sequential fp32 accumulation order may flip a bf16 rounding of a table entry, so compare with the reference)
    assert_close_bf16(att2, ref, ulps=2.0, what=f"sam attention (fused rel-pos) side={side}", outlier_frac=frac, outlier_floor=vmax)
    # V handed over as rows of the q|k|v buffer: the 64 x 64 global kernel reads it through the transposing LDS load (other shapes
    # get their V^T image made by the wrapper) -- the same operands in the same MFMA slots, identical bits
    att4 = torch.full_like(att, float("nan"))
    ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], att4, NB, nH, S, S, hd, strides, strides, (S * C, hd, C), None, causal=False, scale_mode=0,
                  q_scale=hd ** -0.5, rel_h=sd["rel_pos_h"].to(DEV), rel_w=sd["rel_pos_w"].to(DEV), rel_pos_hw=(side, side), v_strides=strides)
    assert torch.equal(att4, att2)
    if side == 14 and hd == 80:
        # the path's own entry for 14 x 14 windows (here: whole windows, no padding) gives the same numbers up to the P*V sum order
        att3 = ops.sam_window_attention(qkv, sd["qkv.bias"].to(DEV), sd["rel_pos_h"].to(DEV), sd["rel_pos_w"].to(DEV), NB, side, side, nH, hd,
                                        side)
        assert_close_bf16(att3, ref, ulps=2.0, what="sam window attention entry", outlier_frac=frac, outlier_floor=vmax)
        e3 = float((att3.float().cpu() - truth).pow(2).mean().sqrt())
        assert e3 <= 1.15 * e_ref, f"rms error vs fp32 truth: window entry {e3:.4g}, reference bf16 path {e_ref:.4g}"
        d = (att3.float() - att2.float()).abs()
        assert float((d > 0).float().mean()) < 0.02, "window entry and generic window kernel differ on more than 2 % of the outputs"


@pytest.mark.parametrize("Sq,Sk,hd", [(6, 4096, 16), (4096, 6, 16), (6, 6, 32), (130, 1500, 64)])
def test_decoder_style_attention(Sq, Sk, hd):
    """SAM decoder attention (scores / sqrt(hd), softmax in bf16) incl. the 4096-key two-pass kernel."""
    ops = pkg("ops")
    n, H = 2, 8 if hd <= 32 else 2
    Di = H * hd
    q, k, v = _rand(n * Sq, Di, seed=11), _rand(n * Sk, Di, seed=12), _rand(n * Sk, Di, seed=13)
    qh = q.view(n, Sq, H, hd).transpose(1, 2)
    kh = k.view(n, Sk, H, hd).transpose(1, 2)
    vh = v.view(n, Sk, H, hd).transpose(1, 2)
    a = torch.softmax((qh @ kh.permute(0, 1, 3, 2)) / math.sqrt(hd), dim=-1)
    ref = (a @ vh).transpose(1, 2).reshape(n * Sq, Di)
    vt = ops.transpose_v(v.to(DEV), Sk * Di, Di, n, Sk, H, hd)
    out = torch.empty(n * Sq, Di, device=DEV, dtype=BF)
    ops.attention(q.to(DEV), k.to(DEV), vt, out, n, H, Sq, Sk, hd, (Sq * Di, hd, Di), (Sk * Di, hd, Di), (Sq * Di, hd, Di), None,
                  causal=False, scale_mode=2, scale=math.sqrt(hd))
    assert_close_bf16(out, ref, ulps=2.0, what=f"decoder attention {Sq}x{Sk}", outlier_frac=2e-3, outlier_floor=float(v.float().abs().max()))


@pytest.mark.parametrize("C,gelu", [(64, True), (256, False)])
def test_layernorm2d_channels_last(C, gelu):
    ops = pkg("ops")
    x = _rand(3, C, 5, 7, seed=14, scale=2.0)
    w, b = (1 + 0.1 * torch.randn(C)).to(BF), (0.1 * torch.randn(C)).to(BF)
    ref = O.layer_norm_2d(x, w, b)
    if gelu:
        ref = F.gelu(ref)
    out = ops.layernorm2d_cl(x.permute(0, 2, 3, 1).contiguous().view(-1, C).to(DEV), w.to(DEV), b.to(DEV), 1e-6, gelu)
    assert_close_bf16(out.view(3, 5, 7, C).permute(0, 3, 1, 2), ref, ulps=1.0, what="LayerNorm2d")


def test_neck_conv3x3_as_gemm():
    ops = pkg("ops")
    B, g, D = 2, 8, 64
    x = _rand(B, D, g, g, seed=15)
    w = _rand(D, D, 3, 3, seed=16, scale=(9 * D) ** -0.5)
    ref = F.conv2d(x, w, padding=1)
    xt = x.permute(0, 2, 3, 1).contiguous().view(-1, D).to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(D, 9 * D).contiguous().to(DEV)
    out = ops.linear(ops.im2col3x3(xt, B, g, g), wp)
    assert_close_bf16(out.view(B, g, g, D).permute(0, 3, 1, 2), ref, what="conv3x3")


def test_bilinear_matches_interpolate():
    ops = pkg("ops")
    m = _rand(3, 256, 256, seed=17)
    ref = O.postprocess_masks(m[:, None], (768, 1024), (480, 640))[:, 0]
    up = ops.bilinear(m.to(DEV), 256, 256, 1024, 1024)
    out = ops.bilinear(up, 768, 1024, 480, 640).cpu()
    assert out.dtype == torch.float32 and tuple(out.shape) == (3, 480, 640)
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def _decoder_engine(fx):
    C, S = pkg("configuration"), pkg("sam")
    cfg = C.SamConfig(depth=0)
    holder = S.build_sam_holder(cfg, device=DEV)
    sd = {k[len("visual_model."):]: v for k, v in fixture_sd(fx, BF).items()}
    missing = holder.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return S.SamEngine(holder, cfg), fixture_sd(fx, BF)


def test_mask_decoder_fixture_g7():
    fx = load_fixture("g7_sam_decoder_bf16.pt")
    eng, sd = _decoder_engine(fx)
    pe = eng.dense_pe().cpu()                                     # token-major [4096, 256]
    ref_pe = O.dense_pe(sd, (64, 64))[0].permute(1, 2, 0).reshape(4096, 256)
    assert torch.equal(pe, ref_pe), "dense PE must be bit-exact (constant folded with the reference's bf16 recipe)"
    g = torch.Generator().manual_seed(fx["image_embedding_seed"])
    emb = torch.randn(1, 256, 64, 64, generator=g).to(BF)
    emb_tm = emb[0].permute(1, 2, 0).reshape(4096, 256).contiguous().to(DEV)
    sd32 = {k: v.float() for k, v in sd.items()}
    for case in fx["cases"]:
        text = case["text_embeds"][:, 0].to(DEV)
        masks, iou = eng.decode(emb_tm, text)
        low = masks[:, 0:1].float().cpu()
        sp, de = O.prompt_encoder_text(sd, case["text_embeds"], (64, 64))
        sp32, de32 = O.prompt_encoder_text(sd32, case["text_embeds"].float(), (64, 64))
        # fp32 "truth": same bf16-rounded weights/inputs and the same bf16-computed dense PE (part of the reference's semantics)
        truth, truth_iou = O.mask_decoder(sd32, emb.float(), O.dense_pe(sd, (64, 64)).float(), sp32, de32, False)
        st = case["low_res_stride"]
        ref_low = case["low_res_masks"].float()
        e_ref, e_hip = rel_err(ref_low, truth[:, :, ::st, ::st]), rel_err(low, truth)
        e_direct = float((low[:, :, ::st, ::st] - ref_low).abs().max()) / case["low_res_max"]
        e_iou = rel_err(iou[:, 0:1], case["iou"])
        print(f"n={case['n']}: mask-logit err vs fp32 truth: reference bf16 {e_ref:.4f}, HIP {e_hip:.4f}; HIP vs reference fixture "
              f"{e_direct:.5f} (max|dlogit| / max|logit|); iou err {e_iou:.4f}")
        RESULTS.append(dict(test="g7_bf16", n=case["n"], hip_vs_reference=e_direct, hip_vs_fp32=e_hip, reference_vs_fp32=e_ref, iou=e_iou))
        assert e_hip <= max(2.0 * e_ref, 0.02)
        # direct bound against the reference's own bf16 output.  Layer 0 of the two-way transformer is reproduced bit for bit
        # (test_mask_decoder_stage_trace_bf16); after it single bf16 rounding flips are amplified (every image row attends to the same
        # 6 tokens), exactly as between two host CPUs running the identical reference code: the reference's own cross-host deviation is
        # measured here (oracle on this host vs the fixture) and HIP must not be further from the fixture than 1.6x that, nor > 2 %
        ocross, _ = O.mask_decoder(sd, emb, O.dense_pe(sd, (64, 64)), sp.to(BF), de, False)
        e_cross = float((ocross[:, :, ::st, ::st].float() - ref_low).abs().max()) / case["low_res_max"]
        RESULTS[-1]["reference_cross_host"] = e_cross
        print(f"      reference cross-host deviation (oracle on this CPU vs fixture): {e_cross:.5f}")
        assert e_direct <= min(max(1.6 * e_cross, 2.0 ** -7), 0.02), (e_direct, e_cross)
        post = eng.postprocess(masks[:, 0].contiguous(), (768, 1024), (480, 640)).cpu()
        assert post.dtype == torch.float32 and tuple(post.shape) == (case["n"], 480, 640)
        ref_post = O.postprocess_masks(masks[:, 0:1].cpu(), (768, 1024), (480, 640))[:, 0]
        assert float((post - ref_post).abs().max()) <= 1e-5 * float(ref_post.abs().max())


def _full_model(fx):
    C, M = pkg("configuration"), pkg("modeling_ullava")
    cfg, cd = fx["cfg"], fx["cfg"]["llm"]
    ucfg = C.UllavaConfig(llm_config=dict(vision_config=cd["vision_config"], vision_hidden_layer=cd["vision_hidden_layer"],
                                          projector_type="mlp", projector_from_scratch=bool(cd.get("projector_from_scratch", False)),   # the fixtures' reference model: False
                                          mm_token_ids=cd["mm_token_ids"], vocab_size=cd["vocab_size"],
                                          hidden_size=cd["hidden_size"], intermediate_size=cd["intermediate_size"],
                                          num_hidden_layers=cd["num_hidden_layers"], num_attention_heads=cd["num_attention_heads"]),
                            seg_token_idx=cfg["seg_token_idx"], loc_token_idx=cfg["loc_token_idx"], sam_config=dict(cfg["sam"]))
    model = M.UllavaForCausalLM(ucfg, device=DEV)
    sd = fixture_sd(fx, BF)
    model.load_state_dict(sd, strict=True)
    return model, sd


def test_full_forward_fixture_g8():
    fx = load_fixture("g8_full_tiny_bf16.pt")
    model, sd = _full_model(fx)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g).to(BF)
    out = model(images_sam=images_sam.to(DEV), images=fx["images"].to(DEV), input_ids=fx["input_ids"].to(DEV), labels=None,
                attention_mask=fx["attention_mask"].to(DEV), mask_list=[None, None], size_list=fx["size_list"],
                resize_list=fx["resize_list"], bbox_list=[None, None], inference=True)
    assert sorted(out.keys()) == fx["dict_keys"]
    assert [m.shape[0] for m in out["pred_masks"]] == [2, 1] and [b.shape[0] for b in out["pred_boxes"]] == [1, 2]   # [SEG]/[LOC] shift
    valid = fx["attention_mask"].bool()
    print("logits err", rel_err(out["logits"].cpu()[valid], fx["logits"][valid]))
    # measured 0.0037 = one bf16 ulp at the top of the logit range (an ulp is 2^-8 .. 2^-7 of the value): bound = two such flips
    assert rel_err(out["logits"].cpu()[valid], fx["logits"][valid]) <= 2.0 ** -7
    emb = model.get_visual_embs(images_sam.to(DEV)).cpu()
    e = rel_err(emb[:, ::16, ::4, ::4], fx["image_embeddings_sample"])
    print("SAM encoder embedding err vs reference", e)
    assert e < 0.0105                       # measured 0.0068 (x 1.5)
    for i in range(2):
        assert out["pred_masks"][i].dtype == torch.float32 and tuple(out["pred_masks"][i].shape) == tuple(fx["pred_mask_shapes"][i])
        em = rel_err(out["pred_masks"][i].cpu()[:, ::8, ::8], fx["pred_mask_samples"][i])
        eb = rel_err(out["pred_boxes"][i], fx["pred_boxes"][i])
        print(f"sample {i}: mask err {em:.4f} box err {eb:.4f}")
        RESULTS.append(dict(test="g8_bf16", sample=i, mask_vs_reference=em, box_vs_reference=eb))
        assert em < 0.018 and eb == 0.0      # measured: masks 0.0106 / 0.0120 (x 1.5), boxes bit-exact


def test_evaluate_greedy_tiny():
    """evaluate(temperature=0): ids identical to the oracle's greedy loop, masks produced for generated [SEG] tokens."""
    fx = load_fixture("g8_full_tiny_bf16.pt")
    model, sd = _full_model(fx)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g).to(BF)[:1]
    ids = fx["input_ids"][:1]
    seq, masks, boxes = model.evaluate(images_sam.to(DEV), fx["images"][:1].to(DEV), ids.to(DEV), [fx["size_list"][0]], [fx["resize_list"][0]],
                                       max_new_tokens=6, temperature=0)
    llm_sd = {k[4:]: v for k, v in sd.items() if k.startswith("llm.")}
    ref_seq, _ = O.greedy_generate(llm_sd, fx["cfg"]["llm"], ids, fx["images"][:1], None, 6)
    assert torch.equal(seq.cpu(), ref_seq), (seq.cpu().tolist(), ref_seq.tolist())
    n_seg = int((ref_seq[0, 1:] == fx["cfg"]["seg_token_idx"]).sum())
    assert masks[0].shape[0] == n_seg and tuple(masks[0].shape[1:]) == tuple(fx["size_list"][0])


def test_loss_kernels_against_reference_formulas():
    """mask_loss_sums / box_losses vs the reference's torch formulas (models/loss.py restated in the oracle), fp32."""
    ops = pkg("ops")
    g = torch.Generator().manual_seed(71)
    for n, (h, w) in ((3, (48, 64)), (1, (333, 500)), (5, (7, 9))):
        x = torch.randn(n, h, w, generator=g) * 4
        t = (torch.rand(n, h, w, generator=g) > 0.6).float()
        s = ops.mask_loss_sums(x.to(DEV), t.to(DEV)).cpu()
        bce = (s[:, 0] / (h * w)).sum() / (n + 1e-8)
        dice = (1 - (2 * s[:, 1] + 1e-6) / (s[:, 2] + s[:, 3] + 1e-6)).sum() / (n + 1e-8)
        torch.testing.assert_close(bce, O.sigmoid_ce_loss(x, t, n), rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(dice, O.dice_loss(x, t, n), rtol=2e-5, atol=1e-6)
    for dt in (torch.float32, BF):
        xy = torch.rand(6, 2, generator=g) * 0.5
        gt = torch.cat([xy, xy + 0.1 + torch.rand(6, 2, generator=g) * 0.4], 1)
        pred = (gt + torch.randn(6, 4, generator=g) * 0.1).to(dt)
        pred[2] = torch.tensor([0.6, 0.2, 0.5, 0.9]).to(dt)                      # x1 < x0: dropped from the GIoU term only
        out = ops.box_losses(pred.to(DEV), gt.to(DEV)).cpu()
        torch.testing.assert_close(out[0] / (6 + 1e-8), O.bbox_l1_loss(pred.float(), gt, 6), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(out[1] / (6 + 1e-8), O.bbox_giou_loss(pred.float(), gt, 6), rtol=1e-5, atol=1e-6)


def test_training_loss_dict_fixture_g10():
    """UllavaForCausalLM.forward(inference=False): same keys as the reference, every loss within the bf16 path's tolerance of the
    reference's bf16 run (G10), "ce_loss" aliased to the total like the reference's in-place accumulation."""
    fx = load_fixture("g10_train_losses_bf16.pt")
    model, sd = _full_model(fx)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g).to(BF)
    g2 = torch.Generator().manual_seed(fx["gt_seed"])
    n_seg = [2, 1]
    gt_masks = [(torch.rand(n_seg[i], *fx["size_list"][i], generator=g2) > 0.7).float() for i in range(2)]
    assert [t.sum().item() for t in gt_masks] == fx["gt_mask_sums"]               # the seeded targets are the generator's
    out = model(images_sam=images_sam.to(DEV), images=fx["images"].to(DEV), input_ids=fx["input_ids"].to(DEV), labels=fx["labels"].to(DEV),
                attention_mask=fx["attention_mask"].to(DEV), mask_list=[m.to(DEV) for m in gt_masks], size_list=fx["size_list"],
                resize_list=fx["resize_list"], bbox_list=[b.to(DEV) for b in fx["gt_boxes"]], inference=False)
    assert sorted(out.keys()) == fx["dict_keys"]
    assert out["ce_loss"] is out["loss"]
    for k, ref in fx["losses"].items():
        got = float(out[k])
        print(f"{k}: {got:.5f} (reference bf16 run {float(ref):.5f})")
        assert abs(got - float(ref)) <= 0.02 * abs(float(ref)) + 1e-3, k


@pytest.mark.parametrize("dt", [BF, torch.float16])
def test_full_width_sam_encoder_against_oracle(dt):
    """SAM ViT-H widths (1280 wide, 16 heads, 14x14 windows, 64x64 global grid, 1024x1024 image) with one windowed and one global
    block: HIP encoder vs the CPU oracle; the embedding must be as close to an fp32 evaluation as the oracle's 16-bit run (x1.5).
    Exercises the resident window kernel, the single-pass global kernel with in-kernel rel-pos, the fused patch embed and the
    neck at production sizes -- in fp16 the neck of image_encoder.py:117-124: fp32 convolutions (1280 -> 256, 3x3 over K = 2304 as a
    two-term split GEMM) and fp32 LayerNorm2d, only the result cast back."""
    C, S, W = pkg("configuration"), pkg("sam"), pkg("weights")
    cfg = C.SamConfig(depth=2, global_attn_indexes=[1])
    holder = S.build_sam_holder(cfg, device=DEV, dtype=dt)
    shapes = {k: tuple(v.shape) for k, v in holder.state_dict().items() if k.startswith("image_encoder.")}
    sd = {k: v.to(dt) for k, v in W.seeded_state_dict(shapes, 91, torch.float32).items()}
    res = holder.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    eng = S.SamEngine(holder, cfg)
    g = torch.Generator().manual_seed(92)
    img = torch.randn(1, 3, 1024, 1024, generator=g).to(dt)
    got = eng.encode(img.to(DEV)).cpu().view(1, 64, 64, 256).permute(0, 3, 1, 2)          # token-major -> NCHW
    scfg = dict(patch_size=16, depth=2, global_attn_indexes=[1], window_size=14, num_heads=16)
    osd = {"visual_model." + k: v for k, v in sd.items()}
    torch.set_num_threads(min(64, __import__("os").cpu_count()))
    ref = O.sam_image_encoder(osd, scfg, img)
    truth = O.sam_image_encoder({k: v.float() for k, v in osd.items()}, scfg, img.float())
    e_ref, e_hip = rel_err(ref, truth), rel_err(got, truth)
    print(f"SAM encoder (full width, 2 blocks, {dt}): HIP err vs fp32 {e_hip:.5f}, oracle 16-bit {e_ref:.5f}, HIP vs oracle {rel_err(got, ref):.5f}")
    assert got.dtype == dt and e_hip <= max(1.5 * e_ref, 2.0 ** -6 if dt == BF else 2.0 ** -9)


def test_forward_with_no_seg_or_loc_tokens():
    """edge case: a batch in which one sample (or every sample) has no [SEG] / [LOC] token -> empty [0, H, W] / [0, 4] predictions,
    like the reference's empty gathers (ullava.py:214-256), and no kernel launch on empty inputs."""
    fx = load_fixture("g8_full_tiny_bf16.pt")
    model, sd = _full_model(fx)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g).to(BF).to(DEV)
    seg, loc = fx["cfg"]["seg_token_idx"], fx["cfg"]["loc_token_idx"]
    for wipe in ("second", "both"):
        ids = fx["input_ids"].clone()
        rows = [1] if wipe == "second" else [0, 1]
        for r in rows:
            ids[r][(ids[r] == seg) | (ids[r] == loc)] = 17
        out = model(images_sam=images_sam, images=fx["images"].to(DEV), input_ids=ids.to(DEV), labels=None,
                    attention_mask=fx["attention_mask"].to(DEV), mask_list=[None, None], size_list=fx["size_list"],
                    resize_list=fx["resize_list"], bbox_list=[None, None], inference=True)
        for r in rows:
            assert tuple(out["pred_masks"][r].shape) == (0, *fx["size_list"][r]) and out["pred_masks"][r].dtype == torch.float32
            assert tuple(out["pred_boxes"][r].shape) == (0, 4)
        if wipe == "second":
            assert out["pred_masks"][0].shape[0] == 2 and out["pred_boxes"][0].shape[0] == 1
            assert bool(torch.isfinite(out["pred_masks"][0]).all())


@pytest.mark.parametrize("name,BF", [("g9_sam_blocks_bf16.pt", BF), ("g9_sam_blocks_fp16.pt", torch.float16)])
def test_sam_encoder_blocks_fixture_g9(name, BF):
    """G9: one windowed + one global ViT-H block at d=1280 on 1024x1024 -- HIP vs the REFERENCE's 16-bit output (strided sample); the fp16
    fixture carries the reference's fp32 neck (image_encoder.py:117-124) at the real widths."""
    C, S = pkg("configuration"), pkg("sam")
    fx = load_fixture(name)
    cfg = C.SamConfig(depth=2, global_attn_indexes=[1])
    holder = S.build_sam_holder(cfg, device=DEV, dtype=BF)
    sd = {k[len("visual_model."):]: v for k, v in fixture_sd(fx, BF).items()}
    res = holder.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    eng = S.SamEngine(holder, cfg)
    g = torch.Generator().manual_seed(fx["image_seed"])
    img = torch.randn(1, 3, 1024, 1024, generator=g).to(BF)
    tr, to = {}, {}
    got = eng.encode(img.to(DEV), trace=tr).cpu().view(1, 64, 64, 256).permute(0, 3, 1, 2)
    # stage-level: fraction of elements that differ from the reference's own tensors (fixture made on the build container's CPU), next to
    # the same fraction for the reference restatement run on THIS host's CPU -- the reference's cross-host reproducibility.  The HIP path
    # must not deviate more than the reference does from itself (x1.25 + 0.2 %): K = 1280 / 5120 reductions flip ~1e-3 of the bf16
    # roundings of every Linear whatever the implementation, and attention spreads each flip over its window.
    torch.set_num_threads(min(32, __import__("os").cpu_count()))
    O.sam_image_encoder(fixture_sd(fx, BF), fx["cfg"], img, trace=to)
    for k, ref in fx["trace"].items():
        mine, host = tr[k].cpu()[:, ::4, ::4, ::8], to[k][:, ::4, ::4, ::8]
        flips, cross = float((mine != ref).float().mean()), float((host != ref).float().mean())
        dmax = float((mine.float() - ref.float()).abs().max() / ref.float().abs().max())
        print(f"G9 stage {k:12s}: differing elements HIP {flips:.5f} / reference cross-host {cross:.5f}, max|d|/max {dmax:.2e}")
        RESULTS.append(dict(test="g9_trace_" + name[14:18], stage=k, frac_differing=flips, reference_cross_host=cross, max_rel=dmax))
        assert flips <= 1.25 * cross + 2e-3, (k, flips, cross)
    e = float((got[:, ::2, ::2, ::2].float() - fx["embedding_sample"].float()).abs().max()) / fx["embedding_max"]
    print(f"G9 SAM blocks (d=1280, {name}): HIP vs reference fixture {e:.5f}")
    RESULTS.append(dict(test="g9_" + name[14:18], hip_vs_reference=e))
    # bf16: measured 0.0112 (x 1.5); fp16: measured 0.00185 (bound 0.0028); the per-stage flip counts above carry the cross-host rule
    assert got.dtype == BF and e < (0.017 if BF == torch.bfloat16 else 0.0028)


def test_evaluate_fixture_g11():
    """evaluate(temperature=0) against the reference-assembled fixture: ids exact, mask / box VALUES compared (not only shapes)."""
    fx = load_fixture("g11_evaluate_bf16.pt")
    model, sd = _full_model(fx)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g).to(BF)[:1]
    for use_cache in (False, True):
        model.llm.config.use_cache = use_cache
        seq, masks, boxes = model.evaluate(images_sam.to(DEV), fx["images"].to(DEV), fx["input_ids"].to(DEV), [fx["size"]], [fx["resize"]],
                                           max_new_tokens=6, temperature=0)
        assert torch.equal(seq.cpu(), fx["sequences"]), (seq.tolist(), fx["sequences"].tolist())
        assert tuple(masks[0].shape) == (fx["low_res_masks"].shape[0], *fx["size"]) and masks[0].dtype == torch.float32
        em = float((masks[0].cpu()[:, ::4, ::4] - fx["pred_mask_sample"]).abs().max()) / fx["pred_mask_max"]
        eb = rel_err(boxes[0], fx["pred_boxes"])
        print(f"evaluate(use_cache={use_cache}): mask err vs reference {em:.4f}, box err {eb:.4f}")
        RESULTS.append(dict(test="g11_evaluate_bf16", use_cache=use_cache, mask_vs_reference=em, box_vs_reference=eb))
        assert em < 0.0155 and eb == 0.0     # measured: masks 0.0103 (x 1.5), boxes bit-exact


def test_evaluate_sampling_path_is_seeded_and_valid():
    """default temperature=0.2 -> do_sample=True (reference ullava.py:343,356): same torch seed -> same ids; ids within vocab."""
    fx = load_fixture("g11_evaluate_bf16.pt")
    model, sd = _full_model(fx)
    g = torch.Generator().manual_seed(fx["images_sam_seed"])
    _ = torch.randn(2, 3, 28, 28, generator=g)
    images_sam = torch.randn(2, 3, 1024, 1024, generator=g).to(BF)[:1].to(DEV)
    runs = []
    for _i in range(2):
        torch.manual_seed(1234)
        seq, masks, boxes = model.evaluate(images_sam, fx["images"].to(DEV), fx["input_ids"].to(DEV), [fx["size"]], [fx["resize"]],
                                           max_new_tokens=5, top_p=0.9)
        runs.append(seq.cpu())
        n_seg = int((seq[0, 1:] == fx["cfg"]["seg_token_idx"]).sum())
        assert masks[0].shape[0] == n_seg and bool(torch.isfinite(masks[0]).all())
    assert torch.equal(runs[0], runs[1]) and int(runs[0].max()) < fx["cfg"]["llm"]["vocab_size"]


def test_mask_decoder_stage_trace_bf16_layer0_bit_exact():
    """Stage-level parity against the reference's own intermediate tensors (G7 trace, n = 1): the first TwoWayAttentionBlock --
    self attention, token->image attention over 4096 keys (incl. at::linear's unfused-bias path for the non-contiguous image keys),
    MLP, image->token attention, four LayerNorms -- must reproduce the reference's bf16 tensors bit for bit, up to isolated
    single-ulp flips from fp32 summation order (< 0.1 % of the elements of any stage; < 0.3 % for the block's last tensor, the
    4096 x 256 keys after norm4, which collects the flips of all stages before it)."""
    fx = load_fixture("g7_sam_decoder_bf16.pt")
    eng, sd = _decoder_engine(fx)
    g = torch.Generator().manual_seed(fx["image_embedding_seed"])
    emb = torch.randn(1, 256, 64, 64, generator=g).to(BF)
    emb_tm = emb[0].permute(1, 2, 0).reshape(4096, 256).contiguous().to(DEV)
    case = fx["cases"][0]
    tr = {}
    eng.decode(emb_tm, case["text_embeds"][:, 0].to(DEV), trace=tr)
    # the same stages from the reference restatement run on THIS host's CPU: how far the reference moves between hosts (the fixture was
    # made on the build container's CPU; torch's vectorised exp / reductions differ by ISA)
    sdo = fixture_sd(fx, BF)
    osp, ode = O.prompt_encoder_text(sdo, case["text_embeds"], (64, 64))
    host = {}
    O.mask_decoder(sdo, emb, O.dense_pe(sdo, (64, 64)), osp.to(BF), ode, False, trace=host)
    rows = []
    for k, ref in case["trace"].items():
        if k not in tr:
            continue
        got = tr[k].cpu()
        got = got.reshape(1, -1, ref.shape[-1])
        hst = host[k].reshape(1, -1, ref.shape[-1])
        if got.shape[1] != ref.shape[1]:
            got, hst = got[:, ::16], hst[:, ::16]
        flips, cross = float((got != ref).float().mean()), float((hst != ref).float().mean())
        dmax = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
        rows.append((k, flips, dmax))
        print(f"{k:12s} differing elements HIP {flips:.5f} / reference cross-host {cross:.5f}  max|d|/max {dmax:.2e}")
        RESULTS.append(dict(test="g7_trace_bf16", stage=k, frac_differing=flips, reference_cross_host=cross, max_rel=dmax))
        if k.startswith("l0."):
            assert flips <= (3e-3 if k == "l0.norm4" else 1e-3) and dmax <= 2.0 ** -7, (k, flips, dmax)
    assert len(rows) >= 12


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_fused_decoder_kernels_match_the_op_by_op_chain(dt):
    """csrc/sam_decoder.hip (one or two launches per attention / MLP block, LDS-resident key tiles) against the op-by-op decode that
    the stage-trace test pins to the reference: same rounding points, so the two differ only by fp32 summation order -- the queries
    after the first block must agree up to isolated one-ulp flips, the final mask logits within the decoder's flip-amplification
    envelope (the same bound as HIP vs reference fixture), IoU predictions likewise.  n = 1 and a batched n = 5 over 2 images."""
    C, S = pkg("configuration"), pkg("sam")
    fx = load_fixture("g7_sam_decoder_bf16.pt")
    cfg = C.SamConfig(depth=0)
    holder = S.build_sam_holder(cfg, device=DEV, dtype=dt)
    sd = {k[len("visual_model."):]: v for k, v in fixture_sd(fx, dt).items()}
    holder.load_state_dict(sd, strict=False)
    eng = S.SamEngine(holder, cfg)
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(2, 4096, 256, generator=g).to(dt).to(DEV)
    text = torch.randn(5, 256, generator=g).to(dt).to(DEV)
    idx = torch.tensor([0, 1, 1, 0, 1], device=DEV)
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    for e_, t_, i_ in ((emb[0], text[:1], None), (emb, text, idx)):
        eng.fused_decoder = True
        mf, iouf = eng.decode(e_, t_, i_)
        eng.fused_decoder = False
        mu, iouu = eng.decode(e_, t_, i_)
        d = (mf.float() - mu.float()).abs()
        mx = float(mu.float().abs().max())
        frac = float((d > 0).float().mean())
        print(f"{dt} n={t_.shape[0]}: fused vs op-by-op mask logits: max {float(d.max()) / mx:.2e} of max|logit|, differing {frac:.3f}; "
              f"iou {rel_err(iouf, iouu):.2e}")
        assert float(d.max()) <= 4 * ulp * mx
        assert rel_err(iouf, iouu) <= 4 * ulp
    # first block in isolation: queries after self-attention + LN, token->image attention + LN, MLP + LN; keys after image->token + LN
    ops = pkg("ops")
    tr = holder.mask_decoder.transformer
    l0 = tr.layers[0]
    md = holder.mask_decoder
    tokens = torch.cat([torch.cat([md.iou_token.weight, md.mask_tokens.weight], 0).unsqueeze(0), text[:1].unsqueeze(1)], dim=1).contiguous()
    keys = ops.add_rows(emb[0], holder.prompt_encoder.no_mask_embed.weight).view(1, 4096, 256)
    pos = eng.dense_pe()
    att0 = ops.sam_self_attn_heads(tokens, tokens, l0.self_attn, first=True)
    q1, (qp,) = ops.sam_out_ln(att0, None, tokens, l0.self_attn.out_proj, l0.norm1, projs=[(l0.cross_attn_token_to_image.q_proj, True)])
    ref1 = ops.layernorm(eng._attn(l0.self_attn, tokens.view(6, 256), tokens.view(6, 256), tokens.view(6, 256), 1, 6, 6), l0.norm1.weight, l0.norm1.bias, 1e-5)
    flip = 1.0 if dt == torch.bfloat16 else 8.0         # fp16's rounding grid is 8x finer: fp32 summation-order noise flips 8x more roundings
    assert float((q1.view(6, 256) != ref1).float().mean()) <= 0.01 * flip
    att1 = ops.sam_t2i_attention(qp, keys, pos, l0.cross_attn_token_to_image, late_bias_kv=True)
    q2, _ = ops.sam_out_ln(att1, q1, tokens, l0.cross_attn_token_to_image.out_proj, l0.norm2)
    qq, kk = ops.add_rows(q1.view(6, 256), tokens.view(6, 256)), ops.add_rows(keys.view(4096, 256), pos)
    ref2 = ops.layernorm(eng._attn(l0.cross_attn_token_to_image, qq, kk, keys.view(4096, 256), 1, 6, 4096, residual=q1.view(6, 256), unfused_bias=("k", "v")),
                         l0.norm2.weight, l0.norm2.bias, 1e-5)
    f2 = float((q2.view(6, 256) != ref2).float().mean())
    q3, (kp, vp) = ops.sam_token_mlp_ln(q2, tokens, l0.mlp.lin1, l0.mlp.lin2, l0.norm3,
                                        projs=[(l0.cross_attn_image_to_token.k_proj, True), (l0.cross_attn_image_to_token.v_proj, False)])
    m_ = ops.linear(q2.view(6, 256), l0.mlp.lin1.weight, l0.mlp.lin1.bias, act="relu")
    ref3 = ops.layernorm(ops.linear(m_, l0.mlp.lin2.weight, l0.mlp.lin2.bias, residual=q2.view(6, 256)), l0.norm3.weight, l0.norm3.bias, 1e-5)
    f3 = float((q3.view(6, 256) != ref3).float().mean())
    k1 = ops.sam_i2t_attention_ln(keys, pos, kp, vp, l0.cross_attn_image_to_token, l0.norm4, late_bias_q=True)
    qq3 = ops.add_rows(q3.view(6, 256), tokens.view(6, 256))
    ref4 = ops.layernorm(eng._attn(l0.cross_attn_image_to_token, kk, qq3, q3.view(6, 256), 1, 4096, 6, residual=keys.view(4096, 256), unfused_bias=("q",)),
                         l0.norm4.weight, l0.norm4.bias, 1e-5)
    f4 = float((k1.view(4096, 256) != ref4).float().mean())
    print(f"{dt} block 0, fraction of elements differing from the op-by-op chain: t2i {f2:.4f}, mlp {f3:.4f}, i2t {f4:.5f}")
    assert f2 <= 0.02 * flip and f3 <= 0.02 * flip and f4 <= 0.002 * flip


def test_evaluate_with_four_sam_images_matches_one_by_one():
    """evaluate() on a batch of 4 (different original sizes; the SAM encoder runs on its own stream beside the LLM, the mask decoder of
    all prompts as one chain of launches) returns, row by row, what four single-image calls return: ids identical, masks / boxes equal."""
    fx = load_fixture("g11_evaluate_bf16.pt")
    model, sd = _full_model(fx)
    g = torch.Generator().manual_seed(77)
    images_sam = torch.randn(4, 3, 1024, 1024, generator=g).to(BF).to(DEV)
    images = torch.cat([fx["images"]] * 4).to(DEV)
    images = (images.float() * torch.tensor([1.0, 0.9, 1.1, 0.8], device=DEV).view(4, 1, 1, 1)).to(BF)
    ids = torch.cat([fx["input_ids"]] * 4).to(DEV)
    sizes = [fx["size"], (37, 53), (64, 48), (50, 50)]
    resizes = [fx["resize"], (712, 1024), (1024, 768), (1024, 1024)]
    for use_cache in (False, True):
        model.llm.config.use_cache = use_cache
        seq4, masks4, boxes4 = model.evaluate(images_sam, images, ids, sizes, resizes, max_new_tokens=6, temperature=0)
        assert len(masks4) == 4 and seq4.shape[0] == 4
        for b in range(4):
            seq1, masks1, boxes1 = model.evaluate(images_sam[b:b + 1], images[b:b + 1], ids[b:b + 1], [sizes[b]], [resizes[b]],
                                                  max_new_tokens=6, temperature=0)
            n = seq1.shape[1]
            assert torch.equal(seq4[b, :n], seq1[0]), (b, seq4[b].tolist(), seq1[0].tolist())
            assert tuple(masks4[b].shape) == tuple(masks1[0].shape) and tuple(masks4[b].shape[1:]) == tuple(sizes[b])
            if masks1[0].numel():
                scale = float(masks1[0].abs().max()) + 1e-6
                assert float((masks4[b] - masks1[0]).abs().max()) <= 2e-2 * scale, (b, float((masks4[b] - masks1[0]).abs().max()), scale)
            if boxes1[0].numel():
                assert float((boxes4[b].float() - boxes1[0].float()).abs().max()) <= 2e-2
