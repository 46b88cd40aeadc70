"""GPU parity of each HIP kernel (called through the C-ABI) against the CPU oracle's restatement of the same
reference op on the same seeded inputs.  Tolerance: a couple of bf16 ulps (2^-8 relative) -- the kernels keep the
reference's bf16 rounding points, so differences come only from fp32 summation order."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import pkg, assert_close_bf16, rel_err
from oracle import ullava_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def _mm_ref(x, w):
    """bf16 matmul with fp32 accumulation, output rounded to bf16 (what torch's bf16 Linear does)."""
    return (x.float() @ w.float().t()).to(BF)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (77, 100, 192), (1, 4, 64), (300, 32011 // 13, 256),
                                   (515, 12288, 4096), (1029, 4096, 11008)])
def test_gemm_plain(M, N, K):
    ops = pkg("ops")
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    y = ops.linear(x.to(DEV), w.to(DEV))
    assert_close_bf16(y, _mm_ref(x, w), what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K,kind", [(256 * 20 + 40, 256 * 13, 512, "plain"), (256 * 33, 256 * 8, 1024, "bias_resid"),
                                          (256 * 9 + 3, 256 * 30, 256, "swiglu"), (4100, 32011, 128, "odd")])
def test_gemm_streamk_tail(M, N, K, kind):
    """Tile counts that are not a multiple of the CU count: the tail tiles are split along K and summed by the finalize kernel."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    x = _rand(M, K, seed=21)
    if kind == "swiglu":
        wg, wu = _rand(N // 2, K, seed=22, scale=K ** -0.5), _rand(N // 2, K, seed=23, scale=K ** -0.5)
        y = ops.linear(x.to(DEV), M_.interleave_gate_up(wg, wu).to(DEV), swiglu=True)
        ref = F.silu(_mm_ref(x, wg)) * _mm_ref(x, wu)
    elif kind == "bias_resid":
        w, b, r = _rand(N, K, seed=22, scale=K ** -0.5), _rand(N, seed=23), _rand(M, N, seed=24)
        y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=r.to(DEV))
        lin = F.linear(x.float(), w.float(), b.float()).to(BF)
        # a 1-ulp flip of the (larger) Linear output survives the residual add: bound those few by the Linear's magnitude
        assert_close_bf16(y, r + lin, what="stream-K bias_resid", outlier_frac=1e-5, outlier_floor=float(lin.float().abs().max()))
        return
    else:
        w = _rand(N, K, seed=22, scale=K ** -0.5)
        y = ops.linear(x.to(DEV), w.to(DEV))
        ref = _mm_ref(x, w)
    assert_close_bf16(y, ref, what=f"stream-K {kind}")


@pytest.mark.parametrize("M", [1, 2, 3, 4])
def test_gemv_decode_shapes(M):
    ops, M_ = pkg("ops"), pkg("modeling_core")
    K, N, I = 1088, 520, 352
    x, w, b, r = _rand(M, K, seed=31), _rand(N, K, seed=32, scale=K ** -0.5), _rand(N, seed=33), _rand(M, N, seed=34)
    assert_close_bf16(ops.linear(x.to(DEV), w.to(DEV)), _mm_ref(x, w), what="gemv plain")
    t = F.linear(x.float(), w.float(), b.float()).to(BF)
    assert_close_bf16(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act="gelu"), F.gelu(t), what="gemv bias+gelu")
    assert_close_bf16(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=r.to(DEV)), r + t, what="gemv bias+residual")
    wg, wu = _rand(I, K, seed=35, scale=K ** -0.5), _rand(I, K, seed=36, scale=K ** -0.5)
    y = ops.linear(x.to(DEV), M_.interleave_gate_up(wg, wu).to(DEV), swiglu=True)
    assert_close_bf16(y, F.silu(_mm_ref(x, wg)) * _mm_ref(x, wu), what="gemv swiglu")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [2, 4, 5, 8, 16])
def test_skinny_gemm_batched_decode_shapes(M, dt):
    """2 <= M <= 16 against LLaMA-sized weights (ull_gemm_skinny: the weight stream on the matrix cores): every epilogue of the
    decode step, a vocabulary that is not a multiple of 16, the fused RMSNorm form -- against fp32 references and against the same
    rows pushed through the M = 1 GEMV one by one (same rounding points; fp32 summation order differs)."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    g = torch.Generator().manual_seed(M)
    def rnd_t(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dt)
    def close(a, b, what):
        a, b = a.float().cpu(), b.float().cpu()
        bound = b.abs().clamp_min(float(b.abs().max()) * 1e-2) * ulp * 2.02
        bad = (a - b).abs() > bound
        assert float(bad.float().mean()) <= 2e-3, f"{what}: {int(bad.sum())}/{bad.numel()} beyond 2 ulp"
    K, N = 4096, 1043                     # N * K >= 2^22, N % 16 != 0
    x, w, b, r = rnd_t(M, K), rnd_t(N, K, scale=K ** -0.5), rnd_t(N), rnd_t(M, N)
    xd, wd = x.to(DEV), w.to(DEV)
    ref = x.double() @ w.double().t()
    close(ops.linear(xd, wd), ref.to(dt), "plain")
    close(ops.linear(xd, wd, b.to(DEV), residual=r.to(DEV)), (r.double() + (ref + b.double()).to(dt).double()).to(dt), "bias + residual")
    close(ops.linear(xd, wd, b.to(DEV), act="gelu"), F.gelu((ref + b.double()).to(dt).float()).to(dt), "bias + gelu")
    out32 = ops.linear(xd, wd, out_f32=True)
    assert out32.dtype == torch.float32 and float((out32.cpu().double() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    one_by_one = torch.cat([ops.linear(xd[i:i + 1], wd, b.to(DEV), residual=r.to(DEV)[i:i + 1]) for i in range(M)])
    close(ops.linear(xd, wd, b.to(DEV), residual=r.to(DEV)), one_by_one, "against the GEMV row by row")
    I = 2048
    wg, wu = rnd_t(I, K, scale=K ** -0.5), rnd_t(I, K, scale=K ** -0.5)
    y = ops.linear(xd, M_.interleave_gate_up(wg, wu).to(DEV), swiglu=True)
    gate, up = (x.double() @ wg.double().t()).to(dt), (x.double() @ wu.double().t()).to(dt)
    close(y, (F.silu(gate.float()).to(dt).float() * up.float()).to(dt), "swiglu")
    # the down projection's K = 11008 (344 k-steps: an uneven eighth per wave)
    K2, N2 = 11008, 400
    x2, w2, r2 = rnd_t(M, K2), rnd_t(N2, K2, scale=K2 ** -0.5), rnd_t(M, N2)
    close(ops.linear(x2.to(DEV), w2.to(DEV), residual=r2.to(DEV)), (r2.double() + (x2.double() @ w2.double().t()).to(dt).double()).to(dt), "K = 11008 + residual")
    nw = rnd_t(K) * 0.1 + 1.0
    xn = O.rms_norm(x, nw, 1e-6) if dt == torch.bfloat16 else None
    if xn is not None:
        close(ops.linear(xd, wd, rms_w=nw.to(DEV), rms_eps=1e-6), (xn.double() @ w.double().t()).to(dt), "rmsnorm + linear")


@pytest.mark.parametrize("M,K", [(1, 4096), (4, 4096), (2, 1088), (1, 11008)])
def test_gemv_fused_rmsnorm(M, K):
    """decode-step input_layernorm -> projection in one launch == the two separate kernels == the reference ops."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    N, I = 264, 96
    x, g, w = _rand(M, K, seed=41), _rand(K, seed=42) + 1.0, _rand(N, K, seed=43, scale=K ** -0.5)
    xn = O.rms_norm(x, g, 1e-6)
    y = ops.linear(x.to(DEV), w.to(DEV), rms_w=g.to(DEV), rms_eps=1e-6)
    assert_close_bf16(y, _mm_ref(xn, w), what="gemv rmsnorm")
    if K <= 8192:                                   # (the stand-alone row-norm kernel holds a row in registers: D <= 8192)
        two = ops.linear(ops.rmsnorm(x.to(DEV), g.to(DEV), 1e-6), w.to(DEV))
        assert_close_bf16(y, two.cpu(), ulps=1.0, what="fused vs rmsnorm kernel + gemv")
    wg, wu = _rand(I, K, seed=44, scale=K ** -0.5), _rand(I, K, seed=45, scale=K ** -0.5)
    y = ops.linear(x.to(DEV), M_.interleave_gate_up(wg, wu).to(DEV), swiglu=True, rms_w=g.to(DEV), rms_eps=1e-6)
    assert_close_bf16(y, F.silu(_mm_ref(xn, wg)) * _mm_ref(xn, wu), what="gemv rmsnorm swiglu")


def test_gemm_transpose_detecting():
    """A = I against an asymmetric W catches swapped operands / C layouts."""
    ops = pkg("ops")
    K = 128
    x = torch.eye(K).to(BF)
    w = (torch.arange(256 * K).reshape(256, K) % 251).float().to(BF) / 16
    y = ops.linear(x.to(DEV), w.to(DEV))
    assert torch.equal(y.cpu().float(), w.float().t()), "C layout / operand swap"


@pytest.mark.parametrize("act", [None, "quick_gelu", "gelu", "relu"])
def test_gemm_bias_act_residual(act):
    ops = pkg("ops")
    M, N, K = 200, 264, 320
    x, w, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=K ** -0.5), _rand(N, seed=5), _rand(M, N, seed=6)
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=act)
    t = F.linear(x.float(), w.float(), b.float()).to(BF)
    ref = {None: lambda v: v, "quick_gelu": O.quick_gelu, "gelu": F.gelu, "relu": F.relu}[act](t)
    assert_close_bf16(y, ref, what=f"bias+{act}")
    y2 = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=r.to(DEV))
    assert_close_bf16(y2, r + t, what="bias+residual")


def test_gemm_swiglu():
    ops, M_ = pkg("ops"), pkg("modeling_core")
    M, I, K = 150, 352, 256
    x, wg, wu = _rand(M, K, seed=7), _rand(I, K, seed=8, scale=K ** -0.5), _rand(I, K, seed=9, scale=K ** -0.5)
    y = ops.linear(x.to(DEV), M_.interleave_gate_up(wg, wu).to(DEV), swiglu=True)
    ref = F.silu(_mm_ref(x, wg)) * _mm_ref(x, wu)
    assert_close_bf16(y, ref, what="swiglu")


def test_gemm_odd_ldc_lm_head_shape():
    ops = pkg("ops")
    M, N, K = 70, 32011, 128
    x, w = _rand(M, K, seed=10), _rand(N, K, seed=11, scale=K ** -0.5)
    y = ops.linear(x.to(DEV), w.to(DEV))
    assert_close_bf16(y, _mm_ref(x, w), what="odd N")


def test_gemm_f32_out_and_strided_x():
    ops = pkg("ops")
    x_full = _rand(90, 3 * 128, seed=12)
    w = _rand(64, 128, seed=13, scale=128 ** -0.5)
    xg = x_full.to(DEV)
    y = ops.linear(xg[:, 128:256], w.to(DEV), out_f32=True)
    ref = x_full[:, 128:256].float() @ w.float().t()
    assert y.dtype == torch.float32
    assert rel_err(y, ref) < 1e-5


@pytest.mark.parametrize("rows,D", [(5, 64), (33, 256), (300, 1024), (129, 1280), (1000, 4096)])
def test_rmsnorm_layernorm(rows, D):
    ops = pkg("ops")
    x, w, b = _rand(rows, D, seed=20, scale=3.0), (1 + 0.1 * torch.randn(D)).to(BF), (0.1 * torch.randn(D)).to(BF)
    y = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6)
    assert_close_bf16(y, O.rms_norm(x, w, 1e-6), ulps=1.0, what="rmsnorm")
    y = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    assert_close_bf16(y, F.layer_norm(x, (D,), w, b, 1e-5), ulps=1.0, what="layernorm")


@pytest.mark.parametrize("hd,H", [(128, 4), (16, 4)])
def test_rope(hd, H):
    ops = pkg("ops")
    B, S = 2, 37
    D = H * hd
    qkv = _rand(B * S, 3 * D, seed=30)
    pos = torch.stack([torch.arange(S), torch.arange(S) + 500])
    cos, sin = O.rope_tables(pos, hd, 10000.0, BF)
    q = qkv[:, :D].view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, D:2 * D].view(B, S, H, hd).transpose(1, 2)
    qr, kr = O.apply_rope(q, k, cos, sin)
    g = qkv.to(DEV)
    inv = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))).to(DEV)
    ops.rope_inplace(g, 3 * D, pos.reshape(-1).to(DEV), inv, B * S, 2 * H, hd)
    out = g.cpu()
    assert_close_bf16(out[:, :D].view(B, S, H, hd).transpose(1, 2), qr, ulps=1.0, what="rope q")
    assert_close_bf16(out[:, D:2 * D].view(B, S, H, hd).transpose(1, 2), kr, ulps=1.0, what="rope k")
    assert torch.equal(out[:, 2 * D:], qkv[:, 2 * D:]), "v must be untouched"


def test_rope_append_matches_rope_plus_copies():
    """decode-step RoPE + KV-cache append in one launch: q/k bit-equal to rope_inplace, cache rows/columns in place."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    B, S, H, hd, smax, past = 2, 3, 4, 128, 128, 37
    D = H * hd
    qkv = _rand(B * S, 3 * D, seed=46)
    pos = (torch.arange(S)[None] + past + torch.tensor([[0], [5]])).reshape(-1)
    inv = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))).to(DEV)
    a = qkv.to(DEV)
    ops.rope_inplace(a, 3 * D, pos.to(DEV), inv, B * S, 2 * H, hd)
    b = qkv.to(DEV)
    kc = torch.full((B, H, smax, hd), 7.0, device=DEV, dtype=BF)
    vtc = torch.full((B, H, hd, smax), 7.0, device=DEV, dtype=BF)
    ops.rope_append(b, 3 * D, pos.to(DEV), inv, B, S, H, hd, kc, vtc, smax, past)
    assert torch.equal(a[:, :D], b[:, :D]), "q"
    ar = a.view(B, S, 3, H, hd)
    for t in range(S):
        assert torch.equal(kc[:, :, past + t], ar[:, t, 1]), "k row"
        assert torch.equal(vtc[:, :, :, M_.KVCache.vt_slot(past + t)], ar[:, t, 2]), "v column"
    untouched = torch.ones(smax, dtype=torch.bool)
    untouched[past:past + S] = False
    assert bool((kc[:, :, untouched.to(DEV)] == 7.0).all())
    cols = torch.ones(smax, dtype=torch.bool)
    cols[[M_.KVCache.vt_slot(past + t) for t in range(S)]] = False
    assert bool((vtc[:, :, :, cols.to(DEV)] == 7.0).all())


@pytest.mark.parametrize("B,S,H,hd,K,dt", [(1, 1, 32, 128, 4096, BF), (2, 2, 4, 128, 512, BF), (3, 1, 6, 64, 256, torch.float16),
                                            (1, 3, 5, 32, 160, BF)])
def test_qkv_gemv_with_fused_rope_append_is_bit_equal(B, S, H, hd, K, dt):
    """decode step: ull_gemv_qkv_rope_append (RMSNorm prologue, RoPE + KV-cache append in the epilogue) == the three-launch path
    (linear with fused RMSNorm, rope_append): queries, K-cache rows and V^T-cache columns bit for bit; nothing else in the caches touched."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    D, T, smax, past = H * hd, B * S, 128, 45
    x = _rand(T, K, seed=47).to(dt).to(DEV)
    w = _rand(3 * D, K, seed=48, scale=K ** -0.5).to(dt).to(DEV)
    ln = (1.0 + 0.1 * _rand(K, seed=49).float()).to(dt).to(DEV)
    pos = (torch.arange(S)[None] + past + torch.arange(B)[:, None] * 3).reshape(-1).to(DEV)
    inv = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))).to(DEV)
    for rms in (ln, None):
        kc0 = torch.full((B, H, smax, hd), 7.0, device=DEV, dtype=dt)
        vt0 = torch.full((B, H, hd, smax), 7.0, device=DEV, dtype=dt)
        qkv = ops.linear(x, w, rms_w=rms, rms_eps=1e-5) if rms is not None else ops.linear(x, w)
        ops.rope_append(qkv, 3 * D, pos, inv, B, S, H, hd, kc0, vt0, smax, past)
        kc1, vt1 = torch.full_like(kc0, 7.0), torch.full_like(vt0, 7.0)
        cs, sn = ops.rope_table(pos, inv, dt)
        q = ops.linear_qkv_rope_append(x, w, cs, sn, B, S, H, hd, kc1, vt1, smax, past, rms_w=rms, rms_eps=1e-5)
        assert torch.equal(q, qkv[:, :D]), "rotated queries"
        assert torch.equal(kc1, kc0), "K cache"
        assert torch.equal(vt1, vt0), "V^T cache"
        assert bool((kc1[:, :, past:past + S] != 7.0).any()) and bool((vt1[:, :, :, M_.KVCache.vt_slot(past)] != 7.0).any())
    with pytest.raises(RuntimeError):
        ops.linear_qkv_rope_append(x, w, cs, sn, B, S, H, hd, kc1, vt1, smax, smax, rms_w=None)       # past + S > smax


def _attn_ref(q, k, v, scale, causal, key_mask):
    """eager attention on bf16 tensors: q,k,v [B,H,S,hd]."""
    w = torch.matmul(q, k.transpose(2, 3)) * scale
    B, H, Sq, Sk = w.shape
    if causal or key_mask is not None:
        allowed = torch.ones(B, 1, Sq, Sk, dtype=torch.bool)
        if causal:
            allowed = allowed & (torch.arange(Sk)[None, :] <= torch.arange(Sq)[:, None] + (Sk - Sq))[None, None]
        if key_mask is not None:
            allowed = allowed & (key_mask[:, None, None, :] != 0)
        w = w + torch.where(allowed, torch.zeros((), dtype=BF), torch.full((), torch.finfo(BF).min, dtype=BF))
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(BF)
    return torch.matmul(w, v)


@pytest.mark.parametrize("B,H,S,hd,causal,masked", [(2, 4, 11, 16, True, True), (1, 2, 5, 16, False, False), (2, 3, 257, 64, False, False),
                                                    (2, 2, 291, 128, True, True), (1, 2, 643, 128, True, False), (1, 1, 196, 80, False, False),
                                                    (1, 2, 1024, 128, True, False), (2, 2, 1500, 128, True, True),
                                                    (1, 1, 2048, 128, True, False)])
def test_attention(B, H, S, hd, causal, masked):
    ops = pkg("ops")
    D = H * hd
    qkv = _rand(B * S, 3 * D, seed=40 + S)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.int32)
        km[-1, S - S // 3:] = 0                      # right padding on the last sample
    q = qkv[:, :D].view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, D:2 * D].view(B, S, H, hd).transpose(1, 2)
    v = qkv[:, 2 * D:].view(B, S, H, hd).transpose(1, 2)
    ref = _attn_ref(q, k, v, hd ** -0.5, causal, km).transpose(1, 2).reshape(B * S, D)
    g = qkv.to(DEV)
    vt = ops.transpose_v(g[:, 2 * D:], S * 3 * D, 3 * D, B, S, H, hd)
    vt_nat = vt.cpu()[..., ops.vt_unpermute_index(vt.shape[-1])]       # undo the 32-key block permutation
    assert torch.equal(vt_nat[..., :S], v.transpose(2, 3)), "V^T"
    assert float(vt_nat[..., S:].abs().sum()) == 0.0, "V^T padding must be zero"
    out = torch.empty(B * S, D, device=DEV, dtype=BF)
    ops.attention(g, g[:, D:], vt, out, B, H, S, S, hd, (S * 3 * D, hd, 3 * D), (S * 3 * D, hd, 3 * D), (S * D, hd, D),
                  None if km is None else km.to(DEV), causal=causal, scale_mode=1, scale=hd ** -0.5)
    got = out.cpu()
    if masked:      # rows of padded queries are don't-care in the reference's callers too; compare the valid ones
        valid = km.bool().reshape(-1)
        got, ref = got[valid], ref[valid]
    assert_close_bf16(got, ref, ulps=2.0, what=f"attention S={S} hd={hd}", outlier_frac=1e-3, outlier_floor=float(v.float().abs().max()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,S,hd,causal,masked", [(2, 3, 643, 128, True, True), (1, 2, 1000, 128, True, False), (2, 2, 130, 128, True, False),
                                                    (2, 3, 577, 64, False, False), (3, 2, 257, 64, False, False), (1, 2, 50, 64, False, False),
                                                    (1, 2, 1500, 128, True, False), (1, 2, 196, 80, False, False), (2, 4, 11, 16, True, True),
                                                    # every CU busy with the 8-wave forms (a block's O rows must not be staged over a tile other waves still read)
                                                    (8, 32, 1000, 128, True, False), (32, 16, 257, 64, False, False)])
def test_attention_takes_v_as_rows(B, H, S, hd, causal, masked, dt):
    """V handed over as rows of the fused q|k|v buffer (C-ABI vt_len = 0): the LLaMA / CLIP prefill kernels read it through the
    transposing LDS load -- same operands in the same MFMA slots as with the V^T image, so the result is identical to the last bit;
    shapes without such a kernel (long rows, other head dims, few queries) get their V^T image made by the wrapper."""
    ops = pkg("ops")
    D = H * hd
    g = torch.Generator().manual_seed(S * 3 + hd)
    qkv = torch.randn(B * S, 3 * D, generator=g).to(dt).to(DEV)
    km = None
    if masked:
        km = torch.ones(B, S, dtype=torch.int32)
        km[-1, S - S // 3:] = 0
        km = km.to(DEV)
    st = (S * 3 * D, hd, 3 * D)
    vt = ops.transpose_v(qkv[:, 2 * D:], S * 3 * D, 3 * D, B, S, H, hd)
    a = torch.empty(B * S, D, device=DEV, dtype=dt)
    b = torch.full((B * S, D), float("nan"), device=DEV, dtype=dt)
    ops.attention(qkv, qkv[:, D:], vt, a, B, H, S, S, hd, st, st, (S * D, hd, D), km, causal=causal, scale_mode=1, scale=hd ** -0.5)
    ops.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], b, B, H, S, S, hd, st, st, (S * D, hd, D), km, causal=causal, scale_mode=1, scale=hd ** -0.5,
                  v_strides=st)
    assert torch.equal(a, b)


@pytest.mark.parametrize("Sq,Sk,left_pad", [(1, 70, 0), (1, 700, 13), (3, 1024, 0), (2, 2000, 100), (16, 4096, 0)])
def test_attention_decode_step_shapes(Sq, Sk, left_pad):
    """<= 16 new queries against a longer key cache (KV-cached generation incl. left-padded prompts; few-query split-key kernel up to
    4096 keys): causal offset Sk - Sq, key mask, K / V^T read by cache strides."""
    ops = pkg("ops")
    B, H, hd = 2, 4, 128
    D = H * hd
    q, k, v = _rand(B, H, Sq, hd, seed=60 + Sk), _rand(B, H, Sk, hd, seed=61 + Sk), _rand(B, H, Sk, hd, seed=62 + Sk)
    km = torch.ones(B, Sk, dtype=torch.int32)
    if left_pad:
        km[1, :left_pad] = 0
    ref = _attn_ref(q, k, v, hd ** -0.5, True, km)                                   # [B, H, Sq, hd]
    smax = ((Sk + 63) // 64) * 64 + 64
    kc = torch.zeros(B, H, smax, hd, dtype=BF)
    kc[:, :, :Sk] = k
    vsrc = v.transpose(1, 2).reshape(B * Sk, D).to(DEV)                              # [B*Sk, H*hd] token-major
    vtc = ops.transpose_v(vsrc, Sk * D, D, B, Sk, H, hd, pitch=smax)
    qd = q.transpose(1, 2).reshape(B * Sq, D).contiguous().to(DEV)                   # [B*Sq, H*hd]
    out = torch.empty(B * Sq, D, device=DEV, dtype=BF)
    ops.attention(qd, kc.to(DEV), vtc, out, B, H, Sq, Sk, hd, (Sq * D, hd, D), (H * smax * hd, smax * hd, hd), (Sq * D, hd, D), km.to(DEV),
                  causal=True, scale_mode=1, scale=hd ** -0.5)
    got = out.cpu().view(B, Sq, H, hd).transpose(1, 2)
    assert_close_bf16(got, ref, ulps=2.0, what=f"decode attention {Sq}x{Sk}", outlier_frac=2e-3, outlier_floor=float(v.float().abs().max()))


def test_patch_embed_and_clip_pre_ln():
    ops = pkg("ops")
    n, C, HW, ps, Dv = 3, 3, 56, 14, 64
    img = _rand(n, C, HW, HW, seed=50)
    w = _rand(Dv, C, ps, ps, seed=51, scale=(C * ps * ps) ** -0.5)
    cls, pos = _rand(Dv, seed=52), _rand((HW // ps) ** 2 + 1, Dv, seed=53)
    lw, lb = (1 + 0.1 * torch.randn(Dv)).to(BF), (0.1 * torch.randn(Dv)).to(BF)
    K = C * ps * ps
    Kp = ((K + 63) // 64) * 64
    cols = ops.im2col(img.to(DEV), ps, Kp)
    ref_cols = F.unfold(img.float(), ps, stride=ps).transpose(1, 2).reshape(-1, K).to(BF)
    assert torch.equal(cols[:, :K].cpu(), ref_cols) and float(cols[:, K:].abs().sum()) == 0.0
    wp = torch.zeros(Dv, Kp, dtype=BF)
    wp[:, :K] = w.reshape(Dv, K)
    patches = ops.linear(cols, wp.to(DEV))
    pe = F.conv2d(img, w, None, stride=ps).flatten(2).transpose(1, 2)
    assert_close_bf16(patches.view(n, -1, Dv), pe, what="patch conv")
    T = pe.shape[1] + 1
    h = ops.clip_embed_ln(patches, cls.to(DEV), pos.to(DEV), lw.to(DEV), lb.to(DEV), n, T, 1e-5)
    ref = F.layer_norm(torch.cat([cls.expand(n, 1, -1), patches.cpu().view(n, -1, Dv)], 1) + pos[None], (Dv,), lw, lb, 1e-5)
    assert_close_bf16(h, ref, ulps=1.0, what="clip embeddings + pre-LN")


def test_embed_splice_and_spans():
    ops = pkg("ops")
    V, D, P = 50, 64, 4
    table = _rand(V, D, seed=60)
    ids = torch.tensor([[1, 7, 8, 9, 10, 11, 12, 13, 14, 15],          # text only
                        [1, 40, 42, 42, 42, 42, 41, 5, 6, 7],          # image at pos 1
                        [1, 2, 43, 45, 45, 45, 45, 45, 44, 9],         # video (5 tokens) at pos 2
                        [1, 3, 3, 40, 42, 42, 42, 42, 41, 0]])         # image at pos 3
    img_feat = _rand(2, P + 1, D, seed=61)      # with a leading CLS row that must be skipped
    vid_feat = _rand(1, 5, D, seed=62)
    spans = ops.mm_spans(ids.to(DEV), 40, 41, 43, 44)
    assert spans.cpu().tolist() == [[0, -1, 0, 0], [1, 1, 0, 0], [2, 2, 0, 0], [1, 3, 1, 0]]
    out = ops.embed_splice(ids.to(DEV), table.to(DEV), img_feat.to(DEV), vid_feat.to(DEV), spans, P, P + 1, 1).cpu()
    ref = table[ids].clone()
    ref[1, 2:6] = img_feat[0, 1:]
    ref[2, 3:8] = vid_feat[0]
    ref[3, 4:8] = img_feat[1, 1:]
    assert torch.equal(out, ref)
    bad = torch.tensor([[1, 40, 42, 41, 41, 5]])
    assert int(ops.mm_spans(bad.to(DEV), 40, 41, 43, 44)[0, 3]) == 1      # start/end count mismatch is flagged


def test_video_pool():
    ops = pkg("ops")
    B, T, N, D = 2, 8, 4, 64
    f = _rand(B * T, N + 1, D, seed=70)
    out = ops.video_pool(f.to(DEV), B, T, N, tok_pitch=N + 1, tok_off=1).cpu()
    ff = f[:, 1:].reshape(B, T, N, D)
    ref = torch.concat([ff.mean(dim=2), ff.mean(dim=1)], dim=1)
    assert_close_bf16(out, ref, ulps=1.0, what="video pool")


def test_gather_and_add():
    ops = pkg("ops")
    src = _rand(40, 128, seed=80)
    idx = torch.tensor([3, 39, 0, 3])
    assert torch.equal(ops.gather_rows(src.to(DEV), idx.to(DEV)).cpu(), src[idx])
    a, b = _rand(6, 4, 128, seed=81), _rand(4, 128, seed=82)
    assert torch.equal(ops.add_rows(a.to(DEV), b.to(DEV)).cpu(), a + b)


def test_gemm_tile_major_weight_is_bit_identical():
    """ULL_EPI_W_TILED: the same GEMM from a tile-major copy of W (incl. a ragged last row tile and the stream-K tail)."""
    ops = pkg("ops")
    M, N, K = 1300, 1000, 320
    x, w, b = _rand(M, K, seed=51), _rand(N, K, seed=52, scale=K ** -0.5), _rand(N, seed=53)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    plain = ops.linear(xd, wd, bd, act="relu")
    ops.register_tiled(wd)
    assert ops._tiled_of(wd) is not None and tuple(ops._tiled_of(wd).shape) == (4, 5, 256, 64)
    tiled = ops.linear(xd, wd, bd, act="relu")
    assert torch.equal(plain, tiled)
    assert_close_bf16(tiled, F.relu(F.linear(x.float(), w.float(), b.float()).to(BF)), ulps=2.0, what="tiled gemm vs reference")
    small = ops.linear(xd[:8], wd, bd)                  # decode / small-M shapes keep using the row-major tensor
    assert_close_bf16(small, F.linear(x[:8].float(), w.float(), b.float()).to(BF), what="small-M with a registered tiled copy")
    ops._TILED.clear()


@pytest.mark.parametrize("n,C,HW,ps,Dv,bias", [(3, 3, 56, 14, 64, False), (2, 3, 64, 16, 160, True), (1, 3, 336, 14, 1024, False), (5, 3, 28, 14, 200, True)])
def test_fused_patchify_matches_conv2d(n, C, HW, ps, Dv, bias):
    """ull_patchify_bf16 (A tiles DMA'd from the pixels, kx padded to 16) == F.conv2d, and nothing behind the image is read."""
    ops = pkg("ops")
    img = _rand(n, C, HW, HW, seed=80 + HW)
    w = _rand(Dv, C, ps, ps, seed=81, scale=(C * ps * ps) ** -0.5)
    b = _rand(Dv, seed=82) if bias else None
    ref = F.conv2d(img.float(), w.float(), None if b is None else b.float(), stride=ps).flatten(2).transpose(1, 2).reshape(-1, Dv)
    wp = ops.pack_patch_weight(w.to(DEV))
    got = ops.patchify(img.to(DEV), wp, ps, None if b is None else b.to(DEV))
    assert_close_bf16(got, ref.to(BF), ulps=1.0, what="fused patchify vs conv2d")
    # poison the memory right behind a private copy of the image: the kernel must not consume it
    buf = torch.full((img.numel() + 64,), float("nan"), dtype=BF, device=DEV)
    buf[:img.numel()] = img.reshape(-1).to(DEV)
    got2 = ops.patchify(buf[:img.numel()].view(n, C, HW, HW), wp, ps, None if b is None else b.to(DEV))
    assert torch.equal(got, got2) and not bool(torch.isnan(got2.float()).any())


def test_attention_fuzz_shapes():
    """random self-attention shapes across the dispatch table (register kernels with 5 / 11 / 16 tiles, two-pass long kernel, few-query
    kernel; head dims 16..128 incl. 80; causal / padded / plain) against the eager bf16 reference."""
    ops = pkg("ops")
    rs = __import__("random").Random(77)
    for trial in range(24):
        hd = rs.choice([16, 32, 64, 80, 128])
        H, B = rs.choice([1, 2, 3]), rs.choice([1, 2])
        S = rs.choice([1, 2, 15, 16, 17, 63, 64, 65, 130, 257, 320, 321, 640, 704, 705, 1023, 1025, 1100])
        causal = rs.random() < 0.5
        masked = rs.random() < 0.4 and S > 4
        D = H * hd
        qkv = _rand(B * S, 3 * D, seed=1000 + trial)
        km = None
        if masked:
            km = torch.ones(B, S, dtype=torch.int32)
            km[-1, S - max(1, S // 4):] = 0
        q = qkv[:, :D].view(B, S, H, hd).transpose(1, 2)
        k = qkv[:, D:2 * D].view(B, S, H, hd).transpose(1, 2)
        v = qkv[:, 2 * D:].view(B, S, H, hd).transpose(1, 2)
        ref = _attn_ref(q, k, v, hd ** -0.5, causal, km).transpose(1, 2).reshape(B * S, D)
        g = qkv.to(DEV)
        vt = ops.transpose_v(g[:, 2 * D:], S * 3 * D, 3 * D, B, S, H, hd)
        out = torch.empty(B * S, D, device=DEV, dtype=BF)
        ops.attention(g, g[:, D:], vt, out, B, H, S, S, hd, (S * 3 * D, hd, 3 * D), (S * 3 * D, hd, 3 * D), (S * D, hd, D),
                      None if km is None else km.to(DEV), causal=causal, scale_mode=1, scale=hd ** -0.5)
        got = out.cpu()
        if masked:
            valid = km.bool().reshape(-1)
            got, ref = got[valid], ref[valid]
        assert_close_bf16(got, ref, ulps=2.0, what=f"attention fuzz B={B} H={H} S={S} hd={hd} causal={causal} masked={masked}", outlier_frac=2e-3,
                          outlier_floor=float(v.float().abs().max()))


def test_linear_fuzz_shapes_and_epilogues():
    """random Linear shapes over all three kernels (GEMV M <= 4, 128x128, 256x256 + stream-K tail) with ragged M / N, K that needs the
    zero-pad path, odd output pitch, and every epilogue combination (bias, quick_gelu / gelu / relu, residual, fp32 out, SwiGLU)."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    rs = __import__("random").Random(99)
    acts = {None: lambda t: t, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t), "gelu": F.gelu, "relu": F.relu}
    for trial in range(28):
        M = rs.choice([1, 3, 4, 5, 63, 129, 300, 1024, 1100, 1500, 2049])
        N = rs.choice([8, 40, 97, 128, 520, 777, 1031, 2048])
        K = rs.choice([64, 72, 192, 320, 1088])
        swiglu = rs.random() < 0.2
        if swiglu:
            N = rs.choice([32, 96, 352, 1024])
        x = _rand(M, K, seed=2000 + trial)
        w = _rand(2 * N if swiglu else N, K, seed=2100 + trial, scale=K ** -0.5)
        if swiglu:
            wg, wu = w[:N], w[N:]
            got = ops.linear(x.to(DEV), M_.interleave_gate_up(wg, wu).to(DEV), swiglu=True)
            ref = F.silu(_mm_ref(x, wg)) * _mm_ref(x, wu)
            assert_close_bf16(got, ref, what=f"fuzz swiglu M={M} N={N} K={K}")
            continue
        act = rs.choice([None, "quick_gelu", "gelu", "relu"])
        use_b, use_r, f32 = rs.random() < 0.6, rs.random() < 0.4, rs.random() < 0.25
        b = _rand(N, seed=2200 + trial) if use_b else None
        r = _rand(M, N, seed=2300 + trial) if use_r else None
        t = F.linear(x.float(), w.float(), None if b is None else b.float())
        if act is not None or use_r or not f32:
            t = t.to(BF)
        t = acts[act](t)
        if use_r:
            t = r + t.to(BF)
        got = ops.linear(x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), act=act, residual=None if r is None else r.to(DEV), out_f32=f32)
        assert got.dtype == (torch.float32 if f32 else BF) and tuple(got.shape) == (M, N)
        # a one-ulp flip inside the activation chain (fast exp vs libm) survives a cancelling residual add as an absolute error:
        # allow 1e-4 of the elements to meet the 2-ulp bound at the tensor's scale instead of their own
        want = t.float() if f32 else t.to(BF)
        assert_close_bf16(got, want, ulps=2.0, what=f"fuzz M={M} N={N} K={K} act={act} bias={use_b} res={use_r} f32={f32}", outlier_frac=1e-4,
                          outlier_floor=float(want.float().abs().max()))


def test_gemm_streamk_two_streams_concurrently():
    """Re-entrancy of the C-ABI (include/ullava_hip.h): two stream-K GEMMs in flight on two HIP streams use two caller-owned
    workspaces and must both equal their single-stream results bit for bit (a shared workspace corrupts one of them)."""
    ops = pkg("ops")
    M, N, K = 256 * 9, 256 * 32, 2048                      # 288 tiles: 32 tail tiles split 8 ways along K
    xs = [_rand(M, K, seed=51 + i).to(DEV) for i in range(2)]
    ws = [_rand(N, K, seed=61 + i, scale=K ** -0.5).to(DEV) for i in range(2)]
    alone = [ops.linear(xs[i], ws[i]) for i in range(2)]
    torch.cuda.synchronize()
    assert_close_bf16(alone[0][:300, :600], _mm_ref(xs[0][:300].cpu(), ws[0][:600].cpu()), what="stream-K GEMM vs oracle")
    assert_close_bf16(alone[1][-300:, -600:], _mm_ref(xs[1][-300:].cpu(), ws[1][-600:].cpu()), what="stream-K GEMM vs oracle")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[torch.empty_like(alone[i]) for _ in range(6)] for i in range(2)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for it in range(6):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                ops.linear(xs[i], ws[i], out=outs[i][it])
    torch.cuda.synchronize()
    assert len({k for k in ops._SK_WS if k[0] == 0}) >= 2, "each stream must own a workspace"
    for i in range(2):
        for it in range(6):
            assert torch.equal(outs[i][it], alone[i]), f"stream {i} iteration {it} differs from the single-stream result"


def test_gemm_without_workspace_never_splits():
    """ws = NULL is legal (no K-split) and gives the same values up to the fp32 summation order of the split."""
    ops = pkg("ops")
    M, N, K = 256 * 9, 256 * 32, 2048
    x, w = _rand(M, K, seed=71).to(DEV), _rand(N, K, seed=72, scale=K ** -0.5).to(DEV)
    with ops.streamk_policy(None):
        y0 = ops.linear(x, w)
    y1 = ops.linear(x, w)
    assert_close_bf16(y0, y1.cpu(), ulps=1.0, what="split vs unsplit")
    assert torch.equal(y0[: 256 * 8], y1[: 256 * 8])         # whole-tile rounds do not depend on the policy


@pytest.mark.parametrize("M,N,K,kind", [(323, 4096, 4096, "resid"), (291, 4096, 11008, "resid"), (643, 1024, 4096, "bias_gelu"),
                                          (130, 1024, 4096, "f32"), (323, 4096, 1024, "swiglu"), (200, 2176, 2048, "plain")])
def test_small_m_gemm_split_k_matches_reference_and_unsplit(M, N, K, kind):
    """single-image prefill shapes (M < 1024, few 128x128 tiles, long K): the 128x128 kernel cuts K into slices over the idle CUs, fp32
    slabs in the caller's workspace, summed in slice order by the finalize launch with the full epilogue (bias / activation / residual /
    SwiGLU / fp32 output).  Against the fp32 reference, against the unsplit launch (fp32 summation order only), and run to run."""
    ops = pkg("ops")
    x = _rand(M, K, seed=300 + M).to(DEV)
    w = _rand(N, K, seed=301 + N, scale=K ** -0.5)
    b = _rand(N, seed=302) if kind == "bias_gelu" else None
    r = _rand(M, N, seed=303) if kind == "resid" else None
    kw = dict(act="gelu" if kind == "bias_gelu" else None, residual=None if r is None else r.to(DEV), out_f32=kind == "f32")
    if kind == "swiglu":
        wd = pkg("modeling_core").interleave_gate_up(w[: N // 2].contiguous(), w[N // 2:].contiguous()).to(DEV)
        ref = (F.silu((x.cpu().float() @ w[: N // 2].float().t()).to(BF)).to(BF) * (x.cpu().float() @ w[N // 2:].float().t()).to(BF)).to(BF)
        run = lambda: ops.linear(x, wd, swiglu=True)
    else:
        wd = w.to(DEV)
        t = F.linear(x.cpu().float(), w.float(), None if b is None else b.float())
        if kind != "f32":
            t = t.to(BF)
        if kind == "bias_gelu":
            t = F.gelu(t)
        if r is not None:
            t = r + t.to(BF)
        ref = t
        run = lambda: ops.linear(x, wd, None if b is None else b.to(DEV), **kw)
    with ops.small_m_split_k(True):                     # opt-in latency mode (default off: batch invariance, see ops.py)
        y1, y2 = run(), run()
    y0 = run()
    assert torch.equal(y1, y2), "split-K must be deterministic"
    assert not torch.equal(y1, y0), "the split must actually have run (another summation order)"
    assert_close_bf16(y1, ref.float() if kind == "f32" else ref.to(BF), ulps=2.0, what=f"split-K {kind} vs reference", outlier_frac=1e-4,
                      outlier_floor=float(ref.float().abs().max()))
    assert_close_bf16(y1, y0.cpu(), ulps=2.0, what=f"split-K {kind} vs unsplit", outlier_frac=1e-4, outlier_floor=float(ref.float().abs().max()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,D", [(300, 256), (643 * 3, 512), (256 * 9 + 17, 1024 + 128)])
def test_qkv_gemm_with_fused_rope_is_bit_identical_to_gemm_then_rope(M, D, dt):
    """ull_gemm_qkv_rope == ull_gemm + ull_rope_inplace, bit for bit: 128x128 kernel, 256x256 kernel and its stream-K tail
    (finalize kernel); q and k columns rotated, v columns untouched; positions with repeats (batched sequences)."""
    ops = pkg("ops")
    hd, Hn = 128, D // 128
    K = 2048 if M > 2000 else 256
    g = torch.Generator().manual_seed(90 + M)
    x = torch.randn(M, K, generator=g).to(dt).to(DEV)
    w = (torch.randn(3 * D, K, generator=g) * K ** -0.5).to(dt).to(DEV)
    pos = (torch.arange(M) % 643).to(DEV)
    inv = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))).to(DEV)
    ref = ops.linear(x, w)
    v_before = ref[:, 2 * D:].clone()
    ops.rope_inplace(ref, 3 * D, pos, inv, M, 2 * Hn, hd)
    cs, sn = ops.rope_table(pos, inv, dt)
    got = ops.linear_qkv_rope(x, w, cs, sn, 2 * D, hd)
    assert torch.equal(got[:, 2 * D:], v_before), "v columns must pass through"
    assert torch.equal(got, ref), f"fused RoPE differs: {int((got != ref).sum())} elements"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind", ["plain", "bias_gelu_resid", "swiglu", "f32", "f32_bias_relu", "rope", "tiled_w"])
def test_gemm_256_tile_on_4_and_8_waves_agree(kind, dt):
    """The two forms of the 256x256 kernel (4 waves of 128x128 with AGPR accumulators, 8 waves of 128x64) are forced through every
    epilogue on shapes with ragged edges, a stream-K tail and a long K: bit-identical to each other, and checked against fp32 math."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    for (M, N, K) in ((256 * 5 + 37, 256 * 3 + 128, 512), (256 * 41 + 5, 256 * 7, 1024), (1029, 1024, 11008)):
        g = torch.Generator().manual_seed(M + K)
        x = torch.randn(M, K, generator=g).to(dt).to(DEV)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).to(DEV)
        b = torch.randn(N, generator=g).to(dt).to(DEV)
        r = torch.randn(M, N, generator=g).to(dt).to(DEV)
        outs = []
        for tune in (ops.GEMM_TUNE_WAVES4, ops.GEMM_TUNE_WAVES8):
            if kind == "plain":
                y = ops.linear(x, w, tune=tune)
            elif kind == "bias_gelu_resid":
                y = ops.linear(x, w, b, act="gelu", residual=r, tune=tune)
            elif kind == "swiglu":
                y = ops.linear(x, w, swiglu=True, tune=tune)
            elif kind == "f32":
                y = ops.linear(x, w, out_f32=True, tune=tune)
            elif kind == "f32_bias_relu":
                y = ops.linear(x, w, b, act="relu", out_f32=True, tune=tune)
            elif kind == "rope":
                pos = (torch.arange(M) % 643).to(DEV)
                inv = (1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).to(DEV)
                cs, sn = ops.rope_table(pos, inv, dt)
                y = ops.linear_qkv_rope(x, w, cs, sn, (N // 128 // 2) * 128 or 128, 128, tune=tune)
            else:
                ops.register_tiled(w)
                y = ops.linear(x, w, b, tune=tune)
            outs.append(y)
        for o in outs[1:]:
            assert torch.equal(outs[0], o), f"{kind} {M}x{N}x{K}: {int((outs[0] != o).sum())} elements differ between the kernel forms"
        if kind == "plain":
            ref = x.float() @ w.float().t()
            assert float((outs[0].float() - ref).abs().max()) <= 2.0 ** (-8 if dt == torch.bfloat16 else -11) * float(ref.abs().max()) * 1.01
        elif kind == "f32":
            ref = x.float() @ w.float().t()
            assert float((outs[0] - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gelu_erf_over_every_16_bit_input(dt):
    """The activation sees 16-bit inputs only, so it is checked on ALL of them against torch's CPU kernel of the same dtype: results may
    differ only where the reference's own 1 + erf(x / sqrt 2) has cancelled (x < -3, |GELU| < 3e-3), and only in a few dozen inputs."""
    ops = pkg("ops")
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(dt)
    keep = torch.isfinite(x.float()) & (x.float().abs() > 1e-4 if dt == torch.float16 else x.float().abs() > 1e-30) & (x.float().abs() < 1e30)
    x = x[keep].contiguous()
    ref = torch.nn.functional.gelu(x)
    got = ops.gelu_fwd(x.to(DEV)).cpu()
    bad = got != ref
    nbad = int(bad.sum())
    print(f"GELU(erf) {dt}: {nbad} of {x.numel()} inputs differ from torch CPU")
    assert nbad <= (64 if dt == torch.bfloat16 else 400), nbad
    if nbad and dt == torch.bfloat16:
        assert float(x[bad].float().max()) < -2.5 and float(ref[bad].float().abs().max()) < 4e-3
        assert float((got[bad].float() - ref[bad].float()).abs().max()) < 4e-5
    elif nbad:
        # fp16 keeps 11 bits: a 1e-7 relative difference flips a rounding now and then anywhere -- by one unit in the last place
        ulp = torch.maximum(ref[bad].float().abs(), torch.tensor(6.2e-5)) * 2.0 ** -10
        tail = x[bad].float() < -2.5
        assert bool(((got[bad].float() - ref[bad].float()).abs() <= torch.where(tail, torch.tensor(4e-5), ulp * 1.01)).all())


@pytest.mark.parametrize("name,dt", [("g6_per_op_bf16.pt", torch.bfloat16), ("g6_per_op_fp16.pt", torch.float16)])
def test_per_op_fixture_g6_real_dims(name, dt):
    """G6: the HIP ops against the REFERENCE modules' outputs at BASELINE dims (strided samples from the fixture; inputs regenerated
    from the seed).  Elementwise chains with identical rounding points must reproduce the reference to the last bit except where a
    transcendental (exp / cos / sin / rsqrt) differs in its last fp32 bit between the CPU's and the GPU's library."""
    from helpers import per_op_inputs, load_fixture
    ops, M_ = pkg("ops"), pkg("modeling_core")
    fx = load_fixture(name)
    x, out = per_op_inputs(fx["seed"], dt), fx["outputs"]
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10

    def check(got, key, max_flip_frac, excuse=None, ulps=1.0, floor_frac=1e-3):
        """excuse: boolean mask (same sampling) of elements that may differ freely within the bound (GELU's cancelled tail).  ulps: 2 for
        chains with a rounding behind the transcendental (a flipped sigmoid, multiplied and rounded again, can land two units away)."""
        d = out[key]
        got = got.cpu().contiguous().view(-1)[::d["step"]]
        ref = d["sample"]
        assert got.shape == ref.shape, (key, got.shape, ref.shape)
        diff = got != ref
        err = (got.float() - ref.float()).abs()
        bound = ref.float().abs().clamp_min(float(ref.float().abs().max()) * floor_frac) * ulp * 1.01 * ulps
        print(f"G6 {dt} {key:18s}: {float(diff.float().mean()) * 100:.4f} % of elements differ from the reference module's output")
        if excuse is not None:
            ex = excuse.contiguous().view(-1)[::d["step"]]
            diff = diff & ~ex
            assert bool((err[ex] <= 4e-5 + 16 * bound[ex]).all()), key          # excused: tiny values, still close in absolute terms
            err = err[~ex]; bound = bound[~ex]
        assert bool((err <= bound).all()), (key, float((err / bound).max()))
        assert float(diff.float().mean()) <= max_flip_frac, (key, float(diff.float().mean()))

    check(ops.rmsnorm(x["rms_x"].to(DEV), x["rms_w"].to(DEV), 1e-6), "rmsnorm", 2e-3)
    # RoPE: [1, H, S, hd] -> token-major [S, H * hd] rows, rotated in place
    inv = (1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float) / 128))).to(DEV)
    pos = torch.arange(1024, dtype=torch.int64, device=DEV)
    for key, src in (("rope_q", "rope_q"), ("rope_k", "rope_k")):
        t = x[src][0].permute(1, 0, 2).reshape(1024, 256).contiguous().to(DEV)
        ops.rope_inplace(t, 256, pos, inv, 1024, 2, 128)
        check(t.view(1024, 2, 128).permute(1, 0, 2).unsqueeze(0), key, 5e-3)
    gu = M_.interleave_gate_up(x["gate"].t().contiguous(), x["up"].t().contiguous()).t().contiguous()       # columns interleaved in 16-groups
    check(ops.swiglu_fwd(gu.to(DEV)), "swiglu", 2e-3, ulps=2.0)
    eye = torch.eye(1024, dtype=dt, device=DEV)
    xa = x["act_x"].to(DEV)
    xa_big = torch.cat([xa] * 7)[:1024 + 96]                  # >= 1024 rows: the 256x256 kernel's epilogue
    for rows in (xa, xa_big):
        got = ops.linear(rows, eye, act="quick_gelu")[:160]
        # (fp16: sigmoid(1.702 t) is a subnormal half below t ~ -5.7 and coarse well before: one unit there is percents of the product)
        check(got, "quick_gelu", 2e-3, ulps=2.0, excuse=(x["act_x"].float() < -4.0) if dt == torch.float16 else None)
        check(ops.linear(rows, eye, act="gelu")[:160], "gelu", 2e-3, excuse=x["act_x"].float() < -2.5)
    # (x < -2.5: 1 + erf(x / sqrt 2) has cancelled in the reference's own fp32 evaluation; see test_gelu_erf_over_every_16_bit_input)
    check(ops.gelu_fwd(xa), "gelu", 2e-3, excuse=x["act_x"].float() < -2.5)
    wp = ops.pack_patch_weight(x["patch_w"].to(DEV))
    for isz, px, posk in ((224, "px224", "pos224"), (336, "px336", "pos336")):
        n = x[px].shape[0]
        patches = ops.patchify(x[px].to(DEV), wp, 14)
        tokens = (isz // 14) ** 2 + 1
        got = ops.clip_embed_ln(patches, x["cls"].to(DEV), x[posk].to(DEV), x["ln_w"].to(DEV), x["ln_b"].to(DEV), n, tokens, 1e-5)
        # LayerNorm subtracts the row mean: a one-unit flip of an input near the mean is many units of the output -- absolute bound
        check(got, f"clip_embed_ln_{isz}", 2e-2, floor_frac=0.25, ulps=2.0)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("R,n,scale", [(2, 8, 0.5), (3, 1001, 1.0 / 3), (8, 4099, 0.125), (7, 262147, 1.0 / 7), (1, 77, 1.0)])
def test_sum_slabs_against_fp32_sum(dt, R, n, scale):
    """`ull_sum_slabs` (the local reduction of the direct-exchange gradient all-reduce, dist.allreduce_gradients): out[i] =
    rnd(scale * sum_r x[r, i]) with fp32 accumulation in slab order -- bit-exact against torch's fp32 sum over the same order for
    R <= 8 (odd n, every world size of one node)."""
    ops = pkg("ops")
    g = torch.Generator().manual_seed(R * 1000 + n)
    x = torch.randn(R, n, generator=g).to(dt)
    want = torch.zeros(n, dtype=torch.float32)
    for r in range(R):                                       # the kernel's order: slab 0, 1, ... accumulated in fp32
        want = want + x[r].float()
    want = (want * torch.tensor(scale, dtype=torch.float32)).to(dt)
    got = ops.sum_slabs(x.to(DEV), scale)
    assert got.dtype == dt and tuple(got.shape) == (n,)
    assert torch.equal(got.cpu(), want), float((got.cpu().float() - want.float()).abs().max())
    with pytest.raises(RuntimeError):
        ops.sum_slabs(x.float().to(DEV), scale)              # no fp32 build: allreduce_gradients routes fp32 buckets to all_reduce


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_greedy_step_kernel_matches_torch_bookkeeping(dt):
    """ull_greedy_step (one launch per generated token) against the torch ops it replaces in HF-style greedy search: argmax with the
    FIRST index on ties (16-bit logits tie often), pad fill of finished rows, EOS tracking, append, the unfinished-row counter."""
    ops = pkg("ops")
    g = torch.Generator().manual_seed(9)
    B, V, L = 7, 32011, 12
    logits = (torch.randn(B, 3, V, generator=g) * 2).to(dt)
    logits[0, -1, 100] = logits[0, -1].max() + 1                 # a clear winner
    logits[1, -1, [5, 17000, 32010]] = logits[1, -1].float().max().item() + 2   # a three-way tie: index 5 wins
    logits[2, -1, :] = 0.5                                        # every entry ties: index 0
    eos = torch.tensor([2, 100], dtype=torch.int64)
    unfinished = torch.tensor([1, 1, 0, 1, 1, 0, 1], dtype=torch.int32)
    seq = torch.full((B, L), -7, dtype=torch.int64)
    want_tok = logits[:, -1].float().argmax(-1)
    assert int(want_tok[1]) == 5 and int(want_tok[2]) == 0
    pad = 31999
    want_tok = torch.where(unfinished.bool(), want_tok, torch.full_like(want_tok, pad))
    want_unf = unfinished.bool() & ~torch.isin(want_tok, eos)
    ld = logits.to(DEV)
    u_d, s_d, alive = unfinished.to(DEV), seq.to(DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.greedy_step(ld[:, -1], u_d, eos.to(DEV), pad, s_d, 4, alive)
    assert torch.equal(s_d[:, 4].cpu(), want_tok) and torch.equal(u_d.cpu().bool(), want_unf) and int(alive) == int(want_unf.sum())
    assert bool((s_d.cpu()[:, :4] == -7).all()) and bool((s_d.cpu()[:, 5:] == -7).all())       # only column `pos` is written
    # no pad id: finished rows keep the argmax; no EOS list: nothing finishes
    u2, s2, a2 = unfinished.to(DEV), seq.to(DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.greedy_step(ld[:, -1], u2, None, None, s2, 0, a2)
    assert torch.equal(s2[:, 0].cpu(), logits[:, -1].float().argmax(-1)) and torch.equal(u2.cpu(), unfinished) and int(a2) == int(unfinished.sum())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_relu_mask_and_swiglu_halves(dt):
    """The two small kernels the full-parameter training path added: ReLU backward as a selection, and SwiGLU on [gate | up] column halves
    (the layout of the shared gate|up buffer) -- bit-identical to the 16-column interleave of the inference pack, forward and backward."""
    ops, M_ = pkg("ops"), pkg("modeling_core")
    g = torch.Generator().manual_seed(4)
    y, dy = torch.randn(37, 129, generator=g).to(dt), torch.randn(37, 129, generator=g).to(dt)
    y[0, :5] = 0
    got = ops.relu_mask(y.to(DEV), dy.to(DEV)).cpu()
    assert torch.equal(got, torch.where(y > 0, dy, torch.zeros_like(dy)))
    Mr, I = 19, 64
    gate, up, da = (torch.randn(Mr, I, generator=g).to(dt) for _ in range(3))
    halves = torch.cat([gate, up], dim=1).to(DEV)
    inter = M_.interleave_gate_up(gate.t().contiguous(), up.t().contiguous()).t().contiguous().to(DEV)      # columns in 16-wide [gate | up] groups
    a_h, a_i = ops.swiglu_fwd(halves, halves=True), ops.swiglu_fwd(inter)
    assert torch.equal(a_h, a_i)
    d_h, d_i = ops.swiglu_bwd(halves, da.to(DEV), halves=True).cpu(), ops.swiglu_bwd(inter, da.to(DEV)).cpu()
    dg_i = d_i.view(Mr, I // 16, 2, 16)
    assert torch.equal(d_h[:, :I], dg_i[:, :, 0].reshape(Mr, I)) and torch.equal(d_h[:, I:], dg_i[:, :, 1].reshape(Mr, I))


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_direct_activation_epilogue_matches_the_staged_one_over_the_value_range(act):
    """QuickGELU / GELU(erf) in the 4-wave kernel's register-direct epilogue against the 8-wave kernel's LDS-staged one: same bits for
    Linear outputs that span tiny values, |x| >= 8, exact zeros and infinities.  (A table-lookup form of the activations -- window in LDS,
    full 65536-entry table in L2, bit-identical by construction -- was built on this test and measured SLOWER than the arithmetic: SAM fc1
    509 vs 431 us, CLIP fc1 227 vs 170: eight dependent 2-byte LDS reads and their index arithmetic per 16 bytes of output cost more
    issue slots than 12.5 VALU instructions per element.  Not shipped.)"""
    ops = pkg("ops")
    g = torch.Generator().manual_seed(21)
    M, N, K = 2048 + 40, 1024, 128
    for scale in (1e-4, 0.05, 1.0, 40.0):
        x = (torch.randn(M, K, generator=g) * scale).to(BF)
        x[:8] = 0                                                    # rows of exact zeros (output = bias)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF)
        b = (torch.randn(N, generator=g) * scale).to(BF)
        b[:4] = 0
        if scale == 40.0:
            x[8, :] = 3e38                                           # overflows to +-inf in the product
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        y4 = ops.linear(xd, wd, bd, act=act, tune=ops.GEMM_TUNE_WAVES4)
        y8 = ops.linear(xd, wd, bd, act=act, tune=ops.GEMM_TUNE_WAVES8)
        same = (y4.view(torch.int16) == y8.view(torch.int16)) | (torch.isnan(y4) & torch.isnan(y8))
        assert bool(same.all()), (act, scale, int((~same).sum()))
        y4n = ops.linear(xd, wd, None, act=act, tune=ops.GEMM_TUNE_WAVES4)
        y8n = ops.linear(xd, wd, None, act=act, tune=ops.GEMM_TUNE_WAVES8)
        samen = (y4n.view(torch.int16) == y8n.view(torch.int16)) | (torch.isnan(y4n) & torch.isnan(y8n))
        assert bool(samen.all()), (act, scale, "no bias")
