import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("u-llava_amd.ops")
for dt in (torch.float16, torch.bfloat16):
    hd, T = 128, 2048
    x = torch.zeros(T, hd, dtype=dt); x[:, :hd // 2] = 1.0
    pos = torch.arange(T)
    inv = (1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float) / hd)))
    ang = pos[:, None].float() * inv[None, :]
    buf = x.cuda()
    ops.rope_inplace(buf, hd, pos.cuda(), inv.cuda(), T, 1, hd)
    got = buf.cpu()
    c_ref, s_ref = ang.cos().to(dt), ang.sin().to(dt)
    cg, sg = ang.cuda().cos().cpu().to(dt), ang.cuda().sin().cpu().to(dt)
    print(dt, "kernel cos != cpu:", int((got[:, :64] != c_ref).sum()), "sin:", int((got[:, 64:] != s_ref).sum()), "of", c_ref.numel(),
          "| torch-gpu cos != cpu:", int((cg != c_ref).sum()), int((sg != s_ref).sum()))
    d = (got[:, :64].float() - c_ref.float()).abs()
    i = d.argmax(); print("  worst:", float(d.max()), "angle", float(ang.flatten()[i]), float(got[:, :64].flatten()[i]), float(c_ref.flatten()[i]))
