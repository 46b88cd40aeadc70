import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("u-llava_amd.ops")
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K in ((1029, 4096, 11008), (1024, 512, 128), (1024, 512, 256), (2048, 1024, 512)):
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
    y4 = ops.linear(x, w)
    y8 = ops.linear(x, w, tune=1 << 21)
    ref = (x.float() @ w.float().t())
    bad = (y4 != y8)
    print(M, N, K, "mismatch frac", float(bad.float().mean()), "max|y4-ref|", float((y4.float() - ref).abs().max()), "max|y8-ref|", float((y8.float() - ref).abs().max()))
    if bad.any():
        idx = bad.nonzero()
        mm, nn = idx[:, 0] % 256, idx[:, 1] % 256
        hm = torch.zeros(16, 16, dtype=torch.long)
        for a, b in zip((mm // 16).tolist(), (nn // 16).tolist()):
            hm[a, b] += 1
        print("mismatches by (m%256//16 rows, n%256//16 cols):\n", hm)
        print("tiles (m//256, n//256) with mismatches:", sorted(set(zip((idx[:, 0] // 256).tolist(), (idx[:, 1] // 256).tolist())))[:20])
