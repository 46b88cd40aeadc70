"""GPU: device pre/post-processing (u-llava_amd/preprocess.py -> ull_resample_u8 / ull_u8_lut_chw / ull_mask_iou_counts) against
the committed fixtures and the CPU oracle.  Byte / integer work: everything must be bit-exact."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from helpers import pkg, load_fixture, ROOT
from oracle import preprocess_oracle as P

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from synth import synth_image  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
sha = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()


def test_clip_processor_fixtures():
    pre = pkg("preprocess")
    fx = load_fixture("p1_preprocess.pt")
    for c in fx["clip"]:
        img = torch.from_numpy(synth_image(*c["image_hw"], c["seed"])).to(DEV)
        out = pre.CLIPProcessor(size=c["size"], aspect_ratio=c["aspect_ratio"], device=DEV)(img)
        assert out.dtype == torch.float32 and tuple(out.shape) == (3, c["size"], c["size"])
        assert sha(out) == c["sha256"], (c["size"], c["image_hw"], c["aspect_ratio"])


def test_seg_toolbox_fixtures():
    pre = pkg("preprocess")
    fx = load_fixture("p1_preprocess.pt")
    for c in fx["sam"]:
        tb = pre.SegToolBox(device=DEV, sam_size=c["long_side"])
        img = torch.from_numpy(synth_image(*c["image_hw"], c["seed"])).to(DEV)
        r = tb.apply_image(img)
        assert tuple(r.shape[:2]) == tuple(c["resized_hw"]) and sha(r) == c["sha256_resized"]
        assert sha(tb.preprocess(r)) == c["sha256"]
        assert sha(tb.preprocess(r, dtype=torch.bfloat16).view(torch.int16)) == c["sha256_bf16"]


@pytest.mark.parametrize("kind", ["bilinear", "bicubic"])
def test_resize_random_sizes_against_oracle(kind):
    pre = pkg("preprocess")
    rs = np.random.RandomState(11)
    for _ in range(10):
        h, w, oh, ow = rs.randint(4, 300), rs.randint(4, 300), rs.randint(2, 260), rs.randint(2, 260)
        img = (rs.rand(h, w, 3) * 255).astype(np.uint8)
        got = pre.resize_u8(torch.from_numpy(img).to(DEV), (oh, ow), kind).cpu().numpy()
        assert np.array_equal(got, P.pil_resize(img, (oh, ow), kind)), (h, w, oh, ow)
    same = torch.from_numpy(img).to(DEV)
    assert torch.equal(pre.resize_u8(same, img.shape[:2], kind), same)                 # both passes skipped -> copy


def test_full_size_photo_roundtrip_properties():
    """production sizes (12 MP photo -> SAM 1024, CLIP 336): constant images stay constant through every pass, and the result equals
    the oracle on a strip of the image (the oracle's python loops are too slow for the whole photo)."""
    pre = pkg("preprocess")
    for val in (0, 37, 255):
        img = torch.full((3024, 4032, 3), val, dtype=torch.uint8, device=DEV)
        r = pre.SegToolBox(device=DEV).apply_image(img)
        assert tuple(r.shape) == (768, 1024, 3) and bool((r == val).all())
    rs = np.random.RandomState(3)
    photo = (rs.rand(1200, 1600, 3) * 255).astype(np.uint8)
    tb = pre.SegToolBox(device=DEV)
    r = tb.apply_image(torch.from_numpy(photo).to(DEV)).cpu().numpy()
    want = P.pil_resize(photo, (768, 1024), P.BILINEAR)
    assert np.array_equal(r, want)
    x = tb.preprocess(torch.from_numpy(r).to(DEV))
    assert tuple(x.shape) == (3, 1024, 1024) and bool((x[:, 768:] == 0).all())
    ref, _ = P.sam_preprocess(photo)
    assert np.array_equal(x.cpu().numpy(), ref)


def test_mask_iou_fixtures_and_oracle():
    pre = pkg("preprocess")
    fx = load_fixture("p1_preprocess.pt")
    for c in fx["iou"]:
        i, u, a = pre.mask_iou_stats(c["logits"].to(DEV), c["target"].to(DEV))
        assert torch.equal(i.cpu(), c["intersection"]) and torch.equal(u.cpu(), c["union"])
        torch.testing.assert_close(a.cpu(), c["acc_iou"], rtol=1e-6, atol=0)
        ai, au, at = pre.intersectionAndUnionGPU(c["logits"][0].to(DEV), c["target"][0].to(DEV))
        oi, ou, ot = P.intersection_and_union((c["logits"][0].numpy() > 0).astype(np.int32), c["target"][0].numpy())
        assert np.array_equal(ai.cpu().numpy(), oi) and np.array_equal(au.cpu().numpy(), ou) and np.array_equal(at.cpu().numpy(), ot)
    # full-resolution masks: counts are exact integers whatever the reduction order
    rs = np.random.RandomState(5)
    lg = torch.from_numpy(rs.randn(4, 1080, 1920).astype(np.float32)).to(DEV)
    tg = torch.from_numpy((rs.rand(4, 1080, 1920) > 0.5).astype(np.uint8)).to(DEV)
    cnt = pre.mask_iou_counts(lg, tg).cpu()
    o = lg > 0
    for m in range(4):
        t1 = tg[m] == 1
        want = [int((~o[m] & ~t1).sum()), int((o[m] & t1).sum()), int((~o[m]).sum()), int(o[m].sum()), int((~t1).sum()), int(t1.sum())]
        assert cnt[m].tolist() == want
