"""CPU: the pre/post-processing oracle (oracle/preprocess_oracle.py) against the committed fixtures produced by transformers'
CLIPImageProcessor / Pillow / the reference's torch formulas (tests/golden/gen_golden_preprocess.py), and -- where Pillow is
importable -- against Pillow directly on random sizes; plus the host-side tap generator of the product path."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import pkg, load_fixture, ROOT
from oracle import preprocess_oracle as P

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from synth import synth_image  # noqa: E402

sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_oracle_clip_branch_matches_clipimageprocessor_fixtures():
    fx = load_fixture("p1_preprocess.pt")
    for c in fx["clip"]:
        img = synth_image(*c["image_hw"], c["seed"])
        got = P.clip_preprocess(img, c["size"], c["aspect_ratio"])
        assert got.dtype == np.float32 and sha(got) == c["sha256"], (c["size"], c["image_hw"], c["aspect_ratio"])
        if "pixel_values" in c:
            assert np.array_equal(got, c["pixel_values"].numpy())


def test_oracle_sam_branch_matches_pillow_and_torch_fixtures():
    fx = load_fixture("p1_preprocess.pt")
    for c in fx["sam"]:
        img = synth_image(*c["image_hw"], c["seed"])
        got, hw = P.sam_preprocess(img, c["long_side"])
        assert tuple(hw) == tuple(c["resized_hw"])
        assert sha(got) == c["sha256"], c["image_hw"]
        assert sha(torch.from_numpy(got).to(torch.bfloat16).view(torch.int16).numpy()) == c["sha256_bf16"]
        if "pixel_values" in c:
            assert np.array_equal(got, c["pixel_values"].numpy())
            r = P.pil_resize(img, hw, P.BILINEAR)
            assert np.array_equal(r, c["resized"].numpy())


def test_oracle_iou_matches_reference_torch_ops():
    fx = load_fixture("p1_preprocess.pt")
    for c in fx["iou"]:
        i, u, a = P.mask_iou_stats(c["logits"].numpy(), c["target"].numpy())
        assert np.array_equal(i, c["intersection"].numpy()) and np.array_equal(u, c["union"].numpy())
        np.testing.assert_allclose(a, c["acc_iou"].numpy(), rtol=1e-6)


def test_oracle_resize_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(7)
    for _ in range(12):
        h, w, oh, ow = rs.randint(4, 260), rs.randint(4, 260), rs.randint(2, 200), rs.randint(2, 200)
        img = (rs.rand(h, w, 3) * 255).astype(np.uint8)
        for kind, res in ((P.BILINEAR, Image.BILINEAR), (P.BICUBIC, Image.BICUBIC)):
            ref = np.array(Image.fromarray(img).resize((ow, oh), res))
            assert np.array_equal(P.pil_resize(img, (oh, ow), kind), ref), (h, w, oh, ow, kind)


def test_host_tap_generator_equals_oracle():
    """the product path computes Pillow's taps itself (u-llava_amd/preprocess.py); same integers as the oracle's restatement."""
    pre = pkg("preprocess")
    for in_size, out_size in ((640, 1024), (1024, 97), (97, 224), (500, 336), (3, 7), (224, 224)):
        for kind in ("bilinear", "bicubic"):
            b, k = pre._taps_host(in_size, out_size, kind)
            ob, ok = P.resample_coeffs(in_size, out_size, kind)
            assert np.array_equal(b, ob) and np.array_equal(k, ok)
    lut = pre.CLIPProcessor.__new__(pre.CLIPProcessor)            # LUT construction without touching a device
    assert np.array_equal(P.clip_lut(), P.clip_lut(pre.OPENAI_CLIP_MEAN, pre.OPENAI_CLIP_STD))
