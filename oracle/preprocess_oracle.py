"""CPU restatement of the image pre/post-processing either side of the u-LLaVA forward path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/, tests/golden/gen_golden_preprocess.py and nothing else.

What the reference does (all on the host CPU, per image, inside the dataloader / inference scripts):
  * CLIP branch  -- dataset/processors/clip_processor.py:82-95: optional pad-to-square (pad_pil, :35-52), then transformers
    CLIPImageProcessor.preprocess = PIL bicubic resize (shortest edge) -> center crop -> x/255 -> (x - mean) / std, fp32 CHW.
  * SAM branch   -- models/segment_anything/utils/transforms.py:27-35 (ResizeLongestSide.apply_image = torchvision
    resize(to_pil_image(img)) = PIL bilinear resize of the uint8 image) and dataset/tools/mask_toolbox.py:15-25
    ((x - mean) / std on the uint8->fp32 tensor, zero pad right/bottom to 1024).
  * evaluation   -- trainers/ullava_trainer.py:40-52 + evaluation/tools.py:29-41: (mask logits > 0) and the class-wise
    intersection / union pixel counts (intersectionAndUnionGPU, K = 2, ignore_index = 255).

The resampling algorithm lives in a third-party dependency that is not part of /root/reference: Pillow's libImaging/Resample.c
(separable convolution with 22-bit fixed-point coefficients, horizontal pass then vertical pass, uint8 intermediate).  It is
restated here from its published algorithm and pinned against the Pillow installed in this image (12.2.0) by
tests/test_preprocess_cpu.py on random sizes (up- and down-scaling, both filters), and against transformers' CLIPImageProcessor
(PIL backend) through the committed fixtures tests/golden/p*_preprocess.pt.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8 bits of pixel, 2 bits of headroom for the accumulation
BILINEAR, BICUBIC = "bilinear", "bicubic"
_SUPPORT = {BILINEAR: 1.0, BICUBIC: 2.0}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
SAM_MEAN = (123.675, 116.28, 103.53)
SAM_STD = (58.395, 57.12, 57.375)


def _filter(kind, x):
    x = -x if x < 0.0 else x
    if kind == BILINEAR:
        return 1.0 - x if x < 1.0 else 0.0
    a = -0.5
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int, kind: str):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box (0, in_size).
    -> (bounds int32 [out, 2] = (first input index, tap count), coeffs int32 [out, ksize])."""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = float(np.float32(in1 - in0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = _SUPPORT[kind] * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_filter(kind, (x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            v = w * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, kind: str, axis: int) -> np.ndarray:
    """one separable pass over `axis` (0 = vertical, 1 = horizontal) of a uint8 H x W x C image."""
    bounds, kk = resample_coeffs(img.shape[axis], out_size, kind)
    src = img.astype(np.int64)
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.empty(shape, np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(shape[:axis] + shape[axis + 1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += np.take(src, xmin + x, axis=axis) * int(kk[xx, x])
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)          # clip8: arithmetic shift, then clamp
        if axis == 0:
            out[xx] = v
        else:
            out[:, xx] = v
    return out


def pil_resize(img: np.ndarray, out_hw, kind: str) -> np.ndarray:
    """PIL.Image.resize((w, h), resample) of a uint8 H x W x C array: horizontal pass first, then vertical; a pass whose size
    does not change is skipped (Resample.c ImagingResample: need_horizontal / need_vertical)."""
    oh, ow = int(out_hw[0]), int(out_hw[1])
    x = img
    if ow != x.shape[1]:
        x = _pass(x, ow, kind, 1)
    if oh != x.shape[0]:
        x = _pass(x, oh, kind, 0)
    return x.copy() if x is img else x


# ---- CLIP branch -----------------------------------------------------------------------------------------------------------
def pad_square(img: np.ndarray, background=(255, 255, 255)) -> np.ndarray:
    """clip_processor.py:35-52 pad_pil."""
    h, w = img.shape[:2]
    if h == w:
        return img
    s = max(h, w)
    out = np.empty((s, s, img.shape[2]), np.uint8)
    out[:] = np.asarray(background, np.uint8)
    if w > h:
        o = (w - h) // 2
        out[o:o + h, :w] = img
    else:
        o = (h - w) // 2
        out[:h, o:o + w] = img
    return out


def shortest_edge_size(h: int, w: int, size: int):
    """transformers.image_transforms.get_resize_output_image_size(default_to_square=False)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_offsets(h: int, w: int, ch: int, cw: int):
    """transformers.image_transforms.center_crop: top = (h - ch) // 2, left = (w - cw) // 2 (images smaller than the crop are zero
    padded there; CLIP's shortest-edge resize makes that impossible)."""
    return (h - ch) // 2, (w - cw) // 2


def clip_lut(mean=CLIP_MEAN, std=CLIP_STD, scale=1.0 / 255.0) -> np.ndarray:
    """[3, 256] fp32: value of (float32(float64(u8) * scale) - mean) / std, the exact numpy op sequence of
    transformers.image_transforms.rescale + normalize."""
    u = np.arange(256, dtype=np.uint8)
    x = (u.astype(np.float64) * scale).astype(np.float32)
    m, s = np.array(mean, np.float32), np.array(std, np.float32)
    return np.stack([(x - m[c]) / s[c] for c in range(3)]).astype(np.float32)


def clip_preprocess(img: np.ndarray, size: int = 224, aspect_ratio=None) -> np.ndarray:
    """uint8 H x W x 3 -> fp32 [3, size, size] (CLIPProcessor.__call__)."""
    if aspect_ratio == "pad":
        img = pad_square(img)
    nh, nw = shortest_edge_size(img.shape[0], img.shape[1], size)
    r = pil_resize(img, (nh, nw), BICUBIC)
    top, left = center_crop_offsets(nh, nw, size, size)
    r = r[top:top + size, left:left + size]
    lut = clip_lut()
    return np.stack([lut[c][r[..., c]] for c in range(3)])


# ---- SAM branch ------------------------------------------------------------------------------------------------------------
def sam_preprocess_shape(h: int, w: int, long_side: int = 1024):
    """transforms.py get_preprocess_shape."""
    scale = long_side * 1.0 / max(h, w)
    return int(h * scale + 0.5), int(w * scale + 0.5)


def sam_lut(mean=SAM_MEAN, std=SAM_STD) -> np.ndarray:
    """[3, 256] fp32 of (float32(u8) - mean) / std with torch.Tensor([..]) fp32 constants (mask_toolbox.py:10-18)."""
    x = np.arange(256, dtype=np.float32)
    m, s = np.array(mean, np.float32), np.array(std, np.float32)
    return np.stack([(x - m[c]) / s[c] for c in range(3)]).astype(np.float32)


def sam_preprocess(img: np.ndarray, long_side: int = 1024):
    """uint8 H x W x 3 -> (fp32 [3, long_side, long_side], (resized h, resized w))."""
    nh, nw = sam_preprocess_shape(img.shape[0], img.shape[1], long_side)
    r = pil_resize(img, (nh, nw), BILINEAR)
    lut = sam_lut()
    out = np.zeros((3, long_side, long_side), np.float32)
    for c in range(3):
        out[c, :nh, :nw] = lut[c][r[..., c]]
    return out, (nh, nw)


# ---- evaluation ------------------------------------------------------------------------------------------------------------
def intersection_and_union(output: np.ndarray, target: np.ndarray, K: int = 2, ignore_index: int = 255):
    """evaluation/tools.py:29-41 on integer label maps -> (area_intersection, area_union, area_target), fp32 [K] each."""
    o = output.reshape(-1).astype(np.int64).copy()
    t = target.reshape(-1).astype(np.int64)
    o[t == ignore_index] = ignore_index
    inter = o[o == t]
    hist = lambda v: np.array([(v == k).sum() for k in range(K)], np.float32)
    ai, ao, at = hist(inter), hist(o), hist(t)
    return ai, ao + at - ai, at


def mask_iou_stats(logits: np.ndarray, target: np.ndarray):
    """trainers/ullava_trainer.py:40-52 for one (n, H, W) stack: threshold at 0, per-mask counts -> (intersection[2], union[2],
    acc_iou[2]) accumulated over the n masks exactly like the trainer."""
    out = (logits > 0).astype(np.int32)
    inter = np.zeros(2, np.float32)
    union = np.zeros(2, np.float32)
    acc = np.zeros(2, np.float32)
    for m, o in zip(target.astype(np.int32), out):
        i, u, _ = intersection_and_union(o, m, 2, 255)
        inter += i
        union += u
        a = i / (u + np.float32(1e-5))
        a[u == 0] += 1.0
        acc += a
    return inter, union, acc / np.float32(target.shape[0])
