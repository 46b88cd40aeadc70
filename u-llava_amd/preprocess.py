"""On-device image pre/post-processing either side of the forward path: the uint8 image is uploaded once and
everything after that -- Pillow-exact resampling, crop / pad, normalisation, the IoU counts of the evaluation loop -- runs in HIP
kernels, so the host CPU no longer sits between the dataloader and 8 GPUs.

Host-side mirror of the reference's processor objects (same names, same call signatures, same results bit for bit):
    CLIPProcessor            dataset/processors/clip_processor.py:23-103  (transformers CLIPImageProcessor inside)
    SegToolBox               dataset/tools/mask_toolbox.py:8-28           (segment_anything ResizeLongestSide inside)
    intersectionAndUnionGPU  evaluation/tools.py:29-41
    mask_iou_stats           trainers/ullava_trainer.py:40-52

What stays on the host is a few KB of constants per image size: Pillow's resampling taps (precompute_coeffs, double precision,
cached per (in, out, filter)) and the 3 x 256 look-up tables that hold the reference's normalisation of every byte value.
There is no CPU fallback: tensors must live on the GPU and the HIP library must load.
"""
import functools
import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import ops
from .ops import _p, _stream

PRECISION_BITS = 32 - 8 - 2
_SUPPORT = {"bilinear": 1.0, "bicubic": 2.0}
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _filter(kind: str, x: float) -> float:
    x = -x if x < 0.0 else x
    if kind == "bilinear":
        return 1.0 - x if x < 1.0 else 0.0
    a = -0.5
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=256)
def _taps_host(in_size: int, out_size: int, kind: str):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc over the whole axis (python floats are C doubles)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = _SUPPORT[kind] * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ws = [_filter(kind, (x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in ws:
            ww += w
        for x, w in enumerate(ws):
            if ww != 0.0:
                w = w / ww
            v = w * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


_TAPS_DEV = {}


def _taps(in_size: int, out_size: int, kind: str, device):
    key = (in_size, out_size, kind, str(device))
    t = _TAPS_DEV.get(key)
    if t is None:
        b, k = _taps_host(in_size, out_size, kind)
        t = _TAPS_DEV[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1])
    return t


def _chk_u8(img: torch.Tensor):
    if not img.is_cuda:
        raise RuntimeError("u-llava_amd.preprocess: the image must live on the GPU (no CPU path exists)")
    if img.dtype != torch.uint8 or img.dim() != 3 or not img.is_contiguous():
        raise RuntimeError("u-llava_amd.preprocess: expected a contiguous uint8 [H, W, C] image")


def resize_u8(img: torch.Tensor, out_hw: Tuple[int, int], kind: str) -> torch.Tensor:
    """PIL.Image.resize((w, h), BILINEAR | BICUBIC) of a uint8 [H, W, C] device image: horizontal pass, then vertical pass;
    a pass that keeps its size is skipped (as ImagingResample does)."""
    _chk_u8(img)
    oh, ow = int(out_hw[0]), int(out_hw[1])
    x = img
    for axis, out in ((1, ow), (0, oh)):
        H, W, C = x.shape
        if out == x.shape[axis]:
            continue
        b, k, ksize = _taps(x.shape[axis], out, kind, x.device)
        dst = torch.empty((H, out, C) if axis == 1 else (out, W, C), device=x.device, dtype=torch.uint8)
        _lib.call("ull_resample_u8", _p(x), H, W, C, axis, out, _p(b), _p(k), ksize, _p(dst), _stream())
        x = dst
    return x.clone() if x is img else x


def _lut_chw(img: torch.Tensor, top: int, left: int, lut: torch.Tensor, out_hw, copy_hw, dtype) -> torch.Tensor:
    H, W, C = img.shape
    out = torch.empty(3, out_hw[0], out_hw[1], device=img.device, dtype=dtype)
    if dtype not in ops.DT_CODE:
        raise NotImplementedError("pixel tensors are produced in fp32 (the reference's dtype) or bf16 / fp16 (its .to(dtype) cast)")
    _lib.call("ull_u8_lut_chw", _p(img), H, W, C, top, left, _p(lut), _p(out), out_hw[0], out_hw[1], copy_hw[0], copy_hw[1],
              ops.DT_CODE[dtype], _stream())
    return out


class CLIPProcessor:
    """dataset/processors/clip_processor.py CLIPProcessor with the CLIPImageProcessor constants passed directly (the reference reads
    them from preprocessor_config.json: shortest_edge = crop = 224 or 336, bicubic, OPENAI_CLIP mean / std, 1/255)."""

    def __init__(self, size: int = 224, aspect_ratio: Optional[str] = None, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD,
                 rescale_factor: float = 1.0 / 255.0, device="cuda:0"):
        self.size, self.aspect_ratio, self.device = int(size), aspect_ratio, device
        u = np.arange(256, dtype=np.uint8)
        x = (u.astype(np.float64) * rescale_factor).astype(np.float32)               # transformers.image_transforms.rescale
        m, s = np.array(image_mean, np.float32), np.array(image_std, np.float32)
        self.lut = torch.from_numpy(np.stack([(x - m[c]) / s[c] for c in range(3)]).astype(np.float32)).to(device)   # normalize

    @staticmethod
    def pad_square(img: torch.Tensor, background_color=(255, 255, 255)) -> torch.Tensor:
        """pad_pil (clip_processor.py:35-52) on a uint8 [H, W, 3] device image."""
        h, w, c = img.shape
        if h == w:
            return img
        s = max(h, w)
        out = torch.empty(s, s, c, device=img.device, dtype=torch.uint8)
        out[:] = torch.tensor(background_color, device=img.device, dtype=torch.uint8)
        if w > h:
            o = (w - h) // 2
            out[o:o + h, :w] = img
        else:
            o = (h - w) // 2
            out[:h, o:o + w] = img
        return out

    def __call__(self, item: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """uint8 [H, W, 3] RGB device image -> [3, size, size] pixel_values."""
        _chk_u8(item)
        if self.aspect_ratio == "pad":
            item = self.pad_square(item)
        h, w = item.shape[:2]
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = self.size, int(self.size * long / short)               # get_resize_output_image_size
        nh, nw = (new_long, new_short) if w <= h else (new_short, new_long)
        r = resize_u8(item, (nh, nw), "bicubic")
        top, left = (nh - self.size) // 2, (nw - self.size) // 2                     # center_crop
        return _lut_chw(r, top, left, self.lut, (self.size, self.size), (self.size, self.size), dtype)


class SegToolBox:
    """dataset/tools/mask_toolbox.py SegToolBox: apply_image = ResizeLongestSide(1024).apply_image, preprocess = normalise + pad."""

    def __init__(self, device="cuda:0", sam_size: int = 1024):
        self.sam_size, self.device = sam_size, device
        x = np.arange(256, dtype=np.float32)
        m, s = np.array([123.675, 116.28, 103.53], np.float32), np.array([58.395, 57.12, 57.375], np.float32)
        self.lut = torch.from_numpy(np.stack([(x - m[c]) / s[c] for c in range(3)]).astype(np.float32)).to(device)

    def get_preprocess_shape(self, oldh: int, oldw: int) -> Tuple[int, int]:
        scale = self.sam_size * 1.0 / max(oldh, oldw)
        return int(oldh * scale + 0.5), int(oldw * scale + 0.5)

    def apply_image(self, image: torch.Tensor) -> torch.Tensor:
        """uint8 [H, W, 3] -> uint8 [h', w', 3], longest side = sam_size (torchvision resize of a PIL image = PIL bilinear)."""
        _chk_u8(image)
        return resize_u8(image, self.get_preprocess_shape(image.shape[0], image.shape[1]), "bilinear")

    def preprocess(self, x: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """uint8 [h', w', 3] (the output of apply_image; the reference permutes it to CHW first) -> [3, sam_size, sam_size]."""
        _chk_u8(x)
        h, w = x.shape[:2]
        return _lut_chw(x, 0, 0, self.lut, (self.sam_size, self.sam_size), (h, w), dtype)


def mask_iou_counts(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = 255) -> torch.Tensor:
    """int32 [n, 6] = {inter0, inter1, out0, out1, tgt0, tgt1} of (logits > 0) against uint8 targets, ignore_index pixels dropped."""
    if not logits.is_cuda or logits.dtype != torch.float32 or not logits.is_contiguous():
        raise RuntimeError("u-llava_amd.preprocess: mask logits must be contiguous fp32 on the GPU (postprocess_masks returns fp32)")
    if target.dtype != torch.uint8 or not target.is_contiguous() or target.shape != logits.shape:
        raise RuntimeError("u-llava_amd.preprocess: targets must be contiguous uint8 of the logits' shape")
    n = logits.shape[0]
    hw = logits[0].numel()
    counts = torch.zeros(n, 6, device=logits.device, dtype=torch.int32)
    _lib.call("ull_mask_iou_counts", _p(logits), _p(target), n, hw, ignore_index, _p(counts), _stream())
    return counts


def intersectionAndUnionGPU(logits: torch.Tensor, target: torch.Tensor, K: int = 2, ignore_index: int = 255):
    """evaluation/tools.py:29-41 for ONE mask, fused with the `> 0` threshold of its caller -> (area_intersection, area_union,
    area_target), fp32 [2] each, like the torch.histc results."""
    if K != 2:
        raise NotImplementedError("the u-LLaVA evaluation uses K = 2 (background / object)")
    c = mask_iou_counts(logits.reshape(1, -1), target.reshape(1, -1), ignore_index)[0].float()
    inter, out, tgt = c[0:2], c[2:4], c[4:6]
    return inter, out + tgt - inter, tgt


def mask_iou_stats(pred_masks: torch.Tensor, gt_masks: torch.Tensor):
    """trainers/ullava_trainer.py:40-52: pred_masks fp32 [n, H, W] logits, gt_masks [n, H, W] -> (intersection[2], union[2],
    acc_iou[2]) accumulated over the n masks."""
    c = mask_iou_counts(pred_masks.contiguous(), gt_masks.to(torch.uint8).contiguous()).float()
    inter, union = c[:, 0:2], c[:, 2:4] + c[:, 4:6] - c[:, 0:2]
    acc = inter / (union + 1e-5)
    acc = acc + (union == 0).float()
    return inter.sum(0), union.sum(0), acc.sum(0) / pred_masks.shape[0]
