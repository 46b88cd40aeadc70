"""Seeded, order-independent weight generator (pure torch-CPU, no HIP).

The GPU box has no reference checkout and no checkpoints, so every parity test
regenerates identical weights from (tensor name, shape, seed).  Each tensor gets its own
CPU generator keyed by crc32(name) ^ seed, so the result does not depend on
state-dict iteration order.  torch's CPU randn stream is a pure function of the
seed for a given torch build (the GPU box runs this same image).

Scale rules keep activations O(1) at any width so that tiny test models are as
sensitive to kernel bugs as full-size ones:
  * norm weights            1 + 0.1 N(0,1)
  * biases (1-D "*.bias")   0.1 N(0,1)
  * tables / tokens         0.5 N(0,1)   (embed_tokens, position/pos embeddings, iou_token/mask_tokens, class_embedding, rel_pos)
  * Gaussian PE buffer      N(0,1)       (SAM positional_encoding_gaussian_matrix)
  * everything else (>=2-D) N(0,1) / sqrt(fan_in)
`hf_init=True` switches to the reference's initializer_range recipe (N(0,0.02) weights,
ones for norms, zero biases) used by bench.py for the 7B random-init model.
"""
import zlib
from typing import Dict, Iterable, Tuple

import torch

_NORM_KEYS = ("norm", "layrnorm", "layer_norm", "neck.1.", "neck.3.", "output_upscaling.1.")
_TABLE_KEYS = ("embed_tokens", "position_embedding", "pos_embed", "iou_token", "mask_tokens", "class_embedding", "rel_pos",
               "no_mask_embed", "point_embeddings", "not_a_point_embed")


def _kind(name: str, shape: Tuple[int, ...]) -> str:
    if "positional_encoding_gaussian_matrix" in name:
        return "gauss"
    if any(k in name for k in _NORM_KEYS) and len(shape) == 1:
        return "norm_b" if name.endswith("bias") else "norm_w"
    if name.endswith(".bias") and len(shape) == 1:
        return "bias"
    if any(k in name for k in _TABLE_KEYS):
        return "table"
    return "weight"


def seeded_tensor(name: str, shape: Iterable[int], seed: int = 0, dtype=torch.float32, hf_init: bool = False) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    t = torch.randn(shape, generator=g, dtype=torch.float32)
    k = _kind(name, shape)
    if hf_init:
        if k == "norm_w":
            t = torch.ones(shape)
        elif k in ("norm_b", "bias"):
            t = torch.zeros(shape)
        elif k != "gauss":
            t = t * 0.02
    else:
        if k == "norm_w":
            t = 1.0 + 0.1 * t
        elif k in ("norm_b", "bias"):
            t = 0.1 * t
        elif k == "table":
            t = 0.5 * t
        elif k == "weight":
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            # ConvTranspose2d weights are [in, out, kh, kw]: fan_in is shape[0]*kh*kw / stride-overlap ~ shape[0]
            if "output_upscaling" in name and len(shape) == 4:
                fan_in = shape[0]
            t = t / max(fan_in, 1) ** 0.5
    return t.to(dtype)


def seeded_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, dtype=torch.float32, hf_init: bool = False) -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, s, seed, dtype, hf_init) for k, s in shapes.items()}
