"""Deterministic synthetic uint8 images shared by gen_golden_preprocess.py and the tests (fixtures store seeds, not pixels)."""
import numpy as np


def synth_image(h: int, w: int, seed: int) -> np.ndarray:
    """smooth gradients + noise, so both the interpolation and the 0/255 clamping paths of the resampler are exercised."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(np.sin(xx / 7.0) + 1) * 100, (np.cos(yy / 5.0) + 1) * 100, ((xx + yy) % 64) * 4.0], -1)
    return np.clip(base + rs.randn(h, w, 3) * 40, 0, 255).astype(np.uint8)
