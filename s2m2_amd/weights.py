"""Deterministic random weights for parity tests and benchmarks.

Pretrained ``CH{C}NTR{n}.pth`` checkpoints are not shipped with the reference (README.md:164-169 links
only), and PyTorch's default init makes the network degenerate (the cost volume
collapses to ~127.97 everywhere).  Every tensor is drawn from its own counter-based Philox stream
keyed by ``crc32(name)`` so the values do not depend on module construction order, torch version
or device:

* tensors with >= 2 dims:  N(0, gain^2 / fan_in), fan_in = shape[1] * prod(shape[2:])  (LeCun normal)
* 1-D ``*.weight`` (GroupNorm / LayerNorm scale):  1 + 0.1 * N(0, 1)
* 1-D ``*.bias``:  bias_std * N(0, 1)
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Mapping, Tuple

import numpy as np
import torch

from .spec import param_table


def _stream(name: str, seed: int) -> np.random.Generator:
    key = (zlib.crc32(name.encode()) << 32) | (seed & 0xFFFFFFFF)
    return np.random.Generator(np.random.Philox(key=key))


def draw_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, gain: float = 1.0,
                bias_std: float = 0.02) -> torch.Tensor:
    g = _stream(name, seed)
    x = g.standard_normal(size=shape, dtype=np.float64)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        x *= gain / np.sqrt(fan_in)
    elif name.endswith(".weight"):
        x = 1.0 + 0.1 * x
    else:
        x *= bias_std
    return torch.from_numpy(x.astype(np.float32))


def seeded_state_dict(feature_channels: int, dim_expansion: int, num_transformer: int, seed: int = 0,
                      gain: float = 1.0, bias_std: float = 0.02) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU state_dict with the reference's key names (see :mod:`s2m2_amd.spec`)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_table(feature_channels, dim_expansion, num_transformer).items():
        out[name] = draw_tensor(name, shape, seed, gain, bias_std)
    return out


def synthetic_pair(height: int, width: int, batch: int = 1, disparity: int = 16, seed: int = 0,
                   noise: float = 55.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Textured stereo pair with a known constant horizontal shift.

    A smooth random texture of width ``W + disparity`` is cut twice: ``left = base[..., d:]``,
    ``right = base[..., :W]`` so that left pixel x matches right pixel x - d.  Values in [0, 255] fp32.
    """
    g = np.random.Generator(np.random.Philox(key=0xC0FFEE ^ seed))
    H, W, D = height, width, disparity
    lo = g.random(size=(batch, 3, H // 8 + 2, (W + D) // 8 + 2), dtype=np.float64)
    lo_t = torch.from_numpy(lo.astype(np.float32))
    base = torch.nn.functional.interpolate(lo_t, size=(H, W + D), mode="bicubic", align_corners=True)
    fine = torch.from_numpy(g.random(size=(batch, 3, H, W + D), dtype=np.float64).astype(np.float32))
    base = base.clamp(0, 1) * (255.0 - noise) + fine * noise
    left = base[..., D:].contiguous()
    right = base[..., :W].contiguous()
    return left, right


def noise_pair(height: int, width: int, batch: int = 1, seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """iid uniform integer images in [0, 255] (throughput input; timing is data independent)."""
    g = np.random.Generator(np.random.Philox(key=0xBEEF ^ seed))
    a = g.integers(0, 256, size=(2, batch, 3, height, width), dtype=np.int64)
    t = torch.from_numpy(a.astype(np.float32))
    return t[0].contiguous(), t[1].contiguous()
