#!/usr/bin/env python3
"""Per-wave timeline of K13 (row_attn_kernel) from shader-clock stamps (experiment build -DS2M2_RA_TRACE=1).
    S2M2_LIB_SUFFIX=_ratrace S2M2_BUILD_DEFINES=-DS2M2_RA_TRACE=1 python -m s2m2_amd.build      (build container)
    S2M2_LIB_SUFFIX=_ratrace python tools/rowattn_trace.py [w=304] [h=256] [heads=1]              (GPU box)"""
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 304
h = int(sys.argv[2]) if len(sys.argv) > 2 else 256
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
C = 128
lib = hip.load()
lib.s2m2_debug_ra_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
g = torch.Generator(device="cuda").manual_seed(0)
weights = pack.rowattn_pack((torch.randn(6 * C, C, device="cuda", generator=g) / math.sqrt(C)).half().contiguous())
vectors = torch.randn(12, C, device="cuda")
x = torch.randn(2, h, w, C, device="cuda", generator=g).half()
for _ in range(3):
    hip.row_attn(x, heads, True, weights, vectors)
torch.cuda.synchronize()
assert lib.s2m2_debug_ra_trace(None, 0, 1) == 0
hip.row_attn(x, heads, True, weights, vectors)
torch.cuda.synchronize()
buf = np.zeros(1024 * 10 * 16, dtype=np.uint64)
assert lib.s2m2_debug_ra_trace(buf.ctypes.data, buf.nbytes, 0) == 0
nwv = (w + 31) // 32
t = buf.reshape(1024, 10, 16).astype(np.float64)[: min(2 * h, 1024), :nwv]
us = 1e6 / 2.4e9                               # ticks of the 100 MHz-derived shader clock counter are reported at the nominal 2.4 GHz
nchunk = (nwv + 4) // 5
names = [(0, 1, "entry -> weights q/k/v + own tokens landed (barrier)"), (1, 2, "Q projection + epilogue"), (2, 3, "barrier"),
         (3, 4, "chunk 0: K / V projection"), (4, 5, "barrier"), (5, 6 if nchunk > 1 else 9, "chunk 0: attention")]
if nchunk > 1:
    names += [(6, 7, "chunk 1: K / V projection"), (7, 8, "barrier"), (8, 9, "chunk 1: attention")]
names += [(9, 10, "barrier (drains proj / ffn.0 DMA)"), (10, 11, "tail vectors + normalise + z reload + barrier"), (11, 12, "proj + residual + LN + ffn.0 + GELU"),
          (12, 13, "barrier (drains ffn.2 DMA)"), (13, 14, "ffn.2 + residual"), (14, 15, "store"), (0, 15, "whole block")]
print(f"row_attn cross, 2 x {h} x {w} x 128, heads {heads}: {t.shape[0]} blocks x {nwv} waves (us at 2.4 GHz ticks per wave: p10 / median / p90)")
for a, b, nm in names:
    d = (t[:, :, b] - t[:, :, a]) * us
    print(f"  {nm:<58}{np.percentile(d, 10):8.2f}{np.percentile(d, 50):8.2f}{np.percentile(d, 90):8.2f}")
print("per wave (median over blocks, us since the block's earliest entry stamp):")
t0 = t[:, :, 0].min(axis=1, keepdims=True)
for s in range(16):
    print(f"  stamp {s:2d}: " + " ".join(f"{np.median((t[:, k, s] - t0[:, 0]) * us):7.2f}" for k in range(nwv)))
