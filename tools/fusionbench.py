#!/usr/bin/env python3
"""Time s2m2_feature_fusion (K10) against the two K5 launches it replaces (hipGraph-timed).   python tools/fusionbench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph  # noqa: E402

SHAPES = [("1/32 x2 C256", 2 * 32 * 38, 256), ("1/16 x2 C256", 2 * 64 * 76, 256), ("1/16 x1 C256", 64 * 76, 256), ("1/8 x2 C128", 2 * 128 * 152, 128),
          ("1/8 x1 C128", 128 * 152, 128), ("1/4 x1 C128", 256 * 304, 128), ("1/4 x2 C128", 2 * 256 * 304, 128)]


def main():
    for name, rows, C in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        z0 = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
        z1 = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
        w1 = (torch.randn(3 * C, 2 * C, 1, 1, device="cuda", generator=g) / math.sqrt(2 * C)).half()
        w2 = (torch.randn(C, 3 * C, 1, 1, device="cuda", generator=g) / math.sqrt(2 * C)).half()
        p1, p2 = pack.pack_conv(w1, torch.float16), pack.pack_conv(w2, torch.float16)
        b1, bg, bf = (pack.pack_bias(torch.randn(n, device="cuda", generator=g), n) for n in (3 * C, C, C))

        def fused():
            return hip.feature_fusion(z0, z1, p1, b1, p2, bg, bf)

        def sep():
            gf = hip.conv2d([z0, z1], p1, b1, 1, 1, 3 * C, act=hip.ACT_GELU)
            return hip.conv2d([gf], p2, bg, 1, 1, C, act=hip.ACT_SIGMOID, epi=hip.EPI_DUALMIX, aux0=z0, aux1=z1, ksplit=C, bias2=bf)

        fl = 2.0 * rows * (2 * C * 3 * C + 3 * C * C)
        tf, ts = timeit_graph(fused, 20, 3), timeit_graph(sep, 20, 3)
        print(f"{name:14s} K10 {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s) | 2 x K5 {ts:7.1f} us ({fl / ts / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
