#!/usr/bin/env python3
"""K10 at the forward's shapes: the LDS-staged form (s2m2_feature_fusion) against the direct form (s2m2_feature_fusion_frag), in-graph, hot
and with the L2s evicted before every call.   python tools/fusionbench_direct.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph, timeit_graph_cold  # noqa: E402

# (label, n, h, w, C, z1 coarse)
SHAPES = [("1/32 x2 C256 same", 2, 32, 38, 256, False), ("1/16 x1 C256 up", 1, 64, 76, 256, True), ("1/16 x2 C256 up", 2, 64, 76, 256, True),
          ("1/16 x1 C128 up", 1, 64, 76, 128, True), ("1/8 x1 C128 up", 1, 128, 152, 128, True), ("1/8 x2 C128 up", 2, 128, 152, 128, True),
          ("1/4 x1 C128 up", 1, 256, 304, 128, True), ("1/4 x2 C128 up", 2, 256, 304, 128, True)]


def main():
    for name, n, h, w, C, coarse in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        z0 = torch.randn(n, h, w, C, device="cuda", generator=g).half()
        z1 = torch.randn(n, h // 2, w // 2, C, device="cuda", generator=g).half() if coarse else torch.randn(n, h, w, C, device="cuda", generator=g).half()
        w1 = pack.pack_conv((torch.randn(3 * C, 2 * C, 1, 1, device="cuda", generator=g) / math.sqrt(2 * C)).half(), torch.float16)
        w2 = pack.pack_conv((torch.randn(C, 3 * C, 1, 1, device="cuda", generator=g) / math.sqrt(2 * C)).half(), torch.float16)
        b1, bg, bf = (torch.randn(k, device="cuda", generator=g) for k in (3 * C, C, C))
        ws = pack.fusion_frag(w1, w2)

        def lds():
            return hip.feature_fusion(z0, z1, w1, b1, w2, bg, bf, z1_coarse=coarse)

        def direct():
            return hip.feature_fusion(z0, z1, ws, b1, None, bg, bf, z1_coarse=coarse, frag=True)

        assert torch.equal(lds(), direct())
        line = f"{name:18s} rows {n * h * w:7d}"
        for lbl, fn in (("lds", lds), ("direct", direct)):
            th, tc = timeit_graph(fn, 20, 3), timeit_graph_cold(fn, 20, 3)
            line += f" | {lbl} hot {th:6.1f} cold {tc:6.1f}"
        print(line + "  (us)", flush=True)


if __name__ == "__main__":
    main()
