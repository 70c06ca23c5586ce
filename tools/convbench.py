#!/usr/bin/env python3
"""Time s2m2_conv2d against PyTorch-ROCm (MIOpen / hipBLASLt) on the hot-path conv shapes (fp16, channels-last).
    python tools/convbench.py [--iters 30]"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph, timeit_graph_cold  # noqa: E402


COLD = False


def timeit(fn, iters):
    return timeit_graph_cold(fn, 20, 3) if COLD else timeit_graph(fn, 20, 3)

SHAPES = [  # name, N,H,W, Cin, Cout, KH, KW
    ("1/4 3x3 128->128", 1, 256, 304, 128, 128, 3, 3),
    ("1/4 3x3 128->128 x2", 2, 256, 304, 128, 128, 3, 3),
    ("1/4 1x1 128->128", 1, 256, 304, 128, 128, 1, 1),
    ("1/4 1x1 256->256", 1, 256, 304, 256, 256, 1, 1),
    ("1/4 1x1 384->256", 1, 256, 304, 384, 256, 1, 1),
    ("1/4 3x3 256->128", 1, 256, 304, 256, 128, 3, 3),
    ("1/4 3x1 256->128", 1, 256, 304, 256, 128, 3, 1),
    ("1/4 lin 128->384 x2", 2, 256, 304, 128, 384, 1, 1),
    ("1/8 3x3 128->128", 1, 128, 152, 128, 128, 3, 3),
    ("1/16 3x3 256->256", 1, 64, 76, 256, 256, 3, 3),
    ("1/16 3x3 256->256 x2", 2, 64, 76, 256, 256, 3, 3),
    ("1/8 3x3 128->128 x2", 2, 128, 152, 128, 128, 3, 3),
    ("1/32 1x1 256->256", 1, 32, 38, 256, 256, 1, 1),
    ("1/32 1x1 256->256 x2", 2, 32, 38, 256, 256, 1, 1),
    ("1/32 1x1 256->768 x2", 2, 32, 38, 256, 768, 1, 1),
    ("1/16 1x1 256->256 x2", 2, 64, 76, 256, 256, 1, 1),
    ("1/16 1x1 512->768 x2", 2, 64, 76, 512, 768, 1, 1),
    ("1/8 1x1 128->128 x2", 2, 128, 152, 128, 128, 1, 1),
    ("1/8 1x1 256->384 x2", 2, 128, 152, 256, 384, 1, 1),
    ("1/2 3x3 128->128", 1, 512, 608, 128, 128, 3, 3),
    ("1/2 3x3 128->128 x2", 2, 512, 608, 128, 128, 3, 3),
    ("1/4 3x3 128->256", 1, 256, 304, 128, 256, 3, 3),
    ("1/4 3x3 256->384", 1, 256, 304, 256, 384, 3, 3),
    ("1/4 1x3 256->128", 1, 256, 304, 256, 128, 1, 3),
    ("1/2 3x3 128->64", 1, 512, 608, 128, 64, 3, 3),
    ("1/1 3x3 48->48", 1, 1024, 1216, 48, 48, 3, 3),
    ("1/4 3x3 128->8", 1, 256, 304, 128, 8, 3, 3),
    ("1/1 3x3 8->32 narrowK", 1, 1024, 1216, 8, 32, 3, 3),
    ("1/4 3x3 8->160 narrowK", 1, 256, 304, 8, 160, 3, 3),
    ("1/1 1x1 8->16 x2 narrowK", 2, 1024, 1216, 8, 16, 1, 1),
    ("1/1 1x1 16->16 x2 narrowK", 2, 1024, 1216, 16, 16, 1, 1),
    ("1/1 1x1 48->16 narrowK", 1, 1024, 1216, 48, 16, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--cold", action="store_true", help="evict the L2s before every call (the state inside the pipeline)")
    ap.add_argument("--tiles", default="")
    a = ap.parse_args()
    global COLD
    COLD = a.cold
    for name, N, H, W, ci, co, kh, kw in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn(N, H, W, ci, device="cuda").half()
        w = (torch.randn(co, ci, kh, kw, device="cuda") / math.sqrt(ci * kh * kw)).half()
        b = torch.randn(co, device="cuda")
        wp, bp = pack.pack_conv(w, torch.float16), pack.pack_bias(b, co)
        xn = x.permute(0, 3, 1, 2)
        wcl = w.contiguous(memory_format=torch.channels_last)
        bh = b.half()
        fl = 2.0 * N * H * W * ci * co * kh * kw
        t_ref = timeit(lambda: F.gelu(F.conv2d(xn, wcl, bh, padding=(kh // 2, kw // 2))), a.iters)
        line = f"{name:24s} torch conv+gelu {t_ref:8.1f} us ({fl / t_ref / 1e6:6.1f} TF/s) |"
        wk = pack.pack_conv(w, torch.float16, korder=1) if ci % 32 == 0 else None
        for tile in (tuple(int(t) for t in a.tiles.split(',')) if a.tiles else ((6, 2, 16, 17, 22, 20) if kh * kw == 1 else (13, 19, 23))):
            t = timeit(lambda: hip.conv2d([x], wp, bp, kh, kw, wp.shape[0], act=hip.ACT_GELU, tile=tile), a.iters)
            line += f" t{tile} {t:8.1f} us ({fl / t / 1e6:6.1f})"
            if wk is not None and kw > 1 and tile < 12 and False:
                t = timeit(lambda: hip.conv2d([x], wk, bp, kh, kw, wk.shape[0], act=hip.ACT_GELU, tile=tile, korder=1), a.iters)
                line += f" k1 {t:8.1f} ({fl / t / 1e6:6.1f})"
        if pack.frag_eligible(wp.shape[0], ci, kh, kw, torch.float16):           # v5 fragment-stream kernel (K order 2)
            wfr = pack.pack_conv_frag(w, torch.float16)
            t = timeit(lambda: hip.conv2d([x], wfr, bp, kh, kw, wp.shape[0], act=hip.ACT_GELU, korder=2), a.iters)
            line += f" | frag {t:8.1f} us ({fl / t / 1e6:6.1f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
