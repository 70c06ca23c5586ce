#!/usr/bin/env python3
"""K11 (s2m2_pw_direct) against the K5 launch it replaces, per layer shape of the S model at 1216x1024 (hipGraph-timed, fp16).
    python tools/pwbench.py        -> the table kept as profiles/r04/pwbench.txt"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph  # noqa: E402

# name, source channels, Cout, act, leading shape, shuffle (ConvTranspose 2x2 s2)
LAYERS = [("refiner corr_feat.0 (x2 block-diag)", (32,), 192, 1, (1, 256, 304), False),
          ("refiner corr_feat.2 (x2 block-diag)", (192,), 128, 0, (1, 256, 304), False),
          ("refiner conf_occ_feat.2", (64,), 32, 0, (1, 256, 304), False),
          ("refiner disp_corr_ctx_cat.0", (96, 128, 128, 32), 256, 1, (1, 256, 304), False),
          ("up_conv 2C->C @1/16 x2", (256,), 128, 0, (2, 64, 76), False),
          ("up_conv 2C->C @1/32 x2", (256,), 128, 0, (2, 32, 38), False),
          ("up_conv 2C->C @1/16 x1", (256,), 128, 0, (1, 64, 76), False),
          ("mask1x conv_concat.2 (48->16, full res)", (48,), 16, 0, (1, 1024, 1216), False),
          ("mask4x conv_x convT (C->64) @1/4", (128,), 256, 0, (1, 256, 304), True),
          ("mask4x conv_concat.2 convT (128->16) @1/2", (128,), 64, 0, (1, 512, 608), True),
          ("mask1x conv_ctx convT (C->16) @1/2", (128,), 64, 0, (1, 512, 608), True)]


def main():
    print(f"{'layer':<44}{'rows':>9}{'K':>5}{'Cout':>6}   {'K5 us':>8}{'K11 us':>8}   bytes in+out -> K11 GB/s")
    for name, cs, cout, act, shp, shuf in LAYERS:
        g = torch.Generator(device="cuda").manual_seed(1)
        K = sum(cs)
        srcs = [(torch.randn(*shp, c, device="cuda", generator=g)).half() for c in cs]
        w = (torch.randn(cout, K, 1, 1, device="cuda", generator=g) / math.sqrt(K)).half()
        wp = pack.pack_conv(w, torch.float16, [(c, c) for c in cs])
        bp = pack.pack_bias(torch.randn(cout, device="cuda", generator=g), cout)
        wf = pack.pw_frag(wp)
        sh = cout // 4 if shuf else 0
        t5 = timeit_graph(lambda: hip.conv2d(srcs, wp, bp, 1, 1, cout, act=act, shuffle2=sh), 20, 3)
        t11 = timeit_graph(lambda: hip.pw_direct(srcs, wf, bp, cout, act=act, shuffle2=sh), 20, 3)
        rows = srcs[0].numel() // cs[0]
        nbytes = rows * (K + cout) * 2
        print(f"{name:<44}{rows:>9}{K:>5}{cout:>6}   {t5:>8.1f}{t11:>8.1f}   {nbytes / 1e6:7.1f} MB -> {nbytes / t11 / 1e3:7.0f}", flush=True)


if __name__ == "__main__":
    main()
