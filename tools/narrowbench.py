#!/usr/bin/env python3
"""K12 (s2m2_conv_narrow) against the K5 launch it replaces, per layer shape of the S model at 1216x1024 (hipGraph-timed, fp16).
    python tools/narrowbench.py        -> the table kept as profiles/r04/narrowbench.txt"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph  # noqa: E402

# name, kernel size, stride, source channels, Cout, act, (N, H, W)
LAYERS = [("mask1x conv_disp|conv_rgb (full res)", 3, 1, (8,), 32, 2, (1, 1024, 1216)),
          ("refiner disp_feat.0|conf_occ_feat.0 @1/4", 3, 1, (8,), 160, 1, (1, 256, 304)),
          ("cnn conv1_down.0 5x5 s2 (both images)", 5, 2, (16,), 64, 1, (2, 1024, 1216)),
          ("mask1x conv_concat.0 (32+16->48, full res)", 3, 1, (32, 16), 48, 2, (1, 1024, 1216)),
          ("refiner disp_feat.2 (96->96) @1/4", 3, 1, (96,), 96, 0, (1, 256, 304)),
          ("refiner disp|conf_occ update.2 (256->16)", 3, 1, (256,), 16, 0, (1, 256, 304)),
          ("global refiner out_feat (128->8) @1/4", 3, 1, (128,), 8, 0, (1, 256, 304))]


def main():
    print(f"{'layer':<44}{'pixels':>9}{'K':>5}{'Cout':>6}   {'K5 us':>8}{'K12 us':>8}   bytes in+out -> K12 GB/s")
    for name, k, stride, cs, cout, act, shp in LAYERS:
        g = torch.Generator(device="cuda").manual_seed(1)
        cin = sum(cs)
        srcs = [torch.randn(*shp, c, device="cuda", generator=g).half() for c in cs]
        w = (torch.randn(cout, cin, k, k, device="cuda", generator=g) / math.sqrt(cin * k * k)).half()
        wp = pack.pack_conv(w, torch.float16, [(c, c) for c in cs])
        bp = pack.pack_bias(torch.randn(cout, device="cuda", generator=g), cout)
        wf = pack.narrow_frag(wp, k * k)
        t5 = timeit_graph(lambda: hip.conv2d(srcs, wp, bp, k, k, cout, act=act, stride=stride), 20, 3)
        t12 = timeit_graph(lambda: hip.conv_narrow(srcs, wf, bp, k, k, cout, stride=stride, act=act), 20, 3)
        npx = srcs[0].numel() // cs[0]
        nbytes = npx * cin * 2 + (npx // (stride * stride)) * cout * 2
        print(f"{name:<44}{npx:>9}{k * k * cin:>5}{cout:>6}   {t5:>8.1f}{t12:>8.1f}   {nbytes / 1e6:7.1f} MB -> {nbytes / t12 / 1e3:7.0f}", flush=True)


if __name__ == "__main__":
    main()
