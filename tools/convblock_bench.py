#!/usr/bin/env python3
"""K14 (s2m2_conv_block: ConvBlock2D in one launch) against the three launches it replaces, on the coarse grids of the S model (fp16, hipGraph replay)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph  # noqa: E402

SHAPES = [(1, 128, 152, 128), (2, 128, 152, 128), (1, 64, 76, 256), (2, 64, 76, 256), (1, 64, 76, 128), (1, 60, 80, 128), (1, 30, 40, 256), (1, 256, 304, 128)]
for N, H, W, C in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    k0 = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / math.sqrt(9 * C)).half()
    k2 = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / math.sqrt(9 * C)).half()
    p0 = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half()
    p2 = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half()
    bs = [torch.randn(C, device="cuda") * 0.3 for _ in range(4)]
    x = torch.randn(N, H, W, C, device="cuda", generator=g).half()
    w0, w2 = pack.pack_conv_frag(k0, torch.float16), pack.pack_conv_frag(k2, torch.float16)
    a0, a2 = pack.chain_frag(pack.pack_conv(p0, torch.float16)), pack.chain_frag(pack.pack_conv(p2, torch.float16))

    def triple():
        b = hip.mlp_chain(x, [(a0, bs[2], hip.ACT_RELU, None), (a2, bs[3], hip.ACT_NONE, None)], frag=True)
        t = hip.conv2d([x], w0, bs[0], 3, 3, C, act=hip.ACT_GELU, korder=2)
        return hip.conv2d([t], w2, bs[1], 3, 3, C, epi=hip.EPI_ADD, aux0=b, korder=2)
    t_old = timeit_graph(triple, 20, 3)
    line = f"({N},{H},{W},{C}): chain + conv + conv {t_old:7.1f} us   conv_block"
    if hip.conv_block_supported(C, H, W, torch.float16) or True:
        for ph in ((2, 4) if C == 128 else (2,)):
            try:
                t_new = timeit_graph(lambda: hip.conv_block(x, w0, bs[0], w2, bs[1], a0, bs[2], a2, bs[3], patch_rows=ph), 20, 3)
                line += f"  {ph}-row patches {t_new:7.1f} us"
            except RuntimeError as e:
                line += f"  {ph}-row: {e}"
    print(line, flush=True)
