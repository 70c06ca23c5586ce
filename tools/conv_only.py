#!/usr/bin/env python3
"""Launch one conv shape N times (for rocprofv3 PMC passes):  python tools/conv_only.py --tile 1 [--shape 0]"""
import argparse, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack
from tools.convbench import SHAPES
ap = argparse.ArgumentParser(); ap.add_argument("--tile", type=int, default=0); ap.add_argument("--shape", type=int, default=0); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
name, N, H, W, ci, co, kh, kw = SHAPES[a.shape]
x = torch.randn(N, H, W, ci, device="cuda").half()
w = (torch.randn(co, ci, kh, kw, device="cuda") / math.sqrt(ci * kh * kw)).half()
wp, bp = pack.pack_conv(w, torch.float16), pack.pack_bias(torch.randn(co, device="cuda"), co)
for _ in range(a.iters):
    hip.conv2d([x], wp, bp, kh, kw, wp.shape[0], act=hip.ACT_GELU, tile=a.tile)
torch.cuda.synchronize()
print("conv_only", name, "tile", a.tile)
