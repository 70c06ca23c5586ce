#!/usr/bin/env python3
"""Launch-to-launch period of the forward's conv_frag_kernel (K5 v5) layer shapes inside a hipGraph of 20 launches (the state the bench runs them
in), for A/B of library switches that are read once per process (S2M2_FRAG_STAGGER, ...):    python tools/convperiod.py [tag]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402

CASES = [  # N, H, W, cin (list = concatenated sources), cout, kh, kw, act, epi, calls per forward
    (1, 256, 304, [128], 128, 3, 3, 1, 0, 10),
    (1, 256, 304, [128], 128, 3, 3, 0, 1, 10),
    (1, 256, 304, [128, 128], 256, 1, 3, 3, 1, 3),      # GRU z | r (epi MUL on the upper half in the engine; here on all couts)
    (1, 256, 304, [256], 128, 3, 3, 0, 0, 3),
    (1, 256, 304, [128], 256, 3, 3, 1, 0, 3),
    (1, 128, 152, [128], 128, 3, 3, 1, 0, 10),
    (1, 128, 152, [128], 128, 3, 3, 0, 1, 10),
    (1, 64, 76, [256], 256, 3, 3, 1, 0, 8),
    (1, 64, 76, [256], 256, 3, 3, 0, 1, 8),
    (2, 256, 304, [128], 128, 3, 3, 1, 0, 1),
    (2, 512, 608, [128], 128, 3, 3, 1, 0, 1),
]


def period(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (n * reps))
    return best


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    tot = 0.0
    for N, H, W, cins, co, kh, kw, act, epi, calls in CASES:
        xs = [torch.randn(N, H, W, c, device="cuda").half() for c in cins]
        ci = sum(cins)
        w = (torch.randn(co, ci, kh, kw, device="cuda") / math.sqrt(ci * kh * kw)).half()
        b = torch.randn(co, device="cuda")
        a0 = torch.rand(N, H, W, co, device="cuda").half() if epi else None
        wf, bp = pack.pack_conv_frag(w, torch.float16, [(c, c) for c in cins]), pack.pack_bias(b, co)
        y = torch.empty(N, H, W, co, device="cuda", dtype=torch.float16)
        t = period(lambda: hip.conv2d(xs, wf, bp, kh, kw, co, act=act, epi=epi, aux0=a0, korder=2, out=y))
        fl = 2.0 * N * H * W * ci * co * kh * kw
        tot += t * calls
        print(f"{tag:14s} {N}x{H}x{W} {'+'.join(map(str, cins)):>7}->{co:<3} k{kh}x{kw} act={act} epi={epi}  {t:8.2f} us  {fl / t / 1e6:7.1f} TF/s   x{calls}")
    print(f"{tag:14s} weighted by calls per forward: {tot:8.1f} us")


if __name__ == "__main__":
    main()
