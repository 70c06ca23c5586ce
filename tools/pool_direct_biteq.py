#!/usr/bin/env python3
"""Is the pooled 1x1 layer on the direct K9 form bit-identical to the K5 pool2 launch it replaces?  (same mean, same k16 order, same epilogue)
And the whole fp16 forward with S2M2_POOL_DIRECT=1 vs 0?   python tools/pool_direct_biteq.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402


def main():
    ok = True
    for C, (N, H, W), n in ((128, (2, 256, 304), 1), (128, (2, 128, 152), 2), (256, (2, 64, 76), 1), (128, (1, 120, 160), 1), (256, (1, 30, 40), 1)):
        g = torch.Generator(device="cuda").manual_seed(C + H)
        x = (torch.randn(N, H, W, C, device="cuda", generator=g) * 1.5 + 0.3).half()
        wp = pack.pack_conv((torch.randn(n * C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half(), torch.float16)
        bp = pack.pack_bias(torch.randn(n * C, device="cuda", generator=g) * 0.3, n * C)
        y = hip.mlp_fan(x, pack.chain_frag(wp), bp, None, frag=True, pool2=True)
        ref = hip.conv2d([x], wp, bp, 1, 1, n * C, pool2=True)
        eq = torch.equal(y, ref)
        ok &= eq
        print(f"C={C} {N}x{H}x{W} -> {n}C: bit-identical to K5 pool2: {eq}  max|diff| {float((y.float() - ref.float()).abs().max()):.3e}", flush=True)
    from s2m2_amd.model import S2M2
    from s2m2_amd.weights import seeded_state_dict, synthetic_pair
    sd = seeded_state_dict(128, 1, 1, 0)
    l, r = synthetic_pair(480, 640, 1, 32, 0)
    l, r = l.cuda(), r.cuda()
    outs = []
    for v in ("1", "0"):
        os.environ["S2M2_POOL_DIRECT"] = v
        m = S2M2(128, 1, 1, use_positivity=True, refine_iter=3)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        with torch.autocast("cuda", dtype=torch.float16):
            outs.append([t.clone() for t in m(l, r)])
    same = all(torch.equal(a, b) for a, b in zip(*outs))
    print("fp16 forward 640x480 refine_iter 3, S2M2_POOL_DIRECT=1 vs 0: bit-identical:", same, flush=True)
    print("ALL BIT-IDENTICAL" if ok and same else "DIFFERENCES")


if __name__ == "__main__":
    main()
