#!/usr/bin/env python3
"""Debug helper for the K order 2 (fragment stream) kernel: error map per cout tile / patch row against torch."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402


def run(N, H, W, cin, cout, kh, kw, mode):
    torch.manual_seed(0)
    if mode == "ones":
        x = torch.ones(N, H, W, cin, device="cuda").half()
        w = torch.ones(cout, cin, kh, kw, device="cuda").half() / (cin * kh * kw)
    else:
        x = torch.randn(N, H, W, cin, device="cuda").half()
        w = (torch.randn(cout, cin, kh, kw, device="cuda") / math.sqrt(cin * kh * kw)).half()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1)
    wf = pack.pack_conv_frag(w, torch.float16)
    out = hip.conv2d([x], wf, None, kh, kw, cout, korder=2).float()
    torch.cuda.synchronize()
    err = (out - ref).abs()
    bad = ~torch.isfinite(out)
    print(f"{mode} {N}x{H}x{W} cin={cin} cout={cout} k={kh}x{kw}: max err {float(err[~bad].max()) if (~bad).any() else -1:.3e}  nonfinite {int(bad.sum())}/{bad.numel()}")
    for ct in range(cout // 32):
        e = err[..., ct * 32:(ct + 1) * 32]
        b = bad[..., ct * 32:(ct + 1) * 32]
        print(f"  cout tile {ct}: nonfinite {int(b.sum()):6d}  max err(finite) {float(e[~b].max()) if (~b).any() else -1:.3e}  mean out {float(out[..., ct*32:(ct+1)*32][~b].mean()):.4f} ref {float(ref[..., ct*32:(ct+1)*32].mean()):.4f}")
    rows = err.amax(dim=(0, 2, 3))
    print("  max err per image row:", [f"{float(v):.2e}" for v in rows[:12]])
    cols = err.amax(dim=(0, 1, 3))
    print("  max err per image col:", [f"{float(v):.2e}" for v in cols[:40]])


if __name__ == "__main__":
    hip.load()
    run(1, 8, 32, 128, 128, 3, 3, "ones")
    run(1, 8, 32, 128, 128, 3, 3, "rand")
    run(1, 17, 23, 128, 128, 3, 3, "rand")
    run(1, 8, 32, 256, 128, 3, 1, "rand")
