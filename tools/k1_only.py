#!/usr/bin/env python3
"""Launch only K1 (ln_corr) at a BASELINE size N times, plus a known-size device copy used to calibrate the
FETCH_SIZE / WRITE_SIZE PMC counters (MI355X_MICROARCH.md, HBM section).  Driven under rocprofv3 by tools/gpu_profile.sh.

    python tools/k1_only.py [--case c3] [--dtype fp16] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip  # noqa: E402

CASES = {"c2": (128, 120, 160), "c3": (128, 256, 304), "c4": (256, 256, 304), "c5": (384, 512, 608)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="c3")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--calib-mib", type=int, default=512)
    ap.add_argument("--form", default="corr", choices=["corr", "ln"],
                    help="corr: the shipped fp16 path (tokens normalised by K9, s2m2_corr, volume rows on 128-byte lines); ln: s2m2_ln_corr, dense rows")
    a = ap.parse_args()
    C, h, w = CASES[a.case]
    dt = torch.float16 if a.dtype == "fp16" else torch.float32
    torch.manual_seed(0)
    feat = torch.randn(2, h, w, C, device="cuda").to(dt)
    g = torch.ones(C, device="cuda")
    b = torch.zeros(C, device="cuda")
    if a.form == "corr":
        tok = torch.nn.functional.layer_norm(feat.float(), (C,)).to(dt)
        cv = hip.cv_alloc(1, h, w, dt, "cuda")
    for _ in range(a.iters):
        cv = hip.corr(tok, out=cv) if a.form == "corr" else hip.ln_corr(feat, g, b)
    torch.cuda.synchronize()
    # calibration: a streaming copy of a known byte count, larger than the 256 MiB Infinity Cache
    n = a.calib_mib * 1024 * 1024 // 2
    src = torch.randn(n // 4, device="cuda").half().repeat(4)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
    print(f"k1_only: form={a.form} pitch={cv.stride(2)} case={a.case} dtype={a.dtype} iters={a.iters} cv={tuple(cv.shape)} calib_bytes={src.numel() * 2}")


if __name__ == "__main__":
    main()
