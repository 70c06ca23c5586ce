#!/usr/bin/env python3
"""Micro-benchmarks of the hand-written kernels at the BASELINE sizes (run on the GPU box).

    python tools/kbench.py [--iters 50]
Prints one line per kernel/config: time (us), algorithmic GB/s or TFLOP/s.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip  # noqa: E402


def timeit_graph(fn, iters=20, reps=5):
    """GPU time per call with the host out of the loop: `iters` launches captured in one hipGraph, replayed `reps` times.
    (Eager back-to-back launches through ctypes cost ~10 us of host time each, more than many of the kernels.)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (iters * reps)


def timeit_graph_cold(fn, iters=20, reps=5, mb=64):
    """like timeit_graph, but every call is preceded by a `mb` MiB fill that evicts the 8 x 4 MiB L2s (the 256 MiB Infinity Cache
    keeps the operands: that is the state a layer finds inside the real pipeline, where ~30 other layers ran since its last use);
    returns (fill + fn) - fill."""
    buf = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    t_f = timeit_graph(lambda: buf.zero_(), iters, reps)
    t = timeit_graph(lambda: (buf.zero_(), fn()), iters, reps)
    return t - t_f


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    hip.load()
    dev = "cuda"
    res = []
    cases = [("c2 S 640x480", 128, 120, 160, 1), ("c3 S 1216x1024", 128, 256, 304, 1), ("c3 S 1216x1024 B=4", 128, 256, 304, 4),
             ("c4 L 1216x1024", 256, 256, 304, 1), ("c5 XL 2432x2048", 384, 512, 608, 1)]
    for name, C, h, w, B in cases:
        for dt in (torch.float16, torch.float32):
            if dt == torch.float32 and C > 128:
                continue
            e = 2 if dt == torch.float16 else 4
            feat = torch.randn(2 * B, h, w, C, device=dev).to(dt)
            g = torch.ones(C, device=dev)
            b = torch.zeros(C, device=dev)
            t = timeit(lambda: hip.ln_corr(feat, g, b), args.iters)
            by = B * (2 * h * w * C * e + h * w * w * e)
            fl = 2.0 * B * h * w * w * C
            res.append(dict(kernel="ln_corr", case=name, dtype=str(dt), us=t, GBs=by / t / 1e3, TFs=fl / t / 1e6))
            print(f"ln_corr   {name:22s} {str(dt):14s} {t:9.1f} us  {by / t / 1e3:8.1f} GB/s  {fl / t / 1e6:7.1f} TF/s", flush=True)
            cv = hip.ln_corr(feat, g, b)
            for pos in (True, False):
                t = timeit(lambda: hip.sinkhorn_regress(cv, pos), max(5, args.iters // 5))
                by = B * (h * w * w * e + 3 * h * w * 4)
                res.append(dict(kernel="sinkhorn", case=name, dtype=str(dt), pos=pos, us=t, GBs=by / t / 1e3))
                print(f"sinkhorn  {name:22s} {str(dt):14s} pos={int(pos)} {t:9.1f} us  {by / t / 1e3:8.1f} GB/s (1 read of cv)", flush=True)
            disp = torch.rand(B, 1, h, w, device=dev) * 60
            t = timeit(lambda: hip.cv_lookup(cv, disp, 4, True, dt), args.iters)
            res.append(dict(kernel="cv_lookup", case=name, dtype=str(dt), us=t))
            print(f"cv_lookup {name:22s} {str(dt):14s} {t:9.1f} us", flush=True)
            del feat, cv
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/kbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
