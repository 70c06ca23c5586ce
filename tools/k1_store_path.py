#!/usr/bin/env python3
"""What does the memory system give K1's store phase?  ("commit a store-only microbench of the same 47.3 MB pattern
from 256 single-row blocks".)  Every number is the kernel's own execution time (start / stop HIP events attached to the dispatch):

  * K1 in its three forms (LayerNorm inside / normalised tokens through LDS / streaming on fragment-ordered tokens; dense / 128-byte-aligned rows),
  * its store loop ALONE (s2m2_debug_store_pattern mode 0: same blocks, same waves, same 128-byte segments, no loads, no MFMA),
  * the same bytes as one linear stream of 16-byte stores from 2048 blocks (mode 2) -- the plain write ceiling of the box,
  * non-temporal variants of both (modes 1 / 3; K1 itself: run with S2M2_K1_NT=1).

    python tools/k1_store_path.py [c3|c2|c4]          # prints the table kept as profiles/r03/k1_store_path.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip  # noqa: E402

CASES = {"c2": (128, 120, 160), "c3": (128, 256, 304), "c4": (256, 256, 304)}


def timed(fn, n=30):
    ts = []
    for k in range(n + 5):
        t = hip.KernelTimer()
        fn(t)
        torch.cuda.synchronize()
        if k >= 5:
            ts.append(t.elapsed_us())
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    case = ([a for a in sys.argv[1:] if not a.startswith("-")] or ["c3"])[0]
    C, h, w = CASES[case]
    lib = hip.load()
    torch.manual_seed(0)
    feat = (torch.randn(2, h, w, C, device="cuda") * 1.5).half()
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    normed = torch.nn.functional.layer_norm(feat.float(), (C,)).half()
    dense = torch.empty((1, h, w, w), device="cuda", dtype=torch.float16)
    padded = hip.cv_alloc(1, h, w, torch.float16, "cuda")
    wbytes = h * w * w * 2
    allbytes = wbytes + 2 * h * w * C * 2
    nt = os.environ.get("S2M2_K1_NT", "0")
    print(f"{case}: C={C} h={h} w={w}: cost volume {wbytes / 1e6:.1f} MB, K1 algorithmic bytes {allbytes / 1e6:.1f} MB; S2M2_K1_NT={nt}; median (min) of 30 dispatches")

    def row(name, fn, nbytes):
        med, mn = timed(fn)
        print(f"  {name:<80}{med:7.2f} us ({mn:6.2f})  {nbytes / med / 1e6:6.2f} TB/s of its bytes")

    row("K1 with its LayerNorm, dense rows (s2m2_ln_corr)", lambda t: hip.ln_corr(feat, g, b, out=dense, timer=t), allbytes)
    row("K1 on normalised tokens, dense rows (s2m2_corr)", lambda t: hip.corr(normed, out=dense, timer=t), allbytes)
    row(f"K1 on normalised tokens, rows on 128-byte lines (pitch {padded.stride(2)})", lambda t: hip.corr(normed, out=padded, timer=t), allbytes)

    tiled = hip.TiledTokens.from_rows(normed)
    row("K1 streaming form on fragment-ordered tokens, dense rows (s2m2_corr_tiled)", lambda t: hip.corr_tiled(tiled, out=dense, timer=t), allbytes)
    row(f"K1 streaming form, rows on 128-byte lines (pitch {padded.stride(2)})", lambda t: hip.corr_tiled(tiled, out=padded, timer=t), allbytes)

    def pat(buf, mode):
        def f(t):
            hip._check(lib.s2m2_debug_store_pattern(buf.data_ptr(), h, w, buf.stride(2), mode, hip._stream(), t.start, t.stop), "s2m2_debug_store_pattern")
        return f
    row("store loop of K1 alone, dense rows", pat(dense, 0), wbytes)
    row("store loop of K1 alone, rows on 128-byte lines", pat(padded, 0), wbytes)
    row("   the same, non-temporal stores", pat(padded, 1), wbytes)
    row("the same bytes as a linear stream from 2048 blocks", pat(padded, 2), h * w * padded.stride(2) * 2)
    row("   the same, non-temporal stores", pat(padded, 3), h * w * padded.stride(2) * 2)
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big.zero_()
    e0.record()
    for _ in range(10):
        big.zero_()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"  {'256 MiB fill (torch zero_, events around 10 launches)':<80}{us:7.2f} us           {big.numel() / us / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
