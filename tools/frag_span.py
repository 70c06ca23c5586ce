#!/usr/bin/env python3
"""Where does a conv_frag_kernel launch spend its time OUTSIDE the blocks' own timelines?  (experiment build -DS2M2_FRAG_TRACE=1)

Per launch: the blocks' entry / exit on the chip-wide 100 MHz counter (start ramp, lifetime, tail), the CU every block ran on (do blocks b and b + 256
share a CU?), against the launch-to-launch period of the same kernel back to back in a stream and inside a small hipGraph.
    S2M2_LIB_SUFFIX=_fragtrace S2M2_BUILD_DEFINES=-DS2M2_FRAG_TRACE=1 python -m s2m2_amd.build      (build container)
    S2M2_LIB_SUFFIX=_fragtrace python tools/frag_span.py                                              (GPU box)"""
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402

NS = 10


def run(N, H, W, ci, co, kh, kw, act, epi, tile=0, other=None):
    lib = hip.load()
    lib.s2m2_debug_frag_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    x = torch.randn(N, H, W, ci, device="cuda").half()
    w = (torch.randn(co, ci, kh, kw, device="cuda") / math.sqrt(ci * kh * kw)).half()
    b = torch.randn(co, device="cuda")
    a0 = torch.rand(N, H, W, co, device="cuda").half() if epi else None
    wf, bp = pack.pack_conv_frag(w, torch.float16), pack.pack_bias(b, co)
    y = torch.empty(N, H, W, co, device="cuda", dtype=torch.float16)

    def launch():
        hip.conv2d([x], wf, bp, kh, kw, co, act=act, epi=epi, aux0=a0, korder=2, tile=tile, out=y)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    # period of the launch back to back (eager) and inside a graph of 20 launches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        launch()
    e1.record()
    torch.cuda.synchronize()
    eager_us = e0.elapsed_time(e1) * 1e3 / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        launch()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20):
                launch()
                if other is not None:
                    other()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    graph_us = e0.elapsed_time(e1) * 1e3 / 100
    assert lib.s2m2_debug_frag_trace_clear() == 0
    launch()
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 4 * NS, dtype=np.uint64)
    assert lib.s2m2_debug_frag_trace(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(4096, 4, NS)
    nb = int((t[:, 0, 6] > 0).sum())
    t = t[:nb]
    rt0 = t[:, :, 7].astype(np.float64) * 0.01                     # us
    rt1 = t[:, :, 8].astype(np.float64) * 0.01
    base = rt0.min()
    st, en = rt0.min(axis=1) - base, rt1.max(axis=1) - base        # per block
    hw = t[:, 0, 9]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
    cu = (hw >> np.uint64(8)) & np.uint64(0xF)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    key = (xcc.astype(np.int64) << 8) | (se.astype(np.int64) << 4) | cu.astype(np.int64)
    ncu = len(set(key.tolist()))
    same = int(sum(1 for b_ in range(nb - 256) if key[b_] == key[b_ + 256])) if nb > 256 else 0
    xcd_rr = float(np.mean(xcc.astype(np.int64) == (np.arange(nb) % 8)))
    print(f"{N}x{H}x{W} {ci}->{co} k{kh}x{kw} act={act} epi={epi} tile={tile or 'auto'}: {nb} blocks on {ncu} distinct CUs; "
          f"block b on XCD b%8: {xcd_rr:.3f}; pairs (b, b+256) on one CU: {same}/{max(nb - 256, 0)}")
    print(f"  period per launch: back to back (eager) {eager_us:7.2f} us   in a hipGraph of 20{' (alternating with another kernel: period of the pair)' if other else ''} {graph_us:7.2f} us")
    q = lambda a, p: float(np.percentile(a, p))
    print(f"  block entry after the first block's   p50 {q(st, 50):6.2f}  p90 {q(st, 90):6.2f}  max {st.max():6.2f} us")
    print(f"  block lifetime (100 MHz counter)      p10 {q(en - st, 10):6.2f}  p50 {q(en - st, 50):6.2f}  p90 {q(en - st, 90):6.2f}  max {(en - st).max():6.2f} us")
    print(f"  block exit after the first entry      p10 {q(en, 10):6.2f}  p50 {q(en, 50):6.2f}  p90 {q(en, 90):6.2f}  max {en.max():6.2f} us   -> outside the grid's span: {graph_us - en.max():6.2f} us of the graph period")
    tk = t[:, :, :7].astype(np.float64)
    clk = np.median((tk[:, :, 6] - tk[:, :, 0]) / np.maximum((rt1 - rt0), 0.01)) / 1e3
    print(f"  shader clock inside the blocks: {clk:.2f} GHz (s_memtime ticks per 100 MHz tick)")
    for name, a, b_ in (("halo tile", 0, 1), ("K loop", 1, 2), ("barrier", 2, 3), ("bias/act/staging", 3, 4), ("barrier", 4, 5), ("aux + stores issued", 5, 6)):
        d = (tk[:, :, b_] - tk[:, :, a]) / (clk * 1e3)
        print(f"    {name:<22} p10 {q(d, 10):6.2f}  p50 {q(d, 50):6.2f}  p90 {q(d, 90):6.2f} us (real)")


if __name__ == "__main__":
    run(1, 256, 304, 128, 128, 3, 3, 1, 0)                         # 512 blocks of 4x40, one round
    run(1, 256, 304, 128, 128, 3, 3, 0, 1, tile=40)
    if "--quick" in sys.argv:
        sys.exit(0)
    run(1, 128, 152, 128, 128, 3, 3, 1, 0)                         # coarse level: 2x32 patches
    run(2, 64, 76, 256, 256, 3, 3, 1, 0)
    run(1, 256, 152, 128, 128, 3, 3, 1, 0)                         # half the grid: 256 blocks, one per CU
    z = torch.randn(1, 256, 304, 128, device="cuda").half()
    run(1, 256, 304, 128, 128, 3, 3, 1, 0, other=lambda: hip.tanh(z))
