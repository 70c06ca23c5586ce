#!/usr/bin/env python3
"""Why does the 3x3 128->128 layer take 50 us inside the forward and 38 us in the L2-cold micro-benchmark?

Measures the average ENGINE CLOCK a region of the stream ran at: s2m2_debug_clock_probe stamps the shader-clock counter (s_memtime, ticks
at the CU's current clock) and the constant 100 MHz real-time counter around the region; ratio x 100 MHz = clock.  Regions:

  * the layer back to back (hot: the matrix pipes never rest),
  * the layer with an L2-evicting memory-bound kernel between calls (the "cold" micro-benchmark: the chip rests between layers),
  * one whole forward (hipGraph replay), and the same layer timed in both regimes.

    python tools/clock_probe.py            # prints a table, used for profiles/r03/clock_probe.txt
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from s2m2_amd.model import build_model  # noqa: E402
from s2m2_amd.weights import noise_pair  # noqa: E402


def region(fn, reps):
    """-> (us per rep, MHz) of `reps` calls of fn on the current stream"""
    a = torch.zeros(2, dtype=torch.int64, device="cuda")
    b = torch.zeros(2, dtype=torch.int64, device="cuda")
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hip.clock_probe(a)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    hip.clock_probe(b)
    torch.cuda.synchronize()
    d = (b - a).tolist()
    return 1e3 * e0.elapsed_time(e1) / reps, 100.0 * d[0] / max(1, d[1])


def main():
    hip.load()
    dt = torch.float16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1, 256, 304, 128, device="cuda", generator=g).to(dt)
    w = (torch.randn(128, 128, 3, 3, device="cuda", generator=g) / math.sqrt(1152)).to(dt)
    wp = pack.pack_conv_frag(w, dt, None)
    b = torch.zeros(128, device="cuda")
    evict = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")          # larger than L2 + Infinity Cache

    def layer():
        return hip.conv2d([x], wp, b, 3, 3, 128, act=hip.ACT_GELU, korder=2)

    def layer_cold():
        evict.add_(1)                                                         # memory-bound: the matrix pipes rest, L2 / MALL are flushed
        return layer()

    def graphed(fn, n):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n):
                fn()
        return gr.replay

    hot = graphed(layer, 50)
    cold = graphed(layer_cold, 50)
    us_evict, _ = region(graphed(lambda: evict.add_(1), 50), 4)
    us_hot, mhz_hot = region(hot, 20)
    us_cold, mhz_cold = region(cold, 4)
    print("3x3 128->128 fp16 + GELU at 256x304 (K5 v5), 50 launches per graph replay")
    print(f"  back to back:                {us_hot / 50:7.1f} us per layer   engine clock {mhz_hot:6.0f} MHz")
    print(f"  L2-evicting kernel between:  {(us_cold - us_evict) / 50:7.1f} us per layer   engine clock {mhz_cold:6.0f} MHz (average over layer + evict kernel)")
    model = build_model("S", use_positivity=True, refine_iter=3).cuda().eval()
    l, r = (t.cuda() for t in noise_pair(1024, 1216, 1, 0))
    with torch.autocast("cuda", dtype=torch.float16):
        for _ in range(3):
            model(l, r)

        def fwd():
            return model(l, r)
        us_f, mhz_f = region(fwd, 30)
    print(f"  whole forward (S 1216x1024 fp16 r=3, graph replay): {us_f / 1e3:6.3f} ms   engine clock {mhz_f:6.0f} MHz")
    print(f"  the layer's back-to-back time scaled to the forward's clock: {us_hot / 50 * mhz_hot / mhz_f:6.1f} us")
    for name in ("rocm-smi",):
        pass


if __name__ == "__main__":
    main()
