#!/usr/bin/env python3
"""Does a launch's duration depend on how long the chip has been busy (clock ramp / power averaging)?  One process, one box:

  * the 3x3 128->128 layer at 256x304 (K5 v5): 3 ms bursts from idle (what tools/convperiod.py measures) vs one continuous 2 s run, the
    period and the shader clock (s_memtime ticks per 100 MHz tick, s2m2_debug_clock_probe) of every 30 ms chunk of it;
  * the whole S forward (the module's graph replay): every forward of a continuous 3 s run, grouped by time since the start.

    python tools/sustain_probe.py          # used for profiles/r05/sustain_probe.txt
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.convperiod import period  # noqa: E402
from tools.power_probe import graph_of  # noqa: E402


def chunks(replay, launches_per_replay, replays_per_chunk, seconds):
    """continuous run: -> list of (t_since_start_ms, us_per_launch, MHz) per chunk, read back after the run (no host sync inside)"""
    n = int(seconds * 1e3 / 30) + 1
    stamps = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(n + 1)]
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    replay()
    torch.cuda.synchronize()
    time.sleep(0.5)
    hip.clock_probe(stamps[0])
    evs[0].record()
    for i in range(n):
        for _ in range(replays_per_chunk):
            replay()
        hip.clock_probe(stamps[i + 1])
        evs[i + 1].record()
    torch.cuda.synchronize()
    out = []
    for i in range(n):
        d = (stamps[i + 1] - stamps[i]).tolist()
        out.append((evs[0].elapsed_time(evs[i + 1]), 1e3 * evs[i].elapsed_time(evs[i + 1]) / (replays_per_chunk * launches_per_replay),
                    100.0 * d[0] / max(1, d[1])))
    return out


def show(rows, marks=(0, 1, 2, 3, 5, 8, 12, 20, 30, 45, 60, 80, 100)):
    for i in marks:
        if i < len(rows):
            t, us, mhz = rows[i]
            print(f"    chunk {i:3d}  ends {t:8.1f} ms after the start   {us:9.2f} us   clock {mhz:6.0f} MHz")
    tail = rows[len(rows) // 2:]
    print(f"    second half of the run: {sum(r[1] for r in tail) / len(tail):9.2f} us   clock {sum(r[2] for r in tail) / len(tail):6.0f} MHz")


def main():
    hip.load()
    N, H, W, ci, co = 1, 256, 304, 128, 128
    for mode in ("random", "zero"):
        x = torch.randn(N, H, W, ci, device="cuda").half()
        w = (torch.randn(co, ci, 3, 3, device="cuda") / math.sqrt(ci * 9)).half()
        if mode == "zero":
            x.zero_()
            w.zero_()
        wf, bp = pack.pack_conv_frag(w, torch.float16, [(ci, ci)]), pack.pack_bias(torch.randn(co, device="cuda"), co)
        y = torch.empty(N, H, W, co, device="cuda", dtype=torch.float16)

        def layer(x=x, wf=wf, bp=bp, y=y):
            hip.conv2d([x], wf, bp, 3, 3, co, act=0, epi=0, korder=2, out=y)
        time.sleep(0.5)
        print(f"K5 3x3 128->128 256x304 act=0, {mode} operands")
        print(f"  bursts of 100 launches from idle (tools/convperiod.period, best of 3): {period(layer):7.2f} us")
        g = graph_of(layer, 30)
        print("  continuous run, 30 launches per replay, 30 replays per chunk:")
        show(chunks(g.replay, 30, 30, 2.0))
    from s2m2_amd.model import build_model
    from s2m2_amd.weights import noise_pair
    m = build_model("S", use_positivity=True, refine_iter=3).cuda().eval()
    left, right = (t.cuda() for t in noise_pair(1024, 1216, 1, 0))

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return m(left, right)
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    time.sleep(1.0)
    print("S forward 1216x1024 fp16 refine_iter 3 (graph replay), continuous run after 1 s of rest, 4 forwards per chunk:")
    show(chunks(fwd, 1, 4, 3.0))


if __name__ == "__main__":
    main()
