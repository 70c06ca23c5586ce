#!/usr/bin/env python3
"""Per-call breakdown of one eager pass of the engine: every C-ABI entry point the engine uses is wrapped with a pair of
HIP events; calls are grouped by (entry point, shape signature) and sorted by total GPU time.

    python tools/layer_trace.py [--model S] [--h 1024] [--w 1216] [--iters 3] > gpurun_out/layer_trace.txt
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import s2m2_amd.engine as engine_mod  # noqa: E402
from s2m2_amd import hip  # noqa: E402
from s2m2_amd.model import build_model  # noqa: E402
from s2m2_amd.weights import seeded_state_dict  # noqa: E402


class Tracer:
    def __init__(self, real):
        self.real, self.log, self.on = real, [], False

    def __getattr__(self, name):
        f = getattr(self.real, name)
        if not callable(f) or name.startswith("_") or name[0].isupper():
            return f

        def wrapped(*a, **k):
            if not self.on:
                return f(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if self.real.METER is None:
                self.real.METER = {}
            fl0 = sum(v[0] for v in self.real.METER.values())
            e0.record()
            r = f(*a, **k)
            e1.record()
            self.log.append((name, self.sig(name, a, k), e0, e1, sum(v[0] for v in self.real.METER.values()) - fl0))
            return r
        return wrapped

    @staticmethod
    def sig(name, a, k):
        if name == "conv2d":
            srcs, w = a[0], a[1]
            kh, kw, cout = a[3], a[4], a[5]
            s0 = srcs[0]
            return "%dx%d k%dx%d cin=%s->%d s%d act=%d epi=%d%s" % (
                s0.shape[1], s0.shape[2], kh, kw, "+".join(str(s.shape[3]) for s in srcs), cout, k.get("stride", 1),
                k.get("act", 0), k.get("epi", 0), (" shuf" if k.get("shuffle2") else "") + (" frag" if k.get("korder") == 2 else ""))
        shp = [tuple(x.shape) for x in a if torch.is_tensor(x)]
        return " ".join(str(s) for s in shp[:2])


def ab(a):
    """Same-process A/B of an engine switch: alternating passes of two engines built with ENV=v0 / ENV=v1."""
    var, vals = a.ab.split("=")
    v0, v1 = vals.split(",")
    tr = Tracer(hip)
    engine_mod.hip = tr
    m = build_model(a.model, use_positivity=True, refine_iter=a.refine).cuda()
    engs = []
    for v in (v0, v1):
        os.environ[var] = v
        engs.append(engine_mod.Engine(m, torch.float16))
    left = torch.rand(1, 3, a.h, a.w, device="cuda") * 255
    right = torch.rand(1, 3, a.h, a.w, device="cuda") * 255
    for e in engs:
        for _ in range(2):
            e.run(left, right)
    torch.cuda.synchronize()
    logs = [[], []]
    tr.on = True
    for _ in range(a.iters):
        for i, e in enumerate(engs):
            tr.log = logs[i]
            e.run(left, right)
    torch.cuda.synchronize()
    aggs = []
    for lg in logs:
        agg = collections.OrderedDict()
        for name, sig, e0, e1, _fl in lg:
            d = agg.setdefault((name, sig.replace(" frag", "")), [0, 0.0])
            d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3
        aggs.append(agg)
    t0 = sum(v[1] for v in aggs[0].values()) / a.iters
    t1 = sum(v[1] for v in aggs[1].values()) / a.iters
    print(f"{var}={v0}: {t0 / 1e3:.3f} ms per pass   {var}={v1}: {t1 / 1e3:.3f} ms per pass  (eager, event-bracketed)")
    rows = []
    for k, (n, t) in aggs[0].items():
        if k in aggs[1]:
            rows.append(((aggs[1][k][1] - t) / a.iters, k, n // a.iters, t / n, aggs[1][k][1] / aggs[1][k][0]))
    for d, k, n, ta, tb in sorted(rows, key=lambda r: r[0]):
        if abs(d) >= 2:
            print("%-16s %-58s x%3d  %s=%s %7.1f us  %s=%s %7.1f us  delta/pass %+8.1f us" % (k[0], k[1], n, var, v0, ta, var, v1, tb, d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="S"); ap.add_argument("--h", type=int, default=1024); ap.add_argument("--w", type=int, default=1216)
    ap.add_argument("--iters", type=int, default=3); ap.add_argument("--refine", type=int, default=3)
    ap.add_argument("--ab", default="", help="ENV=0,1: two engines in ONE process (same clocks), per-layer comparison of ENV=0 vs ENV=1")
    a = ap.parse_args()
    if a.ab:
        return ab(a)
    tr = Tracer(hip)
    engine_mod.hip = tr
    m = build_model(a.model, use_positivity=True, refine_iter=a.refine).cuda()
    eng = engine_mod.Engine(m, torch.float16)
    left = torch.rand(1, 3, a.h, a.w, device="cuda") * 255
    right = torch.rand(1, 3, a.h, a.w, device="cuda") * 255
    for _ in range(2):
        eng.run(left, right)
    torch.cuda.synchronize()
    tr.on = True
    for _ in range(a.iters):
        eng.run(left, right)
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for name, sig, e0, e1, fl in tr.log:
        d = agg.setdefault((name, sig), [0, 0.0, 0.0])
        d[0] += 1; d[1] += e0.elapsed_time(e1) * 1e3; d[2] += fl
    tot = sum(v[1] for v in agg.values()) / a.iters
    print("total traced GPU time per pass: %.2f ms (eager, event-bracketed; includes ~2-4 us event overhead per call)" % (tot / 1e3))
    by = collections.defaultdict(float)
    for (name, sig), (n, t, fl) in agg.items():
        by[name] += t / a.iters
    for name, t in sorted(by.items(), key=lambda x: -x[1]):
        print("  %-18s %8.1f us  %5.1f%%" % (name, t, 100 * t / tot))
    print()
    # last column: time this row would save per pass if it ran at 700 TFLOP/s (what the large 3x3 layers reach) -- where the MFMA time is lost
    for (name, sig), (n, t, fl) in sorted(agg.items(), key=lambda x: -x[1][1]):
        tf = fl / (t * 1e-6) / 1e12 if fl > 0 else 0.0
        lost = (t - fl / 700e12 * 1e6) / a.iters if fl > 0 else 0.0
        print("%-16s %-58s calls/pass %3d  avg %7.1f us  total %8.1f us  %4.1f%%  %6.1f TF/s  over-700: %7.1f us" %
              (name, sig, n // a.iters, t / n, t / a.iters, 100 * t / a.iters / tot, tf, lost))


if __name__ == "__main__":
    main()
