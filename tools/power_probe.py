#!/usr/bin/env python3
"""Is the forward limited by the package power cap?  Board power and shader clock (hwmon sysfs of the GPU, rocm-smi as a fallback) sampled every
20 ms while one workload is replayed for ~2.5 s:

  idle | the 3x3 128->128 layer at 256x304 (K5 v5) with random / zero operands | K1 | the whole S forward (hipGraph replay)

    python tools/power_probe.py            # prints a table, used for profiles/r05/power_probe.txt
"""
import glob
import json
import math
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402


def hwmon_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for nm in ("power1_average", "power1_input", "power1_cap", "power1_cap_max", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
            p = os.path.join(d, nm)
            if os.path.exists(p) and nm not in out:
                out[nm] = p
    return out


def read_int(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


def smi_once():
    """-> dict of whatever rocm-smi reports (json), or {}"""
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout) if r.stdout.strip().startswith("{") else {"raw": r.stdout[-600:] + r.stderr[-300:]}
    except Exception as e:
        return {"error": repr(e)}


class Sampler(threading.Thread):
    def __init__(self, files, use_smi):
        super().__init__(daemon=True)
        self.files, self.use_smi, self.stop, self.rows, self.smi = files, use_smi, False, [], []

    def run(self):
        while not self.stop:
            if self.files:
                self.rows.append({k: read_int(p) for k, p in self.files.items()})
                time.sleep(0.02)
            if self.use_smi:
                self.smi.append(smi_once())


def stats(vals):
    vals = [v for v in vals if v is not None]
    if not vals:
        return None
    vals.sort()
    return dict(mean=sum(vals) / len(vals), p50=vals[len(vals) // 2], max=vals[-1], n=len(vals))


def run_for(fn, seconds, files, use_smi):
    """replays fn (a callable that enqueues ~1 ms of GPU work) for `seconds`; -> (us per call, samples)"""
    fn()
    torch.cuda.synchronize()
    smp = Sampler(files, use_smi)
    t0 = time.time()
    calls = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    smp.start()
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        calls += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    smp.stop = True
    smp.join(timeout=30)
    return 1e3 * e0.elapsed_time(e1) / max(1, calls), smp


def graph_of(fn, n):
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
    return g


def main():
    hip.load()
    files = hwmon_files()
    print("hwmon files:", {k: v for k, v in files.items()})
    first = smi_once()
    print("rocm-smi:", json.dumps(first)[:900])
    use_smi = not any(k.startswith("power1") for k in files)
    seconds = float(os.environ.get("PROBE_SECONDS", "2.5"))
    work = []
    N, H, W, ci, co = 1, 256, 304, 128, 128
    for mode in ("random", "zero"):
        x = torch.randn(N, H, W, ci, device="cuda").half()
        w = (torch.randn(co, ci, 3, 3, device="cuda") / math.sqrt(ci * 9)).half()
        if mode == "zero":
            x.zero_()
            w.zero_()
        wf, bp = pack.pack_conv_frag(w, torch.float16, [(ci, ci)]), pack.pack_bias(torch.randn(co, device="cuda"), co)
        y = torch.empty(N, H, W, co, device="cuda", dtype=torch.float16)
        g = graph_of(lambda x=x, wf=wf, bp=bp, y=y: hip.conv2d([x], wf, bp, 3, 3, co, act=0, epi=0, korder=2, out=y), 30)
        work.append((f"K5 3x3 128->128 256x304 {mode} operands (30 launches / replay)", g.replay, 30))
    # K1 at the headline geometry
    tok_l = torch.randn(1, 256, 304, 128, device="cuda").half()
    tok_r = torch.randn(1, 256, 304, 128, device="cuda").half()
    try:
        cv = torch.empty(1, 256, 304, 304, device="cuda", dtype=torch.float16)
        tok = torch.cat([tok_l, tok_r], 0)
        g1 = graph_of(lambda: hip.corr(tok, out=cv), 50)
        work.append(("K1 cost volume 256x304x304 (50 launches / replay)", g1.replay, 50))
    except Exception as e:  # the binding's argument names differ between rounds: the probe does not depend on this leg
        print("K1 leg skipped:", repr(e)[:200])
    # the whole forward
    from s2m2_amd.model import build_model
    from s2m2_amd.weights import noise_pair
    m = build_model("S", use_positivity=True, refine_iter=3).cuda().eval()
    left, right = (t.cuda() for t in noise_pair(1024, 1216, 1, 0))

    def fwd():
        with torch.autocast("cuda", dtype=torch.float16):
            return m(left, right)
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    work.append(("S forward 1216x1024 fp16 refine_iter 3 (the module's own graph replay)", fwd, 1))

    time.sleep(1.0)
    smp = Sampler(files, use_smi)
    smp.start()
    time.sleep(1.5)
    smp.stop = True
    smp.join(timeout=30)
    report("idle (after 1 s of rest)", None, smp, files)
    for name, fn, per in work:
        with torch.no_grad():
            us, smp = run_for(fn, seconds, files, use_smi)
        report(name, us / per, smp, files)
        time.sleep(1.0)


def report(name, us, smp, files):
    line = f"{name:78s}"
    if us is not None:
        line += f" {us:9.2f} us/launch"
    pw = None
    for k in ("power1_average", "power1_input"):
        if k in files:
            pw = stats([r[k] for r in smp.rows])
            if pw:
                line += f"  {k} mean {pw['mean'] / 1e6:7.1f} W  max {pw['max'] / 1e6:7.1f} W (n={pw['n']})"
                break
    for k in ("freq1_input", "freq2_input"):
        if k in files:
            f = stats([r[k] for r in smp.rows])
            if f:
                line += f"  {k} mean {f['mean'] / 1e6:7.1f} MHz"
    if "power1_cap" in files:
        line += f"  cap {read_int(files['power1_cap']) / 1e6:.0f} W"
    print(line)
    if smp.smi:
        print("    rocm-smi sample:", json.dumps(smp.smi[len(smp.smi) // 2])[:600])


if __name__ == "__main__":
    main()
