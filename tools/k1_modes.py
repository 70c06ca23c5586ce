#!/usr/bin/env python3
"""Cache policy of K1's cost-volume stores (store_cv in s2m2_amd/csrc/ln_corr.hip, switch S2M2_K1_NT): kernel execution time (start / stop HIP
events attached to the dispatch) of the shipped K1 (s2m2_corr, rows on 128-byte lines) and of K1 with its own LayerNorm, per store mode --
0 default write-back, 1 nt, 2 sc1, 3 sc0 sc1 (write-through), 4 sc0 sc1 nt.  The mode is read once per process, so every mode runs in a
child process.    python tools/k1_modes.py [c3|c2|c4|c5]        -> the table kept as profiles/r04/k1_store_modes.txt"""
import os
import subprocess
import sys

CASES = {"c2": (128, 120, 160), "c3": (128, 256, 304), "c4": (256, 256, 304), "c5": (384, 512, 608)}


def child(case):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from s2m2_amd import hip
    C, h, w = CASES[case]
    torch.manual_seed(0)
    feat = (torch.randn(2, h, w, C, device="cuda") * 1.5).half()
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    normed = torch.nn.functional.layer_norm(feat.float(), (C,)).half()
    cv = hip.cv_alloc(1, h, w, torch.float16, "cuda")
    evict = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")           # larger than the 256 MB Infinity Cache

    def timed(fn, n=30, cold=False):
        ts = []
        for k in range(n + 5):
            if cold:
                evict.zero_()
            t = hip.KernelTimer()
            fn(t)
            torch.cuda.synchronize()
            if k >= 5:
                ts.append(t.elapsed_us())
        ts.sort()
        return ts[len(ts) // 2], ts[0]
    nbytes = h * w * w * 2 + 2 * h * w * C * 2
    ref = hip.corr(normed, out=hip.cv_alloc(1, h, w, torch.float16, "cuda")).clone()
    out = []
    for name, fn in (("s2m2_corr", lambda t: hip.corr(normed, out=cv, timer=t)), ("s2m2_ln_corr", lambda t: hip.ln_corr(feat, g, b, out=cv, timer=t))):
        for cold in (False, True):
            med, mn = timed(fn, cold=cold)
            out.append(f"{name:<13}{'after a 320 MB fill' if cold else 'back to back':<22}{med:7.2f} us (min {mn:6.2f})  {nbytes / med / 1e6:5.2f} TB/s = {nbytes / med / 1e6 / 8:.3f} of 8 TB/s")
    hip.corr(normed, out=cv)
    torch.cuda.synchronize()
    same = bool(torch.equal(cv, ref))
    print(f"mode {os.environ.get('S2M2_K1_NT', '0')}: volume bit-identical to mode-independent reference: {same}")
    for ln in out:
        print("   " + ln)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    case = sys.argv[1] if len(sys.argv) > 1 else "c3"
    C, h, w = CASES[case]
    print(f"{case}: C={C} h={h} w={w} fp16, K1 algorithmic bytes {(h * w * w * 2 + 2 * h * w * C * 2) / 1e6:.1f} MB; median of 30 dispatches; "
          f"S2M2_K1_NT: 0 write-back (default), 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt")
    for mode in ("0", "1", "2", "3", "4"):
        env = dict(os.environ, S2M2_K1_NT=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case], env=env, capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode != 0:
            sys.stdout.write(f"mode {mode}: FAILED\n{r.stderr[-1500:]}\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
