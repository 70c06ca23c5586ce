#!/usr/bin/env python3
"""A/B of K1 (ln_corr) launch variants at BASELINE sizes on the GPU box: one subprocess per environment setting (the library
reads its tuning knobs once).  Prints time per launch (hipGraph-timed, back to back and with the L2s evicted before each call),
algorithmic GB/s and a checksum of the cost volume (all variants must agree bit for bit).

    python tools/k1_variants.py            # parent: runs every variant
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {"c2": (128, 120, 160, 1), "c3": (128, 256, 304, 1), "c3B4": (128, 256, 304, 4), "c4": (256, 256, 304, 1), "c5": (384, 512, 608, 1)}


def child(cases, dtypes):
    import torch
    from s2m2_amd import hip
    from tools.kbench import timeit_graph, timeit_graph_cold
    hip.load()
    for case in cases:
        C, h, w, B = CASES[case]
        for dn in dtypes:
            dt = torch.float16 if dn == "fp16" else torch.float32
            if dt == torch.float32 and C > 128:
                continue
            torch.manual_seed(0)
            feat = (torch.randn(2 * B, h, w, C, device="cuda") * 1.5 + 0.2).to(dt)
            g = 1 + 0.1 * torch.randn(C, device="cuda")
            b = 0.02 * torch.randn(C, device="cuda")
            out = torch.empty((B, h, w, w), device="cuda", dtype=dt)
            fn = lambda: hip.ln_corr(feat, g, b, out=out)          # noqa: E731
            fn()
            torch.cuda.synchronize()
            digest = hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:10]
            t_hot = timeit_graph(fn, 20, 5)
            t_cold = timeit_graph_cold(fn, 20, 5)
            e = 2 if dt == torch.float16 else 4
            nbytes = B * (2 * h * w * C * e + h * w * w * e)
            print(f"  {case:5s} {dn}  hot {t_hot:7.2f} us {nbytes / t_hot / 1e3:7.0f} GB/s   L2-cold {t_cold:7.2f} us {nbytes / t_cold / 1e3:7.0f} GB/s   md5 {digest}", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2].split(","), sys.argv[3].split(","))
    variants = [("one block per row (default)", {}),
                ("two half-row blocks per row (S2M2_LNCORR_NSTRIP=2: 5 waves, 66 KB of LDS each)", {"S2M2_LNCORR_NSTRIP": "2"})]
    cases = sys.argv[1] if len(sys.argv) > 1 else "c2,c3,c3B4"
    dtypes = sys.argv[2] if len(sys.argv) > 2 else "fp16,fp32"
    for name, env in variants:
        print(f"== {name}", flush=True)
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cases, dtypes], env=e, cwd=ROOT)


if __name__ == "__main__":
    main()
