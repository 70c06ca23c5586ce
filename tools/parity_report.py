#!/usr/bin/env python3
"""Per-stage error tables of the HIP forward against the CPU oracle at BASELINE sizes (GPU box).

    python tools/parity_report.py OUT.txt [case ...]        cases: c1 c1r3 c3r1 c2h (default: all four)

c1   = BASELINE configs[0]: S model, 640x480, fp32, refine_iter=1          (HIP fp32 vs oracle fp32)
c1r3 = same, refine_iter=3
c3r1 = S model, 1216x1024 (the benchmark geometry), fp32, refine_iter=1
c2h  = BASELINE configs[1]: S model, 640x480, fp16 deployment mode, refine_iter=3: HIP fp16 vs the oracle's emulation of the
       reference's autocast path, next to the distance of both from the fp32 oracle

The oracle is test infrastructure: this tool (like tests/) uses it only as the checker.
"""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parity_util as PU  # noqa: E402
from oracle import s2m2_oracle as O  # noqa: E402
from s2m2_amd.weights import seeded_state_dict, synthetic_pair  # noqa: E402

CASES = {  # name: (C, ntr, H, W, refine_iter, disparity, seed, fp16)
    "c1": (128, 1, 480, 640, 1, 32, 0, False),
    "c1r3": (128, 1, 480, 640, 3, 32, 0, False),
    "c3r1": (128, 1, 1024, 1216, 1, 48, 1, False),
    "c2h": (128, 1, 480, 640, 3, 32, 0, True),
}


DOWNSTREAM = ["disp_g", "ctx", "hidden", "mask4x", "disp_up4", "mask1x", "disp", "occ", "conf"]


def _only(rows, keep):
    return [r for r in rows if keep(r[0])]


def run_case(name):
    C, ntr, H, W, ri, disp, seed, fp16 = CASES[name]
    sd = seeded_state_dict(C, 1, ntr, seed)
    left, right = synthetic_pair(H, W, 1, disp, seed)
    t0 = time.perf_counter()
    hout, hcap = PU.hip_forward(sd, C, ntr, ri, left, right, fp16)
    t1 = time.perf_counter()
    ocap = {}
    oout = O.forward(sd, left, right, True, ri, False, ocap, precision="fp16" if fp16 else "fp32")
    t2 = time.perf_counter()
    rows, am = PU.compare(hcap, hout, ocap, oout, ri)
    mode = "fp16 (autocast emulation)" if fp16 else "fp32"
    txt = PU.format_table(f"== {name}: S-model {W}x{H} refine_iter={ri} use_positivity=True, HIP {('fp16' if fp16 else 'fp32')} vs oracle {mode}; "
                          f"textured-shift pair d={disp} seed={seed}; torch {torch.__version__}, {torch.get_num_threads()} CPU threads "
                          f"(HIP {t1 - t0:.1f} s incl. packing, oracle {t2 - t1:.1f} s)", rows, am)
    down = lambda n: n in DOWNSTREAM or "_it" in n                                            # noqa: E731
    if not fp16 and am["agree_all"] < 1.0:
        # near-tie argmax flips (allowed: not 'sure' pixels) feed a 2-D global attention: judge the stages after DispInit from
        # the oracle's own disp0 / conf0 / occ0 instead
        inj = {k: ocap[k] for k in ("disp0", "conf0", "occ0")}
        hout2, hcap2 = PU.hip_forward(sd, C, ntr, ri, left, right, fp16, inject=inj)
        rows2, am2 = PU.compare(hcap2, hout2, ocap, oout, ri)
        txt += PU.format_table("-- stages after DispInit, continued from the oracle's disp0/conf0/occ0 (teacher forcing)", _only(rows2, down), am2)
    if fp16:
        ocap32 = {}
        oout32 = O.forward(sd, left, right, True, ri, False, ocap32)
        rows_a, am_a = PU.compare(ocap, oout, ocap32, oout32, ri)
        txt += PU.format_table("-- for scale: oracle fp16 emulation vs oracle fp32 (the reference's own fp16 deployment error)", rows_a, am_a)
        rows_b, am_b = PU.compare(hcap, hout, ocap32, oout32, ri)
        txt += PU.format_table("-- HIP fp16 vs oracle fp32", rows_b, am_b)
        # teacher forcing 1: K1 + K2 from the oracle's fp16 transformer features
        h3o, h3c = PU.hip_forward(sd, C, ntr, ri, left, right, True, inject={"feature_tr_4x": ocap["feature_tr_4x"]})
        rows_c, am_c = PU.compare(h3c, h3o, ocap, oout, ri)
        txt += PU.format_table("-- HIP fp16 K1+K2 from the oracle-fp16 feature_tr_4x (teacher forcing)",
                               _only(rows_c, lambda n: n in ("cv", "disp0", "conf0", "occ0")), am_c)
        # teacher forcing 2: everything after DispInit from the oracle's fp16 cv / disp0 / conf0 / occ0
        inj = {k: ocap[k] for k in ("cv", "disp0", "conf0", "occ0")}
        h4o, h4c = PU.hip_forward(sd, C, ntr, ri, left, right, True, inject=inj)
        rows_d, am_d = PU.compare(h4c, h4o, ocap, oout, ri)
        txt += PU.format_table("-- HIP fp16 stages after DispInit from the oracle-fp16 cv/disp0/conf0/occ0 (teacher forcing)", _only(rows_d, down), am_d)
    return txt


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # the oracle's CPU ops: 128 default threads on the GPU box oversubscribe
    out = sys.argv[1]
    cases = sys.argv[2:] or list(CASES)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        for c in cases:
            txt = run_case(c)
            print(txt, flush=True)
            f.write(txt + "\n")
            f.flush()


if __name__ == "__main__":
    main()
