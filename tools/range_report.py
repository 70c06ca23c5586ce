#!/usr/bin/env python3
"""fp16 headroom of a forward: max |value| that every launch converts to fp16, against the fp16 maximum 65504.

Every parity number of this repository is on seeded random weights (no pretrained checkpoint ships with the reference); a real CH{C}NTR{n}.pth
may drive activations somewhere else.  This tool runs ONE eager fp16 forward on the probe build of the library (-DS2M2_RANGE_CHECK=1: every
fp32 -> fp16 conversion folds |x| into a device word, logged per launch) and prints, per launch, the largest magnitude BEFORE rounding and its
ratio to 65504 -- a ratio >= 1 (or NaN) names the layer that overflows in fp16, before the value turns into inf downstream.

    S2M2_LIB_SUFFIX=_range S2M2_BUILD_DEFINES=-DS2M2_RANGE_CHECK=1 python -m s2m2_amd.build        (build container, once)
    S2M2_LIB_SUFFIX=_range python tools/range_report.py [--model S] [--height 1024 --width 1216] [--ckpt CH128NTR1.pth] [--top 25]
"""
import argparse
import ctypes
import os
import sys

os.environ.setdefault("S2M2_LIB_SUFFIX", "_range")
os.environ["S2M2_GRAPH"] = "0"                 # eager: the probe logs per launch
os.environ["S2M2_REFINE_NATIVE"] = "0"         # every launch through its entry point (a replayed plan bypasses the dispatch hook)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from s2m2_amd import hip  # noqa: E402
from s2m2_amd.model import build_model  # noqa: E402
from s2m2_amd.weights import noise_pair, synthetic_pair  # noqa: E402

FP16_MAX = 65504.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="S")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1216)
    ap.add_argument("--ckpt", default="")
    ap.add_argument("--images", default="textured", choices=["textured", "noise"])
    ap.add_argument("--negative", action="store_true", help="use_positivity=False")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    lib = hip.load()
    if not hasattr(lib, "s2m2_debug_range_log"):
        sys.exit("this library has no range probe: build it with S2M2_LIB_SUFFIX=_range S2M2_BUILD_DEFINES=-DS2M2_RANGE_CHECK=1")
    lib.s2m2_debug_range_log.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.s2m2_debug_range_name.restype = ctypes.c_char_p
    lib.s2m2_debug_range_name.argtypes = [ctypes.c_int]
    m = build_model(a.model, use_positivity=not a.negative, refine_iter=3)
    if a.ckpt:
        sd = torch.load(a.ckpt, map_location="cpu")
        m.my_load_state_dict(sd.get("state_dict", sd))
    m = m.cuda().eval()
    l, r = (synthetic_pair(a.height, a.width, 1, 32, 0) if a.images == "textured" else noise_pair(a.height, a.width, 1, 0))
    l, r = l.cuda(), r.cuda()
    # labels: the engine methods on the Python stack at every library call, in launch order
    labels = []
    import s2m2_amd.engine as E
    efile = E.__file__

    def wrap(name, fn):
        def f(*args):
            fr, path = sys._getframe(1), []
            while fr is not None:
                if fr.f_code.co_filename == efile and fr.f_code.co_name not in ("cconv", "run", "finish", "features"):
                    p = fr.f_locals.get("p")
                    path.append(fr.f_code.co_name + (f"[{p}]" if isinstance(p, str) else ""))
                fr = fr.f_back
            labels.append((name, " < ".join(path[:3])))
            return fn(*args)
        return f
    recorded = {"s2m2_conv2d", "s2m2_mlp_chain", "s2m2_pw_direct", "s2m2_conv_narrow", "s2m2_cost_volume", "s2m2_attention", "s2m2_feature_fusion",
                "s2m2_feature_fusion_frag", "s2m2_cv_lookup", "s2m2_sinkhorn_regress", "s2m2_refine_prep", "s2m2_global_update", "s2m2_refine_update_to",
                "s2m2_tanh", "s2m2_convex_upsample", "s2m2_resample2x", "s2m2_groupnorm_nhwc", "s2m2_layernorm", "s2m2_image_prep", "s2m2_stem_mlp",
                "s2m2_image_pad"}
    with torch.autocast("cuda", dtype=torch.float16):
        m(l, r)                                                    # warm-up: packing, attributes
        torch.cuda.synchronize()
        lib.s2m2_debug_range_reset()
        for name in recorded:
            setattr(lib, name, wrap(name, getattr(lib, name)))
        d, o, c = m(l, r)
    torch.cuda.synchronize()
    buf = np.zeros(4096, dtype=np.float32)
    n = lib.s2m2_debug_range_log(buf.ctypes.data, buf.size)
    names = [lib.s2m2_debug_range_name(i).decode() for i in range(n)]
    if [x[0] for x in labels] != names:
        print(f"# warning: {len(labels)} labelled calls vs {n} logged launches: labels are omitted where the sequences differ")
    print(f"# {a.model}-model {a.width}x{a.height} fp16, {'checkpoint ' + a.ckpt if a.ckpt else 'seeded random weights'}, {a.images} images: {n} launches; "
          f"outputs finite: {bool(torch.isfinite(d).all() and torch.isfinite(o).all() and torch.isfinite(c).all())}")
    rows = []
    for i in range(n):
        lab = labels[i][1] if i < len(labels) and labels[i][0] == names[i] else ""
        rows.append((float(buf[i]), i, names[i], lab))
    worst = sorted(rows, key=lambda t: -(t[0] if t[0] == t[0] else float("inf")))
    print(f"# largest magnitude converted to fp16 over the whole forward: {worst[0][0]:.4g} = {worst[0][0] / FP16_MAX:.3g} of 65504 (launch #{worst[0][1]} {worst[0][2]})")
    print(f"{'launch':>6} {'max |x|':>12} {'/ 65504':>10}  entry point / engine context")
    for v, i, nm, lab in worst[:a.top]:
        flag = "  <-- OVERFLOW" if not (v < FP16_MAX) else ("  <-- above 1/8 of the range" if v > FP16_MAX / 8 else "")
        print(f"{i:6d} {v:12.5g} {v / FP16_MAX:10.3g}  {nm[5:]:22s} {lab}{flag}")
    by = {}
    for v, i, nm, lab in rows:
        by[nm] = max(by.get(nm, 0.0), v)
    print("# per entry point: " + ", ".join(f"{k[5:]} {v:.4g}" for k, v in sorted(by.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
