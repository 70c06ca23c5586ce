"""fp16 deployment-mode goldens of the UNMODIFIED reference at BASELINE.json's large configurations, generated in the build container (CPU):

    python tests/golden/make_golden_fp16_big.py c3     # S  1216x1024 refine_iter 3 use_positivity   (the headline config)     ~2 min
    python tests/golden/make_golden_fp16_big.py c4     # L  1216x1024 refine_iter 3 use_positivity                              ~8 min
    python tests/golden/make_golden_fp16_big.py c5     # XL 2432x2048 refine_iter 3 allow_negative                             ~35 min, 22 GB
    python tests/golden/make_golden_fp16_big.py m      # M  640x480   refine_iter 3 use_positivity                              ~1 min

needs /root/reference; writes tests/golden/e2e_<name>_fp16_r3_sub.npz.  Every stored array is an output of the reference's own modules.

Two legs per configuration, each run twice -- fp32, and the way run_stereo_matching deploys the model (model_utils.py:75-76:
``torch.amp.autocast(device_type=device.type, dtype=float16)``, here with device cpu: the one fp16 mode of the reference that exists
in this container):

* ``n_*``  NATURAL run on the seeded textured pair: final maps at every ``sub``-th pixel (fp32 and fp16 run).  fp16 runs that round at
  different points agree only statistically (near-tie argmax flips), so the fp16 maps are a YARDSTICK: the HIP fp16 forward must sit as
  close to the reference's fp16 maps as the reference's own fp32 run does (tests/test_fp16_reference_autocast.py's criterion, at scale).
  For c5 the fp32 run is also the refine_iter-3 golden the fp32 parity test continues to (DispInit outputs: the r1 golden of
  make_golden_big.py, same weights / pair).
* ``s_*``  SHARP leg: the transformer output (``feature_tr_4x``, the input of DispInit, s2m2.py:150) is replaced by synthetic tokens with ONE
  unambiguous match per pixel (tests/parity_util.sharp_tokens: top-2 gap of the transport probabilities > 0.5) by overriding
  ``model.transformer.forward`` -- the only deviation from the unmodified module, at the same boundary Engine's ``inject`` hook uses; the
  model runs without the positivity mask (circular shifts: wrapped matches are negative disparities).  No argmax can flip, so the fp16
  forward is held TIGHT: integer argmax (bit exact), complete cost-volume rows of the fp16 run (one fp16 ulp), DispInit outputs and
  final maps of both runs.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference/src")

import s2m2.core.model.s2m2 as ref_s2m2  # noqa: E402  (reference, read-only)

from parity_util import sharp_tokens  # noqa: E402
from s2m2_amd.weights import seeded_state_dict, synthetic_pair  # noqa: E402

torch.set_num_threads(8)

#            C   ntr  H     W     pos    ri seed disp gain sub sub0 cvsub shifts            file
CFG = {
    "c3": (128, 1, 1024, 1216, True, 3, 1, 48, 1.0, 4, 1, 32, (37, 5, 120, 64), "e2e_S_1216x1024_fp16_r3_sub.npz"),
    "c4": (256, 3, 1024, 1216, True, 3, 1, 48, 0.9, 4, 1, 64, (37, 5, 120, 64), "e2e_L_1216x1024_fp16_r3_sub.npz"),
    "c5": (384, 3, 2048, 2432, False, 3, 1, 48, 0.9, 8, 2, 256, (37, 5, 300, 64), "e2e_XL_2432x2048_fp16_r3_sub.npz"),
    "m": (192, 2, 480, 640, True, 3, 0, 32, 0.9, 2, 1, 30, (12, 5, 23), "e2e_M_640x480_fp16_r3_sub.npz"),
}


def _run(model, left, right, fp16):
    with torch.inference_mode():
        if fp16:
            with torch.amp.autocast(enabled=True, device_type="cpu", dtype=torch.float16):      # model_utils.py:76 with device cpu
                return model(left, right)
        return model(left, right)


def _dispinit_hook(cap, cvsub, pos):
    """forward hook of DispInit: its outputs + prob_max_ind and the smallest top-2 relative gap, recomputed with the module's own
    ``_optimal_transport`` on blocks of 16 image rows in the dtype the module itself ran in (submodules.py:221-226)."""
    def hook(m, inp, out):
        cv = out[3]
        w = cv.shape[-1]
        amax = torch.empty(cv.shape[:3], dtype=torch.int32)
        mingap = 1.0
        tri = torch.triu(torch.ones(w, w, dtype=torch.bool), 1)
        for y0 in range(0, cv.shape[1], 16):
            blk = cv[:, y0:y0 + 16]
            p = m._optimal_transport(blk.masked_fill(tri, -1e4) if pos else blk)
            if pos:
                p = p.masked_fill(tri, 0)
            top = p.float().topk(2, dim=-1).values
            mingap = min(mingap, float(((top[..., 0] - top[..., 1]) / top[..., 0].clamp_min(1e-30)).min()))
            amax[:, y0:y0 + 16] = p.argmax(dim=3).int()
        cap.update(cv=cv[:, ::cvsub].clone(), disp0=out[0].clone(), conf0=out[1].clone(), occ0=out[2].clone(), argmax=amax, mingap=mingap)
    return hook


def run(name: str):
    C, ntr, H, W, pos, ri, seed, disparity, gain, sub, sub0, cvsub, shifts, fname = CFG[name]
    sd = seeded_state_dict(C, 1, ntr, seed, gain=gain)
    left, right = synthetic_pair(H, W, 1, disparity, seed)
    out = dict(cfg=np.array([C, ntr, H, W, 1, int(pos), ri, disparity, seed]), gain=np.array(gain), sub=np.array(sub), sub0=np.array(sub0),
               cvsub=np.array(cvsub), shifts=np.array(shifts), tok_seed=np.array(4), torch_version=np.array(torch.__version__))
    # ---- natural leg
    model = ref_s2m2.S2M2(C, 1, ntr, use_positivity=pos, output_upsample=False, refine_iter=ri).eval()
    model.load_state_dict(sd, strict=True)
    for tag, fp16 in (("32", False), ("16", True)):
        t0 = time.time()
        d, o, c = _run(model, left, right, fp16)
        print(f"{name}: natural fp{tag} {time.time() - t0:.0f} s  disp [{float(d.min()):.1f}, {float(d.max()):.1f}] finite {bool(torch.isfinite(d).all())}", flush=True)
        out[f"n_disp_{tag}"] = d[..., ::sub, ::sub].float().numpy()
        out[f"n_occ_{tag}"] = o[..., ::sub, ::sub].half().numpy()
        out[f"n_conf_{tag}"] = c[..., ::sub, ::sub].half().numpy()
    dd = (torch.as_tensor(out["n_disp_16"]) - torch.as_tensor(out["n_disp_32"])).abs().flatten()
    print(f"   reference fp16 vs fp32, disparity: median {float(dd.median()):.4f} p90 {float(dd.kthvalue(int(0.9 * dd.numel()))[0]):.4f} "
          f"p99 {float(dd.kthvalue(int(0.99 * dd.numel()))[0]):.4f} px", flush=True)
    del model
    # ---- sharp leg (no positivity mask; the transformer's output replaced at the DispInit boundary)
    model = ref_s2m2.S2M2(C, 1, ntr, use_positivity=False, output_upsample=False, refine_iter=ri).eval()
    model.load_state_dict(sd, strict=True)
    tok = sharp_tokens(C, H // 4, W // 4, shifts, 4)
    cap = {}
    model.disp_init.register_forward_hook(_dispinit_hook(cap, cvsub, False))
    for tag, fp16 in (("32", False), ("16", True)):
        model.transformer.forward = lambda *a, _t=(tok.half() if fp16 else tok), **k: _t     # feature_tr_4x := sharp tokens (s2m2.py:150)
        t0 = time.time()
        d, o, c = _run(model, left, right, fp16)
        print(f"{name}: sharp fp{tag} {time.time() - t0:.0f} s  min top-2 gap {cap['mingap']:.3f}  disp [{float(d.min()):.1f}, {float(d.max()):.1f}]", flush=True)
        out[f"s_argmax_{tag}"] = cap["argmax"].numpy().astype(np.int16)
        out[f"s_mingap_{tag}"] = np.array(cap["mingap"])
        s0 = (slice(None), slice(None), slice(None, None, sub0), slice(None, None, sub0))
        out[f"s_disp0_{tag}"] = cap["disp0"][s0].float().numpy()
        out[f"s_conf0_{tag}"] = cap["conf0"][s0].float().numpy() if not fp16 else cap["conf0"][s0].half().numpy()
        out[f"s_occ0_{tag}"] = cap["occ0"][s0].float().numpy() if not fp16 else cap["occ0"][s0].half().numpy()
        if fp16:
            assert cap["cv"].dtype == torch.float16, cap["cv"].dtype
            out["s_cv_16"] = cap["cv"].numpy()
        out[f"s_disp_{tag}"] = d[..., ::sub, ::sub].float().numpy()
        out[f"s_occ_{tag}"] = o[..., ::sub, ::sub].half().numpy()
        out[f"s_conf_{tag}"] = c[..., ::sub, ::sub].half().numpy()
    assert bool((torch.as_tensor(out["s_argmax_16"]) == torch.as_tensor(out["s_argmax_32"])).all()), "sharp tokens: fp16 and fp32 argmax differ"
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **out)
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    for which in sys.argv[1:]:
        run(which)
