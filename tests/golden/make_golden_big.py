"""Reference outputs at BASELINE.json's two large-model geometries, generated in the build container on the CPU:

    python tests/golden/make_golden_big.py c4      # L  (CH=256, NTR=3) 1216x1024 fp32 refine_iter 3, use_positivity          (~1-2 min,  6 GB)
    python tests/golden/make_golden_big.py c5      # XL (CH=384, NTR=3) 2432x2048 fp32 refine_iter 1, allow_negative            (~8 min,  21 GB)

needs /root/reference; writes tests/golden/e2e_{L_1216x1024_fp32_r3,XL_2432x2048_fp32_r1_neg}_sub.npz.

Stored per file (everything is an output of the UNMODIFIED reference module, captured with forward hooks):

* ``feature_tr_4x`` at every ``fsub``-th 1/4-resolution pixel in both directions (the input of DispInit: the whole backbone,
  feature pyramid and multi-resolution transformer at the widths C = 256 / 384 are upstream of it);
* ``cv`` on every ``cvsub``-th image row (complete rows ``cv[b, y, :, :]``);
* the DispInit outputs ``disp0 / conf0 / occ0`` complete, plus ``gap0`` = top-2 relative gap of the reference's masked
  transport probabilities per pixel (fp16; where it is below ~1e-4 ... 1e-3 the integer argmax is decided by fp32 summation order of
  the C-term correlation sums, SURVEY 8c: the test derives its threshold from the cost-volume error it measures) and
  ``argmax`` = prob_max_ind, both recomputed inside the hook with the module's own ``_optimal_transport`` on blocks of 16 image rows
  (DispInit does not return them; the transport problem of an image row is independent of the other rows);
* ``disp_g`` (GlobalRefiner + clamp) at every ``gsub``-th and the final ``disp / occ / conf`` at every ``sub``-th pixel in both directions.

The GPU test (tests/test_reference_golden_big.py) runs DispInit free, then continues from the reference's own disp0 / conf0 / occ0
(Engine's ``inject`` hook), exactly like tests/test_reference_golden_c3.py does for the S model.

Weights: ``seeded_state_dict(C, 1, ntr, seed, gain)``.  LeCun gain 1.0 at these widths lets the residual streams of the 3 stacked
MRTs grow until the global refiner's x100 output saturates (disparities of 10^4 px on a 1216-px image: measured with THIS script,
the reference does the same as the HIP path, see profiles/r03/golden_big_ranges.txt); the stored runs use the gain given in CFG so
that the disparities stay inside the image and the tolerances mean something.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import s2m2.core.model.s2m2 as ref_s2m2  # noqa: E402  (reference, read-only)

from s2m2_amd.weights import seeded_state_dict, synthetic_pair  # noqa: E402

torch.set_num_threads(8)

#            C   ntr  H     W     pos    ri seed disparity gain  sub fsub cvsub  file
CFG = {
    "c4": (256, 3, 1024, 1216, True, 3, 1, 48, 0.9, 4, 8, 64, "e2e_L_1216x1024_fp32_r3_sub.npz"),
    "c5": (384, 3, 2048, 2432, False, 1, 1, 48, 0.9, 8, 32, 256, "e2e_XL_2432x2048_fp32_r1_neg_sub.npz"),
}


def run(name: str, gain_override=None, dry: bool = False):
    C, ntr, H, W, pos, ri, seed, disparity, gain, sub, fsub, cvsub, fname = CFG[name]
    if gain_override is not None:
        gain = gain_override
    sd = seeded_state_dict(C, 1, ntr, seed, gain=gain)
    model = ref_s2m2.S2M2(C, 1, ntr, use_positivity=pos, output_upsample=False, refine_iter=ri).eval()
    model.load_state_dict(sd, strict=True)
    left, right = synthetic_pair(H, W, 1, disparity, seed)
    cap = {}

    def di_hook(m, inp, out):
        tr = inp[0]
        cv = out[3]
        cap.update(feature_tr_4x=tr[:, :, ::fsub, ::fsub].float().clone(), cv=cv[:, ::cvsub].float().clone(),
                   disp0=out[0].float().clone(), conf0=out[1].float().clone(), occ0=out[2].float().clone())
        # top-2 gap of the masked probabilities, row by row (the full prob tensor is transient inside DispInit: recompute per row block)
        w = cv.shape[-1]
        gaps = torch.empty(cv.shape[:3])
        amax = torch.empty(cv.shape[:3], dtype=torch.int32)
        tri = torch.triu(torch.ones(w, w, dtype=torch.bool), 1)
        for y0 in range(0, cv.shape[1], 16):
            blk = cv[:, y0:y0 + 16].float()
            p = m._optimal_transport(blk.masked_fill(tri, -1e4) if pos else blk)          # submodules.py:221-223 on a block of rows
            if pos:
                p = p.masked_fill(tri, 0)
            top = p.topk(2, dim=-1).values
            gaps[:, y0:y0 + 16] = (top[..., 0] - top[..., 1]) / top[..., 0].clamp_min(1e-30)
            amax[:, y0:y0 + 16] = p.argmax(dim=3).int()                                      # submodules.py:226
        cap["gap0"] = gaps[:, None]
        cap["argmax"] = amax

    model.disp_init.register_forward_hook(di_hook)
    model.global_refiner.register_forward_hook(lambda m, i, o: cap.update(disp_g=(o.clamp(min=0) if pos else o)[..., ::1, ::1].float().clone()))
    t0 = time.time()
    with torch.inference_mode():
        d, o, c = model(left, right)
    dt = time.time() - t0
    rng = {k: (float(v.min()), float(v.max())) for k, v in dict(disp0=cap["disp0"], disp_g=cap["disp_g"], disp=d, occ=o, conf=c).items()}
    print(f"{name}: gain {gain} {dt:.0f} s  ranges {rng}  finite {bool(torch.isfinite(d).all())}", flush=True)
    print(f"   feature_tr_4x |mean| {float(cap['feature_tr_4x'].abs().mean()):.3g} max {float(cap['feature_tr_4x'].abs().max()):.3g}; "
          f"cv range [{float(cap['cv'].min()):.4g}, {float(cap['cv'].max()):.4g}]; gap0<1e-4 on {float((cap['gap0'] < 1e-4).float().mean()):.2e} of pixels",
          flush=True)
    if dry:
        return
    gsub = max(1, sub // 4)
    out = dict(cfg=np.array([C, ntr, H, W, 1, int(pos), ri, disparity, seed]), gain=np.array(gain), sub=np.array(sub), fsub=np.array(fsub),
               cvsub=np.array(cvsub),
               feature_tr_4x=cap["feature_tr_4x"].numpy(), cv=cap["cv"].numpy(),
               gap0=cap["gap0"].clamp(max=1.0).numpy().astype(np.float16), argmax=cap["argmax"].numpy().astype(np.int16),
               disp0=cap["disp0"].numpy(), conf0=cap["conf0"].numpy(), occ0=cap["occ0"].numpy(),
               disp_g=cap["disp_g"][..., ::gsub, ::gsub].numpy(), gsub=np.array(gsub),
               disp=d[..., ::sub, ::sub].float().numpy(), occ=o[..., ::sub, ::sub].float().numpy(), conf=c[..., ::sub, ::sub].float().numpy(),
               torch_version=np.array(torch.__version__))
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **out)
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB", {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim > 1})


if __name__ == "__main__":
    which = sys.argv[1]
    gain = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "dry" else None
    run(which, gain, dry="dry" in sys.argv)
