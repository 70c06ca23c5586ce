"""Golden of the UNMODIFIED reference in its fp16 deployment mode on its own REAL sample pair at the pair's natural (padded) size:

    python tests/golden/make_golden_web_fp16.py     # needs /root/reference; writes tests/golden/e2e_S_web0025_full_fp16_r3.npz

data/samples/Web/0025_{L,R}.png (800 x 1100 RGB, uint8) -> the reference's image_pad(32) (image_utils.py:27-71; 800 x 1120) -> the reference
module under torch.amp.autocast(device_type="cpu", dtype=float16) (model_utils.py:75-76 with device cpu: the only fp16 mode of the reference
that runs here) AND in fp32 (the yardstick: how far the reference's own fp16 deployment sits from its fp32 run on this pair) -> image_crop
(image_utils.py:73-103).  S model, refine_iter 3, use_positivity, seeded LeCun-normal weights (no checkpoint ships).  Stored: the two uint8
images as the reference's demo reads them, the three final maps of both runs at every 2nd pixel (fp16 storage for occ / conf), the average
confidence of both runs (model_utils.py:93-94).  tests/test_fp16_headline.py::test_web0025_full_size_fp16 reads it.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
for name in ("cv2", "open3d"):                                      # import-time dependencies of the utility modules only
    sys.modules.setdefault(name, types.ModuleType(name))

from s2m2.core.utils.image_utils import image_crop, image_pad      # noqa: E402  (reference, read-only)
import s2m2.core.model.s2m2 as ref_s2m2                             # noqa: E402

from s2m2_amd.weights import seeded_state_dict                      # noqa: E402

torch.set_num_threads(8)


def main():
    from PIL import Image
    lr = [np.ascontiguousarray(np.asarray(Image.open(f"/root/reference/data/samples/Web/0025_{s}.png").convert("RGB"))) for s in "LR"]
    H, W = lr[0].shape[:2]
    left = torch.from_numpy(lr[0]).permute(2, 0, 1)[None]
    right = torch.from_numpy(lr[1]).permute(2, 0, 1)[None]
    C, ntr, ri, seed = 128, 1, 3, 0
    model = ref_s2m2.S2M2(C, 1, ntr, use_positivity=True, output_upsample=False, refine_iter=ri).eval()
    model.load_state_dict(seeded_state_dict(C, 1, ntr, seed), strict=True)
    lp, rp = image_pad(left, 32), image_pad(right, 32)
    with torch.inference_mode():
        with torch.amp.autocast(enabled=True, device_type="cpu", dtype=torch.float16):
            o16 = model(lp, rp)
        o32 = model(lp, rp)
    crop = lambda t: image_crop(t, (H, W)).squeeze().float()
    d16, oc16, c16 = (crop(t) for t in o16)
    d32, oc32, c32 = (crop(t) for t in o32)
    m = 100
    out = dict(cfg=np.array([C, ntr, H, W, 1, 1, ri, 0, seed]), padded=np.array(lp.shape[-2:]), left=lr[0], right=lr[1],
               disp_fp16=d16[::2, ::2].numpy(), occ_fp16=oc16[::2, ::2].half().numpy(), conf_fp16=c16[::2, ::2].half().numpy(),
               disp_fp32=d32[::2, ::2].numpy(), occ_fp32=oc32[::2, ::2].half().numpy(), conf_fp32=c32[::2, ::2].half().numpy(),
               avg_conf_fp16=np.array(c16[m:-m, m:-m].mean().item()), avg_conf_fp32=np.array(c32[m:-m, m:-m].mean().item()),
               torch_version=np.array(torch.__version__))
    path = os.path.join(HERE, "e2e_S_web0025_full_fp16_r3.npz")
    np.savez_compressed(path, **out)
    dd = (d16 - d32).abs().flatten()
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB; padded {tuple(lp.shape)}; disp fp32 [{float(d32.min()):.1f}, {float(d32.max()):.1f}]; "
          f"reference fp16 (CPU autocast) vs fp32: median {float(dd.median()):.4f} p90 {float(dd.kthvalue(int(0.9 * dd.numel()))[0]):.4f} "
          f"p99 {float(dd.kthvalue(int(0.99 * dd.numel()))[0]):.4f} px; avg conf {out['avg_conf_fp16']:.5f} / {out['avg_conf_fp32']:.5f}")


if __name__ == "__main__":
    main()
