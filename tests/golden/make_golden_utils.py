"""Goldens of the reference's REAL pre/post-processing functions and of its driver glue on REAL image data, generated
in the build container on the CPU:

    python tests/golden/make_golden_utils.py         # needs /root/reference; writes tests/golden/utils_pad_crop.npz, e2e_S_web0025_crop_fp32_r1.npz

* ``image_pad`` / ``image_crop`` (/root/reference/src/s2m2/core/utils/image_utils.py:27-103) are imported from the reference with ``cv2``
  stubbed in ``sys.modules`` (the module imports OpenCV for its file readers; neither function touches it) and run on seeded inputs of
  five shapes, uint8 included.  The inputs are stored beside the outputs (the 375 x 1242 output at every 4th pixel).
* ``run_stereo_matching`` (model_utils.py:51-95; ``open3d`` stubbed likewise) cannot run on a CPU-only torch build as written (it times
  with ``torch.cuda.Event``); its body without the timers -- image_pad(32) -> model -> image_crop -> squeeze().float() -> mean of the
  confidence inside a 100-px margin -- is executed here with the reference's own functions and the reference module (S model, fp32, no
  autocast, refine_iter 1, seeded weights) on a 350 x 470 window of the reference's sample pair data/samples/Web/0025_{L,R}.png (a size
  that is NOT a multiple of 32, so pad and crop are both active).  Stored: the two uint8 windows, the padded left image at every 4th
  pixel, the three final maps at every 2nd pixel, the average confidence.
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
for name in ("cv2", "open3d"):                                      # import-time dependencies of the two utility modules only
    sys.modules.setdefault(name, types.ModuleType(name))

from s2m2.core.utils.image_utils import image_crop, image_pad      # noqa: E402  (reference, read-only)
import s2m2.core.model.s2m2 as ref_s2m2                             # noqa: E402

from s2m2_amd.weights import seeded_state_dict                      # noqa: E402

torch.set_num_threads(8)

PAD_CASES = [((1, 3, 50, 70), torch.float32), ((2, 3, 64, 96), torch.float32), ((1, 3, 33, 64), torch.float32), ((1, 3, 45, 100), torch.uint8),
             ((1, 3, 375, 1242), torch.uint8)]
CROP_CASES = [((1, 1, 64, 96), (50, 70)), ((2, 1, 64, 96), (64, 96)), ((1, 1, 96, 160), (75, 131)), ((1, 3, 64, 64), (33, 64))]


def pad_crop():
    out = {}
    g = torch.Generator().manual_seed(0)
    for k, (shape, dt) in enumerate(PAD_CASES):
        img = torch.randint(0, 256, shape, generator=g).to(dt)
        sub = 4 if img.numel() > 100000 else 1                       # the KITTI-sized case: output kept at every 4th pixel
        out[f"pad{k}_in"] = img.numpy()
        out[f"pad{k}_sub"] = np.array(sub)
        out[f"pad{k}_out"] = image_pad(img, 32)[..., ::sub, ::sub].numpy()
    for k, (shape, to) in enumerate(CROP_CASES):
        img = torch.rand(shape, generator=g)
        out[f"crop{k}_in"] = img.numpy()
        out[f"crop{k}_shape"] = np.array(to)
        out[f"crop{k}_out"] = image_crop(img, to).numpy()
    path = os.path.join(HERE, "utils_pad_crop.npz")
    np.savez_compressed(path, npad=np.array(len(PAD_CASES)), ncrop=np.array(len(CROP_CASES)), **out)
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB")


def real_pair():
    from PIL import Image
    y0, x0, H, W = 300, 200, 350, 470                                # window of the 800 x 1100 sample (disparities of a few tens of px)
    lr = []
    for side in "LR":
        im = np.asarray(Image.open(f"/root/reference/data/samples/Web/0025_{side}.png").convert("RGB"))
        lr.append(np.ascontiguousarray(im[y0:y0 + H, x0:x0 + W]))
    left = torch.from_numpy(lr[0]).permute(2, 0, 1)[None]            # (1,3,H,W) uint8, as the demos build it (visualize_2d_simple.py)
    right = torch.from_numpy(lr[1]).permute(2, 0, 1)[None]
    C, ntr, ri, seed = 128, 1, 1, 0
    model = ref_s2m2.S2M2(C, 1, ntr, use_positivity=True, output_upsample=False, refine_iter=ri).eval()
    model.load_state_dict(seeded_state_dict(C, 1, ntr, seed), strict=True)
    # body of run_stereo_matching (model_utils.py:69-94) without the CUDA event timers and without autocast (fp32 parity configuration)
    lp, rp = image_pad(left, 32), image_pad(right, 32)
    with torch.inference_mode():
        d, o, c = model(lp, rp)
    d = image_crop(d, (H, W)).squeeze().float()
    o = image_crop(o, (H, W)).squeeze().float()
    c = image_crop(c, (H, W)).squeeze().float()
    margin = 100
    score = c[margin:-margin, margin:-margin].mean().item()
    print(f"padded {tuple(lp.shape)}  disp [{float(d.min()):.2f}, {float(d.max()):.2f}]  avg conf {score:.6f}")
    path = os.path.join(HERE, "e2e_S_web0025_crop_fp32_r1.npz")
    np.savez_compressed(path, cfg=np.array([C, ntr, H, W, 1, 1, ri, 0, seed]), window=np.array([y0, x0, H, W]), left=lr[0], right=lr[1],
                        left_pad_sub4=lp[..., ::4, ::4].numpy(), disp=d[::2, ::2].numpy(), occ=o[::2, ::2].numpy(), conf=c[::2, ::2].numpy(),
                        avg_conf=np.array(score), torch_version=np.array(torch.__version__))
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    pad_crop()
    real_pair()
