"""Golden outputs of the UNMODIFIED reference in its fp16 deployment mode, generated in the build container on CPU.

    python tests/golden/make_golden_fp16.py        # needs /root/reference; writes tests/golden/e2e_S_640x480_fp16_autocast.npz

The reference's run_stereo_matching (model_utils.py:74-82) wraps the forward in torch.amp.autocast(device_type=device.type,
dtype=torch.float16); with device cpu that is CPU autocast, the only fp16 mode of the reference that can run here.  Its op list is not
CUDA autocast's and CPU fp16 GEMMs round differently, so this pins the fp16 paths (the oracle's emulation, the HIP fp16 mode)
STATISTICALLY: they must sit as close to these outputs as the reference's own fp32 run does (stored beside them as the yardstick).
BASELINE configs[1] geometry: S model, 640x480, refine_iter 3, seeded weights / inputs as in tests/test_hip_parity_baseline.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import s2m2.core.model.s2m2 as ref_s2m2  # noqa: E402  (reference, read-only)

from s2m2_amd.weights import seeded_state_dict, synthetic_pair  # noqa: E402

torch.set_num_threads(8)


def main():
    C, ntr, H, W, ri, seed, disparity = 128, 1, 480, 640, 3, 0, 32
    sd = seeded_state_dict(C, 1, ntr, seed)
    model = ref_s2m2.S2M2(C, 1, ntr, use_positivity=True, output_upsample=False, refine_iter=ri).eval()
    model.load_state_dict(sd, strict=True)
    left, right = synthetic_pair(H, W, 1, disparity, seed)
    with torch.inference_mode():
        with torch.amp.autocast(enabled=True, device_type="cpu", dtype=torch.float16):       # model_utils.py:76 with device cpu
            d16, o16, c16 = model(left, right)
        d32, o32, c32 = model(left, right)
    out = dict(cfg=np.array([C, ntr, H, W, 1, 1, ri, disparity, seed]), disp_fp16=d16.float().numpy(), occ_fp16=o16.half().numpy(),
               conf_fp16=c16.half().numpy(), disp_fp32=d32.float().numpy(), occ_fp32=o32.half().numpy(), conf_fp32=c32.half().numpy(),
               torch_version=np.array(torch.__version__))
    path = os.path.join(HERE, "e2e_S_640x480_fp16_autocast.npz")
    np.savez_compressed(path, **out)
    dd = (d16.float() - d32.float()).abs().flatten()
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB; reference fp16 (CPU autocast) vs reference fp32: disparity median "
          f"{float(dd.median()):.4f} p90 {float(dd.kthvalue(int(0.9 * dd.numel()))[0]):.4f} p99 {float(dd.kthvalue(int(0.99 * dd.numel()))[0]):.4f} px")


if __name__ == "__main__":
    main()
