"""CPU checks of the driver-glue counterparts that are pure tensor logic: image_crop against outputs of the reference's REAL function
(tests/golden/utils_pad_crop.npz, written by tests/golden/make_golden_utils.py with the reference imported under a stubbed cv2)."""
import os

import numpy as np
import torch


def test_image_crop_matches_the_reference_function():
    from s2m2_amd import utils
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "utils_pad_crop.npz"))
    for k in range(int(g["ncrop"])):
        img = torch.from_numpy(g[f"crop{k}_in"])
        out = utils.image_crop(img, tuple(int(v) for v in g[f"crop{k}_shape"]))
        assert torch.equal(out, torch.from_numpy(g[f"crop{k}_out"])), k
