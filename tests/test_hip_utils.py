"""GPU parity of the driver-glue counterparts (s2m2_amd/utils.py) against the reference formulas restated with PyTorch ops
(src/s2m2/core/utils/image_utils.py:27-103 -- the module itself needs OpenCV, which is not installed)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_image_pad(img, factor=32):
    H, W = img.shape[-2:]
    Hn, Wn = math.ceil(H / factor) * factor, math.ceil(W / factor) * factor
    ph, pw = Hn - H, Wn - W
    x = F.pad(img, (pw // 2, pw - pw // 2, 0, 0), "constant", 0)
    x = F.pad(x, (0, 0, ph // 2, ph - ph // 2), "constant", 0)
    down = F.adaptive_avg_pool2d(x.float(), output_size=[H // factor, W // factor])
    out = F.interpolate(down, size=[Hn, Wn], mode="bilinear")
    out[:, :, ph // 2: ph // 2 + H, pw // 2: pw // 2 + W] = img
    return out


@pytest.mark.parametrize("shape", [(1, 3, 100, 150), (2, 3, 97, 64), (1, 3, 64, 96), (1, 3, 375, 1242)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8])
def test_image_pad_matches_reference_formula(shape, dtype):
    from s2m2_amd import utils
    g = torch.Generator().manual_seed(shape[2])
    img = torch.randint(0, 256, shape, generator=g).to(dtype)
    out = utils.image_pad(img.cuda(), 32).cpu()
    ref = _ref_image_pad(img.float(), 32)
    assert tuple(out.shape) == tuple(ref.shape)
    assert float((out - ref).abs().max()) < 2e-3


def test_run_stereo_matching_on_a_non_x32_pair():
    from s2m2_amd import utils
    from s2m2_amd.model import build_model
    from s2m2_amd.weights import synthetic_pair
    m = build_model("S", use_positivity=True, refine_iter=1).cuda().eval()
    l, r = synthetic_pair(288, 352, 1, 8, 2)
    l, r = l[..., :270, :330], r[..., :270, :330]                      # not multiples of 32
    d, o, c, score, ms = utils.run_stereo_matching(m, l, r, torch.device("cuda"), N_repeat=2)
    assert tuple(d.shape) == (270, 330) and tuple(o.shape) == (270, 330) and tuple(c.shape) == (270, 330)
    assert torch.isfinite(d).all() and 0.0 <= score <= 1.0 and ms > 0
    assert utils.image_crop(torch.zeros(1, 1, 288, 352), (270, 330)).shape[-2:] == (270, 330)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_stem_mlp_matches_the_two_conv_layers(dtype):
    """K8 stem (conv0 = Conv1x1(3,16)-GELU-Conv1x1(16,16), submodules.py:68-71, per pixel on the VALU) == the two K5 launches it
    replaces and the fp32 formula with the intermediate rounded to the I/O dtype."""
    import torch.nn.functional as F
    from s2m2_amd import hip, pack
    g = torch.Generator(device="cuda").manual_seed(3)
    x8 = torch.zeros(2, 37, 53, 8, device="cuda", dtype=dtype)
    x8[..., 1:4] = (torch.rand(2, 37, 53, 3, device="cuda", generator=g) * 2 - 1).to(dtype)
    w0 = torch.zeros(16, 8, 1, 1, device="cuda")
    w0[:, 1:4] = torch.randn(16, 3, 1, 1, device="cuda", generator=g)
    w1 = torch.randn(16, 16, 1, 1, device="cuda", generator=g) / 4
    b0, b1 = torch.randn(16, device="cuda", generator=g), torch.randn(16, device="cuda", generator=g)
    p0, p1 = pack.pack_conv(w0.to(dtype), dtype), pack.pack_conv(w1.to(dtype), dtype)
    y = hip.stem_mlp(x8, p0.float().contiguous(), pack.pack_bias(b0, 16), p1.float().contiguous(), pack.pack_bias(b1, 16))
    t = hip.conv2d([x8], p0, pack.pack_bias(b0, 16), 1, 1, 16, act=hip.ACT_GELU)
    y2 = hip.conv2d([t], p1, pack.pack_bias(b1, 16), 1, 1, 16)
    h = F.gelu(F.linear(x8.float(), p0.float(), b0)).to(dtype).float()
    ref = F.linear(h, p1.float(), b1)
    tol = 2e-5 if dtype == torch.float32 else 4e-3
    assert y.shape == (2, 37, 53, 16) and y.dtype == dtype
    assert float((y.float() - ref).abs().max()) < tol
    assert float((y.float() - y2.float()).abs().max()) < tol


# ---- pinned to the reference's REAL functions (tests/golden/make_golden_utils.py imports image_pad / image_crop from the reference with a
# stubbed cv2 and runs the body of run_stereo_matching on a window of the reference's own sample pair) ---------------------------------
def _gold(name):
    import os
    import numpy as np
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))


def test_image_pad_matches_the_reference_function():
    from s2m2_amd import utils
    g = _gold("utils_pad_crop.npz")
    for k in range(int(g["npad"])):
        img = torch.from_numpy(g[f"pad{k}_in"])
        sub = int(g[f"pad{k}_sub"])
        out = utils.image_pad(img.cuda(), 32).cpu()[..., ::sub, ::sub]
        ref = torch.from_numpy(g[f"pad{k}_out"])
        assert tuple(out.shape) == tuple(ref.shape), k
        # the interior is a copy (exact); the blurred border is adaptive_avg_pool2d + bilinear on values up to 255: fp32 summation order
        assert float((out - ref).abs().max()) < 2e-3, (k, float((out - ref).abs().max()))


def test_run_stereo_matching_body_on_the_references_sample_pair():
    """image_pad -> forward -> image_crop -> average confidence (model_utils.py:69-94) in fp32 on a 350 x 470 window of
    data/samples/Web/0025_{L,R}.png (uint8, not a multiple of 32) against the reference's own functions + module on the same data."""
    from s2m2_amd import utils
    from s2m2_amd.model import S2M2
    from s2m2_amd.weights import seeded_state_dict
    g = _gold("e2e_S_web0025_crop_fp32_r1.npz")
    C, ntr, H, W, B, pos, ri, _, seed = [int(v) for v in g["cfg"]]
    left = torch.from_numpy(g["left"]).permute(2, 0, 1)[None].cuda()          # uint8, as the demos hand it over
    right = torch.from_numpy(g["right"]).permute(2, 0, 1)[None].cuda()
    m = S2M2(C, 1, ntr, use_positivity=bool(pos), refine_iter=ri)
    m.load_state_dict(seeded_state_dict(C, 1, ntr, seed), strict=True)
    m = m.cuda().eval()
    lp, rp = utils.image_pad(left, 32), utils.image_pad(right, 32)
    assert float((lp.cpu()[..., ::4, ::4] - torch.from_numpy(g["left_pad_sub4"])).abs().max()) < 2e-3
    with torch.inference_mode():
        d, o, c = m(lp, rp)
    d, o, c = (utils.image_crop(t, (H, W)).squeeze().float().cpu() for t in (d, o, c))
    assert tuple(d.shape) == (H, W)
    for name, t, lim in (("disp", d, None), ("occ", o, 5e-4), ("conf", c, 5e-4)):
        ref = torch.from_numpy(g[name])
        e = (t[::2, ::2] - ref).abs()
        frac = float((e > 1e-3 + 1e-4 * ref.abs()).float().mean())
        # free running on real image content: a near-tie argmax of DispInit may fall the other way and move its neighbourhood
        assert frac <= 5e-3 and float(e.median()) < (2e-4 if name == "disp" else 2e-5), (name, frac, float(e.median()), float(e.max()))
    score = float(c[100:-100, 100:-100].mean())
    assert abs(score - float(g["avg_conf"])) < 2e-4, (score, float(g["avg_conf"]))
    # the deployed entry point (autocast fp16, timers) on the same pair: same shapes, a score close to the fp32 one
    out = utils.run_stereo_matching(m, left, right, torch.device("cuda"), N_repeat=1)
    assert tuple(out[0].shape) == (H, W) and abs(out[3] - float(g["avg_conf"])) < 0.05
