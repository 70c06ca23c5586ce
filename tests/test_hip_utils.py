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
