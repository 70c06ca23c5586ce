"""GPU parity of K12 (s2m2_conv_narrow: 3x3 / 5x5-stride-2 layers on 8- / 16-channel tensors in the direct style) against a plain PyTorch
fp32 reference of the same layer and against the K5 launch it replaces (reference submodules.py:124-129,139-141 -- UpsampleMask1x
conv_disp.0 | conv_rgb.0; refinenet.py:93-101,141-142 -- LocalRefiner disp_feat.0 | conf_occ_feat.0; submodules.py:69-71 -- conv1_down.0).

Tolerance: fp16 operands, fp32 accumulation, one rounding of the result: |err| <= 2^-9 * max(1, |ref|max) against fp32 math on the fp16
operands; against K5 (same operands, fp32 accumulation in another order) a few fp16 ulps on a handful of elements."""
import math

import pytest
import torch
import torch.nn.functional as F

from s2m2_amd import pack

pytestmark = pytest.mark.gpu
ACT = {0: lambda t: t, 1: F.gelu, 2: F.relu}


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


# KH, stride, Cin, Cout, act, (N, H, W), bias, forced rows per block (S2M2_NARROW_MT is read once per process: both tile heights are
# reached through the grid size instead -- 8-row blocks from 1024 blocks on)
CASES = [
    (3, 1, 8, 32, 2, (1, 64, 96), True),                 # mask1x conv_disp | conv_rgb: 8 -> 32, ReLU (4-row blocks)
    (3, 1, 8, 32, 2, (1, 512, 544), True),               # ... on a grid large enough for the 8-row blocks
    (3, 1, 8, 160, 1, (1, 64, 76), True),                # disp_feat.0 | conf_occ_feat.0: 8 -> 160, GELU, five cout groups
    (3, 1, 8, 160, 1, (2, 37, 45), True),                # ragged patch rows / columns, two images
    (3, 1, 8, 24, 0, (1, 9, 11), False),                 # a partial cout tile, no bias, one ragged block
    (5, 2, 16, 64, 1, (2, 64, 96), True),                # conv1_down.0: 16 -> 64, 5x5 stride 2, GELU
    (5, 2, 16, 64, 1, (1, 1024, 1216), True),            # ... at the size the engine launches it (8-row blocks)
    (5, 2, 16, 64, 0, (1, 37, 51), True),                # odd sizes: Ho = ceil(H / 2)
    (5, 2, 16, 128, 2, (1, 20, 70), False),              # four cout tiles = two groups
]


@pytest.mark.parametrize("k,stride,cin,cout,act,shp,has_bias", CASES)
def test_conv_narrow_vs_torch_and_k5(hip, k, stride, cin, cout, act, shp, has_bias):
    dtype = torch.float16
    assert hip.conv_narrow_supported(k, k, stride, cin, cout, dtype)
    g = torch.Generator(device="cuda").manual_seed(k * 100 + cout + shp[1])
    wide = (torch.randn(*shp, cin + 8, device="cuda", generator=g) * 1.5 + 0.2).to(dtype)
    x = wide[..., :cin] if cin == 8 else wide[..., 8:8 + cin]          # a channel slice of a wider tensor: pixel stride != Cin
    w = (torch.randn(cout, cin, k, k, device="cuda", generator=g) / math.sqrt(cin * k * k)).to(dtype)
    b = torch.randn(cout, device="cuda", generator=g) * 0.5 if has_bias else None
    wp = pack.pack_conv(w, dtype, [(cin, cin)])
    bp = pack.pack_bias(b, cout) if has_bias else None
    wf = pack.narrow_frag(wp, k * k)
    ref = ACT[act](F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=stride, padding=k // 2)).permute(0, 2, 3, 1)
    if stride == 2:                                                    # Ho = ceil(H / 2) (torch gives floor((H - 1) / 2) + 1: the same)
        assert ref.shape[1] == (shp[1] + 1) // 2 and ref.shape[2] == (shp[2] + 1) // 2
    scale = max(1.0, float(ref.abs().max()))
    for _ in range(3):                                                 # repeated: a kernel without block barriers in its epilogue is a kernel that can race
        y = hip.conv_narrow(x, wf, bp, k, k, cout, stride=stride, act=act)
        assert tuple(y.shape) == tuple(ref.shape) and y.dtype == dtype
        assert float((y.float() - ref).abs().max()) < 2 ** -9 * scale, (float((y.float() - ref).abs().max()), scale)
    cp = (cout + 7) // 8 * 8
    k5 = hip.conv2d([x.contiguous()], wp, bp, k, k, cp, act=act, stride=stride)
    d = (y.float() - k5.float()[..., :cout]).abs()
    assert float(d.max()) <= 2 ** -8 * scale and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))


# source channels, Cout, act, (N, H, W), bias        (second form: 3x3 layers with few output channels on wider inputs)
PX_CASES = [
    ((32, 16), 48, 2, (1, 64, 96), True),                 # mask1x conv_concat.0: cat(32, 16) -> 48, ReLU (4-row blocks)
    ((32, 16), 48, 2, (1, 520, 544), True),               # ... on a grid large enough for the 8-row blocks
    ((96,), 96, 0, (1, 64, 76), True),                    # disp_feat.2: 96 -> 96 (reads a channel slice of a wider tensor)
    ((96,), 96, 1, (2, 37, 45), False),                   # ragged, two images, GELU, no bias
    ((256,), 16, 0, (1, 64, 76), True),                   # disp_update.2 | conf_occ_update.2: four chunks of 64 channels, 16 couts
    ((128,), 8, 0, (1, 33, 50), True),                    # GlobalRefiner out_feat: 128 -> 8
    ((128, 128), 24, 2, (1, 20, 70), True),               # two sources meeting at a chunk boundary
    ((16, 32), 64, 1, (2, 9, 11), True),                  # two full cout tiles, one ragged block
]


@pytest.mark.parametrize("cs,cout,act,shp,has_bias", PX_CASES)
def test_conv_px_vs_torch_and_k5(hip, cs, cout, act, shp, has_bias):
    dtype = torch.float16
    cin = sum(cs)
    assert hip.conv_narrow_supported(3, 3, 1, cin, cout, dtype)
    g = torch.Generator(device="cuda").manual_seed(cin + cout + shp[1])
    srcs = []
    for c in cs:
        wide = (torch.randn(*shp, c + 16, device="cuda", generator=g) * 1.2 + 0.1).to(dtype)
        srcs.append(wide[..., 8:8 + c])                               # channel slices: pixel stride != channel count
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / math.sqrt(cin * 9)).to(dtype)
    b = torch.randn(cout, device="cuda", generator=g) * 0.5 if has_bias else None
    wp = pack.pack_conv(w, dtype, [(c, c) for c in cs])
    bp = pack.pack_bias(b, cout) if has_bias else None
    wf = pack.narrow_frag(wp, 9)
    ref = ACT[act](F.conv2d(torch.cat([t.float() for t in srcs], -1).permute(0, 3, 1, 2), w.float(), b, padding=1)).permute(0, 2, 3, 1)
    scale = max(1.0, float(ref.abs().max()))
    for _ in range(3):
        y = hip.conv_narrow(srcs, wf, bp, 3, 3, cout, act=act)
        assert tuple(y.shape) == tuple(ref.shape) and y.dtype == dtype
        assert float((y.float() - ref).abs().max()) < 2 ** -9 * scale * max(1.0, math.sqrt(cin / 128)), (float((y.float() - ref).abs().max()), scale)
    k5 = hip.conv2d([t.contiguous() for t in srcs], wp, bp, 3, 3, (cout + 7) // 8 * 8, act=act)
    d = (y.float() - k5.float()[..., :cout]).abs()
    assert float(d.max()) <= 2 ** -8 * scale and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("shp", [(1, 64, 96), (2, 37, 45), (1, 520, 544)])
def test_conv_px_fused_head_vs_two_launches_and_torch(hip, shp):
    """UpsampleMask1x conv_concat.0 -> ReLU -> conv_concat.2 (1x1, 48 -> 9 padded to 16) as ONE K12 launch (head=...): equal to the K12 launch
    followed by the K11 launch up to the fp32 summation order inside a k16 step (single fp16 ulps), and to fp32 math on the fp16 operands with the
    intermediate rounded to fp16 where the two-launch path stores it."""
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(7 + shp[1])
    a = (torch.randn(*shp, 32, device="cuda", generator=g) * 1.1).to(dtype)
    b = (torch.randn(*shp, 16, device="cuda", generator=g) * 0.9 + 0.2).to(dtype)
    w0 = (torch.randn(48, 48, 3, 3, device="cuda", generator=g) / math.sqrt(48 * 9)).to(dtype)
    b0 = torch.randn(48, device="cuda", generator=g) * 0.3
    w2 = (torch.randn(9, 48, 1, 1, device="cuda", generator=g) / math.sqrt(48)).to(dtype)
    b2 = torch.randn(9, device="cuda", generator=g) * 0.2
    wp0, bp0 = pack.pack_conv(w0, dtype, [(32, 32), (16, 16)]), pack.pack_bias(b0, 48)
    wp2, bp2 = pack.pack_conv(w2, dtype), pack.pack_bias(b2, 9)              # (16, 48): Cout 9 padded to 16
    assert tuple(wp2.shape) == (16, 48)
    wf0 = pack.narrow_frag(wp0, 9)
    y = hip.conv_narrow([a, b], wf0, bp0, 3, 3, 48, act=hip.ACT_RELU)
    two = hip.pw_direct([y], pack.pw_frag(wp2), bp2, 16)
    for _ in range(3):
        one = hip.conv_narrow([a, b], wf0, bp0, 3, 3, 48, act=hip.ACT_RELU, head=(pack.head_frag(wp2), bp2, 16))
        assert tuple(one.shape) == (*shp, 16) and one.dtype == dtype
        d = (one.float() - two.float()).abs()
        scale = max(1.0, float(two.float().abs().max()))
        assert float(d.max()) <= 2 ** -8 * scale and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))
    x = torch.cat([a, b], -1).float().permute(0, 3, 1, 2)
    mid = F.relu(F.conv2d(x, w0.float(), b0, padding=1)).half().float()
    ref = F.conv2d(mid, w2.float(), b2).permute(0, 2, 3, 1)
    assert float((one.float()[..., :9] - ref).abs().max()) < 2 ** -8 * scale
    assert bool((one[..., 9:] == 0).all())                                    # padding channels of the head: zero weight rows, zero bias


def test_conv_narrow_rejects_what_it_does_not_take(hip):
    dtype = torch.float16
    assert not hip.conv_narrow_supported(3, 3, 1, 16, 32, dtype) and not hip.conv_narrow_supported(3, 3, 2, 8, 32, dtype)
    assert not hip.conv_narrow_supported(3, 3, 1, 128, 64, dtype) and not hip.conv_narrow_supported(3, 3, 1, 64, 32, dtype)
    assert not hip.conv_narrow_supported(5, 5, 2, 16, 32, dtype) and not hip.conv_narrow_supported(3, 3, 1, 8, 32, torch.float32)
    x = torch.randn(1, 8, 8, 8, device="cuda").half()
    wf = pack.pw_frag(torch.randn(32, 72, device="cuda").half())
    with pytest.raises(ValueError, match="pack.narrow_frag"):
        hip.conv_narrow(x, wf, None, 3, 3, 64)                          # fragment tensor of another Cout
    with pytest.raises(RuntimeError, match="act"):
        hip.conv_narrow(x, wf, None, 3, 3, 32, act=hip.ACT_TANH)
    with pytest.raises(RuntimeError, match="not supported"):
        hip.conv_narrow(torch.randn(1, 8, 8, 16, device="cuda").half(), pack.pw_frag(torch.randn(32, 144, device="cuda").half()), None, 3, 3, 32)
