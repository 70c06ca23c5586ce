"""GPU parity of K11 (s2m2_pw_direct: a 1x1 layer with any channel counts in the direct style -- weights as MFMA fragments from global memory
straight into the operand registers) against a plain PyTorch fp32 reference of the same layer and against the K5 launch it replaces
(reference refinenet.py:87-106,138-146 -- the 1x1 layers of LocalRefiner; unet.py:32-37 -- up_conv; submodules.py:104-113,131-144 -- the
ConvTranspose heads of the upsampling masks).

Tolerance: fp16 operands, fp32 accumulation, one rounding of the result: |err| <= 2^-9 * max(1, |ref|max) against fp32 math on the fp16
operands; against K5 (same operands, same fp32 accumulation, possibly another summation order) at most a few fp16 ulps, on a handful of elements."""
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


# source channel counts, Cout, activation, leading shape, bias      (the shapes the engine launches, plus ragged / tiny ones)
CASES = [
    ((32,), 192, 1, (1, 256, 304), True),                 # corr_feat*.0 (both levels block-diagonal): 32 -> 192, GELU
    ((192,), 128, 0, (1, 256, 304), True),                # corr_feat*.2
    ((64,), 32, 0, (1, 256, 304), True),                  # conf_occ_feat.2 (reads a channel slice)
    ((96, 128, 128, 32), 256, 1, (1, 256, 304), True),    # disp_corr_ctx_cat.0: four sources -> 2C, GELU
    ((256,), 128, 0, (2, 32, 38), True),                  # up_conv 2C -> C on the coarse grid
    ((256,), 128, 0, (1, 63, 77), False),                 # ragged row count, no bias
    ((48,), 16, 0, (1, 128, 160), True),                  # conv_concat.2 of UpsampleMask1x: Cout = 16 (half a cout tile)
    ((128,), 256, 2, (1, 5, 7), True),                    # a single ragged block, ReLU
    ((8, 24), 96, 1, (2, 9, 11), True),                   # narrow sources, three cout tiles
    ((384,), 192, 0, (1, 40, 50), True),                  # K = 384, six cout tiles
]


@pytest.mark.parametrize("cs,cout,act,shp,has_bias", CASES)
def test_pw_direct_vs_torch_and_k5(hip, cs, cout, act, shp, has_bias):
    dtype = torch.float16
    K = sum(cs)
    assert hip.pw_direct_supported(K, cout, dtype)
    g = torch.Generator(device="cuda").manual_seed(K + cout)
    # sources as channel slices of wider tensors (strided rows), like dc[..., 96:160] in the engine
    srcs = []
    for c in cs:
        wide = (torch.randn(*shp, c + 16, device="cuda", generator=g) * 1.5 + 0.2).to(dtype)
        srcs.append(wide[..., 8:8 + c])
    w = (torch.randn(cout, K, 1, 1, device="cuda", generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(cout, device="cuda", generator=g) * 0.5 if has_bias else None
    splits = [(c, c) for c in cs]
    wp = pack.pack_conv(w, dtype, splits)
    bp = pack.pack_bias(b, cout) if has_bias else None
    wf = pack.pw_frag(wp)
    assert tuple(wf.shape) == ((cout + 31) // 32, (K + 15) // 16, 64, 8)
    ref = ACT[act](F.linear(torch.cat([t.float() for t in srcs], -1), w.reshape(cout, K).float(), b))
    scale = max(1.0, float(ref.abs().max()))
    for _ in range(3):                                             # repeated: a kernel with few barriers is a kernel that can race
        y = hip.pw_direct(srcs, wf, bp, cout, act=act)
        assert y.shape == (*shp, cout) and y.dtype == dtype
        assert float((y.float() - ref).abs().max()) < 2 ** -9 * scale, (float((y.float() - ref).abs().max()), scale)
    k5 = hip.conv2d([t.contiguous() for t in srcs], wp, bp, 1, 1, (cout + 7) // 8 * 8, act=act)
    d = (y.float() - k5.float()[..., :cout]).abs()
    assert float(d.max()) <= 2 ** -8 * scale and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("cin,cout,shp", [(128, 64, (1, 64, 76)), (128, 16, (2, 33, 40)), (128, 9, (1, 16, 24))])
def test_pw_direct_convT_2x2_stride2(hip, cin, cout, shp):
    """ConvTranspose2d(k 2, s 2) as the pixel-shuffle GEMM (UpsampleMask4x.conv_x / conv_concat.2, UpsampleMask1x.conv_ctx): K11 == K5's
    shuffle2 launch within fp16 rounding and == F.conv_transpose2d in fp32."""
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(cin + cout)
    x = (torch.randn(*shp, cin, device="cuda", generator=g) * 1.2).to(dtype)
    w = (torch.randn(cin, cout, 2, 2, device="cuda", generator=g) / math.sqrt(cin)).to(dtype)
    b = torch.randn(cout, device="cuda", generator=g) * 0.3
    wp, cp = pack.pack_convT_2x2s2(w, dtype)
    bp = pack.pack_bias_shuffle(b, cout)
    assert hip.pw_direct_supported(cin, 4 * cp, dtype)
    y = hip.pw_direct([x], pack.pw_frag(wp), bp, 4 * cp, shuffle2=cp)
    assert tuple(y.shape) == (shp[0], 2 * shp[1], 2 * shp[2], cp)
    ref = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=2).permute(0, 2, 3, 1)
    scale = max(1.0, float(ref.abs().max()))
    assert float((y.float()[..., :cout] - ref).abs().max()) < 2 ** -9 * scale
    assert bool((y[..., cout:] == 0).all())                        # padding channels of C' stay zero (zero weight rows, zero bias)
    k5 = hip.conv2d([x], wp, bp, 1, 1, 4 * cp, shuffle2=cp)
    assert float((y.float() - k5.float()).abs().max()) <= 2 ** -8 * scale


def test_pw_direct_rejects_what_it_does_not_take(hip):
    dtype = torch.float16
    assert not hip.pw_direct_supported(80, 128, dtype) and not hip.pw_direct_supported(128, 160, dtype)        # 5 k16 steps / 5 cout tiles
    assert not hip.pw_direct_supported(128, 128, torch.float32) and not hip.pw_direct_supported(512, 128, dtype)
    x = torch.randn(4, 128, device="cuda").half()
    wf = pack.pw_frag(torch.randn(128, 128, device="cuda").half())
    with pytest.raises(ValueError, match="pack.pw_frag"):
        hip.pw_direct([x], wf, None, 96)                            # fragment tensor of another Cout
    with pytest.raises(RuntimeError, match="act"):
        hip.pw_direct([x], wf, None, 128, act=hip.ACT_TANH)
