"""GPU parity of K14 (s2m2_conv_block: a whole ConvBlock2D, reference attentions.py:255-281, in one launch) against the three launches it
replaces -- K9 two-stage chain (1x1 branch), K5 v5 (3x3 + GELU), K5 v5 with the residual epilogue -- BIT FOR BIT (same arithmetic in the same
order; the kernel only keeps the intermediates on the CU), and against a plain PyTorch fp32 restatement with fp16 rounding points."""
import math

import pytest
import torch
import torch.nn.functional as F

from s2m2_amd import pack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


def _layers(C, seed, bias=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    k0 = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / math.sqrt(9 * C)).half()
    k2 = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / math.sqrt(9 * C)).half()
    p0 = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half()
    p2 = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half()
    bs = [torch.randn(C, device="cuda", generator=g) * 0.3 if bias else None for _ in range(4)]
    return k0, k2, p0, p2, bs


def _triple(hip, x, k0, k2, p0, p2, bs):
    C = x.shape[-1]
    w0, w2 = pack.pack_conv_frag(k0, torch.float16), pack.pack_conv_frag(k2, torch.float16)
    a0, a2 = pack.chain_frag(pack.pack_conv(p0, torch.float16)), pack.chain_frag(pack.pack_conv(p2, torch.float16))
    b = hip.mlp_chain(x, [(a0, bs[2], hip.ACT_RELU, None), (a2, bs[3], hip.ACT_NONE, None)], frag=True)
    t = hip.conv2d([x], w0, bs[0], 3, 3, C, act=hip.ACT_GELU, korder=2)
    return hip.conv2d([t], w2, bs[1], 3, 3, C, epi=hip.EPI_ADD, aux0=b, korder=2), (w0, w2, a0, a2)


CASES = [  # N, H, W, C, patch_rows, bias
    (1, 128, 152, 128, 0, True),       # the 1/8 level of 1216 x 1024 (refiners)
    (2, 128, 152, 128, 0, True),       # ... of the feature pyramid (both views)
    (1, 64, 76, 256, 0, True),         # the 1/16 level: two 128-channel chunks, eight waves
    (2, 64, 76, 256, 0, False),
    (1, 64, 76, 128, 0, True),         # GlobalRefiner's 1/16 level
    (1, 64, 76, 128, 4, True),
    (1, 60, 80, 128, 2, True),         # 640 x 480
    (1, 37, 45, 128, 4, True),         # ragged: partial patches on both edges
    (1, 37, 45, 128, 2, False),
    (1, 5, 7, 256, 0, True),           # a grid smaller than one patch
    (3, 9, 33, 128, 0, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"n{c[0]}-{c[1]}x{c[2]}-C{c[3]}-p{c[4]}")
def test_conv_block_equals_the_three_launches_bit_for_bit(hip, case):
    N, H, W, C, ph, bias = case
    assert hip.conv_block_supported(C, H, W, torch.float16)
    k0, k2, p0, p2, bs = _layers(C, H + W, bias)
    x = (torch.randn(N, H, W, C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(H)) * 1.3 + 0.2).half()
    ref, (w0, w2, a0, a2) = _triple(hip, x, k0, k2, p0, p2, bs)
    got = hip.conv_block(x, w0, bs[0], w2, bs[1], a0, bs[2], a2, bs[3], patch_rows=ph)
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())


def test_conv_block_vs_torch(hip):
    N, H, W, C = 1, 30, 41, 128
    k0, k2, p0, p2, bs = _layers(C, 3)
    x = torch.randn(N, H, W, C, device="cuda").half()
    _, (w0, w2, a0, a2) = _triple(hip, x, k0, k2, p0, p2, bs)
    got = hip.conv_block(x, w0, bs[0], w2, bs[1], a0, bs[2], a2, bs[3]).float()
    xn = x.float().permute(0, 3, 1, 2)
    rd = lambda t: t.half().float()                                   # noqa: E731
    t = rd(F.gelu(F.conv2d(xn, k0.float(), bs[0], padding=1)))
    main = rd(F.conv2d(t, k2.float(), bs[1], padding=1))
    br = rd(F.conv2d(rd(F.relu(F.conv2d(xn, p0.float(), bs[2]))), p2.float(), bs[3]))
    ref = rd(main + br).permute(0, 2, 3, 1)
    e = (got - ref).abs()
    assert float((e > 4e-3 * ref.abs().clamp(min=1.0)).float().mean()) <= 5e-3 and float(e.max()) <= 3e-2, (float(e.max()),)


def test_conv_block_rejects_what_it_does_not_take(hip):
    assert not hip.conv_block_supported(192, 64, 76, torch.float16) and not hip.conv_block_supported(128, 256, 304, torch.float16)
    assert not hip.conv_block_supported(128, 64, 76, torch.float32)
