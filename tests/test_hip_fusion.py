"""GPU parity of K10 (s2m2_feature_fusion: FeatureFusion with 1x1 kernels in one launch, feature_fusion.py:4-33) against the fp32
formula (intermediates rounded to the I/O dtype where the kernel rounds them) and against the K5 launches it replaces.

Tolerances: fp32 -- exact-fp32 MFMA chains vs torch's summation order through two layers: 2e-4 absolute on O(1-3) values.
fp16 -- h, the gate, the mix and the fusion term are rounded to fp16 and h feeds the second GEMM: 1.5e-2 absolute."""
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


def _weights(C, dtype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    w1 = (torch.randn(3 * C, 2 * C, 1, 1, device="cuda", generator=g) / math.sqrt(2 * C)).to(dtype)
    wg = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
    wf = (torch.randn(C, 2 * C, 1, 1, device="cuda", generator=g) / math.sqrt(2 * C)).to(dtype)
    b1, bg, bf = (torch.randn(n, device="cuda", generator=g) * 0.5 for n in (3 * C, C, C))
    return w1, wg, wf, b1, bg, bf


def _fusion_ref(z0, z1, w1, wg, wf, b1, bg, bf, dtype):
    """feature_fusion.py:24-31 in fp32, every intermediate rounded to the I/O dtype where the kernel rounds it"""
    C = z0.shape[-1]

    def rd(t):
        return t.to(dtype).float()
    h = rd(F.gelu(F.linear(torch.cat([z0.float(), z1.float()], -1), w1.float().reshape(3 * C, 2 * C), b1)))
    gate = rd(torch.sigmoid(F.linear(h[..., :C], wg.float().reshape(C, C), bg))).clamp(0.01, 0.99)
    return rd(rd(F.linear(h[..., C:], wf.float().reshape(C, 2 * C), bf)) + rd(gate * z0.float() + (1 - gate) * z1.float()))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(128, (2, 50, 61)), (128, (1, 256, 40)), (128, (1, 1, 1)), (256, (2, 32, 38)), (256, (1, 9, 7)), (256, (2, 64, 76))])
def test_feature_fusion_vs_torch_and_k5(hip, dtype, shape):
    C, shp = shape
    assert hip.feature_fusion_supported(C, dtype)
    g = torch.Generator(device="cuda").manual_seed(C + shp[1])
    wide = torch.randn(*shp, 2 * C + 8, device="cuda", generator=g).to(dtype)
    z0 = wide[..., :C]                                            # strided rows (a channel slice of a wider tensor)
    z1 = (torch.randn(*shp, C, device="cuda", generator=g) * 1.5).to(dtype)
    w1, wg, wf, b1, bg, bf = _weights(C, dtype, 11 * C)
    p1 = pack.pack_conv(w1, dtype)
    p2 = torch.cat([pack.pack_conv(wg, dtype), pack.pack_conv(wf, dtype)], dim=1).contiguous()
    y = hip.feature_fusion(z0, z1, p1, pack.pack_bias(b1, 3 * C), p2, pack.pack_bias(bg, C), pack.pack_bias(bf, C))
    assert y.shape == z1.shape and y.dtype == dtype

    def rd(t):
        return t.to(dtype).float()
    h = rd(F.gelu(F.linear(torch.cat([z0.float(), z1.float()], -1), w1.float().reshape(3 * C, 2 * C), b1)))
    gate = rd(torch.sigmoid(F.linear(h[..., :C], wg.float().reshape(C, C), bg))).clamp(0.01, 0.99)
    ref = rd(rd(F.linear(h[..., C:], wf.float().reshape(C, 2 * C), bf)) + rd(gate * z0.float() + (1 - gate) * z1.float()))
    tol = 2e-4 if dtype == torch.float32 else 1.5e-2
    assert float((y.float() - ref).abs().max()) < tol
    # the launches it replaces: K5 (concat + GELU), K5 dual-GEMM gate mix
    z0c = z0.contiguous()
    gf = hip.conv2d([z0c.reshape(1, 1, -1, C), z1.reshape(1, 1, -1, C)], p1, pack.pack_bias(b1, 3 * C), 1, 1, 3 * C, act=hip.ACT_GELU)
    y2 = hip.conv2d([gf], p2, pack.pack_bias(bg, C), 1, 1, C, act=hip.ACT_SIGMOID, epi=hip.EPI_DUALMIX, aux0=z0c.reshape(1, 1, -1, C),
                    aux1=z1.reshape(1, 1, -1, C), ksplit=C, bias2=pack.pack_bias(bf, C))
    assert float((y.float().reshape(-1) - y2.float().reshape(-1)).abs().max()) < tol


def test_feature_fusion_rejects_unsupported_width(hip):
    assert not hip.feature_fusion_supported(192, torch.float16)
    z = torch.zeros(4, 192, device="cuda").half()
    with pytest.raises(RuntimeError, match="not supported"):
        hip.feature_fusion(z, z, torch.zeros(576, 384, device="cuda").half(), torch.zeros(576, device="cuda"),
                           torch.zeros(192, 576, device="cuda").half(), torch.zeros(192, device="cuda"), torch.zeros(192, device="cuda"))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(128, (2, 13, 21)), (256, (1, 16, 19)), (128, (1, 1, 1))])
def test_feature_fusion_reads_z1_through_the_bilinear_upsampling(hip, dtype, shape):
    """z1_coarse: K10 fed with the coarse tensor == K7 resample2x (nn.Upsample bilinear x2, align_corners=False) followed by K10,
    bit for bit (same arithmetic, same rounding point), and the torch interpolation within rounding."""
    C, (n, hc, wc) = shape
    g = torch.Generator(device="cuda").manual_seed(C + hc)
    z0 = torch.randn(n, 2 * hc, 2 * wc, C, device="cuda", generator=g).to(dtype)
    zc = torch.randn(n, hc, wc, C, device="cuda", generator=g).to(dtype)
    w1, wg, wf, b1, bg, bf = _weights(C, dtype, 5 * C)
    p1 = pack.pack_conv(w1, dtype)
    p2 = torch.cat([pack.pack_conv(wg, dtype), pack.pack_conv(wf, dtype)], dim=1).contiguous()
    args = (p1, pack.pack_bias(b1, 3 * C), p2, pack.pack_bias(bg, C), pack.pack_bias(bf, C))
    y = hip.feature_fusion(z0, zc, *args, z1_coarse=True)
    up = hip.resample2x(zc, 1)
    y2 = hip.feature_fusion(z0, up, *args)
    assert torch.equal(y, y2)
    ref_up = F.interpolate(zc.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    assert float((up.float() - ref_up).abs().max()) < (1e-5 if dtype == torch.float32 else 4e-3)


# ---- direct form (s2m2_feature_fusion_frag: both layers' weights as one fragment stream, straight into the MFMA operand registers) -------
@pytest.mark.parametrize("C,shp,coarse", [(256, (1, 64, 76), False), (256, (2, 32, 38), True), (256, (2, 64, 76), True),     # 9728 rows: 64-row tiles
                                          (128, (1, 128, 152), True), (128, (2, 128, 152), False),                          # 38912 rows: 64-row tiles
                                          (128, (2, 13, 21), False), (128, (1, 1, 1), False), (256, (1, 9, 7), True)])
def test_feature_fusion_direct_form_equals_lds_form_bit_for_bit(hip, C, shp, coarse):
    """Same fp16 operands, same k16 order of every MFMA chain, same epilogues and rounding points -> bit-identical outputs, incl. ragged
    last tiles, strided z0 rows and the bilinear z1 path; repeated launches (the form has 6 block barriers: a race would be a flaky mismatch)."""
    dtype = torch.float16
    assert hip.feature_fusion_frag_supported(C, dtype) and not hip.feature_fusion_frag_supported(C, torch.float32)
    g = torch.Generator(device="cuda").manual_seed(C + shp[1] + 3)
    n, h, w = shp
    if coarse:
        wide = torch.randn(n, 2 * h, 2 * w, C + 8, device="cuda", generator=g).to(dtype)
        z1 = (torch.randn(n, h, w, C, device="cuda", generator=g) * 1.5).to(dtype)
    else:
        wide = torch.randn(n, h, w, C + 8, device="cuda", generator=g).to(dtype)
        z1 = (torch.randn(n, h, w, C, device="cuda", generator=g) * 1.5).to(dtype)
    z0 = wide[..., :C]                                            # strided rows
    w1, wg, wf, b1, bg, bf = _weights(C, dtype, 7 * C)
    p1 = pack.pack_conv(w1, dtype)
    p2 = torch.cat([pack.pack_conv(wg, dtype), pack.pack_conv(wf, dtype)], dim=1).contiguous()
    bs = (pack.pack_bias(b1, 3 * C), pack.pack_bias(bg, C), pack.pack_bias(bf, C))
    ref = hip.feature_fusion(z0, z1, p1, bs[0], p2, bs[1], bs[2], z1_coarse=coarse)
    stream = pack.fusion_frag(p1, p2)
    assert stream.numel() == 9 * C * C
    for _ in range(5):
        y = hip.feature_fusion(z0, z1, stream, bs[0], None, bs[1], bs[2], z1_coarse=coarse, frag=True)
        assert y.shape == ref.shape and torch.equal(y, ref)


def test_feature_fusion_direct_form_rejects_unsupported(hip):
    z = torch.zeros(4, 640, device="cuda").half()
    with pytest.raises(RuntimeError, match="not supported"):
        hip.feature_fusion(z, z, torch.zeros(9 * 640 * 640, device="cuda").half(), torch.zeros(3 * 640, device="cuda"), None,
                           torch.zeros(640, device="cuda"), torch.zeros(640, device="cuda"), frag=True)


@pytest.mark.parametrize("C,shp,coarse", [(192, (2, 50, 61), False), (192, (1, 128, 152), True), (192, (2, 128, 152), False), (192, (1, 1, 1), False),
                                          (384, (2, 32, 38), True), (384, (1, 64, 76), False), (384, (1, 9, 7), True),
                                          (512, (2, 32, 38), True), (512, (2, 64, 76), False), (512, (1, 9, 7), True)])
def test_feature_fusion_direct_form_m_xl_widths(hip, C, shp, coarse):
    """C = 192 / 384 (the M and XL models) and C = 512 (the L model's 1/16 level, r06) exist in the direct form only: against the fp32 reference of the block (rounded where the kernel
    rounds) and against the K5 launches it replaces; repeated launches must be bit-identical to each other."""
    dtype = torch.float16
    assert hip.feature_fusion_frag_supported(C, dtype) and not hip.feature_fusion_supported(C, dtype)
    g = torch.Generator(device="cuda").manual_seed(C + shp[1] + 5)
    n, h, w = shp
    if coarse:
        wide = torch.randn(n, 2 * h, 2 * w, C + 8, device="cuda", generator=g).to(dtype)
        zc = (torch.randn(n, h, w, C, device="cuda", generator=g) * 1.5).to(dtype)
        z1_full = hip.resample2x(zc, 1)
    else:
        wide = torch.randn(n, h, w, C + 8, device="cuda", generator=g).to(dtype)
        zc = z1_full = (torch.randn(n, h, w, C, device="cuda", generator=g) * 1.5).to(dtype)
    z0 = wide[..., :C]
    w1, wg, wf, b1, bg, bf = _weights(C, dtype, 7 * C)
    p1 = pack.pack_conv(w1, dtype)
    p2 = torch.cat([pack.pack_conv(wg, dtype), pack.pack_conv(wf, dtype)], dim=1).contiguous()
    bs = (pack.pack_bias(b1, 3 * C), pack.pack_bias(bg, C), pack.pack_bias(bf, C))
    stream = pack.fusion_frag(p1, p2)
    first = None
    for _ in range(4):
        y = hip.feature_fusion(z0, zc, stream, bs[0], None, bs[1], bs[2], z1_coarse=coarse, frag=True)
        first = y if first is None else first
        assert torch.equal(y, first)
    ref = _fusion_ref(z0, z1_full, w1, wg, wf, b1, bg, bf, dtype)
    assert y.shape == ref.shape and float((y.float() - ref).abs().max()) < 2e-2
