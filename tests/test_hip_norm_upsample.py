"""GPU parity of K6 (LayerNorm rows, GroupNorm NHWC) and K7 (convex upsampling) through the C ABI against plain PyTorch fp32
references (F.layer_norm, F.group_norm, and the reference's unfold/softmax/sum formulation, s2m2.py:101-133)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C", [128, 256, 192, 384, 768, 8])
def test_layernorm_rows(hip, dtype, C):
    g = torch.Generator(device="cuda").manual_seed(C)
    x = (torch.randn(3, 37, C, device="cuda", generator=g) * 3 + 1.5).to(dtype)
    y = hip.layernorm(x)
    ref = F.layer_norm(x.float(), (C,))
    tol = 2e-5 if dtype == torch.float32 else 2e-3
    assert float((y.float() - ref).abs().max()) < tol
    # strided rows: a channel slice of a wider tensor
    wide = (torch.randn(50, 2 * C, device="cuda", generator=g)).to(dtype)
    y2 = hip.layernorm(wide[:, C:])
    assert float((y2.float() - F.layer_norm(wide[:, C:].float(), (C,))).abs().max()) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 24, 40, 128, 8), (1, 33, 17, 128, 8), (2, 16, 16, 256, 8), (1, 20, 28, 192, 8), (1, 12, 20, 384, 8)])
def test_groupnorm_nhwc(hip, dtype, shape):
    N, H, W, C, G = shape
    g = torch.Generator(device="cuda").manual_seed(H)
    x = (torch.randn(N, H, W, C, device="cuda", generator=g) * 2 + 0.7).to(dtype)
    gamma = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    beta = 0.1 * torch.randn(C, device="cuda", generator=g)
    y = hip.groupnorm_nhwc(x, G, gamma, beta)
    ref = F.group_norm(x.float().permute(0, 3, 1, 2), G, gamma, beta).permute(0, 2, 3, 1)
    assert float((y.float() - ref).abs().max()) < (3e-5 if dtype == torch.float32 else 4e-3)


def _neigh9(x):
    B, _, h, w = x.shape
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
    return torch.cat([xp[:, :, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("factor", [4, 1])
def test_convex_upsample(hip, dtype, factor):
    B, h, w = 2, 13, 19
    g = torch.Generator(device="cuda").manual_seed(factor)
    maps = [torch.randn(B, 1, h, w, device="cuda", generator=g) * s for s in (30.0, 1.0, 1.0)]
    logits = torch.zeros(B, h * factor, w * factor, 16, device="cuda", dtype=dtype)
    logits[..., :9] = (torch.randn(B, h * factor, w * factor, 9, device="cuda", generator=g) * 3).to(dtype)
    logits[..., 9:] = 50.0                                    # padding channels must be ignored
    outs = hip.convex_upsample(maps, logits, factor, scales=[4.0, 1.0, 1.0])
    wgt = logits[..., :9].float().permute(0, 3, 1, 2).softmax(1)
    for m, o, s in zip(maps, outs, (4.0, 1.0, 1.0)):
        n = _neigh9(m)
        if factor > 1:
            n = F.interpolate(n, scale_factor=factor, mode="nearest")
        ref = (n * wgt).sum(1, keepdim=True) * s
        assert tuple(o.shape) == tuple(ref.shape)
        assert float((o - ref).abs().max()) < 2e-5 * float(ref.abs().max() + 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_convex_upsample_output_upsample_branch(hip, dtype):
    """output_upsample=True branch of upsample1x (s2m2.py:123-127): nearest x2 of the neighbours, bilinear x2 of the logits."""
    B, h, w = 1, 14, 22
    g = torch.Generator(device="cuda").manual_seed(7)
    m = torch.randn(B, 1, h, w, device="cuda", generator=g) * 20
    logits = torch.zeros(B, h, w, 16, device="cuda", dtype=dtype)
    logits[..., :9] = (torch.randn(B, h, w, 9, device="cuda", generator=g) * 2).to(dtype)
    (o,) = hip.convex_upsample([m], logits, 2, logit_up2=True)
    lg = F.interpolate(logits[..., :9].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    ref = (F.interpolate(_neigh9(m), scale_factor=2, mode="nearest") * lg.float().softmax(1)).sum(1, keepdim=True)
    assert float((o - ref).abs().max()) < (5e-5 if dtype == torch.float32 else 5e-2) * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("mode", [0, 1])
def test_resample2x(hip, dtype, mode):
    g = torch.Generator(device="cuda").manual_seed(mode)
    x = torch.randn(2, 14, 22, 128, device="cuda", generator=g).to(dtype)
    y = hip.resample2x(x, mode)
    xn = x.float().permute(0, 3, 1, 2)
    ref = F.avg_pool2d(xn, 2) if mode == 0 else F.interpolate(xn, scale_factor=2, mode="bilinear", align_corners=False)
    assert float((y.float() - ref.permute(0, 2, 3, 1)).abs().max()) < (1e-6 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("img_dtype", [torch.float32, torch.uint8])
def test_image_prep(hip, dtype, img_dtype):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(0, 256, (2, 3, 18, 26), device="cuda", generator=g).to(img_dtype)
    b = torch.randint(0, 256, (2, 3, 18, 26), device="cuda", generator=g).to(img_dtype)
    x8 = hip.image_prep(a, b, dtype)
    ref = ((torch.cat([a, b]).float() / 255.0 - 0.5) * 2).permute(0, 2, 3, 1)
    assert tuple(x8.shape) == (4, 18, 26, 8)
    assert float((x8[..., 1:4].float() - ref).abs().max()) < (1e-6 if dtype == torch.float32 else 1e-3)
    assert float(x8[..., 0].abs().max()) == 0 and float(x8[..., 4:].abs().max()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_refiner_pointwise_stages(hip, dtype):
    g = torch.Generator(device="cuda").manual_seed(1)
    B, h, w = 2, 11, 17
    disp = torch.rand(B, 1, h, w, device="cuda", generator=g) * 40 - 2
    conf = torch.rand(B, 1, h, w, device="cuda", generator=g)
    occ = torch.rand(B, 1, h, w, device="cuda", generator=g)
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    s0 = hip.refine_prep(disp, conf, None, 0, dtype).float()
    mask = (conf > 0.2).float()
    assert float((s0[..., 0] - (disp / 1e2 * mask)[:, 0]).abs().max()) < tol
    assert float((s0[..., 1] - torch.logit(mask * conf, eps=1e-1)[:, 0]).abs().max()) < tol * 4
    assert float(s0[..., 2:].abs().max()) == 0
    s1 = hip.refine_prep(disp, conf, occ, 1, dtype).float()
    assert float((s1[..., 1] - torch.logit(conf, eps=1e-2)[:, 0]).abs().max()) < tol * 8
    assert float((s1[..., 2] - torch.logit(occ, eps=1e-2)[:, 0]).abs().max()) < tol * 8
    upd = torch.randn(B, h, w, 8, device="cuda", generator=g).to(dtype)
    out = hip.global_update(upd, disp, conf, True)
    ref = (mask * disp + (1 - mask) * upd[..., 0].float().unsqueeze(1) * 1e2).clamp(min=0)
    assert float((out - ref).abs().max()) < 1e-4
    dco = torch.randn(B, h, w, 16, device="cuda", generator=g).to(dtype)
    d2, c2, o2 = hip.refine_update(dco, disp, conf, occ, True)
    rd = (disp + dco[..., 0].float().unsqueeze(1)).clamp(min=0)
    rc = torch.sigmoid(dco[..., 8].float().unsqueeze(1) + torch.logit(conf, eps=1e-2))
    ro = torch.sigmoid(dco[..., 9].float().unsqueeze(1) + torch.logit(occ, eps=1e-2))
    ro = ro * (torch.arange(w, device="cuda").float().reshape(1, 1, 1, w) - rd >= 0)
    assert float((d2 - rd).abs().max()) < 1e-5 and float((c2 - rc).abs().max()) < 1e-5 and float((o2 - ro).abs().max()) < 1e-5
    # the same launch can also write the next iteration's side input: bit-identical to refine_prep of its own outputs, and the
    # in-place C entry (s2m2_refine_update) is the same kernel with the outputs aliased to the inputs
    d3, c3, o3, small = hip.refine_update(dco, disp, conf, occ, True, want_small=True)
    assert torch.equal(d3, d2) and torch.equal(c3, c2) and torch.equal(o3, o2)
    assert torch.equal(small, hip.refine_prep(d2, c2, o2, 1, dtype))
    di, ci, oi = disp.clone(), conf.clone(), occ.clone()
    rc_ = hip.load().s2m2_refine_update(dco.data_ptr(), 16, di.data_ptr(), ci.data_ptr(), oi.data_ptr(), di.numel(), w, 1,
                                        0 if dtype == torch.float32 else 1, None)
    torch.cuda.synchronize()
    assert rc_ == 0 and torch.equal(di, d2) and torch.equal(ci, c2) and torch.equal(oi, o2)
    x = torch.randn(3, 5, 7, 128, device="cuda", generator=g).to(dtype)
    assert float((hip.tanh(x).float() - torch.tanh(x.float())).abs().max()) < (1e-6 if dtype == torch.float32 else 1e-3)
