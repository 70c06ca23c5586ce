"""GPU parity of K5 (implicit-GEMM conv / linear with fused concat + epilogues, s2m2_conv2d through the C ABI) against plain
PyTorch fp32 references of the same ops (F.conv2d / F.conv_transpose2d / F.linear + the elementwise tail), on the operand
values the kernel sees (fp16 mode: operands rounded to fp16 first, fp32 accumulate).

Tolerances: fp32 mode -- exact-fp32 MFMA chain vs the vendor's summation order: 2e-5 * sqrt(K) * scale absolute.
fp16 mode -- output rounded to fp16 (rel 2^-11) plus accumulate-order noise: 2e-3 * scale."""
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


def _act(x, a):
    return [lambda t: t, lambda t: F.gelu(t), F.relu, torch.sigmoid, torch.tanh][a](x)


def _ref_conv(srcs, w, b, act, scale=1.0):
    x = torch.cat([s.float() for s in srcs], -1).permute(0, 3, 1, 2)
    kh, kw = w.shape[2:]
    y = F.conv2d(x, w.float(), None if b is None else b.float(), padding=(kh // 2, kw // 2))
    return (_act(y, act) * scale).permute(0, 2, 3, 1)


def _tol(dtype, K, scale):
    return (2e-5 * math.sqrt(K) if dtype == torch.float32 else 2.5e-3) * scale


CASES = [
    # N, H, W, [src channels (real)], Cout, KH, KW, act, tile
    (1, 17, 23, [128], 128, 3, 3, 1, 0),
    (2, 16, 24, [128], 128, 1, 1, 0, 1),
    (1, 33, 19, [128, 128], 128, 3, 1, 3, 2),
    (1, 19, 33, [128, 128], 128, 1, 3, 4, 0),
    (1, 24, 40, [96, 64, 64, 160], 256, 1, 1, 1, 0),
    (1, 20, 28, [1], 96, 3, 3, 1, 0),
    (1, 20, 28, [9], 96, 1, 1, 1, 4),
    (1, 20, 28, [2, 128], 128, 3, 3, 1, 0),
    (1, 40, 56, [128], 1, 3, 3, 0, 0),
    (1, 40, 56, [48], 9, 1, 1, 0, 3),
    (1, 12, 20, [256], 256, 3, 3, 2, 0),
    (1, 9, 13, [512], 512, 1, 1, 1, 0),
    (3, 8, 8, [64], 32, 1, 1, 0, 0),
]
# every block-tile variant (v1 with 64-byte K rows: 5, 6; v2 LDS-direct ring: 7..11) on shapes with M / K / Cout tails,
# multi-source concatenation and zero padding
for _t in (5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27):
    CASES += [(1, 17, 23, [128], 128, 3, 3, 1, _t), (2, 13, 21, [96, 64, 64, 160], 256, 1, 1, 1, _t), (1, 20, 28, [2, 128], 136, 3, 3, 3, _t),
              (1, 33, 19, [128, 128], 128, 3, 1, 4, _t), (1, 9, 11, [8], 96, 3, 3, 1, _t), (2, 19, 70, [128, 128], 128, 1, 3, 0, _t)]

# v4 persistent pointwise kernel (1x1 only): pixel-count tails, Cout tails, concatenated sources, three K chunks
for _t in (14, 15):
    CASES += [(2, 37, 53, [128], 128, 1, 1, 1, _t), (1, 40, 56, [96, 64, 64, 160], 256, 1, 1, 3, _t), (1, 64, 70, [256], 136, 1, 1, 0, _t),
              (3, 16, 16, [128, 128], 72, 1, 1, 4, _t)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_conv_vs_torch(hip, case, dtype):
    N, H, W, cs, Cout, KH, KW, act, tile = case
    if tile == 14 and dtype == torch.float32 and sum(cs) >= 256:
        pytest.skip("weight slice of 128 fp32 output channels x 384 does not fit the LDS (the dispatcher never picks it)")
    g = torch.Generator(device="cuda").manual_seed(CASES.index(case))
    srcs_real = [torch.randn(N, H, W, c, device="cuda", generator=g).to(dtype) for c in cs]
    cin = sum(cs)
    w = (torch.randn(Cout, cin, KH, KW, device="cuda", generator=g) / math.sqrt(cin * KH * KW)).to(dtype)
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    ref = _ref_conv(srcs_real, w, b, act, 0.5)
    srcs = []
    for t in srcs_real:                                    # zero-pad every source to a multiple of 8 channels
        cp = pack.pad8(t.shape[-1])
        buf = torch.zeros(N, H, W, cp, device="cuda", dtype=dtype)
        buf[..., :t.shape[-1]] = t
        srcs.append(buf)
    wp = pack.pack_conv(w, dtype, [(c, pack.pad8(c)) for c in cs])
    bp = pack.pack_bias(b, Cout)
    out = hip.conv2d(srcs, wp, bp, KH, KW, wp.shape[0], act=act, out_scale=0.5, tile=tile)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (N, H, W, pack.pad8(Cout))
    err = float((out[..., :Cout].float() - ref).abs().max())
    assert err < _tol(dtype, cin * KH * KW, max(1.0, float(ref.abs().max()))), err
    if pack.pad8(Cout) > Cout:                              # padded output channels: act(0) * scale
        pad = out[..., Cout:].float()
        assert float((pad - _act(torch.zeros(()), act) * 0.5).abs().max()) < 1e-6
    cin_p = sum(pack.pad8(c) for c in cs)
    if cin_p % pack.chunk_channels(dtype) == 0:             # K order 1 (horizontal taps adjacent in K): same result, other packing
        if tile in (12, 13, 19, 23, 24, 25, 26):
            return                                           # the halo kernel takes K order 0 only
        wk = pack.pack_conv(w, dtype, [(c, pack.pad8(c)) for c in cs], korder=1)
        out1 = hip.conv2d(srcs, wk, bp, KH, KW, wk.shape[0], act=act, out_scale=0.5, tile=tile, korder=1)
        assert float((out1[..., :Cout].float() - ref).abs().max()) < _tol(dtype, cin * KH * KW, max(1.0, float(ref.abs().max())))


FRAG_CASES = [
    # N, H, W, [src channels], Cout, KH, KW, act, epi      (K order 2: weights as an MFMA fragment stream, conv_frag_kernel)
    (1, 17, 23, [128], 128, 3, 3, 1, 0),                   # ragged patch rows / columns
    (2, 19, 70, [128, 128], 128, 1, 3, 4, 2),              # GRU pass (1x3) with two sources, two channel chunks, tanh * aux
    (1, 33, 19, [128, 128], 128, 3, 1, 3, 2),              # GRU pass (3x1), sigmoid * aux
    (1, 12, 20, [256], 256, 3, 3, 2, 1),                   # two cout blocks, residual add
    (1, 21, 37, [96, 64, 32], 384, 3, 3, 1, 0),            # Cin = 192: a source boundary inside a chunk, half-empty last chunk
    (1, 64, 76, [128], 128, 3, 3, 0, 0),                   # several patches per row, exact tiling
    (2, 37, 45, [128], 256, 3, 3, 0, 1),                   # residual add without activation (ConvBlock2D's last conv), ragged, 2 cout blocks
    (2, 19, 70, [128, 128], 128, 1, 3, 4, 3),              # GRU pass (1x3): tanh + GRU blend, two epilogue operands (64-pixel blocks only)
    (1, 8, 40, [128, 128, 64, 64], 128, 3, 3, 1, 0),       # four sources, three chunks
    # the M model's widths: 192-cout blocks (six waves) on 192-channel chunks
    (1, 17, 23, [192], 192, 3, 3, 1, 0),                   # ragged
    (2, 37, 45, [192], 192, 3, 3, 0, 1),                   # residual add
    (1, 33, 19, [192, 192], 192, 3, 1, 3, 2),              # GRU gate: two sources = two chunks, sigmoid * aux
    (2, 19, 70, [192, 192], 192, 1, 3, 4, 3),              # GRU candidate: two-operand blend (64-pixel blocks only)
    (1, 64, 76, [96, 96], 384 + 192, 3, 3, 1, 0),          # three cout blocks of 192, a source boundary inside the chunk
]


# tile: 4 / 2 = patch height of the 32-wide patches (128- / 64-pixel blocks), 40 = 4x40 patches (160-pixel blocks, 5 MFMA tiles that run
# across patch rows); the dispatcher picks by grid size and wave quantisation
@pytest.mark.parametrize("ph", [4, 2, 40])
@pytest.mark.parametrize("case", FRAG_CASES + [(1, 256, 304, [128], 128, 3, 3, 1, 0), (1, 45, 83, [256], 128, 3, 1, 0, 0)])
def test_conv_frag_stream(hip, case, ph):
    N, H, W, cs, Cout, KH, KW, act, epi = case
    if epi == 3 and ph == 4:
        pytest.skip("two-operand epilogues run on 64-pixel blocks only")
    if epi == 3 and ph == 40:
        pytest.skip("160-pixel blocks take one-operand epilogues only")
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(100 + H + W)
    srcs = [torch.randn(N, H, W, c, device="cuda", generator=g).to(dtype) for c in cs]
    cin = sum(cs)
    w = (torch.randn(Cout, cin, KH, KW, device="cuda", generator=g) / math.sqrt(cin * KH * KW)).to(dtype)
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    assert pack.frag_eligible(Cout, cin, KH, KW, dtype)
    ref = _ref_conv(srcs, w, b, act)
    a0 = torch.rand(N, H, W, Cout, device="cuda", generator=g).to(dtype) if epi else None
    a1 = torch.randn(N, H, W, Cout, device="cuda", generator=g).to(dtype) if epi == 3 else None
    if epi == 1:
        ref = ref.half().float() + a0.float()
    elif epi == 2:
        ref = ref.half().float() * a0.float()
    elif epi == 3:
        ref = (1 - a0.float()) * a1.float() + a0.float() * ref.half().float()
    wf = pack.pack_conv_frag(w, dtype, [(c, c) for c in cs])
    bp = pack.pack_bias(b, Cout)
    out = hip.conv2d(srcs, wf, bp, KH, KW, Cout, act=act, epi=epi, aux0=a0, aux1=a1, korder=2, tile=ph)
    # same layer through the v3 halo tile (K order 0): both must agree with the reference, and with each other to fp16 rounding
    w0 = pack.pack_conv(w, dtype, [(c, c) for c in cs])
    out0 = hip.conv2d(srcs, w0, bp, KH, KW, Cout, act=act, epi=epi, aux0=a0, aux1=a1)
    torch.cuda.synchronize()
    tol = _tol(dtype, cin * KH * KW, max(1.0, float(ref.abs().max())))
    assert float((out.float() - ref).abs().max()) < tol
    assert float((out.float() - out0.float()).abs().max()) < tol
    assert float((out != out0).float().mean()) < 0.02          # only fp32 summation order differs: rare 1-ulp flips of the fp16 outputs


def test_conv_frag_argument_checks(hip):
    x = torch.zeros(1, 8, 8, 128, device="cuda", dtype=torch.float16)
    w = pack.pack_conv_frag(torch.zeros(128, 128, 3, 3), torch.float16)
    with pytest.raises(RuntimeError, match="K order 2"):
        hip.conv2d([x], w, None, 3, 3, 128, korder=2, stride=2)
    x32 = x.float()
    with pytest.raises((RuntimeError, ValueError)):
        hip.conv2d([x32], w.float(), None, 3, 3, 128, korder=2)
    a = torch.zeros(1, 8, 8, 128, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError, match="one-operand"):            # two-operand epilogues: 64-pixel blocks only
        hip.conv2d([x], w, None, 3, 3, 128, korder=2, epi=3, aux0=a, aux1=a, tile=4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("epi", [1, 2, 3, 4])
def test_conv_epilogues(hip, dtype, epi):
    N, H, W, C = 1, 18, 26, 128
    g = torch.Generator(device="cuda").manual_seed(epi)
    x = torch.randn(N, H, W, 2 * C, device="cuda", generator=g).to(dtype)
    w = (torch.randn(C, 2 * C, 3, 3, device="cuda", generator=g) / math.sqrt(18 * C)).to(dtype)
    a0 = torch.rand(N, H, W, C, device="cuda", generator=g).to(dtype)
    a1 = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
    act = {1: 0, 2: 3, 3: 4, 4: 3}[epi]
    v = _ref_conv([x], w, None, act)
    v = v.to(dtype).float()                                  # the kernel stages the activated value in the I/O dtype
    if epi == 1:
        ref = v + a0.float()
    elif epi == 2:
        ref = v * a0.float()
    elif epi == 3:
        ref = (1 - a0.float()) * a1.float() + a0.float() * v
    else:
        gt = v.clamp(0.01, 0.99)
        ref = gt * a0.float() + (1 - gt) * a1.float()
    # sources given as two channel-slice views of one wider tensor (exercises pixel strides)
    out = hip.conv2d([x[..., :C], x[..., C:]], pack.pack_conv(w, dtype), None, 3, 3, C, act=act, epi=epi, aux0=a0, aux1=a1)
    err = float((out.float() - ref).abs().max())
    assert err < _tol(dtype, 18 * C, 4.0), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("cout", [64, 9, 16])
def test_convT_2x2_stride2(hip, dtype, cout):
    N, H, W, C = 2, 13, 21, 128
    g = torch.Generator(device="cuda").manual_seed(cout)
    x = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
    w = (torch.randn(C, cout, 2, 2, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    ref = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=2).permute(0, 2, 3, 1)
    wp, cp = pack.pack_convT_2x2s2(w, dtype)
    out = hip.conv2d([x], wp, pack.pack_bias_shuffle(b, cout), 1, 1, 4 * cp, shuffle2=cp)
    assert tuple(out.shape) == (N, 2 * H, 2 * W, cp)
    err = float((out[..., :cout].float() - ref).abs().max())
    assert err < _tol(dtype, C, 4.0), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_convT_3x3_stride1_and_output_slice(hip, dtype):
    N, H, W = 1, 22, 30
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(N, H, W, 3, device="cuda", generator=g).to(dtype)
    w = (torch.randn(3, 16, 3, 3, device="cuda", generator=g) / 5).to(dtype)
    b = torch.randn(16, device="cuda", generator=g) * 0.1
    ref = F.relu(F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1)).permute(0, 2, 3, 1)
    xp = torch.zeros(N, H, W, 8, device="cuda", dtype=dtype)
    xp[..., :3] = x
    wide = torch.full((N, H, W, 48), 7.0, device="cuda", dtype=dtype)          # write into channels 16..31 of a wider tensor
    hip.conv2d([xp], pack.pack_conv(pack.convT_s1_as_conv(w), dtype, [(3, 8)]), pack.pack_bias(b, 16), 3, 3, 16, act=2,
               out=wide[..., 16:32])
    assert float((wide[..., 16:32].float() - ref).abs().max()) < _tol(dtype, 27, 4.0)
    assert float((wide[..., :16].float() - 7).abs().max()) == 0 and float((wide[..., 32:].float() - 7).abs().max()) == 0


def test_linear_tokens_big(hip):
    """nn.Linear on (tokens, C) = 1x1 conv on a (1, 1, tokens, C) image; BASELINE-sized token count, fp16."""
    T, C = 2 * 256 * 304, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(1, 1, T, C, device="cuda", generator=g).half()
    w = (torch.randn(3 * C, C, device="cuda", generator=g) / math.sqrt(C)).half()
    b = torch.randn(3 * C, device="cuda", generator=g)
    out = hip.conv2d([x], pack.pack_conv(w, torch.float16), pack.pack_bias(b, 3 * C), 1, 1, 3 * C)
    ref = F.linear(x.float(), w.float(), b)
    assert float((out.float() - ref).abs().max()) < 2.5e-3 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("tile", [0, 2, 3, 6, 20])
@pytest.mark.parametrize("shape", [(2, 32, 38, 256, 768, 0), (1, 64, 76, 128, 128, 1), (2, 9, 13, 384, 384, 1), (1, 50, 61, 128, 384, 0)])
def test_pre_layernorm_folded_into_1x1(hip, dtype, tile, shape):
    """LayerNorm (no affine) + Linear as one launch (attentions.py:117 pre-norm) == F.layer_norm -> F.linear -> act in fp32, on rows
    with a mean several times their standard deviation (the subtraction rstd*(W.x - mean*rowsum(W)) must survive it)."""
    n, h, w, cin, cout, act = shape
    if tile == 3 and cout > 128:
        pytest.skip("narrow-head tile")
    g = torch.Generator(device="cuda").manual_seed(cin + cout + tile)
    x = (torch.randn(n, h, w, cin, device="cuda", generator=g) * (0.5 + torch.rand(n, h, w, 1, device="cuda", generator=g) * 3)
         + torch.randn(n, h, w, 1, device="cuda", generator=g) * 4).to(dtype)
    wt = (torch.randn(cout, cin, 1, 1, device="cuda", generator=g) / math.sqrt(cin)).to(dtype)
    b = torch.randn(cout, device="cuda", generator=g)
    wp, bp = pack.pack_conv(wt, dtype), pack.pack_bias(b, cout)
    y = hip.conv2d([x], wp, bp, 1, 1, cout, act=act, tile=tile, ln_wsum=wp.float().sum(1).contiguous())
    ref = _act(F.linear(F.layer_norm(x.float(), (cin,)), wt.float().reshape(cout, cin), b), act)
    # reference path of the engine before the fusion: separate K6 LayerNorm (output rounded to the I/O dtype) + plain K5
    y2 = hip.conv2d([hip.layernorm(x)], wp, bp, 1, 1, cout, act=act, tile=tile)
    tol = 2e-4 if dtype == torch.float32 else 6e-3
    assert float((y.float() - ref).abs().max()) < tol
    assert float((y.float() - ref).abs().max()) <= max(2 * float((y2.float() - ref).abs().max()), tol / 4)


def test_pre_layernorm_rejects_spatial_kernels(hip):
    x = torch.randn(1, 8, 8, 128, device="cuda").half()
    wp = torch.randn(128, 9 * 128, device="cuda").half()
    with pytest.raises(RuntimeError, match="pre-LayerNorm"):
        hip.conv2d([x], wp, None, 3, 3, 128, ln_wsum=torch.zeros(128, device="cuda"))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("tile", [0, 2, 6, 20])
@pytest.mark.parametrize("shape", [(1, 50, 61, 128, 256, 128), (2, 16, 19, 256, 512, 256), (1, 9, 13, 64, 64, 96)])
def test_dual_gemm_gate_mix(hip, dtype, tile, shape):
    """S2M2_EPI_DUALMIX (the two heads of FeatureFusion, feature_fusion.py:15-31, in one launch) == the two separate launches
    and the fp32 formula: (W2.h[ks:] + b2) + g*z0 + (1-g)*z1, g = clamp(sigmoid(W1.h[:ks] + b1), .01, .99)."""
    n, h, w, ks, k2, cout = shape
    g = torch.Generator(device="cuda").manual_seed(ks + k2 + tile)
    hdn = torch.randn(n, h, w, ks + k2, device="cuda", generator=g).to(dtype)
    z0 = torch.randn(n, h, w, cout, device="cuda", generator=g).to(dtype)
    z1 = torch.randn(n, h, w, cout, device="cuda", generator=g).to(dtype)
    w1 = (torch.randn(cout, ks, 1, 1, device="cuda", generator=g) / math.sqrt(ks)).to(dtype)
    w2 = (torch.randn(cout, k2, 1, 1, device="cuda", generator=g) / math.sqrt(k2)).to(dtype)
    b1, b2 = torch.randn(cout, device="cuda", generator=g), torch.randn(cout, device="cuda", generator=g)
    p1, p2 = pack.pack_conv(w1, dtype), pack.pack_conv(w2, dtype)
    wcat = torch.cat([p1, p2], dim=1).contiguous()
    y = hip.conv2d([hdn], wcat, pack.pack_bias(b1, cout), 1, 1, cout, act=hip.ACT_SIGMOID, epi=hip.EPI_DUALMIX, aux0=z0, aux1=z1,
                   ksplit=ks, bias2=pack.pack_bias(b2, cout), tile=tile)
    gate = torch.sigmoid(F.linear(hdn[..., :ks].float(), w1.float().reshape(cout, ks), b1)).clamp(0.01, 0.99)
    ref = F.linear(hdn[..., ks:].float(), w2.float().reshape(cout, k2), b2) + gate * z0.float() + (1 - gate) * z1.float()
    m = hip.conv2d([hdn[..., :ks]], p1, pack.pack_bias(b1, cout), 1, 1, cout, act=hip.ACT_SIGMOID, epi=hip.EPI_GATEMIX, aux0=z0, aux1=z1)
    y2 = hip.conv2d([hdn[..., ks:]], p2, pack.pack_bias(b2, cout), 1, 1, cout, epi=hip.EPI_ADD, aux0=m)
    tol = 1e-4 if dtype == torch.float32 else 8e-3
    assert float((y.float() - ref).abs().max()) < tol
    assert float((y.float() - y2.float()).abs().max()) < tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 32, 38, 128, 256), (1, 9, 15, 256, 256), (2, 64, 76, 128, 128), (1, 2, 2, 128, 128), (1, 16, 24, 640, 128)])
def test_conv_pool2_equals_avgpool_then_conv(hip, shape, dtype):
    """pool2 (AvgPool2d(2) folded into the operand load of a 1x1 layer, the down_convs of unet.py:24-29 / stacked_MRT.py:21-26): bit-identical
    to the stand-alone K7 pooling launch followed by the same layer, odd sizes drop the last row / column like nn.AvgPool2d."""
    N, H, W, cin, cout = shape
    g = torch.Generator(device="cuda").manual_seed(H * W + cin)
    x = torch.randn(N, H, W, cin, device="cuda", generator=g).to(dtype)
    w = (torch.randn(cout, cin, 1, 1, device="cuda", generator=g) / math.sqrt(cin)).to(dtype)
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    wp, bp = pack.pack_conv(w, dtype), pack.pack_bias(b, cout)
    fused = hip.conv2d([x], wp, bp, 1, 1, cout, act=1, pool2=True)
    pooled = hip.resample2x(x[:, :H // 2 * 2, :W // 2 * 2].contiguous(), 0)       # (the K7 launch takes even sizes only)
    assert tuple(pooled.shape) == (N, H // 2, W // 2, cin)
    ref = hip.conv2d([pooled], wp, bp, 1, 1, cout, act=1, tile=6 if cin <= 512 else 2)
    torch.cuda.synchronize()
    assert tuple(fused.shape) == (N, H // 2, W // 2, cout)
    assert torch.equal(fused, ref)
    tref = F.gelu(F.conv2d(F.avg_pool2d(x.float().permute(0, 3, 1, 2), 2).to(dtype).float(), w.float(), b)).permute(0, 2, 3, 1)
    assert float((fused.float() - tref).abs().max()) < (2e-2 if dtype == torch.float16 else 2e-4)
    with pytest.raises(RuntimeError, match="pool2"):
        hip.conv2d([x], pack.pack_conv(torch.zeros(cout, cin, 3, 3), dtype), None, 3, 3, cout, pool2=True)


@pytest.mark.parametrize("shape,k", [((1, 37, 83, 128), (3, 1)), ((2, 16, 40, 128), (1, 3)), ((1, 256, 304, 128), (3, 1))])
@pytest.mark.parametrize("tile", [0, 2, 40])
def test_conv_frag_two_stacked_layers_with_different_epilogues(hip, shape, k, tile):
    """epi_cout0: z = sigmoid(convz(cat(h, x))) and r*h = sigmoid(convr(cat(h, x))) * h of ConvGRU (refinenet.py:24-29) as ONE launch with
    the two layers stacked along Cout -- bit-identical to the two separate launches (same K order, same epilogue arithmetic)."""
    N, H, W, C = shape
    KH, KW = k
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(H + W)
    h = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
    x = torch.randn(N, H, W, C, device="cuda", generator=g).to(dtype)
    ws = [(torch.randn(C, 2 * C, KH, KW, device="cuda", generator=g) / math.sqrt(2 * C * KH * KW)).to(dtype) for _ in range(2)]
    bs = [torch.randn(C, device="cuda", generator=g) * 0.1 for _ in range(2)]
    sp = [(C, C), (C, C)]
    z = hip.conv2d([h, x], pack.pack_conv_frag(ws[0], dtype, sp), pack.pack_bias(bs[0], C), KH, KW, C, act=hip.ACT_SIGMOID, korder=2)
    rh = hip.conv2d([h, x], pack.pack_conv_frag(ws[1], dtype, sp), pack.pack_bias(bs[1], C), KH, KW, C, act=hip.ACT_SIGMOID, epi=hip.EPI_MUL,
                    aux0=h, korder=2)
    wz = pack.pack_conv_frag(torch.cat(ws, 0), dtype, sp)
    both = hip.conv2d([h, x], wz, pack.pack_bias(torch.cat(bs), 2 * C), KH, KW, 2 * C, act=hip.ACT_SIGMOID, epi=hip.EPI_MUL, aux0=h,
                      korder=2, epi_cout0=C, tile=tile)
    assert torch.equal(both[..., :C], z) and torch.equal(both[..., C:], rh)
    ref = torch.sigmoid(_ref_conv([h, x], ws[1], bs[1], 0)).half().float() * h.float()
    assert float((both[..., C:].float() - ref).abs().max()) < 2e-2
    with pytest.raises(RuntimeError, match="epi_cout0"):           # not a multiple of the 128-cout block
        wide = torch.cat([h, h[..., :64]], -1).contiguous()
        hip.conv2d([h, x], wz, pack.pack_bias(torch.cat(bs), 2 * C), KH, KW, 2 * C, act=hip.ACT_SIGMOID, epi=hip.EPI_MUL, aux0=wide, korder=2, epi_cout0=64)
