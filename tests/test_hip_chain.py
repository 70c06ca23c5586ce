"""GPU parity of K9 (s2m2_mlp_chain: chains of 1x1 layers with the row tile resident in LDS) against a plain PyTorch fp32
reference of the same layers, rounding every stage output to the I/O dtype where the kernel does, and against the separate
K5/K6 launches it replaces (reference attentions.py:311-321,347-355 -- proj + residual + pre-LN FFN; :269-275 -- 1x1 branch).

Tolerances: fp32 -- exact-fp32 MFMA chain vs torch's summation order through up to three layers: 3e-4 absolute on O(1) values.
fp16 -- every stage output is rounded to fp16 (rel 2^-11) and feeds the next GEMM: 1.5e-2 absolute on O(1-4) values."""
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


def _make(C, nst, dtype, ln_stage, acts, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    raw, packed = [], []
    for s in range(nst):
        w = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
        b = torch.randn(C, device="cuda", generator=g) * 0.5 if s != 1 else None
        wp = pack.pack_conv(w, dtype)
        raw.append((w.reshape(C, C).float(), b, acts[s], s == ln_stage))
        packed.append((wp, None if b is None else pack.pack_bias(b, C), acts[s], wp.float().sum(1).contiguous() if s == ln_stage else None))
    return raw, packed


def _ref(x, raw, res, res_stage, carry, dtype):
    t, ys = x.float(), []
    for s, (w, b, act, ln) in enumerate(raw):
        a = F.layer_norm(t, (t.shape[-1],)) if ln else t
        y = ACT[act](F.linear(a, w, b)).to(dtype).float()
        if s == res_stage:
            y = (y + res.float()).to(dtype).float()
        if carry and s == 2:
            y = (y + ys[0]).to(dtype).float()
        ys.append(y)
        t = y
    return t


CASES = [  # C, dtype, rows-shape, nstage, ln_stage, acts, res_stage, carry
    (128, torch.float16, (2, 50, 61), 3, 1, (0, 1, 0), 0, True),          # proj + residual + pre-LN FFN
    (256, torch.float16, (2, 32, 38), 3, 1, (0, 1, 0), 0, True),
    (256, torch.float16, (1, 1, 1), 3, 1, (0, 1, 0), 0, True),
    (128, torch.float16, (1, 37, 41), 2, -1, (2, 0), -1, False),          # ConvBlock2D 1x1 branch
    (256, torch.float16, (1, 16, 19), 2, -1, (2, 0), -1, False),
    (128, torch.float16, (3, 7, 9), 1, 0, (1,), 0, False),
    (128, torch.float16, (1, 9, 30), 2, 0, (1, 0), 1, False),             # residual on the last stage
    (384, torch.float16, (2, 16, 19), 3, 1, (0, 1, 0), 0, True),
    (512, torch.float16, (2, 16, 19), 3, 1, (0, 1, 0), 0, True),
    (512, torch.float16, (1, 5, 7), 2, -1, (2, 0), -1, False),
    (128, torch.float32, (2, 20, 31), 3, 1, (0, 1, 0), 0, True),
    (256, torch.float32, (2, 16, 19), 3, 1, (0, 1, 0), 0, True),
    (256, torch.float32, (1, 6, 11), 2, -1, (2, 0), -1, False),
    (128, torch.float32, (1, 9, 30), 2, 0, (1, 0), 1, False),
    # long row counts at C = 128 fp16 (64-row tiles), incl. ragged row counts
    (128, torch.float16, (2, 256, 304), 3, 1, (0, 1, 0), 0, True),
    (128, torch.float16, (1, 255, 303), 2, -1, (2, 0), -1, False),
    (128, torch.float16, (1, 250, 301), 1, 0, (1,), 0, False),
    (128, torch.float16, (1, 200, 300), 2, 0, (1, 0), 1, False),
    (128, torch.float16, (3, 111, 101), 3, 1, (0, 1, 0), 0, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"C{c[0]}-{'h' if c[1] == torch.float16 else 'f'}-{'x'.join(map(str, c[2]))}-n{c[3]}")
def test_chain_vs_torch(hip, case):
    C, dtype, shp, nst, ln_stage, acts, res_stage, carry = case
    assert hip.mlp_chain_supported(C, dtype)
    g = torch.Generator(device="cuda").manual_seed(C + nst)
    x = torch.randn(*shp, C, device="cuda", generator=g).to(dtype)
    res = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 1).to(dtype) if res_stage >= 0 else None
    raw, packed = _make(C, nst, dtype, ln_stage, acts, 7 * C + nst)
    y = hip.mlp_chain(x, packed, res=res, res_stage=res_stage, carry=carry)
    ref = _ref(x, raw, res, res_stage, carry, dtype)
    assert y.shape == x.shape and y.dtype == dtype
    tol = 3e-4 if dtype == torch.float32 else 1.5e-2
    assert float((y.float() - ref).abs().max()) < tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_chain_matches_separate_launches_and_strided_rows(hip, dtype):
    """the launches K9 replaces: K5 proj with EPI_ADD, K5 pre-LN FFN layer with GELU, K5 with EPI_ADD -- on rows that are a
    channel slice of a wider tensor (row stride 3C, like the q/k/v views)"""
    C, shp = 128, (2, 24, 40)
    g = torch.Generator(device="cuda").manual_seed(5)
    wide = torch.randn(*shp, 3 * C, device="cuda", generator=g).to(dtype)
    o = wide[..., C:2 * C]
    z = torch.randn(*shp, C, device="cuda", generator=g).to(dtype)
    raw, st = _make(C, 3, dtype, 1, (0, 1, 0), 11)
    y = hip.mlp_chain(o, st, res=z, res_stage=0, carry=True)
    z1 = hip.conv2d([o.contiguous()], st[0][0], st[0][1], 1, 1, C, epi=hip.EPI_ADD, aux0=z)
    h = hip.conv2d([z1], st[1][0], st[1][1], 1, 1, C, act=hip.ACT_GELU, ln_wsum=st[1][3])
    y2 = hip.conv2d([h], st[2][0], st[2][1], 1, 1, C, epi=hip.EPI_ADD, aux0=z1)
    assert float((y.float() - y2.float()).abs().max()) < (1e-4 if dtype == torch.float32 else 1.5e-2)


def test_chain_rejects_bad_arguments(hip):
    x = torch.randn(4, 192, device="cuda").half()
    w = torch.randn(192, 192, device="cuda").half()
    assert not hip.mlp_chain_supported(192, torch.float16)
    assert not hip.mlp_chain_supported(512, torch.float32)
    with pytest.raises(RuntimeError, match="not supported"):
        hip.mlp_chain(x, [(w, None, 0, None)])
    x = torch.randn(4, 128, device="cuda").half()
    w = torch.randn(128, 128, device="cuda").half()
    with pytest.raises(RuntimeError, match="carry"):
        hip.mlp_chain(x, [(w, None, 0, None), (w, None, 0, None)], carry=True)
    with pytest.raises(RuntimeError, match="act"):
        hip.mlp_chain(x, [(w, None, 3, None)])


@pytest.mark.parametrize("C,dtype,shp,nst", [(128, torch.float16, (2, 50, 61), 3), (256, torch.float16, (2, 32, 38), 3), (512, torch.float16, (1, 5, 7), 2),
                                              (128, torch.float16, (2, 256, 304), 3), (128, torch.float16, (1, 201, 299), 2),
                                              (128, torch.float32, (2, 20, 31), 3), (256, torch.float32, (1, 6, 11), 2), (128, torch.float16, (1, 1, 1), 1)])
def test_chain_layernorm_second_output(hip, C, dtype, shp, nst):
    """ln_out: LayerNorm(out rows) * gamma + beta as a second output of the last stage (DispInit's layer_norm, submodules.py:165,216,
    folded into the launch that writes feature_tr_4x) == F.layer_norm of the STORED rows, fp32 statistics, rounded to the I/O dtype;
    the first output is bit-identical to the launch without it."""
    assert hip.mlp_chain_ln_out_supported(C, dtype)
    g = torch.Generator(device="cuda").manual_seed(C + nst)
    x = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 0.3).to(dtype)
    res = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 1).to(dtype)
    acts = (0, 1, 0)[:nst] if nst == 3 else ((2, 0) if nst == 2 else (1,))
    raw, packed = _make(C, nst, dtype, 1 if nst == 3 else -1, acts, 3 * C + nst)
    gam = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    bet = 0.05 * torch.randn(C, device="cuda", generator=g)
    plain = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=nst == 3)
    y, yn = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=nst == 3, ln_out=(gam, bet, 1e-5))
    assert torch.equal(y, plain)
    ref = F.layer_norm(y.float(), (C,), gam, bet, 1e-5)
    err = float((yn.float() - ref).abs().max())
    # fp32: summation order of the statistics only;  fp16: one rounding of an O(1..4) value
    assert err < (2e-5 if dtype == torch.float32 else 4e-3), err
    assert hip.mlp_chain_ln_out_supported(384, torch.float16) and hip.mlp_chain_ln_out_supported(192, torch.float16)
    assert not hip.mlp_chain_ln_out_supported(192, torch.float32)


def test_chain_xcd_placement_hint_changes_nothing_but_the_block_order(hip):
    """xcd_group_rows only permutes which block computes which row tile: outputs are bit-identical."""
    C, dtype = 128, torch.float16
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2, 64, 152, C, device="cuda", generator=g).to(dtype)             # 2 images x 8 groups x (8 rows x 152) = 1216 rows = 19 tiles of 64
    res = torch.randn(2, 64, 152, C, device="cuda", generator=g).to(dtype)
    raw, packed = _make(C, 3, dtype, 1, (0, 1, 0), 11)
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    a = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=True, ln_out=(gam, bet, 1e-5))
    b = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=True, ln_out=(gam, bet, 1e-5), xcd_group_rows=8 * 152)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    c = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=True, xcd_group_rows=100)      # not a multiple of the tile height: hint ignored
    assert torch.equal(a[0], c)


@pytest.mark.parametrize("C,dtype,shp,nst,nfan,ln", [(128, torch.float16, (2, 50, 61), 3, 3, True), (256, torch.float16, (2, 32, 38), 3, 3, True),
                                                      (256, torch.float16, (1, 64, 76), 3, 3, True), (384, torch.float16, (1, 16, 19), 3, 3, True),
                                                      (512, torch.float16, (1, 5, 7), 2, 2, False), (128, torch.float32, (2, 20, 31), 3, 3, True),
                                                      (256, torch.float32, (1, 6, 11), 3, 3, True), (128, torch.float16, (1, 1, 1), 1, 1, True),
                                                      (128, torch.float16, (2, 256, 304), 3, 3, True)])
def test_chain_fan_out_stages(hip, C, dtype, shp, nst, nfan, ln):
    """Fan-out stages (the fused Q | K | V projection of the next attention): nfan C -> C layers on the chain's OUTPUT rows with the
    pre-LayerNorm folded in == the stand-alone K5 launch with the same folded LayerNorm on the stored rows (what the engine ran before),
    and == F.linear(F.layer_norm(out)) in fp32; the chain's own output is bit-identical to the launch without fan-out stages."""
    g = torch.Generator(device="cuda").manual_seed(C + nst + nfan)
    x = (torch.randn(*shp, C, device="cuda", generator=g) * 1.5).to(dtype)
    res = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 1).to(dtype)
    acts = (0, 1, 0)[:nst] if nst == 3 else ((2, 0) if nst == 2 else (1,))
    raw, packed = _make(C, nst, dtype, 1 if nst == 3 else -1, acts, 5 * C + nst)
    wq = (torch.randn(nfan * C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
    bq = torch.randn(nfan * C, device="cuda", generator=g) * 0.3
    wp = pack.pack_conv(wq, dtype)
    bp = pack.pack_bias(bq, nfan * C)
    wsum = wp.float().sum(1).contiguous() if ln else None
    plain = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=nst == 3)
    y, q = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=nst == 3, fan=(wp, bp, wsum))
    assert torch.equal(y, plain) and q.shape == (*shp, nfan * C)
    a = F.layer_norm(y.float(), (C,)) if ln else y.float()
    ref = F.linear(a, wq.reshape(nfan * C, C).float(), bq)
    err = float((q.float() - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert err < (3e-4 if dtype == torch.float32 else 4e-3 * scale), (err, scale)
    sep = hip.conv2d([y.reshape(1, 1, -1, C)], wp, bp, 1, 1, nfan * C, ln_wsum=wsum).reshape(q.shape)
    assert float((q.float() - sep.float()).abs().max()) < (3e-4 if dtype == torch.float32 else 2 ** -7 * scale)
    # with a LayerNorm second output as well: three results, unchanged
    if hip.mlp_chain_ln_out_supported(C, dtype):
        gam = torch.ones(C, device="cuda")
        bet = torch.zeros(C, device="cuda")
        y3, n3, q3 = hip.mlp_chain(x, packed, res=res, res_stage=0, carry=nst == 3, ln_out=(gam, bet, 1e-5), fan=(wp, bp, wsum))
        assert torch.equal(y3, y) and torch.equal(q3, q)
        assert float((n3.float() - F.layer_norm(y.float(), (C,))).abs().max()) < (2e-5 if dtype == torch.float32 else 4e-3)


# ---- direct form (s2m2_chain_desc.weight_frag: weights in MFMA-fragment order, straight into the operand registers) ------------------------
def _frag(packed):
    return [(pack.chain_frag(w), b, act, ws) for w, b, act, ws in packed]


@pytest.mark.parametrize("C,shp,nst,res_stage,carry,ln_stage,acts,nfan,fan_ln", [
    (256, (2, 32, 38), 3, 0, True, 1, (0, 1, 0), 3, True),          # the 1/32 attention block of the S model with the next Q|K|V
    (256, (1, 64, 76), 3, 0, True, 1, (0, 1, 0), 0, False),
    (128, (2, 20, 31), 3, 0, True, 1, (0, 1, 0), 3, True),          # ragged last tile
    (128, (1, 128, 152), 2, -1, False, -1, (2, 0), 0, False),       # > 8192 rows: the row-major form runs 64-row tiles
    (128, (1, 1, 1), 1, 0, False, 0, (1,), 1, False),
    (256, (1, 5, 7), 2, 1, False, 0, (1, 2), 2, True),
    (256, (2, 64, 76), 3, 0, True, 1, (0, 1, 0), 3, True),          # 304 blocks
    (128, (2, 128, 152), 3, 0, True, 1, (0, 1, 0), 3, True),        # 1216 blocks
])
def test_chain_direct_form_equals_row_major_form_bit_for_bit(hip, C, shp, nst, res_stage, carry, ln_stage, acts, nfan, fan_ln):
    """Same fp16 operands, same k16 order of every MFMA chain, same epilogues: the direct form (fragments from global memory into the operand
    registers, no weight tile in LDS) must reproduce the row-major form BIT FOR BIT -- chain output and fan-out stages."""
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(3 * C + nst + nfan)
    x = (torch.randn(*shp, C, device="cuda", generator=g) * 1.5).to(dtype)
    res = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 1).to(dtype)
    raw, packed = _make(C, nst, dtype, ln_stage, acts, 13 * C + nst)
    kw = dict(res=res if res_stage >= 0 else None, res_stage=res_stage, carry=carry)
    fan = fanf = None
    if nfan:
        wq = (torch.randn(nfan * C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
        wp = pack.pack_conv(wq, dtype)
        bp = pack.pack_bias(torch.randn(nfan * C, device="cuda", generator=g) * 0.3, nfan * C)
        ws = wp.float().sum(1).contiguous() if fan_ln else None
        fan, fanf = (wp, bp, ws), (pack.chain_frag(wp), bp, ws)
    a = hip.mlp_chain(x, packed, fan=fan, **kw)
    a = a if nfan else (a,)
    fpacked = _frag(packed)
    for _ in range(6):                                              # (repeated: the form has no barrier inside a stage -- a race would show as a flaky mismatch)
        b = hip.mlp_chain(x, fpacked, fan=fanf, frag=True, **kw)
        b = b if nfan else (b,)
        assert len(a) == len(b)
        for u, v in zip(a, b):
            assert u.shape == v.shape and torch.equal(u, v)
    ref = _ref(x, raw, res, res_stage, carry, dtype)
    assert float((b[0].float() - ref).abs().max()) < 1.5e-2


def test_chain_direct_form_layernorm_output_and_strided_rows(hip):
    C, dtype = 128, torch.float16
    g = torch.Generator(device="cuda").manual_seed(17)
    wide = (torch.randn(2, 24, 40, 3 * C, device="cuda", generator=g)).to(dtype)
    o = wide[..., C:2 * C]                                          # row stride 3C
    z = torch.randn(2, 24, 40, C, device="cuda", generator=g).to(dtype)
    raw, packed = _make(C, 3, dtype, 1, (0, 1, 0), 23)
    gam = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    bet = 0.05 * torch.randn(C, device="cuda", generator=g)
    y0, n0 = hip.mlp_chain(o, packed, res=z, res_stage=0, carry=True, ln_out=(gam, bet, 1e-5))
    y1, n1 = hip.mlp_chain(o, _frag(packed), res=z, res_stage=0, carry=True, ln_out=(gam, bet, 1e-5), frag=True)
    assert torch.equal(y0, y1) and torch.equal(n0, n1)


def test_chain_direct_form_rejects_unsupported(hip):
    assert all(hip.mlp_chain_frag_supported(c, torch.float16) for c in (128, 192, 256, 384, 512))
    assert not hip.mlp_chain_frag_supported(640, torch.float16) and not hip.mlp_chain_frag_supported(128, torch.float32)
    x = torch.randn(4, 640, device="cuda").half()
    w = torch.randn(640, 640, device="cuda").half()
    with pytest.raises(RuntimeError, match="weight_frag"):
        hip.mlp_chain(x, [(w, None, 0, None)], frag=True)


# ---- the widths of the M and XL models (C = 192 / 384: six / twelve waves per block, rows of 24 / 48 16-byte pieces) ---------------------------
@pytest.mark.parametrize("C,shp,nst,res_stage,carry,ln_stage,acts,nfan,fan_ln", [
    (384, (2, 16, 19), 3, 0, True, 1, (0, 1, 0), 3, True),          # the 1/32 attention block of the M model with the next Q|K|V
    (384, (1, 64, 76), 3, 0, True, 1, (0, 1, 0), 0, False),
    (384, (1, 5, 7), 2, 1, False, 0, (1, 2), 2, True),
    (384, (1, 1, 1), 1, 0, False, 0, (1,), 1, False),
])
def test_chain_direct_form_c384_equals_row_major_form_bit_for_bit(hip, C, shp, nst, res_stage, carry, ln_stage, acts, nfan, fan_ln):
    """C = 384 has both forms: the direct form (half a stage of fragments in flight, 12 waves) reproduces the LDS-staged one bit for bit"""
    test_chain_direct_form_equals_row_major_form_bit_for_bit(hip, C, shp, nst, res_stage, carry, ln_stage, acts, nfan, fan_ln)


# ---- the L model's coarse levels (C = 512, r06: sixteen waves per block, four fragments in flight per wave) -------------------------------------------
@pytest.mark.parametrize("C,shp,nst,res_stage,carry,ln_stage,acts,nfan,fan_ln", [
    (512, (2, 32, 38), 3, 0, True, 1, (0, 1, 0), 3, True),          # the 1/32 attention block of the L model with the next Q|K|V
    (512, (2, 64, 76), 3, 0, True, 1, (0, 1, 0), 0, False),         # 9728 rows
    (512, (1, 5, 7), 2, 1, False, 0, (1, 2), 2, True),
    (512, (1, 1, 1), 1, 0, False, 0, (1,), 1, False),
])
def test_chain_direct_form_c512_equals_row_major_form_bit_for_bit(hip, C, shp, nst, res_stage, carry, ln_stage, acts, nfan, fan_ln):
    """C = 512 has both forms as well"""
    test_chain_direct_form_equals_row_major_form_bit_for_bit(hip, C, shp, nst, res_stage, carry, ln_stage, acts, nfan, fan_ln)


@pytest.mark.parametrize("C,shp,nst,ln_stage,acts,res_stage,carry", [
    (192, (2, 50, 61), 3, 1, (0, 1, 0), 0, True),                    # proj + residual + pre-LN FFN (32-row tiles)
    (192, (2, 128, 152), 3, 1, (0, 1, 0), 0, True),                  # 38912 rows: 64-row tiles
    (192, (1, 37, 41), 2, -1, (2, 0), -1, False),                    # ConvBlock2D 1x1 branch
    (192, (1, 9, 30), 2, 0, (1, 0), 1, False),
    (192, (1, 1, 1), 1, 0, (1,), 0, False),
    (384, (2, 256, 304), 2, -1, (2, 0), -1, False),                  # the XL model's 1/4 level: 155648 rows
])
def test_chain_direct_form_m_xl_widths_vs_torch_and_k5(hip, C, shp, nst, ln_stage, acts, res_stage, carry):
    """C = 192 exists in the direct form only: against the fp32 reference of the same layers (every stage rounded where the kernel rounds) and
    against the separate K5 launches it replaces; repeated (no barrier inside a stage: a race would be a flaky mismatch)"""
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(C + nst + shp[1])
    x = torch.randn(*shp, C, device="cuda", generator=g).to(dtype)
    res = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 1).to(dtype) if res_stage >= 0 else None
    raw, packed = _make(C, nst, dtype, ln_stage, acts, 7 * C + nst)
    ref = _ref(x, raw, res, res_stage, carry, dtype)
    first = None
    for _ in range(4):
        y = hip.mlp_chain(x, _frag(packed), res=res, res_stage=res_stage, carry=carry, frag=True)
        assert y.shape == x.shape and y.dtype == dtype
        assert float((y.float() - ref).abs().max()) < 1.5e-2
        first = y if first is None else first
        assert torch.equal(y, first)
    t, outs = x, []
    for s, (wp, bp, act, ws) in enumerate(packed):                   # the K5 launches: same operands, same rounding points
        kw = {}
        if ws is not None:
            kw["ln_wsum"] = ws
        if s == res_stage:
            kw.update(epi=hip.EPI_ADD, aux0=res)
        t = hip.conv2d([t], wp, bp, 1, 1, C, act=act, **kw)
        if carry and s == 2:
            t = (t.float() + outs[0].float()).to(dtype)
        outs.append(t)
    assert float((y.float() - t.float()).abs().max()) < 1.5e-2


@pytest.mark.parametrize("C,shp,nst,frag", [(192, (2, 50, 61), 3, True), (192, (2, 128, 152), 3, True), (192, (1, 1, 1), 1, True),
                                            (384, (2, 32, 38), 3, True), (384, (2, 32, 38), 3, False), (384, (1, 5, 7), 2, True)])
def test_chain_layernorm_second_output_m_xl_widths(hip, C, shp, nst, frag):
    """rows of 24 / 48 pieces: the LayerNorm output takes its statistics in a second pass over the stored tile (8 lanes per row)"""
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(C + nst)
    x = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 0.3).to(dtype)
    res = (torch.randn(*shp, C, device="cuda", generator=g) * 2 + 1).to(dtype)
    acts = (0, 1, 0)[:nst] if nst == 3 else ((2, 0) if nst == 2 else (1,))
    raw, packed = _make(C, nst, dtype, 1 if nst == 3 else -1, acts, 3 * C + nst)
    st = _frag(packed) if frag else packed
    gam = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    bet = 0.05 * torch.randn(C, device="cuda", generator=g)
    plain = hip.mlp_chain(x, st, res=res, res_stage=0, carry=nst == 3, frag=frag)
    for _ in range(3):
        y, yn = hip.mlp_chain(x, st, res=res, res_stage=0, carry=nst == 3, ln_out=(gam, bet, 1e-5), frag=frag)
        assert torch.equal(y, plain)
        ref = F.layer_norm(y.float(), (C,), gam, bet, 1e-5)
        assert float((yn.float() - ref).abs().max()) < 4e-3


@pytest.mark.parametrize("C,shp,nfan,ln,pool", [(192, (2, 128, 152), 3, True, False), (192, (1, 7, 9), 1, False, False), (384, (2, 32, 38), 3, True, False),
                                                (192, (2, 64, 76), 2, False, True), (384, (1, 9, 7), 1, False, True)])
def test_fan_only_direct_form_m_xl_widths(hip, C, shp, nfan, ln, pool):
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(C + nfan)
    wide = (torch.randn(*shp, C + 8, device="cuda", generator=g) * 1.5 + 0.3).to(dtype)
    x = wide[..., :C]
    wq = (torch.randn(nfan * C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
    wp = pack.pack_conv(wq, dtype)
    bp = pack.pack_bias(torch.randn(nfan * C, device="cuda", generator=g) * 0.3, nfan * C)
    ws = wp.float().sum(1).contiguous() if ln else None
    assert hip.mlp_fan_supported(C, nfan, dtype)
    if pool:
        ref = hip.conv2d([x.contiguous()], wp, bp, 1, 1, nfan * C, pool2=True)
        y = hip.mlp_fan(x, pack.chain_frag(wp), bp, None, frag=True, pool2=True)
        assert ref.shape == y.shape and float((y.float() - ref.float()).abs().max()) < 1.5e-2
        return
    ref = hip.conv2d([x.contiguous()], wp, bp, 1, 1, nfan * C, **({"ln_wsum": ws} if ln else {}))
    for _ in range(3):
        y = hip.mlp_fan(x, pack.chain_frag(wp), bp, ws, frag=True)
        assert y.shape == ref.shape and float((y.float() - ref.float()).abs().max()) < 1.5e-2
    t = F.layer_norm(x.float(), (C,)) if ln else x.float()
    full = F.linear(t, wq.float().reshape(nfan * C, C), bp[:nfan * C])
    assert float((y.float() - full).abs().max()) < 2e-2


@pytest.mark.parametrize("C,shp,nfan,ln", [(256, (2, 32, 38), 3, True), (256, (2, 64, 76), 3, True), (128, (2, 128, 152), 3, True),
                                           (128, (1, 256, 304), 3, True), (128, (1, 7, 9), 1, False), (256, (1, 5, 3), 4, False),
                                           (512, (2, 32, 38), 3, True)])
def test_fan_only_direct_form(hip, C, shp, nfan, ln):
    """nstage = 0 with weight_frag: the fan-out layers alone (a block's first Q | K | V projection) in the direct form == the K5 launch with the
    folded pre-LayerNorm within fp16 rounding; the form exists for fp16 at the direct form's widths only."""
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(C + nfan)
    wide = (torch.randn(*shp, C + 8, device="cuda", generator=g) * 1.5 + 0.3).to(dtype)
    x = wide[..., :C]                                               # strided rows
    wq = (torch.randn(nfan * C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
    wp = pack.pack_conv(wq, dtype)
    bp = pack.pack_bias(torch.randn(nfan * C, device="cuda", generator=g) * 0.3, nfan * C)
    ws = wp.float().sum(1).contiguous() if ln else None
    wf = pack.chain_frag(wp)
    ref = hip.conv2d([x.contiguous()], wp, bp, 1, 1, nfan * C, **({"ln_wsum": ws} if ln else {}))
    for _ in range(4):
        y = hip.mlp_fan(x, wf, bp, ws, frag=True)
        assert y.shape == ref.shape
        assert float((y.float() - ref.float()).abs().max()) < 1.5e-2
    assert hip.mlp_fan_supported(C, nfan, dtype) and not hip.mlp_fan_supported(640, nfan, dtype) and not hip.mlp_fan_supported(C, nfan, torch.float32)
    with pytest.raises(ValueError, match="direct form"):
        hip.mlp_fan(x, wp, bp, ws, frag=False)
    t = F.layer_norm(x.float(), (C,)) if ln else x.float()
    full = F.linear(t, wq.float().reshape(nfan * C, C), bp[:nfan * C])
    assert float((y.float() - full).abs().max()) < 2e-2


@pytest.mark.parametrize("C,shp,n", [(128, (2, 256, 304), 1), (128, (1, 128, 152), 2), (256, (2, 64, 76), 1), (128, (1, 9, 7), 2), (256, (1, 2, 2), 1)])
def test_fan_only_direct_form_with_avgpool(hip, C, shp, n):
    """pool2: nn.AvgPool2d(2) (floor) folded into the tile load of the fan-out-only direct form == the K5 pool2 launch bit for bit (same mean, same rounding
    of the mean, same k16 order and epilogue), and == F.linear(F.avg_pool2d(x)) on the rounded mean; odd sizes drop the last row / column."""
    dtype = torch.float16
    g = torch.Generator(device="cuda").manual_seed(C + shp[1])
    N, H, W = shp
    wide = (torch.randn(N, H, W, C + 8, device="cuda", generator=g) * 1.5 + 0.3).to(dtype)
    x = wide[..., :C]                                               # strided pixels
    wq = (torch.randn(n * C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).to(dtype)
    wp = pack.pack_conv(wq, dtype)
    bp = pack.pack_bias(torch.randn(n * C, device="cuda", generator=g) * 0.3, n * C)
    y = hip.mlp_fan(x, pack.chain_frag(wp), bp, None, frag=True, pool2=True)
    assert tuple(y.shape) == (N, H // 2, W // 2, n * C)
    ref = hip.conv2d([x.contiguous()], wp, bp, 1, 1, n * C, pool2=True)
    assert ref.shape == y.shape and torch.equal(y, ref)            # same mean, same k16 order, same epilogue: bit for bit (tools/pool_direct_biteq.py)
    xc = x[:, :H // 2 * 2, :W // 2 * 2].float().reshape(N, H // 2, 2, W // 2, 2, C)
    mean = ((xc[:, :, 0, :, 0] + xc[:, :, 0, :, 1] + xc[:, :, 1, :, 0] + xc[:, :, 1, :, 1]) * 0.25).to(dtype)
    assert torch.equal(y, hip.mlp_fan(mean, pack.chain_frag(wp), bp, None, frag=True))     # the pooled tile == the explicit mean, bit for bit
    full = F.linear(mean.float(), wq.float().reshape(n * C, C), bp[:n * C])
    assert float((y.float() - full).abs().max()) < 2e-2
