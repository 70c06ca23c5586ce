"""GPU parity of K1 (LayerNorm+correlation), K2 (Sinkhorn+argmax+regression), K3 (cost-volume lookup) -- through the
C ABI -- against (a) golden outputs of the reference, (b) the CPU oracle on seeded inputs, (c) size-independent
properties at the BASELINE sizes.  Integer argmax: bit exact wherever the reference's own top-2 relative gap
exceeds 1e-4 and on exact ties (first max wins); fp32 tolerances are written next to each assert."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


def _nhwc(feat_nchw, dtype):
    return feat_nchw.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)


@pytest.mark.parametrize("name", ["op_dispinit_pos", "op_dispinit_neg"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_ln_corr_vs_reference_golden(hip, name, dtype):
    g = load_golden(name + ".npz")
    feat = _nhwc(T(g["feat"]), dtype)
    cv = hip.ln_corr(feat, T(g["gamma"]).cuda(), T(g["beta"]).cuda()).float().cpu()
    ref = T(g["cv"])
    err = float((cv - ref).abs().max())
    # fp32: exact fp32 MFMA chain vs MKL sgemm ordering, |cv| ~ 1e2;  fp16: operands + output rounded to fp16 (ulp 0.06-0.125 at 128-256)
    assert err < (3e-4 if dtype == torch.float32 else 0.35), err


@pytest.mark.parametrize("name", ["op_dispinit_pos", "op_dispinit_neg", "e2e_S_64x96_pos_r2", "e2e_S_96x160_neg_r1_b2"])
def test_sinkhorn_regress_vs_reference_golden(hip, name):
    g = load_golden(name + ".npz")
    pos = bool(g["cfg"][4] if name.startswith("op_") else g["cfg"][5])
    cv = T(g["cv"]).cuda()
    disp, conf, occ, am = hip.sinkhorn_regress(cv, pos, 3, want_argmax=True)
    disp, conf, occ, am = disp.cpu(), conf.cpu(), occ.cpu(), am.cpu()
    rd, rc, ro = (T(g[k]) for k in (("disp", "conf", "occ") if name.startswith("op_") else ("disp0", "conf0", "occ0")))
    top1, top2 = g["top2"][..., 0], g["top2"][..., 1]
    # (an exact tie top1 == top2 in the reference's fp32 plan is NOT "sure": op_dispinit_pos holds one between two columns whose scores differ by
    # 1.4 -- a coincidence of its rounding, which any other summation order resolves either way; ties between identical scores are pinned by
    # test_sinkhorn_first_maximum_wins_on_exact_ties)
    sure = T((top1 - top2) > 1e-4 * top1)
    same = am == T(g["argmax"])
    assert bool(same[sure].all()), f"{int((~same[sure]).sum())} argmax mismatches on well separated pixels"
    assert float(sure.float().mean()) > 0.97
    # conf/disp are functions of the argmax: compare where it agrees (the rest are the reference's own near ties)
    assert float(((conf - rc).abs()[:, 0][same]).max()) < 5e-5
    assert float((occ - ro).abs().max()) < 5e-5
    assert float(((disp - rd).abs()[:, 0][same]).max()) < 2e-4
    assert float(same.float().mean()) > 0.995


@pytest.mark.parametrize("out_dtype", [torch.float32])
@pytest.mark.parametrize("channels_last", [False, True])
def test_lookup_vs_reference_golden(hip, out_dtype, channels_last):
    g = load_golden("op_lookup.npz")
    c1, c2 = hip.cv_lookup(T(g["cv"]).cuda(), T(g["disp"]).cuda(), 4, channels_last, out_dtype)
    if channels_last:
        c1, c2 = c1.permute(0, 3, 1, 2), c2.permute(0, 3, 1, 2)
    # identical inputs; the grid_sample coordinate round trip is reproduced -> one ulp of |cv| ~ 1e2
    assert float((c1.cpu() - T(g["corr1"])).abs().max()) < 4e-5
    assert float((c2.cpu() - T(g["corr2"])).abs().max()) < 4e-5


@pytest.mark.parametrize("C,h,w,B", [(128, 3, 304, 1), (64, 2, 40, 2), (192, 2, 72, 1), (256, 2, 152, 1), (384, 1, 608, 1), (128, 2, 8, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_ln_corr_vs_oracle_shapes(hip, C, h, w, B, dtype):
    """Ragged strips/chunks (w not a multiple of 32/64/128), all supported channel counts, smallest legal width."""
    from oracle import s2m2_oracle as O
    g = torch.Generator().manual_seed(C + w)
    feat = torch.randn(2 * B, C, h, w, generator=g) * 1.5 + 0.2
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.05 * torch.randn(C, generator=g)
    if dtype == torch.float16:
        feat = feat.half().float()
    ref = O.ln_corr(feat, gamma, beta)
    cv = hip.ln_corr(_nhwc(feat, dtype), gamma.cuda(), beta.cuda()).float().cpu()
    scale = float(ref.abs().max())
    err = float((cv - ref).abs().max())
    assert err < (1e-5 * scale + 1e-4 if dtype == torch.float32 else 4e-3 * scale + 0.05), (err, scale)


def test_ln_corr_swap_is_exact_transpose(hip):
    """Size-independent property at the BASELINE size (S, 1216x1024: h=256, w=304, C=128): exchanging left and
    right images transposes every row matrix bit-exactly (the k-order of the MFMA chain is symmetric)."""
    g = torch.Generator().manual_seed(7)
    feat = torch.randn(2, 256, 304, 128, generator=g).cuda().half()
    gamma = (1 + 0.1 * torch.randn(128, generator=g)).cuda()
    beta = (0.05 * torch.randn(128, generator=g)).cuda()
    a = hip.ln_corr(feat, gamma, beta)
    b = hip.ln_corr(feat.flip(0).contiguous(), gamma, beta)
    assert torch.equal(a, b.transpose(2, 3))
    # against fp32 math on the same fp16 inputs
    f = torch.nn.functional.layer_norm(feat.float(), (128,), gamma, beta)
    ref = torch.matmul(f[:1], f[1:].transpose(-1, -2))
    assert float((a.float() - ref).abs().max()) < 0.3


@pytest.mark.parametrize("pos", [True, False])
def test_sinkhorn_full_size_vs_oracle_and_marginals(hip, pos):
    """h=8 rows of the 1216-wide case (w=304) against the oracle, plus the transport-plan property: every
    regular row of P has mass <= 1 (the rest went to the dustbin) and mass >= 0."""
    from oracle import s2m2_oracle as O
    g = torch.Generator().manual_seed(11)
    w = 304
    cv = torch.randn(1, 8, w, w, generator=g) * 3 + 110
    idx = torch.randint(0, w, (8, w), generator=g)
    cv[0, torch.arange(8)[:, None], torch.arange(w)[None, :], idx] += 12          # one strong match per left pixel
    P = O.sinkhorn_prob(cv, pos)
    rd, rc, ro, rind = O.regress(P)
    disp, conf, occ, am = hip.sinkhorn_regress(cv.cuda(), pos, 3, want_argmax=True)
    top = P.topk(2, 3).values
    sure = (top[..., 0] - top[..., 1]) > 1e-4 * top[..., 0]
    same = am.cpu() == rind.int()
    assert bool(same[sure].all())
    assert float((conf.cpu() - rc).abs()[:, 0][same].max()) < 5e-5
    assert float((occ.cpu() - ro).abs().max()) < 5e-5
    assert float((disp.cpu() - rd).abs()[:, 0][same].max()) < 2e-4
    assert float(occ.max()) <= 1 + 1e-5 and float(occ.min()) >= 0


def test_sinkhorn_fp16_input_matches_fp32_on_same_values(hip):
    g = torch.Generator().manual_seed(5)
    cv = (torch.randn(2, 4, 160, 160, generator=g) * 4 + 100).half()
    a = hip.sinkhorn_regress(cv.cuda(), True, 3, want_argmax=True)
    b = hip.sinkhorn_regress(cv.float().cuda(), True, 3, want_argmax=True)
    # same values, different column->lane mapping (8 fp16 vs 4 fp32 per 16-byte piece): identical up to summation order
    assert float((a[3] == b[3]).float().mean()) > 0.999
    same = a[3] == b[3]
    assert float((a[0] - b[0]).abs()[:, 0][same].max()) < 1e-4
    assert float((a[1] - b[1]).abs()[:, 0][same].max()) < 1e-5 and float((a[2] - b[2]).abs().max()) < 1e-5


def test_lookup_full_size_vs_oracle(hip):
    from oracle import s2m2_oracle as O
    g = torch.Generator().manual_seed(13)
    B, h, w = 1, 6, 304
    cv = torch.randn(B, h, w, w, generator=g) * 10 + 100
    disp = torch.rand(B, 1, h, w, generator=g) * 200 - 10
    r1, r2 = O.cv_lookup(cv, disp)
    c1, c2 = hip.cv_lookup(cv.cuda(), disp.cuda(), 4)
    assert float((c1.cpu() - r1).abs().max()) < 4e-5
    assert float((c2.cpu() - r2).abs().max()) < 4e-5
    # fp16 volume / fp16 output
    c1h, _ = hip.cv_lookup(cv.half().cuda(), disp.cuda(), 4, True, torch.float16)
    assert float((c1h.float().cpu().permute(0, 3, 1, 2) - r1).abs().max()) < 0.2


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(1, 6, 304, 128), (2, 3, 160, 128), (1, 2, 608, 64), (1, 4, 304, 256)])
def test_ln_corr_banded_equals_full_inside_the_band(hip, shape, dtype):
    """s2m2_ln_corr_banded (use_positivity models): bit-identical to the full volume for j <= i + band, nothing written in tiles that
    lie entirely beyond it (lookups and the masked Sinkhorn never read j > i + 11)."""
    B, h, w, C = shape
    g = torch.Generator(device="cuda").manual_seed(w + C)
    feat = (torch.randn(2 * B, h, w, C, device="cuda", generator=g) * 1.3 + 0.1).to(dtype)
    gam = 1 + 0.1 * torch.randn(C, device="cuda", generator=g)
    bet = 0.05 * torch.randn(C, device="cuda", generator=g)
    full = hip.ln_corr(feat, gam, bet)
    sentinel = -777.0
    out = torch.full((B, h, w, w), sentinel, device="cuda", dtype=dtype)
    hip.ln_corr(feat, gam, bet, out=out, band=11)
    torch.cuda.synchronize()
    i = torch.arange(w, device="cuda")[:, None]
    j = torch.arange(w, device="cuda")[None, :]
    inside = (j <= i + 11).expand(B, h, w, w)
    assert torch.equal(out[inside], full[inside])
    written = out != sentinel
    # store granule: 64 columns (counted from the chunk's first column) x the 32 rows of a wave -> nothing is written 64 or more
    # columns beyond the last column (i | 31) + 11 that the wave's rows need
    limit = (i | 31) + 11 + 64
    assert not bool((written & (j >= limit).expand(B, h, w, w)).any())
    if w >= 256:
        assert float(written.float().mean()) < 0.75          # a real saving on wide rows


def test_forward_with_banded_cost_volume_is_bit_identical(monkeypatch):
    from s2m2_amd.model import S2M2
    from s2m2_amd.weights import seeded_state_dict, synthetic_pair
    sd = seeded_state_dict(128, 1, 1, 0)
    l, r = synthetic_pair(128, 640, 1, 24, 3)
    l, r = l.cuda(), r.cuda()
    outs = []
    for band in ("0", "1"):
        monkeypatch.setenv("S2M2_CV_BAND", band)
        m = S2M2(128, 1, 1, use_positivity=True, refine_iter=2)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        with torch.autocast("cuda", dtype=torch.float16):
            m(l, r)
            outs.append([t.clone() for t in m(l, r)])          # second call: graph replay, cv buffer reused
        assert (m.engine(torch.float16).cv_band == 11) == (band == "1")
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape,dtype,pos", [((1, 24, 160, 160), torch.float16, True), ((1, 8, 304, 304), torch.float16, False),
                                             ((2, 6, 72, 72), torch.float32, True)])
def test_sinkhorn_is_run_to_run_deterministic(hip, shape, dtype, pos):
    """K2 keeps u, v, the column partials and the fallback flags in LDS across passes: 40 runs on the same volume must agree bit for
    bit (tools/k2_determinism.py runs the longer version)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    cv = (torch.randn(*shape, device="cuda", generator=g) * 8).to(dtype)
    ref = [t.clone() for t in hip.sinkhorn_regress(cv, pos, 3, want_argmax=True)]
    for _ in range(40):
        out = hip.sinkhorn_regress(cv, pos, 3, want_argmax=True)
        assert all(torch.equal(a, b) for a, b in zip(ref, out))


# ---- K2 beyond w = 384: the 32- and 64-lane row groups (sinkhorn.hip dispatch: GL = 16 / 32 / 64 for w <= 384 / 768 / above) ----------
def _wide_case(w, h, pos, dtype, seed, spread=3.0, peak=12.0):
    g = torch.Generator().manual_seed(seed)
    cv = torch.randn(1, h, w, w, generator=g) * spread + 110
    idx = torch.randint(0, w, (h, w), generator=g)
    if pos:
        idx = torch.minimum(idx, torch.arange(w)[None, :].expand(h, w))              # the match of left pixel i lies at j <= i
    cv[0, torch.arange(h)[:, None], torch.arange(w)[None, :], idx] += peak            # one strong match per left pixel
    if dtype == torch.float16:
        cv = cv.half().float()
    return cv


@pytest.mark.parametrize("w,h", [(400, 3), (608, 4), (768, 2), (776, 2), (800, 2), (1200, 2), (392, 5)])
@pytest.mark.parametrize("pos", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_sinkhorn_wide_rows_vs_oracle(hip, w, h, pos, dtype):
    """submodules.py:147-152,169-241 at the widths of BASELINE configs[4] (XL 2432x2048: w = 608) and on both sides of every
    dispatch boundary of K2 (384 | 392..768 | 776..1200)."""
    from oracle import s2m2_oracle as O
    cv = _wide_case(w, h, pos, dtype, seed=w + 17 * pos)
    P = O.sinkhorn_prob(cv, pos)
    rd, rc, ro, rind = O.regress(P)
    disp, conf, occ, am = (t.cpu() for t in hip.sinkhorn_regress(cv.to("cuda", dtype), pos, 3, want_argmax=True))
    top = P.topk(2, 3).values
    sure = (top[..., 0] - top[..., 1]) > 1e-4 * top[..., 0]
    same = am == rind.int()
    assert float(sure.float().mean()) > 0.9
    assert bool(same[sure].all()), f"{int((~same[sure]).sum())} argmax mismatches on well separated pixels"
    assert float((conf - rc).abs()[:, 0][same].max()) < 5e-5
    assert float((occ - ro).abs().max()) < 5e-5
    # disp = i - corr: |corr| reaches w, fp32 ulp at 1200 is 1.2e-4
    assert float((disp - rd).abs()[:, 0][same].max()) < 2e-4 + 2e-7 * w
    assert float(occ.max()) <= 1 + 1e-5 and float(occ.min()) >= 0


@pytest.mark.parametrize("w", [304, 608, 1200])
def test_sinkhorn_first_maximum_wins_on_exact_ties(hip, w):
    """argmax semantics of submodules.py:226 (first maximal index): duplicated columns have identical potentials, so P ties EXACTLY in
    the oracle and must in K2 (without the positivity mask; with it the two columns see different masks and cannot tie)."""
    from oracle import s2m2_oracle as O
    cv, rows, cols = _tie_case(w)
    P = O.sinkhorn_prob(cv, False)
    rind = O.regress(P)[3]
    top = P.topk(2, 3).values
    tied = top[..., 0] == top[..., 1]
    assert bool(tied[:, :, rows].all()) and bool((rind[0, :, rows] == cols).all()), "the construction must tie exactly in the oracle"
    am = hip.sinkhorn_regress(cv.cuda(), False, 3, want_argmax=True)[3].cpu()
    assert bool((am[0, :, rows] == cols.int()).all())
    assert bool((am == rind.int())[tied | ((top[..., 0] - top[..., 1]) > 1e-4 * top[..., 0])].all())


def _tie_case(w, h=2, seed=3):
    """8 column pairs (j, j+5) that are identical in EVERY row of the volume (so their potentials are identical and P ties exactly
    wherever the row maximum sits on the pair); each pair is the strong match of one left pixel."""
    g = torch.Generator().manual_seed(seed + w)
    cv = torch.randn(1, h, w, w, generator=g) * 3 + 110
    cols = torch.arange(8) * (w // 9) + 11
    rows = (cols * 7 + 3) % w
    cv[:, :, rows, cols] = 135.0
    cv[..., cols + 5] = cv[..., cols]
    return cv, rows, cols


@pytest.mark.parametrize("shape,dtype,pos", [((1, 6, 608, 608), torch.float16, False), ((1, 3, 800, 800), torch.float32, True)])
def test_sinkhorn_wide_rows_run_to_run_deterministic(hip, shape, dtype, pos):
    g = torch.Generator(device="cuda").manual_seed(9)
    cv = (torch.randn(*shape, device="cuda", generator=g) * 8).to(dtype)
    ref = [t.clone() for t in hip.sinkhorn_regress(cv, pos, 3, want_argmax=True)]
    assert all(bool(torch.isfinite(t.float()).all()) for t in ref)
    for _ in range(40):
        out = hip.sinkhorn_regress(cv, pos, 3, want_argmax=True)
        assert all(torch.equal(a, b) for a, b in zip(ref, out))


def test_sinkhorn_fp16_wide_input_matches_fp32_on_same_values(hip):
    g = torch.Generator().manual_seed(6)
    cv = (torch.randn(1, 4, 608, 608, generator=g) * 4 + 100).half()
    a = hip.sinkhorn_regress(cv.cuda(), False, 3, want_argmax=True)
    b = hip.sinkhorn_regress(cv.float().cuda(), False, 3, want_argmax=True)
    same = a[3] == b[3]
    assert float(same.float().mean()) > 0.999
    assert float((a[0] - b[0]).abs()[:, 0][same].max()) < 3e-4
    assert float((a[1] - b[1]).abs()[:, 0][same].max()) < 1e-5 and float((a[2] - b[2]).abs().max()) < 1e-5


# ---- K1 without the LayerNorm (s2m2_corr: tokens normalised by the launch that produced them) and row-padded cost volumes ----------------
@pytest.mark.parametrize("C,h,w,B", [(128, 3, 304, 1), (128, 2, 160, 2), (256, 2, 304, 1), (64, 2, 40, 1), (128, 2, 8, 1), (384, 1, 608, 1), (192, 2, 72, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_corr_on_normalised_tokens_equals_ln_corr(hip, C, h, w, B, dtype):
    """s2m2_corr(LayerNorm(x)) against s2m2_ln_corr(x): the same kernel body minus statistics and affine.  The tokens are normalised
    here with the oracle's arithmetic order in fp32 and rounded like K1 rounds its MFMA operands; in fp16 the two volumes agree to the
    rounding of the last bit of a token (a different but equally valid fp32 summation order inside the LayerNorm)."""
    from oracle import s2m2_oracle as O
    g = torch.Generator().manual_seed(C + w)
    feat = torch.randn(2 * B, C, h, w, generator=g) * 1.5 + 0.2
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.05 * torch.randn(C, generator=g)
    if dtype == torch.float16:
        feat = feat.half().float()
    ref = O.ln_corr(feat, gamma, beta)
    tok = torch.nn.functional.layer_norm(feat.permute(0, 2, 3, 1), (C,), gamma, beta, 1e-5).to(dtype).cuda().contiguous()
    cv = hip.corr(tok)
    assert cv.shape == (B, h, w, w) and cv.stride(2) % (64 if dtype == torch.float16 else 32) == 0 and cv.stride(2) >= w
    scale = float(ref.abs().max())
    err = float((cv.float().cpu() - ref).abs().max())
    assert err < (1e-5 * scale + 1e-4 if dtype == torch.float32 else 4e-3 * scale + 0.05), (err, scale)
    # dense output buffer and banded variant: same values
    dense = torch.empty((B, h, w, w), device="cuda", dtype=dtype)
    hip.corr(tok, out=dense)
    assert torch.equal(dense, cv)
    banded = torch.full_like(dense, -777.0)
    hip.corr(tok, out=banded, band=11)
    i = torch.arange(w, device="cuda")[:, None]
    j = torch.arange(w, device="cuda")[None, :]
    inside = (j <= i + 11).expand(B, h, w, w)
    assert torch.equal(banded[inside], cv[inside])
    # the padding columns of the row-padded allocation are never written
    if cv.stride(2) > w:
        base = torch.full((B, h, w, cv.stride(2)), 5.0, device="cuda", dtype=dtype)
        hip.corr(tok, out=base[..., :w])
        assert bool((base[..., w:] == 5.0).all()) and torch.equal(base[..., :w], cv)


@pytest.mark.parametrize("w,dtype,pos", [(304, torch.float16, True), (160, torch.float32, False), (608, torch.float16, False), (72, torch.float32, True)])
def test_sinkhorn_and_lookup_read_row_padded_volumes(hip, w, dtype, pos):
    """K2 / K3 on the row-padded view (pitch = w rounded up to 128 bytes, garbage in the padding) == on the dense volume, bit for bit."""
    g = torch.Generator(device="cuda").manual_seed(w)
    h = 5
    dense = (torch.randn(1, h, w, w, device="cuda", generator=g) * 4 + 100).to(dtype)
    padded = hip.cv_alloc(1, h, w, dtype, "cuda")
    assert padded.stride(2) > w or w % (64 if dtype == torch.float16 else 32) == 0
    padded._base.fill_(float("nan")) if padded._base is not None else None
    padded.copy_(dense)
    a = hip.sinkhorn_regress(dense, pos, 3, want_argmax=True)
    b = hip.sinkhorn_regress(padded, pos, 3, want_argmax=True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    disp = torch.rand(1, 1, h, w, device="cuda", generator=g) * 60 - 5
    c = hip.cv_lookup(dense, disp, 4)
    d = hip.cv_lookup(padded, disp, 4)
    assert torch.equal(c[0], d[0]) and torch.equal(c[1], d[1])
    buf1 = torch.zeros(1, h, w, 32, device="cuda", dtype=dtype)
    buf2 = torch.zeros_like(buf1)
    hip.cv_lookup_into(dense, disp, buf1, 0, 16, 4)
    hip.cv_lookup_into(padded, disp, buf2, 0, 16, 4)
    assert torch.equal(buf1, buf2)


def test_forward_with_folded_layernorm_matches_k1_with_its_own(monkeypatch):
    """S2M2_FUSE_K1LN (default on): DispInit's LayerNorm as the second output of the last K9 chain + the correlation
    alone on a row-padded volume, against K1 normalising the tokens itself on a dense volume.  fp32: the cost volumes agree to
    summation order, everything downstream to the parity tolerance; fp16: both are valid fp16 forwards."""
    from s2m2_amd.model import S2M2
    from s2m2_amd.weights import seeded_state_dict, synthetic_pair
    sd = seeded_state_dict(128, 1, 1, 0)
    l, r = synthetic_pair(128, 608, 1, 24, 3)                 # w = 152: 608-byte fp32 rows, padded to 640 bytes (pitch 160)
    l, r = l.cuda(), r.cuda()
    caps = []
    for fold in ("1", "0"):
        monkeypatch.setenv("S2M2_FUSE_K1LN", fold)
        m = S2M2(128, 1, 1, use_positivity=True, refine_iter=2)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        cap = {}
        out = m(l, r, capture=cap)
        eng = m.engine(torch.float32)
        assert (eng._tokens_normed is not None) == (fold == "1")
        assert cap["cv"].stride(2) > cap["cv"].shape[3]            # both forms write rows on 128-byte lines (s2m2_ln_corr_pitched)
        caps.append((cap, out, [t.clone() for t in m(l, r)], [t.clone() for t in m(l, r)]))          # eager, eager (first plain call), graph replay
    (ca, oa, ea, ga), (cb, ob, eb, gb) = caps
    assert float((ca["cv"] - cb["cv"]).abs().max()) < 2e-4 * float(cb["cv"].abs().max())
    agree = float((ca["argmax"] == cb["argmax"]).float().mean())
    assert agree > 0.999
    if agree == 1.0:
        assert float((oa[0] - ob[0]).abs().max()) < 2e-2
    for x, y in zip(ea, ga):                                  # graph replay == eager on the folded path
        assert torch.equal(x, y)
