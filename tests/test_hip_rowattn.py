"""GPU parity of K13 (s2m2_row_attn: one whole 1-D attention step of BasicAttnBlock per launch, reference attentions.py:99-161,229-250,
347-355) against (a) a plain PyTorch fp32 restatement of the step that rounds to fp16 exactly where the kernel does
(tests/fake_hip.py: row_attn_reference) and (b) the launch triple it replaces -- K9 fan-out Q | K | V, K4 attention, K9 chain -- on the same
tensors.

Tolerances (fp16 I/O, values of magnitude O(1 - 4)): an fp16 ulp at 4 is 3.9e-3; six GEMMs, a softmax and two LayerNorms whose intermediates are
rounded to fp16 feed each other, so a last-bit difference of one intermediate (fp32 summation order inside the MFMAs) moves a few outputs by
an ulp or two: 99.5 % of the elements within 4e-3 * max(1, |ref|), none beyond 3e-2 -- against the triple, whose rounding points are the
same, the same bounds."""
import math

import pytest
import torch

from fake_hip import row_attn_reference
from s2m2_amd import pack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


def _layers(C, seed, with_bias=True, ln_affine=None):
    """-> (plain stacked Q | K | V weight, its bias, plain proj / ffn.0 / ffn.2, b0, b2, the entry point's packed weights and vectors)"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = (torch.randn(3 * C, C, device="cuda", generator=g) / math.sqrt(C)).half().contiguous()        # stacked Q | K | V (plain packing)
    bqkv = torch.zeros(3 * C, device="cuda")
    bqkv[2 * C:] = torch.randn(C, device="cuda", generator=g) * 0.3                                      # v has a bias, q / k none (attentions.py:24-28)
    rest = [(torch.randn(C, C, device="cuda", generator=g) / math.sqrt(C)).half().contiguous() for _ in range(3)]
    b0 = torch.randn(C, device="cuda", generator=g) * 0.3 if with_bias else None
    b2 = torch.randn(C, device="cuda", generator=g) * 0.3 if with_bias else None
    ws = qkv.float().sum(1)
    weights = pack.rowattn_pack(torch.cat([qkv] + rest, 0))
    vectors = pack.rowattn_vectors((ws[:C], ws[C:2 * C], ws[2 * C:], rest[1].float().sum(1)),
                                   (None, None, bqkv[2 * C:], None, b0, b2), ln_affine)
    return qkv, bqkv, rest, b0, b2, weights, vectors


def _tokens(nimg, h, w, C, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    base = torch.randn(nimg, h, w, C, device="cuda", generator=g)
    # correlated left / right rows with a distinct best match per token: peaked attention rows next to flat ones
    base[nimg // 2:] = 0.7 * base[: nimg - nimg // 2].roll(3, 2) + 0.3 * base[nimg // 2:]
    return (1.5 * base + 0.2).half().contiguous()


def _close(a, b, what):
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all(), what
    e = (a - b).abs()
    lim = 4e-3 * b.abs().clamp(min=1.0)
    frac = float((e > lim).float().mean())
    assert frac <= 5e-3 and float(e.max()) <= 3e-2, (what, frac, float(e.max()))


CASES = [  # nimg, h, w, heads, cross, ln_out, bias
    (2, 8, 304, 1, True, False, True),          # the 1/4 level of 1216 x 1024: two key chunks, ten waves, a half-empty last tile
    (2, 8, 304, 1, False, True, True),
    (2, 16, 160, 1, True, True, False),         # 640 x 480: one chunk, five waves, two projection items per wave
    (2, 8, 152, 2, True, False, True),          # the 1/8 level: two heads of 64
    (2, 8, 152, 2, False, False, True),
    (4, 3, 300, 1, True, False, True),          # two pairs per launch; ragged width (not a multiple of 8 tokens per tile)
    (2, 5, 76, 2, True, True, True),
    (2, 4, 40, 1, False, False, False),
    (2, 2, 320, 2, True, False, True),          # the widest row: two full chunks
    (6, 2, 24, 1, True, False, True),           # a single wave per block, three pairs
    (2, 24, 200, 1, True, True, True),          # h a multiple of 8: the XCD placement permutes the blocks
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"n{c[0]}-h{c[1]}-w{c[2]}-H{c[3]}-{'x' if c[4] else 's'}{'-ln' if c[5] else ''}")
def test_row_attn_vs_torch(hip, case):
    nimg, h, w, heads, cross, want_ln, with_bias = case
    C = 128
    assert hip.row_attn_supported(C, heads, w, torch.float16)
    gam = torch.randn(C, device="cuda") * 0.2 + 1.0
    bet = torch.randn(C, device="cuda") * 0.2
    *_, weights, vectors = _layers(C, 1 + w, with_bias, (gam, bet) if want_ln else None)
    x = _tokens(nimg, h, w, C, 7 + h)
    eps = 1e-5 if want_ln else None
    got = hip.row_attn(x, heads, cross, weights, vectors, ln_out_eps=eps)
    ref = row_attn_reference(x, heads, cross, weights, vectors, ln_out_eps=eps, rounding=torch.float16)
    if want_ln:
        _close(got[0], ref[0], "out")
        # the second output is the LayerNorm of the rows the kernel STORED
        ln_ref = torch.nn.functional.layer_norm(got[0].float(), (C,), gam, bet, 1e-5)
        assert float((got[1].float() - ln_ref).abs().max()) <= 8e-3
    else:
        _close(got, ref, "out")
    # deterministic, and the placement hint changes the block order only
    again = hip.row_attn(x, heads, cross, weights, vectors, ln_out_eps=eps, xcd_hint=False)
    a0, b0 = (got[0], again[0]) if want_ln else (got, again)
    assert torch.equal(a0, b0)


@pytest.mark.parametrize("shape", [(2, 8, 304, 1), (2, 6, 152, 2), (2, 16, 160, 1)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("cross", [True, False], ids=["cross", "self"])
def test_row_attn_vs_the_launch_triple(hip, shape, cross):
    """the same step as K9 fan-out (pre-LN folded Q | K | V) -> K4 -> K9 chain (proj + residual, pre-LN FFN, carry): identical rounding points"""
    nimg, h, w, heads = shape
    C = 128
    qkv, bqkv, rest, b0, b2, weights, vectors = _layers(C, 11 + w)
    x = _tokens(nimg, h, w, C, 3)
    got = hip.row_attn(x, heads, cross, weights, vectors)
    f3 = hip.mlp_fan(x, pack.chain_frag(qkv), bqkv, qkv.float().sum(1).contiguous())
    v3 = f3.reshape(nimg * h, w, 3 * C)
    o = hip.attention(v3[..., :C], v3[..., C:2 * C], v3[..., 2 * C:], heads, swap_halves=cross).reshape(nimg, h, w, C)
    st = [(pack.chain_frag(rest[0]), None, hip.ACT_NONE, None), (pack.chain_frag(rest[1]), b0, hip.ACT_GELU, rest[1].float().sum(1).contiguous()),
          (pack.chain_frag(rest[2]), b2, hip.ACT_NONE, None)]
    ref = hip.mlp_chain(o, st, res=x, res_stage=0, carry=True, frag=True)
    _close(got, ref, "vs triple")


def test_row_attn_rejects_what_it_does_not_take(hip):
    *_, weights, vectors = _layers(128, 5)
    x = _tokens(2, 2, 336, 128, 1)
    assert not hip.row_attn_supported(128, 1, 336, torch.float16) and not hip.row_attn_supported(256, 4, 76, torch.float16)
    with pytest.raises(RuntimeError, match="row_attn"):
        hip.row_attn(x, 1, True, weights, vectors)
    with pytest.raises(RuntimeError, match="even number"):
        hip.row_attn(_tokens(3, 2, 64, 128, 1), 1, True, weights, vectors)


def test_forward_with_and_without_row_fusion(monkeypatch):
    """S model, fp16: the whole forward with the 1-D attention steps as K13 launches against the same forward on the launch triples.  Judged
    where the two paths part and meet again -- feature_tr_4x, the transformer's output tokens: both paths round the same intermediates to
    fp16, so the tokens agree to a few fp16 ulps; the final maps are compared loosely (the randomly initialised refiners amplify last-bit
    differences of the tokens, as they do between any two fp16 runs: tests/test_fp16_headline.py has the yardsticks)."""
    import parity_util as PU
    from s2m2_amd.weights import seeded_state_dict, synthetic_pair
    l, r = synthetic_pair(128, 192, 1, 16, 3)
    sd = seeded_state_dict(128, 1, 1, 0)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("S2M2_ROWFUSE", flag)
        res[flag] = PU.hip_forward(sd, 128, 1, 2, l, r, True)
    t1, t0 = res["1"][1]["feature_tr_4x"].float(), res["0"][1]["feature_tr_4x"].float()
    e = (t1 - t0).abs()
    lim = 8e-3 * t0.abs().clamp(min=1.0)              # 12 fused steps deep (two transformer levels x enc / dec x cross / self): a few ulps
    assert float((e > lim).float().mean()) <= 2e-2 and float(e.max()) <= 0.25, (float((e > lim).float().mean()), float(e.max()))
    d = (res["1"][0][0] - res["0"][0][0]).abs()
    assert float(d.median()) < 0.1, float(d.median())
