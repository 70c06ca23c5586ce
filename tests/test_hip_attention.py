"""GPU parity of K4 (flash attention on MFMA, s2m2_attention through the C ABI) against explicit PyTorch fp32 attention
(softmax(QK^T*scale)V and the reference's positional-encoding einsum, attentions.py:42-48) on the operand values the kernel
sees.  fp32 mode: 2e-5 (exact-fp32 MFMA + online softmax vs a dense softmax); fp16 mode: P and V rounded to fp16 for the PV
product and the output rounded to fp16 -> 4e-3 * max|V|."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


def _ref(q, k, v, heads, swap):
    nb, N, C = q.shape
    d = C // heads
    sp = lambda t: t.float().reshape(t.shape[0], t.shape[1], heads, d).transpose(1, 2)
    qh, kh, vh = sp(q), sp(k), sp(v)
    if swap:
        kh, vh = kh.roll(nb // 2, 0), vh.roll(nb // 2, 0)
    a = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1)
    return (a @ vh).transpose(1, 2).reshape(nb, N, C), a


CASES = [  # nb, heads, N, D, swap
    (6, 1, 304, 128, False), (4, 1, 304, 128, True), (4, 2, 152, 64, True), (4, 4, 76, 64, False), (2, 8, 300, 32, False),
    (2, 8, 1216, 32, True), (1, 8, 300, 16, False), (3, 1, 8, 128, False), (2, 1, 33, 64, True), (2, 2, 95, 48, False),
    (2, 1, 70, 256, True), (2, 2, 65, 96, False), (1, 8, 100, 24, False),
    # the wide heads of the M / L / XL models' 1/4 level (two-sweep softmax, accumulators in AGPRs; d = 384 fp16: queries in LDS)
    (2, 1, 304, 192, False), (2, 1, 70, 192, True), (2, 1, 304, 256, False), (2, 1, 608, 384, False), (4, 1, 100, 384, True), (1, 1, 8, 384, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", CASES)
def test_attention_vs_torch(hip, case, dtype):
    nb, heads, N, D, swap = case
    g = torch.Generator(device="cuda").manual_seed(CASES.index(case))
    C = heads * D
    qkv = torch.randn(nb, N, 3 * C, device="cuda", generator=g)
    qkv[..., :2 * C] *= 1.5                                     # sharper softmax than iid unit scores
    qkv = qkv.to(dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]            # strided views of a fused QKV buffer
    out = hip.attention(q, k, v, heads, swap_halves=swap)
    ref, _ = _ref(q, k, v, heads, swap)
    err = float((out.float() - ref).abs().max())
    assert err < (3e-5 if dtype == torch.float32 else 4e-3 * float(v.float().abs().max())), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", [(2, 1, 304, 128), (2, 1, 304, 192), (2, 1, 304, 256), (2, 1, 608, 384), (2, 8, 1216, 32)])
def test_attention_row_maximum_late_in_the_key_sequence(hip, case, dtype):
    """the dominant keys of every query sit in the LAST stage (and a second group in the middle): the running maximum of the online form moves
    late, the two-sweep form of the wide heads must land on the same softmax"""
    nb, heads, N, D = case
    g = torch.Generator(device="cuda").manual_seed(N + D)
    C = heads * D
    q = torch.randn(nb, N, C, device="cuda", generator=g)
    k = torch.randn(nb, N, C, device="cuda", generator=g) * 0.5
    v = torch.randn(nb, N, C, device="cuda", generator=g)
    k[:, N // 2: N // 2 + 3] *= 3.0
    k[:, -5:] *= 6.0
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    out = hip.attention(q, k, v, heads)
    ref, _ = _ref(q, k, v, heads, False)
    err = float((out.float() - ref).abs().max())
    assert err < (5e-5 if dtype == torch.float32 else 4e-3 * float(v.float().abs().max())), err


def _tables(h, w, device):
    from s2m2_amd.engine import pe_tables
    return pe_tables(h, w, device)


# (gh, gw, D, nb): the 1/32 token grids of BASELINE's configs -- S 640x480 (15x20, d 32), S / L 1216x1024 (32x38, d 32 / 64), XL 2432x2048
# (64x76, d 96: the (3, 2) bin tiles) -- plus ragged grids, a grid taller than 32 rows with >= 32 (batch, head) pairs (no key split, the
# waves-per-block clamp of the PE variant) and the largest instantiated grid (96 x 96 cells, (3, 3) tiles)
PE_CASES = [(5, 8, 32, 2), (15, 20, 32, 2), (3, 11, 32, 2), (32, 38, 32, 2), (32, 38, 64, 2), (32, 38, 96, 1), (64, 76, 96, 1), (64, 76, 64, 1),
            (63, 64, 32, 4), (40, 33, 48, 2), (96, 96, 32, 1)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", PE_CASES)
def test_attention_with_positional_encoding(hip, dtype, case):
    gh, gw, D, nb = case
    N, heads = gh * gw, 8
    ok, why = hip.attention_supported(nb, heads, N, D, dtype, grid=(gw, gh))
    assert ok, why
    g = torch.Generator(device="cuda").manual_seed(gh)
    C = heads * D
    qkv = (torch.randn(nb, N, 3 * C, device="cuda", generator=g) * 1.3).to(dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    px, py = _tables(gh, gw, "cuda")
    out, pe_sum = hip.attention(q, k, v, heads, pe=(px, py, gw, gh))
    ref, a = _ref(q, k, v, heads, False)
    ys, xs = torch.meshgrid(torch.arange(gh, device="cuda"), torch.arange(gw, device="cuda"), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    pref = torch.empty(nb, heads, N, 32, device="cuda")
    for i0 in range(0, N, 512):                                  # the reference's einsum 'nij,ijc->nic' (attentions.py:47), in query blocks
        sl = slice(i0, min(i0 + 512, N))
        pe = 0.5 * torch.cat([px[xs[sl, None] - xs[None, :] + gw - 1], py[ys[sl, None] - ys[None, :] + gh - 1]], 2)   # (nq,N,32)
        pref[:, :, sl] = torch.einsum("bhij,ijc->bhic", a[:, :, sl], pe)
    pref = pref.transpose(1, 2).reshape(nb, N, heads * 32)
    tol = 3e-5 if dtype == torch.float32 else 4e-3
    assert float((out.float() - ref).abs().max()) < tol * max(1.0, float(v.float().abs().max()))
    assert float((pe_sum.float() - pref).abs().max()) < (3e-5 if dtype == torch.float32 else 2e-3)


def test_attention_supported_matches_the_launch(hip):
    """s2m2_attention_supported plans with the launch's own code: what it rejects the launch rejects with the same message."""
    ok, why = hip.attention_supported(2, 8, 100 * 50, 32, torch.float16, grid=(100, 50))
    assert not ok and "96 x 96" in why
    q = torch.zeros(2, 100 * 50, 3 * 256, device="cuda", dtype=torch.float16)
    from s2m2_amd.engine import pe_tables
    px, py = pe_tables(50, 100, "cuda")
    with pytest.raises(RuntimeError, match="96 x 96"):
        hip.attention(q[..., :256], q[..., 256:512], q[..., 512:], 8, pe=(px, py, 100, 50))
