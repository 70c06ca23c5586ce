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


def _tables(h, w, device):
    from s2m2_amd.engine import pe_tables
    return pe_tables(h, w, device)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("grid", [(5, 8), (15, 20), (3, 11)])
def test_attention_with_positional_encoding(hip, dtype, grid):
    gh, gw = grid
    N, heads, D, nb = gh * gw, 8, 32, 2
    g = torch.Generator(device="cuda").manual_seed(gh)
    C = heads * D
    qkv = (torch.randn(nb, N, 3 * C, device="cuda", generator=g) * 1.3).to(dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    px, py = _tables(gh, gw, "cuda")
    out, pe_sum = hip.attention(q, k, v, heads, pe=(px, py, gw, gh))
    ref, a = _ref(q, k, v, heads, False)
    ys, xs = torch.meshgrid(torch.arange(gh, device="cuda"), torch.arange(gw, device="cuda"), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    pe = 0.5 * torch.cat([px[xs[:, None] - xs[None, :] + gw - 1], py[ys[:, None] - ys[None, :] + gh - 1]], 2)   # (N,N,32)
    pref = torch.einsum("bhij,ijc->bhic", a, pe).transpose(1, 2).reshape(nb, N, heads * 32)
    tol = 3e-5 if dtype == torch.float32 else 4e-3
    assert float((out.float() - ref).abs().max()) < tol * max(1.0, float(v.float().abs().max()))
    assert float((pe_sum.float() - pref).abs().max()) < (3e-5 if dtype == torch.float32 else 2e-3)
