"""Host-side weight packing (s2m2_amd/pack.py) checked on the CPU against the layouts include/s2m2_hip.h documents."""
import torch

from s2m2_amd import pack


def test_pack_conv_k_order_0_and_padding():
    w = torch.arange(16 * 11 * 3 * 3, dtype=torch.float32).reshape(16, 11, 3, 3)
    p = pack.pack_conv(w, torch.float32)
    assert tuple(p.shape) == (16, 3 * 3 * 16)                                   # Cin 11 -> 16 (multiples of 8), K = (ky, kx, c)
    q = p.reshape(16, 3, 3, 16)
    assert torch.equal(q[..., :11], w.permute(0, 2, 3, 1)) and float(q[..., 11:].abs().max()) == 0
    # two sources, individually padded: channels of source 1 start at the padded end of source 0
    p2 = pack.pack_conv(w, torch.float32, [(3, 8), (8, 8)]).reshape(16, 3, 3, 16)
    assert torch.equal(p2[..., :3], w.permute(0, 2, 3, 1)[..., :3]) and torch.equal(p2[..., 8:16], w.permute(0, 2, 3, 1)[..., 3:11])
    assert float(p2[..., 3:8].abs().max()) == 0
    assert tuple(pack.pack_conv(w[:10], torch.float32).shape) == (16, 144)      # Cout 10 -> 16 zero rows
    assert float(pack.pack_bias(torch.ones(10), 10)[10:].abs().max()) == 0


def test_pack_conv_frag_is_the_documented_fragment_stream():
    """K order 2 (s2m2_conv_desc.korder): [Cout/32][chunk][tap][k16 step][lane][8], lane l = cout 32t + l % 32,
    channel 128 c + 16 s + 8 (l // 32) + e, zero beyond Cin."""
    cout, cin, kh, kw = 128, 192, 3, 1
    w = torch.randn(cout, cin, kh, kw).half()
    assert pack.frag_eligible(cout, cin, kh, kw, torch.float16)
    f = pack.pack_conv_frag(w, torch.float16)
    nchunk = 2
    assert tuple(f.shape) == (cout, kh * kw * nchunk * 128)
    s = f.reshape(cout // 32, nchunk, kh * kw, 8, 64, 8)
    for t, c, tap, st, lane, e in [(0, 0, 0, 0, 0, 0), (3, 1, 2, 3, 63, 7), (1, 0, 1, 7, 31, 5), (2, 1, 0, 4, 40, 2), (0, 1, 2, 7, 63, 7)]:
        co, ch = 32 * t + lane % 32, 128 * c + 16 * st + 8 * (lane // 32) + e
        want = w[co, ch, tap // kw, tap % kw] if ch < cin else torch.tensor(0.0).half()
        assert s[t, c, tap, st, lane, e] == want, (t, c, tap, st, lane, e)
    # every real weight appears exactly once, the rest is padding
    assert abs(float(s.float().abs().sum()) - float(w.float().abs().sum())) < 1e-2 * float(w.float().abs().sum())
    assert not pack.frag_eligible(64, 128, 3, 3, torch.float16) and not pack.frag_eligible(128, 64, 3, 3, torch.float16)
    assert not pack.frag_eligible(128, 128, 1, 1, torch.float16) and not pack.frag_eligible(128, 128, 3, 3, torch.float32)


def test_pack_conv_frag_192_channel_chunks():
    """the layers of 192-channel models (Cout % 192 == 0, not % 128, Cin % 192 == 0): chunks of 192 channels, 12 k16 steps per tap"""
    for cout, cin, kh, kw in ((192, 192, 3, 3), (192, 384, 1, 3)):
        assert pack.frag_chunk(cout, cin) == 192 and pack.frag_eligible(cout, cin, kh, kw, torch.float16)
        w = torch.randn(cout, cin, kh, kw).half()
        f = pack.pack_conv_frag(w, torch.float16, [(192, 192)] * (cin // 192))
        nchunk = cin // 192
        assert tuple(f.shape) == (cout, kh * kw * cin)
        s = f.reshape(cout // 32, nchunk, kh * kw, 12, 64, 8)
        for t, c, tap, st, lane, e in [(0, 0, 0, 0, 0, 0), (5, nchunk - 1, kh * kw - 1, 11, 63, 7), (1, 0, 1, 7, 31, 5), (2, nchunk - 1, 0, 4, 40, 2)]:
            co, ch = 32 * t + lane % 32, 192 * c + 16 * st + 8 * (lane // 32) + e
            assert s[t, c, tap, st, lane, e] == w[co, ch, tap // kw, tap % kw], (t, c, tap, st, lane, e)
        assert torch.equal(f.float().abs().sum(), w.float().abs().sum()) or abs(float(f.float().abs().sum()) - float(w.float().abs().sum())) < 1e-2 * float(w.float().abs().sum())
    # everything else keeps 128-channel chunks (384 and 256 are multiples of 128; 192 <- 128 + 64 is not a multiple of 192 channels)
    assert pack.frag_chunk(384, 384) == 128 and pack.frag_chunk(256, 192) == 128 and pack.frag_chunk(128, 192) == 128
    assert pack.frag_chunk(192, 320) == 128 and not pack.frag_eligible(192, 320, 3, 3, torch.float16)


def test_transposed_conv_packings():
    w = torch.randn(6, 10, 3, 3)                                                 # ConvTranspose2d weight (Cin, Cout, KH, KW), stride 1
    c = pack.convT_s1_as_conv(w)
    x = torch.randn(1, 6, 7, 9)
    ref = torch.nn.functional.conv_transpose2d(x, w, padding=1)
    assert float((torch.nn.functional.conv2d(x, c, padding=1) - ref).abs().max()) < 2e-6 * float(ref.abs().max()) + 1e-6   # fp32 round-off of two summation orders
    w2 = torch.randn(8, 5, 2, 2)                                                 # ConvTranspose2d(k=2, s=2)
    g, cp = pack.pack_convT_2x2s2(w2, torch.float32)
    assert cp == 8 and tuple(g.shape) == (32, 8)
    y = torch.nn.functional.conv_transpose2d(torch.randn(1, 8, 3, 4), w2, stride=2)
    assert tuple(y.shape) == (1, 5, 6, 8)
    xin = torch.randn(1, 8, 3, 4)
    y = torch.nn.functional.conv_transpose2d(xin, w2, stride=2)
    z = torch.einsum("rk,nkyx->nryx", g, xin).reshape(1, 2, 2, 8, 3, 4)          # rows ordered (dy, dx, c')
    z = z.permute(0, 3, 4, 1, 5, 2).reshape(1, 8, 6, 8)[:, :5]
    assert float((z - y).abs().max()) < 2e-6 * float(y.abs().max()) + 1e-6


def test_chain_frag_is_a_permutation():
    """pack.chain_frag: slot ((o/32) * (C/16) + q/2) * 64 + (q%2) * 32 + o%32 holds the 16-byte piece (cout o, channels 8q .. 8q+7)"""
    C = 128
    w = torch.arange(2 * C * C, dtype=torch.float32).reshape(2 * C, C)
    f = pack.chain_frag(w).reshape(2, C * C // 8, 8)
    for layer, o, q in ((0, 0, 0), (0, 33, 5), (1, 127, 15), (1, 64, 2)):
        slot = ((o // 32) * (C // 16) + q // 2) * 64 + (q % 2) * 32 + o % 32
        assert torch.equal(f[layer, slot], w[layer * C + o, 8 * q:8 * q + 8])


def test_fusion_frag_stream_order():
    """pack.fusion_frag: fragment j of cout tile t at 16-byte slot (t * 9C/16 + j) * 64 + lane; slice s: 2C/16 fragments of w1 rows sC + 32t ..,
    then C/16 fragments of w2 columns sC .. (include/s2m2_hip.h, s2m2_feature_fusion_frag)"""
    C = 128
    w1 = torch.arange(3 * C * 2 * C, dtype=torch.float32).reshape(3 * C, 2 * C)
    w2 = -torch.arange(C * 3 * C, dtype=torch.float32).reshape(C, 3 * C) - 1
    f = pack.fusion_frag(w1, w2).reshape(C // 32, 9 * C // 16, 64, 8)
    per = 3 * C // 16
    for t, s, step, lane in ((0, 0, 0, 0), (1, 0, 15, 33), (3, 2, 7, 63), (2, 1, 0, 31)):
        row, k = 32 * t + lane % 32, 16 * step + 8 * (lane // 32)
        assert torch.equal(f[t, s * per + step, lane], w1[s * C + row, k:k + 8])
    for t, s, step, lane in ((0, 0, 0, 0), (1, 1, 3, 40), (3, 2, 7, 63)):
        row, k = 32 * t + lane % 32, s * C + 16 * step + 8 * (lane // 32)
        assert torch.equal(f[t, s * per + 2 * C // 16 + step, lane], w2[row, k:k + 8])


def test_pw_frag_layout_and_padding():
    """pack.pw_frag (s2m2_pw_desc.weight_frag): 16-byte slot (t * steps + s) * 64 + l holds row 32 t + l % 32, columns 16 s + 8 (l // 32) .. + 7 of
    the (Cout, K) matrix zero-padded to (32 * tiles, 16 * steps)"""
    cout, k = 40, 72
    w = torch.arange(cout * k, dtype=torch.float32).reshape(cout, k) + 1
    f = pack.pw_frag(w)
    assert tuple(f.shape) == (2, 5, 64, 8)
    for t, s, l in ((0, 0, 0), (0, 4, 63), (1, 2, 7), (1, 4, 40), (0, 3, 32)):
        row, col = 32 * t + l % 32, 16 * s + 8 * (l // 32)
        want = torch.zeros(8)
        if row < cout:
            n = max(0, min(8, k - col))
            want[:n] = w[row, col:col + n]
        assert torch.equal(f[t, s, l], want), (t, s, l)
    assert float(f.sum()) == float(w.sum())                                      # every weight once, zeros elsewhere


def _conv_from_fragments(x, f, cout, ntap_w, cin, chunk):
    """3x3 (or kxk) convolution evaluated the way conv_narrow.hip walks pack.narrow_frag: K index of slot (tile, kstep, lane, e) =
    16 kstep + 8 (lane // 32) + e, read as (chunk, tap, channel within the chunk) for Cin >= 128 and as (tap, channel) below."""
    n, h, wd, _ = x.shape
    kk = ntap_w * ntap_w
    xp = torch.nn.functional.pad(x, (0, 0, ntap_w // 2, ntap_w // 2, ntap_w // 2, ntap_w // 2))
    out = torch.zeros(n, h, wd, cout, dtype=torch.float64)
    tiles, steps = f.shape[:2]
    for t in range(tiles):
        for s in range(steps):
            for half in range(2):
                k0 = 16 * s + 8 * half
                for e in range(8):
                    k = k0 + e
                    if k >= kk * cin:
                        continue
                    if chunk:
                        c, rem = divmod(k, kk * chunk)
                        tap, ch = divmod(rem, chunk)
                        ch += c * chunk
                    else:
                        tap, ch = divmod(k, cin)
                    ky, kx = divmod(tap, ntap_w)
                    wv = f[t, s, half * 32:half * 32 + 32, e].double()          # 32 couts of this tile
                    co = min(32, cout - 32 * t)
                    if co > 0:
                        out[..., 32 * t:32 * t + co] += xp[:, ky:ky + h, kx:kx + wd, ch, None].double() * wv[:co]
    return out


def test_narrow_frag_walked_like_the_kernel_is_the_convolution():
    """pack.narrow_frag against F.conv2d: natural K order below 128 input channels, chunk-major (64 channels) from 128 on (include/s2m2_hip.h, K12)"""
    torch.manual_seed(3)
    for cin, cout, k in ((8, 40, 3), (16, 64, 5), (48, 48, 3), (128, 8, 3), (256, 16, 3)):
        w = torch.randn(cout, cin, k, k)
        x = torch.randn(1, 5, 6, cin)
        wp = pack.pack_conv(w, torch.float32, [(cin, cin)])
        f = pack.narrow_frag(wp, k * k)
        assert tuple(f.shape) == ((wp.shape[0] + 31) // 32, (k * k * cin + 15) // 16, 64, 8)
        got = _conv_from_fragments(x, f, cout, k, cin, 64 if cin >= 128 else 0)
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=k // 2).permute(0, 2, 3, 1)
        assert float((got - ref).abs().max()) < 1e-9, (cin, cout, k)


def test_head_frag_matches_the_accumulator_layout_of_a_32x32_mfma():
    """pack.head_frag (s2m2_narrow_desc.head_frag): the 1x1 head's K order is the order in which lane (pixel, half) of the 3x3 layer's accumulator
    tile j holds its channels -- 32 j + 8 g + 4 half + e in quad g -- so quads 2p, 2p + 1 are the head's pixel fragment of k16 step (j, p).  Emulates
    D2[m][pixel] = sum_k A[m][k] B[k][pixel] with A lane (m, half) element e8 <-> k = 8 half + e8 on both operands."""
    torch.manual_seed(0)
    cout2, k = 16, 48
    w2 = torch.randn(cout2, k)
    hf = pack.head_frag(w2)
    assert tuple(hf.shape) == (1, 4, 64, 8)
    y = torch.randn(32, 64)
    y[:, k:] = 0                                                              # channels beyond Cout: relu(0 + 0)
    d2 = torch.zeros(32, 32)
    for j in range(2):
        for p in range(2):
            for half in range(2):
                for e8 in range(8):
                    q, e = divmod(e8, 4)
                    ch = 32 * j + 8 * (2 * p + q) + 4 * half + e              # what lane (pixel, half) holds in quad 2p + q, element e
                    d2 += hf[0, 2 * j + p, 32 * half:32 * half + 32, e8][:, None] * y[:, ch][None, :]
    assert float((d2[:cout2] - w2 @ y[:, :k].T).abs().max()) < 1e-5 and float(d2[cout2:].abs().max()) == 0


def test_row_attn_packing_is_an_involution():
    from s2m2_amd import pack
    w = torch.arange(4 * 32, dtype=torch.float32).reshape(4, 32)
    p = pack.rowattn_cols(w)
    assert p[0, :16].tolist() == [0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15] and torch.equal(pack.rowattn_cols(p), w)
    w2 = torch.randn(256, 128)
    pk = pack.rowattn_pack(w2)
    assert torch.equal(pack.rowattn_unpack(pk), w2)
    # unit u of row r of the second layer at element (u * 128 + r) * 8 of its block
    assert torch.equal(pk[128:].reshape(-1)[(3 * 128 + 5) * 8:(3 * 128 + 5) * 8 + 8], pack.rowattn_cols(w2)[128 + 5, 24:32])




def test_row_attn_vectors_and_reference_on_cpu():
    """pack.rowattn_vectors' order is what the CPU restatement of s2m2_row_attn (tests/fake_hip.py) and the kernel read; the restatement itself
    against a direct evaluation of the step with the plain weights (fp32, no rounding)."""
    import torch.nn.functional as F
    from fake_hip import row_attn_reference
    from s2m2_amd import pack
    g = torch.Generator().manual_seed(0)
    C = 128
    ws = [torch.randn(C, C, generator=g) / C ** 0.5 for _ in range(6)]
    bv, b0, b2 = (torch.randn(C, generator=g) * 0.3 for _ in range(3))
    vec = pack.rowattn_vectors((ws[0].sum(1), ws[1].sum(1), ws[2].sum(1), ws[4].sum(1)), (None, None, bv, None, b0, b2))
    assert tuple(vec.shape) == (12, C) and torch.equal(vec[4], bv) and torch.equal(vec[5], ws[2].sum(1)) and torch.equal(vec[8], ws[4].sum(1))
    x = torch.randn(2, 3, 20, C, generator=g)
    out = row_attn_reference(x, 2, True, pack.rowattn_pack(torch.cat(ws, 0)), vec)
    ln = lambda t: F.layer_norm(t, (C,))                                   # noqa: E731
    src = x.roll(1, 0)
    q, k, v = F.linear(ln(x), ws[0]), F.linear(ln(src), ws[1]), F.linear(ln(src), ws[2], bv)
    sp = lambda t: t.reshape(6, 20, 2, 64).transpose(1, 2)                 # noqa: E731
    o = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / 8.0, -1) @ sp(v)).transpose(1, 2).reshape(2, 3, 20, C)
    z1 = x + F.linear(o, ws[3])
    ref = z1 + F.linear(F.gelu(F.linear(ln(z1), ws[4], b0)), ws[5], b2)
    assert float((out - ref).abs().max()) < 1e-4
