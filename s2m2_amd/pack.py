"""Weight packing for the implicit-GEMM kernel (s2m2_conv2d): done once per (model, dtype) by the engine.

All layouts are (Cout_padded, KH*KW*Cin_padded), K = (tap, channel) with channel fastest.  Channel counts are padded to
multiples of 8 with zero weights so that every activation row is made of whole 16-byte pieces.

The MFMA-fragment orders of the direct-form kernels (chain_frag, pw_frag, narrow_frag, head_frag, fusion_frag, the K order 2 of
pack_conv_frag) are produced by the LIBRARY (s2m2_pack_frag, csrc/pack.hip) whenever the plain packing sits on the device: a caller of the
C ABI gets the same streams without this module.  The torch formulas below remain for CPU tensors; tests/test_hip_pack.py holds the two
bit for bit against each other.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def _native(w: torch.Tensor) -> bool:
    """the library packs: fp16 plain packing on a GPU (S2M2_PACK_TORCH=1 forces the torch formulas: A/B and tests)"""
    import os
    return w.is_cuda and w.dtype == torch.float16 and os.environ.get("S2M2_PACK_TORCH") != "1"


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def chunk_channels(dtype: torch.dtype) -> int:
    """CH of K order 1: 64 bytes of channels."""
    return 64 // torch.empty((), dtype=dtype).element_size()


def reorder_k(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(Cout, KH, KW, Cin) -> K order 1 of s2m2_conv2d: (Cout, KH, Cin/CH, KW, CH)."""
    co, kh, kw, ci = w.shape
    ch = chunk_channels(dtype)
    assert ci % ch == 0, (ci, ch)
    return w.reshape(co, kh, kw, ci // ch, ch).permute(0, 1, 3, 2, 4)


def pack_conv(w: torch.Tensor, dtype: torch.dtype, splits: Optional[Sequence[Tuple[int, int]]] = None,
              cout_pad: Optional[int] = None, korder: int = 0) -> torch.Tensor:
    """nn.Conv2d / nn.Linear weight (Cout, Cin[, KH, KW]) -> packed (Cout_p, KH*KW*Cin_p).

    splits: [(real, padded), ...] how the Cin input channels are laid out over the (concatenated, individually padded)
    sources; default one source padded to a multiple of 8."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    co, ci, kh, kw = w.shape
    if splits is None:
        splits = [(ci, pad8(ci))]
    assert sum(r for r, _ in splits) == ci, (splits, ci)
    cols: List[torch.Tensor] = []
    off = 0
    wt = w.permute(0, 2, 3, 1)                                   # (Cout, KH, KW, Cin)
    for real, padded in splits:
        blk = wt[..., off:off + real]
        if padded > real:
            blk = torch.cat([blk, blk.new_zeros(co, kh, kw, padded - real)], -1)
        cols.append(blk)
        off += real
    wp = torch.cat(cols, -1)                                      # (Cout, KH, KW, Cin_p)
    if korder:
        wp = reorder_k(wp, dtype)
    wp = wp.reshape(co, -1)
    cop = cout_pad if cout_pad is not None else pad8(co)
    if cop > co:
        wp = torch.cat([wp, wp.new_zeros(cop - co, wp.shape[1])], 0)
    return wp.to(dtype).contiguous()


FRAG_CH = 128        # channels per chunk of K order 2 (conv_frag_kernel: one halo tile in LDS per chunk) ...


def frag_chunk(cout_p: int, cin_p: int) -> int:
    """... except for the layers of 192-channel models: include/s2m2_hip.h s2m2_conv_frag_chunk (same rule, tests/test_cabi.py compares them)"""
    return 192 if (cout_p % 128 != 0 and cout_p % 192 == 0 and cin_p % 192 == 0) else FRAG_CH


def frag_eligible(cout_p: int, cin_p: int, kh: int, kw: int, dtype: torch.dtype) -> bool:
    """Layers the fragment-stream kernel (K order 2) takes: fp16, stride-1 3x3 / 3x1 / 1x3, Cout % 128 == 0, Cin % 64 == 0 and
    >= 128 (a half-empty last chunk of 128 channels costs 25 % at Cin = 192, 17 % at 320, nothing at 128 / 256 / 384; Cin = 64
    would waste half of the MFMAs and stays on the v3 tiles) -- or Cout % 192 == 0 with Cin % 192 == 0 (192-cout blocks on 192-channel chunks)."""
    if not (dtype == torch.float16 and kh * kw > 1 and kh in (1, 3) and kw in (1, 3)):
        return False
    if frag_chunk(cout_p, cin_p) == 192:
        import os
        return os.environ.get("S2M2_FRAG192", "1") != "0"          # A/B switch: 0 = these layers on the v3 halo tiles (as up to round 5)
    return cout_p % 128 == 0 and cin_p % 64 == 0 and cin_p >= 128


def pack_conv_frag(w: torch.Tensor, dtype: torch.dtype, splits: Optional[Sequence[Tuple[int, int]]] = None) -> torch.Tensor:
    """nn.Conv2d weight (Cout, Cin, KH, KW) -> K order 2 of s2m2_conv2d: the stream of MFMA A-fragments conv_frag_kernel consumes,
    [Cout/32][chunk][tap][k16 step][lane][8] with lane l holding cout 32t + l % 32, channels CK*chunk + 16*step + 8*(l // 32) + e
    (zero beyond Cin; CK = frag_chunk(Cout, Cin): 128 or 192).  Returned as (Cout, KH*KW*nchunk*CK): same row count as K order 0, K padded to whole chunks."""
    assert dtype == torch.float16
    kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    ntap = kh * kw
    if w.is_cuda and _native(w.new_empty(0, dtype=dtype)):
        from . import hip
        plain = pack_conv(w, dtype, splits)                       # rounded to fp16 first: the same values the torch formula rounds last
        assert plain.shape[0] % 32 == 0, plain.shape
        return hip.pack_frag(hip.PACK_CONV_FRAG, plain, ntap=ntap).reshape(plain.shape[0], -1)
    wp = pack_conv(w, torch.float32, splits)                      # (Cout_p, KH*KW*Cin_p), K = (tap, channel)
    cop, cin = wp.shape[0], wp.shape[1] // ntap
    assert cop % 32 == 0, cop
    ck = frag_chunk(cop, cin)
    nchunk = (cin + ck - 1) // ck
    t = wp.reshape(cop, ntap, cin)
    if nchunk * ck > cin:
        t = torch.cat([t, t.new_zeros(cop, ntap, nchunk * ck - cin)], -1)
    ks = ck // 16
    t = t.reshape(cop // 32, 32, ntap, nchunk, ks, 2, 8)          # cout tile, cout, tap, chunk, step, half, e
    t = t.permute(0, 3, 2, 4, 5, 1, 6)                            # cout tile, chunk, tap, step, half, cout, e  (lane = half*32 + cout)
    return t.reshape(cop, ntap * nchunk * ck).to(dtype).contiguous()


def chain_frag(wp: torch.Tensor) -> torch.Tensor:
    """packed 1x1 weight (n*C, C) (pack_conv: row = cout, column = input channel) -> the MFMA-fragment order of s2m2_chain_desc.weight_frag,
    per C x C layer [cout / 32][k16 step][lane][8] with lane l holding cout 32t + l % 32, channels 16*step + 8*(l // 32) + e.  Same
    shape and values, permuted."""
    rows, C = wp.shape
    assert rows % C == 0 and C % 32 == 0, (rows, C)
    if _native(wp):
        from . import hip
        return hip.pack_frag(hip.PACK_ROWS, wp).reshape(rows, C)
    t = wp.reshape(rows // C, C // 32, 32, C // 16, 2, 8).permute(0, 1, 3, 4, 2, 5)       # (layer, tile, step, half, l % 32, e)
    return t.contiguous().reshape(rows, C)


def pw_frag(wp: torch.Tensor) -> torch.Tensor:
    """packed 1x1 weight (Cout, K) (pack_conv / pack_convT_2x2s2: row = cout, column = concatenated input channel) -> the fragment order of
    s2m2_pw_direct: zero-padded to (32 * tiles, 16 * steps), then [tile][step][lane][8] with lane l holding row 32 t + l % 32, columns
    16 s + 8 (l // 32) + e."""
    cout, k = wp.shape
    if _native(wp):
        from . import hip
        return hip.pack_frag(hip.PACK_ROWS, wp).reshape((cout + 31) // 32, (k + 15) // 16, 64, 8)
    full = wp.new_zeros(((cout + 31) // 32 * 32, (k + 15) // 16 * 16))
    full[:cout, :k] = wp
    return _frag_rows(full).contiguous()


def narrow_frag(wp: torch.Tensor, ntap: int) -> torch.Tensor:
    """packed spatial weight (Cout, ntap * Cin) in K order 0 (pack_conv: K = (tap, channel)) -> the fragment order of s2m2_conv_narrow.
    Below 128 input channels that is pack.pw_frag of the matrix as it is; Cin = 128 / 256 run as chunks of 64 channels, whose K columns come
    chunk-major: K = (chunk, tap, channel within the chunk)."""
    cout, k = wp.shape
    cin = k // ntap
    assert cin * ntap == k
    if _native(wp):
        from . import hip
        assert cin < 128 or cin % 64 == 0, cin
        return hip.pack_frag(hip.PACK_NARROW, wp, ntap=ntap).reshape((cout + 31) // 32, (k + 15) // 16, 64, 8)
    if cin >= 128:
        assert cin % 64 == 0, cin
        wp = wp.reshape(cout, ntap, cin // 64, 64).permute(0, 2, 1, 3).reshape(cout, k)
    return pw_frag(wp)


def head_frag(w2: torch.Tensor) -> torch.Tensor:
    """packed 1x1 weight (Cout2 <= 32, K) of a layer fused behind a K12 3x3 layer (s2m2_narrow_desc.head_frag) -> (1, 2 * ceil(K / 32), 64, 8):
    k16 step (j, p), lane l = 32 * half + row, element 4 q + e holds W2[row, 32 j + 8 (2 p + q) + 4 half + e] -- the channels lane (pixel, half)
    of the 3x3 layer's accumulator tile j holds in its quads 2 p and 2 p + 1 (D[cout][pixel] of the 32x32 MFMA), so those registers are the
    head's pixel fragments as they are."""
    r, k = w2.shape
    assert r <= 32, r
    nj = (k + 31) // 32
    if _native(w2):
        from . import hip
        return hip.pack_frag(hip.PACK_HEAD, w2).reshape(1, nj * 2, 64, 8)
    full = w2.new_zeros((32, nj * 32))
    full[:r, :k] = w2
    t = full.reshape(32, nj, 2, 2, 2, 4)                          # row, j, p, q, half, e   (channel = 32 j + 8 (2 p + q) + 4 half + e)
    return t.permute(1, 2, 4, 0, 3, 5).reshape(1, nj * 2, 64, 8).contiguous()   # (j, p), half, row, (q, e)


def _frag_rows(w: torch.Tensor) -> torch.Tensor:
    """(R, K) -> (R/32, K/16, 64, 8): per 32-row tile and k16 step the MFMA A-fragment (lane l: row 32t + l % 32, k 16*step + 8*(l // 32) + e)"""
    R, K = w.shape
    return w.reshape(R // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(R // 32, K // 16, 64, 8)


def fusion_frag(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """K10 weights, w1 packed (3C, 2C) and w2 packed (C, 3C) = [Wg | Wf] -> the fragment stream of s2m2_feature_fusion_frag: per 32-cout
    tile t, for slice s = 0, 1, 2: the fragments of w1[sC + 32t .. + 32, :] (2C/16 steps), then of w2[32t .. + 32, sC : (s + 1)C] (C/16 steps).
    Flat tensor of 9*C*C values."""
    C = w2.shape[0]
    assert tuple(w1.shape) == (3 * C, 2 * C) and tuple(w2.shape) == (C, 3 * C) and C % 32 == 0
    if _native(w1) and _native(w2):
        from . import hip
        return hip.pack_frag(hip.PACK_FUSION, w1, w2=w2, rows=C)
    parts = []
    for s in range(3):
        parts.append(_frag_rows(w1[s * C:(s + 1) * C]))                          # (C/32, 2C/16, 64, 8)
        parts.append(_frag_rows(w2[:, s * C:(s + 1) * C].contiguous()))          # (C/32, C/16, 64, 8)
    return torch.cat(parts, dim=1).contiguous().reshape(-1)                      # (C/32, 9C/16, 64, 8)


def pack_bias(b: Optional[torch.Tensor], cout: int, cout_pad: Optional[int] = None) -> Optional[torch.Tensor]:
    if b is None:
        return None
    cop = cout_pad if cout_pad is not None else pad8(cout)
    out = b.new_zeros(cop, dtype=torch.float32)
    out[:cout] = b.float()
    return out.contiguous()


def convT_s1_as_conv(w: torch.Tensor) -> torch.Tensor:
    """nn.ConvTranspose2d weight (Cin, Cout, KH, KW), stride 1, padding (K-1)/2  ->  equivalent Conv2d weight
    (Cout, Cin, KH, KW): spatially flipped, channel axes swapped."""
    return w.flip(2, 3).permute(1, 0, 2, 3).contiguous()


def pack_convT_2x2s2(w: torch.Tensor, dtype: torch.dtype, cin_pad: Optional[int] = None) -> Tuple[torch.Tensor, int]:
    """nn.ConvTranspose2d(k=2, s=2) weight (Cin, Cout, 2, 2) -> GEMM weight (4*C', Cin_p) with rows ordered (dy, dx, c'),
    C' = Cout padded to a multiple of 8.  out[n, 2y+dy, 2x+dx, c'] = sum_ci in[n,y,x,ci] * w[ci, c', dy, dx]."""
    ci, co = w.shape[:2]
    cp = pad8(co)
    cip = cin_pad if cin_pad is not None else pad8(ci)
    g = w.new_zeros(2, 2, cp, cip)
    g[:, :, :co, :ci] = w.permute(2, 3, 1, 0)
    return g.reshape(4 * cp, cip).to(dtype).contiguous(), cp


def pack_bias_shuffle(b: Optional[torch.Tensor], cout: int) -> Optional[torch.Tensor]:
    if b is None:
        return None
    cp = pad8(cout)
    out = b.new_zeros(4, cp, dtype=torch.float32)
    out[:, :cout] = b.float()[None]
    return out.reshape(-1).contiguous()


def rowattn_cols(w: torch.Tensor) -> torch.Tensor:
    """The column order of K13's operands: per 16 input channels the four quads of 4 channels in the order (0, 2, 1, 3) -- k-slot 8a + 4b + e
    of a 16-channel group holds channel 8b + 4a + e, which makes the fp16 accumulator tile of one layer the B operand of the next
    (csrc/rowattn.hip).  An involution: applying it twice returns the input."""
    r, k = w.shape
    assert k % 16 == 0, "row_attn packing: K must be a multiple of 16"
    return w.reshape(r, k // 16, 2, 2, 4).transpose(2, 3).reshape(r, k).contiguous()


def rowattn_pack(w: torch.Tensor) -> torch.Tensor:
    """Plain packed (n * 128, 128) 1x1 weight(s) -> the "row_attn packing" of s2m2_row_attn (include/s2m2_hip.h), per block of 128 rows:
    columns in K13's order (rowattn_cols), then UNIT-MAJOR -- the 16-byte unit u (columns 8u .. 8u+7) of row r at element (u * 128 + r) * 8 --
    so that the kernel's LDS-DMA is a linear copy and its fragment reads are conflict-free.  Same shape, contiguous."""
    r, k = w.shape
    assert k == 128 and r % 128 == 0, "row_attn packing: (n * 128, 128) layers"
    return rowattn_cols(w).reshape(r // 128, 128, 16, 8).transpose(1, 2).contiguous().reshape(r, k)


def rowattn_unpack(wp: torch.Tensor) -> torch.Tensor:
    """inverse of rowattn_pack (tests, the CPU stand-in)"""
    r, k = wp.shape
    return rowattn_cols(wp.reshape(r // 128, 16, 128, 8).transpose(1, 2).contiguous().reshape(r, k))


def rowattn_vectors(wsums, biases, ln_affine=None) -> torch.Tensor:
    """The (12, 128) fp32 vector block of s2m2_row_attn: wsums = row sums of (q, k, v, ffn.0), biases = (q, k, v, proj, ffn.0, ffn.2) or None
    each, ln_affine = (gamma, beta) of the optional LayerNorm output or None."""
    dev = wsums[0].device
    z = torch.zeros(128, device=dev)
    f = lambda t: z if t is None else t.float().reshape(128).to(dev)      # noqa: E731
    bq, bk, bv, bp, b0, b2 = (f(b) for b in biases)
    g, b = (f(ln_affine[0]), f(ln_affine[1])) if ln_affine is not None else (z + 1.0, z)
    return torch.stack([bq, f(wsums[0]), bk, f(wsums[1]), bv, f(wsums[2]), bp, b0, f(wsums[3]), b2, g, b]).contiguous()
