"""Weight packing for the implicit-GEMM kernel (s2m2_conv2d): done once per (model, dtype) by the engine.

All layouts are (Cout_padded, KH*KW*Cin_padded), K = (tap, channel) with channel fastest.  Channel counts are padded to
multiples of 8 with zero weights so that every activation row is made of whole 16-byte pieces.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


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
