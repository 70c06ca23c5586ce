"""Parameter table of the S2M2 network (names + shapes), built from a compact description.

The product does not mirror the reference's ``nn.Module`` classes.  Instead the whole network is
described by a flat table ``{dotted_name: shape}`` that is *key-for-key identical* to
``reference S2M2(...).state_dict()`` (src/s2m2/core/model/s2m2.py:14-67 and the sub-module
constructors it calls), so real ``CH{C}NTR{n}.pth`` checkpoints drop in
(src/s2m2/core/utils/model_utils.py:27,39-40).  ``tests/golden/state_dict_spec_*.json`` holds the
key/shape list dumped from the reference itself; ``tests/test_spec.py`` compares this builder with it.

Shapes follow PyTorch conventions: Conv2d (out, in, kh, kw); ConvTranspose2d (in, out, kh, kw);
Linear (out, in).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

Shape = Tuple[int, ...]

# model_type -> (feature_channels, num_transformer); src/s2m2/core/utils/model_utils.py:12-17
MODEL_CONFIGS = {
    "S": (128, 1),
    "M": (192, 2),
    "L": (256, 3),
    "XL": (384, 3),
}


class _Table:
    def __init__(self) -> None:
        self.t: "OrderedDict[str, Shape]" = OrderedDict()

    def conv(self, name: str, cin: int, cout: int, kh: int, kw: int | None = None, bias: bool = True) -> None:
        kw = kh if kw is None else kw
        self.t[name + ".weight"] = (cout, cin, kh, kw)
        if bias:
            self.t[name + ".bias"] = (cout,)

    def convT(self, name: str, cin: int, cout: int, k: int) -> None:
        self.t[name + ".weight"] = (cin, cout, k, k)
        self.t[name + ".bias"] = (cout,)

    def linear(self, name: str, cin: int, cout: int, bias: bool = True) -> None:
        self.t[name + ".weight"] = (cout, cin)
        if bias:
            self.t[name + ".bias"] = (cout,)

    def affine(self, name: str, c: int) -> None:
        self.t[name + ".weight"] = (c,)
        self.t[name + ".bias"] = (c,)


def _fusion(t: _Table, p: str, dim: int, k: int) -> None:
    # feature_fusion.py:15-21
    t.conv(p + ".feature_gate.0", 2 * dim, dim, k)
    t.conv(p + ".feature_gate.2", dim, dim, 1)
    t.conv(p + ".feature_fusion.0", 2 * dim, 2 * dim, k)
    t.conv(p + ".feature_fusion.2", 2 * dim, dim, 1)


def _attn(t: _Table, p: str, dim: int, e: int, heads: int, pe: bool) -> None:
    # attentions.py:24-30 / 71-74   (q,k,proj: no bias; v: bias; pe_proj: Linear(32, head_dim))
    t.linear(p + ".q", dim, e * dim, bias=False)
    t.linear(p + ".k", dim, e * dim, bias=False)
    t.linear(p + ".v", dim, e * dim, bias=True)
    t.linear(p + ".proj", e * dim, dim, bias=False)
    if pe:
        t.linear(p + ".pe_proj", 32, e * dim // heads)


def _ffn(t: _Table, p: str, dim: int, e: int) -> None:
    t.linear(p + ".ffn.0", dim, e * dim)
    t.linear(p + ".ffn.2", e * dim, dim)


def _global_block(t: _Table, p: str, dim: int, e: int, heads: int, cross: bool, pe: bool) -> None:
    # attentions.py:284-309 (registration order: self_attn, cross_attn, ffn_c, ffn)
    _attn(t, p + ".self_attn.attn", dim, e, heads, pe)
    if cross:
        _attn(t, p + ".cross_attn.attn", dim, e, heads, False)
        _ffn(t, p + ".ffn_c", dim, e)
    _ffn(t, p + ".ffn", dim, e)


def _basic_block(t: _Table, p: str, dim: int, e: int, heads: int) -> None:
    # attentions.py:324-345 (registration order: cross_attn, self_attn, ffn_c, ffn)
    _attn(t, p + ".cross_attn.attn", dim, e, heads, False)
    _attn(t, p + ".self_attn.attn", dim, e, heads, False)
    _ffn(t, p + ".ffn_c", dim, e)
    _ffn(t, p + ".ffn", dim, e)


def _conv_block(t: _Table, p: str, dim: int, e: int) -> None:
    # attentions.py:269-275
    t.conv(p + ".convs.0", dim, e * dim, 3)
    t.conv(p + ".convs.2", e * dim, dim, 3)
    t.conv(p + ".convs_1x.0", dim, e * dim, 1)
    t.conv(p + ".convs_1x.2", e * dim, dim, 1)


def _resample_convs(t: _Table, p: str, d: Tuple[int, int, int]) -> None:
    # stacked_MRT.py:22-34 == unet.py:25-37
    t.conv(p + ".down_conv0.1", d[0], d[1], 1)
    t.conv(p + ".down_conv1.1", d[1], d[2], 1)
    t.conv(p + ".down_conv2.1", d[2], d[2], 1)
    t.conv(p + ".up_conv0.1", d[1], d[0], 1)
    t.conv(p + ".up_conv1.1", d[2], d[1], 1)
    t.conv(p + ".up_conv2.1", d[2], d[2], 1)


def _unet(t: _Table, p: str, d: Tuple[int, int, int], e: int, n_attn: int, pe: bool) -> None:
    # unet.py:13-63
    _resample_convs(t, p, d)
    for i in range(3):
        _fusion(t, f"{p}.concat_conv{i}", d[i], 1)
    for i in range(3):
        _conv_block(t, f"{p}.enc{i}", d[i], e)
    for i in range(n_attn):
        _global_block(t, f"{p}.enc3s.{i}", d[2], e, 8, False, pe)
    for i in range(3):
        _conv_block(t, f"{p}.dec{i}", d[i], e)
    for i in range(n_attn):
        _global_block(t, f"{p}.dec3s.{i}", d[2], e, 8, False, False)


def _mrt(t: _Table, p: str, d: Tuple[int, int, int], e: int, nh: int) -> None:
    # stacked_MRT.py:10-86
    _resample_convs(t, p, d)
    _fusion(t, p + ".down_concat1", d[1], 1)
    _fusion(t, p + ".down_concat2", d[2], 1)
    _fusion(t, p + ".down_concat3", d[2], 1)
    _fusion(t, p + ".up_concat0", d[0], 1)
    _fusion(t, p + ".up_concat1", d[1], 1)
    _fusion(t, p + ".up_concat2", d[2], 1)
    _basic_block(t, p + ".enc_attn0", d[0], e, 1 * nh)
    _basic_block(t, p + ".enc_attn1", d[1], e, 2 * nh)
    _basic_block(t, p + ".enc_attn2", d[2], e, 4 * nh)
    for i in range(2):
        _global_block(t, f"{p}.enc_attn3s.{i}", d[2], e, 8 * nh, True, False)
    _basic_block(t, p + ".dec_attn0", d[0], e, 1 * nh)
    _basic_block(t, p + ".dec_attn1", d[1], e, 2 * nh)
    _basic_block(t, p + ".dec_attn2", d[2], e, 4 * nh)
    for i in range(2):
        _global_block(t, f"{p}.dec_attn3s.{i}", d[2], e, 8 * nh, True, False)


def param_table(feature_channels: int, dim_expansion: int, num_transformer: int) -> "OrderedDict[str, Shape]":
    """Ordered ``{name: shape}`` identical to the reference ``S2M2.state_dict()`` (s2m2.py:30-67)."""
    C, e, ntr = feature_channels, dim_expansion, num_transformer
    t = _Table()
    # cnn_backbone: submodules.py:70-86
    p = "cnn_backbone"
    t.conv(p + ".conv0.0", 3, 16, 1)
    t.conv(p + ".conv0.2", 16, 16, 1)
    t.conv(p + ".conv1_down.0", 16, 64, 5)
    t.conv(p + ".conv1_down.2", 64, C, 3)
    t.affine(p + ".norm1", C)
    t.conv(p + ".conv2.0", C, C, 3)
    t.conv(p + ".conv2.2", C, C, 3)
    t.conv(p + ".conv2_down.0", C, C, 3)
    # feat_pyramid: s2m2.py:34-38
    _unet(t, "feat_pyramid", (C, C, 2 * C), e, 2 * ntr, True)
    # transformer: s2m2.py:40-44
    for i in range(ntr):
        _mrt(t, f"transformer.uformer_list.{i}", (C, C, 2 * C), e, 1)
    # disp_init: submodules.py:165
    t.affine("disp_init.layer_norm", C)
    # upsample_mask_1x: submodules.py:127-135
    p = "upsample_mask_1x"
    t.convT(p + ".conv_disp.0", 1, 16, 3)
    t.convT(p + ".conv_rgb.0", 3, 16, 3)
    t.convT(p + ".conv_ctx", C, 16, 2)
    t.conv(p + ".conv_concat.0", 48, 48, 3)
    t.convT(p + ".conv_concat.2", 48, 9, 1)
    # upsample_mask_4x_refine: submodules.py:104-108
    p = "upsample_mask_4x_refine"
    t.convT(p + ".conv_x", C, 64, 2)
    t.conv(p + ".conv_y", C, 64, 3)
    t.conv(p + ".conv_concat.0", 128, 128, 3)
    t.convT(p + ".conv_concat.2", 128, 9, 2)
    # global_refiner: refinenet.py:47-57
    p = "global_refiner"
    t.conv(p + ".init_feat.0", 2 + C, C, 3)
    t.conv(p + ".init_feat.2", C, C, 1)
    _unet(t, p + ".refine_unet", (C, C, C), 1, 1, False)
    t.conv(p + ".out_feat.0", C, 1, 3)
    # feat_fusion_layer: s2m2.py:59
    _fusion(t, "feat_fusion_layer", C, 3)
    # refiner: refinenet.py:87-124
    p = "refiner"
    t.conv(p + ".disp_feat.0", 1, 96, 3)
    t.conv(p + ".disp_feat.2", 96, 96, 3)
    t.conv(p + ".corr_feat1.0", 9, 96, 1)
    t.conv(p + ".corr_feat1.2", 96, 64, 1)
    t.conv(p + ".corr_feat2.0", 9, 96, 1)
    t.conv(p + ".corr_feat2.2", 96, 64, 1)
    t.conv(p + ".conf_occ_feat.0", 2, 64, 3)
    t.conv(p + ".conf_occ_feat.2", 64, 32, 1)
    t.conv(p + ".disp_corr_ctx_cat.0", 256 + C, 2 * C, 1)
    t.conv(p + ".disp_corr_ctx_cat.2", 2 * C, C, 3)
    _unet(t, p + ".refine_unet", (C, C, 2 * C), e, 1, False)
    t.conv(p + ".disp_update.0", C, C, 3)
    t.conv(p + ".disp_update.2", C, 1, 3, bias=False)
    t.conv(p + ".conf_occ_update.0", C, C, 3)
    t.conv(p + ".conf_occ_update.2", C, 2, 3, bias=False)
    for g in ("z", "r", "q"):
        t.conv(f"{p}.gru.conv{g}1", 2 * C, C, 3, 1)
    for g in ("z", "r", "q"):
        t.conv(f"{p}.gru.conv{g}2", 2 * C, C, 1, 3)
    # ctx_feat: s2m2.py:65-67
    t.conv("ctx_feat.0", C, C, 1)
    t.conv("ctx_feat.2", C, C, 1)
    return t.t


def num_parameters(feature_channels: int, dim_expansion: int, num_transformer: int) -> int:
    n = 0
    for shape in param_table(feature_channels, dim_expansion, num_transformer).values():
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
