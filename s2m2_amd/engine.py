"""Inference engine behind ``S2M2.forward`` on MI355X.

Activations are channels-last everywhere (NCHW-shaped tensors with ``torch.channels_last`` strides, so the
``(B,H,W,C)`` token view the attention blocks and the HIP kernels want is free).  Weights are cast/packed once per
(dtype, weight version).  Stage map (reference file:line -> what runs here):

* CNN backbone (submodules.py:63-93): PyTorch-ROCm convolutions (MIOpen) -- stays on the vendor library by design.
* LayerNorm + correlation, Sinkhorn/argmax/regression, cost-volume lookups (submodules.py:19-60,154-243):
  hand-written HIP kernels K1-K3 through the C ABI (:mod:`s2m2_amd.hip`).
* Everything else is being moved to HIP kernels stage by stage; until a stage has its kernel it runs as PyTorch-ROCm
  ops here (never on the CPU, never through oracle/).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import hip

Tensor = torch.Tensor


class Engine:
    def __init__(self, model, dtype: torch.dtype):
        hip.load()                                   # fail loudly if the kernel library is missing
        self.dtype = dtype
        self.C = model.feature_channels
        self.ntr = model.num_transformer
        self.use_positivity = model.use_positivity
        self.refine_iter = model.refine_iter
        self.output_upsample = model.output_upsample
        self.device = next(model.parameters()).device
        self.w: Dict[str, Tensor] = {}
        for name, p in model.named_parameters():
            t = p.detach().to(self.device, dtype)
            if t.dim() == 4:
                t = t.contiguous(memory_format=torch.channels_last)
            self.w[name] = t
        # LayerNorm affine of DispInit is consumed in fp32 by K1
        self.ln_w = model.get_parameter("disp_init.layer_norm.weight").detach().float().contiguous()
        self.ln_b = model.get_parameter("disp_init.layer_norm.bias").detach().float().contiguous()
        self._pe_cache: Dict[Tuple[int, int], Tensor] = {}
        self.k1_events = None                        # bench.py: list collecting (start, end) HIP events around K1

    # ---- dense helpers ------------------------------------------------------------------------------
    def conv(self, p: str, x: Tensor, stride: int = 1, pad=0) -> Tensor:
        return F.conv2d(x, self.w[p + ".weight"], self.w.get(p + ".bias"), stride=stride, padding=pad)

    def convT(self, p: str, x: Tensor, stride: int = 1, pad: int = 0) -> Tensor:
        return F.conv_transpose2d(x, self.w[p + ".weight"], self.w.get(p + ".bias"), stride=stride, padding=pad)

    def lin(self, p: str, x: Tensor) -> Tensor:
        return F.linear(x, self.w[p + ".weight"], self.w.get(p + ".bias"))

    @staticmethod
    def ln(x: Tensor) -> Tensor:
        return F.layer_norm(x, (x.shape[-1],))

    def down(self, p: str, x: Tensor) -> Tensor:
        return self.conv(p + ".1", F.avg_pool2d(x, 2))

    def up(self, p: str, x: Tensor) -> Tensor:
        return self.conv(p + ".1", F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False))

    def conv_block(self, p: str, z: Tensor) -> Tensor:
        a = self.conv(p + ".convs.2", F.gelu(self.conv(p + ".convs.0", z, 1, 1)), 1, 1)
        b = self.conv(p + ".convs_1x.2", F.relu(self.conv(p + ".convs_1x.0", z)))
        return a + b

    def fusion(self, p: str, z0: Tensor, z1: Tensor) -> Tensor:
        z = torch.cat([z0, z1], 1)
        k = self.w[p + ".feature_gate.0.weight"].shape[-1]
        g = torch.sigmoid(self.conv(p + ".feature_gate.2", F.gelu(self.conv(p + ".feature_gate.0", z, 1, k // 2))))
        g = g.clamp(0.01, 0.99)
        f = self.conv(p + ".feature_fusion.2", F.gelu(self.conv(p + ".feature_fusion.0", z, 1, k // 2)))
        return f + g * z0 + (1 - g) * z1

    # ---- attention ----------------------------------------------------------------------------------
    @staticmethod
    def _heads(x: Tensor, nh: int) -> Tensor:
        b, n, c = x.shape
        return x.reshape(b, n, nh, c // nh).transpose(1, 2)

    def pe(self, h: int, w: int) -> Tensor:
        key = (h, w)
        if key not in self._pe_cache:
            self._pe_cache[key] = _dense_pe(h, w, self.device).to(self.dtype)
        return self._pe_cache[key]

    def self_attn(self, p: str, x: Tensor, nh: int, pe: Optional[Tensor]) -> Tensor:
        b, n, c = x.shape
        q, k, v = (self._heads(self.lin(p + "." + t, x), nh) for t in ("q", "k", "v"))
        if (p + ".pe_proj.weight") in self.w:
            s = torch.matmul(q * (q.shape[-1] ** -0.5), k.transpose(-1, -2))
            a = torch.softmax(s.float(), dim=-1).to(self.dtype)
            o = torch.matmul(a, v) + self.lin(p + ".pe_proj", torch.einsum("bhij,ijc->bhic", a, pe))
        else:
            o = F.scaled_dot_product_attention(q, k, v)
        return self.lin(p + ".proj", o.transpose(1, 2).reshape(b, n, -1))

    def cross_attn(self, p: str, x: Tensor, y: Tensor, nh: int) -> Tuple[Tensor, Tensor]:
        b, n, c = x.shape
        qx, kx, vx = (self._heads(self.lin(p + "." + t, x), nh) for t in ("q", "k", "v"))
        qy, ky, vy = (self._heads(self.lin(p + "." + t, y), nh) for t in ("q", "k", "v"))
        ox = F.scaled_dot_product_attention(qx, ky, vy).transpose(1, 2).reshape(b, n, -1)
        oy = F.scaled_dot_product_attention(qy, kx, vx).transpose(1, 2).reshape(b, n, -1)
        return self.lin(p + ".proj", ox), self.lin(p + ".proj", oy)

    def ffn(self, p: str, z: Tensor) -> Tensor:
        return self.lin(p + ".ffn.2", F.gelu(self.lin(p + ".ffn.0", self.ln(z)))) + z

    def cross_block(self, p: str, z: Tensor, nh: int, two_d: bool) -> Tensor:
        zn = self.ln(z)
        x, y = zn.chunk(2, 0)
        b, h, w, c = x.shape
        shp = (b, h * w, c) if two_d else (b * h, w, c)
        ox, oy = self.cross_attn(p + ".attn", x.reshape(shp), y.reshape(shp), nh)
        return torch.cat([ox.reshape(b, h, w, c), oy.reshape(b, h, w, c)], 0) + z

    def self_block(self, p: str, z: Tensor, nh: int, two_d: bool, pe: Optional[Tensor]) -> Tensor:
        b, h, w, c = z.shape
        zz = z.reshape((b, h * w, c) if two_d else (b * h, w, c))
        return (self.self_attn(p + ".attn", self.ln(zz), nh, pe) + zz).reshape(b, h, w, c)

    def attn_block(self, p: str, z: Tensor, nh: int, two_d: bool, pe: Optional[Tensor] = None) -> Tensor:
        """BasicAttnBlock (1-D, attentions.py:347-355) / GlobalAttnBlock (2-D, :311-321); z NCHW-shaped channels-last."""
        t = z.permute(0, 2, 3, 1)
        if (p + ".cross_attn.attn.q.weight") in self.w:
            t = self.ffn(p + ".ffn_c", self.cross_block(p + ".cross_attn", t, nh, two_d))
        t = self.ffn(p + ".ffn", self.self_block(p + ".self_attn", t, nh, two_d, pe))
        return t.permute(0, 3, 1, 2)

    def _count(self, prefix: str) -> int:
        n = 0
        while f"{prefix}.{n}.ffn.ffn.0.weight" in self.w:
            n += 1
        return n

    # ---- U-Net / MRT --------------------------------------------------------------------------------
    def unet(self, p: str, z: Tensor):
        use_pe = (p + ".enc3s.0.self_attn.attn.pe_proj.weight") in self.w
        pe = self.pe(z.shape[-2] // 8, z.shape[-1] // 8) if use_pe else None
        z0 = self.conv_block(p + ".enc0", z)
        z1 = self.conv_block(p + ".enc1", self.down(p + ".down_conv0", z0))
        z2 = self.conv_block(p + ".enc2", self.down(p + ".down_conv1", z1))
        z3 = self.down(p + ".down_conv2", z2)
        for i in range(self._count(p + ".enc3s")):
            z3 = self.attn_block(f"{p}.enc3s.{i}", z3, 8, True, pe)
        for i in range(self._count(p + ".dec3s")):
            z3 = self.attn_block(f"{p}.dec3s.{i}", z3, 8, True, None)
        n2 = self.conv_block(p + ".dec2", self.fusion(p + ".concat_conv2", z2, self.up(p + ".up_conv2", z3)))
        n1 = self.conv_block(p + ".dec1", self.fusion(p + ".concat_conv1", z1, self.up(p + ".up_conv1", n2)))
        n0 = self.conv_block(p + ".dec0", self.fusion(p + ".concat_conv0", z0, self.up(p + ".up_conv0", n1)))
        return n0, n1, n2, z3

    def mrt(self, p: str, z0: Tensor, z1: Tensor, z2: Tensor, z3: Tensor):
        z0 = self.attn_block(p + ".enc_attn0", z0, 1, False)
        z1 = self.attn_block(p + ".enc_attn1", self.fusion(p + ".down_concat1", z1, self.down(p + ".down_conv0", z0)), 2, False)
        z2 = self.attn_block(p + ".enc_attn2", self.fusion(p + ".down_concat2", z2, self.down(p + ".down_conv1", z1)), 4, False)
        z3 = self.fusion(p + ".down_concat3", z3, self.down(p + ".down_conv2", z2))
        for i in range(2):
            z3 = self.attn_block(f"{p}.enc_attn3s.{i}", z3, 8, True)
        for i in range(2):
            z3 = self.attn_block(f"{p}.dec_attn3s.{i}", z3, 8, True)
        z2 = self.attn_block(p + ".dec_attn2", self.fusion(p + ".up_concat2", z2, self.up(p + ".up_conv2", z3)), 4, False)
        z1 = self.attn_block(p + ".dec_attn1", self.fusion(p + ".up_concat1", z1, self.up(p + ".up_conv1", z2)), 2, False)
        z0 = self.attn_block(p + ".dec_attn0", self.fusion(p + ".up_concat0", z0, self.up(p + ".up_conv0", z1)), 1, False)
        return z0, z1, z2, z3

    # ---- refiners -----------------------------------------------------------------------------------
    def global_refiner(self, p: str, ctx: Tensor, disp: Tensor, conf: Tensor) -> Tensor:
        mask = (conf > 0.2).float()
        x = torch.cat([(disp / 1e2 * mask).to(self.dtype), torch.logit(mask * conf, eps=1e-1).to(self.dtype), ctx], 1)
        f = self.conv(p + ".init_feat.2", F.gelu(self.conv(p + ".init_feat.0", x, 1, 1)))
        f = self.unet(p + ".refine_unet", f)[0]
        upd = self.conv(p + ".out_feat.0", f, 1, 1).float() * 1e2
        return mask * disp + (1 - mask) * upd

    def gru(self, p: str, h: Tensor, x: Tensor) -> Tensor:
        for sfx, pad in (("1", (1, 0)), ("2", (0, 1))):
            hx = torch.cat([h, x], 1)
            z = torch.sigmoid(self.conv(f"{p}.convz{sfx}", hx, 1, pad))
            r = torch.sigmoid(self.conv(f"{p}.convr{sfx}", hx, 1, pad))
            q = torch.tanh(self.conv(f"{p}.convq{sfx}", torch.cat([r * h, x], 1), 1, pad))
            h = (1 - z) * h + z * q
        return h

    def local_refiner(self, p: str, hidden: Tensor, ctx: Tensor, disp: Tensor, conf: Tensor, occ: Tensor, cv: Tensor, cap, it):
        cl = torch.logit(conf, eps=1e-2)
        ol = torch.logit(occ, eps=1e-2)
        c1, c2 = hip.cv_lookup(cv, disp.contiguous(), 4, channels_last=True, out_dtype=self.dtype)          # K3
        c1, c2 = c1.permute(0, 3, 1, 2), c2.permute(0, 3, 1, 2)
        if cap is not None:
            cap[f"corr1_it{it}"], cap[f"corr2_it{it}"] = c1, c2
        f1 = self.conv(p + ".corr_feat1.2", F.gelu(self.conv(p + ".corr_feat1.0", c1 / 16)))
        f2 = self.conv(p + ".corr_feat2.2", F.gelu(self.conv(p + ".corr_feat2.0", c2 / 16)))
        fd = self.conv(p + ".disp_feat.2", F.gelu(self.conv(p + ".disp_feat.0", (disp / 1e2).to(self.dtype), 1, 1)), 1, 1)
        fc = self.conv(p + ".conf_occ_feat.2", F.gelu(self.conv(p + ".conf_occ_feat.0", torch.cat([cl, ol], 1).to(self.dtype), 1, 1)))
        x = torch.cat([fd, f1, f2, ctx, fc], 1)
        x = self.conv(p + ".disp_corr_ctx_cat.2", F.gelu(self.conv(p + ".disp_corr_ctx_cat.0", x)), 1, 1)
        x = self.unet(p + ".refine_unet", x)[0]
        hn = self.gru(p + ".gru", hidden, x)
        dd = self.conv(p + ".disp_update.2", F.gelu(self.conv(p + ".disp_update.0", hn, 1, 1)), 1, 1).float()
        co = self.conv(p + ".conf_occ_update.2", F.gelu(self.conv(p + ".conf_occ_update.0", hn, 1, 1)), 1, 1).float()
        return hn, disp + dd, torch.sigmoid(co[:, 0:1] + cl), torch.sigmoid(co[:, 1:2] + ol)

    # ---- upsampling ---------------------------------------------------------------------------------
    def mask4x(self, p: str, hidden: Tensor, f2x: Tensor) -> Tensor:
        a = self.convT(p + ".conv_x", hidden, 2)
        b = self.conv(p + ".conv_y", f2x, 1, 1)
        y = F.relu(self.conv(p + ".conv_concat.0", torch.cat([a, b], 1), 1, 1))
        return self.convT(p + ".conv_concat.2", y, 2)

    def mask1x(self, p: str, disp: Tensor, rgb: Tensor, f2x: Tensor) -> Tensor:
        a = F.relu(self.convT(p + ".conv_disp.0", disp.to(self.dtype), 1, 1))
        b = F.relu(self.convT(p + ".conv_rgb.0", rgb, 1, 1))
        c = self.convT(p + ".conv_ctx", f2x, 2)
        y = F.relu(self.conv(p + ".conv_concat.0", torch.cat([a, b, c], 1), 1, 1))
        return self.convT(p + ".conv_concat.2", y)

    @staticmethod
    def _neigh9(x: Tensor) -> Tensor:
        B, _, h, w = x.shape
        xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
        return torch.cat([xp[:, :, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], 1)

    def upsample4x(self, x: Tensor, logits: Tensor) -> Tensor:
        n = F.interpolate(self._neigh9(x), scale_factor=4, mode="nearest")
        return (n * logits.float().softmax(1)).sum(1, keepdim=True)

    def upsample1x(self, x: Tensor, logits: Tensor) -> Tensor:
        n = self._neigh9(x)
        if self.output_upsample:
            n = F.interpolate(n, scale_factor=2, mode="nearest")
            logits = F.interpolate(logits, scale_factor=2, mode="bilinear", align_corners=False)
        return (n * logits.float().softmax(1)).sum(1, keepdim=True)

    # ---- whole forward ------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, img0: Tensor, img1: Tensor, cap: Optional[dict] = None):
        dt = self.dtype
        B = img0.shape[0]
        x = torch.cat([img0, img1], 0).to(torch.float32)
        x = ((x / 255.0 - 0.5) * 2).to(dt).contiguous(memory_format=torch.channels_last)
        # backbone (PyTorch-ROCm / MIOpen)
        p = "cnn_backbone"
        t = self.conv(p + ".conv0.2", F.gelu(self.conv(p + ".conv0.0", x)))
        f2 = self.conv(p + ".conv1_down.2", F.gelu(self.conv(p + ".conv1_down.0", t, 2, 2)), 1, 1)
        f2 = F.group_norm(f2, 8, self.w[p + ".norm1.weight"], self.w[p + ".norm1.bias"])
        f2 = self.conv(p + ".conv2.2", F.gelu(self.conv(p + ".conv2.0", f2, 1, 1)), 1, 1) + f2
        f4 = self.conv(p + ".conv2_down.0", f2, 2, 1)
        f2_left = f2[:B]
        # feature pyramid + multi-resolution transformer
        py = self.unet("feat_pyramid", f4)
        z = py
        for i in range(self.ntr):
            z = self.mrt(f"transformer.uformer_list.{i}", *z)
        tr = z[0]
        tokens = tr.permute(0, 2, 3, 1).contiguous()                        # (2B,h,w,C): free for channels-last
        # K1 + K2: cost volume, optimal transport, initial disparity
        if self.k1_events is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        cv = hip.ln_corr(tokens, self.ln_w, self.ln_b)
        if self.k1_events is not None:
            ev1.record()
            self.k1_events.append((ev0, ev1))
        disp, conf, occ, amax = hip.sinkhorn_regress(cv, self.use_positivity, 3, want_argmax=True)
        if cap is not None:
            cap.update(feature_tr_4x=tr, feature_py_4x=py[0], cv=cv, argmax=amax, disp0=disp, conf0=conf, occ0=occ)
        tr0 = tr[:B]
        disp = self.global_refiner("global_refiner", tr0, disp, conf)
        if self.use_positivity:
            disp = disp.clamp(min=0)
        if cap is not None:
            cap["disp_g"] = disp
        fus = self.fusion("feat_fusion_layer", tr0, py[0][:B])
        ctx = self.conv("ctx_feat.2", F.gelu(self.conv("ctx_feat.0", fus)))
        hidden = torch.tanh(ctx)
        w = disp.shape[-1]
        xs = torch.arange(w, device=disp.device, dtype=torch.float32).reshape(1, 1, 1, w)
        for it in range(self.refine_iter):
            hidden, disp, conf, occ = self.local_refiner("refiner", hidden, ctx, disp, conf, occ, cv, cap, it)
            if self.use_positivity:
                disp = disp.clamp(min=0)
            occ = occ * (xs - disp >= 0)
            if cap is not None:
                cap[f"disp_it{it}"], cap[f"conf_it{it}"], cap[f"occ_it{it}"] = disp, conf, occ
        m4 = self.mask4x("upsample_mask_4x_refine", hidden, f2_left)
        d_up = self.upsample4x(disp * 4, m4)
        o_up = self.upsample4x(occ, m4)
        c_up = self.upsample4x(conf, m4)
        m1 = self.mask1x("upsample_mask_1x", d_up, x[:B], f2_left)
        d_up = self.upsample1x(d_up, m1)
        o_up = self.upsample1x(o_up, m1)
        c_up = self.upsample1x(c_up, m1)
        if self.output_upsample:
            d_up = 2 * d_up
        return d_up, o_up, c_up


def _dense_pe(h: int, w: int, device, pe_dim: int = 32) -> Tensor:
    """Dense (N,N,32) sinc relative positional encoding (reference utils.py:32-60); fp32.
    TODO(kernel): never materialise -- the attention-with-PE kernel consumes the two separable tables."""
    def table(n: int) -> Tensor:
        L = 2 * n + 1
        sig = 5 / pe_dim
        pos = torch.linspace(-3, 3, L, device=device).tanh()
        dim_t = torch.linspace(-1, 1, pe_dim // 2, device=device)
        x = (dim_t[None, :] - pos[:, None]) / sig
        s = torch.where(x.abs() < 1e-6, torch.ones_like(x), torch.sin(3.1415 * x) / (3.1415 * x))
        return F.normalize(s, p=2, dim=-1)
    px, py = table(w), table(h)
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    return 0.5 * torch.cat([px[xs[:, None] - xs[None, :] + w - 1], py[ys[:, None] - ys[None, :] + h - 1]], dim=2)
