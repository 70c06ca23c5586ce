"""ctypes binding of libs2m2_hip.so (include/s2m2_hip.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the stream; every function below enqueues hand-written
gfx950 kernels on ``torch.cuda.current_stream()`` through the C ABI.  There is NO fallback: if the shared
library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libs2m2_hip%s.so" % os.environ.get("S2M2_LIB_SUFFIX", ""))   # suffix: experiment builds only

F32, F16 = 0, 1
_DT = {torch.float32: F32, torch.float16: F16}

_lib: Optional[ctypes.CDLL] = None

_vp, _i, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong

ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
EPI_NONE, EPI_ADD, EPI_MUL, EPI_GRU, EPI_GATEMIX, EPI_DUALMIX = 0, 1, 2, 3, 4, 5


class ConvDesc(ctypes.Structure):
    """mirror of s2m2_conv_desc (include/s2m2_hip.h)"""
    _fields_ = [("src", _vp * 4), ("src_c", _i * 4), ("src_stride", _i * 4), ("nsrc", _i), ("weight", _vp), ("bias", _vp),
                ("out", _vp), ("out_stride", _i), ("N", _i), ("H", _i), ("W", _i), ("KH", _i), ("KW", _i), ("Cout", _i),
                ("act", _i), ("epi", _i), ("aux0", _vp), ("aux1", _vp), ("aux0_stride", _i), ("aux1_stride", _i),
                ("out_scale", ctypes.c_float), ("shuffle2", _i), ("tile", _i), ("dtype", _i), ("korder", _i), ("stride", _i),
                ("ln_wsum", _vp), ("ln_eps", ctypes.c_float), ("ksplit", _i), ("bias2", _vp), ("pool2", _i), ("epi_cout0", _i)]


class ChainDesc(ctypes.Structure):
    """mirror of s2m2_chain_desc (include/s2m2_hip.h)"""
    _fields_ = [("x", _vp), ("res", _vp), ("out", _vp), ("x_stride", _ll), ("res_stride", _ll), ("out_stride", _ll), ("rows", _ll),
                ("C", _i), ("nstage", _i), ("weight", _vp * 3), ("bias", _vp * 3), ("ln_wsum", _vp * 3), ("act", _i * 3),
                ("res_stage", _i), ("carry", _i), ("ln_eps", ctypes.c_float), ("dtype", _i),
                ("ln_out", _vp), ("ln_out_stride", _ll), ("ln_gamma", _vp), ("ln_beta", _vp), ("ln_out_eps", ctypes.c_float),
                ("fan_weight", _vp), ("fan_bias", _vp), ("fan_ln_wsum", _vp), ("fan_out", _vp), ("fan_out_stride", _ll),
                ("nfan", _i), ("xcd_group_rows", _ll), ("weight_frag", _i), ("pool_h", _i), ("pool_w", _i)]


class RowAttnDesc(ctypes.Structure):
    """mirror of s2m2_rowattn_desc (include/s2m2_hip.h): K13, one 1-D attention step per launch"""
    _fields_ = [("x", _vp), ("x_stride", _ll), ("out", _vp), ("out_stride", _ll), ("nimg", _i), ("h", _i), ("w", _i), ("C", _i), ("heads", _i),
                ("cross", _i), ("weights", _vp), ("vectors", _vp), ("ln_eps", ctypes.c_float),
                ("ln_out", _vp), ("ln_out_stride", _ll), ("ln_out_eps", ctypes.c_float), ("xcd_hint", _i), ("dtype", _i)]


class ConvBlockDesc(ctypes.Structure):
    """mirror of s2m2_convblock_desc (include/s2m2_hip.h): K14, a whole ConvBlock2D per launch"""
    _fields_ = [("x", _vp), ("x_stride", _ll), ("out", _vp), ("out_stride", _ll), ("N", _i), ("H", _i), ("W", _i), ("C", _i),
                ("w_conv0", _vp), ("w_conv2", _vp), ("w_1x0", _vp), ("w_1x2", _vp), ("b_conv0", _vp), ("b_conv2", _vp), ("b_1x0", _vp),
                ("b_1x2", _vp), ("patch_rows", _i), ("dtype", _i)]


class PwDesc(ctypes.Structure):
    """mirror of s2m2_pw_desc (include/s2m2_hip.h)"""
    _fields_ = [("src", _vp * 4), ("src_c", _i * 4), ("src_stride", _ll * 4), ("nsrc", _i), ("rows", _ll), ("weight_frag", _vp), ("bias", _vp),
                ("out", _vp), ("out_stride", _ll), ("Cout", _i), ("act", _i), ("shuffle2", _i), ("Ho", _i), ("Wo", _i), ("dtype", _i)]


class NarrowDesc(ctypes.Structure):
    """mirror of s2m2_narrow_desc (include/s2m2_hip.h)"""
    _fields_ = [("x", _vp), ("x_stride", _ll), ("N", _i), ("H", _i), ("W", _i), ("Cin", _i), ("x1", _vp), ("x1_stride", _ll), ("Cin1", _i),
                ("weight_frag", _vp), ("bias", _vp), ("out", _vp), ("out_stride", _ll), ("Cout", _i), ("KH", _i), ("KW", _i), ("stride", _i),
                ("act", _i), ("dtype", _i), ("head_frag", _vp), ("head_bias", _vp), ("head_cout", _i)]


class CorrDesc(ctypes.Structure):
    """mirror of s2m2_corr_desc (include/s2m2_hip.h): every form of K1"""
    _fields_ = [("tokens", _vp), ("ln_weight", _vp), ("ln_bias", _vp), ("cv", _vp), ("B", _i), ("h", _i), ("w", _i), ("C", _i),
                ("cv_pitch", _i), ("band", _i), ("token_dtype", _i), ("cv_dtype", _i), ("start_event", _vp), ("stop_event", _vp)]


class PackDesc(ctypes.Structure):
    """mirror of s2m2_pack_desc (include/s2m2_hip.h): weight packing into the fragment orders of the direct-form kernels"""
    _fields_ = [("kind", _i), ("w", _vp), ("w2", _vp), ("rows", _i), ("cols", _i), ("ld", _i), ("ld2", _i), ("ntap", _i), ("out", _vp),
                ("out_elems", _ll)]


PACK_ROWS, PACK_NARROW, PACK_CONV_FRAG, PACK_FUSION, PACK_HEAD = range(5)


# name -> (restype, argtypes); must list every symbol declared in include/s2m2_hip.h
ABI_VERSION = 600                     # include/s2m2_hip.h: S2M2_ABI_VERSION (checked in load())

SIGNATURES = {
    "s2m2_version": (_i, []),
    "s2m2_last_error": (ctypes.c_char_p, []),
    "s2m2_ln_corr_kernel_name": (ctypes.c_char_p, [_i, _i, _i]),
    "s2m2_cost_volume": (_i, [ctypes.POINTER(CorrDesc), _vp]),
    "s2m2_plan_begin": (_i, [ctypes.POINTER(_vp)]),
    "s2m2_plan_end": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_size_t), _i]),
    "s2m2_plan_abort": (_i, [_vp]),
    "s2m2_plan_launches": (_i, [_vp]),
    "s2m2_plan_patches": (_i, [_vp, _i]),
    "s2m2_plan_run": (_i, [_vp, ctypes.POINTER(_vp), _i, _vp]),
    "s2m2_plan_destroy": (_i, [_vp]),
    "s2m2_refine_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "s2m2_pack_frag_elems": (_ll, [ctypes.POINTER(PackDesc)]),
    "s2m2_pack_frag": (_i, [ctypes.POINTER(PackDesc), _vp]),
    "s2m2_event_create": (_i, [ctypes.POINTER(_vp)]),
    "s2m2_event_destroy": (_i, [_vp]),
    "s2m2_event_elapsed_us": (_i, [_vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    "s2m2_sinkhorn_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i, _i]),
    "s2m2_pw_direct_supported": (_i, [_i, _i, _i]),
    "s2m2_pw_direct": (_i, [ctypes.POINTER(PwDesc), _vp]),
    "s2m2_conv_narrow_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "s2m2_conv_narrow": (_i, [ctypes.POINTER(NarrowDesc), _vp]),
    "s2m2_sinkhorn_regress": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "s2m2_cv_lookup": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _ll, _ll, _ll, _i, _vp]),
    "s2m2_conv2d": (_i, [ctypes.POINTER(ConvDesc), _vp]),
    "s2m2_mlp_chain_supported": (_i, [_i, _i]),
    "s2m2_mlp_chain_frag_supported": (_i, [_i, _i]),
    "s2m2_mlp_fan_supported": (_i, [_i, _i, _i]),
    "s2m2_mlp_chain": (_i, [ctypes.POINTER(ChainDesc), _vp]),
    "s2m2_conv_block_supported": (_i, [_i, _i, _i, _i]),
    "s2m2_conv_block": (_i, [ctypes.POINTER(ConvBlockDesc), _vp]),
    "s2m2_row_attn_supported": (_i, [_i, _i, _i, _i]),
    "s2m2_row_attn": (_i, [ctypes.POINTER(RowAttnDesc), _vp]),
    "s2m2_feature_fusion_supported": (_i, [_i, _i]),
    "s2m2_feature_fusion": (_i, [_vp, _vp, _vp, _ll, _ll, _ll, _ll, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "s2m2_feature_fusion_frag_supported": (_i, [_i, _i]),
    "s2m2_feature_fusion_frag": (_i, [_vp, _vp, _vp, _ll, _ll, _ll, _ll, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "s2m2_convex_upsample": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _ll, _i, _vp]),
    "s2m2_attention": (_i, [_vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _i, ctypes.c_float, _i, _vp, _vp, _vp, _ll,
                            _i, _i, _i, _vp]),
    "s2m2_attention_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "s2m2_resample2x": (_i, [_vp, _vp, _i, _i, _i, _i, _ll, _ll, _i, _i, _vp]),
    "s2m2_image_prep": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "s2m2_refine_prep": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "s2m2_global_update": (_i, [_vp, _i, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "s2m2_refine_update": (_i, [_vp, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "s2m2_debug_poison_lds": (_i, [_vp]),
    "s2m2_debug_clock_probe": (_i, [_vp, _vp]),
    "s2m2_refine_update_to": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp]),
    "s2m2_tanh": (_i, [_vp, _vp, _ll, _i, _vp]),
    "s2m2_stem_mlp": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp]),
    "s2m2_image_pad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "s2m2_layernorm": (_i, [_vp, _vp, _ll, _i, _ll, _ll, _i, _vp]),
    "s2m2_groupnorm_workspace_bytes": (ctypes.c_size_t, [_i, _i]),
    "s2m2_groupnorm_nhwc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _i, ctypes.c_float, _i, _vp]),
}


def load() -> ctypes.CDLL:
    """dlopen the kernel library (once).  Raises if it has not been built (python -m s2m2_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m s2m2_amd.build` "
                               "(the S2M2 hot path has no PyTorch fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        ver = lib.s2m2_version()
        if ver != ABI_VERSION:                        # exact: descriptor layouts change in place between patch versions as well
            raise RuntimeError(f"{LIB_PATH} reports ABI version {ver}, this binding was written against {ABI_VERSION} "
                               "(include/s2m2_hip.h: S2M2_ABI_VERSION): rebuild the library with `python -m s2m2_amd.build`")
        _lib = lib
    return _lib


# Optional work meter (bench.py): family -> [flops, launches] of the multiply-accumulate work the launches execute (padded channel
# counts, 2 flops per MAC); ATTN_EVENTS: list collecting (start event, end event, flops, shape tag) around every K4 launch.
METER: Optional[dict] = None
ATTN_EVENTS: Optional[list] = None
ROW_EVENTS: Optional[list] = None     # like ATTN_EVENTS around every K13 launch: (start, end, flops, unique HBM bytes, shape tag)


def _meter(family: str, flops: float) -> None:
    if METER is not None:
        e = METER.setdefault(family, [0.0, 0])
        e[0] += flops
        e[1] += 1


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().s2m2_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError("s2m2_amd.hip: tensors must be contiguous device tensors")


def pack_frag(kind: int, w: torch.Tensor, w2: Optional[torch.Tensor] = None, ntap: int = 1, rows: Optional[int] = None) -> torch.Tensor:
    """s2m2_pack_frag: the plain packing ``w`` (rows, K) fp16 on the device -> flat fp16 tensor holding the fragment stream of the direct-form
    kernel named by ``kind`` (PACK_ROWS / PACK_NARROW / PACK_CONV_FRAG / PACK_FUSION / PACK_HEAD; include/s2m2_hip.h).  One-time set-up."""
    if w.dtype != torch.float16 or not w.is_cuda or w.dim() != 2 or w.stride(1) != 1:
        raise ValueError("pack_frag: w must be a 2-D fp16 device tensor with contiguous rows")
    d = PackDesc()
    d.kind, d.w, d.rows, d.cols, d.ld, d.ntap = kind, w.data_ptr(), rows if rows is not None else w.shape[0], w.shape[1], w.stride(0), ntap
    if w2 is not None:
        if w2.dtype != torch.float16 or not w2.is_cuda or w2.dim() != 2 or w2.stride(1) != 1:
            raise ValueError("pack_frag: w2 must be a 2-D fp16 device tensor with contiguous rows")
        d.w2, d.ld2 = w2.data_ptr(), w2.stride(0)
    lib = load()
    n = lib.s2m2_pack_frag_elems(ctypes.byref(d))
    if n < 0:
        raise RuntimeError(f"s2m2_pack_frag_elems failed: {lib.s2m2_last_error().decode()}")
    out = torch.empty(n, device=w.device, dtype=torch.float16)
    d.out, d.out_elems = out.data_ptr(), n
    with torch.cuda.device(w.device):
        _check(lib.s2m2_pack_frag(ctypes.byref(d), _stream()), "s2m2_pack_frag")
    return out


class Plan:
    """A recorded launch plan (s2m2_plan_*): ``with Plan.record(externals) as p: ...`` records every library call this thread makes inside the
    block (they run as always); afterwards ``p.run(externals)`` re-issues the sequence natively with the external tensors somewhere else.
    externals: list of tensors (or None).  The tensors allocated inside the block must be kept alive by the caller (engine: a private MemPool)."""

    def __init__(self):
        self.h = _vp()
        self.n = 0
        self.sealed = False

    class _Rec:
        def __init__(self, plan, externals):
            self.plan, self.ext = plan, externals

        def __enter__(self):
            _check(load().s2m2_plan_begin(ctypes.byref(self.plan.h)), "s2m2_plan_begin")
            return self.plan

        def __exit__(self, et, ev, tb):
            lib = load()
            if et is not None:
                lib.s2m2_plan_abort(self.plan.h)
                return False
            n = len(self.ext)
            base = (_vp * max(n, 1))(*[(t.data_ptr() if t is not None else None) for t in self.ext])
            size = (ctypes.c_size_t * max(n, 1))(*[(t.numel() * t.element_size() if t is not None else 0) for t in self.ext])
            _check(lib.s2m2_plan_end(self.plan.h, base, size, n), "s2m2_plan_end")
            self.plan.n, self.plan.sealed = n, True
            return False

    def record(self, externals):
        return Plan._Rec(self, externals)

    @property
    def launches(self) -> int:
        return load().s2m2_plan_launches(self.h)

    def patches(self, slot: int = -1) -> int:
        return load().s2m2_plan_patches(self.h, slot)

    def run(self, externals) -> None:
        n = len(externals)
        ptrs = (_vp * max(n, 1))(*[(t.data_ptr() if t is not None else None) for t in externals])
        _check(load().s2m2_plan_run(self.h, ptrs, n, _stream()), "s2m2_plan_run")

    def refine_step(self, hidden, ctx, disp, conf, occ, cv, side) -> None:
        """s2m2_refine_step: this plan as one refinement iteration, externals in the ABI's fixed order"""
        p = [t.data_ptr() if t is not None else None for t in (hidden, ctx, disp, conf, occ, cv, side)]
        _check(load().s2m2_refine_step(self.h, *p, _stream()), "s2m2_refine_step")

    def __del__(self):
        try:
            if self.h:
                load().s2m2_plan_destroy(self.h)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class KernelTimer:
    """A start / stop HIP event pair attached to ONE kernel dispatch (s2m2_corr_desc.start_event / stop_event): elapsed_us() is the kernel's own execution
    time, what a rocprofv3 kernel trace reports, without the dispatch gaps that events recorded around a launch include."""

    def __init__(self):
        self.start, self.stop = _vp(), _vp()
        _check(load().s2m2_event_create(ctypes.byref(self.start)), "s2m2_event_create")
        _check(load().s2m2_event_create(ctypes.byref(self.stop)), "s2m2_event_create")

    def elapsed_us(self) -> float:
        us = ctypes.c_float()
        _check(load().s2m2_event_elapsed_us(self.start, self.stop, ctypes.byref(us)), "s2m2_event_elapsed_us")
        return float(us.value)

    def __del__(self):
        try:
            load().s2m2_event_destroy(self.start)
            load().s2m2_event_destroy(self.stop)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def _cost_volume(tokens: torch.Tensor, ln, cv: torch.Tensor, timer, band: int, what: str) -> None:
    """one launch of K1 through its descriptor (s2m2_cost_volume); ln = (weight, bias) fp32 or None (tokens normalised already)"""
    twoB, h, w, C = tokens.shape
    d = CorrDesc()
    d.tokens, d.cv = tokens.data_ptr(), cv.data_ptr()
    keep = None
    if ln is not None:
        keep = (ln[0].float().contiguous(), ln[1].float().contiguous())
        d.ln_weight, d.ln_bias = keep[0].data_ptr(), keep[1].data_ptr()
    d.B, d.h, d.w, d.C = twoB // 2, h, w, C
    d.cv_pitch, d.band = _cv_pitch(cv, what), band if band >= 0 else -1
    d.token_dtype, d.cv_dtype = _DT[tokens.dtype], _DT[cv.dtype]
    if timer is not None:
        d.start_event, d.stop_event = timer.start, timer.stop
    _check(load().s2m2_cost_volume(ctypes.byref(d), _stream()), "s2m2_cost_volume")
    _meter("ln_corr", 2.0 * (twoB // 2) * h * w * w * C)


def ln_corr(feat: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, cv_dtype: Optional[torch.dtype] = None,
            out: Optional[torch.Tensor] = None, timer: Optional[KernelTimer] = None, band: int = -1) -> torch.Tensor:
    """feat (2B,h,w,C) channels-last tokens (left = first B) -> cv (B,h,w,w), LayerNorm inside the kernel.  [A4]   timer: see KernelTimer.
    band >= 0: only columns j <= i + band are written, the rest of ``cv`` keeps whatever it held.  ``out`` may be a row-padded view (cv_alloc)."""
    _dev(feat, ln_w, ln_b)
    twoB, h, w, C = feat.shape
    B = twoB // 2
    cv_dtype = cv_dtype or feat.dtype
    cv = out if out is not None else torch.empty((B, h, w, w), device=feat.device, dtype=cv_dtype)
    if tuple(cv.shape) != (B, h, w, w):
        raise ValueError("ln_corr: out must be a (B,h,w,w) tensor")
    _cost_volume(feat, (ln_w, ln_b), cv, timer, band, "ln_corr")
    return cv


def _cv_pitch(cv: torch.Tensor, what: str) -> int:
    """(B,h,w,w) cost volume, columns contiguous, volume rows ``pitch`` elements apart (a padded allocation viewed ``[..., :w]``, or
    plain contiguous: pitch = w), image rows and batch entries dense behind that."""
    if cv.dim() != 4 or not cv.is_cuda or cv.shape[2] != cv.shape[3]:
        raise ValueError(f"{what}: cv must be a (B,h,w,w) device tensor, got {tuple(cv.shape)}")
    B, h, w, _ = cv.shape
    pitch = cv.stride(2)
    if cv.stride(3) != 1 or pitch < w or pitch % 8 or (h > 1 and cv.stride(1) != w * pitch) or (B > 1 and cv.stride(0) != h * w * pitch):
        raise ValueError(f"{what}: cv strides {cv.stride()} are not a row-padded (B,h,w,w) volume")
    return pitch


def cv_alloc(B: int, h: int, w: int, dtype: torch.dtype, device, aligned: bool = True) -> torch.Tensor:
    """(B,h,w,w) cost volume whose rows start on 128-byte lines: allocated (B,h,w,pitch) with pitch = w rounded up to 128 bytes and
    returned as the ``[..., :w]`` view (K1's 64-column store granules are then whole lines: no partial-line writes)."""
    per = 128 // (2 if dtype == torch.float16 else 4)
    pitch = (w + per - 1) // per * per if aligned else w
    return torch.empty((B, h, w, pitch), device=device, dtype=dtype)[..., :w]


def corr(tokens: torch.Tensor, cv_dtype: Optional[torch.dtype] = None, out: Optional[torch.Tensor] = None,
         timer: Optional[KernelTimer] = None, band: int = -1) -> torch.Tensor:
    """tokens (2B,h,w,C) ALREADY LayerNorm'ed (left = first B) -> cv (B,h,w,w), a row-padded view (cv_alloc) unless ``out`` is given.
    [A4 without the LayerNorm: s2m2_cost_volume with ln_weight = NULL]"""
    _dev(tokens)
    twoB, h, w, C = tokens.shape
    B = twoB // 2
    cv = out if out is not None else cv_alloc(B, h, w, cv_dtype or tokens.dtype, tokens.device)
    if tuple(cv.shape) != (B, h, w, w):
        raise ValueError("corr: out must be a (B,h,w,w) tensor")
    _cost_volume(tokens, None, cv, timer, band, "corr")
    return cv


def sinkhorn_regress(cv: torch.Tensor, use_positivity: bool, ot_iter: int = 3, want_argmax: bool = False):
    """cv (B,h,w,w) (row-padded views accepted) -> disp, conf, occ (B,1,h,w) fp32 [, argmax (B,h,w) int32].  [A5+A6]"""
    pitch = _cv_pitch(cv, "sinkhorn_regress")
    B, h, w, _ = cv.shape
    out = torch.empty((3, B, 1, h, w), device=cv.device, dtype=torch.float32)
    am = torch.empty((B, h, w), device=cv.device, dtype=torch.int32) if want_argmax else None
    _check(load().s2m2_sinkhorn_regress(cv.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                        am.data_ptr() if am is not None else None, B, h, w, ot_iter, int(use_positivity),
                                        _DT[cv.dtype], pitch, None, _stream()), "s2m2_sinkhorn_regress")
    return (out[0], out[1], out[2], am) if want_argmax else (out[0], out[1], out[2])


def cv_lookup(cv: torch.Tensor, disp: torch.Tensor, radius: int = 4, channels_last: bool = False,
              out_dtype: torch.dtype = torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """cv (B,h,w,w), disp (B,1,h,w) fp32 -> corr1, corr2: (B,2r+1,h,w) planar or (B,h,w,2r+1) channels-last.  [A9+A10]"""
    _dev(disp)
    pitch = _cv_pitch(cv, "cv_lookup")
    B, h, w, _ = cv.shape
    T = 2 * radius + 1
    if channels_last:
        both = torch.empty((B, h, w, 2 * T), device=cv.device, dtype=out_dtype)
        c1, c2 = both[..., :T], both[..., T:]
        bs, ps, ts = h * w * 2 * T, 2 * T, 1
    else:
        both = torch.empty((2, B, T, h, w), device=cv.device, dtype=out_dtype)
        c1, c2 = both[0], both[1]
        bs, ps, ts = T * h * w, 1, h * w
    _check(load().s2m2_cv_lookup(cv.data_ptr(), disp.float().data_ptr(), c1.data_ptr(), c2.data_ptr(), B, h, w, radius,
                                 _DT[cv.dtype], _DT[out_dtype], bs, ps, ts, pitch, _stream()), "s2m2_cv_lookup")
    return c1, c2


def _nhwc(t: torch.Tensor):
    """(N,H,W,C) view, channels contiguous, pixels dense with a common pixel stride (a channel slice of a wider tensor is fine)."""
    if t.dim() != 4 or not t.is_cuda or t.stride(3) != 1:
        raise ValueError(f"s2m2_amd.hip: expected an (N,H,W,C) device tensor with contiguous channels, got {tuple(t.shape)} {t.stride()}")
    n, h, w, c = t.shape
    ps = t.stride(2)
    if (h > 1 and t.stride(1) != w * ps) or (n > 1 and t.stride(0) != h * w * ps):
        raise ValueError(f"s2m2_amd.hip: pixels are not dense: shape {tuple(t.shape)} strides {t.stride()}")
    return ps


def conv2d(srcs, weight: torch.Tensor, bias: Optional[torch.Tensor], KH: int, KW: int, Cout: int, act: int = ACT_NONE,
           epi: int = EPI_NONE, aux0: Optional[torch.Tensor] = None, aux1: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, out_scale: float = 1.0, shuffle2: int = 0, tile: int = 0,
           stride: int = 1, korder: int = 0, ln_wsum: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
           ksplit: int = 0, bias2: Optional[torch.Tensor] = None, pool2: bool = False, epi_cout0: int = 0) -> torch.Tensor:
    """Implicit-GEMM convolution / linear layer (s2m2_conv2d).  srcs: list of (N,H,W,Cs) tensors, concatenated along C;
    weight: packed (Cout, KH*KW*sum(Cs)) (see s2m2_amd.pack); bias fp32 (Cout) or None.  Returns (N,H,W,Cout), or
    (N,2H,2W,shuffle2) for the 2x2-stride-2 transposed-conv GEMM.  ln_wsum (fp32 (Cout) row sums of the packed weight): the 1x1 layer
    is preceded by LayerNorm(Cin, no affine, eps=ln_eps) of the raw input rows, folded into the kernel.  pool2: the 1x1 layer is preceded
    by AvgPool2d(2), folded into its operand load -> (N, H//2, W//2, Cout).  epi_cout0 > 0: the one-operand epilogue applies to couts
    >= epi_cout0 only and aux0 has Cout - epi_cout0 channels (two stacked layers, one launch: s2m2_conv_desc.epi_cout0)."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    d = ConvDesc()
    x0 = srcs[0]
    n, h, w, _ = x0.shape
    dt = x0.dtype
    cin = 0
    for i, t in enumerate(srcs):
        if t.dtype != dt or tuple(t.shape[:3]) != (n, h, w):
            raise ValueError("conv2d: sources must share dtype and (N,H,W)")
        d.src[i] = t.data_ptr()
        d.src_c[i] = t.shape[3]
        d.src_stride[i] = _nhwc(t)
        cin += t.shape[3]
    ck = 192 if (Cout % 128 != 0 and Cout % 192 == 0 and cin % 192 == 0) else 128        # s2m2_conv_frag_chunk
    kcols = KH * KW * (-(-cin // ck) * ck) if korder == 2 else KH * KW * cin           # K order 2: K padded to whole chunks
    if weight.dtype != dt or not weight.is_contiguous() or tuple(weight.shape) != (Cout, kcols):
        raise ValueError(f"conv2d: packed weight must be {(Cout, kcols)} {dt}, got {tuple(weight.shape)} {weight.dtype}")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != Cout or not bias.is_contiguous()):
        raise ValueError("conv2d: bias must be fp32 (Cout)")
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    if pool2:
        ho, wo = h // 2, w // 2
    exp_shape = (n, 2 * h, 2 * w, shuffle2) if shuffle2 else (n, ho, wo, Cout)
    if out is None:
        out = torch.empty(exp_shape, device=x0.device, dtype=dt)
    if tuple(out.shape) != exp_shape or out.dtype != dt:
        raise ValueError(f"conv2d: out must be {exp_shape} {dt}, got {tuple(out.shape)} {out.dtype}")
    d.nsrc = len(srcs)
    d.weight = weight.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.out = out.data_ptr()
    d.out_stride = _nhwc(out)
    d.N, d.H, d.W, d.KH, d.KW, d.Cout = n, h, w, KH, KW, Cout
    d.act, d.epi = act, epi
    for name, a in (("aux0", aux0), ("aux1", aux1)):
        if a is not None:
            ac = Cout - epi_cout0 if name == "aux0" else Cout
            if a.dtype != dt or tuple(a.shape) != (n, ho, wo, ac):
                raise ValueError(f"conv2d: {name} must be {(n, ho, wo, ac)} {dt}, got {tuple(a.shape)} {a.dtype}")
            setattr(d, name, a.data_ptr())                       # (aux0 of an epi_cout0 launch: its own base; the library applies the cout offset)
            setattr(d, name + "_stride", _nhwc(a))
    d.epi_cout0 = epi_cout0
    d.out_scale = out_scale
    d.shuffle2 = shuffle2
    d.tile = tile
    d.stride = stride
    d.korder = korder
    d.pool2 = int(pool2)
    if ln_wsum is not None:
        if ln_wsum.dtype != torch.float32 or ln_wsum.numel() != Cout or not ln_wsum.is_contiguous():
            raise ValueError("conv2d: ln_wsum must be fp32 (Cout)")
        d.ln_wsum = ln_wsum.data_ptr()
        d.ln_eps = ln_eps
    if epi == EPI_DUALMIX:
        if bias2 is not None and (bias2.dtype != torch.float32 or bias2.numel() != Cout or not bias2.is_contiguous()):
            raise ValueError("conv2d: bias2 must be fp32 (Cout)")
        d.ksplit = ksplit
        d.bias2 = bias2.data_ptr() if bias2 is not None else None
    d.dtype = _DT[dt]
    _check(load().s2m2_conv2d(ctypes.byref(d), _stream()), "s2m2_conv2d")
    _meter("conv2d", 2.0 * n * ho * wo * Cout * KH * KW * cin)
    return out


def _token_rows(x: torch.Tensor, what: str):
    """(rows, row stride) of a (..., C) tensor with contiguous channels and a uniform row stride"""
    C = x.shape[-1]
    if x.stride(-1) != 1:
        raise ValueError(f"{what}: channels must be contiguous")
    xs = x.stride(-2) if x.dim() > 1 else C
    for d in range(x.dim() - 2):
        if x.shape[d] > 1 and x.stride(d) != x.stride(d + 1) * x.shape[d + 1]:
            raise ValueError(f"{what}: rows must have a uniform stride")
    return x.numel() // C, xs


def mlp_chain_supported(C: int, dtype: torch.dtype) -> bool:
    return bool(load().s2m2_mlp_chain_supported(C, _DT[dtype]))


def mlp_fan_supported(C: int, nfan: int, dtype: torch.dtype) -> bool:
    return bool(load().s2m2_mlp_fan_supported(C, nfan, _DT[dtype]))


def mlp_fan(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], ln_wsum: Optional[torch.Tensor], ln_eps: float = 1e-5,
            frag: bool = True, pool2: bool = False) -> torch.Tensor:
    """n stacked C -> C layers on the rows of x (..., C) -> (..., n*C) in one pass over the rows (s2m2_mlp_chain with nstage = 0: the
    fan-out-only launch of the direct form; pre-LayerNorm folded in when ln_wsum is given).  weight (n*C, C) in MFMA-fragment order
    (pack.chain_frag; mlp_fan_supported: fp16, C = 128 / 256, any row count, n <= 4).  pool2 (x (N,H,W,C)): nn.AvgPool2d(2) in front of
    the layers, folded into the tile load -> (N, H//2, W//2, n*C)."""
    if not frag:
        raise ValueError("mlp_fan: the fan-out-only launch exists in the direct form only (weight in fragment order, frag=True)")
    C = x.shape[-1]
    rows, xs = _token_rows(x, "mlp_fan")
    n = weight.shape[0] // C
    oshape = tuple(x.shape[:-1])
    if pool2:
        if not frag or x.dim() != 4 or x.shape[1] < 2 or x.shape[2] < 2:
            raise ValueError("mlp_fan: pool2 needs frag and an (N,H,W,C) tensor of at least 2x2 pixels")
        oshape = (x.shape[0], x.shape[1] // 2, x.shape[2] // 2)
        rows = oshape[0] * oshape[1] * oshape[2]
    if weight.dtype != x.dtype or tuple(weight.shape) != (n * C, C) or not weight.is_contiguous() or not weight.is_cuda or not x.is_cuda:
        raise ValueError(f"mlp_fan: weight must be a packed (n*{C}, {C}) {x.dtype} device matrix")
    for name, t in (("bias", bias), ("ln_wsum", ln_wsum)):
        if t is not None and (t.dtype != torch.float32 or t.numel() != n * C or not t.is_contiguous() or not t.is_cuda):
            raise ValueError(f"mlp_fan: {name} must be fp32 ({n * C}) on the device")
    out = torch.empty(oshape + (n * C,), device=x.device, dtype=x.dtype)
    d = ChainDesc()
    d.x, d.x_stride, d.rows, d.C, d.nstage, d.dtype, d.ln_eps = x.data_ptr(), xs, rows, C, 0, _DT[x.dtype], ln_eps
    d.res_stage = -1
    d.weight_frag = int(bool(frag))
    if pool2:
        d.pool_h, d.pool_w = x.shape[1], x.shape[2]
    d.fan_weight, d.fan_out, d.fan_out_stride, d.nfan = weight.data_ptr(), out.data_ptr(), n * C, n
    d.fan_bias = bias.data_ptr() if bias is not None else None
    d.fan_ln_wsum = ln_wsum.data_ptr() if ln_wsum is not None else None
    _check(load().s2m2_mlp_chain(ctypes.byref(d), _stream()), "s2m2_mlp_chain")
    _meter("mlp_chain", 2.0 * rows * C * C * n)
    return out


def mlp_chain_frag_supported(C: int, dtype: torch.dtype) -> bool:
    """the direct form of mlp_chain (weights in MFMA-fragment order, pack.chain_frag) exists for this width"""
    return bool(load().s2m2_mlp_chain_frag_supported(C, _DT[dtype]))


def mlp_chain_ln_out_supported(C: int, dtype: torch.dtype) -> bool:
    """the optional LayerNorm output of mlp_chain needs a row of 8 k <= 64 16-byte pieces, in either form of the kernel"""
    ppr = C * (2 if dtype == torch.float16 else 4) // 16
    return (mlp_chain_supported(C, dtype) or mlp_chain_frag_supported(C, dtype)) and ppr % 8 == 0 and ppr <= 64


def mlp_chain(x: torch.Tensor, stages, res: Optional[torch.Tensor] = None, res_stage: int = -1, carry: bool = False,
              ln_eps: float = 1e-5, ln_out: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None, xcd_group_rows: int = 0,
              fan=None, frag: bool = False, pool2: bool = False):
    """Up to three C -> C 1x1 layers on the rows of x (..., C) in one launch (s2m2_mlp_chain).  stages: list of
    (packed weight (C, C), fp32 bias (C) or None, activation, ln_wsum fp32 (C) or None = pre-LayerNorm of that stage's input);
    res (same shape as x) is added to the output of stage res_stage; carry adds the output of stage 0 to the last of 3 stages.
    ln_out = (gamma fp32 (C), beta fp32 (C), eps): also return LayerNorm(out) * gamma + beta  -> (out, normalised).
    xcd_group_rows: placement hint (s2m2_chain_desc): x is images of 8 groups of that many rows, group g runs on XCD g.
    fan = (packed weight (n*C, C), fp32 bias (n*C) or None, ln_wsum fp32 (n*C) or None): n further C -> C layers on the OUTPUT rows
    (pre-LayerNorm folded in when ln_wsum is given), returned as one (..., n*C) tensor: the fused QKV projection of the next attention.
    frag: every weight (stages and fan) is in MFMA-fragment order (pack.chain_frag) -> the direct form of the kernel, meant for short row
    counts (mlp_chain_frag_supported).  pool2 (frag, x (N,H,W,C), no res / carry / ln_out): nn.AvgPool2d(2) in front of the first stage,
    folded into the tile load -> outputs on the (N, H//2, W//2) grid.
    Return value: out, or a tuple (out[, normalised][, fan_out]) in that order."""
    C = x.shape[-1]
    rows, xs = _token_rows(x, "mlp_chain")
    oshape = tuple(x.shape[:-1])
    d = ChainDesc()
    if pool2:
        if not frag or x.dim() != 4 or x.shape[1] < 2 or x.shape[2] < 2 or res_stage >= 0 or carry or ln_out is not None:
            raise ValueError("mlp_chain: pool2 needs frag, an (N,H,W,C) tensor of at least 2x2 pixels and no res / carry / ln_out")
        oshape = (x.shape[0], x.shape[1] // 2, x.shape[2] // 2)
        rows = oshape[0] * oshape[1] * oshape[2]
        d.pool_h, d.pool_w = x.shape[1], x.shape[2]
    d.x, d.x_stride, d.rows, d.C, d.nstage, d.dtype = x.data_ptr(), xs, rows, C, len(stages), _DT[x.dtype]
    if not 1 <= len(stages) <= 3:
        raise ValueError("mlp_chain: 1..3 stages")
    keep = [x]
    for i, (w, b, act, wsum) in enumerate(stages):
        if w.dtype != x.dtype or tuple(w.shape) != (C, C) or not w.is_contiguous():
            raise ValueError(f"mlp_chain: weight[{i}] must be a packed ({C}, {C}) {x.dtype} matrix, got {tuple(w.shape)} {w.dtype}")
        for name, t in (("bias", b), ("ln_wsum", wsum)):
            if t is not None and (t.dtype != torch.float32 or t.numel() != C or not t.is_contiguous()):
                raise ValueError(f"mlp_chain: {name}[{i}] must be fp32 ({C})")
        d.weight[i], d.act[i] = w.data_ptr(), act
        d.bias[i] = b.data_ptr() if b is not None else None
        d.ln_wsum[i] = wsum.data_ptr() if wsum is not None else None
        keep += [w, b, wsum]
    if not all(t.is_cuda for t in keep if t is not None):
        raise ValueError("mlp_chain: tensors must live on the GPU")
    d.res_stage, d.carry, d.ln_eps = res_stage, int(carry), ln_eps
    d.xcd_group_rows = int(xcd_group_rows)
    d.weight_frag = int(bool(frag))
    if res_stage >= 0:
        if res is None or res.dtype != x.dtype or tuple(res.shape) != tuple(x.shape):
            raise ValueError("mlp_chain: res must match x")
        d.res, d.res_stride = res.data_ptr(), _token_rows(res, "mlp_chain")[1]
    out = torch.empty(oshape + (C,), device=x.device, dtype=x.dtype)
    d.out, d.out_stride = out.data_ptr(), C
    normed = None
    if ln_out is not None:
        gam, bet, eps = ln_out
        _dev(gam, bet)
        if gam.dtype != torch.float32 or bet.dtype != torch.float32 or gam.numel() != C or bet.numel() != C:
            raise ValueError(f"mlp_chain: ln_out gamma / beta must be fp32 ({C})")
        normed = torch.empty(x.shape, device=x.device, dtype=x.dtype)
        ln_ptr = normed.data_ptr()
        d.ln_out, d.ln_out_stride, d.ln_gamma, d.ln_beta, d.ln_out_eps = ln_ptr, C, gam.data_ptr(), bet.data_ptr(), float(eps)
    fan_out = None
    nfan = 0
    if fan is not None:
        fw, fb, fws = fan
        nfan = fw.shape[0] // C
        if fw.dtype != x.dtype or fw.dim() != 2 or fw.shape[1] != C or fw.shape[0] != nfan * C or not fw.is_contiguous() or not fw.is_cuda:
            raise ValueError(f"mlp_chain: fan weight must be a packed (n*{C}, {C}) {x.dtype} device matrix")
        for name, t in (("fan bias", fb), ("fan ln_wsum", fws)):
            if t is not None and (t.dtype != torch.float32 or t.numel() != nfan * C or not t.is_contiguous() or not t.is_cuda):
                raise ValueError(f"mlp_chain: {name} must be fp32 ({nfan * C}) on the device")
        fan_out = torch.empty(oshape + (nfan * C,), device=x.device, dtype=x.dtype)
        d.fan_weight, d.fan_out, d.fan_out_stride, d.nfan = fw.data_ptr(), fan_out.data_ptr(), nfan * C, nfan
        d.fan_bias = fb.data_ptr() if fb is not None else None
        d.fan_ln_wsum = fws.data_ptr() if fws is not None else None
    _check(load().s2m2_mlp_chain(ctypes.byref(d), _stream()), "s2m2_mlp_chain")
    _meter("mlp_chain", 2.0 * rows * C * C * (len(stages) + nfan))
    res_t = (out,) + ((normed,) if normed is not None else ()) + ((fan_out,) if fan_out is not None else ())
    return res_t[0] if len(res_t) == 1 else res_t


def conv_block_supported(C: int, H: int, W: int, dtype: torch.dtype) -> bool:
    """K14 (conv_block) takes a ConvBlock2D of this width on an H x W grid (fp16, C = 128 / 256, coarse grids)"""
    return bool(load().s2m2_conv_block_supported(C, H, W, _DT[dtype]))


def conv_block(x: torch.Tensor, w_conv0: torch.Tensor, b_conv0, w_conv2: torch.Tensor, b_conv2, w_1x0: torch.Tensor, b_1x0, w_1x2: torch.Tensor, b_1x2,
               patch_rows: int = 0) -> torch.Tensor:
    """K14: ConvBlock2D (attentions.py:255-281) on x (N,H,W,C) in one launch: convs.2(GELU(convs.0(x))) + convs_1x.2(ReLU(convs_1x.0(x))).
    w_conv0 / w_conv2: the 3x3 layers as K5 v5 fragment streams (pack.pack_conv_frag), w_1x0 / w_1x2: the 1x1 layers in K9's fragment order
    (pack.chain_frag); biases fp32 (C) or None."""
    if x.dim() != 4 or x.dtype != torch.float16 or not x.is_cuda or x.stride(3) != 1:
        raise ValueError("conv_block: x must be an (N,H,W,C) fp16 device tensor with contiguous channels")
    N, H, W, C = x.shape
    xs = x.stride(2)
    if x.stride(1) != W * xs or (N > 1 and x.stride(0) != H * W * xs):
        raise ValueError("conv_block: the pixels of x must be evenly strided")
    d = ConvBlockDesc()
    out = torch.empty((N, H, W, C), device=x.device, dtype=x.dtype)
    d.x, d.x_stride, d.out, d.out_stride, d.N, d.H, d.W, d.C = x.data_ptr(), xs, out.data_ptr(), C, N, H, W, C
    for name, w, n in (("w_conv0", w_conv0, 9 * C * C), ("w_conv2", w_conv2, 9 * C * C), ("w_1x0", w_1x0, C * C), ("w_1x2", w_1x2, C * C)):
        if w.dtype != x.dtype or w.numel() != n or not w.is_contiguous() or not w.is_cuda:
            raise ValueError(f"conv_block: {name} must be a contiguous fp16 device tensor of {n} elements")
        setattr(d, name, w.data_ptr())
    for name, b in (("b_conv0", b_conv0), ("b_conv2", b_conv2), ("b_1x0", b_1x0), ("b_1x2", b_1x2)):
        if b is not None and (b.dtype != torch.float32 or b.numel() != C or not b.is_contiguous() or not b.is_cuda):
            raise ValueError(f"conv_block: {name} must be fp32 ({C}) on the device or None")
        setattr(d, name, b.data_ptr() if b is not None else None)
    d.patch_rows, d.dtype = patch_rows, _DT[x.dtype]
    _check(load().s2m2_conv_block(ctypes.byref(d), _stream()), "s2m2_conv_block")
    _meter("conv_block", 2.0 * N * H * W * C * C * 20)
    return out


def row_attn_supported(C: int, heads: int, w: int, dtype: torch.dtype) -> bool:
    """K13 (row_attn) exists for rows of w tokens x C channels with this many heads (fp16, C = 128, 1 or 2 heads, w <= 320)"""
    return bool(load().s2m2_row_attn_supported(C, heads, w, _DT[dtype]))


def row_attn(x: torch.Tensor, heads: int, cross: bool, weights: torch.Tensor, vectors: torch.Tensor, ln_eps: float = 1e-5,
             ln_out_eps: Optional[float] = None, xcd_hint: bool = True):
    """K13: one 1-D attention step of BasicAttnBlock (attentions.py:347-355) on the token rows of x (nimg, h, w, 128) in one launch:
    z' = z + proj(attention(LN(z), LN(s)));  out = z' + ffn(LN(z'))  with s = the same line of image (n + nimg/2) % nimg (cross) or z itself.
    weights: (6 * 128, 128) fp16 -- q, k, v, proj, ffn.0, ffn.2 in the row_attn packing (pack.rowattn_pack); vectors: (12, 128) fp32 --
    bias q, row sums q, bias k, row sums k, bias v, row sums v, bias proj, bias ffn.0, row sums ffn.0, bias ffn.2, ln_out gamma, ln_out beta
    (pack.rowattn_vectors).  ln_out_eps: also return LayerNorm(out) * gamma + beta with that eps -> (out, normalised)."""
    if x.dim() != 4 or x.dtype != torch.float16 or not x.is_cuda or x.stride(3) != 1:
        raise ValueError("row_attn: x must be an (nimg, h, w, C) fp16 device tensor with contiguous channels")
    nimg, h, w, C = x.shape
    xs = x.stride(2)
    if x.stride(1) != w * xs or (nimg > 1 and x.stride(0) != h * w * xs):
        raise ValueError("row_attn: the tokens of x must be evenly strided")
    if weights.dtype != x.dtype or tuple(weights.shape) != (6 * C, C) or not weights.is_contiguous() or not weights.is_cuda:
        raise ValueError(f"row_attn: weights must be a contiguous ({6 * C}, {C}) {x.dtype} device tensor (pack.rowattn_pack)")
    if vectors.dtype != torch.float32 or tuple(vectors.shape) != (12, C) or not vectors.is_contiguous() or not vectors.is_cuda:
        raise ValueError(f"row_attn: vectors must be a contiguous (12, {C}) fp32 device tensor (pack.rowattn_vectors)")
    d = RowAttnDesc()
    out = torch.empty((nimg, h, w, C), device=x.device, dtype=x.dtype)
    d.x, d.x_stride, d.out, d.out_stride = x.data_ptr(), xs, out.data_ptr(), C
    d.nimg, d.h, d.w, d.C, d.heads, d.cross, d.ln_eps, d.dtype = nimg, h, w, C, heads, int(bool(cross)), ln_eps, _DT[x.dtype]
    d.xcd_hint = int(bool(xcd_hint))
    d.weights, d.vectors = weights.data_ptr(), vectors.data_ptr()
    normed = None
    if ln_out_eps is not None:
        normed = torch.empty((nimg, h, w, C), device=x.device, dtype=x.dtype)
        d.ln_out, d.ln_out_stride, d.ln_out_eps = normed.data_ptr(), C, float(ln_out_eps)
    ev = None
    if ROW_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _check(load().s2m2_row_attn(ctypes.byref(d), _stream()), "s2m2_row_attn")
    rows = nimg * h * w
    flops = 2.0 * rows * C * C * 6 + 4.0 * nimg * h * w * w * C
    if ev is not None:
        ev[1].record()
        ROW_EVENTS.append((ev[0], ev[1], flops, 2.0 * rows * C * (3 if normed is not None else 2),      # unique bytes: rows read once, written once (+ ln_out)
                           f"({nimg},{h},{w},{C}) heads {heads} {'cross' if cross else 'self'}"))
    _meter("row_attn", flops)
    return out if normed is None else (out, normed)


def pw_direct_supported(K: int, Cout: int, dtype: torch.dtype) -> bool:
    """K11 exists for this layer shape (K = concatenated input channels, multiple of 8)"""
    return bool(load().s2m2_pw_direct_supported(K, Cout, _DT[dtype]))


def pw_direct(srcs, weight_frag: torch.Tensor, bias: Optional[torch.Tensor], Cout: int, act: int = ACT_NONE, shuffle2: int = 0) -> torch.Tensor:
    """K11: a 1x1 layer on the channel concatenation of ``srcs`` ((N,H,W,Ci) / (..., Ci) tensors with contiguous channels and the same leading
    shape), weight in the fragment order of pack.pw_frag, fp32 bias (Cout) or None -> (..., Cout).  shuffle2 = C' > 0: the ConvTranspose2d(2,
    stride 2) store, Cout = 4 * C' -> (N, 2H, 2W, C')."""
    x0 = srcs[0]
    d = PwDesc()
    d.nsrc = len(srcs)
    K = 0
    rows = None
    for i, x in enumerate(srcs):
        r, xs = _token_rows(x, "pw_direct")
        if rows is not None and r != rows:
            raise ValueError("pw_direct: sources must have the same number of rows")
        rows = r
        if x.dtype != x0.dtype or not x.is_cuda:
            raise ValueError("pw_direct: sources must be device tensors of one dtype")
        d.src[i], d.src_c[i], d.src_stride[i] = x.data_ptr(), x.shape[-1], xs
        K += x.shape[-1]
    if weight_frag.dtype != x0.dtype or not weight_frag.is_cuda or not weight_frag.is_contiguous() or weight_frag.dim() != 4 or \
            tuple(weight_frag.shape[2:]) != (64, 8) or weight_frag.shape[0] != (Cout + 31) // 32 or weight_frag.shape[1] != (K + 15) // 16:
        raise ValueError(f"pw_direct: weight must be pack.pw_frag of a ({Cout}, {K}) matrix, got {tuple(weight_frag.shape)} {weight_frag.dtype}")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() < Cout or not bias.is_cuda):
        raise ValueError(f"pw_direct: bias must be fp32 ({Cout}) on the device")
    if shuffle2:
        if x0.dim() != 4 or Cout != 4 * shuffle2:
            raise ValueError("pw_direct: shuffle2 needs (N,H,W,C) sources and Cout = 4 * shuffle2")
        out = torch.empty((x0.shape[0], 2 * x0.shape[1], 2 * x0.shape[2], shuffle2), device=x0.device, dtype=x0.dtype)
        d.shuffle2, d.Ho, d.Wo, d.out_stride = shuffle2, x0.shape[1], x0.shape[2], shuffle2
    else:
        out = torch.empty(tuple(x0.shape[:-1]) + (Cout,), device=x0.device, dtype=x0.dtype)
        d.out_stride = Cout
    d.rows, d.weight_frag, d.bias, d.out = rows, weight_frag.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr()
    d.Cout, d.act, d.dtype = Cout, act, _DT[x0.dtype]
    _check(load().s2m2_pw_direct(ctypes.byref(d), _stream()), "s2m2_pw_direct")
    _meter("conv2d", 2.0 * rows * K * Cout)                          # (as K5 counts the same layer: no MFMA-tile padding)
    return out


def conv_narrow_supported(KH: int, KW: int, stride: int, Cin: int, Cout: int, dtype: torch.dtype) -> bool:
    """K12 exists for this layer shape (Cin = ALL input channels of the one or two sources)"""
    return bool(load().s2m2_conv_narrow_supported(KH, KW, stride, Cin, Cout, _DT[dtype]))


def conv_narrow(srcs, weight_frag: torch.Tensor, bias: Optional[torch.Tensor], KH: int, KW: int, Cout: int, stride: int = 1,
                act: int = ACT_NONE, head=None) -> torch.Tensor:
    """K12: a KH x KW convolution (padding K // 2) in the pixel-split direct form on one (N,H,W,Cin) tensor -- or the channel concatenation
    of two -- weight = pack.narrow_frag of the K-order-0 matrix (Cout, KH*KW*Cin), fp32 bias (Cout) or None
    -> (N, ceil(H/stride), ceil(W/stride), Cout).  Shapes: conv_narrow_supported.
    head = (pack.head_frag of a (Cout2, Cout) 1x1 weight, fp32 bias (Cout2) or None, Cout2): that 1x1 layer applied to the activated output
    inside the same launch (Cin = 48 form) -> (..., Cout2); the Cout-channel tensor is not produced."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    if not 1 <= len(srcs) <= 2 or any(t.shape[:3] != srcs[0].shape[:3] or t.dtype != srcs[0].dtype for t in srcs):
        raise ValueError("conv_narrow: one or two (N,H,W,C) sources of the same grid and dtype")
    x = srcs[0]
    xs = _nhwc(x)
    n, h, w, _ = x.shape
    cin = sum(t.shape[-1] for t in srcs)
    K = KH * KW * cin
    if weight_frag.dtype != x.dtype or not weight_frag.is_cuda or not weight_frag.is_contiguous() or weight_frag.dim() != 4 or \
            tuple(weight_frag.shape[2:]) != (64, 8) or weight_frag.shape[0] != (Cout + 31) // 32 or weight_frag.shape[1] != (K + 15) // 16:
        raise ValueError(f"conv_narrow: weight must be pack.narrow_frag of a ({Cout}, {K}) matrix, got {tuple(weight_frag.shape)} {weight_frag.dtype}")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() < Cout or not bias.is_cuda):
        raise ValueError(f"conv_narrow: bias must be fp32 ({Cout}) on the device")
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    cout_final = Cout
    if head is not None:
        hf, hb, cout_final = head
        if hf.dtype != x.dtype or not hf.is_cuda or not hf.is_contiguous() or tuple(hf.shape) != (1, 2 * ((Cout + 31) // 32), 64, 8):
            raise ValueError(f"conv_narrow: head weight must be pack.head_frag of a (Cout2, {Cout}) matrix, got {tuple(hf.shape)} {hf.dtype}")
        if hb is not None and (hb.dtype != torch.float32 or hb.numel() < cout_final or not hb.is_cuda):
            raise ValueError(f"conv_narrow: head bias must be fp32 ({cout_final}) on the device")
    out = torch.empty((n, ho, wo, cout_final), device=x.device, dtype=x.dtype)
    d = NarrowDesc()
    if head is not None:
        d.head_frag, d.head_bias, d.head_cout = hf.data_ptr(), hb.data_ptr() if hb is not None else None, cout_final
    d.x, d.x_stride, d.N, d.H, d.W, d.Cin = x.data_ptr(), xs, n, h, w, cin
    if len(srcs) == 2:
        d.x1, d.x1_stride, d.Cin1 = srcs[1].data_ptr(), _nhwc(srcs[1]), srcs[1].shape[-1]
    d.weight_frag, d.bias, d.out, d.out_stride = weight_frag.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), cout_final
    d.Cout, d.KH, d.KW, d.stride, d.act, d.dtype = Cout, KH, KW, stride, act, _DT[x.dtype]
    _check(load().s2m2_conv_narrow(ctypes.byref(d), _stream()), "s2m2_conv_narrow")
    _meter("conv2d", 2.0 * n * ho * wo * (K * Cout + (Cout * cout_final if head is not None else 0)))   # (as K5 counts the same layers: no MFMA-tile padding)
    return out


def feature_fusion_supported(C: int, dtype: torch.dtype) -> bool:
    return bool(load().s2m2_feature_fusion_supported(C, _DT[dtype]))


def feature_fusion_frag_supported(C: int, dtype: torch.dtype) -> bool:
    """the direct form of K10 (weights as a fragment stream, pack.fusion_frag) exists for this width"""
    return bool(load().s2m2_feature_fusion_frag_supported(C, _DT[dtype]))


def feature_fusion(z0: torch.Tensor, z1: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: Optional[torch.Tensor], bg: torch.Tensor,
                   bf: torch.Tensor, z1_coarse: bool = False, frag: bool = False) -> torch.Tensor:
    """FeatureFusion with 1x1 kernels in one launch (s2m2_feature_fusion): z0, z1 (..., C); w1 packed (3C, 2C) = [gate.0; fusion.0],
    w2 packed (C, 3C) = [gate.2 | fusion.2]; biases fp32.  z1_coarse: z0 is (N, 2h, 2w, C) and z1 the coarse (N, h, w, C) tensor,
    read through the bilinear x2 resampling.  frag: w1 is the fragment stream of BOTH layers (pack.fusion_frag(w1, w2), 9*C*C values),
    w2 is None -> the direct form (s2m2_feature_fusion_frag), meant for short row counts."""
    C = z0.shape[-1]
    if z1.dtype != z0.dtype or z1.shape[-1] != C:
        raise ValueError("feature_fusion: z0 and z1 must match")
    hc = wc = 0
    if z1_coarse:
        if z0.dim() != 4 or z1.dim() != 4 or tuple(z0.shape[:3]) != (z1.shape[0], 2 * z1.shape[1], 2 * z1.shape[2]):
            raise ValueError("feature_fusion: z1_coarse needs z0 (N,2h,2w,C) and z1 (N,h,w,C)")
        hc, wc = z1.shape[1], z1.shape[2]
    elif z1.shape != z0.shape:
        raise ValueError("feature_fusion: z0 and z1 must match")
    if frag:
        if w2 is not None or w1.numel() != 9 * C * C or w1.dtype != z0.dtype or not w1.is_contiguous():
            raise ValueError(f"feature_fusion: frag needs the {9 * C * C}-value fragment stream as w1 and w2 = None")
    elif tuple(w1.shape) != (3 * C, 2 * C) or tuple(w2.shape) != (C, 3 * C) or w1.dtype != z0.dtype or w2.dtype != z0.dtype:
        raise ValueError(f"feature_fusion: w1 must be ({3 * C}, {2 * C}) and w2 ({C}, {3 * C}) in {z0.dtype}")
    for t, n in ((b1, 3 * C), (bg, C), (bf, C)):
        if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
            raise ValueError("feature_fusion: biases must be fp32 (3C), (C), (C)")
    _dev(w1, b1, bg, bf) if frag else _dev(w1, w2, b1, bg, bf)
    rows, s0 = _token_rows(z0, "feature_fusion")
    _, s1 = _token_rows(z1, "feature_fusion")
    out = torch.empty(z0.shape, device=z0.device, dtype=z0.dtype)
    if frag:
        _check(load().s2m2_feature_fusion_frag(z0.data_ptr(), z1.data_ptr(), out.data_ptr(), s0, s1, C, rows, C, w1.data_ptr(), b1.data_ptr(),
                                               bg.data_ptr(), bf.data_ptr(), hc, wc, _DT[z0.dtype], _stream()), "s2m2_feature_fusion_frag")
        _meter("feature_fusion", 2.0 * rows * C * C * 9)
        return out
    _check(load().s2m2_feature_fusion(z0.data_ptr(), z1.data_ptr(), out.data_ptr(), s0, s1, C, rows, C, w1.data_ptr(), b1.data_ptr(),
                                      w2.data_ptr(), bg.data_ptr(), bf.data_ptr(), hc, wc, _DT[z0.dtype], _stream()), "s2m2_feature_fusion")
    _meter("feature_fusion", 2.0 * rows * C * C * 9)                # (2C -> 3C) + (C -> C) + (2C -> C)
    return out


def layernorm(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm without affine over the last axis of a (..., C) tensor with contiguous channels and a uniform row stride."""
    C = x.shape[-1]
    if x.stride(-1) != 1:
        raise ValueError("layernorm: channels must be contiguous")
    rows = x.numel() // C
    xs = x.stride(-2) if x.dim() > 1 else C
    for d in range(x.dim() - 2):
        if x.shape[d] > 1 and x.stride(d) != x.stride(d + 1) * x.shape[d + 1]:
            raise ValueError("layernorm: rows must have a uniform stride")
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    _check(load().s2m2_layernorm(x.data_ptr(), out.data_ptr(), rows, C, xs, C, _DT[x.dtype], _stream()), "s2m2_layernorm")
    return out


def groupnorm_nhwc(x: torch.Tensor, groups: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.GroupNorm on a contiguous (N,H,W,C) tensor; gamma/beta fp32."""
    _dev(x, gamma, beta)
    n, h, w, c = x.shape
    ws = torch.empty(load().s2m2_groupnorm_workspace_bytes(n, groups) // 8, device=x.device, dtype=torch.float64)
    out = torch.empty_like(x)
    _check(load().s2m2_groupnorm_nhwc(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), n, h * w, c,
                                      groups, eps, _DT[x.dtype], _stream()), "s2m2_groupnorm_nhwc")
    return out


def convex_upsample(maps, logits: torch.Tensor, factor: int, scales=None, logit_up2: bool = False,
                    chan_out: Optional[torch.Tensor] = None):
    """maps: list of (B,1,hs,ws) or (B,hs,ws) fp32 tensors; logits (B,Ho,Wo,>=16) NHWC (9 used).  -> list of (B,1,Ho,Wo) fp32.
    chan_out: optional (B,Ho,Wo) view (one channel of an NHWC tensor, dense pixels) that also receives map 0 in its dtype."""
    B, hs, ws = maps[0].shape[0], maps[0].shape[-2], maps[0].shape[-1]
    n = len(maps)
    maps = [m.float().contiguous() for m in maps]
    _dev(*maps)
    ls = _nhwc(logits)
    Ho, Wo = hs * factor, ws * factor
    exp_l = (B, hs, ws) if logit_up2 else (B, Ho, Wo)
    if tuple(logits.shape[:3]) != exp_l:
        raise ValueError(f"convex_upsample: logits must be {exp_l + ('>=16',)}, got {tuple(logits.shape)}")
    base = torch.empty((n, B, 1, Ho, Wo), device=logits.device, dtype=torch.float32)      # one allocation: callers can copy all maps at once
    outs = [base[k] for k in range(n)]
    xp = (_vp * n)(*[m.data_ptr() for m in maps])
    op = (_vp * n)(*[o.data_ptr() for o in outs])
    sc = (ctypes.c_float * n)(*[float(v) for v in (scales or [1.0] * n)])
    if chan_out is not None and (chan_out.dtype != logits.dtype or tuple(chan_out.shape) != (B, Ho, Wo)):
        raise ValueError("convex_upsample: chan_out must be a (B,Ho,Wo) channel view in the logits dtype")
    _check(load().s2m2_convex_upsample(xp, op, sc, n, logits.data_ptr(), ls, B, hs, ws, factor, int(logit_up2),
                                       chan_out.data_ptr() if chan_out is not None else None,
                                       chan_out.stride(2) if chan_out is not None else 0, _DT[logits.dtype], _stream()),
           "s2m2_convex_upsample")
    return outs


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, swap_halves: bool = False,
              pe: Optional[Tuple[torch.Tensor, torch.Tensor, int, int]] = None, scale: Optional[float] = None):
    """q, k, v: (nb, N, heads*D) views with contiguous channels and dense token rows (e.g. slices of a fused QKV buffer).
    -> out (nb, Nq, heads*D) [, pe_sum (nb, Nq, heads*32) when pe = (px (2w-1,16) fp32, py (2h-1,16) fp32, w, h)]."""
    nb, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    for t in (q, k, v):
        if t.stride(2) != 1 or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1)) or not t.is_cuda:
            raise ValueError("attention: q/k/v must be device tensors with contiguous channels and dense token rows")
    out = torch.empty((nb, Nq, C), device=q.device, dtype=q.dtype)
    pe_out = None
    px = py = None
    gw = gh = 0
    if pe is not None:
        px, py, gw, gh = pe
        _dev(px, py)
        pe_out = torch.empty((nb, Nq, heads * 32), device=q.device, dtype=q.dtype)
    flops = 4.0 * nb * heads * Nq * Nk * D                            # QK^T + PV (4 N^2 d per batch x head)
    ev = None
    if ATTN_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _check(load().s2m2_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(1), k.stride(1), v.stride(1), C,
                                 nb, heads, Nq, Nk, D, float(scale if scale is not None else D ** -0.5), int(swap_halves),
                                 px.data_ptr() if px is not None else None, py.data_ptr() if py is not None else None,
                                 pe_out.data_ptr() if pe_out is not None else None, heads * 32, gw, gh, _DT[q.dtype], _stream()),
           "s2m2_attention")
    if ev is not None:
        ev[1].record()
        ATTN_EVENTS.append((ev[0], ev[1], flops, f"({nb},{heads},{Nq},{D}){'+pe' if pe is not None else ''}",
                            2.0 * nb * heads * D * (2 * Nq + 2 * Nk)))       # unique bytes: q, k, v read + o written (fp16)
    _meter("attention", flops)
    return (out, pe_out) if pe is not None else out


def attention_supported(nb: int, heads: int, N: int, D: int, dtype: torch.dtype, grid: Optional[Tuple[int, int]] = None) -> Tuple[bool, str]:
    """Would s2m2_attention take this launch?  grid = (w, h) of the token grid for the PE variant.  -> (ok, reason)."""
    gw, gh = grid if grid is not None else (0, 0)
    lib = load()
    ok = bool(lib.s2m2_attention_supported(nb, heads, N, D, gw, gh, _DT[dtype]))
    return ok, ("" if ok else lib.s2m2_last_error().decode())


def resample2x(x: torch.Tensor, mode: int) -> torch.Tensor:
    """(N,H,W,C) -> AvgPool2d(2) (mode 0) or bilinear x2, align_corners=False (mode 1)."""
    n, h, w, c = x.shape
    xs = _nhwc(x)
    out = torch.empty((n, h // 2, w // 2, c) if mode == 0 else (n, 2 * h, 2 * w, c), device=x.device, dtype=x.dtype)
    _check(load().s2m2_resample2x(x.data_ptr(), out.data_ptr(), n, h, w, c, xs, c, mode, _DT[x.dtype], _stream()), "s2m2_resample2x")
    return out


def cv_lookup_into(cv: torch.Tensor, disp: torch.Tensor, buf: torch.Tensor, off1: int, off2: int, radius: int = 4) -> None:
    """K3 writing straight into channel slots of a wider NHWC tensor: taps of level 0 -> buf[..., off1:off1+2r+1], level 1 ->
    buf[..., off2:off2+2r+1] (buf (B,h,w,Cb) contiguous; the other channels are left untouched)."""
    _dev(disp, buf)
    pitch = _cv_pitch(cv, "cv_lookup")
    B, h, w, _ = cv.shape
    cb = buf.shape[-1]
    es = buf.element_size()
    _check(load().s2m2_cv_lookup(cv.data_ptr(), disp.data_ptr(), buf.data_ptr() + off1 * es, buf.data_ptr() + off2 * es, B, h, w,
                                 radius, _DT[cv.dtype], _DT[buf.dtype], h * w * cb, cb, 1, pitch, _stream()), "s2m2_cv_lookup")


_IMG_DT = {torch.float32: 0, torch.float16: 1, torch.uint8: 2}


def image_prep(img0: torch.Tensor, img1: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """img0, img1 (B,3,H,W) fp32 / fp16 / uint8 in [0,255] -> x8 (2B,H,W,8): channels 1..3 = normalised RGB, others 0 (``out``: written in place)."""
    if img0.dtype not in _IMG_DT:
        img0 = img0.float()
    img1 = img1.to(img0.dtype)
    img0, img1 = img0.contiguous(), img1.contiguous()
    _dev(img0, img1)
    B, _, H, W = img0.shape
    if out is not None:
        if tuple(out.shape) != (2 * B, H, W, 8) or out.dtype != dtype or not out.is_contiguous() or out.device != img0.device:
            raise RuntimeError(f"image_prep: out must be a contiguous {(2 * B, H, W, 8)} {dtype} tensor on {img0.device}")
        x8 = out
    else:
        x8 = torch.empty((2 * B, H, W, 8), device=img0.device, dtype=dtype)
    _check(load().s2m2_image_prep(img0.data_ptr(), img1.data_ptr(), x8.data_ptr(), B, H, W, _IMG_DT[img0.dtype], _DT[dtype], _stream()),
           "s2m2_image_prep")
    return x8


def clock_probe(out: torch.Tensor) -> None:
    """Measurement aid: out (2,) int64 device tensor <- {shader-clock ticks, 100 MHz real-time ticks} when the stream reaches this point."""
    _dev(out)
    _check(load().s2m2_debug_clock_probe(out.data_ptr(), _stream()), "s2m2_debug_clock_probe")


def poison_lds() -> None:
    """Test aid: quiet-NaN patterns in the LDS of every CU (s2m2_debug_poison_lds), on the current stream."""
    _check(load().s2m2_debug_poison_lds(_stream()), "s2m2_debug_poison_lds")


def refine_prep(disp: torch.Tensor, conf: torch.Tensor, occ: Optional[torch.Tensor], mode: int, dtype: torch.dtype) -> torch.Tensor:
    """(B,1,h,w) fp32 maps -> (B,h,w,8) side input of the global (mode 0) / local (mode 1) refiner."""
    _dev(disp, conf, occ)
    B, _, h, w = disp.shape
    out = torch.empty((B, h, w, 8), device=disp.device, dtype=dtype)
    _check(load().s2m2_refine_prep(disp.data_ptr(), conf.data_ptr(), occ.data_ptr() if occ is not None else None, out.data_ptr(),
                                   B * h * w, mode, _DT[dtype], _stream()), "s2m2_refine_prep")
    return out


def global_update(upd: torch.Tensor, disp: torch.Tensor, conf: torch.Tensor, clamp0: bool) -> torch.Tensor:
    """upd (B,h,w,C>=1) NHWC (channel 0 used), disp/conf (B,1,h,w) fp32 -> refined disparity (B,1,h,w) fp32."""
    _dev(disp, conf)
    out = torch.empty_like(disp)
    _check(load().s2m2_global_update(upd.data_ptr(), _nhwc(upd), disp.data_ptr(), conf.data_ptr(), out.data_ptr(), disp.numel(),
                                     int(clamp0), _DT[upd.dtype], _stream()), "s2m2_global_update")
    return out


def refine_update(dco: torch.Tensor, disp: torch.Tensor, conf: torch.Tensor, occ: torch.Tensor, use_positivity: bool,
                  want_small: bool = False):
    """dco (B,h,w,>=10) NHWC deltas; disp/conf/occ (B,1,h,w) fp32 -> new (disp, conf, occ) (fresh tensors) [, small (B,h,w,8): the
    refine_prep(mode 1) side input of the next iteration, from the same launch]."""
    _dev(disp, conf, occ)
    outs = torch.empty((3,) + tuple(disp.shape), device=disp.device, dtype=torch.float32)
    small = torch.empty((disp.shape[0], disp.shape[-2], disp.shape[-1], 8), device=disp.device, dtype=dco.dtype) if want_small else None
    _check(load().s2m2_refine_update_to(dco.data_ptr(), _nhwc(dco), disp.data_ptr(), conf.data_ptr(), occ.data_ptr(),
                                        outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                        small.data_ptr() if small is not None else None, disp.numel(), disp.shape[-1],
                                        int(use_positivity), _DT[dco.dtype], _stream()), "s2m2_refine_update_to")
    return (outs[0], outs[1], outs[2], small) if want_small else (outs[0], outs[1], outs[2])


def tanh(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    y = torch.empty_like(x)
    _check(load().s2m2_tanh(x.data_ptr(), y.data_ptr(), x.numel(), _DT[x.dtype], _stream()), "s2m2_tanh")
    return y


def stem_mlp(x8: torch.Tensor, w0: torch.Tensor, b0: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor) -> torch.Tensor:
    """(N,H,W,8) -> (N,H,W,16): Conv1x1(8->16) - GELU - Conv1x1(16->16) per pixel (s2m2_stem_mlp); fp32 weights (16,8), (16,16), biases (16)."""
    _dev(x8, w0, b0, w1, b1)
    if x8.shape[-1] != 8 or tuple(w0.shape) != (16, 8) or tuple(w1.shape) != (16, 16) or b0.numel() != 16 or b1.numel() != 16:
        raise ValueError("stem_mlp: x8 (...,8), w0 (16,8), w1 (16,16), biases (16)")
    if any(t.dtype != torch.float32 for t in (w0, b0, w1, b1)):
        raise ValueError("stem_mlp: weights and biases must be fp32")
    out = torch.empty(tuple(x8.shape[:-1]) + (16,), device=x8.device, dtype=x8.dtype)
    _check(load().s2m2_stem_mlp(x8.data_ptr(), w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), out.data_ptr(),
                                x8.numel() // 8, _DT[x8.dtype], _stream()), "s2m2_stem_mlp")
    return out


def image_pad(img: torch.Tensor, factor: int = 32) -> torch.Tensor:
    """Reference image_pad (image_utils.py:27-71) on the device: (B,C,H,W) fp32/fp16/uint8 -> (B,C,Hn,Wn) fp32."""
    if img.dtype not in _IMG_DT:
        img = img.float()
    img = img.contiguous()
    _dev(img)
    B, C, H, W = img.shape
    Hn, Wn = -(-H // factor) * factor, -(-W // factor) * factor
    pooled = torch.empty((B, C, H // factor, W // factor), device=img.device, dtype=torch.float32)
    out = torch.empty((B, C, Hn, Wn), device=img.device, dtype=torch.float32)
    _check(load().s2m2_image_pad(img.data_ptr(), pooled.data_ptr(), out.data_ptr(), B, C, H, W, factor, _IMG_DT[img.dtype], _stream()),
           "s2m2_image_pad")
    return out
