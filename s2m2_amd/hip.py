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

# name -> (restype, argtypes); must list every symbol declared in include/s2m2_hip.h
SIGNATURES = {
    "s2m2_version": (_i, []),
    "s2m2_last_error": (ctypes.c_char_p, []),
    "s2m2_ln_corr_kernel_name": (ctypes.c_char_p, [_i, _i, _i]),
    "s2m2_ln_corr": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "s2m2_sinkhorn_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i, _i]),
    "s2m2_sinkhorn_regress": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "s2m2_cv_lookup": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _ll, _ll, _ll, _vp]),
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
        _lib = lib
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().s2m2_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError("s2m2_amd.hip: tensors must be contiguous device tensors")


def ln_corr(feat: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, cv_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """feat (2B,h,w,C) channels-last tokens (left = first B) -> cv (B,h,w,w).  [A4]"""
    _dev(feat, ln_w, ln_b)
    twoB, h, w, C = feat.shape
    B = twoB // 2
    cv_dtype = cv_dtype or feat.dtype
    cv = torch.empty((B, h, w, w), device=feat.device, dtype=cv_dtype)
    _check(load().s2m2_ln_corr(feat.data_ptr(), ln_w.float().data_ptr(), ln_b.float().data_ptr(), cv.data_ptr(),
                               B, h, w, C, _DT[feat.dtype], _DT[cv_dtype], _stream()), "s2m2_ln_corr")
    return cv


def sinkhorn_regress(cv: torch.Tensor, use_positivity: bool, ot_iter: int = 3, want_argmax: bool = False):
    """cv (B,h,w,w) -> disp, conf, occ (B,1,h,w) fp32 [, argmax (B,h,w) int32].  [A5+A6]"""
    _dev(cv)
    B, h, w, _ = cv.shape
    out = torch.empty((3, B, 1, h, w), device=cv.device, dtype=torch.float32)
    am = torch.empty((B, h, w), device=cv.device, dtype=torch.int32) if want_argmax else None
    _check(load().s2m2_sinkhorn_regress(cv.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                        am.data_ptr() if am is not None else None, B, h, w, ot_iter, int(use_positivity),
                                        _DT[cv.dtype], None, _stream()), "s2m2_sinkhorn_regress")
    return (out[0], out[1], out[2], am) if want_argmax else (out[0], out[1], out[2])


def cv_lookup(cv: torch.Tensor, disp: torch.Tensor, radius: int = 4, channels_last: bool = False,
              out_dtype: torch.dtype = torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """cv (B,h,w,w), disp (B,1,h,w) fp32 -> corr1, corr2: (B,2r+1,h,w) planar or (B,h,w,2r+1) channels-last.  [A9+A10]"""
    _dev(cv, disp)
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
                                 _DT[cv.dtype], _DT[out_dtype], bs, ps, ts, _stream()), "s2m2_cv_lookup")
    return c1, c2
