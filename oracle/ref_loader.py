"""Loader for ``oracle/_ref`` (the unmodified reference model as bytecode, built by oracle/make_ref.py).

TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__ and bench.py's cpu_baseline leg import this.

    from oracle import ref_loader
    if ref_loader.available():
        S2M2 = ref_loader.reference_class()           # the reference's own nn.Module, s2m2.py:13-197
        model = ref_loader.reference_model(sd, C, ntr, use_positivity, refine_iter)
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import os
import sys
from typing import Mapping, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "_ref")
PKG = "s2m2_reference_model"


def why_not() -> Optional[str]:
    """None when the bytecode package is importable by THIS interpreter, else the reason."""
    d = os.path.join(ROOT, PKG)
    meta = os.path.join(d, "BUILD.json")
    if not os.path.exists(os.path.join(d, "s2m2.pyc")) or not os.path.exists(meta):
        return f"{d} not built (python oracle/make_ref.py in the build container, where /root/reference exists)"
    magic = json.load(open(meta)).get("magic")
    if magic != importlib.util.MAGIC_NUMBER.hex():
        return f"{d} was byte-compiled by another CPython (magic {magic}, this interpreter {importlib.util.MAGIC_NUMBER.hex()})"
    return None


def available() -> bool:
    return why_not() is None


def reference_module():
    """the reference's ``s2m2.core.model.s2m2`` module (sourceless import of oracle/_ref/s2m2_reference_model/s2m2.pyc)"""
    why = why_not()
    if why:
        raise ImportError(why)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module(PKG + ".s2m2")


def reference_class():
    return reference_module().S2M2


def reference_model(sd: Mapping[str, "object"], feature_channels: int, num_transformer: int, use_positivity: bool, refine_iter: int,
                    output_upsample: bool = False):
    """reference ``S2M2(C, 1, ntr, ...)`` in eval mode with ``sd`` loaded strictly (model_utils.py:30-43 without the checkpoint file)"""
    m = reference_class()(feature_channels, 1, num_transformer, use_positivity=use_positivity, output_upsample=output_upsample,
                          refine_iter=refine_iter).eval()
    m.load_state_dict(sd, strict=True)
    return m
