"""Build recipe for ``oracle/_ref``: the UNMODIFIED reference model, compiled from its sources where they lie.

    python oracle/make_ref.py            # needs /root/reference (the build container); writes oracle/_ref/s2m2_reference_model/*.pyc

TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE (same rule as oracle/s2m2_oracle.py: only tests/, __graft_entry__ and bench.py's
cpu_baseline leg may load what this builds; nothing under s2m2_amd/ does).

The reference's hot path is pure Python over torch (/root/reference/src/s2m2/core/model/*.py, SURVEY.md 8c: the model files import
only torch), so "compiling the reference" means byte-compiling those files: ``py_compile`` reads every source IN PLACE under
/root/reference and writes ONLY CPython bytecode (``.pyc``, the interpreter's binary format) into the git-ignored ``oracle/_ref/``.
No reference source text is copied into this repository or its history; the directory ships to the GPU box with the gpurun snapshot
like the in-tree ``libs2m2_hip.so`` does (git-ignored, not gpurun-ignored), where /root/reference does not exist.  The bytecode is
tied to the interpreter that wrote it (``importlib.util.MAGIC_NUMBER``): the GPU box runs the same image; oracle/ref_loader.py checks
the magic number and says so if it ever differs.

Used by: bench.py ``cpu_baseline`` (kind "reference": the reference's own CPU forward timed beside the HIP path, north_star),
tests/test_live_reference.py (HIP fp32 vs the live reference, no oracle in between; the oracle vs the live reference on the CPU).
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("S2M2_REFERENCE_SRC", "/root/reference/src/s2m2/core/model")
PKG = "s2m2_reference_model"
OUT = os.path.join(HERE, "_ref", PKG)


def build(verbose: bool = True) -> str:
    """-> package directory.  Raises FileNotFoundError when the reference sources are not on this machine."""
    if not os.path.isdir(SRC):
        raise FileNotFoundError(f"{SRC}: the reference sources are not on this machine (oracle/_ref is built in the build container "
                                f"and travels to the GPU box)")
    os.makedirs(OUT, exist_ok=True)
    names = sorted(f for f in os.listdir(SRC) if f.endswith(".py"))
    digest = {}
    for f in names:
        src = os.path.join(SRC, f)
        dst = os.path.join(OUT, f + "c")                       # sourceless import: <module>.pyc next to where <module>.py would be
        py_compile.compile(src, cfile=dst, dfile=f"<reference>/src/s2m2/core/model/{f}", doraise=True, optimize=0)
        digest[f] = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    json.dump({"built_from": SRC, "python": sys.version.split()[0], "magic": importlib.util.MAGIC_NUMBER.hex(), "sha256_16": digest},
              open(os.path.join(OUT, "BUILD.json"), "w"), indent=1)
    if verbose:
        print(f"[oracle/make_ref] {len(names)} modules byte-compiled from {SRC} -> {OUT}")
    return OUT


if __name__ == "__main__":
    build()
