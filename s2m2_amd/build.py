"""Build libs2m2_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m s2m2_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  Objects are cached per source file (mtime based) under s2m2_amd/csrc/_obj.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
SUFFIX = os.environ.get("S2M2_LIB_SUFFIX", "")          # experiment builds: extra -D flags -> libs2m2_hip<suffix>.so
DEFINES = os.environ.get("S2M2_BUILD_DEFINES", "").split()
LIB = os.path.join(LIBDIR, f"libs2m2_hip{SUFFIX}.so")
if SUFFIX:
    OBJ = os.path.join(CSRC, "_obj" + SUFFIX)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on", "-fno-fast-math"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _defines_file() -> str:
    return os.path.join(OBJ, "defines.txt")


def effective_defines(extra=()) -> list:
    """S2M2_BUILD_DEFINES + `extra` + what an earlier build of this object directory persisted (the tracked-loads fallback of
    __graft_entry__.build(): once taken, every later incremental compile of a single file must use it too -- objects with and without
    -DS2M2_UNTRACKED_LOADS=0 must never be linked together)."""
    keep = []
    explicit = any(d.startswith("-DS2M2_UNTRACKED_LOADS=") for d in list(DEFINES) + list(extra))
    if os.path.exists(_defines_file()) and not explicit:           # an explicit value (environment / caller) overrides the persisted one
        keep = [d for d in open(_defines_file()).read().split() if d.startswith("-DS2M2_UNTRACKED_LOADS=")]
    out = list(DEFINES)
    for d in list(extra) + keep:
        if d not in out:
            out.append(d)
    return out


def _compile(src: str, force: bool, hdr_mtime: float, defines=()) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(spath)
            and os.path.getmtime(obj) > hdr_mtime):
        return obj
    cmd = [HIPCC, *FLAGS, *defines, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def config_key() -> str:
    """what the objects under OBJ were compiled with: library suffix, extra defines, compiler version (keys the check_isa stamp)"""
    import hashlib
    try:
        ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    except OSError:
        ver = "?"
    return hashlib.sha256((SUFFIX + "|" + " ".join(effective_defines()) + "|" + ver).encode()).hexdigest()[:12]


def build(force: bool = False, verbose: bool = True, extra_defines=()) -> str:
    """extra_defines: appended to S2M2_BUILD_DEFINES for this call and PERSISTED next to the objects (a change of the effective defines
    forces a full rebuild)"""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    defines = effective_defines(extra_defines)
    before = open(_defines_file()).read().split() if os.path.exists(_defines_file()) else None
    if before is not None and before != defines:
        force = True                                               # objects of another configuration: never mix
    if before is None and (extra_defines or any(f.endswith(".o") for f in os.listdir(OBJ))):
        force = True                                               # objects of an unknown configuration (an interrupted forced rebuild)
    if force and os.path.exists(_defines_file()):
        os.remove(_defines_file())                                 # written again only after EVERY object compiled (below): a failed
    srcs = sources()                                               # forced rebuild leaves no stamp, so the next build forces again
    hdr = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr, tuple(defines)), srcs))
    tmp = _defines_file() + ".tmp"
    with open(tmp, "w") as f:
        f.write(" ".join(defines) + "\n")
    os.replace(tmp, _defines_file())                               # atomically, after all objects exist
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[s2m2_amd.build] {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
