#!/usr/bin/env python3
"""Static audit of the compiled kernels (no GPU): compiles every csrc/*.hip to gfx950 assembly and lists, per kernel, the registers it takes,
what the compiler spilled (scratch bytes, scratch instructions) and how many v_accvgpr moves it emitted -- the two pathologies that cost the
wide-head attention 2.5x in round 5 and that no profile names (Q fragments kept in scratch; a copy of every accumulator tile between VGPRs and
AGPRs around each stage's MFMAs).  Kernels with neither are summarised in one line per file.

    python tools/isa_audit.py [file.hip ...] [--all] [--fp32]          # default: fp16 instantiations with scratch or > 64 accvgpr moves
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "s2m2_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-fno-fast-math", "--cuda-device-only", "-S",
         "-I" + os.path.join(ROOT, "include")]


def compile_s(src, out):
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *os.environ.get("S2M2_BUILD_DEFINES", "").split(), os.path.join(CSRC, src), "-o", out],
                       capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
    return out


def audit(path):
    txt = open(path).read()
    stats, cur = {}, None
    for line in txt.splitlines():
        if line.startswith("_Z") and "; @_Z" in line:
            cur = line.split(":")[0]
            stats[cur] = dict(acc_r=0, acc_w=0, scratch_ops=0, mfma=0)
        elif cur:
            if "v_accvgpr_read" in line: stats[cur]["acc_r"] += 1
            elif "v_accvgpr_write" in line: stats[cur]["acc_w"] += 1
            elif "scratch_" in line: stats[cur]["scratch_ops"] += 1
            elif "v_mfma" in line: stats[cur]["mfma"] += 1
            if "s_endpgm" in line: cur = None
    meta = {}
    i = txt.find("amdhsa.kernels:")
    for b in re.split(r"\n  - \.agpr_count:", txt[i:])[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
        meta[g("name")] = dict(agpr=b.strip().split("\n")[0].strip(), regs=g("vgpr_count"), spill=g("vgpr_spill_count"),
                               scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    return stats, meta


def demangle(n):
    try:
        return subprocess.run(["c++filt", "-p", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    show_all, fp32 = "--all" in sys.argv, "--fp32" in sys.argv
    srcs = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    tmp = tempfile.mkdtemp(prefix="isa_audit_")
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        outs = list(ex.map(lambda s: compile_s(s, os.path.join(tmp, s.replace(".hip", ".s"))), srcs))
    for src, out in zip(srcs, outs):
        stats, meta = audit(out)
        flagged = 0
        for k, v in stats.items():
            m = meta.get(k, {})
            is32 = bool(re.search(r"I[a-zA-Z0-9_]*f(Li|EE|L)", k)) and "DF16_" not in k
            if is32 and not fp32:
                continue
            bad = v["scratch_ops"] > 0 or v["acc_r"] + v["acc_w"] > 64
            if bad or show_all:
                flagged += bad
                print(f"{src:16s} regs={m.get('regs', '?'):>3} (agpr {m.get('agpr', '?'):>3}) spilled={m.get('spill', '?'):>3} scratch={m.get('scratch', '?'):>4} B"
                      f" | scratch instr {v['scratch_ops']:3d}  accvgpr r/w {v['acc_r']:3d}/{v['acc_w']:3d}  mfma {v['mfma']:3d} | {demangle(k)[:150]}")
        print(f"# {src}: {len(stats)} kernels, {flagged} listed above")


if __name__ == "__main__":
    main()
