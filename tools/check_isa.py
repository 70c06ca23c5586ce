#!/usr/bin/env python3
"""Build-time guard of the scheduling assumptions behind K5 v5 (conv_frag_kernel, s2m2_amd/csrc/conv.hip), K14 (conv_block_kernel, convblock.hip: the
same ring, check_ring_kernel), the direct form of K10
(feature_fusion_direct_kernel, s2m2_amd/csrc/fusion.hip) and the shipped K1 (ln_corr_kernel<PRENORM>, s2m2_amd/csrc/ln_corr.hip: s2m2_corr).

The kernel prefetches its weight fragments with loads the compiler does NOT track (common.h: global_load16_async) and waits for them with
hand-counted ``s_waitcnt vmcnt(N)``.  That is only correct while, inside the K loop,

  1. the compiler inserts no ``s_waitcnt vmcnt(0)`` of its own (it would drain the ring on every tap: slow, and a sign that a tracked
     load got mixed in), and
  2. no instruction other than an MFMA reads a ring register (a ``v_mov`` / ``v_accvgpr_write`` copy of a register whose load may still be
     in flight reads stale data: DESIGN.md section 4 records two bugs and one memory fault of exactly this kind).

This script compiles conv.hip to gfx950 assembly (device only), finds the K-loop basic blocks of every conv_frag_kernel instantiation
(blocks that hold both MFMAs and 16-byte global loads) and fails when either property is violated.  ``__graft_entry__.build()`` runs it
when the library is (re)built;  python tools/check_isa.py [--keep-asm PATH]  runs it alone (about a minute: one device-only compile).

feature_fusion_direct_kernel is one straight-line block (fully unrolled k16 steps): there the checked region runs from the first MFMA to the
last fragment request -- while the ring is being refilled every ring register is either in flight or about to feed an MFMA -- and, besides
the two properties above, every ``s_waitcnt vmcnt(N)`` in the region must carry N = ring depth - 1 (the D of FusionDirectCfg<C, BM, NW, D>: the
prefetch distance the kernel was written for is what the generated code has).

ln_corr_kernel<..., PRENORM = true> (the default K1 of the forward, and its hybrid left-fragment branch) requests all of its tokens with the
same untracked loads and counted waits.  Checked over the whole kernel with the in-order memory model the compiler itself uses on gfx9
(vmcnt counts loads and stores in issue order; ``s_waitcnt vmcnt(N)`` retires all but the youngest N):
  a. no instruction reads or overwrites the destination of a load that may still be in flight;
  b. no load is in flight across a label or a branch (a register copy at a control-flow merge would read it early);
  c. the token burst is really in flight together: the deepest queue reaches RIF * PPL (* 2 with EARLY_B) loads -- the optimiser has not
     sunk the requests next to their LDS stores again (29 us instead of 22).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "s2m2_amd", "csrc", "conv.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-fno-fast-math", "--cuda-device-only", "-S"]


def regs_of(tok: str):
    """'v[12:15]' -> {12..15}, 'v7' -> {7}; anything else -> empty"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check_function(name: str, lines):
    """-> list of problems found in the K-loop blocks of one kernel"""
    blocks, cur = [], []
    for ln in lines:
        if re.match(r"^\.LBB\S*:", ln):
            blocks.append(cur)
            cur = []
        else:
            cur.append(ln.strip())
    blocks.append(cur)
    problems, nloops = [], 0
    for blk in blocks:
        ins = [l.split(";")[0].strip() for l in blk if l and not l.startswith((".", ";"))]
        mfma = [l for l in ins if l.startswith("v_mfma")]
        loads = [l for l in ins if l.startswith("global_load_dwordx4")]
        if len(mfma) < 16 or not loads:
            continue
        nloops += 1
        ring = set()
        for l in loads:
            ring |= regs_of(l.split()[1].rstrip(","))
        for l in ins:
            if re.search(r"s_waitcnt\b.*vmcnt\(0\)", l):
                problems.append(f"{name}: 's_waitcnt vmcnt(0)' inside a K-loop block ({len(mfma)} MFMAs): {l}")
            if l.startswith(("v_mfma", "global_load_dwordx4", "s_", "ds_", "buffer_")):
                continue
            ops = [t.strip().rstrip(",") for t in l.split()[1:]]
            srcs = ops[1:] if ops else []
            for t in srcs:
                if regs_of(t) & ring:
                    problems.append(f"{name}: non-MFMA instruction reads a ring register inside the K loop: {l}")
    if nloops == 0:
        problems.append(f"{name}: no K-loop block found (the check no longer matches the generated code)")
    return problems, nloops


def check_straight_line(name: str, lines, depth=None, min_mfma=16, exact=True):
    """feature_fusion_direct_kernel / conv_narrow_kernel: region = first MFMA .. last 16-byte global load of the kernel's instruction stream.
    depth: ring depth (default: parsed from FusionDirectCfg); exact: the counted waits inside the region are exactly {depth - 1}, else depth - 1
    is one of them (the unrolled tail of conv_narrow's rings counts down, and the compiler adds waits of its own for tracked loads)"""
    ins = [l.split(";")[0].strip() for l in lines if l.strip() and not l.strip().startswith((".", ";"))]
    mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    ld = [i for i, l in enumerate(ins) if l.startswith("global_load_dwordx4")]
    if len(mf) < min_mfma or not ld or (ld[-1] < mf[0] and exact):
        return [f"{name}: no MFMA / fragment-load region found (the check no longer matches the generated code)"], 0
    lo, hi = mf[0], ld[-1]
    if ld[-1] < mf[0]:                                            # a ring as deep as the whole K loop: every request precedes the first MFMA
        lo, hi = ld[-1], mf[-1]
    # registers whose fragment request may still be in flight: set by a 16-byte load, cleared by the MFMA that consumes them (the allocator
    # moves a ring slot to other registers between refills, so the set is tracked instruction by instruction)
    flying = set()
    problems, waits = [], set()
    for idx, l in enumerate(ins[:hi + 1]):
        inside = idx >= lo
        m = re.search(r"s_waitcnt\b.*vmcnt\((\d+)\)", l)
        if m and int(m.group(1)) == 0:
            flying.clear()                                        # everything requested so far has landed (the drain after a K loop)
            continue
        if m and inside:
            waits.add(int(m.group(1)))
        ops = [t.strip().rstrip(",") for t in l.split()[1:]]
        if l.startswith("global_load_dwordx4"):
            flying |= regs_of(ops[0])
            continue
        if l.startswith("v_mfma"):
            for t in ops[1:3]:
                flying -= regs_of(t)
            continue
        if l.startswith("ds_write"):                              # (the prologue's tracked loads -- row tile, bias vectors -- end in LDS)
            for t in ops[1:]:
                flying -= regs_of(t)
            continue
        if l.startswith(("s_", "buffer_")):
            continue
        srcs, dsts = ([], ops[:1]) if l.startswith("ds_read") else (ops[1:], ops[:1])
        if not inside:                                            # prologue: tracked loads (bilinear z1 path) are consumed by plain arithmetic
            for t in srcs:
                flying -= regs_of(t)
            continue
        for t in srcs:
            if regs_of(t) & flying:
                problems.append(f"{name}: non-MFMA instruction reads a register whose fragment may be in flight: {l}")
        for t in dsts:
            if regs_of(t) & flying:
                problems.append(f"{name}: instruction overwrites a register whose fragment may be in flight: {l}")
    if depth is None:
        m = re.search(r"FusionDirectCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        depth = int(m.group(4)) if m else 9
    if exact and waits != {depth - 1}:
        problems.append(f"{name}: vmcnt waits inside the refill region are {sorted(waits)} (expected exactly {depth - 1}: ring depth - 1)")
    if not exact and depth - 1 not in waits:
        problems.append(f"{name}: vmcnt waits inside the refill region are {sorted(waits)} (expected {depth - 1} = ring depth - 1 among them)")
    return problems, 1


def check_inflight(name: str, lines, expect_depth: int):
    """properties a-c of the module docstring for one ln_corr_kernel<PRENORM> instantiation"""
    problems, queue, deepest = [], [], 0                             # queue: destination register sets of loads (empty set: a store)
    for raw in lines:
        ln = raw.strip()
        if re.match(r"^\.LBB\S*:", ln) or ln.startswith(("s_cbranch", "s_branch", "s_barrier")):
            if any(q for q in queue):
                problems.append(f"{name}: {sum(1 for q in queue if q)} load(s) in flight across control flow at: {ln.split(';')[0].strip()}")
            continue
        l = ln.split(";")[0].strip()
        if not l or l.startswith("."):
            continue
        m = re.search(r"s_waitcnt\b.*vmcnt\((\d+)\)", l)
        if m:
            n = int(m.group(1))
            queue = queue[len(queue) - n:] if n < len(queue) else queue
            if n == 0:
                queue = []
            continue
        if l.startswith("s_waitcnt") or l.startswith("s_"):
            continue
        ops = [t.strip().rstrip(",") for t in l.split()[1:]]
        if l.startswith(("global_load", "buffer_load")):
            flying = set().union(*queue) if queue else set()
            for t in ops[1:]:
                if regs_of(t) & flying:
                    problems.append(f"{name}: address of a load reads a register whose load may be in flight: {l}")
            queue.append(regs_of(ops[0]))
            deepest = max(deepest, sum(1 for q in queue if q))
            continue
        flying = set().union(*queue) if queue else set()
        if l.startswith(("global_store", "buffer_store")):
            for t in ops:
                if regs_of(t) & flying:
                    problems.append(f"{name}: store reads a register whose load may be in flight: {l}")
            queue.append(set())
            continue
        if not flying:
            continue
        for t in ops:                                                 # sources and destinations alike: neither may touch a flying register
            if regs_of(t) & flying:
                problems.append(f"{name}: instruction touches a register whose load may be in flight: {l}")
    if deepest < expect_depth:
        problems.append(f"{name}: at most {deepest} loads in flight together, expected the whole token burst ({expect_depth})")
    return problems, deepest


def check_ln_corr(text):
    """every ln_corr_kernel<LnCorrCfg<T, TO, C, NWCAP, RIF, EARLY_B, TPW, PRENORM = true>> instantiation"""
    bad, seen = [], 0
    for name, lines in functions(text, "_ZN4s2m214ln_corr_kernel").items():
        # (parsed from the mangled name: this c++filt does not know DF16_ = _Float16)
        m = re.search(r"LnCorrCfgI(DF16_|f)(DF16_|f)Li(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)ELb([01])E", name)
        if not m or m.group(8) != "1":
            continue
        seen += 1
        a = ["fp16" if m.group(1) != "f" else "fp32", "fp16" if m.group(2) != "f" else "fp32"]
        C, rif, early = int(m.group(3)), int(m.group(5)), m.group(6) == "1"
        vec = 4 if m.group(1) == "f" else 8
        expect = rif * (C // vec // 8) * (2 if early else 1)
        problems, deepest = check_inflight(name, lines, expect)
        print(f"check_isa: ln_corr_kernel<{a[0]} -> {a[1]}, C={C}, PRENORM>: {deepest} loads in flight (expected {expect}), {len(problems)} problem(s)")
        bad += problems
    if seen == 0:
        bad.append("ln_corr.hip: no PRENORM instantiation of ln_corr_kernel found in the assembly")
    return bad


def _asm_marked(lines):
    """-> [(instruction text, is inside an ;;#ASMSTART / ;;#ASMEND pair, is a label)] of a function body"""
    out, in_asm = [], False
    for raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if re.match(r"^\.LBB\S*:", raw):
            out.append(("", False, True))
            continue
        l = raw.split(";")[0].strip()
        if l and not l.startswith((".", ";")):
            out.append((l, in_asm, False))
    return out


def check_ring_kernel(name: str, lines):
    """K14 (conv_block_kernel): the untracked ring of K5 v5 inside a kernel that also issues tracked 16-byte loads, so the ring is told apart by
    the inline-asm markers.  Checked: ring starts = asm requests followed by straight-line code -- until the next wait, MFMA, branch or label nothing
    writes or reads a register whose request is in flight (the allocator reuses the destination of a request whose value the program never consumes: the
    round-6 memory fault of this kernel)."""
    ins = _asm_marked(lines)
    problems, nloops, starts = [], 0, 0
    # (2) ring starts
    flying = set()
    for l, a, lab in ins:
        if lab or l.startswith("v_mfma") or re.search(r"s_waitcnt\b.*vmcnt\(", l) or l.startswith(("s_cbranch", "s_branch", "s_barrier")):
            flying.clear()
            continue
        ops = [t.strip().rstrip(",") for t in l.split()[1:]]
        if a and l.startswith("global_load_dwordx4"):
            if not flying:
                starts += 1
            flying |= regs_of(ops[0])
            continue
        if not flying or not ops:
            continue
        if l.startswith(("ds_write", "global_store", "scratch_store")):
            if any(regs_of(t) & flying for t in ops):
                problems.append(f"{name}: '{l}' reads a ring register whose request is in flight")
        elif regs_of(ops[0]) & flying:
            problems.append(f"{name}: '{l}' overwrites a ring register whose request is in flight")
        elif any(regs_of(t) & flying for t in ops[1:]):
            problems.append(f"{name}: '{l}' reads a ring register whose request is in flight")
    if starts < 2:
        problems.append(f"{name}: fewer than two ring starts found (the check no longer matches the generated code)")
    return problems, nloops, starts


def compile_asm(src: str, asm: str):
    defines = os.environ.get("S2M2_BUILD_DEFINES", "").split()
    r = subprocess.run([HIPCC, *FLAGS, *defines, src, "-o", asm], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        return None
    return open(asm).read().splitlines()


def functions(text, prefix: str):
    funcs, cur = {}, None
    for ln in text:
        m = re.match(r"^(" + prefix + r"\S*):", ln)
        if m:
            cur = []
            funcs[m.group(1)] = cur
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                cur.append(ln)
    return funcs


def main() -> int:
    keep = sys.argv[sys.argv.index("--keep-asm") + 1] if "--keep-asm" in sys.argv else None
    with tempfile.TemporaryDirectory() as td:
        text = compile_asm(SRC, keep or os.path.join(td, "conv.s"))
        ftext = compile_asm(os.path.join(ROOT, "s2m2_amd", "csrc", "fusion.hip"), os.path.join(td, "fusion.s"))
        ktext = compile_asm(os.path.join(ROOT, "s2m2_amd", "csrc", "ln_corr.hip"), os.path.join(td, "ln_corr.s"))
        ntext = compile_asm(os.path.join(ROOT, "s2m2_amd", "csrc", "conv_narrow.hip"), os.path.join(td, "conv_narrow.s"))
        btext = compile_asm(os.path.join(ROOT, "s2m2_amd", "csrc", "convblock.hip"), os.path.join(td, "convblock.s"))
        if text is None or ftext is None or ktext is None or ntext is None or btext is None:
            return 2
    bad = check_ln_corr(ktext)
    bfuncs = functions(btext, "_ZN4s2m217conv_block_kernel")
    if not bfuncs:
        print("check_isa: no conv_block_kernel instantiation found in the assembly")
        return 1
    for name, lines in bfuncs.items():
        p1, nloops, starts = check_ring_kernel(name, lines)
        m = re.search(r"CbCfgILi(\d+)ELi(\d+)E", name)
        print(f"check_isa: conv_block_kernel<C {m.group(1)}, PH {m.group(2)}>: {starts} ring (re)start window(s), {len(p1)} problem(s) (its K loops are held bit for bit against K5 v5 by tests/test_hip_convblock.py)")
        bad += p1
    nfuncs = functions(ntext, "_ZN4s2m218conv_narrow_kernel")
    if not nfuncs:
        print("check_isa: no conv_narrow_kernel instantiation found in the assembly")
        return 1
    for name, lines in nfuncs.items():
        m = re.search(r"NarrowCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        kh, kw, _, cin, mt, ntl, _ = (int(g) for g in m.groups())
        ns = (kh * kw * cin + 15) // 16
        problems, n = check_straight_line(name, lines, depth=min(ns * ntl, 8), min_mfma=ns * mt * ntl, exact=False)
        print(f"check_isa: conv_narrow_kernel<{kh}x{kw}, Cin {cin}, MT {mt}, NTL {ntl}>: {n} refill region(s), {len(problems)} problem(s)")
        bad += problems
    pfuncs = functions(ntext, "_ZN4s2m214conv_px_kernel")
    if not pfuncs:
        print("check_isa: no conv_px_kernel instantiation found in the assembly")
        return 1
    for name, lines in pfuncs.items():
        m = re.search(r"PxCfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        ch, ntl, mt, nw = (int(g) for g in m.groups())
        problems, n = check_straight_line(name, lines, depth=8, min_mfma=9 * (ch // 16) * ntl * mt, exact=False)
        print(f"check_isa: conv_px_kernel<CH {ch}, NTL {ntl}, MT {mt}, NW {nw}>: {n} refill region(s), {len(problems)} problem(s)")
        bad += problems
    ffuncs = functions(ftext, "_ZN4s2m228feature_fusion_direct_kernel")
    if not ffuncs:
        print("check_isa: no feature_fusion_direct_kernel instantiation found in the assembly")
        return 1
    for name, lines in ffuncs.items():
        problems, n = check_straight_line(name, lines)
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:110]
        print(f"check_isa: {short}: {n} refill region(s), {len(problems)} problem(s)")
        bad += problems
    funcs, cur, name = {}, None, None
    for ln in text:
        m = re.match(r"^(_ZN4s2m216conv_frag_kernel\S*):", ln)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is not None:
            if ln.startswith(".Lfunc_end"):
                cur = None
            else:
                cur.append(ln)
    if not funcs:
        print("check_isa: no conv_frag_kernel instantiation found in the assembly")
        return 1
    for name, lines in funcs.items():
        problems, nloops = check_function(name, lines)
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0][:110]
        print(f"check_isa: {short}: {nloops} K-loop block(s), {len(problems)} problem(s)")
        bad += problems
    for p in bad[:20]:
        print("  " + p)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
