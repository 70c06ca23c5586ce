"""CPU-side checks of the drop-in boundary: libs2m2_hip.so loads and exports every symbol of include/s2m2_hip.h.
No compute calls (there is no GPU in the build container); argument validation paths only."""
import ctypes
import os
import re

import pytest

from s2m2_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from s2m2_amd.build import build
    build(verbose=False)
    return hip.load()


def _declared():
    txt = open(os.path.join(ROOT, "include", "s2m2_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^static inline[^\n]*$", "", txt, flags=re.M)       # one-line helpers defined in the header itself (s2m2_conv_frag_chunk): not exports
    return set(re.findall(r"\b(s2m2_[a-z0-9_]+)\s*\(", txt))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/s2m2_hip.h but not exported"
    assert names == set(hip.SIGNATURES)


def test_version_and_error_reporting(lib):
    assert lib.s2m2_version() >= 100
    kd = hip.CorrDesc()
    kd.B, kd.h, kd.w, kd.C, kd.band, kd.token_dtype, kd.cv_dtype = 1, 1, 8, 128, -1, hip.F16, hip.F16
    assert lib.s2m2_cost_volume(ctypes.byref(kd), None) != 0
    assert b"null pointer" in lib.s2m2_last_error()
    assert lib.s2m2_sinkhorn_regress(None, None, None, None, None, 1, 1, 8, 3, 1, 1, 0, None, None) != 0
    assert lib.s2m2_cv_lookup(None, None, None, None, 1, 1, 8, 4, 1, 0, 0, 0, 0, 0, None) != 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libs2m2_hip.so")
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        hip.load()


def test_argument_validation_of_the_gemm_entry_points(lib):
    """Validation returns an error code + message before any device call (no GPU needed): descriptor ranges, supported widths,
    the pre-LayerNorm / dual-GEMM preconditions."""
    import ctypes
    assert lib.s2m2_conv2d(None, None) != 0 and b"null descriptor" in lib.s2m2_last_error()
    d = hip.ConvDesc()
    d.nsrc = 5
    assert lib.s2m2_conv2d(ctypes.byref(d), None) != 0 and b"nsrc" in lib.s2m2_last_error()
    # a plausible 3x3 descriptor with a pre-LayerNorm request: rejected (1x1 layers only)
    d = hip.ConvDesc()
    d.nsrc, d.N, d.H, d.W, d.KH, d.KW, d.Cout, d.out_stride, d.stride = 1, 1, 8, 8, 3, 3, 128, 128, 1
    d.src[0], d.src_c[0], d.src_stride[0] = 4096, 128, 128
    d.weight, d.out, d.ln_wsum, d.ln_eps, d.dtype = 4096, 4096, 4096, 1e-5, hip.F16
    assert lib.s2m2_conv2d(ctypes.byref(d), None) != 0 and b"pre-LayerNorm" in lib.s2m2_last_error()
    d.ln_wsum, d.KH, d.KW, d.epi, d.act, d.aux0, d.aux1, d.ksplit, d.out_scale = None, 1, 1, hip.EPI_DUALMIX, hip.ACT_SIGMOID, 4096, 4096, 40, 1.0
    assert lib.s2m2_conv2d(ctypes.byref(d), None) != 0 and b"DUALMIX" in lib.s2m2_last_error()

    assert lib.s2m2_mlp_chain_supported(128, hip.F16) == 1 and lib.s2m2_mlp_chain_supported(192, hip.F16) == 0
    assert lib.s2m2_mlp_chain_supported(512, hip.F32) == 0
    assert lib.s2m2_mlp_chain(None, None) != 0 and b"null descriptor" in lib.s2m2_last_error()
    c = hip.ChainDesc()
    c.x, c.out, c.C, c.nstage, c.dtype, c.rows, c.x_stride, c.out_stride = 4096, 4096, 128, 4, hip.F16, 8, 128, 128
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"nstage" in lib.s2m2_last_error()
    c.nstage, c.C = 2, 192
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"not supported" in lib.s2m2_last_error()
    c.C, c.carry, c.res_stage = 128, 1, -1
    c.weight[0] = c.weight[1] = 4096
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"carry" in lib.s2m2_last_error()
    # the direct form (weight_frag) exists for fp16 C = 128 / 192 / 256 / 384 / 512 only; the pooled tile load needs it and whole pooled images
    c = hip.ChainDesc()
    c.x, c.out, c.C, c.nstage, c.dtype, c.rows, c.x_stride, c.out_stride, c.res_stage = 4096, 4096, 640, 1, hip.F16, 8, 640, 640, -1
    c.weight[0], c.weight_frag = 4096, 1
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"weight_frag" in lib.s2m2_last_error()
    c.C, c.x_stride, c.out_stride, c.dtype = 128, 128, 128, hip.F32
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"weight_frag" in lib.s2m2_last_error()
    c.dtype, c.weight_frag, c.pool_h, c.pool_w = hip.F16, 0, 4, 4
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"pool_h" in lib.s2m2_last_error()
    c.weight_frag, c.rows = 1, 7                                    # 7 rows are not N * 2 * 2 pooled pixels
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"pool_h" in lib.s2m2_last_error()
    c.rows, c.nstage, c.res_stage, c.res = 8, 1, 0, 4096            # the pooled tile load (ABI 600: chain stages allowed) takes no residual,
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"pooled tile load" in lib.s2m2_last_error()
    c.res_stage, c.nstage, c.nfan, c.xcd_group_rows = -1, 0, 1, 4
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"pooled tile load" in lib.s2m2_last_error()     # ... no XCD placement hint,
    c.xcd_group_rows, c.ln_out = 0, 4096
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"pooled tile load" in lib.s2m2_last_error()     # ... no LayerNorm output
    assert lib.s2m2_version() == hip.ABI_VERSION
    c = hip.ChainDesc()                                            # fan-out only, direct form: nfan 1..4
    c.x, c.C, c.nstage, c.dtype, c.rows, c.x_stride, c.res_stage, c.weight_frag, c.nfan = 4096, 256, 0, hip.F16, 8, 256, -1, 1, 5
    assert lib.s2m2_mlp_chain(ctypes.byref(c), None) != 0 and b"nfan" in lib.s2m2_last_error()

    # round 4: K11 (rectangular direct 1x1): shapes, sources, activation
    assert lib.s2m2_pw_direct_supported(384, 256, hip.F16) == 1 and lib.s2m2_pw_direct_supported(80, 128, hip.F16) == 0
    assert lib.s2m2_pw_direct(None, None) != 0 and b"null descriptor" in lib.s2m2_last_error()
    w = hip.PwDesc()
    w.nsrc, w.rows, w.weight_frag, w.out, w.out_stride, w.Cout, w.dtype = 1, 8, 4096, 4096, 128, 128, hip.F16
    w.src[0], w.src_c[0], w.src_stride[0] = 4096, 12, 128
    assert lib.s2m2_pw_direct(ctypes.byref(w), None) != 0 and b"multiples of 8" in lib.s2m2_last_error()
    w.src_c[0] = 128
    w.act = hip.ACT_SIGMOID
    assert lib.s2m2_pw_direct(ctypes.byref(w), None) != 0 and b"act=" in lib.s2m2_last_error()
    w.act, w.shuffle2 = hip.ACT_NONE, 24
    assert lib.s2m2_pw_direct(ctypes.byref(w), None) != 0 and b"shuffle2" in lib.s2m2_last_error()
    # round 4: K12 (spatial layers on 8- / 16-channel tensors, direct form): shapes, strides, activation
    assert lib.s2m2_conv_narrow_supported(3, 3, 1, 8, 160, hip.F16) == 1 and lib.s2m2_conv_narrow_supported(5, 5, 2, 16, 64, hip.F16) == 1
    assert lib.s2m2_conv_narrow_supported(5, 5, 2, 16, 32, hip.F16) == 0 and lib.s2m2_conv_narrow_supported(3, 3, 1, 16, 32, hip.F16) == 0
    assert lib.s2m2_conv_narrow_supported(3, 3, 1, 8, 32, hip.F32) == 0 and lib.s2m2_conv_narrow_supported(3, 3, 1, 8, 12, hip.F16) == 0
    assert lib.s2m2_conv_narrow(None, None) != 0 and b"null descriptor" in lib.s2m2_last_error()
    nd = hip.NarrowDesc()
    nd.x, nd.x_stride, nd.N, nd.H, nd.W, nd.Cin, nd.weight_frag, nd.out, nd.out_stride = 4096, 8, 1, 8, 8, 8, 4096, 4096, 32
    nd.Cout, nd.KH, nd.KW, nd.stride, nd.act, nd.dtype = 32, 3, 3, 2, hip.ACT_NONE, hip.F16
    assert lib.s2m2_conv_narrow(ctypes.byref(nd), None) != 0 and b"not supported" in lib.s2m2_last_error()       # 3x3 with a stride
    nd.stride, nd.x_stride = 1, 12
    assert lib.s2m2_conv_narrow(ctypes.byref(nd), None) != 0 and b"multiples of 8" in lib.s2m2_last_error()
    nd.x_stride, nd.out_stride = 16, 24
    assert lib.s2m2_conv_narrow(ctypes.byref(nd), None) != 0 and b"out_stride" in lib.s2m2_last_error()          # narrower than Cout
    nd.out_stride, nd.act = 32, hip.ACT_TANH
    assert lib.s2m2_conv_narrow(ctypes.byref(nd), None) != 0 and b"act=" in lib.s2m2_last_error()
    nd.act, nd.head_cout, nd.head_frag = hip.ACT_RELU, 16, 4096                                                  # the fused head: 48-channel form only
    assert lib.s2m2_conv_narrow(ctypes.byref(nd), None) != 0 and b"fused 1x1 head" in lib.s2m2_last_error()
    # K1 through one descriptor (the positional ABI 400 entry points were removed with ABI 600)
    assert lib.s2m2_cost_volume(None, None) != 0 and b"null descriptor" in lib.s2m2_last_error()
    kd = hip.CorrDesc()
    kd.tokens, kd.cv, kd.B, kd.h, kd.w, kd.C, kd.band, kd.token_dtype, kd.cv_dtype = 4096, 4096, 1, 2, 12, 128, -1, hip.F16, hip.F16
    assert lib.s2m2_cost_volume(ctypes.byref(kd), None) != 0 and b"multiple of 8" in lib.s2m2_last_error()
    kd.w, kd.cv_pitch = 16, 12
    assert lib.s2m2_cost_volume(ctypes.byref(kd), None) != 0 and b"cv_pitch" in lib.s2m2_last_error()
    kd.cv_pitch, kd.ln_weight = 0, 4096
    assert lib.s2m2_cost_volume(ctypes.byref(kd), None) != 0 and b"come together" in lib.s2m2_last_error()
    kd.ln_weight, kd.band = None, -2
    assert lib.s2m2_cost_volume(ctypes.byref(kd), None) != 0 and b"band" in lib.s2m2_last_error()
    kd.band, kd.token_dtype = -1, hip.F32
    assert lib.s2m2_cost_volume(ctypes.byref(kd), None) != 0 and b"dtype pair" in lib.s2m2_last_error()
    assert lib.s2m2_feature_fusion_supported(128, hip.F16) == 1 and lib.s2m2_feature_fusion_supported(384, hip.F16) == 0
    assert lib.s2m2_feature_fusion(4096, 4096, 4096, 128, 128, 128, 64, 192, 4096, 4096, 4096, 4096, 4096, 0, 0, hip.F16, None) != 0
    assert b"not supported" in lib.s2m2_last_error()
    assert lib.s2m2_feature_fusion(4096, 4096, 4096, 128, 128, 128, 60, 128, 4096, 4096, 4096, 4096, 4096, 4, 4, hip.F16, None) != 0
    assert b"whole number" in lib.s2m2_last_error()
    assert lib.s2m2_feature_fusion_frag_supported(256, hip.F16) == 1 and lib.s2m2_feature_fusion_frag_supported(512, hip.F16) == 1 and lib.s2m2_feature_fusion_frag_supported(128, hip.F32) == 0
    assert lib.s2m2_feature_fusion_frag(4096, 4096, 4096, 128, 128, 128, 64, 128, 4096, 4096, 4096, 4096, 0, 0, hip.F32, None) != 0
    assert b"not supported" in lib.s2m2_last_error()
    assert lib.s2m2_mlp_chain_frag_supported(128, hip.F16) == 1 and lib.s2m2_mlp_chain_frag_supported(384, hip.F16) == 1 and lib.s2m2_mlp_chain_frag_supported(512, hip.F16) == 1 and lib.s2m2_mlp_chain_frag_supported(640, hip.F16) == 0
    assert lib.s2m2_stem_mlp(None, None, None, None, None, None, 8, hip.F16, None) != 0


def test_attention_planning_query_runs_without_a_device(lib):
    """s2m2_attention_supported: the launch planner (head-dim instantiation, PE bin tiles, waves per block, LDS budget) is host code."""
    import torch
    F16, F32 = torch.float16, torch.float32
    assert hip.attention_supported(2, 8, 1216, 32, F16, grid=(38, 32))[0]             # S 1216x1024
    assert hip.attention_supported(2, 8, 4864, 96, F32, grid=(76, 64))[0]             # XL 2432x2048, parity mode: the (3, 2) bin tiles
    assert hip.attention_supported(4, 8, 63 * 64, 32, F16, grid=(64, 63))[0]          # B = 2 at 2048x2016: waves per block clamped
    assert hip.attention_supported(512, 1, 304, 128, F16)[0]
    ok, why = hip.attention_supported(2, 8, 5000, 32, F16, grid=(100, 50))
    assert not ok and "96 x 96" in why
    ok, why = hip.attention_supported(2, 8, 9216, 96, F32, grid=(96, 96))
    assert not ok and "LDS" in why
    from s2m2_amd.engine import check_limits
    check_limits(2048, 2432, 384, 1, F32)
    with pytest.raises(ValueError, match="token grid"):
        check_limits(1600, 3200, 128, 1, F16)


def test_descriptor_structs_have_the_layout_of_the_header(tmp_path):
    """the ctypes mirrors of the descriptor structs (s2m2_amd/hip.py) against include/s2m2_hip.h compiled by gcc: sizes and the offsets of the
    last fields -- a field added to one side only shifts everything behind it"""
    import ctypes
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [("s2m2_conv_desc", hip.ConvDesc), ("s2m2_chain_desc", hip.ChainDesc), ("s2m2_pw_desc", hip.PwDesc), ("s2m2_narrow_desc", hip.NarrowDesc),
             ("s2m2_corr_desc", hip.CorrDesc)]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "s2m2_hip.h"', "int main(void) {"]
    for cname, py in pairs:
        last = py._fields_[-1][0]
        lines.append(f'  printf("{cname} %zu %zu\\n", sizeof({cname}), offsetof({cname}, {last}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for (cname, py), line in zip(pairs, out):
        name, size, off = line.split()
        assert name == cname and int(size) == ctypes.sizeof(py), (cname, size, ctypes.sizeof(py))
        assert int(off) == getattr(py, py._fields_[-1][0]).offset, (cname, off)


def test_conv_frag_chunk_rule_of_the_header_is_the_rule_of_the_packers(tmp_path):
    """s2m2_conv_frag_chunk (static inline in include/s2m2_hip.h, compiled by gcc) against pack.frag_chunk on a grid of layer shapes: the
    kernel dispatcher, the library's packer and the Python packer all chunk the K-order-2 stream by this one rule"""
    import shutil
    import subprocess
    from s2m2_amd import pack
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = [(co, ci) for co in (64, 128, 192, 256, 320, 384, 576, 768) for ci in (64, 128, 192, 256, 320, 384, 576)]
    body = "".join(f'  printf("%d\\n", s2m2_conv_frag_chunk({co}, {ci}));\n' for co, ci in shapes)
    src = tmp_path / "chunk.c"
    src.write_text('#include <stdio.h>\n#include "s2m2_hip.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "chunk"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out == [pack.frag_chunk(co, ci) for co, ci in shapes]
    assert pack.frag_chunk(192, 192) == 192 and pack.frag_chunk(192, 384) == 192 and pack.frag_chunk(384, 384) == 128


def test_plan_pointer_masks_list_every_pointer_field_of_the_recorded_descriptors():
    """Plans (s2m2_plan_*, csrc/plan.h) relocate external pointers in the POINTER words of a recorded call only; the descriptors that go through
    plans name those words by hand (S2M2_PLAN_PTRS).  A pointer field added to include/s2m2_hip.h without its entry would silently stop following
    the external buffers on replay: the lists are checked against the header here."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "s2m2_hip.h")).read()
    plan_h = open(os.path.join(root, "s2m2_amd", "csrc", "plan.h")).read()
    recorded = set()
    for src in os.listdir(os.path.join(root, "s2m2_amd", "csrc")):
        if src.endswith(".hip"):
            recorded |= set(re.findall(r"plan_dispatch_desc<(s2m2_\w+)>", open(os.path.join(root, "s2m2_amd", "csrc", src)).read()))
    assert recorded >= {"s2m2_conv_desc", "s2m2_chain_desc", "s2m2_corr_desc", "s2m2_rowattn_desc", "s2m2_convblock_desc", "s2m2_pw_desc", "s2m2_narrow_desc"}
    for name in sorted(recorded):
        body = header[header.index("typedef struct %s {" % name):header.index("} %s;" % name)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        assert not re.search(r"\*\s*\w+\s*,", body), f"{name}: one pointer declarator per line, please (this check parses them)"
        fields = set()
        for m in re.finditer(r"\*\s*(\w+)\s*(\[(\d+)\])?\s*;", body):
            fields |= {(m.group(1), i) for i in range(int(m.group(3)))} if m.group(3) else {(m.group(1), None)}
        blk = plan_h[plan_h.index("S2M2_PLAN_PTRS(%s," % name):]
        blk = blk[:blk.index("))\n") + 2]
        listed = {(m.group(2), int(m.group(4)) if m.group(4) else None) for m in re.finditer(r"S2M2_OFF(_I)?\(%s, (\w+)(, (\d))?\)" % name, blk)}
        assert fields == listed, (name, fields ^ listed)
