#!/usr/bin/env python3
"""EXPERIMENT (csrc/convpipe.hip): the pipelined persistent fragment-stream convolution against K5 v5 on the same operands -- results and time.
    python tools/convpipe_bench.py        (prints one line per shape; run it under `timeout`: a protocol error between the waves would hang)"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402

lib = hip.load()
lib.s2m2_debug_conv_pipe.restype = ctypes.c_int
vp, ll, i32 = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
lib.s2m2_debug_conv_pipe.argtypes = [vp, ll, vp, ll, vp, ll, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]


def frag64(w):
    """(Cout, Cin, 3, 3) -> the stream [Cout/32][chunk of 64][tap][k16 step][lane][8] (pack.pack_conv_frag's torch formula with CK = 64)"""
    wp = pack.pack_conv(w, torch.float32)
    cop, ntap = wp.shape[0], 9
    cin = wp.shape[1] // ntap
    ck, ks = 64, 4
    t = wp.reshape(cop, ntap, cin).reshape(cop // 32, 32, ntap, cin // ck, ks, 2, 8).permute(0, 3, 2, 4, 5, 1, 6)
    return t.reshape(-1).half().contiguous()


def time_us(f, reps=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


shapes = [(1, 32, 64, 128, 1, False, 32), (1, 256, 256, 128, 1, False, 32), (1, 256, 256, 128, 0, True, 32), (2, 256, 256, 128, 1, False, 32), (4, 256, 256, 128, 1, False, 32), (1, 256, 304, 128, 1, False, 40), (1, 256, 304, 128, 0, True, 40), (2, 256, 304, 128, 1, False, 40), (1, 256, 304, 128, 1, False, 32),
          (2, 512, 608, 128, 1, False, 32), (1, 256, 304, 256, 1, False, 40)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for N, H, W, cout, act, add, pw in shapes:
    g = torch.Generator(device="cuda").manual_seed(H + cout)
    x = torch.randn(N, H, W, 128, device="cuda", generator=g).half()
    w = (torch.randn(cout, 128, 3, 3, device="cuda", generator=g) / math.sqrt(128 * 9)).half()
    b = pack.pack_bias(torch.randn(cout, device="cuda", generator=g) * 0.3, cout)
    aux = torch.randn(N, H, W, cout, device="cuda", generator=g).half() if add else None
    wf5 = pack.pack_conv_frag(w, torch.float16)
    wf6 = frag64(w)
    kw = dict(act=hip.ACT_GELU if act else hip.ACT_NONE, korder=2)
    if add:
        kw.update(epi=hip.EPI_ADD, aux0=aux)
    ref = hip.conv2d([x], wf5, b, 3, 3, cout, **kw)
    out = torch.full_like(ref, float("nan"))

    def pipe():
        rc = lib.s2m2_debug_conv_pipe(x.data_ptr(), 128, out.data_ptr(), cout, aux.data_ptr() if add else None, cout, N, H, W, 128, cout,
                                      wf6.data_ptr(), b.data_ptr(), 1 if act else 0, pw, None)
        assert rc == 0, lib.s2m2_last_error()

    pipe()
    torch.cuda.synchronize()
    err = float((out.float() - ref.float()).abs().max())
    t5 = time_us(lambda: hip.conv2d([x], wf5, b, 3, 3, cout, **kw))
    t6 = time_us(pipe)
    print(f"N={N} {H}x{W} 128->{cout} act={act} add={int(add)} pw={pw}: max |pipe - v5| {err:.4f} (ref max {float(ref.float().abs().max()):.2f})   v5 {t5:7.2f} us   pipe {t6:7.2f} us   {t6 / t5:.3f}", flush=True)
