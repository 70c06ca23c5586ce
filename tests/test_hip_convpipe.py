"""GPU test of the EXPERIMENTAL pipelined fragment-stream convolution (csrc/convpipe.hip, entry point s2m2_debug_conv_pipe -- measured in
profiles/r06/convpipe_bench.txt, not dispatched by the engine): same values as K5 v5 up to the K order of the fp32 accumulation."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W,cout,act,add,pw", [(1, 32, 64, 128, 1, False, 32), (2, 37, 50, 256, 0, True, 32), (1, 20, 90, 128, 2, True, 40), (3, 9, 33, 128, 1, False, 40)])
def test_conv_pipe_matches_v5(N, H, W, cout, act, add, pw):
    from s2m2_amd import hip, pack
    lib = hip.load()
    vp, ll, i32 = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    lib.s2m2_debug_conv_pipe.restype = ctypes.c_int
    lib.s2m2_debug_conv_pipe.argtypes = [vp, ll, vp, ll, vp, ll, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]
    g = torch.Generator(device="cuda").manual_seed(H + W + cout)
    x = torch.randn(N, H, W, 128, device="cuda", generator=g).half()
    w = (torch.randn(cout, 128, 3, 3, device="cuda", generator=g) / math.sqrt(128 * 9)).half()
    b = pack.pack_bias(torch.randn(cout, device="cuda", generator=g) * 0.3, cout)
    aux = torch.randn(N, H, W, cout, device="cuda", generator=g).half() if add else None
    kw = dict(act=act, korder=2)
    if add:
        kw.update(epi=hip.EPI_ADD, aux0=aux)
    ref = hip.conv2d([x], pack.pack_conv_frag(w, torch.float16), b, 3, 3, cout, **kw)
    wp = pack.pack_conv(w, torch.float32)                         # the stream in chunks of 64 channels: (cout tile, chunk, tap, k16 step, half, cout, e)
    wf = wp.reshape(cout, 9, 128).reshape(cout // 32, 32, 9, 2, 4, 2, 8).permute(0, 3, 2, 4, 5, 1, 6).reshape(-1).half().contiguous()
    out = torch.full_like(ref, float("nan"))
    for _ in range(3):
        rc = lib.s2m2_debug_conv_pipe(x.data_ptr(), 128, out.data_ptr(), cout, aux.data_ptr() if add else None, cout, N, H, W, 128, cout,
                                      wf.data_ptr(), b.data_ptr(), act, pw, None)
        assert rc == 0, lib.s2m2_last_error()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert float((out.float() - ref.float()).abs().max()) < 1.5e-2
