"""Conv ablation: times one conv shape for experiment builds libs2m2_hip_cdbg<N>.so (compile-time S2M2_CONV_DBG=N).
Build:  for d in 2 8 16 32 64; do S2M2_LIB_SUFFIX=_cdbg$d S2M2_BUILD_DEFINES=-DS2M2_CONV_DBG=$d python -m s2m2_amd.build; done"""
import os, sys, subprocess
for suffix in sys.argv[1:] or [""]:
    env = dict(os.environ, S2M2_LIB_SUFFIX=suffix)
    code = ("import torch,sys,math; sys.path.insert(0,'.'); from s2m2_amd import hip, pack; from tools.kbench import timeit;"
            "x=torch.randn(1,256,304,128,device='cuda').half(); w=(torch.randn(128,128,3,3,device='cuda')/34).half();"
            "wp=pack.pack_conv(w,torch.float16); bp=pack.pack_bias(torch.zeros(128,device='cuda'),128);"
            "r=[timeit(lambda: hip.conv2d([x],wp,bp,3,3,128,act=1,tile=t),30) for t in (1,2)];"
            "print('lib%-8s 3x3 128->128 @1/4: tile128 %.1f us  tile64 %.1f us' % (sys.argv[1], r[0], r[1]))")
    subprocess.run([sys.executable, "-c", code, suffix], env=env)
