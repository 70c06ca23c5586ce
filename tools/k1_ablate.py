"""K1 ablation: times c3 fp16 ln_corr for experiment builds libs2m2_hip_dbg<N>.so (compile-time S2M2_LNCORR_DBG=N).
Build them first:  for d in 1 2 3 4 8 12; do S2M2_LIB_SUFFIX=_dbg$d S2M2_BUILD_DEFINES=-DS2M2_LNCORR_DBG=$d python -m s2m2_amd.build; done"""
import os, sys, subprocess
combos = [x.split(":") for x in sys.argv[1:]] or [["", "1"]]
for suffix, ns in combos:
    env = dict(os.environ, S2M2_LIB_SUFFIX=suffix, S2M2_LNCORR_NSTRIP=str(ns))
    code = ("import torch,sys; sys.path.insert(0,'.'); from s2m2_amd import hip; from tools.kbench import timeit;"
            "f=torch.randn(2,256,304,128,device='cuda').half(); g=torch.ones(128,device='cuda'); b=torch.zeros(128,device='cuda');"
            "cv=torch.empty(1,256,304,304,device='cuda',dtype=torch.half);"
            "lib=hip.load(); st=torch.cuda.current_stream().cuda_stream;"
            "fn=lambda: lib.s2m2_ln_corr(f.data_ptr(),g.data_ptr(),b.data_ptr(),cv.data_ptr(),1,256,304,128,1,1,st);"
            "print('lib%-8s nstrip=%s  %.1f us' % (sys.argv[1], sys.argv[2], timeit(fn, 50)))")
    subprocess.run([sys.executable, "-c", code, suffix, str(ns)], env=env)
