"""K1 A/B: pipelined (S2M2_LNCORR_PIPE=1) vs plain (default), c3/c2 fp16, back-to-back timing."""
import os, sys, subprocess
for env_extra in ({"S2M2_LNCORR_PIPE": "1"}, {}):
    env = dict(os.environ, **env_extra)
    code = ("import torch,sys; sys.path.insert(0,'.'); from s2m2_amd import hip; from tools.kbench import timeit;\n"
            "for name,C,h,w in (('c3',128,256,304),('c2',128,120,160),('c4',256,256,304)):\n"
            "    f=torch.randn(2,h,w,C,device='cuda').half(); g=torch.ones(C,device='cuda'); b=torch.zeros(C,device='cuda')\n"
            "    cv=torch.empty(1,h,w,w,device='cuda',dtype=torch.half)\n"
            "    t=min(timeit(lambda: hip.ln_corr(f,g,b,out=cv),50) for _ in range(3))\n"
            "    by=2*h*w*C*2+h*w*w*2\n"
            "    print('%s %s: %.1f us  %.0f GB/s  frac %.3f' % (sys.argv[1], name, t, by/t/1e3, by/t/1e3/8000))\n")
    subprocess.run([sys.executable, "-c", code, "pipe  " if env_extra else "plain "], env=env)
