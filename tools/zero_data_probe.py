import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack
from tools.convperiod import period
N,H,W,ci,co,kh,kw=1,256,304,128,128,3,3
for mode in ("random","zero_x","zero_w","zero_both","const_x"):
    x = torch.randn(N,H,W,ci,device="cuda").half()
    w = (torch.randn(co,ci,kh,kw,device="cuda")/math.sqrt(ci*kh*kw)).half()
    if mode in ("zero_x","zero_both"): x.zero_()
    if mode in ("zero_w","zero_both"): w.zero_()
    if mode=="const_x": x.fill_(0.5)
    b = torch.randn(co,device="cuda")
    wf,bp = pack.pack_conv_frag(w,torch.float16,[(ci,ci)]), pack.pack_bias(b,co)
    y = torch.empty(N,H,W,co,device="cuda",dtype=torch.float16)
    for act in (1,0):
        t = period(lambda: hip.conv2d([x],wf,bp,kh,kw,co,act=act,epi=0,korder=2,out=y))
        print(f"{mode:10s} act={act}  {t:7.2f} us")
