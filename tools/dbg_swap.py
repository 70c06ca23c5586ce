import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip
g = torch.Generator().manual_seed(7)
for dt in (torch.float16, torch.float32):
    for (h, w, C) in ((4, 304, 128), (4, 64, 128), (4, 32, 64)):
        feat = torch.randn(2, h, w, C, generator=g).cuda().to(dt)
        gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
        beta = (0.05 * torch.randn(C, generator=g)).cuda()
        a = hip.ln_corr(feat, gamma, beta, torch.float32)
        b = hip.ln_corr(feat.flip(0).contiguous(), gamma, beta, torch.float32).transpose(2, 3)
        ne = (a != b)
        print(dt, h, w, C, "mismatch frac", float(ne.float().mean()), "maxdiff", float((a - b).abs().max()),
              "rows with mismatch", ne.any(3).any(0).sum(1).tolist(), "cols", int(ne.any(2).any(0).sum()))
        # identical left/right: must be exactly symmetric too
        f2 = torch.cat([feat[:1], feat[:1]], 0).contiguous()
        s = hip.ln_corr(f2, gamma, beta, torch.float32)
        print("   self-symmetric mismatch", float((s != s.transpose(2, 3)).float().mean()))
