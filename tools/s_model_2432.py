import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from s2m2_amd.model import build_model
from s2m2_amd.weights import noise_pair
H, W = 2048, 2432
m = build_model("S", use_positivity=True, refine_iter=3).cuda().eval()
l, r = noise_pair(H, W, 1, 0); l, r = l.cuda(), r.cuda()
with torch.autocast("cuda", dtype=torch.float16):
    for _ in range(3): out = m(l, r)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = m(l, r)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("S 2432x2048 ROWFUSE=%s: %.2f ms/pair finite=%s" % (os.environ.get("S2M2_ROWFUSE", "1"), dt * 1e3, all(bool(torch.isfinite(o).all()) for o in out)))
