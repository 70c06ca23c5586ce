"""List the PyTorch ops (aten::*) that still run inside one eager forward (they should be allocation-only)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2M2_GRAPH"] = "0"
from s2m2_amd.model import build_model
from s2m2_amd.weights import noise_pair
m = build_model("S", True, 3).cuda().eval()
l, r = noise_pair(256, 320, 1, 0)
l, r = l.cuda(), r.cuda()
with torch.autocast("cuda", dtype=torch.float16):
    m(l, r); m(l, r)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
        m(l, r)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.device_time_total > 0 and e.stack:
        site = next((s for s in e.stack if "s2m2_amd" in s), e.stack[0] if e.stack else "?")
        cnt[(e.name, site.split("/")[-1][:80])] += 1
for k, v in cnt.most_common(40):
    print(v, k)
