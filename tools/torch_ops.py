"""List the PyTorch ops (aten::*) that still run inside one eager forward (they should be allocation-only)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2M2_GRAPH"] = "0"
from s2m2_amd.model import build_model
from s2m2_amd.weights import noise_pair
m = build_model("S", True, 3).cuda().eval()
l, r = noise_pair(1024, 1216, 1, 0)
l, r = l.cuda(), r.cuda()
with torch.autocast("cuda", dtype=torch.float16):
    m(l, r); m(l, r)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
        m(l, r)
        torch.cuda.synchronize()
ka = prof.key_averages()
print("device kernels that are not ours:")
for e in sorted(ka, key=lambda e: -e.self_device_time_total):
    if e.self_device_time_total > 0 and "s2m2" not in e.key:
        print("  %-90s calls %4d  device %8.1f us" % (e.key[:90], e.count, e.self_device_time_total))
print("aten ops by call site (ops that launch device work):")
cnt, tim = collections.Counter(), collections.Counter()
for e in prof.key_averages(group_by_stack_n=12):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        site = next((s for s in e.stack if "s2m2_amd" in s), e.stack[0] if e.stack else "?")
        k = (e.key, site.split("/")[-1][:90])
        cnt[k] += e.count
        tim[k] += e.self_device_time_total
for k, v in sorted(cnt.items(), key=lambda kv: -tim[kv[0]]):
    print("  %4d  %8.1f us  %s  @ %s" % (v, tim[k], k[0], k[1]))
