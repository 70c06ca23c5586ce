import os, sys, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from s2m2_amd.model import S2M2
from s2m2_amd.weights import seeded_state_dict, synthetic_pair
sd = seeded_state_dict(128, 1, 1, 0)
def build():
    m = S2M2(128, 1, 1, use_positivity=True, refine_iter=3); m.load_state_dict(sd, strict=True); return m.cuda().eval()
pairs = [tuple(t.cuda() for t in synthetic_pair(96, 160, 1, 8 + 4 * k, k)) for k in range(4)]
os.environ["S2M2_GRAPH"] = "0"; os.environ["S2M2_REFINE_NATIVE"] = "0"
rm = build()
with torch.autocast("cuda", dtype=torch.float16):
    ref = [tuple(o.clone() for o in rm(l, r)) for l, r in pairs]
os.environ["S2M2_REFINE_NATIVE"] = "1"
nm = build()
# same pair repeatedly first
with torch.autocast("cuda", dtype=torch.float16):
    for rep in range(4):
        out = nm(*pairs[0])
        print("same pair rep", rep, [bool(torch.equal(a, b)) for a, b in zip(out, ref[0])], [float((a - b).abs().max()) for a, b in zip(out, ref[0])])
    for k in range(4):
        out = nm(*pairs[k])
        print("pair", k, [bool(torch.equal(a, b)) for a, b in zip(out, ref[k])], [float((a - b).abs().max()) for a, b in zip(out, ref[k])])
eng = next(iter(nm._engines.values()))
for k_, v in {**eng._bufs, **eng._plans}.items():
    if isinstance(k_, tuple) and k_[0] == "refine_plan":
        print(k_[1], type(v), v[0].launches if isinstance(v, tuple) else v, [v[0].patches(i) for i in range(7)] if isinstance(v, tuple) else '')
os.environ["S2M2_GRAPH"] = "1"
try:
    gm = build()
    with torch.autocast("cuda", dtype=torch.float16):
        for k in range(4):
            out = gm(*pairs[k])
            print("graph pair", k, [bool(torch.equal(a, b)) for a, b in zip(out, ref[k])])
except Exception:
    traceback.print_exc()
