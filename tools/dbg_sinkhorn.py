import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip
from oracle import s2m2_oracle as O
for name in ["op_dispinit_pos", "op_dispinit_neg"]:
    g = np.load(f"tests/golden/{name}.npz")
    pos = bool(g["cfg"][4])
    cv = torch.from_numpy(g["cv"])
    disp, conf, occ, am = [x.cpu() for x in hip.sinkhorn_regress(cv.cuda(), pos, 3, want_argmax=True)]
    for k, a in (("disp", disp), ("conf", conf), ("occ", occ)):
        e = (a - torch.from_numpy(g[k])).abs()
        idx = np.unravel_index(int(e.argmax()), e.shape)
        print(name, k, "maxerr", float(e.max()), "at", idx, "mine", float(a[idx]), "ref", float(torch.from_numpy(g[k])[idx]), "n>1e-4", int((e > 1e-4).sum()))
    print(" argmax mism", int((am != torch.from_numpy(g["argmax"])).sum()))
    e = (conf - torch.from_numpy(g["conf"])).abs()[:, 0]
    bad = torch.nonzero(e > 1e-4)
    print(" bad conf idx", bad[:20].tolist(), "argmax there", [int(am[tuple(b)]) for b in bad[:20]])
