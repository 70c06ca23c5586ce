"""Stage-by-stage comparison of the HIP forward with the CPU oracle (test infrastructure; used by tests/test_hip_parity_*.py
and tools/parity_report.py, which writes the error tables committed under profiles/).

Every stage boundary that both sides capture is compared: the reference's own intermediate tensors of
``S2M2.forward`` (s2m2.py:136-197) in the order they are produced.  The per-stage statistics are

* ``max``, ``p99.9``   absolute error
* ``frac_out``         fraction of elements with |err| > atol + rtol * |ref|   (atol = 1e-3, rtol = 1e-4: north_star's 1e-3 px
                       on the disparity, relaxed by 1e-4 relative because disparities reach hundreds of px)
* ``n_abs``            number of elements with |err| > atol alone (the plain 1e-3 of north_star, no relative term), so that the
                       relaxation is visible in every table; tools/parity_report.py prints the reference's own 1-vs-32-thread
                       noise at the same sizes beside it
* ``scale``            mean |ref|, to read the absolute numbers against

and, for the integer argmax, the agreement on the pixels where the oracle's top-2 relative gap exceeds ``GAP`` (elsewhere fp32
summation order legitimately decides) -- there it must be bit exact.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

ATOL, RTOL, GAP = 1e-3, 1e-4, 1e-4

# (name, kind): kind "disp" = pixel-valued map, "prob" = [0,1] map, "feat" = feature tensor / logits / correlation values
STAGE_ORDER = [("feature_py_4x", "feat"), ("feature_tr_4x", "feat"), ("cv", "feat"), ("disp0", "disp"), ("conf0", "prob"),
               ("occ0", "prob"), ("disp_g", "disp"), ("ctx", "feat")]
FINAL = [("hidden", "feat"), ("mask4x", "feat"), ("disp_up4", "disp"), ("mask1x", "feat"), ("disp", "disp"), ("occ", "prob"),
         ("conf", "prob")]


def stage_list(refine_iter: int) -> List[Tuple[str, str]]:
    it = []
    for i in range(refine_iter):
        it += [(f"corr1_it{i}", "feat"), (f"corr2_it{i}", "feat"), (f"disp_it{i}", "disp"), (f"conf_it{i}", "prob"),
               (f"occ_it{i}", "prob")]
    return STAGE_ORDER + it + FINAL


def stats(test: torch.Tensor, ref: torch.Tensor, atol: float = ATOL, rtol: float = RTOL) -> Dict[str, float]:
    t, r = test.detach().float().cpu(), ref.detach().float().cpu()
    if t.shape != r.shape:
        raise AssertionError(f"shape mismatch {tuple(t.shape)} vs {tuple(r.shape)}")
    e = (t - r).abs().reshape(-1)
    n = e.numel()
    k = max(1, int(round(0.999 * n)))
    p999 = float(e.kthvalue(k).values) if n > 1 else float(e.max())
    return dict(max=float(e.max()), p999=p999, median=float(e.median()), p99=float(e.kthvalue(max(1, int(round(0.99 * n)))).values),
                frac_out=float((e > atol + rtol * r.abs().reshape(-1)).float().mean()),
                n_abs=int((e > atol).sum()),                 # elements beyond the PLAIN absolute tolerance (north_star: 1e-3), no relative term
                scale=float(r.abs().mean()), n=n, finite=bool(torch.isfinite(t).all()))


def argmax_stats(hip_idx: torch.Tensor, ora_idx: torch.Tensor, prob: torch.Tensor, gap: float = GAP) -> Dict[str, float]:
    """prob: the oracle's masked transport probabilities (B,h,w,w).  'sure' pixels: (p1 - p2) > gap * p1."""
    top = prob.topk(2, dim=3).values
    sure = (top[..., 0] - top[..., 1]) > gap * top[..., 0]
    same = hip_idx.cpu().long() == ora_idx.cpu().long()
    return dict(agree_all=float(same.float().mean()), sure_frac=float(sure.float().mean()),
                agree_sure=float(same[sure].float().mean()) if bool(sure.any()) else 1.0,
                mismatch_sure=int((~same[sure]).sum()))


def compare(hip_cap: Dict[str, torch.Tensor], hip_out: Sequence[torch.Tensor], ora_cap: Dict[str, torch.Tensor],
            ora_out: Sequence[torch.Tensor], refine_iter: int) -> Tuple[List[Tuple[str, str, Dict[str, float]]], Dict[str, float]]:
    h = dict(hip_cap)
    o = dict(ora_cap)
    for k, th, to in zip(("disp", "occ", "conf"), hip_out, ora_out):
        h[k], o[k] = th, to
    rows = []
    for name, kind in stage_list(refine_iter):
        if name in h and name in o:
            rows.append((name, kind, stats(h[name], o[name])))
    am = argmax_stats(h["argmax"], o["argmax"], o["prob"])
    return rows, am


def format_table(title: str, rows, am, extra: Optional[str] = None) -> str:
    out = [title, f"tolerance per element: |err| <= {ATOL:g} + {RTOL:g}*|ref|;  argmax 'sure' pixels: oracle top-2 relative gap > {GAP:g}",
           f"{'stage':<16}{'elements':>11}{'mean|ref|':>12}{'median err':>12}{'p99 err':>12}{'p99.9 err':>12}{'max err':>12}{'frac > tol':>12}{'n > 1e-3 abs':>14}"]
    for name, kind, s in rows:
        out.append(f"{name:<16}{s['n']:>11d}{s['scale']:>12.4g}{s['median']:>12.3e}{s['p99']:>12.3e}{s['p999']:>12.3e}{s['max']:>12.3e}"
                   f"{s['frac_out']:>12.3e}{s['n_abs']:>14d}")
    out.append(f"argmax: agreement {am['agree_all']:.6f} on all pixels; 'sure' pixels {am['sure_frac']:.4f} of all, "
               f"agreement there {am['agree_sure']:.6f} ({am['mismatch_sure']} mismatches)")
    if extra:
        out.append(extra)
    return "\n".join(out) + "\n"


def hip_forward(sd, C: int, ntr: int, refine_iter: int, left: torch.Tensor, right: torch.Tensor, fp16: bool,
                inject: Optional[Dict[str, torch.Tensor]] = None, use_positivity: bool = True):
    """One captured forward of the product on cuda:0 -> (outputs, captures), both on the CPU.  ``inject``: stage-boundary tensors
    of the checker to continue from (Engine.finish / Engine.run, parity tests only)."""
    from s2m2_amd.model import S2M2
    m = S2M2(C, 1, ntr, use_positivity=use_positivity, refine_iter=refine_iter)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    cap: Dict[str, object] = {}
    if inject:
        cap["inject"] = inject
    with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
        out = m(left.cuda(), right.cuda(), capture=cap)
    torch.cuda.synchronize()
    cap.pop("inject", None)
    return [t.float().cpu() for t in out], {k: v.detach().cpu() for k, v in cap.items()}


def select(rows, names) -> Dict[str, Dict[str, float]]:
    d = {n: s for n, _, s in rows}
    return {n: d[n] for n in names if n in d}


def lookup_from_checker_disparity(ocap: Dict[str, torch.Tensor], refine_iter: int) -> Dict[str, float]:
    """K3 judged at OPERATOR tolerance inside an end-to-end comparison: the lookups of iteration k are recomputed by the HIP kernel
    from the checker's OWN cost volume and the checker's own disparity entering iteration k (disp_g, then disp_it{k-1}), so that the
    disparity error of the stages upstream (times a cost slope of ~100 per px) is not part of what is compared.  -> max |err| per map."""
    from s2m2_amd import hip
    out = {}
    cv = ocap["cv"].float().cuda()
    for it in range(refine_iter):
        d = (ocap["disp_g"] if it == 0 else ocap[f"disp_it{it - 1}"]).float().cuda()
        c1, c2 = hip.cv_lookup(cv, d, 4)
        out[f"corr1_it{it}"] = float((c1.cpu() - ocap[f"corr1_it{it}"].float()).abs().max())
        out[f"corr2_it{it}"] = float((c2.cpu() - ocap[f"corr2_it{it}"].float()).abs().max())
    return out


def sharp_tokens(C, h, w, shifts, seed, noise=0.05):
    """Synthetic transformer output (2,C,h,w) with ONE unambiguous match per left pixel: left tokens iid N(0,1) per
    channel, right token j = left token (j + d) mod w + noise, d = shifts[row band] -- after LayerNorm the matched score is ~C (128)
    against N(0, sqrt C) for every other column, so neither fp16 rounding of the volume (ulp 0.125 at 128) nor summation order can
    move an argmax.  Circular shift: every pixel has its match (run without the positivity mask, wrapped matches are negative
    disparities)."""
    g = torch.Generator().manual_seed(seed)
    left = torch.randn(1, C, h, w, generator=g)
    right = torch.empty_like(left)
    band = (h + len(shifts) - 1) // len(shifts)
    for k, d in enumerate(shifts):
        rows = slice(k * band, min(h, (k + 1) * band))
        right[:, :, rows] = torch.roll(left[:, :, rows], shifts=-d, dims=3)
    right = right + noise * torch.randn(1, C, h, w, generator=g)
    return torch.cat([left, right], 0)
