"""Drop-in counterparts of the reference's driver glue around ``S2M2.forward`` (src/s2m2/core/utils/model_utils.py,
image_utils.py) -- SURVEY.md section 8f rows 1-2: same names, arguments and return values, pre/post-processing on the device.

* ``load_model``           model_utils.py:11-48 (model-size table, ``CH{C}NTR{n}.pth`` checkpoint, ``my_load_state_dict``)
* ``image_pad``            image_utils.py:27-71 (HIP kernel ``s2m2_image_pad``)
* ``image_crop``           image_utils.py:73-103 (pure slicing)
* ``run_stereo_matching``  model_utils.py:51-95 (pad -> autocast fp16 forward -> crop -> average confidence)
* ``compute_confidence_score``   model_utils.py:98-101
* ``compute_confidence_scores``  the calibration objective (calibration/base.py:15-36 called 20 x 5 times one pair at a time by
  calibration/cem.py:66-72) for a whole population of rectified pairs at once: one batched forward, or sharded over the ranks of a
  ``torch.distributed`` group
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import hip
from .model import S2M2
from .spec import MODEL_CONFIGS


def load_model(pretrain_path: str, model_type: str, use_positivity: bool = True, refine_iter: int = 3, device=None) -> Optional[S2M2]:
    """Same contract as the reference: returns the model in eval mode on ``device``, or None (after printing) when loading fails."""
    if model_type not in MODEL_CONFIGS:
        print("model type should be one of [S, M, L, XL]")
        raise SystemExit(1)
    c, ntr = MODEL_CONFIGS[model_type]
    ckpt_path = os.path.join(pretrain_path, f"CH{c}NTR{ntr}.pth")
    model = S2M2(feature_channels=c, dim_expansion=1, num_transformer=ntr, use_positivity=use_positivity, refine_iter=refine_iter)
    try:
        checkpoint = torch.load(ckpt_path, weights_only=True)
        model.my_load_state_dict(checkpoint["state_dict"])
        model.eval()
        if device:
            model = model.to(device)
        print("Model loaded")
        return model
    except Exception as e:  # noqa: BLE001  (the reference swallows and reports)
        print(f"Error loading model: {e}")
        return None


def image_pad(img: torch.Tensor, factor: int = 32) -> torch.Tensor:
    """(B,C,H,W) -> (B,C,ceil(H/f)*f,ceil(W/f)*f) fp32 with the reference's blurred border; device tensors only."""
    with torch.cuda.device(img.device):                  # the C ABI launches on the current stream of the current device
        return hip.image_pad(img, factor)


def image_crop(img: torch.Tensor, img_shape: Tuple[int, int]) -> torch.Tensor:
    H, W = img.shape[-2:]
    Hn, Wn = img_shape
    ch = H - Hn
    if ch > 0:
        img = img[:, :, ch // 2: -(ch - ch // 2)]
    cw = W - Wn
    if cw > 0:
        img = img[:, :, :, cw // 2: -(cw - cw // 2)]
    return img


@torch.no_grad()
def run_stereo_matching(model: S2M2, left_torch: torch.Tensor, right_torch: torch.Tensor, device, N_repeat: int = 1):
    """-> (pred_disp, pred_occ, pred_conf, avg_conf_score, run_time_ms); images (1,3,H,W) of any size (padded to x32 here)."""
    img_height, img_width = left_torch.shape[-2:]
    left_pad = image_pad(left_torch.to(device), 32)
    right_pad = image_pad(right_torch.to(device), 32)
    with torch.inference_mode():
        with torch.amp.autocast(enabled=True, device_type=torch.device(device).type, dtype=torch.float16):
            if N_repeat > 1:
                # run-time estimation: the first call with a new geometry runs eagerly and the second captures the hipGraph; keep both
                # out of the timed loop (a single-shot call, N_repeat = 1, is timed as it is)
                # (bounded: with S2M2_GRAPH_CACHE=0, or a batch that forward() slices, no graph for this shape ever becomes resident)
                for _ in range(3):
                    if model.is_warm(left_pad.shape, torch.float16) or os.environ.get("S2M2_GRAPH", "1") == "0":
                        break
                    model(left_pad, right_pad)
            starter, ender = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            starter.record()
            for _ in range(N_repeat):
                pred_disp, pred_occ, pred_conf = model(left_pad, right_pad)
            ender.record()
            torch.cuda.synchronize()
    run_time = starter.elapsed_time(ender) / N_repeat
    pred_disp = image_crop(pred_disp, (img_height, img_width)).squeeze().float()
    pred_occ = image_crop(pred_occ, (img_height, img_width)).squeeze().float()
    pred_conf = image_crop(pred_conf, (img_height, img_width)).squeeze().float()
    margin = 100
    avg_conf_score = pred_conf[margin:-margin, margin:-margin].mean().item()
    return pred_disp, pred_occ, pred_conf, avg_conf_score, run_time


def compute_confidence_score(model: S2M2, left_torch: torch.Tensor, right_torch: torch.Tensor, device) -> float:
    return run_stereo_matching(model, left_torch, right_torch, device, N_repeat=1)[3]


@torch.no_grad()
def compute_confidence_scores(model: S2M2, lefts: torch.Tensor, rights: torch.Tensor, device, batch: Optional[int] = None,
                              dist=None, group=None, margin: int = 100) -> torch.Tensor:
    """Average confidence of N rectified pairs ``lefts`` / ``rights`` (N,3,H,W), same number per pair as ``compute_confidence_score``
    returns for it (pad to x32 -> autocast fp16 forward -> crop -> mean of the confidence map inside a ``margin`` border), evaluated
    ``batch`` pairs per forward (default: all at once; pairs are independent along the batch axis).

    With ``dist`` (an initialised ``torch.distributed`` module, one rank per GPU) the pairs are sharded round-robin over the ranks
    of ``group`` -- every rank passes the full population -- and every rank receives all N scores: the only collective is one
    all-gather of ceil(N/world) floats per rank (N need not divide: short shards are padded and the padding dropped).  Returns a float32 CPU tensor (N,)."""
    N = lefts.shape[0]
    if rights.shape != lefts.shape or lefts.dim() != 4:
        raise ValueError(f"expected two (N,3,H,W) batches, got {tuple(lefts.shape)} and {tuple(rights.shape)}")
    H, W = lefts.shape[-2:]
    if H <= 2 * margin or W <= 2 * margin:
        raise ValueError(f"images of {W}x{H} have no interior inside a {margin}-px margin")
    idx = list(range(N))
    world = rank = 1
    if dist is not None:
        from .shard import padded_shard, shard_indices
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        idx, _ = padded_shard(N, rank, world)                # N % world != 0: pad and drop, the all-gather stays fixed-size
    step = batch or max(1, len(idx))
    scores = []
    for s0 in range(0, len(idx), step):
        sel = idx[s0:s0 + step]
        lp = image_pad(lefts[sel].to(device), 32)
        rp = image_pad(rights[sel].to(device), 32)
        with torch.amp.autocast(enabled=True, device_type=torch.device(device).type, dtype=torch.float16):
            conf = model(lp, rp)[2]
        conf = image_crop(conf, (H, W)).float()
        scores.append(conf[:, 0, margin:-margin, margin:-margin].mean(dim=(1, 2)))
    mine = torch.cat(scores) if scores else torch.empty(0, device=device)
    if dist is None:
        return mine.cpu()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = torch.empty(N, dtype=torch.float32)
    for r in range(world):
        real = shard_indices(N, r, world)
        out[real] = parts[r][:len(real)].float().cpu()
    return out
