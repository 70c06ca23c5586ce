#!/usr/bin/env python3
"""Headline benchmark: stereo pairs/s of the S2M2 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json metric / configs[2] per GPU): S model (C=128, NTR=1), 1216x1024, fp16 compute (autocast, the
reference's deployment mode), refine_iter=3, use_positivity=True, ONE stereo pair per GPU per step (weak scaling: pairs shard
across ranks with no data-path collective; for N>1 each step ends with the RCCL gather of the three output maps to rank 0).
Inputs are synthetic uint8-valued images already resident in HBM; weights are the seeded random init (no checkpoints ship).

Prints ONE JSON line on rank 0 with `roofline` (K1 = LayerNorm+correlation kernel, HBM bound, measured with HIP events
around its launches inside the timed region) and `cpu_baseline` (the CPU oracle = port of the reference forward, timed on
this box's host cores on one pair of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="S")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1216)
    ap.add_argument("--pairs-per-gpu", type=int, default=1)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--refine-iter", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    return ap.parse_args()


def cpu_baseline(model_type, H, W, refine_iter):
    """Oracle (CPU restatement of the reference forward, fp32) on this box's host cores: one 1216x1024 pair (~10-30 s)."""
    from oracle import s2m2_oracle as O
    from s2m2_amd.spec import MODEL_CONFIGS
    from s2m2_amd.weights import noise_pair, seeded_state_dict
    C, ntr = MODEL_CONFIGS[model_type]
    sd = seeded_state_dict(C, 1, ntr, 0)
    l, r = noise_pair(64, 96, 1, 1)
    O.forward(sd, l, r, True, 1)                                   # warm up oneDNN primitives
    l, r = noise_pair(H, W, 1, 0)
    t0 = time.perf_counter()
    O.forward(sd, l, r, True, refine_iter)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "pairs/s", "seconds_per_pair": dt, "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"1 pair {W}x{H} {model_type}-model fp32 refine_iter={refine_iter}, oracle/s2m2_oracle.py (torch CPU ops), 1 timed run after a 96x64 warm-up"}


def pmc_traffic(model_type, H, W, use_fp16, B):
    """HBM bytes per K1 launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs of
    tools/k1_only.py, corrected as MI355X_MICROARCH.md prescribes -> profiles/rNN/k1_<case>_<dtype>_pmc.json).  PMC collection
    serialises kernels and cannot run inside the timed bench, so the newest committed summary for this exact K1 shape is quoted."""
    import glob
    case = {("S", 1024, 1216): "c3", ("S", 480, 640): "c2", ("L", 1024, 1216): "c4", ("XL", 2048, 2432): "c5"}.get((model_type, H, W))
    if case is None or B != 1:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"k1_{case}_{'fp16' if use_fp16 else 'fp32'}_pmc.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d.get("traffic_bytes"), os.path.relpath(files[-1], ROOT)


def main():
    a = parse()
    if a.no_graph:
        os.environ["S2M2_GRAPH"] = "0"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_gather = os.environ.get("S2M2_BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ   # single-GPU check of the RCCL gather path
    if world > 1 or force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when the communicator is created: stdout must carry exactly one JSON line, so the
        # communicator is created (eager init + one barrier) with fd 1 pointing at stderr
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from s2m2_amd.model import build_model
    from s2m2_amd.shard import gather_outputs_async
    from s2m2_amd.weights import noise_pair

    model = build_model(a.model, use_positivity=True, refine_iter=a.refine_iter).to(dev).eval()
    B = a.pairs_per_gpu
    left, right = noise_pair(a.height, a.width, B, seed=rank)
    left, right = left.to(dev), right.to(dev)
    use_fp16 = a.dtype == "fp16"
    eng = model.engine(torch.float16 if use_fp16 else torch.float32)

    pending = [None]                                  # the output gather of the previous step (N > 1), still in flight
    # S2M2_BENCH_FORCE_GATHER=1 (under torchrun with one rank): exercise the RCCL gather path on a single GPU
    gather = world > 1 or force_gather

    def step():
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
            out = model(left, right)
        if gather:                                    # gather of step i overlaps the forward of step i+1 (RCCL runs on its own stream)
            if pending[0] is not None:
                pending[0].wait(stack=False)
            pending[0] = gather_outputs_async(out, dist, dst=0)
        return out

    def drain():
        if pending[0] is not None:
            pending[0].wait(stack=False)
            pending[0] = None

    # HIP events around every K1 launch (the forward is replayed as two hipGraphs cut around K1, see engine.GraphRunner); enabled
    # before the warm-up so that graph capture happens there: call 1 runs eagerly, call 2 captures, later calls replay
    eng.k1_events = []
    for _ in range(max(a.warmup, 2)):
        step()
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    eng.k1_events.clear()                             # keep only the launches of the timed region
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()                                           # the last gather belongs to the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k1_ms = [s.elapsed_time(e) for s, e in eng.k1_events]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        h, w, C = a.height // 4, a.width // 4, model.feature_channels
        e = 2 if use_fp16 else 4
        k1_bytes = B * (2 * h * w * C * e + h * w * w * e)         # read both feature maps once + write cv once
        k1_us = 1e3 * sum(k1_ms) / max(1, len(k1_ms))
        achieved = k1_bytes / (k1_us * 1e-6) / 1e9 if k1_us > 0 else 0.0
        pairs = a.steps * B * world
        traffic, traffic_src = pmc_traffic(a.model, a.height, a.width, use_fp16, B)
        line = {
            "metric": "stereo pairs/sec, S-model 1216x1024 fp16 refine_iter=3 (ms/pair = 1000*n_gpus/value)",
            "value": pairs / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "ms_per_pair": 1e3 * elapsed / (a.steps * B),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if use_fp16 else "f32", "data": "synthetic",
            "config": {"workload": f"{a.model}-model {a.width}x{a.height} refine_iter={a.refine_iter} use_positivity=True, "
                                   f"{B} pair(s) per GPU per step, random-init weights (seeded LeCun normal)",
                       "pairs_per_gpu": B, "parallelism": f"dp{world} (pairs sharded, RCCL gather of outputs to rank 0)"},
            "roofline": {"kernel": "ln_corr_kernel (K1: LayerNorm + all-pairs correlation -> cost volume)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": k1_bytes, "avg_launch_us": k1_us,
                         "launches_timed": len(k1_ms)},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.model, a.height, a.width, a.refine_iter)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
