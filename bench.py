#!/usr/bin/env python3
"""Headline benchmark: stereo pairs/s of the S2M2 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank per GPU;
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or started plainly, in which case bench.py launches its N
ranks itself through torch.distributed.run on 127.0.0.1 and rank 0 prints the one JSON line.  ``--gpus N`` with fewer than N
visible devices fails loudly.

Workload (BASELINE.json metric / configs[2] per GPU): S model (C=128, NTR=1), 1216x1024, fp16 compute (autocast, the
reference's deployment mode), refine_iter=3, use_positivity=True, ONE stereo pair per GPU per step (weak scaling: pairs shard
across ranks with no data-path collective; for N>1 each step ends with the RCCL gather of the three output maps to rank 0).
Inputs are synthetic uint8-valued images already resident in HBM; weights are the seeded random init (no checkpoints ship).

The JSON line carries, next to the contract's fields:
* ``roofline``            K1 = LayerNorm + correlation kernel, HBM bound: a start / stop HIP event pair attached to each of its
                          dispatches inside the timed region (hipExtLaunchKernel: the kernel's own execution time, the quantity
                          a rocprofv3 kernel trace reports);
* ``roofline_attention``  K4 attention kernels, MFMA bound: one instrumented eager forward AFTER the timed region with HIP events
                          around every K4 launch, analytic FLOPs 4*N^2*d*heads*batch -> fraction of the dense
                          fp16 MFMA peak (north_star: "MFMA utilisation on the attention GEMMs");
* ``forward``             multiply-accumulate work of one forward as executed (counted per launch) / time per pair;
* ``cpu_baseline``        the reference's own CPU forward (oracle/_ref = the unmodified reference model byte-compiled by
                          oracle/make_ref.py; kind "reference") on this box's host cores: thread-count sweep, 1 warm-up + 3 timed
                          runs at the best count.

``--dry`` (CPU, gloo, no model): exercises launcher, rendezvous, gather and max-over-ranks timing without a GPU (tests).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3


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
    ap.add_argument("--no-secondary", action="store_true", help="skip the 640x480 block measured after the timed region")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--dry", action="store_true", help="CPU/gloo plumbing check of the multi-rank path: no GPU, no model")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` -> N ranks under torch.distributed.run
# ---------------------------------------------------------------------------------------------------------------------------
def launch_ranks(a) -> int:
    if not a.dry:
        import torch
        n = torch.cuda.device_count()
        if n < a.gpus:
            sys.stderr.write(f"bench.py: --gpus {a.gpus} requested but only {n} GPU(s) visible; refusing to run a smaller job\n")
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's own CPU forward (oracle/_ref), thread sweep
# ---------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(model_type, H, W, refine_iter):
    """north_star: "next to the reference's own CPU forward timed on the host cores of the same box in the same run".  The UNMODIFIED
    reference module (oracle/_ref: its model files byte-compiled by oracle/make_ref.py in the build container, shipped with the
    snapshot; s2m2.py:136-197 under torch.no_grad(), eval, fp32 -- SURVEY.md 8d) on this box's host cores, one pair of the benchmark
    workload per run, the same seeded weights and images as the GPU leg: 1 warm-up, one timed run per thread count in the sweep, 2 more
    at the best count (value = median of its 3).  Default intra-op thread counts (128 on the GPU box) oversubscribe the oneDNN / ATen
    kernels: measured 35.7 s per pair at 128 threads in round 1.  kind "reference"; only if oracle/_ref is missing (a checkout that was
    never built where /root/reference exists) the repo's own restatement (oracle/s2m2_oracle.py) is timed instead and labelled "port"."""
    import torch
    from oracle import ref_loader
    from s2m2_amd.spec import MODEL_CONFIGS
    from s2m2_amd.weights import noise_pair, seeded_state_dict
    C, ntr = MODEL_CONFIGS[model_type]
    sd = seeded_state_dict(C, 1, ntr, 0)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    sweep = sorted({t for t in (8, 16, 32) if t <= ncpu} or {ncpu})
    l, r = noise_pair(H, W, 1, 0)
    why = ref_loader.why_not()
    if why is None:
        kind = "reference"
        ref = ref_loader.reference_model(sd, C, ntr, True, refine_iter)
        what = "the unmodified reference S2M2.forward (oracle/_ref, byte-compiled from /root/reference/src/s2m2/core/model by oracle/make_ref.py)"

        def fwd():
            with torch.no_grad():
                return ref(l, r)
    else:
        if os.environ.get("S2M2_REQUIRE_REF") == "1":
            raise RuntimeError(f"S2M2_REQUIRE_REF=1 but oracle/_ref is not usable: {why}")
        sys.stderr.write(f"bench.py: WARNING: oracle/_ref is not usable ({why}): the CPU baseline is the repo's restatement (kind 'port'), NOT the reference\n")
        kind = "port"
        from oracle import s2m2_oracle as O
        what = f"oracle/s2m2_oracle.py (torch CPU restatement; the reference itself is not on this box: {why})"

        def fwd():
            return O.forward(sd, l, r, True, refine_iter)

    def one():
        t0 = time.perf_counter()
        fwd()
        return time.perf_counter() - t0

    torch.set_num_threads(sweep[len(sweep) // 2])
    one()                                                          # warm-up at full size (oneDNN primitive creation, page faults)
    times = {}
    for t in sweep:
        torch.set_num_threads(t)
        times[t] = [one()]
    best = min(times, key=lambda t: times[t][0])
    torch.set_num_threads(best)
    times[best] += [one(), one()]
    dt = sorted(times[best])[1]
    torch.set_num_threads(default_threads)
    return {"value": 1.0 / dt, "unit": "pairs/s", "seconds_per_pair": dt, "cores": best, "host_cpus": ncpu, "kind": kind,
            "thread_sweep_seconds": {str(t): [round(x, 3) for x in v] for t, v in times.items()},
            "sample": f"1 pair {W}x{H} {model_type}-model fp32 refine_iter={refine_iter} use_positivity=True per run, {what}: "
                      f"1 warm-up, 1 run per thread count, median of 3 at the best count"}


def gpu_state(local_rank=0):
    """Best-effort snapshot of the board's power cap / average power / clocks (an extra of the line: a power-capped or down-clocked box of
    the pool then shows in the record next to its ms/pair).  amdgpu hwmon files (no privileges needed); rocm-smi --json as a second source."""
    import glob
    import json as _json
    import subprocess
    out = {}
    try:
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        if cards:
            hw = cards[min(local_rank, len(cards) - 1)]

            def rd(name):
                try:
                    return int(open(os.path.join(hw, name)).read().strip())
                except (OSError, ValueError):
                    return None
            for key, name, scale in (("power_cap_w", "power1_cap", 1e-6), ("power_avg_w", "power1_average", 1e-6), ("power_input_w", "power1_input", 1e-6),
                                     ("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6), ("temp_c", "temp1_input", 1e-3)):
                v = rd(name)
                if v is not None:
                    out[key] = round(v * scale, 1)
    except Exception as e:  # noqa: BLE001
        out["hwmon_error"] = str(e)[:100]
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
        if r.returncode == 0 and r.stdout.strip().startswith("{"):
            js = _json.loads(r.stdout)
            card = js.get(f"card{local_rank}") or next(iter(js.values()))
            out["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "performance"))}
    except Exception as e:  # noqa: BLE001
        out["rocm_smi_error"] = str(e)[:100]
    return out


def k1_algorithmic(h, w, C, e, B=1):
    """read both feature maps once + write the volume once (bytes), 2*h*w*w*C flops"""
    return B * (2 * h * w * C * e + h * w * w * e), 2.0 * B * h * w * w * C


def attention_and_work(eng, left, right, use_fp16, B, ms_per_pair):
    """One instrumented EAGER forward (outside every timed region): the work meter of the binding (2 flops per multiply-accumulate of every
    GEMM-shaped launch) and HIP events around every K4 launch -> (roofline_attention, forward) blocks of the JSON line."""
    import torch
    from s2m2_amd import hip
    saved = eng.k1_events
    eng.k1_events = None
    hip.METER, hip.ATTN_EVENTS, hip.ROW_EVENTS = {}, [], []
    with torch.autocast("cuda", enabled=False):
        eng.run(left, right, None)
    torch.cuda.synchronize()
    meter, attn_ev, row_ev = hip.METER, hip.ATTN_EVENTS, hip.ROW_EVENTS
    hip.METER = hip.ATTN_EVENTS = hip.ROW_EVENTS = None
    eng.k1_events = saved
    peak_tf = MFMA_F16_PEAK_TFLOPS if use_fp16 else MFMA_F32_PEAK_TFLOPS
    a_us = sum(1e3 * s.elapsed_time(e_) for s, e_, *_ in attn_ev)
    a_fl = sum(f for _, _, f, *_ in attn_ev)
    by_shape = {}
    for s, e_, f, tag, nbytes in attn_ev:
        d = by_shape.setdefault(tag, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += 1e3 * s.elapsed_time(e_)
        d[2] += f
        d[3] += nbytes

    def roof(us, flops, nbytes):
        """a launch (or a sum of launches) against min(HBM, MFMA): the time the binding roofline allows over the time measured"""
        t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9) * 1e6, flops / (peak_tf * 1e12) * 1e6
        return {"us": round(us, 1), "tflops": round(flops / (us * 1e-6) / 1e12, 1) if us > 0 else 0.0, "roofline_us": round(max(t_hbm, t_mfma), 2),
                "bound": "hbm" if t_hbm > t_mfma else "mfma", "frac_of_roofline": round(max(t_hbm, t_mfma) / us, 4) if us > 0 else 0.0}

    rows_by = {}
    for s, e_, f, nbytes, tag in row_ev:
        d = rows_by.setdefault(tag, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += 1e3 * s.elapsed_time(e_)
        d[2] += f
        d[3] += nbytes
    fwd_flops = sum(v[0] for v in meter.values()) / B
    attn = {"kernel": "attention_kernel (K4: every QK^T / PV contraction of a forward)", "bound": "mfma",
            "achieved": a_fl / (a_us * 1e-6) / 1e12 if a_us > 0 else 0.0, "peak": peak_tf, "unit": "TFLOP/s",
            "frac": (a_fl / (a_us * 1e-6) / 1e12 / peak_tf) if a_us > 0 else 0.0,
            "flops_per_forward": a_fl, "us_per_forward": a_us, "launches": len(attn_ev),
            "how": "HIP events around each K4 launch in one eager forward after the timed region (adds ~2 us of dispatch per launch)",
            "by_shape(batch,heads,N,d)": {k: dict(launches=v[0], **roof(v[1], v[2], v[3])) for k, v in by_shape.items()},
            "row_attn(K13: Q|K|V, attention, proj, FFN of a 1-D attention step in one launch; flops = all six layers + QK^T + PV)":
                {k: dict(launches=v[0], **roof(v[1], v[2], v[3])) for k, v in rows_by.items()}}
    fwd = {"flops_per_pair_executed": fwd_flops, "achieved_tflops": fwd_flops / (ms_per_pair * 1e-3) / 1e12,
           "frac_of_mfma_peak": fwd_flops / (ms_per_pair * 1e-3) / 1e12 / peak_tf,
           "flops_by_family": {k: v[0] / B for k, v in meter.items()}, "launches_by_family": {k: v[1] for k, v in meter.items()},
           "note": "2 flops per multiply-accumulate of every GEMM-shaped launch, padded channel counts; per GPU"}
    return attn, fwd


def secondary_config(tag, model_type, H, W, positivity, dev, use_fp16, refine_iter, steps, warmup=2, detail=True):
    """Another BASELINE.json configuration (c4: L 1216x1024 -- "stresses MFMA attn-aggregation path"; c5: XL 2432x2048 allow_negative; M) on
    rank 0's GPU after the headline's timed region, one pair per step through the same drop-in module: hipGraph replay cut around K1, whose
    dispatches carry start / stop HIP events; then one instrumented eager forward for the attention / whole-forward MFMA fractions.  Not
    part of `value`.  The model is freed afterwards."""
    import torch
    from s2m2_amd.model import build_model
    from s2m2_amd.weights import noise_pair
    t_build = time.perf_counter()
    model = build_model(model_type, use_positivity=positivity, refine_iter=refine_iter).to(dev).eval()
    left, right = (t.to(dev) for t in noise_pair(H, W, 1, seed=3))
    eng = model.engine(torch.float16 if use_fp16 else torch.float32)
    eng.k1_events = []

    def step():
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
            return model(left, right)

    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    eng.k1_events.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k1_us = [t.elapsed_us() for t in eng.k1_events]
    finite = bool(all(torch.isfinite(o).all() for o in out))
    h, w, C = H // 4, W // 4, model.feature_channels
    k1_bytes, k1_flops = k1_algorithmic(h, w, C, 2 if use_fp16 else 4)
    us = sum(k1_us) / max(1, len(k1_us))
    gbs = k1_bytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
    peak_tf = MFMA_F16_PEAK_TFLOPS if use_fp16 else MFMA_F32_PEAK_TFLOPS
    ms_pair = 1e3 * dt / steps
    res = {"config": tag, "workload": f"{model_type}-model {W}x{H} refine_iter={refine_iter} use_positivity={positivity}, 1 pair per step, n_gpus=1 (rank 0), "
                                     f"random-init weights (seeded LeCun normal)",
           "value": steps / dt, "unit": "pairs/s", "ms_per_pair": ms_pair, "steps": steps, "warmup": max(warmup, 2), "dtype": "f16" if use_fp16 else "f32",
           "outputs_finite": finite, "setup_seconds": round(t_build, 1),
           "roofline": {"kernel": "ln_corr_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "variant": "tokens normalised by the producing K9 launch (s2m2_corr)" if eng._tokens_normed is not None else "LayerNorm inside K1 (s2m2_ln_corr)",
                        "algorithmic_bytes_per_launch": k1_bytes, "avg_launch_us": us, "launches_timed": len(k1_us),
                        "mfma_tflops": k1_flops / (us * 1e-6) / 1e12 if us > 0 else 0.0,
                        "mfma_frac": (k1_flops / (us * 1e-6) / 1e12 / peak_tf) if us > 0 else 0.0}}
    if detail:
        attn, fwd = attention_and_work(eng, left, right, use_fp16, 1, ms_pair)
        res["roofline_attention"] = {k: attn[k] for k in attn if k not in ("kernel", "bound", "how")}
        res["forward"] = {k: fwd[k] for k in ("flops_per_pair_executed", "achieved_tflops", "frac_of_mfma_peak")}
    res["peak_mem_gib"] = round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)
    eng.k1_events = None
    del model, eng, left, right, out
    torch.cuda.empty_cache()
    return res


def k1_ln_inside_in_forward(a, dev, use_fp16, steps=20):
    """The OTHER form of the judged kernel inside a forward: a second module whose K1 normalises the tokens itself (S2M2_FUSE_K1LN=0: the
    LayerNorm + correlation kernel SURVEY.md 8d names, s2m2_ln_corr, sc1 volume stores), warm graph, `steps` forwards with the same
    dispatch-attached events as the headline's K1."""
    import torch
    from s2m2_amd.model import build_model
    from s2m2_amd.weights import noise_pair
    old = os.environ.get("S2M2_FUSE_K1LN")
    os.environ["S2M2_FUSE_K1LN"] = "0"
    try:
        model = build_model(a.model, use_positivity=True, refine_iter=a.refine_iter).to(dev).eval()
        eng = model.engine(torch.float16 if use_fp16 else torch.float32)
    finally:
        if old is None:
            os.environ.pop("S2M2_FUSE_K1LN", None)
        else:
            os.environ["S2M2_FUSE_K1LN"] = old
    left, right = (t.to(dev) for t in noise_pair(a.height, a.width, 1, seed=0))
    eng.k1_events = []

    def step():
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
            return model(left, right)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    eng.k1_events.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    us = [t.elapsed_us() for t in eng.k1_events]
    eng.k1_events = None
    h, w, C = a.height // 4, a.width // 4, model.feature_channels
    k1_bytes, _ = k1_algorithmic(h, w, C, 2 if use_fp16 else 4)
    avg = sum(us) / max(1, len(us))
    del model, eng
    torch.cuda.empty_cache()
    return {"kernel": "ln_corr_kernel with the LayerNorm inside (s2m2_ln_corr, S2M2_FUSE_K1LN=0), in-forward, dispatch-attached events",
            "avg_launch_us": avg, "launches_timed": len(us), "frac": (k1_bytes / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS) if avg > 0 else 0.0,
            "ms_per_pair_of_that_forward": 1e3 * dt / steps}


def secondary_640x480(model, eng, dev, use_fp16, refine_iter, model_type, steps=50, warmup=3):
    """north_star's second size (BASELINE configs[1]: 640x480, refine_iter 3, one pair) measured in the SAME run on rank 0's GPU after
    the headline's timed region: hipGraph replay, K1 with its own start / stop events on the dispatch.  Not part of `value`."""
    import torch
    from s2m2_amd.weights import noise_pair
    H, W = 480, 640
    left, right = (t.to(dev) for t in noise_pair(H, W, 1, seed=7))
    eng.k1_events = []

    def step():
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
            return model(left, right)

    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    eng.k1_events.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k1_us = [t.elapsed_us() for t in eng.k1_events]
    eng.k1_events = None
    h, w, C = H // 4, W // 4, model.feature_channels
    e = 2 if use_fp16 else 4
    k1_bytes = 2 * h * w * C * e + h * w * w * e
    us = sum(k1_us) / max(1, len(k1_us))
    gbs = k1_bytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
    return {"workload": f"{model_type}-model {W}x{H} refine_iter={refine_iter} use_positivity=True, 1 pair per step, n_gpus=1 (rank 0)",
            "value": steps / dt, "unit": "pairs/s", "ms_per_pair": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
            "dtype": "f16" if use_fp16 else "f32",
            "roofline": {"kernel": "ln_corr_kernel", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": k1_bytes, "avg_launch_us": us, "launches_timed": len(k1_us),
                         "note": "h = 120 image rows = 120 workgroups on 256 CUs: this size cannot fill the chip with one pair"}}


def secondary_batched(model, dev, use_fp16, a, pairs=2, steps=10, warmup=3):
    """Information only (not `value`, whose workload is ONE pair per GPU per step as BASELINE's configs[2] shards them): the same model and
    size with `pairs` pairs per launch sequence on rank 0's GPU -- what batching buys when a caller has more than one pair per GPU
    (r06: the module runs a batch as two chunks of ceil(B / 2) pairs on two side streams, model.py: _forward_pairs -- the CUs one chunk leaves
    idle at the coarse pyramid levels run the other chunk's launches; S2M2_PAIR_STREAMS=0 is the single batched launch sequence of r05)."""
    import torch
    from s2m2_amd.weights import noise_pair
    left, right = (t.to(dev) for t in noise_pair(a.height, a.width, pairs, seed=11))

    def step():
        with torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
            return model(left, right)

    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": f"{a.model}-model {a.width}x{a.height} refine_iter={a.refine_iter}, {pairs} pairs per step, n_gpus=1 (rank 0)",
            "value": steps * pairs / dt, "unit": "pairs/s", "ms_per_pair": 1e3 * dt / (steps * pairs), "steps": steps, "warmup": warmup,
            "pair_streams": int(os.environ.get("S2M2_PAIR_STREAMS", "2"))}


def secondary_two_streams(model, dev, use_fp16, a, steps=20, warmup=3):
    """Information only (not `value`): the SAME workload -- one pair per forward -- with two forwards in flight, consecutive pairs enqueued
    alternately on two HIP streams (the module's calling contract: any stream, one captured graph and scratch set per stream).  The small
    launches of one forward fill the CUs the other leaves idle at the coarse pyramid levels: what a serving loop with more than one
    request queued gets without batching; the latency of a pair roughly doubles, the throughput is what is reported."""
    import torch
    from s2m2_amd.weights import noise_pair
    pairs = [tuple(t.to(dev) for t in noise_pair(a.height, a.width, 1, seed=21 + k)) for k in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    cur = torch.cuda.current_stream(dev)
    for s in streams:
        s.wait_stream(cur)

    def step(i):
        with torch.cuda.stream(streams[i & 1]), torch.autocast("cuda", dtype=torch.float16, enabled=use_fp16):
            return model(*pairs[i & 1])

    for i in range(2 * max(warmup, 3)):                   # every stream: eager call, capture, replays
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    finite = bool(all(torch.isfinite(o).all() for o in out))
    return {"workload": f"{a.model}-model {a.width}x{a.height} refine_iter={a.refine_iter}, 1 pair per forward, TWO forwards in flight (two HIP streams), n_gpus=1 (rank 0)",
            "value": steps / dt, "unit": "pairs/s", "ms_per_pair_throughput": 1e3 * dt / steps, "steps": steps, "outputs_finite": finite}


def k1_source_hash():
    """sha256 over the sources K1 is compiled from -- what a PMC summary is valid for (tools/pmc_summary.py stores it)"""
    import hashlib
    hsh = hashlib.sha256()
    for f in ("ln_corr.hip", "common.h"):
        hsh.update(open(os.path.join(ROOT, "s2m2_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()[:16]


def pmc_traffic(model_type, H, W, use_fp16, B):
    """HBM bytes per K1 launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs of
    tools/k1_only.py, corrected as MI355X_MICROARCH.md prescribes -> profiles/rNN/k1_<case>_<dtype>_pmc.json).  PMC collection
    serialises kernels and cannot run inside the timed bench, so the newest committed summary for this exact K1 shape is quoted --
    and ONLY if it was measured on the kernel source of this checkout (`k1_source_sha256_16` in the summary): a summary of an older
    ln_corr.hip is refused (traffic null, the reason in traffic_source)."""
    import glob
    case = {("S", 1024, 1216): "c3", ("S", 480, 640): "c2", ("L", 1024, 1216): "c4", ("XL", 2048, 2432): "c5"}.get((model_type, H, W))
    if case is None or B != 1:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"k1_{case}_{'fp16' if use_fp16 else 'fp32'}_pmc.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if d.get("k1_source_sha256_16") != k1_source_hash():
        return None, f"{rel}: REFUSED (measured on another ln_corr.hip / common.h: {d.get('k1_source_sha256_16')} vs {k1_source_hash()} here)"
    return d.get("traffic_bytes"), rel


def _quiet_init(dist, backend, rank, world, dev):
    """RCCL prints a version banner on STDOUT when the communicator is created: stdout must carry exactly one JSON line, so the
    communicator is created (eager init + one barrier) with fd 1 pointing at stderr."""
    import torch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()
        if backend == "nccl":
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)


def dry_run(a, rank, world):
    """The N>1 control flow on CPU: rendezvous (gloo), per-step async gather to rank 0 overlapped with the next step, barrier-bracketed
    timing, max over ranks, one JSON line from rank 0.  The "forward" is a stand-in that only depends on the rank's own pair."""
    import torch
    import torch.distributed as dist
    from s2m2_amd.shard import gather_outputs_async
    if world > 1:
        _quiet_init(dist, "gloo", rank, world, None)
    g = torch.Generator().manual_seed(rank)
    left = torch.rand(a.pairs_per_gpu, 3, 32, 64, generator=g)
    right = torch.rand(a.pairs_per_gpu, 3, 32, 64, generator=g)
    pending = [None]
    got = []

    def step():
        d = (left - right).mean(dim=1, keepdim=True)
        out = (d, d * 0.5, d.abs())
        if world > 1:
            if pending[0] is not None:
                got.append(pending[0].wait())
            pending[0] = gather_outputs_async(out, dist, dst=0)

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    if pending[0] is not None:
        got.append(pending[0].wait())
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ok = world == 1 or all(x is not None and tuple(x[0].shape) == (world * a.pairs_per_gpu, 1, 32, 64) for x in got)
        pairs = a.steps * a.pairs_per_gpu * world
        print(json.dumps({"metric": "stereo pairs/sec (DRY RUN: launcher / gather plumbing on CPU, no model)", "value": pairs / elapsed,
                          "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "dry": True, "gathers_ok": bool(ok), "config": {"workload": "dry run", "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    if "RANK" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {a.gpus}, or plainly)")
    if a.dry:
        return dry_run(a, rank, world)
    if a.no_graph:
        os.environ["S2M2_GRAPH"] = "0"
    import torch
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_gather = os.environ.get("S2M2_BENCH_FORCE_GATHER") == "1" and "RANK" in os.environ   # single-GPU check of the RCCL gather path
    if world > 1 or force_gather:
        import torch.distributed as dist
        _quiet_init(dist, "nccl", rank, world, dev)

    from s2m2_amd import hip
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
    state_before = gpu_state(local_rank) if rank == 0 else None
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()                                           # the last gather belongs to the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    state_after = gpu_state(local_rank) if rank == 0 else None
    k1_ms = [t.elapsed_us() * 1e-3 for t in eng.k1_events]
    own_elapsed = elapsed
    per_rank_ms = [1e3 * elapsed / a.steps]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                                  # per-rank step time: a straggler is visible in the line
        per_rank_ms = [1e3 * float(x.item()) / a.steps for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    devices = None
    if dist is not None:
        # which physical device every rank ran on: a scaling record then proves N distinct GPUs took part
        me = str(getattr(torch.cuda.get_device_properties(dev), "uuid", f"{socket.gethostname()}:{local_rank}"))
        devices = [None] * world
        dist.all_gather_object(devices, me)

    if rank == 0:
        h, w, C = a.height // 4, a.width // 4, model.feature_channels
        e = 2 if use_fp16 else 4
        k1_bytes = B * (2 * h * w * C * e + h * w * w * e)         # read both feature maps once + write cv once
        k1_variant = "full volume (B,h,w,w)"
        if eng._tokens_normed is not None:                         # DispInit's LayerNorm ran in the K9 launch that wrote the tokens (engine.features)
            k1_variant += ", tokens normalised by the producing K9 launch (s2m2_corr), volume rows on 128-byte lines"
        else:
            k1_variant += ", LayerNorm inside K1 (s2m2_ln_corr)"
        if eng.cv_band >= 0:                                       # opt-in banded store (S2M2_CV_BAND=1): only j <= i + band must be written
            k1_bytes = B * (2 * h * w * C * e + h * e * sum(min(w, i + 1 + eng.cv_band) for i in range(w)))
            k1_variant = f"banded volume j <= i + {eng.cv_band}"
        k1_us = 1e3 * sum(k1_ms) / max(1, len(k1_ms))
        achieved = k1_bytes / (k1_us * 1e-6) / 1e9 if k1_us > 0 else 0.0
        pairs = a.steps * B * world
        ms_per_pair_gpu = 1e3 * elapsed / (a.steps * B)             # one GPU's time per pair
        traffic, traffic_src = pmc_traffic(a.model, a.height, a.width, use_fp16, B) if eng.cv_band < 0 else (None, None)
        # the OTHER variant of the judged kernel in the same process, same dispatch-attached events: K1 with the LayerNorm inside (s2m2_ln_corr)
        # on the very tokens of the last forward (back to back: the tokens sit in L2 / Infinity Cache, the in-forward figure above does not have
        # that luxury), and the shipped variant measured the same way for comparison.  Same-box end-to-end A/B of the two:
        # profiles/r04/ab_k1_fold.txt (5 alternating runs: 8.819 vs 8.815 ms per pair -- the choice does not move the forward)
        k1_both = None
        try:
            runner = next(iter(model._graphs.values()), None)
            tok = runner.state[0] if runner is not None and getattr(runner, "split", False) else None
            if tok is not None and eng.cv_band < 0:
                def k1_loop(fn):
                    ts = []
                    for _ in range(25):
                        t = hip.KernelTimer()
                        fn(t)
                        ts.append(t)
                    torch.cuda.synchronize()
                    us = sorted(t.elapsed_us() for t in ts[5:])
                    return us[len(us) // 2]
                own = k1_loop(lambda t: hip.ln_corr(tok, eng.ln_w, eng.ln_b, out=runner.cv, timer=t))
                k1_both = {"ln_inside_k1_us_back_to_back": own, "ln_inside_k1_frac": k1_bytes / (own * 1e-6) / 1e9 / HBM_PEAK_GBS}
                if runner.normed is not None:
                    shp = k1_loop(lambda t: hip.corr(runner.normed, out=runner.cv, timer=t))
                    k1_both.update({"shipped_us_back_to_back": shp, "shipped_frac_back_to_back": k1_bytes / (shp * 1e-6) / 1e9 / HBM_PEAK_GBS})
        except Exception as e:  # noqa: BLE001  (an extra, never the line)
            k1_both = {"error": str(e)[:200]}
        # ---- after the timed region: one instrumented eager forward (work meter + HIP events around every K4 launch)
        attn_block, fwd_block = attention_and_work(eng, left, right, use_fp16, B, ms_per_pair_gpu)
        eng.k1_events = None
        line = {
            "metric": "stereo pairs/sec, S-model 1216x1024 fp16 refine_iter=3 (ms/pair = 1000*n_gpus/value)",
            "value": pairs / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "ms_per_pair": ms_per_pair_gpu,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if use_fp16 else "f32", "data": "synthetic",
            "config": {"workload": f"{a.model}-model {a.width}x{a.height} refine_iter={a.refine_iter} use_positivity=True, "
                                   f"{B} pair(s) per GPU per step, random-init weights (seeded LeCun normal)",
                       "pairs_per_gpu": B, "parallelism": f"dp{world} (pairs sharded, RCCL gather of outputs to rank 0)"},
            "roofline": {"kernel": "ln_corr_kernel (K1: all-pairs correlation of the LayerNorm'ed tokens -> cost volume)", "bound": "hbm", "variant": k1_variant,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": k1_bytes, "avg_launch_us": k1_us,
                         "launches_timed": len(k1_ms), "store_policy": "sc1 write-through (S2M2_K1_NT=%s)" % os.environ.get("S2M2_K1_NT", "2"),
                         "both_variants": k1_both},
            "roofline_attention": attn_block,
            "forward": fwd_block,
        }
        line["per_rank_ms_per_step"] = [round(x, 4) for x in per_rank_ms]
        line["gpu_state"] = {"before_timed_region": state_before, "after_timed_region": state_after,
                             "note": "board power cap / average power / clocks read by rank 0 outside the timed region (hwmon, rocm-smi); boxes of "
                                     "the pool differ by 2 - 5 % end to end on the same build"}
        if not a.no_secondary and (a.height, a.width) != (480, 640):
            line["secondary"] = secondary_640x480(model, eng, dev, use_fp16, a.refine_iter, a.model)
            if B == 1:
                line["secondary_batched"] = secondary_batched(model, dev, use_fp16, a, pairs=2)
                if world == 1:
                    try:
                        line["secondary_two_streams"] = secondary_two_streams(model, dev, use_fp16, a)
                    except Exception as e:  # noqa: BLE001  (an extra, never the line)
                        line["secondary_two_streams"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        if devices is not None:
            line["rccl_ranks"], line["distinct_gpus"], line["gpu_uuids"] = world, len(set(devices)), devices
        if not a.no_secondary and world == 1 and (a.model, a.height, a.width, B) == ("S", 1024, 1216, 1) and use_fp16:
            # the judged kernel's other form inside a forward, and the other BASELINE configurations (driver-run record of c4 / c5 / M)
            try:
                line["roofline"]["ln_inside_k1_in_forward"] = k1_ln_inside_in_forward(a, dev, use_fp16)
            except Exception as e:  # noqa: BLE001  (an extra, never the line)
                line["roofline"]["ln_inside_k1_in_forward"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
            for key, (tag, mt, H2, W2, pos, st) in (("secondary_L", ("c4", "L", 1024, 1216, True, 10)), ("secondary_M", ("M", "M", 1024, 1216, True, 10)),
                                                    ("secondary_XL", ("c5", "XL", 2048, 2432, False, 3))):
                try:
                    line[key] = secondary_config(tag, mt, H2, W2, pos, dev, use_fp16, a.refine_iter, st)
                except Exception as e:  # noqa: BLE001
                    line[key] = {"config": tag, "error": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.model, a.height, a.width, a.refine_iter)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
