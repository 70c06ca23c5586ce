"""Drop-in ``S2M2`` module: same constructor, ``forward`` and ``state_dict`` keys as the reference
(/root/reference/src/s2m2/core/model/s2m2.py:13-197), different inside.

* Parameters are generated from the flat table of :mod:`s2m2_amd.spec` (no per-layer ``nn.Module`` classes);
  they live in a tree of plain containers only so that ``state_dict()`` / ``load_state_dict()`` produce the
  reference's dotted names and real ``CH{C}NTR{n}.pth`` checkpoints load unchanged.
* ``forward`` hands the images to :class:`s2m2_amd.engine.Engine`, which runs the hot path on gfx950 through
  ``libs2m2_hip.so``.  There is no CPU implementation: inputs must be CUDA tensors and the kernel library must be
  built, otherwise a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import os
import threading
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .spec import MODEL_CONFIGS, param_table


class _Params(nn.Module):
    """Anonymous container node of the parameter tree (children are named by the dotted-path components)."""

    def extra_repr(self) -> str:
        return f"{sum(p.numel() for p in self.parameters(recurse=False))} params"


class S2M2(nn.Module):
    def __init__(self, feature_channels: int, dim_expansion: int, num_transformer: int, use_positivity: bool = False,
                 output_upsample: bool = False, refine_iter: int = 3):
        super().__init__()
        self.feature_channels = feature_channels
        self.dim_expansion = dim_expansion
        self.num_transformer = num_transformer
        self.use_positivity = use_positivity
        self.refine_iter = refine_iter
        self.output_upsample = output_upsample
        self._table = param_table(feature_channels, dim_expansion, num_transformer)
        for name, shape in self._table.items():
            node: nn.Module = self
            *path, leaf = name.split(".")
            for comp in path:
                if comp not in node._modules:
                    node.add_module(comp, _Params())
                node = node._modules[comp]
            node.register_parameter(leaf, nn.Parameter(torch.empty(shape), requires_grad=False))
        self.reset_parameters()
        self._engines: Dict[Tuple, "object"] = {}
        self._graphs: "OrderedDict[Tuple, object]" = OrderedDict()      # LRU of captured hipGraphs, see forward()
        self._seen = set()
        self._side_streams: Dict[object, list] = {}                     # per (device, caller stream): the side streams of _forward_pairs
        self._epoch = 0                                                 # bumped by invalidate()
        self._lock = threading.RLock()                                  # host-side enqueue of one forward at a time per module

    # caches and the lock are process state, not model state: copy.deepcopy / pickling (torch.save(model)) drop them
    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("_engines", "_graphs", "_seen", "_lock", "_side_streams"):
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._engines, self._graphs, self._seen, self._lock = {}, OrderedDict(), set(), threading.RLock()
        self._side_streams = {}

    # -- weights -------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self, seed: int = 0) -> None:
        """Well-conditioned deterministic random weights (see s2m2_amd.weights); real use loads a checkpoint."""
        from .weights import draw_tensor
        for name, p in self.named_parameters():
            p.copy_(draw_tensor(name, tuple(p.shape), seed))

    def my_load_state_dict(self, state_dict) -> None:
        """Shape-mismatch tolerant load, same behaviour as the reference (s2m2.py:69-78)."""
        own = self.state_dict()
        for k in state_dict:
            if k in own and state_dict[k].shape != own[k].shape:
                print(f"Skip loading parameter: {k}, required shape: {own[k].shape}, loaded shape: {state_dict[k].shape}")
                state_dict[k] = own[k]
        self.load_state_dict(state_dict, strict=False)

    def invalidate(self) -> None:
        """Drop the packed weights and captured graphs.  ``load_state_dict`` / ``my_load_state_dict`` / ``.to()`` / ``.half()`` and any
        in-place update that bumps a parameter's version counter are detected automatically; call this after writes that bypass it
        (``p.data.copy_()``, ``p.data = ...``, raw pointers)."""
        self._epoch += 1

    def _weights_version(self) -> Tuple:
        """Cheap fingerprint of the parameter storage: autograd version counters (where they exist: inference tensors have none),
        data pointers, device, dtype and the explicit epoch."""
        def ver(p):
            try:
                return p._version
            except RuntimeError:
                return -1
        ps = list(self.parameters())
        return tuple((ver(p), p.data_ptr()) for p in ps) + (ps[0].device, ps[0].dtype, self._epoch)

    # -- inference -----------------------------------------------------------------------------------
    def engine(self, dtype: torch.dtype):
        from .engine import Engine
        key = (dtype, self._weights_version())
        eng = self._engines.get(key)
        if eng is None:
            self._engines.clear()                          # weights changed -> drop stale packed copies
            self._graphs.clear()
            self._seen.clear()
            eng = Engine(self, dtype)
            self._engines[key] = eng
        return eng

    def forward(self, img0: torch.Tensor, img1: torch.Tensor, capture: Optional[dict] = None):
        """img0/img1: (B,3,H,W) in [0,255], H and W multiples of 32 -> (disp, occ, conf), each (B,1,H,W) fp32
        ((B,1,2H,2W), disparity x2, with output_upsample).  Compute dtype: fp16 under ``torch.autocast(float16)``
        (how the reference is deployed, model_utils.py:76) or when the parameters are fp16, else fp32.

        Like the reference's module this is a pure function of (weights, img0, img1) that may be called from any thread, on any
        current stream and for a model on any device: launches go to the current stream of the IMAGES' device, captured graphs and
        scratch buffers are per stream, and a lock serialises the host-side enqueue per module.  ``torch.compile(model)`` (applied by
        the reference's demos, visualize_2d_simple.py:36-37) is accepted: the body is opaque to the tracer (there is nothing for
        a compiler to fuse -- every op is already a hand-written kernel)."""
        return self._forward_impl(img0, img1, capture)

    @torch.compiler.disable
    @torch.no_grad()
    def _forward_impl(self, img0: torch.Tensor, img1: torch.Tensor, capture: Optional[dict] = None, _nosplit: bool = False):
        if not img0.is_cuda:
            raise RuntimeError("s2m2_amd.S2M2 runs on MI355X only: move the model and the images to a CUDA(HIP) device "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
        p0 = next(self.parameters())
        if p0.device != img0.device or img1.device != img0.device:
            raise RuntimeError(f"model parameters on {p0.device}, images on {img0.device} / {img1.device}")
        if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16:
            dtype = torch.float16
        else:
            dtype = torch.float16 if p0.dtype == torch.float16 else torch.float32
        if img0.shape != img1.shape or img0.dim() != 4 or img0.shape[1] != 3:
            raise ValueError(f"expected two (B,3,H,W) images, got {tuple(img0.shape)} and {tuple(img1.shape)}")
        if img0.shape[-1] % 32 or img0.shape[-2] % 32:
            raise ValueError("image height and width must be multiples of 32 (pad with image_pad first)")
        from .engine import GraphRunner, check_limits, max_batch
        with self._lock, torch.cuda.device(img0.device):         # the planner queries the images' device
            check_limits(img0.shape[2], img0.shape[3], self.feature_channels, img0.shape[0], dtype,
                         use_pe="feat_pyramid.enc3s.0.self_attn.attn.pe_proj.weight" in self._table)
        if img0.shape[0] >= 2 and capture is None and not _nosplit and self._pair_streams() >= 2:
            return self._forward_pairs(img0, img1)
        nb = max_batch(img0.shape[2], img0.shape[3])
        if img0.shape[0] > nb:                                   # K5 indexes pixels with 24 bits: large batches run in slices
            if capture is not None:
                raise ValueError(f"capture needs a batch of at most {nb} pairs at this resolution")
            parts = [self._forward_impl(img0[i:i + nb], img1[i:i + nb]) for i in range(0, img0.shape[0], nb)]
            return tuple(torch.cat([p[k] for p in parts], 0) for k in range(3))
        # the tensors' device becomes current: the C ABI launches on the current stream and keeps per-device state
        with self._lock, torch.cuda.device(img0.device), torch.autocast("cuda", enabled=False):
            eng = self.engine(dtype)
            if capture is not None or os.environ.get("S2M2_GRAPH", "1") == "0":
                return eng.run(img0, img1, capture)
            # hipGraph replay from the second call with the same geometry on (the first call runs eagerly and warms everything up)
            stream = torch.cuda.current_stream(img0.device).cuda_stream
            key = (tuple(img0.shape), dtype, eng.k1_events is not None, stream)
            runner = self._graphs.get(key)
            if runner is None:
                if key not in self._seen:
                    self._seen.add(key)
                    return eng.run(img0, img1, None)
                # graph state (static input / output buffers) must be ordinary tensors even when the caller is in inference_mode
                with torch.inference_mode(False), torch.no_grad():
                    runner = GraphRunner(eng, img0.shape[0], img0.shape[2], img0.shape[3], split_k1=eng.k1_events is not None)
                self._graphs[key] = runner
                while len(self._graphs) > int(os.environ.get("S2M2_GRAPH_CACHE", "8")):       # LRU: each graph pins its activations
                    self._graphs.popitem(last=False)
            else:
                self._graphs.move_to_end(key)
            return runner(img0, img1)

    @staticmethod
    def _pair_streams() -> int:
        """S2M2_PAIR_STREAMS=n (default 2; 0 / 1: off): a batch of B >= 2 pairs runs as n chunks on n side streams instead of one batched launch
        sequence.  The pairs of a batch are independent, one forward leaves most CUs idle at the coarse pyramid levels, and a
        second forward in flight fills them (profiles/r06/ab_pair_streams.txt: 1216 x 1024 B = 2 6.87 -> 6.42 ms per pair)."""
        try:
            return int(os.environ.get("S2M2_PAIR_STREAMS", "2"))
        except ValueError:
            return 2

    def _forward_pairs(self, img0: torch.Tensor, img1: torch.Tensor):
        """The batch in n chunks, each a (batched) forward on its own side stream (one captured graph and scratch set per stream, as for any
        caller stream): B = 2 is two single-pair forwards in flight, B = 8 two batches of four; the caller's stream waits for them before it
        touches the results."""
        dev = img0.device
        n = self._pair_streams()
        cur = torch.cuda.current_stream(dev)
        # side streams per CALLER stream: the chunk graphs replay into static output buffers, and only the wait_stream pairs below order
        # a replay behind the caller's reads of the previous results -- two callers on two streams must not share them
        skey = (dev, cur.cuda_stream)
        with self._lock:
            streams = self._side_streams.get(skey)
            if streams is None or len(streams) != n:
                streams = self._side_streams[skey] = [torch.cuda.Stream(device=dev) for _ in range(n)]
        for s in streams:
            s.wait_stream(cur)                                     # the images were produced on the caller's stream
        parts = []
        B = img0.shape[0]
        per = (B + n - 1) // n                                     # n chunks of the batch, each a batched forward on its own stream
        for k, b0 in enumerate(range(0, B, per)):
            with torch.cuda.stream(streams[k % n]):
                parts.append(self._forward_impl(img0[b0:b0 + per], img1[b0:b0 + per], None, _nosplit=True))
        for s in streams:
            cur.wait_stream(s)
        for p in parts:
            for t in p:
                t.record_stream(cur)                               # allocated on a side stream, consumed (and freed) on the caller's
        return tuple(torch.cat([p[k] for p in parts], 0) for k in range(3))

    def is_warm(self, img_shape, dtype: torch.dtype = torch.float16) -> bool:
        """True when a captured graph exists for (B,3,H,W) ``img_shape`` on the current stream (next call = pure replay)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            return False
        B, n = int(img_shape[0]), self._pair_streams()
        if B >= 2 and n >= 2:                              # the batch runs as chunks on the side streams (_forward_pairs)
            streams = self._side_streams.get((dev, torch.cuda.current_stream(dev).cuda_stream))
            if streams is None or len(streams) != n:
                return False
            per = (B + n - 1) // n
            want = [((min(per, B - b0),) + tuple(img_shape[1:]), streams[k % n].cuda_stream) for k, b0 in enumerate(range(0, B, per))]
        else:
            want = [(tuple(img_shape), torch.cuda.current_stream(dev).cuda_stream)]
        return all(any(k[0] == shp and k[1] == dtype and k[3] == st for k in self._graphs) for shp, st in want)


def build_model(model_type: str, use_positivity: bool = True, refine_iter: int = 3, output_upsample: bool = False) -> S2M2:
    """Model-size table of the reference's ``load_model`` (model_utils.py:12-17) without the checkpoint IO."""
    c, ntr = MODEL_CONFIGS[model_type]
    return S2M2(c, 1, ntr, use_positivity=use_positivity, output_upsample=output_upsample, refine_iter=refine_iter)
