"""N>1 path on CPU: world_size-2 gloo processes shard pairs and gather the outputs to rank 0 in pair order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from s2m2_amd.shard import gather_outputs, gather_outputs_async, run_sharded, shard_indices


def _fake_forward(l, r):
    # deterministic per-pair "outputs": depends only on that pair's data (like the real forward)
    d = (l - r).mean(dim=1, keepdim=True)
    return d, d * 0.5, d.abs()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    left = torch.rand(4, 3, 8, 16, generator=g)
    right = torch.rand(4, 3, 8, 16, generator=g)
    out = run_sharded(_fake_forward, left, right, dist, dst=0)
    if rank == 0:
        ref = _fake_forward(left, right)
        q.put(all(torch.allclose(a, b) for a, b in zip(out, ref)))
        one = gather_outputs(_fake_forward(left[:1] + rank, right[:1]), dist, 0)
        q.put(tuple(one[0].shape) == (world, 1, 8, 16))
    else:
        assert out is None
        gather_outputs(_fake_forward(left[:1] + rank, right[:1]), dist, 0)
    for n_odd in (3, 1):                                      # uneven shards (n % world != 0; n < world): pad and drop, pair order kept
        o = run_sharded(_fake_forward, left[:n_odd], right[:n_odd], dist, dst=0)
        if rank == 0:
            q.put(all(torch.allclose(a, b) and a.shape[0] == n_odd for a, b in zip(o, _fake_forward(left[:n_odd], right[:n_odd]))))
        else:
            assert o is None
    # overlapped form used by bench.py: two gathers started back to back, each completed later, results in rank order
    src = [_fake_forward(left[:1] + rank + 10 * k, right[:1]) for k in range(2)]
    handles = [gather_outputs_async(o, dist, 0) for o in src]
    got = [h.wait() for h in handles]
    if rank == 0:
        ok = True
        for k in range(2):
            exp = torch.cat([_fake_forward(left[:1] + r + 10 * k, right[:1])[0] for r in range(world)], 0)
            ok = ok and torch.allclose(got[k][0], exp)
        q.put(ok)
    else:
        assert got == [None, None]
    dist.barrier()
    dist.destroy_process_group()


def _conf_worker(rank, world, port, q):
    """utils.compute_confidence_scores sharded over 2 gloo ranks == the unsharded evaluation (calibration objective).
    The device pieces (HIP image_pad, the model) are replaced by CPU stand-ins: this test is about the sharding arithmetic."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import s2m2_amd.utils as U
    U.image_pad = lambda img, factor=32: torch.nn.functional.pad(img.float(), (0, (-img.shape[-1]) % factor, 0, (-img.shape[-2]) % factor))
    U.image_crop = lambda img, shape: img[..., :shape[0], :shape[1]]

    def model(l, r):
        c = torch.sigmoid((l - r).mean(dim=1, keepdim=True) / 50)
        return c, c, c
    g = torch.Generator().manual_seed(1)
    lefts = torch.rand(6, 3, 230, 250, generator=g) * 255
    rights = torch.rand(6, 3, 230, 250, generator=g) * 255
    sharded = U.compute_confidence_scores(model, lefts, rights, "cpu", dist=dist)
    single = U.compute_confidence_scores(model, lefts, rights, "cpu", batch=4)
    ref = torch.stack([model(lefts[i:i + 1], rights[i:i + 1])[2][0, 0, 100:-100, 100:-100].mean() for i in range(6)])
    q.put(bool(torch.allclose(sharded, ref, atol=1e-6)) and bool(torch.allclose(single, ref, atol=1e-6)) and sharded.shape == (6,))
    odd = U.compute_confidence_scores(model, lefts[:5], rights[:5], "cpu", dist=dist)        # 5 pairs over 2 ranks: pad and drop
    q.put(bool(torch.allclose(odd, ref[:5], atol=1e-6)) and odd.shape == (5,))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world2_gloo_sharded_confidence_objective():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_conf_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(4)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [True] * 4


def test_shard_indices_cover_all_pairs():
    from s2m2_amd.shard import padded_shard
    for n, w in [(8, 8), (8, 2), (6, 3), (5, 2), (3, 8), (9, 8)]:
        seen = sorted(p for r in range(w) for p in shard_indices(n, r, w))
        assert seen == list(range(n))
        per = -(-n // w)
        for r in range(w):
            idx, real = padded_shard(n, r, w)
            assert len(idx) == per and idx[:real] == shard_indices(n, r, w) and all(0 <= i < n for i in idx)


@pytest.mark.timeout(120)
def test_world2_gloo_gather_in_pair_order():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(5)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [True] * 5
