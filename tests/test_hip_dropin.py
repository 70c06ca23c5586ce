"""GPU tests of the drop-in boundary around ``S2M2.forward``: checkpoint loading
(model_utils.py:11-48, s2m2.py:69-78), the batched calibration objective (calibration/base.py:15-36, cem.py:66-72),
``torch.compile(model)`` (visualize_2d_simple.py:36-37), re-entrancy across streams / threads / inference_mode, weight updates."""
import threading

import pytest
import torch

from s2m2_amd import utils as U
from s2m2_amd.model import S2M2
from s2m2_amd.weights import noise_pair, seeded_state_dict, synthetic_pair

pytestmark = pytest.mark.gpu


def _direct(seed=0, ri=1):
    m = S2M2(128, 1, 1, use_positivity=True, refine_iter=ri)
    m.load_state_dict(seeded_state_dict(128, 1, 1, seed), strict=True)
    return m.cuda().eval()


def _fwd16(m, l, r):
    with torch.autocast("cuda", dtype=torch.float16):
        return m(l, r)


def test_load_model_checkpoint_round_trip(tmp_path, capsys):
    """A seeded {"state_dict": ...} saved as CH128NTR1.pth and loaded through utils.load_model gives bit-identical outputs to the
    directly initialised model; a deliberately mis-shaped tensor takes my_load_state_dict's skip branch (s2m2.py:72-77) and an
    unknown key is ignored (strict=False)."""
    sd = seeded_state_dict(128, 1, 1, 3)
    torch.save({"state_dict": sd}, tmp_path / "CH128NTR1.pth")
    m = U.load_model(str(tmp_path), "S", use_positivity=True, refine_iter=1, device="cuda")
    assert m is not None and not m.training and next(m.parameters()).is_cuda
    assert "Model loaded" in capsys.readouterr().out
    l, r = synthetic_pair(64, 96, 1, 8, 3)
    l, r = l.cuda(), r.cuda()
    ref = _direct(3)(l, r)
    out = m(l, r)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    # fp16 deployment mode through the same checkpoint
    for a, b in zip(_fwd16(m, l, r), _fwd16(_direct(3), l, r)):
        assert torch.equal(a, b)
    # mis-shaped + unknown keys
    bad = dict(sd)
    name = "refiner.disp_update.2.weight"
    bad[name] = torch.zeros(3, 5)
    bad["not.a.parameter"] = torch.zeros(1)
    (tmp_path / "b").mkdir()
    torch.save({"state_dict": bad}, tmp_path / "b" / "CH128NTR1.pth")
    m2 = U.load_model(str(tmp_path / "b"), "S", use_positivity=True, refine_iter=1, device="cuda")
    txt = capsys.readouterr().out
    assert m2 is not None and f"Skip loading parameter: {name}" in txt
    fresh = S2M2(128, 1, 1)                                    # the skipped tensor keeps the constructor's value
    assert torch.equal(m2.state_dict()[name].cpu(), fresh.state_dict()[name])
    other = "refiner.disp_update.0.weight"
    assert torch.equal(m2.state_dict()[other].cpu(), sd[other])
    assert torch.isfinite(m2(l, r)[0]).all()
    # missing file: the reference prints and returns None
    assert U.load_model(str(tmp_path / "nowhere"), "S") is None
    assert "Error loading model" in capsys.readouterr().out


def test_weight_updates_are_seen():
    """load_state_dict and in-place updates re-pack; writes through .data need invalidate() (documented)."""
    m = _direct(0)
    l, r = synthetic_pair(64, 96, 1, 8, 1)
    l, r = l.cuda(), r.cuda()
    a = m(l, r)[0].clone()
    a2 = m(l, r)[0].clone()                                    # second call: captured graph
    m.load_state_dict(seeded_state_dict(128, 1, 1, 5), strict=True)
    b = m(l, r)[0].clone()
    assert not torch.equal(a, b)
    assert torch.equal(b, _direct(5)(l, r)[0])
    m.load_state_dict(seeded_state_dict(128, 1, 1, 0), strict=True)
    assert torch.equal(m(l, r)[0], a) and torch.equal(a, a2)
    with torch.no_grad():
        for p, q in zip(m.parameters(), _direct(5).parameters()):
            p.data.copy_(q.data)                               # bypasses the version counter
    m.invalidate()
    assert torch.equal(m(l, r)[0], b)


def test_batched_confidence_objective_equals_sequential():
    """compute_confidence_scores (one batched forward / slices of it) == N calls of compute_confidence_score, images that are
    not multiples of 32 (pad + crop on the device), fp16 deployment mode like the reference's objective."""
    m = _direct(2, ri=1)
    N, H, W = 4, 250, 300
    l, r = noise_pair(H, W, N, 7)
    dev = torch.device("cuda")
    seq = torch.tensor([U.compute_confidence_score(m, l[i:i + 1], r[i:i + 1], dev) for i in range(N)])
    all_at_once = U.compute_confidence_scores(m, l, r, dev)
    by_two = U.compute_confidence_scores(m, l, r, dev, batch=2)
    assert all_at_once.shape == (N,) and all_at_once.dtype == torch.float32
    assert float((all_at_once - seq).abs().max()) < 1e-5, (all_at_once, seq)
    assert float((by_two - seq).abs().max()) < 1e-5
    assert float(seq.std()) > 0                                # the pairs really differ


def test_graph_replay_batch_gt_one_and_batch_slicing(monkeypatch):
    m = _direct(0)
    l, r = synthetic_pair(64, 96, 3, 8, 2)
    l, r = l.cuda(), r.cuda()
    eager = m(l, r)
    replay = m(l, r)
    replay2 = m(l, r)
    for a, b, c in zip(eager, replay, replay2):
        assert torch.equal(a, b) and torch.equal(a, c)
    singles = [m(l[i:i + 1], r[i:i + 1])[0] for i in range(3)]
    assert float((torch.cat(singles) - eager[0]).abs().max()) < 1e-3
    # batches beyond the 24-bit pixel index of K5 run in slices
    import s2m2_amd.engine as E
    monkeypatch.setattr(E, "max_batch", lambda H, W: 2)
    sliced = m(l, r)
    assert float((sliced[0] - eager[0]).abs().max()) < 1e-3 and sliced[0].shape == eager[0].shape


def test_batch_in_chunks_on_two_streams_equals_one_batched_forward(monkeypatch):
    """S2M2_PAIR_STREAMS (model.py: _forward_pairs): a batch runs as two chunks on two side streams; per pair the results are those of the
    batched launch sequence (pairs of a batch are independent; a chunk of one pair equals the single-pair forward bit for bit), the caller's
    stream is ordered behind the side streams, and is_warm reports the chunk graphs."""
    m = _direct(0)
    for B in (2, 3, 5):
        l, r = synthetic_pair(64, 96, B, 8, 20 + B)
        l, r = l.cuda(), r.cuda()
        monkeypatch.setenv("S2M2_PAIR_STREAMS", "0")
        whole = [t.clone() for t in m(l, r)]
        monkeypatch.setenv("S2M2_PAIR_STREAMS", "2")
        assert not m.is_warm(l.shape, torch.float32)
        for _ in range(3):                                     # eager, capture, replay of both chunks
            got = m(l, r)
            total = float(got[0].sum())                        # consumed on the caller's stream right away
            for a, b in zip(got, whole):
                assert a.shape == b.shape and float((a - b).abs().max()) < 1e-3
            assert abs(total - float(whole[0].sum())) < 1e-1 * B
        assert m.is_warm(l.shape, torch.float32)
        if B == 2:
            single = m(l[1:2], r[1:2])
            assert torch.equal(single[0], got[0][1:2])
    # two host threads, each on its own stream, each with batches of two: every caller stream has its own side streams (the chunk graphs
    # replay into static buffers that only the caller's wait_stream pairs protect)
    data = {}
    for tag, seed in (("a", 31), ("b", 32)):
        l, r = synthetic_pair(64, 96, 2, 8, seed)
        data[tag] = (l.cuda(), r.cuda())
        data[tag + "_ref"] = m(*data[tag])[0].clone()
    torch.cuda.synchronize()
    res = {}

    def work(tag):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(6):
                out = m(*data[tag])
                acc = out[0].clone()                           # read right away on the caller's stream
            res[tag] = acc
        st.synchronize()
    ts = [threading.Thread(target=work, args=(t,)) for t in ("a", "b")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert torch.equal(res["a"], data["a_ref"]) and torch.equal(res["b"], data["b_ref"])


def test_torch_compile_wrapper_is_accepted():
    m = _direct(0)
    l, r = synthetic_pair(64, 96, 1, 8, 4)
    l, r = l.cuda(), r.cuda()
    ref = m(l, r)
    cm = torch.compile(m)
    for _ in range(3):
        out = cm(l, r)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)


def test_inference_mode_then_plain_call_and_streams_and_threads():
    m = _direct(0)
    l, r = synthetic_pair(64, 96, 1, 8, 6)
    l, r = l.cuda(), r.cuda()
    ref = [t.clone() for t in m(l, r)]
    with torch.inference_mode():                               # run_stereo_matching's context: eager, capture, replay
        for _ in range(3):
            o = m(l, r)
    assert torch.equal(o[0], ref[0])
    o = m(l, r)                                                # plain no_grad call replays the graph captured above
    assert torch.equal(o[0], ref[0])
    # a second stream gets its own graph and scratch buffers
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            o2 = m(l, r)
    s.synchronize()
    assert torch.equal(o2[0], ref[0])
    # two host threads on two streams, different inputs, interleaved
    l2, r2 = synthetic_pair(64, 96, 1, 8, 7)
    l2, r2 = l2.cuda(), r2.cuda()
    ref2 = [t.clone() for t in m(l2, r2)]
    torch.cuda.synchronize()
    res = {}

    def work(tag, a, b):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(6):
                out = m(a, b)
            res[tag] = out[0].clone()
        st.synchronize()
    ts = [threading.Thread(target=work, args=("a", l, r)), threading.Thread(target=work, args=("b", l2, r2))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert torch.equal(res["a"], ref[0]) and torch.equal(res["b"], ref2[0])


def test_limits_are_validated_up_front():
    m = _direct(0)
    big = torch.zeros(1, 3, 32, 4864, device="cuda")          # w/4 + 1 > 1204: K2's LDS row budget
    with pytest.raises(ValueError, match="wide"):
        m(big, big)
    with pytest.raises(ValueError, match="multiples of 32"):
        m(big[..., :100], big[..., :100])
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))
