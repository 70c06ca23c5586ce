"""GPU tests of the recorded launch plans (include/s2m2_hip.h: s2m2_plan_*; csrc/plan.h, runtime.hip): externals follow their buffers on replay,
and -- round 6 -- only POINTER arguments / descriptor fields are relocated: a size or a stride that happens to fall inside an external's address
range is left alone."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


class _Range:
    """an 'external buffer' given by address and size only"""

    def __init__(self, base, nbytes):
        self.base, self.nbytes = base, nbytes

    def data_ptr(self):
        return self.base

    def numel(self):
        return self.nbytes

    def element_size(self):
        return 1


def test_externals_follow_their_buffers_and_sizes_are_never_relocated(hip):
    C, rows = 128, 3000
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(rows, C, device="cuda", generator=g).half()
    out = torch.empty_like(x)
    ref = hip.layernorm(x)
    # a decoy external whose "address range" [64, 64 + 1 MB) holds the call's rows (3000), channel count and strides (128): word-scanning
    # relocation would rewrite them; the pointer mask of the positional pack (plan.h: mark_pack) names the two pointer arguments only
    plan = hip.Plan()
    with plan.record([x, out, _Range(64, 1 << 20)]):
        hip.layernorm(x, out=out)
    assert plan.launches == 1
    assert plan.patches(0) == 1 and plan.patches(1) == 1 and plan.patches(2) == 0, (plan.patches(0), plan.patches(1), plan.patches(2))
    x2 = torch.randn(rows, C, device="cuda", generator=g).half()
    out2 = torch.zeros_like(x2)
    plan.run([x2, out2, _Range(1 << 40, 0)])                       # the decoy "moved": nothing may follow it
    assert torch.equal(out2, hip.layernorm(x2)) and torch.equal(out, ref)


def test_descriptor_calls_relocate_their_pointer_fields_only(hip):
    """K9 one-stage chain through s2m2_chain_desc: x / out are externals; the decoy range covers the descriptor's small integers (rows, C, strides)."""
    from s2m2_amd import pack
    C, rows = 128, 2048
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(rows, C, device="cuda", generator=g).half()
    w = (torch.randn(C, C, device="cuda", generator=g) / 11.3).half().contiguous()
    st = [(pack.chain_frag(w), None, hip.ACT_GELU, None)]
    ref = hip.mlp_chain(x, st, frag=True)
    plan = hip.Plan()
    with plan.record([x, _Range(1, 1 << 16)]):
        y = hip.mlp_chain(x, st, frag=True)
    assert plan.patches(0) >= 1 and plan.patches(1) == 0
    x2 = torch.randn(rows, C, device="cuda", generator=g).half()
    plan.run([x2, _Range(1 << 41, 0)])
    assert torch.equal(y, hip.mlp_chain(x2, st, frag=True)) and not torch.equal(y, ref)
