"""GPU tests of the split-bf16 arithmetic of the fused MLP kernel (cs_set_mlp_arith).

Every float32 operand is split exactly into three bfloat16 terms and a product is accumulated in float32
from the six leading term products, so the results are float32-equivalent, not bit-identical to the
exact-float32 matrix instructions.  The claim is checked against a float64 evaluation of the same
network: the error of the split arithmetic must be of the size of the float32 path's own rounding error.
(The parity tests of the samplers run under both arithmetics: see `arith_fixture` in tests/helpers.py.)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def to64(w):
    f = lambda t: t.double()
    return {"W0": f(w["W0"]), "b0": f(w["b0"]), "blocks": [tuple(f(t) for t in b) for b in w["blocks"]],
            "Wf": f(w["Wf"]), "bf": f(w["bf"])}


@pytest.mark.parametrize("shape", [(43, 64, 5, 256, 4096), (32, 86, 2, 256, 4096), (21, 32, 2, 256, 1000),
                                   (8, 32, 1, 128, 777), (70, 128, 0, 64, 333), (11, 10, 2, 64, 37)])
def test_split_arithmetic_is_float32_equivalent(shape):
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    in_dim, out_dim, nb, H, rows = shape
    w = O.make_resfc_weights(in_dim, out_dim, nb, H, seed=in_dim + rows)
    net = ResidualFCNetwork(in_dim, out_dim, nb, H).load_weight_dict(w).cuda().eval()
    x = torch.randn(rows, in_dim, generator=torch.Generator().manual_seed(5)) * 1.5
    y64 = O.resfc_forward(to64(w), x.double())
    out = {}
    prev = _lib.get_mlp_arith()
    try:
        for arith in ("f32", "bf16x3"):
            _lib.set_mlp_arith(arith)
            assert _lib.get_mlp_arith() == arith
            out[arith] = net(x.cuda()).cpu().double()
    finally:
        _lib.set_mlp_arith(prev)
    scale = y64.abs().max()
    err = {k: float((v - y64).abs().max() / scale) for k, v in out.items()}
    cpu = float((O.resfc_forward(w, x).double() - y64).abs().max() / scale)   # torch CPU float32, the reference's arithmetic
    print(f"shape {shape}: max error / max|y| vs float64: exact-f32 kernel {err['f32']:.2e}, split-bf16 kernel {err['bf16x3']:.2e}, "
          f"torch CPU float32 {cpu:.2e}")
    assert err["f32"] < 2e-6 and err["bf16x3"] < 2e-6
    # of the size of float32 rounding: within a small factor of the other float32 evaluations
    assert err["bf16x3"] <= 3 * max(err["f32"], cpu) + 1e-7
    # and the two kernels agree to the tolerance of the parity tests, with room to spare
    assert float((out["f32"] - out["bf16x3"]).abs().max() / scale) < 2e-6


def test_arithmetic_setting_is_validated():
    from vae_conditional_sampling_amd import _lib
    prev = _lib.get_mlp_arith()
    assert _lib.lib().cs_set_mlp_arith(7) != 0
    assert b"unknown arithmetic" in _lib.lib().cs_last_error()
    assert _lib.get_mlp_arith() == prev


def test_arithmetic_is_selectable_per_network_handle():
    """cs_mlp_set_arith: a handle's own arithmetic overrides the process-wide setting (two networks with different
    arithmetics in one process), CS_ARITH_DEFAULT hands it back."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    w = O.make_resfc_weights(43, 64, 2, 256, seed=9)
    x = torch.randn(500, 43, generator=torch.Generator().manual_seed(1)).cuda()
    prev = _lib.get_mlp_arith()
    try:
        ref = {}
        for arith in ("f32", "bf16x3"):
            _lib.set_mlp_arith(arith)
            ref[arith] = ResidualFCNetwork(43, 64, 2, 256).load_weight_dict(w).cuda().eval()(x).clone()
        assert not torch.equal(ref["f32"], ref["bf16x3"])
        _lib.set_mlp_arith("bf16x3")
        a = ResidualFCNetwork(43, 64, 2, 256).load_weight_dict(w).cuda().eval()
        b = ResidualFCNetwork(43, 64, 2, 256).load_weight_dict(w).cuda().eval()
        a.packed().set_arith("f32")
        assert a.packed().get_arith() == "f32" and b.packed().get_arith() == "bf16x3"
        assert torch.equal(a(x), ref["f32"]) and torch.equal(b(x), ref["bf16x3"])
        _lib.set_mlp_arith("f32")          # the process-wide setting moves; the pinned handle does not follow
        b.packed().set_arith("bf16x3")
        assert torch.equal(b(x), ref["bf16x3"]) and torch.equal(a(x), ref["f32"])
        b.packed().set_arith(None)
        assert torch.equal(b(x), ref["f32"])
    finally:
        _lib.set_mlp_arith(prev)


def test_reserved_layer_program_workspace_gives_the_same_results():
    """cs_net_reserve allocates a layer program's activation workspace ahead of the first launch (and is a no-op
    for fused-MLP handles); results do not depend on who allocated it."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    w = O.make_resfc_weights(5, 6, 1, 512, seed=3)          # hidden 512: runs as a layer program
    x = torch.randn(300, 5, generator=torch.Generator().manual_seed(2)).cuda()
    lazy = ResidualFCNetwork(5, 6, 1, 512).load_weight_dict(w).cuda().eval()
    y0 = lazy(x).clone()
    eager = ResidualFCNetwork(5, 6, 1, 512).load_weight_dict(w).cuda().eval()
    eager.packed().reserve(300)
    eager.packed().reserve(10)     # never shrinks
    assert torch.equal(eager(x), y0)
    fused = ResidualFCNetwork(5, 6, 1, 64).load_weight_dict(O.make_resfc_weights(5, 6, 1, 64, seed=3)).cuda().eval()
    fused.packed().reserve(1000)
