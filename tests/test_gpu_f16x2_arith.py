"""GPU tests of the fp16-split arithmetic of the fused MLP kernel (CS_ARITH_F16X2, the default).

Every float32 operand is split into two fp16 terms (11 + 11 significand bits) and a product is accumulated in
float32 from three term products; the neglected term is below 2^-22 |a||b|.  Checked like the default arithmetic:
against a float64 evaluation of the same network its error must be of the size of the float32 evaluations' own
rounding error.  fp16 has 5 exponent bits: every operand row and every layer's weights are normalised by exact powers
of two, and the tests below probe that (inputs and weights of any scale, rows that are small throughout, and the one
case a per-row scale cannot help).  Nothing here is pinned by a fixture of the reference ("parity unpinned"); the
samplers' parity tests run under this arithmetic too (tests/helpers.arith_fixture)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def to64(w):
    f = lambda t: t.double()
    return {"W0": f(w["W0"]), "b0": f(w["b0"]), "blocks": [tuple(f(t) for t in b) for b in w["blocks"]],
            "Wf": f(w["Wf"]), "bf": f(w["bf"])}


def _errors(shape, x_scale=1.5):
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    in_dim, out_dim, nb, H, rows = shape
    w = O.make_resfc_weights(in_dim, out_dim, nb, H, seed=in_dim + rows)
    net = ResidualFCNetwork(in_dim, out_dim, nb, H).load_weight_dict(w).cuda().eval()
    x = torch.randn(rows, in_dim, generator=torch.Generator().manual_seed(5)) * x_scale
    y64 = O.resfc_forward(to64(w), x.double())
    out = {}
    for arith in ("f32", "bf16x3", "f16x2"):
        net.packed().set_arith(arith)
        assert net.packed().get_arith() == arith
        out[arith] = net(x.cuda()).cpu().double()
    net.packed().set_arith(None)
    scale = y64.abs().max()
    err = {k: float((v - y64).abs().max() / scale) for k, v in out.items()}
    err["cpu32"] = float((O.resfc_forward(w, x).double() - y64).abs().max() / scale)
    return err, out


@pytest.mark.parametrize("shape", [(43, 64, 5, 256, 4096), (32, 86, 2, 256, 4096), (21, 32, 2, 256, 1000),
                                   (8, 32, 1, 128, 777), (70, 128, 0, 64, 333), (11, 10, 2, 64, 37), (43, 64, 5, 256, 40000)])
def test_fp16_split_is_float32_sized_error(shape):
    err, out = _errors(shape)
    print(f"shape {shape}: max error / max|y| vs float64: f16x2 {err['f16x2']:.2e}, bf16x3 {err['bf16x3']:.2e}, "
          f"exact f32 {err['f32']:.2e}, torch CPU float32 {err['cpu32']:.2e}")
    assert err["f16x2"] < 3e-6
    assert err["f16x2"] <= 4 * max(err["f32"], err["cpu32"]) + 2e-7
    assert not torch.equal(out["f16x2"], out["bf16x3"])          # it is its own kernel


@pytest.mark.parametrize("x_scale", [1e-3, 30.0])
def test_fp16_split_away_from_unit_scale(x_scale):
    err, _ = _errors((43, 64, 5, 256, 2048), x_scale)
    print(f"input scale {x_scale}: f16x2 {err['f16x2']:.2e}, bf16x3 {err['bf16x3']:.2e}, exact f32 {err['f32']:.2e}")
    assert err["f16x2"] < 1e-5


@pytest.mark.parametrize("case", ["inputs", "weights"])
def test_fp16_split_has_no_range_limit_of_its_own(case):
    """fp16 has 5 exponent bits, but every operand row is normalised by a power of two (its largest magnitude into
    [1, 2)) and every layer's weights by another (largest weight into [2^14, 2^15)), exactly: activations and weights
    far beyond 65504 or far below 2^-14 are handled like ordinary ones -- rows with an outlier feature, rows that are
    tiny or huge throughout, a weight of 7e4, an output row of weights scaled by 1e-7.  Only a non-finite weight is
    refused.  (What stays out of reach is an outlier input that meets an outlier weight of another input in the same
    row: the other input then keeps 11 bits at the outlier's scale; see the wide_weights case below.)"""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    w = O.make_resfc_weights(43, 64, 2, 256, seed=4)
    x = torch.randn(64, 43, generator=torch.Generator().manual_seed(1))
    if case == "inputs":
        x[7, 3] = 1.0e5
        x[9] *= 1.0e-9
        x[11] *= 1.0e12
    else:
        w["W0"][0, 0] = 7.0e4
        w["Wf"][3] *= 1.0e-7
    net = ResidualFCNetwork(43, 64, 2, 256).load_weight_dict(w).cuda().eval()
    y64 = O.resfc_forward(to64(w), x.double())
    net.packed().set_arith("f16x2")
    y = net(x.cuda()).cpu().double()
    net.packed().set_arith("bf16x3")
    y3 = net(x.cuda()).cpu().double()
    assert torch.isfinite(y).all()
    rowscale = y64.abs().amax(dim=1, keepdim=True)
    e2, e3 = float(((y - y64).abs() / rowscale).max()), float(((y3 - y64).abs() / rowscale).max())
    print(f"{case}: per-row max error / max|y_row| vs float64: f16x2 {e2:.2e}, bf16x3 {e3:.2e}")
    assert e2 < 4e-6 and e2 <= 4 * e3 + 2e-7
    if case == "weights":
        w["W0"][1, 1] = float("inf")
        bad = ResidualFCNetwork(43, 64, 2, 256).load_weight_dict(w).cuda().eval()
        bad.packed().set_arith("f16x2")
        with pytest.raises(_lib.CsError, match="not finite"):
            bad(x.cuda())


def test_fp16_split_outlier_input_beside_an_outlier_weight():
    """An input outlier of 1e5 in a row whose other inputs meet a weight of 7e4: the ordinary inputs sit 2^-17 below
    their row's largest and still carry a normal low-order term; the ordinary weights sit 2^-19 below their layer's
    largest and keep 2^-21 of themselves."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    w = O.make_resfc_weights(43, 64, 2, 256, seed=4)
    w["W0"][0, 0] = 7.0e4
    x = torch.randn(256, 43, generator=torch.Generator().manual_seed(1))
    x[::2, 3] = 1.0e5
    net = ResidualFCNetwork(43, 64, 2, 256).load_weight_dict(w).cuda().eval()
    y64 = O.resfc_forward(to64(w), x.double())
    err = {}
    for arith in ("f32", "bf16x3", "f16x2"):
        net.packed().set_arith(arith)
        y = net(x.cuda()).cpu().double()
        err[arith] = float(((y - y64).abs() / y64.abs().amax(dim=1, keepdim=True)).max())
    net.packed().set_arith(None)
    print(f"outlier input beside an outlier weight: per-row max error / max|y_row| vs float64: {err}")
    assert err["f16x2"] < 4e-6 and err["f16x2"] <= 3 * max(err["f32"], err["bf16x3"]) + 2e-7


@pytest.mark.parametrize("case", ["wide_inputs", "wide_weights", "small_everything"])
def test_fp16_split_dynamic_range_within_a_row(case):
    """Operands of very different sizes.  Rows and layers are normalised by powers of two, so inputs of any scale and
    layers that are small throughout cost nothing.  What a per-row scale cannot undo is a wide range INSIDE a row that the
    next layer's weights compensate (second case: hidden units scaled by 1e-3 ... 10 and un-scaled by the next layer's
    columns): operands 1e-4 of their row's largest keep 11 bits, the error rises to a few 1e-6 -- still inside 1e-5."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    g = torch.Generator().manual_seed(7)
    w = O.make_resfc_weights(43, 64, 2, 256, seed=11)
    x = torch.randn(2048, 43, generator=g)
    if case == "wide_inputs":
        x = x * torch.pow(10.0, torch.rand(43, generator=g) * 6 - 4)          # features scaled by 1e-4 ... 1e2
    elif case == "wide_weights":
        s = torch.pow(10.0, torch.rand(256, generator=g) * 4 - 3)             # hidden units scaled by 1e-3 ... 10,
        w["W0"] = w["W0"] * s[:, None]                                        # undone by the next layer's columns
        w["b0"] = w["b0"] * s
        W1, b1, W2, b2 = w["blocks"][0]
        w["blocks"][0] = (W1 / s[None, :], b1, W2, b2)
    else:
        x = x * 1e-3
        w["b0"] = w["b0"] * 1e-3
        w["blocks"] = [(W1, b1 * 1e-3, W2, b2 * 1e-3) for W1, b1, W2, b2 in w["blocks"]]
        w["bf"] = w["bf"] * 1e-3
    net = ResidualFCNetwork(43, 64, 2, 256).load_weight_dict(w).cuda().eval()
    y64 = O.resfc_forward(to64(w), x.double())
    err = {}
    for arith in ("f32", "bf16x3", "f16x2"):
        net.packed().set_arith(arith)
        err[arith] = float((net(x.cuda()).cpu().double() - y64).abs().max() / y64.abs().max())
    net.packed().set_arith(None)
    print(f"{case}: max error / max|y| vs float64: f16x2 {err['f16x2']:.2e}, bf16x3 {err['bf16x3']:.2e}, exact f32 {err['f32']:.2e}")
    assert err["f32"] < 3e-6 and err["bf16x3"] < 3e-6
    # hidden units 1e-4 of their row's largest keep a normal low-order term (rows are normalised into fp16's top
    # binade): the in-row spread costs nothing
    assert err["f16x2"] < 1e-6 and err["f16x2"] <= 1.5 * max(err["f32"], err["bf16x3"])


def test_fp16_split_rows_and_layers_of_zeros():
    """Rows that are exactly zero (and stay zero through layers whose biases are zero) carry no scale: the exponent
    bookkeeping must not drift with them.  Deep net (8 blocks), zero input rows, zero biases in the first five layers."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    w = O.make_resfc_weights(21, 32, 8, 128, seed=2)
    w["b0"] = torch.zeros_like(w["b0"])
    w["blocks"] = [(W1, torch.zeros_like(b1) if i < 2 else b1, W2, torch.zeros_like(b2) if i < 2 else b2)
                   for i, (W1, b1, W2, b2) in enumerate(w["blocks"])]
    x = torch.randn(96, 21, generator=torch.Generator().manual_seed(3))
    x[::3] = 0.0
    net = ResidualFCNetwork(21, 32, 8, 128).load_weight_dict(w).cuda().eval()
    y64 = O.resfc_forward(to64(w), x.double())
    net.packed().set_arith("f16x2")
    y = net(x.cuda()).cpu().double()
    net.packed().set_arith(None)
    assert torch.isfinite(y).all()
    assert float((y - y64).abs().max() / y64.abs().max()) < 2e-6
    assert float((y[::3] - y64[::3]).abs().max() / y64[::3].abs().max()) < 2e-6
