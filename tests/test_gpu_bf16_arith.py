"""GPU tests of the opt-in plain-bf16 arithmetic of the fused MLP kernel (CS_ARITH_BF16, `CSVAE_MLP_ARITH=bf16`).

Operands are rounded to bfloat16, one matrix instruction per product, float32 accumulation.  It is a throughput
mode, not float32-equivalent, so its yardstick is not the reference's float32 evaluation but a bfloat16-rounded
restatement of the same network (oracle.cs_oracle.resfc_forward_bf16).  The two differ by float32 summation
order only, except where such a last-bit difference moves an activation across a bfloat16 rounding boundary (one
bfloat16 unit, 2^-8 relative, in that activation): hence a tolerance of 2.5e-3 of the output scale on the largest
difference, and -- such crossings being rare -- a root-mean-square difference below a fifth of either one's
distance from the float32 result (the distance itself is checked too, so that the test cannot pass by running the
float32-equivalent kernel).  Nothing here has a counterpart in the reference: "parity unpinned" by its fixtures.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(43, 64, 5, 256, 4096), (32, 86, 2, 256, 4096), (21, 32, 2, 256, 1000), (8, 32, 1, 128, 777),
          (70, 128, 0, 64, 333), (11, 10, 2, 64, 37), (43, 64, 5, 256, 40000)]


@pytest.mark.parametrize("shape", SHAPES)
def test_plain_bf16_follows_the_bf16_rounded_restatement(shape):
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    in_dim, out_dim, nb, H, rows = shape
    w = O.make_resfc_weights(in_dim, out_dim, nb, H, seed=in_dim + rows)
    net = ResidualFCNetwork(in_dim, out_dim, nb, H).load_weight_dict(w).cuda().eval()
    x = torch.randn(rows, in_dim, generator=torch.Generator().manual_seed(5)) * 1.5
    y32 = O.resfc_forward(w, x)
    yb = O.resfc_forward_bf16(w, x)
    net.packed().set_arith("bf16")
    assert net.packed().get_arith() == "bf16"
    out = net(x.cuda()).cpu()
    net.packed().set_arith(None)
    scale = float(y32.abs().max())
    e_oracle = float((out - yb).abs().max()) / scale
    e_f32 = float((out - y32).abs().max()) / scale
    d_oracles = float((yb - y32).abs().max()) / scale
    print(f"shape {shape}: |gpu - bf16 restatement| {e_oracle:.2e}, |gpu - float32| {e_f32:.2e}, "
          f"|bf16 restatement - float32| {d_oracles:.2e} (of max|y|)")
    assert e_oracle < 2.5e-3
    rms = lambda t: float(t.double().pow(2).mean().sqrt())
    assert rms(out - yb) < 0.2 * rms(out - y32)
    assert e_f32 > 1e-4          # it really is the reduced-precision path
    assert e_f32 < 10 * d_oracles + 1e-3


def test_plain_bf16_lair_step_runs_and_keeps_observed_values():
    """One LAIR iteration at the MiniBooNE shape with both networks in plain bf16: finite outputs, observed entries
    untouched, indices inside their range; log-weights close to the float32-equivalent path's (they are sums of 43
    log-densities of bf16-accurate parameters)."""
    from oracle import cs_oracle as O
    from tests.helpers import make_model, seeded_inputs
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    B, K, R, D, Zd, H = 512, 4, 1, 43, 32, 256
    enc = O.make_resfc_weights(D, 2 * Zd, 5, H, 1)
    dec = O.make_resfc_weights(Zd, 2 * D, 2, H, 2)
    model = make_model(enc, dec, device="cuda:0")
    X, M = seeded_inputs(B, K, D, 3)
    g = torch.Generator().manual_seed(4)
    draws = {"eps_z": torch.randn(1, B, K + R, Zd, generator=g),
             "u_res": torch.rand(B, K + R, generator=g, dtype=torch.float64),
             "eps_x": torch.randn(B, K, D, generator=g)}
    ref = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=R, draws=draws)
    ref = [t.clone() if torch.is_tensor(t) else t for t in ref]
    for net in (model.var_latent_network, model.generator_network):
        net.packed().set_arith("bf16")
    try:
        out = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=R, draws=draws)
        Xn, lw = out[0].cpu(), out[7].cpu()
    finally:
        for net in (model.var_latent_network, model.generator_network):
            net.packed().set_arith(None)
    assert torch.isfinite(Xn).all() and torch.isfinite(lw).all()
    Mb = M.bool().expand_as(X)
    assert torch.equal(Xn[Mb], X[Mb])
    lw32 = ref[7].cpu()
    assert not torch.equal(lw, lw32)
    assert float((lw - lw32).abs().max() / (lw32.abs().max() + 1)) < 5e-2
