"""GPU tests of the small-launch kernel of the fused MLP (fp16 split): a workgroup per 16-row tile, the layer's output
features split over its wavefronts.  Its network outputs must reproduce the kernel that keeps a row tile on one wavefront BIT FOR BIT
(same operand scaling, same term products in the same order), in every mode: plain forward, encoder (proposal
parameters), decoder (proposal sampling, generator parameters), Bernoulli decoder, the AC-MWG step's gathers and
overrides; the masked log-likelihood, a sum over the row, agrees to its last bits."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def narrow():
    """Callable that sets the kernel choice (0: never, 1: whenever an instance exists); restored afterwards."""
    from vae_conditional_sampling_amd import _lib
    L = _lib.lib()
    prev_arith = _lib.set_mlp_arith("f16x2")
    prev = L.cs_debug_set_narrow(-1)
    yield lambda mode: L.cs_debug_set_narrow(mode)
    L.cs_debug_set_narrow(prev)
    _lib.set_mlp_arith(prev_arith)


@pytest.mark.parametrize("shape", [(43, 64, 5, 256, 2048), (32, 86, 2, 256, 2560), (21, 32, 2, 256, 1000), (16, 42, 2, 256, 37),
                                   (8, 32, 1, 128, 777), (70, 128, 0, 128, 333), (11, 10, 2, 64, 37), (100, 64, 3, 128, 515), (60, 64, 3, 64, 100),
                                   (8, 16, 2, 256, 16), (43, 64, 5, 256, 9000)])
def test_forward_is_bit_identical_to_the_row_tile_per_wave_kernel(narrow, shape):
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    in_dim, out_dim, nb, H, rows = shape
    w = O.make_resfc_weights(in_dim, out_dim, nb, H, seed=in_dim * 3 + rows)
    net = ResidualFCNetwork(in_dim, out_dim, nb, H).load_weight_dict(w).cuda().eval()
    x = (torch.randn(rows, in_dim, generator=torch.Generator().manual_seed(9)) * 1.7).cuda()
    narrow(0)
    y_wide = net(x).clone()
    narrow(1)
    y_narrow = net(x).clone()
    assert torch.isfinite(y_wide).all()
    assert torch.equal(y_wide, y_narrow), float((y_wide - y_narrow).abs().max())
    # and within float32 rounding of the oracle
    ref = O.resfc_forward(w, x.cpu())
    assert float((y_narrow.cpu() - ref).abs().max() / ref.abs().max()) < 3e-6


@pytest.mark.parametrize("cfg", [
    dict(B=512, K=4, R=1, I=1, D=43, Zd=32, H=256, nbe=5, nbd=2),      # MiniBooNE configuration's batch
    dict(B=256, K=16, R=1, I=1, D=21, Zd=16, H=256, nbe=2, nbd=2),     # HEPMASS shape
    dict(B=33, K=3, R=2, I=3, D=12, Zd=8, H=128, nbe=1, nbd=1),
    dict(B=7, K=2, R=0, I=1, D=5, Zd=3, H=64, nbe=1, nbd=0),
    dict(B=40, K=4, R=1, I=1, D=43, Zd=32, H=256, nbe=5, nbd=2, gen="bern"),
])
def test_lair_step_is_bit_identical_with_either_kernel(narrow, cfg):
    """A whole LAIR iteration on counter-RNG draws: proposal parameters, proposals, generator parameters, masked
    log-likelihood, log-weights, indices and imputations, bit for bit."""
    from oracle import cs_oracle as O
    from tests.helpers import make_model, seeded_inputs
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling import engine
    B, K, R, I, D, Zd, H = (cfg[k] for k in ("B", "K", "R", "I", "D", "Zd", "H"))
    bern = cfg.get("gen") == "bern"
    enc = O.make_resfc_weights(D, 2 * Zd, cfg["nbe"], H, 41)
    dec = O.make_resfc_weights(Zd, D if bern else 2 * D, cfg["nbd"], H, 42)
    model = make_model(enc, dec, gen_dist="bern" if bern else "normal_with_eps")
    X0, M = seeded_inputs(B, K, D, 43)
    if bern:
        X0 = (X0 > 0).float()
    M8 = engine.mask_u8(M.cuda())
    ws = engine.lair_workspace(model, B, K, R, I, torch.device("cuda", 0))
    lcfg = engine.lair_cfg(model, ws, weighting="dmis", scheme="gr", method="multinomial", cluster_size=0, weight_alpha=None)
    outs = []
    for mode in (0, 1):
        narrow(mode)
        X = X0.cuda().contiguous().clone()
        engine.lair_step(model, ws, X, M8, lcfg, rng=_lib.make_rng(3, 5, 100))
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (ws.q, ws.Z, ws.gen, ws.gen_ll, ws.log_w, ws.idx, X)])
    for name, a, b in zip(("q", "Z", "gen", "gen_ll", "log_w", "idx", "X"), *outs):
        if name in ("q", "Z", "gen"):
            assert torch.equal(a, b), f"{name} differs between the kernels: {float((a.double() - b.double()).abs().max())}"
        elif name in ("gen_ll", "log_w"):
            # the masked log-likelihood is a sum over the row that the two kernels round in different places (one wave
            # accumulates it in registers, the other collects the terms of its waves): last-bit differences
            rel = float(((a.double() - b.double()).abs() / (b.double().abs() + 1.0)).max())
            assert rel < 5e-7, f"{name}: {rel}"
        else:
            # indices and imputations follow the log-weights: equal unless a uniform sits within that last bit of a CDF step
            same = (outs[0][5] == outs[1][5])
            assert same.float().mean() > 0.999
            if name == "X":
                ok = same[:, :K].unsqueeze(-1).expand_as(a)
                assert torch.equal(a[ok], b[ok])
    assert torch.isfinite(outs[0][6]).all()


def test_acmwg_sampler_is_bit_identical_with_either_kernel(narrow):
    """AC-MWG (history gathers through in_off, prior-mixture overrides of the proposal) with the counter RNG."""
    from oracle import cs_oracle as O
    from tests.helpers import make_model, seeded_inputs
    from tests.test_gpu_samplers import BASE_KW
    from vae_conditional_sampling_amd.sampling import test_step_vae_sampling as TS
    B, K, D, Zd, H = 48, 5, 43, 32, 256
    model = make_model(O.make_resfc_weights(D, 2 * Zd, 5, H, 51), O.make_resfc_weights(Zd, 2 * D, 2, H, 52))
    X0, M = seeded_inputs(B, K, D, 53)
    res = []
    for mode in (0, 1):
        narrow(mode)
        s = TS.TestVAEAdaptiveCollapsedMetropolisWithinGibbs(
            num_pseudo_warmup_iterations=2, num_mwg_iterations=12, prior_mixture_probability=0.3, warmup_method="irwg",
            num_k_per_cluster=4, glr_cluster_size=5, num_prior_replenish_proposals=5, num_copies=K, **BASE_KW)
        s.set_model(model)
        s.on_test_start()
        s.rng_seed = 77
        outs = [o["X"].clone() for o in s.sampler(X0.cuda().clone(), M.cuda())]
        res.append(torch.stack(outs))
    # accept decisions compare log-likelihoods that agree to the last bit or two: the chains coincide unless a uniform
    # falls into that gap (none does here)
    assert float((res[0] == res[1]).float().mean()) > 0.999
