"""Student-t and logit-normal likelihoods (irwg/models/vae.py:220-229, 244-247) beyond the golden LAIR iterations of
tests/test_gpu_lair.py: the counter-RNG draws have the right laws, the LAIR sampler class runs end to end with them,
and the entry points that do not implement them say so."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(gen_dist, D=12, Zd=5, H=64, seed=5):
    from oracle import cs_oracle as O
    from tests.helpers import make_model
    from vae_conditional_sampling_amd import _lib
    enc = O.make_resfc_weights(D, 2 * Zd, 1, H, seed)
    dec = O.make_resfc_weights(Zd, _lib.gen_nparams(gen_dist) * D, 1, H, seed + 1)
    if gen_dist.startswith("studentt"):
        dec["bf"][:D] += 5.0   # degrees of freedom around 5
    return make_model(enc, dec, gen_dist=gen_dist)


def test_student_t_counter_rng_draws_follow_the_law():
    """x = loc + scale * n * rsqrt(c / df) with counter-RNG n ~ N(0,1), c ~ chi2(df): quantiles of the standardised
    draws against scipy's Student-t."""
    from scipy import stats
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling import engine
    model = _model("studentt", D=8)
    rows, D, df = 200_000, 8, 6.5
    gen = torch.empty(rows, 3 * D, device="cuda")
    gen[:, :D] = 1.5          # loc
    gen[:, D:2 * D] = 0.7     # scale
    gen[:, 2 * D:] = df
    X = torch.empty(rows, 1, D, device="cuda")
    engine.sample_impute(model, gen, X, None, 1, rng=_lib.make_rng(123, 0, 0))
    y = ((X.reshape(-1) - 1.5) / 0.7).double().cpu().numpy()
    qs = np.array([0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99])
    assert np.allclose(np.quantile(y, qs), stats.t.ppf(qs, df), atol=0.02, rtol=0.02)


def test_logit_normal_draws_stay_inside_the_unit_interval():
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling import engine
    model = _model("logitnormal_with_eps", D=8)
    rows, D = 50_000, 8
    gen = torch.empty(rows, 2 * D, device="cuda")
    gen[:, :D] = 0.3
    gen[:, D:] = 40.0         # saturates the sigmoid on both sides
    X = torch.empty(rows, 1, D, device="cuda")
    engine.sample_impute(model, gen, X, None, 1, rng=_lib.make_rng(7, 0, 0))
    assert float(X.min()) >= np.finfo(np.float32).tiny and float(X.max()) <= 1.0 - np.finfo(np.float32).eps
    assert 0.45 < float((X > 0.5).float().mean()) < 0.56


@pytest.mark.parametrize("gen_dist", ["studentt", "studentt_with_eps22", "logitnormal_with_eps"])
def test_lair_sampler_class_runs_with_the_family(gen_dist):
    from tests.test_gpu_samplers import BASE_KW
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAELatentAdaptiveImportanceResampling
    model = _model(gen_dist)
    assert isinstance(model.generator_network.packed(), _lib.PackedProgram)   # these likelihoods run as layer programs
    B, K, D = 9, 4, 12
    g = torch.Generator().manual_seed(3)
    X = torch.randn(B, 1, D, generator=g)
    if gen_dist == "logitnormal_with_eps":
        X = torch.sigmoid(X)
    M = torch.rand(B, 1, D, generator=g) > 0.5
    M[:, :, 0] = True
    s = TestVAELatentAdaptiveImportanceResampling(
        num_iterations=6, num_imp_samples=1, weighting_scheme="dmis", resampling_scheme="gr", resampling_method="multinomial",
        num_prior_replenish_proposals=1, resample_final_imps=False, num_copies=K, **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    s.rng_seed = 17
    s.on_test_batch_start(None, 0)
    X0 = s.basic_imputation(X.repeat(1, K, 1).cuda(), M.cuda())
    outs = [d["X"].clone() for d in s.sampler(X0, M.cuda())]
    assert len(outs) == 6
    Mb = M.cuda().expand(B, K, D)
    for o in outs:
        assert torch.isfinite(o).all()
        assert torch.equal(o[Mb], X0[Mb])
    assert not torch.equal(outs[0], outs[-1])


@pytest.mark.parametrize("gen_dist", ["studentt", "logitnormal_with_eps"])
def test_acmwg_sampler_class_runs_with_the_family_on_the_counter_rng(gen_dist):
    from tests.test_gpu_samplers import BASE_KW
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAEAdaptiveCollapsedMetropolisWithinGibbs
    model = _model(gen_dist)
    s = TestVAEAdaptiveCollapsedMetropolisWithinGibbs(num_pseudo_warmup_iterations=0, num_mwg_iterations=6,
                                                      prior_mixture_probability=0.1, warmup_method="none", num_copies=3,
                                                      **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    s.rng_seed = 1
    g = torch.Generator().manual_seed(2)
    X = torch.randn(4, 3, 12, generator=g)
    if gen_dist == "logitnormal_with_eps":
        X = torch.sigmoid(X)
    M = torch.rand(4, 1, 12, generator=g) > 0.5
    outs = [d["X"].clone() for d in s.sampler(X.cuda(), M.cuda())]
    Mb = M.cuda().expand(4, 3, 12)
    assert len(outs) == 6 and all(torch.isfinite(o).all() for o in outs)
    assert all(torch.equal(o[Mb], X.cuda()[Mb]) for o in outs)
    assert not torch.equal(outs[0], outs[-1])
