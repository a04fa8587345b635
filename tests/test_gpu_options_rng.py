"""GPU property tests of the sampler options on the built-in counter RNG (no injected draws):
lr / ir resampling, prior proposals, tempering / annealing, historical proposals, the plain
Metropolis-within-Gibbs and pseudo-Gibbs classes.  The parity of each option against the reference is
covered by the golden tests (injected draws); here the device-side draws run, and the invariants the
domain offers are checked: observed entries bit-identical, finite imputations, index ranges,
normalised weights, results independent of how examples are batched.
"""
import numpy as np
import pytest
import torch

from tests.helpers import make_model, seeded_inputs

pytestmark = pytest.mark.gpu

from tests.helpers import arith_fixture  # noqa: E402
mlp_arith = arith_fixture()

BASE_KW = dict(store_last_n_iterations=-1, store_every_n_iterations=1, imputation_metric_quantiles=[],
               compute_f1_metric=False, compute_inv_f1_metric=False, compute_bin_accuracy_metric=False,
               compute_rmse_metric=False, compute_ssim_metric=False, compute_latent_wass_metric=False,
               imputation_fn="vae_samples")


@pytest.fixture(scope="module")
def model():
    from oracle import cs_oracle as O
    return make_model(O.make_resfc_weights(9, 8, 1, 64, 5), O.make_resfc_weights(4, 18, 1, 64, 6))


def check_imputations(X_new, X0, M):
    Mb = M.expand_as(X0)
    assert torch.equal(X_new[Mb], X0[Mb]), "an observed entry changed"
    assert torch.isfinite(X_new).all()
    assert not torch.equal(X_new[~Mb], X0[~Mb]), "no missing entry was imputed"


@pytest.mark.parametrize("scheme,method,I", [("lr", "multinomial", 4), ("lr", "systematic", 3), ("ir", "multinomial", 5),
                                             ("ir", "systematic", 2), ("gr", "multinomial", 2)])
@pytest.mark.parametrize("prior_proposals", [False, True])
def test_single_draw_schemes_on_the_counter_rng(model, scheme, method, I, prior_proposals):
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.vae_imputation import (importance_resampling_gibbs_iteration,
                                                                        standard_importance_resampling)
    B, K, D = 301, 3, 9
    X, M = seeded_inputs(B, K, D, 21)
    X, M = X.cuda(), M.cuda()
    rng = _lib.make_rng(7, 3, 0)
    if prior_proposals:
        out = standard_importance_resampling(model, X, M, I, weighting="smis", resampling_scheme=scheme,
                                             resampling_method=method, rng=rng)
    else:
        out = importance_resampling_gibbs_iteration(model, X, M, I, weighting="dmis", resampling_scheme=scheme,
                                                    resampling_method=method, rng=rng)
    check_imputations(out[0], X, M)
    ws = model._cs_workspaces[("lair", B, K, 0, I, "cuda:0")]
    idx = ws.idx.cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < I * K
    if scheme == "lr":      # chain k draws among its own I candidates: flattened index i*K + k
        assert torch.equal(idx % K, torch.arange(K).expand(B, K))
    if scheme == "ir":      # slot k draws from the k-th run of I consecutive entries of the '(i k)' axis
        assert torch.equal(idx // I, torch.arange(K).expand(B, K))
    assert torch.isfinite(out[7]).all() and (out[1] >= 1 - 1e-4).all() and (out[1] <= I * K + 1e-3).all()
    # sharding invariance: the second half of the batch on its own, keyed by its global offset
    h = B // 2
    fn = standard_importance_resampling if prior_proposals else importance_resampling_gibbs_iteration
    kw = dict(weighting="smis" if prior_proposals else "dmis", resampling_scheme=scheme, resampling_method=method)
    out2 = fn(model, X[h:].contiguous(), M[h:].contiguous(), I, rng=_lib.make_rng(7, 3, h), **kw)
    assert torch.equal(out2[0], out[0][h:]), "results depend on the batching"


@pytest.mark.parametrize("temperature,anneal", [(1.5, None), (None, "linear"), (0.6, "linear_0.25")])
def test_tempering_and_annealing_transform(model, temperature, anneal):
    """cs_proposal_transform against the closed form it should be close to: scale*T, and the
    annealing interpolation towards N(0, 1); rows of replenishing proposals are left alone."""
    import ctypes as C
    from vae_conditional_sampling_amd import _lib
    B, K, R, Zd, D = 37, 3, 2, 4, 9
    g = torch.Generator().manual_seed(3)
    q = torch.empty(B, K + R, 2, Zd)
    q[:, :, 0] = torch.randn(B, K + R, Zd, generator=g)
    q[:, :, 1] = 0.05 + 2 * torch.rand(B, K + R, Zd, generator=g)
    M = torch.rand(B, D, generator=g) > 0.4
    qd, Md = q.cuda().contiguous(), M.to(torch.uint8).cuda().contiguous()
    t, a = _lib.proposal_transform_args(temperature, anneal)
    _lib.check(_lib.lib().cs_proposal_transform(_lib.ptr(qd), _lib.ptr(Md), B, K, K + R, Zd, D, _lib.CS_DIST["normal_with_eps"],
                                                0, t, a, _lib.current_stream()))
    out = qd.cpu()
    assert torch.equal(out[:, K:], q[:, K:])
    loc, sc = q[:, :K, 0].double(), q[:, :K, 1].double()
    if temperature:
        sc = sc * temperature
    if anneal:
        mf = ((~M).sum(-1).double() / D)[:, None, None]
        c = 0.25 if anneal == "linear_0.25" else 1.0
        sc = sc * ((1 - mf) + (1 / sc) * c * mf)
        loc = loc * (1 - mf)
    assert torch.allclose(out[:, :K, 0].double(), loc, rtol=1e-5, atol=1e-6)
    assert torch.allclose(out[:, :K, 1].double(), sc, rtol=2e-5, atol=1e-6)


def run_sampler(cls, model, X, M, seed=11, offset=0, **kw):
    s = cls(num_copies=X.shape[1], **kw, **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    s.rng_seed, s.example_offset = seed, offset
    s.on_test_batch_start(None, 0)
    return s, [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in s.sampler(X, M)]


def test_lair_with_history_on_the_counter_rng(model):
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAELatentAdaptiveImportanceResampling as S
    B, K, D = 130, 4, 9
    X, M = seeded_inputs(B, K, D, 31)
    X, M = X.cuda(), M.cuda()
    kw = dict(num_iterations=9, num_imp_samples=1, weighting_scheme="dmis", resampling_scheme="gr",
              resampling_method="multinomial", num_prior_replenish_proposals=1, num_pseudo_warmup_iterations=2,
              num_historical_proposals=3, no_historic_proposals_iterations=4, imputation_history_length=5)
    s, outs = run_sampler(S, model, X, M, **kw)
    assert len(outs) == 11
    for o in outs:
        check_imputations(o["X"], X, M)
    assert outs[-1]["total_num_props"] == K + 3 + 1 and outs[2]["total_num_props"] == K + 1
    # the chains of two consecutive iterations differ (the sampler moves), and a second run repeats the first
    assert not torch.equal(outs[-1]["X"], outs[-2]["X"])
    _, again = run_sampler(S, model, X, M, **kw)
    assert all(torch.equal(a["X"], b["X"]) for a, b in zip(outs, again))
    # global-local resampling with the history: padded groups first, per-proposal chains after
    kw.update(resampling_scheme="glr", weighting_scheme="dmis_within_groups", glr_cluster_size=4, num_prior_replenish_proposals=0,
              num_historical_proposals=4, no_historic_proposals_iterations=5, imputation_history_length=6, num_iterations=10)
    s, outs = run_sampler(S, model, X, M, **kw)
    for o in outs:
        check_imputations(o["X"], X, M)
    assert outs[-1]["total_num_props"] == K + 4 and outs[3]["total_num_props"] == K


@pytest.mark.parametrize("warmup", ["none", "pseudo_gibbs", "irwg"])
def test_metropolis_within_gibbs_on_the_counter_rng(model, warmup):
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAEMetropolisWithinGibbs as S
    B, K, D = 257, 4, 9
    X, M = seeded_inputs(B, K, D, 41)
    X, M = X.cuda(), M.cuda()
    kw = dict(num_pseudo_warmup_iterations=0 if warmup == "none" else 2, num_mwg_iterations=12, warmup_method=warmup,
              var_proposal_temperature=1.2, var_proposal_anneal_type_to_prior="linear")
    if warmup == "irwg":
        kw.update(num_k_per_cluster=3, glr_cluster_size=4, num_prior_replenish_proposals=4)
    s, outs = run_sampler(S, model, X, M, **kw)
    assert len(outs) == s.max_iterations
    mwg = [o for o in outs if "_accepted" in o]
    assert len(mwg) == 12
    for o in mwg:
        check_imputations(o["X"], X, M)
        assert torch.isfinite(o["_log_accept"]).all()
    acc = float(torch.stack([o["acceptance"] for o in mwg]).mean())
    assert 0.02 < acc < 0.98, f"implausible acceptance rate {acc}"
    # sharding invariance of the whole chain: the upper half of the batch alone, with its global offset
    h = B // 2 + 1
    _, part = run_sampler(S, model, X[h:].contiguous(), M[h:].contiguous(), offset=h, **kw)
    assert torch.equal(part[-1]["X"], outs[-1]["X"][h:])
    assert torch.equal(part[-1]["_accepted"], outs[-1]["_accepted"][h:])


def test_pseudo_gibbs_and_importance_resampling_classes_on_the_counter_rng(model):
    from vae_conditional_sampling_amd.sampling import test_step_vae_sampling as TS
    B, K, D = 65, 3, 9
    X, M = seeded_inputs(B, K, D, 51)
    X, M = X.cuda(), M.cuda()
    _, outs = run_sampler(TS.TestVAEPseudoGibbs, model, X, M, num_iterations=5, clip_imputations=True)
    for o in outs:
        check_imputations(o["X"], X, M)
    lim = (X[:, 0] * M[:, 0]).abs().max() * 2 + 1e-6
    assert float(outs[-1]["X"].abs().max()) <= float(lim)        # clipped to twice the observed range
    s, outs = run_sampler(TS.TestVAEImportanceResampling, model, X, M, num_iterations=4, num_imp_samples=3,
                          weighting_scheme="dmis", resampling_scheme="ir", resampling_method="multinomial")
    for o in outs:
        check_imputations(o["X"], X, M)
        assert 1 - 1e-4 <= float(o["is_ess_standard"]) <= 3 * K + 1e-3
    imps, Z, idx = s.final_resampling(X[:, 0], M)
    assert imps.shape == (1, B, 4 * K, D) and int(idx.max()) < 4 * 3 * K and int(idx.min()) >= 0
    Mb = M.expand(B, 4 * K, D)
    assert torch.equal(imps[0][Mb], X[:, :1].expand(B, 4 * K, D)[Mb])
