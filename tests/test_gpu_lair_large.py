"""GPU parity tests of LAIR with more than 64 proposals per example (K' = K + R in the hundreds or
thousands: configs/mnist_gmm/samples/vae_convresnet3_k100_irwg_i1_dmisg_glr5_mult.yaml and
configs/toy_grid2dmog4/sampling/vae_comptrain/irwg_k500_*.yaml, irwg_k5000_*.yaml of the reference), which take
the three-launch path of the library, against golden vectors recorded from the reference and against the CPU
oracle on the same injected draws.

Bar: log-weights, proposals and imputations within 1e-5 (float32); resampling indices bit-exact except where a
float64 uniform lies within 1e-5 of a step of the reference's CDF.  With S proposals and n draws per example an
index can only differ when a uniform falls between the two sides' CDF steps; the CDFs agree to ~1e-6 (they are
sums of exponentials of log-weights that agree to 1e-5 relative), so about n * S * 1e-6 draws per example are
expected to do so -- none at the sizes of the original fixtures (S <= 25), a handful at S = 500.  Every such draw is
checked to be one: the uniform within 1e-5 of the steps that separate the two choices.
"""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, check_indices, load_golden, make_model, weights_from_fixture

pytestmark = pytest.mark.gpu
DIAG_RTOL = 1e-4


LARGE_CASES = ["lair_k100_glr5_dmisg", "lair_k96_r4_gr_dmis_lat", "lair_k40_i3_gr_syst", "lair_k30_i4_lr_smis",
               "lair_k24_i5_ir", "lair_k120_glr10_dmis"]


@pytest.mark.parametrize("name", LARGE_CASES)
def test_large_proposal_sets_match_reference_golden(name):
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    fx = load_golden(name)
    B, K, R, I, D, Zd, H, cluster, iters = [int(v) for v in fx["cfg"]]
    assert I * (K + R) > 64
    model = make_model(weights_from_fixture(fx, "enc"), weights_from_fixture(fx, "dec"))
    scheme, weighting, method = str(fx["scheme"]), str(fx["weighting"]), str(fx["method"])
    want_lat = bool(fx["resampled_latents"])
    X = torch.from_numpy(fx["X0"]).cuda()
    M = torch.from_numpy(fx["M"]).cuda()
    for t in range(iters):
        draws = {"eps_z": torch.from_numpy(fx[f"t{t}_eps_z"]), "u_res": torch.from_numpy(fx[f"t{t}_u_res"]),
                 "eps_x": torch.from_numpy(fx[f"t{t}_eps_x"])}
        if want_lat:
            draws["u_res2"] = torch.from_numpy(fx[f"t{t}_u_res2"])
        out = importance_resampling_gibbs_iteration(
            model, X, M, I, glr_cluster_size=cluster or None, weighting=weighting, resampling_method=method,
            resampling_scheme=scheme, num_prior_replenish_proposals=R, return_resampled_latents=want_lat, draws=draws)
        ws = model._cs_workspaces[("lair", B, K, R, I, str(X.device))]
        assert_close(out[6], fx[f"t{t}_Z_imp"], "Z_imp")
        assert_close(out[7], fx[f"t{t}_log_weights"], "log_weights")
        assert_close(out[5], fx[f"t{t}_norm_log_weights"], "norm_log_weights")
        assert_close(out[1], fx[f"t{t}_ess"], "ess", DIAG_RTOL)
        assert_close(out[3], fx[f"t{t}_perplexity"], "perplexity", DIAG_RTOL)
        assert out[4] == I * (K + R)
        idx = ws.idx.cpu().numpy()
        if method == "multinomial" and scheme in ("gr", "glr"):
            same = check_indices(idx, fx[f"t{t}_idx"], fx[f"t{t}_log_weights"], fx[f"t{t}_u_res"], scheme, I, K + R,
                                 cluster, f"{name} t{t}")
        else:
            same = idx == fx[f"t{t}_idx"]
            assert same.all(), "resampling indices differ"
        ok = torch.from_numpy(same[:, :K]).cuda()
        assert_close(out[0][ok], fx[f"t{t}_X_new"][same[:, :K]], "X_new")
        if want_lat:
            assert_close(out[9], fx[f"t{t}_latents_resampled"], "latents_resampled")
        Mb = M.expand_as(X)
        assert torch.equal(out[0][Mb], X[Mb])
        X = torch.from_numpy(fx[f"t{t}_X_new"]).cuda()


@pytest.mark.parametrize("name", ["toy_lair_k500_dmis_gr", "toy_lair_k500_dmis_glr5", "toy_lair_k500_smis_glr5"])
def test_toy_configuration_with_500_chains_matches_reference_golden(name):
    """Beta posterior, uniform prior, K = 500 on the 14 toy test points (the reference's irwg_k500_* configurations)."""
    from tests.test_gpu_toy import toy_model
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    fx = load_golden(name)
    B, K, R, I, D, Zd, H, iters = [int(v) for v in fx["cfg"]]
    scheme, weighting, cluster = str(fx["scheme"]), str(fx["weighting"]), int(fx["cluster"])
    model = toy_model(fx, D, Zd, H)
    X = torch.from_numpy(fx["X0"]).cuda()
    M = torch.from_numpy(fx["M"]).cuda()
    for t in range(iters):
        draws = {"eps_z": torch.from_numpy(fx[f"t{t}_eps_z"]), "u_res": torch.from_numpy(fx[f"t{t}_u_res"]),
                 "eps_x": torch.from_numpy(fx[f"t{t}_eps_x"])}
        out = importance_resampling_gibbs_iteration(model, X, M, I, glr_cluster_size=cluster or None, weighting=weighting,
                                                    resampling_method="multinomial", resampling_scheme=scheme,
                                                    num_prior_replenish_proposals=R, draws=draws)
        ws = model._cs_workspaces[("lair", B, K, R, I, str(X.device))]
        assert torch.equal(out[6].cpu(), torch.from_numpy(fx[f"t{t}_Z_imp"])), "injected Beta samples must pass through"
        assert_close(out[7], fx[f"t{t}_log_weights"], "log_weights")
        assert_close(out[5], fx[f"t{t}_norm_log_weights"], "norm_log_weights")
        assert_close(out[1], fx[f"t{t}_ess"], "ess", DIAG_RTOL)
        assert_close(out[3], fx[f"t{t}_perplexity"], "perplexity", DIAG_RTOL)
        same = check_indices(ws.idx.cpu().numpy(), fx[f"t{t}_idx"], fx[f"t{t}_log_weights"], fx[f"t{t}_u_res"], scheme, I,
                             K + R, cluster, f"{name} t{t}")
        ok = torch.from_numpy(same[:, :K]).cuda()
        assert_close(out[0][ok], fx[f"t{t}_X_new"][same[:, :K]], "X_new")
        X = torch.from_numpy(fx[f"t{t}_X_new"]).cuda()


@pytest.mark.parametrize("cfg", [
    dict(B=3, K=5000, R=1, Zd=1, D=2, H=64, weighting="dmis"),                              # irwg_k5000_dmis_gr
    dict(B=3, K=5000, R=0, Zd=1, D=2, H=64, weighting="smis", scheme="glr", cluster=5),     # irwg_k5000_smis_glr5
    dict(B=5, K=100, R=0, Zd=25, D=43, H=128, weighting="dmis_within_groups", scheme="glr", cluster=5),   # MNIST-GMM latent size
    dict(B=4, K=700, R=4, Zd=32, D=43, H=256, weighting="dmis"),                            # MiniBooNE sizes, many chains
    dict(B=2, K=9000, R=0, Zd=2, D=3, H=64, weighting="smis"),                              # more than 32 KiB of CDF per example
])
def test_large_proposal_sets_match_oracle(cfg):
    """Sizes beyond the committed fixtures (up to the reference's K = 5000), normal posterior, against the CPU oracle."""
    from oracle import cs_oracle as O
    from tests.helpers import seeded_inputs
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    B, K, R, Zd, D, H = (cfg[k] for k in ("B", "K", "R", "Zd", "D", "H"))
    scheme, weighting, cluster = cfg.get("scheme", "gr"), cfg["weighting"], cfg.get("cluster")
    I, Kp = 1, K + R
    enc = O.make_resfc_weights(D, 2 * Zd, 1, H, 21)
    dec = O.make_resfc_weights(Zd, 2 * D, 1, H, 22)
    model = make_model(enc, dec)
    X, M = seeded_inputs(B, K, D, 23)
    g = torch.Generator().manual_seed(24)
    G = Kp // cluster if scheme == "glr" else 1
    draws = {"eps_z": torch.randn(I, B, Kp, Zd, generator=g), "eps_x": torch.randn(B, K, D, generator=g),
             "u_res": torch.rand(B * G, Kp // G, generator=g, dtype=torch.float64)}
    o = O.lair_iteration(enc, dec, X, M, draws, num_imp_samples=I, weighting=weighting, resampling_scheme=scheme,
                         glr_cluster_size=cluster, num_prior_replenish_proposals=R)
    out = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), I, glr_cluster_size=cluster, weighting=weighting,
                                                resampling_scheme=scheme, num_prior_replenish_proposals=R, draws=draws)
    ws = model._cs_workspaces[("lair", B, K, R, I, "cuda:0")]
    assert_close(out[7], o["log_weights"], "log_weights")
    # random networks give log-weights in the hundreds: the normalised ones carry their absolute error
    assert_close(out[5], o["norm_log_weights"], "norm_log_weights", floor=o["log_weights"].abs().amax(dim=(0, 2))[:, None])
    assert_close(out[1], o["ess"], "ess", DIAG_RTOL)
    assert_close(out[3], o["perplexity"], "perplexity", DIAG_RTOL)
    same = check_indices(ws.idx.cpu().numpy(), o["idx_global"].numpy(), o["log_weights"].numpy(), draws["u_res"].numpy(),
                         scheme, I, Kp, cluster, str(cfg))
    ok = torch.from_numpy(same[:, :K])
    assert_close(out[0].cpu()[ok], o["X_new"][ok], "X_new")


def test_counter_rng_results_do_not_depend_on_batching_with_large_sets():
    """Philox draws are keyed by the global example: a batch of 6 equals two batches of 3, bit for bit (K' = 201)."""
    from oracle import cs_oracle as O
    from tests.helpers import seeded_inputs
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    B, K, R, D, Zd = 6, 200, 1, 9, 4
    model = make_model(O.make_resfc_weights(D, 2 * Zd, 1, 64, 31), O.make_resfc_weights(Zd, 2 * D, 1, 64, 32))
    X, M = seeded_inputs(B, K, D, 33)
    X, M = X.cuda(), M.cuda()
    whole = importance_resampling_gibbs_iteration(model, X, M, 1, num_prior_replenish_proposals=R, rng=_lib.make_rng(5, 2, 0))
    halves = [importance_resampling_gibbs_iteration(model, X[h:h + 3], M[h:h + 3], 1, num_prior_replenish_proposals=R,
                                                    rng=_lib.make_rng(5, 2, h)) for h in (0, 3)]
    assert torch.equal(whole[0], torch.cat([h[0] for h in halves]))
    assert torch.equal(whole[7], torch.cat([h[7] for h in halves], dim=1))
    assert torch.isfinite(whole[0]).all()


@pytest.mark.parametrize("name", ["lair_sampler_history_gr", "lair_sampler_history_glr"])
def test_sampler_class_with_history_through_the_large_set_path(name):
    """Historical proposals (padded glr layout, chains that only propose) through the three-launch path."""
    from tests.test_gpu_samplers import test_lair_sampler_with_historical_proposals_matches_reference_golden as run
    from vae_conditional_sampling_amd import _lib
    prev = _lib.lib().cs_debug_force_large_resample(1)
    try:
        run(name)
    finally:
        _lib.lib().cs_debug_force_large_resample(prev)


def test_history_concat_with_more_than_64_historical_proposals():
    """cs_history_concat moves its slot table in chunks of 64 proposals: 150 of them against plain indexing."""
    import ctypes as C
    from vae_conditional_sampling_amd import _lib
    B, K, D, H, T = 5, 3, 7, 150, 40
    g = torch.Generator().manual_seed(3)
    X = torch.randn(B, K, D, generator=g).cuda()
    ring = torch.randn(T, B, K, D, generator=g).cuda()
    slots = torch.randint(0, T, (H,), generator=g, dtype=torch.int32)
    chains = torch.randint(0, K, (H,), generator=g, dtype=torch.int32)
    out = torch.empty(B, K + H, D, device="cuda")
    sl = (C.c_int32 * H)(*slots.tolist())
    ch = (C.c_int32 * H)(*chains.tolist())
    _lib.check(_lib.lib().cs_history_concat(_lib.ptr(X), _lib.ptr(ring), B, K, D, H, sl, ch, None, None, _lib.ptr(out),
                                            _lib.current_stream()))
    want = torch.cat([X, torch.stack([ring[int(s), :, int(c)] for s, c in zip(slots, chains)], dim=1)], dim=1)
    assert torch.equal(out, want)


@pytest.mark.parametrize("config,K,D,B,T,override", [
    ("configs/mnist_gmm/samples/lair_k100_dmisg_glr5_mult.yaml", 100, 196, 10, 3, ["test_class.init_args.num_iterations=3"]),
    ("configs/toy_grid2dmog4/sampling/lair_k500_dmis_gr_mult_it10.yaml", 500, 2, 1, 10, ["data.batch_size=1"]),
    ("configs/toy_grid2dmog4/sampling/lair_k5000_smis_glr5_mult_it3.yaml", 5000, 2, 14, 3, ["data.batch_size=14"]),
    ("configs/toy_grid2dmog4/sampling/lair_k5000_dmis_gr_mult_it3.yaml", 5000, 2, 14, 3, ["data.batch_size=14"]),
])
def test_yaml_driver_end_to_end_configs_with_many_chains(tmp_path, config, K, D, B, T, override):
    """The reference's LAIR configurations with 100 / 500 / 5000 chains (full-size networks, random weights, counter
    RNG) through the driver: artefact shapes, finite imputations, observed entries untouched."""
    import os
    from tests.helpers import ROOT
    from vae_conditional_sampling_amd import driver
    cfg = driver.load_config(os.path.join(ROOT, config), [f"trainer.default_root_dir={tmp_path}"] + override)
    tester, outs = driver.run(cfg)
    f0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    imps = f0["imputations"]
    assert imps.shape == (T + 1, B, K, D) and np.isfinite(imps).all()
    obs = np.broadcast_to(f0["masks"], (B, K, D))
    truth = np.broadcast_to(np.nan_to_num(f0["true_X"]), (B, K, D))
    for t in range(T + 1):
        assert np.array_equal(imps[t][obs], truth[obs])
    # the chains move: resampling and fresh imputations change the missing entries every iteration
    assert not np.array_equal(imps[1][~obs], imps[T][~obs])


@pytest.mark.parametrize("Zd,K,R,I", [(32, 96, 4, 1), (16, 130, 0, 1), (8, 70, 1, 2), (32, 300, 1, 1)])
def test_log_weight_kernels_of_the_large_path_agree_bit_for_bit(Zd, K, R, I):
    """The 16-byte-access log-weight kernel (latent in registers for Zd = 16 / 32, in LDS otherwise) computes the
    terms of the scalar kernel in the same order: identical log-weights, statistics and imputations."""
    from oracle import cs_oracle as O
    from tests.helpers import seeded_inputs
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    B, D = 7, 11
    model = make_model(O.make_resfc_weights(D, 2 * Zd, 1, 64, 41), O.make_resfc_weights(Zd, 2 * D, 1, 64, 42))
    X, M = seeded_inputs(B, K, D, 43)
    X, M = X.cuda(), M.cuda()

    def run():
        return importance_resampling_gibbs_iteration(model, X, M, I, num_prior_replenish_proposals=R,
                                                     rng=_lib.make_rng(9, 1, 0))
    vec = [t.clone() if torch.is_tensor(t) else t for t in run()]
    prev = _lib.lib().cs_debug_force_scalar_logw(1)
    try:
        ref = run()
    finally:
        _lib.lib().cs_debug_force_scalar_logw(prev)
    for i in (0, 1, 3, 5, 7):
        assert torch.equal(vec[i], ref[i]), i
    assert torch.isfinite(vec[7]).all()
