"""CPU: the oracle restatement reproduces the golden vectors recorded from the reference.

The fixtures under tests/golden/ were written by oracle/check_against_reference.py, which ran the
reference package itself with a recorded draw tape (there the match is bitwise).  Here the same
draws are replayed on whatever host runs the tests, so floats are compared to 1e-5 relative
(torch's vectorised exp/log may differ by an ulp between CPU models) and indices exactly.
"""
import numpy as np
import pytest
import torch

from oracle import cs_oracle as O
from oracle import cs_oracle_samplers as S
from tests.helpers import assert_close, load_golden, weights_from_fixture

LAIR_CASES = ["lair_gr_mult_small", "lair_gr_mult_i2_lat", "lair_gr_mult_noreplenish", "lair_gr_syst",
              "lair_glr_groups", "lair_miniboone_shape", "lair_tempered_annealed", "lair_annealed_025_normal",
              "lair_tempered", "lair_lr_mult", "lair_lr_syst", "lair_ir_mult", "lair_ir_mult_i6", "lair_ir_syst",
              "lair_layernorm_nets", "lair_batchnorm_nets",
              # Student-t and logit-normal likelihoods (irwg/models/vae.py:220-229, 244-247)
              "lair_gen_studentt", "lair_gen_studentt_eps22", "lair_gen_logitnormal"]


@pytest.mark.parametrize("name", LAIR_CASES)
def test_oracle_lair_iteration(name):
    fx = load_golden(name)
    B, K, R, I, D, Zd, H, cluster, iters = [int(v) for v in fx["cfg"]]
    enc, dec = weights_from_fixture(fx, "enc"), weights_from_fixture(fx, "dec")
    X = torch.from_numpy(fx["X0"]); M = torch.from_numpy(fx["M"])
    want_lat = bool(fx["resampled_latents"])
    for t in range(iters):
        draws = {k: torch.from_numpy(fx[f"t{t}_{k}"]) for k in ("eps_z", "u_res", "eps_x")}
        if want_lat:
            draws["u_res2"] = torch.from_numpy(fx[f"t{t}_u_res2"])
        if f"t{t}_chi2_x" in fx:
            draws["chi2_x"] = torch.from_numpy(fx[f"t{t}_chi2_x"])
        o = O.lair_iteration(enc, dec, X, M, draws, num_imp_samples=I, weighting=str(fx["weighting"]),
                             gen_dist=str(fx["gen_dist"]) if "gen_dist" in fx else "normal_with_eps",
                             resampling_scheme=str(fx["scheme"]), resampling_method=str(fx["method"]),
                             glr_cluster_size=cluster or None, num_prior_replenish_proposals=R,
                             return_resampled_latents=want_lat, latent_dist=str(fx["latent_dist"]),
                             var_proposal_temperature=float(fx["temperature"]) or None,
                             var_proposal_anneal_type_to_prior=str(fx["anneal"]) or None)
        assert_close(o["log_weights"], fx[f"t{t}_log_weights"], "log_weights")
        assert np.array_equal(o["idx_global"].numpy(), fx[f"t{t}_idx"])
        assert_close(o["X_new"], fx[f"t{t}_X_new"], "X_new")
        assert_close(o["norm_log_weights"], fx[f"t{t}_norm_log_weights"], "norm_log_weights")
        assert_close(o["ess"], fx[f"t{t}_ess"], "ess", 1e-4)
        if want_lat:
            assert_close(o["latents_resampled"], fx[f"t{t}_latents_resampled"], "latents")
        X = o["X_new"]


MWG_CASES = ["acmwg_miniboone_config", "acmwg_tempered_annealed", "mwg_tempered_annealed", "acmwg_irwg_warmup", "acmwg_pg_warmup", "acmwg_nowarmup_noprior", "acmwg_ablation_mwg_target",
             "acmwg_ablation_pg_warmup", "mwg_pg_warmup_clip", "mwg_nowarmup", "mwg_irwg_warmup",
             "acmwg_gen_studentt", "acmwg_gen_logitnormal"]


@pytest.mark.parametrize("name", MWG_CASES)
def test_oracle_acmwg_sampler(name):
    fx = load_golden(name)
    B, K, D, Zd, H, W, T, nkpc, cluster, R = [int(v) for v in fx["cfg"]]
    enc, dec = weights_from_fixture(fx, "enc"), weights_from_fixture(fx, "dec")
    src = S.TapeSource(S.tape_from_npz_dict(fx))
    outs = list(S.acmwg_sampler(enc, dec, torch.from_numpy(fx["X0"]), torch.from_numpy(fx["M"]), src,
                                num_pseudo_warmup_iterations=W, num_mwg_iterations=T,
                                prior_mixture_probability=float(fx["p"]), warmup_method=str(fx["warmup"]),
                                num_k_per_cluster=nkpc or None, glr_cluster_size=cluster or None,
                                num_prior_replenish_proposals=R, plain_mwg=bool(fx["plain_mwg"]),
                                ablation_use_mwg_target=bool(fx["ablation"]),
                                clip_imputations_during_pg_warmup=bool(fx["clip"]),
                                var_proposal_temperature=float(fx["temperature"]) or None,
                                var_proposal_anneal_type_to_prior=str(fx["anneal"]) or None,
                                gen_dist=str(fx["gen_dist"]) if "gen_dist" in fx else "normal_with_eps"))
    assert src.pos == len(src.entries)
    mwg = [o for o in outs if "accepted" in o]
    assert np.array_equal(np.stack([o["accepted"].numpy() for o in mwg]), fx["accepted"])
    if "slots" in fx:
        assert np.array_equal(np.stack([o["slots"].numpy() for o in mwg]), fx["slots"])
    for t, o in enumerate(outs):
        assert_close(o["X"], fx["X_seq"][t], f"X[{t}]")


@pytest.mark.parametrize("name", ["pseudo_gibbs_clip", "pseudo_gibbs_noclip"])
def test_oracle_pseudo_gibbs_sampler(name):
    fx = load_golden(name)
    B, K, D, Zd, H, T = [int(v) for v in fx["cfg"]]
    enc, dec = weights_from_fixture(fx, "enc"), weights_from_fixture(fx, "dec")
    src = S.TapeSource(S.tape_from_npz_dict(fx))
    outs = list(S.pseudo_gibbs_sampler(enc, dec, torch.from_numpy(fx["X0"]), torch.from_numpy(fx["M"]), src,
                                       num_iterations=T, clip_imputations=bool(fx["clip"])))
    assert src.pos == len(src.entries) and len(outs) == T
    for t, o in enumerate(outs):
        assert_close(o["X"], fx["X_seq"][t], f"X[{t}]")
        assert_close(o["Z"], fx["Z_seq"][t], f"Z[{t}]")


@pytest.mark.parametrize("name", ["lair_sampler_final", "lair_sampler_final_n5"])
def test_oracle_lair_sampler_final(name):
    fx = load_golden(name)
    B, K, R, D, Zd, H, T, n_final, freq = [int(v) for v in fx["cfg"]]
    enc, dec = weights_from_fixture(fx, "enc"), weights_from_fixture(fx, "dec")
    src = S.TapeSource(S.tape_from_npz_dict(fx))
    M = torch.from_numpy(fx["M"])
    outs = list(S.lair_sampler(enc, dec, torch.from_numpy(fx["X0"]), M, src, num_iterations=T,
                               num_prior_replenish_proposals=R, prior_replenish_iteration_frequency=freq,
                               store_latents=True, collect_proposals=True))
    for t, o in enumerate(outs):
        assert_close(o["X"], fx["X_seq"][t], f"X[{t}]")
    idx, Zf, imps = S.lair_final_resample(dec, [o["Z_prop"] for o in outs], [o["log_weights"] for o in outs],
                                          torch.from_numpy(fx["X_true"]), M, src, num_samples=n_final or None)
    assert np.array_equal(idx.numpy(), fx["final_idx"])
    assert_close(imps.unsqueeze(0), fx["final_imputations"], "final imputations")


@pytest.mark.parametrize("name", ["ir_sampler_gr", "ir_sampler_ir_smis", "ir_sampler_lr"])
def test_oracle_importance_resampling_sampler(name):
    fx = load_golden(name)
    B, K, I, D, Zd, H, T, cluster, n_final = [int(v) for v in fx["cfg"]]
    dec = weights_from_fixture(fx, "dec")
    src = S.TapeSource(S.tape_from_npz_dict(fx))
    M = torch.from_numpy(fx["M"])
    outs = list(S.ir_sampler(dec, torch.from_numpy(fx["X0"]), M, src, num_iterations=T, num_imp_samples=I,
                             weighting_scheme=str(fx["weighting"]), resampling_scheme=str(fx["scheme"]),
                             resampling_method=str(fx["method"]), glr_cluster_size=cluster or None))
    for t, o in enumerate(outs):
        assert_close(o["X"], fx["X_seq"][t], f"X[{t}]")
        assert abs(float(o["is_ess_standard"]) - float(fx["is_ess"][t])) < 1e-4 * float(fx["is_ess"][t])
    idx, Zf, imps = S.lair_final_resample(dec, [o["Z_prop"] for o in outs], [o["log_weights"] for o in outs],
                                          torch.from_numpy(fx["X_true"]), M, src, num_samples=n_final or None)
    assert src.pos == len(src.entries)
    assert np.array_equal(idx.numpy(), fx["final_idx"])
    assert_close(imps.unsqueeze(0), fx["final_imputations"], "final imputations")


@pytest.mark.parametrize("name", ["lair_sampler_history_gr", "lair_sampler_history_glr"])
def test_oracle_lair_sampler_with_history(name):
    fx = load_golden(name)
    B, K, R, D, Zd, H, T, n_final, freq = [int(v) for v in fx["cfg"]]
    pg, num_hist, no_hist, hist_len, cluster, final = [int(v) for v in fx["hist_cfg"]]
    enc, dec = weights_from_fixture(fx, "enc"), weights_from_fixture(fx, "dec")
    src = S.TapeSource(S.tape_from_npz_dict(fx))
    outs = list(S.lair_sampler(enc, dec, torch.from_numpy(fx["X0"]), torch.from_numpy(fx["M"]), src, num_iterations=T,
                               num_prior_replenish_proposals=R, store_latents=True,
                               resampling_scheme=str(fx["scheme"]), weighting_scheme=str(fx["weighting"]),
                               glr_cluster_size=cluster or None, num_pseudo_warmup_iterations=pg,
                               num_historical_proposals=num_hist, no_historic_proposals_iterations=no_hist,
                               imputation_history_length=hist_len))
    assert src.pos == len(src.entries) and len(outs) == T + pg
    for t, o in enumerate(outs):
        assert_close(o["X"], fx["X_seq"][t], f"X[{t}]")
        assert_close(o["Z"], fx[f"Z_t{t}"], f"Z[{t}]")


@pytest.mark.parametrize("name", ["mlp_layernorm", "mlp_layernorm_h256", "mlp_batchnorm"])
def test_oracle_mlp_with_block_normalisation(name):
    fx = load_golden(name)
    w = weights_from_fixture(fx, "net")
    assert_close(O.resfc_forward(w, torch.from_numpy(fx["x"])), fx["y"], "forward")


@pytest.mark.parametrize("net", ["gmm_like", "omni_like", "gmm_full", "omni_full"])
def test_oracle_resnet_forward(net):
    """ResNet encoder / decoder restatement against the reference's recorded outputs (*_full: the MNIST-GMM and
    Omniglot networks at their real sizes).  torch's CPU convolution may pick another algorithm on another host,
    hence a tolerance instead of the bitwise match of the generating run."""
    from oracle import cs_oracle_conv as OC
    from tests.helpers import conv_weights
    fx = load_golden(f"conv_forward_{net}")
    enc_w, dec_w = conv_weights(net, int(fx["seed"]))
    with torch.no_grad():
        assert_close(OC.resnet_encoder_forward(enc_w, torch.from_numpy(fx["x"])), fx["enc_out"], "encoder")
        assert_close(OC.resnet_decoder_forward(dec_w, torch.from_numpy(fx["z"])), fx["dec_out"], "decoder")
