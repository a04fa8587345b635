"""GPU parity tests of the convolutional ResNet path (layer programs through the C ABI) against
the golden vectors recorded from the reference and the CPU oracle.

The conv nets are ~20 layers deep with BatchNorm folded into the weights and a different summation
order from torch's CPU convolutions, so raw network outputs are compared at CONV_RTOL; quantities
of the sampling path keep the 1e-5 bar where they hold it (indices bit-exact).
"""
import numpy as np
import pytest
import torch

from tests.helpers import CONV_NETS, assert_close, conv_weights, load_golden, make_conv_model, rel_err

pytestmark = pytest.mark.gpu

from tests.helpers import arith_fixture  # noqa: E402
mlp_arith = arith_fixture()


@pytest.fixture(autouse=True, params=["small_blocks_for_few_images", "large_blocks_always", "passes_in_two_halves"])
def block_shapes(request):
    """Launches of a few images run in blocks of 32 output channels (so that they reach more CUs); every case here
    runs with that rule and with the large blocks that full-size launches use.  Passes with enough work run as two
    halves on two streams: the third setting divides every pass of 8 rows and more that way."""
    from vae_conditional_sampling_amd import _lib
    prev = _lib.lib().cs_debug_set_small_launch_steps(18 if request.param.startswith("small") else 0)
    prev2 = _lib.lib().cs_debug_set_prog_two_streams(-8 if request.param == "passes_in_two_halves" else 0)
    yield
    _lib.lib().cs_debug_set_small_launch_steps(prev)
    _lib.lib().cs_debug_set_prog_two_streams(prev2)

CONV_RTOL = 1e-5


@pytest.mark.parametrize("net", ["gmm_like", "omni_like", "gmm_full", "omni_full"])
def test_resnet_forward_matches_reference_golden(net):
    """*_full are the reference's MNIST-GMM / Omniglot networks at their real sizes (256 / 512 channels)."""
    fx = load_golden(f"conv_forward_{net}")
    enc_w, dec_w = conv_weights(net, int(fx["seed"]))
    model = make_conv_model(net, enc_w, dec_w, "normal", "normal" if net.startswith("gmm") else "bern")
    x, z = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["z"]).cuda()
    y = model.var_latent_network(x)
    assert y.shape == fx["enc_out"].shape
    assert_close(y, fx["enc_out"], "encoder output", CONV_RTOL)
    g = model.generator_network(z)
    assert g.shape == fx["dec_out"].shape
    assert_close(g, fx["dec_out"], "decoder output", CONV_RTOL)


def test_resnet_forward_many_rows_matches_oracle():
    """More rows than one 128-pixel tile boundary pattern, odd row count."""
    from oracle import cs_oracle_conv as OC
    enc_w, dec_w = conv_weights("gmm_like", 77)
    model = make_conv_model("gmm_like", enc_w, dec_w, "normal", "normal")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(37, 196, generator=g)
    z = torch.randn(37, 8, generator=g)
    with torch.no_grad():
        assert_close(model.var_latent_network(x.cuda()), OC.resnet_encoder_forward(enc_w, x), "encoder", CONV_RTOL)
        assert_close(model.generator_network(z.cuda()), OC.resnet_decoder_forward(dec_w, z), "decoder", CONV_RTOL)


def test_wide_mlp_program_matches_oracle():
    """Hidden width 512 (toy configuration) runs as a layer program of matrix-core GEMMs."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.models import ResidualFCNetwork
    for (i, o, nb, rows) in ((2, 2, 4, 14), (1, 4, 4, 70), (40, 64, 1, 300)):
        w = O.make_resfc_weights(i, o, nb, 512, seed=11 + i)
        net = ResidualFCNetwork(i, o, nb, 512).load_weight_dict(w).cuda().eval()
        x = torch.randn(rows, i, generator=torch.Generator().manual_seed(3))
        assert_close(net(x.cuda()), O.resfc_forward(w, x), "wide mlp forward")


@pytest.mark.parametrize("name", ["conv_lair_gmm_like", "conv_lair_gmm_full"])
def test_lair_iteration_conv_matches_reference_golden(name):
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    fx = load_golden(name)
    B, K, R, I, D, Zd, iters, seed = [int(v) for v in fx["cfg"]]
    net = str(fx["net"])
    enc_w, dec_w = conv_weights(net, seed)
    model = make_conv_model(net, enc_w, dec_w, "normal_with_eps", "normal_with_eps")
    X = torch.from_numpy(fx["X0"]).cuda()
    M = torch.from_numpy(fx["M"]).cuda()
    for t in range(iters):
        draws = {"eps_z": torch.from_numpy(fx[f"t{t}_eps_z"]), "u_res": torch.from_numpy(fx[f"t{t}_u_res"]),
                 "eps_x": torch.from_numpy(fx[f"t{t}_eps_x"])}
        out = importance_resampling_gibbs_iteration(model, X, M, I, glr_cluster_size=None, weighting="dmis",
                                                    resampling_method="multinomial", resampling_scheme="gr",
                                                    num_prior_replenish_proposals=R, draws=draws)
        ws = model._cs_workspaces[("lair", B, K, R, I, str(X.device))]
        assert_close(ws.q[:, :, 0], fx[f"t{t}_q_loc"], "q loc", CONV_RTOL)
        assert_close(ws.q[:, :, 1], fx[f"t{t}_q_scale"], "q scale", CONV_RTOL)
        assert_close(out[6], fx[f"t{t}_Z_imp"], "Z_imp", CONV_RTOL)
        assert_close(out[7], fx[f"t{t}_log_weights"], "log_weights", CONV_RTOL)
        assert np.array_equal(ws.idx.cpu().numpy(), fx[f"t{t}_idx"]), "resampling indices differ"
        assert_close(out[0], fx[f"t{t}_X_new"], "X_new", CONV_RTOL)
        X = out[0].clone()  # the engine's own state carries over, as in a run


@pytest.mark.parametrize("name", ["conv_acmwg_omni_like", "conv_acmwg_omni_full"])
def test_acmwg_sampler_conv_bernoulli_matches_reference_golden(name):
    """Omniglot-like / full-size Omniglot AC-MWG (ResNet nets, Bernoulli likelihood, irwg warm-up): the recorded draw tape
    of the reference's sampler class is replayed into the HIP engine.  Binary imputations, accept
    decisions and history slots are compared bit for bit."""
    from tests.test_gpu_samplers import BASE_KW, tape_source
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import \
        TestVAEAdaptiveCollapsedMetropolisWithinGibbs as Sampler
    fx = load_golden(name)
    B, K, D, Zd, W, T, nkpc, cluster, R, seed = [int(v) for v in fx["cfg"]]
    net = str(fx["net"])
    enc_w, dec_w = conv_weights(net, seed)
    model = make_conv_model(net, enc_w, dec_w, "normal", "bern")
    s = Sampler(num_pseudo_warmup_iterations=W, num_mwg_iterations=T, prior_mixture_probability=float(fx["p"]),
                warmup_method="irwg", num_copies=K, num_k_per_cluster=nkpc, glr_cluster_size=cluster,
                num_prior_replenish_proposals=R, **BASE_KW)
    s.set_model(model)
    s.on_test_start()
    s.draw_source = tape_source(fx)
    X = torch.from_numpy(fx["X0"]).cuda()
    M = torch.from_numpy(fx["M"]).cuda()
    outs = []
    for o in s.sampler(X, M):
        outs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
    assert len(outs) == s.max_iterations == fx["X_seq"].shape[0]
    assert s.draw_source.pos == len(s.draw_source.entries), "the engine did not consume the whole tape"
    mwg = [o for o in outs if "_accepted" in o]
    acc = np.stack([o["_accepted"].cpu().numpy().astype(bool) for o in mwg])
    assert np.array_equal(acc, fx["accepted"]), "accept decisions differ"
    slots = np.stack([o["_slots"].cpu().numpy() for o in mwg])
    assert np.array_equal(slots, fx["slots"]), "history slots differ"
    for t, o in enumerate(outs):
        assert np.array_equal(o["X"].cpu().numpy(), fx["X_seq"][t].astype(np.float32)), f"binary imputations differ at {t}"
    assert_close(outs[-1]["Z"], fx["Z_last"], "Z", CONV_RTOL)


def test_yaml_driver_end_to_end_conv_configs(tmp_path):
    """Omniglot AC-MWG and MNIST-GMM LAIR YAML configurations (full-size nets, random weights,
    synthetic data, a few iterations) through the driver."""
    import os
    from tests.helpers import ROOT
    from vae_conditional_sampling_amd import driver
    cfg = driver.load_config(os.path.join(ROOT, "configs/omniglot28x28_bin/q50/acmwg_k5_005prior_irwg_warmup.yaml"),
                             ["data.synthetic_num_test=12", "data.batch_size=8", f"trainer.default_root_dir={tmp_path}",
                              "test_class.init_args.num_pseudo_warmup_iterations=2",
                              "test_class.init_args.num_mwg_iterations=5", "test_class.init_args.store_last_n_iterations=-1"])
    tester, outs = driver.run(cfg)
    g0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    assert g0["imputations"].shape == (1 + 2 * 4 + 5, 8, 5, 784)
    assert set(np.unique(g0["imputations"]).tolist()) <= {0.0, 1.0}
    obs = np.broadcast_to(g0["masks"], (8, 5, 784))
    for t in range(g0["imputations"].shape[0]):
        assert np.array_equal(g0["imputations"][t][obs], np.broadcast_to(g0["true_X"], (8, 5, 784))[obs])
    assert all(0.0 <= a <= 1.0 for a in outs[0]["acceptance"])

    cfg = driver.load_config(os.path.join(ROOT, "configs/mnist_gmm/samples/lair_k4_dmis_gr_mult_replenish1_finalresample.yaml"),
                             [f"trainer.default_root_dir={tmp_path}", "test_class.init_args.num_iterations=4"])
    tester, outs = driver.run(cfg)
    f0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    assert f0["imputations"].shape == (5, 10, 4, 196) and np.isfinite(f0["imputations"]).all()
    f1 = np.load(os.path.join(tester.out_dir, "irwg_imputations_after_final_resampling_0.npz"))
    assert f1["imputations"].shape == (1, 10, 4 * 5, 196) and np.isfinite(f1["imputations"]).all()


def test_row_chunking_of_the_workspace_gives_identical_results():
    """Layer programs process rows in chunks bounded by a workspace budget; a budget of a few rows
    must reproduce the single-pass results bit for bit (row offsets of every stage / finish kernel)."""
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    fx = load_golden("conv_lair_gmm_like")
    B, K, R, I, D, Zd, iters, seed = [int(v) for v in fx["cfg"]]
    enc_w, dec_w = conv_weights(str(fx["net"]), seed)
    X = torch.from_numpy(fx["X0"]).cuda()
    M = torch.from_numpy(fx["M"]).cuda()
    outs = []
    try:
        for budget in (0, 600_000):  # default; ~4 rows per pass of these nets
            _lib.check(_lib.lib().cs_debug_set_workspace_budget(budget))
            model = make_conv_model(str(fx["net"]), enc_w, dec_w, "normal_with_eps", "normal_with_eps")
            o = importance_resampling_gibbs_iteration(model, X, M, I, num_prior_replenish_proposals=R,
                                                      rng=_lib.make_rng(5, 1, 0))
            outs.append([o[0].clone(), o[6].clone(), o[7].clone()])
    finally:
        _lib.check(_lib.lib().cs_debug_set_workspace_budget(0))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_multi_channel_images_take_the_channel_major_path():
    """3-channel inputs / outputs (not in the reference's configurations, allowed by its classes)."""
    from oracle import cs_oracle_conv as OC
    from vae_conditional_sampling_amd.models.vae_resnet import ResNetDecoder, ResNetEncoder
    ecfg = dict(input_shape=(3, 8, 8), layers=[1, 1], layer_widths=[32, 64], latent_dim=12)
    dcfg = dict(output_shape=(3, 8, 8), layers=[1, 1], layer_upscale=[None, None], layer_widths=[32, 32],
                downscaled_resolution=(4, 4), latent_dim=6, upscale_factor=2)
    enc_w = OC.make_resnet_encoder_weights(seed=3, **ecfg)
    dec_w = OC.make_resnet_decoder_weights(seed=4, **dcfg)
    enc, dec = ResNetEncoder(**ecfg), ResNetDecoder(**dcfg)
    enc.load_state_dict(OC.state_dict_of(enc_w), strict=True)
    dec.load_state_dict(OC.state_dict_of(dec_w), strict=True)
    enc.out_split = dec.out_split = 1
    g = torch.Generator().manual_seed(1)
    x, z = torch.randn(5, 192, generator=g), torch.randn(5, 6, generator=g)
    assert_close(enc.cuda().eval()(x.cuda()), OC.resnet_encoder_forward(enc_w, x), "encoder", CONV_RTOL)
    assert_close(dec.cuda().eval()(z.cuda()), OC.resnet_decoder_forward(dec_w, z), "decoder", CONV_RTOL)


def test_repeated_conv_forwards_are_bit_identical():
    """Same networks, same input, launch after launch (under both arithmetics via the module fixture): a race or
    a missed wait between the last matrix instruction of a convolution kernel and its epilogue would show here."""
    enc_w, dec_w = conv_weights("omni_like", 31)
    model = make_conv_model("omni_like", enc_w, dec_w, "normal", "bern")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1024, 784, generator=g).cuda()
    z = torch.randn(1024, 10, generator=g).cuda()
    ref_e, ref_d = model.var_latent_network(x).clone(), model.generator_network(z).clone()
    for rep in range(12):
        e, d = model.var_latent_network(x), model.generator_network(z)
        assert torch.equal(e, ref_e), f"encoder forward {rep} differs in {int((e != ref_e).sum())} entries"
        assert torch.equal(d, ref_d), f"decoder forward {rep} differs in {int((d != ref_d).sum())} entries"


def _single_conv_program(cin, cout, hw, *, upsample=1, res=False, seed=0):
    """One 3x3 convolution (bias, ReLU, optionally a residual input that is the layer's own input) as a layer program."""
    from vae_conditional_sampling_amd import _lib
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g) * 0.1
    out_hw = hw * upsample
    b = _lib.ProgramBuilder(cin * hw * hw, cout * out_hw * out_hw, 1)
    b.conv(0, _lib.CS_NET_BUF_OUT, w, bias, hin=hw, win=hw, upsample=upsample, relu=True, res=0 if res else -1)
    return b.build(), w, bias


@pytest.mark.parametrize("case", ["same_size_256ch_7x7", "upsampled_512ch_4x4"])
def test_launches_divided_over_the_images_match_single_launches(case):
    """A convolution whose blocks would leave the last round of a 256-CU launch nearly empty runs as two launches
    (the first images in large blocks, the rest in small ones).  The image counts below trigger that on 256 CUs;
    every image must come out as when it is computed in a call of a few images (no division), bit for bit, and
    agree with torch's convolution of the same layer."""
    if mlp_arith_is_f32():
        pytest.skip("the exact-f32 kernels launch once")
    if case == "same_size_256ch_7x7":
        cin, cout, hw, up, rows = 256, 256, 7, 1, 674   # 130 x 2 blocks of 256 pixels x 128 channels: 1 round + 4
    else:
        cin, cout, hw, up, rows = 512, 256, 4, 2, 528   # 33 x 2 x 4 phase blocks: 1 round + 8
    net, w, bias = _single_conv_program(cin, cout, hw, upsample=up, res=(up == 1), seed=5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(rows, cin * hw * hw, generator=g)
    xd = x.cuda()
    y = net.forward(xd).clone()
    for lo, hi in ((0, 6), (rows // 2, rows // 2 + 5), (rows - 9, rows)):
        part = net.forward(xd[lo:hi].contiguous())
        assert torch.equal(part, y[lo:hi]), f"images {lo}..{hi} differ from their own launch"
    # NHWC activations: the program's rows are [pixel][channel]
    sel = torch.cat([torch.arange(0, 4), torch.arange(rows - 4, rows)])
    xi = x[sel].reshape(len(sel), hw, hw, cin).permute(0, 3, 1, 2).double()
    if up == 2:
        xi = torch.nn.functional.interpolate(xi, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xi, w.double(), bias.double(), padding=1)
    if up == 1:
        ref = ref + x[sel].reshape(len(sel), hw, hw, cin).permute(0, 3, 1, 2).double()
    ref = torch.relu(ref).permute(0, 2, 3, 1).reshape(len(sel), -1).float()
    assert_close(y[sel.cuda()], ref, "divided launch against torch", CONV_RTOL)


def mlp_arith_is_f32():
    from vae_conditional_sampling_amd import _lib
    return _lib.get_mlp_arith() == "f32"
