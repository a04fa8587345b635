"""Pin the ResNet part of the oracle against the reference and write its golden vectors.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).

    python oracle/check_conv_against_reference.py

1. Network forwards: the reference's ResNetEncoder / ResNetDecoder (irwg/models/vae_resnet.py)
   are instantiated in eval mode, loaded (strict) from the oracle's weight dicts -- which also
   pins the state-dict key layout -- and compared bitwise with oracle/cs_oracle_conv.py.
2. LAIR iteration (normal likelihood, MNIST-GMM-like nets) and AC-MWG steps (Bernoulli
   likelihood, Omniglot-like nets) of the reference with recorded draws; the oracle must
   reproduce them; the reference's outputs become tests/golden/conv_*.npz.  Weights are not
   stored: tests regenerate them from the seed with the oracle's generator.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import cs_oracle as O  # noqa: E402
from oracle import cs_oracle_conv as OC  # noqa: E402
from oracle import cs_oracle_samplers as OS  # noqa: E402
from oracle.ref_harness import DrawTape, load_reference  # noqa: E402
from oracle.check_against_reference import close, make_inputs  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# name: (encoder args, decoder args); shapes chosen to cover every kernel variant (32/64/128-wide
# channel tiles, stride 2, 1x1 shortcuts, x2 resize-convolutions, size resizes, 1 and 2 output
# channels) at fixture-friendly sizes
NETS = {
    "gmm_like": dict(enc=dict(input_shape=(1, 14, 14), layers=[2, 2, 2, 2], layer_widths=[32, 64, 128, 128], latent_dim=2 * 8),
                     dec=dict(output_shape=(2, 14, 14), layers=[2, 2, 2, 2], layer_upscale=[4, 7, None, None],
                              layer_widths=[64, 32, 32, 32], downscaled_resolution=(1, 1), latent_dim=8, upscale_factor=4)),
    "omni_like": dict(enc=dict(input_shape=(1, 28, 28), layers=[1, 1, 1, 1], layer_widths=[32, 32, 64, 64], latent_dim=2 * 10),
                      dec=dict(output_shape=(1, 28, 28), layers=[1, 1, 1, 1], layer_upscale=[7, None, None, None],
                               layer_widths=[64, 32, 32, 32], downscaled_resolution=(4, 4), latent_dim=10, upscale_factor=7)),
    # the networks of the reference's two image configurations at their real sizes
    # (configs/mnist_gmm/vae_convresnet3.yaml, configs/omniglot28x28_bin/vae_convresnet2.yaml)
    "gmm_full": dict(enc=dict(input_shape=(1, 14, 14), layers=[2, 2, 2, 2], layer_widths=[32, 64, 128, 256], latent_dim=2 * 25),
                     dec=dict(output_shape=(2, 14, 14), layers=[2, 2, 2, 2], layer_upscale=[4, 7, None, None],
                              layer_widths=[128, 64, 32, 32], downscaled_resolution=(1, 1), latent_dim=25, upscale_factor=4)),
    "omni_full": dict(enc=dict(input_shape=(1, 28, 28), layers=[2, 2, 2, 2], layer_widths=[64, 128, 256, 512], latent_dim=2 * 50),
                      dec=dict(output_shape=(1, 28, 28), layers=[2, 2, 2, 2], layer_upscale=[7, None, None, None],
                               layer_widths=[256, 128, 64, 64], downscaled_resolution=(4, 4), latent_dim=50, upscale_factor=7)),
}
ONLY = []   # command-line arguments: substrings of the fixture names to (re)generate; empty = all


def wanted(name):
    return not ONLY or any(o in name for o in ONLY)


def make_nets(name, seed):
    n = NETS[name]
    return (OC.make_resnet_encoder_weights(seed=seed, **n["enc"]), OC.make_resnet_decoder_weights(seed=seed + 1, **n["dec"]))


def ref_nets(ns, name, enc_w, dec_w):
    n = NETS[name]
    e = ns.vae_resnet.ResNetEncoder(**n["enc"]).eval()
    d = ns.vae_resnet.ResNetDecoder(**n["dec"]).eval()
    e.load_state_dict(OC.state_dict_of(enc_w), strict=True)
    d.load_state_dict(OC.state_dict_of(dec_w), strict=True)
    return e, d


def check_forwards(ns):
    for name in NETS:
        if not wanted(f"conv_forward_{name}"):
            continue
        enc_w, dec_w = make_nets(name, 100)
        e, d = ref_nets(ns, name, enc_w, dec_w)
        g = torch.Generator().manual_seed(5)
        D = int(np.prod(NETS[name]["enc"]["input_shape"]))
        x = torch.randn(3, 2, D, generator=g)
        z = torch.randn(2, 3, 2, NETS[name]["dec"]["latent_dim"], generator=g)
        with torch.no_grad():
            assert torch.equal(e(x), OC.resnet_encoder_forward(enc_w, x)), f"{name}: encoder differs"
            assert torch.equal(d(z), OC.resnet_decoder_forward(dec_w, z)), f"{name}: decoder differs"
            np.savez_compressed(os.path.join(GOLDEN, f"conv_forward_{name}.npz"), seed=np.int64(100), x=x.numpy(),
                                z=z.numpy(), enc_out=e(x).numpy(), dec_out=d(z).numpy())
        print(f"[ok] forward {name}")


def ref_vae(ns, name, enc_w, dec_w, latent_dist, gen_dist):
    e, d = ref_nets(ns, name, enc_w, dec_w)
    Dn = ns.vae.DISTRIBUTION
    return ns.vae.VAE(d, Dn[gen_dist], e, Dn[latent_dist], num_latent_samples=1, lr_generator=1e-3, lr_latent=1e-3).eval()


def case_lair_conv(ns, name, net, *, B, K, R, I, seed, iters=2):
    if not wanted(name):
        return
    enc_w, dec_w = make_nets(net, seed)
    vae = ref_vae(ns, net, enc_w, dec_w, "normal_with_eps", "normal_with_eps")
    D = int(np.prod(NETS[net]["enc"]["input_shape"]))
    X, M = make_inputs(B, K, D, seed + 2)
    fx = dict(X0=X.numpy(), M=M.numpy(), cfg=np.array([B, K, R, I, D, NETS[net]["dec"]["latent_dim"], iters, seed], dtype=np.int64),
              net=np.array(net))
    tape = DrawTape(seed + 3)
    Xr = X.clone()
    for t in range(iters):
        n0 = len(tape.entries)
        with torch.no_grad(), tape.recording():
            ret = ns.vae_imputation.importance_resampling_gibbs_iteration(
                vae, Xr, M, I, glr_cluster_size=None, weighting="dmis", resampling_method="multinomial",
                resampling_scheme="gr", num_prior_replenish_proposals=R)
        ent = tape.entries[n0:]
        assert [k for k, _ in ent] == ["normal", "multinomial", "normal"]
        draws = {"eps_z": torch.from_numpy(ent[0][1]), "u_res": torch.from_numpy(ent[1][1]), "eps_x": torch.from_numpy(ent[2][1])}
        with torch.no_grad():
            o = O.lair_iteration(enc_w, dec_w, Xr, M, draws, num_imp_samples=I, num_prior_replenish_proposals=R)
        assert torch.equal(o["X_new"], ret[0]) and torch.equal(o["log_weights"], ret[7]), "oracle differs from the reference"
        close(o["ess"], ret[1], "ess")
        fx.update({f"t{t}_eps_z": ent[0][1], f"t{t}_u_res": ent[1][1], f"t{t}_eps_x": ent[2][1],
                   f"t{t}_X_new": ret[0].numpy(), f"t{t}_ess": ret[1].numpy(), f"t{t}_perplexity": ret[3].numpy(),
                   f"t{t}_norm_log_weights": ret[5].numpy(), f"t{t}_Z_imp": ret[6].numpy(),
                   f"t{t}_log_weights": ret[7].numpy(), f"t{t}_idx": o["idx_global"].numpy(),
                   f"t{t}_q_loc": o["q_loc"].numpy(), f"t{t}_q_scale": o["q_scale"].numpy(),
                   f"t{t}_gen_params": o["generator_params"].numpy()})
        Xr = ret[0]
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


def case_acmwg_conv(ns, name, net, *, B, K, seed, W, T, p, nkpc, cluster, R):
    from oracle.check_against_reference import BASE_KW
    if not wanted(name):
        return
    enc_w, dec_w = make_nets(net, seed)
    vae = ref_vae(ns, net, enc_w, dec_w, "normal", "bern")
    D = int(np.prod(NETS[net]["enc"]["input_shape"]))
    g = torch.Generator().manual_seed(seed + 2)
    X_true = (torch.rand(B, D, generator=g) < 0.2).float()
    M = torch.ones(B, 28, 28, dtype=torch.bool)
    for b in range(B):  # two of the four quadrants missing
        q = torch.randperm(4, generator=g)[:2]
        for qi in q.tolist():
            M[b, (qi // 2) * 14:(qi // 2) * 14 + 14, (qi % 2) * 14:(qi % 2) * 14 + 14] = False
    M = M.reshape(B, 1, D)
    X = X_true.unsqueeze(1).repeat(1, K, 1) * M
    kw = dict(num_pseudo_warmup_iterations=W, num_mwg_iterations=T, prior_mixture_probability=p, warmup_method="irwg",
              num_copies=K, num_k_per_cluster=nkpc, glr_cluster_size=cluster, num_prior_replenish_proposals=R, **BASE_KW)
    tester = ns.test_step.TestVAEAdaptiveCollapsedMetropolisWithinGibbs(**kw)
    tester.set_model(vae)
    tape = DrawTape(seed + 3)
    ref_out = []
    with torch.no_grad(), tape.recording():
        for o in tester.sampler(X.clone(), M):
            ref_out.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
    entries = OS.compact_tape(tape.entries)
    src = OS.TapeSource(entries)
    with torch.no_grad():
        ora = list(OS.acmwg_sampler(enc_w, dec_w, X.clone(), M, src, num_pseudo_warmup_iterations=W, num_mwg_iterations=T,
                                    prior_mixture_probability=p, warmup_method="irwg", num_k_per_cluster=nkpc,
                                    glr_cluster_size=cluster, num_prior_replenish_proposals=R, latent_dist="normal",
                                    gen_dist="bern"))
    assert src.pos == len(entries), "oracle did not consume the whole tape"
    assert len(ora) == len(ref_out)
    accs, slots = [], []
    for t, (a, r) in enumerate(zip(ora, ref_out)):
        assert torch.equal(a["X"], r["X"]), f"X[{t}] not bitwise equal"
        if "accepted" in a:
            accs.append(a["accepted"].numpy()); slots.append(a["slots"].numpy())
    fx = dict(OS.tape_to_npz_dict(entries))
    fx.update(X0=X.numpy(), M=M.numpy(), net=np.array(net),
              cfg=np.array([B, K, D, NETS[net]["dec"]["latent_dim"], W, T, nkpc, cluster, R, seed], dtype=np.int64),
              p=np.float64(p), X_seq=np.stack([r["X"].numpy() for r in ref_out]).astype(np.uint8),
              Z_last=ref_out[-1]["Z"].numpy(),
              acceptance=np.array([float(r["acceptance"]) for r in ref_out], dtype=np.float32),
              accepted=np.stack(accs), slots=np.stack(slots))
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}  (iterations {len(ref_out)}, mean acceptance {np.mean(accs):.2f})")


def main():
    ONLY[:] = sys.argv[1:]
    ns = load_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    check_forwards(ns)
    case_lair_conv(ns, "conv_lair_gmm_like", "gmm_like", B=3, K=2, R=1, I=1, seed=300)
    case_acmwg_conv(ns, "conv_acmwg_omni_like", "omni_like", B=2, K=5, seed=400, W=2, T=8, p=0.3, nkpc=4, cluster=5, R=5)
    case_lair_conv(ns, "conv_lair_gmm_full", "gmm_full", B=3, K=4, R=1, I=1, seed=500)
    case_acmwg_conv(ns, "conv_acmwg_omni_full", "omni_full", B=2, K=5, seed=600, W=1, T=6, p=0.2, nkpc=4, cluster=5, R=5)


if __name__ == "__main__":
    main()
