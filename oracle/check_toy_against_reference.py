"""Pin the Beta-posterior / uniform-prior part of the oracle (toy configuration) against the
reference and write its golden vectors.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).

    python oracle/check_toy_against_reference.py

The reference's LAIR iteration and LAIR sampler class (with final resampling) run on a VAE with
``var_latent_distribution=beta`` and ``use_uniform_prior=True``
(configs/toy_grid2dmog4/vae_comptrain.yaml:13-45) on the toy test points with their NaN pattern
(irwg/data/toy.py:605-621); the draw tape records the Beta samples themselves.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import cs_oracle as O  # noqa: E402
from oracle import cs_oracle_samplers as S  # noqa: E402
from oracle.check_against_reference import BASE_KW, close, ref_resfc  # noqa: E402
from oracle.ref_harness import DrawTape, load_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

NA = float("nan")
TOY_POINTS = torch.tensor([[NA, -2.], [NA, -1.], [NA, -.5], [NA, 0.], [NA, .5], [NA, 1.], [NA, 2.],
                           [-2., NA], [-1., NA], [-.5, NA], [0., NA], [.5, NA], [1., NA], [2., NA]])


def ref_vae(ns, enc, dec):
    D = ns.vae.DISTRIBUTION
    return ns.vae.VAE(ref_resfc(ns, dec), D["normal_with_eps"], ref_resfc(ns, enc), D["beta"], num_latent_samples=1,
                      lr_generator=1e-3, lr_latent=1e-3, use_uniform_prior=True).eval()


def case_toy_lair(ns, name, *, H, nb, K, R, I, Zd, seed, iters=2, scheme="gr", weighting="dmis", cluster=None):
    D = 2
    enc = O.make_resfc_weights(D, 2 * Zd, nb, H, seed)
    dec = O.make_resfc_weights(Zd, 2 * D, nb, H, seed + 1)
    vae = ref_vae(ns, enc, dec)
    M = ~torch.isnan(TOY_POINTS)
    B = M.shape[0]
    g = torch.Generator().manual_seed(seed + 2)
    X = torch.nan_to_num(TOY_POINTS).unsqueeze(1).repeat(1, K, 1)
    X = X * M.unsqueeze(1) + torch.randn(B, K, D, generator=g) * (~M.unsqueeze(1))
    M = M.unsqueeze(1)
    # weights are not stored: tests regenerate them with O.make_resfc_weights from (seed, nb)
    fx = dict(X0=X.numpy(), M=M.numpy(), cfg=np.array([B, K, R, I, D, Zd, H, iters], dtype=np.int64),
              wseed=np.array([seed, nb], dtype=np.int64), scheme=np.array(scheme), weighting=np.array(weighting),
              cluster=np.int64(cluster or 0))
    tape = DrawTape(seed + 3)
    Xr = X.clone()
    for t in range(iters):
        n0 = len(tape.entries)
        with torch.no_grad(), tape.recording():
            ret = ns.vae_imputation.importance_resampling_gibbs_iteration(
                vae, Xr, M, I, glr_cluster_size=cluster, weighting=weighting, resampling_method="multinomial",
                resampling_scheme=scheme, num_prior_replenish_proposals=R)
        ent = tape.entries[n0:]
        assert [k for k, _ in ent] == ["beta", "multinomial", "normal"], [k for k, _ in ent]
        draws = {"eps_z": torch.from_numpy(ent[0][1]), "u_res": torch.from_numpy(ent[1][1]), "eps_x": torch.from_numpy(ent[2][1])}
        o = O.lair_iteration(enc, dec, Xr, M, draws, num_imp_samples=I, num_prior_replenish_proposals=R,
                             latent_dist="beta", prior="uniform", weighting=weighting, resampling_scheme=scheme,
                             glr_cluster_size=cluster)
        assert torch.equal(o["Z_imp"], ret[6])
        assert torch.equal(o["log_weights"], ret[7]), float((o["log_weights"] - ret[7]).abs().max())
        assert torch.equal(o["X_new"], ret[0])
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


def case_toy_sampler(ns, name, *, H, nb, K, R, seed, T):
    """The sampler class end to end: initial imputation from the uniform prior, T LAIR iterations,
    final resampling."""
    D, Zd = 2, 1
    enc = O.make_resfc_weights(D, 2 * Zd, nb, H, seed)
    dec = O.make_resfc_weights(Zd, 2 * D, nb, H, seed + 1)
    vae = ref_vae(ns, enc, dec)
    X_true = TOY_POINTS.clone()
    M = ~torch.isnan(X_true)
    B = M.shape[0]
    tester = ns.test_step.TestVAELatentAdaptiveImportanceResampling(
        num_iterations=T, num_imp_samples=1, weighting_scheme="dmis", resampling_scheme="gr",
        resampling_method="multinomial", num_prior_replenish_proposals=R, resample_final_imps=True,
        store_latents=True, num_copies=K, **BASE_KW)
    tester.set_model(vae)
    tmp = tempfile.mkdtemp()
    tester.logger = types.SimpleNamespace(experiment=types.SimpleNamespace(get_logdir=lambda: tmp))
    batch = (X_true, M, torch.arange(B))
    tape = DrawTape(seed + 3)
    ref_out = []
    with torch.no_grad(), tape.recording():
        tester.on_test_batch_start(batch, 0, 0)
        Xk = X_true.unsqueeze(1).repeat(1, K, 1)
        X0 = tester.basic_imputation(Xk, M.unsqueeze(1))
        for o in tester.sampler(X0.clone(), M.unsqueeze(1)):
            ref_out.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
        tester.on_test_batch_end(None, batch, 0, 0)
    final = dict(np.load(os.path.join(tmp, "irwg_imputations_after_final_resampling_0.npz")))
    entries = tape.entries
    src = S.TapeSource(entries)
    Mk = M.unsqueeze(1)
    Xo = S.initial_imputation(dec, Xk, Mk, src, prior="uniform")
    assert torch.equal(Xo, X0), "initial imputation differs"
    ora = list(S.lair_sampler(enc, dec, Xo.clone(), Mk, src, num_iterations=T, num_prior_replenish_proposals=R,
                              store_latents=True, collect_proposals=True, latent_dist="beta", prior="uniform"))
    for t, (a, r) in enumerate(zip(ora, ref_out)):
        assert torch.equal(a["X"], r["X"]), f"X[{t}]"
    idx, Zf, imps = S.lair_final_resample(dec, [a["Z_prop"] for a in ora], [a["log_weights"] for a in ora],
                                          torch.nan_to_num(X_true), Mk, src)
    assert src.pos == len(entries)
    close(imps.unsqueeze(0), final["imputations"], "final imputations")
    fx = dict(S.tape_to_npz_dict(entries))
    fx.update(X_true=X_true.numpy(), M=Mk.numpy(), cfg=np.array([B, K, R, D, Zd, H, T], dtype=np.int64),
              wseed=np.array([seed, nb], dtype=np.int64),
              X_init=X0.numpy(), X_seq=np.stack([r["X"].numpy() for r in ref_out]),
              Z_seq=np.stack([r["Z"].numpy() for r in ref_out]),
              is_ess=np.array([float(r["is_ess"]) for r in ref_out], dtype=np.float32),
              final_imputations=final["imputations"], final_latents=final["latents"], final_idx=idx.numpy())
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


def main():
    ns = load_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    case_toy_lair(ns, "toy_lair_beta_uniform_h64", H=64, nb=2, K=4, R=1, I=1, Zd=1, seed=500)
    case_toy_lair(ns, "toy_lair_beta_uniform_zd3_i2", H=64, nb=1, K=3, R=2, I=2, Zd=3, seed=510)
    case_toy_lair(ns, "toy_lair_beta_uniform_h512", H=512, nb=4, K=4, R=1, I=1, Zd=1, seed=520, iters=1)
    # the toy configurations with hundreds of chains (configs/toy_grid2dmog4/sampling/vae_comptrain/irwg_k500_dmis_gr_mult_it10.yaml,
    # irwg_k500_dmis_glr5_mult_it10.yaml, irwg_k500_smis_glr5_mult_it100.yaml)
    case_toy_lair(ns, "toy_lair_k500_dmis_gr", H=64, nb=1, K=500, R=0, I=1, Zd=1, seed=540, iters=2)
    case_toy_lair(ns, "toy_lair_k500_dmis_glr5", H=64, nb=1, K=500, R=0, I=1, Zd=1, seed=550, iters=1, scheme="glr", cluster=5)
    case_toy_lair(ns, "toy_lair_k500_smis_glr5", H=64, nb=1, K=500, R=0, I=1, Zd=1, seed=560, iters=1, scheme="glr",
                  weighting="smis", cluster=5)
    case_toy_sampler(ns, "toy_sampler_beta_uniform", H=64, nb=1, K=4, R=1, seed=530, T=5)


if __name__ == "__main__":
    main()
