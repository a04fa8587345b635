"""Pin the oracle against the reference itself and write the golden vectors.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference).

    python oracle/check_against_reference.py            # check + (re)write tests/golden/*.npz
    python oracle/check_against_reference.py miniboone  # only the cases whose name contains "miniboone"

For every case it runs the reference package (through ``oracle/ref_harness.py``) on seeded
inputs while recording the draw tape, feeds the same draws to ``oracle/cs_oracle.py`` and
asserts equality of every intermediate (bitwise for indices / decisions; the float
arithmetic is the same torch-CPU ops in the same order, so floats are asserted to 1e-6
relative and are in practice identical).  The inputs, the tape and the *reference's*
outputs are then saved as small ``.npz`` fixtures.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import cs_oracle as O  # noqa: E402
from oracle.ref_harness import DrawTape, load_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
ONLY = []   # command-line arguments: substrings of the case names to (re)generate; empty = all


def selectable(fn):
    """Skip the case when case-name filters were given on the command line and none matches."""
    import functools

    @functools.wraps(fn)
    def wrapper(ns, name, **kw):
        if ONLY and not any(o in name for o in ONLY):
            return None
        return fn(ns, name, **kw)
    return wrapper


def ref_resfc(ns, w):
    H, in_dim = w["W0"].shape
    out_dim = w["Wf"].shape[0]
    norms = w.get("norms")
    kind = norms[0]["kind"] if norms else None
    net = ns.neural_nets.ResidualFCNetwork(in_dim, out_dim, len(w["blocks"]), H, use_layer_norm=kind == "layer",
                                           use_batch_norm=kind == "batch")
    with torch.no_grad():
        for blk, n in zip(net.blocks, norms or []):
            mod = blk.layer_norm if kind == "layer" else blk.batch_norm
            mod.weight.copy_(n["weight"]); mod.bias.copy_(n["bias"])
            if kind == "batch":
                mod.running_mean.copy_(n["mean"]); mod.running_var.copy_(n["var"])
        net.initial_layer.weight.copy_(w["W0"]); net.initial_layer.bias.copy_(w["b0"])
        for blk, (W1, b1, W2, b2) in zip(net.blocks, w["blocks"]):
            blk.linear_layers[0].weight.copy_(W1); blk.linear_layers[0].bias.copy_(b1)
            blk.linear_layers[1].weight.copy_(W2); blk.linear_layers[1].bias.copy_(b2)
        net.final_layer.weight.copy_(w["Wf"]); net.final_layer.bias.copy_(w["bf"])
    return net


def ref_vae(ns, enc, dec, latent_dist="normal_with_eps", gen_dist="normal_with_eps"):
    D = ns.vae.DISTRIBUTION
    vae = ns.vae.VAE(ref_resfc(ns, dec), D[gen_dist], ref_resfc(ns, enc), D[latent_dist],
                     num_latent_samples=1, lr_generator=1e-3, lr_latent=1e-3)
    return vae.eval()


def flat_weights(prefix, w):
    d = {f"{prefix}_W0": w["W0"].numpy(), f"{prefix}_b0": w["b0"].numpy(),
         f"{prefix}_Wf": w["Wf"].numpy(), f"{prefix}_bf": w["bf"].numpy(),
         f"{prefix}_nb": np.int64(len(w["blocks"]))}
    for i, (W1, b1, W2, b2) in enumerate(w["blocks"]):
        d[f"{prefix}_blk{i}_W1"] = W1.numpy(); d[f"{prefix}_blk{i}_b1"] = b1.numpy()
        d[f"{prefix}_blk{i}_W2"] = W2.numpy(); d[f"{prefix}_blk{i}_b2"] = b2.numpy()
    for i, n in enumerate(w.get("norms") or []):
        d[f"{prefix}_norm_kind"] = np.array(n["kind"])
        for k in ("weight", "bias", "mean", "var"):
            if k in n:
                d[f"{prefix}_norm{i}_{k}"] = n[k].numpy()
    return d


def close(a, b, what, rtol=1e-6, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind in "iub":
        assert np.array_equal(a, b), f"{what}: integer/bool mismatch"
    else:
        assert np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True), \
            f"{what}: max abs diff {np.nanmax(np.abs(a - b))}"


def make_inputs(B, K, D, seed, miss=0.5):
    g = torch.Generator().manual_seed(seed)
    X_true = torch.randn(B, D, generator=g)
    M = torch.rand(B, D, generator=g) > miss
    M[:, 0] = True  # no fully-missing rows (the reference filters them)
    X = X_true.unsqueeze(1).repeat(1, K, 1)
    X = X * M.unsqueeze(1) + torch.randn(B, K, D, generator=g) * (~M.unsqueeze(1))
    return X.contiguous(), M.unsqueeze(1).contiguous()


# ----------------------------------------------------------------------------------------
@selectable
def case_lair(ns, name, *, B, K, R, I, D, Zd, H, nb_enc, nb_dec, seed, scheme="gr",
              weighting="dmis", method="multinomial", cluster=None, resampled_latents=False,
              iters=2, temperature=None, anneal=None, latent_dist="normal_with_eps", norm=None,
              gen_dist="normal_with_eps"):
    studentt = gen_dist in ("studentt", "studentt_with_eps22")
    enc = O.make_resfc_weights(D, 2 * Zd, nb_enc, H, seed, norm=norm)
    dec = O.make_resfc_weights(Zd, (3 if studentt else 2) * D, nb_dec, H, seed + 1, norm=norm)
    if studentt:  # degrees of freedom around 5: with the df < 1 of a zero-centred raw output one sample in a few hundred
        dec["bf"][:D] += 5.0  # is beyond 1e6 and the next iteration's encoder overflows
    vae = ref_vae(ns, enc, dec, latent_dist=latent_dist, gen_dist=gen_dist)
    X, M = make_inputs(B, K, D, seed + 2)
    if gen_dist == "logitnormal_with_eps":  # data of the open unit interval
        X = torch.sigmoid(X)
    fx = dict(flat_weights("enc", enc)); fx.update(flat_weights("dec", dec))
    fx.update(X0=X.numpy(), M=M.numpy(),
              cfg=np.array([B, K, R, I, D, Zd, H, cluster or 0, iters], dtype=np.int64),
              scheme=np.array(scheme), weighting=np.array(weighting), method=np.array(method),
              resampled_latents=np.array(resampled_latents), temperature=np.float64(temperature or 0.0),
              anneal=np.array(anneal or ""), latent_dist=np.array(latent_dist), gen_dist=np.array(gen_dist))
    tape = DrawTape(seed + 3)
    Xr = X.clone()
    for t in range(iters):
        n0 = len(tape.entries)
        with torch.no_grad(), tape.recording():
            ret = ns.vae_imputation.importance_resampling_gibbs_iteration(
                vae, Xr, M, I, glr_cluster_size=cluster, weighting=weighting,
                resampling_method=method, resampling_scheme=scheme,
                num_prior_replenish_proposals=R, return_resampled_latents=resampled_latents,
                var_proposal_temperature=temperature, var_proposal_anneal_type_to_prior=anneal)
        ent = tape.entries[n0:]
        kinds = [k for k, _ in ent]
        res_kind = "multinomial" if method == "multinomial" else "rand"
        if method == "multinomial" and scheme in ("lr", "ir"):
            res_kind = "exponential"   # one index per chain: ATen's single-sample multinomial
        expect = ["normal", res_kind] + ([res_kind] if resampled_latents else []) + ["normal"] + (["std_gamma"] if studentt else [])
        assert kinds == expect, kinds
        draws = {"eps_z": torch.from_numpy(ent[0][1]), "u_res": torch.from_numpy(ent[1][1]),
                 "eps_x": torch.from_numpy(ent[-2 if studentt else -1][1])}
        if studentt:  # Chi2.rsample = Gamma(df / 2, rate 0.5).rsample: _standard_gamma / 0.5, clamped to tiny
            draws["chi2_x"] = (torch.from_numpy(ent[-1][1]) / 0.5).clamp(min=torch.finfo(torch.float32).tiny)
        if resampled_latents:
            draws["u_res2"] = torch.from_numpy(ent[2][1])
        o = O.lair_iteration(enc, dec, Xr, M, draws, num_imp_samples=I, weighting=weighting,
                             resampling_scheme=scheme, resampling_method=method,
                             glr_cluster_size=cluster, num_prior_replenish_proposals=R,
                             return_resampled_latents=resampled_latents, latent_dist=latent_dist, gen_dist=gen_dist,
                             var_proposal_temperature=temperature, var_proposal_anneal_type_to_prior=anneal)
        close(o["X_new"], ret[0], "X_new"); close(o["ess"], ret[1], "ess")
        close(o["transformed_ess"], ret[2], "tess"); close(o["perplexity"], ret[3], "perplexity")
        assert o["total_num_props"] == ret[4]
        close(o["norm_log_weights"], ret[5], "norm_log_weights")
        close(o["Z_imp"], ret[6], "Z_imp"); close(o["log_weights"], ret[7], "log_weights")
        assert torch.equal(o["X_new"], ret[0]), "X_new not bitwise equal"
        assert torch.equal(o["log_weights"], ret[7]), "log_weights not bitwise equal"
        if resampled_latents:
            close(o["latents_resampled"][:, :K] if False else o["latents_resampled"], ret[9], "latents_resampled")
        # the reference does not return indices: recover them from its outputs for the fixture
        if studentt:
            fx[f"t{t}_chi2_x"] = draws["chi2_x"].numpy()
        fx.update({f"t{t}_eps_z": ent[0][1], f"t{t}_u_res": ent[1][1], f"t{t}_eps_x": draws["eps_x"].numpy(),
                   f"t{t}_X_new": ret[0].numpy(), f"t{t}_ess": ret[1].numpy(),
                   f"t{t}_perplexity": ret[3].numpy(), f"t{t}_norm_log_weights": ret[5].numpy(),
                   f"t{t}_Z_imp": ret[6].numpy(), f"t{t}_log_weights": ret[7].numpy(),
                   f"t{t}_idx": o["idx_global"].numpy(),
                   f"t{t}_q_loc": o["q_loc"].numpy(), f"t{t}_q_scale": o["q_scale"].numpy(),
                   f"t{t}_gen_params": o["generator_params"].numpy()})
        if resampled_latents:
            fx[f"t{t}_u_res2"] = ent[2][1]
            fx[f"t{t}_latents_resampled"] = ret[9].numpy()
        Xr = ret[0]
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


BASE_KW = dict(store_last_n_iterations=-1, store_every_n_iterations=1, imputation_metric_quantiles=[],
               compute_f1_metric=False, compute_inv_f1_metric=False, compute_bin_accuracy_metric=False,
               compute_rmse_metric=False, compute_ssim_metric=False, compute_latent_wass_metric=False,
               imputation_fn="vae_samples")


def seeded_weights(prefix, in_dim, out_dim, nb, H, seed):
    """Fixture entry naming a network by its shape and seed instead of storing its weights: the tests regenerate
    them with O.make_resfc_weights (full-size networks would make the fixtures megabytes of random floats)."""
    return {f"{prefix}_seeded": np.array([in_dim, out_dim, nb, H, seed], dtype=np.int64)}


@selectable
def case_acmwg(ns, name, *, B, K, D, Zd, H, seed, warmup, W, T, p, nkpc=None, cluster=None, R=0, plain_mwg=False,
               ablation=False, clip=False, temperature=None, anneal=None, nb_enc=1, nb_dec=1, store_weights=True,
               gen_dist="normal_with_eps"):
    """AC-MWG sampler class (or, with plain_mwg, TestVAEMetropolisWithinGibbs) against the oracle."""
    from oracle import cs_oracle_samplers as S
    studentt = gen_dist in ("studentt", "studentt_with_eps22")
    enc = O.make_resfc_weights(D, 2 * Zd, nb_enc, H, seed)
    dec = O.make_resfc_weights(Zd, (3 if studentt else 2) * D, nb_dec, H, seed + 1)
    if studentt:
        dec["bf"][:D] += 5.0  # degrees of freedom around 5 (see case_lair)
    vae = ref_vae(ns, enc, dec, gen_dist=gen_dist)
    X, M = make_inputs(B, K, D, seed + 2)
    if gen_dist == "logitnormal_with_eps":
        X = torch.sigmoid(X)
    kw = dict(num_pseudo_warmup_iterations=W, num_mwg_iterations=T,
              warmup_method=warmup, num_copies=K, clip_imputations_during_pg_warmup=clip,
              var_proposal_temperature=temperature, var_proposal_anneal_type_to_prior=anneal, **BASE_KW)
    if warmup == "irwg":
        kw.update(num_k_per_cluster=nkpc, glr_cluster_size=cluster, num_prior_replenish_proposals=R)
    if plain_mwg:
        tester = ns.test_step.TestVAEMetropolisWithinGibbs(**kw)
    else:
        tester = ns.test_step.TestVAEAdaptiveCollapsedMetropolisWithinGibbs(
            prior_mixture_probability=p, ablation_use_mwg_target=ablation, **kw)
    tester.set_model(vae)
    tape = DrawTape(seed + 3)
    ref_out = []
    with torch.no_grad(), tape.recording():
        for o in tester.sampler(X.clone(), M):
            ref_out.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
    assert len(ref_out) == tester.max_iterations
    entries = S.compact_tape(tape.entries)
    src = S.TapeSource(entries)
    okw = dict(num_pseudo_warmup_iterations=W, num_mwg_iterations=T, prior_mixture_probability=p,
               warmup_method=warmup, num_k_per_cluster=nkpc, glr_cluster_size=cluster,
               num_prior_replenish_proposals=R, plain_mwg=plain_mwg, ablation_use_mwg_target=ablation,
               clip_imputations_during_pg_warmup=clip, var_proposal_temperature=temperature,
               var_proposal_anneal_type_to_prior=anneal, gen_dist=gen_dist)
    ora = list(S.acmwg_sampler(enc, dec, X.clone(), M, src, **okw))
    assert src.pos == len(entries), "oracle did not consume the whole tape"
    assert len(ora) == len(ref_out)
    if store_weights:
        fx = dict(flat_weights("enc", enc)); fx.update(flat_weights("dec", dec))
    else:
        fx = dict(seeded_weights("enc", D, 2 * Zd, nb_enc, H, seed)); fx.update(seeded_weights("dec", Zd, 2 * D, nb_dec, H, seed + 1))
    fx.update(S.tape_to_npz_dict(entries))
    accs, slots = [], []
    for t, (a, r) in enumerate(zip(ora, ref_out)):
        close(a["X"], r["X"], f"X[{t}]")
        assert torch.equal(a["X"], r["X"]), f"X[{t}] not bitwise equal"
        if "Z" in r:
            close(a["Z"], r["Z"], f"Z[{t}]")
        close(torch.as_tensor(a["acceptance"], dtype=torch.float32), torch.as_tensor(r["acceptance"], dtype=torch.float32), "acc")
        if "accepted" in a:
            accs.append(a["accepted"].numpy())
            if "slots" in a:
                slots.append(a["slots"].numpy())
    fx.update(X0=X.numpy(), M=M.numpy(),
              cfg=np.array([B, K, D, Zd, H, W, T, nkpc or 0, cluster or 0, R], dtype=np.int64),
              warmup=np.array(warmup), p=np.float64(p), plain_mwg=np.array(plain_mwg), ablation=np.array(ablation),
              clip=np.array(clip), temperature=np.float64(temperature or 0.0), anneal=np.array(anneal or ""),
              gen_dist=np.array(gen_dist),
              X_seq=np.stack([r["X"].numpy() for r in ref_out]),
              Z_last=ref_out[-1]["Z"].numpy(),
              acceptance=np.array([float(r["acceptance"]) for r in ref_out], dtype=np.float32),
              accepted=np.stack(accs))
    if slots:
        fx["slots"] = np.stack(slots)
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}  (iterations {len(ref_out)}, mean acceptance {np.mean(accs):.2f})")


@selectable
def case_pseudo_gibbs(ns, name, *, B, K, D, Zd, H, seed, T, clip):
    """TestVAEPseudoGibbs.sampler (test_step_vae_sampling.py:1023-1042) against the oracle."""
    from oracle import cs_oracle_samplers as S
    enc = O.make_resfc_weights(D, 2 * Zd, 1, H, seed)
    dec = O.make_resfc_weights(Zd, 2 * D, 1, H, seed + 1)
    vae = ref_vae(ns, enc, dec)
    X, M = make_inputs(B, K, D, seed + 2)
    tester = ns.test_step.TestVAEPseudoGibbs(num_iterations=T, clip_imputations=clip, num_copies=K, **BASE_KW)
    tester.set_model(vae)
    tape = DrawTape(seed + 3)
    ref_out = []
    with torch.no_grad(), tape.recording():
        for o in tester.sampler(X.clone(), M):
            ref_out.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
    src = S.TapeSource(tape.entries)
    ora = list(S.pseudo_gibbs_sampler(enc, dec, X.clone(), M, src, num_iterations=T, clip_imputations=clip))
    assert src.pos == len(tape.entries) and len(ora) == len(ref_out) == T
    for t, (a, r) in enumerate(zip(ora, ref_out)):
        assert torch.equal(a["X"], r["X"]), f"X[{t}] not bitwise equal"
        assert torch.equal(a["Z"], r["Z"]), f"Z[{t}] not bitwise equal"
    fx = dict(flat_weights("enc", enc)); fx.update(flat_weights("dec", dec))
    fx.update(S.tape_to_npz_dict(tape.entries))
    fx.update(X0=X.numpy(), M=M.numpy(), cfg=np.array([B, K, D, Zd, H, T], dtype=np.int64), clip=np.array(clip),
              X_seq=np.stack([r["X"].numpy() for r in ref_out]), Z_seq=np.stack([r["Z"].numpy() for r in ref_out]))
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


@selectable
def case_ir_sampler(ns, name, *, B, K, I, D, Zd, H, seed, T, scheme="gr", weighting="dmis", method="multinomial",
                    cluster=None, n_final=None):
    """TestVAEImportanceResampling (sampler + the final resampling of on_test_batch_end)."""
    import tempfile, types
    from oracle import cs_oracle_samplers as S
    enc = O.make_resfc_weights(D, 2 * Zd, 1, H, seed)
    dec = O.make_resfc_weights(Zd, 2 * D, 1, H, seed + 1)
    vae = ref_vae(ns, enc, dec)
    X, M = make_inputs(B, K, D, seed + 2)
    X_true = X[:, 0].clone()
    tester = ns.test_step.TestVAEImportanceResampling(
        num_iterations=T, num_imp_samples=I, weighting_scheme=weighting, resampling_scheme=scheme,
        resampling_method=method, glr_cluster_size=cluster, resample_final_imps_num_samples=n_final,
        num_copies=K, **BASE_KW)
    tester.set_model(vae)
    tmp = tempfile.mkdtemp()
    tester.logger = types.SimpleNamespace(experiment=types.SimpleNamespace(get_logdir=lambda: tmp))
    batch = (X_true, M[:, 0], torch.arange(B))
    tape = DrawTape(seed + 3)
    ref_out = []
    with torch.no_grad(), tape.recording():
        tester.on_test_batch_start(batch, 0, 0)
        for o in tester.sampler(X.clone(), M):
            ref_out.append({k: (v.clone() if torch.is_tensor(v) else np.array(v)) for k, v in o.items()})
        tester.on_test_batch_end(None, batch, 0, 0)
    final = dict(np.load(os.path.join(tmp, "imputations_0.npz")))
    src = S.TapeSource(tape.entries)
    ora = list(S.ir_sampler(dec, X.clone(), M, src, num_iterations=T, num_imp_samples=I, weighting_scheme=weighting,
                            resampling_scheme=scheme, resampling_method=method, glr_cluster_size=cluster))
    for t, (a, r) in enumerate(zip(ora, ref_out)):
        assert torch.equal(a["X"], r["X"]), f"X[{t}] not bitwise equal"
        close(a["is_ess_standard"], r["is_ess_standard"], "ess")
        close(a["perplexity_standard"], r["perplexity_standard"], "perplexity")
    idx, Zf, imps = S.lair_final_resample(dec, [a["Z_prop"] for a in ora], [a["log_weights"] for a in ora],
                                          X_true, M, src, num_samples=n_final)
    assert src.pos == len(tape.entries)
    close(imps.unsqueeze(0), final["imputations"], "final imputations")
    fx = dict(flat_weights("enc", enc)); fx.update(flat_weights("dec", dec))
    fx.update(S.tape_to_npz_dict(tape.entries))
    fx.update(X0=X.numpy(), M=M.numpy(), X_true=X_true.numpy(),
              cfg=np.array([B, K, I, D, Zd, H, T, cluster or 0, n_final or 0], dtype=np.int64),
              scheme=np.array(scheme), weighting=np.array(weighting), method=np.array(method),
              X_seq=np.stack([r["X"].numpy() for r in ref_out]),
              is_ess=np.array([float(r["is_ess_standard"]) for r in ref_out], dtype=np.float32),
              perplexity=np.array([float(r["perplexity_standard"]) for r in ref_out], dtype=np.float32),
              final_imputations=final["imputations"], final_idx=idx.numpy())
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


def check_single_sample_multinomial():
    """The tape's restatement of ATen's single-sample multinomial equals the real call."""
    from oracle.ref_harness import multinomial_single_from_exponentials
    p = torch.softmax(torch.randn(50, 7, generator=torch.Generator().manual_seed(1)), -1)
    g = torch.Generator().manual_seed(2)
    real = torch.multinomial(p, 1, True, generator=g)
    g = torch.Generator().manual_seed(2)
    e = torch.empty_like(p).exponential_(1, generator=g)
    assert torch.equal(real, multinomial_single_from_exponentials(p, e))
    print("[ok] single-sample multinomial = argmax(p / Exp(1))")


@selectable
def case_lair_sampler(ns, name, *, B, K, R, D, Zd, H, seed, T, n_final=None, freq=1, final=True, scheme="gr",
                      weighting="dmis", cluster=None, pg_warmup=0, num_hist=None, no_hist_its=None, hist_len=None):
    import tempfile, types
    from oracle import cs_oracle_samplers as S
    enc = O.make_resfc_weights(D, 2 * Zd, 1, H, seed)
    dec = O.make_resfc_weights(Zd, 2 * D, 1, H, seed + 1)
    vae = ref_vae(ns, enc, dec)
    X, M = make_inputs(B, K, D, seed + 2)
    g = torch.Generator().manual_seed(seed + 5)
    X_true = X[:, 0].clone()
    tester = ns.test_step.TestVAELatentAdaptiveImportanceResampling(
        num_iterations=T, num_imp_samples=1, weighting_scheme=weighting, resampling_scheme=scheme,
        resampling_method="multinomial", num_prior_replenish_proposals=R, glr_cluster_size=cluster,
        prior_replenish_iteration_frequency=freq, resample_final_imps=final,
        resample_final_imps_num_samples=n_final, store_latents=True, num_copies=K,
        num_pseudo_warmup_iterations=pg_warmup, num_historical_proposals=num_hist,
        no_historic_proposals_iterations=no_hist_its, imputation_history_length=hist_len, **BASE_KW)
    tester.set_model(vae)
    tmp = tempfile.mkdtemp()
    tester.logger = types.SimpleNamespace(experiment=types.SimpleNamespace(get_logdir=lambda: tmp))
    batch = (X_true, M[:, 0], torch.arange(B))
    tape = DrawTape(seed + 3)
    ref_out = []
    with torch.no_grad(), tape.recording():
        tester.on_test_batch_start(batch, 0, 0)
        for o in tester.sampler(X.clone(), M):
            ref_out.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
        tester.on_test_batch_end(None, batch, 0, 0)
    final_res = final
    final = dict(np.load(os.path.join(tmp, "irwg_imputations_after_final_resampling_0.npz"))) if final_res else {}
    entries = S.compact_tape(tape.entries)
    src = S.TapeSource(entries)
    ora = list(S.lair_sampler(enc, dec, X.clone(), M, src, num_iterations=T, num_prior_replenish_proposals=R,
                              prior_replenish_iteration_frequency=freq, store_latents=True, collect_proposals=True,
                              resampling_scheme=scheme, weighting_scheme=weighting, glr_cluster_size=cluster,
                              num_pseudo_warmup_iterations=pg_warmup, num_historical_proposals=num_hist,
                              no_historic_proposals_iterations=no_hist_its, imputation_history_length=hist_len))
    assert len(ora) == len(ref_out) == T + pg_warmup
    for t, (a, r) in enumerate(zip(ora, ref_out)):
        assert torch.equal(a["X"], r["X"]), f"X[{t}]"
        close(a["Z"], r["Z"], "Z"); close(torch.as_tensor(a["is_ess"]), torch.as_tensor(r["is_ess"]), "ess")
    if not final_res:
        assert src.pos == len(entries), "oracle did not consume the whole tape"
        idx = torch.zeros(0)
        final = dict(imputations=np.zeros(0, np.float32), latents=np.zeros(0, np.float32))
    elif freq == 1:
        idx, Zf, imps = S.lair_final_resample(dec, [a["Z_prop"] for a in ora], [a["log_weights"] for a in ora],
                                              X_true, M, src, num_samples=n_final)
        assert src.pos == len(entries)
        close(imps.unsqueeze(0), final["imputations"], "final imputations")
        close(Zf, final["latents"], "final latents")
    fx = dict(flat_weights("enc", enc)); fx.update(flat_weights("dec", dec))
    fx.update(S.tape_to_npz_dict(entries))
    fx.update(X0=X.numpy(), M=M.numpy(), X_true=X_true.numpy(),
              cfg=np.array([B, K, R, D, Zd, H, T, n_final or 0, freq], dtype=np.int64),
              hist_cfg=np.array([pg_warmup, -1 if num_hist is None else num_hist, -1 if no_hist_its is None else no_hist_its,
                                 hist_len or 0, cluster or 0, int(final_res)], dtype=np.int64),
              scheme=np.array(scheme), weighting=np.array(weighting),
              X_seq=np.stack([r["X"].numpy() for r in ref_out]),
              is_ess=np.array([float(r["is_ess"]) for r in ref_out], dtype=np.float32),
              final_imputations=final["imputations"], final_latents=final["latents"],
              final_idx=idx.numpy())
    if len({tuple(r["Z"].shape) for r in ref_out}) == 1:
        fx["Z_seq"] = np.stack([r["Z"].numpy() for r in ref_out])
    else:   # the proposal count changes when the history joins: one array per iteration
        fx.update({f"Z_t{t}": r["Z"].numpy() for t, r in enumerate(ref_out)})
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


@selectable
def case_mlp_norm(ns, name, *, in_dim, out_dim, nb, H, rows, seed, norm):
    """ResidualFCNetwork.forward with LayerNorm / BatchNorm residual blocks (neural_nets.py:87-103), eval mode."""
    w = O.make_resfc_weights(in_dim, out_dim, nb, H, seed, norm=norm)
    net = ref_resfc(ns, w).eval()
    x = torch.randn(3, rows, in_dim, generator=torch.Generator().manual_seed(seed + 1))
    with torch.no_grad():
        y = net(x)
    assert torch.equal(y, O.resfc_forward(w, x)), "forward not bitwise equal"
    fx = dict(flat_weights("net", w))
    fx.update(x=x.numpy(), y=y.numpy())
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


@selectable
def case_metrics(ns, name, *, B, K, D, Zd, H, seed):
    """Per-iteration diagnostics (irwg/sampling/imputation_metrics.py:85-102,133-143,200-213)."""
    import importlib
    im = importlib.import_module("irwg.sampling.imputation_metrics")
    enc = O.make_resfc_weights(D, 2 * Zd, 2, H, seed)
    dec = O.make_resfc_weights(Zd, 2 * D, 1, H, seed + 1)
    vae = ref_vae(ns, enc, dec)
    X, M = make_inputs(B, K, D, seed + 2)
    g = torch.Generator().manual_seed(seed + 3)
    X_true = X[:, :1] * M + torch.randn(B, 1, D, generator=g) * (~M)
    M[0] = True  # a fully observed example: the count of missing entries is replaced by 1
    with torch.no_grad():
        rmse, mae = im.imputation_rmse_metric(X, X_true, M), im.imputation_mae_metric(X, X_true, M)
        wass = im.imputation_latent_wass(X, X_true, vae)
        assert torch.equal(rmse, O.imputation_rmse_metric(X, X_true, M))
        assert torch.equal(mae, O.imputation_mae_metric(X, X_true, M))
        assert torch.equal(wass, O.imputation_latent_wass(enc, X, X_true))
    fx = dict(flat_weights("enc", enc)); fx.update(flat_weights("dec", dec))
    fx.update(X=X.numpy(), X_true=X_true.numpy(), M=M.numpy(), rmse=rmse.numpy(), mae=mae.numpy(), latent_wass=wass.numpy())
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **fx)
    print(f"[ok] {name}")


def main():
    ONLY[:] = sys.argv[1:]
    os.makedirs(GOLDEN, exist_ok=True)
    ns = load_reference()
    torch.manual_seed(0)
    case_metrics(ns, "metrics_small", B=7, K=3, D=70, Zd=5, H=64, seed=31)
    case_lair(ns, "lair_gr_mult_small", B=6, K=4, R=1, I=1, D=11, Zd=5, H=64, nb_enc=2, nb_dec=1, seed=101)
    case_lair(ns, "lair_gr_mult_i2_lat", B=5, K=3, R=2, I=2, D=9, Zd=4, H=64, nb_enc=1, nb_dec=2, seed=202,
              resampled_latents=True)
    case_lair(ns, "lair_gr_mult_noreplenish", B=4, K=5, R=0, I=1, D=8, Zd=16, H=64, nb_enc=1, nb_dec=1, seed=303)
    case_lair(ns, "lair_gr_syst", B=6, K=4, R=1, I=1, D=11, Zd=5, H=64, nb_enc=1, nb_dec=1, seed=404,
              method="systematic")
    case_lair(ns, "lair_glr_groups", B=4, K=20, R=5, I=1, D=10, Zd=6, H=64, nb_enc=1, nb_dec=1, seed=505,
              scheme="glr", weighting="dmis_within_groups", cluster=5)
    case_lair(ns, "lair_miniboone_shape", B=3, K=4, R=1, I=1, D=43, Zd=32, H=256, nb_enc=5, nb_dec=2, seed=606,
              iters=1)
    case_acmwg(ns, "acmwg_gen_studentt", B=3, K=4, D=9, Zd=4, H=64, seed=821, warmup="irwg", W=2, T=8, p=0.2,
               nkpc=3, cluster=4, R=4, gen_dist="studentt")
    case_acmwg(ns, "acmwg_gen_logitnormal", B=3, K=4, D=8, Zd=4, H=64, seed=822, warmup="pseudo_gibbs", W=3, T=8, p=0.1,
               gen_dist="logitnormal_with_eps")
    # likelihood families no configuration uses (irwg/models/vae.py:220-229, 244-247)
    case_lair(ns, "lair_gen_studentt", B=5, K=4, R=1, I=1, D=11, Zd=5, H=64, nb_enc=1, nb_dec=1, seed=811,
              gen_dist="studentt")
    case_lair(ns, "lair_gen_studentt_eps22", B=4, K=3, R=2, I=2, D=9, Zd=4, H=64, nb_enc=1, nb_dec=2, seed=812,
              gen_dist="studentt_with_eps22")
    case_lair(ns, "lair_gen_logitnormal", B=5, K=4, R=1, I=1, D=10, Zd=5, H=64, nb_enc=1, nb_dec=1, seed=813,
              gen_dist="logitnormal_with_eps")
    case_acmwg(ns, "acmwg_irwg_warmup", B=3, K=5, D=9, Zd=4, H=64, seed=707, warmup="irwg", W=2, T=12, p=0.3,
               nkpc=4, cluster=5, R=5)
    # the MiniBooNE AC-MWG configuration (configs/uci_miniboone/sampling_u50/..k5_mwgm_usehistrestricted_005prior_
    # with_irwg_warmup_replenish1.yaml): D=43, Zd=32, hidden 256, encoder 5 / decoder 2 blocks, K=5, p=0.05,
    # irwg warm-up on 20 particles + 5 replenishing proposals in clusters of 5; shortened to 2 warm-up iterations
    # (8 yields) and 14 MWG steps
    case_acmwg(ns, "acmwg_miniboone_config", B=16, K=5, D=43, Zd=32, H=256, seed=717, warmup="irwg", W=2, T=14,
               p=0.05, nkpc=4, cluster=5, R=5, nb_enc=5, nb_dec=2, store_weights=False)
    case_acmwg(ns, "acmwg_pg_warmup", B=4, K=3, D=7, Zd=3, H=64, seed=808, warmup="pseudo_gibbs", W=3, T=10, p=0.05)
    case_acmwg(ns, "acmwg_nowarmup_noprior", B=4, K=2, D=6, Zd=3, H=64, seed=909, warmup="none", W=0, T=8, p=0.0)
    case_mlp_norm(ns, "mlp_layernorm", in_dim=11, out_dim=10, nb=2, H=64, rows=9, seed=545, norm="layer")
    case_mlp_norm(ns, "mlp_layernorm_h256", in_dim=43, out_dim=64, nb=2, H=256, rows=5, seed=555, norm="layer")
    case_mlp_norm(ns, "mlp_batchnorm", in_dim=11, out_dim=10, nb=2, H=64, rows=9, seed=565, norm="batch")
    case_lair(ns, "lair_layernorm_nets", B=5, K=4, R=1, I=1, D=11, Zd=5, H=64, nb_enc=2, nb_dec=1, seed=575, norm="layer")
    case_lair(ns, "lair_batchnorm_nets", B=5, K=4, R=1, I=1, D=11, Zd=5, H=64, nb_enc=1, nb_dec=2, seed=585, norm="batch")
    check_single_sample_multinomial()
    case_lair(ns, "lair_lr_mult", B=5, K=3, R=1, I=4, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=464, scheme="lr",
              weighting="smis", resampled_latents=True)
    case_lair(ns, "lair_lr_syst", B=5, K=4, R=0, I=3, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=474, scheme="lr",
              method="systematic")
    case_lair(ns, "lair_ir_mult", B=5, K=3, R=0, I=3, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=484, scheme="ir")
    case_lair(ns, "lair_ir_mult_i6", B=4, K=3, R=0, I=6, D=8, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=494, scheme="ir",
              weighting="smis")
    case_lair(ns, "lair_ir_syst", B=4, K=4, R=0, I=4, D=8, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=504, scheme="ir",
              method="systematic")
    case_ir_sampler(ns, "ir_sampler_gr", B=4, K=3, I=2, D=8, Zd=4, H=64, seed=514, T=4)
    case_ir_sampler(ns, "ir_sampler_ir_smis", B=3, K=4, I=4, D=8, Zd=4, H=64, seed=524, T=3, scheme="ir",
                    weighting="smis", n_final=5)
    case_ir_sampler(ns, "ir_sampler_lr", B=3, K=2, I=5, D=7, Zd=3, H=64, seed=534, T=3, scheme="lr")
    case_lair(ns, "lair_tempered_annealed", B=6, K=4, R=1, I=1, D=11, Zd=5, H=64, nb_enc=1, nb_dec=1, seed=414,
              temperature=1.3, anneal="linear")
    case_lair(ns, "lair_annealed_025_normal", B=5, K=3, R=0, I=2, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=424,
              anneal="linear_0.25", latent_dist="normal")
    case_lair(ns, "lair_tempered", B=5, K=3, R=2, I=1, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=434, temperature=0.7)
    case_acmwg(ns, "acmwg_tempered_annealed", B=4, K=3, D=7, Zd=3, H=64, seed=444, warmup="irwg", W=2, T=10, p=0.1,
               nkpc=2, cluster=3, R=3, temperature=1.2, anneal="linear_0.25")
    case_acmwg(ns, "mwg_tempered_annealed", B=4, K=3, D=7, Zd=3, H=64, seed=454, warmup="pseudo_gibbs", W=2, T=10,
               p=0.0, plain_mwg=True, temperature=0.8, anneal="linear")
    case_acmwg(ns, "acmwg_ablation_mwg_target", B=4, K=3, D=7, Zd=3, H=64, seed=919, warmup="none", W=0, T=10, p=0.1,
               ablation=True)
    case_acmwg(ns, "acmwg_ablation_pg_warmup", B=3, K=2, D=6, Zd=3, H=64, seed=929, warmup="pseudo_gibbs", W=2, T=8,
               p=0.1, ablation=True)
    case_acmwg(ns, "mwg_pg_warmup_clip", B=4, K=3, D=7, Zd=3, H=64, seed=939, warmup="pseudo_gibbs", W=3, T=10, p=0.0,
               plain_mwg=True, clip=True)
    case_acmwg(ns, "mwg_nowarmup", B=4, K=2, D=6, Zd=3, H=64, seed=949, warmup="none", W=0, T=10, p=0.0, plain_mwg=True)
    case_acmwg(ns, "mwg_irwg_warmup", B=3, K=5, D=9, Zd=4, H=64, seed=959, warmup="irwg", W=2, T=8, p=0.0,
               nkpc=4, cluster=5, R=5, plain_mwg=True)
    case_pseudo_gibbs(ns, "pseudo_gibbs_clip", B=4, K=3, D=7, Zd=3, H=64, seed=969, T=6, clip=True)
    case_pseudo_gibbs(ns, "pseudo_gibbs_noclip", B=3, K=2, D=6, Zd=3, H=64, seed=979, T=5, clip=False)
    # proposal sets of more than 64 per example (configs/mnist_gmm/samples/vae_convresnet3_k100_irwg_i1_dmisg_glr5_mult.yaml:
    # K=100 in clusters of 5 with grouped weights; the stage after the networks is the same for any network family)
    case_lair(ns, "lair_k100_glr5_dmisg", B=3, K=100, R=0, I=1, D=11, Zd=5, H=64, nb_enc=1, nb_dec=1, seed=616,
              scheme="glr", weighting="dmis_within_groups", cluster=5)
    case_lair(ns, "lair_k96_r4_gr_dmis_lat", B=3, K=96, R=4, I=1, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=626,
              resampled_latents=True)
    case_lair(ns, "lair_k40_i3_gr_syst", B=3, K=40, R=2, I=3, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=636,
              method="systematic")
    case_lair(ns, "lair_k30_i4_lr_smis", B=3, K=30, R=0, I=4, D=9, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=646,
              scheme="lr", weighting="smis")
    case_lair(ns, "lair_k24_i5_ir", B=3, K=24, R=0, I=5, D=8, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=656, scheme="ir")
    case_lair(ns, "lair_k120_glr10_dmis", B=2, K=120, R=0, I=1, D=8, Zd=4, H=64, nb_enc=1, nb_dec=1, seed=666,
              scheme="glr", weighting="dmis", cluster=10)
    case_lair_sampler(ns, "lair_sampler_final", B=4, K=3, R=1, D=8, Zd=4, H=64, seed=111, T=5)
    case_lair_sampler(ns, "lair_sampler_final_n5", B=3, K=4, R=1, D=8, Zd=4, H=64, seed=222, T=4, n_final=5)
    case_lair_sampler(ns, "lair_sampler_history_gr", B=4, K=3, R=1, D=8, Zd=4, H=64, seed=333, T=7, final=False,
                      pg_warmup=2, num_hist=2, no_hist_its=3, hist_len=4)
    case_lair_sampler(ns, "lair_sampler_history_glr", B=3, K=5, R=0, D=8, Zd=4, H=64, seed=444, T=12, final=False,
                      scheme="glr", weighting="dmis_within_groups", cluster=5, pg_warmup=2, num_hist=10,
                      no_hist_its=10, hist_len=12)


if __name__ == "__main__":
    main()
