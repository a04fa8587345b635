"""CPU oracle of the two sampler loops (LAIR and AC-MWG) built from ``cs_oracle``.

TEST INFRASTRUCTURE ONLY (see the header of ``oracle/cs_oracle.py``).

Restates the control flow of
  * ``TestVAELatentAdaptiveImportanceResampling.sampler`` and its final resampling
    (irwg/sampling/test_step_vae_sampling.py:1853-1977, 1979-2045),
  * ``TestVAEAdaptiveCollapsedMetropolisWithinGibbs.sampler`` with the ``none``,
    ``pseudo_gibbs`` and ``irwg`` warm-ups (:1512-1768),
  * ``TestVAEMetropolisWithinGibbs.sampler`` (:1090-1282) and ``TestVAEPseudoGibbs.sampler``
    (:1023-1042),
  * the 'vae_samples' initial imputation (:279-287),
with every random draw taken, in the reference's call order, from a *draw source*: either a
recorded tape (parity against the reference / against the HIP engine) or a seeded generator
(CPU baseline timing).
"""
import numpy as np
import torch

from . import cs_oracle as O


def x_draws(src, gen_dist, *shape):
    """Draws of one generator sample of the given shape, in the reference's order: uniforms for Bernoulli, standard
    normals otherwise, and for the Student-t families the chi-square draws behind them (StudentT.rsample)."""
    if gen_dist == "bern":
        return {"eps_x": src.uniform(*shape)}
    d = {"eps_x": src.normal(*shape)}
    if gen_dist in ("studentt", "studentt_with_eps22"):
        d["chi2_x"] = src.chi2(*shape)
    return d


class TapeSource:
    """Replays a recorded draw tape [(kind, ndarray), ...] in order."""

    def __init__(self, entries):
        self.entries = list(entries)
        self.pos = 0

    def _take(self, kinds, shape=None):
        kind, arr = self.entries[self.pos]
        assert kind in kinds, f"tape entry {self.pos}: expected {kinds}, found {kind}"
        if shape is not None:
            assert tuple(arr.shape) == tuple(shape), f"tape entry {self.pos}: shape {arr.shape} != {shape}"
        self.pos += 1
        return torch.from_numpy(np.array(arr))

    def normal(self, *shape):
        return self._take(("normal", "randn"), shape)

    def uniform(self, *shape):
        return self._take(("rand", "rand_like", "bernoulli"), shape)

    def beta(self, *shape):
        """Samples of a Beta posterior as torch's Dirichlet sampler returned them."""
        return self._take(("beta",), shape)

    def chi2(self, *shape):
        """Chi-square draws of StudentT.rsample: Chi2(df).rsample() = _standard_gamma(df / 2) / 0.5, clamped to tiny
        (torch.distributions.Gamma.rsample); the tape holds the standard-gamma variates."""
        g = self._take(("std_gamma",), shape)
        return (g / 0.5).clamp(min=torch.finfo(torch.float32).tiny)

    def uniform64(self, *shape):
        return self._take(("multinomial",), shape)

    def exponential(self, *shape):
        """Exp(1) variates of a single-sample multinomial (argmax(p / e))."""
        return self._take(("exponential",), shape)

    def randint_array(self, high, n):
        """n integers in [0, high) drawn by ONE randint call of the reference (a size-1 call may
        have been merged into a 'randint_block' by compact_tape)."""
        kind, arr = self.entries[self.pos]
        if kind == "randint" and arr.size == n and not (n == 1 and kind == "randint_block"):
            out = self._take(("randint",), (n,)).to(torch.int64)
        else:
            out = torch.tensor([self.randint(high) for _ in range(n)], dtype=torch.int64)
        assert bool(((out >= 0) & (out < high)).all())
        return out

    def randint(self, high):
        """One integer in [0, high) per call, as the reference's history queue draws them.
        Consecutive single draws may be stored merged as one 'randint_block' entry."""
        kind, arr = self.entries[self.pos]
        if kind == "randint_block":
            off = getattr(self, "_blk_off", 0)
            v = int(arr[off])
            off += 1
            if off == len(arr):
                self.pos += 1
                off = 0
            self._blk_off = off
        else:
            v = int(self._take(("randint",), (1,))[0])
        assert 0 <= v < high
        return v


def compact_tape(entries):
    """Merge runs of single-integer 'randint' entries into 'randint_block' entries."""
    out = []
    for kind, arr in entries:
        if kind == "randint" and arr.size == 1:
            if out and out[-1][0] == "randint_block":
                out[-1][1].append(int(arr.reshape(-1)[0]))
            else:
                out.append(["randint_block", [int(arr.reshape(-1)[0])]])
        else:
            out.append([kind, arr])
    return [(k, np.asarray(a, dtype=np.int64) if k == "randint_block" else a) for k, a in out]


def tape_to_npz_dict(entries, prefix="tape"):
    d = {f"{prefix}_n": np.int64(len(entries))}
    for i, (kind, arr) in enumerate(entries):
        d[f"{prefix}_{i:04d}_{kind}"] = arr
    return d


def tape_from_npz_dict(d, prefix="tape"):
    n = int(d[f"{prefix}_n"])
    keys = sorted(k for k in d if k.startswith(prefix + "_") and k != f"{prefix}_n")
    assert len(keys) == n
    return [(k[len(prefix) + 6:], d[k]) for k in keys]


class SeededSource:
    """Fresh draws from a torch generator (same interface as TapeSource)."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)

    def normal(self, *shape):
        return torch.randn(*shape, generator=self.g)

    def uniform(self, *shape):
        return torch.rand(*shape, generator=self.g)

    def uniform64(self, *shape):
        return torch.rand(*shape, generator=self.g, dtype=torch.float64)

    def exponential(self, *shape):
        return torch.empty(*shape).exponential_(1, generator=self.g)

    def randint(self, high):
        return int(torch.randint(0, int(high), (1,), generator=self.g)[0])

    def randint_array(self, high, n):
        return torch.randint(0, int(high), (n,), generator=self.g)


def initial_imputation(dec, X, M, src, gen_dist="normal_with_eps", prior="std_normal"):
    B, K, D = X.shape
    Zd = O.net_input_dim(dec)
    draws = {"z": src.uniform(B * K, Zd) if prior == "uniform" else src.normal(B * K, Zd), **x_draws(src, gen_dist, B * K, D)}
    return O.vae_samples_imputation(dec, X, M, draws, gen_dist=gen_dist, prior=prior)


def _lair_draws(src, I, B, K, R, D, Zd, *, scheme, method, cluster, gen_dist, second, latent_dist="normal",
                K_out=None, pad_to=None):
    """Draws of one LAIR iteration in the reference's order.  K counts every chain that proposes
    (historical ones included), K_out the chains that are imputed, pad_to the padded proposal
    count of glr before the history is used."""
    Kp = K + R
    d = {"eps_z": src.beta(I, B, Kp, Zd) if latent_dist == "beta" else src.normal(I, B, Kp, Zd)}
    G = (pad_to or Kp) // cluster if scheme == "glr" else 1
    K_out = K if K_out is None else K_out

    def res():
        if scheme in ("lr", "ir"):
            return src.exponential(B * Kp, I) if method == "multinomial" else src.uniform(B, Kp, 1)
        if method == "multinomial":
            return src.uniform64(B * G, (pad_to or Kp) // G)
        return src.uniform(B, 1) if scheme == "gr" else src.uniform(B, G, 1)

    d["u_res"] = res()
    if second:
        d["u_res2"] = res()
    d.update(x_draws(src, gen_dist, B, K_out, D))
    return d


def lair_sampler(enc, dec, X, M, src, *, num_iterations, num_imp_samples=1, weighting_scheme="dmis",
                 resampling_scheme="gr", resampling_method="multinomial", glr_cluster_size=None,
                 num_prior_replenish_proposals=0, prior_replenish_iteration_frequency=1, store_latents=False,
                 clip_imputations=False, weight_transform=None, latent_dist="normal_with_eps",
                 gen_dist="normal_with_eps", collect_proposals=False, prior="std_normal",
                 var_proposal_temperature=None, var_proposal_anneal_type_to_prior=None,
                 num_pseudo_warmup_iterations=0, num_historical_proposals=None,
                 no_historic_proposals_iterations=None, imputation_history_length=None):
    """Generator over LAIR iterations; yields the dict the reference yields plus, when
    ``collect_proposals``, 'Z_prop' and 'log_weights' for the final resampling.  With
    ``num_historical_proposals`` every imputation is stored in a history ring and, once
    ``no_historic_proposals_iterations`` have passed, H stratified samples of it join the proposals
    (:1866-1870, 1881-1882, 1894-1895, 1961-1962)."""
    B, K, D = X.shape
    Zd = O.net_input_dim(dec)
    H = num_historical_proposals
    history = O.ImputationHistory(imputation_history_length, X.shape, X.dtype) if H is not None else None
    for t in range(num_pseudo_warmup_iterations):
        draws = {"eps_z": src.normal(B, K, Zd), **x_draws(src, gen_dist, B, K, D)}
        X, Zpg = O.pseudo_gibbs_iteration(enc, dec, X, M, draws, latent_dist=latent_dist, gen_dist=gen_dist)
        if history is not None:
            history.enqueue(X)
        yield {"X": X, "Z": Zpg, "is_ess": 0.0}
    if clip_imputations:
        Xt = X[:, 0].clone() * M.squeeze(1)
        clip_max = torch.max(Xt, dim=0)[0] * 2
        clip_min = torch.min(Xt, dim=0)[0] * 2
    for t in range(num_iterations):
        R = num_prior_replenish_proposals if t % prior_replenish_iteration_frequency == 0 else 0
        alpha = None
        if weight_transform == "poly":
            alpha = min((t + 10) / num_iterations, 1)
        use_hist = (no_historic_proposals_iterations is not None
                    and (num_pseudo_warmup_iterations + t) > no_historic_proposals_iterations)
        hist_X, pad_to = None, None
        if use_hist:
            r = src.randint_array(history.window(H), H)
            if resampling_scheme == "glr":
                hist_X = history.sample_with_chain_idx(r, O.glr_history_chain_idx(K, H, glr_cluster_size), H)
            else:
                hist_X = history.sample(r, src.randint_array(K, B), H)
            hist_X = hist_X.permute(1, 0, 2).to(X.dtype)
        elif no_historic_proposals_iterations is not None and resampling_scheme == "glr" and H is not None:
            pad_to = K + H                                    # vae_imputation.py:546-548
        Kh = K + (H if use_hist else 0)
        draws = _lair_draws(src, num_imp_samples, B, Kh, R, D, Zd, scheme=resampling_scheme,
                            method=resampling_method, cluster=glr_cluster_size, gen_dist=gen_dist,
                            second=store_latents, latent_dist=latent_dist, K_out=K, pad_to=pad_to)
        o = O.lair_iteration(enc, dec, X, M, draws, num_imp_samples=num_imp_samples, weighting=weighting_scheme, prior=prior,
                             hist_X=hist_X, glr_pad_to=pad_to,
                             resampling_scheme=resampling_scheme, resampling_method=resampling_method,
                             glr_cluster_size=glr_cluster_size, num_prior_replenish_proposals=R,
                             latent_dist=latent_dist, gen_dist=gen_dist, return_resampled_latents=store_latents,
                             weight_alpha=alpha, var_proposal_temperature=var_proposal_temperature,
                             var_proposal_anneal_type_to_prior=var_proposal_anneal_type_to_prior)
        X_m = o["X_new"]
        if clip_imputations:
            X_m = torch.min(torch.max(X_m, clip_min), clip_max)
            bad = ~X_m.isfinite()
            X_m = torch.where(bad, X, X_m)
        X = X_m
        if history is not None:
            history.enqueue(X)
        out = {"X": X, "Z": o["latents_resampled"], "is_ess": o["ess"].mean(), "is_ess_not_reduced": o["ess"],
               "transformed_ess": o["transformed_ess"].mean(), "perplexity": o["perplexity"].mean(),
               "total_num_props": o["total_num_props"], "acceptance_x": torch.tensor(1.)}
        if collect_proposals:
            out["Z_prop"] = o["Z_imp"]
            out["log_weights"] = o["log_weights"]
        yield out


def lair_final_resample(dec, Z_props, log_ws, X_true, M, src, *, num_samples=None, gen_dist="normal_with_eps"):
    """test_step_vae_sampling.py:1991-2032: resample n = T*K' (or num_samples) latents out of all
    stored proposals, decode them in chunks of K', sample, and set the observed values (K' is
    the proposal count per iteration, replenishing proposals included, as the reference reads
    it off the stored tensors).  Z_props / log_ws: lists over iterations of [I,B,K',Zd] /
    [I,B,K'] (equal K' required)."""
    T = len(Z_props)
    B, Kp = Z_props[0].shape[1], Z_props[0].shape[2]
    n = T * Kp if num_samples is None else num_samples
    lp = torch.stack(Z_props)   # T I B K' Zd
    lw = torch.stack(log_ws)
    u = src.uniform64(B, n)
    idx, Z = O.final_resample(lp, lw, u, n)
    Xz = X_true.clone().unsqueeze(1)
    Xz[~M] = 0.0
    imps = []
    for c0 in range(0, n, Kp):
        Zb = Z[:, c0:c0 + Kp]
        params = O.resfc_forward(dec, Zb)
        D = Xz.shape[-1]
        dx = x_draws(src, gen_dist, B, Zb.shape[1], D)
        X_m = O.gen_sample(params, gen_dist, dx["eps_x"], dx.get("chi2_x"))
        imps.append(Xz * M + X_m * (~M))
    return idx, Z, torch.cat(imps, dim=1)


def acmwg_sampler(enc, dec, X, M, src, *, num_pseudo_warmup_iterations, num_mwg_iterations,
                  prior_mixture_probability, warmup_method="pseudo_gibbs", num_k_per_cluster=None,
                  glr_cluster_size=None, num_prior_replenish_proposals=0, num_imp_samples=1,
                  weighting_scheme="dmis_within_groups", resampling_scheme="glr",
                  resampling_method="multinomial", enqueue_allwarmup_samples_to_history=False,
                  latent_dist="normal_with_eps", gen_dist="normal_with_eps", ablation_use_mwg_target=False,
                  plain_mwg=False, clip_imputations_during_pg_warmup=False,
                  var_proposal_temperature=None, var_proposal_anneal_type_to_prior=None):
    """Generator over AC-MWG iterations (warm-up iterations included), yielding
    {'X','Z' (absent during the irwg warm-up),'acceptance'} like the reference; MWG steps also
    carry 'accepted', 'log_accept' and 'slots' for the parity tests.

    ``plain_mwg`` restates TestVAEMetropolisWithinGibbs.sampler instead (:1090-1282): same
    warm-ups, no history and no prior mixture, the proposal conditions on the current
    imputations and the target is the joint of latents and missing values (log-likelihoods over
    all dimensions)."""
    B, K, D = X.shape
    Zd = O.net_input_dim(dec)
    W = num_pseudo_warmup_iterations
    tkw = dict(var_proposal_temperature=var_proposal_temperature,
               var_proposal_anneal_type_to_prior=var_proposal_anneal_type_to_prior)
    if warmup_method == "irwg":
        max_it = W * num_k_per_cluster + num_mwg_iterations
    else:
        max_it = W + num_mwg_iterations
    hist = O.RestrictedHistory(max_it, X.shape, X.dtype)
    if plain_mwg:
        assert prior_mixture_probability == 0.0
    clip_min = clip_max = None
    if clip_imputations_during_pg_warmup:  # clipping_mode 'batch_data' (:1097-1102)
        Xt = X[:, 0].clone() * M.squeeze(1)
        clip_max = torch.max(Xt, dim=0)[0] * 2
        clip_min = torch.min(Xt, dim=0)[0] * 2

    def propose_from(Xc):
        q_loc, q_scale = O.split_loc_scale(O.resfc_forward(enc, Xc), latent_dist)
        Z = q_loc + q_scale * src.normal(B, K, Zd)
        return Z, O.resfc_forward(dec, Z)

    if warmup_method == "none":
        hist.enqueue(X)
        Z_new, mis_params = propose_from(X)
    elif warmup_method == "pseudo_gibbs":
        for t in range(W):
            draws = {"eps_z": src.normal(B, K, Zd),
                     **x_draws(src, gen_dist, B, K, D)}
            q_loc, q_scale = O.split_loc_scale(O.resfc_forward(enc, X), latent_dist)
            Z_new = q_loc + q_scale * draws["eps_z"]
            mis_params = O.resfc_forward(dec, Z_new)
            X_m = O.gen_sample(mis_params, gen_dist, draws["eps_x"], draws.get("chi2_x"))
            if clip_imputations_during_pg_warmup:
                X_m = torch.min(torch.max(X_m, clip_min), clip_max)
                X_m = torch.where(~X_m.isfinite(), X, X_m)
            X = X * M + X_m * (~M)
            if enqueue_allwarmup_samples_to_history or t == W - 1:
                hist.enqueue(X)
            yield {"X": X, "Z": Z_new, "acceptance": 1.0}
    elif warmup_method == "irwg":
        c, R = glr_cluster_size, num_prior_replenish_proposals
        Kw = K * num_k_per_cluster
        Xw = X[:, :1].expand(B, Kw, D).contiguous()
        Xw = initial_imputation(dec, Xw, M, src, gen_dist)
        for t in range(W):
            draws = _lair_draws(src, num_imp_samples, B, Kw, R, D, Zd, scheme=resampling_scheme,
                                method=resampling_method, cluster=c, gen_dist=gen_dist, second=False)
            Xw = O.lair_iteration(enc, dec, Xw, M, draws, num_imp_samples=num_imp_samples,
                                  weighting=weighting_scheme, resampling_scheme=resampling_scheme,
                                  resampling_method=resampling_method, glr_cluster_size=c,
                                  num_prior_replenish_proposals=R, latent_dist=latent_dist,
                                  gen_dist=gen_dist, **tkw)["X_new"]
            # 'b (c k) -> b k c' with c = glr_cluster_size (:1611): k' = cc*(Kw/c) + kk
            Xt = Xw.reshape(B, c, Kw // c, D).permute(0, 2, 1, 3).contiguous()   # b k c d
            if R > 0:
                need = K - Xt.shape[1]
                Xt = torch.cat([Xt, Xt[:, :need]], dim=1)
            for i in range(num_k_per_cluster):
                Xi = Xt[:, :, i]
                if enqueue_allwarmup_samples_to_history or t == W - 1:
                    hist.enqueue(Xi)
                yield {"X": Xi, "acceptance": 1.0}
            if R > 0 and t == W - 1:
                # 'b k c -> b (c k)' on the padded tensor (:1628)
                Xw = Xt.permute(0, 2, 1, 3).reshape(B, c * Xt.shape[1], D)
        Xc = Xw.reshape(B, c, Xw.shape[1] // c, D).permute(0, 2, 1, 3)[:, :, 0].contiguous()
        X = Xc
        Z_new, mis_params = propose_from(X)
    else:
        raise NotImplementedError(warmup_method)

    x_lp = O.gen_log_prob(X, mis_params, gen_dist)
    state = dict(X=X, Z_old=Z_new, mis_params_old=mis_params,
                 Z_old_prior_logprob=O.std_normal_log_prob(Z_new).sum(-1),
                 X_old_logprob=(x_lp if (plain_mwg or ablation_use_mwg_target) else x_lp * M).sum(-1))  # :1216-1217, :1657-1660
    if plain_mwg:
        for t in range(num_mwg_iterations):
            draws = {"eps_z": src.normal(B, K, Zd), "u_accept": src.uniform(B, K),
                     **x_draws(src, gen_dist, B, K, D)}
            o = O.acmwg_step(enc, dec, state, M, state["X"], draws, prior_mixture_probability=0.0,
                             latent_dist=latent_dist, gen_dist=gen_dist, joint_target_new=True, joint_target_old=True,
                             **tkw)
            state = o["state"]
            yield {"X": state["X"], "Z": state["Z_old"], "acceptance": o["acceptance"], "accepted": o["accepted"],
                   "log_accept": o["log_accept"]}
        return
    hist.update_available(torch.ones(B, K, dtype=torch.bool))
    for t in range(num_mwg_iterations):
        bounds = hist.draw_bounds()
        r = torch.empty(B, K, dtype=torch.int64)
        for b in range(B):
            for k in range(K):
                r[b, k] = src.randint(int(bounds[b, k]))
        slots = hist.slots_from_draws(r)
        hist_X = hist.gather(slots).to(X.dtype)
        draws = {"eps_z": src.normal(B, K, Zd)}
        if prior_mixture_probability > 0.0:
            draws["z_prior"] = src.normal(B, K, Zd)
            draws["u_prior"] = src.uniform(B, K)
        draws["u_accept"] = src.uniform(B, K)
        draws.update(x_draws(src, gen_dist, B, K, D))
        o = O.acmwg_step(enc, dec, state, M, hist_X, draws, prior_mixture_probability=prior_mixture_probability,
                         latent_dist=latent_dist, gen_dist=gen_dist, joint_target_new=ablation_use_mwg_target, **tkw)
        hist.update_available(o["accepted"])
        state = o["state"]
        hist.enqueue(state["X"])
        yield {"X": state["X"], "Z": state["Z_old"], "acceptance": o["acceptance"], "accepted": o["accepted"],
               "log_accept": o["log_accept"], "slots": slots, "r": r}


def pseudo_gibbs_sampler(enc, dec, X, M, src, *, num_iterations, clip_imputations=False,
                         latent_dist="normal_with_eps", gen_dist="normal_with_eps"):
    """TestVAEPseudoGibbs.sampler (test_step_vae_sampling.py:1023-1042), clipping_mode
    'batch_data'.  Yields {'X','Z'}."""
    B, K, D = X.shape
    Zd = O.net_input_dim(dec)
    clip_min = clip_max = None
    if clip_imputations:
        Xt = X[:, 0].clone() * M.squeeze(1)
        clip_max = torch.max(Xt, dim=0)[0] * 2
        clip_min = torch.min(Xt, dim=0)[0] * 2
    for t in range(num_iterations):
        draws = {"eps_z": src.normal(B, K, Zd),
                 **x_draws(src, gen_dist, B, K, D)}
        X, Z = O.pseudo_gibbs_iteration(enc, dec, X, M, draws, latent_dist=latent_dist, gen_dist=gen_dist,
                                        clip_min=clip_min, clip_max=clip_max)
        yield {"X": X, "Z": Z}


def ir_sampler(dec, X, M, src, *, num_iterations, num_imp_samples, weighting_scheme="dmis", resampling_scheme="gr",
               resampling_method="multinomial", glr_cluster_size=None, gen_dist="normal_with_eps"):
    """TestVAEImportanceResampling.sampler (test_step_vae_sampling.py:2105-2131): importance
    resampling with the prior as the proposal.  Yields the reference's dict plus 'Z_prop' and
    'log_weights' (what the class stores for its final resampling)."""
    B, K, D = X.shape
    Zd = O.net_input_dim(dec)
    for t in range(num_iterations):
        draws = _lair_draws(src, num_imp_samples, B, K, 0, D, Zd, scheme=resampling_scheme, method=resampling_method,
                            cluster=glr_cluster_size, gen_dist=gen_dist, second=False)
        o = O.lair_iteration(None, dec, X, M, draws, num_imp_samples=num_imp_samples, weighting=weighting_scheme,
                             resampling_scheme=resampling_scheme, resampling_method=resampling_method,
                             glr_cluster_size=glr_cluster_size, gen_dist=gen_dist, prior_proposals=True)
        X = o["X_new"]
        yield {"X": X, "is_ess_standard": o["ess"].mean(), "is_ess_not_reduced": o["ess"],
               "perplexity_standard": o["perplexity"].mean(), "Z_prop": o["Z_imp"], "log_weights": o["log_weights"]}
