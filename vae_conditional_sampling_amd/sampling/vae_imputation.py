"""Per-iteration functions of the samplers with the reference's signatures, computed by the
HIP library.

Counterparts of ``irwg/sampling/vae_imputation.py``:
``importance_resampling_gibbs_iteration`` (:484-726), ``standard_importance_resampling``
(:904-998), ``pseudo_gibbs_iteration`` (:19-45), ``expand_M_dim`` (:14-17).  Two keyword-only extensions exist on every stochastic function:
``rng`` (a ``_lib.cs_rng`` counter-RNG key) and ``draws`` (dict of injected draws, used by
the parity tests to feed this engine and the reference the same numbers).
"""
import torch

from .. import _lib
from . import engine


def expand_M_dim(M, *, data_channel_dim=None):
    if data_channel_dim is not None:
        return M.unsqueeze(data_channel_dim)
    return M


def _check_common(miss_model, accept_reject_xm, data_channel_dim):
    if miss_model is not None or accept_reject_xm:
        raise NotImplementedError("missingness models (MNAR) are not implemented")
    if data_channel_dim is not None:
        raise NotImplementedError("data_channel_dim is not implemented")


def _fresh(out):
    """The reference returns new tensors from every call; the kernels write into the model's per-shape workspace.
    Without ``inplace=True`` (the sampler classes' fast path: views of the workspace, overwritten by the next call of
    the same shape) the workspace tensors are therefore copied.  ``out[0]`` is already the caller's own tensor."""
    return (out[0],) + tuple(t.clone() if torch.is_tensor(t) else t for t in out[1:])


def _weight_alpha(weight_transform):
    """The multiplier of a log-weight transform.  The reference passes a callable, called as
    ``weight_transform(log_weights=...)`` (vae_imputation.py:643-644); the only transform it has multiplies the
    log-weights by a number (``polynomial_transform``, test_step_vae_sampling.py:2059-2061), and that is what the
    resampling kernel applies.  A callable is therefore probed on a few values and accepted when it is such a
    multiplication; anything else raises.  ``None`` and plain numbers pass through."""
    if weight_transform is None or isinstance(weight_transform, (int, float)):
        return weight_transform
    if callable(weight_transform):
        probe = torch.tensor([0.0, 1.0, -2.0, 3.5, -700.0])
        out = torch.as_tensor(weight_transform(log_weights=probe), dtype=torch.float32).reshape(-1)
        alpha = float(out[1])
        if out.numel() == probe.numel() and torch.allclose(out, probe * alpha, rtol=1e-6, atol=0.0):
            return alpha
        raise NotImplementedError("weight_transform: only multiplication of the log-weights by a number is built")
    raise NotImplementedError("weight_transform must be None, a number or a callable")


def importance_resampling_gibbs_iteration(model, X, M, num_imp_samples, *, glr_cluster_size=None,
                                          weighting='dmis', resampling_method='multinomial', resampling_scheme='gr',
                                          num_prior_replenish_proposals=0,
                                          weight_transform=None,
                                          use_historical_proposals=None,
                                          num_historical_proposals=None,
                                          history=None,
                                          miss_model=None,
                                          accept_reject_xm=False,
                                          data_channel_dim=None,
                                          var_proposal_temperature=None,
                                          var_proposal_anneal_type_to_prior=None,
                                          return_resampled_latents=False,
                                          rng=None, draws=None, clip_min=None, clip_max=None, inplace=False):
    """One LAIR iteration.  Returns the reference's 9-tuple (10 with resampled latents):
    ``X_new[B,K,D], ess[B], transformed_ess[B], perplexity[B], total_num_props,
    norm_log_weights[B,I*K'], Z_imp[I,B,K',Zd], log_weights[I,B,K'], acceptance_x
    (, latents_resampled[B,K',Zd])``.

    ``weight_transform`` may be ``None``, a float alpha, or -- as the reference passes it -- a callable
    ``f(log_weights=...)`` that multiplies the log-weights by a number (the 'poly' transform of the sampler class):
    the log-weights are multiplied by it before resampling.  As in the reference every call returns new tensors;
    with ``inplace=True`` (an extension: ``X`` is updated in place) the statistics, proposals and log-weights are views
    of the model's workspace instead, overwritten by the next call of the same shape."""
    _check_common(miss_model, accept_reject_xm, data_channel_dim)
    _lib.proposal_transform_args(var_proposal_temperature, var_proposal_anneal_type_to_prior)
    if weighting not in ("dmis", "dmis_within_groups", "smis"):
        raise NotImplementedError()
    if resampling_scheme not in ("gr", "glr", "lr", "ir"):
        raise NotImplementedError()
    if resampling_method not in ("multinomial", "systematic"):
        raise NotImplementedError()
    weight_transform = _weight_alpha(weight_transform)
    if weighting == "dmis_within_groups" or resampling_scheme == "glr":
        assert glr_cluster_size is not None and glr_cluster_size > 1
    B, K, D = X.shape
    R = int(num_prior_replenish_proposals)
    X_new = X if inplace else X.clone()
    if not X_new.is_contiguous():
        X_new = X_new.contiguous()
    H, pad_to, X_in = 0, 0, X_new
    if use_historical_proposals is not None and use_historical_proposals:
        # H stratified samples of the history join the K chains as proposals (:504-545)
        from .utils import glr_history_chain_idx
        H = int(num_historical_proposals)
        d = draws or {}
        w = history.window_size(H)
        if d.get("hist_r") is not None:
            r = [int(v) for v in d["hist_r"]]
        else:  # common to the whole batch, like the reference's single randint call
            import random
            seed = ((rng.seed if rng is not None else 0) << 20) ^ (rng.iteration if rng is not None else 0)
            r = [random.Random(seed * 64 + h).randrange(w) for h in range(H)]
        slots = history.slots_from_window_draws(r, H)
        hist_ws = model.__dict__.setdefault("_cs_hist_concat", {})
        key = (B, K, H, D, str(X.device))
        if key not in hist_ws:
            hist_ws.clear()
            hist_ws[key] = torch.empty(B, K + H, D, device=X.device, dtype=torch.float32)
        if resampling_scheme == 'glr':
            X_in = history.concat_history(X_new, slots, chain_of_proposal=glr_history_chain_idx(K, H, glr_cluster_size),
                                          out=hist_ws[key])
        else:
            X_in = history.concat_history(X_new, slots, chain_of_example=d.get("hist_chain"), rng=rng, out=hist_ws[key])
    elif use_historical_proposals is not None and resampling_scheme == 'glr' and num_historical_proposals is not None:
        pad_to = K + int(num_historical_proposals)   # :546-548
    ws = engine.lair_workspace(model, B, K + H, R, int(num_imp_samples), X.device, slots=pad_to or None)
    cfg = engine.lair_cfg(model, ws, weighting=weighting, scheme=resampling_scheme, method=resampling_method,
                          cluster_size=glr_cluster_size, weight_alpha=weight_transform,
                          temperature=var_proposal_temperature, anneal_type=var_proposal_anneal_type_to_prior)
    cfg.K_impute = K if H else 0
    cfg.glr_pad_to = pad_to
    engine.lair_step(model, ws, X_in, engine.mask_u8(M), cfg, rng=rng, draws=draws,
                     return_resampled_latents=return_resampled_latents, clip_min=clip_min, clip_max=clip_max,
                     X_out=X_new if H else None)
    acceptance_x = torch.tensor(1.)
    out = (X_new, ws.stats[0], ws.stats[1], ws.stats[2], ws.I * ws.Kp, ws.norm_log_w, ws.Z, ws.log_w, acceptance_x)
    if return_resampled_latents:
        out = out + (ws.latents_resampled,)
    return out if inplace else _fresh(out)


def standard_importance_resampling(model, X, M, num_imp_samples, *, glr_cluster_size=None,
                                   weighting='dmis', resampling_method='multinomial', resampling_scheme='gr',
                                   return_weights=False, weight_transform=None, data_channel_dim=None,
                                   rng=None, draws=None, inplace=False):
    """Importance resampling with the prior as the proposal of every chain (vae_imputation.py:904-998).
    Returns the reference's 8-tuple ``X_new, ess, transformed_ess, perplexity, total_num_props,
    norm_log_weights, Z_imp[I,B,K,Zd], log_weights[I,B,K]`` (new tensors; views of the model's workspace with
    ``inplace=True``).
    With the uniform prior the injected 'eps_z' holds the samples of Uniform(0, 1 + 1e-8)."""
    if data_channel_dim is not None:
        raise NotImplementedError("data_channel_dim is not implemented")
    if weighting not in ("dmis", "smis"):
        raise NotImplementedError()
    if resampling_scheme not in ("gr", "glr", "lr", "ir"):
        raise NotImplementedError()
    if resampling_method not in ("multinomial", "systematic"):
        raise NotImplementedError()
    weight_transform = _weight_alpha(weight_transform)
    if resampling_scheme == "glr":
        assert glr_cluster_size is not None and glr_cluster_size > 1
    B, K, D = X.shape
    ws = engine.lair_workspace(model, B, K, 0, int(num_imp_samples), X.device)
    cfg = engine.lair_cfg(model, ws, weighting=weighting, scheme=resampling_scheme, method=resampling_method,
                          cluster_size=glr_cluster_size, weight_alpha=weight_transform)
    if (cfg.prior == _lib.CS_PRIOR["uniform"]) != (cfg.latent_dist == _lib.CS_DIST["beta"]):
        # the uniform prior is proposed as Beta(1, 1), which needs the Beta posterior's kernels
        raise NotImplementedError("prior proposals: uniform prior with a non-Beta posterior (or the reverse)")
    X_new = (X if inplace else X.clone()).contiguous()
    engine.lair_step(model, ws, X_new, engine.mask_u8(M), cfg, rng=rng, draws=draws, prior_proposals=True)
    out = (X_new, ws.stats[0], ws.stats[1], ws.stats[2], ws.I * ws.Kp, ws.norm_log_w, ws.Z, ws.log_w)
    return out if inplace else _fresh(out)


def pseudo_gibbs_iteration(model, X, M, clip_values_min=None, clip_values_max=None, *, data_channel_dim=None,
                           rng=None, draws=None, inplace=False):
    """z ~ q(z|x), x_m ~ p(x_m|z), impute (vae_imputation.py:19-45).  Returns ``(X_new, Z)``.
    Non-finite x_m are replaced by the previous value, as in the reference."""
    if data_channel_dim is not None:
        raise NotImplementedError("data_channel_dim is not implemented")
    import ctypes as C
    draws = draws or {}
    B, K, D = X.shape
    ws = engine.lair_workspace(model, B, K, 0, 1, X.device)
    cfg = engine.lair_cfg(model, ws, weighting="smis", scheme="gr", method="multinomial", cluster_size=0,
                          weight_alpha=None)
    X_new = (X if inplace else X.clone()).contiguous()
    M8 = engine.mask_u8(M)
    eps_z = draws["eps_z"].to(X.device, torch.float32).contiguous() if draws.get("eps_z") is not None else None
    bufs = _lib.make_bufs(X=X_new, M=M8, eps_z=eps_z, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll)
    rng = rng if rng is not None else _lib.make_rng(0)
    L = _lib.lib()
    st = _lib.current_stream()
    _lib.check(L.cs_lair_encode(model.var_latent_network.packed().handle, C.byref(cfg), C.byref(bufs), st))
    _lib.check(L.cs_lair_decode(model.generator_network.packed().handle, C.byref(cfg), C.byref(bufs), C.byref(rng), st))
    engine.sample_impute(model, ws.gen, X_new, M8, K, eps=draws.get("eps_x"), chi2=draws.get("chi2_x"), rng=rng, stream_id=_lib.STREAM_X,
                         clip_min=clip_values_min, clip_max=clip_values_max, nonfinite_keep=True)
    return (X_new, ws.Z[0]) if inplace else (X_new, ws.Z[0].clone())
