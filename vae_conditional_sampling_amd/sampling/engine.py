"""Thin host layer between the reference-shaped Python API and the C ABI.

Holds the per-shape device workspaces (so the sampler loop allocates nothing per
iteration) and turns Python arguments into ``cs_lair_cfg`` / ``cs_lair_bufs`` structs.
"""
import ctypes as C

import torch

from .. import _lib


def mask_u8(M):
    """[B,1,D] / [B,D] bool mask -> contiguous uint8 [B,D] (1 = observed)."""
    if M.dim() == 3:
        M = M[:, 0]
    return M.to(torch.uint8).contiguous()


class LairWorkspace:
    """Device buffers of one LAIR configuration (B, K, R, I)."""

    def __init__(self, model, B, K, R, I, device, slots=None):
        self.B, self.K, self.R, self.I = B, K, R, I
        self.Kp = K + R
        self.NS = slots or self.Kp   # output slots of the resampling (glr with a padded layout: more than K')
        self.D = model.generator_network.output_dim // _lib.gen_nparams(model.hparams.generator_distribution.name)
        self.Zd = model.latent_dim
        self.P = model.generator_network.output_dim
        f = dict(device=device, dtype=torch.float32)
        Kp, Zd, D, P = self.Kp, self.Zd, self.D, self.P
        self.q = torch.empty(B, Kp, 2, Zd, **f)
        self.Z = torch.empty(I, B, Kp, Zd, **f)
        self.gen = torch.empty(I, B, Kp, P, **f)
        self.gen_ll = torch.empty(I, B, Kp, **f)
        self.log_w = torch.empty(I, B, Kp, **f)
        self.norm_log_w = torch.empty(B, I * Kp, **f)
        self.idx = torch.empty(B, self.NS, device=device, dtype=torch.int64)
        self.stats = torch.empty(3, B, **f)  # ess, transformed ess, perplexity
        self.latents_resampled = None
        self.idx2 = None

    def want_latents(self):
        if self.latents_resampled is None:
            self.latents_resampled = torch.empty(self.B, self.NS, self.Zd, device=self.q.device, dtype=torch.float32)
            self.idx2 = torch.empty(self.B, self.NS, device=self.q.device, dtype=torch.int64)


def lair_workspace(model, B, K, R, I, device, slots=None):
    cache = model.__dict__.setdefault("_cs_workspaces", {})
    key = ("lair", B, K, R, I, str(device)) if slots is None else ("lair", B, K, R, I, str(device), slots)
    ws = cache.get(key)
    if ws is None:
        if len(cache) > 8:
            cache.clear()
        ws = cache[key] = LairWorkspace(model, B, K, R, I, device, slots)
        # layer-program networks: allocate their activation workspace for this shape now, so that no iteration
        # allocates inside a forward call (a no-op for the fused MLP kernels)
        model.var_latent_network.packed().reserve(B * K)
        model.generator_network.packed().reserve(I * B * (K + R))
    return ws


def lair_cfg(model, ws, *, weighting, scheme, method, cluster_size, weight_alpha, temperature=None, anneal_type=None):
    latent, gen = model.engine_dists()
    try:
        w = _lib.CS_WEIGHTING[weighting]
        s = _lib.CS_SCHEME[scheme]
        m = _lib.CS_METHOD[method]
    except KeyError:
        raise NotImplementedError(f"{weighting=}, {scheme=}, {method=}")
    temp, anneal = _lib.proposal_transform_args(temperature, anneal_type)
    return _lib.cs_lair_cfg(ws.B, ws.K, ws.R, ws.I, ws.D, ws.Zd, latent, gen, w, s, m,
                            int(cluster_size or 0), float(1.0 if weight_alpha is None else weight_alpha),
                            model.engine_prior(), temp, anneal)


def _lair_bufs(ws, X, M8, cfg, draws, return_resampled_latents, clip_min, clip_max, X_out):
    """cs_lair_bufs of one iteration (or, with draws stacked along a leading T axis, of cs_lair_run); the second
    value keeps the converted draw tensors alive."""
    draws = draws or {}
    dev = X.device
    g = lambda k, dt: (draws[k].to(dev, dt).contiguous() if draws.get(k) is not None else None)
    keep = [g("eps_z", torch.float32), g("eps_x", torch.float32)]
    chi2 = g("chi2_x", torch.float32)  # Student-t generators: chi-square draws beside the normals of eps_x
    systematic = cfg.method == _lib.CS_METHOD["systematic"]
    if systematic:
        keep += [g("u_res", torch.float32), g("u_res2", torch.float32)]
        ud = dict(u_sys=keep[2], u_sys2=keep[3])
    elif cfg.scheme in (_lib.CS_SCHEME["lr"], _lib.CS_SCHEME["ir"]):
        # one index per chain: the draws are the Exp(1) variates of torch.multinomial(p, 1)
        keep += [g("u_res", torch.float32), g("u_res2", torch.float32)]
        ud = dict(e_res=keep[2], e_res2=keep[3])
    else:
        keep += [g("u_res", torch.float64), g("u_res2", torch.float64)]
        ud = dict(u_res=keep[2], u_res2=keep[3])
    if return_resampled_latents:
        ws.want_latents()
    bufs = _lib.make_bufs(X=X, M=M8, eps_z=keep[0], eps_x=keep[1], q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll,
                          log_w=ws.log_w, norm_log_w=ws.norm_log_w, idx=ws.idx, ess=ws.stats[0], tess=ws.stats[1],
                          perplexity=ws.stats[2],
                          latents_resampled=ws.latents_resampled if return_resampled_latents else None,
                          idx2=ws.idx2 if return_resampled_latents else None,
                          clip_min=clip_min, clip_max=clip_max, X_out=X_out, chi2_x=chi2, **ud)
    return bufs, keep + [chi2]


def lair_step(model, ws, X, M8, cfg, *, rng=None, draws=None, return_resampled_latents=False,
              clip_min=None, clip_max=None, prior_proposals=False, X_out=None):
    """One LAIR iteration in place on ``X`` [B,K,D] (cs_lair_step).  ``prior_proposals``: every
    chain proposes from the prior instead of the encoder (standard importance resampling)."""
    bufs, keep = _lair_bufs(ws, X, M8, cfg, draws, return_resampled_latents, clip_min, clip_max, X_out)
    rng = rng if rng is not None else _lib.make_rng(0)
    dec = model.generator_network.packed()
    L, st = _lib.lib(), _lib.current_stream()
    if prior_proposals:
        _lib.check(L.cs_prior_proposals(_lib.ptr(ws.q), ws.B, ws.Kp, ws.Zd, cfg.latent_dist, st))
        _lib.check(L.cs_lair_decode(dec.handle, C.byref(cfg), C.byref(bufs), C.byref(rng), st))
        _lib.check(L.cs_lair_resample_impute(C.byref(cfg), C.byref(bufs), C.byref(rng), st))
        return ws
    enc = model.var_latent_network.packed()
    _lib.check(L.cs_lair_step(enc.handle, dec.handle, C.byref(cfg), C.byref(bufs), C.byref(rng), st))
    return ws


class RunObserver:
    """What the harness does with the chains after every iteration, handed to a run of several iterations
    (cs_run_observe): ``X_store`` [T,B,K,D] keeps the imputations, ``rmse`` / ``mae`` / ``wass`` [T,B,K] receive the
    diagnostics.  ``X_true`` [B,D] is needed by rmse / mae, ``q_true`` [B,1,2,Zd] (the posterior of the true data) by
    wass.  ``window(t, n)`` is the observer of iterations t .. t+n-1 of the same stores."""

    def __init__(self, *, X_store=None, X_true=None, rmse=None, mae=None, q_true=None, wass=None):
        self.X_store, self.X_true, self.rmse, self.mae, self.q_true, self.wass = X_store, X_true, rmse, mae, q_true, wass
        self._q_work = None

    def window(self, t, n):
        cut = lambda a: None if a is None else a[t:t + n]
        o = RunObserver(X_store=cut(self.X_store), X_true=self.X_true, rmse=cut(self.rmse), mae=cut(self.mae),
                        q_true=self.q_true, wass=cut(self.wass))
        o._q_work = self._q_work
        return o

    def struct(self, B, K, Zd, device):
        if self.wass is not None and self._q_work is None:
            self._q_work = torch.empty(B, K, 2, Zd, device=device, dtype=torch.float32)
        return _lib.cs_run_observe(_lib.ptr(self.X_store), _lib.ptr(self.X_true), _lib.ptr(self.rmse), _lib.ptr(self.mae),
                                   _lib.ptr(self.q_true), _lib.ptr(self._q_work) if self.wass is not None else None,
                                   _lib.ptr(self.wass))


def lair_run(model, ws, X, M8, cfg, T, *, rng=None, draws=None, Z_store=None, log_w_store=None, stats_store=None,
             clip_min=None, clip_max=None, observe=None):
    """``T`` LAIR iterations in place on ``X`` [B,K,D] in one library call (cs_lair_run_observed): iteration t is
    ``lair_step`` with ``rng.iteration + t``.  ``draws``: the injected draws of all iterations stacked along a
    leading T axis.  ``Z_store`` [T,I,B,K',Zd] / ``log_w_store`` [T,I,B,K'] receive every iteration's proposals and
    log-weights, ``stats_store`` [3,T,B] its ESS, transformed ESS and perplexity; ``observe`` (RunObserver) keeps
    the imputations and computes the diagnostics of every iteration."""
    bufs, keep = _lair_bufs(ws, X, M8, cfg, draws, False, clip_min, clip_max, None)
    rng = rng if rng is not None else _lib.make_rng(0)
    out = _lib.cs_lair_run_out(_lib.ptr(Z_store), _lib.ptr(log_w_store),
                               *(_lib.ptr(stats_store[i]) if stats_store is not None else None for i in range(3)))
    obs = observe.struct(ws.B, ws.K, ws.Zd, X.device) if observe is not None else None
    enc, dec = model.var_latent_network.packed(), model.generator_network.packed()
    _lib.check(_lib.lib().cs_lair_run_observed(enc.handle, dec.handle, C.byref(cfg), C.byref(bufs), C.byref(rng), int(T),
                                               C.byref(out), C.byref(obs) if obs is not None else None,
                                               _lib.current_stream()))
    return ws


def acmwg_run(model, cfg, bufs, T, *, rng=None, accepted_store=None, log_accept_store=None, slots_store=None,
              Z_store=None, observe=None, device=None):
    """``T`` Metropolis-within-Gibbs steps in one library call (cs_acmwg_run): step t is cs_acmwg_step with
    ``rng.iteration + t`` and the history tail advanced t slots.  The stores [T,B,K] (``Z_store`` [T,B,K,Zd]) receive
    every step's outputs; the caller advances its history scalars ``T`` times afterwards."""
    rng = rng if rng is not None else _lib.make_rng(0)
    out = _lib.cs_acmwg_run_out(_lib.ptr(accepted_store), _lib.ptr(log_accept_store), _lib.ptr(slots_store), _lib.ptr(Z_store))
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    obs = observe.struct(cfg.B, cfg.K, cfg.Zd, device) if observe is not None else None
    enc, dec = model.var_latent_network.packed(), model.generator_network.packed()
    _lib.check(_lib.lib().cs_acmwg_run(enc.handle, dec.handle, C.byref(cfg), C.byref(bufs), C.byref(rng), int(T),
                                       C.byref(out), C.byref(obs) if obs is not None else None, _lib.current_stream()))


def decode_params(model, Z):
    """Generator parameterisation (mean|scale or logits) of latents Z[rows,Zd]."""
    _, gen = model.engine_dists()
    Z2 = Z.reshape(-1, Z.shape[-1]).contiguous().float()
    out = torch.empty(Z2.shape[0], model.generator_network.output_dim, device=Z.device, dtype=torch.float32)
    dec = model.generator_network.packed()
    _lib.check(_lib.lib().cs_decode_params(dec.handle, _lib.ptr(Z2), Z2.shape[0], gen, _lib.ptr(out),
                                           _lib.current_stream()))
    return out


def sample_impute(model, gen, X, M8, K, *, eps=None, rng=None, stream_id=_lib.STREAM_X, clip_min=None,
                  clip_max=None, zero_missing=False, nonfinite_keep=False, chi2=None):
    """X[B,K,D] <- X*M + sample(gen)*(1-M) in place (M8 None: X <- sample).  ``chi2``: the injected chi-square draws
    of a Student-t generator (beside the standard normals in ``eps``)."""
    _, gd = model.engine_dists()
    D = X.shape[-1]
    B = X.numel() // (K * D)
    eps = eps.to(X.device, torch.float32).contiguous() if eps is not None else None
    chi2 = chi2.to(X.device, torch.float32).contiguous() if chi2 is not None else None
    rng = rng if rng is not None else _lib.make_rng(0)
    _lib.check(_lib.lib().cs_sample_impute2(_lib.ptr(gen), gd, _lib.ptr(eps), _lib.ptr(chi2), C.byref(rng), stream_id,
                                            _lib.ptr(X), _lib.ptr(M8), B, K, D, _lib.ptr(clip_min), _lib.ptr(clip_max),
                                            int(zero_missing) | (2 if nonfinite_keep else 0),
                                            _lib.current_stream()))
    return X


def rng_normal(rng, stream_id, rows, cols, device):
    out = torch.empty(rows, cols, device=device, dtype=torch.float32)
    _lib.check(_lib.lib().cs_rng_normal(C.byref(rng), stream_id, rows, cols, _lib.ptr(out), _lib.current_stream()))
    return out


def rng_uniform(rng, stream_id, rows, cols, device):
    out = torch.empty(rows, cols, device=device, dtype=torch.float32)
    _lib.check(_lib.lib().cs_rng_uniform(C.byref(rng), stream_id, rows, cols, _lib.ptr(out), _lib.current_stream()))
    return out


def rng_uniform64(rng, stream_id, rows, cols, device):
    out = torch.empty(rows, cols, device=device, dtype=torch.float64)
    _lib.check(_lib.lib().cs_rng_uniform64(C.byref(rng), stream_id, rows, cols, _lib.ptr(out), _lib.current_stream()))
    return out


def vae_sample(model, num_samples, *, rng=None, draws=None, return_latents=False):
    """x ~ p(x|z), z ~ N(0,I) or z = u * (1 + 1e-8), u ~ U[0,1) under the uniform prior
    (irwg/models/vae.py:166-178).  An injected 'z' holds the normals / the uniforms u."""
    draws = draws or {}
    dev = model.device
    rng = rng if rng is not None else _lib.make_rng(0)
    Zd = model.latent_dim
    uniform = bool(model.hparams.use_uniform_prior)
    if draws.get("z") is not None:
        Z = draws["z"].to(dev, torch.float32).contiguous()
    elif uniform:
        Z = rng_uniform(rng, _lib.STREAM_INIT_Z, num_samples, Zd, dev)
    else:
        Z = rng_normal(rng, _lib.STREAM_INIT_Z, num_samples, Zd, dev)
    if uniform:
        Z = Z * (1. + 1e-8)
    gen = decode_params(model, Z)
    D = gen.shape[-1] // _lib.gen_nparams(model.hparams.generator_distribution.name)
    X = torch.empty(num_samples, D, device=dev, dtype=torch.float32)
    sample_impute(model, gen, X, None, 1, eps=draws.get("eps_x"), chi2=draws.get("chi2_x"), rng=rng,
                  stream_id=_lib.STREAM_INIT_X)
    return (X, Z) if return_latents else X
