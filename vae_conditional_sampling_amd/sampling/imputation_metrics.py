"""Per-iteration imputation diagnostics with the reference's function names, on the device.

Mirrors ``imputation_rmse_metric`` (irwg/sampling/imputation_metrics.py:85-102),
``imputation_mae_metric`` (:133-143) and ``imputation_latent_wass`` (:200-213).  Inputs are the
harness's tensors: ``X_imp`` [B,K,D], ``X_true`` [B,1,D] (or [B,D]), ``M`` bool [B,1,D]
(True = observed); results are [B,K] float32 device tensors.  An ``out=`` tensor lets the harness
write straight into its per-iteration accumulation buffer.
"""
import ctypes as C

import torch

from .. import _lib
from . import engine


class PreparedTruth:
    """The per-batch constants of the diagnostics, prepared once: X_true [B,D] as it is (a truth that is NaN
    at missing entries, as in `nan_miss` data, gives NaN errors exactly as the reference's functions do), the
    mask as bytes [B,D], and (lazily) the posterior of the true data for the latent distance."""

    def __init__(self, X_true, M, like):
        B, K, D = like.shape
        self.Xt = X_true.reshape(B, D).to(like.device, torch.float32).contiguous()
        self.M8 = engine.mask_u8(M)
        self.q_true = None


def imputation_errors(X_imp, truth, *, rmse=None, mae=None):
    """RMSE and / or MAE of one iteration in one launch, written into the given [B,K] tensors."""
    B, K, D = X_imp.shape
    _lib.check(_lib.lib().cs_imputation_errors(_lib.ptr(X_imp.contiguous()), _lib.ptr(truth.Xt), _lib.ptr(truth.M8), B, K, D,
                                               _lib.ptr(rmse), _lib.ptr(mae), _lib.current_stream()))


def _errors(X_imp, X_true, M, want_rmse, out):
    B, K, D = X_imp.shape
    out = out if out is not None else torch.empty(B, K, device=X_imp.device, dtype=torch.float32)
    imputation_errors(X_imp, PreparedTruth(X_true, M, X_imp), **({"rmse": out} if want_rmse else {"mae": out}))
    return out


def imputation_rmse_metric(X_imp, X_true, M, *, out=None):
    return _errors(X_imp, X_true, M, True, out)


def imputation_mae_metric(X_imp, X_true, M, *, out=None):
    return _errors(X_imp, X_true, M, False, out)


def encode_posterior(vae, X):
    """q[B,K,2,Zd] = (loc, scale) of q(z | x) for X[B,K,D] (VAE.compute_var_distr with an all-ones
    mask, vae.py:461-463)."""
    B, K, D = X.shape
    latent, _ = vae.engine_dists()
    if latent == _lib.CS_DIST["beta"]:
        raise NotImplementedError("latent Wasserstein distance is implemented for Gaussian posteriors only")
    Zd = vae.latent_dim
    q = torch.empty(B, K, 2, Zd, device=X.device, dtype=torch.float32)
    cfg = _lib.cs_lair_cfg(B, K, 0, 1, D, Zd, latent, 0, 0, 0, 0, 0, 1.0, vae.engine_prior())
    bufs = _lib.make_bufs(X=X.contiguous(), q=q)
    _lib.check(_lib.lib().cs_lair_encode(vae.var_latent_network.packed().handle, C.byref(cfg), C.byref(bufs),
                                         _lib.current_stream()))
    return q


def imputation_latent_wass(X_imp, X_true, vae, *, q_true=None, out=None):
    """``q_true`` (from ``encode_posterior`` of the true data) may be passed in: it does not change
    over the iterations of a batch, so the harness encodes it once."""
    B, K, D = X_imp.shape
    if q_true is None:
        q_true = encode_posterior(vae, X_true.reshape(B, 1, D).to(X_imp.device, torch.float32))
    q_imp = encode_posterior(vae, X_imp)
    out = out if out is not None else torch.empty(B, K, device=X_imp.device, dtype=torch.float32)
    _lib.check(_lib.lib().cs_latent_wasserstein(_lib.ptr(q_imp), _lib.ptr(q_true), B, K, vae.latent_dim, _lib.ptr(out),
                                                _lib.current_stream()))
    return out
