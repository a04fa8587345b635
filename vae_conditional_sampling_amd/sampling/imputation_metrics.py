"""Per-iteration imputation diagnostics with the reference's function names, on the device.

Mirrors ``imputation_rmse_metric`` (irwg/sampling/imputation_metrics.py:85-102),
``imputation_mae_metric`` (:133-143) and ``imputation_latent_wass`` (:200-213).  Inputs are the
harness's tensors: ``X_imp`` [B,K,D], ``X_true`` [B,1,D] (or [B,D]), ``M`` bool [B,1,D]
(True = observed); results are [B,K] float32 device tensors.  An ``out=`` tensor lets the harness
write straight into its per-iteration accumulation buffer.
"""
import ctypes as C

import numpy as np
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


def _order_statistics(x, ranks):
    """out[row, j] = the ranks[j]-th smallest value of row `row` of the device matrix x (cs_row_order_statistics)."""
    rows, m = x.shape
    L = _lib.lib()
    nq = len(ranks)
    out = torch.empty(rows, nq, device=x.device, dtype=torch.float32)
    x = x.contiguous()
    for r0 in range(0, rows, 65535):
        r1 = min(rows, r0 + 65535)
        ws = torch.empty(L.cs_row_order_statistics_workspace(r1 - r0, nq), dtype=torch.uint8, device=x.device)
        _lib.check(L.cs_row_order_statistics(_lib.ptr(x[r0:r1]), r1 - r0, m, (C.c_int64 * nq)(*[int(r) for r in ranks]), nq,
                                             _lib.ptr(out[r0:r1]), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    return out


def pointwise_summary(values, quantiles, device=None):
    """Per-iteration summary of a point-wise metric over the whole test set (the aggregation of
    irwg/sampling/test_step_vae_sampling.py:824-858): ``values`` is the list of per-batch series ``[t, n_b, k]``
    (numpy arrays or tensors on any device);
    returns ``(mean[t], std[t], quant[len(quantiles), t])`` where mean / std are taken over the chains k of the mean
    over the data and the quantiles over all (n k) values of an iteration, numpy's default (linear) interpolation.

    The reduction runs in torch on ``device`` (the sampler's GPU in the driver; CPU tensors work the same) over
    chunks of iterations: at a whole test set (1e5 examples x 16 chains x 600 iterations) numpy's per-column
    partition on the host took longer than the sampling itself."""
    T = values[0].shape[0]
    n_total = sum(v.shape[1] for v in values)
    K = values[0].shape[2]
    dev = torch.device(device) if device is not None else torch.device("cpu")
    qs = torch.tensor(list(quantiles), dtype=torch.float64)
    mean = np.empty(T, dtype=np.float32)
    std = np.empty(T, dtype=np.float32)
    quant = np.empty((len(quantiles), T), dtype=np.float32)
    m = n_total * K
    # positions of numpy's linear method: virtual index q (m - 1), neighbours floor / floor + 1
    vidx = qs * (m - 1)
    lo = vidx.floor().clamp(0, m - 1).long()
    hi = (lo + 1).clamp(max=m - 1)
    g = (vidx - lo.double()).to(torch.float32).to(dev)
    lo, hi = lo.to(dev), hi.to(dev)
    step = max(1, (1 << 27) // max(m, 1))          # <= 512 MiB of float32 per chunk
    for t0 in range(0, T, step):
        t1 = min(T, t0 + step)
        x = torch.cat([torch.as_tensor(v[t0:t1]).to(dev, torch.float32) for v in values], dim=1)
        over_data = x.mean(dim=1)                  # [tc, k]
        mean[t0:t1] = over_data.mean(dim=1).cpu().numpy()
        std[t0:t1] = (over_data.std(dim=1, unbiased=True) if K != 1 else torch.ones(t1 - t0, device=dev)).cpu().numpy()
        if len(quantiles):
            flat = x.reshape(t1 - t0, m)
            if flat.is_cuda and 2 * len(quantiles) <= 16:
                # the two neighbours of every quantile by radix selection (cs_row_order_statistics): three reads of
                # the values instead of a sort
                ab = _order_statistics(flat, torch.cat([lo, hi]).tolist())
                a, b = ab[:, :len(quantiles)], ab[:, len(quantiles):]
                has_nan = flat.isnan().any(dim=1, keepdim=True)
            else:
                srt = flat.sort(dim=1).values
                a, b = srt[:, lo], srt[:, hi]       # [tc, q]
                has_nan = srt[:, -1:].isnan()       # the sort puts NaNs last
            d = b - a
            # numpy's _lerp: a + d g, taken from the other end when g >= 0.5
            r = torch.where(g >= 0.5, b - d * (1 - g), a + d * g)
            # np.quantile returns NaN for an iteration that holds a NaN (test_step_vae_sampling.py:847)
            r = torch.where(has_nan, torch.full_like(r, float("nan")), r)
            quant[:, t0:t1] = r.t().cpu().numpy()
    return mean, std, quant
