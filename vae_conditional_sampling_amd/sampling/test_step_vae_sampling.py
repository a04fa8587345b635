"""Sampler classes with the reference's names, constructor arguments and generator protocol,
running on the HIP library.

Counterparts in irwg/sampling/test_step_vae_sampling.py:
  TestVAESamplerBase                                 :102-1000 (the parts LAIR / AC-MWG use)
  TestVAELatentAdaptiveImportanceResampling (LAIR)   :1770-2061
  TestVAEAdaptiveCollapsedMetropolisWithinGibbs      :1457-1768
  TestVAEPseudoGibbs :1002-1042, TestVAEMetropolisWithinGibbs :1045-1282,
  TestVAEImportanceResampling :2063-2194            (baselines on the same kernels)

Extensions that do not exist in the reference (all optional):
  * ``rng_seed`` / ``example_offset`` attributes key the counter RNG (global example index, so a
    sharded run draws the same numbers as an unsharded one);
  * ``draw_source``: an object with ``normal/uniform/uniform64/randint`` methods that supplies
    every random draw in the reference's call order (used by the parity tests to replay a tape
    recorded from the reference);
  * ``out_dir``: where ``.npz`` artefacts go (the reference uses the Lightning logger's dir);
  * ``collector``: when set (sharded runs), the per-batch artefacts are handed to it instead of being
    written: the driver gathers them over the ranks and rank 0 writes the files of a one-process run.
"""
import math
import os
from collections import defaultdict
from typing import Any, List

import numpy as np
import torch

from .. import _lib
from ..utils.artifacts import AsyncNpzWriter
from ..utils.hparams import HyperparametersMixin
from . import engine, imputation_metrics
from .utils import ImputationHistoryQueue, ImputationHistoryQueue_with_restrictedavailability
from .vae_imputation import (importance_resampling_gibbs_iteration, pseudo_gibbs_iteration,
                             standard_importance_resampling)

# diagnostics of the reference's harness that are NOT implemented here (the point-wise RMSE, MAE and
# latent-Wasserstein metrics are: sampling/imputation_metrics.py)
_DIAGNOSTIC_FLAGS = ("compute_f1_metric", "compute_inv_f1_metric", "compute_bin_accuracy_metric",
                     "compute_ssim_metric", "estimate_sampler_ess", "estimate_mmd_with_model",
                     "estimate_mmd_for_full_chain", "estimate_iwelbo", "estimate_kl_div_for_full_chain",
                     "estimate_kl_div", "estimate_fid_chain", "estimate_kid_chain", "estimate_mog_kldivs",
                     "estimate_sampler_variance", "save_img_snapshot_in_tb")


def _check_if_bool(X):
    return bool(((X == 0).sum() + (X == 1).sum()) == X.numel())


def _x_draws(src, model, *shape):
    """Injected draws of one generator sample, in the reference's order: uniforms for Bernoulli, standard normals
    otherwise, and for the Student-t families the chi-square draws behind them (StudentT.rsample)."""
    name = model.hparams.generator_distribution.name
    if name == "bern":
        return {"eps_x": src.uniform(*shape)}
    d = {"eps_x": src.normal(*shape)}
    if name in ("studentt", "studentt_with_eps22"):
        d["chi2_x"] = src.chi2(*shape)
    return d


class TestVAESamplerBase(HyperparametersMixin):
    """Samplers for VAE, base class (harness: chain set-up, initial imputation, loop over the
    sampler generator, storage of the imputation sequence)."""
    __test__ = False  # not a pytest class
    # False for a sampler whose on_test_batch_end writes imputations_{b}.npz itself (the reference's IR
    # class overwrites the harness's file of the same name: only the later content survives)
    _harness_writes_imputations = True

    def __init__(self,
                 store_last_n_iterations: int,
                 store_every_n_iterations: int,

                 imputation_metric_quantiles: List[float],
                 compute_f1_metric: bool,
                 compute_inv_f1_metric: bool,
                 compute_bin_accuracy_metric: bool,
                 compute_rmse_metric: bool,
                 compute_ssim_metric: bool,
                 compute_latent_wass_metric: bool,

                 num_copies: int,
                 imputation_fn: str,

                 store_latents: bool = False,
                 dont_store_imputations: bool = False,

                 sgd_map_init_num_particles: int = None,
                 sgd_map_init_sgd_steps: int = None,
                 sgd_map_init_sgd_lr: float = None,
                 sgd_map_init_use_lr_sched: bool = False,

                 num_sequential_runs: int = 1,

                 save_img_snapshot_in_tb: bool = False,
                 snapshot_image_dims: List[int] = [28, 28],

                 estimate_sampler_ess: bool = False,
                 sampler_ess_estimator_warmup: int = -1,
                 ess_evaluation_batchsize: int = -1,
                 normalise_trajectory_ess_by_is_ess: bool = False,
                 store_is_norm_ess: bool = False,

                 estimate_mmd_with_model: bool = False,
                 mmd_estimator_kernel: str = 'rbf',
                 mmd_estimator_size: int = None,
                 mmd_estimator_frequency: int = None,
                 mmd_avg_of_chains: bool = False,
                 estimate_mmd_for_full_chain: bool = False,
                 estimate_mmd_target_num_repeats: int = 1,
                 mmd_full_chain_linear_estimator_repeats: int = 1,

                 estimate_iwelbo: bool = False,
                 iwelbo_estimator_frequency: int = None,
                 iwelbo_num_importance_samples: int = None,
                 iwelbo_batchsize: int = -1,

                 estimate_kl_div_for_full_chain: bool = False,
                 kl_div_kde_kernel: str = 'rbf',
                 kl_div_iwelbo_num_importance_samples: int = None,
                 estimate_kl_div: bool = False,
                 kl_div_estimator_frequency: int = None,

                 estimate_fid_chain: bool = False,
                 fid_estimator_size: int = None,

                 estimate_kid_chain: bool = False,
                 kid_estimator_kernel: str = None,
                 kid_full_chain_linear_estimator_repeats: int = None,

                 estimate_mog_kldivs: bool = False,
                 num_mog_em_iterations: int = None,
                 num_mog_kl_samples: int = None,
                 mog_kldivs_skip_first_num_imps: int = None,
                 mog_kldivs_use_every_nth: int = 1,
                 mog_em_use_solver: bool = False,
                 estimate_mog_jsd: bool = False,

                 estimate_sampler_variance: bool = False,

                 compute_mae_metric: bool = False,

                 data_channel_dim: int = None,

                 ssim_image_dims: List[int] = [1, 28, 28],
                 ):
        super().__init__()
        self.save_hyperparameters()
        assert not estimate_sampler_ess or sampler_ess_estimator_warmup > 0
        # engine-side state (not in the reference)
        self.rng_seed = 0
        self.example_offset = 0
        self.draw_source = None
        self.out_dir = None
        self.collector = None
        self.skip_unimplemented_diagnostics = False
        self._iteration = 0
        # engine-side: while test_step iterates over the sampler, what it does with every iteration's state
        # (_HarnessPlan; None outside test_step).  A sampler may take that work over and run up to
        # max_iterations_per_call iterations in one library call.
        self._harness = None
        self.max_iterations_per_call = 64

    # ---- wiring -------------------------------------------------------------------------
    def set_model(self, model):
        self.model = model

    def set_datamodule(self, datamodule):
        self.datamodule = datamodule

    @property
    def device(self):
        return self.model.device

    def on_test_start(self):
        enabled = [f for f in _DIAGNOSTIC_FLAGS if self.hparams.get(f)]
        if enabled and not self.skip_unimplemented_diagnostics:
            raise NotImplementedError(f"per-iteration diagnostics are not implemented: {enabled} "
                                      "(set skip_unimplemented_diagnostics=True to run without them)")
        if self.hparams.data_channel_dim is not None:
            raise NotImplementedError("data_channel_dim is not implemented")
        if self.hparams.num_sequential_runs != 1:
            raise NotImplementedError("num_sequential_runs != 1 is not implemented")

    def on_test_epoch_start(self):
        self.metrics = defaultdict(list)
        self._writer = AsyncNpzWriter()

    def _stores_artifacts(self):
        return self.out_dir is not None or self.collector is not None

    def _save_npz(self, name, **arrays):
        """np.savez_compressed(out_dir/name) on a background thread (joined in test_epoch_end); with a
        collector the arrays are kept for the gather over the ranks instead."""
        if self.collector is not None:
            self.collector.add(name, arrays)
            return
        os.makedirs(self.out_dir, exist_ok=True)
        self._write_npz(os.path.join(self.out_dir, name), **arrays)

    def _write_npz(self, path, **arrays):
        writer = getattr(self, "_writer", None)
        if writer is None:
            np.savez_compressed(path, **arrays)
        else:
            writer.submit(path, **arrays)

    def _keep_pointwise(self, series):
        """Where a batch's point-wise diagnostic series [t, n_b, k] waits for the end of the epoch.  A one-process
        run leaves it in device memory, where test_epoch_end reduces it (a whole UCI test set at 16 chains and 600
        iterations is 13 GB for two metrics: moving that to the host and back took a third as long as sampling it);
        it goes to the host as a numpy array when ranks will exchange it, or when the device is short of room."""
        if self.collector is None and series.is_cuda:
            free, _ = torch.cuda.mem_get_info(series.device)
            if free > 8 * series.numel() * series.element_size():
                return series
        return series.cpu().numpy()

    def on_test_batch_start(self, batch: Any, batch_idx: int, dataloader_idx: int = 0) -> None:
        pass

    def on_test_batch_end(self, outputs, batch: Any, batch_idx: int, dataloader_idx: int = 0) -> None:
        pass

    def test_epoch_end(self, outputs):
        """Aggregation of test_step_vae_sampling.py:824-858 without TensorBoard: scalar series are
        averaged over batches weighted by batch size; point-wise metrics [t, n, k] give per-iteration
        mean / std over chains (of the mean over data) and quantiles over (n k).  The summary is
        returned and, when ``out_dir`` is set, saved as ``metrics_summary.npz``."""
        num_X = np.array(self.metrics.get('num_X', []), dtype=np.float64)
        summary = {}
        for key, values in self.metrics.items():
            if key == 'num_X' or len(values) == 0:
                continue
            if key.startswith('imp_metric/'):
                name = key.split('imp_metric/')[1]
                qs = self.hparams.imputation_metric_quantiles
                dev = getattr(getattr(self, "model", None), "device", None)
                mean, std, quant = imputation_metrics.pointwise_summary(values, qs, dev)   # values: [t, n_b, k] per batch
                summary[f'{name}/mean/test'] = mean
                summary[f'{name}/std/test'] = std
                for i, q in enumerate(qs):
                    summary[f'{name}/q{q}/test'] = quant[i]
            else:
                m = np.array(values, dtype=np.float64)
                summary[f'{key}/test'] = (m * num_X[:, None]).sum(axis=0) / num_X.sum()
        self.metrics_summary = summary
        if getattr(self, "_writer", None) is not None:
            self._writer.close()      # every artefact of the epoch is on disk when the epoch ends
            self._writer = None
        if self.out_dir is not None and summary:
            os.makedirs(self.out_dir, exist_ok=True)
            np.savez_compressed(os.path.join(self.out_dir, 'metrics_summary.npz'),
                                **{k.replace('/', '__'): v for k, v in summary.items()})
        return outputs

    # ---- random draws -------------------------------------------------------------------
    def _rng(self, stream_iteration=None):
        it = self._iteration if stream_iteration is None else stream_iteration
        return _lib.make_rng(self.rng_seed, it, self.example_offset)

    def _next_iteration(self):
        self._iteration += 1

    # ---- to be implemented by subclasses ------------------------------------------------
    def sampler(self, X, M):
        raise NotImplementedError()

    @property
    def max_iterations(self):
        raise NotImplementedError()

    # ---- initial imputation (test_step_vae_sampling.py:276-297) -------------------------
    def basic_imputation(self, X, M):
        B, K, D = X.shape
        src = self.draw_source
        if self.hparams.imputation_fn in ('rand_bin', 'std_gauss'):
            # irwg/utils/basic_imputation.py:3-16: uniform draws u of X's shape; missing entries become u > 0.5
            # ('rand_bin') or u itself ('std_gauss': the reference draws torch.rand_like there too); X*M + X_imp*(~M),
            # so a NaN at a missing entry of X stays NaN, as in the reference
            if src is not None:
                u = src.uniform(B, K, D).to(X.device, torch.float32)
            else:   # counter RNG rows (example_offset + b)*K + k: independent of batching
                u = engine.rng_uniform(_lib.make_rng(self.rng_seed, 0, self.example_offset * K), _lib.STREAM_INIT_X,
                                       B * K, D, X.device).reshape(B, K, D)
            X_imp = (u > 0.5).to(X.dtype) if self.hparams.imputation_fn == 'rand_bin' else u
            return X * M + X_imp * (~M)
        if self.hparams.imputation_fn != 'vae_samples':
            raise NotImplementedError(f"imputation_fn={self.hparams.imputation_fn!r} ('vae_samples', 'rand_bin', 'std_gauss')")
        draws = None
        if src is not None:
            bern = self.model.hparams.generator_distribution.name == "bern"
            uni = bool(self.model.hparams.use_uniform_prior)  # VAE.sample draws torch.rand then
            draws = {"z": (src.uniform if uni else src.normal)(B * K, self.model.latent_dim),
                     **_x_draws(src, self.model, B * K, D)}
        # rng rows are (example_offset + b)*K + k of a dedicated stream: independent of batching
        rng = _lib.make_rng(self.rng_seed, 0, self.example_offset * K)
        if draws is not None:
            Z = draws["z"].to(X.device, torch.float32).contiguous()
        elif self.model.hparams.use_uniform_prior:
            Z = engine.rng_uniform(rng, _lib.STREAM_INIT_Z, B * K, self.model.latent_dim, X.device)
        else:
            Z = engine.rng_normal(rng, _lib.STREAM_INIT_Z, B * K, self.model.latent_dim, X.device)
        if self.model.hparams.use_uniform_prior:
            Z = Z * (1. + 1e-8)  # vae.py:168
        gen = engine.decode_params(self.model, Z)
        Xn = X.clone().contiguous()
        rng_x = _lib.make_rng(self.rng_seed, 0, self.example_offset)
        engine.sample_impute(self.model, gen, Xn, engine.mask_u8(M), K, eps=draws["eps_x"] if draws else None,
                             chi2=draws.get("chi2_x") if draws else None,
                             rng=rng_x, stream_id=_lib.STREAM_INIT_X, zero_missing=True)
        return Xn

    # ---- harness (test_step_vae_sampling.py:344-416, 488-496, 766-791) -------------------
    def test_step(self, batch, batch_idx: int):
        X_true, M = batch[:2]
        I = batch[-1]
        dev = self.device
        X_true = X_true.to(dev, torch.float32)
        M = M.to(dev).bool()
        bool_data = _check_if_bool(X_true[M])
        K = self.hparams.num_copies
        X = X_true.unsqueeze(1).repeat(1, K, *([1] * (X_true.dim() - 1))).contiguous()
        M = M.unsqueeze(1)
        X_true1 = X_true.unsqueeze(1)
        hp = self.hparams
        imputations, latents = [], ([] if hp.store_latents else None)
        metrics = defaultdict(list)
        # point-wise diagnostics accumulate in device buffers [T, B, K]; one transfer at the end
        T_max, B = self.max_iterations, X.shape[0]
        point = {}
        for flag, name in (('compute_rmse_metric', 'rmse'), ('compute_mae_metric', 'mae'),
                           ('compute_latent_wass_metric', 'latent_wass')):
            if hp.get(flag):
                point[name] = torch.empty(T_max, B, K, device=dev, dtype=torch.float32)
        truth = imputation_metrics.PreparedTruth(X_true1, M, X) if point else None
        q_true = None
        if 'latent_wass' in point:
            q_true = imputation_metrics.encode_posterior(self.model, truth.Xt.unsqueeze(1))
        self._iteration = 0
        X = self.basic_imputation(X, M)
        imputations.append(X.clone())
        t = -1
        outputs = None

        def stored(t_):
            return ((hp.store_last_n_iterations == -1 or (self.max_iterations - hp.store_last_n_iterations) <= t_)
                    and t_ % hp.store_every_n_iterations == 0)
        # The chain state is read after iteration t when a diagnostic is computed from it and when it is stored.  A
        # sampler may take both over (self._harness.observer) and run many iterations in one library call; it then
        # marks what it yields as '_observed' (diagnostics of this iteration are in the buffers) and its 'X' as
        # '_X_private' (a tensor of its own that nothing overwrites: kept without a copy).
        self._harness = _HarnessPlan(point, truth, q_true, stored)
        for t, outputs in enumerate(self.sampler(X, M)):
            X = outputs['X']
            if not outputs.get('_observed'):
                if 'rmse' in point or 'mae' in point:  # one launch for both
                    imputation_metrics.imputation_errors(X, truth, rmse=point['rmse'][t] if 'rmse' in point else None,
                                                         mae=point['mae'][t] if 'mae' in point else None)
                if 'latent_wass' in point:
                    imputation_metrics.imputation_latent_wass(X, X_true1, self.model, q_true=q_true,
                                                              out=point['latent_wass'][t])
            if stored(t):
                imputations.append(X if outputs.get('_X_private') else X.clone())
                if hp.store_latents:
                    latents.append(outputs['Z'].clone())
            for key, val in outputs.items():
                if key in ('X', 'Z', 'is_ess_not_reduced', 'total_num_props') or key.startswith('_'):
                    continue
                metrics[key].append(val)
        if t >= 0 and ((hp.store_last_n_iterations == -1 or (self.max_iterations - hp.store_last_n_iterations) <= t)
                       and not (t % hp.store_every_n_iterations == 0)):
            imputations.append(X.clone())
            if hp.store_latents:
                latents.append(outputs['Z'].clone())
        self._harness = None
        # one device->host transfer for the whole sequence
        imps = _to_host(torch.stack(imputations))
        if bool_data:
            imps = imps.astype(bool)
        lat = _to_host(torch.stack(latents)) if latents else []
        self.last_imputations = imps
        if not hp.dont_store_imputations and self._stores_artifacts() and self._harness_writes_imputations:
            self._save_npz(f'imputations_{batch_idx}.npz',
                           imputations=imps, latents=lat, true_X=X_true1.cpu().numpy(),
                           masks=M.cpu().numpy(), data_idx=np.asarray(I.cpu() if torch.is_tensor(I) else I))
        out = {k: [float(v) for v in vals] for k, vals in metrics.items()}   # device scalars: read once, here
        for name, buf in point.items():
            out[f'imp_metric/{name}'] = self._keep_pointwise(buf[:t + 1])
        for key, value in out.items():
            self.metrics[key].append(value)
        self.metrics['num_X'].append(X_true.shape[0])
        return out


def _to_host(t):
    """Device tensor -> numpy array.  Large ones go through page-locked memory (a batch's kept imputations are
    about a gigabyte, which the pageable path of ``.cpu()`` moves at a few GB/s); the framework's host allocator
    keeps such blocks for the next batch once the artefact writer has let go of the array."""
    if not t.is_cuda or t.numel() * t.element_size() < (8 << 20):
        return t.cpu().numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t)
    return h.numpy()


class _HarnessPlan:
    """What test_step does with the state after every iteration: the point-wise diagnostic buffers [T,B,K] it fills
    (``point``: 'rmse' / 'mae' / 'latent_wass'), what they are computed against, and which iterations it keeps.
    ``observer(t, n, ...)`` hands the same work for the iterations t .. t+n-1 to a run of n iterations in one
    library call (engine.RunObserver)."""

    def __init__(self, point, truth, q_true, stored):
        self.point, self.truth, self.q_true, self.stored = point, truth, q_true, stored

    def observer(self, t, n, shape, device):
        B, K, D = shape
        cut = lambda name: self.point[name][t:t + n] if name in self.point else None
        keep = any(self.stored(t + i) for i in range(n))
        return engine.RunObserver(
            X_store=torch.empty(n, B, K, D, device=device, dtype=torch.float32) if keep else None,
            X_true=self.truth.Xt if self.truth is not None else None, rmse=cut('rmse'), mae=cut('mae'),
            q_true=self.q_true, wass=cut('latent_wass'))


def _means_of_three(a, b, c):
    """Means of three per-example statistics.  The iteration returns them as consecutive rows of one workspace
    array: one reduction launch then instead of three (at 65 536 examples the three launches were 3 % of a LAIR
    iteration)."""
    n = a.numel()
    if (a.dim() == 1 and a.is_contiguous() and b.is_contiguous() and c.is_contiguous() and b.numel() == n and c.numel() == n
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + n and c.storage_offset() == a.storage_offset() + 2 * n):
        m = torch.as_strided(a, (3, n), (n, 1), a.storage_offset()).mean(dim=1)
        return m[0], m[1], m[2]
    return a.mean(), b.mean(), c.mean()


def _final_resampling(self, proposals, X_true, M, num_samples):
    """Categorical over all stored proposals of every example, ``n`` draws, decode in one launch,
    sample and set the observed values (test_step_vae_sampling.py:1991-2032, 2151-2185).
    ``proposals`` = (Z[T,I,B,K',Zd], log_w[T,I,B,K']) on the device."""
    import ctypes as C
    Zp, lw = proposals
    T, I, B, Kp, Zd = Zp.shape
    n = T * Kp if num_samples is None else num_samples
    dev = Zp.device
    src = self.draw_source
    u = src.uniform64(B, n).to(dev).contiguous() if src is not None else None
    cdf = torch.empty(B, T * I * Kp, device=dev)
    idx = torch.empty(B, n, device=dev, dtype=torch.int64)
    Zout = torch.empty(B, n, Zd, device=dev)
    rng = _lib.make_rng(self.rng_seed, 0, self.example_offset)
    _lib.check(_lib.lib().cs_final_resample(_lib.ptr(lw), _lib.ptr(Zp), T * I, B, Kp, Zd, _lib.ptr(u),
                                            C.byref(rng), n, _lib.ptr(cdf), _lib.ptr(idx), _lib.ptr(Zout),
                                            _lib.current_stream()))
    gen = engine.decode_params(self.model, Zout.reshape(B * n, Zd))
    D = X_true.shape[-1]
    Xz = X_true.to(dev, torch.float32).unsqueeze(1).repeat(1, n, 1).contiguous()
    eps = chi2 = None
    if src is not None:
        chunks = [_x_draws(src, self.model, B, min(Kp, n - c0), D) for c0 in range(0, n, Kp)]
        eps = torch.cat([c["eps_x"] for c in chunks], dim=1)
        if "chi2_x" in chunks[0]:
            chi2 = torch.cat([c["chi2_x"] for c in chunks], dim=1)
    engine.sample_impute(self.model, gen, Xz, engine.mask_u8(M), n, eps=eps, chi2=chi2, rng=rng,
                         stream_id=_lib.STREAM_UFINAL, zero_missing=True)
    return Xz.unsqueeze(0), Zout, idx


class TestVAELatentAdaptiveImportanceResampling(TestVAESamplerBase):
    """IRWG sampling, called LAIR (latent-adaptive importance resampling) in the paper."""

    def __init__(self,
                 num_iterations: int,
                 num_imp_samples: int,
                 weighting_scheme: str,
                 resampling_scheme: str,
                 resampling_method: str,
                 *args,
                 glr_cluster_size: int = None,
                 num_pseudo_warmup_iterations: int = 0,
                 num_prior_replenish_proposals: int = 0,
                 prior_replenish_iteration_frequency: int = 1,

                 var_proposal_temperature: float = None,
                 var_proposal_anneal_type_to_prior: str = None,

                 weight_transform: str = None,
                 clip_imputations: bool = False,

                 num_historical_proposals: int = None,
                 no_historic_proposals_iterations: int = None,
                 imputation_history_length: int = None,

                 use_miss_model: bool = False,
                 accept_reject_xm: bool = False,

                 store_norm_log_weights: bool = False,

                 resample_final_imps: bool = False,
                 resample_final_imps_num_samples: int = None,
                 **kwargs
                 ):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()

    @property
    def max_iterations(self):
        return self.hparams.num_iterations

    def on_test_start(self):
        super().on_test_start()
        hp = self.hparams
        if hp.use_miss_model or hp.accept_reject_xm:
            raise NotImplementedError("missingness models (MNAR) are not implemented")
        _lib.proposal_transform_args(hp.var_proposal_temperature, hp.var_proposal_anneal_type_to_prior)
        if hp.weight_transform not in (None, 'poly'):
            raise NotImplementedError()

    def on_test_epoch_start(self):
        super().on_test_epoch_start()
        if self.hparams.store_norm_log_weights:
            self.norm_log_weights = []

    def on_test_batch_start(self, batch: Any, batch_idx: int, dataloader_idx: int = 0) -> None:
        super().on_test_batch_start(batch, batch_idx, dataloader_idx)
        self._proposals = None  # device store [T,I,B,K',Zd] / [T,I,B,K'] for the final resampling

    def polynomial_transform(self, log_weights, iter, max_iter):
        alpha = min((iter + 10) / max_iter, 1)
        return log_weights * alpha

    def _draws(self, B, K, R, D, second, use_hist=False, history=None):
        """Injected draws of one iteration in the reference's call order: history window / chain
        integers (:539-541), proposals, resampling (twice with store_latents), x_m."""
        src = self.draw_source
        if src is None:
            return None
        hp = self.hparams
        d = {}
        K_out, pad_to = K, None
        if use_hist:
            H = hp.num_historical_proposals
            d["hist_r"] = src.randint_array(history.window_size(H), H)
            if hp.resampling_scheme != 'glr':
                d["hist_chain"] = src.randint_array(K, B)
            K = K + H
        elif (hp.no_historic_proposals_iterations is not None and hp.resampling_scheme == 'glr'
              and hp.num_historical_proposals is not None):
            pad_to = K + hp.num_historical_proposals
        I, Kp, Zd = hp.num_imp_samples, K + R, self.model.latent_dim
        bern = self.model.hparams.generator_distribution.name == "bern"
        # a Beta posterior is sampled by torch's Dirichlet sampler: the tape holds the samples
        beta = self.model.hparams.var_latent_distribution.name == "beta"
        d["eps_z"] = src.beta(I, B, Kp, Zd) if beta else src.normal(I, B, Kp, Zd)
        G = (pad_to or Kp) // hp.glr_cluster_size if hp.resampling_scheme == 'glr' else 1

        def res():
            if hp.resampling_scheme in ('lr', 'ir'):  # one index per chain among its I candidates
                return src.exponential(B * Kp, I) if hp.resampling_method == 'multinomial' else src.uniform(B, Kp, 1)
            if hp.resampling_method == 'multinomial':
                return src.uniform64(B * G, (pad_to or Kp) // G)
            return src.uniform(B, 1) if hp.resampling_scheme == 'gr' else src.uniform(B, G, 1)
        d["u_res"] = res()
        if second:
            d["u_res2"] = res()
        d.update(_x_draws(src, self.model, B, K_out, D))
        return d

    def sampler(self, X, M):
        hp = self.hparams
        B, K, D = X.shape
        history = None
        if hp.num_historical_proposals is not None:   # :1866-1870
            history = ImputationHistoryQueue(max_history_length=hp.imputation_history_length, batch_shape=X.shape,
                                             dtype=torch.bool if _check_if_bool(X) else torch.float32, device=X.device)
        clip_min = clip_max = None
        if hp.clip_imputations:
            X_temp = X[:, 0].clone()
            X_temp *= M.squeeze(1)
            clip_max = (torch.max(X_temp, dim=0)[0] * 2).contiguous()
            clip_min = (torch.min(X_temp, dim=0)[0] * 2).contiguous()
            del X_temp
        X = X.clone().contiguous()

        if hp.num_pseudo_warmup_iterations > 0:
            for t in range(hp.num_pseudo_warmup_iterations):
                draws = None
                if self.draw_source is not None:
                    draws = {"eps_z": self.draw_source.normal(B, K, self.model.latent_dim),
                             **_x_draws(self.draw_source, self.model, B, K, D)}
                X, Z_imp = pseudo_gibbs_iteration(self.model, X, M, rng=self._rng(), draws=draws, inplace=True)
                self._next_iteration()
                if history is not None:
                    history.enqueue_batch(X)
                yield {'X': X, 'Z': Z_imp.clone(), 'is_ess': 0., 'is_ess_not_reduced': torch.zeros(B),
                       'total_num_props': 1.}

        all_norm_log_weights = []
        T = hp.num_iterations
        # Under the harness of test_step the iterations run max_iterations_per_call at a time in one library call
        # (cs_lair_run_observed), which also does the harness's work on every iteration's state: same kernels, same
        # counter-RNG iteration numbers, so the chain is bit for bit the one of single steps.
        can_run = (self._harness is not None and self.draw_source is None and history is None
                   and not hp.store_latents and hp.weight_transform is None and not hp.store_norm_log_weights
                   and hp.prior_replenish_iteration_frequency == 1 and self.max_iterations_per_call > 1)
        t = 0
        while t < T:
            n = min(T - t, self.max_iterations_per_call) if can_run else 1
            if n == 1:
                yield from self._single_step(t, T, X, M, history, clip_min, clip_max, all_norm_log_weights)
                t += 1
                continue
            R = hp.num_prior_replenish_proposals
            ws = engine.lair_workspace(self.model, B, K, R, hp.num_imp_samples, X.device)
            cfg = engine.lair_cfg(self.model, ws, weighting=hp.weighting_scheme, scheme=hp.resampling_scheme,
                                  method=hp.resampling_method, cluster_size=hp.glr_cluster_size, weight_alpha=None,
                                  temperature=hp.var_proposal_temperature, anneal_type=hp.var_proposal_anneal_type_to_prior)
            if hp.resample_final_imps and self._proposals is None:
                self._proposals = (torch.empty((T,) + tuple(ws.Z.shape), device=X.device),
                                   torch.empty((T,) + tuple(ws.log_w.shape), device=X.device))
            stats = torch.empty(3, n, B, device=X.device)
            obs = self._harness.observer(hp.num_pseudo_warmup_iterations + t, n, (B, K, D), X.device)
            engine.lair_run(self.model, ws, X, engine.mask_u8(M), cfg, n, rng=self._rng(),
                            Z_store=self._proposals[0][t:t + n] if hp.resample_final_imps else None,
                            log_w_store=self._proposals[1][t:t + n] if hp.resample_final_imps else None,
                            stats_store=stats, clip_min=clip_min, clip_max=clip_max, observe=obs)
            self._iteration += n
            means = stats.mean(dim=2).tolist()   # [3][n]: one read for the stretch
            for i in range(n):
                kept = obs.X_store is not None
                yield {'X': obs.X_store[i] if kept else X, '_X_private': kept, '_observed': True, 'Z': None,
                       'is_ess': means[0][i], 'is_ess_not_reduced': stats[0, i],
                       'transformed_ess': means[1][i], 'perplexity': means[2][i],
                       'total_num_props': hp.num_imp_samples * (K + R), 'acceptance_x': 1.0}
            t += n

        if hp.store_norm_log_weights:
            self.norm_log_weights.append(torch.stack(all_norm_log_weights, dim=1).cpu().numpy())

    def _single_step(self, t, T, X, M, history, clip_min, clip_max, all_norm_log_weights):
        """Iteration t of the sampler as one call of importance_resampling_gibbs_iteration (in place on X); yields its
        output dict."""
        hp = self.hparams
        B, K, D = X.shape
        weight_transform = None
        if hp.weight_transform == 'poly':
            weight_transform = min((t + 10) / self.max_iterations, 1)
        R = hp.num_prior_replenish_proposals if t % hp.prior_replenish_iteration_frequency == 0 else 0
        use_hist = (hp.no_historic_proposals_iterations is not None
                    and (hp.num_pseudo_warmup_iterations + t) > hp.no_historic_proposals_iterations)
        ret = importance_resampling_gibbs_iteration(
            self.model, X, M, hp.num_imp_samples, glr_cluster_size=hp.glr_cluster_size,
            weighting=hp.weighting_scheme, resampling_method=hp.resampling_method,
            resampling_scheme=hp.resampling_scheme, num_prior_replenish_proposals=R,
            use_historical_proposals=use_hist, num_historical_proposals=hp.num_historical_proposals,
            history=history,
            weight_transform=weight_transform, return_resampled_latents=bool(hp.store_latents),
            var_proposal_temperature=hp.var_proposal_temperature,
            var_proposal_anneal_type_to_prior=hp.var_proposal_anneal_type_to_prior,
            rng=self._rng(), draws=self._draws(B, K, R, D, bool(hp.store_latents), use_hist, history),
            clip_min=clip_min, clip_max=clip_max, inplace=True)
        self._next_iteration()
        if history is not None:   # :1961-1962
            history.enqueue_batch(ret[0])
        X, ess, transformed_ess, perplexity, total_num_props, norm_log_weights, Z_prop, log_weights = ret[:8]
        latents = ret[9].clone() if hp.store_latents else None  # (the reference fails without store_latents)

        if hp.resample_final_imps:
            if self._proposals is None:
                if ((hp.prior_replenish_iteration_frequency != 1 and hp.num_prior_replenish_proposals > 0)
                        or hp.no_historic_proposals_iterations is not None):
                    raise NotImplementedError("final resampling with a varying number of proposals")
                self._proposals = (torch.empty((T,) + tuple(Z_prop.shape), device=X.device),
                                   torch.empty((T,) + tuple(log_weights.shape), device=X.device))
            self._proposals[0][t].copy_(Z_prop)
            self._proposals[1][t].copy_(log_weights)
        if hp.store_norm_log_weights:
            all_norm_log_weights.append(norm_log_weights.clone())

        ess_mean, tess_mean, perp_mean = _means_of_three(ess, transformed_ess, perplexity)
        yield {'X': X,
               'Z': latents,
               'is_ess': ess_mean,
               'is_ess_not_reduced': ess,
               'transformed_ess': tess_mean,
               'perplexity': perp_mean,
               'total_num_props': total_num_props,
               'acceptance_x': 1.0,
               }

    def final_resampling(self, X_true, M):
        """Resample the latents across all iterations and decode (:1991-2032).  Returns
        ``(imputations[1,B,n,D], latents[B,n,Zd], idx[B,n])`` as device tensors."""
        return _final_resampling(self, self._proposals, X_true, M, self.hparams.resample_final_imps_num_samples)

    def on_test_batch_end(self, outputs, batch: Any, batch_idx: int, dataloader_idx: int = 0) -> None:
        super().on_test_batch_end(outputs, batch, batch_idx, dataloader_idx)
        if not self.hparams.resample_final_imps:
            return
        X_true, M = batch[:2]
        I = batch[-1]
        dev = self.device
        M1 = M.to(dev).bool().unsqueeze(1)
        imps, Z, _ = self.final_resampling(X_true, M1)
        # float32 also for binary data: the reference decides its bool cast from the stored latents
        # (test_step_vae_sampling.py:1988-1989, never 0/1), so this file always holds float32
        imps = _to_host(imps)
        self.last_final_imputations = imps
        if self._stores_artifacts():
            self._save_npz(f'irwg_imputations_after_final_resampling_{batch_idx}.npz',
                           imputations=imps, latents=_to_host(Z) if self.hparams.store_latents else [],
                           true_X=X_true.unsqueeze(1).cpu().numpy(), masks=M1.cpu().numpy(),
                           data_idx=np.asarray(I.cpu() if torch.is_tensor(I) else I))

    def test_epoch_end(self, outputs):
        out = super().test_epoch_end(outputs)
        if self.hparams.store_norm_log_weights and self.out_dir is not None:
            np.savez_compressed(os.path.join(self.out_dir, 'is_norm_log_weights.npz'),
                                is_norm_log_weights=np.concatenate(self.norm_log_weights))
        return out


class TestVAEImportanceResampling(TestVAESamplerBase):
    """IR sampling: importance resampling with the prior as the proposal
    (test_step_vae_sampling.py:2063-2194).  After the last iteration of a batch the latents of all
    iterations are resampled and decoded, and ``imputations_{batch}.npz`` is rewritten with them
    (the reference's on_test_batch_end overwrites the harness's file of the same name)."""
    _harness_writes_imputations = False

    def __init__(self,
                 num_iterations: int,
                 num_imp_samples: int,
                 weighting_scheme: str,
                 resampling_scheme: str,
                 resampling_method: str,
                 *args,
                 glr_cluster_size: int = None,

                 resample_final_imps_num_samples: int = None,
                 **kwargs
                 ):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()

    @property
    def max_iterations(self):
        return self.hparams.num_iterations

    def on_test_batch_start(self, batch: Any, batch_idx: int, dataloader_idx: int = 0) -> None:
        super().on_test_batch_start(batch, batch_idx, dataloader_idx)
        self._proposals = None

    def _draws(self, B, K, D):
        src = self.draw_source
        if src is None:
            return None
        hp = self.hparams
        I, Zd = hp.num_imp_samples, self.model.latent_dim
        bern = self.model.hparams.generator_distribution.name == "bern"
        uni = bool(self.model.hparams.use_uniform_prior)   # Uniform.sample draws torch.rand
        d = {"eps_z": (src.uniform if uni else src.normal)(I, B, K, Zd)}
        if hp.resampling_method == 'multinomial':
            if hp.resampling_scheme in ('lr', 'ir'):
                d["u_res"] = src.exponential(B * K, I)
            else:
                G = K // hp.glr_cluster_size if hp.resampling_scheme == 'glr' else 1
                d["u_res"] = src.uniform64(B * G, K // G)
        elif hp.resampling_scheme == 'gr':
            d["u_res"] = src.uniform(B, 1)
        elif hp.resampling_scheme == 'glr':
            d["u_res"] = src.uniform(B, K // hp.glr_cluster_size, 1)
        else:
            d["u_res"] = src.uniform(B, K, 1)
        d.update(_x_draws(src, self.model, B, K, D))
        return d

    def sampler(self, X, M):
        hp = self.hparams
        B, K, D = X.shape
        X = X.clone().contiguous()
        T = hp.num_iterations
        for t in range(T):
            ret = standard_importance_resampling(
                self.model, X, M, hp.num_imp_samples, glr_cluster_size=hp.glr_cluster_size,
                weighting=hp.weighting_scheme, resampling_method=hp.resampling_method,
                resampling_scheme=hp.resampling_scheme, rng=self._rng(), draws=self._draws(B, K, D), inplace=True)
            self._next_iteration()
            X, ess, transformed_ess, perplexity, total_num_props, _, Z_prop, log_weights = ret
            if self._proposals is None:
                self._proposals = (torch.empty((T,) + tuple(Z_prop.shape), device=X.device),
                                   torch.empty((T,) + tuple(log_weights.shape), device=X.device))
            self._proposals[0][t].copy_(Z_prop)
            self._proposals[1][t].copy_(log_weights)
            yield {'X': X, 'is_ess_standard': ess.mean(), 'is_ess_not_reduced': ess,
                   'perplexity_standard': perplexity.mean()}

    def final_resampling(self, X_true, M):
        return _final_resampling(self, self._proposals, X_true, M, self.hparams.resample_final_imps_num_samples)

    def on_test_batch_end(self, outputs, batch: Any, batch_idx: int, dataloader_idx: int = 0) -> None:
        super().on_test_batch_end(outputs, batch, batch_idx, dataloader_idx)
        X_true, M = batch[:2]
        I = batch[-1]
        M1 = M.to(self.device).bool().unsqueeze(1)
        imps, _, _ = self.final_resampling(X_true, M1)
        imps = _to_host(imps)
        self.last_final_imputations = imps
        if self._stores_artifacts():
            self._save_npz(f'imputations_{batch_idx}.npz', imputations=imps,
                           true_X=X_true.unsqueeze(1).cpu().numpy(), masks=M1.cpu().numpy(),
                           data_idx=np.asarray(I.cpu() if torch.is_tensor(I) else I))


class TestVAEPseudoGibbs(TestVAESamplerBase):
    """Pseudo-Gibbs sampling (test_step_vae_sampling.py:1002-1042)."""

    def __init__(self,
                 num_iterations: int,
                 *args,
                 clip_imputations: bool = False,
                 clipping_mode: str = 'batch_data',
                 **kwargs
                 ):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()

    @property
    def max_iterations(self):
        return self.hparams.num_iterations

    def sampler(self, X, M):
        hp = self.hparams
        B, K, D = X.shape
        bern = self.model.hparams.generator_distribution.name == "bern"
        clip_min = clip_max = None
        if hp.clip_imputations:
            clip_max, clip_min = _clip_values(self, X, M, hp.clipping_mode)
        X = X.clone().contiguous()
        for t in range(hp.num_iterations):
            draws = None
            if self.draw_source is not None:
                src = self.draw_source
                draws = {"eps_z": src.normal(B, K, self.model.latent_dim),
                         **_x_draws(src, self.model, B, K, D)}
            # the reference calls the same function with and without clip values (:1037-1040); the
            # non-finite safety applies in both cases
            X, Z_imp = pseudo_gibbs_iteration(self.model, X, M, clip_min, clip_max, rng=self._rng(), draws=draws,
                                              inplace=True)
            self._next_iteration()
            yield {'X': X, 'Z': Z_imp.clone()}


def _clip_values(self, X, M, clipping_mode):
    """(max, min) bounds of the pseudo-Gibbs clipping (test_step_vae_sampling.py:1024-1034,1097-1107)."""
    if clipping_mode == 'batch_data':
        X_temp = X[:, 0].clone()
        X_temp *= M.squeeze(1)
        return ((torch.max(X_temp, dim=0)[0] * 2).contiguous(), (torch.min(X_temp, dim=0)[0] * 2).contiguous())
    if clipping_mode == 'dataset':
        core = self.datamodule.test_data_core
        return (torch.tensor(core.data_max * 2, device=X.device, dtype=torch.float32).contiguous(),
                torch.tensor(core.data_min * 2, device=X.device, dtype=torch.float32).contiguous())
    raise NotImplementedError(f'Invalid {clipping_mode=}')


class _MetropolisWithinGibbsCommon(TestVAESamplerBase):
    """Warm-up phases and chain-state set-up shared by the two Metropolis-within-Gibbs samplers
    (TestVAEMetropolisWithinGibbs :1090-1217, TestVAEAdaptiveCollapsedMetropolisWithinGibbs
    :1512-1662).  ``on_warmup_imputation(X, last)`` is called with every warm-up imputation in the
    order the reference produces them (AC-MWG enqueues some of them to its history)."""

    @property
    def max_iterations(self):
        if self.hparams.warmup_method != 'irwg':
            return self.hparams.num_pseudo_warmup_iterations + self.hparams.num_mwg_iterations
        return (self.hparams.num_pseudo_warmup_iterations * self.hparams.num_k_per_cluster) + self.hparams.num_mwg_iterations

    def on_test_start(self):
        super().on_test_start()
        hp = self.hparams
        _lib.proposal_transform_args(hp.var_proposal_temperature, hp.var_proposal_anneal_type_to_prior)
        if hp.warmup_method not in ('none', 'pseudo_gibbs', 'irwg'):
            raise NotImplementedError(f'{hp.warmup_method=}')

    def _lair_warmup_draws(self, B, Kw, R, D):
        src = self.draw_source
        if src is None:
            return None
        hp = self.hparams
        I, Kp, Zd = hp.num_imp_samples, Kw + R, self.model.latent_dim
        d = {"eps_z": src.normal(I, B, Kp, Zd)}
        G = Kp // hp.glr_cluster_size if hp.resampling_scheme == 'glr' else 1
        if hp.resampling_method == 'multinomial':
            d["u_res"] = src.uniform64(B * G, Kp // G)
        else:
            d["u_res"] = src.uniform(B, 1) if hp.resampling_scheme == 'gr' else src.uniform(B, G, 1)
        bern = self.model.hparams.generator_distribution.name == "bern"
        d.update(_x_draws(src, self.model, B, Kw, D))
        return d

    def _pseudo_gibbs_warmup(self, X, M, M8, on_warmup_imputation):
        """Generator over the pseudo-Gibbs warm-up steps (:1119-1155, :1548-1591); returns X."""
        hp, src, model = self.hparams, self.draw_source, self.model
        B, K, D = X.shape
        Zd = model.latent_dim
        bern = model.hparams.generator_distribution.name == "bern"
        clip_max = clip_min = None
        if hp.clip_imputations_during_pg_warmup:
            clip_max, clip_min = _clip_values(self, X, M, hp.clipping_mode)
        W = hp.num_pseudo_warmup_iterations
        for t in range(W):
            draws = None
            if src is not None:
                draws = {"eps_z": src.normal(B, K, Zd), **_x_draws(src, self.model, B, K, D)}
            if hp.clip_imputations_during_pg_warmup:
                X, Z_new = pseudo_gibbs_iteration(model, X, M, clip_min, clip_max, rng=self._rng(), draws=draws,
                                                  inplace=True)
            else:
                X, Z_new = self._pg_step_noclip(X, M, M8, draws)
            self._next_iteration()
            on_warmup_imputation(X, t == W - 1)
            yield {'X': X, 'Z': Z_new.clone(), 'acceptance': 1.}
        return X

    def _irwg_warmup(self, X, M, on_warmup_imputation):
        """Generator over the LAIR warm-up (:1156-1197, :1592-1636); returns the K chains' X."""
        hp, model = self.hparams, self.model
        B, K, D = X.shape
        c, R = hp.glr_cluster_size, hp.num_prior_replenish_proposals
        Kw = K * hp.num_k_per_cluster
        X_w = X[:, :1].expand(B, Kw, D).contiguous()
        X_w = self.basic_imputation(X_w, M)
        W = hp.num_pseudo_warmup_iterations
        for t in range(W):
            ret = importance_resampling_gibbs_iteration(
                model, X_w, M, hp.num_imp_samples, glr_cluster_size=c, weighting=hp.weighting_scheme,
                resampling_method=hp.resampling_method, resampling_scheme=hp.resampling_scheme,
                num_prior_replenish_proposals=R, var_proposal_temperature=hp.var_proposal_temperature,
                var_proposal_anneal_type_to_prior=hp.var_proposal_anneal_type_to_prior,
                rng=self._rng(), draws=self._lair_warmup_draws(B, Kw, R, D), inplace=True)
            self._next_iteration()
            X_w = ret[0]
            # 'b (c k) -> b k c' (:1611), padded to K chains by repeating the first ones (:1613-1615)
            X_temp = X_w.reshape(B, c, Kw // c, D).permute(0, 2, 1, 3).contiguous()
            if R > 0:
                needed = K - X_temp.shape[1]
                X_temp = torch.cat([X_temp, X_temp[:, :needed]], dim=1)
            for i in range(hp.num_k_per_cluster):
                X_temp_i = X_temp[:, :, i].contiguous()
                on_warmup_imputation(X_temp_i, t == W - 1)
                yield {'X': X_temp_i, 'acceptance': 1.0}
            if R > 0 and t == W - 1:
                X_w = X_temp.permute(0, 2, 1, 3).reshape(B, c * X_temp.shape[1], D)
        return X_w.reshape(B, c, X_w.shape[1] // c, D).permute(0, 2, 1, 3)[:, :, 0].contiguous()

    def _state_after_pseudo_gibbs(self, st, X, M8, joint):
        """Chain state = the last warm-up step's Z and generator parameters (:1214-1217,
        :1653-1660): its buffers are still in the LAIR workspace of that step."""
        model = self.model
        B, K, D = X.shape
        if self.hparams.num_pseudo_warmup_iterations < 1:
            raise ValueError("pseudo_gibbs warm-up needs at least one iteration")
        ws = engine.lair_workspace(model, B, K, 0, 1, X.device)
        st["Z_old"].copy_(ws.Z[0])
        st["gen_old"].copy_(ws.gen[0])
        L, stream = _lib.lib(), _lib.current_stream()
        _, gen_dist = model.engine_dists()
        # log-likelihood of the *new* imputations under those parameters; with the observed-only sum
        # this equals the value the step's decoder launch computed, but one rule covers both targets
        _lib.check(L.cs_gen_loglik_rows(_lib.ptr(st["gen_old"]), gen_dist, _lib.ptr(X), None if joint else _lib.ptr(M8),
                                        B, K, D, _lib.ptr(st["x_old_lp"]), stream))
        _lib.check(L.cs_stdnormal_logprob_rows(_lib.ptr(st["Z_old"]), B * K, model.latent_dim,
                                               _lib.ptr(st["z_old_prior_lp"]), stream))

    def _pg_step_noclip(self, X, M, M8, draws):
        """Pseudo-Gibbs warm-up step without the clipping / non-finite safety (:1553-1580)."""
        import ctypes as C
        B, K, D = X.shape
        model = self.model
        ws = engine.lair_workspace(model, B, K, 0, 1, X.device)
        cfg = engine.lair_cfg(model, ws, weighting="smis", scheme="gr", method="multinomial", cluster_size=0,
                              weight_alpha=None)
        eps_z = draws["eps_z"].to(X.device, torch.float32).contiguous() if draws else None
        bufs = _lib.make_bufs(X=X, M=M8, eps_z=eps_z, q=ws.q, Z=ws.Z, gen=ws.gen, gen_ll=ws.gen_ll)
        rng = self._rng()
        L = _lib.lib()
        stq = _lib.current_stream()
        _lib.check(L.cs_lair_encode(model.var_latent_network.packed().handle, C.byref(cfg), C.byref(bufs), stq))
        _lib.check(L.cs_lair_decode(model.generator_network.packed().handle, C.byref(cfg), C.byref(bufs), C.byref(rng), stq))
        engine.sample_impute(model, ws.gen, X, M8, K, eps=draws["eps_x"] if draws else None,
                             chi2=draws.get("chi2_x") if draws else None, rng=rng,
                             stream_id=_lib.STREAM_X)
        return X, ws.Z[0]

    def _warmup_yields(self):
        """States the warm-up yields before the first MWG step (the harness counts them as iterations)."""
        hp = self.hparams
        if hp.warmup_method == 'pseudo_gibbs':
            return hp.num_pseudo_warmup_iterations
        if hp.warmup_method == 'irwg':
            return hp.num_pseudo_warmup_iterations * hp.num_k_per_cluster
        return 0

    def _steps_per_call(self, remaining):
        """How many of the remaining MWG steps the next library call runs: under the harness of test_step, with
        counter-RNG draws, max_iterations_per_call at a time (cs_acmwg_run, which also does the harness's work on
        every step's state); one otherwise."""
        if (self._harness is None or self.draw_source is not None or self.hparams.store_latents
                or self.max_iterations_per_call <= 1):
            return 1
        return min(remaining, self.max_iterations_per_call)

    def _run_steps(self, n, t, X, M8, cfg, bufs, st, history):
        """MWG steps t .. t+n-1 in one library call; yields what the single steps yield.  Same kernels, same
        counter-RNG iteration numbers and history slots, so the chains are bit for bit those of single steps."""
        B, K, D = X.shape
        dev = X.device
        acc = torch.empty(n, B, K, device=dev, dtype=torch.uint8)
        log_acc = torch.empty(n, B, K, device=dev, dtype=torch.float32)
        slots = torch.empty(n, B, K, device=dev, dtype=torch.int64) if history is not None else None
        obs = self._harness.observer(self._warmup_yields() + t, n, (B, K, D), dev)
        engine.acmwg_run(self.model, cfg, bufs, n, rng=self._rng(), accepted_store=acc, log_accept_store=log_acc,
                         slots_store=slots, observe=obs, device=dev)
        self._iteration += n
        if history is not None:
            for _ in range(n):
                history.advance()
        means = acc.float().mean(dim=(1, 2)).tolist()   # one read for the stretch
        kept = obs.X_store is not None
        for i in range(n):
            out = {'X': obs.X_store[i] if kept else X, '_X_private': kept, '_observed': True, 'Z': st["Z_old"],
                   'acceptance': means[i], '_accepted': acc[i], '_log_accept': log_acc[i]}
            if slots is not None:
                out['_slots'] = slots[i]
            yield out

    def _chain_buffers(self, B, K, D, Zd, P, dev):
        f = dict(device=dev, dtype=torch.float32)
        st = dict(Z_old=torch.empty(B, K, Zd, **f), gen_old=torch.empty(B, K, P, **f),
                  z_old_prior_lp=torch.empty(B, K, **f), x_old_lp=torch.empty(B, K, **f))
        wk = dict(q=torch.empty(B, K, 2, Zd, **f), Z_new=torch.empty(B, K, Zd, **f), gen_new=torch.empty(B, K, P, **f),
                  ll_new=torch.empty(B, K, **f), accepted=torch.empty(B, K, device=dev, dtype=torch.uint8),
                  log_accept=torch.empty(B, K, **f))
        return st, wk


class TestVAEMetropolisWithinGibbs(_MetropolisWithinGibbsCommon):
    """Metropolis-within-Gibbs sampling (Mattei & Frellsen 2018) targeting the joint p(z, x_m | x_o)
    (test_step_vae_sampling.py:1045-1282)."""

    def __init__(self,
                 num_pseudo_warmup_iterations: int,
                 num_mwg_iterations: int,
                 *args,
                 clip_imputations_during_pg_warmup: bool = False,
                 clipping_mode: str = 'batch_data',

                 var_proposal_temperature: float = None,
                 var_proposal_anneal_type_to_prior: str = None,
                 warmup_method: str = 'pseudo_gibbs',

                 # IRWG warmup
                 num_k_per_cluster: int = None,
                 num_imp_samples: int = 1,
                 weighting_scheme: str = 'dmis_within_groups',
                 resampling_scheme: str = 'glr',
                 resampling_method: str = 'multinomial',
                 glr_cluster_size: int = None,
                 num_prior_replenish_proposals: int = 0,

                 **kwargs
                 ):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        if self.hparams.warmup_method == 'irwg':
            assert (self.hparams.num_k_per_cluster
                    + self.hparams.num_prior_replenish_proposals / self.hparams.glr_cluster_size) == self.hparams.glr_cluster_size

    def sampler(self, X, M):
        import ctypes as C
        hp, model, src = self.hparams, self.model, self.draw_source
        B, K, D = X.shape
        Zd = model.latent_dim
        dev = X.device
        latent, gen_dist = model.engine_dists()
        bern = gen_dist == _lib.CS_DIST["bern"]
        P = model.generator_network.output_dim
        M8 = engine.mask_u8(M)
        X = X.clone().contiguous()
        nothing = lambda X, last: None

        if hp.warmup_method == 'pseudo_gibbs':
            X = yield from self._pseudo_gibbs_warmup(X, M, M8, nothing)
        elif hp.warmup_method == 'irwg':
            X = yield from self._irwg_warmup(X, M, nothing)

        st, wk = self._chain_buffers(B, K, D, Zd, P, dev)
        L = _lib.lib()
        model.var_latent_network.packed().reserve(B * K)   # layer programs: workspace allocated before the loop
        model.generator_network.packed().reserve(B * K)
        h_enc = model.var_latent_network.packed().handle
        h_dec = model.generator_network.packed().handle
        stream = _lib.current_stream()
        flags = _lib.CS_MWG_PROPOSE_FROM_CURRENT | _lib.CS_MWG_JOINT_TARGET_NEW | _lib.CS_MWG_JOINT_TARGET_OLD
        temp, anneal = _lib.proposal_transform_args(hp.var_proposal_temperature, hp.var_proposal_anneal_type_to_prior)
        cfg = _lib.cs_acmwg_cfg(B, K, D, Zd, latent, gen_dist, 1, 0, 0, flags, 0.0, temp, anneal)

        if hp.warmup_method == 'pseudo_gibbs':
            self._state_after_pseudo_gibbs(st, X, M8, joint=True)
        else:  # 'none' / 'irwg': one proposal from the current imputations (:1110-1118, :1200-1208)
            eps = src.normal(B, K, Zd).to(dev).contiguous() if src is not None else None
            bufs = _lib.make_acmwg_bufs(X=X, M=M8, eps_z=eps, q=wk["q"], **st)
            rng = self._rng()
            _lib.check(L.cs_acmwg_init_state(h_enc, h_dec, C.byref(cfg), C.byref(bufs), C.byref(rng), stream))
            self._next_iteration()

        t, T = 0, hp.num_mwg_iterations
        while t < T:
            n = self._steps_per_call(T - t)
            if n > 1:
                yield from self._run_steps(n, t, X, M8, cfg, _lib.make_acmwg_bufs(X=X, M=M8, **st, **wk), st, None)
                t += n
                continue
            d = {}
            if src is not None:
                d["eps_z"] = src.normal(B, K, Zd).to(dev).contiguous()
                d["u_accept"] = src.uniform(B, K).to(dev).contiguous()
                d.update({k: v.to(dev).contiguous() for k, v in _x_draws(src, self.model, B, K, D).items()})
            bufs = _lib.make_acmwg_bufs(X=X, M=M8, **st, **wk, **d)
            rng = self._rng()
            _lib.check(L.cs_acmwg_step(h_enc, h_dec, C.byref(cfg), C.byref(bufs), C.byref(rng), stream))
            self._next_iteration()
            yield {'X': X, 'Z': st["Z_old"], 'acceptance': wk["accepted"].float().mean(),
                   '_accepted': wk["accepted"], '_log_accept': wk["log_accept"]}
            t += 1


class TestVAEAdaptiveCollapsedMetropolisWithinGibbs(_MetropolisWithinGibbsCommon):
    """Metropolis-within-Gibbs sampling with marginalised missing values, called AC-MWG
    (adaptive collapsed-Metropolis-within-Gibbs) in the paper."""

    def __init__(self,
                 num_pseudo_warmup_iterations: int,
                 num_mwg_iterations: int,
                 prior_mixture_probability: float,

                 *args,
                 enqueue_allwarmup_samples_to_history: bool = False,
                 warmup_method: str = 'pseudo_gibbs',
                 clip_imputations_during_pg_warmup: bool = False,
                 clipping_mode: str = 'batch_data',
                 var_proposal_temperature: float = None,
                 var_proposal_anneal_type_to_prior: str = None,
                 num_prior_replenish_proposals: int = 0,

                 # IRWG warmup
                 num_k_per_cluster: int = None,
                 num_imp_samples: int = 1,
                 weighting_scheme: str = 'dmis_within_groups',
                 resampling_scheme: str = 'glr',
                 resampling_method: str = 'multinomial',
                 glr_cluster_size: int = None,

                 ablation_use_mwg_target: bool = False,

                 **kwargs,
                 ):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        if self.hparams.warmup_method == 'irwg':
            assert (self.hparams.num_k_per_cluster
                    + self.hparams.num_prior_replenish_proposals / self.hparams.glr_cluster_size) == self.hparams.glr_cluster_size

    def sampler(self, X, M):
        import ctypes as C
        hp = self.hparams
        model = self.model
        src = self.draw_source
        B, K, D = X.shape
        Zd = model.latent_dim
        dev = X.device
        latent, gen_dist = model.engine_dists()
        bern = gen_dist == _lib.CS_DIST["bern"]
        P = model.generator_network.output_dim
        M8 = engine.mask_u8(M)
        X = X.clone().contiguous()

        bool_data = _check_if_bool(X)
        history = ImputationHistoryQueue_with_restrictedavailability(
            max_history_length=self.max_iterations, batch_shape=X.shape,
            dtype=X.dtype if not bool_data else torch.bool, device=dev,
            compact=isinstance(model.var_latent_network.packed(), _lib.PackedProgram))
        self.history = history

        def enqueue(Xw, last):
            if hp.enqueue_allwarmup_samples_to_history or last:
                history.enqueue_batch(Xw)

        if hp.warmup_method == 'none':
            history.enqueue_batch(X)
        elif hp.warmup_method == 'pseudo_gibbs':
            X = yield from self._pseudo_gibbs_warmup(X, M, M8, enqueue)
        elif hp.warmup_method == 'irwg':
            X = yield from self._irwg_warmup(X, M, enqueue)

        # ---- chain state after the warm-up (:1639-1662) ----
        st, wk = self._chain_buffers(B, K, D, Zd, P, dev)
        wk.update(slots=torch.empty(B, K, device=dev, dtype=torch.int64),
                  in_off=torch.empty(B, K, device=dev, dtype=torch.int64),
                  use_prior=torch.empty(B, K, device=dev, dtype=torch.uint8),
                  z_alt=torch.empty(B, K, Zd, device=dev, dtype=torch.float32))
        L = _lib.lib()
        model.var_latent_network.packed().reserve(B * K)   # layer programs: workspace allocated before the loop
        model.generator_network.packed().reserve(B * K)
        h_enc = model.var_latent_network.packed().handle
        h_dec = model.generator_network.packed().handle
        stream = _lib.current_stream()
        # the ablation scores the proposal on all dimensions of the current imputation, and so does
        # the state right after the warm-up (:1657-1658), but every step then keeps the observed-only
        # sum for the state it carries on (:1722-1725 vs :1755)
        ablation = bool(hp.ablation_use_mwg_target)
        flags = _lib.CS_MWG_JOINT_TARGET_NEW if ablation else 0

        temp, anneal = _lib.proposal_transform_args(hp.var_proposal_temperature, hp.var_proposal_anneal_type_to_prior)

        def cfg_now(flags=flags):
            return _lib.cs_acmwg_cfg(B, K, D, Zd, latent, gen_dist, history.max_history_length,
                                     max(history.tail_idx, 0), int(history.full), flags,
                                     float(hp.prior_mixture_probability), temp, anneal, int(history.u8))

        if hp.warmup_method == 'pseudo_gibbs':
            self._state_after_pseudo_gibbs(st, X, M8, joint=ablation)
        else:
            eps = src.normal(B, K, Zd).to(dev).contiguous() if src is not None else None
            bufs = _lib.make_acmwg_bufs(X=X, M=M8, eps_z=eps, q=wk["q"], **st)
            rng = self._rng()
            cfg0 = cfg_now(_lib.CS_MWG_JOINT_TARGET_OLD if ablation else 0)
            _lib.check(L.cs_acmwg_init_state(h_enc, h_dec, C.byref(cfg0), C.byref(bufs), C.byref(rng), stream))
            self._next_iteration()
        history.update_available_timesteps(None)

        t, T = 0, hp.num_mwg_iterations
        while t < T:
            n = self._steps_per_call(T - t)
            if n > 1:
                bufs = _lib.make_acmwg_bufs(X=X, M=M8, ring=history.history_queue, avail=history.available_timesteps,
                                            **st, **wk)
                yield from self._run_steps(n, t, X, M8, cfg_now(), bufs, st, history)
                t += n
                continue
            t += 1
            d = {}
            if src is not None:
                bounds = torch.clamp(history.available_timesteps, min=1).cpu()
                r = torch.empty(B, K, dtype=torch.int64)
                for b in range(B):
                    for k in range(K):
                        r[b, k] = src.randint(int(bounds[b, k]))
                d["r_hist"] = r.to(dev)
                d["eps_z"] = src.normal(B, K, Zd).to(dev).contiguous()
                if hp.prior_mixture_probability > 0.:
                    d["z_prior"] = src.normal(B, K, Zd).to(dev).contiguous()
                    d["u_prior"] = src.uniform(B, K).to(dev).contiguous()
                d["u_accept"] = src.uniform(B, K).to(dev).contiguous()
                d.update({k: v.to(dev).contiguous() for k, v in _x_draws(src, self.model, B, K, D).items()})
            bufs = _lib.make_acmwg_bufs(X=X, M=M8, ring=history.history_queue, avail=history.available_timesteps,
                                        **st, **wk, **d)
            rng = self._rng()
            _lib.check(L.cs_acmwg_step(h_enc, h_dec, C.byref(cfg_now()), C.byref(bufs), C.byref(rng), stream))
            history.advance()
            self._next_iteration()
            yield {'X': X, 'Z': st["Z_old"], 'acceptance': wk["accepted"].float().mean(),
                   '_accepted': wk["accepted"], '_slots': wk["slots"], '_log_accept': wk["log_accept"]}
