"""CPU: host-side logic around the kernels (no GPU calls): hyper-parameter capture, iteration
counts, unsupported-option errors, YAML driver parsing, data module, sharding."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE_KW = dict(store_last_n_iterations=-1, store_every_n_iterations=1, imputation_metric_quantiles=[],
               compute_f1_metric=False, compute_inv_f1_metric=False, compute_bin_accuracy_metric=False,
               compute_rmse_metric=False, compute_ssim_metric=False, compute_latent_wass_metric=False,
               imputation_fn="vae_samples", num_copies=4)


def test_sampler_classes_capture_hyperparameters_like_lightning():
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import (
        TestVAEAdaptiveCollapsedMetropolisWithinGibbs, TestVAELatentAdaptiveImportanceResampling)
    s = TestVAELatentAdaptiveImportanceResampling(num_iterations=7, num_imp_samples=1, weighting_scheme="dmis",
                                                  resampling_scheme="gr", resampling_method="multinomial",
                                                  num_prior_replenish_proposals=1, **BASE_KW)
    assert s.max_iterations == 7 and s.hparams.num_copies == 4 and s.hparams.resample_final_imps is False
    assert s.hparams.num_prior_replenish_proposals == 1 and s.hparams.store_latents is False
    a = TestVAEAdaptiveCollapsedMetropolisWithinGibbs(num_pseudo_warmup_iterations=100, num_mwg_iterations=2900,
                                                      prior_mixture_probability=0.05, warmup_method="irwg",
                                                      num_k_per_cluster=4, glr_cluster_size=5,
                                                      num_prior_replenish_proposals=5, **BASE_KW)
    assert a.max_iterations == 100 * 4 + 2900
    b = TestVAEAdaptiveCollapsedMetropolisWithinGibbs(num_pseudo_warmup_iterations=10, num_mwg_iterations=5,
                                                      prior_mixture_probability=0.0, **BASE_KW)
    assert b.max_iterations == 15 and b.hparams.warmup_method == "pseudo_gibbs"
    with pytest.raises(AssertionError):  # num_k_per_cluster + R/cluster must equal cluster (reference :1503)
        TestVAEAdaptiveCollapsedMetropolisWithinGibbs(num_pseudo_warmup_iterations=1, num_mwg_iterations=1,
                                                      prior_mixture_probability=0.0, warmup_method="irwg",
                                                      num_k_per_cluster=3, glr_cluster_size=5,
                                                      num_prior_replenish_proposals=5, **BASE_KW)


def test_unsupported_options_raise_not_implemented():
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import TestVAELatentAdaptiveImportanceResampling
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    kw = dict(BASE_KW)
    kw["compute_ssim_metric"] = True  # a diagnostic that is not implemented (RMSE, MAE, latent-Wasserstein are)
    s = TestVAELatentAdaptiveImportanceResampling(num_iterations=1, num_imp_samples=1, weighting_scheme="dmis",
                                                  resampling_scheme="gr", resampling_method="multinomial", **kw)
    with pytest.raises(NotImplementedError):
        s.on_test_start()
    s.skip_unimplemented_diagnostics = True
    s.on_test_start()
    X = torch.zeros(2, 2, 3); M = torch.ones(2, 1, 3, dtype=torch.bool)
    for bad in (dict(resampling_scheme="rr"), dict(weighting="nope"),
                dict(resampling_method="residual"),
                dict(var_proposal_anneal_type_to_prior="quadratic"), dict(accept_reject_xm=True)):
        with pytest.raises(NotImplementedError):
            importance_resampling_gibbs_iteration(None, X, M, 1, **bad)


def test_models_mirror_reference_state_dict_layout():
    from vae_conditional_sampling_amd.models import DISTRIBUTION, VAE, ResidualFCNetwork
    enc = ResidualFCNetwork(43, 64, 5, 256, dropout_probability=0.5, activation="relu")
    dec = ResidualFCNetwork(32, 86, 2, 256)
    vae = VAE(dec, "normal_with_eps", enc, DISTRIBUTION.normal_with_eps)
    keys = set(vae.state_dict().keys())
    assert "var_latent_network.initial_layer.weight" in keys
    assert "var_latent_network.blocks.4.linear_layers.1.bias" in keys
    assert "generator_network.final_layer.weight" in keys
    assert vae.state_dict()["generator_network.final_layer.weight"].shape == (86, 256)
    assert vae.hparams.generator_distribution is DISTRIBUTION.normal_with_eps and vae.latent_dim == 32
    ln = ResidualFCNetwork(4, 4, 1, 64, use_layer_norm=True)
    bn = ResidualFCNetwork(4, 4, 1, 64, use_batch_norm=True)
    assert "blocks.0.layer_norm.bias" in ln.state_dict() and "blocks.0.batch_norm.num_batches_tracked" in bn.state_dict()
    with pytest.raises(NotImplementedError):
        ResidualFCNetwork(4, 4, 1, 64, activation="tanh")
    p = vae.split_params(torch.arange(12.).reshape(2, 6), 2)
    assert p.shape == (2, 2, 3) and torch.equal(vae.join_params(p), torch.arange(12.).reshape(2, 6))


def test_driver_config_and_data_module(tmp_path):
    from vae_conditional_sampling_amd import driver
    from vae_conditional_sampling_amd.data import MissingDataModule
    cfg = driver.load_config(os.path.join(ROOT, "configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml"),
                             ["data.batch_size=128", "test_class.init_args.num_iterations=3"])
    assert cfg["data"]["batch_size"] == 128 and cfg["test_class"]["init_args"]["num_iterations"] == 3
    assert driver.construct_experiment_subdir(cfg).endswith("seed_m20220118_d20220118")
    cls = driver.resolve_test_class(cfg["test_class"]["class_path"])
    assert cls.__name__ == "TestVAELatentAdaptiveImportanceResampling"
    tester = cls(**cfg["test_class"]["init_args"])
    assert tester.max_iterations == 3
    with pytest.raises(NotImplementedError):
        driver.resolve_test_class("irwg.sampling.test_step_vae_sampling.TestVAEMetropolisWithinGibbsMNAR")
    assert driver.resolve_test_class("irwg.sampling.test_step_vae_sampling.TestVAEPseudoGibbs").__name__ == "TestVAEPseudoGibbs"
    dm = MissingDataModule(**cfg["data"])
    dm.setup("test")
    X, M, I = dm.test_data
    assert X.shape == (3000, 43) and M.dtype == torch.bool and I.dtype == torch.int64
    assert M.any(dim=1).all()                      # fully-missing rows were filtered
    assert abs(float((~M).float().mean()) - 0.5) < 0.02
    batches = list(dm.test_dataloader())
    assert len(batches) == len(dm) == 24 and batches[0][0].shape == (128, 43)
    # sharding: contiguous blocks that tile the set, deterministic data regardless of world size
    parts = []
    for r in range(3):
        d = MissingDataModule(**cfg["data"], rank=r, world_size=3)
        d.setup("test")
        parts.append(d.test_data[0])
        assert d.shard_offset == sum(p.shape[0] for p in parts[:-1])
    assert torch.equal(torch.cat(parts), X)


def test_uniform_mask_has_exact_missing_count():
    from vae_conditional_sampling_amd.data.missing_data_module import shard_range, uniform_missingness_mask
    m = uniform_missingness_mask((100, 7), 0.5, torch.Generator().manual_seed(1))
    assert int((~m).sum()) == 350
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_range(0, 0, 2) == (0, 0)


def test_resnet_modules_mirror_reference_state_dict_layout():
    """The oracle's weight dicts use the reference's key names (pinned by a strict load into the
    reference modules when the fixtures were made); the product modules must accept them strictly."""
    from oracle import cs_oracle_conv as OC
    from tests.helpers import CONV_NETS
    from vae_conditional_sampling_amd.models.vae_resnet import ResNetDecoder, ResNetEncoder
    for name, n in CONV_NETS.items():
        enc_w = OC.make_resnet_encoder_weights(seed=1, **n["enc"])
        dec_w = OC.make_resnet_decoder_weights(seed=2, **n["dec"])
        enc, dec = ResNetEncoder(**n["enc"]), ResNetDecoder(**n["dec"])
        enc.load_state_dict(OC.state_dict_of(enc_w), strict=True)
        dec.load_state_dict(OC.state_dict_of(dec_w), strict=True)
        assert enc.input_dim == int(np.prod(n["enc"]["input_shape"])) and enc.output_dim == n["enc"]["latent_dim"]
        assert dec.input_dim == n["dec"]["latent_dim"] and dec.output_dim == int(np.prod(n["dec"]["output_shape"]))
    keys = set(dec.state_dict().keys())
    assert "layers.0.0.conv2.1.weight" in keys and "layers.0.0.upsample.0.1.weight" in keys
    assert "layers.3.0.conv2.weight" in keys and "layers.0.0.upsample.1.running_var" in keys
    with pytest.raises(NotImplementedError):
        ResNetEncoder((1, 28, 28), [1], [32], 8, first_conv=True)


def test_quadrant_and_nan_missingness():
    from vae_conditional_sampling_amd.data import MissingDataModule
    from vae_conditional_sampling_amd.data.missing_data_module import quadrant_missingness_mask
    g = torch.Generator().manual_seed(3)
    M = quadrant_missingness_mask(50, (28, 28), 0.5, g).reshape(50, 28, 28)
    for b in range(50):
        quads = [M[b, r:r + 14, c:c + 14] for r in (0, 14) for c in (0, 14)]
        assert sorted(int(q.all()) for q in quads) == [0, 0, 1, 1]          # two quadrants fully observed
        assert sorted(int((~q).all()) for q in quads) == [0, 0, 1, 1]       # two fully missing
    assert quadrant_missingness_mask(4, (28, 28), 0.0, g).all()
    dm = MissingDataModule("mnist_gmm_custtest0", 20, missingness="nan_miss", synthetic=True, setup_seed=5)
    dm.setup("test")
    X, Mb, I = next(iter(dm.test_dataloader()))
    assert X.shape == (10, 196) and torch.equal(Mb, ~torch.isnan(X)) and Mb[:, :98].all() and not Mb[:, 98:].any()


def test_history_queue_host_logic_and_glr_chain_index():
    """Stratified slots of ImputationHistoryQueue (irwg/sampling/utils.py:33-39) and the chain index of
    historical proposals under global-local resampling, whose known answer the reference documents
    in its comments (vae_imputation.py:508-515): K=5, 10 historical samples, cluster size 5."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd.sampling.utils import ImputationHistoryQueue, glr_history_chain_idx
    assert glr_history_chain_idx(5, 10, 5) == [2, 0, 1, 2, 3, 4, 2, 0, 1, 2]
    for K, H, c in [(5, 10, 5), (4, 4, 4), (3, 3, 2), (6, 6, 4)]:
        assert glr_history_chain_idx(K, H, c) == O.glr_history_chain_idx(K, H, c).tolist()
    q = ImputationHistoryQueue(7, (2, 3, 4), device="cpu")
    ref = O.ImputationHistory(7, (2, 3, 4))
    for step in range(11):      # wraps after 7
        q._increment_tail()
        ref.enqueue(torch.zeros(2, 3, 4))
        assert (q.tail_idx, q.full, len(q)) == (ref.tail, ref.full, len(ref))
        for H in (1, 2, 3):
            if len(q) < H:
                continue
            w = q.window_size(H)
            assert w == ref.window(H)
            r = [(step * 5 + h * 3) % w for h in range(H)]
            assert q.slots_from_window_draws(r, H) == ref.slots(torch.tensor(r), H).tolist()


import enum as _enum


class _TrainingOnlyDist(_enum.Enum):
    normal_with_eps = 1


class _TrainingOnlyNet(torch.nn.Module):   # stands for a class of the training code that the reader must not need
    def __init__(self):
        super().__init__()
        self.l = torch.nn.Linear(3, 2)


def test_lightning_checkpoint_reader_executes_nothing_from_the_file(tmp_path):
    """A checkpoint as the reference's training writes it: state_dict next to hyper_parameters that
    pickle nn.Module instances and enums (vae.py:99-122).  weights_only loading refuses such a file;
    the reader returns the state_dict without importing or constructing any class the file names."""
    import collections
    import pickle
    from vae_conditional_sampling_amd.models import VAE, ResidualFCNetwork
    from vae_conditional_sampling_amd.utils.checkpoint import load_state_dict_from_checkpoint
    Dist = _TrainingOnlyDist
    vae = VAE(ResidualFCNetwork(4, 14, 1, 64), "normal_with_eps", ResidualFCNetwork(7, 8, 2, 64), "normal_with_eps")
    sd = collections.OrderedDict((k, v.clone()) for k, v in vae.state_dict().items())
    ckpt = {"epoch": 7, "global_step": 123, "pytorch-lightning_version": "1.6.5", "state_dict": sd,
            "hyper_parameters": {"generator_network": _TrainingOnlyNet(), "generator_distribution": Dist.normal_with_eps,
                                 "lr_generator": 1e-3},
            "optimizer_states": [{"state": {0: {"exp_avg": torch.randn(2, 3)}}, "param_groups": [{"lr": 1e-3}]}],
            "callbacks": {"ModelCheckpoint": {"best_model_score": torch.tensor(1.5)}}}
    path = str(tmp_path / "epoch=7-step=123.ckpt")
    torch.save(ckpt, path)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)
    got = load_state_dict_from_checkpoint(path)
    assert list(got.keys()) == list(sd.keys())
    assert all(torch.equal(got[k], sd[k]) and got[k].dtype == sd[k].dtype for k in sd)
    vae2 = VAE(ResidualFCNetwork(4, 14, 1, 64), "normal_with_eps", ResidualFCNetwork(7, 8, 2, 64), "normal_with_eps")
    vae2.load_state_dict(got, strict=True)
    # a bare state dict (weights_only succeeds) and a non-checkpoint are handled too
    bare = str(tmp_path / "bare.pt")
    torch.save(sd, bare)
    assert torch.equal(load_state_dict_from_checkpoint(bare)["generator_network.final_layer.bias"],
                       sd["generator_network.final_layer.bias"])
    bad = str(tmp_path / "bad.ckpt")
    torch.save({"epoch": 1}, bad)
    with pytest.raises(ValueError):
        load_state_dict_from_checkpoint(bad)


def test_async_npz_writer(tmp_path):
    from vae_conditional_sampling_amd.utils.artifacts import AsyncNpzWriter
    w = AsyncNpzWriter(workers=2, max_pending=2)
    arrays = [np.random.default_rng(i).normal(size=(50, 40)).astype(np.float32) for i in range(5)]
    for i, a in enumerate(arrays):
        w.submit(str(tmp_path / f"imputations_{i}.npz"), imputations=a, data_idx=np.arange(50))
    w.wait()
    for i, a in enumerate(arrays):
        f = np.load(tmp_path / f"imputations_{i}.npz")
        assert np.array_equal(f["imputations"], a) and f["data_idx"].shape == (50,)
    w.submit(str(tmp_path / "no_such_dir" / "x.npz"), a=arrays[0])     # errors surface in wait()
    with pytest.raises(OSError):
        w.wait()
    w.close()


def test_data_index_counts_positions_among_the_kept_rows(tmp_path):
    """The reference wraps DatasetWithIndices around FullyMissingDataFilter (missing_data_module.py:436-440):
    the index of an example is its position after fully-missing rows were dropped, 0..n_kept-1."""
    import numpy as np
    import torch
    from vae_conditional_sampling_amd.data import MissingDataModule
    os.makedirs(tmp_path / "d2", exist_ok=True)
    np.save(tmp_path / "d2" / "test.npy", np.arange(400, dtype=np.float32).reshape(200, 2))
    dm = MissingDataModule("d2", 16, missingness="uniform", total_miss_test=0.5, setup_seed=3, data_root=str(tmp_path),
                           filter_fully_missing_test=True, num_first_datapoints_test=50)
    dm.setup("test")
    X, M, I = dm.test_data
    assert M.any(dim=1).all() and X.shape[0] == 50
    assert torch.equal(I, torch.arange(50))
    # rows were really dropped: the 50 kept rows are not the first 50 rows of the file
    assert not torch.equal(X[:, 0], torch.arange(0, 100, 2, dtype=torch.float32))


REFERENCE_CONFIGS = "/root/reference/configs"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_CONFIGS), reason="the reference tree exists in the build container only")
def test_every_reference_yaml_of_the_built_sampler_classes_constructs():
    """Boundary check: each of the reference's sampling configurations whose test_class is one of the five sampler
    classes built here constructs that class from its init_args unchanged (the YAML files are read as data)."""
    import glob
    import yaml
    from vae_conditional_sampling_amd.sampling import test_step_vae_sampling as TS
    built = {"TestVAELatentAdaptiveImportanceResampling", "TestVAEAdaptiveCollapsedMetropolisWithinGibbs",
             "TestVAEMetropolisWithinGibbs", "TestVAEPseudoGibbs", "TestVAEImportanceResampling"}
    n, seen = 0, set()
    for path in sorted(glob.glob(os.path.join(REFERENCE_CONFIGS, "**", "*.yaml"), recursive=True)):
        with open(path) as f:
            try:
                cfg = yaml.safe_load(f)
            except yaml.YAMLError:
                continue
        tc = (cfg or {}).get("test_class") if isinstance(cfg, dict) else None
        if not isinstance(tc, dict) or "class_path" not in tc:
            continue
        name = tc["class_path"].split(".")[-1]
        if name not in built:
            continue
        obj = getattr(TS, name)(**(tc.get("init_args") or {}))
        assert obj.max_iterations > 0, path
        n += 1
        seen.add(name)
    assert n >= 200 and seen == built, (n, seen)


@pytest.mark.parametrize("K", [1, 4])
def test_pointwise_metric_summary_matches_numpy(K):
    """The per-iteration mean / std / quantiles of a point-wise metric (test_epoch_end) against the numpy
    formulation of the reference's aggregation (irwg/sampling/test_step_vae_sampling.py:824-858: np.quantile with
    its default linear interpolation over all (n k) values of an iteration), ragged batches."""
    from vae_conditional_sampling_amd.sampling.imputation_metrics import pointwise_summary
    rs = np.random.RandomState(3)
    T = 7
    values = [rs.rand(T, n, K).astype(np.float32) ** 3 for n in (33, 64, 5)]
    qs = [0.1, 0.25, 0.5, 0.75, 0.9, 0.0, 1.0]
    mean, std, quant = pointwise_summary(values, qs)
    v = np.concatenate(values, axis=1).transpose(1, 2, 0)          # n k t
    over_data = v.mean(axis=0)
    assert np.allclose(mean, over_data.mean(axis=0), rtol=1e-5, atol=1e-7)
    ref_std = over_data.std(axis=0, ddof=1) if K != 1 else np.ones(T)
    assert np.allclose(std, ref_std, rtol=1e-4, atol=1e-7)
    ref_q = np.quantile(v.reshape(-1, T), axis=0, q=qs)
    assert quant.shape == ref_q.shape and np.allclose(quant, ref_q, rtol=1e-6, atol=1e-7)
    m2, s2, q2 = pointwise_summary(values, [])
    assert q2.shape == (0, T) and np.allclose(m2, mean)
    # NaN truths (nan_miss data) give NaN errors for some values of some iterations: numpy's quantiles of such an
    # iteration are NaN, the others are untouched
    values[1][2, 5, 0] = np.nan
    values[0][4, :, :] = np.nan
    _, _, quant = pointwise_summary(values, qs)
    ref_q = np.quantile(np.concatenate(values, axis=1).transpose(1, 2, 0).reshape(-1, T), axis=0, q=qs)
    assert np.isnan(ref_q[:, [2, 4]]).all() and np.array_equal(np.isnan(quant), np.isnan(ref_q))
    assert np.allclose(quant, ref_q, rtol=1e-6, atol=1e-7, equal_nan=True)


def test_parallel_npz_writer_matches_numpy(tmp_path):
    """utils.artifacts.savez_compressed_parallel / AsyncNpzWriter write the container np.savez_compressed writes:
    np.load returns the same keys, dtypes, shapes and contents (several compression pieces and write slices, a
    bool array, an empty list, a scalar, Fortran order)."""
    from concurrent.futures import ThreadPoolExecutor

    from vae_conditional_sampling_amd.utils import artifacts
    rs = np.random.RandomState(0)
    big = np.repeat(rs.randn(3, 200000).astype(np.float32), 7, axis=0)           # 16.8 MB, compressible
    arrays = dict(imputations=big, masks=rs.rand(50, 1, 43) > 0.5, latents=[], data_idx=np.arange(50),
                  scalar=np.float64(2.5), fort=np.asfortranarray(rs.rand(5, 6)))
    old_piece, old_slice = artifacts._PIECE, artifacts._SLICE
    artifacts._PIECE, artifacts._SLICE = 1 << 20, 5 << 20                          # 17 pieces, 4 write slices
    try:
        with ThreadPoolExecutor(4) as pool:
            artifacts.savez_compressed_parallel(tmp_path / "a.npz", pool, **arrays)
    finally:
        artifacts._PIECE, artifacts._SLICE = old_piece, old_slice
    np.savez_compressed(tmp_path / "b.npz", **arrays)
    w = artifacts.AsyncNpzWriter(workers=2, max_pending=2, compress_threads=3)
    w.submit(str(tmp_path / "c.npz"), **arrays)
    w.close()
    ref = np.load(tmp_path / "b.npz")
    for name in ("a.npz", "c.npz"):
        got = np.load(tmp_path / name)
        assert sorted(got.files) == sorted(ref.files)
        for k in ref.files:
            assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k])
    size = lambda n: (tmp_path / n).stat().st_size
    assert size("a.npz") < 1.1 * size("b.npz")     # independent pieces cost little compression


def test_row_encoder_writes_what_numpy_reads(tmp_path):
    """Float arrays of short rows (the kept imputations [T,B,K,D]) are deflated by the library's own encoder
    (cs_deflate_rows, host code): np.load must return them bit for bit, including NaNs, across pieces, write slices
    and 256 KiB block boundaries, and the files must not be larger than zlib's by more than a few percent."""
    from concurrent.futures import ThreadPoolExecutor

    from vae_conditional_sampling_amd.utils import artifacts
    rs = np.random.RandomState(1)
    T, B, K, D = 40, 300, 4, 43
    Xt = rs.randn(1, B, 1, D).astype(np.float32)
    M = rs.rand(1, B, 1, D) < 0.5
    imps = np.where(M, Xt, rs.randn(T, B, K, D).astype(np.float32)).astype(np.float32)      # 8.3 MB
    imps[3, 7, 2, 5] = np.nan
    wide = rs.randn(40, 8192).astype(np.float32)                   # rows of exactly 32768 bytes: the largest distance
    dbl = np.repeat(rs.randn(50, 1, 30), 90, axis=1)               # float64, every row repeats: long matches
    const = np.zeros((600, 500), np.float32)                       # one symbol only
    arrays = dict(imputations=imps, wide=wide, dbl=dbl, const=const, masks=M[0], latents=[])
    assert artifacts.row_stride_of(imps) == 172 and artifacts.row_stride_of(wide) == 32768
    assert artifacts.row_stride_of(rs.randn(4, 10000)) is None and artifacts.row_stride_of(M[0]) is None
    assert artifacts.row_stride_of(imps[:2]) is None               # small arrays stay with zlib
    old_piece, old_slice = artifacts._PIECE, artifacts._SLICE
    for piece, slc, name in ((old_piece, old_slice, "a.npz"), (300000, 1000000, "b.npz"), (4097, 4097 * 5, "c.npz")):
        artifacts._PIECE, artifacts._SLICE = piece, slc
        try:
            with ThreadPoolExecutor(4) as pool:
                artifacts.savez_compressed_parallel(tmp_path / name, pool, **arrays)
        finally:
            artifacts._PIECE, artifacts._SLICE = old_piece, old_slice
        got = np.load(tmp_path / name)
        assert sorted(got.files) == sorted(arrays)
        for k, v in arrays.items():
            v = np.asarray(v)
            assert got[k].dtype == v.dtype and got[k].shape == v.shape
            assert got[k].tobytes() == v.tobytes(), (name, k)
    np.savez_compressed(tmp_path / "ref.npz", **arrays)
    size = lambda n: (tmp_path / n).stat().st_size
    assert size("a.npz") < 1.05 * size("ref.npz")
    # a piece on its own is a valid raw deflate stream once closed
    import zlib
    raw = imps.tobytes()[:100000]
    piece = artifacts.deflate_rows(memoryview(raw), 172, 0) + zlib.compressobj(6, zlib.DEFLATED, -15).flush(zlib.Z_FINISH)
    assert zlib.decompress(piece, -15) == raw


def test_crc32_and_its_combination_match_zlib():
    """cs_crc32 / cs_crc32_combine (host functions of the library) against zlib.crc32: unaligned starts, lengths around
    the 8-byte step, continuation from a running value, and pieces folded together in order."""
    import zlib

    from vae_conditional_sampling_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(3)
    for n in (0, 1, 7, 8, 9, 63, 1000, 123457):
        b = rs.randint(0, 256, n).astype(np.uint8)
        for off in (0, 3):
            buf = np.empty(n + 8, np.uint8)
            v = buf[off:off + n]
            v[:] = b
            assert L.cs_crc32(v.ctypes.data if n else None, n, 0) == zlib.crc32(b.tobytes())
            assert L.cs_crc32(v.ctypes.data if n else None, n, 0xDEADBEEF) == zlib.crc32(b.tobytes(), 0xDEADBEEF)
        cuts = sorted(set([0, n // 3, n // 2, n]))
        crc = 0
        for lo, hi in zip([0] + cuts, cuts + [n]):
            crc = L.cs_crc32_combine(crc, zlib.crc32(b[lo:hi].tobytes()), hi - lo)
        assert crc == zlib.crc32(b.tobytes())


def test_sharded_artifact_merge_passes_switched_off_entries_through(tmp_path):
    """An option that is off is stored as an empty list in the artefact files (``latents`` without
    ``store_latents``), also where the example axis of that key is 0 (final-resampling files): merging shards,
    gathering them and cutting them into the one-process batches must hand such an entry on unchanged, while a rank
    without examples still contributes its (empty) per-example arrays."""
    from vae_conditional_sampling_amd import distributed as D
    stem = "irwg_imputations_after_final_resampling"
    mk = lambda lo, hi: {stem: dict(imputations=np.arange(lo, hi, dtype=np.float32)[None, :, None, None] + np.zeros((1, 1, 3, 2), np.float32),
                                    latents=np.asarray([]), true_X=np.zeros((hi - lo, 1, 2), np.float32),
                                    masks=np.ones((hi - lo, 1, 2), bool), data_idx=np.arange(lo, hi))}
    merged = D.merge_shards([mk(0, 5), mk(5, 5), mk(5, 9)])          # the middle rank holds no example
    assert merged[stem]["latents"].shape == (0,) and merged[stem]["imputations"].shape == (1, 9, 3, 2)
    assert np.array_equal(merged[stem]["data_idx"], np.arange(9))
    same = D.gather_shard_arrays(mk(0, 9), 9, "cpu")                 # no process group: the identity
    assert same[stem]["latents"].shape == (0,) and same[stem]["imputations"].shape == (1, 9, 3, 2)
    paths = D.write_batches(merged, 4, str(tmp_path), save=np.savez)
    assert len(paths) == 3
    f = np.load(paths[2])
    assert f["latents"].shape == (0,) and f["imputations"].shape == (1, 1, 3, 2) and f["data_idx"].tolist() == [8]


def test_weight_transform_accepts_the_reference_callable():
    """The function-level entry points take ``weight_transform`` as the reference passes it: a callable called with
    ``log_weights=`` that multiplies them by a number (its polynomial transform); other callables raise."""
    import functools

    from vae_conditional_sampling_amd.sampling.vae_imputation import _weight_alpha

    def polynomial_transform(log_weights, iter, max_iter):        # the reference's transform, restated
        alpha = min((iter + 10) / max_iter, 1)
        return log_weights * alpha

    assert _weight_alpha(None) is None and _weight_alpha(0.25) == 0.25
    assert abs(_weight_alpha(functools.partial(polynomial_transform, iter=5, max_iter=60)) - 0.25) < 1e-7
    assert _weight_alpha(functools.partial(polynomial_transform, iter=500, max_iter=60)) == 1.0
    with pytest.raises(NotImplementedError):
        _weight_alpha(lambda log_weights: log_weights - 1.0)
    with pytest.raises(NotImplementedError):
        _weight_alpha("poly")
