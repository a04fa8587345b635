"""CPU, world_size 2 (gloo): the multi-process layout of the path -- contiguous example shards,
no collective inside the loop, one all-gather of the final imputations -- reassembles the
global order, including unequal shard sizes."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_conditional_sampling_amd.data.missing_data_module import shard_range
    from vae_conditional_sampling_amd.distributed import gather_sharded
    lo, hi = shard_range(n_total, rank, world)
    full = torch.arange(n_total * 2 * 3, dtype=torch.float32).reshape(n_total, 2, 3)
    local = full[lo:hi].clone() + 0.0   # this rank's "final imputations" [n_local,K,D]
    got = gather_sharded(local, n_total)
    assert torch.equal(got, full), f"rank {rank}: gathered tensor differs"
    lw = torch.arange(n_total, dtype=torch.float32)[lo:hi]
    assert torch.equal(gather_sharded(lw, n_total), torch.arange(n_total, dtype=torch.float32))
    torch.save(got, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_gather_of_sharded_imputations_world2(tmp_path):
    for n_total in (7, 8):
        mp.spawn(_worker, args=(2, _free_port(), n_total, str(tmp_path)), nprocs=2, join=True)
        a = torch.load(tmp_path / "r0.pt"); b = torch.load(tmp_path / "r1.pt")
        assert torch.equal(a, b) and a.shape[0] == n_total


LAIR_KW = dict(num_iterations=3, num_imp_samples=1, weighting_scheme="dmis", resampling_scheme="gr",
               resampling_method="multinomial", num_prior_replenish_proposals=1, resample_final_imps=True,
               store_norm_log_weights=True, store_last_n_iterations=-1, store_every_n_iterations=1,
               imputation_metric_quantiles=[0.5], compute_f1_metric=False, compute_inv_f1_metric=False,
               compute_bin_accuracy_metric=False, compute_rmse_metric=True, compute_ssim_metric=False,
               compute_latent_wass_metric=False, num_copies=2, imputation_fn="vae_samples")


def _fake_epoch(tester, lo, hi, batch_size, T=3, K=2, D=5):
    """What test_step / on_test_batch_end / sampler leave behind for the examples [lo, hi): arrays whose
    values are a function of the GLOBAL example index only (as the counter RNG makes the real ones)."""
    import numpy as np
    tester.on_test_epoch_start()
    for b, s in enumerate(range(lo, hi, batch_size)):
        e = min(s + batch_size, hi)
        ex = np.arange(s, e, dtype=np.float32)
        imps = (ex[None, :, None, None] * 100 + np.arange(T + 1, dtype=np.float32)[:, None, None, None] * 10
                + np.arange(K, dtype=np.float32)[None, None, :, None] + np.zeros((1, 1, 1, D), np.float32))
        tester._save_npz(f"imputations_{b}.npz", imputations=imps, latents=[], true_X=ex[:, None, None] + np.zeros((1, 1, D), np.float32),
                         masks=(np.arange(D)[None, None, :] + ex[:, None, None].astype(int)) % 2 == 0, data_idx=np.arange(s, e))
        tester._save_npz(f"irwg_imputations_after_final_resampling_{b}.npz", imputations=imps[:1].repeat(3, axis=2),
                         latents=ex[:, None, None] + np.zeros((1, 6, 4), np.float32), true_X=ex[:, None, None] + np.zeros((1, 1, D), np.float32),
                         masks=np.ones((e - s, 1, D), bool), data_idx=np.arange(s, e))
        tester.metrics["is_ess"].append([float(ex.mean()) + t for t in range(T)])
        tester.metrics["imp_metric/rmse"].append(ex[None, :, None] + np.zeros((T, 1, K), np.float32))
        tester.metrics["num_X"].append(e - s)
        tester.norm_log_weights.append(ex[:, None, None] + np.zeros((1, T, 3), np.float32))


ACMWG_KW = dict(num_pseudo_warmup_iterations=1, num_mwg_iterations=2, prior_mixture_probability=0.05,
                warmup_method="pseudo_gibbs", store_last_n_iterations=-1, store_every_n_iterations=1,
                imputation_metric_quantiles=[0.5], compute_f1_metric=False, compute_inv_f1_metric=False,
                compute_bin_accuracy_metric=False, compute_rmse_metric=True, compute_ssim_metric=False,
                compute_latent_wass_metric=False, num_copies=2, imputation_fn="vae_samples")


def _fake_acmwg_epoch(tester, lo, hi, batch_size, T=3, K=2, D=5):
    """The artefact set of an AC-MWG run (imputations_{b}.npz with latents, the acceptance series): values a function
    of the global example index only."""
    import numpy as np
    tester.on_test_epoch_start()
    for b, s in enumerate(range(lo, hi, batch_size)):
        e = min(s + batch_size, hi)
        ex = np.arange(s, e, dtype=np.float32)
        imps = (ex[None, :, None, None] * 100 + np.arange(T + 1, dtype=np.float32)[:, None, None, None] * 10
                + np.arange(K, dtype=np.float32)[None, None, :, None] + np.zeros((1, 1, 1, D), np.float32))
        tester._save_npz(f"imputations_{b}.npz", imputations=imps, latents=imps[:, :, :, :3] * 0.5,
                         true_X=ex[:, None, None] + np.zeros((1, 1, D), np.float32),
                         masks=(np.arange(D)[None, None, :] + ex[:, None, None].astype(int)) % 2 == 0, data_idx=np.arange(s, e))
        tester.metrics["acceptance"].append([0.25 + 0.0 * t for t in range(T)])
        tester.metrics["imp_metric/rmse"].append(ex[None, :, None] + np.zeros((T, 1, K), np.float32))
        tester.metrics["num_X"].append(e - s)


def _driver_worker(rank, world, port, n_total, batch_size, root, sampler="lair"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    from vae_conditional_sampling_amd import driver
    from vae_conditional_sampling_amd.data.missing_data_module import shard_range
    from vae_conditional_sampling_amd.sampling.test_step_vae_sampling import (
        TestVAEAdaptiveCollapsedMetropolisWithinGibbs, TestVAELatentAdaptiveImportanceResampling)
    if rank == 1:
        time.sleep(1.0)   # a late rank must not open a version directory of its own
    if sampler == "lair":
        tester = TestVAELatentAdaptiveImportanceResampling(**LAIR_KW)
    else:
        tester = TestVAEAdaptiveCollapsedMetropolisWithinGibbs(**ACMWG_KW)
    driver.prepare_outputs(tester, root, rank, world > 1)
    lo, hi = shard_range(n_total, rank, world)
    (_fake_epoch if sampler == "lair" else _fake_acmwg_epoch)(tester, lo, hi, batch_size)
    driver.finish_run(tester, [], n_total, batch_size, rank, world > 1, torch.device("cpu"))
    if world > 1:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("sampler,n_total,world", [("lair", 11, 2), ("acmwg", 11, 2), ("lair", 1, 2), ("acmwg", 2, 3)])
def test_sharded_driver_writes_the_files_of_a_one_process_run(tmp_path, sampler, n_total, world):
    """prepare_outputs / finish_run of the driver under gloo, world size 2 (3), unequal shards, shard boundaries
    that do not fall on batch boundaries: rank 0 alone writes, into ONE version directory, and every file
    equals the one a single process writes for the same examples -- for the artefact sets of LAIR and of AC-MWG,
    and when there are fewer examples than ranks (a rank with an empty shard takes part in every collective)."""
    import glob
    import numpy as np
    bs = 4
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    _driver_worker(0, 1, 0, n_total, bs, one, sampler)
    mp.spawn(_driver_worker, args=(world, _free_port(), n_total, bs, two, sampler), nprocs=world, join=True)
    assert sorted(os.listdir(os.path.join(two, "lightning_logs"))) == ["version_0"]
    d1, d2 = (os.path.join(r, "lightning_logs", "version_0") for r in (one, two))
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d2)) and "imputations_0.npz" in names and "metrics_summary.npz" in names
    if sampler == "lair":
        assert "is_norm_log_weights.npz" in names and "irwg_imputations_after_final_resampling_0.npz" in names
    if n_total == 11:
        assert "imputations_2.npz" in names
    assert not glob.glob(os.path.join(two, "**", "rank_*"), recursive=True)
    for name in names:
        a, b = np.load(os.path.join(d1, name)), np.load(os.path.join(d2, name))
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            if name == "metrics_summary.npz" and (k.startswith("is_ess") or k.startswith("acceptance")):
                # a mean of per-batch means weighted by batch size: the batches differ, the mean does not
                np.testing.assert_allclose(a[k], b[k], rtol=1e-12)
            else:
                assert np.array_equal(a[k], b[k]), (name, k)
