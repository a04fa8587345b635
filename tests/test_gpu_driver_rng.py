"""GPU: counter-RNG mode of the engine, sharding invariance and the YAML driver end to end."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import assert_close, make_model, seeded_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host(series):
    """A batch's point-wise diagnostic series as a numpy array (a one-process run keeps it in device memory)."""
    return series.cpu().numpy() if torch.is_tensor(series) else np.asarray(series)


def test_philox_uniforms_are_bit_exact_with_the_c_restatement():
    from oracle import c_binding
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling import engine
    L = c_binding.load()
    rng = _lib.make_rng(0x1234567890ABCDEF, 7, 1000)
    dev = torch.device("cuda")
    u64 = engine.rng_uniform64(rng, _lib.STREAM_URES, 33, 9, dev).cpu().numpy()
    assert np.array_equal(u64, c_binding.uniform64_rows(L, 0x1234567890ABCDEF, _lib.STREAM_URES, 7, 1000, 33, 9))
    u32 = engine.rng_uniform(rng, _lib.STREAM_UACC, 17, 5, dev).cpu().numpy()
    assert np.array_equal(u32, c_binding.uniform_rows(L, 0x1234567890ABCDEF, _lib.STREAM_UACC, 7, 1000, 17, 5))
    assert (u64 >= 0).all() and (u64 < 1).all() and (u32 > 0).all() and (u32 < 1).all()
    n = engine.rng_normal(rng, _lib.STREAM_Z, 4096, 32, dev)
    assert abs(float(n.mean())) < 0.02 and abs(float(n.std()) - 1.0) < 0.02


def test_lair_counter_rng_mode_matches_oracle_on_the_same_draws():
    """Without injected draws the kernels generate them (Philox).  The same numbers, dumped
    through cs_rng_*, are fed to the oracle: every output must agree."""
    from oracle import cs_oracle as O
    from vae_conditional_sampling_amd import _lib
    from vae_conditional_sampling_amd.sampling import engine
    from vae_conditional_sampling_amd.sampling.vae_imputation import importance_resampling_gibbs_iteration
    B, K, R, D, Zd, H = 96, 4, 1, 43, 32, 256
    Kp = K + R
    enc = O.make_resfc_weights(D, 2 * Zd, 2, H, 21)
    dec = O.make_resfc_weights(Zd, 2 * D, 1, H, 22)
    model = make_model(enc, dec)
    X, M = seeded_inputs(B, K, D, 23)
    off, it, seed = 500, 3, 99
    rng = _lib.make_rng(seed, it, off)
    out = importance_resampling_gibbs_iteration(model, X.cuda(), M.cuda(), 1, num_prior_replenish_proposals=R, rng=rng)
    X_gpu, lw_gpu = out[0].clone(), out[7].clone()
    ws = model._cs_workspaces[("lair", B, K, R, 1, "cuda:0")]
    idx_gpu = ws.idx.cpu().clone()
    dev = torch.device("cuda")
    draws = {
        "eps_z": engine.rng_normal(_lib.make_rng(seed, it, off * Kp), _lib.STREAM_Z, B * Kp, Zd, dev).reshape(1, B, Kp, Zd).cpu(),
        "u_res": engine.rng_uniform64(_lib.make_rng(seed, it, off), _lib.STREAM_URES, B, Kp, dev).cpu(),
        "eps_x": engine.rng_normal(_lib.make_rng(seed, it, off * K), _lib.STREAM_X, B * K, D, dev).reshape(B, K, D).cpu(),
    }
    o = O.lair_iteration(enc, dec, X, M, draws, num_prior_replenish_proposals=R)
    assert_close(ws.Z, o["Z_imp"], "Z_imp")
    assert_close(lw_gpu, o["log_weights"], "log_weights")
    assert torch.equal(idx_gpu, o["idx_global"])
    assert_close(X_gpu, o["X_new"], "X_new")

    # sharding invariance: two half batches with their global offsets give the same result bit for bit
    h = B // 2
    o1 = importance_resampling_gibbs_iteration(model, X[:h].cuda(), M[:h].cuda(), 1, num_prior_replenish_proposals=R,
                                               rng=_lib.make_rng(seed, it, off))[0].clone()
    o2 = importance_resampling_gibbs_iteration(model, X[h:].cuda(), M[h:].cuda(), 1, num_prior_replenish_proposals=R,
                                               rng=_lib.make_rng(seed, it, off + h))[0].clone()
    assert torch.equal(torch.cat([o1, o2]), X_gpu)


def test_yaml_driver_end_to_end_lair_and_acmwg(tmp_path):
    from vae_conditional_sampling_amd import driver
    common = ["data.synthetic_num_test=150", "data.num_first_datapoints_test=100", "data.batch_size=64",
              f"trainer.default_root_dir={tmp_path}"]
    cfg = driver.load_config(os.path.join(ROOT, "configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml"),
                             common + ["test_class.init_args.num_iterations=6"])
    tester, outs = driver.run(cfg)
    assert len(outs) == 2
    f0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    assert f0["imputations"].shape == (7, 64, 4, 43) and f0["masks"].shape == (64, 1, 43)
    assert f0["true_X"].shape == (64, 1, 43) and f0["data_idx"].shape == (64,)
    obs = np.broadcast_to(f0["masks"], (64, 4, 43))
    for t in range(7):  # observed entries never change
        assert np.array_equal(f0["imputations"][t][obs], np.broadcast_to(f0["true_X"], (64, 4, 43))[obs])
    assert np.isfinite(f0["imputations"]).all()
    f1 = np.load(os.path.join(tester.out_dir, "irwg_imputations_after_final_resampling_1.npz"))
    assert f1["imputations"].shape == (1, 36, 6 * 5, 43)
    assert all(0 < e <= 5 for e in outs[0]["is_ess"])
    # the configuration enables the RMSE / MAE / latent-Wasserstein diagnostics: [T, B, K] per batch
    for name in ("rmse", "mae", "latent_wass"):
        v = _host(outs[0][f"imp_metric/{name}"])
        assert v.shape == (6, 64, 4) and np.isfinite(v).all() and (v >= 0).all()
    ms = np.load(os.path.join(tester.out_dir, "metrics_summary.npz"))
    assert ms["rmse__mean__test"].shape == (6,) and ms["rmse__q0.5__test"].shape == (6,)
    assert ms["is_ess__test"].shape == (6,)
    # cross-check one entry of the RMSE against the stored imputations
    imp, tx, mk = f0["imputations"][3], f0["true_X"], f0["masks"]
    ref = np.sqrt((((imp - tx) ** 2) * ~mk).sum(-1) / np.maximum((~mk).sum(-1), 1))
    assert np.allclose(_host(outs[0]["imp_metric/rmse"])[2], ref, rtol=1e-5, atol=1e-6)

    cfg = driver.load_config(os.path.join(ROOT, "configs/uci_miniboone/sampling_u50/acmwg_k5_005prior_irwg_warmup.yaml"),
                             common + ["test_class.init_args.num_pseudo_warmup_iterations=2",
                                       "test_class.init_args.num_mwg_iterations=9"])
    tester, outs = driver.run(cfg)
    g0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    assert g0["imputations"].shape == (1 + 2 * 4 + 9, 64, 5, 43) and np.isfinite(g0["imputations"]).all()
    acc = outs[0]["acceptance"]
    assert len(acc) == 17 and all(0.0 <= a <= 1.0 for a in acc) and acc[0] == 1.0
    # the device history evolved like the reference queue: only the last warm-up iteration's
    # num_k_per_cluster imputations and the 9 MWG steps were enqueued (ring capacity 17)
    assert not tester.history.full and tester.history.tail_idx == 4 + 9 - 1
    av = tester.history.available_timesteps.cpu()
    assert int(av.min()) >= 4 and int(av.max()) <= 4 + 9 - 1


@pytest.mark.parametrize("dataset,D", [("hepmass", 21), ("gas", 8)])
def test_yaml_driver_whole_dataset_configs_k16(tmp_path, dataset, D):
    """The HEPMASS / GAS configurations (K=16 chains + 1 replenishing proposal over the whole test set, last
    iteration and 5 final-resampling imputations stored) through the YAML driver, cut down to 300 synthetic rows."""
    from vae_conditional_sampling_amd import driver
    cfg = driver.load_config(
        os.path.join(ROOT, f"configs/uci_{dataset}/sampling_u50/lair_k16_dmis_gr_mult_1replenish_wholedataset.yaml"),
        ["data.synthetic_num_test=300", "data.batch_size=128", f"trainer.default_root_dir={tmp_path}",
         "test_class.init_args.num_iterations=7"])
    tester, outs = driver.run(cfg)
    f0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    B0 = f0["true_X"].shape[0]
    # store_last_n_iterations: 1 keeps the yields t >= max_iterations - 1, i.e. the last two (the reference's rule,
    # irwg/sampling/test_step_vae_sampling.py:386)
    assert f0["imputations"].shape == (2, B0, 16, D) and np.isfinite(f0["imputations"]).all()
    obs = np.broadcast_to(f0["masks"], (B0, 16, D))
    assert np.array_equal(f0["imputations"][-1][obs], np.broadcast_to(f0["true_X"], (B0, 16, D))[obs])
    f1 = np.load(os.path.join(tester.out_dir, "irwg_imputations_after_final_resampling_0.npz"))
    assert f1["imputations"].shape == (1, B0, 5, D) and np.isfinite(f1["imputations"]).all()
    assert all(0 < e <= 17 for e in outs[0]["is_ess"])
    for name in ("rmse", "mae"):
        v = _host(outs[0][f"imp_metric/{name}"])
        assert v.shape == (7, B0, 16) and np.isfinite(v).all()


@pytest.mark.parametrize("config,extra", [
    ("configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml",
     ["test_class.init_args.num_iterations=5", "test_class.init_args.store_latents=True"]),
    ("configs/uci_miniboone/sampling_u50/acmwg_k5_005prior_irwg_warmup.yaml",
     ["test_class.init_args.num_pseudo_warmup_iterations=2", "test_class.init_args.num_mwg_iterations=7"])])
def test_sharded_driver_runs_reassemble_the_one_process_artifacts(tmp_path, config, extra):
    """driver.run as rank 0 of 2 and rank 1 of 2 (one after the other in this process, no process group: the
    gather is distributed.merge_shards) writes, once merged and cut into the one-process batches, exactly the
    files of the unsharded run: the counter RNG is keyed by the global example index, shard boundaries need
    not fall on batch boundaries (101 examples, batches of 32: shards of 51 and 50)."""
    from vae_conditional_sampling_amd import distributed, driver
    common = ["data.synthetic_num_test=150", "data.num_first_datapoints_test=101", "data.batch_size=32"] + extra
    cfg = lambda root: driver.load_config(os.path.join(ROOT, config), common + [f"trainer.default_root_dir={root}"])
    whole, _ = driver.run(cfg(tmp_path / "one"))
    shards = []
    for rank in range(2):
        t, _ = driver.run(cfg(tmp_path / "two"), rank=rank, world_size=2)
        assert (t.out_dir is None) == (rank == 1)
        shards.append(t.collector.shard_arrays())
    out2 = str(tmp_path / "two" / "merged")
    distributed.write_batches(distributed.merge_shards(shards), 32, out2)
    names = sorted(n for n in os.listdir(whole.out_dir) if n.startswith(("imputations_", "irwg_")))
    assert names == sorted(os.listdir(out2)) and len(names) >= 4
    for name in names:
        a, b = np.load(os.path.join(whole.out_dir, name)), np.load(os.path.join(out2, name))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), f"{name}[{k}] differs between the runs"


def test_sampler_runs_stretches_in_one_call_with_identical_results(tmp_path):
    """Under the harness the LAIR sampler runs max_iterations_per_call iterations per library call (cs_lair_run_observed);
    stored imputations, final-resampling imputations and the ESS series must equal those of the step-by-step run
    (bit for bit for the chain, to rounding for the batch means).  Here: only the last iterations stored, no
    per-iteration diagnostics."""
    from vae_conditional_sampling_amd import driver
    from vae_conditional_sampling_amd.sampling import test_step_vae_sampling as TS
    over = ["data.synthetic_num_test=150", "data.num_first_datapoints_test=70", "data.batch_size=64",
            "test_class.init_args.num_iterations=23", "test_class.init_args.store_last_n_iterations=2",
            "test_class.init_args.compute_rmse_metric=False", "test_class.init_args.compute_mae_metric=False",
            "test_class.init_args.compute_latent_wass_metric=False"]
    res = []
    calls = []
    orig_run = TS.engine.lair_run
    for cap in (1, 8):
        cfg = driver.load_config(os.path.join(ROOT, "configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml"),
                                 over + [f"trainer.default_root_dir={tmp_path / str(cap)}"])
        TS.engine.lair_run = lambda *a, **k: (calls.append((cap, a[5])), orig_run(*a, **k))[1]
        old_init = TS.TestVAESamplerBase.on_test_start

        def patched(self, _cap=cap, _old=old_init):
            _old(self)
            self.max_iterations_per_call = _cap
        TS.TestVAESamplerBase.on_test_start = patched
        try:
            tester, outs = driver.run(cfg)
        finally:
            TS.TestVAESamplerBase.on_test_start = old_init
            TS.engine.lair_run = orig_run
        files = {n: dict(np.load(os.path.join(tester.out_dir, n))) for n in sorted(os.listdir(tester.out_dir)) if n.endswith(".npz")}
        res.append((files, outs))
    assert not [c for c in calls if c[0] == 1] and [c[1] for c in calls if c[0] == 8][:3] == [8, 8, 7]   # 23 iterations
    (f1, o1), (f8, o8) = res
    assert sorted(f1) == sorted(f8) and "irwg_imputations_after_final_resampling_1.npz" in f1
    for n in f1:
        for k in f1[n]:
            if n == "metrics_summary.npz":
                np.testing.assert_allclose(f1[n][k], f8[n][k], rtol=1e-6)
            else:
                assert np.array_equal(f1[n][k], f8[n][k]), f"{n}[{k}] differs"
    assert f1["imputations_0.npz"]["imputations"].shape == (3, 64, 4, 43)
    np.testing.assert_allclose(o1[0]["is_ess"], o8[0]["is_ess"], rtol=1e-6)


def test_sampler_keeps_the_stored_iterations_of_a_stretch(tmp_path):
    """store_every_n_iterations = 10: all 31 iterations run in one library call, which keeps every iteration's
    imputations for the harness to pick the stored ones (0, 10, 20, 30) from."""
    from vae_conditional_sampling_amd import driver
    from vae_conditional_sampling_amd.sampling import test_step_vae_sampling as TS
    cfg = driver.load_config(os.path.join(ROOT, "configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml"),
                             ["data.synthetic_num_test=40", "data.num_first_datapoints_test=40", "data.batch_size=64",
                              "test_class.init_args.num_iterations=31", "test_class.init_args.store_every_n_iterations=10",
                              "test_class.init_args.compute_rmse_metric=False", "test_class.init_args.compute_mae_metric=False",
                              "test_class.init_args.compute_latent_wass_metric=False",
                              f"trainer.default_root_dir={tmp_path}"])
    calls = []
    orig_run = TS.engine.lair_run
    TS.engine.lair_run = lambda *a, **k: (calls.append(a[5]), orig_run(*a, **k))[1]
    try:
        tester, outs = driver.run(cfg)
    finally:
        TS.engine.lair_run = orig_run
    assert calls == [31], calls
    f0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    assert f0["imputations"].shape == (1 + 4, 40, 4, 43) and np.isfinite(f0["imputations"]).all()
    obs = np.broadcast_to(f0["masks"], (40, 4, 43))
    assert all(np.array_equal(f0["imputations"][i][obs], np.broadcast_to(f0["true_X"], (40, 4, 43))[obs]) for i in range(5))
    assert not np.array_equal(f0["imputations"][1], f0["imputations"][2])


@pytest.mark.parametrize("config,extra,batches", [
    ("configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml", ["test_class.init_args.num_iterations=21"], 2),
    ("configs/uci_miniboone/sampling_u50/acmwg_k5_005prior_irwg_warmup.yaml",
     ["test_class.init_args.num_pseudo_warmup_iterations=3", "test_class.init_args.num_mwg_iterations=19"], 2),
    ("configs/uci_miniboone/sampling_u50/acmwg_k5_005prior_irwg_warmup.yaml",
     ["test_class.class_path=irwg.sampling.test_step_vae_sampling.TestVAEMetropolisWithinGibbs",
      "test_class.init_args.prior_mixture_probability=null", "test_class.init_args.enqueue_allwarmup_samples_to_history=null",
      "test_class.init_args.warmup_method=pseudo_gibbs", "test_class.init_args.num_pseudo_warmup_iterations=2",
      "test_class.init_args.num_mwg_iterations=11"], 1)])
def test_observed_runs_equal_single_steps_with_all_diagnostics(tmp_path, config, extra, batches):
    """The published configurations keep every iteration and compute RMSE / MAE / latent distance on each: run
    8 iterations per library call (the run keeps the imputations and computes the diagnostics itself) against one
    step per call (the harness does).  Every artefact and every diagnostic must be identical bit for bit; batch
    means of the acceptance / ESS series to rounding."""
    from vae_conditional_sampling_amd import driver
    from vae_conditional_sampling_amd.sampling import test_step_vae_sampling as TS
    over = ["data.synthetic_num_test=100", f"data.num_first_datapoints_test={40 * batches}", "data.batch_size=40"] + extra
    over = [o for o in over if not o.endswith("=null")]
    drop = [o.split("=")[0].rsplit(".", 1)[1] for o in extra if o.endswith("=null")]
    res = []
    for cap in (1, 8):
        cfg = driver.load_config(os.path.join(ROOT, config), over + [f"trainer.default_root_dir={tmp_path / str(cap)}"])
        for k in drop:
            cfg["test_class"]["init_args"].pop(k, None)
        old_init = TS.TestVAESamplerBase.on_test_start

        def patched(self, _cap=cap, _old=old_init):
            _old(self)
            self.max_iterations_per_call = _cap
        TS.TestVAESamplerBase.on_test_start = patched
        try:
            tester, outs = driver.run(cfg)
        finally:
            TS.TestVAESamplerBase.on_test_start = old_init
        files = {n: dict(np.load(os.path.join(tester.out_dir, n))) for n in sorted(os.listdir(tester.out_dir)) if n.endswith(".npz")}
        res.append((files, outs))
    (f1, o1), (f8, o8) = res
    assert sorted(f1) == sorted(f8) and "imputations_0.npz" in f1 and len(o1) == len(o8) == batches
    for n in f1:
        for k in f1[n]:
            if n == "metrics_summary.npz" and not any(m in k for m in ("rmse", "mae", "latent_wass")):
                np.testing.assert_allclose(f1[n][k], f8[n][k], rtol=1e-6)
            else:
                assert np.array_equal(f1[n][k], f8[n][k], equal_nan=True), f"{n}[{k}] differs"
    T = f1["imputations_0.npz"]["imputations"].shape[0] - 1
    for a, b in zip(o1, o8):
        for name in ("rmse", "mae", "latent_wass"):
            va, vb = _host(a[f"imp_metric/{name}"]), _host(b[f"imp_metric/{name}"])
            assert va.shape == (T, 40, va.shape[2]) and np.array_equal(va, vb), name
        for key in a:
            if not key.startswith("imp_metric/"):
                np.testing.assert_allclose(a[key], b[key], rtol=1e-6, err_msg=key)


def test_yaml_driver_selects_the_arithmetic_per_run(tmp_path):
    """`mlp_arith: f16x2` in a configuration pins the arithmetic on the run's network handles (not process-wide)."""
    from vae_conditional_sampling_amd import _lib, driver
    prev = _lib.get_mlp_arith()
    cfg = driver.load_config(os.path.join(ROOT, "configs/uci_miniboone/sampling_u50/lair_k4_dmis_gr_mult_1replenish.yaml"),
                             ["data.synthetic_num_test=70", "data.num_first_datapoints_test=64", "data.batch_size=64",
                              f"trainer.default_root_dir={tmp_path}", "test_class.init_args.num_iterations=4",
                              "mlp_arith=f16x2"])
    tester, outs = driver.run(cfg)
    assert tester.model.var_latent_network.packed().get_arith() == "f16x2"
    assert tester.model.generator_network.packed().get_arith() == "f16x2"
    assert _lib.get_mlp_arith() == prev
    f0 = np.load(os.path.join(tester.out_dir, "imputations_0.npz"))
    assert np.isfinite(f0["imputations"]).all()


def test_bench_multi_rank_launch_dry_run():
    """`bench.py --gpus N --dry-run`: fresh child processes, one per rank (the parent never touches a GPU), each bound to
    its device and joined in a process group.  On this one-GPU box the two ranks share GPU 0 over gloo (rehearsal);
    without the rehearsal switch the same command must refuse to report a smaller run and exit non-zero, and a rank that
    cannot join its process group must make the whole launch fail."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, CSVAE_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dry_run"] and line["n_gpus"] == 2 and [x["rank"] for x in line["ranks"]] == [0, 1]
    assert all(x["device"] == "cuda:0" for x in line["ranks"])
    if torch.cuda.device_count() < 2:
        env2 = {k: v for k, v in env.items() if k != "CSVAE_BENCH_REHEARSAL"}
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env2, capture_output=True, text=True, timeout=170)
        assert r.returncode != 0 and "GPU(s) are visible" in r.stderr
    # a rank whose rendezvous cannot complete (it waits for a peer that never starts) ends non-zero
    env3 = dict(env, RANK="1", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1",
                CSVAE_BENCH_INIT_TIMEOUT="5")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env3, capture_output=True, text=True, timeout=170)
    assert r.returncode != 0
