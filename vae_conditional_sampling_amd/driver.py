"""``python test.py --config=<yaml>`` driver: config -> data module -> sampler class -> batches.

Counterpart of the reference's ``test.py`` (:16-180) without Lightning / jsonargparse: the same
YAML keys (``seed_everything``, ``experiment_subdir_base``, ``data.*``, ``model``, ``model_path``,
``load_best``, ``test_class.{class_path,init_args}``, ``trainer.*``), the same experiment
directory ``{base}/seed_m{seed}_d{data.setup_seed}`` (irwg/utils/arg_utils.py:1-2) and the same
life-cycle of hooks around every batch.  ``test_class.class_path`` may name the reference's
module (``irwg.sampling.test_step_vae_sampling.<Class>``); it is resolved to this package's class
of the same name.

Sharded runs (one process per GPU, ``world_size`` > 1): every rank samples a contiguous block of the
test set with no collective in the loop and keeps its per-batch artefacts in memory; after the loop
they are gathered in global example order (one all-gather per stored array), rank 0 cuts them into
the batches of the one-process data loader and writes the same files a one-process run writes
(``imputations_{b}.npz``, ``irwg_imputations_after_final_resampling_{b}.npz``,
``metrics_summary.npz``, ``is_norm_log_weights.npz``) into one ``version_N`` directory chosen by
rank 0.  Without a process group (ranks run one after another in one process, as the tests do) the
gather is left to the caller: ``tester.collector.shard_arrays()`` + ``distributed.merge_shards``.

Keys that exist only here:
  ``model_config``  YAML with the network architecture (the reference's *training* config, whose
                    ``model.init_args`` block is read); the reference stores it in the checkpoint.
  ``random_init_seed``  build the model with seeded random weights when no checkpoint exists
                    (benchmarks / plumbing tests).
  ``data.synthetic``  shape-faithful synthetic data when the data files are absent.
"""
import argparse
import glob
import os

import torch
import yaml

from . import distributed
from .data import MissingDataModule
from .models import VAE, ResidualFCNetwork, ResNetDecoder, ResNetEncoder
from .sampling import test_step_vae_sampling
from .utils.checkpoint import load_state_dict_from_checkpoint
from .utils.hparams import AttrDict


def construct_experiment_subdir(hparams):
    return f"{hparams['experiment_subdir_base']}/seed_m{hparams['seed_everything']}_d{hparams['data']['setup_seed']}"


def load_config(path, overrides=None):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    for ov in overrides or []:
        key, val = ov.split("=", 1)
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    cfg.setdefault("trainer", {})
    cfg["trainer"] = cfg["trainer"] or {}
    return cfg


def resolve_test_class(class_path):
    name = class_path.split(".")[-1]
    cls = getattr(test_step_vae_sampling, name, None)
    if cls is None or not isinstance(cls, type):
        raise NotImplementedError(f"sampler class {class_path!r} is not implemented")
    return cls


def build_network(spec):
    name = spec["class_path"].split(".")[-1]
    a = dict(spec["init_args"])
    if name == "ResNetEncoder":
        return ResNetEncoder(**a)
    if name == "ResNetDecoder":
        return ResNetDecoder(**a)
    if name != "ResidualFCNetwork":
        raise NotImplementedError(f"network class {spec['class_path']!r} is not implemented")
    return ResidualFCNetwork(a.pop("input_dim"), a.pop("output_dim"), a.pop("num_residual_blocks"),
                             a.pop("residual_block_dim"), **a)


def build_model(model_cfg_path, device):
    with open(model_cfg_path) as f:
        mc = yaml.safe_load(f)["model"]["init_args"]
    gen = build_network(mc["generator_network"])
    enc = build_network(mc["var_latent_network"])
    model = VAE(gen, mc["generator_distribution"], enc, mc["var_latent_distribution"],
                encoder_use_mis_mask=mc.get("encoder_use_mis_mask", False),
                use_uniform_prior=mc.get("use_uniform_prior", False))
    return model.to(device)


def find_checkpoint(model_path, load_best):
    """test.py:111-123: latest ``version_{}``; a directory must hold exactly one checkpoint once
    ``last.ckpt`` is dropped."""
    root = model_path.split("version_")[0]
    versions = sorted(int(f.split("_")[1]) for f in os.listdir(root) if f.startswith("version_"))
    if len(versions) > 1:
        print("More than one version is available:", versions, ". Loading ", versions[-1])
    path = model_path.format(versions[-1])
    if os.path.isdir(path):
        models = os.listdir(path)
        if load_best and "last.ckpt" in models:
            models.remove("last.ckpt")
        if len(models) > 1:
            raise ValueError(f"Too many models in path: {path}")
        path = os.path.join(path, models[0])
    return path


def run(cfg, *, device=None, rank=0, world_size=1, gather_device=None):
    torch.manual_seed(int(cfg["seed_everything"]))
    device = device or torch.device("cuda", torch.cuda.current_device())
    trainer = cfg.get("trainer") or {}
    subdir = construct_experiment_subdir(cfg)
    root = trainer.get("default_root_dir")
    experiment_dir = f"./{subdir}" if root is None else f"{root}/{subdir}"

    datamodule = MissingDataModule(**cfg["data"], rank=rank, world_size=world_size)
    if cfg.get("model") != "VAE":
        raise NotImplementedError(f"model {cfg.get('model')!r} (only 'VAE')")
    TestClass = resolve_test_class(cfg["test_class"]["class_path"])
    tester = TestClass(**cfg["test_class"]["init_args"])
    if "model_config" not in cfg:
        raise KeyError("model_config: path of the YAML holding model.init_args (network architecture)")
    model = build_model(cfg["model_config"], device)
    if cfg.get("random_init_seed") is not None:
        g = torch.Generator().manual_seed(int(cfg["random_init_seed"]))
        with torch.no_grad():
            for p in model.parameters():
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / max(p.shape[-1], 1) ** 0.5)
    else:
        ckpt = find_checkpoint(cfg["model_path"], cfg.get("load_best"))
        print("Model path:", ckpt)
        model.load_state_dict(load_state_dict_from_checkpoint(ckpt), strict=True)
    if cfg.get("mlp_arith") is not None:
        # arithmetic of the matrix kernels for this run's networks (bf16x3 default | f32 | f16x2 | bf16, see csvae.h):
        # set on the handles, not process-wide
        for net in (model.var_latent_network, model.generator_network):
            net.packed().set_arith(str(cfg["mlp_arith"]))
    tester.set_model(model)
    tester.set_datamodule(datamodule)
    datamodule.set_model(model)
    tester.rng_seed = int(cfg["seed_everything"])
    tester.skip_unimplemented_diagnostics = bool(cfg.get("skip_unimplemented_diagnostics", False))
    sharded = world_size > 1
    prepare_outputs(tester, experiment_dir, rank, sharded)

    datamodule.setup("test")
    tester.on_test_start()
    tester.on_test_epoch_start()
    outputs = []
    with torch.no_grad():
        offset = datamodule.shard_offset
        for batch_idx, batch in enumerate(datamodule.test_dataloader()):
            tester.example_offset = offset
            tester.on_test_batch_start(batch, batch_idx, 0)
            out = tester.test_step(batch, batch_idx)
            tester.on_test_batch_end(out, batch, batch_idx, 0)
            outputs.append(out)
            offset += batch[0].shape[0]
    finish_run(tester, outputs, datamodule.num_test_total, datamodule.hparams.batch_size, rank, sharded,
               gather_device or device)
    return tester, outputs


def prepare_outputs(tester, experiment_dir, rank, sharded):
    """One ``version_N`` directory per run: rank 0 chooses it and tells the others before any rank
    creates a directory; only rank 0 writes.  Sharded ranks keep their artefacts in a collector."""
    version = _next_version(experiment_dir) if rank == 0 else None
    if sharded:
        version = distributed.broadcast_object(version)
        tester.collector = distributed.ArtifactCollector()
    tester.out_dir = (os.path.join(experiment_dir, "lightning_logs", f"version_{version}") if rank == 0 else None)
    tester.run_version = version


def finish_run(tester, outputs, num_test_total, batch_size, rank, sharded, device):
    """End of the epoch.  Sharded with a process group: the only exchange of the path -- stored
    imputations / final-resampling imputations / latents (one all-gather each, global example
    order) and the per-batch metric records -- then rank 0 writes the files of a one-process run."""
    if sharded and distributed.group_ready():
        merged = distributed.gather_shard_arrays(tester.collector.shard_arrays(), num_test_total, device)
        tester.metrics = distributed.gather_lists(dict(tester.metrics))
        if getattr(tester, "norm_log_weights", None) is not None:
            tester.norm_log_weights = distributed.gather_lists(list(tester.norm_log_weights))
        if rank == 0:
            distributed.write_batches(merged, batch_size, tester.out_dir, save=tester._write_npz)
    tester.test_epoch_end(outputs)


def _next_version(experiment_dir):
    existing = glob.glob(os.path.join(experiment_dir, "lightning_logs", "version_*"))
    nums = [int(p.rsplit("_", 1)[1]) for p in existing if p.rsplit("_", 1)[1].isdigit()]
    return max(nums) + 1 if nums else 0


def main(argv=None):
    ap = argparse.ArgumentParser(description="Conditional sampling of a VAE (LAIR / AC-MWG) on MI355X")
    ap.add_argument("--config", required=True)
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override a config entry, e.g. --set data.batch_size=4096")
    args = ap.parse_args(argv)
    cfg = load_config(args.config, args.set)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the sampling path runs in HIP kernels only (no CPU fallback)")
    # CSVAE_DIST_REHEARSAL=1: the multi-process flow on a box with one GPU -- every rank computes on GPU 0 and the
    # final gathers run over gloo on host copies (a check of the control flow, not a way to run faster)
    rehearsal = os.environ.get("CSVAE_DIST_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    try:
        run(cfg, rank=rank, world_size=world, gather_device=torch.device("cpu") if rehearsal else None)
    finally:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
