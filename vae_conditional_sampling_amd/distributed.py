"""Multi-GPU layout of the sampling path: one process per GPU, examples sharded in contiguous
blocks, no collective inside the sampler loop; the only exchange is one all-gather per stored
array (imputations, final-resampling imputations, latents, log-weights) after the loop (RCCL over
xGMI on GPUs; gloo in CPU tests).

The reference is single-process and writes one ``imputations_{b}.npz`` /
``irwg_imputations_after_final_resampling_{b}.npz`` per batch
(irwg/sampling/test_step_vae_sampling.py:766-791, 2039-2045).  A sharded run keeps each rank's
per-batch arrays in memory (``ArtifactCollector``), gathers them in global example order and lets
rank 0 cut them into the batches of the unsharded data loader again, so the files it writes have
the names, keys and -- the counter RNG being keyed by the global example index -- the contents
of a one-process run.
"""
import os
import re
from collections import defaultdict

import numpy as np
import torch
import torch.distributed as dist

from .data.missing_data_module import shard_range

# artefact file stem -> key -> axis of the example dimension (keys as written by the sampler classes)
EXAMPLE_AXIS = {
    "imputations": {"imputations": 1, "latents": 1, "true_X": 0, "masks": 0, "data_idx": 0},
    "irwg_imputations_after_final_resampling": {"imputations": 1, "latents": 0, "true_X": 0, "masks": 0,
                                                "data_idx": 0},
}


def _is_placeholder(key, a, ax):
    """An option that is off is stored as an empty list (e.g. ``latents`` without ``store_latents``): such an entry
    has no example axis and is passed through as it is.  (``data_idx`` is the one genuinely 1-d per-example array:
    empty only on a rank without examples.)"""
    return a.ndim <= ax or (a.ndim == 1 and a.size == 0 and key != "data_idx")


def group_ready():
    return dist.is_available() and dist.is_initialized()


def gather_sharded(local, n_total, group=None):
    """All-gather per-rank blocks ``local[n_local, ...]`` (split by ``shard_range``) into the
    full ``[n_total, ...]`` tensor in global example order, on every rank.  Blocks are padded to
    the largest block so a single ``all_gather_into_tensor`` suffices."""
    if not group_ready():
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    n_max = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.reshape((world, n_max) + tuple(local.shape[1:]))
    return torch.cat([out[r, :hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


class ArtifactCollector:
    """Keeps the arrays of the per-batch ``.npz`` files of one rank instead of writing them."""

    def __init__(self):
        self.files = defaultdict(list)   # stem -> [(batch_idx, {key: array})]

    def add(self, path, arrays):
        m = re.match(r"(.*)_(\d+)\.npz$", os.path.basename(path))
        if m is None or m.group(1) not in EXAMPLE_AXIS:
            raise ValueError(f"no sharding rule for artefact {path!r}")
        self.files[m.group(1)].append((int(m.group(2)), {k: np.asarray(v) for k, v in arrays.items()}))

    def shard_arrays(self):
        """{stem: {key: array}} with this rank's batches concatenated along the example axis."""
        out = {}
        for stem, items in self.files.items():
            items = sorted(items, key=lambda it: it[0])
            out[stem] = _concat([a for _, a in items], stem)
        return out


def _concat(dicts, stem):
    merged = {}
    for key in dicts[0]:
        ax = EXAMPLE_AXIS[stem][key]
        parts = [d[key] for d in dicts]
        merged[key] = parts[0] if _is_placeholder(key, parts[0], ax) else np.concatenate(parts, axis=ax)
    return merged


def merge_shards(shards):
    """Concatenate the ``shard_arrays()`` of all ranks (in rank order) along the example axis."""
    return {stem: _concat([s[stem] for s in shards], stem) for stem in shards[0]}


def _shard_schema(shard):
    """[(stem, key, example axis, trailing shape, dtype, placeholder array or None)] of a rank's arrays."""
    rows = []
    for stem in sorted(shard):
        for key in sorted(shard[stem]):
            a = shard[stem][key]
            ax = EXAMPLE_AXIS[stem][key]
            if _is_placeholder(key, a, ax):
                rows.append((stem, key, ax, None, str(a.dtype), a))
            else:
                rows.append((stem, key, ax, tuple(np.moveaxis(a, ax, 0).shape[1:]), str(a.dtype), None))
    return rows


def gather_shard_arrays(shard, n_total, device, group=None):
    """The collective form of ``merge_shards``: every array of ``shard`` goes through one
    ``gather_sharded``.  Returns the merged structure (on every rank).

    The ranks first agree on the list of arrays (names, trailing shapes, dtypes): a rank whose shard is empty
    (fewer examples than ranks) has collected nothing, and takes part in every collective with zero-length blocks
    of the shapes the other ranks report."""
    schema = _shard_schema(shard)
    if group_ready():
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, schema, group=group)
        schema = max(parts, key=len)   # a rank with examples has the full list; ranks with examples agree
    out = {}
    for stem, key, ax, trailing, dtype, placeholder in schema:
        out.setdefault(stem, {})
        if trailing is None:
            out[stem][key] = placeholder
            continue
        a = shard.get(stem, {}).get(key)
        if a is None or _is_placeholder(key, a, ax):
            t = torch.from_numpy(np.zeros((0,) + tuple(trailing), dtype=np.dtype(dtype)))
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(a, ax, 0)))
        as_bool = t.dtype == torch.bool
        t = t.to(torch.uint8) if as_bool else t
        g = gather_sharded(t.to(device).contiguous(), n_total, group).cpu()
        g = g.bool() if as_bool else g
        out[stem][key] = np.moveaxis(g.numpy(), 0, ax)
    return out


def write_batches(merged, batch_size, out_dir, save=np.savez_compressed):
    """Cut the merged arrays into the batches of the unsharded data loader and write
    ``{stem}_{b}.npz`` with the reference's keys."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for stem, arrays in merged.items():
        n = arrays["true_X"].shape[0]
        for b, lo in enumerate(range(0, n, int(batch_size))):
            part = {}
            for key, a in arrays.items():
                ax = EXAMPLE_AXIS[stem][key]
                part[key] = (a if _is_placeholder(key, a, ax)
                             else np.take(a, np.arange(lo, min(lo + int(batch_size), n)), axis=ax))
            path = os.path.join(out_dir, f"{stem}_{b}.npz")
            save(path, **part)
            paths.append(path)
    return paths


def gather_lists(obj, group=None):
    """Per-batch host-side records (metric series, batch sizes) of all ranks, concatenated in rank
    order: ``obj`` is a dict of lists or a list."""
    if not group_ready():
        return obj
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, obj, group=group)
    if isinstance(obj, dict):
        keys = []
        for p in parts:
            keys += [k for k in p if k not in keys]
        return {k: [v for p in parts for v in p.get(k, [])] for k in keys}
    return [v for p in parts for v in p]


def broadcast_object(obj, src=0, group=None):
    if not group_ready():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]
