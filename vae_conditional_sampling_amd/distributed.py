"""Multi-GPU layout of the sampling path: one process per GPU, examples sharded in contiguous
blocks, no collective inside the sampler loop; the only exchange is one all-gather of the
final imputations (and log-weights) after the loop (RCCL over xGMI on GPUs; gloo in CPU tests).
"""
import torch
import torch.distributed as dist

from .data.missing_data_module import shard_range


def gather_sharded(local, n_total, group=None):
    """All-gather per-rank blocks ``local[n_local, ...]`` (split by ``shard_range``) into the
    full ``[n_total, ...]`` tensor in global example order, on every rank.  Blocks are padded to
    the largest block so a single ``all_gather_into_tensor`` suffices."""
    if not (dist.is_available() and dist.is_initialized()):
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
