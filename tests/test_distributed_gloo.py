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
