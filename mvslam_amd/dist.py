"""Pair-level sharding over the GPUs of one node.

Image pairs are independent units (front-end/image-pair.hpp:56-57: two immutable frames), so the only
exchange step of the path is one all-gather of the fixed-size pose records; point clouds stay on their GPU.
`torch.distributed` is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np


def shard_range(n_total, rank, world_size):
    """Contiguous block of pairs owned by `rank` (config 4: 4096 pairs = 8 x 512)."""
    base, rem = divmod(n_total, world_size)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def gather_records(local, world_size):
    """All-gather equal-size uint8 record blocks: local [n_local * record_bytes] -> [world, n_local * record_bytes].

    `local` is a torch uint8 tensor on the device of the active backend (cuda for nccl/RCCL, cpu for gloo).
    The payload is ~57 kB per rank at 512 pairs: latency-bound, so a single flat all-gather is used.
    """
    import torch
    import torch.distributed as dist

    if world_size == 1 or not dist.is_initialized():
        return local.reshape(1, -1)
    out = torch.empty((world_size, local.numel()), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.reshape(-1), local.reshape(-1))
    return out


def records_to_numpy(gathered, record_dtype):
    """[world, n_local * record_bytes] uint8 tensor -> structured numpy array [world * n_local]."""
    a = gathered.detach().cpu().numpy().reshape(-1)
    return np.frombuffer(a.tobytes(), dtype=record_dtype)
