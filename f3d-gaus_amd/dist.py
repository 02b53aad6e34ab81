"""Multi-GPU layer of the hot path: the batch of input images shards embarrassingly -- one
process per GPU, rank r owns a contiguous slice of the images and runs the whole per-image pipeline on it with NO
data-path collective -- and the only exchange is the final gather of the rendered frames to rank 0 (RCCL over xGMI
when the backend is "nccl"; "gloo" on CPU for the tests). The reference has no distributed code at all
(SURVEY section 0.5); this is the build's one addition.

xGMI note: a gather-to-root on a fully connected 8-GPU node lands on 7 distinct links of the root, so its time is
~ bytes_per_rank / 153 GB/s; frames are sent as ONE contiguous tensor per rank (fewer, larger messages).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (as torchrun sets them).
    Returns (rank, local_rank, world_size). No-op for world_size 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous [start, end) slice of ``n_items`` images owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_items, world):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def gather_frames(frames, n_items_total=None, dst=0, group=None):
    """Gather per-rank frame tensors [n_local, ...] to ``dst`` in image order. Returns the concatenated
    [n_total, ...] tensor on ``dst`` and None elsewhere. Ragged shards are padded to the largest shard so that
    every rank sends one contiguous message of equal size (a single collective)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return frames
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if n_items_total is None:
        n = torch.tensor([frames.shape[0]], dtype=torch.int64, device=frames.device)
        dist.all_reduce(n, group=group)
        n_items_total = int(n.item())
    sizes = shard_sizes(n_items_total, world)
    assert frames.shape[0] == sizes[rank], (frames.shape[0], sizes[rank], "shards must follow shard_range()")
    longest = max(sizes)
    send = frames.contiguous()
    if send.shape[0] < longest:
        pad = torch.zeros((longest - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][:sizes[r]] for r in range(world)], 0)
