"""N>1 path on CPU: image sharding + the final gather over torch.distributed (gloo, world_size 2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from f3dgaus_amd import dist as fdist


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 65, 511, 512):
        for world in (1, 2, 3, 8):
            spans = [fdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == fdist.shard_sizes(n, world)


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = fdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    s, e = fdist.shard_range(n_items, rank, world)
    # every rank "renders" its own images: frame i is filled with the value i (shape [n_local, 3, 4, 4])
    frames = torch.stack([torch.full((3, 4, 4), float(i)) for i in range(s, e)]) if e > s else torch.zeros(0, 3, 4, 4)
    out = fdist.gather_frames(frames, n_items_total=n_items, dst=0)
    if rank == 0:
        q.put(out.clone())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 8])
def test_gather_frames_gloo_world2(n_items):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out.shape == (n_items, 3, 4, 4)
    assert torch.equal(out[:, 0, 0, 0], torch.arange(n_items, dtype=torch.float32))
