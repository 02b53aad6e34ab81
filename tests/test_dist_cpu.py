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



def _run_world2(target, extra_args, attempts=3):
    """Two spawned ranks on a free local port; rank 0's result from the queue. An ephemeral port can be taken between its
    probe and the rendezvous (or a loaded host can miss a timeout): two retries on fresh ports, a moment apart, before failing."""
    import queue as _queue
    import time as _time
    last = None
    for attempt in range(attempts):
        if attempt:
            _time.sleep(1.0 + attempt)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=(r, 2, port) + tuple(extra_args) + (q,)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            out = q.get(timeout=120)
        except _queue.Empty as e:
            out, last = None, e
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.terminate()
        if out is not None and all(p.exitcode == 0 for p in procs):
            return out
        last = last or RuntimeError("exit codes %s" % [p.exitcode for p in procs])
    raise AssertionError("world_size-2 gloo run failed %d times: %r" % (attempts, last))


@pytest.mark.parametrize("n_items", [5, 8])
def test_gather_frames_gloo_world2(n_items):
    out = _run_world2(_worker, (n_items,))
    assert out.shape == (n_items, 3, 4, 4)
    assert torch.equal(out[:, 0, 0, 0], torch.arange(n_items, dtype=torch.float32))


def _c4_worker(rank, world, port, n_images, q):
    """The N > 1 step of bench.py --workload c4 with the renderer replaced by a formula: shard the batch of images, 'render' 8
    frames per image, hand them to bench.Gatherer (asynchronous gather, one in flight, barrier inside the timed region)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    fdist.init_from_env(backend="gloo")
    s, e = fdist.shard_range(n_images, rank, world)
    assert e - s == n_images // world                       # bench.py gives every rank the same number of images (weak scaling)
    V = 8
    gat = bench.Gatherer(dist, world, rank, ((e - s) * V, 4, 4, 3), torch.device("cpu"))
    for step in range(3):                                   # three steps: the gathers overlap the next step's work
        frames = torch.stack([torch.full((4, 4, 3), (img * V + v + step) % 251, dtype=torch.uint8) for img in range(s, e) for v in range(V)])
        gat.submit(frames)
    gat.barrier()
    assert not gat.pending
    if rank == 0:
        q.put(torch.cat(gat.buf).clone())
    dist.barrier()
    dist.destroy_process_group()


def test_c4_step_gather_gloo_world2():
    n_images = 6
    out = _run_world2(_c4_worker, (n_images,))
    assert out.shape == (n_images * 8, 4, 4, 3)
    assert torch.equal(out[:, 0, 0, 0].long(), (torch.arange(n_images * 8) + 2) % 251)      # the last step's frames, in image order
