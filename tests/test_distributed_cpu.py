"""CPU, world_size 2 over gloo: the sample-sharding + frame all_gather of the multi-GPU path
(holo_diffusion_amd/generate.py).  On the GPU node the same code runs over RCCL."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from holo_diffusion_amd.generate import gather_frames, shard_indices


def test_shard_indices_partition():
    for n in (1, 5, 8, 9):
        for w in (1, 2, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = (3, 2, 4)
        local = {i: torch.full(shape, float(i) + 0.5) for i in shard_indices(n_items, rank, world)}
        out = gather_frames(local, n_items, shape, torch.device("cpu"))
        ok = all(torch.all(out[i] == i + 0.5).item() for i in range(n_items)) and out.shape == (n_items,) + shape
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gather_frames_gloo_world2():
    ctx = mp.get_context("spawn")
    for n_items in (5, 2, 1):
        q = ctx.Queue()
        port = 29500 + (os.getpid() + n_items) % 2000
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
        assert res == [(0, True), (1, True)]
