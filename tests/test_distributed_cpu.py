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


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher must start 2 ranks itself and report the LIVE world size
    (--dry-run: rendezvous + reductions only, gloo on CPU here, RCCL on the GPU node)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["requested_gpus"] == 2 and line["max_over_ranks"] == 2.0
    assert line["exchanges_ok"] is True  # frame all_gather + bucketed gradient all-reduce on small tensors
    assert len(set(line["pids"])) == 2
    # a single-rank run reports 1
    res = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--dry-run"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 1


# ---- data-parallel gradient exchange of the training branch (holo_diffusion_amd/ddp.py) ---------------
def test_plan_buckets_is_order_preserving_and_size_bounded():
    from holo_diffusion_amd.ddp import plan_buckets
    g = {f"p{i}": torch.zeros(n) for i, n in enumerate((10, 300, 5, 5, 1000, 1))}
    b = plan_buckets(g, bucket_bytes=4 * 320)
    assert [k for names in b for k in names] == list(g)  # dict order, nothing dropped
    assert b == [["p0", "p1", "p2", "p3"], ["p4"], ["p5"]]  # (40 + 1200 + 20 + 20 bytes fill 1280) a tensor above the bucket size gets its own
    assert plan_buckets(g, bucket_bytes=1 << 30) == [list(g)]


def _ddp_worker(rank, world, port, bucket_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from holo_diffusion_amd.ddp import allreduce_gradients, allreduce_training_gradients
        gen = torch.Generator().manual_seed(7)
        shapes = {"a.weight": (64, 32, 27), "a.bias": (64,), "b.weight": (3, 283), "c": (1,), "d.weight": (257, 256)}
        base = {k: torch.randn(s, generator=gen) for k, s in shapes.items()}
        mine = {k: v * (rank + 1) for k, v in base.items()}  # rank r holds (r + 1) * base: the mean is 1.5 * base
        allreduce_gradients(mine, bucket_bytes=bucket_bytes)
        ok = all(torch.allclose(mine[k], 1.5 * base[k], rtol=1e-6, atol=1e-6) for k in base)
        summed = {k: v * (rank + 1) for k, v in base.items()}
        allreduce_gradients(summed, bucket_bytes=bucket_bytes, average=False)
        ok = ok and all(torch.allclose(summed[k], 3.0 * base[k], rtol=1e-6, atol=1e-6) for k in base)
        out = {"unet": {k: v * (rank + 1) for k, v in base.items()}, "render_mlp": {"w": torch.full((5,), float(rank))},
               "voxel_grid": torch.full((2,), float(rank))}
        allreduce_training_gradients(out, bucket_bytes=bucket_bytes)
        ok = ok and torch.allclose(out["render_mlp"]["w"], torch.full((5,), 0.5)) and \
            torch.allclose(out["unet"]["c"], 1.5 * base["c"]) and torch.all(out["voxel_grid"] == float(rank)).item()
        # the encoder side (pool_views_backward): mapper + learnt aggregator are parameters, the feature maps are not
        enc = {"image_features": {"res": torch.full((3,), float(rank))},
               "pooled_feature_mapper": {"weight": torch.full((4, 6), float(2 * rank)), "bias": None},
               "feature_aggregator": {"_last.weight": base["b.weight"] * (rank + 1)}}
        out2 = {"unet": {"c": base["c"] * (rank + 1)}, "render_mlp": {}, "voxel_features": torch.full((2,), float(rank))}
        allreduce_training_gradients(out2, bucket_bytes=bucket_bytes, encoder=enc)
        ok = ok and torch.allclose(enc["pooled_feature_mapper"]["weight"], torch.full((4, 6), 1.0)) and \
            enc["pooled_feature_mapper"]["bias"] is None and \
            torch.allclose(enc["feature_aggregator"]["_last.weight"], 1.5 * base["b.weight"], rtol=1e-6, atol=1e-6) and \
            torch.all(enc["image_features"]["res"] == float(rank)).item() and torch.allclose(out2["unet"]["c"], 1.5 * base["c"]) and \
            torch.all(out2["voxel_features"] == float(rank)).item()
        # one parameter name in BOTH dicts (advisor, round 5): distinct tensors are both averaged, a shared one exactly once
        shared = torch.full((3,), float(4 * rank))
        out3 = {"pooled_feature_mapper": {"weight": torch.full((2,), float(rank)), "bias": shared}}
        enc3 = {"pooled_feature_mapper": {"weight": torch.full((2,), float(10 * rank)), "bias": shared}}
        allreduce_training_gradients(out3, bucket_bytes=bucket_bytes, encoder=enc3)
        ok = ok and torch.allclose(out3["pooled_feature_mapper"]["weight"], torch.full((2,), 0.5)) and \
            torch.allclose(enc3["pooled_feature_mapper"]["weight"], torch.full((2,), 5.0)) and \
            torch.allclose(shared, torch.full((3,), 2.0))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    """Two ranks with different gradients: every tensor comes back as the mean (or the sum), whatever the bucket size -
    one bucket for everything, buckets smaller than the largest tensor, one tensor per bucket."""
    ctx = mp.get_context("spawn")
    for j, bucket_bytes in enumerate((1 << 30, 4 * 9000, 16)):
        q = ctx.Queue()
        port = 31500 + (os.getpid() + j) % 2000
        procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
        assert res == [(0, True), (1, True)], (bucket_bytes, res)


def test_gradient_allreduce_single_process_is_identity():
    from holo_diffusion_amd.ddp import allreduce_gradients
    g = {"w": torch.arange(6.0).reshape(2, 3)}
    assert allreduce_gradients(g) is g and torch.equal(g["w"], torch.arange(6.0).reshape(2, 3))


# ---- camera sharding of one grid's turntable (optional row; generate.render_views_sharded) --------------
class _TurntableStub(torch.nn.Module):
    """A deterministic stand-in for HoloDiffusionModel.render_views (CPU): every frame is a function of the grid and of ITS
    camera only - the property the real renderer has (test_teddybear_30_view_turntable_in_one_call)."""
    render_image_height, render_image_width = 6, 5

    def __init__(self, n_steps=3, normals=False):
        super().__init__()
        self.n_steps, self.normals = n_steps, normals

    def render_views(self, vf, cams):
        n, H, W = len(cams), self.render_image_height, self.render_image_width
        base = (vf.double().sum() * 1e-3 + cams.T.double().sum(dim=1) + cams.R.double().reshape(n, -1)[:, 1]).float()
        img = base[:, None, None, None] + torch.arange(3 * H * W, dtype=torch.float32).reshape(1, 3, H, W) / 7.0
        out = {"images_render": img, "depths_render": img[:, :1] * 2.0, "masks_render": torch.sigmoid(img[:, 1:2])}
        if self.normals:
            out["normals_render"] = -img
        return out

    def sample_random_voxel_features_progressive(self, **kw):
        g = torch.Generator().manual_seed(5)
        x = torch.randn(1, 4, 3, 3, 3, generator=g)
        for k in range(self.n_steps):
            x = x * 0.5 + k
            yield x.clone()


def _turntable_worker(rank, world, port, n_views, normals, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from holo_diffusion_amd.cameras import get_simple_360_camera_trajectory
        from holo_diffusion_amd.generate import render_progressive_turntable_sharded, render_views_sharded
        import math
        dev = torch.device("cpu")
        model = _TurntableStub(normals=normals)
        cams = get_simple_360_camera_trajectory(2 * math.pi, n_views, -30.0 * (2 * math.pi / 360), 10, (0.0, -1.0, 0.0), 3.2)  # (the driver's defaults)
        vf = torch.arange(108, dtype=torch.float32).reshape(1, 4, 3, 3, 3) if rank == 0 else None
        got = render_views_sharded(model, vf, cams, src_rank=0, device=dev)
        want = model.render_views(torch.arange(108, dtype=torch.float32).reshape(1, 4, 3, 3, 3), cams)
        ok = set(got) == set(want) | {"voxel_features"} and all(torch.equal(got[k], want[k]) for k in want)
        ok = ok and torch.equal(got["voxel_features"], torch.arange(108, dtype=torch.float32).reshape(1, 4, 3, 3, 3))
        # the progressive driver: chain on rank 0, every rank receives every step's frames and the broadcast grid
        steps = list(render_progressive_turntable_sharded(model, n_views=n_views, steps_per_render=1, device=dev))
        ref_grids = list(model.sample_random_voxel_features_progressive())
        ok = ok and len(steps) == len(ref_grids)
        for s, g in zip(steps, ref_grids):
            w = model.render_views(g, cams)
            ok = ok and all(torch.equal(s[k], w[k]) for k in w)
            ok = ok and s["voxel_features"] is not None and torch.equal(s["voxel_features"], g)  # the broadcast grid, on EVERY rank
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_camera_sharded_turntable_gloo_world2_equals_single_rank():
    """30-view turntable of one grid over 2 ranks: broadcast of the grid, cameras k mod N, one all_gather per output -
    bit for bit the single-rank call, also when a rank has no camera (1 view) and with rendered normals."""
    ctx = mp.get_context("spawn")
    for j, (n_views, normals) in enumerate(((30, False), (5, True), (1, False))):
        q = ctx.Queue()
        port = 33100 + (os.getpid() + 7 * j) % 2000
        procs = [ctx.Process(target=_turntable_worker, args=(r, 2, port, n_views, normals, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=180) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
        assert res == [(0, True), (1, True)], (n_views, normals, res)
