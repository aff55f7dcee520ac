#!/usr/bin/env python3
"""bench.py — denoise-steps/s + rendered-rays/s of the HoloDiffusion hot path on MI355X.

Workload (BASELINE.json configs[1]): apple.yaml single-sample DDPM, 64^3 x 32 voxel grid, UNet
model_channels 64, mult (1,1,2,4,8), attention at ds 4/8, then 400x400 frames (64 coarse + 128 fine samples
per ray), fp32, synthetic weights/noise/cameras (no network: random-init weights of that architecture).

A "step" is one DDPM ancestral step: UNet forward + clamp/posterior + fresh Gaussian noise, exactly
gaussian_diffusion.py:459-508 of the reference.  After W warm-up steps EXACTLY K steps are timed between
barrier + synchronize on both sides; with N>1 every rank (one process per GPU, launched by torch.distributed.run)
runs its own independent chain (weak scaling, no data-path collective) and the MAX time over ranks is used.
The render leg (frames of the same grid) is timed the same way and reported in the same JSON line.

  roofline      the dominant kernel (the 64^3-level LDS voxel-halo conv3d), every launch of one UNet forward timed
                with hipEvents on the launch stream (holo_unet_time_ops): algorithmic FLOPs / average launch
                time vs the 157.3 TFLOP/s fp32 MFMA peak; `traffic` = HBM bytes per launch from the committed PMC
                pass (profiles/pmc_traffic.json); all conv launches together under `all_conv_launches`
  cpu_baseline  the CPU oracle (torch-CPU restatement proven bit-equal to the reference, oracle/) on a bounded
                sample: a few UNet forwards + posterior at 64^3x32 and one small frame, all host cores
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
import warnings

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0
UNET_FLOPS_PER_STEP = 1180.8e9  # BASELINE.md §2 (reference module trace, 2*MACs) at 64^3x32

NORTH = dict(resol=64, feature_size=32, model_channels=64, channel_mult=(1, 1, 2, 4, 8), attention_resolutions=(4, 8))
SMALL = dict(resol=32, feature_size=16, model_channels=64, channel_mult=(1, 1, 2, 4, 8), attention_resolutions=(4, 8))
DONUT = dict(resol=128, feature_size=32, model_channels=64, channel_mult=(1, 1, 2, 4, 8), attention_resolutions=(4, 8))
FLOPS_PER_STEP = {64: 1180.8e9, 32: 139.1e9, 128: 11926.9e9}  # BASELINE.md / SURVEY.md 8d (reference module trace)


def build_model(w, H, W, device, n_fine=64, compute_dtype="f32"):
    import holo_diffusion_amd as hda
    from holo_diffusion_amd.structure import unet_param_shapes
    from holo_diffusion_amd.weights import synth_state_dict
    model = hda.HoloDiffusionModel(
        resol=w["resol"], feature_size=w["feature_size"], render_image_width=W, render_image_height=H,
        net_3d_SimpleUnet3D_args=dict(model_channels=w["model_channels"], channel_mult=w["channel_mult"],
                                      attention_resolutions=w["attention_resolutions"], compute_dtype=compute_dtype),
        diffusion_args=dict(num_steps=1000),
        renderer_HoloMultiPassEmissionAbsorptionRenderer_args=dict(n_pts_per_ray_fine_evaluation=n_fine))
    usd = synth_state_dict(unet_param_shapes(w["resol"], w["feature_size"], w["feature_size"], w["model_channels"], 2,
                                             w["channel_mult"], w["attention_resolutions"]), 1234)
    mlp = model._implicit_functions[0]._fn.render_mlp
    msd = synth_state_dict({k: tuple(v.shape) for k, v in mlp.state_dict().items()}, 4321)
    full = {"net_3d._net." + k: v for k, v in usd.items()}
    for i in range(model.num_passes):
        full.update({f"_implicit_functions.{i}._fn.render_mlp." + k: v for k, v in msd.items()})
    model.load_state_dict(full)
    if compute_dtype == "f32_bf16x3":
        model.renderer.compute_dtype = "f32_bf16x3"
    return model.to(device), usd, msd


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x: float, world: int, device) -> float:
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(w, usd, msd, budget_s=30.0):
    """Bounded CPU sample of the same workload through the oracle (kind 'port'): DDPM steps at the
    workload's grid size with torch on ALL host hardware threads, one 128x128 frame of the same grid."""
    from oracle import diffusion_oracle as do
    from oracle import render_oracle as ro
    from oracle import unet_oracle as uo
    from oracle.common import np_noise
    hw_threads = os.cpu_count() or 1
    try:  # physical cores (threads / SMT siblings), reported next to the thread count actually used
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        smt = len([x for part in sib.split(",") for x in (range(int(part.split("-")[0]), int(part.split("-")[-1]) + 1))])
        phys = max(1, hw_threads // max(smt, 1))
    except (OSError, ValueError):
        phys = hw_threads
    cfg = uo.UNetCfg(image_size=w["resol"], in_channels=w["feature_size"], out_channels=w["feature_size"],
                     model_channels=w["model_channels"], num_res_blocks=2, channel_mult=w["channel_mult"],
                     attention_resolutions=w["attention_resolutions"], num_heads=2)
    orc = do.DiffusionOracle(1000)
    shape = (1, w["feature_size"]) + (w["resol"],) * 3
    eps = torch.from_numpy(np_noise(2, shape))
    model = lambda a, b: uo.unet_forward(usd, cfg, a, b)  # noqa: E731

    def steps(threads, budget):
        torch.set_num_threads(threads)
        x = torch.from_numpy(np_noise(1, shape))
        t0 = time.time()
        orc.p_sample(model, x, torch.tensor([999]), eps)  # warm-up
        warm = time.time() - t0
        n = max(1, min(5, int(budget / max(warm, 1e-3)) - 1))
        t0 = time.time()
        for i in range(n):
            x = orc.p_sample(model, x, torch.tensor([998 - i]), eps)["sample"]
        return n / (time.time() - t0), n

    # all hardware threads (the 8d definition) vs <= 32 threads: torch's CPU conv3d / GEMM stop scaling far below the thread
    # count of the GPU box (round 4: ONE step at 256 threads took ~80 s of a 177 s driver run, 40x slower than at 32).  A probe
    # at 16^3 (1/64 of the work, same net) decides whether the all-threads sample is worth taking at full size: it is taken
    # only when the probe says all threads are not clearly slower; otherwise the probe's ratio is recorded and the <= 32
    # thread sample is the baseline.  The FASTER of the samples taken is reported.
    by_threads, probe, n2 = {}, None, 0
    n = 0
    if hw_threads > 32:
        pcfg = uo.UNetCfg(image_size=16, in_channels=w["feature_size"], out_channels=w["feature_size"],
                          model_channels=w["model_channels"], num_res_blocks=2, channel_mult=w["channel_mult"],
                          attention_resolutions=w["attention_resolutions"], num_heads=2)
        px = torch.from_numpy(np_noise(3, (1, w["feature_size"], 16, 16, 16)))
        probe = {}
        for th in (32, hw_threads):
            torch.set_num_threads(th)
            uo.unet_forward(usd, pcfg, px, torch.tensor([500]))
            t0 = time.time()
            uo.unet_forward(usd, pcfg, px, torch.tensor([499]))
            probe[str(th)] = 1.0 / max(time.time() - t0, 1e-6)
        sps2, n2 = steps(32, budget_s * 0.5)
        by_threads[32] = sps2
    if probe is None or probe[str(hw_threads)] >= 0.5 * probe["32"]:
        sps, n = steps(hw_threads, budget_s * 0.35)
        by_threads[hw_threads] = sps
    cores = max(by_threads, key=by_threads.get)
    steps_per_s = by_threads[cores]
    torch.set_num_threads(cores)
    Hs = Ws = 128  # 8d: one 128x128 frame (16 384 rays, 64 + n_fine new samples per ray)
    rcfg = ro.RenderCfg(resol=w["resol"], feature_size=w["feature_size"], image_height=Hs, image_width=Ws)
    grid = torch.tanh(torch.from_numpy(np_noise(7, shape)))
    cams = ro.simple_360_cameras(4)
    t0 = time.time()
    ro.render(grid, msd, {k: v[1:2] for k, v in cams.items()}, rcfg)
    rays_per_s = Hs * Ws / (time.time() - t0)
    return {"value": steps_per_s, "unit": "denoise-steps/s", "cores": cores, "kind": "port",
            "rays_per_sec": rays_per_s, "host_hw_threads": hw_threads, "host_physical_cores": phys,
            "steps_per_s_by_threads": {str(k): v for k, v in by_threads.items()},
            "thread_scaling_probe_16cubed_forwards_per_s": probe,
            "sample": f"oracle DDPM steps (UNet fwd + posterior) at {w['resol']}^3x{w['feature_size']} after 1 warm-up: "
                      + (f"{n} timed at {hw_threads} threads" if n else
                         f"the all-threads ({hw_threads}) sample was skipped - a 16^3 probe ran {probe['32'] / probe[str(hw_threads)]:.1f}x "
                         f"slower there than at 32 threads")
                      + (f", {n2} timed at 32 threads" if n2 else "")
                      + f" (reported: {cores} threads, the faster); one {Hs}x{Ws} frame (64 coarse + 128 fine samples"
                      f"/ray) of a {w['resol']}^3 grid at {cores} threads; torch CPU"}


def side_donut128(usd, device, warm=3, timed=5):
    """BASELINE configs[4] (donut.yaml size): 128^3x32 grid, bf16 storage mode (bf16 activations in HBM, bf16 products /
    fp32 accumulate, fp32 GroupNorm statistics and network input/output), batch-1 DDPM steps on one MI355X."""
    import holo_diffusion_amd as hda
    w = DONUT
    net = hda.SimpleUnet3D(image_size=w["resol"], in_channels=w["feature_size"], out_channels=w["feature_size"],
                           model_channels=w["model_channels"], channel_mult=w["channel_mult"],
                           attention_resolutions=w["attention_resolutions"], compute_dtype="bf16")
    net.load_state_dict({"_net." + k: v for k, v in usd.items()})
    net = net.to(device)
    diff = hda.ImplicitronGaussianDiffusion(num_steps=1000, device_noise_seed=42, device_noise_stream=7)
    shape = (1, w["feature_size"]) + (w["resol"],) * 3
    ts = torch.arange(999, 999 - (warm + timed), -1, device=device, dtype=torch.int64)[:, None].contiguous()
    with torch.no_grad():
        # the sampler's perf chain, as the reported line times it at 64^3: channels-last grid, in-kernel Philox noise
        x = torch.randn(*shape, device=device).permute(0, 2, 3, 4, 1).contiguous()
        for k in range(warm + timed):
            if k == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            out = net.forward_channels_last(x, ts[k])
            x = diff._step_device_noise(x, ts[k], out, 999 - k, True, want_pred=False, channels_last=True)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(x).all()
        # the reference's draw (torch.randn_like) on NCDHW tensors: two layout passes and a randn launch per step more
        x = torch.randn(*shape, device=device)
        for k in range(warm + timed):
            if k == warm:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            out = net(x, ts[k])
            x, _ = diff._step(x, ts[k], out, torch.randn_like(x), True)
        torch.cuda.synchronize()
        dt_ref = time.perf_counter() - t1
    assert torch.isfinite(x).all()
    sps = timed / dt
    ws = net.workspace_bytes(1, device)
    del net
    torch.cuda.empty_cache()
    return {"denoise_steps_per_s": sps, "ms_per_step": 1e3 * dt / timed, "steps": timed, "warmup": warm,
            "unet_tflops": FLOPS_PER_STEP[128] * sps / 1e12, "frac_of_bf16_peak": FLOPS_PER_STEP[128] * sps / 1e12 / PEAK_BF16_MFMA_TFLOPS,
            "unet_workspace_bytes": ws, "dtype": "bf16 storage, bf16 products / f32 accumulate",
            "steps_per_s_ncdhw_torch_noise": timed / dt_ref,
            "workload": "donut.yaml size: 128^3x32 grid, batch-1 DDPM steps (UNet forward + posterior + noise); timed: the sampler's "
                        "perf chain (channels-last grid, in-kernel Philox noise) like the reported line; steps_per_s_ncdhw_torch_noise: "
                        "NCDHW tensors + torch.randn_like (rounds 1-5 reported this form)"}


def side_batched_chains(net, diff, w, device, batches=(2, 4), warm=5, timed=20):
    """B independent ancestral chains advanced together (one UNet call on a batch of B grids per step): the tail of tiny
    launches (1x1 convs, 4^3 / 8^3 levels, GroupNorm finalisation) is paid once per call rather than once per grid."""
    res = {}
    for B in batches:
        x = torch.randn(B, w["feature_size"], *(w["resol"],) * 3, device=device)
        ts = torch.arange(999, 999 - (warm + timed), -1, device=device, dtype=torch.int64)[:, None].repeat(1, B).contiguous()
        with torch.no_grad():
            for k in range(warm + timed):
                if k == warm:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                out = net(x, ts[k])
                x, _ = diff._step(x, ts[k], out, torch.randn_like(x), True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert torch.isfinite(x).all()
        res[f"batch{B}"] = {"grid_steps_per_s": B * timed / dt, "ms_per_call": 1e3 * dt / timed, "calls": timed, "warmup": warm,
                            "unet_workspace_bytes": net.workspace_bytes(B, device)}
        del x
    res["workload"] = ("north-star net, B independent chains per GPU advanced by one batched UNet call per step "
                       "(generate_samples draws its samples one chain at a time; this is the throughput form)")
    torch.cuda.empty_cache()
    return res


def side_teddybear_turntable(model, device, n_views=30, warm=2, timed=6):
    """BASELINE configs[3] (teddybear.yaml): a 30-camera turntable @400x400 after EVERY denoise step
    (progressive_sampling_steps_per_render = 1, flyaround.py:240-253).  Per step, all inside the timed region: one DDPM step of
    the 64^3 x 32 chain, the clip, the refinement tanh(net_3d(vf, t = 0)) (holo_diffusion_model.py:420-426 - a second UNet
    forward, NOT cached: the grid is new every step) and all 30 cameras in ONE holo_render call - the product driver
    generate.render_progressive_turntable."""
    from holo_diffusion_amd.generate import render_progressive_turntable
    H, W = model.render_image_height, model.render_image_width
    gen = render_progressive_turntable(model, n_views=n_views, steps_per_render=1, device=device)
    with torch.no_grad():
        for _ in range(warm):
            out = next(gen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(timed):
            out = next(gen)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gen.close()
        # the pieces on their own (same grid, same cameras): refinement + render of one step
        vf = out["voxel_features"]
        cams = hda_cams(n_views, device)
        model.invalidate_refined_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.invalidate_refined_cache()
            model.render_views(vf, cams)
        torch.cuda.synchronize()
        dt_rr = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            model.render_views(vf, cams)  # (cached refinement: the fixed-grid figure of the main render leg)
        torch.cuda.synchronize()
        dt_r = (time.perf_counter() - t0) / 3
    assert torch.isfinite(out["images_render"]).all() and out["images_render"].shape[0] == n_views
    rays = n_views * H * W
    return {"steps_per_s": timed / dt, "ms_per_step": 1e3 * dt / timed, "views_per_step": n_views,
            "rays_per_sec_whole_step": rays * timed / dt,
            "rays_per_sec_refine_plus_render": rays / dt_rr, "ms_refine_plus_render": 1e3 * dt_rr,
            "rays_per_sec_render_only_fixed_grid": rays / dt_r, "ms_render_only": 1e3 * dt_r,
            "what": f"per denoise step: DDPM step + clip + tanh(net_3d(vf, 0)) + {n_views} frames @{H}x{W} in one holo_render call "
                    "(generate.render_progressive_turntable); rays_per_sec_whole_step counts the step's rays over the WHOLE step, "
                    "rays_per_sec_refine_plus_render leaves the DDPM step out, rays_per_sec_render_only_fixed_grid is the main "
                    "render leg's definition (refinement cached)"}


def hda_cams(n, device):
    import holo_diffusion_amd as hda
    return hda.get_simple_360_camera_trajectory(2 * math.pi, n, -30.0 * (2 * math.pi / 360), 10, (0.0, -1.0, 0.0), 3.2).to(device)


def side_training_step(net, w, device, warm=1, timed=3):
    """forward + backward of the denoiser at the north-star size through holo_unet_backward (the taped
    forward re-run, dgrad on the forward's convolution kernels with transposed weights, row-staged wgrad, GroupNorm /
    attention backward; every parameter gradient + the input gradient).  The optimiser step is the caller's."""
    x = torch.randn(1, w["feature_size"], *(w["resol"],) * 3, device=device)
    g = torch.randn_like(x)
    t = torch.tensor([500], device=device)
    names = ["out.2.weight"]  # (all gradients are computed; one is fetched)
    for _ in range(warm):
        net.backward(x, t, g, params=names)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(timed):
        net.backward(x, t, g, params=names)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / timed
    res = {"forward_backward_ms": 1e3 * dt, "calls": timed, "warmup": warm,
           "tflops_algorithmic": 3 * FLOPS_PER_STEP[w["resol"]] / dt / 1e12,
           "workload": "north-star denoiser, batch 1: taped forward + input / parameter gradients (3x the forward's multiply-adds)"}
    net.__dict__.pop("_holo_train_ws", None)
    torch.cuda.empty_cache()
    return res


def respawn_under_torchrun(n: int) -> None:
    """``python bench.py --gpus N`` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_run(args) -> None:
    """Launch plumbing only (CPU-testable): rendezvous, barrier, max-over-ranks reduction, one JSON line from rank 0
    with the LIVE world size.  No kernels run and nothing is measured."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if use_gpu else "gloo")
    dev = "cuda" if use_gpu else "cpu"
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
    pids = [os.getpid()]
    live = 1
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        live = dist.get_world_size()
        mine = torch.tensor([os.getpid()], dtype=torch.int64, device=dev)
        parts = [torch.zeros_like(mine) for _ in range(live)]
        dist.all_gather(parts, mine)
        pids = [int(x.item()) for x in parts]
        # the two exchanges of the N > 1 run on small tensors: frame all_gather and the gradient all-reduce
        from holo_diffusion_amd.ddp import allreduce_gradients
        from holo_diffusion_amd.generate import gather_frames
        fr = torch.full((2, 5, 4, 4), float(rank), device=dev)
        allf = gather_frames({rank: fr}, live, tuple(fr.shape), torch.device(dev))
        g = {"a": torch.full((7, 3), float(rank + 1), device=dev), "b": torch.full((5,), float(rank + 1), device=dev)}
        allreduce_gradients(g, bucket_bytes=64)
        exch_ok = bool(torch.equal(allf[rank], fr)) and all(bool((v == (live + 1) / 2.0).all()) for v in g.values())
        dist.barrier()
        dist.destroy_process_group()
    else:
        exch_ok = True
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": live, "requested_gpus": args.gpus, "max_over_ranks": float(t.item()),
                          "pids": pids, "backend": "nccl" if use_gpu else "gloo", "exchanges_ok": exch_ok}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=40,
                    help="timed 400x400 frames in the render leg: one render call of a whole fly-around (render_flyaround's "
                         "default n_flyaround_poses = 40, flyaround.py:50; generate_samples.py uses 75)")
    ap.add_argument("--flyaround-frames", type=int, default=8,
                    help="frame count of the SECONDARY render leg (0 = skip; the profiling scripts skip it so that every "
                         "render dispatch of a profile has --frames frames).  8 = the call size round 1 reported: 40 000 "
                         "wave tiles on 3 072 resident waves are 13.02 rounds, i.e. 7 % lost to the last round")
    ap.add_argument("--image-size", type=int, default=400)
    ap.add_argument("--workload", choices=["north", "small", "donut128"], default="north",
                    help="north = BASELINE configs[1] (the reported line); small / donut128 = configs[0] / [4] grid "
                         "sizes on the same fp32 path (side measurements, never the reported line)")
    ap.add_argument("--compute-dtype", choices=["f32", "bf16", "f32_bf16x3"], default="f32",
                    help="f32 = the reported line (reference arithmetic); bf16 = opt-in bf16 products / fp32 accumulate in "
                         "the 3x3x3 convolutions (side measurement for the bf16 configurations)")
    ap.add_argument("--noise", choices=["device", "torch"], default="device",
                    help="per-step noise of the timed chain: device = drawn inside the step kernel (holo_ddpm_step_philox, the "
                         "perf mode); torch = torch.randn_like + holo_ddpm_step (the reference's draw; also timed as a side figure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the 128^3 bf16 side workload (BASELINE configs[4] size)")
    ap.add_argument("--no-opt-in", action="store_true", help="skip the side measurements of the opt-in arithmetic modes")
    ap.add_argument("--dry-run", action="store_true", help="launch plumbing only (rendezvous + reductions), no kernels")
    ap.add_argument("--conv-iters", type=int, default=3)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)  # does not return
    if args.dry_run:
        return dry_run(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"# bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting the live world size",
              file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
    w = {"north": NORTH, "small": SMALL, "donut128": DONUT}[args.workload]
    H = W = args.image_size
    warnings.simplefilter("ignore")
    model, usd, msd = build_model(w, H, W, device, compute_dtype=args.compute_dtype)
    net, diff = model.net_3d, model.diffusion
    shape = (1, w["feature_size"]) + (w["resol"],) * 3
    torch.manual_seed(42 + rank)

    # ---------------- denoise leg: K ancestral steps of one chain per GPU
    K, Wm = args.steps, args.warmup
    img = torch.randn(*shape, device=device)
    ts = torch.arange(999, 999 - (K + Wm), -1, device=device, dtype=torch.int64).clamp_min(0)[:, None].contiguous()

    diff.device_noise_seed, diff.device_noise_stream = 42, rank  # (perf mode: Philox noise inside the step kernel)
    t_host = [max(999 - k, 0) for k in range(K + Wm)]

    cl_chain = True  # (every arithmetic mode: the chain stays channels-last; the bf16 storage mode casts its input element-wise)

    def one_step(x, k, mode=args.noise):
        t = ts[k]
        if mode == "device" and not cl_chain:
            return diff._step_device_noise(x, t, net(x, t), t_host[k], True, want_pred=False)[0]
        if mode == "device":
            # the perf chain of ImplicitronGaussianDiffusion.p_sample_loop(device_noise_seed=...): the grid stays in the
            # library's channels-last layout (no layout pass either side of the UNet), one step kernel: clamp + posterior
            # mean + in-kernel Philox noise; pred_xstart is not materialised
            out = net.forward_channels_last(x, t)
            return diff._step_device_noise(x, t, out, t_host[k], True, want_pred=False)[0]
        out = net(x, t)
        eps = torch.randn_like(x)
        sample, _ = diff._step(x, t, out, eps, True)
        return sample

    if args.noise == "device" and cl_chain:
        img = img.permute(0, 2, 3, 4, 1).contiguous()  # (the chain's one conversion; back after the timed steps)
    with torch.no_grad():
        for k in range(Wm):
            img = one_step(img, k)
        barrier_sync(world)
        t0 = time.perf_counter()
        for k in range(Wm, Wm + K):
            img = one_step(img, k)
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0  # this rank's own K steps (before it waits for the others)
        barrier_sync(world)
        dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, device)
    if args.noise == "device" and cl_chain:
        img = img.permute(0, 4, 1, 2, 3).contiguous()
    assert torch.isfinite(img).all()
    steps_per_s = world * K / dt
    # the other noise path over the same K timesteps (side figure, this rank only)
    other_mode = "torch" if args.noise == "device" else "device"
    with torch.no_grad():
        xo = torch.randn(*shape, device=device)
        if other_mode == "device" and cl_chain:
            xo = xo.permute(0, 2, 3, 4, 1).contiguous()
        for k in range(min(3, Wm)):
            xo = one_step(xo, k, other_mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(Wm, Wm + K):
            xo = one_step(xo, k, other_mode)
        torch.cuda.synchronize()
        steps_per_s_other = K / (time.perf_counter() - t0)
    per_rank_steps_per_s = [K / dt_own]
    if world > 1:  # every rank's own rate, gathered for the line rank 0 prints
        tr = torch.tensor([K / dt_own], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(tr) for _ in range(world)]
        dist.all_gather(allr, tr)
        per_rank_steps_per_s = [float(a.item()) for a in allr]

    # ---------------- render leg: frames of one grid per GPU (UNet-at-t=0 refinement hoisted, cached)
    import holo_diffusion_amd as hda
    F = args.frames
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, max(F, 2), -30.0 * (2 * math.pi / 360), 10,
                                                (0.0, -1.0, 0.0), 3.2).to(device)
    vf = torch.clamp(img, -1, 1)
    cams_f = cams[list(range(F))]
    with torch.no_grad():
        # warm-up with the SAME camera count as the timed call (workspace, output tensors and the cached
        # tanh(net_3d(vf, 0)) all reach their steady state before the clock starts)
        for _ in range(2):
            model.render_views(vf, cams_f)
        barrier_sync(world)
        t0 = time.perf_counter()
        out = model.render_views(vf, cams_f)
        barrier_sync(world)
        dtr = time.perf_counter() - t0
    dtr = max_over_ranks(dtr, world, device)
    assert torch.isfinite(out["images_render"]).all()
    rays_per_s = world * F * H * W / dtr
    # the same leg at a second call size (default: the 8-frame call round 1 reported): a secondary figure
    F40 = args.flyaround_frames
    rays_per_s_40, dtr40 = None, None
    if F40 > 0:
        cams40 = hda.get_simple_360_camera_trajectory(2 * math.pi, max(F40, 2), -30.0 * (2 * math.pi / 360), 10,
                                                      (0.0, -1.0, 0.0), 3.2).to(device)[list(range(F40))]
        with torch.no_grad():
            model.render_views(vf, cams40)
            barrier_sync(world)
            t0 = time.perf_counter()
            model.render_views(vf, cams40)
            barrier_sync(world)
            dtr40 = max_over_ranks(time.perf_counter() - t0, world, device)
        rays_per_s_40 = world * F40 * H * W / dtr40

    # the reference's call shape: one render call PER CAMERA (flyaround.py:247-253 loops over the cameras), same grid, the
    # t = 0 refinement cached as above
    n1 = min(F, 8)
    with torch.no_grad():
        singles = [cams[[i]] for i in range(n1)]
        model.render_views(vf, singles[0])
        barrier_sync(world)
        t0 = time.perf_counter()
        for c1 in singles:
            model.render_views(vf, c1)
        barrier_sync(world)
        dts = max_over_ranks(time.perf_counter() - t0, world, device)
    rays_per_s_single = world * n1 * H * W / dts

    # the same call with rendered normals (the released YAMLs' render_normals: true): normals of both passes composited
    # inside render2_kernel<.., NRM> (density scalar field + octahedral normals in LDS)
    rays_per_s_nrm = None
    try:
        fn0 = model._implicit_functions[0]._fn
        fn0.render_normals = True
        with torch.no_grad():
            model.render_views(vf, cams_f)
            barrier_sync(world)
            t0 = time.perf_counter()
            outn = model.render_views(vf, cams_f)
            barrier_sync(world)
            dtn = max_over_ranks(time.perf_counter() - t0, world, device)
        assert "normals_render" in outn and torch.isfinite(outn["normals_render"]).all()
        rays_per_s_nrm = world * F * H * W / dtn
    finally:
        model._implicit_functions[0]._fn.render_normals = False

    # ---------------- roofline of the dominant kernel, hipEvents on the launch stream (holo_unet_time_ops)
    roof = None
    if rank == 0:
        all_ops = net.time_ops(1, args.conv_iters, device)
        ops = [o for o in all_ops if o["op"] == "conv" and o["ksz"] == 3]
        HALO = ("conv_halo_kernel", "conv_wino_kernel", "conv_wino2_kernel", "conv_bf16t_kernel", "conv_wino3_kernel",
                "conv_bf16p_kernel")
        variants = {}
        for o in ops:
            key = (o["kernel"], o["tile_depth"], o["fused_skip"], o["out_dim"], 4 if o["cout"] >= 64 else 2) \
                if o["kernel"] in HALO else (o["kernel"], 0, False, 0, 0)
            v = variants.setdefault(key, dict(ms=0.0, flops=0.0, fexec=0.0, n=0))
            v["ms"] += o["ms"]; v["flops"] += o["flops"]; v["fexec"] += o["flops_executed"]; v["n"] += 1
        for o in ops:  # algorithmic bytes of a launch: input (+ fused skip input) + output + weights, each touched once
            vin = o["out_dim"] ** 3 * (o["stride"] ** 3) / (8 if o["upsample"] else 1)
            o["bytes"] = 4.0 * (vin * o["cin"] + o["out_dim"] ** 3 * o["cout"] + 27 * o["cin"] * o["cout"])
        (kname, tz, sk, od, nwn), dom = max(variants.items(), key=lambda kv: kv[1]["ms"])
        dom_ops = [o for o in ops if o["kernel"] == kname and (kname not in HALO or (
            o["tile_depth"] == tz and o["fused_skip"] == sk and o["out_dim"] == od and (4 if o["cout"] >= 64 else 2) == nwn))]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12        # algorithmic: the reference's multiply-adds x 2
        ach_exec = dom["fexec"] / (dom["ms"] * 1e-3) / 1e12   # what the matrix pipe was actually given
        all_ms = sum(o["ms"] for o in ops)
        all_fl = sum(o["flops"] for o in ops)
        all_fx = sum(o["flops_executed"] for o in ops)
        if kname == "conv_halo_kernel":
            label = f"{kname}<{nwn}, {tz}, {'true' if sk else 'false'}> at {od}^3 output"
            what = "3x3x3 conv3d, LDS voxel-halo implicit GEMM"
        elif kname == "conv_wino_kernel":
            label = f"{kname}<{'true' if sk else 'false'}> at {od}^3 output"
            what = ("3x3x3 conv3d, LDS voxel-halo implicit GEMM in Winograd F(2,3) form along depth: 36 pseudo-taps per two "
                    "output planes instead of 54")
        elif kname == "conv_wino2_kernel":
            label = f"{kname}<{'true' if sk else 'false'}> at {od}^3 output"
            what = ("3x3x3 conv3d, LDS voxel-halo implicit GEMM in Winograd F(2x2,3x3) form over (depth, height): 48 "
                    "pseudo-taps per 2x2 outputs instead of 108")
        elif kname == "conv_wino3_kernel":
            label = f"{kname}<{'true' if sk else 'false'}> at {od}^3 output"
            what = ("3x3x3 conv3d, LDS voxel-halo implicit GEMM in Winograd F(2x2x2,3x3x3) form over all three axes: 64 "
                    "pseudo-taps per 2x2x2 outputs instead of 216; one persistent 512-register wave per SIMD")
        elif kname == "conv_bf16t_kernel":
            label = f"{kname}<{2 if nwn == 4 else 1}, {'true' if sk else 'false'}, true> at {od}^3 output"
            what = ("3x3x3 conv3d on bf16 activations, 8x8x8 output tiles, LDS voxel-halo implicit GEMM, "
                    "v_mfma_f32_32x32x16_bf16 with 4x2 register blocking")
        elif kname == "conv_bf16p_kernel":
            label = f"{kname}<{2 if nwn == 4 else 1}, {'true' if sk else 'false'}> at {od}^3 output"
            what = ("3x3x3 conv3d on bf16 activations, 8x8x8 output tiles, persistent wave-specialised workgroups (4 producer waves "
                    "stage the activated halo into two LDS buffers, 4 consumer waves run v_mfma_f32_32x32x16_bf16 with 4x2 "
                    "register blocking)")
        else:
            label, what = kname, "conv3d"
        traffic = None
        try:  # HBM bytes per launch of this kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
            pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
            ent = pmc.get(label)
            if ent and args.workload == "north":
                traffic = {"bytes_per_launch": ent["fetch_bytes"] + ent["write_bytes"], "fetch_bytes": ent["fetch_bytes"],
                           "write_bytes": ent["write_bytes"],
                           "algorithmic_bytes_per_launch": sum(o["bytes"] for o in dom_ops) / len(dom_ops),
                           "source": ent.get("source")}
        except (OSError, ValueError):
            pass
        # the same kernel's average duration from the committed rocprofv3 kernel trace, per level (every level launches the same
        # persistent grid, so --stats mixes them: scripts/wino3_trace_join.py joins the trace with the plan's op order)
        rocprof_ms = None
        try:
            if kname == "conv_wino3_kernel" and args.workload == "north":
                for l in open(os.path.join(REPO, "profiles", "r06_wino3_64cubed_trace.csv")):
                    f_ = l.strip().split(",")
                    if f_[0] == "# level" and int(f_[1]) == od and bool(int(f_[2])) == bool(sk):
                        rocprof_ms = {"avg_launch_ms": float(f_[4]) * 1e-3, "launches_per_forward": int(f_[3]),
                                      "source": "profiles/r06_wino3_64cubed_trace.csv (rocprofv3 --kernel-trace of this command, "
                                                "dispatches joined with the plan's op order)"}
        except (OSError, ValueError, IndexError):
            pass
        peak = PEAK_FP32_MFMA_TFLOPS if args.compute_dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        # `achieved` / `frac` = what the matrix pipe was actually given (a fraction of a roofline cannot pass 1): the
        # Winograd kernels issue 4/9 (2/3) of the 27-tap multiply-adds.  The reference's algorithmic multiply-adds per
        # second - the figure that decides the step time - are reported as `effective_*`.
        roof = {"bound": "mfma", "kernel": label + f" ({what}, {args.compute_dtype} MFMA)",
                "achieved": ach_exec, "peak": peak, "unit": "TFLOP/s", "frac": ach_exec / peak,
                "effective_tflops": ach, "effective_frac": ach / peak,
                "note": ("achieved/frac count the multiply-adds ISSUED to the matrix pipe (executed flops / average launch "
                         "time, hipEvents on the launch stream); effective_* count the ALGORITHMIC flops of the convolution "
                         "(27 taps, what the reference computes) - the Winograd kernels issue 8/27 (F(2x2x2)) or 4/9 (F(2x2)) of them, so "
                         "effective_frac may pass 1 while frac cannot"),
                "traffic": traffic, "launches_per_forward": dom["n"], "avg_launch_ms": dom["ms"] / dom["n"],
                "rocprofv3_trace": rocprof_ms,
                "algorithmic_gflop_per_launch": dom["flops"] / dom["n"] / 1e9,
                "executed_gflop_per_launch": dom["fexec"] / dom["n"] / 1e9,
                "share_of_conv_time": dom["ms"] / all_ms,
                "all_conv_launches": {"launches_per_forward": len(ops), "ms_per_forward": all_ms,
                                      "algorithmic_gflop_per_forward": all_fl / 1e9,
                                      "executed_gflop_per_forward": all_fx / 1e9,
                                      "achieved": all_fx / (all_ms * 1e-3) / 1e12,
                                      "frac": all_fx / (all_ms * 1e-3) / 1e12 / peak,
                                      "effective_tflops": all_fl / (all_ms * 1e-3) / 1e12,
                                      "effective_frac": all_fl / (all_ms * 1e-3) / 1e12 / peak},
                "by_variant": [{"kernel": k[0], "wave_cols": k[4], "tile_depth": k[1], "fused_skip": k[2], "out_dim": k[3],
                                "launches": v["n"], "ms": v["ms"], "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                "tflops_executed": v["fexec"] / (v["ms"] * 1e-3) / 1e12}
                               for k, v in sorted(variants.items(), key=lambda kv: -kv[1]["ms"])]}
        # the HBM-bound kernel of the step: the streaming 1x1x1 skip connections of the 64^3 level (conv1x1_stream_kernel):
        # algorithmic bytes = input + output + weights, each once, over the average launch time (hipEvents, as above)
        st = [o for o in all_ops if o["op"] == "conv" and o.get("kernel") == "conv1x1_stream_kernel"]
        if st:
            by = sum(4.0 * (o["out_dim"] ** 3 * (o["cin"] + o["cout"]) + o["cin"] * o["cout"]) for o in st) / len(st)
            ms_ = sum(o["ms"] for o in st) / len(st)
            tr = None
            try:
                ent = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))).get(f"conv1x1_stream_kernel<{st[0]['cin'] // 32}>")
                if ent and args.workload == "north":
                    tr = {"bytes_per_launch": ent["fetch_bytes"] + ent["write_bytes"], "fetch_bytes": ent["fetch_bytes"],
                          "write_bytes": ent["write_bytes"], "algorithmic_bytes_per_launch": by, "source": ent.get("source")}
            except (OSError, ValueError):
                pass
            roof["hbm_bound_kernel"] = {"bound": "hbm", "kernel": f"conv1x1_stream_kernel<{st[0]['cin'] // 32}> at {st[0]['out_dim']}^3 "
                                        f"({st[0]['cin']} -> {st[0]['cout']}: a ResBlock's 1x1x1 skip_connection as a launch of its own)",
                                        "achieved": by / (ms_ * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                        "frac": by / (ms_ * 1e-3) / 1e9 / PEAK_HBM_GBPS, "traffic": tr,
                                        "launches_per_forward": len(st), "avg_launch_ms": ms_,
                                        "algorithmic_bytes_per_launch": by,
                                        "tflops": sum(o["flops"] for o in st) / sum(o["ms"] for o in st) / 1e9}
        if os.environ.get("HOLO_BENCH_OPS"):
            for o in all_ops:
                print("# op", json.dumps({**o, "tflops": o["flops"] / (o["ms"] * 1e-3) / 1e12}), file=sys.stderr)
            by_op = {}
            for o in all_ops:
                k = o["op"] if o["op"] != "conv" else f"conv:{o.get('kernel')}"
                by_op.setdefault(k, [0, 0.0])
                by_op[k][0] += 1
                by_op[k][1] += o["ms"]
            print("# per-op totals (launches, ms/forward):", json.dumps({k: (v[0], round(v[1], 4)) for k, v in
                                                                        sorted(by_op.items(), key=lambda kv: -kv[1][1])}),
                  "sum", round(sum(v[1] for v in by_op.values()), 3), file=sys.stderr)
    # ---------------- side measurements (N=1 only, never the reported value): the opt-in matrix-core modes
    alt = None
    if world == 1 and args.compute_dtype == "f32" and args.workload == "north" and not args.no_opt_in:
        alt = {}
        for mode in ("f32_bf16x3", "bf16"):
            net.compute_dtype = mode
            x2 = torch.randn(*shape, device=device)
            ts2 = torch.arange(999, 987, -1, device=device, dtype=torch.int64)[:, None].contiguous()

            def step2(x, k):
                out = net(x, ts2[k])
                sample, _ = diff._step(x, ts2[k], out, torch.randn_like(x), True)
                return sample

            with torch.no_grad():
                for k in range(2):
                    x2 = step2(x2, k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(2, 12):
                    x2 = step2(x2, k)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
            extra = {}
            if mode == "f32_bf16x3":  # the renderer has the same opt-in arithmetic
                model.renderer.compute_dtype = mode
                with torch.no_grad():
                    model.render_views(vf, cams_f)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    model.render_views(vf, cams_f)
                    torch.cuda.synchronize()
                    dtr2 = time.perf_counter() - t0
                model.renderer.compute_dtype = "f32"
                extra = {"rays_per_sec": F * H * W / dtr2, "ms_per_frame": 1e3 * dtr2 / F}
            alt[mode] = {"denoise_steps_per_s": 10 / dt2, "ms_per_step": 1e3 * dt2 / 10, **extra,
                         "arithmetic": {"f32_bf16x3": "3x3x3 convs: fp32 operands split exactly into 3 bf16 terms, 6 bf16 "
                                                      "MFMAs per product, fp32 accumulate (meets the fp32 parity "
                                                      "tolerances); everything else fp32",
                                        "bf16": "bf16 activations in HBM; convolutions and long-sequence attention: bf16 "
                                                "products, fp32 accumulate; fp32 GroupNorm statistics and network "
                                                "input/output (rtol 2e-2)"}[mode]}
        net.compute_dtype = "f32"

    # ---------------- side workload (N=1 only): BASELINE configs[4] donut.yaml size, 128^3x32 in the bf16 storage mode,
    # same weights (the UNet's parameters do not depend on the grid size): 3 warm + 5 timed DDPM steps
    side = None
    if world == 1 and args.workload == "north" and args.compute_dtype == "f32" and not args.no_side:
        side = {"teddybear_turntable30": side_teddybear_turntable(model, device),
                "donut128_bf16": side_donut128(usd, device), "batched_chains_f32": side_batched_chains(net, diff, w, device),
                "training_step_f32": side_training_step(net, w, device)}

    # ---------------- the one exchange of the path: all_gather of the rendered frames over RCCL / xGMI.
    # Every rank contributes the frames of its own sample (the timed render call's output); timed separately from the
    # render leg so that `rays_per_sec` stays the compute figure.
    gather = None
    if world > 1:
        from holo_diffusion_amd.generate import gather_frames
        frames = torch.cat([out["images_render"], out["depths_render"], out["masks_render"]], dim=1)  # (F,5,H,W)
        shape5 = tuple(frames.shape)
        gather_frames({rank: frames}, world, shape5, device)  # warm-up (RCCL communicator / xGMI rings)
        barrier_sync(world)
        t0 = time.perf_counter()
        allf = gather_frames({rank: frames}, world, shape5, device)
        barrier_sync(world)
        dtg = max_over_ranks(time.perf_counter() - t0, world, device)
        ok = bool(torch.equal(allf[rank], frames)) and tuple(allf.shape) == (world,) + shape5
        okt = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        nbytes = frames.numel() * 4
        gather = {"gather_ms": 1e3 * dtg, "rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(),
                  "bytes_per_rank": nbytes, "all_gather_GBps_per_rank": nbytes * (world - 1) / dtg / 1e9,
                  "verified": bool(okt.item() == 1.0),
                  "what": f"all_gather of {F} frames x (rgb, depth, mask) @{H}x{W} fp32 per rank (generate.gather_frames)"}

    # ---------------- the optional row of SURVEY 8e: ONE grid's 30-view turntable with the cameras sharded over the ranks
    # (generate.render_views_sharded: broadcast of the 33.5 MB grid from rank 0, cameras k mod N per rank in one render call,
    # one all_gather per output) against the same turntable rendered by rank 0 alone
    sharded = None
    if world > 1 and args.workload == "north":
        from holo_diffusion_amd.generate import render_views_sharded
        cams30 = hda_cams(30, device)
        with torch.no_grad():
            render_views_sharded(model, vf if rank == 0 else None, cams30, src_rank=0, device=device)  # warm-up
            barrier_sync(world)
            t0 = time.perf_counter()
            sh = render_views_sharded(model, vf if rank == 0 else None, cams30, src_rank=0, device=device)
            barrier_sync(world)
            dtsh = max_over_ranks(time.perf_counter() - t0, world, device)
            model.render_views(sh["voxel_features"], cams30)
            barrier_sync(world)
            t0 = time.perf_counter()
            alone = model.render_views(sh["voxel_features"], cams30)
            torch.cuda.synchronize()
            dtal = time.perf_counter() - t0
            barrier_sync(world)
        oks = torch.tensor([1.0 if torch.equal(sh["images_render"], alone["images_render"]) else 0.0], device=device)
        dist.all_reduce(oks, op=dist.ReduceOp.MIN)
        sharded = {"ms_sharded": 1e3 * dtsh, "ms_one_rank_alone": 1e3 * dtal, "speedup": dtal / dtsh, "views": 30,
                   "rays_per_sec_sharded": 30 * H * W / dtsh, "bit_equal_to_one_rank": bool(oks.item() == 1.0),
                   "what": "30-view turntable of ONE grid: broadcast (33.5 MB) + cameras k mod N per rank + all_gather of the "
                           "frames (generate.render_views_sharded), every rank ends up with all 30 frames"}

    # ---------------- the training branch's exchange: all-reduce of the denoiser's parameter gradients
    # (165 M fp32 = 0.66 GB at the north-star size) in 64 MB buckets over RCCL / xGMI (holo_diffusion_amd/ddp.py); synthetic
    # gradients of the real parameter shapes, timed apart from the denoise and render legs
    grad_exchange = None
    if world > 1 and args.workload == "north":
        from holo_diffusion_amd.ddp import DEFAULT_BUCKET_BYTES, allreduce_gradients, plan_buckets
        gsd = {k: torch.full(tuple(v.shape), float(rank + 1), device=device) for k, v in usd.items()}
        nb = sum(v.numel() for v in gsd.values()) * 4
        allreduce_gradients({k: v.clone() for k, v in list(gsd.items())[:4]})  # warm-up (communicator, rings)
        barrier_sync(world)
        t0 = time.perf_counter()
        allreduce_gradients(gsd)
        barrier_sync(world)
        dta = max_over_ranks(time.perf_counter() - t0, world, device)
        mean = (world + 1) / 2.0
        okg = torch.tensor([1.0 if all(bool((v == mean).all()) for v in list(gsd.values())[:8]) else 0.0], device=device)
        dist.all_reduce(okg, op=dist.ReduceOp.MIN)
        grad_exchange = {"allreduce_ms": 1e3 * dta, "bytes": nb, "buckets": len(plan_buckets(gsd, DEFAULT_BUCKET_BYTES)),
                         "bucket_bytes": DEFAULT_BUCKET_BYTES, "rccl_world_size": dist.get_world_size(),
                         "backend": dist.get_backend(), "algbw_GBps": nb / dta / 1e9,
                         "busbw_GBps": nb * 2 * (world - 1) / world / dta / 1e9, "verified": bool(okg.item() == 1.0),
                         "what": "average of the north-star denoiser's fp32 parameter gradients over the ranks (pack + bucketed "
                                 "all_reduce + scale + unpack)"}
        del gsd
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, usd, msd)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # render roofline (logical gather bytes and collapsed-MLP flops per ray)
        # the reference evaluates 64 coarse + 128 fine points per ray; the kernel evaluates each of the 128 distinct
        # points once (the fine pass re-uses the 64 coarse values it already holds), so the executed work is 128
        C = w["feature_size"]
        samples_ref, samples = 64 + 128, 128
        gather_bytes_per_ray = samples * 8 * C * 4
        mlp_flops_per_ray = samples * (2 * 257 * C + 2 * 3 * 256 + 2 * 3 * 27)
        render_traffic = None
        try:  # fabric-side bytes per FRAME of the render kernel from the committed PMC pass
            pmc_all = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
            ent = pmc_all.get("render2_kernel<16, 64, false, 12, false>") or pmc_all.get("render2_kernel<16, 64, false, 12>")
            if ent and args.workload == "north":
                fpl = ent.get("frames_per_launch", 1)
                render_traffic = {"bytes_per_frame": (ent["fetch_bytes"] + ent["write_bytes"]) / fpl,
                                  "fetch_bytes_per_frame": ent["fetch_bytes"] / fpl,
                                  "write_bytes_per_frame": ent["write_bytes"] / fpl,
                                  "compulsory_bytes_per_frame": 4.0 * (w["feature_size"] * w["resol"] ** 3 + 10 * H * W),
                                  "source": ent.get("source")}
        except (OSError, ValueError):
            pass
        line = {
            "metric": "denoise-steps/sec + rendered-rays/sec, 64^3x32 grid @400^2 render",
            "value": steps_per_s, "unit": "denoise-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "bf16": "bf16 activations in HBM, bf16 products / f32 accumulate (convolutions, long-sequence "
                              "attention), f32 GroupNorm statistics, f32 network input/output",
                      "f32_bf16x3": "f32 operands split into 3 bf16 terms, 6 bf16 MFMAs per product, f32 accumulate "
                                    "(3x3x3 convs); f32 elsewhere"}[args.compute_dtype],
            "data": "synthetic",
            "config": {"workload": ({"north": "apple.yaml single-sample DDPM, 64^3x32 grid, 1 MI355X per chain; ",
                                     "small": "32^3x16 plumbing grid; ",
                                     "donut128": f"128^3x32 grid (donut.yaml size), compute_dtype={args.compute_dtype}; "}[args.workload])
                       + f"{F} frames @{H}x{W}, 64 coarse + 128 fine samples/ray",
                       "parallelism": f"{world} independent chains (sample sharding), no data-path collective"},
            "rays_per_sec": rays_per_s, "ms_per_frame": 1e3 * dtr / F, "frames": F,
            "rays_per_sec_with_normals": rays_per_s_nrm,
            "rays_per_sec_single_frame_calls": rays_per_s_single, "ms_per_single_frame_call": 1e3 * dts / n1,
            "rays_per_sec_second_call_size": rays_per_s_40, "second_call_frames": F40,
            "ms_per_frame_second_call_size": (1e3 * dtr40 / F40) if F40 > 0 else None,
            "unet_tflops": FLOPS_PER_STEP[w["resol"]] * steps_per_s / world / 1e12,
            "unet_workspace_bytes": net.workspace_bytes(1, device),
            "roofline": roof,
            "roofline_render": {"bound": "mfma+gather", "traffic": render_traffic,
                                "kernel": "render2_kernel<16, 64, false, 12, false> (persistent: one 12-wave workgroup per CU walks the "
                                          "4-ray x 8-depth wave tiles of all frames of the call; per-ray values stay in LDS)",
                                "evaluations_per_ray_executed": samples, "evaluations_per_ray_reference": samples_ref,
                                "logical_gather_GBps": gather_bytes_per_ray * rays_per_s / world / 1e9,
                                "peak_hbm_GBps": PEAK_HBM_GBPS,
                                "mlp_tflops_collapsed": mlp_flops_per_ray * rays_per_s / world / 1e12,
                                "peak_tflops": PEAK_FP32_MFMA_TFLOPS,
                                "frac_mfma": mlp_flops_per_ray * rays_per_s / world / 1e12 / PEAK_FP32_MFMA_TFLOPS},
            "cpu_baseline": cpu,
            "frame_gather": gather, "grad_exchange": grad_exchange, "camera_sharded_turntable": sharded,
            "rccl_world_size": (gather or {}).get("rccl_world_size", 1), "gather_ms": (gather or {}).get("gather_ms"),
            "per_rank_steps_per_s": per_rank_steps_per_s,
            "step_noise": {"timed": args.noise, "what": {"device": "the sampler's perf chain: grid kept channels-last (holo_unet_forward_cl, no layout "
                                                                   "passes), Philox4x32-10 + Box-Muller noise inside the step kernel "
                                                                   "(holo_ddpm_step_philox; no randn launch, pred_xstart not written)",
                                                         "torch": "torch.randn_like + holo_ddpm_step (the reference's draw)"}[args.noise],
                           f"steps_per_s_with_{other_mode}_noise_one_rank": steps_per_s_other},
            "side_workloads": side,
            "opt_in_modes_not_reported": alt,
        }
        print(json.dumps(line))


if __name__ == "__main__":
    main()
