#!/usr/bin/env python3
"""TEST-ONLY: drive the host-emulated build of the kernel sources (tests/emu/libholo_emu.so)
through the real C ABI and compare with the oracle / golden vectors.

Purpose: catch indexing / layout / planning bugs in the GPU-less development container before
spending GPU minutes.  Build with ``make -C holo_diffusion_amd/csrc emu``.  Slow (host threads
stand in for GPU lanes); not part of the default pytest run.

Usage: python tests/emu/emu_check.py [unet] [ddpm] [render] [ops]
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

import numpy as np
import torch

os.environ["HOLO_KEEP_INTERMEDIATES"] = "1"

from holo_diffusion_amd import _lib  # noqa: E402
from holo_diffusion_amd.weights import synth_state_dict  # noqa: E402
from oracle import unet_oracle as uo, render_oracle as ro, diffusion_oracle as do  # noqa: E402
from oracle.common import TINY_CFG, seeded_input, np_noise  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
lib = _lib.bind(C.CDLL(os.path.join(REPO, "tests", "emu", "libholo_emu.so")))


def ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def report(name, got, ref, tol=1e-4):
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    ok = err <= tol * max(scale, 1.0)
    print(f"  {'OK ' if ok else 'BAD'} {name}: max|d|={err:.3e} max|ref|={scale:.3e}")
    return ok


def make_ctx():
    ctx = C.c_void_p()
    _lib.check(lib, lib.holo_ctx_create(0, C.byref(ctx)), "ctx_create")
    return ctx


def make_unet(ctx, cfg: uo.UNetCfg, sd):
    c = _lib.make_unet_cfg(cfg.image_size, cfg.in_channels, cfg.out_channels, cfg.model_channels, cfg.num_res_blocks,
                           cfg.channel_mult, cfg.attention_resolutions, cfg.num_heads)
    net = C.c_void_p()
    _lib.check(lib, lib.holo_unet_create(ctx, C.byref(c), C.byref(net)), "unet_create")
    n = lib.holo_unet_num_params(net)
    shapes = uo.unet_param_shapes(cfg)
    assert n == len(shapes), (n, len(shapes))
    name = C.create_string_buffer(256)
    shp = (C.c_int64 * 8)()
    nd = C.c_int()
    for i in range(n):
        _lib.check(lib, lib.holo_unet_param_info(net, i, name, 256, shp, C.byref(nd)), "param_info")
        k = name.value.decode()
        assert tuple(shp[:nd.value]) == tuple(shapes[k]), (k, tuple(shp[:nd.value]), shapes[k])
        t = sd[k].contiguous()
        _lib.check(lib, lib.holo_unet_set_param(net, k.encode(), ptr(t), 0, t.dim(), _lib.shape_array(t.shape), None),
                   f"set_param {k}")
    return net


def unet_forward(net, cfg, x, t):
    B = x.shape[0]
    ws_bytes = lib.holo_unet_workspace_bytes(net, B)
    ws = torch.zeros(ws_bytes // 4 + 64, dtype=torch.float32)
    y = torch.empty(B, cfg.out_channels, *x.shape[2:])
    _lib.check(lib, lib.holo_unet_forward(net, B, ptr(x), ptr(t), ptr(y), ptr(ws), ws_bytes, None), "unet_forward")
    return y, ws


def check_unet():
    print("== tiny UNet forward through the emulated kernels vs reference golden")
    cfg = TINY_CFG
    sd = synth_state_dict(uo.unet_param_shapes(cfg), 1234)
    gold = np.load(os.path.join(GOLD, "tiny_unet.npz"))
    ctx = make_ctx()
    net = make_unet(ctx, cfg, sd)
    x = seeded_input(cfg, 7 + 500)
    t = torch.tensor([500], dtype=torch.int64)
    t0 = time.time()
    y, ws = unet_forward(net, cfg, x, t)
    print(f"  forward took {time.time() - t0:.1f}s (emulated)")
    ok = True
    n_in = sum(1 for k in gold.files if k.startswith("t500.input_blocks"))
    n_out = sum(1 for k in gold.files if k.startswith("t500.output_blocks"))
    tags = [f"input_blocks.{i}" for i in range(n_in)] + ["middle_block"] + [f"output_blocks.{i}" for i in range(n_out)]
    for tag in tags:
        ref = torch.from_numpy(gold[f"t500.{tag}"])
        dst = torch.empty_like(ref)
        numel = C.c_int64()
        _lib.check(lib, lib.holo_unet_fetch_block(net, tag.encode(), ptr(dst), dst.numel(), C.byref(numel), ptr(ws),
                                                  None), f"fetch {tag}")
        ok &= report(tag, dst, ref)
    ok &= report("y", y, torch.from_numpy(gold["t500.y"]))
    return ok


def check_unet_wide(image=8, mc=64, mult=(1, 2), attn=(2,), in_ch=16, bf16=False, split=False):
    """Wider tiny net: model_channels 64 puts the ResBlocks of the top level on the LDS-halo kernel with the fused
    1x1x1 skip connection and split-K (no reference golden at this size: compared with the pinned oracle)."""
    print(f"== wide tiny UNet (image {image}, mc {mc}, mult {mult}) through the emulated kernels vs oracle")
    cfg = uo.UNetCfg(image_size=image, in_channels=in_ch, out_channels=in_ch, model_channels=mc, num_res_blocks=2,
                     channel_mult=mult, attention_resolutions=attn, num_heads=2)
    sd = synth_state_dict(uo.unet_param_shapes(cfg), 99)
    ctx = make_ctx()
    net = make_unet(ctx, cfg, sd)
    if bf16:
        print("  (bf16 products in the halo convolutions, fp32 accumulate: tolerance 2e-2)")
        _lib.check(lib, lib.holo_unet_set_compute_dtype(net, _lib.HOLO_DTYPE_BF16), "set_compute_dtype")
    if split:
        print("  (fp32 operands split into three bf16 terms, six bf16 MFMAs per product: fp32 tolerance)")
        _lib.check(lib, lib.holo_unet_set_compute_dtype(net, _lib.HOLO_DTYPE_F32_BF16X3), "set_compute_dtype")
    x = torch.from_numpy(np_noise(11, (1, in_ch, image, image, image)))
    t = torch.tensor([321], dtype=torch.int64)
    trace = {}
    ref = uo.unet_forward(sd, cfg, x, t, trace)
    t0 = time.time()
    y, ws = unet_forward(net, cfg, x, t)
    print(f"  forward took {time.time() - t0:.1f}s (emulated)")
    ok = True
    for tag, r in trace.items():
        if not (tag.startswith("input_blocks") or tag.startswith("output_blocks") or tag == "middle_block"):
            continue
        dst = torch.empty_like(r)
        numel = C.c_int64()
        _lib.check(lib, lib.holo_unet_fetch_block(net, tag.encode(), ptr(dst), dst.numel(), C.byref(numel), ptr(ws),
                                                  None), f"fetch {tag}")
        ok &= report(tag, dst, r, 2e-2 if bf16 else 1e-4)
    ok &= report("y", y, ref, 2e-2 if bf16 else 1e-4)
    return ok


def check_ddpm():
    print("== ddpm step vs oracle")
    ctx = make_ctx()
    orc = do.DiffusionOracle(1000)
    T = orc.tables
    tab = np.stack([T["posterior_mean_coef1"], T["posterior_mean_coef2"], T["posterior_log_variance_clipped"],
                    np.zeros(1000)], axis=1).astype(np.float32)
    tab_t = torch.from_numpy(tab).contiguous()
    ok = True
    for tt in (999, 500, 1, 0):
        shape = (2, 4, 4, 4, 4)
        x = torch.from_numpy(np_noise(1, shape))
        mo = torch.from_numpy(np_noise(2, shape)) * 1.5
        nz = torch.from_numpy(np_noise(3, shape))
        t = torch.tensor([tt, tt], dtype=torch.int64)
        ref = orc.p_sample(lambda a, b: mo, x, t, nz, True)
        s = torch.empty(shape)
        p = torch.empty(shape)
        _lib.check(lib, lib.holo_ddpm_step(ctx, ptr(tab_t), 1000, ptr(t), 2, 256, ptr(x), ptr(mo), ptr(nz), 1, ptr(s),
                                           ptr(p), None), "ddpm_step")
        ok &= report(f"sample t={tt}", s, ref["sample"], 1e-6)
        ok &= report(f"pred_xstart t={tt}", p, ref["pred_xstart"], 1e-7)
    return ok


def check_render(C_feat=32, R=8, H=8, W=16, n_fine=64, split=False):
    print(f"== fused renderer vs oracle (R={R}, C={C_feat}, {H}x{W}, n_fine={n_fine}{', bf16x3 split' if split else ''})")
    ctx = make_ctx()
    rcfg = ro.RenderCfg(resol=R, feature_size=C_feat, image_height=H, image_width=W, n_pts_fine=n_fine)
    shapes = ro.render_mlp_param_shapes(rcfg)
    sd = synth_state_dict(shapes, 4321)
    # make densities interesting: positive bias on the density row
    sd["_density_net.mlp.3.0.bias"][-1] += float(os.environ.get("EMU_DBIAS", "0.0"))
    grid = torch.tanh(torch.from_numpy(np_noise(7, (1, C_feat, R, R, R))))
    cams = ro.simple_360_cameras(4)
    cam = {k: v[1:2] for k, v in cams.items()}
    ref = ro.render(grid, sd, cam, rcfg, return_coarse=True)
    c = _lib.make_render_cfg(R, C_feat, H, W, n_pts_fine=n_fine)
    r = C.c_void_p()
    _lib.check(lib, lib.holo_renderer_create(ctx, C.byref(c), C.byref(r)), "renderer_create")
    for k, v in sd.items():
        v = v.contiguous()
        _lib.check(lib, lib.holo_renderer_set_param(r, k.encode(), ptr(v), 0, v.dim(), _lib.shape_array(v.shape), None),
                   f"set {k}")
    _lib.check(lib, lib.holo_renderer_commit(r, None), "commit")
    if split:
        _lib.check(lib, lib.holo_renderer_set_compute_dtype(r, _lib.HOLO_DTYPE_F32_BF16X3), "set_compute_dtype")
    hc = _lib.HoloCamera()
    for i, v in enumerate(cam["R"].reshape(-1).tolist()):
        hc.R[i] = v
    for i, v in enumerate(cam["T"].reshape(-1).tolist()):
        hc.T[i] = v
    for i in range(2):
        hc.focal[i] = float(cam["focal"].reshape(-1)[i])
        hc.principal_point[i] = float(cam["pp"].reshape(-1)[i])
    img, dep, msk = torch.empty(1, 3, H, W), torch.empty(1, 1, H, W), torch.empty(1, 1, H, W)
    imgc, depc, mskc = torch.empty(1, 3, H, W), torch.empty(1, 1, H, W), torch.empty(1, 1, H, W)
    wsb = lib.holo_render_workspace_bytes(r, 1, 0)
    ws = torch.zeros(wsb // 4 + 64)
    t0 = time.time()
    _lib.check(lib, lib.holo_render(r, ptr(grid), C.byref(hc), 1, ptr(img), ptr(dep), ptr(msk), ptr(imgc), ptr(depc),
                                    ptr(mskc), None, None, ptr(ws), wsb, None), "render")
    print(f"  render took {time.time() - t0:.1f}s (emulated)")
    ok = True
    ok &= report("coarse rgb", imgc, ref["images_coarse"], 2e-4)
    ok &= report("coarse depth", depc, ref["depths_coarse"], 2e-4)
    ok &= report("coarse mask", mskc, ref["masks_coarse"], 2e-4)
    ok &= report("fine rgb", img, ref["images_render"], 2e-4)
    ok &= report("fine depth", dep, ref["depths_render"], 2e-4)
    ok &= report("fine mask", msk, ref["masks_render"], 2e-4)
    print(f"  (mask range {ref['masks_render'].min().item():.3f}..{ref['masks_render'].max().item():.3f})")
    return ok


if __name__ == "__main__":
    what = sys.argv[1:] or ["ddpm", "render", "unet"]
    allok = True
    torch.set_num_threads(1)
    if "ddpm" in what:
        allok &= check_ddpm()
    if "render" in what:
        allok &= check_render()
    if "render_split" in what:
        allok &= check_render(split=True)
    if "render16" in what:
        allok &= check_render(C_feat=16, n_fine=16)
    if "unet" in what:
        allok &= check_unet()
    if "unet_wide" in what:
        allok &= check_unet_wide()
    if "unet_wide16" in what:  # 16^3 top level: 8 spatial tiles per conv, several items per workgroup
        allok &= check_unet_wide(image=16)
    if "unet_wide_bf16" in what:
        allok &= check_unet_wide(bf16=True)
    if "unet_wide_split" in what:
        allok &= check_unet_wide(split=True)
    if "unet_attn_bf16" in what:  # attention at T = 512 through the bf16 flash kernel (HOLO_BF16_FLASH_MIN_T=0)
        os.environ["HOLO_BF16_FLASH_MIN_T"] = "0"
        allok &= check_unet_wide(attn=(1,), bf16=True)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
