"""GPU parity at the sizes BASELINE.json's configs name (the cases round 1 left to property checks):

  configs[0]  unet_with_no_diffusion.yaml  32^3 x 16 grid, 1 camera @128x128, 64 coarse + 16 new fine samples/ray
  configs[1]  apple.yaml                   64^3 x 32 grid @400x400: >= 4096 rays of one frame (borders included)
                                           against the oracle evaluated on exactly those rays
  configs[3]  teddybear.yaml               30-camera turntable in ONE render call (several launch groups, ragged
                                           last group), progressive denoising with a render after every step
  (configs[4], donut.yaml 128^3 bf16, lives in test_gpu_unet.py::test_128_cubed_forward_vs_oracle)

plus known-answer cases the unpinned renderer half lacked: non-square image with a non-zero principal point, a camera
closer to the scene centre than scene_extent (the depth-bound clamp), an inverse-CDF sample exactly on a CDF knot.

Tolerances (fp32): rgb / mask 2e-4 absolute, depth 2e-4 x far plane (renderer alone); 1e-3 when the frame goes
through the UNet first.  HOLO_TEST_EMU=1 (development container, host emulation of the kernels) shrinks the sizes.
"""
import math
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import holo_diffusion_amd as hda  # noqa: E402
from holo_diffusion_amd.generate import render_flyaround  # noqa: E402
from holo_diffusion_amd.render import EvaluationMode  # noqa: E402
from oracle import diffusion_oracle as do  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from oracle.common import np_noise  # noqa: E402

EMU = os.environ.get("HOLO_TEST_EMU") == "1"
TINY_UNET = dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2))
NORTH_UNET = dict(model_channels=64, channel_mult=(1, 1, 2, 4, 8), attention_resolutions=(4, 8))
FAR = 14.0


@pytest.fixture(scope="module")
def gu():
    import tests.gpu_utils as g
    return g


def _cams(n, up=(0.0, -1.0, 0.0)):
    return hda.get_simple_360_camera_trajectory(2 * math.pi, n, -30.0 * (2 * math.pi / 360), 10, up, 3.2)


def _border_and_random_rays(H, W, n_random, seed):
    """Flat pixel indices: the four corners, the full first/last row and column, and n_random interior pixels."""
    idx = {0, W - 1, (H - 1) * W, H * W - 1}
    idx.update(range(W))
    idx.update(range((H - 1) * W, H * W))
    idx.update(range(0, H * W, W))
    idx.update(range(W - 1, H * W, W))
    g = torch.Generator().manual_seed(seed)
    idx.update(torch.randint(0, H * W, (n_random,), generator=g).tolist())
    return torch.tensor(sorted(idx), dtype=torch.long)


def _check_rays(preds, ref, idx, H, W, tol=2e-4, coarse=None):
    """rgb and mask: EVERY ray within `tol`.  Depth: 99 % of the rays within tol x far, every ray within 5e-3 x far.
    Why depth gets a quantile: sample_pdf switches `den = cdf_a - cdf_b` to 1 when it is below eps = 1e-5, and for an
    opaque ray an empty bin has den = 1e-5 / (1 + 6e-4) - 0.6 % under the threshold, while the fp32 cdf values it is the
    difference of are quantised to 6e-8 (0.6 % of it).  Which side an empty bin falls on is therefore decided by the
    LAST BIT of the cumulative sums; any two correct evaluations (the reference on CPU vs on CUDA, the oracle vs the
    oracle on a grid perturbed by 1e-6: 0.7 % of the rays move a new depth by > 1e-3) disagree on it for a few per cent
    of the rays, where one importance sample then sits elsewhere inside an (almost) empty interval.  The colour and the
    mask barely notice; sum w z does, at the 1e-3 level, on those rays."""
    flat = lambda t: t.reshape(t.shape[1], H * W).t().cpu()[idx]  # noqa: E731  (1,c,H,W) -> (n,c)
    for k, rk in (("images_render", "rgb"), ("masks_render", "mask")):
        err = (flat(preds[k]) - ref[rk]).abs().max().item()
        assert err < tol, (k, err)
    e = (flat(preds["depths_render"]) - ref["depth"]).abs().flatten()
    q99 = e.kthvalue(max(1, int(0.99 * e.numel())))[0].item()
    assert q99 < tol * FAR and e.max().item() < 5e-3 * FAR, ("depths_render", q99, e.max().item(), e.median().item())
    if coarse is not None:  # the coarse pass has no resampling: every ray, every output
        cf = lambda t: t.permute(0, 3, 1, 2).reshape(t.shape[3], H * W).t().cpu()[idx]  # noqa: E731
        assert (cf(coarse.features) - ref["rgb_c"]).abs().max().item() < tol
        assert (cf(coarse.masks) - ref["mask_c"]).abs().max().item() < tol
        assert (cf(coarse.depths) - ref["depth_c"]).abs().max().item() < tol * FAR


def test_north_star_ray_subset_vs_oracle(gu):
    """configs[1]: 64^3 x 32 grid at 400x400; every border pixel + 4096 random pixels of one frame against the oracle
    evaluated on exactly those rays (the oracle takes arbitrary rays; a full frame would take minutes)."""
    R, C, H, W, nrand = (8, 32, 40, 40, 256) if EMU else (64, 32, 400, 400, 4096)
    model, _, _, rcfg, msd = gu.make_model(R, C, H, W, TINY_UNET if EMU else NORTH_UNET)
    model.net_3d_enabled = False
    grid = torch.tanh(torch.from_numpy(np_noise(7, (1, C, R, R, R))))
    cams = _cams(4)
    preds = model(camera=cams[1].to(gu.DEV), evaluation_mode=EvaluationMode.EVALUATION, voxel_features=grid.to(gu.DEV))
    idx = _border_and_random_rays(H, W, nrand, 5)
    assert len(idx) >= nrand // 2 + 2 * (H + W) - 4
    o, d, l = ro.make_rays(gu.cam_dict(cams, 1), rcfg)
    ref = ro.render_rays(grid, msd, o[idx], d[idx], l[idx], rcfg)
    _check_rays(preds, ref, idx, H, W, coarse=preds["rendered"].prev_stage)
    # ---- the control experiment behind the depth quantile of _check_rays -------------------------------------------
    # sample_pdf switches den = cdf_a - cdf_b to 1 below eps = 1e-5.  A ray is FRAGILE when one of its importance
    # samples has a raw den within 0.3 eps of that switch: the cdf is a normalised running sum of coarse weights that two
    # correct evaluations reproduce to ~1e-6 (fast exp / sigmoid, reassociated dot products), so there the low bits
    # decide on which side of the switch the sample falls - and the sample lands elsewhere in an (almost) empty bin.
    # (1) every ray the HIP path has outside 2e-4 x far is fragile, i.e. every non-fragile ray is inside, on EVERY ray;
    # (2) the SAME holds between the oracle and the oracle on a grid perturbed by 1e-6 (no HIP code involved), with a
    #     disagreement rate of the same order: the quantile is a property of sample_pdf, not of the kernel.
    margin = 0.3 * rcfg.sample_pdf_eps
    gap = (ref["pdf_denom"] - rcfg.sample_pdf_eps).abs().min(dim=1)[0]
    fragile = gap <= margin
    flat = lambda t: t.reshape(t.shape[1], H * W).t().cpu()[idx]  # noqa: E731
    e_hip = (flat(preds["depths_render"]) - ref["depth"]).abs().flatten()
    bad_hip = e_hip >= 2e-4 * FAR
    pert = grid + 1e-6 * torch.from_numpy(np_noise(8, tuple(grid.shape)))
    ref2 = ro.render_rays(pert, msd, o[idx], d[idx], l[idx], rcfg)
    gap2 = torch.minimum(gap, (ref2["pdf_denom"] - rcfg.sample_pdf_eps).abs().min(dim=1)[0])
    bad_ctl = (ref2["depth"] - ref["depth"]).abs().flatten() >= 2e-4 * FAR
    print(f"depth control: {int(fragile.sum())}/{len(idx)} rays fragile (a den within {margin:.1e} of eps); outside 2e-4*far: "
          f"HIP vs oracle {int(bad_hip.sum())} (largest gap among them {float(gap[bad_hip].max()) if bad_hip.any() else 0:.2e}), "
          f"oracle vs oracle(grid + 1e-6) {int(bad_ctl.sum())} (largest gap {float(gap2[bad_ctl].max()) if bad_ctl.any() else 0:.2e})")
    for j in torch.nonzero(bad_hip).flatten().tolist():  # diagnostics of every ray outside the depth tolerance
        dh, dr_ = float(flat(preds["depths_render"])[j]), float(ref["depth"][j])
        print(f"  bad ray {j} (pixel {int(idx[j])}): depth HIP {dh:.5f} oracle {dr_:.5f} perturbed-oracle {float(ref2['depth'][j]):.5f} | "
              f"mask {float(ref['mask'][j]):.5f} (HIP d {float((flat(preds['masks_render'])[j] - ref['mask'][j]).abs()):.1e}) | "
              f"rgb d {float((flat(preds['images_render'])[j] - ref['rgb'][j]).abs().max()):.1e} | coarse depth d "
              f"{float((preds['rendered'].prev_stage.depths.reshape(-1)[idx[j]].cpu() - ref['depth_c'][j]).abs()):.1e} | gap {float(gap[j]):.2e} "
              f"| smallest den {float(ref['pdf_denom'][j].min()):.3e}")
    assert not (bad_hip & ~fragile).any(), ("a non-fragile ray misses the depth tolerance", int((bad_hip & ~fragile).sum()),
                                            float(gap[bad_hip].max()))
    assert not (bad_ctl & (gap2 > margin)).any(), ("control: a non-fragile ray moved", float(gap2[bad_ctl].max()))
    assert (ref2["rgb"] - ref["rgb"]).abs().max() < 2e-4  # colour and mask barely notice
    if not EMU:
        assert int(bad_hip.sum()) <= 3 * int(bad_ctl.sum()) + 0.01 * len(idx)  # same order as the control's own rate


def test_north_star_full_frame_vs_oracle(gu):
    """configs[1], EVERY ray: one whole 400 x 400 frame of the 64^3 x 32 grid against ``ro.render_rays`` on all 160 000
    rays (about a minute of host time).  Every NON-FRAGILE ray: rgb / mask within 2e-4, depth within 2e-4 x far, both
    passes.  A ray is fragile when the reference algorithm itself is discontinuous at it within float32 rounding:
      (a) an importance sample with a raw ``den`` within 0.3 eps of sample_pdf's ``den < eps -> 1`` switch (see
          test_north_star_ray_subset_vs_oracle), or
      (b) a raw density of the LAST sample (the far bound, whose interval is the raymarcher's background_opacity = 1e10)
          within 1e-4 of zero: density_relu * 1e10 makes the ray fully opaque or leaves it as it was on the SIGN of that
          value - seen on 1 of 160 000 rays of this frame (oracle mask 1.0 / depth 11.6, kernel 0.553 / 5.3).
      (c) any other ray outside the tolerance must be one the ORACLE itself moves by a comparable amount (>= a quarter of
          the kernel's distance) under a perturbation of float32-rounding size, run on exactly those rays: the grid
          + 1e-6 noise (the control of the subset test), or the inverse-CDF abscissae u +- 2e-6.  The latter is the
          rounding of the cdf itself - a running sum of 64 float32 terms near 1, reproducible to ~2e-6 between two correct
          evaluations - and it is what these rays amplify: an importance sample in a bin of raw ``den`` ~ 1.5e-5 (just
          ABOVE the eps switch, so not of kind (a)) moves by 2e-6 / den = 13 % of the bin.  Seen on 2 ... 14 rays of the
          frame depending on the host that runs the oracle (torch's CPU reductions round differently with the thread
          count): on pixel 81 331 the ORACLE ITSELF gives depth 5.9304 on the 256-thread GPU host, 5.9165 in the 8-thread
          development container, 5.9154 in float64 (kernel: 5.9353); every such ray moves by 2e-3 ... 1.5e-2 under
          u +- 2e-6, ordinary rays by 1e-5 ... 5e-5.
    Fragile rays must be few (<= 0.1 %) and stay within 5e-3 x far unless of kind (b).  At most TWO rays of the frame may
    stay unexplained, within the same loose bound."""
    R, C, H, W = (8, 32, 24, 24) if EMU else (64, 32, 400, 400)
    model, _, _, rcfg, msd = gu.make_model(R, C, H, W, TINY_UNET if EMU else NORTH_UNET)
    model.net_3d_enabled = False
    grid = torch.tanh(torch.from_numpy(np_noise(17, (1, C, R, R, R))))
    cams = _cams(8)
    preds = model(camera=cams[3].to(gu.DEV), evaluation_mode=EvaluationMode.EVALUATION, voxel_features=grid.to(gu.DEV))
    o, d, l = ro.make_rays(gu.cam_dict(cams, 3), rcfg)
    ref = ro.render_rays(grid, msd, o, d, l, rcfg)
    assert ref["rgb"].shape[0] == H * W
    margin = 0.3 * rcfg.sample_pdf_eps
    frag_pdf = (ref["pdf_denom"] - rcfg.sample_pdf_eps).abs().min(dim=1)[0] <= margin
    dens_last, _ = ro.implicit_function(grid, msd, o, d, l[:, -1:], rcfg)
    frag_far = dens_last.reshape(-1).abs() <= 1e-4
    fragile = frag_pdf | frag_far
    flat = lambda t, c: t.reshape(c, H * W).t().cpu()  # noqa: E731  (1,c,H,W) -> (rays,c)
    coarse = preds["rendered"].prev_stage
    cflat = lambda t: t.permute(0, 3, 1, 2).reshape(t.shape[3], H * W).t().cpu()  # noqa: E731
    errs = {
        "rgb": ((flat(preds["images_render"], 3) - ref["rgb"]).abs().max(dim=1)[0], 2e-4),
        "mask": ((flat(preds["masks_render"], 1) - ref["mask"]).abs().reshape(-1), 2e-4),
        "depth": ((flat(preds["depths_render"], 1) - ref["depth"]).abs().reshape(-1), 2e-4 * FAR),
        "rgb_c": ((cflat(coarse.features) - ref["rgb_c"]).abs().max(dim=1)[0], 2e-4),
        "mask_c": ((cflat(coarse.masks) - ref["mask_c"]).abs().reshape(-1), 2e-4),
        "depth_c": ((cflat(coarse.depths) - ref["depth_c"]).abs().reshape(-1), 2e-4 * FAR),
    }
    # (c): rays outside a tolerance that neither rule explains -> the oracle on a grid perturbed by 1e-6, on those rays only
    unexplained = torch.zeros(H * W, dtype=torch.bool)
    for k, (e, tol) in errs.items():
        unexplained |= (e >= tol) & ~fragile
    if unexplained.any():
        idx = torch.nonzero(unexplained).flatten()
        assert len(idx) <= 0.001 * H * W + 2, len(idx)
        pert = grid + 1e-6 * torch.from_numpy(np_noise(8, tuple(grid.shape)))
        u0 = torch.linspace(0.0, 1.0, ref["pdf_denom"].shape[1])[None].expand(len(idx), -1)
        controls = [ro.render_rays(pert, msd, o[idx], d[idx], l[idx], rcfg)]
        controls += [ro.render_rays(grid, msd, o[idx], d[idx], l[idx], rcfg, u_fine=(u0 + du).clamp(0.0, 1.0)) for du in (2e-6, -2e-6)]
        moved = {}
        for k in errs:
            per = [(c[k] - ref[k][idx]).abs() for c in controls]
            moved[k] = torch.stack([(m.max(dim=1)[0] if m.dim() > 1 and m.shape[1] > 1 else m.reshape(-1)) for m in per]).max(dim=0)[0]
        # third voter: the oracle's own arithmetic in float64 on exactly these rays.  Rule (c) alone is defined by the float32
        # oracle's instability and could hide a kernel that is wrong ONLY on ill-conditioned rays; against the float64 value
        # the float32 oracle and the kernel are two roundings of the same ill-conditioned number, so the kernel may sit no
        # further from it than a small multiple of what the float32 oracle (or its perturbed runs) sits from it.
        kvals = {"rgb": flat(preds["images_render"], 3), "mask": flat(preds["masks_render"], 1), "depth": flat(preds["depths_render"], 1),
                 "rgb_c": cflat(coarse.features), "mask_c": cflat(coarse.masks), "depth_c": cflat(coarse.depths)}
        torch.set_default_dtype(torch.float64)
        try:
            ref64 = ro.render_rays(grid.double(), {k: v.double() for k, v in msd.items()}, o[idx].double(), d[idx].double(),
                                   l[idx].double(), rcfg)
        finally:
            torch.set_default_dtype(torch.float32)
        red = lambda m: (m.max(dim=1)[0] if m.dim() > 1 and m.shape[1] > 1 else m.reshape(-1))  # noqa: E731
        for k, (e, tol) in errs.items():
            r64 = ref64[k].float()
            k_vs_64 = red((kvals[k][idx] - r64).abs())
            o_vs_64 = torch.stack([red((ref[k][idx] - r64).abs())] + [red((c[k] - r64).abs()) for c in controls]).max(dim=0)[0]
            bound = 8.0 * o_vs_64 + 8.0 * tol
            worst = int(torch.argmax(k_vs_64 - bound))
            assert (k_vs_64 <= bound).all(), (k, "kernel further from the float64 oracle than the float32 oracle's own spread",
                                              int(idx[worst]), float(k_vs_64[worst]), float(o_vs_64[worst]))
        for j, pix in enumerate(idx.tolist()):
            sens = any(float(moved[k][j]) >= 0.25 * float(errs[k][0][pix]) for k in errs if float(errs[k][0][pix]) >= errs[k][1])
            print(f"  ray {pix}: " + ", ".join(f"{k} kernel-vs-oracle {float(errs[k][0][pix]):.2e} / oracle-vs-perturbed-oracle {float(moved[k][j]):.2e}"
                                              for k in ("rgb", "mask", "depth")) + ("  -> the reference itself is unstable here" if sens else ""))
            if sens:
                fragile[pix] = True
    unexplained &= ~fragile
    assert int(unexplained.sum()) <= 2, int(unexplained.sum())
    summary = []
    for k, (e, tol) in errs.items():
        bad = e >= tol
        summary.append(f"{k} {int(bad.sum())} outside (max {float(e.max()):.2e}, max non-fragile {float(e[~fragile].max()):.2e})")
        assert not (bad & ~fragile & ~unexplained).any(), (k, int((bad & ~fragile).sum()), float(e[~fragile].max()))
        loose = 5e-3 * (FAR if "depth" in k else 1.0)
        assert not ((e >= loose) & ~frag_far).any(), (k, "a ray away from the far-sample sign switch misses the loose bound")
    print(f"full frame: {H * W} rays, fragile {int(frag_pdf.sum())} (sample_pdf switch) + {int(frag_far.sum())} (far-sample sign); "
          + "; ".join(summary))
    assert int(fragile.sum()) <= 0.001 * H * W + 2
    # the frame is not trivial: opaque and semi-transparent rays both occur
    m = ref["mask"].reshape(-1)
    assert float(m.max()) > 0.95 and int(((m > 0.2) & (m < 0.8)).sum()) > 0.001 * H * W


@pytest.mark.parametrize("T,max_iter", [(1000, 4), (250, 8)])
def test_north_star_sampler_chain_vs_oracle(gu, T, max_iter):
    """configs[1] as a CHAIN: the first ``max_iter`` DDPM steps (``p_sample_loop_progressive``, gaussian_diffusion.py:568-643)
    of the 64^3 x 32 net with injected noise against ``DiffusionOracle`` driving the pinned UNet oracle - the sizes where
    conv_wino2_kernel, split-K and the 4^3 weight-streaming level actually run (the recorded reference trajectories and
    the wide-net chains are 8^3 grids).  Default planner; every step's sample and pred_xstart within 5e-3."""
    from oracle.common import NORTH_CFG, TINY_CFG
    cfg = TINY_CFG if EMU else NORTH_CFG
    net, sd = gu.make_unet(cfg)
    with np.errstate(divide="ignore"):
        diff = hda.ImplicitronGaussianDiffusion(num_steps=T)
    shape = (1, cfg.in_channels, cfg.image_size, cfg.image_size, cfg.image_size)
    ns_dev = lambda t, shp, device=None: torch.from_numpy(np_noise(77 * 100003 + t, tuple(shp))).to(gu.DEV)  # noqa: E731
    ns_cpu = lambda t, shp, device=None: torch.from_numpy(np_noise(77 * 100003 + t, tuple(shp)))  # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        steps = [{k: v.cpu() for k, v in s.items() if k in ("sample", "pred_xstart")}
                 for s in diff.p_sample_loop_progressive(net, shape, clip_denoised=True, noise_sampler=ns_dev, max_iter=max_iter)]
        ref = list(do.DiffusionOracle(T).p_sample_loop_progressive(lambda x, t: uo.unet_forward(sd, cfg, x, t), shape,
                                                                   ns_cpu, True, max_iter))
    assert len(steps) == len(ref) == max_iter
    worst = 0.0
    for i, (s, r) in enumerate(zip(steps, ref)):
        for k in ("sample", "pred_xstart"):
            e = gu.rel_err(s[k], r[k])
            worst = max(worst, e)
            assert e < 5e-3, (k, i, e)
    print(f"north-star chain T={T}, {max_iter} steps: worst relative error {worst:.2e}")
    if not EMU:  # the planner's own choice: the kernel the bench line credits is the one this chain ran on
        ops = [o_ for o_ in net.time_ops(1, 1, gu.DEV) if o_["op"] == "conv"]
        w3 = [o_ for o_ in ops if o_["kernel"] == "conv_wino3_kernel"]
        # every stride-1 3x3x3 convolution of the 64^3 level with >= 64 output channels in the F(2x2x2, 3x3x3) form
        wide64 = [o_ for o_ in ops if o_["out_dim"] == 64 and o_["ksz"] == 3 and o_["stride"] == 1 and o_["cout"] >= 64]
        assert len(wide64) >= 12 and all(o_["kernel"] == "conv_wino3_kernel" for o_ in wide64), \
            [(o_["kernel"], o_["cin"], o_["cout"]) for o_ in wide64]
        # the up path's 1x1x1 skip connections of the 64^3 level: their own streaming launches (the second convolution's residual)
        skips64 = [o_ for o_ in ops if o_["out_dim"] == 64 and o_["ksz"] == 1]
        assert len(skips64) == 3 and all(o_["kernel"] == "conv1x1_stream_kernel" for o_ in skips64), skips64
        assert any(o_["fused_skip"] for o_ in w3)  # (fused into the convolution on the smaller levels)
        assert any(o_["nsplit"] > 1 for o_ in w3), "no split-K launch of conv_wino3_kernel in the default plan"
        assert any(o_["nsplit"] > 1 and o_["kernel"] == "conv_small_kernel" for o_ in ops)  # the weight-streaming 4^3 level


def test_north_star_perf_mode_chain_vs_oracle(gu):
    """The path bench.py times, at the size it times it: the sampler's PERF MODE (``device_noise_seed``: the chain stays in
    channels-last tensors, ``holo_unet_forward_cl`` + ``holo_ddpm_step_philox``, gaussian_diffusion.py:568-643) on the 64^3 x 32
    net.  (1) ``forward_channels_last(x_cl)`` is ``forward(x)`` BIT for bit at this size - conv_wino3_kernel at 64^3 and
    conv1x1_stream_kernel read / write the caller's tensors there; (2) four steps of the chain with the per-step noise read
    back from the step kernel (``want_noise``) against ``DiffusionOracle`` driving the pinned UNet oracle with exactly that
    noise, every step's sample and pred_xstart within 5e-3; (3) ``p_sample_loop_progressive`` in the perf mode - the product
    loop - yields the same samples as the hand-written loop, bit for bit, and so does the NCDHW fallback route with the same
    seed (the Philox draw is keyed on the logical element, not on the layout)."""
    from oracle.common import NORTH_CFG, TINY_CFG
    cfg = TINY_CFG if EMU else NORTH_CFG
    T, n_steps, seed, stream = 1000, 4, 20240917, 5
    net, sd = gu.make_unet(cfg)
    shape = (1, cfg.in_channels, cfg.image_size, cfg.image_size, cfg.image_size)
    x_T = torch.from_numpy(np_noise(4242, shape))
    # (1) the two forward entries
    t0 = torch.tensor([T - 1], device=gu.DEV)
    x_dev = x_T.to(gu.DEV)
    x_cl = x_dev.permute(0, 2, 3, 4, 1).contiguous()
    with torch.no_grad():
        y = net(x_dev, t0)
        y_cl = net.forward_channels_last(x_cl, t0)
    assert torch.equal(y_cl.permute(0, 4, 1, 2, 3), y)
    # (2) the hand-written perf chain (bench.py's one_step), noise read back
    with np.errstate(divide="ignore"):
        diff = hda.ImplicitronGaussianDiffusion(num_steps=T, device_noise_seed=seed, device_noise_stream=stream)
    indices = diff._indices(n_steps)
    steps, noises = [], {}
    img_cl = x_cl
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for idx in indices:
            t = torch.tensor([idx], device=gu.DEV)
            out_cl = net.forward_channels_last(img_cl, t)
            s_cl, p_cl, e_cl = diff._step_device_noise(img_cl, t, out_cl, idx, True, want_pred=True, want_noise=True,
                                                       channels_last=True)
            steps.append({"sample": s_cl.permute(0, 4, 1, 2, 3).contiguous().cpu(), "pred_xstart": p_cl.permute(0, 4, 1, 2, 3).contiguous().cpu()})
            noises[idx] = e_cl.permute(0, 4, 1, 2, 3).contiguous().cpu()
            img_cl = s_cl
        ns_cpu = lambda t, shp, device=None: (x_T if t == T else noises[t])  # noqa: E731  (t == T: the oracle's x_T draw)
        ref = list(do.DiffusionOracle(T).p_sample_loop_progressive(lambda x, t: uo.unet_forward(sd, cfg, x, t), shape,
                                                                   ns_cpu, True, n_steps))
    assert len(ref) == n_steps
    worst = 0.0
    for i, (s, r) in enumerate(zip(steps, ref)):
        for k in ("sample", "pred_xstart"):
            e = gu.rel_err(s[k], r[k])
            worst = max(worst, e)
            assert e < 5e-3, (k, i, e)
    # the in-kernel draws are standard normals (the oracle chain above would not notice a wrong variance on its own)
    e_all = torch.cat([v.reshape(-1) for v in noises.values()]).double()
    se = 5.0 / e_all.numel() ** 0.5  # five standard errors of the mean (the variance's is sqrt(2) times that)
    assert abs(float(e_all.mean())) < se and abs(float(e_all.var()) - 1.0) < 1.5 * se
    # (3) the product loop, and the NCDHW fallback route (a wrapped model has no forward_channels_last)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prod = [s["sample"].cpu() for s in diff.p_sample_loop_progressive(net, shape, noise=x_dev, max_iter=n_steps)]
        fallback = [s["sample"].cpu() for s in diff.p_sample_loop_progressive(lambda a, b: net(a, b), shape, noise=x_dev,
                                                                             max_iter=n_steps, device=gu.DEV)]
    for i in range(n_steps):
        assert torch.equal(prod[i], steps[i]["sample"]), i
        assert torch.equal(fallback[i], steps[i]["sample"]), i
    print(f"north-star perf-mode chain (channels-last, in-kernel Philox noise), {n_steps} steps: worst relative error {worst:.2e}")


def test_config0_plumbing_frame_vs_oracle(gu):
    """configs[0] (unet_with_no_diffusion.yaml: diffusion disabled, the path is tanh(net_3d(vf, 0)) + render): 32^3 x 16
    grid, model_channels 64, one camera at 128x128, n_pts_per_ray_fine_evaluation = 16 (configs/...yaml:155-156): the
    whole frame against the oracle pipeline."""
    if EMU:
        R, C, H, W, unet = 8, 16, 16, 16, TINY_UNET
    else:
        R, C, H, W, unet = 32, 16, 128, 128, NORTH_UNET
    model, ucfg, usd, rcfg, msd = gu.make_model(R, C, H, W, unet, n_fine=16)
    vf = torch.tanh(torch.from_numpy(np_noise(3, (1, C, R, R, R))))
    cams = _cams(4)
    preds = model(camera=cams[2].to(gu.DEV), evaluation_mode=EvaluationMode.EVALUATION, voxel_features=vf.to(gu.DEV))
    grid_ref = torch.tanh(uo.unet_forward(usd, ucfg, vf, torch.zeros(1, dtype=torch.long)))
    ref = ro.render(grid_ref, msd, gu.cam_dict(cams, 2), rcfg)
    assert preds["images_render"].shape == (1, 3, H, W)
    for k, tol in (("images_render", 1e-3), ("masks_render", 1e-3), ("depths_render", 1e-3 * FAR)):
        assert (preds[k].cpu() - ref[k]).abs().max().item() < tol, k


def test_teddybear_30_view_turntable_in_one_call(gu):
    """configs[3]: a 30-camera turntable of one grid in ONE render_views call (more cameras than one launch group
    takes, ragged last group) equals 30 single-camera forward() calls bit for bit; the LAST camera (in the ragged
    group) is also checked against the oracle on a ray subset."""
    R, C, H, W, n, nrand = (8, 32, 16, 24, 11, 96) if EMU else (64, 32, 400, 400, 30, 1024)
    model, _, _, rcfg, msd = gu.make_model(R, C, H, W, TINY_UNET if EMU else NORTH_UNET)
    model.net_3d_enabled = False
    grid_c = torch.tanh(torch.from_numpy(np_noise(17, (1, C, R, R, R))))
    grid = grid_c.to(gu.DEV)
    cams = _cams(n)
    dcams = cams.to(gu.DEV)
    allv = model.render_views(grid, dcams)
    assert allv["images_render"].shape == (n, 3, H, W)
    for i in range(n):
        one = model(camera=dcams[i], voxel_features=grid)
        for k in ("images_render", "depths_render", "masks_render"):
            assert torch.equal(one[k][0], allv[k][i]), (k, i)
    last = n - 1
    idx = _border_and_random_rays(H, W, nrand, 9)
    o, d, l = ro.make_rays(gu.cam_dict(cams, last), rcfg)
    ref = ro.render_rays(grid_c, msd, o[idx], d[idx], l[idx], rcfg)
    _check_rays({k: v[last:last + 1] for k, v in allv.items()}, ref, idx, H, W)


def test_progressive_denoise_render_every_step_vs_oracle(gu):
    """flyaround.py:236-245 with progressive_sampling_steps_per_render = 1: camera n renders the grid after n + 1
    denoising steps (clipped to [-1, 1], then tanh(net_3d(., 0))).  16^3 grid, 3 steps, recorded noise: every frame
    against the oracle pipeline (DDPM steps -> clip -> UNet at t=0 -> tanh -> render)."""
    R, C, H, W, steps = (8, 32, 10, 12, 3) if EMU else (16, 32, 24, 32, 3)
    model, ucfg, usd, rcfg, msd = gu.make_model(R, C, H, W, TINY_UNET, diffusion_args=dict(num_steps=1000))
    ns = lambda t, shp, dev=None: torch.from_numpy(np_noise(31 * 100003 + t, tuple(shp)))  # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fly = render_flyaround(model, n_flyaround_poses=steps, device=gu.DEV, progressive_sampling_steps_per_render=1,
                               sampler_kwargs=dict(max_iter=steps, noise_sampler=lambda t, s, d: ns(t, s).to(gu.DEV)))
        orc = do.DiffusionOracle(1000)
        unet = lambda x, t: uo.unet_forward(usd, ucfg, x, t)  # noqa: E731
        chain = list(orc.p_sample_loop_progressive(unet, (1, C, R, R, R), ns, True, steps))
    assert fly["images_render"].shape == (steps, 3, H, W)
    cams = _cams(steps)
    for n in range(steps):
        vf = chain[n]["sample"].clamp(-1, 1)
        grid_ref = torch.tanh(unet(vf, torch.zeros(1, dtype=torch.long)))
        ref = ro.render(grid_ref, msd, gu.cam_dict(cams, n), rcfg)
        # the chain's own error (5e-3 of the dynamic range per step, test_gpu_diffusion) feeds the frame
        assert (fly["images_render"][n:n + 1].cpu() - ref["images_render"]).abs().max().item() < 5e-3, n
        assert (fly["masks_render"][n:n + 1].cpu() - ref["masks_render"]).abs().max().item() < 5e-3, n
    assert gu.rel_err(fly["voxel_features"], chain[-1]["sample"].clamp(-1, 1)) < 5e-3


def test_turntable_per_denoise_step_driver(gu):
    """configs[3] stress form: ALL cameras of the turntable rendered after every denoising
    step, one batched render call per step; frames of step k equal render_views of the k-th progressive grid."""
    from holo_diffusion_amd.generate import render_progressive_turntable
    R, C, H, W, n_views, steps = 8, 32, 8, 12, 5, 3
    model, *_ = gu.make_model(R, C, H, W, TINY_UNET, diffusion_args=dict(num_steps=1000))
    ns = lambda t, shp, dev=None: torch.from_numpy(np_noise(41 * 100003 + t, tuple(shp))).to(gu.DEV)  # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        per_step = list(render_progressive_turntable(model, n_views=n_views, steps_per_render=1, device=gu.DEV,
                                                     sampler_kwargs=dict(max_iter=steps, noise_sampler=ns)))
        grids = list(model.sample_random_voxel_features_progressive(max_iter=steps, noise_sampler=ns))
    assert len(per_step) == steps
    cams = _cams(n_views).to(gu.DEV)
    for k in range(steps):
        assert per_step[k]["images_render"].shape == (n_views, 3, H, W)
        assert torch.equal(per_step[k]["voxel_features"], grids[k])
        again = model.render_views(grids[k], cams)
        assert torch.equal(per_step[k]["images_render"], again["images_render"])


# ---------------------------------------------------------------------------------------------------------------
# known-answer / edge cases of the (unpinned) renderer half
# ---------------------------------------------------------------------------------------------------------------
def test_non_square_image_and_principal_point(gu):
    """Non-square frames (the longer side spans +-(long/short) in NDC) with a non-zero principal point and fx != fy,
    both portrait and landscape, against the oracle."""
    for (H, W) in ((12, 20), (22, 10)):
        model, _, _, rcfg, msd = gu.make_model(8, 32, H, W, TINY_UNET)
        model.net_3d_enabled = False
        grid = torch.tanh(torch.from_numpy(np_noise(13, (1, 32, 8, 8, 8))))
        base = _cams(5)
        cams = hda.PerspectiveCameras(R=base.R, T=base.T, focal_length=torch.tensor([[3.0, 3.6]]).expand(5, 2),
                                      principal_point=torch.tensor([[0.12, -0.07]]).expand(5, 2))
        preds = model(camera=cams[3].to(gu.DEV), voxel_features=grid.to(gu.DEV))
        ref = ro.render(grid, msd, gu.cam_dict(cams, 3), rcfg)
        for k, tol in (("images_render", 2e-4), ("masks_render", 2e-4), ("depths_render", 2e-4 * FAR)):
            assert (preds[k].cpu() - ref[k]).abs().max().item() < tol, (k, H, W)


def test_camera_inside_scene_extent_clamps_depth_bounds(gu):
    """AdaptiveRaySampler: dist = max(|C - centre|, scene_extent + 1e-3) — a camera at radius 2.5 < scene_extent = 4
    starts sampling at depth 1e-3 instead of behind the camera."""
    H, W = 14, 14
    model, _, _, rcfg, msd = gu.make_model(8, 32, H, W, TINY_UNET)
    model.net_3d_enabled = False
    grid = torch.tanh(torch.from_numpy(np_noise(19, (1, 32, 8, 8, 8))))
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 4, -0.4, 2.5, (0.0, -1.0, 0.0), 1.5)
    zmin, zmax = ro.depth_bounds(cams.R[1], cams.T[1], rcfg)
    assert abs(zmin - 1e-3) < 1e-5 and abs(zmax - 8.001) < 1e-4
    preds = model(camera=cams[1].to(gu.DEV), voxel_features=grid.to(gu.DEV))
    ref = ro.render(grid, msd, gu.cam_dict(cams, 1), rcfg)
    for k, tol in (("images_render", 2e-4), ("masks_render", 2e-4), ("depths_render", 2e-4 * FAR)):
        assert (preds[k].cpu() - ref[k]).abs().max().item() < tol, k


def test_inverse_cdf_sample_on_a_cdf_knot(gu):
    """Zero density everywhere: the coarse weights are exactly 0, the pdf is uniform over the 62 inner bins and the
    reference's cdf (torch CPU cumsum) is exactly 0.5 at its 32nd entry; with n_fine = 3 the middle sample u = 0.5 falls
    exactly on that knot (searchsorted right=True), u = 0 and u = 1 on the end knots (the den < eps -> 1 branch).  The
    frame must be pure background with zero depth and mask - no NaN from the degenerate interval - like the oracle's."""
    H, W = 8, 8
    model, _, _, rcfg, msd = gu.make_model(8, 32, H, W, TINY_UNET, n_fine=3)
    model.net_3d_enabled = False
    sd = model.state_dict()
    for i in range(2):
        pre = f"_implicit_functions.{i}._fn.render_mlp."
        sd[pre + "_density_net.mlp.3.0.weight"][-1] = 0.0
        sd[pre + "_density_net.mlp.3.0.bias"][-1] = -1.0
    model.load_state_dict(sd)
    msd = {k[len("_implicit_functions.0._fn.render_mlp."):]: v.cpu() for k, v in sd.items()
           if k.startswith("_implicit_functions.0._fn.render_mlp.")}
    grid = torch.tanh(torch.from_numpy(np_noise(29, (1, 32, 8, 8, 8))))
    cams = _cams(3)
    w = torch.full((62,), 1e-5)
    cdf = torch.cumsum(w / w.sum(), 0)
    assert cdf[30].item() == 0.5 and cdf[61].item() == 1.0  # the knots u = 0.5 and u = 1 are hit exactly
    zf = ro.refine_lengths(torch.linspace(6.0, 14.0, 64)[None], torch.zeros(1, 64), rcfg)
    assert zf.shape == (1, 67) and torch.isfinite(zf).all()
    preds = model(camera=cams[1].to(gu.DEV), voxel_features=grid.to(gu.DEV))
    ref = ro.render(grid, msd, gu.cam_dict(cams, 1), rcfg)
    assert torch.all(preds["masks_render"] == 0) and torch.all(ref["masks_render"] == 0)
    torch.testing.assert_close(preds["images_render"].cpu(), ref["images_render"], rtol=0, atol=1e-6)
    assert torch.isfinite(preds["depths_render"]).all() and preds["depths_render"].abs().max() == 0
