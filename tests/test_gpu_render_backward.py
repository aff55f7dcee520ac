"""GPU parity of the renderer's backward pass (second half): holo_render_rays_backward - gradients of both
passes' rgb / depth / mask w.r.t. the voxel grid (grid_sample's scatter-add) and every RenderMLP parameter - against
autograd through the oracle's training-mode renderer with the same injected random streams (the importance sampling is
detached on both sides, as PyTorch3D's RayPointRefiner samples under torch.no_grad())."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import holo_diffusion_amd as hda  # noqa: E402
from holo_diffusion_amd.render import EvaluationMode  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from oracle.common import np_noise  # noqa: E402

EMU = os.environ.get("HOLO_TEST_EMU") == "1"
TINY_UNET = dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2))


@pytest.fixture(scope="module")
def gu():
    import tests.gpu_utils as g
    return g


def _streams(n_cam, n_rays, P, Pf, seed):
    u = lambda s, shp: torch.from_numpy(np_noise(s, shp)).mul(0.5).erf().add(1).mul(0.5).clamp(0, 0.999999)  # noqa: E731  U[0,1)
    return {"u_coarse": u(seed, (n_cam, n_rays, P)), "u_fine": u(seed + 1, (n_cam, n_rays, Pf)),
            "noise_coarse": torch.from_numpy(np_noise(seed + 2, (n_cam, n_rays, P))),
            "noise_fine": torch.from_numpy(np_noise(seed + 3, (n_cam, n_rays, P + Pf)))}


def _rel(got, want, floor):
    return ((got - want).abs().max() / max(float(want.abs().max()), floor)).item()


CASES = [(12, 10, 16, 8, 3, 13, "all")] if EMU else [(12, 10, 16, 8, 3, 13, "all"), (24, 20, 32, 16, 3, 37, "all"),
                                                     (12, 10, 64, 8, 2, 9, "all"), (12, 10, 16, 8, 33, 5, "all"),
                                                     (64, 64, 32, 32, 2, 300, "all"), (24, 20, 16, 16, 2, 41, "fine_only"),
                                                     (16, 100, 16, 16, 2, 29, "no_noise"), (64, 64, 32, 32, 4, 700, "chunks")]


@pytest.mark.parametrize("P,Pf,C,R,n_cam,n_rays,which", CASES)
def test_render_rays_backward_vs_oracle_autograd(gu, P, Pf, C, R, n_cam, n_rays, which):
    """Losses on all six outputs (random cotangents), stratified depths / importance samples and density noise of std 1
    injected; `chunks`: more points than one 65 536-point chunk, `fine_only`: no gradient on the coarse outputs (NULL
    pointers), `no_noise`: deterministic depths, no density noise (16 + 100 samples: a 116-point merged list)."""
    model, _, _, _, msd = gu.make_model(R, C, 16, 16, TINY_UNET, n_fine=64)
    model.raysampler.n_pts_per_ray_training = P
    model.renderer.n_pts_per_ray_fine_training = Pf
    rcfg = ro.RenderCfg(resol=R, feature_size=C, image_height=16, image_width=16, n_pts_coarse=P, n_pts_fine=Pf)
    grid = torch.tanh(torch.from_numpy(np_noise(7, (1, C, R, R, R))))
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, n_cam, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    xys = (torch.from_numpy(np_noise(11, (n_cam, n_rays, 2))).clamp(-2, 2) * 0.45).contiguous()
    rs = _streams(n_cam, n_rays, P, Pf, 500 + P)
    std = 1.0
    if which == "no_noise":
        model.raysampler.stratified_point_sampling_training = False
        model.renderer.stratified_sampling_coarse_training = False
        model.renderer.density_noise_std_train = std = 0.0
        rs = {}
    keys = {"features": ("rgb", 3), "depths": ("depth", 1), "masks": ("mask", 1), "features_coarse": ("rgb_c", 3),
            "depths_coarse": ("depth_c", 1), "masks_coarse": ("mask_c", 1)}
    if which == "fine_only":
        keys = {k: v for k, v in keys.items() if not k.endswith("_coarse")}
    cot = {k: torch.from_numpy(np_noise(900 + i, (n_cam, n_rays, 1, c))) * (0.05 if "depth" in k else 1.0)
           for i, (k, (_, c)) in enumerate(keys.items())}
    for fn in model._implicit_functions:
        fn.bind_args(voxel_grid_features=grid.to(gu.DEV))
    bundle = model.raysampler(cams.to(gu.DEV), EvaluationMode.TRAINING, xys=xys.to(gu.DEV))
    dev_rs = {k: v.to(gu.DEV) for k, v in rs.items()}
    ggrid, pg, zm, zf = model.renderer.backward_training(bundle, list(model._implicit_functions), dev_rs,
                                                         {k: v.to(gu.DEV) for k, v in cot.items()}, return_merged=True)
    assert ggrid.shape == grid.shape and torch.isfinite(ggrid).all()
    zm, zf = zm.cpu(), zf.cpu()
    assert zm.shape == (n_cam, n_rays, P + Pf) and int(zf.sum()) == n_cam * n_rays * Pf and (zm.diff(dim=-1) >= 0).all()

    def oracle(fixed, f64=False):
        """autograd per camera, summed; fixed: the fine pass uses the kernel's own merged depth list; f64: the oracle's
        arithmetic in float64 (d colour / d density is a small difference of O(1) terms: the float32 oracle itself sits
        ~5e-4 from the float64 one on the grid at the large sizes)"""
        cv = (lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t) if f64 else (lambda t: t)
        wg, wp, dz = torch.zeros_like(cv(grid)), None, []
        rays = [ro.rays_from_xys(gu.cam_dict(cams, i), xys[i], rcfg) for i in range(n_cam)]  # (float32 on both sides)
        if f64:
            torch.set_default_dtype(torch.float64)
        try:
            for i in range(n_cam):
                o, d, l = rays[i]
                og = {keys[k][0]: cv(cot[k][i]) for k in keys}
                g, p, out = ro.render_rays_grad(cv(grid), {k: cv(v) for k, v in msd.items()}, cv(o), cv(d), cv(l), rcfg, og,
                                                u_coarse=cv(rs["u_coarse"][i]) if rs else None,
                                                u_fine=cv(rs["u_fine"][i]) if rs else None,
                                                noise_coarse=cv(rs["noise_coarse"][i]) if rs else None,
                                                noise_fine=cv(rs["noise_fine"][i]) if rs else None, noise_std=std,
                                                fine_lengths=cv(zm[i]) if fixed else None)
                wg += g
                wp = p if wp is None else {k: wp[k] + p[k] for k in p}
                dz.append((out["fine_lengths"] - zm[i]).abs())
        finally:
            torch.set_default_dtype(torch.float32)
        return wg.float(), {k: v.float() for k, v in wp.items()}, torch.stack(dz).float()

    def worst_of(want_grid, want_p, norm="max"):
        """worst error over the grid and the parameter tensors, relative to each tensor's scale: max norm or L2"""
        if norm == "l2":
            worst = ("grid", float((ggrid.cpu() - want_grid).norm() / want_grid.norm()))
        else:
            worst = ("grid", _rel(ggrid.cpu(), want_grid, 1e-6))
        scale = sorted(float(v.abs().max()) for v in want_p.values())[len(want_p) // 2]
        assert set(pg) == set(want_p)
        for k in want_p:
            if norm == "l2":
                e = float((pg[k].cpu() - want_p[k]).norm() / max(float(want_p[k].norm()), 1e-2 * scale * want_p[k].numel() ** 0.5))
            else:
                e = _rel(pg[k].cpu(), want_p[k], 1e-2 * scale)
            if e > worst[1]:
                worst = (k, e)
        return worst

    # (1) the backward arithmetic, with the sample placement of the forward kernel held fixed on both sides, against the
    # float64 oracle.  LeakyReLU / ReLU kinks: a pre-activation within float32 rounding of zero takes the other branch on
    # one side, which changes ONE hidden unit of ONE sample by its whole slope difference - sparse, O(1e-3) of a tensor's
    # max per event (measured at 4 x 700 rays, 92 M pre-activations: 10 of 32 768 voxel positions and 3 of 257 rows of
    # d W3 above 1e-3, the float32 torch oracle against the float64 one shows the same events: 4 positions, 1 row, same
    # maximum 4.0e-3 / 2.5e-3).  The small cases see no such event and are held to 1e-3 in the max norm; the large ones to
    # 2e-3 in L2 and 1e-2 in the max norm.
    want_grid, want_p, _ = oracle(True, f64=True)
    assert float(want_grid.abs().max()) > 1e-3
    worst = worst_of(want_grid, want_p)
    worst_l2 = worst_of(want_grid, want_p, "l2")
    # (2) the fully independent oracle (its own refiner): the importance samples agree except where the inverse cdf sits on
    # its `denominator < eps` switch (the forward tests' depth tolerance); a moved sample moves one sample's scatter
    free_grid, free_p, dz = oracle(False)
    moved = float((dz > 1e-3).float().mean())
    l2 = float((ggrid.cpu() - free_grid).norm() / free_grid.norm())
    worst_free = worst_of(free_grid, free_p)
    print(f"\nrender backward P={P} Pf={Pf} C={C} R={R} rays={n_cam}x{n_rays} [{which}]: worst relative gradient error "
          f"{worst[1]:.2e} ({worst[0]}) max norm, {worst_l2[1]:.2e} ({worst_l2[0]}) L2, vs the float64 oracle at the kernel's "
          f"sample placement; independent float32 oracle: {100 * moved:.3f} % of the samples moved by > 1e-3, grid gradient L2 "
          f"error {l2:.2e}, worst max-norm error {worst_free[1]:.2e} ({worst_free[0]})")
    many = n_cam * n_rays * (P + Pf) > 50000
    if not many:
        assert worst[1] < 1e-3 and worst_l2[1] < 1e-3, (worst, worst_l2)
    else:
        # kink events are whole-sample events on the grid (the eight corners of one cell; the density's ReLU flips a
        # sample's entire density gradient: 1.8e-2 of the grid's max at 4 x 700 rays, eight positions forming one cell):
        # up to 64 voxel positions (eight events) are set aside, the rest of the grid is held to 2e-3; parameter tensors sum
        # over all samples and are held to 1e-2 / 2e-3 (max / L2) whole
        err = (ggrid.cpu() - want_grid).abs().amax(dim=1).flatten() / float(want_grid.abs().max())
        rest = float(torch.topk(err, 65).values[64])
        n_events = int((err > 2e-3).sum())
        wp_max = worst_of(torch.zeros_like(want_grid) + ggrid.cpu(), want_p)  # parameters only (grid error zeroed)
        print(f"   grid: {n_events} voxel positions above 2e-3 (set aside: 64), the rest within {rest:.2e}; parameters {wp_max[1]:.2e} ({wp_max[0]})")
        assert rest < 2e-3 and worst_l2[1] < 3e-3 and wp_max[1] < 1e-2, (rest, worst_l2, wp_max)
    assert moved < 5e-3 and l2 < 2e-2 and worst_free[1] < 5e-2, (moved, l2, worst_free)


def test_render_backward_needs_the_forward_draws(gu):
    """The backward pass must see the draws of the forward pass: a missing stream is an error, not a fresh draw."""
    model, _, _, _, _ = gu.make_model(8, 16, 16, 16, TINY_UNET, n_fine=64)
    model.raysampler.n_pts_per_ray_training = 12
    model.renderer.n_pts_per_ray_fine_training = 10
    grid = torch.zeros(1, 16, 8, 8, 8, device=gu.DEV)
    for fn in model._implicit_functions:
        fn.bind_args(voxel_grid_features=grid)
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 2, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    bundle = model.raysampler(cams.to(gu.DEV), EvaluationMode.TRAINING, xys=torch.zeros(2, 5, 2, device=gu.DEV))
    with pytest.raises(hda._lib.HoloError):
        model.renderer.backward_training(bundle, list(model._implicit_functions), {}, {"features": torch.zeros(2, 5, 1, 3)})


DET_CASES = [(12, 10, 16, 8, 2, 13)] if EMU else [(12, 10, 16, 8, 2, 13), (24, 20, 32, 16, 3, 37), (64, 64, 32, 32, 4, 700)]


@pytest.mark.parametrize("P,Pf,C,R,n_cam,n_rays", DET_CASES)
def test_deterministic_scatter_mode_of_the_render_backward(gu, P, Pf, C, R, n_cam, n_rays):
    """holo_ctx_set_deterministic: the grid gradient's trilinear scatter-add as 64-bit fixed-point sums.  Two calls are
    bit-identical (the last case spans several 65 536-point chunks, each with its own binary point), the result sits within
    fp32 summation noise of the default mode's (hardware fp32 atomics, order-dependent), whose own run-to-run spread is
    printed next to it, and the parameter gradients - fixed-order sums in both modes - do not change at all."""
    from holo_diffusion_amd import runtime
    model, _, _, _, _ = gu.make_model(R, C, 16, 16, TINY_UNET, n_fine=64)
    model.raysampler.n_pts_per_ray_training = P
    model.renderer.n_pts_per_ray_fine_training = Pf
    grid = torch.tanh(torch.from_numpy(np_noise(7, (1, C, R, R, R))))
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, n_cam, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    xys = (torch.from_numpy(np_noise(11, (n_cam, n_rays, 2))).clamp(-2, 2) * 0.45).contiguous()
    rs = {k: v.to(gu.DEV) for k, v in _streams(n_cam, n_rays, P, Pf, 500 + P).items()}
    cot = {k: (torch.from_numpy(np_noise(900 + i, (n_cam, n_rays, 1, c))) * (0.05 if "depth" in k else 1.0)).to(gu.DEV)
           for i, (k, c) in enumerate((("features", 3), ("depths", 1), ("masks", 1), ("features_coarse", 3)))}
    for fn in model._implicit_functions:
        fn.bind_args(voxel_grid_features=grid.to(gu.DEV))
    bundle = model.raysampler(cams.to(gu.DEV), EvaluationMode.TRAINING, xys=xys.to(gu.DEV))

    def run():
        g, pg = model.renderer.backward_training(bundle, list(model._implicit_functions), rs, cot)
        return g.clone(), {k: v.clone() for k, v in pg.items()}

    with runtime.deterministic(True, gu.DEV):
        g1, p1 = run()
        g2, p2 = run()
    with runtime.deterministic(False, gu.DEV):
        a1, pa = run()
        a2, _ = run()
    scale = float(g1.abs().max())
    assert scale > 1e-3 and torch.isfinite(g1).all()
    assert torch.equal(g1, g2), float((g1 - g2).abs().max()) / scale
    for k in p1:
        assert torch.equal(p1[k], p2[k]) and torch.equal(p1[k], pa[k]), k
    d_modes, d_runs = float((g1 - a1).abs().max()) / scale, float((a1 - a2).abs().max()) / scale
    print(f"\ndeterministic scatter P={P} Pf={Pf} C={C} R={R} rays={n_cam}x{n_rays}: fixed-point vs atomics {d_modes:.2e}, "
          f"atomics run to run {d_runs:.2e} (of the gradient's max)")
    assert d_modes < 2e-5, d_modes


@pytest.mark.skipif(EMU, reason="the UNet legs are too slow for the host emulation")
@pytest.mark.parametrize("bootstrap", [False, True])
def test_training_backward_chain_vs_oracle_autograd(gu, bootstrap):
    """HoloDiffusionModel.training_backward: renderer backward -> clamp of pred_xstart -> denoiser backward -> q_sample
    (-> the bootstrap round's clamp and denoiser once more), against autograd through the oracle pipeline (diffusion
    oracle + UNet oracle + training-mode render oracle) with every draw injected: gradients of the UNet parameters (the
    two rounds accumulate), the RenderMLP parameters and the clean grid."""
    from oracle import diffusion_oracle as do
    from oracle import unet_oracle as uo
    R, C, P, Pf, n_rays = 8, 16, 16, 16, 33
    model, ucfg, usd, _, msd = gu.make_model(R, C, 16, 16, TINY_UNET, n_fine=64)
    model.n_train_target_views = 2
    model.raysampler.n_pts_per_ray_training = P
    model.renderer.n_pts_per_ray_fine_training = Pf
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 4, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    vf = torch.tanh(torch.from_numpy(np_noise(3, (1, C, R, R, R))))
    xys = (torch.from_numpy(np_noise(12, (2, n_rays, 2))).clamp(-2, 2) * 0.45).contiguous()
    rs = _streams(2, n_rays, P, Pf, 900)
    rs.update({"xys": xys, "timesteps": torch.tensor([700]), "q_noise": torch.from_numpy(np_noise(31, tuple(vf.shape))),
               "bootstrap": bootstrap, "timesteps2": torch.tensor([150]), "q_noise2": torch.from_numpy(np_noise(32, tuple(vf.shape)))})
    keys = {"features": ("rgb", 3), "masks": ("mask", 1), "depths": ("depth", 1), "features_coarse": ("rgb_c", 3)}
    cot = {k: torch.from_numpy(np_noise(700 + i, (2, n_rays, 1, c))) * (0.05 if "depth" in k else 1.0)
           for i, (k, (_, c)) in enumerate(keys.items())}
    dev_rs = {k: (v.to(gu.DEV) if torch.is_tensor(v) else v) for k, v in rs.items()}
    out = model.training_backward(camera=cams.to(gu.DEV), voxel_features=vf.to(gu.DEV), rng_streams=dev_rs,
                                  grads={k: v.to(gu.DEV) for k, v in cot.items()})
    # ---- oracle: autograd through the whole training branch
    orc = do.DiffusionOracle(1000)
    rcfg = ro.RenderCfg(resol=R, feature_size=C, image_height=16, image_width=16, n_pts_coarse=P, n_pts_fine=Pf)
    with torch.enable_grad():
        x0 = vf.clone().requires_grad_(True)
        pu = {k: v.detach().clone().requires_grad_(True) for k, v in usd.items()}
        pm = {k: v.detach().clone().requires_grad_(True) for k, v in msd.items()}
        g = x0
        for tk, nk in (("timesteps", "q_noise"), ("timesteps2", "q_noise2"))[: 2 if bootstrap else 1]:
            g = uo.unet_forward.__wrapped__(pu, ucfg, orc.q_sample(g, rs[tk], rs[nk]), rs[tk]).clamp(-1, 1)
        loss = 0.0
        for i in range(2):
            o, d, l = ro.rays_from_xys(gu.cam_dict(cams, i), xys[i], rcfg)
            r = ro.render_rays.__wrapped__(g, pm, o, d, l, rcfg, "", u_coarse=rs["u_coarse"][i], u_fine=rs["u_fine"][i],
                                           noise_coarse=rs["noise_coarse"][i], noise_fine=rs["noise_fine"][i], noise_std=1.0)
            loss = loss + sum((r[keys[k][0]] * cot[k][i].reshape(r[keys[k][0]].shape)).sum() for k in keys)
        mnames = [k for k in pm if k.startswith("_density_net") or k.startswith("_radiance_net")]
        unames = list(pu)
        gs = torch.autograd.grad(loss, [x0] + [pu[k] for k in unames] + [pm[k] for k in mnames], allow_unused=True)
    want_x0 = gs[0]
    want_u = {k: (v if v is not None else torch.zeros_like(pu[k])) for k, v in zip(unames, gs[1:1 + len(unames)])}
    want_m = dict(zip(mnames, gs[1 + len(unames):]))
    worst = ("voxel_features", _rel(out["voxel_features"].cpu(), want_x0, 1e-9))
    for group, want in (("unet", want_u), ("render_mlp", want_m)):
        scale = sorted(float(v.abs().max()) for v in want.values())[len(want) // 2]
        assert set(out[group]) >= set(want), (group, set(want) - set(out[group]))
        for k in want:
            e = _rel(out[group][k].cpu(), want[k], 1e-2 * scale)
            if e > worst[1]:
                worst = (group + "." + k, e)
    print(f"\ntraining backward chain (bootstrap={bootstrap}): worst relative gradient error {worst[1]:.2e} ({worst[0]})")
    assert worst[1] < 1e-3, worst


@pytest.mark.skipif(EMU, reason="the UNet legs are too slow for the host emulation")
def test_training_step_with_a_torch_loss_vs_oracle_autograd(gu):
    """HoloDiffusionModel.training_step: an ordinary torch loss on the rendered outputs (rgb MSE on both passes, mask binary
    cross-entropy, an L1 depth term - the kind of objective Implicitron's ViewMetrics builds) is differentiated by torch
    on the per-ray tensors and carried through the HIP backward; loss value and every gradient against the same loss
    under autograd through the oracle pipeline."""
    import torch.nn.functional as F
    from oracle import diffusion_oracle as do
    from oracle import unet_oracle as uo
    R, C, P, Pf, n_rays = 8, 16, 16, 16, 29
    model, ucfg, usd, _, msd = gu.make_model(R, C, 16, 16, TINY_UNET, n_fine=64)
    model.n_train_target_views = 2
    model.raysampler.n_pts_per_ray_training = P
    model.renderer.n_pts_per_ray_fine_training = Pf
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 3, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    vf = torch.tanh(torch.from_numpy(np_noise(5, (1, C, R, R, R))))
    xys = (torch.from_numpy(np_noise(13, (2, n_rays, 2))).clamp(-2, 2) * 0.45).contiguous()
    rs = _streams(2, n_rays, P, Pf, 1200)
    rs.update({"xys": xys, "timesteps": torch.tensor([420]), "q_noise": torch.from_numpy(np_noise(41, tuple(vf.shape))),
               "bootstrap": False})
    tgt_rgb = torch.from_numpy(np_noise(51, (2, 3, n_rays, 1))).mul(0.3).add(0.5).clamp(0, 1)
    tgt_msk = (torch.from_numpy(np_noise(52, (2, 1, n_rays, 1))) > 0).float()
    tgt_dep = torch.from_numpy(np_noise(53, (2, 1, n_rays, 1))).abs() + 9.0

    def loss_fn(p, dev="cpu"):
        m = p["masks_render"].clamp(1e-4, 1 - 1e-4)
        return (F.mse_loss(p["images_render"], tgt_rgb.to(dev)) + F.mse_loss(p["images_render_coarse"], tgt_rgb.to(dev))
                + 0.1 * F.binary_cross_entropy(m, tgt_msk.to(dev)) + 0.01 * (p["depths_render"] - tgt_dep.to(dev)).abs().mean())

    dev_rs = {k: (v.to(gu.DEV) if torch.is_tensor(v) else v) for k, v in rs.items()}
    out = model.training_step(camera=cams.to(gu.DEV), voxel_features=vf.to(gu.DEV), rng_streams=dev_rs,
                              loss_fn=lambda p: loss_fn(p, gu.DEV))
    # ---- oracle
    orc = do.DiffusionOracle(1000)
    rcfg = ro.RenderCfg(resol=R, feature_size=C, image_height=16, image_width=16, n_pts_coarse=P, n_pts_fine=Pf)
    with torch.enable_grad():
        x0 = vf.clone().requires_grad_(True)
        pu = {k: v.detach().clone().requires_grad_(True) for k, v in usd.items()}
        pm = {k: v.detach().clone().requires_grad_(True) for k, v in msd.items()}
        g = uo.unet_forward.__wrapped__(pu, ucfg, orc.q_sample(x0, rs["timesteps"], rs["q_noise"]), rs["timesteps"]).clamp(-1, 1)
        rr = []
        for i in range(2):
            o, d, l = ro.rays_from_xys(gu.cam_dict(cams, i), xys[i], rcfg)
            rr.append(ro.render_rays.__wrapped__(g, pm, o, d, l, rcfg, "", u_coarse=rs["u_coarse"][i], u_fine=rs["u_fine"][i],
                                                 noise_coarse=rs["noise_coarse"][i], noise_fine=rs["noise_fine"][i], noise_std=1.0))
        st = lambda k, c: torch.stack([r[k].reshape(n_rays, c) for r in rr]).permute(0, 2, 1)[..., None]  # noqa: E731 (2,c,n_rays,1)
        preds = {"images_render": st("rgb", 3), "depths_render": st("depth", 1), "masks_render": st("mask", 1),
                 "images_render_coarse": st("rgb_c", 3)}
        loss = loss_fn(preds)
        mnames = [k for k in pm if k.startswith("_density_net") or k.startswith("_radiance_net")]
        gs = torch.autograd.grad(loss, [x0] + [pu[k] for k in pu] + [pm[k] for k in mnames], allow_unused=True)
    loss = loss.detach()
    assert abs(float(out["loss"]) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    want_u = {k: (v if v is not None else torch.zeros_like(pu[k])) for k, v in zip(pu, gs[1:1 + len(pu)])}
    want_m = dict(zip(mnames, gs[1 + len(pu):]))
    worst = ("voxel_features", _rel(out["voxel_features"].cpu(), gs[0], 1e-12))
    for group, want in (("unet", want_u), ("render_mlp", want_m)):
        scale = sorted(float(v.abs().max()) for v in want.values())[len(want) // 2]
        for k in want:
            e = _rel(out[group][k].cpu(), want[k], 1e-2 * scale)
            if e > worst[1]:
                worst = (group + "." + k, e)
    print(f"\ntraining step (torch loss {float(loss):.5f}): worst relative gradient error {worst[1]:.2e} ({worst[0]})")
    assert worst[1] < 1e-3, worst


@pytest.mark.skipif(EMU, reason="the UNet legs are too slow for the host emulation")
def test_training_forward_is_differentiable_like_the_reference(gu):
    """The reference's training loop calls ``preds = model(**batch)`` and ``preds["objective"].backward()``.  Here
    ``forward(evaluation_mode=TRAINING)`` under autograd returns tensors hanging on the HIP autograd nodes (denoiser and
    renderer; q_sample and the clamp in torch), so an ordinary ``loss.backward()`` fills the ``.grad`` of the plugin's
    parameters - compared with the explicit chain (training_step, itself checked against oracle autograd), bootstrap on."""
    import torch.nn.functional as F
    R, C, P, Pf, n_rays = 8, 16, 16, 16, 23
    model, _, _, _, _ = gu.make_model(R, C, 16, 16, TINY_UNET, n_fine=64)
    model.n_train_target_views = 2
    model.raysampler.n_pts_per_ray_training = P
    model.renderer.n_pts_per_ray_fine_training = Pf
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 3, -0.5, 10, (0.0, -1.0, 0.0), 3.2).to(gu.DEV)
    vf = torch.tanh(torch.from_numpy(np_noise(5, (1, C, R, R, R)))).to(gu.DEV)
    rs = _streams(2, n_rays, P, Pf, 1500)
    rs.update({"xys": (torch.from_numpy(np_noise(14, (2, n_rays, 2))).clamp(-2, 2) * 0.45).contiguous(),
               "timesteps": torch.tensor([380]), "q_noise": torch.from_numpy(np_noise(61, tuple(vf.shape))), "bootstrap": True,
               "timesteps2": torch.tensor([90]), "q_noise2": torch.from_numpy(np_noise(62, tuple(vf.shape)))})
    rs = {k: (v.to(gu.DEV) if torch.is_tensor(v) else v) for k, v in rs.items()}
    tgt = torch.from_numpy(np_noise(71, (2, 3, n_rays, 1))).mul(0.3).add(0.5).clamp(0, 1).to(gu.DEV)

    def loss_fn(p):
        return F.mse_loss(p["images_render"], tgt) + 0.2 * p["masks_render"].mean() + 0.01 * p["depths_render"].mean()

    from holo_diffusion_amd import runtime

    def both_routes():
        """worst difference (tensor, relative to its scale) between loss.backward() through forward(TRAINING) and the
        explicit chain"""
        want = model.training_step(camera=cams, voxel_features=vf, rng_streams=rs, loss_fn=loss_fn)
        model.requires_grad_(True)  # (the plugin's parameters are created frozen: inference is the default use)
        model.zero_grad(set_to_none=True)
        x = vf.clone().requires_grad_(True)
        preds = model(camera=cams, evaluation_mode=EvaluationMode.TRAINING, voxel_features=x, rng_streams=rs)
        assert preds["images_render"].requires_grad
        loss = loss_fn(preds)
        loss.backward()
        assert abs(float(loss.detach()) - float(want["loss"])) < 1e-6
        worst = ("voxel_features", _rel(x.grad.cpu(), want["voxel_features"].cpu(), 1e-12))
        named = dict(model.named_parameters())
        for group, prefix in (("unet", "net_3d._net."), ("render_mlp", "_implicit_functions.0._fn.render_mlp.")):
            scale = sorted(float(v.abs().max()) for v in want[group].values())[len(want[group]) // 2]
            for k, g in want[group].items():
                got = named[prefix + k].grad
                assert got is not None, prefix + k
                e = _rel(got.cpu(), g.cpu(), 1e-2 * scale)
                if e > worst[1]:
                    worst = (group + "." + k, e)
        return worst

    # Both routes are the HIP path; what may differ between them is the ORDER of the grid gradient's scatter-add (fp32
    # atomics by default: d(grid) differs in its last bits from run to run, and the denoiser's backward amplifies that).
    # (1) In the deterministic mode (fixed-point scatter, holo_ctx_set_deterministic) the two routes run the same kernels
    # on the same inputs in an order-independent way: they must agree to rounding of the few torch ops between them.
    with runtime.deterministic(True):
        det = both_routes()
        det2 = both_routes()
    # (2) In the default (atomics) mode the difference is a sample of the run-to-run spread of the gradients themselves:
    # it is measured over three pairs of evaluations and held to the tolerance the oracle comparisons of this file use
    # (1e-3; round 5 held ONE sample of it to 1e-4 and the driver's box drew 1.18e-4).
    with runtime.deterministic(False):
        spread = [both_routes() for _ in range(3)]
    print(f"\nloss.backward() through forward(TRAINING) vs the explicit chain: deterministic mode {det[1]:.2e} ({det[0]}), again "
          f"{det2[1]:.2e}; atomics mode " + ", ".join(f"{w[1]:.2e}" for w in spread) + f" ({spread[0][0]})")
    assert det[1] == det2[1], (det, det2)
    assert det[1] < 5e-6, det
    assert max(w[1] for w in spread) < 1e-3, spread


@pytest.mark.skipif(EMU, reason="the UNet legs are too slow for the host emulation")
def test_three_sgd_steps_follow_the_oracle_trained_the_same_way(gu, request):
    """End to end: three plain SGD steps of the TRAINING branch (fresh draws per step, injected on both sides) through
    ``forward`` + ``loss.backward()`` + an in-place parameter update - which the plugin must notice and re-upload (and
    re-transpose for the next backward) - against the oracle pipeline trained with the same rule by torch autograd: the
    loss of every step and the parameters after the third."""
    import torch.nn.functional as F
    from oracle import diffusion_oracle as do
    from oracle import unet_oracle as uo
    R, C, P, Pf, n_rays, lr = 8, 16, 12, 12, 19, 2e-2
    model, ucfg, usd, _, msd = gu.make_model(R, C, 16, 16, TINY_UNET, n_fine=64)
    model.n_train_target_views = 2
    model.raysampler.n_pts_per_ray_training = P
    model.renderer.n_pts_per_ray_fine_training = Pf
    model.requires_grad_(True)
    from holo_diffusion_amd import runtime
    prev_mode = runtime.set_deterministic(True, gu.DEV)  # (the grid gradient's scatter-add in fixed point: the trajectory of the
    # three updates is then the same on every run; with fp32 atomics it moves by ~1e-5 between runs)
    request.addfinalizer(lambda: runtime.set_deterministic(prev_mode, gu.DEV))
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 3, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
    vf = torch.tanh(torch.from_numpy(np_noise(5, (1, C, R, R, R))))
    orc = do.DiffusionOracle(1000)
    rcfg = ro.RenderCfg(resol=R, feature_size=C, image_height=16, image_width=16, n_pts_coarse=P, n_pts_fine=Pf)
    pu = {k: v.detach().clone().requires_grad_(True) for k, v in usd.items()}
    pm = {k: v.detach().clone().requires_grad_(True) for k, v in msd.items()}
    mnames = [k for k in pm if k.startswith("_density_net") or k.startswith("_radiance_net")]
    pu_prev = {k: v.detach().clone() for k, v in pu.items()}  # the denoiser ONE update behind: what a stale forward would use
    losses = []
    for step in range(3):
        rs = _streams(2, n_rays, P, Pf, 2000 + 10 * step)
        xys = (torch.from_numpy(np_noise(300 + step, (2, n_rays, 2))).clamp(-2, 2) * 0.45).contiguous()
        tt = torch.tensor([600 - 200 * step])
        qn = torch.from_numpy(np_noise(400 + step, tuple(vf.shape)))
        tgt = torch.from_numpy(np_noise(500 + step, (2, 3, n_rays, 1))).mul(0.3).add(0.5).clamp(0, 1)
        # ---- HIP: forward (autograd nodes), loss.backward(), in-place SGD update
        dev_rs = {k: v.to(gu.DEV) for k, v in rs.items()}
        dev_rs.update({"xys": xys.to(gu.DEV), "timesteps": tt.to(gu.DEV), "q_noise": qn.to(gu.DEV), "bootstrap": False})
        model.zero_grad(set_to_none=True)
        preds = model(camera=cams.to(gu.DEV), evaluation_mode=EvaluationMode.TRAINING, voxel_features=vf.to(gu.DEV), rng_streams=dev_rs)
        loss = F.mse_loss(preds["images_render"], tgt.to(gu.DEV)) + 0.1 * preds["masks_render"].mean()
        loss.backward()
        with torch.no_grad():
            for p_ in model.parameters():
                if p_.grad is not None:
                    p_.add_(p_.grad, alpha=-lr)
        # ---- oracle: the same step under torch autograd
        with torch.enable_grad():
            g = uo.unet_forward.__wrapped__(pu, ucfg, orc.q_sample(vf, tt, qn), tt).clamp(-1, 1)
            rr = []
            for i in range(2):
                o, d, l = ro.rays_from_xys(gu.cam_dict(cams, i), xys[i], rcfg)
                rr.append(ro.render_rays.__wrapped__(g, pm, o, d, l, rcfg, "", u_coarse=rs["u_coarse"][i], u_fine=rs["u_fine"][i],
                                                     noise_coarse=rs["noise_coarse"][i], noise_fine=rs["noise_fine"][i], noise_std=1.0))
            st = lambda k, c: torch.stack([r[k].reshape(n_rays, c) for r in rr]).permute(0, 2, 1)[..., None]  # noqa: E731
            lo = F.mse_loss(st("rgb", 3), tgt) + 0.1 * st("mask", 1).mean()
            params = [pu[k] for k in pu] + [pm[k] for k in mnames]
            gs = torch.autograd.grad(lo, params, allow_unused=True)
        with torch.no_grad():
            pu_prev = {k: v.detach().clone() for k, v in pu.items()}
            for p_, g_ in zip(params, gs):
                if g_ is not None:
                    p_.add_(g_, alpha=-lr)
        losses.append((float(loss.detach()), float(lo.detach())))
        assert abs(losses[-1][0] - losses[-1][1]) < 2e-5 * max(1.0, abs(losses[-1][1])), (step, losses)
        # the loss alone cannot tell a stale forward from a fresh one (one update moves it by 0.1 of the tolerance above -
        # the round-3 advisor's point); the denoiser's output can: 0.8 % per update at this learning rate
        with torch.no_grad():
            x_t = orc.q_sample(vf, tt, qn)
            y_new = uo.unet_forward.__wrapped__(pu, ucfg, x_t, tt)
            y_old = uo.unet_forward.__wrapped__(pu_prev, ucfg, x_t, tt)
            y_hip = model.net_3d(x_t.to(gu.DEV), tt.to(gu.DEV)).cpu()
        scale = float(y_new.abs().max())
        assert float((y_new - y_old).abs().max()) > 5e-3 * scale, "the update is too small for this check to notice stale weights"
        assert float((y_hip - y_new).abs().max()) < 5e-4 * scale, ("forward after the in-place update", step,
                                                                 float((y_hip - y_new).abs().max()) / scale,
                                                                 float((y_hip - y_old).abs().max()) / scale)
    named = dict(model.named_parameters())
    worst = ("", 0.0)
    for prefix, ref in (("net_3d._net.", pu), ("_implicit_functions.0._fn.render_mlp.", {k: pm[k] for k in mnames})):
        for k, v in ref.items():
            e = _rel(named[prefix + k].detach().cpu(), v.detach(), 1e-6)
            if e > worst[1]:
                worst = (prefix + k, e)
    print(f"\nthree SGD steps: losses (HIP, oracle) {[(round(a, 6), round(b, 6)) for a, b in losses]}; worst parameter difference "
          f"after the third {worst[1]:.2e} ({worst[0]})")
    assert worst[1] < 1e-4, worst
