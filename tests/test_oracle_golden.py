"""CPU: the oracle (oracle/*.py) reproduces the golden vectors recorded from the REAL reference
(oracle/make_golden.py, development container).  This is what pins the denoiser half of the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as do
from oracle import unet_oracle as uo
from oracle.common import TINY_CFG, PLUMB_CFG, digest, np_noise, seeded_input
from holo_diffusion_amd.weights import synth_state_dict


def test_schedule_tables_bit_equal(golden_dir):
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    for T in (1000, 250, 20):
        with np.errstate(divide="ignore"):
            tab = do.schedule_tables(do.linear_betas(T))
        for k, v in tab.items():
            assert np.array_equal(v, g[f"T{T}.{k}"]), (T, k)
    # known-answer values measured on the reference
    t = do.schedule_tables(do.linear_betas(1000))
    assert t["posterior_variance"][0] == 0.0
    np.testing.assert_allclose(t["posterior_variance"][1], 5.45318766e-05, rtol=1e-8)
    np.testing.assert_allclose(t["posterior_log_variance_clipped"][0], -9.81672514, rtol=1e-8)
    np.testing.assert_allclose(t["posterior_mean_coef1"][999], 1.2835148717e-4, rtol=1e-8)
    np.testing.assert_allclose(t["posterior_mean_coef2"][999], 0.98994867827, rtol=1e-8)


def test_timestep_embedding(golden_dir):
    g = np.load(os.path.join(golden_dir, "timestep_embedding.npz"))
    ts = torch.from_numpy(g["t"])
    for dim in (32, 64):
        assert torch.equal(uo.timestep_embedding(ts, dim), torch.from_numpy(g[f"emb{dim}"]))


def test_tiny_unet_and_blocks(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_unet.npz"))
    sd = synth_state_dict(uo.unet_param_shapes(TINY_CFG), 1234)
    for t in (0, 500, 999):
        trace = {}
        y = uo.unet_forward(sd, TINY_CFG, seeded_input(TINY_CFG, 7 + t), torch.tensor([t]), trace)
        torch.testing.assert_close(y, torch.from_numpy(g[f"t{t}.y"]), rtol=1e-5, atol=1e-5)
        if t == 500:
            for k in g.files:
                if k.startswith("t500.") and k not in ("t500.y",):
                    torch.testing.assert_close(trace[k[5:]], torch.from_numpy(g[k]), rtol=1e-5, atol=1e-5)


def test_tiny_unet_batch2(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_unet_b2.npz"))
    sd = synth_state_dict(uo.unet_param_shapes(TINY_CFG), 1234)
    x2 = torch.cat([seeded_input(TINY_CFG, 100), seeded_input(TINY_CFG, 101)])
    y = uo.unet_forward(sd, TINY_CFG, x2, torch.tensor([17, 803]))
    torch.testing.assert_close(y, torch.from_numpy(g["y"]), rtol=1e-5, atol=1e-5)


def test_tiny_ops(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_ops.npz"))
    sd = synth_state_dict(uo.unet_param_shapes(TINY_CFG), 1234)
    emb = torch.from_numpy(g["emb"])
    T = lambda k: torch.from_numpy(g[k])  # noqa: E731
    torch.testing.assert_close(uo.res_block(sd, "input_blocks.4.0", T("res.x"), emb), T("res.y"), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(uo.downsample(sd, "input_blocks.3.0", T("down.x"), True), T("down.y"), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(uo.attention_block(sd, "input_blocks.4.1", T("attn.x"), 2), T("attn.y"), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(uo.upsample(sd, "output_blocks.2.2", T("up.x"), True), T("up.y"), rtol=1e-5, atol=1e-5)


def test_sampler_trajectories(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_sampler.npz"))
    sd = synth_state_dict(uo.unet_param_shapes(TINY_CFG), 1234)
    cfg = TINY_CFG
    shape = (1, cfg.in_channels, cfg.image_size, cfg.image_size, cfg.image_size)
    model = lambda x, t: uo.unet_forward(sd, cfg, x, t)  # noqa: E731
    ns = lambda t, shp: torch.from_numpy(np_noise(900 * 100003 + t, tuple(shp)))  # noqa: E731
    for tag, T, max_iter in (("T1000_iter4", 1000, 4),):
        with np.errstate(divide="ignore"):
            orc = do.DiffusionOracle(T)
        assert orc.indices(max_iter) == g[f"{tag}.indices"].tolist()
        steps = list(orc.p_sample_loop_progressive(model, shape, ns, True, max_iter))
        for i, s in enumerate(steps):
            torch.testing.assert_close(s["sample"], torch.from_numpy(g[f"{tag}.samples"][i]), rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(s["pred_xstart"], torch.from_numpy(g[f"{tag}.pred_xstart"][i]), rtol=1e-4, atol=1e-4)
    assert do.DiffusionOracle(1000).indices(4) == [999, 666, 333, 0]


@pytest.mark.slow
def test_plumbing_unet_digest(golden_dir):
    g = np.load(os.path.join(golden_dir, "full_unet_digests.npz"))
    sd = synth_state_dict(uo.unet_param_shapes(PLUMB_CFG), 1234)
    y = uo.unet_forward(sd, PLUMB_CFG, seeded_input(PLUMB_CFG, 7 + 500), torch.tensor([500]))
    d = digest(y)
    np.testing.assert_allclose(d["head"], g["plumb32x16.t500.head"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(d["mean"], g["plumb32x16.t500.mean"], rtol=1e-4, atol=1e-5)


def test_render_mlp_oracle_vs_reference_class(golden_dir):
    """oracle.render_oracle.render_mlp against outputs of the reference's OWN RenderMLP / MLPWithInputSkips class
    bodies (executed from the reference source by oracle/make_golden_render.py, PyTorch3D scaffolding stood in):
    pins the parameter names, the density-net construction quirk (LeakyReLU after the last layer only), the output
    split and the sigmoid.  What stays unpinned on this half is listed in that script's header."""
    from holo_diffusion_amd.weights import synth_state_dict
    from oracle import render_oracle as ro
    g = np.load(os.path.join(golden_dir, "ref_render_mlp.npz"))
    for C in (16, 32):
        rcfg = ro.RenderCfg(feature_size=C)
        sd = synth_state_dict(ro.render_mlp_param_shapes(rcfg), int(g[f"C{C}.seed"]))
        sd["_density_net.mlp.3.0.bias"][-1] += 0.05
        feats, dirs = torch.from_numpy(g[f"C{C}.features"]), torch.from_numpy(g[f"C{C}.dirs"])
        dens, col = ro.render_mlp(sd, feats, dirs, rcfg)
        assert (dens - torch.from_numpy(g[f"C{C}.densities"])).abs().max() <= 2e-6
        assert (col - torch.from_numpy(g[f"C{C}.colours"])).abs().max() <= 2e-6
        assert (dens < 0).any() and (dens > 0).any()  # both branches of the LeakyReLU are exercised


def test_shaded_from_normals_vs_reference(golden_dir):
    """flyaround_output.make_shaded_from_normals against the reference's _make_shaded_from_normals (flyaround.py:400-420),
    executed from the reference source: bit-equal."""
    from holo_diffusion_amd.flyaround_output import images_from_preds, make_shaded_from_normals
    g = np.load(os.path.join(golden_dir, "ref_shaded_from_normals.npz"))
    n, m, ref = (torch.from_numpy(g[k]) for k in ("normals", "mask", "shaded"))
    assert torch.equal(make_shaded_from_normals(n, m), ref)
    ims = images_from_preds({"normals_render": n, "masks_render": m}, ("_shaded_depth_render",))
    assert ims["_shaded_depth_render"].shape == (4, 3, 9, 13) and torch.equal(ims["_shaded_depth_render"][:, 1:2], ref)


# ---- round 3: more of the render half executed from the reference source (oracle/make_golden_render.py) -------------
def _implicit_sd(ro, R, C, Fd, seed):
    from holo_diffusion_amd.weights import synth_state_dict
    from oracle.common import np_noise
    rcfg = ro.RenderCfg(resol=R, feature_size=C, feature_dim=Fd)
    sd = synth_state_dict(ro.render_mlp_param_shapes(rcfg), seed)
    sd["_density_net.mlp.3.0.bias"][-1] += 0.05
    sd["_feature_net.mlp.0.0.bias"] = 0.1 * torch.from_numpy(np_noise(6, (Fd,)))
    return rcfg, sd


def test_render_mlp_defaults_vs_reference_class(golden_dir):
    """RenderMLP() with the reference's DEFAULTS (128 input features, 64 view-point independent output features - the
    configuration of holo_diffusion/tests/test_voxel_grid_implicit_function.py:17-26): densities, colours and the third
    output against the reference class."""
    from holo_diffusion_amd.weights import synth_state_dict
    from oracle import render_oracle as ro
    from oracle.common import np_noise
    g = np.load(os.path.join(golden_dir, "ref_render_mlp.npz"))
    rcfg = ro.RenderCfg(feature_size=128, feature_dim=64)
    sd = synth_state_dict(ro.render_mlp_param_shapes(rcfg), int(g["C128.seed"]))
    sd["_feature_net.mlp.0.0.bias"] = 0.1 * torch.from_numpy(np_noise(5, (64,)))
    feats, dirs = torch.from_numpy(g["C128.features"]), torch.from_numpy(g["C128.dirs"])
    dens, col = ro.render_mlp(sd, feats, dirs, rcfg)
    vp = ro.render_mlp_vp_features(sd, feats)
    assert (dens - torch.from_numpy(g["C128.densities"])).abs().max() <= 3e-6
    assert (col - torch.from_numpy(g["C128.colours"])).abs().max() <= 3e-6
    assert vp.shape[-1] == 64 and (vp - torch.from_numpy(g["C128.vp_features"])).abs().max() <= 3e-6


@pytest.mark.parametrize("tag", ["small", "defaults"])
def test_implicit_function_forward_and_normals_vs_reference_body(golden_dir, tag):
    """The in-tree body of HoloVoxelGridImplicitFunction.forward (holo_voxel_grid_implicit_function.py:182-269, both the
    pts_3d and the ray-bundle entry) and RenderMLP.get_normals (:131-145), executed from the reference source."""
    from oracle import render_oracle as ro
    from oracle.common import np_noise
    g = np.load(os.path.join(golden_dir, "ref_implicit_function.npz"))
    R, C, Fd = (int(v) for v in g[f"{tag}.cfg"])
    rcfg, sd = _implicit_sd(ro, R, C, Fd, int(g[f"{tag}.seed"]))
    grid = torch.tanh(torch.from_numpy(np_noise(int(g[f"{tag}.grid_seed"]), (1, C, R, R, R))))
    pts = torch.from_numpy(g[f"{tag}.pts"])
    dens, feats = ro.implicit_function_pts(grid, sd, pts, rcfg)
    assert feats.shape[-1] == 3 + Fd
    assert (dens - torch.from_numpy(g[f"{tag}.densities"])).abs().max() <= 3e-6
    assert (feats - torch.from_numpy(g[f"{tag}.features"])).abs().max() <= 3e-6
    if f"{tag}.normals" in g.files:
        nrm = ro.implicit_normals(grid, sd, pts, rcfg)
        assert (nrm - torch.from_numpy(g[f"{tag}.normals"])).abs().max() <= 1e-5
    o, d, l = (torch.from_numpy(g[f"{tag}.ray_{k}"]) for k in ("origins", "directions", "lengths"))
    od, oc = ro.implicit_function(grid, sd, o.reshape(-1, 3), d.reshape(-1, 3), l.reshape(-1, l.shape[-1]), rcfg)
    assert (od.reshape(g[f"{tag}.ray_densities"].shape) - torch.from_numpy(g[f"{tag}.ray_densities"])).abs().max() <= 3e-6
    assert (oc.reshape(*l.shape, 3) - torch.from_numpy(g[f"{tag}.ray_features"])[..., :3]).abs().max() <= 3e-6


@pytest.mark.parametrize("case", ["ones", "soft"])
def test_mlp_mean_aggregator_vs_reference_class(golden_dir, case):
    """oracle.viewpool_oracle.mlp_mean_aggregate against the reference's OWN MLPMeanFeatureAggregator and
    _get_point_to_source_camera_ray_dirs (custom_modules.py:162-334), executed from the reference source: parameter
    names, the doubly applied aggregation weights, the single-layer-is-the-last-layer LeakyReLU, the softmax over views."""
    from holo_diffusion_amd.weights import synth_state_dict
    from oracle import viewpool_oracle as vo
    from oracle.common import np_noise
    g = np.load(os.path.join(golden_dir, "ref_mlp_mean_aggregator.npz"))
    n_src, P, nh, do = (int(v) for v in g["dims"])
    feats = [torch.from_numpy(g[k])[0] for k in sorted(k for k in g.files if k.startswith("feats."))]
    D = sum(f.shape[-1] for f in feats) + 21
    shapes = vo.mlp_mean_param_shapes(D, nh, do)
    sd = synth_state_dict(shapes, int(g["seed"]))
    for k in shapes:
        if k.endswith("bias"):
            sd[k] = 0.1 * torch.from_numpy(np_noise(len(k), shapes[k]))
    pts, Rm, T = torch.from_numpy(g["pts"])[0], torch.from_numpy(g["R"]), torch.from_numpy(g["T"])
    dirs = torch.stack([vo.ray_dirs_to_cameras(pts, Rm[v], T[v]) for v in range(n_src)])
    assert (dirs - torch.from_numpy(g["ray_dirs"])[0]).abs().max() <= 1e-6
    out = vo.mlp_mean_aggregate(feats, dirs, torch.from_numpy(g[f"{case}.masks"])[0, ..., 0], sd, 3)
    assert out.shape == (P, do) and (out - torch.from_numpy(g[f"{case}.aggregated"])[0, 0]).abs().max() <= 3e-6


def test_images_from_preds_vs_reference_body(golden_dir):
    """flyaround_output.images_from_preds against the reference's _images_from_preds / _stack_images (flyaround.py:
    422-502) executed from the reference source (make_depth_image stood in by the restatement): every key bit-equal,
    incl. the nearest-resized mask on a depth map of another size, the source-image mosaic and the first-mask shading."""
    from holo_diffusion_amd.flyaround_output import images_from_preds
    g = np.load(os.path.join(golden_dir, "ref_images_from_preds.npz"))
    preds = {k[len("preds."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("preds.")}
    want = {k[len("out."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("out.")}
    got = images_from_preds(preds, list(want))
    assert set(got) == set(want) and len(want) == 8
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert set(images_from_preds({"images_render": preds["images_render"], "masks_render": preds["masks_render"]})) == \
        {"images_render", "masks_render"}  # default keys: the ones the predictions lack are skipped


def test_oracle_autograd_vs_reference_module_gradients(golden_dir):
    """Autograd through the oracle's forward reproduces the gradients of the REFERENCE UNetModel (ref_unet_backward.npz):
    this is what lets the `-m gpu` backward tests use the oracle as their gradient reference for other net shapes."""
    g = np.load(os.path.join(golden_dir, "ref_unet_backward.npz"))
    cfg = TINY_CFG
    sd = synth_state_dict(uo.unet_param_shapes(cfg), int(g["seed"]))
    x = torch.from_numpy(np_noise(int(g["x_seed"]), (2, cfg.in_channels) + (cfg.image_size,) * 3))
    t = torch.from_numpy(g["t"])
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        uo.unet_forward.__wrapped__(sdr, cfg, xr, t).mean().backward()
    for k in list(sd) + ["grad_x"]:
        f = (xr.grad if k == "grad_x" else sdr[k].grad).reshape(-1)
        if f.numel() > 4096:
            f = f[torch.cat([torch.arange(2048), torch.linspace(2048, f.numel() - 1, 2048).long()])]
        scale = max(float(g[f"mean.scale.{k}"]), float(g["mean.floor"]))
        assert (f - torch.from_numpy(g[f"mean.{k}"])).abs().max().item() <= 1e-5 * scale, k
