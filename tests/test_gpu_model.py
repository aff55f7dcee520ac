"""GPU parity: HoloDiffusionModel end to end (sample -> tanh(net_3d(vf,0)) -> render) vs the oracle pipeline."""
import math
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import holo_diffusion_amd as hda  # noqa: E402
from holo_diffusion_amd.generate import generate_samples, render_flyaround  # noqa: E402
from oracle import diffusion_oracle as do  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from oracle.common import np_noise  # noqa: E402

TINY_UNET = dict(model_channels=32, channel_mult=(1, 2), attention_resolutions=(1, 2))


@pytest.fixture(scope="module")
def gu():
    import tests.gpu_utils as g
    return g


def test_sample_then_render_matches_oracle_pipeline(gu):
    H, W = 12, 20
    model, ucfg, usd, rcfg, msd = gu.make_model(8, 32, H, W, TINY_UNET, diffusion_args=dict(num_steps=1000))
    ns = lambda t, shp, dev=None: torch.from_numpy(np_noise(77 * 100003 + t, tuple(shp)))  # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vf = model.sample_random_voxel_features(max_iter=3,
                                                noise_sampler=lambda t, s, d: ns(t, s).to(gu.DEV))
        orc = do.DiffusionOracle(1000)
        unet = lambda x, t: uo.unet_forward(usd, ucfg, x, t)  # noqa: E731
        steps = list(orc.p_sample_loop_progressive(unet, (1, 32, 8, 8, 8), ns, True, 3))
    vf_ref = steps[-1]["sample"]
    assert gu.rel_err(vf, vf_ref) < 5e-3
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 4, -30.0 * (2 * math.pi / 360), 10, (0.0, -1.0, 0.0), 3.2)
    preds = model(camera=cams[1].to(gu.DEV), voxel_features=vf_ref.to(gu.DEV))
    grid_ref = torch.tanh(uo.unet_forward(usd, ucfg, vf_ref, torch.zeros(1, dtype=torch.long)))
    ref = ro.render(grid_ref, msd, gu.cam_dict(cams, 1), rcfg)
    assert (preds["images_render"].cpu() - ref["images_render"]).abs().max() < 1e-3
    assert (preds["masks_render"].cpu() - ref["masks_render"]).abs().max() < 1e-3


def test_progressive_generator_clips_and_flyaround_shapes(gu):
    model, *_ = gu.make_model(8, 32, 8, 8, TINY_UNET, diffusion_args=dict(num_steps=1000))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.manual_seed(0)
        outs = list(model.sample_random_voxel_features_progressive(max_iter=3))
        assert len(outs) == 3 and all(o.min() >= -1 and o.max() <= 1 for o in outs)
        torch.manual_seed(0)
        fly = render_flyaround(model, n_flyaround_poses=3, device=gu.DEV, sampler_kwargs=dict(max_iter=2))
        assert fly["images_render"].shape == (3, 3, 8, 8) and fly["voxel_features"].shape == (1, 32, 8, 8, 8)
        torch.manual_seed(0)
        fly2 = render_flyaround(model, n_flyaround_poses=3, device=gu.DEV, sampler_kwargs=dict(max_iter=2),
                                batched=False)
        assert torch.equal(fly["images_render"], fly2["images_render"])
        prog = render_flyaround(model, n_flyaround_poses=2, device=gu.DEV, progressive_sampling_steps_per_render=1,
                                sampler_kwargs=dict(max_iter=2))
        assert prog["images_render"].shape == (2, 3, 8, 8)
        res = generate_samples(model, num_samples=2, n_eval_cameras=2, seed=3, device=gu.DEV,
                               sampler_kwargs=dict(max_iter=2))
        assert res["images_render"].shape == (2, 2, 3, 8, 8) and torch.isfinite(res["images_render"]).all()


def test_experiment_directory_drives_the_hip_path(gu, tmp_path):
    """expconfig.yaml + model_epoch_*.pth -> model on the GPU -> one frame equals the oracle run on the checkpoint's
    own tensors (a reference checkpoint drives the HIP path through the reference's key names)."""
    import os

    import yaml

    from holo_diffusion_amd import checkpoint as ck
    from tests.test_checkpoint_loading import _expconfig, _reference_like_state
    d = str(tmp_path)
    with open(os.path.join(d, "expconfig.yaml"), "w") as f:
        yaml.safe_dump(_expconfig(resol=8, feat=16, mc=32), f)
    cfg, _ = ck.read_expconfig(d)
    kw, _ = ck.model_args_from_expconfig(cfg)
    state = _reference_like_state(hda.HoloDiffusionModel(**kw), 11)
    torch.save(state, os.path.join(d, "model_epoch_00000042.pth"))
    H, W = 10, 14
    model, rep = ck.load_experiment(d, render_size=(W, H), device=gu.DEV)
    assert rep.checkpoint_file.endswith("model_epoch_00000042.pth")
    usd = {k[len("net_3d._net."):]: v for k, v in state.items() if k.startswith("net_3d._net.")}
    msd = {k[len("_implicit_functions.0._fn.render_mlp."):]: v for k, v in state.items()
           if k.startswith("_implicit_functions.0._fn.render_mlp.")}
    ucfg = uo.UNetCfg(image_size=8, in_channels=16, out_channels=16, model_channels=32, num_res_blocks=2,
                      channel_mult=(1, 2), attention_resolutions=(2,), num_heads=2)
    rcfg = ro.RenderCfg(resol=8, feature_size=16, image_height=H, image_width=W, n_pts_fine=16)
    vf = torch.from_numpy(np_noise(5, (1, 16, 8, 8, 8))).clamp(-1, 1)
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 4, -30.0 * (2 * math.pi / 360), 10, (0.0, -1.0, 0.0), 3.2)
    preds = model(camera=cams[2].to(gu.DEV), voxel_features=vf.to(gu.DEV))
    grid_ref = torch.tanh(uo.unet_forward(usd, ucfg, vf, torch.zeros(1, dtype=torch.long)))
    ref = ro.render(grid_ref, msd, gu.cam_dict(cams, 2), rcfg)
    assert (preds["images_render"].cpu() - ref["images_render"]).abs().max() < 1e-3
    assert (preds["masks_render"].cpu() - ref["masks_render"]).abs().max() < 1e-3


def test_generate_samples_from_experiment_end_to_end(gu, tmp_path):
    """exp_dir -> sampled grids -> fly-around frames on disk (generate_samples.py:37-138 on the HIP path): 2 samples,
    250-step schedule of the config, 3 cameras; reproducible per-sample seeds."""
    import os

    import yaml

    from holo_diffusion_amd import checkpoint as ck
    from holo_diffusion_amd.generate import generate_samples_from_experiment
    from tests.test_checkpoint_loading import _expconfig, _reference_like_state
    d = str(tmp_path)
    with open(os.path.join(d, "expconfig.yaml"), "w") as f:
        yaml.safe_dump(_expconfig(resol=8, feat=16, mc=32), f)
    cfg, _ = ck.read_expconfig(d)
    kw, _ = ck.model_args_from_expconfig(cfg)
    torch.save(_reference_like_state(hda.HoloDiffusionModel(**kw), 3), os.path.join(d, "model_epoch_00000001.pth"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = generate_samples_from_experiment(d, render_size=(12, 10), n_eval_cameras=3, num_samples=2, seed=5,
                                               device=gu.DEV)
        out2 = generate_samples_from_experiment(d, render_size=(12, 10), n_eval_cameras=3, num_samples=2, seed=5,
                                                device=gu.DEV, save_frames=False)
    assert out["images_render"].shape == (2, 3, 3, 10, 12) and torch.isfinite(out["images_render"]).all()
    assert torch.equal(out["images_render"], out2["images_render"])          # seeded per sample: reproducible
    assert not torch.equal(out["images_render"][0], out["images_render"][1])  # different samples differ
    gen = os.path.join(d, "generated_samples")
    assert os.path.isfile(os.path.join(gen, "sample_00001_frames.pt"))
    assert sorted(os.listdir(os.path.join(gen, "sample_00000_images_render"))) == [f"frame_{i:05d}.ppm" for i in range(3)]
    assert out["load_report"].checkpoint_file.endswith("model_epoch_00000001.pth")
