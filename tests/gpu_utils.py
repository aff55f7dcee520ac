"""Helpers shared by the -m gpu parity tests."""
import torch

import holo_diffusion_amd as hda
from holo_diffusion_amd.weights import synth_state_dict
from oracle import render_oracle as ro
from oracle import unet_oracle as uo

import os

# HOLO_TEST_EMU=1 (tests/conftest.py): the same tests on the host emulation of the kernels, CPU tensors
EMU = os.environ.get("HOLO_TEST_EMU") == "1"
DEV = torch.device("cpu") if EMU else torch.device("cuda", 0)


def make_unet(cfg: uo.UNetCfg, seed: int = 1234, compute_dtype: str = "f32"):
    """SimpleUnet3D on the GPU + the identical CPU state dict (reference names)."""
    sd = synth_state_dict(uo.unet_param_shapes(cfg), seed)
    net = hda.SimpleUnet3D(image_size=cfg.image_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                           model_channels=cfg.model_channels, num_res_blocks=cfg.num_res_blocks,
                           channel_mult=cfg.channel_mult, attention_resolutions=cfg.attention_resolutions,
                           num_heads=cfg.num_heads, compute_dtype=compute_dtype)
    net.load_state_dict({"_net." + k: v for k, v in sd.items()})
    return net.to(DEV), sd


def make_model(resol, feature_size, H, W, unet_args, diffusion_args=None, n_fine=64, seed=1234, mlp_seed=4321,
               density_bias=0.0):
    model = hda.HoloDiffusionModel(
        resol=resol, feature_size=feature_size, render_image_width=W, render_image_height=H,
        net_3d_SimpleUnet3D_args=unet_args, diffusion_args=diffusion_args or {},
        renderer_HoloMultiPassEmissionAbsorptionRenderer_args=dict(n_pts_per_ray_fine_evaluation=n_fine))
    ucfg = uo.UNetCfg(image_size=resol, in_channels=feature_size, out_channels=feature_size,
                      model_channels=unet_args["model_channels"], num_res_blocks=unet_args.get("num_res_blocks", 2),
                      channel_mult=tuple(unet_args["channel_mult"]),
                      attention_resolutions=tuple(unet_args["attention_resolutions"]),
                      num_heads=unet_args.get("num_heads", 2))
    usd = synth_state_dict(uo.unet_param_shapes(ucfg), seed)
    rcfg = ro.RenderCfg(resol=resol, feature_size=feature_size, image_height=H, image_width=W, n_pts_fine=n_fine)
    msd = synth_state_dict(ro.render_mlp_param_shapes(rcfg), mlp_seed)
    msd["_density_net.mlp.3.0.bias"][-1] += density_bias
    full = {"net_3d._net." + k: v for k, v in usd.items()}
    for i in range(model.num_passes):
        full.update({f"_implicit_functions.{i}._fn.render_mlp." + k: v for k, v in msd.items()})
    model.load_state_dict(full)
    return model.to(DEV), ucfg, usd, rcfg, msd


def cam_dict(cams, i):
    return {"R": cams.R[i:i + 1].cpu(), "T": cams.T[i:i + 1].cpu(), "focal": cams.focal_xy()[i:i + 1].cpu(),
            "pp": cams.principal_point[i:i + 1].cpu()}


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    return ((got.detach().cpu().float() - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
