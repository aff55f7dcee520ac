"""Development probe: where does the renderer backward differ from the oracle at the large configuration?"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import holo_diffusion_amd as hda  # noqa: E402
import tests.gpu_utils as gu  # noqa: E402
from holo_diffusion_amd.render import EvaluationMode  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from oracle.common import np_noise  # noqa: E402
from tests.test_gpu_render_backward import TINY_UNET, _streams  # noqa: E402

P, Pf, C, R, n_cam, n_rays = 64, 64, 32, 32, 4, 700
model, _, _, _, msd = gu.make_model(R, C, 16, 16, TINY_UNET, n_fine=64)
model.raysampler.n_pts_per_ray_training = P
model.renderer.n_pts_per_ray_fine_training = Pf
rcfg = ro.RenderCfg(resol=R, feature_size=C, image_height=16, image_width=16, n_pts_coarse=P, n_pts_fine=Pf)
grid = torch.tanh(torch.from_numpy(np_noise(7, (1, C, R, R, R))))
cams = hda.get_simple_360_camera_trajectory(2 * math.pi, n_cam, -0.5, 10, (0.0, -1.0, 0.0), 3.2)
xys = (torch.from_numpy(np_noise(11, (n_cam, n_rays, 2))).clamp(-2, 2) * 0.45).contiguous()
rs = _streams(n_cam, n_rays, P, Pf, 500 + P)
for fn in model._implicit_functions:
    fn.bind_args(voxel_grid_features=grid.to(gu.DEV))
bundle = model.raysampler(cams.to(gu.DEV), EvaluationMode.TRAINING, xys=xys.to(gu.DEV))
dev_rs = {k: v.to(gu.DEV) for k, v in rs.items()}
fwd = model.renderer(ray_bundle=bundle, implicit_functions=list(model._implicit_functions), evaluation_mode=EvaluationMode.TRAINING,
                     rng_streams=dev_rs)

torch.set_printoptions(precision=3, linewidth=200)
keys = {"features": ("rgb", 3), "depths": ("depth", 1), "masks": ("mask", 1), "features_coarse": ("rgb_c", 3),
        "depths_coarse": ("depth_c", 1), "masks_coarse": ("mask_c", 1)}
cot = {k: torch.from_numpy(np_noise(900 + i, (n_cam, n_rays, 1, c))) * (0.05 if "depth" in k else 1.0)
       for i, (k, (_, c)) in enumerate(keys.items())}
ggrid, pg, zm, zf = model.renderer.backward_training(bundle, list(model._implicit_functions), dev_rs,
                                                     {k: v.to(gu.DEV) for k, v in cot.items()}, return_merged=True)
rays = [ro.rays_from_xys(gu.cam_dict(cams, i), xys[i], rcfg) for i in range(n_cam)]
res = {}
for f64 in (False, True):
    cv = (lambda t: t.double()) if f64 else (lambda t: t)
    if f64:
        torch.set_default_dtype(torch.float64)
    wg = None
    for i in range(n_cam):
        o, d, l = rays[i]
        g, p, out = ro.render_rays_grad(cv(grid), {k: cv(v) for k, v in msd.items()}, cv(o), cv(d), cv(l), rcfg,
                                        {keys[k][0]: cv(cot[k][i]) for k in keys},
                                        u_coarse=cv(rs["u_coarse"][i]), u_fine=cv(rs["u_fine"][i]), noise_coarse=cv(rs["noise_coarse"][i]),
                                        noise_fine=cv(rs["noise_fine"][i]), noise_std=1.0, fine_lengths=cv(zm[i].cpu()))
        wg = g if wg is None else wg + g
    torch.set_default_dtype(torch.float32)
    res[f64] = wg.float()
for label, a_g in (("HIP vs f64", ggrid.cpu()), ("torch f32 vs f64", res[False])):
    b_g = res[True]
    e = (a_g - b_g).abs().amax(dim=1).flatten() / b_g.abs().max()
    top = torch.topk(e, 12)
    print(label, "grid: max", float(e.max()), "positions > 1e-3:", int((e > 1e-3).sum()), "> 1e-4:", int((e > 1e-4).sum()),
          "L2", float((a_g - b_g).norm() / b_g.norm()))
    print("   top positions", top.indices.tolist())
    print("   top errors   ", [round(float(v), 5) for v in top.values])
