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
only, name, c, cam_sel = "features", "rgb", 3, 0
cot = torch.zeros(n_cam, n_rays, 1, c)
cot[cam_sel] = torch.from_numpy(np_noise(900, (n_rays, 1, c)))
ggrid, pg, zm, zf = model.renderer.backward_training(bundle, list(model._implicit_functions), dev_rs, {only: cot.to(gu.DEV)},
                                                     return_merged=True)
i = cam_sel
o, d, l = ro.rays_from_xys(gu.cam_dict(cams, i), xys[i], rcfg)
res = {}
for f64 in (False, True):
    cv = (lambda t: t.double()) if f64 else (lambda t: t)
    if f64:
        torch.set_default_dtype(torch.float64)
    g, p, out = ro.render_rays_grad(cv(grid), {k: cv(v) for k, v in msd.items()}, cv(o), cv(d), cv(l), rcfg, {name: cv(cot[i])},
                                    u_coarse=cv(rs["u_coarse"][i]), u_fine=cv(rs["u_fine"][i]), noise_coarse=cv(rs["noise_coarse"][i]),
                                    noise_fine=cv(rs["noise_fine"][i]), noise_std=1.0, fine_lengths=cv(zm[i].cpu()))
    torch.set_default_dtype(torch.float32)
    res[f64] = (g.float(), {k: v.float() for k, v in p.items()})
for label, (a_g, a_p), (b_g, b_p) in (("HIP vs f64", (ggrid.cpu(), {k: v.cpu() for k, v in pg.items()}), res[True]),
                                      ("torch f32 vs f64", res[False], res[True])):
    e = (a_g - b_g).abs().amax(dim=1).flatten() / b_g.abs().max()
    print(label, "grid: max", float(e.max()), "positions > 1e-3:", int((e > 1e-3).sum()), "> 1e-4:", int((e > 1e-4).sum()), "> 1e-5:",
          int((e > 1e-5).sum()), "of", e.numel(), "L2", float((a_g - b_g).norm() / b_g.norm()))
    k = "_density_net.mlp.3.0.weight"
    er = (a_p[k] - b_p[k]).abs().amax(dim=1) / b_p[k].abs().max()
    print(label, k, "rows: max", float(er.max()), "rows > 1e-3:", int((er > 1e-3).sum()), "> 1e-4:", int((er > 1e-4).sum()), "of", er.numel(),
          "L2", float((a_p[k] - b_p[k]).norm() / b_p[k].norm()))
    k = "_density_net.mlp.3.0.bias"
    er = (a_p[k] - b_p[k]).abs() / b_p[k].abs().max()
    print(label, k, "max", float(er.max()), "> 1e-4:", int((er > 1e-4).sum()), "top", torch.topk(er, 5).values)
