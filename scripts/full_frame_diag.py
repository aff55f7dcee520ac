#!/usr/bin/env python3
"""Which rays of the north-star full frame disagree with the oracle, and how (development diagnostic of
tests/test_gpu_configs.py::test_north_star_full_frame_vs_oracle)."""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import holo_diffusion_amd as hda  # noqa: E402
from holo_diffusion_amd.render import EvaluationMode  # noqa: E402
from oracle import render_oracle as ro  # noqa: E402
from oracle.common import np_noise  # noqa: E402
import tests.gpu_utils as gu  # noqa: E402

R, C, H, W = 64, 32, 400, 400
seed, ncam, icam = int(sys.argv[1]) if len(sys.argv) > 1 else 17, int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 3
model, _, _, rcfg, msd = gu.make_model(R, C, H, W, dict(model_channels=64, channel_mult=(1, 1, 2, 4, 8), attention_resolutions=(4, 8)))
model.net_3d_enabled = False
grid = torch.tanh(torch.from_numpy(np_noise(seed, (1, C, R, R, R))))
cams = hda.get_simple_360_camera_trajectory(2 * math.pi, ncam, -30.0 * (2 * math.pi / 360), 10, (0.0, -1.0, 0.0), 3.2)
preds = model(camera=cams[icam].to(gu.DEV), evaluation_mode=EvaluationMode.EVALUATION, voxel_features=grid.to(gu.DEV))
rgb = preds["images_render"].reshape(3, -1).t().cpu()
dep = preds["depths_render"].reshape(-1).cpu()
msk = preds["masks_render"].reshape(-1).cpu()
o, d, l = ro.make_rays(gu.cam_dict(cams, icam), rcfg)
e_all = []
bad_total = 0
for s in range(0, H * W, 20000):
    sl = slice(s, min(s + 20000, H * W))
    ref = ro.render_rays(grid, msd, o[sl], d[sl], l[sl], rcfg)
    e = (rgb[sl] - ref["rgb"]).abs().max(dim=1)[0]
    bad = torch.nonzero(e > 2e-4).flatten()
    bad_total += len(bad)
    for j in bad[:6].tolist():
        pix = s + j
        c_rgb = preds["rendered"].prev_stage.features.reshape(-1, 3)[pix].cpu()
        print(f"pixel {pix} (row {pix // W}, col {pix % W}): rgb err {float(e[j]):.4f} | HIP rgb {rgb[pix].tolist()} oracle {ref['rgb'][j].tolist()} | "
              f"mask HIP {float(msk[pix]):.5f} oracle {float(ref['mask'][j]):.5f} | depth HIP {float(dep[pix]):.4f} oracle {float(ref['depth'][j]):.4f} | "
              f"coarse rgb err {float((c_rgb - ref['rgb_c'][j]).abs().max()):.2e} | lengths {float(l[sl][j][0]):.3f}..{float(l[sl][j][-1]):.3f}")
    e_all.append(e)
e = torch.cat(e_all)
print(f"rays with rgb error > 2e-4: {bad_total} of {H * W}; max {float(e.max()):.4f}; rows of bad rays: "
      f"{sorted(set((torch.nonzero(e > 2e-4).flatten() // W).tolist()))[:40]}")
