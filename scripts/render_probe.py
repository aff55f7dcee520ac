"""Render-leg probe (development): ms/frame of the persistent renderer for several frame counts per call, with the
per-phase timeline, for the knobs given in the environment (HOLO_RENDER_XCD, HOLO_RENDER_WGS).  Usage on the GPU box:
  python scripts/render_probe.py [frames ...]"""
import math
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import holo_diffusion_amd as hda  # noqa: E402

warnings.simplefilter("ignore")
dev = torch.device("cuda", 0)
frames = [int(a) for a in sys.argv[1:]] or [1, 8, 30]
model, _, _ = bench.build_model(bench.NORTH, 400, 400, dev)
model.net_3d_enabled = False
vf = torch.tanh(torch.randn(1, 32, 64, 64, 64, device=dev))
for normals in (False, True):
    model._implicit_functions[0]._fn.render_normals = normals
    for F in frames:
        cams = hda.get_simple_360_camera_trajectory(2 * math.pi, max(F, 2), -30.0 * (2 * math.pi / 360), 10,
                                                    (0.0, -1.0, 0.0), 3.2).to(dev)[list(range(F))]
        with torch.no_grad():
            model.render_views(vf, cams)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                model.render_views(vf, cams)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
        print(f"normals {int(normals)} frames {F:3d}: {F * 160000 / dt / 1e6:6.2f} M rays/s  {dt / F * 1e3:6.3f} ms/frame "
              f"(xcd={os.environ.get('HOLO_RENDER_XCD', 'default')}, wgs={os.environ.get('HOLO_RENDER_WGS', 'default')})",
              flush=True)
