"""Render-leg probe: rays/s for different grid resolutions (is the renderer bound by the gather or by arithmetic?)."""
import math, sys, time, warnings
import torch
sys.path.insert(0, "/root/repo")
import bench
warnings.simplefilter("ignore")
dev = torch.device("cuda", 0)
import holo_diffusion_amd as hda
for resol in (8, 16, 32, 64):
    w = dict(resol=resol, feature_size=32, model_channels=64, channel_mult=(1, 1), attention_resolutions=())
    model, _, _ = bench.build_model(w, 400, 400, dev)
    cams = hda.get_simple_360_camera_trajectory(2 * math.pi, 5, -30.0 * (2 * math.pi / 360), 10, (0.0, -1.0, 0.0), 3.2).to(dev)
    vf = torch.tanh(torch.randn(1, 32, resol, resol, resol, device=dev))
    model.net_3d_enabled_backup = True
    with torch.no_grad():
        model.render_views(vf, cams[[0]])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.render_views(vf, cams[list(range(5))])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"resol {resol}: {5*160000/dt/1e6:.2f} M rays/s, {dt/5*1e3:.2f} ms/frame")
