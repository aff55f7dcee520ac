"""North-star backward probe: forward + backward wall time (holo_unet_backward) of the 64^3x32 net.  Usage on the GPU box:
  python scripts/backward_probe.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import holo_diffusion_amd as hda  # noqa: E402
from holo_diffusion_amd.structure import unet_param_shapes  # noqa: E402
from holo_diffusion_amd.weights import synth_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
w = bench.NORTH
net = hda.SimpleUnet3D(image_size=w["resol"], in_channels=32, out_channels=32, model_channels=64, channel_mult=w["channel_mult"],
                       attention_resolutions=w["attention_resolutions"])
net.load_state_dict({"_net." + k: v for k, v in synth_state_dict(unet_param_shapes(64, 32, 32, 64, 2, w["channel_mult"],
                                                                                   w["attention_resolutions"]), 1234).items()})
net.to(dev)
x, g, t = torch.randn(1, 32, 64, 64, 64, device=dev), torch.randn(1, 32, 64, 64, 64, device=dev), torch.tensor([500], device=dev)
names = ["out.2.weight"]
net.backward(x, t, g, params=names)
torch.cuda.synchronize()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t0 = time.perf_counter()
for _ in range(iters):
    net.backward(x, t, g, params=names)
torch.cuda.synchronize()
print(f"forward + backward: {(time.perf_counter() - t0) / iters * 1e3:.1f} ms per call")
