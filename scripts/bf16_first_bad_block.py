"""Find the first block whose output differs between repeated runs of the bf16 denoiser (development probe)."""
import os
import sys

import torch

os.environ["HOLO_KEEP_INTERMEDIATES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.gpu_utils as gu  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from oracle.common import np_noise  # noqa: E402

image = 32
cfg = uo.UNetCfg(image_size=image, in_channels=16, out_channels=16, model_channels=128, num_res_blocks=2, channel_mult=(1, 2),
                 attention_resolutions=(2,), num_heads=2)
x = torch.from_numpy(np_noise(13, (1, 16, image, image, image)))
t = torch.tensor([77], dtype=torch.int64)
net, sd = gu.make_unet(cfg, seed=7, compute_dtype="bf16")
trace = {}
uo.unet_forward(sd, cfg, x, t, trace)
tags = [k for k in trace if k.startswith(("input_blocks", "output_blocks")) or k == "middle_block"]
good = None
xd, td = x.to(gu.DEV), t.to(gu.DEV)
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    with torch.no_grad():
        y = net(xd, td)
    blocks = {k: net.fetch_block(k, tuple(trace[k].shape)).cpu() for k in tags}
    if good is None:
        good = blocks
        continue
    for k in tags:
        if not torch.equal(blocks[k], good[k]):
            d = (blocks[k] - good[k]).abs()
            nz = (d > 0).nonzero()
            ch = sorted(set(nz[:, 1].tolist()))
            print(f"run {run}: first differing block {k} {tuple(d.shape)}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}, "
                  f"channels {ch[:8]}..{ch[-3:]} ({len(ch)}), z range {int(nz[:, 2].min())}..{int(nz[:, 2].max())}, "
                  f"y {int(nz[:, 3].min())}..{int(nz[:, 3].max())}, x {int(nz[:, 4].min())}..{int(nz[:, 4].max())}")
            break
print("done")
