"""bf16 denoiser at batch 1 / 2 against the fp32 oracle per sample (development probe of the wide-tile bf16 kernel).
usage: python scripts/bf16_batch_check.py [image=32] [repeats=6]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.gpu_utils as gu  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from oracle.common import np_noise  # noqa: E402

image = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
resident = len(sys.argv) > 3 and sys.argv[3] == "resident"
cfg = uo.UNetCfg(image_size=image, in_channels=16, out_channels=16, model_channels=128, num_res_blocks=2, channel_mult=(1, 2),
                 attention_resolutions=(2,), num_heads=2)
x = torch.from_numpy(np_noise(13, (2, 16, image, image, image)))
t = torch.tensor([77, 901], dtype=torch.int64)
net, sd = gu.make_unet(cfg, seed=7, compute_dtype="bf16")
ref = uo.unet_forward(sd, cfg, x, t)
for B in (1, 2):
    errs = []
    xr, tr = x[:B].to(gu.DEV), t[:B].to(gu.DEV)
    for r in range(reps):
        with torch.no_grad():
            y = (net(xr, tr) if resident else net(x[:B].to(gu.DEV), t[:B].to(gu.DEV))).cpu()
        errs.append([float((y[b] - ref[b]).abs().max() / ref[b].abs().max()) if not torch.isnan(y[b]).any() else float("nan") for b in range(B)])
    print(f"batch {B}: relative error per sample over {reps} runs:", [[round(e, 4) for e in er] for er in errs])
