"""Run-to-run repeatability of the bf16 denoiser (development probe): the same input through the same net N times, fresh
pageable copies each time; counts outputs that differ from the first.
usage: python scripts/bf16_repeat_stress.py [N=300] [image=16] [attn_ds=2] [min_t env or -1 for default] [sync 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
image = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ads = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mint = int(sys.argv[4]) if len(sys.argv) > 4 else -1
sync = int(sys.argv[5]) if len(sys.argv) > 5 else 0
if mint >= 0:
    os.environ["HOLO_BF16_FLASH_MIN_T"] = str(mint)
import tests.gpu_utils as gu  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from oracle.common import np_noise  # noqa: E402

cfg = uo.UNetCfg(image_size=image, in_channels=16, out_channels=16, model_channels=128, num_res_blocks=2, channel_mult=(1, 2),
                 attention_resolutions=(ads,), num_heads=2)
x = torch.from_numpy(np_noise(13, (2, 16, image, image, image)))
t = torch.tensor([77, 901], dtype=torch.int64)
net, sd = gu.make_unet(cfg, seed=7, compute_dtype="bf16")
first = None
bad = 0
for i in range(N):
    junk = torch.full((1 + (i * 7919) % 5_000_000,), float("nan"), device=gu.DEV)
    with torch.no_grad():
        xa, ta = x.to(gu.DEV), t.to(gu.DEV)
        if sync:
            torch.cuda.synchronize()
        y = net(xa, ta)
    if first is None:
        first = y.clone()
    elif not torch.equal(y, first):
        bad += 1
        if bad <= 3:
            d = (y - first).abs()
            nz = (d > 0).nonzero()
            print(f"  iter {i}: {int((d > 0).sum())} of {d.numel()} elements differ, max {float(d.max()):.3e}, nan {bool(torch.isnan(y).any())}, "
                  f"first {nz[0].tolist()} last {nz[-1].tolist()}")
    del junk
T = (image // ads) ** 3
print(f"image {image}, attention at {image // ads}^3 = {T} tokens, HOLO_BF16_FLASH_MIN_T={'default' if mint < 0 else mint}, sync {sync}: "
      f"{bad} of {N} outputs differ from the first")
