"""K ancestral DDPM steps of the north-star net (for profiling): python scripts/denoise_loop.py [steps=12]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
w = bench.NORTH
model, usd, msd = bench.build_model(w, 64, 64, dev)
net, diff = model.net_3d, model.diffusion
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = torch.randn(1, w["feature_size"], *(w["resol"],) * 3, device=dev)
ts = torch.arange(999, 999 - K, -1, device=dev, dtype=torch.int64)[:, None].contiguous()
with torch.no_grad():
    for k in range(K):
        out = net(x, ts[k])
        x, _ = diff._step(x, ts[k], out, torch.randn_like(x), True)
torch.cuda.synchronize()
print("done", float(x.abs().mean()))
