"""Platform probe (no holo kernels): does a torch kernel that consumes a fresh pageable host->device copy ever see stale data
when the destination block was last written by another kernel?"""
import sys

import torch

dev = torch.device("cuda", 0)
n = 131072  # 512 KB fp32, the size of the unet stress inputs
src = torch.randn(n)
src_dev = src.to(dev)
other = torch.randn(n, device=dev)
torch.cuda.synchronize()
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for i in range(N):
    out = other * 2.0          # a kernel writes a fresh block
    s = float(out[0])           # (forces completion sometimes)
    if i % 3:
        del out                 # block returns to the allocator; the copy below tends to get it
    a = src.to(dev)             # pageable host -> device
    b = a + 0.0                 # a kernel reads it at once
    if not torch.equal(b, src_dev):
        bad += 1
        d = (b - src_dev).abs()
        print(f"iter {i}: {int((d > 0).sum())} stale elements, first at {int((d > 0).nonzero()[0])}")
print(f"torch-only: {bad} bad of {N}")
