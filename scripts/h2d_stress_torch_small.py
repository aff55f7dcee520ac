"""Platform probe (no holo kernels): tiny pageable host->device copies consumed by a torch kernel at once, the small-pool
block having been read by earlier kernels while it held other values."""
import sys

import torch

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
bad = 0
big = torch.randn(1 << 20, device=dev)
for i in range(N):
    v = torch.tensor([i * 7 + 1, i * 13 + 5], dtype=torch.int64)
    old = torch.tensor([-(i + 1), -(i + 2)], dtype=torch.int64).to(dev)   # previous owner of a small block
    r0 = (old + 1).sum()                                                  # a kernel reads it (lines into L2)
    del old
    t = v.to(dev)                                                         # tends to reuse the block
    big.mul_(1.0001)                                                      # some unrelated GPU work in flight
    got = (t * 1).cpu()                                                   # a kernel consumes the fresh copy
    if not torch.equal(got, v):
        bad += 1
        if bad < 6:
            print(f"iter {i}: got {got.tolist()} want {v.tolist()}")
print(f"torch-only small tensors: {bad} bad of {N}")
