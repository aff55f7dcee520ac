"""Root-cause probe of the round-3 small-tensor finding ("a coherence rule"): the denoiser is fed FRESH pageable
host->device copies of its 16-byte timestep tensor, and time_embed_kernel reads it one of three ways
(HOLO_DEBUG_TIMESTEP_LOAD): 0 system-scope load (the product), 1 scalar load (s_load_dwordx2, what the compiler emits for
t[blockIdx.x]), 2 per-lane vector load.  A wrong timestep shows as an output far from the reference.
usage: python scripts/h2d_stress_kinds.py [iterations per kind = 600] [compute dtype = f32 | bf16]
(bf16 + HOLO_BF16_FLASH_MIN_T=0 is the configuration of the round-3 stress runs that showed 1-3 % wrong outputs.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.gpu_utils as gu  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from oracle.common import np_noise  # noqa: E402

cfg = uo.UNetCfg(image_size=16, in_channels=16, out_channels=16, model_channels=128, num_res_blocks=2, channel_mult=(1, 2),
                 attention_resolutions=(2,), num_heads=2)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
x = torch.from_numpy(np_noise(13, (2, 16, 16, 16, 16)))
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
if dtype == "bf16":
    os.environ["HOLO_BF16_FLASH_MIN_T"] = "0"
net, sd = gu.make_unet(cfg, seed=7, compute_dtype=dtype)
refs = {}
for kind in (1, 2, 0):
    os.environ["HOLO_DEBUG_TIMESTEP_LOAD"] = str(kind)
    bad = 0
    for i in range(n):
        tv = (17 + 5 * i) % 1000, (901 - 3 * i) % 1000
        t = torch.tensor(tv, dtype=torch.int64)
        if tv not in refs:
            with torch.no_grad():
                td = t.to(gu.DEV)
                torch.cuda.synchronize()
                refs[tv] = net(x.to(gu.DEV), td).clone()  # the resident, synchronised call is the reference
        junk = torch.full((1 + (i * 7919) % 5_000_000,), float(i), device=gu.DEV)  # vary what the allocator hands out
        with torch.no_grad():
            y = net(x.to(gu.DEV), t.to(gu.DEV))  # fresh pageable copies, no synchronisation before the launch
        if not torch.equal(y, refs[tv]):
            bad += 1
            if bad <= 3:
                print(f"  kind {kind} iter {i}: output differs from the synchronised call (max {float((y - refs[tv]).abs().max()):.3e})")
        del junk
    print(f"timestep read kind {kind} ({['system-scope load', 'scalar load', 'vector load'][kind]}): {bad} wrong of {n}")
