#!/usr/bin/env python3
"""Register / LDS / scratch usage of every kernel in a gfx950 assembly listing (hipcc -save-temps).
usage: python scripts/kernel_resources.py holo_diffusion_amd/csrc/<file>-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)

    def g(k):
        mm = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
        return mm.group(1) if mm else None
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dem = name
    dem = dem.replace("holo::(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in dem:
        continue
    print(f"{dem[:70]:70s} vgpr {g('next_free_vgpr'):>4} accum_off {g('accum_offset'):>4} sgpr {g('next_free_sgpr'):>4} "
          f"lds {g('group_segment_fixed_size'):>7} scratch {g('private_segment_fixed_size'):>5}")
